"""rocprofv3 PMC passes over a command, aggregated per kernel (separate passes per counter set; --pmc only ever with --kernel-trace,
never with a trace domain).  Usage (GPU box):
  python tools/pmc_run.py OUT.json --filter k_conv3x3_halo [--sets sq lds misc cache hbm] -- python tools/bench_conv.py ...
Output: {kernel name (shortened): {counter: per-launch average, launches, derived fractions}}; HBM bytes per launch =
(2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE counts 128-B requests as 64 B)."""
import argparse, collections, csv, glob, json, os, re, shutil, subprocess, sys
SETS = {
    'sq': "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU",
    'lds': "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU GRBM_GUI_ACTIVE",
    'misc': "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_WAVES",
    'cache': "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum",
    'fetch': "FETCH_SIZE GRBM_GUI_ACTIVE",
    'write': "WRITE_SIZE",
}
ap = argparse.ArgumentParser()
ap.add_argument('out')
ap.add_argument('--filter', default='.', help='regex on the kernel name')
ap.add_argument('--sets', nargs='*', default=['sq', 'lds', 'misc', 'cache', 'fetch', 'write'])
ap.add_argument('--keep', action='store_true')
argv = sys.argv[1:]
cut = argv.index('--') if '--' in argv else len(argv)
a = ap.parse_args(argv[:cut])
cmd = argv[cut + 1:]
assert cmd, "usage: pmc_run.py OUT.json [options] -- command ..."
root = os.environ.get('GRAFT_REPO_ROOT', os.getcwd())
work = os.path.join('/tmp', 'pmc_run_%d' % os.getpid())
os.makedirs(work, exist_ok=True)
env = dict(os.environ, TMPDIR='/tmp')
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for sname in a.sets:
    d = os.path.join(work, sname)
    r = subprocess.run(['rocprofv3', '--kernel-trace', '--pmc'] + SETS[sname].split() + ['-d', d, '-o', 'pmc', '--output-format', 'csv', '--'] + cmd,
                       cwd='/tmp', env=dict(env, PYTHONPATH=root), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        print(f"pass {sname} failed:\n" + r.stdout[-1500:], file=sys.stderr)
        continue
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            if re.search(a.filter, row['Kernel_Name']):
                acc[row['Kernel_Name']][row['Counter_Name']].append(float(row['Counter_Value']))
res = {}
for k, dct in acc.items():
    short = re.sub(r'\(.*', '', k)
    short = re.sub(r'^.*?(k_[a-z0-9_]+)', r'\1', short)[:100]
    c = {n: sum(v) / len(v) for n, v in dct.items()}
    c['launches'] = max(len(v) for v in dct.values())
    if 'SQ_WAVE_CYCLES' in c:
        for n in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_VMEM'):
            if n in c:
                c['frac_of_wave_cycles.' + n] = round(c[n] / c['SQ_WAVE_CYCLES'], 4)
    if 'SQ_LDS_IDX_ACTIVE' in c and 'SQ_LDS_BANK_CONFLICT' in c:
        c['lds_conflict_frac_of_active'] = round(c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1), 4)
    if 'SQ_LDS_IDX_ACTIVE' in c and 'SQ_BUSY_CU_CYCLES' in c:
        c['lds_active_frac_of_cu_busy'] = round(c['SQ_LDS_IDX_ACTIVE'] / max(c['SQ_BUSY_CU_CYCLES'], 1), 4)
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'GRBM_GUI_ACTIVE' in c:
        c['mfma_util'] = round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['GRBM_GUI_ACTIVE'] / 8 * 256 * 4), 4)       # gfx94x MfmaUtil formula; GUI_ACTIVE is summed over the 8 XCDs
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'SQ_BUSY_CU_CYCLES' in c:
        c['mfma_busy_frac_of_workgroup_lifetime'] = round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / 4 / max(c['SQ_BUSY_CU_CYCLES'], 1), 4)
    if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
        c['hbm_bytes_per_launch'] = (2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024
    if 'TCC_HIT_sum' in c and 'TCC_MISS_sum' in c:
        c['l2_hit_rate'] = round(c['TCC_HIT_sum'] / max(c['TCC_HIT_sum'] + c['TCC_MISS_sum'], 1), 4)
    res[short] = c
out = dict(command=' '.join(cmd), filter=a.filter, kernels=res,
           note="per-launch averages; SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves, SQ_VALU_MFMA_BUSY_CYCLES cycles; FETCH_SIZE doubled in hbm_bytes_per_launch")
json.dump(out, open(a.out, 'w'), indent=1)
if not a.keep:
    shutil.rmtree(work, ignore_errors=True)
print(json.dumps(out, indent=1)[:5000])
