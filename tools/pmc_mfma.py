"""Matrix-pipe occupancy and wave-state split of a kernel: ONE rocprofv3 --kernel-trace --pmc pass (GRBM_GUI_ACTIVE + SQ counters), durations from
the same pass's kernel trace.  (GRBM_GUI_ACTIVE / duration is NOT a clock reading for launches of tens of microseconds: the counter window is
wider than the dispatch; the clock under load is sampled from sysfs by bench.py.)  Usage (GPU box):
  python tools/pmc_mfma.py --filter k_conv_ht -- python /abs/path/tools/bench_ht.py --shapes 2"""
import argparse, collections, csv, glob, os, re, subprocess, sys
ap = argparse.ArgumentParser()
ap.add_argument('--filter', default='.')
ap.add_argument('--counters', nargs='*', default="GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA".split())
argv = sys.argv[1:]
cut = argv.index('--')
a = ap.parse_args(argv[:cut]); cmd = argv[cut + 1:]
root = os.environ.get('GRAFT_REPO_ROOT', os.getcwd())
d = '/tmp/pmc_mfma_%d' % os.getpid()
r = subprocess.run(['rocprofv3', '--kernel-trace', '--pmc'] + a.counters + ['-d', d, '-o', 'p', '--output-format', 'csv', '--'] + cmd, cwd='/tmp',
                   env=dict(os.environ, TMPDIR='/tmp', PYTHONPATH=root), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
if r.returncode != 0:
    print(r.stdout[-2000:]); sys.exit(1)
dur = {}
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        dur[row['Dispatch_Id']] = (row['Kernel_Name'], int(row['End_Timestamp']) - int(row['Start_Timestamp']))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if re.search(a.filter, row['Kernel_Name']):
            key = (re.sub(r'\(.*', '', row['Kernel_Name'])[-60:], row['Grid_Size'])
            acc[key][row['Counter_Name']].append(float(row['Counter_Value']))
            if row['Dispatch_Id'] in dur:
                acc[key]['_dur_ns_' + row['Dispatch_Id']] = [dur[row['Dispatch_Id']][1]]
for key, c in acc.items():
    durs = [v[0] for k, v in c.items() if k.startswith('_dur_ns_')]
    cnt = {k: sum(v) / len(v) for k, v in c.items() if not k.startswith('_dur')}
    t = sum(durs) / max(len(durs), 1)
    line = f"{key[0]} grid {key[1]}: {len(durs)} launches, avg {t / 1e3:.2f} us"
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in cnt and 'GRBM_GUI_ACTIVE' in cnt:
        line += f", MFMA busy / (GUI_ACTIVE/8 x 1024 SIMDs) {cnt['SQ_VALU_MFMA_BUSY_CYCLES'] / (cnt['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f}"
    if 'SQ_INSTS_MFMA' in cnt and 'SQ_VALU_MFMA_BUSY_CYCLES' in cnt:
        line += f", busy cycles per MFMA {cnt['SQ_VALU_MFMA_BUSY_CYCLES'] / max(cnt['SQ_INSTS_MFMA'], 1):.1f}"
    if 'SQ_WAVE_CYCLES' in cnt:
        line += "".join(f", {n[3:]} {cnt[n] / cnt['SQ_WAVE_CYCLES']:.3f}" for n in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY') if n in cnt)
    print(line)
