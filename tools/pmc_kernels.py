"""Per-kernel PMC summary from the tools/pmc_bench.sh passes (all kernels of a `bench.py --ddnm-steps 2` shape):
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB (gfx950: FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads,
MI355X_MICROARCH.md), duration from the same pass's timestamps -> achieved GB/s; MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES /
(GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs).  Usage: python tools/pmc_kernels.py gpurun_out/pmc_bench profiles/r01_pmc_kernels.json"""
import csv, glob, json, sys, collections, re
root = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pmc_bench'
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(root + '/*/pmc_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name']
        short = re.sub(r'\(.*', '', name)
        m = re.search(r'(k_[a-z0-9_]+)', name)
        short = m.group(1) if m else short[:40]
        acc[short][r['Counter_Name']].append(float(r['Counter_Value']))
        if r['Counter_Name'] in ('FETCH_SIZE',):
            dur[short].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3)
out = {}
for k, d in acc.items():
    if not k.startswith('k_'):
        continue
    e = dict(launches=len(d.get('FETCH_SIZE', d.get('WRITE_SIZE', []))))
    if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d and dur[k]:
        fb = sum(d['FETCH_SIZE']) / len(d['FETCH_SIZE']) * 2 * 1024
        wb = sum(d['WRITE_SIZE']) / len(d['WRITE_SIZE']) * 1024
        us = sum(dur[k]) / len(dur[k])
        e.update(hbm_read_MB=round(fb / 1e6, 3), hbm_write_MB=round(wb / 1e6, 3), avg_us=round(us, 2), hbm_GBps=round((fb + wb) / us / 1e3, 1))
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in d and 'GRBM_GUI_ACTIVE' in d and sum(d['GRBM_GUI_ACTIVE']) > 0:
        e['mfma_util'] = round(sum(d['SQ_VALU_MFMA_BUSY_CYCLES']) / (sum(d['GRBM_GUI_ACTIVE']) / 8 * 256 * 4), 4)
    out[k] = e
res = dict(note="per-launch averages over one `bench.py --steps 1 --warmup 0 --ddnm-steps 2 --no-cpu-baseline` run under rocprofv3 --pmc "
                "(separate passes for FETCH_SIZE, WRITE_SIZE, SQ/GRBM); profiled passes run ~3 % slower clocks than un-profiled ones",
           kernels=dict(sorted(out.items(), key=lambda kv: -kv[1].get('hbm_read_MB', 0) * kv[1]['launches'])))
print(json.dumps(res, indent=1))
if len(sys.argv) > 2:
    json.dump(res, open(sys.argv[2], 'w'), indent=1)
