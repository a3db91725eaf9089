"""Per-kernel register / scratch table from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
Usage: hipcc ... -Rpass-analysis=kernel-resource-usage -c unit.hip -o unit.o 2> remarks.txt ; python tools/res_usage.py remarks.txt [--all]"""
import re, sys, subprocess
t = open(sys.argv[1]).read()
show_all = '--all' in sys.argv
for b in re.split(r'remark: [^\n]*Function Name: ', t)[1:]:
    name = b.split()[0]
    try:
        name = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', name], capture_output=True, text=True).stdout.strip().split('(')[0]
    except Exception:
        pass
    g = lambda k: int(re.search(k + r': (\d+)', b).group(1))
    v, sg, sp, sc, occ = g('VGPRs'), g('SGPRs'), g('VGPRs Spill'), g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]')
    if show_all or sp or sc:
        print(f"{name[-90:]:90s} VGPR {v:3d} SGPR {sg:3d} spill {sp:3d} scratch {sc:4d} occ {occ}")
