# compile csrc/hpr.hip alone and print the resource usage of its kernels; the ISA goes to /tmp/hx/hpr.s
mkdir -p /tmp/hx && cd /root/repo/pointdreamer_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 "$@" -c hpr.hip -o /tmp/hx/hpr.o -save-temps=obj 2>&1 | grep -v "^$" | head -40
mv /tmp/hx/hpr-hip-amdgcn-amd-amdhsa-gfx950.s /tmp/hx/hpr.s; rm -f /tmp/hx/hpr-h*
grep -E "^\s+\.(vgpr_count|sgpr_count|private_segment_fixed_size|name):" /tmp/hx/hpr.s | paste - - - - | awk '{print substr($2,1,40),$4,$6,$8}'
