// ds_read_b64_tr_b16 semantics probe: LDS holds u16 element e at element index e; lane l supplies byte address perm[l] * 8.
// Output: the four u16 each lane receives.  Usage: tools/ub/ub_tr.py
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ void k_tr(const int* __restrict__ slot_of_lane, unsigned short* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned short*)lds + slot_of_lane[threadIdx.x] * 8;
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
    out[threadIdx.x * 4 + 0] = v.x & 0xffff; out[threadIdx.x * 4 + 1] = v.x >> 16;
    out[threadIdx.x * 4 + 2] = v.y & 0xffff; out[threadIdx.x * 4 + 3] = v.y >> 16;
}
extern "C" int ub_tr(const int* slots, unsigned short* out, hipStream_t s) {
    k_tr<<<1, 64, 0, s>>>(slots, out);
    return (int)hipGetLastError();
}
