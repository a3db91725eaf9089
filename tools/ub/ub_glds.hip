// Micro-benchmark: per-CU L2 -> LDS / L2 -> VGPR fill rate on gfx950, by load kind and waves per CU.
// Every workgroup (one per CU, `waves` waves) streams `bytes_per_wg` bytes from an L2-resident window of `window` bytes, `reps` times.
//   mode 0: global_load_lds dwordx4 (LDS-DMA, 1 KiB per wave-instruction)     mode 1: global_load_dwordx4 -> VGPR (accumulated)
//   mode 2: global_load_dwordx4 -> VGPR -> ds_write_b128
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/ub/ub_glds.hip -o tools/ub/ub_glds.so ; run: python tools/ub/ub_glds.py
#include <hip/hip_runtime.h>
#include <cstdint>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
typedef __attribute__((ext_vector_type(4))) float f4;

template <int MODE>
__global__ __launch_bounds__(1024) void k_fill(const char* __restrict__ src, size_t window, int bytes_per_wg, int reps, float* __restrict__ sink, int same) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const size_t base = same ? 0 : ((size_t)blockIdx.x * bytes_per_wg) % window;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const int per_wave = bytes_per_wg / nw;              // contiguous slice per wave
    for (int r = 0; r < reps; ++r) {
        const char* p = src + (base + (size_t)wave * per_wave) % window + lane * 16;
        char* l = smem + wave * 8192;
        for (int off = 0; off < per_wave; off += 8192) {     // 8 pieces of 1 KiB in flight per wave
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (MODE == 0) __builtin_amdgcn_global_load_lds((gbl_void*)(p + off + u * 1024), (lds_void*)(l + u * 1024), 16, 0, 0);
                else {
                    const f4 v = *reinterpret_cast<const f4*>(p + off + u * 1024);
                    if (MODE == 1) acc += v;
                    else *reinterpret_cast<f4*>(l + u * 1024 + lane * 16) = v;
                }
            }
            if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    if (MODE != 1) { __syncthreads(); acc = *reinterpret_cast<f4*>(smem + threadIdx.x * 16); }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}

extern "C" int ub_fill(int mode, const void* src, size_t window, int wgs, int waves, int bytes_per_wg, int reps, float* sink, int same, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const size_t smem = (size_t)waves * 8192;
    auto k = mode == 0 ? k_fill<0> : mode == 1 ? k_fill<1> : k_fill<2>;
    if (smem > 65536) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    k<<<wgs, waves * 64, smem, s>>>((const char*)src, window, bytes_per_wg, reps, sink, same);
    return (int)hipGetLastError();
}
