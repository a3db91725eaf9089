// Micro-benchmark: LDS-DMA fill rate when a 1 KiB wave-instruction gathers 8 ROWS of 128 bytes at a given row stride (what the conv
// kernels' tile loaders do: activation rows are Cin*2 bytes apart, weight rows K*2 bytes apart) vs 1 KiB contiguous.
// Every workgroup (4 or 8 waves, 1 per CU) streams `rows_per_wg` rows of 128 B, `reps` times, 8 pieces in flight per wave.
#include <hip/hip_runtime.h>
#include <cstdint>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
__global__ __launch_bounds__(1024) void k_rows(const char* __restrict__ src, size_t window, long long row_stride, int rows_per_wg, int reps,
                                               float* __restrict__ sink, int shared_src, int kstep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int lrow = lane >> 3, lpos = lane & 7;
    const size_t base = shared_src ? 0 : ((size_t)blockIdx.x * rows_per_wg * row_stride) % window;   // (once per workgroup)
    char* l = smem + wave * 8192;
    for (int r = 0; r < reps; ++r) {
        // K-steps: the same rows, 128 B further along each row (like a conv's K loop), kstep of them per repetition
        for (int ks = 0; ks < kstep; ++ks) {
            for (int p0 = wave * 8; p0 < rows_per_wg; p0 += nw * 64) {       // 8 pieces (64 rows) per wave per batch
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int row = p0 + u * 8 + lrow;                          // (host: rows_per_wg % 64 == 0, everything inside the window)
                    const char* g = src + base + (size_t)row * row_stride + (size_t)ks * 128 + lpos * 16;
                    __builtin_amdgcn_global_load_lds((gbl_void*)g, (lds_void*)(l + u * 1024), 16, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
    }
    __syncthreads();
    const float v = *reinterpret_cast<float*>(smem + threadIdx.x * 4);
    if (v == 123.456f) sink[0] = v;
}
extern "C" int ub_rows(const void* src, size_t window, long long row_stride, int wgs, int waves, int rows_per_wg, int reps, float* sink,
                       int shared_src, int kstep, void* stream) {
    k_rows<<<wgs, waves * 64, waves * 8192, (hipStream_t)stream>>>((const char*)src, window, row_stride, rows_per_wg, reps, sink, shared_src, kstep);
    return (int)hipGetLastError();
}
