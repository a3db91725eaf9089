// Micro-benchmark: what the LDS of one gfx950 CU delivers to the operand pattern of k_conv_sk's 128x128 tile.
//   A: ds_read_b128 fragment reads with the kernel's own layout (128-byte rows, 16-byte slot ^= (row >> 1) & 7; 16 reads per batch,
//      counted or full lgkmcnt waits) at 4 and 8 waves per CU, next to the un-swizzled layout (8-way conflicts) as a control;
//   B: LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction) landing rate from an L2-resident 32 KB window per CU, 4 / 8 waves;
//   C: A and B at once -- 4 reader waves (one per SIMD) beside 4 loader waves, the kernel's loader-specialised shape without its MFMAs;
//   D: C with 32 v_mfma_f32_16x16x32_f16 per batch behind counted waits (the K-step of the kernel, no barriers);
//   E: latency of one batch: cycles from the first ds_read_b128 of 16 to lgkmcnt(11) (first MFMA group can start) and to lgkmcnt(0),
//      4 waves starting together behind a barrier, with and without loaders running.
// Cycle counts are s_memtime deltas of wave 0 of workgroup 8 (shader clock), rates are per CU.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ub/ub_lds.hip -o tools/ub/ub_lds.bin ; run on the GPU box: tools/ub/ub_lds.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(8))) _Float16 h8;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

__device__ __forceinline__ unsigned long long clk() { return __builtin_readcyclecounter(); }

// one batch = 16 ds_read_b128 of a 64 x 64 wave tile's K-step: 2 k-halves x (4 B rows blocks + 4 A row blocks), as k_conv_sk issues them
#define DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define BATCH16(F, A0, B0, A1, B1)                                                                     \
    DSR(F[0], B0, 0); DSR(F[1], B0, 2048); DSR(F[2], B0, 4096); DSR(F[3], B0, 6144);                    \
    DSR(F[4], A0, 0); DSR(F[5], A0, 2048); DSR(F[6], A0, 4096); DSR(F[7], A0, 6144);                    \
    DSR(F[8], B1, 0); DSR(F[9], B1, 2048); DSR(F[10], B1, 4096); DSR(F[11], B1, 6144);                  \
    DSR(F[12], A1, 0); DSR(F[13], A1, 2048); DSR(F[14], A1, 4096); DSR(F[15], A1, 6144);

// MODE bit 0: readers run; bit 1: loaders run; bit 2: readers also issue 32 MFMAs per batch; bit 3: a sched_barrier behind every MFMA
// group (hipcc otherwise sinks group (0,3) below the NEXT group's wait: "lgkmcnt(8); lgkmcnt(3); 8 MFMAs" in the .s); SWZ: the kernel's slot swizzle on / off
// readers = waves [0, nread), loaders = waves [nread, nread + nload)
template <int MODE, int SWZ>
__global__ __launch_bounds__((MODE & 4) ? 768 : 1024) void k_lds(const char* __restrict__ src, int reps, int nread, int nload, unsigned long long* __restrict__ out, float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];          // 3 stages x 32 KB like the kernel
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 96 * 1024 / 4; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    unsigned long long t0 = 0, t1 = 0;
    f4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
    if (wave < nread) {
        if (MODE & 1) {
            const int w4 = wave & 3, wm = w4 >> 1, wn = w4 & 1;
            const int sw = SWZ ? ((lane & 15) >> 1) & 7 : 0;
            uint32_t fo[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) fo[kk] = (lane & 15) * 128 + ((((lane >> 4) + 4 * kk) ^ sw) << 4);
            const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
            t0 = clk();
            int st = 0;
            for (int r = 0; r < reps; ++r) {
                const uint32_t a_ad = base + st * 32768 + wm * 64 * 128, b_ad = base + st * 32768 + 16384 + wn * 64 * 128;
                h8 F[16];
                const uint32_t a0 = a_ad + fo[0], b0 = b_ad + fo[0], a1 = a_ad + fo[1], b1 = b_ad + fo[1];
                BATCH16(F, a0, b0, a1, b1)
                if (MODE & 4) {
#define GRP(W, AI, B0_)                                                                                                           \
    asm volatile("s_waitcnt lgkmcnt(" #W ")" : "+v"(F[AI]), "+v"(F[B0_]), "+v"(F[B0_ + 1]), "+v"(F[B0_ + 2]), "+v"(F[B0_ + 3]));  \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[(AI) & 3][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(F[B0_ + j], F[AI], acc[(AI) & 3][j], 0, 0, 0); \
    if (MODE & 8) __builtin_amdgcn_sched_barrier(0);
                    GRP(11, 4, 0) GRP(10, 5, 0) GRP(9, 6, 0) GRP(8, 7, 0) GRP(3, 12, 8) GRP(2, 13, 8) GRP(1, 14, 8) GRP(0, 15, 8)
#undef GRP
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int i = 0; i < 16; ++i) asm volatile("" ::"v"(F[i]));
                }
                st = st == 2 ? 0 : st + 1;
            }
            t1 = clk();
        }
    } else if (wave < nread + nload) {
        if (MODE & 2) {
            // each loader wave: 8 pieces of 1 KiB per "K-step" into its quarter of the stage, from this CU's own 32 KB window (L2 hits)
            const int w = wave - nread;
            const char* p = src + ((size_t)(blockIdx.x & 255) * 32768) + (size_t)(w & 3) * 8192 + lane * 16;
            t0 = clk();
            int st = 0;
            for (int r = 0; r < reps; ++r) {
                char* l = smem + st * 32768 + (w & 3) * 8192;
#pragma unroll
                for (int u = 0; u < 8; ++u) __builtin_amdgcn_global_load_lds((gbl_void*)(p + u * 1024), (lds_void*)(l + u * 1024), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // one K-step of pieces stays in flight
                st = st == 2 ? 0 : st + 1;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            t1 = clk();
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 123.456f) sink[0] = s;
    if (lane == 0 && blockIdx.x == 8) out[wave] = t1 - t0;
}

// E: latency of ONE batch behind a barrier.  Stamps by s_memtime without waiting for the result (a waited s_memtime is an lgkmcnt(0)):
// SMEM returns out of order with respect to LDS, so an outstanding stamp can only make a counted wait conservative, never early.
template <int LOADERS>
__global__ __launch_bounds__(512) void k_lat(const char* __restrict__ src, int reps, unsigned long long* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 96 * 1024 / 4; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = 0.f;
    __syncthreads();
    unsigned long long s_issue = 0, s_first = 0, s_all = 0;
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int w4 = wave & 3, wm = w4 >> 1, wn = w4 & 1, sw = ((lane & 15) >> 1) & 7;
    const uint32_t fo0 = (lane & 15) * 128 + ((((lane >> 4) + 0) ^ sw) << 4), fo1 = (lane & 15) * 128 + ((((lane >> 4) + 4) ^ sw) << 4);
    const char* p = src + ((size_t)(blockIdx.x & 255) * 32768) + (size_t)w4 * 8192 + lane * 16;
    int st = 0;
    for (int r = 0; r < reps; ++r) {
        __builtin_amdgcn_s_barrier();
        if (wave < 4) {
            const uint32_t a_ad = base + st * 32768 + wm * 64 * 128, b_ad = base + st * 32768 + 16384 + wn * 64 * 128;
            const uint32_t a0 = a_ad + fo0, b0 = b_ad + fo0, a1 = a_ad + fo1, b1 = b_ad + fo1;
            h8 F[16];
            unsigned long long ta, tb, tc, td;
            asm volatile("s_memtime %0" : "=s"(ta));
            BATCH16(F, a0, b0, a1, b1)
            asm volatile("s_memtime %0" : "=s"(tb));
            asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");      // (11 reads + the stamp in flight at most: reads 0..4 are back)
            asm volatile("s_memtime %0" : "=s"(tc));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("s_memtime %0" : "=s"(td));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ta), "+s"(tb), "+s"(tc), "+s"(td));    // (tied: no stamp is used above this line)
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("" ::"v"(F[i]));
            s_issue += tb - ta; s_first += tc - ta; s_all += td - ta;
        } else if (LOADERS) {
            char* l = smem + ((st + 1) % 3) * 32768 + w4 * 8192;
#pragma unroll
            for (int u = 0; u < 8; ++u) __builtin_amdgcn_global_load_lds((gbl_void*)(p + u * 1024), (lds_void*)(l + u * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        st = st == 2 ? 0 : st + 1;
    }
    if (lane == 0 && blockIdx.x == 8 && wave < 4) { out[wave * 3] = s_issue; out[wave * 3 + 1] = s_first; out[wave * 3 + 2] = s_all; }
}

template <int MODE, int SWZ>
static int run(const char* tag, const char* src, int nread, int nload, int reps, unsigned long long* d_out, float* sink) {
    auto k = k_lds<MODE, SWZ>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    for (int it = 0; it < 2; ++it) { k<<<256, (nread + nload) * 64, 96 * 1024>>>(src, reps, nread, nload, d_out, sink); }
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(16);
    CK(hipMemcpy(h.data(), d_out, 16 * 8, hipMemcpyDeviceToHost));
    double rc = 0, lc = 0;
    for (int w = 0; w < nread; ++w) rc += (double)h[w] / nread;
    for (int w = nread; w < nread + nload; ++w) lc += (double)h[w] / (nload ? nload : 1);
    printf("%-58s", tag);
    if (MODE & 1) {
        const double per_batch = rc / reps;                 // cycles per 16-read batch of ONE wave; nread waves run concurrently
        printf(" readers %d: %7.1f cyc / batch of 16 = %5.2f LDS cyc per wave-instruction, %6.1f B/clk/CU", nread, per_batch, per_batch / (16.0 * nread),
               16.0 * 1024 * nread / per_batch);
    }
    if (MODE & 2) {
        const double per_step = lc / reps;
        printf(" | loaders %d: %7.1f cyc / 8 pieces, %6.1f B/clk/CU landed", nload, per_step, 8.0 * 1024 * nload / per_step);
    }
    printf("\n");
    return 0;
}

int main() {
    char* src; unsigned long long* d_out; float* sink;
    CK(hipMalloc(&src, 256 * 32768 + 65536)); CK(hipMemset(src, 0, 256 * 32768 + 65536));
    CK(hipMalloc(&d_out, 64 * 8)); CK(hipMalloc(&sink, 64));
    const int R = 2000;
    printf("== A: ds_read_b128, k_conv_sk fragment layout (16 reads per batch, lgkmcnt(0) per batch)\n");
    run<1, 1>("swizzled, 4 waves (1 per SIMD)", src, 4, 0, R, d_out, sink);
    run<1, 1>("swizzled, 8 waves (2 per SIMD)", src, 8, 0, R, d_out, sink);
    run<1, 1>("swizzled, 16 waves", src, 16, 0, R, d_out, sink);
    run<1, 0>("NOT swizzled (control: 8-way), 4 waves", src, 4, 0, R, d_out, sink);
    printf("== B: LDS-DMA landing, 8 pieces of 1 KiB per wave and step, one step kept in flight, L2-resident source\n");
    run<2, 1>("loaders only, 4 waves", src, 0, 4, R, d_out, sink);
    run<2, 1>("loaders only, 8 waves", src, 0, 8, R, d_out, sink);
    printf("== C: readers + loaders together (4 + 4: the loader-specialised workgroup without MFMAs)\n");
    run<3, 1>("4 readers + 4 loaders", src, 4, 4, R, d_out, sink);
    run<3, 1>("8 readers + 4 loaders", src, 8, 4, R, d_out, sink);
    printf("== D: readers issue the K-step's 32 MFMAs behind counted waits (no barriers): the compute phase alone\n");
    run<5, 1>("4 readers + MFMA, no loaders", src, 4, 0, R, d_out, sink);
    run<7, 1>("4 readers + MFMA + 4 loaders", src, 4, 4, R, d_out, sink);
    run<13, 1>("4 readers + MFMA (groups pinned), no loaders", src, 4, 0, R, d_out, sink);
    run<15, 1>("4 readers + MFMA (groups pinned) + 4 loaders", src, 4, 4, R, d_out, sink);
    run<5, 1>("8 readers + MFMA, no loaders", src, 8, 0, R, d_out, sink);
    run<7, 1>("8 readers + MFMA + 4 loaders", src, 8, 4, R, d_out, sink);
    printf("== E: one batch behind a barrier: cycles from the first ds_read_b128 to [16 issued | reads 0-4 back | all back], wave 0..3\n");
    for (int ld = 0; ld < 2; ++ld) {
        if (ld == 0) { CK(hipFuncSetAttribute((const void*)k_lat<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)); k_lat<0><<<256, 512, 96 * 1024>>>(src, R, d_out); }
        else { CK(hipFuncSetAttribute((const void*)k_lat<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)); k_lat<1><<<256, 512, 96 * 1024>>>(src, R, d_out); }
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(12);
        CK(hipMemcpy(h.data(), d_out, 12 * 8, hipMemcpyDeviceToHost));
        printf("%s:", ld ? "4 loaders issuing 8 pieces each per step" : "loader waves idle                        ");
        for (int w = 0; w < 4; ++w) printf("  [%5.0f | %5.0f | %5.0f]", (double)h[w * 3] / R, (double)h[w * 3 + 1] / R, (double)h[w * 3 + 2] / R);
        printf("\n");
    }
    return 0;
}
