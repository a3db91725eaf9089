// Row U1: implicit-GEMM convolution on the gfx950 matrix cores -- the dominant kernel of the DDNM path
// (94.5 % of the UNet's 1 119.8 GMAC are 3x3 convolutions).  Replaces the cuDNN fp16
// conv calls behind nn.Conv2d / nn.Conv1d(k=1) in models/DDNM/guided_diffusion/unet.py:143-305.
//
//   Y[m, n] = sum_k A[m, k] * Wt[n, k] + bias[n] (+ residual[m, n])
//   m = (image, y, x) pixel, n = output channel, k = (tap, input channel), A gathered on the fly from the
//   NHWC f16 activation with zero padding -- no im2col buffer.
//
// Structure (MI355X-first): 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 each =
// 4x4 v_mfma_f32_16x16x32_f16 accumulators), K-step BKT = 64 (32 when Cin % 64 != 0), both operand tiles staged
// HBM->LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, 1 KiB per wave-instruction), double-buffered;
// the LDS image is lane-linear so the bank swizzle is applied on the per-lane SOURCE address and mirrored on the
// ds_read_b128 side (cdna guide rule 21).  Out-of-image taps read a 16-byte zero page instead of branching.
// Workgroup ids are remapped so that the n-tiles sharing one activation tile run back-to-back on one XCD (L2 reuse).
// Epilogue: accumulators -> LDS transpose -> 16-byte coalesced NHWC stores with bias and residual fused.
// Small-M layers: split-K over blockIdx.y with f32 partial tiles + a fixed-order reduce kernel.
#include "nn_common.h"
using namespace pdhip;
namespace pdnn {


#ifdef PD_LAB_STAMP                                        // (lab builds only: per-block phase timestamps, s_memtime)
__device__ unsigned long long g_lab_stamps[8 * 4096];
#define PD_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 4096 && blockIdx.y == 0) g_lab_stamps[blockIdx.x * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define PD_STAMP(k) do {} while (0)
#endif
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
#ifndef PD_LAB_NODMA                                       // (lab builds only: mainloop without the HBM->LDS traffic)
    __builtin_amdgcn_global_load_lds((gbl_void*)gsrc, (lds_void*)lds_wave_base, 16, 0, 0);
#endif
}

// 16-byte slot permutation inside a tile row: slot = chunk ^ swz(row).  Chosen so that every ds_read_b128
// 16-lane group ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32) touches 16 distinct 16-B slots of the 256-B bank row:
//   64-B rows  (BKT 32): s = [0,2,3,1][(row >> 2) & 3]      128-B rows (BKT 64): s = (row >> 1) & 7
template <int BKT>
__device__ __forceinline__ int swz(int row) {
    if (BKT == 32) return (0x78 >> (2 * ((row >> 2) & 3))) & 3;
    return (row >> 1) & 7;
}

// Tile geometry: WM x WN waves, each wave owns TM x 4 accumulator tiles of 16x16 (wave tile TM*16 rows x 64 columns).
//   <2,2,4> 128x128 / 4 waves      <4,2,4> 256x128 / 8 waves      <2,4,8> 256x256 / 8 waves (wave tile 128x64)
template <int TAPS, int BKT, int NSTAGE, int WM, int WN, int TM>
__global__ __launch_bounds__(WM * WN * 64) void k_conv_igemm(const half_t* __restrict__ X, const half_t* __restrict__ Wt,
                                                    const float* __restrict__ bias, const half_t* __restrict__ residual,
                                                    half_t* __restrict__ Y, int N, int H, int W, int Cin, int Cout,
                                                    int n_tiles, int total_tiles, const half_t* __restrict__ zero_page,
                                                    int splits, float* __restrict__ partial, float* __restrict__ gn_part,
                                                    const half_t* __restrict__ X2, int Cin1) {
    constexpr int ROWB = BKT * 2;                 // bytes per tile row
    constexpr int CPR = BKT / 8;                  // 16-byte chunks per row
    constexpr int RPI = 1024 / ROWB;              // rows per wave-instruction (1 KiB)
    constexpr int NWAVES = WM * WN;
    constexpr int BMT = WM * TM * 16;             // output rows per workgroup
    constexpr int BNT = WN * 64;                  // output columns per workgroup
    constexpr int AROWS = BMT / NWAVES;           // A rows staged per wave
    constexpr int BROWS = BNT / NWAVES;           // B rows staged per wave
    constexpr int LPO = AROWS / RPI;              // A loads per lane per K-step
    constexpr int LPB = BROWS / RPI;              // B loads per lane per K-step
    static_assert(AROWS % RPI == 0 && BROWS % RPI == 0 && LPO >= 1 && LPB >= 1, "tile rows must split into 1 KiB wave-instructions");
    constexpr int A_BYTES = BMT * ROWB;
    constexpr int TILE_BYTES = A_BYTES;           // offset of the B tile inside a stage
    constexpr int STAGE_BYTES = A_BYTES + BNT * ROWB;
    constexpr int CS_LD = BNT + 8;                // epilogue tile leading dimension (halfs)
    constexpr int KSTEPS = BKT / 32;              // MFMA k-steps per K-step
    PD_STAMP(0);
    constexpr bool PIPE = NSTAGE == 12;           // NSTAGE 12 = two LDS stages + register-pipelined fragment schedule
    constexpr int NST = PIPE ? 2 : NSTAGE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware tile id: workgroup b runs on XCD b % 8; give each XCD a contiguous run of tile ids
    int tile;
    {
        const int b = blockIdx.x, q = total_tiles >> 3, r = total_tiles & 7, xcd = b & 7, i = b >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
    }
    const int m0 = (tile / n_tiles) * BMT, n0 = (tile % n_tiles) * BNT;
    const long long M = (long long)N * H * W;
    const int K = TAPS * Cin;
    const int kc = Cin / BKT;                // K-steps per tap
    const int KI = TAPS * kc;

    // ---- loader role: this lane stages rows {wave*32 + i*RPI + lane/CPR}, 16-byte slot (lane % CPR) of each tile.
    // Running source pointers: advanced by BKT halfs per K-step, re-derived once per filter tap (uniform branch).
    // Two-source input (1x1 convs over a channel concat that is never materialised): channels [0, Cin1) come from X
    // (pixel stride Cin1), the rest from X2 (pixel stride Cin - Cin1); X2 == nullptr -> one tensor, Cin1 == Cin.
    const int lrow = lane / CPR, lpos = lane % CPR;
    int py[LPO], pxx[LPO];
    long long pbase[LPO], pbase2[LPO];
    const half_t* bp[LPB];
    const half_t* ap[LPO];
    int astep[LPO];
    const long long zoff = zero_page - X;
#pragma unroll
    for (int i = 0; i < LPO; ++i) {
        const int r = wave * AROWS + i * RPI + lrow;
        const int c = lpos ^ swz<BKT>(r);
        const long long m = (long long)m0 + r;
        const bool inm = m < M;
        const long long mm = inm ? m : 0;
        const int img = (int)(mm / ((long long)H * W));
        const int rem = (int)(mm - (long long)img * H * W);
        py[i] = inm ? rem / W : -100000;
        pxx[i] = rem - (rem / W) * W;
        pbase[i] = (((long long)img * H + rem / W) * W + pxx[i]) * Cin1 + c * 8;
        if (TAPS == 1) pbase2[i] = inm ? (((long long)img * H + rem / W) * W + pxx[i]) * (Cin - Cin1) + c * 8 : (long long)(zero_page - X2);
    }
#pragma unroll
    for (int i = 0; i < LPB; ++i) {
        const int r = wave * BROWS + i * RPI + lrow;
        bp[i] = Wt + (size_t)(n0 + r) * K + (lpos ^ swz<BKT>(r)) * 8;
    }
    auto set_tap = [&](int tap) {
        const int dy = (TAPS == 1) ? 0 : tap / 3 - 1, dx = (TAPS == 1) ? 0 : tap - (tap / 3) * 3 - 1;
#pragma unroll
        for (int i = 0; i < LPO; ++i) {
            const int yy = py[i] + dy, xx = pxx[i] + dx;
            const bool ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W);
            const long long off = pbase[i] + ((long long)dy * W + dx) * Cin1;
            ap[i] = X + (ok ? off : zoff);
            astep[i] = ok ? BKT : 0;
        }
    };
    char* const wave_dst = smem + wave * (AROWS * ROWB);
    char* const wave_dst_b = smem + TILE_BYTES + wave * (BROWS * ROWB);
    // split-K: this workgroup reduces K-steps [it0, it1) (blockIdx.y = split); splits == 1 -> the whole K range
    const int it0 = (int)((long long)KI * blockIdx.y / splits), it1 = (int)((long long)KI * (blockIdx.y + 1) / splits);
    int ntap = it0 / kc, nc = it0 - (it0 / kc) * kc;
    set_tap(ntap);
#pragma unroll
    for (int i = 0; i < LPO; ++i) ap[i] += astep[i] * nc;
#pragma unroll
    for (int i = 0; i < LPB; ++i) bp[i] += (size_t)it0 * BKT;
    // one 1 KiB LDS-DMA piece of a stage (p < LPO: activation rows, else weight rows) / the K-step bookkeeping after the last
    auto issue_piece = [&](int stage, int p) {
        if (p < LPO) {
#ifndef PD_LAB_NOA
            glds16(ap[p], wave_dst + stage * STAGE_BYTES + p * 1024);
#endif
            ap[p] += astep[p];
        } else {
#ifndef PD_LAB_NOB
            glds16(bp[p - LPO], wave_dst_b + stage * STAGE_BYTES + (p - LPO) * 1024);
#endif
            bp[p - LPO] += BKT;
        }
    };
    auto issue_advance = [&]() {
        if (TAPS == 1) {
            if (X2 != nullptr && ++nc == Cin1 / BKT) {     // the concat's first tensor is exhausted: continue in the second
#pragma unroll
                for (int i = 0; i < LPO; ++i) ap[i] = X2 + pbase2[i];
            }
        } else if (++nc == kc) {
            nc = 0;
            if (++ntap < TAPS) set_tap(ntap);
        }
    };
    auto issue = [&](int stage) {
#pragma unroll
        for (int p = 0; p < LPO + LPB; ++p) issue_piece(stage, p);
        issue_advance();
    };

    // ---- consumer role: fragment byte offset inside a 16-row group (row = lane&15, k-chunk = lane>>4 (+4 per k-step))
    int frag_off[KSTEPS];
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk)
        frag_off[kk] = (lane & 15) * ROWB + ((((lane >> 4) + 4 * kk) ^ swz<BKT>(lane & 15)) << 4);
    float4_t acc[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    constexpr int OPS_ = LPO + LPB;
    PD_STAMP(1);
    if constexpr (PIPE) {
        // Register-pipelined schedule.  A K-step is G = 16 groups of 4 MFMAs (one A fragment x 4 B fragments; groups 0..7 are
        // MFMA k-step 0, 8..15 k-step 1).  The A fragment of group g+3 is read from LDS at the start of group g into the ring
        // slot group g-1 just released (ring of 4 registers), the B fragments of k-step 1 are read during groups 0..3.  The
        // stage hand-over (vmcnt wait + barrier, then the first fragment reads of K-step it+1) sits in front of the LAST THREE
        // groups, whose 12 MFMAs cover the barrier skew and the LDS latency of the next step's first reads.  The LDS-DMA pieces
        // of K-step it+2 follow one per group (an LDS-DMA issue costs ~60+ cycles of the wave's issue slot: behind a group's
        // MFMAs it is free, in a block it stalls the pipe).
        // hipcc only emits lgkmcnt(0) around LDS-DMA kernels, which would expose every read's latency, so the fragment reads
        // are inline-asm ds_read_b128 with hand-counted s_waitcnt lgkmcnt(N): LDS reads return in issue order, N = number of
        // reads issued after the one a group needs (table below).  No scalar loads may sit in this loop (they share lgkmcnt).
        static_assert(BKT == 64 && TM == 8, "pipelined schedule: BKT 64, TM 8");
        static_assert(OPS_ == 8 || OPS_ == 12, "unexpected loads per stage");
        constexpr int G = KSTEPS * TM;
        constexpr int HO = G - 3;                             // hand-over group
        constexpr int PPG = OPS_ == 12 ? 2 : 1;              // LDS-DMA pieces per group
        constexpr int NPG = OPS_ / PPG;                       // groups that carry pieces: 3 from the hand-over + NPG-3 after it
#ifdef PD_LAB_STAMP
        unsigned long long lab_t0 = 0, lab_wait = 0, lab_bar = 0;
        const bool lab_trace = blockIdx.x == 8 && blockIdx.y == 0 && (wave & 3) == 0 && lane == 0;    // waves 0 and 4 share SIMD 0
#define PD_LAB_T0 lab_t0 = __builtin_readcyclecounter()
#define PD_LAB_T1 { const unsigned long long t_ = __builtin_readcyclecounter(); lab_wait += t_ - lab_t0; lab_t0 = t_; }
#define PD_LAB_T2 lab_bar += __builtin_readcyclecounter() - lab_t0
#else
#define PD_LAB_T0
#define PD_LAB_T1
#define PD_LAB_T2
#endif
#ifdef PD_LAB_NOBARRIER
#define PD_LAB_BARRIER
#else
#define PD_LAB_BARRIER __builtin_amdgcn_s_barrier()
#endif
#define PD_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
        const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
        const uint32_t a_base = lds0 + (wm * TM * 16) * ROWB, b_base = lds0 + TILE_BYTES + (wn * 64) * ROWB;
        half8 ar[4], bf[KSTEPS][4];
        issue(0);
        if (it0 + 1 < it1) {
#pragma unroll
            for (int p = 0; p < 3 * PPG; ++p) issue_piece(1, p);
            if (PPG == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        {
            const uint32_t b0 = b_base + frag_off[0], a0 = a_base + frag_off[0];
            PD_DSR(bf[0][0], b0, 0); PD_DSR(bf[0][1], b0, 16 * ROWB); PD_DSR(bf[0][2], b0, 32 * ROWB); PD_DSR(bf[0][3], b0, 48 * ROWB);
            PD_DSR(ar[0], a0, 0); PD_DSR(ar[1], a0, 16 * ROWB); PD_DSR(ar[2], a0, 32 * ROWB);
        }
        int cur = 0;
        for (int it = it0; it < it1; ++it) {
            const uint32_t a_ad0 = a_base + cur * STAGE_BYTES + frag_off[0], a_ad1 = a_base + cur * STAGE_BYTES + frag_off[1];
            const uint32_t b_ad1 = b_base + cur * STAGE_BYTES + frag_off[1];
            const uint32_t a_nx0 = a_base + (cur ^ 1) * STAGE_BYTES + frag_off[0], b_nx0 = b_base + (cur ^ 1) * STAGE_BYTES + frag_off[0];
            const bool more = it + 1 < it1, more2 = it + 2 < it1;
            // The two waves of a SIMD (w, w+4) run the same stream and meet at the barrier every K-step.  Under the default
            // oldest-first issue arbitration one of them races ahead and then idles at the barrier while the other runs alone
            // with all its stalls exposed; alternating the priority by half K-steps keeps both in the pipe for the whole step.
#ifdef PD_LAB_STAMP
#define PD_LAB_G(g) if (lab_trace && it == it0 + 10) g_lab_stamps[7 * 4096 + (wave >> 2) * 32 + (g)] = __builtin_readcyclecounter();
#else
#define PD_LAB_G(g)
#endif
#ifdef PD_LAB_NOPRIO
#define PD_PRIO(g)
#else
#define PD_PRIO(g)                                                                                                       \
        if constexpr ((g) == 0) { if (wave < NWAVES / 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0); }   \
        if constexpr ((g) == G / 2) { if (wave < NWAVES / 2) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(2); }
#endif
            // reads still allowed in flight when group g starts: A(g+1..g+3) plus the B reads interleaved after A(g)
#define PD_WAITN(g_) ((g_) == 0 ? 4 : (g_) == 1 ? 5 : (g_) == 2 ? 6 : (g_) == 3 ? 7 : (g_) == 4 ? 6 : (g_) == 5 ? 5 : (g_) == 6 ? 4 : 3)
#define PD_GROUP(g)                                                                                                      \
    {                                                                                                                    \
        PD_PRIO(g)                                                                                                       \
        PD_LAB_G(g)                                                                                                      \
        if constexpr ((g) + 3 < G) {                                                                                     \
            if constexpr (((g) + 3) / TM == 0) PD_DSR(ar[((g) + 3) & 3], a_ad0, (((g) + 3) % TM) * 16 * ROWB);           \
            else PD_DSR(ar[((g) + 3) & 3], a_ad1, (((g) + 3) % TM) * 16 * ROWB);                                         \
        }                                                                                                                \
        if constexpr ((g) < 4) PD_DSR(bf[1][(g) & 3], b_ad1, ((g) & 3) * 16 * ROWB);                                     \
        if constexpr ((g) == HO) {                                                                                       \
            if (more) {                                                                                                  \
                PD_LAB_T0;                                                                                               \
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(ar[(g) & 3]), "+v"(ar[((g) + 1) & 3]), "+v"(ar[((g) + 2) & 3]) :: "memory"); \
                PD_LAB_T1;                                                                                               \
                PD_LAB_BARRIER;                                                                                          \
                asm volatile("" ::: "memory");                                                                           \
                PD_LAB_T2;                                                                                               \
                PD_DSR(bf[0][0], b_nx0, 0); PD_DSR(bf[0][1], b_nx0, 16 * ROWB);                                          \
                PD_DSR(bf[0][2], b_nx0, 32 * ROWB); PD_DSR(bf[0][3], b_nx0, 48 * ROWB);                                  \
                PD_DSR(ar[0], a_nx0, 0);                                                                                 \
            } else {                                                                                                     \
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ar[(g) & 3]), "+v"(ar[((g) + 1) & 3]), "+v"(ar[((g) + 2) & 3])); \
            }                                                                                                            \
        } else if constexpr ((g) == HO + 1) {                                                                            \
            if (more) PD_DSR(ar[1], a_nx0, 16 * ROWB);                                                                   \
        } else if constexpr ((g) == HO + 2) {                                                                            \
            if (more) PD_DSR(ar[2], a_nx0, 32 * ROWB);                                                                   \
        } else {                                                                                                         \
            asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(ar[(g) & 3]) : "n"(PD_WAITN(g)));                                \
            if constexpr ((g) == 0) asm volatile("" : "+v"(bf[0][0]), "+v"(bf[0][1]), "+v"(bf[0][2]), "+v"(bf[0][3]));    \
            if constexpr ((g) == TM) asm volatile("" : "+v"(bf[1][0]), "+v"(bf[1][1]), "+v"(bf[1][2]), "+v"(bf[1][3]));   \
        }                                                                                                                \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                    \
            acc[(g) % TM][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[(g) / TM][j], ar[(g) & 3], acc[(g) % TM][j], 0, 0, 0); \
        if constexpr ((g) >= HO) {                       /* pieces 0..2 of K-step it+2 into the stage just drained */      \
            if (more2) { _Pragma("unroll") for (int p = 0; p < PPG; ++p) issue_piece(cur, ((g) - HO) * PPG + p); }       \
        } else if constexpr ((g) < NPG - 3) {            /* the remaining pieces of K-step it+1 */                       \
            if (more) {                                                                                                  \
                _Pragma("unroll") for (int p = 0; p < PPG; ++p) issue_piece(cur ^ 1, (3 + (g)) * PPG + p);               \
                if constexpr ((g) == NPG - 4) issue_advance();                                                           \
            }                                                                                                            \
        }                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
    }
            PD_GROUP(0) PD_GROUP(1) PD_GROUP(2) PD_GROUP(3) PD_GROUP(4) PD_GROUP(5) PD_GROUP(6) PD_GROUP(7)
            PD_GROUP(8) PD_GROUP(9) PD_GROUP(10) PD_GROUP(11) PD_GROUP(12) PD_GROUP(13) PD_GROUP(14) PD_GROUP(15)
            PD_LAB_G(16)
#undef PD_GROUP
#undef PD_LAB_G
#undef PD_PRIO
#undef PD_WAITN
            cur ^= 1;
        }
#undef PD_DSR
#ifdef PD_LAB_STAMP
        if (threadIdx.x == 0 && blockIdx.x < 4096 && blockIdx.y == 0) { g_lab_stamps[blockIdx.x * 8 + 6] = lab_wait; g_lab_stamps[blockIdx.x * 8 + 7] = lab_bar; }
#endif
    } else {
    // Software pipeline: NSTAGE LDS stages, loads run NSTAGE-1 K-steps ahead and stay in flight ACROSS the barrier
    // (counted s_waitcnt vmcnt + raw s_barrier; __syncthreads() would drain the LDS-DMA queue -- cdna guide T3/T4).
    constexpr int D = NST - 1;                          // prefetch distance
    constexpr int OPS = LPO + LPB;                         // VMEM ops per lane per stage
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (it0 + d < it1) issue(d);
    int cur = 0, nxt = D % NST;
    for (int it = it0; it < it1; ++it) {
        const int inflight = min(D, it1 - it);             // stages issued and not yet consumed (incl. this one)
        static_assert(OPS == 3 || OPS == 4 || OPS == 6 || OPS == 8 || OPS == 12, "unexpected loads per stage");
#ifndef PD_LAB_NOWAIT                                      // (lab builds only: timing without the load dependency)
        if (D >= 3 && inflight >= 3) {
            if (OPS == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (OPS == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (OPS == 6) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (OPS == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        } else if (D >= 2 && inflight == 2) {
            if (OPS == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else if (OPS == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (OPS == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (OPS == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#ifndef PD_LAB_NOBARRIER
        __builtin_amdgcn_s_barrier();                      // stage `cur` landed for every wave; stage `nxt` is free again
#endif
        asm volatile("" ::: "memory");
        if (it + D < it1) issue(nxt);
        const char* As = smem + cur * STAGE_BYTES + (wm * TM * 16) * ROWB;
        const char* Bs = smem + cur * STAGE_BYTES + TILE_BYTES + (wn * 64) * ROWB;
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            half8 a[TM], b[4];
#ifdef PD_LAB_NOLDS                                        // (lab builds only: MFMA-only ceiling, fragments from registers)
#pragma unroll
            for (int j = 0; j < 4; ++j) { b[j] = (half8){(half_t)j, 1, 2, 3, 4, 5, 6, (half_t)lane}; asm volatile("" : "+v"(b[j])); }
#pragma unroll
            for (int i = 0; i < TM; ++i) { a[i] = (half8){(half_t)i, 1, 2, 3, 4, 5, 6, (half_t)kk}; asm volatile("" : "+v"(a[i])); }
#else
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const half8*>(Bs + j * 16 * ROWB + frag_off[kk]);
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const half8*>(As + i * 16 * ROWB + frag_off[kk]);
#endif
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j], a[i], acc[i][j], 0, 0, 0);
        }
        cur = (cur + 1 == NST) ? 0 : cur + 1;
        nxt = (nxt + 1 == NST) ? 0 : nxt + 1;
    }
    }
    PD_STAMP(2);
    __syncthreads();                                       // all fragment reads done before the tile is reused
#ifdef PD_LAB_NOEPI                                        // (lab builds only: mainloop without the epilogue)
    {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (t == 123456.789f) Y[0] = (half_t)t;
        return;
    }
#endif

    // The MFMAs are issued as (weights x activations), i.e. the accumulator tile is the TRANSPOSE of the usual layout: lane
    // holds pixel m = 16 i + (lane & 15) and the four CONSECUTIVE channels n = 16 j + 4 (lane >> 4) + r -- 8-/16-byte packed
    // stores along the channel dimension instead of scalar ones.
    if (partial != nullptr) {                              // split-K: raw f32 partial tile, reduced by k_splitk_reduce
        float* P = partial + (size_t)blockIdx.y * (size_t)M * Cout;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const long long m = (long long)m0 + wm * TM * 16 + i * 16 + (lane & 15);
                if (m < M && n < Cout) *reinterpret_cast<float4_t*>(P + (size_t)m * Cout + n) = acc[i][j];
            }
        }
        return;
    }
    // ---- epilogue: acc (+bias) -> f16 -> LDS [BMT][CS_LD] -> coalesced 16-byte rows (+residual)
    half_t* Cs = reinterpret_cast<half_t*>(smem);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int nl = wn * 64 + j * 16 + (lane >> 4) * 4;
        float4_t bv = (float4_t){0.f, 0.f, 0.f, 0.f};
        if (bias != nullptr && n0 + nl < Cout) bv = *reinterpret_cast<const float4_t*>(bias + n0 + nl);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int ml = wm * TM * 16 + i * 16 + (lane & 15);
            half4 h;
#pragma unroll
            for (int r = 0; r < 4; ++r) h[r] = (half_t)(acc[i][j][r] + bv[r]);
            *reinterpret_cast<half4*>(&Cs[ml * CS_LD + nl]) = h;
        }
    }
    PD_STAMP(3);
    __syncthreads();
    PD_STAMP(4);
    constexpr int CT = BNT / 8;                            // column threads (one channel octet each)
    constexpr int RPP = NWAVES * 64 / CT;                  // rows per pass
    const int col8 = (tid % CT) * 8;
    float gs = 0.f, gq = 0.f;                              // fused GroupNorm partial statistics of this thread's 8 channels
#pragma unroll
    for (int p = 0; p < BMT / RPP; ++p) {
        const int row = p * RPP + tid / CT;
        const long long m = (long long)m0 + row;
        if (m < M && n0 + col8 < Cout) {
            half8 v = *reinterpret_cast<const half8*>(&Cs[row * CS_LD + col8]);
            const size_t o = (size_t)m * Cout + n0 + col8;
            if (residual != nullptr) {
                const half8 rv = *reinterpret_cast<const half8*>(residual + o);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] + (float)rv[e]);
            }
#ifdef PD_LAB_NOSTORE                                      // (lab builds only: epilogue without the global stores)
            if (v[0] == (half_t)12345.f) *reinterpret_cast<half8*>(Y + o) = v;
#else
            *reinterpret_cast<half8*>(Y + o) = v;
#endif
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; gs += f; gq += f * f; }
        }
    }
    PD_STAMP(5);
    if (gn_part != nullptr) {
        // Fused GroupNorm statistics: per-(image, row-chunk, channel octet) sum / sum of squares of the tile just written,
        // layout [img][chunk][Cout/8][2].  Any GroupNorm(32) whose group size is a multiple of 8 channels -- also over a
        // channel concat of two such tensors -- is assembled from these by gn_finalize_oct without re-reading the tensor.
        // Host guarantees the whole tile lies inside one image.
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);
        red[tid * 2] = gs; red[tid * 2 + 1] = gq;
        __syncthreads();
        if (tid < CT && n0 + tid * 8 < Cout) {
            float s1 = 0.f, q1 = 0.f;
            for (int r = 0; r < RPP; ++r) { s1 += red[(r * CT + tid) * 2]; q1 += red[(r * CT + tid) * 2 + 1]; }
            const int hw = H * W, chunks = hw / BMT;
            const int img = m0 / hw, chunk = (m0 - img * hw) / BMT;
            float* dst = gn_part + (((size_t)img * chunks + chunk) * (Cout >> 3) + (n0 >> 3) + tid) * 2;
            dst[0] = s1; dst[1] = q1;
        }
    }
}

// sum of split-K partials (fixed order) + bias -> f16 (+ residual) -> Y, plus the same fused GroupNorm octet partials as
// the direct epilogue.  Block = SK_ROWS consecutive pixels x 64 channel octets; thread (octet, lane q) takes rows q, q+4, ...
#define SK_ROWS 16
__global__ __launch_bounds__(256) void k_splitk_reduce(const float* __restrict__ partial, int splits, long long M, int Cout,
                                                       const float* __restrict__ bias, const half_t* __restrict__ residual,
                                                       half_t* __restrict__ Y, float* __restrict__ gn_part, int hw) {
    __shared__ float s_st[4][64][2];
    const int ol = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int oct = blockIdx.y * 64 + ol;
    const bool live = oct < (Cout >> 3);
    const long long m0 = (long long)blockIdx.x * SK_ROWS;
    const size_t MC = (size_t)M * Cout;
    float gs = 0.f, gq = 0.f;
    if (live) {
        float bv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = bias ? bias[oct * 8 + e] : 0.f;
        float a[SK_ROWS / 4][8];
#pragma unroll
        for (int j = 0; j < SK_ROWS / 4; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) a[j][e] = 0.f;
        // four splits' loads in flight per row (the sum order stays sp = 0, 1, 2, ...: deterministic); one split at a time was a
        // chain of L2 round trips -- 10-14 us per launch at batch 1, where 68 of these run per forward
        for (int sp0 = 0; sp0 < splits; sp0 += 4) {
#pragma unroll
            for (int j = 0; j < SK_ROWS / 4; ++j) {
                const long long m = m0 + q + 4 * j;
                if (m < M) {
                    float4 v0[4], v1[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int sp = min(sp0 + u, splits - 1);
                        const float* src = partial + (size_t)sp * MC + (size_t)m * Cout + (size_t)oct * 8;
                        v0[u] = *reinterpret_cast<const float4*>(src);
                        v1[u] = *reinterpret_cast<const float4*>(src + 4);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (sp0 + u < splits) {
                            a[j][0] += v0[u].x; a[j][1] += v0[u].y; a[j][2] += v0[u].z; a[j][3] += v0[u].w;
                            a[j][4] += v1[u].x; a[j][5] += v1[u].y; a[j][6] += v1[u].z; a[j][7] += v1[u].w;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < SK_ROWS / 4; ++j) {
            const long long m = m0 + q + 4 * j;
            if (m >= M) continue;
            const size_t o = (size_t)m * Cout + (size_t)oct * 8;
            half8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (half_t)(a[j][e] + bv[e]);
            if (residual != nullptr) {
                const half8 rv = *reinterpret_cast<const half8*>(residual + o);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] + (float)rv[e]);
            }
            *reinterpret_cast<half8*>(Y + o) = v;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; gs += f; gq += f * f; }
        }
    }
    if (gn_part == nullptr) return;                        // (uniform) host guarantees hw % SK_ROWS == 0: one image per block
    s_st[q][ol][0] = gs; s_st[q][ol][1] = gq;
    __syncthreads();
    if (q == 0 && live) {
        gs = (s_st[0][ol][0] + s_st[1][ol][0]) + (s_st[2][ol][0] + s_st[3][ol][0]);
        gq = (s_st[0][ol][1] + s_st[1][ol][1]) + (s_st[2][ol][1] + s_st[3][ol][1]);
        const int chunks = hw / SK_ROWS;
        const long long img = m0 / hw;
        const int chunk = (int)((m0 - img * hw) / SK_ROWS);
        float* dst = gn_part + (((size_t)img * chunks + chunk) * (Cout >> 3) + oct) * 2;
        dst[0] = gs; dst[1] = gq;
    }
}

thread_local int g_force_bk = 0;       // tuning hook: 32 / 64 forces the K-step, 0 = automatic
thread_local int g_force_stages = 0;   // tuning hook: 2 / 3 / 4 LDS stages, 0 = automatic
thread_local int g_force_splits = 0;   // tuning hook: >= 1 forces the split-K factor, 0 = automatic
thread_local float* g_dbg_splitk_ws = nullptr; thread_local size_t g_dbg_splitk_floats = 0;   // split-K workspace for the stand-alone conv entry point
thread_local int g_force_wmw = 0;      // tuning hook, tile geometry: 2 = 128x128/4 waves, 4 = 256x128/8 waves, 8 = 256x256/8 waves, 0 = automatic

template <int TAPS, int BKT, int NSTAGE, int WM, int WN, int TM>
static int launch_conv(dim3 grid, size_t smem, hipStream_t s, const half_t* X, const half_t* Wt, const float* bias,
                       const half_t* residual, half_t* Y, int N, int H, int W, int Cin, int Cout, int n_tiles, int total,
                       const half_t* zero_page, int splits, float* partial, float* gnp, const half_t* X2, int Cin1) {
    auto kern = k_conv_igemm<TAPS, BKT, NSTAGE, WM, WN, TM>;
    if (smem > 65536) PD_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    kern<<<grid, WM * WN * 64, smem, s>>>(X, Wt, bias, residual, Y, N, H, W, Cin, Cout, n_tiles, total, zero_page, splits, partial, gnp, X2, Cin1);
    return PDHIP_OK;
}

// does conv_igemm route this layer to the halo-resident kernel? (tuning hook: tile geometry 32 forces it, any other forced
// geometry / K-step / stage count disables it)
bool conv_uses_halo(int N, int H, int W, int Cin, int Cout, int Cout_pad, int taps, size_t splitk_ws_floats) {
    if (taps != 9 || !conv3x3_halo_eligible(N, H, W, Cin, Cout_pad)) return false;
    if (g_force_wmw == 32) return true;
    // automatic use only when the tiles alone fill the chip: with the chunk split (small-M layers) the halo kernel measures
    // no better than the 128x128 split-K path (tools/bench_splitk.py: 54 vs 51 us at 32x32 / 512 -> 512)
    return g_force_wmw == 0 && g_force_bk == 0 && g_force_stages == 0 && g_force_splits == 0 &&
           conv3x3_halo_splits(N, H, W, Cin, Cout, Cout_pad, 0) == 1;
}

// THE routing decision of conv_igemm() -- the one copy of it (run_res predicts the route of conv2 to decide whether the
// up-sampled x branch needs materialising; a second copy of the predicate could drift from the one that launches): 0 = halo-resident
// kernel, 1 = k_conv_sk (plan in *plan), 2 = the implicit-GEMM kernel, 3 = k_conv_ht (256 x 64 halo tiles: the 64^2 / 128^2 levels at batch 1-2).
int conv_route(int N, int H, int W, int Cin, int Cout, int Cout_pad, int taps, size_t splitk_ws_floats, bool two_source, bool in_up,
               bool apply, SkPlan* plan) {
    if (!two_source && conv_uses_halo(N, H, W, Cin, Cout, Cout_pad, taps, splitk_ws_floats)) return 0;
    if (taps == 9 && !two_source && !in_up && !apply && g_force_wmw == 0 && g_force_bk == 0 && g_force_stages == 0 && g_force_splits == 0 &&
        conv_ht_routes(N, H, W, Cin, Cout, Cout_pad, splitk_ws_floats)) return 3;
    if (!in_up && !apply && g_force_wmw == 0 && g_force_bk == 0 && g_force_stages == 0 && g_force_splits == 0) {
        const SkPlan pl = conv_sk_plan(N, H, W, Cin, Cout, Cout_pad, taps, two_source, splitk_ws_floats);
        if (pl.bm > 0) {
            if (plan) *plan = pl;
            return 1;
        }
    }
    return 2;
}
bool conv_routes_small(int N, int H, int W, int Cin, int Cout, int Cout_pad, int taps, size_t splitk_ws_floats) {   // k_conv_sk or k_conv_ht
    const int r = conv_route(N, H, W, Cin, Cout, Cout_pad, taps, splitk_ws_floats, false, false, false, nullptr);
    return r == 1 || r == 3;
}

int splitk_reduce(const float* partial, int splits, long long M, int Cout, const float* bias, const half_t* residual, half_t* Y,
                  float* gn_part, int hw, hipStream_t s) {
    dim3 gr((unsigned)((M + SK_ROWS - 1) / SK_ROWS), (unsigned)(((Cout >> 3) + 63) / 64));
    k_splitk_reduce<<<gr, 256, 0, s>>>(partial, splits, M, Cout, bias, residual, Y, gn_part, hw);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}

int conv_igemm(const half_t* X, const half_t* Wt, const float* bias, const half_t* residual, half_t* Y, int N, int H,
               int W, int Cin, int Cout, int Cout_pad, int taps, const half_t* zero_page, hipStream_t s, float* splitk_ws,
               size_t splitk_ws_floats, float* gn_part, int* gn_fused, const half_t* X2, int Cin1, const float* apply_table, int res_up, int in_up) {
    PD_REQUIRE(taps == 1 || taps == 9, "conv_igemm: taps must be 1 or 9");
    if (X2 == nullptr) Cin1 = Cin;
    PD_REQUIRE(X2 == nullptr || (taps == 1 && Cin1 > 0 && Cin1 < Cin && Cin1 % 64 == 0 && (Cin - Cin1) % 64 == 0),
               "conv_igemm: a two-source input needs a 1x1 conv and channel counts that are multiples of 64");
    PD_REQUIRE(Cin % 32 == 0 && Cout % 8 == 0 && Cout_pad % 128 == 0 && Cout_pad >= Cout,
               "conv_igemm: need Cin %% 32 == 0, Cout %% 8 == 0, padded Cout %% 128 == 0 (Cin=%d Cout=%d pad=%d)", Cin, Cout, Cout_pad);
    const long long M = (long long)N * H * W;
    // large-image 3x3 layers: halo-resident kernel (2.1x less L2 -> LDS traffic per flop) once it fills the chip
    // (tuning hook: tile geometry 32 forces it, any other forced geometry disables it)
    SkPlan pl{0, 0, 0, 0};
    const int route = conv_route(N, H, W, Cin, Cout, Cout_pad, taps, splitk_ws ? splitk_ws_floats : 0, X2 != nullptr, in_up != 0, apply_table != nullptr, &pl);
    if (route == 0) {
        // (the first PD_SK_TICKET_FLOATS words of the workspace are k_conv_sk's self-resetting ticket counters: the halo kernel's f32
        // partials -- only under the forced-split tuning hooks -- must not land on them, or the next split k_conv_sk launch never
        // sees its last ticket)
        float* hws = splitk_ws != nullptr && splitk_ws_floats > PD_SK_TICKET_FLOATS ? splitk_ws + PD_SK_TICKET_FLOATS : nullptr;
        const size_t hfl = hws != nullptr ? splitk_ws_floats - PD_SK_TICKET_FLOATS : 0;
        return conv3x3_halo(X, Wt, bias, residual, Y, N, H, W, Cin, Cout, Cout_pad, zero_page, s, gn_part, gn_fused, hws, hfl, apply_table,
                            res_up, in_up);
    }
    if (route == 3) {
        int chunks = 0;
        const int rc = conv_ht(X, Wt, bias, residual, Y, N, H, W, Cin, Cout, Cout_pad, zero_page, s, gn_part, &chunks, res_up, splitk_ws, splitk_ws_floats);
        if (gn_fused) *gn_fused = chunks;
        return rc;
    }
    // small-M layers (output tiles do not fill the chip): small tiles, deep staging, split-K combined inside the launch
    if (route == 1) {
        float* gnp = (gn_part != nullptr && (((long long)H * W) % pl.bm == 0 || pl.bm == 2 * H * W)) ? gn_part : nullptr;
        return conv_sk(pl, X, Wt, bias, residual, Y, N, H, W, Cin, Cout, Cout_pad, taps, zero_page, s, splitk_ws, splitk_ws_floats, gnp,
                       gn_fused, X2, Cin1, res_up);
    }
    // (the first PD_SK_TICKET_FLOATS words of the split-K workspace are k_conv_sk's ticket counters)
    if (splitk_ws != nullptr) {
        if (splitk_ws_floats > PD_SK_TICKET_FLOATS) { splitk_ws += PD_SK_TICKET_FLOATS; splitk_ws_floats -= PD_SK_TICKET_FLOATS; }
        else { splitk_ws = nullptr; splitk_ws_floats = 0; }
    }
    PD_REQUIRE(in_up == 0, "conv_igemm: an up-sampled input needs a layer the halo-resident kernel takes");
    PD_REQUIRE(res_up == 0, "conv_igemm: an up-sampled residual needs a layer the halo-resident kernel (unsplit) or k_conv_sk takes");
    PD_REQUIRE(apply_table == nullptr, "conv_igemm: an input transform needs a layer the halo-resident kernel takes");
    const int bk = (g_force_bk == 32 || Cin % 64 != 0) ? 32 : 64;
    // tile geometry: 256x256 (wave tile 128x64: 25 % fewer LDS reads per MFMA, half the L2 traffic) once it still yields
    // at least one workgroup per CU; 128x128 otherwise
    int geo = g_force_wmw;
    if (geo != 2 && geo != 4 && geo != 8 && geo != 16)
        geo = (Cout_pad % 256 == 0 && ((M + 255) / 256) * (Cout_pad / 256) >= 256) ? 8 : 2;
    if (geo == 8 && Cout_pad % 256 != 0) geo = 2;
    const int bmt = geo == 2 ? 128 : 256, bnt = geo == 8 ? 256 : 128;    // geo 16: 256x128, 4 waves with 128x64 wave tiles
    const int m_tiles = (int)((M + bmt - 1) / bmt), n_tiles = Cout_pad / bnt;
    const int total = m_tiles * n_tiles;
    // small-M 3x3 layers (32x32 .. 8x8 levels) leave most of the 256 CUs idle: split the K loop across workgroups, aiming at
    // ~512 work items but never more than 8 splits -- the f32 partial round trip costs more than it buys beyond that, and
    // never for 1x1 convs, whose K loop is too short to amortise it (tools/bench_splitk.py holds the measured table)
    const int KI = taps * (Cin / bk);
    int splits = 1;
    if (splitk_ws != nullptr && taps == 9 && total <= 256) {
        // up to 8 splits in general; tiny-M layers (the 8x8 / 16x16 levels at batch 1-2: 8-16 tiles, a weight stream of 19-38 MB
        // per conv) go on splitting while the f32 partials stay small (<= 8 MB: L2-resident), up to 32 -- measured at batch 1
        // (tools/time_unet.py): 22-24 us -> see DESIGN section 8
        int cap = 8;
        while (cap < 32 && total * cap < 256 && (size_t)(2 * cap) * M * Cout * sizeof(float) <= (size_t)8 << 20) cap *= 2;
        splits = std::min(std::min((512 + total / 2) / total, KI / 2), cap);
        while (splits > 1 && (size_t)splits * M * Cout > splitk_ws_floats) --splits;
    }
    if (g_force_splits >= 1 && splitk_ws != nullptr)
        splits = (int)std::min<size_t>(std::min(g_force_splits, std::max(KI / 2, 1)), splitk_ws_floats / ((size_t)M * Cout));
    if (splits < 1 || X2 != nullptr) splits = 1;
    float* partial = splits > 1 ? splitk_ws : nullptr;
    dim3 grid(total, splits);
    // fused GroupNorm partial statistics: only when a tile never straddles two images
    const bool fuse = gn_part != nullptr && splits == 1 && ((long long)H * W) % bmt == 0;
    const bool fuse_sk = gn_part != nullptr && splits > 1 && ((long long)H * W) % SK_ROWS == 0;   // stats from the reduce kernel
    if (gn_fused) *gn_fused = fuse ? (int)(((long long)H * W) / bmt) : fuse_sk ? (int)(((long long)H * W) / SK_ROWS) : 0;
    float* gnp = fuse ? gn_part : nullptr;
    // 12 = two stages + register-pipelined schedule: the default for the 256x256 tile (+8..16 % over the compiler's schedule)
    int stages = g_force_stages ? g_force_stages : (bk == 64 ? (geo == 8 ? 12 : 2) : 3);
    const bool pipe = stages == 12 && bk == 64 && (geo == 8 || geo == 16);   // written for the 128-row wave tile
    if (stages == 12) stages = 2;
    if (stages < 2) stages = 2;
    if (stages > 4) stages = 4;
    const size_t stage_bytes = (size_t)(bmt + bnt) * bk * 2;
    while (stages > 2 && stages * stage_bytes > 160 * 1024) --stages;
    const size_t smem = std::max<size_t>((size_t)stages * stage_bytes, (size_t)bmt * (bnt + 8) * 2);
    if (pipe) stages = 12;
#define ARGS grid, smem, s, X, Wt, bias, residual, Y, N, H, W, Cin, Cout, n_tiles, total, zero_page, splits, partial, gnp, X2, Cin1
#define BY_STAGE(T, B, WM_, WN_, TM_)                                                   \
    (stages == 2 ? launch_conv<T, B, 2, WM_, WN_, TM_>(ARGS) : stages == 3 ? launch_conv<T, B, 3, WM_, WN_, TM_>(ARGS) \
                                                              : launch_conv<T, B, 4, WM_, WN_, TM_>(ARGS))
#define BY_STAGE64(T, WM_, WN_, TM_) (stages == 12 ? launch_conv<T, 64, 12, WM_, WN_, TM_>(ARGS) : BY_STAGE(T, 64, WM_, WN_, TM_))
#define BY_GEO(T, B) (geo == 2 ? BY_STAGE(T, B, 2, 2, 4) : geo == 4 ? BY_STAGE(T, B, 4, 2, 4) : geo == 8 ? BY_STAGE(T, B, 2, 4, 8) : BY_STAGE(T, B, 2, 2, 8))
#define BY_GEO64(T) (geo == 2 ? BY_STAGE(T, 64, 2, 2, 4) : geo == 4 ? BY_STAGE(T, 64, 4, 2, 4) : geo == 8 ? BY_STAGE64(T, 2, 4, 8) : BY_STAGE64(T, 2, 2, 8))
    int rc;
    if (taps == 9) rc = (bk == 64) ? BY_GEO64(9) : BY_GEO(9, 32);
    else rc = (bk == 64) ? BY_GEO64(1) : BY_GEO(1, 32);
#undef BY_GEO64
#undef BY_GEO
#undef BY_STAGE64
#undef BY_STAGE
#undef ARGS
    if (rc) return rc;
    if (splits > 1) {
        return splitk_reduce(partial, splits, M, Cout, bias, residual, Y, fuse_sk ? gn_part : nullptr, H * W, s);
    }
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}

#ifdef PD_LAB_STAMP
extern "C" int pdhip_lab_read_stamps(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_lab_stamps), sizeof(unsigned long long) * n);
}
#endif
}  // namespace pdnn
