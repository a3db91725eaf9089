// Row U1, small-M layers: 3x3 / 1x1 convolutions whose output tile count does not fill the 256 CUs -- the 128^2 ... 8^2 levels of
// the UNet at batch 1 (view-parallel: one view per GPU), the 32^2 ... 8^2 levels at batch 8, every qkv / proj / skip 1x1 of the
// small levels (models/DDNM/guided_diffusion/unet.py:143-305, SURVEY Appendix A "small-M tail").
//
// Same operand formats and fragment layout as k_conv_igemm (nn_gemm.hip): NHWC f16 activations gathered on the fly, weights
// [Cout_pad][taps*Cin], K-step 64, LDS-DMA staging with the bank swizzle on the source address.  What differs is the decomposition:
//   * tile shapes 128x128 / 128x64 / 64x64 / 64x32 (4 waves as 2x2) chosen per layer so that tiles x K-splits is 256..512 workgroups
//     with the SMALLEST split factor -- a tiny-M layer is a weight stream (19-38 MB per conv at the 16^2 / 8^2 levels), so it is cut
//     along Cout first (64x32 tiles: every CU streams its own 74 KB of weights) and along K only as far as needed;
//   * 3-4 LDS stages (2 for the 128x128 tile) so that a workgroup has its next K-steps in flight instead of exposing one memory
//     round trip per step (the 2-stage igemm at 4.5 K-steps per workgroup ran at 1.6 TB/s);
//   * the split-K combine happens INSIDE the launch (cdna guide section 5, "in-launch split-K reduction"): every K-slice writes its f32
//     accumulators write-through (sc1) in register order -- 16 bytes per lane, fully coalesced -- drains, and takes a ticket; the
//     workgroup that draws the last ticket re-reads all slices in slice order (fixed order: deterministic, independent of which
//     workgroup arrives last) with sc1 loads and runs the ordinary epilogue: bias, f16, residual, coalesced NHWC stores and the fused
//     GroupNorm octet partials.  No k_splitk_reduce launch, no f32 round trip through a second kernel (0.97 ms of a 6.6 ms batch-1
//     forward), and 1x1 convs can be split too.
#include "nn_common.h"
using namespace pdhip;
namespace pdnn {

typedef __attribute__((address_space(3))) void sk_lds_void;
typedef const __attribute__((address_space(1))) void sk_gbl_void;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

namespace {

template <int AUX = 0>
__device__ __forceinline__ void sk_glds16(const void* gsrc, void* lds_wave_base) {
#ifndef PD_LAB_SK_NOGLDS                                   // (lab builds only: the loop without its L2 -> LDS traffic)
    __builtin_amdgcn_global_load_lds((sk_gbl_void*)gsrc, (sk_lds_void*)lds_wave_base, 16, 0, AUX);
#endif
}
#ifndef PD_SK_B_AUX
#define PD_SK_B_AUX 0                                       // cache policy of the weight stream (lab: 2 = nt, 16 = sc1: both bypass the CU's L1)
#endif
__device__ __forceinline__ int sk_swz(int row) { return (row >> 1) & 7; }       // 128-byte rows: 16-byte slot ^= (row >> 1) & 7

#ifdef PD_LAB_SK_STAMP                                      // (lab builds only: where one wave's loop time goes, s_memtime cycles)
__device__ unsigned long long g_sk_stamps[16 * 64];
#define SK_CLK() __builtin_readcyclecounter()
#endif
// s_waitcnt lgkmcnt(n) for a compile-time-unrolled n in [0, 15], tying the A fragment and the B fragments of the group to the wait
// ("+v": the MFMAs that read them cannot be scheduled above it -- cdna guide section 5.7, form (ii))
template <int TN_>
__device__ __forceinline__ void sk_wait_lgkm_tied(int n, half8& a, half8 (&b)[TN_]) {
#define SK_W(NN)                                                                                                  \
    case NN:                                                                                                      \
        if constexpr (TN_ == 4) asm volatile("s_waitcnt lgkmcnt(" #NN ")" : "+v"(a), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3 % TN_])); \
        else if constexpr (TN_ == 2) asm volatile("s_waitcnt lgkmcnt(" #NN ")" : "+v"(a), "+v"(b[0]), "+v"(b[1 % TN_]));                     \
        else asm volatile("s_waitcnt lgkmcnt(" #NN ")" : "+v"(a), "+v"(b[0]));                                   \
        break;
    switch (n) {
        SK_W(0) SK_W(1) SK_W(2) SK_W(3) SK_W(4) SK_W(5) SK_W(6) SK_W(7) SK_W(8) SK_W(9) SK_W(10) SK_W(11) SK_W(12) SK_W(13) SK_W(14) SK_W(15)
        default: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b[0])); break;
    }
#undef SK_W
}
template <int N> __device__ __forceinline__ void sk_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// grid: total_tiles * splits workgroups (1-D) of KG * 256 threads.  tickets[tile] must be zero at launch; the last arriver resets it.
// KG "K-groups" of 4 waves share one output tile: group g runs K-steps it0 + g, it0 + g + KG, ... of the workgroup's K slice on its
// own ring of NST LDS stages, the groups' accumulators are summed through LDS (fixed order) after the loop.  One wave's K-step is a
// serial chain -- LDS-DMA issue (~80 cycles of blocked issue slot per 1 KiB piece: 250-650 cycles), fragment reads, 2-32 MFMAs
// (tools/lab_sk.sh stamp: 800 cycles per step for the 64x32 tile, 1 600 for 128x128, at one wave per SIMD) -- so a second / fourth
// wave per SIMD working on ANOTHER K-step is what overlaps it, without a second workgroup's slab traffic.
// LS ("loader-specialised", KG == 1): the workgroup has EIGHT waves -- waves 0-3 only run the fragment reads + MFMAs of the tile
// (2x2 as before), waves 4-7 only issue the LDS-DMA pieces (wave 4 + w stages what wave w staged): a SIMD then holds one compute
// wave and one loader wave, and the ~80 cycles a wave's issue slot is blocked per LDS-DMA piece (640 of a 128x128 step's 1 600
// cycles) run beside the MFMAs of the partner instead of in front of them.
// NL (LS only): loader waves per workgroup, 4 or 8.  Round 5 (tools/ub/ub_lds.hip, profiles/r05_ub_lds.txt): a wave's issue slot is blocked
// ~100 cycles per LDS-DMA piece, so FOUR loaders deliver a 128x128 step's 32 pieces in ~810 cycles (40 B/clk) against 544 cycles of MFMA
// -- the loaders, not the LDS port (ds_read_b128 measured at 256 B/clk, 25 % busy), were the pole of the K-step; EIGHT loaders (two per
// SIMD beside one compute wave, 12 waves at <= 168 VGPRs) issue in parallel up to the ~60 B/clk the CU's L2 -> LDS path delivers.
template <int TAPS, int BM, int BN, int NST, int KG, bool LS, int NL = 4>
__global__ __launch_bounds__(LS ? (4 + NL) * 64 : KG * 256) void k_conv_sk(const half_t* __restrict__ X, const half_t* __restrict__ Wt, const float* __restrict__ bias,
                                                      const half_t* __restrict__ residual, half_t* __restrict__ Y, int N, int H, int W, int Cin,
                                                      int Cout, int n_tiles, int total_tiles, const half_t* __restrict__ zero_page, int splits,
                                                      float* __restrict__ slabs, unsigned* __restrict__ tickets, float* __restrict__ gn_part,
                                                      const half_t* __restrict__ X2, int Cin1, int m_fast,
                                                      const half_t* __restrict__ XS, const half_t* __restrict__ XS2, int Cs1, int Cskip, int res_up) {
    // TAPS == 10 (round 4): a 3x3 conv with the ResBlock's skip 1x1 conv appended to its K loop -- out = W2 * im2col(h) + Wskip * x
    // (unet.py:255 `return self.skip_connection(x) + h`): nine taps over X (Cin channels) and a tenth, centre-only "tap" over the
    // block input x = [XS (Cs1 channels) | XS2 (Cskip - Cs1)] (a never-materialised concat, XS2 may be null); weights
    // [Cout_pad][9 Cin + Cskip], bias = b2 + bskip.  One launch and one rounding instead of conv -> f16 -> + f16(skip conv).
    constexpr bool SKIPK = TAPS == 10;
    constexpr int XT = SKIPK ? 9 : TAPS;                // taps over X
    constexpr int ROWB = 128, RPI = 8;                 // bytes per tile row (K-step 64), rows per 1 KiB wave-instruction
    static_assert(NL == 4 || (LS && NL == 8 && BM >= 64 && BN >= 64), "loader waves: 4, or 8 in the loader-specialised form");
    constexpr int LPO = BM / (8 * NL), LPB = BN / (8 * NL);   // LDS-DMA pieces per loading wave per K-step: activation rows, weight rows
#if defined(PD_LAB_SK_NOA)                                 // (lab: fill diagnostics -- one operand stream compiled out)
    constexpr int OPS = LPB;
#elif defined(PD_LAB_SK_NOB)
    constexpr int OPS = LPO;
#else
    constexpr int OPS = LPO + LPB;
#endif
    constexpr int TM = BM / 32, TN = BN / 32;          // accumulator tiles per wave (wave tile BM/2 x BN/2)
    constexpr int A_BYTES = BM * ROWB, STAGE_BYTES = (BM + BN) * ROWB;
    constexpr int CS_LD = BN + 8;
    constexpr int D = NST - 1;                         // prefetch distance
    static_assert(!LS || KG == 1, "loader specialisation replaces the K-groups");
    constexpr int NT = LS ? (4 + NL) * 64 : KG * 256;
    constexpr int RING_BYTES = NST * STAGE_BYTES;
    constexpr int FLAG_OFF = KG * RING_BYTES;          // "I drew the last ticket" (ONE __shared__ object: cdna guide section 5 trap 4a)
    static_assert(NST >= 2 && NST <= 8 && OPS * (NST - 2) <= 63, "stage count / vmcnt range");
    static_assert(BM * CS_LD * 2 <= KG * RING_BYTES && NT * 4 * 4 <= KG * RING_BYTES && (KG - 1) * BM * BN * 4 <= KG * RING_BYTES,
                  "epilogue staging must fit the stages");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // (provably wave-uniform: LDS-DMA bases stay in SGPRs)
    const int grp = LS ? 0 : wave >> 2, w4 = wave & 3, t4 = tid & 255;
    const int lw = LS ? (wave >= 4 ? wave - 4 : 0) : w4;   // index among the NL loading waves
    const bool loader = !LS || wave >= 4, consumer = !LS || wave < 4;      // LS: the role of this wave
    const int wm = w4 >> 1, wn = w4 & 1;
    // XCD-aware work id: workgroup b runs on XCD b % 8; every XCD gets a contiguous run of (tile, split) ids, so the K-slices of a
    // tile and the n-tiles sharing an activation tile sit on one L2
    int wid;
    {
        const int total = total_tiles * splits;
        const int b = blockIdx.x, q = total >> 3, r = total & 7, xcd = b & 7, i = b >> 3;
        wid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
    }
    const int tile = wid / splits, split = wid - tile * splits;
    // tile order inside an XCD's run: n-tiles of one pixel tile back-to-back (they share the activation tile), or -- m_fast, the
    // weight-heavy layers of the 16^2 / 8^2 levels, 19-38 MB of weights against <= 4 MB of activations -- the pixel tiles of one
    // n-tile back-to-back, so that a weight slice is fetched from HBM once and its other readers hit the XCD's L2
    const int m_tiles = total_tiles / n_tiles;
    const int m0 = (m_fast ? tile % m_tiles : tile / n_tiles) * BM, n0 = (m_fast ? tile / m_tiles : tile % n_tiles) * BN;
    const int M = N * H * W, HWp = H * W;              // (host: M < 2^31)
    const int K = XT * Cin + (SKIPK ? Cskip : 0);
    const int kc = Cin >> 6;                           // K-steps per tap
    const int KI = XT * kc + (SKIPK ? (Cskip >> 6) : 0);

    // ---- loader role: lane stages 16-byte slot (lane % 8) of row (lane / 8) of each of its pieces.  Source addresses are formed
    // from (tap, channel chunk) at every issue, branch-free: per piece a base pointer and a 9-bit "tap inside the image" mask,
    // the tap's pixel offset is wave-uniform; out-of-image taps and rows beyond M read the zero page.
    // Two-source input (1x1 convs over a never-materialised channel concat): channels [0, Cin1) come from X, the rest from X2.
    const int lrow = lane >> 3, lpos = lane & 7;
    const half_t* abase[LPO];
    const half_t* abase2[LPO];
    unsigned amask[LPO];
    unsigned apix[LPO];                                 // (SKIPK) pixel index of the row: the skip source's address is formed at issue
    const half_t* bp[LPB];
#pragma unroll
    for (int i = 0; i < LPO; ++i) {
        const int r = lw * (BM / NL) + i * RPI + lrow;
        const int c = lpos ^ sk_swz(r);
        const int m = m0 + r;
        const bool inm = m < M;
        const unsigned mm = inm ? (unsigned)m : 0u;
        const unsigned img = mm / (unsigned)HWp, rem = mm - img * (unsigned)HWp;
        const int y = (int)(rem / (unsigned)W), x = (int)(rem - (rem / (unsigned)W) * (unsigned)W);
        abase[i] = X + (size_t)mm * Cin1 + c * 8;
        abase2[i] = (TAPS == 1 && X2 != nullptr) ? X2 + (size_t)mm * (Cin - Cin1) + c * 8 : nullptr;
        unsigned msk = 0;
        if (TAPS == 1) msk = inm ? 1u : 0u;
        else {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                if (inm && yy >= 0 && yy < H && xx >= 0 && xx < W) msk |= 1u << t;
            }
            if (SKIPK && inm) msk |= 1u << 9;
        }
        amask[i] = msk;
        apix[i] = mm;
    }
#pragma unroll
    for (int i = 0; i < LPB; ++i) {
        const int r = lw * (BN / NL) + i * RPI + lrow;
        bp[i] = Wt + (size_t)(n0 + r) * K + (lpos ^ sk_swz(r)) * 8;
    }
    const int it0 = (int)((long long)KI * split / splits), it1 = (int)((long long)KI * (split + 1) / splits);
    char* const ring = smem + grp * RING_BYTES;
    char* const wave_dst_a = ring + lw * ((BM / NL) * ROWB);
    char* const wave_dst_b = ring + A_BYTES + lw * ((BN / NL) * ROWB);
    // (tap, channel chunk) of the group's next K-step to issue: scalars advanced by KG steps per issue (no division in the loop)
    int is_tap = 0, is_nc = 0;
    auto issue = [&](int stage, int it) {                 // all pieces of K-step `it` (wave-uniform) into `stage` of this group's ring
        const int tap = is_tap, nc = is_nc;
#ifndef PD_LAB_SK_TAPMAJOR
        // K order (round 5): chunk-major -- the nine taps of a 64-channel chunk back to back, then the next chunk (weights stay
        // [Cout_pad][tap * Cin + c]: only the order in which the K-steps are walked changes).  Consecutive steps then re-read the same
        // activation lines shifted by one pixel / one row; tap-major walked all Cin / 64 chunks of a tap first, 64 KB of other lines
        // between two uses of a line.  1-8 % on every 3x3 layer (profiles/r05_sk_order.txt); the f32 summation order is a fixed
        // function of the layer either way.
        if (TAPS == 1 || is_tap >= XT) is_nc += KG;
        else { is_tap += KG; while (is_tap >= XT && is_nc + 1 < kc) { is_tap -= XT; ++is_nc; } if (is_tap >= XT) { is_nc = is_tap - XT; is_tap = XT; } }
#else
        is_nc += KG;
        while (TAPS != 1 && is_tap < XT && is_nc >= kc) { is_nc -= kc; ++is_tap; }
#endif
        const int dy = (TAPS == 1) ? 0 : tap / 3 - 1, dx = (TAPS == 1) ? 0 : tap - (tap / 3) * 3 - 1;
        const bool second = TAPS == 1 && X2 != nullptr && nc * 64 >= Cin1;
        const long long koff = second ? (long long)nc * 64 - Cin1 : ((long long)dy * W + dx) * Cin1 + (long long)nc * 64;
        const bool skp = SKIPK && tap == 9;               // (wave-uniform) the appended 1x1 over the block input
        const bool skp2 = skp && nc * 64 >= Cs1;
#ifndef PD_LAB_SK_NOA
#pragma unroll
        for (int p = 0; p < LPO; ++p) {
            const half_t* src = (second ? abase2[p] : abase[p]) + koff;
            if (skp) {
                const int cc = (lpos ^ sk_swz(lw * (BM / NL) + p * RPI + lrow)) * 8;
                src = skp2 ? XS2 + (size_t)apix[p] * (unsigned)(Cskip - Cs1) + (nc * 64 - Cs1) + cc : XS + (size_t)apix[p] * (unsigned)Cs1 + nc * 64 + cc;
            }
            if (((amask[p] >> tap) & 1u) == 0u) src = zero_page + ((lpos ^ sk_swz(lw * (BM / NL) + p * RPI + lrow)) * 8 & 63);
            sk_glds16(src, wave_dst_a + stage * STAGE_BYTES + p * 1024);
        }
#endif
#ifndef PD_LAB_SK_NOB
#ifdef PD_LAB_SK_ROT                                        // (lab: every workgroup walks the weight rows' K range from another start)
        const int itb = (it + wid * PD_LAB_SK_ROT) % KI;
#else
#ifndef PD_LAB_SK_TAPMAJOR
        const int itb = (TAPS == 1 || tap >= XT) ? it : tap * kc + nc;
#else
        const int itb = it;
#endif
#endif
#pragma unroll
        for (int p = 0; p < LPB; ++p) sk_glds16<PD_SK_B_AUX>(bp[p] + (size_t)itb * 64, wave_dst_b + stage * STAGE_BYTES + p * 1024);
#endif
    };

    // ---- consumer role
    int frag_off[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) frag_off[kk] = (lane & 15) * ROWB + ((((lane >> 4) + 4 * kk) ^ sk_swz(lane & 15)) << 4);
    float4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    // this group's K-steps: g0, g0 + KG, ...; `mine` of them.  Every group runs `iters` loop trips (the barrier counts must match).
    const int g0 = it0 + grp;
#ifndef PD_LAB_SK_TAPMAJOR
    if (TAPS == 1) is_nc = g0; else if (g0 >= XT * kc) { is_tap = XT; is_nc = g0 - XT * kc; } else { is_nc = g0 / XT; is_tap = g0 - is_nc * XT; }
#else
    if (TAPS == 1) is_nc = g0; else { is_tap = min(g0 / kc, XT); is_nc = g0 - is_tap * kc; }
#endif
    const int mine = g0 < it1 ? (it1 - g0 + KG - 1) / KG : 0;
    const int iters = (it1 - it0 + KG - 1) / KG;
    if (loader) {
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (d < mine) issue(d, g0 + d * KG);
    }
    int cur = 0, nxt = D % NST;
#ifdef PD_LAB_SK_STAMP
    unsigned long long lab_w = 0, lab_b = 0, lab_i = 0, lab_c = 0, lab_t, lab_u, lab_f[4] = {0, 0, 0, 0};
    const unsigned long long lab_start = SK_CLK();
#endif
    for (int k = 0; k < iters; ++k) {
#ifdef PD_LAB_SK_STAMP
        lab_t = SK_CLK();
#endif
        const int infl = min(D, mine - k);                 // this group's stages in flight, the current one included (<= 0: idle trip)
        if (D >= 7 && infl >= 7) sk_wait_vm<(D >= 7 ? 6 : 0) * OPS>();
        else if (D >= 6 && infl >= 6) sk_wait_vm<(D >= 6 ? 5 : 0) * OPS>();
        else if (D >= 5 && infl >= 5) sk_wait_vm<(D >= 5 ? 4 : 0) * OPS>();
        else if (D >= 4 && infl >= 4) sk_wait_vm<(D >= 4 ? 3 : 0) * OPS>();
        else if (D >= 3 && infl >= 3) sk_wait_vm<2 * OPS>();
        else if (D >= 2 && infl >= 2) sk_wait_vm<OPS>();
        else sk_wait_vm<0>();
#ifdef PD_LAB_SK_STAMP
        lab_u = SK_CLK(); lab_w += lab_u - lab_t; lab_t = lab_u;
#endif
#ifndef PD_LAB_SK_NOBARRIER
        __builtin_amdgcn_s_barrier();                      // stage `cur` landed for every wave of the group; stage `nxt` is free again
#endif
        asm volatile("" ::: "memory");
#ifdef PD_LAB_SK_STAMP
        lab_u = SK_CLK(); lab_b += lab_u - lab_t; lab_t = lab_u;
#endif
        // The groups of a SIMD's waves run in OPPOSITE phases: even groups issue their LDS-DMA pieces (the wave's issue slot is
        // blocked ~80 cycles per piece, the matrix pipe idles) while odd groups run their MFMAs, then the roles swap -- all groups
        // doing the same thing behind the common barrier just queue at the LDS-DMA port and then at the matrix pipe.
        // (measured, tools/bench_sk.py: worth 3-7 % on the 128x64 ... 64x32 tiles; the 128x128 tile is 7 % faster in lock-step)
        const bool issue_first = KG == 1 || (grp & 1) == 0 || (BM == 128 && BN == 128);
        if (loader && issue_first && k + D < mine) issue(nxt, g0 + (k + D) * KG);
#ifdef PD_LAB_SK_STAMP
        lab_u = SK_CLK(); lab_i += lab_u - lab_t; lab_t = lab_u;
#endif
#ifdef PD_LAB_SK_NOCOMPUTE
        if (false) {                                       // (lab: the loop without its fragment reads and MFMAs -- the fill alone)
#else
        if (consumer && k < mine) {
#endif
            // Fragment reads by inline asm with counted lgkmcnt waits in front of each MFMA group (hipcc waits lgkmcnt(0) before every
            // batch of MFMAs; LDS returns in issue order).  Round 5: only k-half 0's TM + TN reads go out before the first MFMA; the reads
            // of k-half 1 follow the groups of k-half 0 in equal shares.  Issued all at once (round 3) the wave sat ~210 cycles in the
            // ISSUE of 16 ds_read_b128 -- four waves share a port that serves one read per 4 cycles -- before its first wait could even
            // execute (fine stamps, profiles/r05_sk_loop_stamps.txt: first group at cycle 220, MFMAs back to back from there).
            const uint32_t a_ad = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(ring + cur * STAGE_BYTES + (wm * (BM / 2)) * ROWB);
            const uint32_t b_ad = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(ring + cur * STAGE_BYTES + A_BYTES + (wn * (BN / 2)) * ROWB);
            half8 fa[2][TM], fb[2][TN];
            constexpr int RPK = TM + TN;
#ifdef PD_LAB_SK_STAMP
            // fine stamps: s_memtime WITHOUT a wait (a waited stamp is an lgkmcnt(0) in the middle of the counted waits; an outstanding
            // one can only make a counted wait conservative), read behind the phase's final lgkmcnt(0)
            unsigned long long f_a, f_b, f_c = 0, f_d = 0, f_e = 0;
            asm volatile("s_memtime %0" : "=s"(f_a));
#endif
#define SK_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
            const uint32_t ba0 = b_ad + frag_off[0], aa0 = a_ad + frag_off[0], ba1 = b_ad + frag_off[1], aa1 = a_ad + frag_off[1];
#pragma unroll
            for (int j = 0; j < TN; ++j) SK_DSR(fb[0][j], ba0, j * 16 * ROWB);
#pragma unroll
            for (int i = 0; i < TM; ++i) SK_DSR(fa[0][i], aa0, i * 16 * ROWB);
#ifdef PD_LAB_SK_READS_UPFRONT
#pragma unroll
            for (int j = 0; j < TN; ++j) SK_DSR(fb[1][j], ba1, j * 16 * ROWB);
#pragma unroll
            for (int i = 0; i < TM; ++i) SK_DSR(fa[1][i], aa1, i * 16 * ROWB);
#endif
#ifdef PD_LAB_SK_STAMP
            asm volatile("s_memtime %0" : "=s"(f_b));
#endif
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#ifdef PD_LAB_SK_READS_UPFRONT
                    const int q_lo = RPK, q_hi = RPK;
#else
                    const int q_lo = (i * RPK) / TM, q_hi = ((i + 1) * RPK) / TM;      // k-half 1 reads issued behind group (0, i)
#endif
                    // outstanding reads this group may leave in flight: everything issued behind its own A fragment
                    const int allowed = kk == 0 ? RPK + q_lo - (TN + i + 1) : TM - 1 - i;
                    sk_wait_lgkm_tied<TN>(allowed, fa[kk][i], fb[kk]);
#ifdef PD_LAB_SK_STAMP
                    if (kk == 0 && i == 0) asm volatile("s_memtime %0" : "=s"(f_c));
                    if (kk == 1 && i == 0) asm volatile("s_memtime %0" : "=s"(f_d));
                    if (kk == 1 && i == TM - 1) asm volatile("s_memtime %0" : "=s"(f_e));
#endif
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[kk][j], fa[kk][i], acc[i][j], 0, 0, 0);
                    if (kk == 0) {
#pragma unroll
                        for (int q = 0; q < RPK; ++q) {
                            if (q < q_lo || q >= q_hi) continue;
                            if (q < TN) SK_DSR(fb[1][q < TN ? q : 0], ba1, (q < TN ? q : 0) * 16 * ROWB);
                            else SK_DSR(fa[1][q >= TN ? q - TN : 0], aa1, (q >= TN ? q - TN : 0) * 16 * ROWB);
                        }
                    }
                }
            }
#undef SK_DSR
#ifdef PD_LAB_SK_STAMP
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(f_a), "+s"(f_b), "+s"(f_c), "+s"(f_d), "+s"(f_e));   // (the stamps have landed; tied: no use above this line)
            lab_f[0] += f_b - f_a; lab_f[1] += f_c - f_a; lab_f[2] += f_d - f_a; lab_f[3] += f_e - f_a;
#endif
        }
#ifdef PD_LAB_SK_STAMP
        asm volatile("" : "+v"(acc[0][0]));
        lab_u = SK_CLK(); lab_c += lab_u - lab_t;
#endif
        if (loader && !issue_first && k + D < mine) issue(nxt, g0 + (k + D) * KG);
        cur = (cur + 1 == NST) ? 0 : cur + 1;
        nxt = (nxt + 1 == NST) ? 0 : nxt + 1;
    }
#ifdef PD_LAB_SK_STAMP
    if (lane == 0 && blockIdx.x == 8 && wave < 8) {
        unsigned long long* o = g_sk_stamps + wave * 16;
        o[0] = lab_w; o[1] = lab_b; o[2] = lab_i; o[3] = lab_c; o[4] = SK_CLK() - lab_start; o[5] = iters; o[6] = lab_start;
        o[8] = lab_f[0]; o[9] = lab_f[1]; o[10] = lab_f[2]; o[11] = lab_f[3];
    }
#endif
    __syncthreads();                                       // all fragment reads done before the stages are reused

    // ---- the K-groups' accumulators: groups 1 .. KG-1 park theirs in LDS (register order, 16 bytes per lane), group 0 adds them in
    // group order
    if (KG > 1) {
        constexpr int F = TM * TN;
        float4_t* R = reinterpret_cast<float4_t*>(smem);
        if (grp > 0) {
#pragma unroll
            for (int f = 0; f < F; ++f) R[((grp - 1) * F + f) * 256 + t4] = acc[f / TN][f % TN];
        }
        __syncthreads();
        if (grp == 0) {
#pragma unroll
            for (int g = 1; g < KG; ++g)
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    const float4_t v = R[((g - 1) * F + f) * 256 + t4];
                    float4_t& a = acc[f / TN][f % TN];
                    a[0] += v[0]; a[1] += v[1]; a[2] += v[2]; a[3] += v[3];
                }
        }
        __syncthreads();
    }

    // ---- in-launch split-K combine (group 0 holds the workgroup's slice)
    if (splits > 1) {
        constexpr int F = TM * TN;                         // float4 pieces per thread
        constexpr int SLAB_BYTES = BM * BN * 4;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(slabs + (size_t)tile * splits * (BM * BN), 0, splits * SLAB_BYTES, 0x00020000);
        if (grp == 0 && consumer) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rs,
                                                           split * SLAB_BYTES + ((i * TN + j) * 256 + t4) * 16, 0, /*sc1: write-through*/ 16);
        }
        // hand-off form R1 of the cdna guide (section 6, guideline 16; microarch "valid forms"): the payload is stored WRITE-THROUGH (sc1:
        // it is in memory, not in this XCD's L2, once the store has completed), every storing wave drains with an asm vmcnt(0) the
        // compiler cannot drop, the workgroup joins, and only then one lane takes the ticket with a relaxed agent-scope atomic; the last
        // arriver reads the slabs with sc1 loads (L1 bypass).  No release / acquire fence: a buffer_wbl2 / buffer_inv pair per workgroup
        // measured 3.9x slower in the guide's own table and orders nothing the write-through + drain does not.  Contract (pdhip.h,
        // pdhip_unet_forward): ONE forward in flight per handle -- tickets and slabs live in the handle's workspace.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains ...
        __syncthreads();                                   // ... before ONE lane takes the ticket
        volatile int* flag = reinterpret_cast<volatile int*>(smem + FLAG_OFF);
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = old == (unsigned)(splits - 1);
            if (last) __hip_atomic_store(tickets + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
            *flag = last;
        }
        __syncthreads();
        if (*flag == 0) return;
        // the last arriver: sum the slices in slice order (its own included -- the order must not depend on who is last);
        // U slices' loads in flight at a time, clamped index + conditional add (no branch around a load: cdna guide trap 4c)
        if (grp == 0 && consumer) {
            constexpr int U = 16 / F >= 1 ? 16 / F : 1;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};
            for (int sp0 = 0; sp0 < splits; sp0 += U) {
                float4_t v[U][F];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int sp = min(sp0 + u, splits - 1);
#pragma unroll
                    for (int f = 0; f < F; ++f)
                        v[u][f] = __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, sp * SLAB_BYTES + (f * 256 + t4) * 16, 0, /*sc1*/ 16));
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
#pragma unroll
                    for (int f = 0; f < F; ++f) {
                        float4_t& a = acc[f / TN][f % TN];
                        if (sp0 + u < splits) { a[0] += v[u][f][0]; a[1] += v[u][f][1]; a[2] += v[u][f][2]; a[3] += v[u][f][3]; }
                    }
                }
            }
        }
    }

    // ---- epilogue: acc (+bias) -> f16 -> LDS [BM][CS_LD] -> coalesced 16-byte rows (+residual) + GroupNorm octet partials.
    // Transposed accumulators (weights x activations): lane holds pixel 16 i + (lane & 15), channels 16 j + 4 (lane >> 4) + 0..3
    half_t* Cs = reinterpret_cast<half_t*>(smem);
    if (grp == 0 && consumer) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nl = wn * (BN / 2) + j * 16 + (lane >> 4) * 4;
            float4_t bv = (float4_t){0.f, 0.f, 0.f, 0.f};
            if (bias != nullptr && n0 + nl < Cout) bv = *reinterpret_cast<const float4_t*>(bias + n0 + nl);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int ml = wm * (BM / 2) + i * 16 + (lane & 15);
                half4 h;
#pragma unroll
                for (int r = 0; r < 4; ++r) h[r] = (half_t)(acc[i][j][r] + bv[r]);
                *reinterpret_cast<half4*>(&Cs[ml * CS_LD + nl]) = h;
            }
        }
    }
    __syncthreads();
    constexpr int CT = BN / 8;                             // column threads (one channel octet each)
    constexpr int RPP = NT / CT;                           // rows per pass
    constexpr int PASSES = (BM + RPP - 1) / RPP;
    const int col8 = (tid % CT) * 8;
    // GroupNorm octet partials per image segment of the tile: a 128-row tile of an 8x8 level covers TWO images (host: H * W % BM == 0
    // or BM % (H * W) == 0 with at most two images per tile)
    const int seg_rows = min(BM, HWp);
    float gs[2] = {0.f, 0.f}, gq[2] = {0.f, 0.f};
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int row = p * RPP + tid / CT;
        const int m = m0 + row;
        if (row < BM && m < M && n0 + col8 < Cout) {
            half8 v = *reinterpret_cast<const half8*>(&Cs[row * CS_LD + col8]);
            const size_t o = (size_t)m * Cout + n0 + col8;
            if (residual != nullptr) {
                // res_up: the residual is the half-resolution tensor [N, H/2, W/2, Cout] read with nearest x2 (the x branch of an up-ResBlock,
                // unet.py:190-195: no k_resample pass, no up-sampled copy)
                size_t ro = o;
                if (res_up) {
                    const int img = m / HWp, rem = m - img * HWp, yy = rem / W, xx = rem - yy * W;
                    ro = ((size_t)(img * (H >> 1) + (yy >> 1)) * (W >> 1) + (xx >> 1)) * Cout + n0 + col8;
                }
                const half8 rv = *reinterpret_cast<const half8*>(residual + ro);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] + (float)rv[e]);
            }
#ifdef PD_LAB_SK_NTSTORE                                    // (lab: streaming output stores -- does a clean L2 shorten the kernel boundary?)
            __builtin_nontemporal_store(v, reinterpret_cast<half8*>(Y + o));
#else
            *reinterpret_cast<half8*>(Y + o) = v;
#endif
            float s8 = 0.f, q8 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; s8 += f; q8 += f * f; }
            const bool hi = row >= seg_rows;
            gs[0] += hi ? 0.f : s8; gq[0] += hi ? 0.f : q8;
            gs[1] += hi ? s8 : 0.f; gq[1] += hi ? q8 : 0.f;
        }
    }
    if (gn_part != nullptr) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);       // [segment][thread][2]
        red[tid * 2] = gs[0]; red[tid * 2 + 1] = gq[0];
        red[(NT + tid) * 2] = gs[1]; red[(NT + tid) * 2 + 1] = gq[1];
        __syncthreads();
        // two-level, fixed-order column sums (round 5).  One thread per (segment, octet) summing all RPP rows was a serial chain that
        // hipcc unrolled into RPP x 2 loads in flight -- 256 VGPRs and a spill on the 64x32 tiles, ~2 000 cycles at the tail of every
        // launch; now 8 threads per (segment, octet) take every 8th row each, one thread adds the 8 sub-sums in order.
        const int nseg = BM / seg_rows;
        constexpr int SUB = 8;
        static_assert(RPP % SUB == 0 && (2 * NT + 2 * 2 * CT * SUB) * 8 <= (LS ? 1 : KG) * RING_BYTES, "GroupNorm partial reduce staging");
        float* red2 = red + 2 * NT * 2;                    // [segment][octet][SUB][2]
        if (tid < CT * nseg * SUB) {
            const int j = tid % SUB, c = (tid / SUB) % CT, seg = tid / (SUB * CT);
            float s1 = 0.f, q1 = 0.f;
#pragma unroll
            for (int r = 0; r < RPP / SUB; ++r) { s1 += red[(seg * NT + (r * SUB + j) * CT + c) * 2]; q1 += red[(seg * NT + (r * SUB + j) * CT + c) * 2 + 1]; }
            red2[tid * 2] = s1; red2[tid * 2 + 1] = q1;
        }
        __syncthreads();
        if (tid < CT * nseg) {
            const int seg = tid / CT, c = tid - seg * CT;
            const int img = m0 / HWp + seg;
            if (n0 + c * 8 < Cout && img < N) {
                float s1 = 0.f, q1 = 0.f;
#pragma unroll
                for (int j = 0; j < SUB; ++j) { s1 += red2[(tid * SUB + j) * 2]; q1 += red2[(tid * SUB + j) * 2 + 1]; }
                const int chunks = HWp >= BM ? HWp / BM : 1;
                const int chunk = HWp >= BM ? (m0 - (m0 / HWp) * HWp) / BM : 0;
                float* dst = gn_part + (((size_t)img * chunks + chunk) * (Cout >> 3) + (n0 >> 3) + c) * 2;
                dst[0] = s1; dst[1] = q1;
            }
        }
    }
}

template <int TAPS, int BM, int BN, int NST, int KG, bool LS = false, int NL = 4>
int launch_sk(int grid, hipStream_t s, const half_t* X, const half_t* Wt, const float* bias, const half_t* residual, half_t* Y, int N, int H,
              int W, int Cin, int Cout, int n_tiles, int total, const half_t* zero_page, int splits, float* slabs, unsigned* tickets,
              float* gnp, const half_t* X2, int Cin1, int m_fast, const half_t* XS = nullptr, const half_t* XS2 = nullptr, int Cs1 = 0, int Cs = 0, int res_up = 0) {
    auto kern = k_conv_sk<TAPS, BM, BN, NST, KG, LS, NL>;
    constexpr size_t smem = (size_t)KG * NST * (BM + BN) * 128 + 16;
    static_assert(smem <= 160 * 1024, "LDS budget");
    if (smem > 65536) PD_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    kern<<<grid, LS ? (4 + NL) * 64 : KG * 256, smem, s>>>(X, Wt, bias, residual, Y, N, H, W, Cin, Cout, n_tiles, total, zero_page, splits, slabs, tickets, gnp, X2, Cin1, m_fast, XS, XS2, Cs1, Cs, res_up);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}

}  // namespace

#ifdef PD_LAB_SK_STAMP
extern "C" int pdhip_lab_sk_read_stamps(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_sk_stamps), sizeof(unsigned long long) * n);
}
#endif
thread_local int g_sk_mode = 1;        // tuning / test hook: 0 = never route to k_conv_sk, 1 = automatic, 2 = every eligible layer
thread_local int g_sk_tile = 0;        // tuning hook: 0 = automatic, 1 = 128x128, 2 = 128x64, 3 = 64x64, 4 = 64x32
thread_local int g_sk_splits = 0;      // tuning hook: >= 1 forces the split factor
thread_local int g_sk_stages = 0;      // lab hook: LDS stages per K-group (2, 3, 4 where the tile allows); 0 = default
thread_local int g_sk_kg = 0;          // lab hook: K-groups per workgroup (1, 2, 4; 8 = loader-specialised); 0 = default
thread_local int g_sk_order = 0;       // lab hook: tile order inside an XCD's run: 0 automatic, 1 pixel tiles fastest, 2 n-tiles fastest

// the plan for one layer; bm == 0: not a layer for this kernel.  Rules read off tools/bench_sk.py tables (profiles/r03_sk_bench.txt):
//   * a tile shape whose tiles alone fill the chip (>= 224) runs unsplit -- the largest such shape (fewest L2 -> LDS bytes);
//   * otherwise 3x3 layers are split along K to ~256 workgroups: the largest tile that needs <= 4 slices, else the 64x32 tile with
//     up to 8 (a tile's slabs are re-read by ONE workgroup: 16 x 8 KB slices cost more than the K loop they shorten);
//   * 1x1 layers (K loops of 4-32 steps) never split: the combine costs more than their whole loop.
SkPlan conv_sk_plan(int N, int H, int W, int Cin, int Cout, int Cout_pad, int taps, bool two_source, size_t ws_floats, int k_extra) {
    SkPlan p{0, 0, 0, 0};
    if (g_sk_mode == 0 || Cin % 64 != 0 || Cout % 8 != 0 || Cout_pad % 128 != 0 || k_extra % 64 != 0) return p;
    const long long M = (long long)N * H * W, hw = (long long)H * W;
    if (M >= (1LL << 31) / 2) return p;
    const int KI = taps * (Cin / 64) + k_extra / 64;      // (k_extra: channels of a skip 1x1 appended to the K loop, conv_sk_skip)
    static const int BMs[4] = {128, 128, 64, 64}, BNs[4] = {128, 64, 64, 32};
    long long tiles[4];
    bool ok[4];
    for (int t = 0; t < 4; ++t) {
        tiles[t] = ((M + BMs[t] - 1) / BMs[t]) * (Cout_pad / BNs[t]);
        ok[t] = hw % BMs[t] == 0 || BMs[t] == 2 * hw;      // a tile is part of one image, or exactly two whole images (fused GroupNorm partials)
    }
    int pick = -1, ps = 1;
    if (g_sk_tile != 0) {
        pick = g_sk_tile - 1;
        if (pick < 0 || pick > 3 || !ok[pick]) return p;
        ps = g_sk_splits >= 1 ? g_sk_splits : 1;
    } else {
        if (g_sk_mode != 2 && ok[0] && tiles[0] > 640) return p;   // enough 128x128 tiles: the big-tile kernels own that regime
        // (long K loops amortise a two-slice combine: 128 tiles of 128x128 x 2 slices beat 256 unsplit 128x64 tiles at K = 9216 --
        // 50 vs 55 us at 16^2 batch 8 -- and lose at K = 4608, 31 vs 29 us at 64^2 batch 1)
        if (taps == 9 && KI >= 128 && ok[0] && tiles[0] >= 112 && tiles[0] < 224) { pick = 0; ps = 2; }
        // (a weight-heavy layer re-reads its 19-38 MB of weights once per PIXEL tile: 64-row tiles double that traffic -- 39 us unsplit
        // on 64x32 tiles against 24 us on 128x64 tiles x 4 slices at the 8^2 level, batch 8)
        const bool heavy = (size_t)Cout_pad * ((size_t)taps * Cin + k_extra) * 2 >= ((size_t)8 << 20) && M >= 256;
        for (int t = 0; t < 4 && pick < 0; ++t)
            if (ok[t] && tiles[t] >= 224 && !(heavy && BMs[t] == 64 && (ok[0] || ok[1]))) { pick = t; ps = 1; }
        if (pick < 0 && taps == 9) {
            for (int t = 0; t < 4 && pick < 0; ++t) {
                const int s = (int)((256 + tiles[t] / 2) / std::max<long long>(tiles[t], 1));
                if (ok[t] && s <= 4) { pick = t; ps = std::max(s, 1); }
            }
            if (pick < 0 && ok[3]) { pick = 3; ps = (int)std::min<long long>(8, (256 + tiles[3] / 2) / std::max<long long>(tiles[3], 1)); }
        }
        if (pick < 0) {                                    // 1x1: unsplit, the smallest tile the images allow
            for (int t = 3; t >= 0 && pick < 0; --t) if (ok[t]) pick = t;
            ps = 1;
        }
        if (pick < 0) return p;
        if (g_sk_splits >= 1) ps = g_sk_splits;
    }
    ps = std::max(1, std::min(std::min(ps, std::max(KI / 2, 1)), 64));
    if (ws_floats == 0 || tiles[pick] > 4096) ps = 1;
    while (ps > 1 && (size_t)tiles[pick] * ps * BMs[pick] * BNs[pick] + PD_SK_TICKET_FLOATS > ws_floats) --ps;
    (void)two_source;
    p.bm = BMs[pick]; p.bn = BNs[pick]; p.splits = ps; p.tile_id = pick + 1;
    return p;
}

int conv_sk(const SkPlan& pl, const half_t* X, const half_t* Wt, const float* bias, const half_t* residual, half_t* Y, int N, int H, int W,
            int Cin, int Cout, int Cout_pad, int taps, const half_t* zero_page, hipStream_t s, float* ws, size_t ws_floats, float* gn_part,
            int* gn_fused, const half_t* X2, int Cin1, int res_up) {
    PD_REQUIRE(pl.bm > 0 && (taps == 1 || taps == 9), "conv_sk: no plan / bad taps");
    PD_REQUIRE(res_up == 0 || (residual != nullptr && H % 2 == 0 && W % 2 == 0), "conv_sk: an up-sampled residual needs even H, W");
    if (X2 == nullptr) Cin1 = Cin;
    PD_REQUIRE(X2 == nullptr || (taps == 1 && Cin1 % 64 == 0 && (Cin - Cin1) % 64 == 0 && Cin1 > 0 && Cin1 < Cin), "conv_sk: bad two-source split");
    const long long M = (long long)N * H * W;
    const int m_tiles = (int)((M + pl.bm - 1) / pl.bm), n_tiles = Cout_pad / pl.bn, total = m_tiles * n_tiles;
    PD_REQUIRE(pl.splits == 1 || (ws != nullptr && (size_t)total * pl.splits * pl.bm * pl.bn + PD_SK_TICKET_FLOATS <= ws_floats && total <= 4096),
               "conv_sk: split-K workspace too small");
    // workspace: [0, 4096) ticket words (zeroed at allocation, self-resetting), then the slabs
    unsigned* tickets = reinterpret_cast<unsigned*>(ws);
    float* slabs = ws ? ws + PD_SK_TICKET_FLOATS : nullptr;
    if (gn_fused) *gn_fused = gn_part ? std::max((int)(((long long)H * W) / pl.bm), 1) : 0;
    const int grid = total * pl.splits;
    // (lab hook only: sharing the weight slices of the weight-heavy 16^2 / 8^2 layers through one L2 measured no better than the
    // default order -- the 256 MB Infinity Cache already absorbs the re-reads -- and 18 % worse at 512 tiles; profiles/r03_sk_bench.txt)
    const int m_fast = g_sk_order == 1 && m_tiles > 1 ? 1 : 0;
#define SK_ARGS grid, s, X, Wt, bias, residual, Y, N, H, W, Cin, Cout, n_tiles, total, zero_page, pl.splits, slabs, tickets, gn_part, X2, Cin1, m_fast, nullptr, nullptr, 0, 0, res_up
    // (lab hooks: K-groups g_sk_kg, stages g_sk_stages; 0 = the tile's default)
    int kg = g_sk_kg;
    const int st = g_sk_stages;
    // K-groups: two (four for long unsplit loops of the 64x32 tile) when the launch is ONE round of workgroups -- they are what puts a
    // second wave on every SIMD; with >= 1.5 rounds co-resident workgroups do that already and the lock-step of the groups only
    // costs (tools/bench_sk.py, 512 tiles of 128x128: 148 us with one group, 184 with two)
    // The 128x128 tile takes the loader-specialised form instead (8): 26.2 vs 26.8 us at 128^2 batch 1, 42.4 vs 44.3 at 32^2 batch 8.
    // (four K-groups exist for the 64x32 tile only: on the 64x64 tile the form needed 138 spilled VGPRs and was never routed)
    if (kg == 4 && pl.tile_id != 4) kg = 2;
    // (round 5: the 128x128 tile's loader-specialised form runs with EIGHT loader waves -- 12 waves, 133 VGPRs: 26.5 -> 25.7 us at 128^2
    // batch 1, 43.7 -> 42.2 at 32^2 batch 8, 51.6 -> 50.2 at 16^2 batch 8, profiles/r05_sk_probe.txt)
    if (kg == 0) kg = grid >= 384 ? 1 : pl.tile_id == 1 ? 12 : (pl.tile_id == 4 && taps * (Cin / 64) / pl.splits >= 32) ? 4 : 2;
#define SK_L(T, BM_, BN_, NST_, KG_) launch_sk<T, BM_, BN_, NST_, KG_>(SK_ARGS)
#define SK_LS(T, BM_, BN_, NST_) launch_sk<T, BM_, BN_, NST_, 1, true>(SK_ARGS)
#define SK_LS8(T, BM_, BN_, NST_) launch_sk<T, BM_, BN_, NST_, 1, true, 8>(SK_ARGS)
    if (kg == 12 && pl.tile_id > 2) kg = 8;                // (the 8-loader form exists for the 128-row tiles)
    if (kg == 12) {                                        // loader-specialised, EIGHT loader waves (hook value 12): 128x128 and 128x64 tiles
        if (taps == 9) return pl.tile_id == 1 ? (st == 2 ? SK_LS8(9, 128, 128, 2) : st == 4 ? SK_LS8(9, 128, 128, 4) : SK_LS8(9, 128, 128, 3))
                                              : (st == 3 ? SK_LS8(9, 128, 64, 3) : SK_LS8(9, 128, 64, 4));
        return pl.tile_id == 1 ? (st == 2 ? SK_LS8(1, 128, 128, 2) : st == 4 ? SK_LS8(1, 128, 128, 4) : SK_LS8(1, 128, 128, 3))
                               : (st == 3 ? SK_LS8(1, 128, 64, 3) : SK_LS8(1, 128, 64, 4));
    }
    if (kg == 8) {                                         // loader-specialised form (hook value 8)
        if (taps == 9) return pl.tile_id == 1 ? (st == 2 ? SK_LS(9, 128, 128, 2) : st == 4 ? SK_LS(9, 128, 128, 4) : SK_LS(9, 128, 128, 3))
                            : pl.tile_id == 2 ? (st == 2 ? SK_LS(9, 128, 64, 2) : st == 3 ? SK_LS(9, 128, 64, 3) : SK_LS(9, 128, 64, 4))
                            : pl.tile_id == 3 ? (st == 2 ? SK_LS(9, 64, 64, 2) : st == 6 ? SK_LS(9, 64, 64, 6) : SK_LS(9, 64, 64, 4))
                                              : (st == 2 ? SK_LS(9, 64, 32, 2) : st == 6 ? SK_LS(9, 64, 32, 6) : SK_LS(9, 64, 32, 4));
        return pl.tile_id == 1 ? (st == 2 ? SK_LS(1, 128, 128, 2) : st == 4 ? SK_LS(1, 128, 128, 4) : SK_LS(1, 128, 128, 3))
             : pl.tile_id == 2 ? (st == 2 ? SK_LS(1, 128, 64, 2) : st == 3 ? SK_LS(1, 128, 64, 3) : SK_LS(1, 128, 64, 4))
             : pl.tile_id == 3 ? (st == 2 ? SK_LS(1, 64, 64, 2) : st == 6 ? SK_LS(1, 64, 64, 6) : SK_LS(1, 64, 64, 4))
                               : (st == 2 ? SK_LS(1, 64, 32, 2) : st == 6 ? SK_LS(1, 64, 32, 6) : SK_LS(1, 64, 32, 4));
    }
#define SK_TILE(T)                                                                                                                   \
    (pl.tile_id == 1 ? (kg == 1 ? (st == 3 ? SK_L(T, 128, 128, 3, 1) : st == 4 ? SK_L(T, 128, 128, 4, 1) : SK_L(T, 128, 128, 2, 1)) : SK_L(T, 128, 128, 2, 2))        \
     : pl.tile_id == 2 ? (kg == 1 ? (st == 2 ? SK_L(T, 128, 64, 2, 1) : st == 4 ? SK_L(T, 128, 64, 4, 1) : SK_L(T, 128, 64, 3, 1))                                   \
                          : (st == 2 ? SK_L(T, 128, 64, 2, 2) : SK_L(T, 128, 64, 3, 2)))                                                                          \
     : pl.tile_id == 3 ? (kg == 1 ? (st == 2 ? SK_L(T, 64, 64, 2, 1) : SK_L(T, 64, 64, 4, 1))                                                                      \
                          : (st == 2 ? SK_L(T, 64, 64, 2, 2) : st == 3 ? SK_L(T, 64, 64, 3, 2) : SK_L(T, 64, 64, 4, 2)))                                             \
                       : (kg == 1 ? (st == 2 ? SK_L(T, 64, 32, 2, 1) : SK_L(T, 64, 32, 4, 1)) : kg == 4 ? (st == 2 ? SK_L(T, 64, 32, 2, 4) : SK_L(T, 64, 32, 3, 4))    \
                          : (st == 2 ? SK_L(T, 64, 32, 2, 2) : st == 3 ? SK_L(T, 64, 32, 3, 2) : SK_L(T, 64, 32, 4, 2))))
    return taps == 9 ? SK_TILE(9) : SK_TILE(1);
#undef SK_TILE
#undef SK_L
#undef SK_LS
#undef SK_LS8
#undef SK_ARGS
}

// 3x3 conv (Cin -> Cout) + the ResBlock's skip 1x1 (Cs -> Cout over xs = [XS | XS2]) as ONE K loop; Wt [Cout_pad][9 Cin + Cs], bias summed.
// The default tile configurations only (the lab hooks for stages / K-groups belong to conv_sk).
int conv_sk_skip(const SkPlan& pl, const half_t* X, const half_t* Wt, const float* bias, half_t* Y, int N, int H, int W, int Cin, int Cout,
                 int Cout_pad, const half_t* XS, const half_t* XS2, int Cs1, int Cs, const half_t* zero_page, hipStream_t s, float* ws,
                 size_t ws_floats, float* gn_part, int* gn_fused) {
    PD_REQUIRE(pl.bm > 0 && XS != nullptr && Cs > 0 && Cs % 64 == 0, "conv_sk_skip: no plan / bad skip source");
    if (XS2 == nullptr) Cs1 = Cs;
    PD_REQUIRE(XS2 == nullptr || (Cs1 % 64 == 0 && Cs1 > 0 && Cs1 < Cs), "conv_sk_skip: bad two-source split");
    const long long M = (long long)N * H * W;
    const int m_tiles = (int)((M + pl.bm - 1) / pl.bm), n_tiles = Cout_pad / pl.bn, total = m_tiles * n_tiles;
    PD_REQUIRE(pl.splits == 1 || (ws != nullptr && (size_t)total * pl.splits * pl.bm * pl.bn + PD_SK_TICKET_FLOATS <= ws_floats && total <= 4096),
               "conv_sk_skip: split-K workspace too small");
    unsigned* tickets = reinterpret_cast<unsigned*>(ws);
    float* slabs = ws ? ws + PD_SK_TICKET_FLOATS : nullptr;
    if (gn_fused) *gn_fused = gn_part ? std::max((int)(((long long)H * W) / pl.bm), 1) : 0;
    const int grid = total * pl.splits;
    const int kg = grid >= 384 ? 1 : pl.tile_id == 1 ? 8 : (pl.tile_id == 4 && (9 * (Cin / 64) + Cs / 64) / pl.splits >= 32) ? 4 : 2;
#define SKS_ARGS grid, s, X, Wt, bias, nullptr, Y, N, H, W, Cin, Cout, n_tiles, total, zero_page, pl.splits, slabs, tickets, gn_part, nullptr, Cin, 0, XS, XS2, Cs1, Cs
    if (pl.tile_id == 1) return kg == 8 ? launch_sk<10, 128, 128, 3, 1, true, 8>(SKS_ARGS) : kg == 1 ? launch_sk<10, 128, 128, 2, 1>(SKS_ARGS) : launch_sk<10, 128, 128, 2, 2>(SKS_ARGS);
    if (pl.tile_id == 2) return kg == 1 ? launch_sk<10, 128, 64, 3, 1>(SKS_ARGS) : launch_sk<10, 128, 64, 3, 2>(SKS_ARGS);
    if (pl.tile_id == 3) return kg == 1 ? launch_sk<10, 64, 64, 4, 1>(SKS_ARGS) : launch_sk<10, 64, 64, 4, 2>(SKS_ARGS);
    return kg == 1 ? launch_sk<10, 64, 32, 4, 1>(SKS_ARGS) : kg == 4 ? launch_sk<10, 64, 32, 3, 4>(SKS_ARGS) : launch_sk<10, 64, 32, 4, 2>(SKS_ARGS);
#undef SKS_ARGS
}

// fused weights of conv_sk_skip: dst [Cout_pad][K9 + Cs] = [w3 [Cout_pad][K9] | w1 [Cout_pad][Cs]], bias = b3 + b1
__global__ void k_fuse_skip_weights(const half_t* __restrict__ w3, int K9, const half_t* __restrict__ w1, int Cs, int rows, half_t* __restrict__ dst) {
    const int KT = (K9 + Cs) >> 3;
    const long long total = (long long)rows * KT;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / KT), k8 = (int)(i - (long long)r * KT) * 8;
        const half8 v = k8 < K9 ? *reinterpret_cast<const half8*>(w3 + (size_t)r * K9 + k8) : *reinterpret_cast<const half8*>(w1 + (size_t)r * Cs + (k8 - K9));
        *reinterpret_cast<half8*>(dst + (size_t)r * (K9 + Cs) + k8) = v;
    }
}
__global__ void k_add_bias(const float* __restrict__ a, const float* __restrict__ b, int n, float* __restrict__ o) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = a[i] + b[i];
}
int fuse_skip_weights(const half_t* w3, int K9, const half_t* w1, int Cs, int Cout_pad, const float* b3, const float* b1, int Cout, half_t* dst,
                      float* bdst, hipStream_t s) {
    PD_REQUIRE(K9 % 8 == 0 && Cs % 8 == 0, "fuse_skip_weights: K must be a multiple of 8");
    k_fuse_skip_weights<<<(int)std::min<long long>(((long long)Cout_pad * ((K9 + Cs) >> 3) + 255) / 256, 8192), 256, 0, s>>>(w3, K9, w1, Cs, Cout_pad, dst);
    k_add_bias<<<(Cout + 255) / 256, 256, 0, s>>>(b3, b1, Cout, bdst);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}

}  // namespace pdnn
