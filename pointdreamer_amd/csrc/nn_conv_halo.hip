// Row U1: 3x3 convolution with an LDS-resident activation halo -- the large-image layers of the UNet (W = 64 / 128 / 256).
// Same contract as k_conv_igemm (nn_gemm.hip) for taps == 9: Y[m, n] = sum_{tap, c} X[m + d(tap), c] * Wt[n, tap*Cin + c] + bias
// (+ residual), NHWC f16, f32 accumulation, fused GroupNorm octet partials.
//
// Why a second kernel: k_conv_igemm re-loads the activation tile once per filter tap, so every 64-deep K-step moves
// 32 KB (activations) + 32 KB (weights) L2 -> LDS for 8.4 MFLOP.  Measured on MI355X (tools/lab_build.sh, NODMA / NOA / NOB
// variants) that traffic is what bounds it: the same kernel without the LDS-DMA runs at 1.9 PFLOP/s, with half of it at
// 1.45-1.7, with all of it at 1.1-1.2.  Here the tile is 512 pixels (whole image rows) x 128 channels and the K loop is
// CHUNK-major: per 32 input channels the (rows + 2) x (W + 2) halo is staged ONCE and all nine taps read it at shifted
// addresses, so a 32-deep step moves 8 KB of weights + ~7 KB of halo: 2.1x less L2 -> LDS traffic per flop, and half the
// LDS-DMA instructions per MFMA.  Zero padding is resolved when the halo is staged (zero page) -- no per-tap predicates.
//
// LDS: 2 halo buffers (chunk c / c+1) + 3 weight stages of 8 KB.  8 waves as 4 (pixels) x 2 (channels), wave tile 128 x 64,
// the register-pipelined fragment schedule of nn_gemm.hip (inline-asm ds_read_b128, counted waits) with 8 groups per step.
// Halo row = 64 B (32 channels); 16-byte slot swizzle s(hp) = 2 * ((hp >> 2) & 1): conflict-free ds_read_b128 for ANY
// start pixel (the tap shift moves the 16-pixel fragment window by +-1), mirrored on the LDS-DMA source address.
#include <type_traits>
#include "nn_common.h"
using namespace pdhip;
namespace pdnn {
thread_local int g_halo_strips = 0;                                    // tuning / test hook (pdhip_debug_set_conv_halo_strips)
namespace {

typedef __attribute__((address_space(3))) void lds_void_h;
typedef const __attribute__((address_space(1))) void gbl_void_h;
__device__ __forceinline__ void glds16h(const void* gsrc, void* lds_wave_base) {
#ifndef PD_LAB_NODMA                                       // (lab builds only: mainloop without the L2 -> LDS traffic)
    __builtin_amdgcn_global_load_lds((gbl_void_h*)gsrc, (lds_void_h*)lds_wave_base, 16, 0, 0);
#endif
}
__device__ __forceinline__ int swz_b(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }   // weight tile: aligned 16-row windows
__device__ __forceinline__ int swz_a(int hp) { return ((hp >> 2) & 1) << 1; }                     // halo: any window start

// WLOG: log2 of the TILE width; ILOG >= WLOG: log2 of the image width.  ILOG > WLOG: the image is cut into column strips of the
// tile width (a 512-pixel tile = 8 rows x 64 columns instead of 2 rows x 256: 660 halo pixels per chunk instead of 1 032).
// APPLY: the conv's input is silu(A x + B) of X with one (A, B) per (image, channel) -- GroupNorm (+ FiLM) + SiLU folded by
// gn_table() -- applied IN LDS after the halo pieces have landed, while the MFMAs of the current chunk run: the eight pieces
// (128 halo pixels) staged in tap step T-1 are fetched behind the hand-over barrier of step T, transformed under groups 0-3 of
// step T+1 and stored under group 4.  Work split: wave (o, h) = (wave & 3, wave >> 2) owns channel octet o of pixel half h of
// every slot, one pixel per lane -- so the 16 constants of a chunk are WAVE-UNIFORM: they sit in lanes 0-15 of one VGPR
// (one ds_read_b32 per chunk) and reach the VALU through v_readlane, no per-lane constant registers (the kernel has none to
// spare).  The separate GroupNorm-apply pass over the tensor (read + write of every activation, 13 % of a DDNM step)
// disappears.  Zero padding belongs to the TRANSFORMED image: a validity byte per halo pixel (LDS, written once per tile)
// zeroes the out-of-image pixels again.  Duplicate tail pieces go to a scratch KiB per wave so that no raw copy can land on a
// transformed piece.
template <int WLOG, int ILOG, bool APPLY>
__global__ __launch_bounds__(512) void k_conv3x3_halo(const half_t* __restrict__ X, const half_t* __restrict__ Wt,
                                                      const float* __restrict__ bias, const half_t* __restrict__ residual,
                                                      half_t* __restrict__ Y, int N, int H, int Cin, int Cout, int n_tiles,
                                                      int total_tiles, const half_t* __restrict__ zero_page,
                                                      float* __restrict__ gn_part, int splits, float* __restrict__ partial,
                                                      const float* __restrict__ ap_table, int res_up, int in_up) {
    constexpr int W = 1 << WLOG, BMT = 512, BNT = 128, TM = 8, ROWB = 64, NWAVES = 8;
    constexpr int RT = BMT / W, HW2 = W + 2, HP = (RT + 2) * HW2;     // tile rows, halo row length, halo pixels
    constexpr int NPA = (HP + 15) / 16, PA = (NPA + 7) / 8;           // 1 KiB halo pieces per chunk, per wave
    constexpr int AH_BYTES = NPA * 1024, B_BYTES = BNT * ROWB;
    constexpr int RPW = W >= 128 ? 1 : 128 / W;                       // image rows inside one wave's 128 pixels
    constexpr int FPR = TM / RPW;                                     // A fragments per such row
    constexpr int CS_LD = BNT + 8;
    static_assert(PA <= 9 && RPW <= 4, "halo schedule: at most one halo piece per wave per tap, W >= 32");
    static_assert(!APPLY || PA <= 7, "APPLY: piece T-1 is finished under step T+1 <= 8");
    constexpr int SCR_OFF = 2 * AH_BYTES + 3 * B_BYTES;              // APPLY: 1 KiB per wave for the duplicate tail pieces
    constexpr int TAB_OFF = SCR_OFF + NWAVES * 1024;                 // APPLY: (A0..A7, B0..B7) per channel octet of this image
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform: keeps the piece / stage arithmetic on the scalar unit
    const int wm = wave >> 1, wn = wave & 1;
    int tile;
    {
        const int b = blockIdx.x, q = total_tiles >> 3, r = total_tiles & 7, xcd = b & 7, i = b >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
    }
    const int pt = tile / n_tiles, n0 = (tile % n_tiles) * BNT;      // pixel tile, output-channel tile
    const int HWp = H << ILOG;
    const int tpi = HWp / BMT;                                        // pixel tiles per image
    const int img = pt / tpi, tin = pt - img * tpi;                   // image, tile inside the image (= its GroupNorm chunk)
    constexpr int STRIPS = 1 << (ILOG - WLOG);
    const int ty0 = (tin / STRIPS) * RT, x0 = (tin % STRIPS) << WLOG;
    // pixel index (row of the [N*H*W, C] activation matrix) of tile pixel p
    auto pix = [&](int p) -> long long { return (((long long)img * H + ty0 + (p >> WLOG)) << ILOG) + x0 + (p & (W - 1)); };
    const int K = 9 * Cin;
    // split over the channel chunks (small-M layers): blockIdx.y owns chunks [cb, cb + NC) and writes an f32 partial tile
    const int NCT = Cin >> 5;
    const int cb = (int)((long long)NCT * blockIdx.y / splits), NC = (int)((long long)NCT * (blockIdx.y + 1) / splits) - cb;

    // ---- loader role.  Halo piece slot t of this wave = piece min(8 t + wave, NPA - 1) (the tail slots of the last waves
    // re-load the last piece: every wave issues the same number of LDS-DMA instructions, which keeps the vmcnt counts static).
    // The source address of a piece is recomputed when it is issued (~20 VALU ops behind 32 MFMAs) rather than kept in 9
    // registers: the kernel sits at the 256-VGPR limit of two waves per SIMD.
    // Byte offset of this lane's 16 bytes of halo piece slot t at chunk 0 (~0u = zero page: outside the image / past the halo).
    // Slots < PA_TAB keep it in a register (4 VALU ops to form the source pointer), the rest recompute it when issued (~20):
    // the kernel sits at the 256-VGPR limit of two waves per SIMD, and VALU work next to the MFMAs is not free here either.
    constexpr int PA_TAB = PA > 6 ? 6 : PA;
    const int cq = lane & 3;
    auto halo_off = [&](int t) -> uint32_t {
        const int pi = min(t * 8 + wave, NPA - 1);
        int lq = lane >> 2;
        asm volatile("" : "+v"(lq));                                  // recompute here, every time (no loop-invariant hoisting)
        const int hp = pi * 16 + lq;
        const int hy = hp / HW2, hx = hp - hy * HW2;
        const int y = ty0 - 1 + hy, x = x0 + hx - 1;
        const bool ok = (hp < HP) & (y >= 0) & (y < H) & (x >= 0) & (x < (1 << ILOG));
        // in_up (0 / 1): X is the HALF-resolution tensor and the conv's input is its nearest x2 up-sampling (the in_layers conv of an
        // up-ResBlock, unet.py:190-192, 237-238) -- index arithmetic here instead of a 4x larger tensor in HBM
        const int spix = ((img * (H >> in_up) + (y >> in_up)) << (ILOG - in_up)) + (x >> in_up);          // source pixel
        return ok ? (uint32_t)((spix * Cin + ((cq ^ swz_a(hp)) << 3)) * 2) : ~0u;   // < 2^32 (host check)
    };
    uint32_t aoff[PA_TAB];
#pragma unroll
    for (int t = 0; t < PA_TAB; ++t) aoff[t] = halo_off(t);
    const char* const Xb = reinterpret_cast<const char*>(X);
    auto halo_src = [&](int t, int chunk) -> const void* {
        uint32_t off = t < PA_TAB ? aoff[t < PA_TAB ? t : 0] : halo_off(t);
        if constexpr (APPLY) asm volatile("" : "+v"(off));           // form the 64-bit pointer when the piece is issued (no hoisted pairs)
        return off == ~0u ? (const void*)zero_page : (const void*)(Xb + off + (uint32_t)chunk * 64u);
    };
    const int brow = wave * 16 + (lane >> 2);
    const half_t* bp = Wt + (size_t)(n0 + brow) * K + cb * 32 + (((lane & 3) ^ swz_b(brow)) << 3);   // next weight slice to stage
    char* const ah_dst = smem;                                        // + buf * AH_BYTES + piece * 1024
    char* const b_dst = smem + 2 * AH_BYTES + wave * 1024;            // + stage * B_BYTES
    // LDS byte offset of halo piece slot t in buffer buf (APPLY: a duplicate tail slot lands in the wave's scratch KiB)
    auto piece_off = [&](int t, int buf) -> int {
        const int pi = t * 8 + wave;
        if (APPLY) return pi < NPA ? buf * AH_BYTES + pi * 1024 : SCR_OFF + wave * 1024;
        return buf * AH_BYTES + min(pi, NPA - 1) * 1024;
    };

    // ---- consumer role
    const int r16 = lane & 15, q4 = lane >> 4;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int p0 = wm * 128;
    const int hp0 = ((p0 >> WLOG) + 1) * HW2 + (p0 & (W - 1)) + 1 + r16;          // halo pixel of (fragment 0, tap centre)
    const uint32_t b_frag = lds0 + 2 * AH_BYTES + (wn * 64 + r16) * ROWB + ((q4 ^ swz_b(r16)) << 4);
    float4_t acc[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

#define HL_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
    // swizzled LDS address of this lane's halo pixel for tap (dy, dx) in halo buffer `buf` (+ second image row for W = 64)
#define HL_AADDR(buf, toff, row2)                                                                                    \
    (lds0 + (buf) * AH_BYTES + (uint32_t)((hpv + (toff) + (row2) * HW2) << 6) + (uint32_t)((q4 ^ swz_a(hpv + (toff) + (row2) * HW2)) << 4))
    half8 ar[4], bf[2][4];

    // ---- APPLY role: wave (o, h) transforms channel octet o of halo pixels t * 128 + h * 64 + lane of every slot t
    half8 ap_d;                                                       // the 8 channels being transformed
    int ap_k = 0;                                                     // lanes 0-7: A of the chunk's octet, lanes 8-15: B (f32 bits)
    uint32_t ap_lds = 0;                                              // where ap_d lives in LDS
    uint32_t ap_v = 0;                                                // validity byte of the lane's halo pixel
    const int ap_o = wave & 3, ap_h = wave >> 2;
    const uint32_t val_lds = lds0 + (uint32_t)TAB_OFF + (uint32_t)Cin * 8u;
    auto ap_fetch = [&](int t, int buf) {                             // issue the LDS reads of slot t (pixels beyond the last piece: scratch)
        int hp = ap_h * 64 + lane;
        asm volatile("" : "+v"(hp));                                  // (keeps the per-slot addresses out of loop-invariant registers)
        hp += t * 128;
        const bool real = hp < NPA * 16;
        ap_lds = real ? lds0 + (uint32_t)(buf * AH_BYTES) + ((uint32_t)hp << 6) + (uint32_t)((ap_o ^ swz_a(hp)) << 4)
                      : lds0 + (uint32_t)SCR_OFF + (uint32_t)(wave * 1024 + lane * 16);
#ifndef PD_LAB_AP_NOFETCH
        HL_DSR(ap_d, ap_lds, 0);
        asm volatile("ds_read_u8 %0, %1" : "=v"(ap_v) : "v"(val_lds + (uint32_t)(real ? hp : 0)));
#endif
    };
    auto ap_consts = [&](int chunk) {                                 // (A0..A7, B0..B7) of (chunk, octet o) -> lanes 0-15 of ap_k
        asm volatile("ds_read_b32 %0, %1" : "=v"(ap_k) : "v"(lds0 + (uint32_t)TAB_OFF + (uint32_t)(((chunk * 4 + ap_o) * 16 + (lane & 15)) << 2)));
    };
    // y = silu(A x + B) on channel pair j: f32 arithmetic, one rounding to f16
    auto ap_pair = [&](int j) {
#ifdef PD_LAB_AP_NOCOMPUTE                                  // (lab builds only: the transform's LDS traffic without its arithmetic)
        return;
#endif
        int kk = ap_k;
        asm volatile("" : "+v"(kk));                                  // read the lanes here (no 16 long-lived SGPRs)
#pragma unroll
        for (int e = 2 * j; e < 2 * j + 2; ++e) {
            const float A = __builtin_bit_cast(float, __builtin_amdgcn_readlane(kk, e));
            const float B = __builtin_bit_cast(float, __builtin_amdgcn_readlane(kk, 8 + e));
            float y = (float)ap_d[e] * A + B;
#ifndef PD_LAB_AP_NOTRANS                                   // (lab builds only: affine map without the SiLU's transcendentals)
            y = y * __builtin_amdgcn_rcpf(1.0f + __expf(-y));
#endif
            ap_d[e] = (half_t)y;
        }
    };
    auto ap_store = [&]() {                                           // zero padding belongs to the transformed image
        typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
        u32x4 v = __builtin_bit_cast(u32x4, ap_d);
        const bool ok = ap_v != 0;
        v[0] = ok ? v[0] : 0u; v[1] = ok ? v[1] : 0u; v[2] = ok ? v[2] : 0u; v[3] = ok ? v[3] : 0u;
#ifndef PD_LAB_AP_NOSTORE
        asm volatile("ds_write_b128 %0, %1" :: "v"(ap_lds), "v"(v) : "memory");
#endif
    };

    // ---- prologue: halo of chunk 0, weight slices 0 and 1
#pragma unroll
    for (int t = 0; t < PA; ++t) glds16h(halo_src(t, cb), ah_dst + piece_off(t, 0));
    glds16h(bp, b_dst);
    bp += Cin;
    glds16h(bp, b_dst + B_BYTES);
    bp += Cin;
    if (PA == 9) {                                                    // keeps the issue order of the steady state (see HL_VMN)
        glds16h(halo_src(PA - 1, cb), ah_dst + piece_off(PA - 1, 0));
#ifndef PD_LAB_NOPROWAIT                                    // (lab builds, WRONG results: the tile without its exposed first-chunk fill -- what a cross-tile prefetch could hide at most)
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
#endif
    } else {
#ifndef PD_LAB_NOPROWAIT
        asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
#endif
    }
    if constexpr (APPLY) {
        // the image's affine table and the validity bytes -> LDS (plain loads: the compiler drains every outstanding LDS-DMA
        // first, which the transform below needs anyway); after the barrier every piece of chunk 0 is visible to every wave
        const float4_t* tg = reinterpret_cast<const float4_t*>(ap_table) + (size_t)img * (Cin >> 1);
        float4_t* tl = reinterpret_cast<float4_t*>(smem + TAB_OFF);
        for (int i = tid; i < (Cin >> 1); i += NWAVES * 64) tl[i] = tg[i];
        uint8_t* vl = reinterpret_cast<uint8_t*>(smem + TAB_OFF) + (size_t)Cin * 8;
        for (int hp = tid; hp < NPA * 16; hp += NWAVES * 64) {
            const int hy = hp / HW2, hx = hp - hy * HW2;
            const int y = ty0 - 1 + hy, x = x0 + hx - 1;
            vl[hp] = ((hp < HP) & (y >= 0) & (y < H) & (x >= 0) & (x < (1 << ILOG))) ? 1 : 0;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        ap_consts(cb);
#pragma unroll
        for (int t = 0; t < PA; ++t) {
            ap_fetch(t, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ap_d), "+v"(ap_v), "+v"(ap_k));
            ap_pair(0); ap_pair(1); ap_pair(2); ap_pair(3);
            ap_store();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {
        const int hpv = hp0;
        const uint32_t a0 = HL_AADDR(0, -HW2 - 1, 0);
        HL_DSR(bf[0][0], b_frag, 0); HL_DSR(bf[0][1], b_frag, 16 * ROWB); HL_DSR(bf[0][2], b_frag, 32 * ROWB); HL_DSR(bf[0][3], b_frag, 48 * ROWB);
        HL_DSR(ar[0], a0, 0); HL_DSR(ar[1], a0, 1024);
        if constexpr (FPR > 2) { HL_DSR(ar[2], a0, 2048); } else { const uint32_t a1 = HL_AADDR(0, -HW2 - 1, 1); HL_DSR(ar[2], a1, 0); }
    }

    int bs = 0;                                                       // weight stage of the current step (s mod 3)
    uint32_t a_carry;                                                 // row-0 fragment base of the step about to start
    { const int hpv = hp0; a_carry = HL_AADDR(0, -HW2 - 1, 0); }
    const int S = 9 * NC;
    for (int c0 = 0; c0 < NC; c0 += 2) {                              // two chunks per trip: buffer / register-set parities static
        // One step = (chunk c, tap T): 8 groups of 4 MFMAs.  Reads: A fragment g+3 at group g (ring of 4), the next step's
        // four weight fragments + A fragments 0..2 from the hand-over on.  LDS-DMA: weight slice s+2 after group 0, halo
        // piece T of chunk c+1 after group 1.  Hand-over in front of group 5: counted vmcnt (weight slice s+1 landed -- issued
        // after it: the previous step's halo piece, this step's weight and halo pieces), drain of my LDS reads, barrier.
    // APPLY: one fewer -- the previous step's halo piece must have landed too, it is fetched right behind the hand-over
#ifdef PD_LAB_AP_LOOSEVM
#define HL_VMN(T) (1 + ((T) < PA ? 1 : 0) + (((((T) + 8) % 9) < PA) ? 1 : 0))
#else
#define HL_VMN(T) (1 + ((T) < PA ? 1 : 0) + ((!APPLY && (((T) + 8) % 9) < PA) ? 1 : 0))
#endif
#define HL_GROUP(P, T, g)                                                                                              \
    {                                                                                                                 \
        if constexpr ((g) + 3 < TM) {                                 /* A fragment g+3: row (g+3)/FPR of the wave's pixels */ \
            if constexpr (RPW > 1 && ((g) == 0 || ((g) + 3) % FPR == 0 || RPW == 2))                                  \
                a_cur = HL_AADDR(ab, (T / 3 - 1) * HW2 + (T % 3 - 1), ((g) + 3) / FPR);                               \
            HL_DSR(ar[((g) + 3) & 3], a_cur, (((g) + 3) % FPR) * 1024);                                               \
        }                                                                                                             \
        if constexpr ((g) == 5) {                                                                                     \
            if (more) {                                                                                               \
                asm volatile("s_waitcnt vmcnt(%3) lgkmcnt(0)" : "+v"(ar[1]), "+v"(ar[2]), "+v"(ar[3]) : "n"(HL_VMN(T)) : "memory"); \
                __builtin_amdgcn_s_barrier();                                                                         \
                asm volatile("" ::: "memory");                                                                        \
                const uint32_t b_nx = b_frag + (bs == 2 ? 0 : bs + 1) * B_BYTES;                                      \
                a_nx0 = HL_AADDR(abn, dyn_ * HW2 + dxn_, 0);                                                          \
                a_carry = a_nx0;                                                                                      \
                HL_DSR(bf[bn][0], b_nx, 0); HL_DSR(bf[bn][1], b_nx, 16 * ROWB);                                       \
                HL_DSR(bf[bn][2], b_nx, 32 * ROWB); HL_DSR(bf[bn][3], b_nx, 48 * ROWB);                               \
                HL_DSR(ar[0], a_nx0, 0);                                                                              \
                if constexpr (APPLY && (T) >= 1 && (T) - 1 < PA) {    /* slot T-1 of chunk c+1 (landed: barrier above) */ \
                    if constexpr ((T) == 1) ap_consts(cb + min(c + 1, NC - 1));                                       \
                    ap_fetch((T) - 1, ab ^ 1);                                                                        \
                }                                                                                                     \
            } else {                                                                                                  \
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ar[1]), "+v"(ar[2]), "+v"(ar[3]));                         \
            }                                                                                                         \
        } else if constexpr ((g) == 6) {                                                                              \
            if (more) HL_DSR(ar[1], a_nx0, 1024);                                                                     \
        } else if constexpr ((g) == 7) {                                                                              \
            if (more) {                                                                                               \
                if constexpr (FPR == 2) a_nx0 = HL_AADDR(abn, dyn_ * HW2 + dxn_, 1);                                  \
                HL_DSR(ar[2], a_nx0, (2 % FPR) * 1024);                                                               \
            }                                                                                                         \
        } else {                                                                                                      \
            asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(ar[(g) & 3]));                                                 \
            if constexpr ((g) == 0) asm volatile("" : "+v"(bf[bc][0]), "+v"(bf[bc][1]), "+v"(bf[bc][2]), "+v"(bf[bc][3])); \
            if constexpr (APPLY && (g) == 0 && (T) >= 2 && (T) - 2 < PA)   /* fetched two groups before that wait: landed */ \
                asm volatile("" : "+v"(ap_d), "+v"(ap_v), "+v"(ap_k));                                                \
        }                                                                                                             \
        if constexpr (APPLY && (T) >= 2 && (T) - 2 < PA) {            /* transform piece T-2 under this step's MFMAs */  \
            if constexpr ((g) <= 3) ap_pair(g);                                                                       \
            else if constexpr ((g) == 4) ap_store();                                                                  \
        }                                                                                                             \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                 \
            acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[bc][j], ar[(g) & 3], acc[g][j], 0, 0, 0);           \
        if constexpr ((g) == 0) {                                     /* weight slice s + 2 into stage (s + 2) % 3 */     \
            const int st2 = bs == 0 ? 2 : bs - 1;                                                                     \
            if (s + 2 < S) {                                                                                          \
                glds16h(bp, b_dst + st2 * B_BYTES);                                                                   \
                if constexpr (((T) + 2) % 9 < 8) bp += Cin; else bp += 32 - 8 * Cin;                                  \
            } else glds16h(zero_page, b_dst + st2 * B_BYTES);         /* past the end: harmless load, static counts */   \
        }                                                                                                             \
        if constexpr ((g) == 1 && (T) < PA) {                         /* halo piece T of chunk c + 1 */                 \
            glds16h(halo_src(T, cb + min(c + 1, NC - 1)), ah_dst + piece_off(T, ab ^ 1));                            \
        }                                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
    }
#define HL_STEP(P, T)                                                                                                 \
    {                                                                                                                 \
        constexpr int tn_ = ((T) + 1) % 9;                                                                            \
        constexpr int dyn_ = tn_ / 3 - 1, dxn_ = tn_ % 3 - 1;                                                         \
        constexpr int ab = (P), bc = ((P) + (T)) & 1, bn = bc ^ 1;    /* halo buffer; fragment register set of this / next step */ \
        const int s = c * 9 + (T);                                                                                    \
        const bool more = s + 1 < S;                                                                                  \
        int hpv = hp0;                                                                                                \
        asm volatile("" : "+v"(hpv));                                 /* keeps the per-tap addresses out of loop-invariant registers */ \
        uint32_t a_cur = a_carry;                                     /* row-0 base of this tap (formed at the previous hand-over); later rows at first use */ \
        const int abn = (T) == 8 ? (ab ^ 1) : ab;                                                                     \
        uint32_t a_nx0 = 0;                                           /* computed at the hand-over (short live range) */ \
        HL_GROUP(P, T, 0) HL_GROUP(P, T, 1) HL_GROUP(P, T, 2) HL_GROUP(P, T, 3)                                        \
        HL_GROUP(P, T, 4) HL_GROUP(P, T, 5) HL_GROUP(P, T, 6) HL_GROUP(P, T, 7)                                        \
        bs = bs == 2 ? 0 : bs + 1;                                                                                    \
    }
#define HL_CHUNK(P) HL_STEP(P, 0) HL_STEP(P, 1) HL_STEP(P, 2) HL_STEP(P, 3) HL_STEP(P, 4) HL_STEP(P, 5) HL_STEP(P, 6) HL_STEP(P, 7) HL_STEP(P, 8)
        { const int c = c0; HL_CHUNK(0) }
        if (c0 + 1 < NC) { const int c = c0 + 1; HL_CHUNK(1) }
#undef HL_CHUNK
#undef HL_STEP
#undef HL_GROUP
#undef HL_VMN
    }
#undef HL_AADDR
#undef HL_DSR
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the harmless tail re-loads must land before the LDS is reused
    __syncthreads();

    if (partial != nullptr) {                              // split over chunks: raw f32 partial tile, reduced by splitk_reduce()
        const long long M = (long long)N * HWp;
        float* P = partial + (size_t)blockIdx.y * (size_t)M * Cout;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const long long m = pix(wm * TM * 16 + i * 16 + (lane & 15));
                if (n < Cout) *reinterpret_cast<float4_t*>(P + (size_t)m * Cout + n) = acc[i][j];
            }
        }
        return;
    }
    // ---- epilogue (as k_conv_igemm: transposed accumulator tile -> f16 -> LDS -> coalesced rows, residual, GN partials)
#ifdef PD_LAB_NOEPI                                        // (lab builds only: how much of a tile is the epilogue)
    if (pt != -12345) { if (acc[0][0][0] == 123.456f) Y[0] = (half_t)1.f; return; }
#endif
    half_t* Cs = reinterpret_cast<half_t*>(smem);
    constexpr int CT = BNT / 8;                            // column threads (one channel octet each)
    constexpr int RPP = NWAVES * 64 / CT;                  // rows per pass
    const int col8 = (tid % CT) * 8;
    // Bias first, then (residual convs) every residual row of this thread in one batch: the main loop's registers are dead
    // here, and the HBM latency of the batch runs under the accumulator staging instead of once per output row.  The body is
    // instantiated per case so that the compiler's wait counts stay exact (no conservative vmcnt(0) at a join).
    float gs = 0.f, gq = 0.f;
    auto epilogue = [&](auto has_res) {
        constexpr bool RES = decltype(has_res)::value;
        float4_t bvs[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nl = wn * 64 + j * 16 + (lane >> 4) * 4;
            bvs[j] = (float4_t){0.f, 0.f, 0.f, 0.f};
            if (bias != nullptr && n0 + nl < Cout) bvs[j] = *reinterpret_cast<const float4_t*>(bias + n0 + nl);
        }
        const bool col_ok = n0 + col8 < Cout;
        const int ccol = col_ok ? n0 + col8 : 0;
        // residual rows in two batches of 8: the first before the accumulators are staged (its HBM latency runs under the staging),
        // the second right behind the staging stores, when the 128 accumulator registers are dead -- all 16 rows in one batch next to
        // the live accumulators cost the W = 32 / strip-64 instances 4-8 spilled VGPRs, and a spilled row is a load + vmcnt(0) + scratch
        // store in the MIDDLE of the batch (round 5: zero scratch on every routed kernel)
        constexpr int NRES = BMT / RPP, NR1 = NRES / 2;
        half8 rres[RES ? NRES : 1];
        auto res_load = [&](int p) {
            // res_up: the residual is the HALF-resolution tensor of an up-ResBlock, nearest-upsampled on the fly (unet.py:237-242
            // h = conv(...) + upsample(x)): the x2 copy of x is never materialised
            const int rp = p * RPP + tid / CT;
            const int ry = ty0 + (rp >> WLOG), rx = x0 + (rp & (W - 1));
            const long long rm = res_up ? ((long long)img * (H >> 1) + (ry >> 1)) * ((1 << ILOG) >> 1) + (rx >> 1) : pix(rp);
            rres[p] = *reinterpret_cast<const half8*>(residual + (size_t)rm * Cout + ccol);
        };
        if (RES) {
#pragma unroll
            for (int p = 0; p < NR1; ++p) res_load(p);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nl = wn * 64 + j * 16 + (lane >> 4) * 4;
            const float4_t bv = bvs[j];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int ml = wm * TM * 16 + i * 16 + (lane & 15);
                half4 h;
#pragma unroll
                for (int r = 0; r < 4; ++r) h[r] = (half_t)(acc[i][j][r] + bv[r]);
                *reinterpret_cast<half4*>(&Cs[ml * CS_LD + nl]) = h;
            }
        }
        if (RES) {
            asm volatile("" ::: "memory");                 // (the second batch is issued HERE, not hoisted above the staging stores)
#pragma unroll
            for (int p = NR1; p < NRES; ++p) res_load(p);
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < BMT / RPP; ++p) {
            const int row = p * RPP + tid / CT;
            const long long m = pix(row);
            half8 v = *reinterpret_cast<const half8*>(&Cs[row * CS_LD + col8]);
            const size_t o = (size_t)m * Cout + n0 + col8;
            if (RES) {
                const half8 rv = rres[p];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] + (float)rv[e]);
            }
            if (col_ok) {
#ifndef PD_LAB_HALO_NOSTORE                                // (lab builds only, WRONG results: the epilogue without its global stores)
                *reinterpret_cast<half8*>(Y + o) = v;
#endif
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; gs += f; gq += f * f; }
            }
        }
    };
    if (residual != nullptr) epilogue(std::true_type{}); else epilogue(std::false_type{});
#ifdef PD_LAB_HALO_NOSTORE
    if (gs == 123.456f && gq == 7.f) Y[0] = (half_t)1.f;   // (keeps the staging reads alive)
#endif
    if (gn_part != nullptr) {                              // [img][chunk][Cout/8][2], chunk = 512-pixel tile of the image
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);
        red[tid * 2] = gs; red[tid * 2 + 1] = gq;
        __syncthreads();
        // two-level fixed-order column sums (as k_conv_sk): 8 threads per octet take every 8th row, one thread adds the 8 sub-sums
        constexpr int SUB = 8;
        static_assert(RPP % SUB == 0, "GroupNorm partial reduce");
        float* red2 = red + NWAVES * 64 * 2;
        if (tid < CT * SUB) {
            const int j = tid % SUB, c = tid / SUB;
            float s1 = 0.f, q1 = 0.f;
#pragma unroll
            for (int r = 0; r < RPP / SUB; ++r) { s1 += red[((r * SUB + j) * CT + c) * 2]; q1 += red[((r * SUB + j) * CT + c) * 2 + 1]; }
            red2[tid * 2] = s1; red2[tid * 2 + 1] = q1;
        }
        __syncthreads();
        if (tid < CT && n0 + tid * 8 < Cout) {
            float s1 = 0.f, q1 = 0.f;
#pragma unroll
            for (int j = 0; j < SUB; ++j) { s1 += red2[(tid * SUB + j) * 2]; q1 += red2[(tid * SUB + j) * 2 + 1]; }
            const int chunks = tpi;
            const int chunk = tin;
            float* dst = gn_part + (((size_t)img * chunks + chunk) * (Cout >> 3) + (n0 >> 3) + tid) * 2;
            dst[0] = s1; dst[1] = q1;
        }
    }
}

template <int WLOG, int ILOG, bool APPLY>
int launch_halo_v(const half_t* X, const half_t* Wt, const float* bias, const half_t* residual, half_t* Y, int N, int H, int Cin,
                  int Cout, int Cout_pad, const half_t* zero_page, hipStream_t s, float* gn_part, int splits, float* partial,
                  const float* ap_table, int res_up, int in_up) {
    constexpr int W = 1 << WLOG, RT = 512 / W, HP = (RT + 2) * (W + 2), NPA = (HP + 15) / 16;
    const size_t loop_bytes = (size_t)2 * NPA * 1024 + 3 * 8192 + (APPLY ? (size_t)8 * 1024 + (size_t)Cin * 8 + (size_t)NPA * 16 + 16 : 0);
    const size_t smem = std::max<size_t>(loop_bytes, (size_t)512 * (128 + 8) * 2);
    PD_REQUIRE(smem <= 160 * 1024, "conv3x3_halo: %zu bytes of LDS (Cin = %d)", smem, Cin);
    auto kern = k_conv3x3_halo<WLOG, ILOG, APPLY>;
    PD_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const int n_tiles = Cout_pad / 128;
    const int total = (int)(((long long)N * H * (1 << ILOG)) / 512) * n_tiles;
    kern<<<dim3(total, splits), 512, smem, s>>>(X, Wt, bias, residual, Y, N, H, Cin, Cout, n_tiles, total, zero_page, gn_part, splits,
                                                partial, ap_table, res_up, in_up);
    return PDHIP_OK;
}
template <int WLOG, int ILOG = WLOG>
int launch_halo(const half_t* X, const half_t* Wt, const float* bias, const half_t* residual, half_t* Y, int N, int H, int Cin,
                int Cout, int Cout_pad, const half_t* zero_page, hipStream_t s, float* gn_part, int splits, float* partial,
                const float* ap_table, int res_up, int in_up) {
    constexpr int W = 1 << WLOG, RT = 512 / W, HP = (RT + 2) * (W + 2), NPA = (HP + 15) / 16, PA = (NPA + 7) / 8;
    if constexpr (PA <= 7) {
        if (ap_table != nullptr)
            return launch_halo_v<WLOG, ILOG, true>(X, Wt, bias, residual, Y, N, H, Cin, Cout, Cout_pad, zero_page, s, gn_part, splits, partial, ap_table, res_up, in_up);
    } else {
        PD_REQUIRE(ap_table == nullptr, "conv3x3_halo: the full-row 256-wide tile has no APPLY variant");
    }
    return launch_halo_v<WLOG, ILOG, false>(X, Wt, bias, residual, Y, N, H, Cin, Cout, Cout_pad, zero_page, s, gn_part, splits, partial, nullptr, res_up, in_up);
}

}  // namespace

bool conv3x3_halo_eligible(int N, int H, int W, int Cin, int Cout_pad) {
    return (W == 32 || W == 64 || W == 128 || W == 256) && ((long long)H * W) % 512 == 0 && Cin % 32 == 0 && Cout_pad % 128 == 0 &&
           (long long)N * H * W * Cin * 2 <= 0xfffffff0LL;
}

// split factor over the 32-channel chunks when the tile count alone would leave CUs idle (0 = this layer should not use the
// halo kernel at all: too few tiles even with the largest useful split)
int conv3x3_halo_splits(int N, int H, int W, int Cin, int Cout, int Cout_pad, size_t splitk_ws_floats) {
    const long long M = (long long)N * H * W;
    const long long tiles = (M / 512) * (Cout_pad / 128);
    if (tiles >= 256) return 1;
    const int nct = Cin / 32;
    int splits = (int)std::min<long long>(std::min<long long>(256 / tiles, nct / 2), 8);
    while (splits > 1 && (size_t)splits * (size_t)M * Cout > splitk_ws_floats) --splits;
    if (splits < 2 || tiles * splits < 128) return 0;
    return splits;
}

// gn_part (optional): fused GroupNorm octet partials; chunks per image returned through gn_fused (H*W/512 from the direct
// epilogue, H*W/16 from the split reduce)
int conv3x3_halo(const half_t* X, const half_t* Wt, const float* bias, const half_t* residual, half_t* Y, int N, int H, int W,
                 int Cin, int Cout, int Cout_pad, const half_t* zero_page, hipStream_t s, float* gn_part, int* gn_fused,
                 float* splitk_ws, size_t splitk_ws_floats, const float* apply_table, int res_up, int in_up) {
    PD_REQUIRE(in_up == 0 || (apply_table == nullptr && H % 2 == 0), "conv3x3_halo: an up-sampled input excludes the in-conv GroupNorm");
    PD_REQUIRE(conv3x3_halo_eligible(N, H, W, Cin, Cout_pad), "conv3x3_halo: unsupported geometry (N=%d H=%d W=%d Cin=%d)", N, H, W, Cin);
    int splits = conv3x3_halo_splits(N, H, W, Cin, Cout, Cout_pad, splitk_ws ? splitk_ws_floats : 0);
    if (splits < 1) splits = 1;
    if (g_force_splits >= 1 && splitk_ws != nullptr)            // tuning / test hook
        splits = (int)std::min<size_t>(std::min(g_force_splits, Cin / 32), splitk_ws_floats / ((size_t)N * H * W * Cout));
    if (splits < 1) splits = 1;
    PD_REQUIRE(!res_up || (splits == 1 && residual != nullptr && H % 2 == 0), "conv3x3_halo: an up-sampled residual needs the direct (unsplit) epilogue");
    float* partial = splits > 1 ? splitk_ws : nullptr;
    const bool fuse_sk = splits > 1 && gn_part != nullptr && (H * W) % 16 == 0;
    if (gn_fused) *gn_fused = gn_part ? (splits > 1 ? (fuse_sk ? (H * W) / 16 : 0) : (int)(((long long)H * W) / 512)) : 0;
    float* gnp = splits > 1 ? nullptr : gn_part;
#define HL_LAUNCH(WL) launch_halo<WL>(X, Wt, bias, residual, Y, N, H, Cin, Cout, Cout_pad, zero_page, s, gnp, splits, partial, apply_table, res_up, in_up)
    int rc;
#define HL_LAUNCH2(WL, IL) launch_halo<WL, IL>(X, Wt, bias, residual, Y, N, H, Cin, Cout, Cout_pad, zero_page, s, gnp, splits, partial, apply_table, res_up, in_up)
    // 256-wide images: column strips of 128 (4 rows x 128 per tile: 780 halo pixels per chunk instead of 1 032) measured +2-3 %
    // over full rows, strips of 64 +1.5-3 % (that instance spilled 8 VGPRs and was never the default: removed in round 5); at 128 wide
    // strips do not pay.  g_halo_strips: 0 automatic, 1 full rows.
    if (W == 256 && g_halo_strips != 1 && H % 4 == 0) rc = HL_LAUNCH2(7, 8);
    else if (W == 256 && apply_table != nullptr) { set_error("conv3x3_halo: APPLY needs H %% 4 == 0 at W = 256"); return PDHIP_E_ARG; }
    else if (W == 256) rc = HL_LAUNCH(8);
    else if (W == 128) rc = HL_LAUNCH(7);
    else if (W == 64) rc = HL_LAUNCH(6);
    else rc = HL_LAUNCH(5);
#undef HL_LAUNCH
#undef HL_LAUNCH2
    if (rc) return rc;
    PD_LAUNCH_CHECK();
    if (splits > 1) return splitk_reduce(partial, splits, (long long)N * H * W, Cout, bias, residual, Y, fuse_sk ? gn_part : nullptr, H * W, s);
    return PDHIP_OK;
}

}  // namespace pdnn

extern "C" int pdhip_debug_set_conv_halo_strips(int mode) {
    const int old = pdnn::g_halo_strips;
    pdnn::g_halo_strips = mode;
    return old;
}
