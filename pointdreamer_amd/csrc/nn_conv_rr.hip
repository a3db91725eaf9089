// Row U1, the 8^2 / 16^2 / 32^2 levels at small batch (view-parallel: ONE view per GPU): "row-resident" 3x3 convolution with
// the GroupNorm (+ FiLM) + SiLU of its input applied while staging, and the weights streamed HBM -> registers.
// Replaces, per ResBlock conv of those levels (models/DDNM/guided_diffusion/unet.py:143-260: `in_layers` = GroupNorm32 -> SiLU -> conv,
// `out_layers` = GroupNorm32 -> FiLM -> SiLU -> conv, `skip_connection`), the pair k_gn_apply + k_conv_sk.
//
// Why another kernel.  At batch 1 these 42 convs are weight streams (19-38 MB of f16 weights against 0.1-1 MB
// of activations) or short GEMMs (M = 1024), and k_conv_sk -- an implicit GEMM that stages BOTH operands through LDS per K-step, behind a
// barrier per step -- moved their weights at 0.9-1.5 TB/s and ran the 32^2 layers at 0.2 PFLOP/s; in front of each sat a 5 us
// k_gn_apply launch.  Here:
//   * the workgroup's activation tile -- MT x 16 pixels as whole image rows, with its 3x3 halo, CS input channels -- is staged ONCE per
//     channel chunk into LDS ([row][column][channel], pixel stride CS * 2 + 32 bytes: the four lane groups of a ds_read_b128 then hit 64
//     distinct banks) and all nine taps read it at shifted addresses: 9x fewer activation bytes from L2 than the implicit GEMM;
//   * GroupNorm statistics are finished in the kernel from the producing conv's octet partials (k_gn_apply's FIN arithmetic, same f64
//     summation order) and the element map gn_elem (nn_common.h) is applied between the global load and the LDS store: the stand-alone
//     pass and its tensor disappear, the result is bit-identical to the two-pass form;
//   * weights live in a fragment-major copy made at load time ([Cout/16][K-step][lane][8 halfs]: one v_mfma_f32_16x16x32_f16 operand =
//     one fully coalesced 1 KiB global_load_dwordx4) and go HBM -> VGPR -> MFMA without touching LDS or a barrier; a wave issues ALL the
//     weight loads of its K share (18-36 KiB in flight per wave, 72-144 KiB per CU) before it touches the activations, so the stream
//     runs under the staging and the transform;
//   * the four waves split K (each owns every fourth 32-channel step, all nine taps, all MT pixel tiles: every weight byte is loaded by
//     exactly one wave of the chip), their accumulators meet in LDS; K is split across workgroups too ("slabs") with the in-launch
//     ticketed combine of k_conv_sk (write-through slices, last arriver sums in slice order: deterministic);
//   * a ResBlock's skip 1x1 (unet.py:255) rides along as extra slabs over the block input, like k_conv_sk<10>.
// Output: bias (+ residual, optionally read at half resolution) -> f16 NHWC + GroupNorm octet partials of the result.
#include "nn_common.h"
#include <algorithm>
#include <type_traits>
using namespace pdhip;
namespace pdnn {

typedef __attribute__((ext_vector_type(4))) unsigned int rr_u32x4;
#define RR_TRY(expr) do { int rc_ = (expr); if (rc_ != PDHIP_OK) return rc_; } while (0)
// raw barrier: LDS traffic of this wave drained, compiler memory order pinned, global loads left in flight (a __syncthreads() would
// drain vmcnt too -- the weight stream must survive every barrier of the main loop)
__device__ __forceinline__ void rr_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// gn_elem (nn_common.h) over the 8 channels of a 16-byte piece, stage by stage: the same operations in the same order on the same values
// as eight gn_elem calls (bit-identical), but written so that the eight dependency chains advance TOGETHER.  Called element by element
// the opaque `asm("" : "+v"(f))` statements of gn_round_f16 -- which keep hipcc from fusing an op with the f16 rounding behind it -- stay in
// program order, i.e. one element's whole chain (14 dependent VALU ops, two of them transcendental) ran before the next one started:
// ~200 cycles per element at one wave per SIMD (12 us of a 16^2 conv).
template <bool FILM>
__device__ __forceinline__ void rr_gn_vec8(half8& hv, const float (&ga)[8], const float (&gb)[8], const float (&t1)[8], const float (&sh)[8], bool silu) {
    float f[8], g[8];
#define RR_ALL(expr) _Pragma("unroll") for (int e = 0; e < 8; ++e) { expr; }
    RR_ALL(f[e] = __builtin_fmaf((float)hv[e], ga[e], gb[e]))
    RR_ALL(asm("" : "+v"(f[e])))
    RR_ALL(f[e] = (float)(half_t)f[e])
    if (FILM) {
        RR_ALL(f[e] = f[e] * t1[e])
        RR_ALL(asm("" : "+v"(f[e])))
        RR_ALL(f[e] = (float)(half_t)f[e])
        RR_ALL(f[e] = f[e] + sh[e])
        RR_ALL(asm("" : "+v"(f[e])))
        RR_ALL(f[e] = (float)(half_t)f[e])
    }
    if (silu) {                                            // (wave-uniform: GroupNorm without SiLU -- the attention norm -- skips the two transcendentals)
        RR_ALL(g[e] = silu_f(f[e]))
        RR_ALL(asm("" : "+v"(g[e])))
        RR_ALL(hv[e] = (half_t)g[e])
    } else {
        RR_ALL(hv[e] = (half_t)f[e])
    }
#undef RR_ALL
}

struct RrSrcK {                    // one K source as the kernel sees it
    const half_t* x; const half_t* x2;      // [N,H,W,Ca] and (virtual channel concat) [N,H,W,C-Ca] or null
    int C, Ca;
    int gn;                        // 0 raw, 1 GroupNorm, 2 GroupNorm + SiLU
    const float* gamma; const float* beta; const float* film; long long film_stride;    // film: row n = (scale[C] | shift[C]) or null
    const float* partA; const float* partB; int chunksA, chunksB; float eps;          // octet partials of x / x2 ([N][chunks][C/8][2])
    int cg, cg_magic, opg, pps_log2;   // channels per group, ceil(2^20 / cg) (x / cg == (x * magic) >> 20 for x < 4096), octets per group, log2 of k_gn_apply's slot count
};
#define RR_MAXQ 8
#ifndef RR_W_AUX
#define RR_W_AUX 0                 // cache policy of the weight stream (lab: 2 = nt)
#endif
template <int B, int E, typename Fn>
__device__ __forceinline__ void rr_static_for(Fn&& fn) {
    if constexpr (B < E) { fn(std::integral_constant<int, B>{}); rr_static_for<B + 1, E>(fn); }
}
struct RrArgs {
    RrSrcK src[2];                 // [0] the conv's own input (taps0 = 9, or 1 for a 1x1 layer), [1] the appended skip 1x1 source (S1 == 0: none)
    int taps0;
    int S0, U0, S1, U1;            // K slabs per source and CS-channel units per slab
    const half_t* wf; int KS;      // fragment-major weights [Cout/16][KS][64][8]; KS = taps0 * C0 / 32 + C1 / 32
    const float* bias; const half_t* residual; int res_up; half_t* Y;
    int N, H, Cout, ntiles, bands;
    float* slabs; unsigned* tickets; float* gn_part;
};

namespace {

#ifdef PD_LAB_RR_STAMP                                      // (lab builds only: where one workgroup's time goes, s_memtime cycles of wave 0)
__device__ unsigned long long g_rr_stamps[256 * 16];
#define RR_STAMP(k) do { if (tid == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); lab_t[k] += t_ - lab_prev; lab_prev = t_; } } while (0)    // time spent in the phase ENDING at stamp k, summed over the units
#else
#define RR_STAMP(k) do {} while (0)
#endif

// W: image width (8, 16, 32); MT: 16-pixel tiles per workgroup (MT * 16 / W whole rows); NF: 16-channel output fragments per workgroup;
// CS: input channels per staged unit.  The four waves split the unit's 32-channel steps (K), every wave runs all MT x NF fragments.
template <int W, int MT, int NF, int CS>
struct RrCfg {
    static constexpr int TR = MT * 16 / W, ROWS = TR + 2;
    static constexpr int RS = W == 8 ? 16 : W + 2;           // LDS pixels per row (W = 8: a 16-pixel tile is two rows -- stride 16 keeps their banks apart)
    static constexpr int PS = CS * 2 + 32;                   // bytes per LDS pixel
    static constexpr int PP = CS / 8, PXI = 256 / PP, UP = ROWS * W / PXI;     // 16-byte pieces per pixel, pixels per staging pass, passes
    static constexpr int JW = CS / 128;                      // 32-channel steps per wave and unit
    static constexpr int ACT_BYTES = ROWS * RS * PS;
    static constexpr int F = MT * NF, FW = F / 4;            // accumulator fragments of the tile; per wave after the exchange
    static constexpr int PARK_BYTES = 4 * F * 1024;
    static constexpr int CS_LD = NF * 16 + 8;
    static constexpr int EPI_BYTES = MT * 16 * CS_LD * 2;
    static constexpr int MAIN_BYTES = ACT_BYTES > PARK_BYTES ? (ACT_BYTES > EPI_BYTES ? ACT_BYTES : EPI_BYTES) : (PARK_BYTES > EPI_BYTES ? PARK_BYTES : EPI_BYTES);
    static constexpr int FIN_OFF = (MAIN_BYTES + 255) & ~255;                // double2 s_fin[17][8]
    static constexpr int ST_OFF = FIN_OFF + 17 * 8 * 16;                      // float s_st[17][2]
    static constexpr int FLAG_OFF = ST_OFF + 17 * 8;
    static constexpr int SMEM = FLAG_OFF + 16;
    static_assert(MT * 16 % W == 0 && (ROWS * W) % PXI == 0 && CS % 128 == 0 && F % 4 == 0, "tile geometry");
    static_assert(W != 8 || MT == 4, "8-wide images: the whole image per workgroup");
    static_assert(SMEM <= 160 * 1024, "LDS budget");
};

// GNF: the in-staging GroupNorm path compiled in (the large-tile variants leave it out: its constants cost ~80 registers they do not have)
template <int W, int MT, int NF, int CS, bool GNF>
__global__ __launch_bounds__(256) void k_conv_rr(const RrArgs a) {
    using G = RrCfg<W, MT, NF, CS>;
    constexpr int TR = G::TR, ROWS = G::ROWS, RS = G::RS, PS = G::PS, PP = G::PP, PXI = G::PXI, UP = G::UP;
    constexpr int JW = G::JW, MTW = MT, F = G::F, FW = G::FW, CS_LD = G::CS_LD;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
#ifdef PD_LAB_RR_STAMP
    unsigned long long lab_t[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long lab_prev = __builtin_readcyclecounter();
    lab_t[15] = lab_prev;
#endif
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave;                                 // this wave's K share: 32-channel steps wk * JW .. of every unit
    const int H = a.H;
    const int S = a.S0 + a.S1;
    const int mtiles = a.N * a.bands;
    // XCD-aware work id (workgroup b runs on XCD b % 8): every XCD gets a contiguous run of ids; id order = (slab, n-tile, pixel tile) with
    // the pixel tile fastest, so the workgroups that read the same weight fragments (and then the same activation chunk) share one L2
    int wid;
    {
        const int total = mtiles * a.ntiles * S;
        const int b = blockIdx.x, q = total >> 3, r = total & 7, xcd = b & 7, i = b >> 3;
        wid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
    }
    const int mt = wid % mtiles, rest = wid / mtiles, nt = rest % a.ntiles, slab = rest / a.ntiles;
    const int tile = nt * mtiles + mt;
    const int img = mt / a.bands, band = mt - img * a.bands, r0 = band * TR;
    const int n0 = nt * (NF * 16);
    const bool second = slab >= a.S0;                    // (wave-uniform) this slab runs over the appended skip source
    const RrSrcK* const sr = &a.src[second ? 1 : 0];    // (fields are read where they are used: two dozen pre-selected scalars cost SGPRs for the whole kernel)
    const int taps = second ? 1 : a.taps0;
    const int units = second ? a.U1 : a.U0;
    const int unit0 = (second ? slab - a.S0 : slab) * units;
    const int ks_src0 = second ? a.taps0 * (a.src[0].C >> 5) : 0;          // first K-step of this source in the weight fragments
    const int sgn = GNF ? sr->gn : 0;

    // ---- halo columns: zero once (staging never writes them); LDS pixel (ry, 0) and (ry, W + 1)
    for (int q = tid; q < ROWS * 2 * PP; q += 256) {
        const int ry = q / (2 * PP), rem = q - ry * (2 * PP), side = rem / PP, oc = rem - side * PP;
        *reinterpret_cast<half8*>(smem + (ry * RS + (side ? W + 1 : 0)) * PS + oc * 16) = (half8){0, 0, 0, 0, 0, 0, 0, 0};
    }
    // ---- this lane's activation fragment base: pixel (lane & 15) of tile 0 of its wave group, 16-byte slot (lane >> 4) of its first 32-channel step
    int lane_px;
    if (W == 8) lane_px = ((lane & 15) >> 3) * RS + (lane & 7);
    else lane_px = lane & 15;
    const char* const frag_base = smem + (lane_px * PS + (lane >> 4) * 16 + wk * JW * 64);

    float4_t acc[MTW][NF];
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    const int oct = tid % PP, psub = tid / PP;           // staging: this thread's channel octet (fixed) and pixel sub-slot

    // One unit = CS input channels of the slab's source: stage (+ transform) the halo image, then the wave's K-steps of it.
    // Code size matters as much as cycles for a 5 us launch (DESIGN_HISTORY: ~0.3 us per KB of straight-line code executed once): the
    // element map runs in a ROLLED loop over the thread's pieces, IN PLACE in LDS (raw store as the loads land, then read - transform -
    // write of the thread's own pieces -- a wave's LDS operations execute in order, no barrier in between), and only the fragment
    // reads + MFMAs are unrolled (register-resident weight fragments cannot be indexed by a loop variable).
    // Units are software-pipelined: the activations (and statistics inputs) of unit u + 1 are requested before the MFMAs of unit u, its
    // weight fragments right behind them (the registers are free then) -- their latency runs under the MFMAs and the staging barriers.
    auto run_slab = [&](auto unit_tap_tag) {
        constexpr int UT = decltype(unit_tap_tag)::value;     // 9 or 1: the slab's tap count (wave-uniform, fixed for the workgroup)
        const int sC = sr->C, sCa = sr->Ca;
        const int cg = sr->cg, opg = sr->opg, ppl = sr->pps_log2, pps = 1 << ppl, mg = sr->cg_magic;
        const float* const sfilm = sr->film;
        const half_t* const sx2 = sr->x2;
        float gam[8], bet[8], fsc[8], fsh[8];
        float2 pv[RR_MAXQ];                                // this thread's octet partials: chunks sl, sl + pps, ... (host: at most RR_MAXQ of them)
        int npv = 0;
        half8 av[UP];
        half8 wf[JW][UT][NF];
        // DBW (the variants with few fragment registers): the NEXT unit's fragments are requested into a shadow set before the MFMAs of this unit and
        // copied over behind them (36 v_mov against ~2 000 cycles of exposed L2 latency per unit when they were requested after the MFMAs)
        constexpr bool DBW = JW * UT * NF * 4 <= 40;
        half8 wf2[DBW ? JW : 1][DBW ? UT : 1][DBW ? NF : 1];

        // statistics inputs of unit u (in-order vmcnt: they are requested BEFORE the unit's other loads, which may still fly when these are needed)
        auto issue_stats = [&](int u) {
            const int c0 = (unit0 + u) * CS;
            const int g_first = (c0 * mg) >> 20, ng = (((c0 + CS - 1) * mg) >> 20) - g_first + 1;
            // statistics threads: (group gl = tid >> 3, r = tid & 7 = (octet k of the group, slot sl)), r < opg * pps <= 8
            if ((tid >> 3) < ng && (tid & 7) < opg * pps) {
                const int gl = tid >> 3, r = tid & 7, k = r >> ppl, sl = r & (pps - 1);
                const int o = (g_first + gl) * opg + k, oa = sCa >> 3;
                const bool inA = o < oa;
                const int chunks = inA ? sr->chunksA : sr->chunksB, os = inA ? oa : ((sC - sCa) >> 3);
                const float2* p = reinterpret_cast<const float2*>(inA ? sr->partA : sr->partB) + (size_t)img * chunks * os + (inA ? o : o - oa);
                npv = (chunks - sl + pps - 1) >> ppl;
#pragma unroll
                for (int q = 0; q < RR_MAXQ; ++q) pv[q] = p[(size_t)min(sl + q * pps, chunks - 1) * os];     // clamped index, conditional add below
            }
            // (no control flow around the FiLM loads -- conditional stores into the register arrays sent them to scratch: without FiLM the two
            // rows alias gamma and are never used)
            const float* const sgamma = sr->gamma + c0 + oct * 8; const float* const sbeta = sr->beta + c0 + oct * 8;
            const float* const f0 = sfilm != nullptr ? sfilm + (size_t)img * sr->film_stride + c0 + oct * 8 : sgamma;
            const float* const f1 = sfilm != nullptr ? f0 + sC : sgamma;
            const float4_t g0 = *reinterpret_cast<const float4_t*>(sgamma), g1 = *reinterpret_cast<const float4_t*>(sgamma + 4);
            const float4_t b0 = *reinterpret_cast<const float4_t*>(sbeta), b1 = *reinterpret_cast<const float4_t*>(sbeta + 4);
            const float4_t c0v = *reinterpret_cast<const float4_t*>(f0), c1v = *reinterpret_cast<const float4_t*>(f0 + 4);
            const float4_t h0 = *reinterpret_cast<const float4_t*>(f1), h1 = *reinterpret_cast<const float4_t*>(f1 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                gam[e] = g0[e]; gam[4 + e] = g1[e]; bet[e] = b0[e]; bet[4 + e] = b1[e];
                fsc[e] = c0v[e]; fsc[4 + e] = c1v[e]; fsh[e] = h0[e]; fsh[4 + e] = h1[e];
            }
        };
        // the unit's activations: rows r0 - 1 .. r0 + TR of the image, CS channels, 16 bytes per (pixel, octet).  Buffer loads with the
        // pass-dependent part of the address in an SGPR (soffset): no 64-bit pointer per piece in VGPRs (18-20 pieces -> they spilled)
        auto issue_act = [&](int u) {
            const int c0 = (unit0 + u) * CS;
            const bool inB = sx2 != nullptr && c0 >= sCa;
            const int cs = sx2 == nullptr ? sC : (inB ? sC - sCa : sCa);
            const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(inB ? sx2 : sr->x), 0, 0x7fffffff, 0x00020000);
            const int cch = inB ? c0 - sCa : c0;
            if (PXI <= W) {
                const int voff = (psub * cs + oct * 8) * 2;
#pragma unroll
                for (int p = 0; p < UP; ++p) {
                    const int ry = (p * PXI) / W, col0 = (p * PXI) % W;
                    const int y = min(max(r0 - 1 + ry, 0), H - 1);
                    av[p] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rx, voff, (((img * H + y) * W + col0) * cs + cch) * 2, 0));
                }
            } else {
#pragma unroll
                for (int p = 0; p < UP; ++p) {
                    const int pix = p * PXI + psub, ry = pix / W, xx = pix - ry * W;
                    const int y = min(max(r0 - 1 + ry, 0), H - 1);
                    av[p] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rx, (((img * H + y) * W + xx) * cs + oct * 8) * 2, cch * 2, 0));
                }
            }
        };
        // the wave's weight fragments of the unit: K-steps (32-channel step j, tap t), all NF channel fragments -- one coalesced 1 KiB buffer
        // load each, lane offset in the VGPR, everything else in the SGPR
        auto issue_w = [&](int u, auto shadow_tag) {
            constexpr bool SH = decltype(shadow_tag)::value;
            const int c0 = (unit0 + u) * CS;
            const auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(a.wf), 0, 0x7fffffff, 0x00020000);
#pragma unroll
            for (int jw = 0; jw < JW; ++jw) {
                const int c32 = (c0 >> 5) + wk * JW + jw;
                const int ks = ks_src0 + (UT == 1 ? c32 : c32 * 9);
#pragma unroll
                for (int t = 0; t < UT; ++t)
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        const half8 v = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16, (((n0 >> 4) + f) * a.KS + ks + t) * 1024, RR_W_AUX));
                        if constexpr (SH) wf2[jw][t][f] = v; else wf[jw][t][f] = v;
                    }
            }
        };

        if (sgn) issue_stats(0);
        issue_act(0);
        issue_w(0, std::false_type{});
        RR_STAMP(1);
#pragma unroll 1
        for (int u = 0; u < units; ++u) {
            const int c0 = (unit0 + u) * CS;             // first channel of the unit in the (virtual concat) source
            // statistics: k_gn_apply's FIN arithmetic (slot partials over the chunks, then the group's octets x slots in order, f64)
            float ga[8], gb[8], t1[8], sh[8];
            if (sgn) {
                const int g_first = (c0 * mg) >> 20, ng = (((c0 + CS - 1) * mg) >> 20) - g_first + 1;
                double* s_fin = reinterpret_cast<double*>(smem + G::FIN_OFF);
                float* s_st = reinterpret_cast<float*>(smem + G::ST_OFF);
                if ((tid >> 3) < ng && (tid & 7) < opg * pps) {
                    double ds = 0.0, dq = 0.0;
#pragma unroll
                    for (int q = 0; q < RR_MAXQ; ++q)
                        if (q < npv) { ds += (double)pv[q].x; dq += (double)pv[q].y; }
                    s_fin[tid * 2] = ds; s_fin[tid * 2 + 1] = dq;          // [group][8 (k, slot)]
                }
                rr_sync();                                 // (lgkmcnt only: the weight loads stay in flight)
                if (tid < ng) {
                    double s1 = 0.0, q1 = 0.0;
                    for (int r = 0; r < opg * pps; ++r) { s1 += s_fin[(tid * 8 + r) * 2]; q1 += s_fin[(tid * 8 + r) * 2 + 1]; }     // k outer, slot inner: k_gn_apply's order
                    const double cnt = (double)H * W * cg;
                    const double mean = s1 / cnt;
                    double var = q1 / cnt - mean * mean;
                    if (var < 0.0) var = 0.0;
                    s_st[2 * tid] = (float)mean; s_st[2 * tid + 1] = (float)(1.0 / sqrt(var + (double)sr->eps));
                }
                rr_sync();
                const int gl = (((c0 + oct * 8) * mg) >> 20) - g_first;
                const float mean = s_st[2 * gl], rstd = s_st[2 * gl + 1];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    ga[e] = rstd * gam[e];
                    gb[e] = bet[e] - mean * ga[e];
                    t1[e] = (float)(half_t)(1.0f + (float)(half_t)fsc[e]); sh[e] = (float)(half_t)fsh[e];      // (used with FiLM only)
                }
            }
            RR_STAMP(2);
            // every wave is done with the previous unit's LDS image -> raw store (zero rows outside the image) -> in-place transform
            rr_sync();
#pragma unroll
            for (int p = 0; p < UP; ++p) {
                const int pix = p * PXI + psub, ry = pix / W, xx = pix - ry * W;
                const int y = r0 - 1 + ry;
                half8 hv = av[p];
                if (y < 0 || y >= H) hv = (half8){0, 0, 0, 0, 0, 0, 0, 0};   // zero padding applies to the TRANSFORMED image: such rows skip the map below
                *reinterpret_cast<half8*>(smem + (ry * RS + xx + 1) * PS + oct * 16) = hv;
            }
            RR_STAMP(3);
            if (sgn) {
                const bool silu = sgn == 2;
                const bool film = sfilm != nullptr;
                char* const pbase = smem + ((psub / W) * RS + psub % W + 1) * PS + oct * 16;
                constexpr int PSTEP = (PXI / W) * RS * PS + (PXI % W) * PS;   // PXI pixels further: PXI / W rows (PXI % W == 0 or PXI < W with W % PXI == 0)
                static_assert(PXI % W == 0 || W % PXI == 0, "staging pass geometry");
#pragma unroll 1
                for (int p = 0; p < UP; ++p) {
                    const int pix = p * PXI + psub, ry = pix / W, xx = pix - ry * W;
                    const int y = r0 - 1 + ry;
                    if (y < 0 || y >= H) continue;
                    half8* const q = reinterpret_cast<half8*>(PXI % W == 0 ? pbase + p * PSTEP : smem + (ry * RS + xx + 1) * PS + oct * 16);
                    half8 hv = *q;
                    if (film) rr_gn_vec8<true>(hv, ga, gb, t1, sh, silu);
                    else rr_gn_vec8<false>(hv, ga, gb, t1, sh, silu);
                    *q = hv;
                }
            }
            // the next unit's statistics inputs and activations: requested now, landed by the time the MFMAs below are through
            if (u + 1 < units) {
                if (sgn) issue_stats(u + 1);
                issue_act(u + 1);
                if constexpr (DBW) issue_w(u + 1, std::true_type{});
            }
            rr_sync();
            RR_STAMP(4);
            // MFMAs: weights (A operand: 16 channels x 32 k) x activations (B operand: 32 k x 16 pixels) -> lane holds pixel (lane & 15),
            // channels 4 (lane >> 4) + 0..3 of the fragment.  One "item" = one activation fragment read + its NF MFMAs; the reads run PD items
            // ahead of the MFMAs (left to hipcc each read sat directly in front of its MFMAs behind an lgkmcnt(0))
            {
                constexpr int T = UT;
                constexpr int R = JW * T * MTW, PD = R < 6 ? R : 6;
                half8 xa[PD + 1];
                auto rd = [&](auto r_tag) {
                    constexpr int r = decltype(r_tag)::value;
                    constexpr int jw = r / (T * MTW), t = (r / MTW) % T, i = r % MTW;
                    constexpr int tap_off = T == 1 ? (RS + 1) * PS : ((t / 3) * RS + t % 3) * PS;
                    constexpr int m_off = W == 8 ? i * 2 * RS * PS : (((i * 16) / W) * RS + (i * 16) % W) * PS;
                    xa[r % (PD + 1)] = *reinterpret_cast<const half8*>(frag_base + (tap_off + m_off + jw * 64));
                };
                rr_static_for<0, PD>([&](auto r_tag) { rd(r_tag); });
                rr_static_for<0, R>([&](auto r_tag) {
                    constexpr int r = decltype(r_tag)::value;
                    constexpr int jw = r / (T * MTW), t = (r / MTW) % T, i = r % MTW;
                    if constexpr (r + PD < R) rd(std::integral_constant<int, r + PD>{});
#pragma unroll
                    for (int f = 0; f < NF; ++f) acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[jw][t][f], xa[r % (PD + 1)], acc[i][f], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);     // pin the order: read r + PD, then the MFMAs of item r
                });
            }
            if (u + 1 < units) {
                if constexpr (DBW) {
#pragma unroll
                    for (int jw = 0; jw < JW; ++jw)
#pragma unroll
                        for (int t = 0; t < UT; ++t)
#pragma unroll
                            for (int f = 0; f < NF; ++f) wf[jw][t][f] = wf2[jw][t][f];
                } else {
                    issue_w(u + 1, std::false_type{});        // (the fragment registers are free again)
                }
            }
            RR_STAMP(5);
        }
    };

    if (taps == 9) run_slab(std::integral_constant<int, 9>{});
    else run_slab(std::integral_constant<int, 1>{});
    rr_sync();                                             // all fragment reads done: the LDS image is free

    // ---- the four K shares meet in LDS: every wave parks its MT x NF accumulator fragments (register order, 16 bytes per lane), then wave w
    // sums fragments w * FW .. w * FW + FW - 1 over the waves in wave order -- all 256 lanes add, nobody waits for one wave's serial chain
    float4_t sum[FW];
    {
        float4_t* R = reinterpret_cast<float4_t*>(smem);
#pragma unroll
        for (int f = 0; f < F; ++f) R[(wave * F + f) * 64 + lane] = acc[f / NF][f % NF];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < FW; ++q) {
            const int f = wave * FW + q;
            sum[q] = R[f * 64 + lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float4_t v = R[(w * F + f) * 64 + lane];
                sum[q][0] += v[0]; sum[q][1] += v[1]; sum[q][2] += v[2]; sum[q][3] += v[3];
            }
        }
        __syncthreads();
    }

    RR_STAMP(6);
    // ---- in-launch split-K combine over the slabs (the protocol of k_conv_sk: write-through slices, drained, one ticket per workgroup; the
    // last arriver sums all slices in slab order -- its own included, so the order never depends on who is last).  Every wave owns FW
    // fragments of the tile: the last arriver's S x FW slice loads are all in flight at once (one wave re-reading 8 x 32 KB slice by
    // slice was 16 us of a 16^2 conv: a handed-off tile arrives at ~65 GB/s per workgroup, microarch guide "handoff-payload")
    if (S > 1) {
        constexpr int SLICE_BYTES = F * 1024;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(a.slabs + (size_t)tile * S * (F * 256), 0, S * SLICE_BYTES, 0x00020000);
#pragma unroll
        for (int q = 0; q < FW; ++q)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(rr_u32x4, sum[q]), rs, slab * SLICE_BYTES + ((wave * FW + q) * 64 + lane) * 16, 0,
                                                   /*sc1: write-through*/ 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        volatile int* flag = reinterpret_cast<volatile int*>(smem + G::FLAG_OFF);
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(a.tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = old == (unsigned)(S - 1);
            if (last) __hip_atomic_store(a.tickets + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *flag = last;
        }
        __syncthreads();
        RR_STAMP(7);
#ifdef PD_LAB_RR_STAMP
        if (*flag == 0 && tid == 0 && blockIdx.x < 256) { for (int k = 0; k < 16; ++k) g_rr_stamps[blockIdx.x * 16 + k] = lab_t[k]; }
#endif
        if (*flag == 0) return;
        constexpr int SB = FW >= 4 ? 4 : 8;                // slices per batch of loads (SB x FW <= 16 x 16-byte loads per lane in flight)
#pragma unroll
        for (int q = 0; q < FW; ++q) sum[q] = (float4_t){0.f, 0.f, 0.f, 0.f};
        for (int sp0 = 0; sp0 < S; sp0 += SB) {
            float4_t v[SB][FW];
#pragma unroll
            for (int uu = 0; uu < SB; ++uu) {
                const int sp = min(sp0 + uu, S - 1);       // clamped index + conditional add: no branch around a load
#pragma unroll
                for (int q = 0; q < FW; ++q)
                    v[uu][q] = __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, sp * SLICE_BYTES + ((wave * FW + q) * 64 + lane) * 16, 0, /*sc1*/ 16));
            }
#pragma unroll
            for (int uu = 0; uu < SB; ++uu)
#pragma unroll
                for (int q = 0; q < FW; ++q)
                    if (sp0 + uu < S) { sum[q][0] += v[uu][q][0]; sum[q][1] += v[uu][q][1]; sum[q][2] += v[uu][q][2]; sum[q][3] += v[uu][q][3]; }
        }
    }

    RR_STAMP(8);
    // ---- epilogue: sum + bias -> f16 -> LDS [pixel][CS_LD] -> 16-byte rows (+ residual) + GroupNorm octet partials of the band
    half_t* Cs = reinterpret_cast<half_t*>(smem);
#pragma unroll
    for (int q = 0; q < FW; ++q) {
        const int f = wave * FW + q, i = f / NF, j = f - i * NF;
        const int nl = j * 16 + (lane >> 4) * 4, ml = i * 16 + (lane & 15);
        float4_t bv = (float4_t){0.f, 0.f, 0.f, 0.f};
        if (a.bias != nullptr) bv = *reinterpret_cast<const float4_t*>(a.bias + n0 + nl);
        half4 h;
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = (half_t)(sum[q][r] + bv[r]);
        *reinterpret_cast<half4*>(&Cs[ml * CS_LD + nl]) = h;
    }
    __syncthreads();
    constexpr int CT = NF * 2;                             // column threads (one channel octet each)
    constexpr int RPP = 256 / CT;                          // rows per pass
    constexpr int MROWS = MT * 16;
    constexpr int PASSES = (MROWS + RPP - 1) / RPP;
    const int col8 = (tid % CT) * 8;
    const int Cout = a.Cout;
    float gs = 0.f, gq = 0.f;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int row = p * RPP + tid / CT;
        if (row < MROWS) {
            half8 v = *reinterpret_cast<const half8*>(&Cs[row * CS_LD + col8]);
            const int yy = r0 + row / W, xx = row % W;
            const size_t o = ((size_t)(img * H + yy) * W + xx) * Cout + n0 + col8;
            if (a.residual != nullptr) {
                size_t ro = o;
                if (a.res_up) ro = ((size_t)(img * (H >> 1) + (yy >> 1)) * (W >> 1) + (xx >> 1)) * Cout + n0 + col8;
                const half8 rv = *reinterpret_cast<const half8*>(a.residual + ro);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] + (float)rv[e]);
            }
            *reinterpret_cast<half8*>(a.Y + o) = v;
            float s8 = 0.f, q8 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; s8 += f; q8 += f * f; }
            gs += s8; gq += q8;
        }
    }
    if (a.gn_part != nullptr) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);       // [thread][2]
        red[tid * 2] = gs; red[tid * 2 + 1] = gq;
        __syncthreads();
        // two-level, fixed-order column sums: SUB threads per octet take every SUB-th row each, one thread adds the SUB sub-sums in order
        constexpr int RUSED = MROWS < RPP ? MROWS : RPP;   // rows that carried data
        constexpr int SUB = RUSED >= 8 ? 8 : RUSED;
        float* red2 = red + 2 * 256;
        if (tid < CT * SUB) {
            const int j = tid % SUB, c = tid / SUB;
            float s1 = 0.f, q1 = 0.f;
#pragma unroll
            for (int r = 0; r < RUSED / SUB; ++r) { s1 += red[((r * SUB + j) * CT + c) * 2]; q1 += red[((r * SUB + j) * CT + c) * 2 + 1]; }
            red2[tid * 2] = s1; red2[tid * 2 + 1] = q1;
        }
        __syncthreads();
        if (tid < CT) {
            float s1 = 0.f, q1 = 0.f;
#pragma unroll
            for (int j = 0; j < SUB; ++j) { s1 += red2[(tid * SUB + j) * 2]; q1 += red2[(tid * SUB + j) * 2 + 1]; }
            float* dst = a.gn_part + (((size_t)img * a.bands + band) * (Cout >> 3) + (n0 >> 3) + tid) * 2;
            dst[0] = s1; dst[1] = q1;
        }
    }
    RR_STAMP(9);
#ifdef PD_LAB_RR_STAMP
    if (tid == 0 && blockIdx.x < 256) { for (int k = 0; k < 16; ++k) g_rr_stamps[blockIdx.x * 16 + k] = lab_t[k]; }
#endif
}

template <int W, int MT, int NF, int CS, bool GNF = true>
int launch_rr(const RrArgs& a, int grid, hipStream_t s) {
    using G = RrCfg<W, MT, NF, CS>;
    PD_REQUIRE(GNF || (a.src[0].gn == 0 && a.src[1].gn == 0), "conv_rr: this tile variant has no in-staging GroupNorm");
    auto kern = k_conv_rr<W, MT, NF, CS, GNF>;
    if (G::SMEM > 65536) PD_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::SMEM));
    kern<<<grid, 256, G::SMEM, s>>>(a);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}

// [Cout_pad][K] (k = tap * Cin + c for the first taps * Cin columns, then the skip 1x1's Cs columns) -> fragment-major
// [Cout / 16][KS][lane][8]: K-step ks < taps * Cin / 32 is (32-channel step ks / taps, tap ks % taps), the skip's steps follow
__global__ void k_pack_rr(const half_t* __restrict__ src, int K, int Cin, int taps, int Cs, int n16, half_t* __restrict__ dst) {
    const int KS = taps * (Cin >> 5) + (Cs >> 5);
    const long long total = (long long)n16 * KS * 64;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const long long r = i >> 6;
        const int ks = (int)(r % KS), nb = (int)(r / KS);
        int k;
        if (ks < taps * (Cin >> 5)) { const int c32 = ks / taps, t = ks - c32 * taps; k = t * Cin + c32 * 32; }
        else k = taps * Cin + (ks - taps * (Cin >> 5)) * 32;
        *reinterpret_cast<half8*>(dst + i * 8) = *reinterpret_cast<const half8*>(src + (size_t)(nb * 16 + (lane & 15)) * K + k + (lane >> 4) * 8);
    }
}

}  // namespace

thread_local int g_rr_mode = 1;        // tuning / test hook (pdhip_debug_set_conv_rr): 0 = never, 1 = automatic, 2 = every eligible layer
thread_local int g_rr_variant = 0;     // 0 = automatic, else force a variant id (see conv_rr_plan)
thread_local int g_rr_slabs = 0;       // 0 = automatic, else force the slab count of the conv source

size_t conv_rr_weight_halfs(int Cin, int taps, int Cs, int Cout) { return (size_t)(Cout / 16) * (taps * (Cin / 32) + Cs / 32) * 512; }

int conv_rr_pack(const half_t* w_packed, int Cin, int taps, int Cs, int Cout, half_t* dst, hipStream_t s) {
    PD_REQUIRE(Cin % 32 == 0 && Cs % 32 == 0 && Cout % 16 == 0 && (taps == 1 || taps == 9), "conv_rr_pack: bad shape");
    const long long total = (long long)(Cout / 16) * (taps * (Cin / 32) + Cs / 32) * 64;
    k_pack_rr<<<(int)std::min<long long>((total + 255) / 256, 8192), 256, 0, s>>>(w_packed, taps * Cin + Cs, Cin, taps, Cs, Cout / 16, dst);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}

// variants: id -> (W, MT, NF, CS)
//   1: W = 8,  whole image (64 pixels),   16 channels, 256-channel units
//   2: W = 16, half image (128 pixels),   32 channels, 128-channel units
//   3: W = 32, 4-row bands (128 pixels),  32 channels, 128-channel units
//   5: W = 8,  whole image,               16 channels, 128-channel units
//   6: W = 32, 4-row bands,               16 channels, 128-channel units (twice the tiles: the 32^2 layers can run unsplit)
//   7: W = 16, half image,                16 channels, 128-channel units
//   8: W = 64, 4-row bands (256 pixels),  32 channels, 128-channel units
//   (W = 128, 2-row bands: measured 40.6 us against k_conv_sk's 25.8 at 128^2 / 256 -> 256 and 34 spilled registers -- not kept)
static bool rr_variant(int v, int* w, int* mt, int* nf, int* cs) {
    switch (v) {
        case 1: *w = 8; *mt = 4; *nf = 1; *cs = 256; return true;
        case 2: *w = 16; *mt = 8; *nf = 2; *cs = 128; return true;
        case 3: *w = 32; *mt = 8; *nf = 2; *cs = 128; return true;
        case 5: *w = 8; *mt = 4; *nf = 1; *cs = 128; return true;
        case 6: *w = 32; *mt = 8; *nf = 1; *cs = 128; return true;
        case 7: *w = 16; *mt = 8; *nf = 1; *cs = 128; return true;
        case 8: *w = 64; *mt = 16; *nf = 2; *cs = 128; return true;
    }
    return false;
}
RrPlan conv_rr_plan(int N, int H, int W, int Cin, int Cout, int taps, int Cs, size_t ws_floats, bool want_gn) {
    RrPlan p{0, 0, 0, 0, 0, 0, 0};
    if (g_rr_mode == 0 || H != W || (taps != 9 && taps != 1)) return p;
    // Candidate tile variants of the image width, smallest channel tile first.  Rule read off tools/bench_rr.py (profiles/r06_rr_bench_standalone.txt):
    // the variant whose tiles come closest to one workgroup per CU with the FEWEST K slabs wins -- every slab beyond the first costs a
    // write-through publish, a ticket round trip and a re-read of the slices by the last arriver (~4 us of a 10 us launch).
    static const int cand8[] = {1, 0}, cand16[] = {7, 2, 0}, cand32[] = {6, 3, 0}, cand64[] = {8, 0};
    const int* cand = W == 8 ? cand8 : W == 16 ? cand16 : W == 32 ? cand32 : W == 64 ? cand64 : nullptr;
    if (cand == nullptr) return p;
    int v = g_rr_variant, vw = 0, mt = 0, nf = 0, cs = 0;
    if (v == 0) {
        for (int k = 0; cand[k] != 0; ++k) {
            int w_, mt_, nf_, cs_;
            rr_variant(cand[k], &w_, &mt_, &nf_, &cs_);
            if (want_gn && (cand[k] == 6 || cand[k] == 7 || cand[k] >= 8)) continue;      // (the variants without the in-staging GroupNorm)
            if (Cin % cs_ != 0 || Cs % cs_ != 0 || Cout % (nf_ * 16) != 0) continue;
            v = cand[k];
            if ((long long)N * (H * W / (mt_ * 16)) * (Cout / (nf_ * 16)) <= 256) break;        // fits one round of workgroups: take it; else try the next (larger) tile
        }
    }
    if (!rr_variant(v, &vw, &mt, &nf, &cs) || vw != W) return p;
    if (Cin % cs != 0 || Cs % cs != 0 || Cout % (nf * 16) != 0) return p;
    const int bands = H * W / (mt * 16);
    const int ntiles = Cout / (nf * 16);
    const long long tiles = (long long)N * bands * ntiles;
    if (tiles > 4096) return p;
    // automatic routing: the launches a same-box A/B of the DDNM step favours (profiles/r06_rr_ab.txt): batch 1-2 at 8^2 ... 32^2 (batch 4 at
    // 8^2 only: from batch 4 up k_conv_sk's larger tiles win back what the slabs cost).  The 64^2 level at batch 1 (512+-channel layers: 27-30 /
    // 40 / 51 us at 512 / 768 / 1 024 input channels) went to k_conv_ht's two-slab form when that arrived (27 / 34 / 42 us, profiles/r06_ht_bench.txt)
    if (g_rr_mode != 2) {
        // (with an appended skip 1x1 the slabs of 1x1 units are the long pole: 16-27 us against k_conv_sk<10>'s 15-23 at every level -- not routed)
        const bool ok = Cs == 0 && W <= 32 && (N <= 2 || (N <= 4 && W == 8));
        if (!ok) return p;
    }
    // slabs: enough workgroups to put ~one on every CU, every slab a whole number of units, at most 8 conv slabs (the last arriver re-reads them all)
    const int u0 = Cin / cs, u1 = Cs / cs;
    int s0 = g_rr_slabs > 0 ? g_rr_slabs : (int)std::max<long long>(1, (256 + tiles / 2) / tiles);
    s0 = std::min(std::min(s0, u0), g_rr_slabs > 0 ? 64 : (mt * nf >= 16 ? 4 : 8));      // the last arriver re-reads S x (MT x NF) KB at ~65 GB/s: <= 64 KB
    while (u0 % s0 != 0) --s0;
    int s1 = 0;
    if (u1 > 0) {
        // the skip's K per slab ~ the conv's K per slab (9 taps x u0 / s0 units): u1 / s1 ~ 9 u0 / s0.  (Cutting it by UNIT count instead --
        // a 1x1 unit costs two barriers and a staging round trip like a 3x3 one -- measured worse: 12 slabs to combine at the 8^2 level.)
        s1 = std::max(1, std::min(u1, (int)((long long)u1 * s0 / ((long long)taps * u0) + 1)));
        while (u1 % s1 != 0) --s1;
    }
    if ((size_t)tiles * (s0 + s1) * mt * nf * 256 + PD_SK_TICKET_FLOATS > ws_floats && s0 + s1 > 1) return p;
    p.variant = v; p.S0 = s0; p.U0 = u0 / s0; p.S1 = s1; p.U1 = s1 ? u1 / s1 : 0; p.bands = bands; p.ntiles = ntiles;
    return p;
}

int conv_rr(const RrPlan& pl, const RrIn& in, const RrIn* skip, int taps, const half_t* wf, const float* bias, const half_t* residual, int res_up,
            half_t* Y, int N, int H, int W, int Cout, float* ws, size_t ws_floats, float* gn_part, int* gn_chunks, hipStream_t s) {
    PD_REQUIRE(pl.variant > 0 && wf != nullptr && Y != nullptr && in.x != nullptr, "conv_rr: no plan / null argument");
    PD_REQUIRE((skip == nullptr) == (pl.S1 == 0), "conv_rr: the plan and the skip source disagree");
    PD_REQUIRE(res_up == 0 || (residual != nullptr && H % 2 == 0 && W % 2 == 0), "conv_rr: an up-sampled residual needs even H, W");
    auto fill = [](const RrIn& i, RrSrcK& k) -> int {
        k.x = i.x; k.x2 = i.x2; k.C = i.C; k.Ca = i.x2 ? i.Ca : i.C; k.gn = i.gn; k.gamma = i.gamma; k.beta = i.beta; k.film = i.film;
        k.film_stride = i.film_stride; k.partA = i.partA; k.partB = i.partB; k.chunksA = i.chunksA; k.chunksB = i.chunksB; k.eps = i.eps;
        k.cg = i.C / 32; k.cg_magic = ((1 << 20) + k.cg - 1) / k.cg; k.opg = k.cg >> 3;
        const int pps = std::max(1, 256 / (i.C >> 3));
        k.pps_log2 = 0; while ((1 << (k.pps_log2 + 1)) <= pps) ++k.pps_log2;
        PD_REQUIRE(i.gn == 0 || ((1 << k.pps_log2) == pps && k.opg * pps <= 8 && i.C <= 4096 &&
                                 (i.chunksA + pps - 1) / pps <= RR_MAXQ && (i.x2 == nullptr || (i.chunksB + pps - 1) / pps <= RR_MAXQ)),
                   "conv_rr: GroupNorm input outside the in-kernel statistics' range (C = %d, chunks %d / %d)", i.C, i.chunksA, i.chunksB);
        PD_REQUIRE(i.gn == 0 || (i.gamma && i.beta && i.partA && ((i.C / 32) % 8) == 0 && (i.C >> 3) <= 256 && (i.x2 == nullptr || i.partB)),
                   "conv_rr: GroupNorm input needs gamma / beta / octet partials and a group size that is a multiple of 8 channels");
        PD_REQUIRE(i.x2 == nullptr || (i.Ca > 0 && i.Ca < i.C), "conv_rr: bad two-source split");
        return PDHIP_OK;
    };
    RrArgs a{};
    RR_TRY(fill(in, a.src[0]));
    if (skip) RR_TRY(fill(*skip, a.src[1]));
    a.taps0 = taps; a.S0 = pl.S0; a.U0 = pl.U0; a.S1 = pl.S1; a.U1 = pl.U1;
    a.wf = wf; a.KS = taps * (in.C / 32) + (skip ? skip->C / 32 : 0);
    a.bias = bias; a.residual = residual; a.res_up = res_up; a.Y = Y;
    a.N = N; a.H = H; a.Cout = Cout; a.ntiles = pl.ntiles; a.bands = pl.bands;
    const int S = pl.S0 + pl.S1;
    const long long tiles = (long long)N * pl.bands * pl.ntiles;
    PD_REQUIRE(tiles <= 4096, "conv_rr: too many tiles for the ticket table");
    a.tickets = reinterpret_cast<unsigned*>(ws);
    a.slabs = ws ? ws + PD_SK_TICKET_FLOATS : nullptr;
    a.gn_part = gn_part;
    if (gn_chunks) *gn_chunks = gn_part ? pl.bands : 0;
    const int grid = (int)(tiles * S);
    int vw = 0, cs = 0, mt = 0, nf = 0;
    PD_REQUIRE(rr_variant(pl.variant, &vw, &mt, &nf, &cs) && vw == W, "conv_rr: variant %d does not serve %d-wide images", pl.variant, W);
    PD_REQUIRE(S == 1 || (ws != nullptr && (size_t)tiles * S * mt * nf * 256 + PD_SK_TICKET_FLOATS <= ws_floats), "conv_rr: split-K workspace too small");
    PD_REQUIRE(in.C % cs == 0 && (in.x2 == nullptr || in.Ca % cs == 0) && (!skip || (skip->C % cs == 0 && (skip->x2 == nullptr || skip->Ca % cs == 0))),
               "conv_rr: channel counts must be multiples of the unit size");
    switch (pl.variant) {
        case 1: return launch_rr<8, 4, 1, 256>(a, grid, s);
        case 2: return launch_rr<16, 8, 2, 128>(a, grid, s);
        case 3: return launch_rr<32, 8, 2, 128>(a, grid, s);
        case 5: return launch_rr<8, 4, 1, 128>(a, grid, s);
        case 6: return launch_rr<32, 8, 1, 128, false>(a, grid, s);
        case 7: return launch_rr<16, 8, 1, 128, false>(a, grid, s);
        case 8: return launch_rr<64, 16, 2, 128, false>(a, grid, s);
    }
    set_error("conv_rr: unknown variant %d", pl.variant);
    return PDHIP_E_ARG;
}

}  // namespace pdnn

// =================================================================================================
// stand-alone surface (unit tests, tools/bench_rr.py)
namespace {
// octet partials of X [N,HW,C] as a conv epilogue leaves them: part[((n * chunks + c) * (C / 8) + o) * 2] = (sum, sum of squares) of the
// f16 values of chunk c's pixels, channels 8 o .. 8 o + 7 (test helper: builds the GroupNorm input of pdhip_conv_rr_f16)
__global__ void k_gn_octet_partials(const pdnn::half_t* __restrict__ X, int HW, int C, int chunks, float* __restrict__ part) {
    const int n = blockIdx.y, c = blockIdx.x, oc = C >> 3, ppc = HW / chunks;
    for (int o = threadIdx.x; o < oc; o += blockDim.x) {
        float s1 = 0.f, q1 = 0.f;
        for (int p = c * ppc; p < (c + 1) * ppc; ++p) {
            const pdnn::half8 v = *reinterpret_cast<const pdnn::half8*>(X + ((size_t)n * HW + p) * C + o * 8);
            float s8 = 0.f, q8 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; s8 += f; q8 += f * f; }
            s1 += s8; q1 += q8;
        }
        part[(((size_t)n * chunks + c) * oc + o) * 2] = s1;
        part[(((size_t)n * chunks + c) * oc + o) * 2 + 1] = q1;
    }
}
}  // namespace

#ifdef PD_LAB_RR_STAMP
extern "C" int pdhip_lab_rr_read_stamps(unsigned long long* host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(pdnn::g_rr_stamps), sizeof(unsigned long long) * n); }
#endif
extern "C" int pdhip_debug_set_conv_rr(int mode, int variant, int slabs) {
    const int old = pdnn::g_rr_mode;
    pdnn::g_rr_mode = mode; pdnn::g_rr_variant = variant; pdnn::g_rr_slabs = slabs;
    return old;
}
extern "C" long long pdhip_conv_rr_weight_halfs(int Cin, int taps, int Cs, int Cout) { return (long long)pdnn::conv_rr_weight_halfs(Cin, taps, Cs, Cout); }
extern "C" int pdhip_conv_rr_pack_f16(const void* w_packed, int Cin, int taps, int Cs, int Cout, void* wf, void* stream) {
    PD_REQUIRE(w_packed && wf, "pdhip_conv_rr_pack_f16: null argument");
    return pdnn::conv_rr_pack((const pdnn::half_t*)w_packed, Cin, taps, Cs, Cout, (pdnn::half_t*)wf, as_stream(stream));
}
extern "C" int pdhip_gn_octet_partials_f16(const void* x, int N, int HW, int C, int chunks, float* part, void* stream) {
    PD_REQUIRE(x && part && C % 8 == 0 && chunks >= 1 && HW % chunks == 0, "pdhip_gn_octet_partials_f16: bad arguments");
    k_gn_octet_partials<<<dim3(chunks, N), 256, 0, as_stream(stream)>>>((const pdnn::half_t*)x, HW, C, chunks, part);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}
extern "C" int pdhip_gn_apply_parts_f16(const void* x, const void* x2, int Ca, int C, const float* partA, int chunksA, const float* partB, int chunksB,
                                        const float* gamma, const float* beta, const float* film, long long film_stride, int N, int H, int W, int silu,
                                        void* y, void* stream) {
    PD_REQUIRE(x && partA && gamma && beta && y, "pdhip_gn_apply_parts_f16: null argument");
    const pdnn::GnPartsArg pa{partA, x2 ? Ca : C, chunksA, partB, x2 ? C - Ca : 0, chunksB, 1e-5f};
    return pdnn::gn_apply((const pdnn::half_t*)x, nullptr, gamma, beta, film, film_stride, N, H, W, C, silu, 0, y, 0, as_stream(stream),
                          (const pdnn::half_t*)x2, x2 ? Ca : 0, nullptr, &pa);
}
extern "C" int pdhip_conv_rr_f16(const void* x, const void* x2, int C, int Ca, int gn_mode, const float* gamma, const float* beta, const float* film,
                                 long long film_stride, const float* partA, int chunksA, const float* partB, int chunksB, const void* xs,
                                 const void* xs2, int Cs, int Cs1, int taps, const void* wf, const float* bias, const void* residual, int res_up,
                                 void* y, int N, int H, int W, int Cout, float* ws, long long ws_floats, float* gn_part, int* gn_chunks,
                                 void* stream) {
    PD_REQUIRE(x && wf && y, "pdhip_conv_rr_f16: null argument");
    const pdnn::RrPlan pl = pdnn::conv_rr_plan(N, H, W, C, Cout, taps, xs ? Cs : 0, ws ? (size_t)ws_floats : 0, gn_mode != 0);
    PD_REQUIRE(pl.variant != 0, "pdhip_conv_rr_f16: not a layer for the row-resident kernel (N=%d H=%d W=%d Cin=%d Cout=%d taps=%d Cs=%d)", N, H, W, C, Cout, taps, xs ? Cs : 0);
    pdnn::RrIn in{(const pdnn::half_t*)x, (const pdnn::half_t*)x2, C, x2 ? Ca : C, gn_mode, gamma, beta, film, film_stride, partA, partB, chunksA, chunksB, 1e-5f};
    pdnn::RrIn sk{(const pdnn::half_t*)xs, (const pdnn::half_t*)xs2, Cs, xs2 ? Cs1 : Cs, 0, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0, 0, 1e-5f};
    return pdnn::conv_rr(pl, in, xs ? &sk : nullptr, taps, (const pdnn::half_t*)wf, bias, (const pdnn::half_t*)residual, res_up, (pdnn::half_t*)y, N, H, W,
                         Cout, ws, ws ? (size_t)ws_floats : 0, gn_part, gn_chunks, as_stream(stream));
}

