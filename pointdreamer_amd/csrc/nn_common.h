// Shared declarations of the UNet engine (row U1/D1) translation units.
#pragma once
#include "common.h"
#include <hip/hip_fp16.h>

namespace pdnn {

typedef _Float16 half_t;
typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(4))) _Float16 half4;
typedef __attribute__((ext_vector_type(4))) float float4_t;

// ---- implicit-GEMM convolution (nn_gemm.hip)
// X [N,H,W,Cin] f16 (NHWC), Wt [Cout_pad][TAPS*Cin] f16 (k = tap*Cin + c, tap = ky*3+kx; Cout_pad % 128 == 0),
// bias f32 [Cout] (may be null), residual [N,H,W,Cout] f16 (may be null), Y [N,H,W,Cout] f16.
// taps: 1 (1x1 conv / linear over pixels) or 9 (3x3, pad 1).  Cin % 32 == 0.
// 3x3 convolution with an LDS-resident activation halo (nn_conv_halo.hip): 512-pixel x 128-channel tiles, W in {64,128,256}
bool conv3x3_halo_eligible(int N, int H, int W, int Cin, int Cout_pad);
int conv3x3_halo_splits(int N, int H, int W, int Cin, int Cout, int Cout_pad, size_t splitk_ws_floats);   // 1 direct, >1 split, 0 = do not use
bool conv_uses_halo(int N, int H, int W, int Cin, int Cout, int Cout_pad, int taps, size_t splitk_ws_floats);   // conv_igemm's routing (nn_gemm.hip)
// apply_table != NULL: the input is silu(A x + B) of X per gn_table (zero padding applies to the TRANSFORMED image)
int conv3x3_halo(const half_t* X, const half_t* Wt, const float* bias, const half_t* residual, half_t* Y, int N, int H, int W,
                 int Cin, int Cout, int Cout_pad, const half_t* zero_page, hipStream_t s, float* gn_part, int* gn_fused,
                 float* splitk_ws, size_t splitk_ws_floats, const float* apply_table = nullptr, int res_up = 0,    // res_up: residual = half-resolution tensor, nearest x2 on the fly
                 int in_up = 0);                                          // in_up: X = half-resolution tensor, the conv sees its nearest x2
// fixed-order sum of split-K partials [splits][M][Cout] f32 + bias (+ residual) -> f16 Y, optional GroupNorm octet partials
int splitk_reduce(const float* partial, int splits, long long M, int Cout, const float* bias, const half_t* residual, half_t* Y,
                  float* gn_part, int hw, hipStream_t s);
extern thread_local int g_force_bk, g_force_stages, g_force_wmw, g_force_splits;     // tuning hooks (nn_gemm.hip)
extern thread_local int g_fuse_gn, g_fold_resample, g_fold_finalize, g_fuse_skip, g_fold_skip;                                                   // tuning hook (nn_unet.hip)
extern thread_local float* g_dbg_splitk_ws; extern thread_local size_t g_dbg_splitk_floats;
int conv_igemm(const half_t* X, const half_t* Wt, const float* bias, const half_t* residual, half_t* Y, int N, int H,
               int W, int Cin, int Cout, int Cout_pad, int taps, const half_t* zero_page, hipStream_t s,
               float* splitk_ws = nullptr, size_t splitk_ws_floats = 0, float* gn_part = nullptr, int* gn_fused = nullptr,
               const half_t* X2 = nullptr, int Cin1 = 0,    // X2: second tensor of a never-materialised channel concat (1x1 only)
               const float* apply_table = nullptr,          // input = silu(A x + B) per gn_table (layers the halo kernel takes only)
               int res_up = 0,                              // residual = half-resolution tensor read with nearest x2 (unsplit halo layers only)
               int in_up = 0);                              // X = half-resolution tensor, input = its nearest x2 (halo layers only)
// ---- small-M convolution with in-launch split-K combine (nn_conv_sk.hip).  ws: [0, 4096) ticket words (zero at allocation,
// self-resetting), slabs behind them.  conv_sk_plan decides whether / how a layer runs there (bm == 0: not this kernel's layer).
struct SkPlan { int bm, bn, splits, tile_id; };
SkPlan conv_sk_plan(int N, int H, int W, int Cin, int Cout, int Cout_pad, int taps, bool two_source, size_t ws_floats, int k_extra = 0);
// 3x3 conv + the ResBlock's skip 1x1 over the block input xs = [XS (Cs1 channels) | XS2 (Cs - Cs1), may be null] as ONE K loop
// (out = W2 * im2col(X) + Wskip * xs + (b2 + bskip): unet.py:255); Wt [Cout_pad][9 Cin + Cs] and bias from fuse_skip_weights
int conv_sk_skip(const SkPlan& pl, const half_t* X, const half_t* Wt, const float* bias, half_t* Y, int N, int H, int W, int Cin, int Cout,
                 int Cout_pad, const half_t* XS, const half_t* XS2, int Cs1, int Cs, const half_t* zero_page, hipStream_t s, float* ws,
                 size_t ws_floats, float* gn_part, int* gn_fused);
int fuse_skip_weights(const half_t* w3, int K9, const half_t* w1, int Cs, int Cout_pad, const float* b3, const float* b1, int Cout, half_t* dst,
                      float* bdst, hipStream_t s);
int conv_sk(const SkPlan& pl, const half_t* X, const half_t* Wt, const float* bias, const half_t* residual, half_t* Y, int N, int H, int W,
            int Cin, int Cout, int Cout_pad, int taps, const half_t* zero_page, hipStream_t s, float* ws, size_t ws_floats, float* gn_part,
            int* gn_fused, const half_t* X2, int Cin1, int res_up = 0);      // res_up: residual = half-resolution tensor read with nearest x2
// conv_igemm's routing decision for a single-source layer without input transform: does it go to k_conv_sk? (nn_gemm.hip)
bool conv_routes_small(int N, int H, int W, int Cin, int Cout, int Cout_pad, int taps, size_t splitk_ws_floats);
extern thread_local int g_sk_mode, g_sk_tile, g_sk_splits, g_sk_stages, g_sk_kg, g_sk_order;                         // tuning / test hooks (pdhip_debug_set_conv_sk)
#define PD_SK_TICKET_FLOATS 4096
// combine per-(chunk, channel-octet) partial sums written by the conv epilogue ([N][chunks][C/8][2]) of one tensor, or of
// the two tensors of a channel concat (A: Ca channels, B: Cb channels), into GroupNorm(32) stats [N][32][2] (mean, rstd).
int gn_finalize_oct(const float* partA, int Ca, int chunksA, const float* partB, int Cb, int chunksB, int N, int HW, float eps,
                    float* stats, hipStream_t s);

// ---- row-resident convolution of the 8^2 / 16^2 / 32^2 levels at small batch (nn_conv_rr.hip): GroupNorm (+ FiLM) (+ SiLU) of the input
// applied while staging, weights streamed HBM -> registers from a fragment-major copy, in-launch split-K over K "slabs".
struct RrIn {                      // one K source: x [N,H,W,Ca] (+ x2 [N,H,W,C-Ca]: never-materialised channel concat)
    const half_t* x; const half_t* x2; int C, Ca;
    int gn;                        // 0 raw, 1 GroupNorm, 2 GroupNorm + SiLU (statistics from the producers' octet partials, k_gn_apply's FIN arithmetic)
    const float* gamma; const float* beta; const float* film; long long film_stride;
    const float* partA; const float* partB; int chunksA, chunksB; float eps;
};
struct RrPlan { int variant, S0, U0, S1, U1, bands, ntiles; };      // variant == 0: not a layer for this kernel
RrPlan conv_rr_plan(int N, int H, int W, int Cin, int Cout, int taps, int Cs /*channels of an appended skip 1x1, 0 = none*/, size_t ws_floats,
                    bool want_gn = false /*the caller will ask for the in-staging GroupNorm: only the variants that carry it*/);
size_t conv_rr_weight_halfs(int Cin, int taps, int Cs, int Cout);
// w_packed: the engine's [Cout_pad][taps * Cin (+ Cs)] layout -> fragment-major [Cout / 16][K-steps][64 lanes][8]
int conv_rr_pack(const half_t* w_packed, int Cin, int taps, int Cs, int Cout, half_t* dst, hipStream_t s);
int conv_rr(const RrPlan& pl, const RrIn& in, const RrIn* skip, int taps, const half_t* wf, const float* bias, const half_t* residual, int res_up,
            half_t* Y, int N, int H, int W, int Cout, float* ws, size_t ws_floats, float* gn_part, int* gn_chunks, hipStream_t s);
extern thread_local int g_rr_mode, g_rr_variant, g_rr_slabs;            // tuning / test hooks (pdhip_debug_set_conv_rr)
#define PD_RR_MAX_PART_LOADS 8    // octet-partial loads per thread of the in-kernel statistics: chunks / slots of a source must not exceed it

// ---- 3x3 conv on 256-pixel x 64-channel halo tiles, K unsplit (nn_conv_ht.hip): the 64^2 / 128^2 levels at batch 1-2
bool conv_ht_routes(int N, int H, int W, int Cin, int Cout, int Cout_pad, size_t ws_floats);
int conv_ht_slabs(int N, int H, int W, int Cin, int Cout_pad, size_t ws_floats);
int conv_ht(const half_t* X, const half_t* Wt, const float* bias, const half_t* residual, half_t* Y, int N, int H, int W, int Cin, int Cout,
            int Cout_pad, const half_t* zero_page, hipStream_t s, float* gn_part, int* gn_chunks, int res_up, float* ws, size_t ws_floats);
extern thread_local int g_ht_mode;

// ---- the GroupNorm (+ FiLM) (+ SiLU) element map (nn_norm.hip's k_gn_apply and nn_conv_rr.hip's staging: ONE definition, bit-identical results)
// x * sigmoid(x); v_rcp_f32 (1 ulp) instead of the IEEE divide sequence: the result is rounded to f16 (or feeds the f32 head,
// tolerance 1e-3) and the divide was half of this HBM-bound kernel's VALU work
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// One element of GroupNorm (+ FiLM) (+ SiLU) with the reference's f16 tensor between the ops.  Every f32 result is made opaque
// before it is rounded to f16: left alone, the compiler fuses an op with the conversion behind it (v_fma_mixlo_f16: ONE rounding) in
// some copies of the loop and not in others -- the 4-pixel main loop and the 1-pixel tail of k_gn_apply differed by one f16 ulp in
// 4e-5 of the elements, so a result depended on the pixels-per-thread a launch happened to pick.
__device__ __forceinline__ float gn_round_f16(float f) { asm("" : "+v"(f)); return (float)(half_t)f; }
template <bool OUT_F32, bool FILM>
__device__ __forceinline__ float gn_elem(float x, float ga, float gb, float t1, float sh, int silu) {
    float f = __builtin_fmaf(x, ga, gb);
    if (!OUT_F32) f = gn_round_f16(f);                                   // GroupNorm32 returns x.dtype (f16)
    if (FILM) {
        f = gn_round_f16(f * t1);
        f = gn_round_f16(f + sh);
    }
    if (silu) { f = silu_f(f); if (!OUT_F32) f = gn_round_f16(f); }
    return f;
}


// ---- normalisation / elementwise (nn_norm.hip)
// GroupNorm(32) statistics of X [N,HW,C] f16 -> stats [N][32][2] (mean, rstd) f32.  ws: N*chunks*32*2 floats.
int gn_stats(const half_t* X, int N, int HW, int C, float eps, float* stats, float* ws, size_t ws_floats, hipStream_t s);
// y = silu?( GN(x)*gamma+beta [*(1+scale)+shift] ), optional 2x resample; RESAMPLE: 0 none, 1 avgpool2, 2 nearest-up2.
// film: rows of (scale[C] | shift[C]) f32, row n at film + n*film_stride, or null.  Output f16 NHWC (or f32 when out_f32).
// parts != NULL (stats == NULL): the statistics are reduced inside the apply kernel from the producing convs' octet partials
// (what gn_finalize_oct would read) -- the small-batch form, one launch less per GroupNorm
struct GnPartsArg { const float* partA; int Ca, chunksA; const float* partB; int Cb, chunksB; float eps; };
int gn_apply(const half_t* X, const float* stats, const float* gamma, const float* beta, const float* film,
             long long film_stride, int N, int H, int W, int C, int silu, int resample, void* Y, int out_f32, hipStream_t s,
             const half_t* XB = nullptr, int Ca = 0,      // XB: second tensor of a never-materialised channel concat
             half_t* Yraw = nullptr,                      // resample 1 only: also AvgPool2d(2) of the RAW input (the x branch of a down-ResBlock)
             const GnPartsArg* parts = nullptr);
// GroupNorm (+ FiLM) as one affine map per (image, channel): table [N][C/8][16] = (A0..A7, B0..B7) per channel octet, y = silu(A x + B) -- the
// input transform of the APPLY variant of the halo conv (the stand-alone gn_apply pass disappears)
int gn_table(const float* stats, const float* gamma, const float* beta, const float* film, long long film_stride, int N, int C,
             float* table, hipStream_t s);
// GroupNorm-apply (+ SiLU) of a ResBlock input AND the block's skip 1x1 conv (C -> 256) in one pass over x = [XA | XB] (channel concat,
// XB may be null when Ca == C): H0 [N,HW,C] = silu(GN(x)), SK [N,HW,256] = f16(Wt x + bias).  stats [N][32][2] finished.
bool gn_skip_eligible(int N, int HW, int Ca, int C, int Cout, int Cout_pad);
int gn_skip(const half_t* XA, const half_t* XB, int Ca, int C, const float* stats, const float* gamma, const float* beta, const half_t* Wt,
            const float* bias, half_t* H0, half_t* SK, int N, int HW, hipStream_t s);
int resample2x(const half_t* X, int N, int H, int W, int C, int mode, half_t* Y, hipStream_t s);
int concat_channels(const half_t* A, int Ca, const half_t* B, int Cb, long long pixels, half_t* Y, hipStream_t s);

// ---- attention (nn_attn.hip): QKVAttentionLegacy on qkv [N,T,3C] f16 (per head h: q|k|v at channel h*3*D), out [N,T,C].
int attention(const half_t* qkv, half_t* out, int N, int T, int C, int D, hipStream_t s, half_t* vt_ws = nullptr);   // vt_ws: N*T*C halfs (transposed V)

// ---- small dense ops (nn_misc.hip)
int conv_in_3x3(const float* x_nchw, const half_t* Wt /*[Cout_pad][32] k=(ky*3+kx)*3+c, k>=27 zero*/, const float* bias, half_t* Y,
                int N, int H, int W, int Cout, int Cout_pad, half_t* im2col_ws /*[N*H*W][32]*/, const half_t* zero_page, hipStream_t s, float* gn_part = nullptr, int* gn_fused = nullptr);
// output head (nn_head.hip): GroupNorm affine -> SiLU -> conv3x3 (C -> 3|6) in f32-equivalent arithmetic, y f32 NCHW.
// wz = head_pack(w f32 [NO][9*C] (k = tap*C + c)): [2][64][C] f16 hi/lo.
int head_pack(const float* w, int NO, int C, half_t* wz, hipStream_t s);
int head_gn_silu_conv3x3(const half_t* X, const float* stats, const float* gamma, const float* beta, const half_t* wz,
                         const float* bias, float* y_nchw, int N, int H, int W, int C, int NO, hipStream_t s);
int timestep_mlp(const float* t, int N, int mc, const float* w0, const float* b0, const float* w2, const float* b2,
                 float* emb_silu /*[N][4mc] = silu(time_embed(t))*/, float* tmp, hipStream_t s);
int gemv_rows(const float* Wm /*[R][K]*/, const float* b, const float* x /*[N][K]*/, float* y /*[N][R]*/, int R, int K, int N,
              hipStream_t s);

// ---- DDNM (nn_misc.hip)
struct DdnmCoef { float sqrt_1m_at, sqrt_at, sqrt_at_next, sigma_t, c1, c2; };
int ddnm_update(float* x /*[N,3,HW] in/out*/, const float* et /*[N,Cet,HW], first 3 used*/, int Cet, const float* y,
                const float* mask /*[N,HW]*/, const float* eps /*null -> philox*/, unsigned long long seed,
                unsigned long long step, DdnmCoef co, int N, int HW, hipStream_t s, unsigned long long quad0 = 0);
int ddnm_prepare(const float* masked_img, const float* mask, float* y, int N, int HW, hipStream_t s);
int ddnm_finish(const float* x, float* out, long long n, hipStream_t s);
// Philox counter = (quad0 + element / 4, stream_id), key = seed.  quad0 = first image key * quads per image makes the noise of an
// image a function of its KEY only (not of its position in the batch or of how views are sharded over ranks).
extern thread_local int g_gn_iters;
int copy16(const void* src, void* dst, long long bytes, int blocks, int unroll, hipStream_t s);   // calibration copy (nn_misc.hip)
int philox_normal(float* out, long long n, unsigned long long seed, unsigned long long stream_id, hipStream_t s,
                  unsigned long long quad0 = 0);

}  // namespace pdnn
