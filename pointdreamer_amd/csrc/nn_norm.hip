// Row U1: GroupNorm32 statistics / apply (+SiLU, FiLM scale-shift, 2x resample), channel concat.
// Replaces nn.GroupNorm (GroupNorm32, nn.py:17-19), nn.SiLU, the scale-shift conditioning of
// ResBlock._forward (unet.py:248-252), Upsample/Downsample without conv (unet.py:100-140) and th.cat
// (unet.py:661).  All HBM-bound streaming kernels over NHWC f16 with 16-byte accesses.
// Rounding points follow the reference's fp16 torso: GN output -> f16, each FiLM op -> f16, SiLU -> f16.
#include "nn_common.h"
using namespace pdhip;
namespace pdnn {

#define GN_PIX_PER_BLOCK 256

// partial sums: grid (chunks, N); each block covers GN_PIX_PER_BLOCK pixels x all channels.
// thread -> (pixel sub-slot, channel octet); deterministic two-level reduction (no float atomics).
__global__ __launch_bounds__(256) void k_gn_partial(const half_t* __restrict__ X, int HW, int C, float* __restrict__ part) {
    extern __shared__ float s_acc[];                 // [pps][C][2]
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int opp = C >> 3;                          // octets per pixel
    const int pps = max(1, 256 / opp);               // pixel sub-slots handled concurrently
    const int tid = threadIdx.x;
    const int sub = tid / opp, oct = tid - sub * opp;
    const int p0 = chunk * GN_PIX_PER_BLOCK;
    const int p1 = min(HW, p0 + GN_PIX_PER_BLOCK);
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
    if (sub < pps) {
        for (int oc = oct; oc < opp; oc += 256) {    // opp > 256 only when C > 2048 (not in this net); keeps it general
            for (int p = p0 + sub; p < p1; p += pps) {
                const half8 v = *reinterpret_cast<const half8*>(X + ((size_t)n * HW + p) * C + oc * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; s[e] += f; q[e] += f * f; }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s_acc[((size_t)sub * C + oc * 8 + e) * 2] = s[e];
                s_acc[((size_t)sub * C + oc * 8 + e) * 2 + 1] = q[e];
                s[e] = 0.f; q[e] = 0.f;
            }
        }
    }
    __syncthreads();
    if (tid < 32) {                                   // one thread per group, fixed summation order
        const int cg = C / 32;
        double ds = 0.0, dq = 0.0;
        for (int sb = 0; sb < pps; ++sb)
            for (int c = tid * cg; c < (tid + 1) * cg; ++c) {
                ds += (double)s_acc[((size_t)sb * C + c) * 2];
                dq += (double)s_acc[((size_t)sb * C + c) * 2 + 1];
            }
        float* o = part + (((size_t)n * gridDim.x + chunk) * 32 + tid) * 2;
        o[0] = (float)ds; o[1] = (float)dq;
    }
}

// one block per image: thread (slice j = tid>>5, group g = tid&31) sums chunks j, j+8, ...; fixed-order combine.
__global__ __launch_bounds__(256) void k_gn_finalize(const float* __restrict__ part, int chunks, int HW, int C, float eps,
                                                     float* __restrict__ stats) {
    __shared__ double s_s[8][32], s_q[8][32];
    const int n = blockIdx.x, g = threadIdx.x & 31, j = threadIdx.x >> 5;
    double ds = 0.0, dq = 0.0;
    for (int c = j; c < chunks; c += 8) {
        ds += (double)part[(((size_t)n * chunks + c) * 32 + g) * 2];
        dq += (double)part[(((size_t)n * chunks + c) * 32 + g) * 2 + 1];
    }
    s_s[j][g] = ds; s_q[j][g] = dq;
    __syncthreads();
    if (j == 0) {
        for (int k = 1; k < 8; ++k) { ds += s_s[k][g]; dq += s_q[k][g]; }
        const double cnt = (double)HW * (C / 32);
        const double mean = ds / cnt;
        double var = dq / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        stats[((size_t)n * 32 + g) * 2] = (float)mean;
        stats[((size_t)n * 32 + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

int gn_stats(const half_t* X, int N, int HW, int C, float eps, float* stats, float* ws, size_t ws_floats, hipStream_t s) {
    PD_REQUIRE(C % 32 == 0, "gn_stats: C %% 32 != 0 (C=%d)", C);
    const int chunks = cdiv(HW, GN_PIX_PER_BLOCK);
    PD_REQUIRE((size_t)N * chunks * 64 <= ws_floats, "gn_stats: workspace too small");
    const int opp = C >> 3;
    const int pps = max(1, 256 / opp);
    const size_t smem = (size_t)pps * C * 2 * sizeof(float);
    PD_REQUIRE(smem <= 64 * 1024, "gn_stats: C too large (%d)", C);
    dim3 g(chunks, N);
    k_gn_partial<<<g, 256, smem, s>>>(X, HW, C, ws);
    k_gn_finalize<<<N, 256, 0, s>>>(ws, chunks, HW, C, eps, stats);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}

// stats from the conv epilogue's octet partials (the two sources of a channel concat may be chunked differently);
// block = image, thread (slice j = tid>>5, group g = tid&31): 32 slices walk the chunks, fixed-order f64 combine
__global__ __launch_bounds__(1024) void k_gn_finalize_oct(const float* __restrict__ partA, int Ca, int chunksA,
                                                          const float* __restrict__ partB, int Cb, int chunksB, int HW, float eps,
                                                          float* __restrict__ stats) {
    __shared__ double s_s[32][33], s_q[32][33];
    const int n = blockIdx.x, g = threadIdx.x & 31, j = threadIdx.x >> 5;
    const int C = Ca + Cb, opg = (C / 32) >> 3;           // octets per group
    const int oa = Ca >> 3, ob = Cb >> 3;
    double ds = 0.0, dq = 0.0;
    for (int k = 0; k < opg; ++k) {
        const int o = g * opg + k;
        const bool inA = o < oa;
        const int chunks = inA ? chunksA : chunksB, os = inA ? oa : ob;
        const float2* src = reinterpret_cast<const float2*>(inA ? partA : partB) + (size_t)n * chunks * os + (inA ? o : o - oa);
#pragma unroll 4
        for (int c = j; c < chunks; c += 32) {
            const float2 v = src[(size_t)c * os];
            ds += (double)v.x; dq += (double)v.y;
        }
    }
    s_s[j][g] = ds; s_q[j][g] = dq;
    __syncthreads();
    if (j == 0) {
        for (int k = 1; k < 32; ++k) { ds += s_s[k][g]; dq += s_q[k][g]; }
        const double cnt = (double)HW * (C / 32);
        const double mean = ds / cnt;
        double var = dq / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        stats[((size_t)n * 32 + g) * 2] = (float)mean;
        stats[((size_t)n * 32 + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

int gn_finalize_oct(const float* partA, int Ca, int chunksA, const float* partB, int Cb, int chunksB, int N, int HW, float eps,
                    float* stats, hipStream_t s) {
    PD_REQUIRE(((Ca + Cb) / 32) % 8 == 0 && Ca % 8 == 0 && Cb % 8 == 0 && (Ca + Cb) % 32 == 0, "gn_finalize_oct: group size must be a multiple of 8");
    k_gn_finalize_oct<<<N, 1024, 0, s>>>(partA, Ca, chunksA, partB, Cb, chunksB, HW, eps, stats);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}

// ---- GroupNorm (+ FiLM) folded to one affine map per (image, channel) for the conv kernels that apply it while staging their
// input (nn_conv_halo.hip, APPLY): y = silu(A x + B), A = rstd gamma t1, B = (beta - mean rstd gamma) t1 + sh with the FiLM terms
// t1 = f16(1 + f16(scale)), sh = f16(shift) exactly as k_gn_apply forms them (1, 0 without FiLM).  The fused form rounds once
// (after the SiLU) where the stand-alone kernel rounds after every op like the reference's f16 tensors do; the difference
// is below the f16 resolution of the result and inside the U1 tolerance (tests compare both with the fp32 reference).
// table [N][C/8][16] = (A0..A7, B0..B7) per channel octet: a wave of the conv owns one octet, so its 16 constants are
// wave-uniform (lanes 0-15 of one register, read with v_readlane).
__global__ __launch_bounds__(256) void k_gn_table(const float* __restrict__ stats, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, const float* __restrict__ film,
                                                  long long film_stride, int C, float* __restrict__ table) {
    const int n = blockIdx.y, cg = C / 32;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < C; c += gridDim.x * blockDim.x) {
        {
            const int grp = c / cg;
            const float mean = stats[((size_t)n * 32 + grp) * 2], rstd = stats[((size_t)n * 32 + grp) * 2 + 1];
            const float ga = rstd * gamma[c], gb = beta[c] - mean * ga;
            float t1 = 1.0f, sh = 0.0f;
            if (film != nullptr) {
                t1 = (float)(half_t)(1.0f + (float)(half_t)film[(size_t)n * film_stride + c]);
                sh = (float)(half_t)film[(size_t)n * film_stride + C + c];
            }
            float* row = table + ((size_t)n * (C / 8) + (c >> 3)) * 16;
            row[c & 7] = ga * t1; row[8 + (c & 7)] = gb * t1 + sh;
        }
    }
}
int gn_table(const float* stats, const float* gamma, const float* beta, const float* film, long long film_stride, int N, int C,
             float* table, hipStream_t s) {
    PD_REQUIRE(C % 32 == 0, "gn_table: C %% 32 != 0");
    k_gn_table<<<dim3(cdiv(C, 256), N), 256, 0, s>>>(stats, gamma, beta, film, film_stride, C, table);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}

// (silu_f, gn_round_f16, gn_elem: nn_common.h -- shared with the row-resident conv, which applies the same element map while staging)

// grid (pixel chunks, N).  thread -> (pixel sub-slot, channel octet): the octet's affine constants live in
// registers for the whole chunk; consecutive threads touch consecutive 16-byte octets (full 128-B lines).
// RES: 0 none, 1 avgpool2 (of activated values), 2 nearest-up2.
#define GNA_ITERS 8        // pixels per thread (round 5: 32 -> 8, -0.5 % on the batch-32 forward, -0.6 % at batch 1; 4 ties, 2 and 1 lose to the per-thread
                           // constant set-up.  The chip's copy ceiling -- 6.2 TB/s -- is only reached by ONE 16-byte element per thread in
                           // hundreds of thousands of workgroups; any per-thread loop copies at 4.8-5.4 TB/s: bench.py calibration.copy16_sweep_gbs)
// The statistics come either finished (stats [N][32][2], FIN = false) or as the producing convs' octet partials (GnParts, FIN = true):
// at small batches every workgroup re-reduces its image's few hundred partials itself (L2-resident, f64, fixed order) and the
// k_gn_finalize_oct launch -- 94 per forward, 5 us each at batch 1 -- disappears.
struct GnParts { const float* a; const float* b; int Ca, Cb, chunksA, chunksB; float eps; };
template <int RES, bool OUT_F32, bool FILM, bool FIN>
__global__ __launch_bounds__(256) void k_gn_apply(const half_t* __restrict__ X, const float* __restrict__ stats,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  const float* __restrict__ film, long long film_stride, int H, int W,
                                                  int C, int silu, void* __restrict__ Yv, const half_t* __restrict__ XB, int Ca, int iters,
                                                  half_t* __restrict__ Yraw, GnParts gp) {
    const int opp = C >> 3, cg = C / 32;
    const int pps = max(1, 256 / opp);
    const int Ho = RES == 1 ? H / 2 : (RES == 2 ? H * 2 : H), Wo = RES == 1 ? W / 2 : (RES == 2 ? W * 2 : W);
    __shared__ double s_fin[FIN ? 512 : 1];
    __shared__ float s_st[FIN ? 64 : 1];
    if (FIN) {
        // thread (slot, octet) sums chunks slot, slot + pps, ... of its octet; thread g < 32 then combines its group's octets
        // over the slots in a fixed order (host guarantees opp <= 256 and a group size that is a multiple of 8 channels)
        const int img = blockIdx.y, fs = threadIdx.x / opp, fo = threadIdx.x - fs * opp;
        if (fs < pps) {
            const int oa = gp.Ca >> 3;
            const bool inA = fo < oa;
            const int chunks = inA ? gp.chunksA : gp.chunksB, os = inA ? oa : (gp.Cb >> 3);
            const float2* src = reinterpret_cast<const float2*>(inA ? gp.a : gp.b) + (size_t)img * chunks * os + (inA ? fo : fo - oa);
            double ds = 0.0, dq = 0.0;
#pragma unroll 8
            for (int c = fs; c < chunks; c += pps) { const float2 v = src[(size_t)c * os]; ds += (double)v.x; dq += (double)v.y; }
            s_fin[(fs * opp + fo) * 2] = ds; s_fin[(fs * opp + fo) * 2 + 1] = dq;
        }
        __syncthreads();
        if (threadIdx.x < 32) {
            const int opg = cg >> 3, g = threadIdx.x;
            double ds = 0.0, dq = 0.0;
            for (int k = 0; k < opg; ++k)
                for (int sl = 0; sl < pps; ++sl) { ds += s_fin[(sl * opp + g * opg + k) * 2]; dq += s_fin[(sl * opp + g * opg + k) * 2 + 1]; }
            const double cnt = (double)H * W * cg;
            const double mean = ds / cnt;
            double var = dq / cnt - mean * mean;
            if (var < 0.0) var = 0.0;
            s_st[2 * g] = (float)mean; s_st[2 * g + 1] = (float)(1.0 / sqrt(var + (double)gp.eps));
        }
        __syncthreads();
    }
    // (lab: descending image / chunk order so that the pass starts on what the producing conv wrote last and ends on what the
    // consuming conv reads first -- measured 1 % slower per DDNM step than the plain order; PD_LAB_GN_REV)
#ifdef PD_LAB_GN_REV
    const int n = gridDim.y - 1 - blockIdx.y, bx = gridDim.x - 1 - blockIdx.x;
#else
    const int n = blockIdx.y, bx = blockIdx.x;
#endif
    const int sub = threadIdx.x / opp;
    if (sub >= pps) return;
    const int p_begin = bx * (pps * iters), p_end = min(Ho * Wo, p_begin + pps * iters);
    for (int oc = threadIdx.x - sub * opp; oc < opp; oc += 256) {
        const int c0 = oc * 8;
        // input of a never-materialised channel concat: channels [0, Ca) live in X (pixel stride Ca), the rest in XB
        const bool second = XB != nullptr && c0 >= Ca;
        const half_t* const Xs = second ? XB + (c0 - Ca) : X + c0;
        const int cs = XB == nullptr ? C : (second ? C - Ca : Ca);
        float ga[8], gb[8], t1[8], sh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = c0 + e, grp = c / cg;
            const float mean = FIN ? s_st[2 * grp] : stats[((size_t)n * 32 + grp) * 2], rstd = FIN ? s_st[2 * grp + 1] : stats[((size_t)n * 32 + grp) * 2 + 1];
            ga[e] = rstd * gamma[c];
            gb[e] = beta[c] - mean * ga[e];
            if (FILM) {
                t1[e] = (float)(half_t)(1.0f + (float)(half_t)film[(size_t)n * film_stride + c]);
                sh[e] = (float)(half_t)film[(size_t)n * film_stride + C + c];
            }
        }
        float raw[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) raw[e] = 0.f;
        auto act = [&](int yi, int xi, float* o) {
            const half8 v = *reinterpret_cast<const half8*>(Xs + (((size_t)n * H + yi) * W + xi) * cs);
            if (RES == 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) raw[e] += (float)v[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o[e] = gn_elem<OUT_F32, FILM>((float)v[e], ga[e], gb[e], FILM ? t1[e] : 1.f, FILM ? sh[e] : 0.f, silu);
            }
        };
        auto finish = [&](const half8& v, float* r) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                r[e] = gn_elem<OUT_F32, FILM>((float)v[e], ga[e], gb[e], FILM ? t1[e] : 1.f, FILM ? sh[e] : 0.f, silu);
            }
        };
        auto store = [&](int p, const float* r) {
            const size_t o = ((size_t)n * Ho * Wo + p) * C + c0;
            if (OUT_F32) {
                float* Y = reinterpret_cast<float*>(Yv);
                *reinterpret_cast<float4*>(Y + o) = make_float4(r[0], r[1], r[2], r[3]);
                *reinterpret_cast<float4*>(Y + o + 4) = make_float4(r[4], r[5], r[6], r[7]);
            } else {
                half8 hv;
#pragma unroll
                for (int e = 0; e < 8; ++e) hv[e] = (half_t)r[e];
                // (a streaming / nontemporal store makes this pass 10-14 % faster on its own -- tools/bench_gn.py -- and the DDNM step
                // 3.6 % SLOWER: the conv that consumes the tensor then misses the part of it the caches would have kept)
#ifdef PD_LAB_GN_NT
                __builtin_nontemporal_store(hv, reinterpret_cast<half8*>(reinterpret_cast<half_t*>(Yv) + o));
#else
                *reinterpret_cast<half8*>(reinterpret_cast<half_t*>(Yv) + o) = hv;
#endif
            }
        };
        if (RES == 0) {
            // four independent 16-byte loads in flight per thread (memory-level parallelism for the HBM stream)
            int p = p_begin + sub;
#ifndef GNA_U
#define GNA_U 4
#endif
            for (; p + (GNA_U - 1) * pps < p_end; p += GNA_U * pps) {
                half8 v[GNA_U];
#pragma unroll
#ifdef PD_LAB_GN_NTL                                       // (lab builds only: streaming loads -- no effect on the step)
                for (int u = 0; u < GNA_U; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(Xs + ((size_t)n * H * W + p + u * pps) * cs));
#else
                for (int u = 0; u < GNA_U; ++u) v[u] = *reinterpret_cast<const half8*>(Xs + ((size_t)n * H * W + p + u * pps) * cs);
#endif
#pragma unroll
                for (int u = 0; u < GNA_U; ++u) { float r[8]; finish(v[u], r); store(p + u * pps, r); }
            }
            for (; p < p_end; p += pps) {
                const half8 v = *reinterpret_cast<const half8*>(Xs + ((size_t)n * H * W + p) * cs);
                float r[8]; finish(v, r); store(p, r);
            }
        } else {
            for (int p = p_begin + sub; p < p_end; p += pps) {
                float r[8];
                const int yo = p / Wo, xo = p - yo * Wo;
                if (RES == 1) {
                    float a[8], b[8], c[8], d[8];
                    act(2 * yo, 2 * xo, a); act(2 * yo, 2 * xo + 1, b); act(2 * yo + 1, 2 * xo, c); act(2 * yo + 1, 2 * xo + 1, d);
#pragma unroll
                    for (int e = 0; e < 8; ++e) r[e] = (a[e] + b[e] + c[e] + d[e]) * 0.25f;
                    if (Yraw != nullptr) {                 // the block's x branch: AvgPool2d(2) of the raw input, same pixels, same sum order as k_resample
                        half8 hv;
#pragma unroll
                        for (int e = 0; e < 8; ++e) { hv[e] = (half_t)(raw[e] * 0.25f); }
                        *reinterpret_cast<half8*>(Yraw + ((size_t)n * Ho * Wo + p) * C + c0) = hv;
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) raw[e] = 0.f;
                } else {
                    act(yo >> 1, xo >> 1, r);
                }
                store(p, r);
            }
        }
    }
}

thread_local int g_gn_iters = 0;       // lab hook (pdhip_debug_set_gn_iters): pixels per thread of k_gn_apply, 0 = default
int gn_apply(const half_t* X, const float* stats, const float* gamma, const float* beta, const float* film,
             long long film_stride, int N, int H, int W, int C, int silu, int resample, void* Y, int out_f32, hipStream_t s,
             const half_t* XB, int Ca, half_t* Yraw, const GnPartsArg* parts) {
    GnParts gp{nullptr, nullptr, 0, 0, 0, 0, 0.f};
    if (parts != nullptr) {
        PD_REQUIRE(stats == nullptr && parts->partA != nullptr && (C >> 3) <= 256 && ((C / 32) % 8) == 0 && parts->Ca + parts->Cb == C &&
                   parts->Ca % 8 == 0 && parts->Cb % 8 == 0 && (parts->Cb == 0 || parts->partB != nullptr),
                   "gn_apply: bad octet partials for the in-kernel statistics");
        gp = GnParts{parts->partA, parts->partB, parts->Ca, parts->Cb, parts->chunksA, parts->chunksB, parts->eps};
    }
    PD_REQUIRE(Yraw == nullptr || (resample == 1 && XB == nullptr), "gn_apply: the raw avg-pool output belongs to resample 1");
    PD_REQUIRE(XB == nullptr || (Ca > 0 && Ca < C && Ca % 8 == 0), "gn_apply: bad two-source split");
    PD_REQUIRE(C % 32 == 0 && resample >= 0 && resample <= 2, "gn_apply: bad arguments");
    PD_REQUIRE(resample != 1 || (H % 2 == 0 && W % 2 == 0), "gn_apply: avgpool needs even H, W");
    PD_REQUIRE(!out_f32 || (resample == 0 && film == nullptr), "gn_apply: f32 output only without resampling / FiLM");
    PD_REQUIRE(film == nullptr || resample == 0, "gn_apply: FiLM only without resampling");
    const int Ho = resample == 1 ? H / 2 : (resample == 2 ? H * 2 : H), Wo = resample == 1 ? W / 2 : (resample == 2 ? W * 2 : W);
    const int opp = C >> 3, pps = max(1, 256 / opp);
    // pixels per thread: GNA_ITERS on the big tensors (constants amortised), fewer on the small ones so that the grid still fills
    // the chip (a 64-pixel 8x8 level with 32 pixels per thread is 8 workgroups walking a serial latency chain)
    int iters = g_gn_iters > 0 ? g_gn_iters : GNA_ITERS;
    // (with the in-kernel statistics every workgroup pays the re-reduction of its image's partials first: fewer, longer workgroups --
    // 512 against 2 048 is -2.3 % on the batch-1 forward, -1 % at batch 8; tools/time_unet.py)
    const int tgt_blocks = parts != nullptr && N <= 8 ? 512 : 2048;
    while (iters > 1 && (long long)cdiv((long long)Ho * Wo, pps * iters) * N < tgt_blocks) iters >>= 1;
    dim3 grid(cdiv((long long)Ho * Wo, pps * iters), N);
#define GNA_ARGS X, stats, gamma, beta, film, film_stride, H, W, C, silu, Y, XB, Ca, iters, Yraw, gp
#define GNA_LAUNCH(R, O, F) do { if (parts) k_gn_apply<R, O, F, true><<<grid, 256, 0, s>>>(GNA_ARGS); else k_gn_apply<R, O, F, false><<<grid, 256, 0, s>>>(GNA_ARGS); } while (0)
    if (out_f32) GNA_LAUNCH(0, true, false);
    else if (film) GNA_LAUNCH(0, false, true);
    else if (resample == 0) GNA_LAUNCH(0, false, false);
    else if (resample == 1) GNA_LAUNCH(1, false, false);
    else GNA_LAUNCH(2, false, false);
#undef GNA_LAUNCH
#undef GNA_ARGS
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}

// x_upd of up/down ResBlocks: plain AvgPool2d(2) (mode 1) or nearest x2 (mode 2) on the raw activation
__global__ __launch_bounds__(256) void k_resample(const half_t* __restrict__ X, int H, int W, int C, int mode,
                                                  half_t* __restrict__ Y, long long total_oct) {
    const int opp = C >> 3;
    const int Ho = mode == 1 ? H / 2 : H * 2, Wo = mode == 1 ? W / 2 : W * 2;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total_oct; idx += (long long)gridDim.x * blockDim.x) {
        const int oc = (int)(idx % opp);
        const long long pix = idx / opp;
        const int xo = (int)(pix % Wo), yo = (int)((pix / Wo) % Ho), n = (int)(pix / ((long long)Wo * Ho));
        half8 out;
        if (mode == 1) {
            const half_t* p = X + (((size_t)n * H + 2 * yo) * W + 2 * xo) * C + oc * 8;
            const half8 a = *reinterpret_cast<const half8*>(p), b = *reinterpret_cast<const half8*>(p + C);
            const half8 c = *reinterpret_cast<const half8*>(p + (size_t)W * C), d = *reinterpret_cast<const half8*>(p + (size_t)W * C + C);
#pragma unroll
            for (int e = 0; e < 8; ++e) out[e] = (half_t)(((float)a[e] + (float)b[e] + (float)c[e] + (float)d[e]) * 0.25f);
        } else {
            out = *reinterpret_cast<const half8*>(X + (((size_t)n * H + (yo >> 1)) * W + (xo >> 1)) * C + oc * 8);
        }
        *reinterpret_cast<half8*>(Y + (((size_t)n * Ho + yo) * Wo + xo) * C + oc * 8) = out;
    }
}

int resample2x(const half_t* X, int N, int H, int W, int C, int mode, half_t* Y, hipStream_t s) {
    PD_REQUIRE((mode == 1 || mode == 2) && C % 8 == 0, "resample2x: bad arguments");
    const int Ho = mode == 1 ? H / 2 : H * 2, Wo = mode == 1 ? W / 2 : W * 2;
    const long long total = (long long)N * Ho * Wo * (C / 8);
    k_resample<<<(int)std::min<long long>((total + 255) / 256, 65536), 256, 0, s>>>(X, H, W, C, mode, Y, total);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}

__global__ __launch_bounds__(256) void k_concat(const half_t* __restrict__ A, int Ca, const half_t* __restrict__ B, int Cb,
                                                long long pixels, half_t* __restrict__ Y) {
    const int oa = Ca >> 3, ob = Cb >> 3, ot = oa + ob;
    const long long total = pixels * ot;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const long long p = idx / ot;
        const int o = (int)(idx - p * ot);
        const half8 v = (o < oa) ? *reinterpret_cast<const half8*>(A + (size_t)p * Ca + o * 8)
                                 : *reinterpret_cast<const half8*>(B + (size_t)p * Cb + (o - oa) * 8);
        *reinterpret_cast<half8*>(Y + (size_t)p * (Ca + Cb) + o * 8) = v;
    }
}

int concat_channels(const half_t* A, int Ca, const half_t* B, int Cb, long long pixels, half_t* Y, hipStream_t s) {
    PD_REQUIRE(Ca % 8 == 0 && Cb % 8 == 0, "concat_channels: channels must be multiples of 8");
    const long long total = pixels * ((Ca + Cb) / 8);
    k_concat<<<(int)std::min<long long>((total + 255) / 256, 65536), 256, 0, s>>>(A, Ca, B, Cb, pixels, Y);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}

// ---- GroupNorm-apply + skip 1x1 of a channel-changing ResBlock in ONE pass over x (unet.py:197-209, 236-256: in_layers' GroupNorm32 ->
// SiLU and skip_connection = conv_nd(1) both read the block input).  The decoder blocks of the 256^2 / 128^2 levels read a 512- /
// 768-channel concat: as two launches (k_gn_apply, k_conv_igemm<1>) x is streamed from HBM twice (2 x 2.15 GB at UNet batch 32, 256^2).
// Here a workgroup owns 128 pixels x all C input channels: per 64-channel chunk every thread loads 4 x 16 B of x ONCE, parks the raw
// values in LDS as the MFMA operand of the skip GEMM (128 x 256 x 64 per chunk, weights chunk from L2) and stores silu(GN(x)) of the same
// registers to h0.  HBM-bound by construction (MFMA time is a quarter of the stream time); two workgroups per CU overlap each other's
// load waits.  Same arithmetic as the two-launch form: gn_elem on the f16 input, skip = f16(acc + bias).
constexpr int GS_BM = 128, GS_BN = 256, GS_CLD = GS_BN + 8;
constexpr int GS_SMEM = GS_BM * GS_CLD * 2;                 // epilogue staging [128][264] f16 = 67 584 B >= K-loop tiles (16 + 32 KiB)

// epilogue shared by the variants: lane holds pixel 16 i + (lane & 15), channels 16 j + 4 (lane >> 4) + r (weights x activations) ->
// f16(acc + bias) -> LDS -> 16-byte coalesced NHWC rows.  The caller has synchronised: the staging area is free.
template <int MI = 4>
__device__ __forceinline__ void gs_epilogue(const float4_t (&acc)[MI][8], char* smem, const float* __restrict__ bias, half_t* __restrict__ SK,
                                            long long m0, int tid) {
    const int lane = tid & 63, wave = tid >> 6, wm = wave & 1, wn = wave >> 1;
    half_t* const Cs = reinterpret_cast<half_t*>(smem);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int nl = wn * 128 + j * 16 + (lane >> 4) * 4;
        float4_t bv = (float4_t){0.f, 0.f, 0.f, 0.f};
        if (bias != nullptr) bv = *reinterpret_cast<const float4_t*>(bias + nl);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int ml = wm * (16 * MI) + i * 16 + (lane & 15);
            half4 h;
#pragma unroll
            for (int r = 0; r < 4; ++r) h[r] = (half_t)(acc[i][j][r] + bv[r]);
            *reinterpret_cast<half4*>(&Cs[ml * GS_CLD + nl]) = h;
        }
    }
    __syncthreads();
    const int col8 = (tid & 31) * 8;
#pragma unroll
    for (int p = 0; p < 4 * MI; ++p) {
        const int row = p * 8 + (tid >> 5);
        *reinterpret_cast<half8*>(SK + (size_t)(m0 + row) * GS_BN + col8) = *reinterpret_cast<const half8*>(&Cs[row * GS_CLD + col8]);
    }
}
// one 64-deep K-step of the skip GEMM on the staged tiles (wave tile 64 px x 128 co); NB weight fragments per batch (register budget)
template <int NB, int MI = 4>
__device__ __forceinline__ void gs_mfma_step(float4_t (&acc)[MI][8], const char* As, const char* Ws, int lane, int wm, int wn) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int fo = ((kk * 4 + (lane >> 4)) ^ (lane & 7)) << 4;
        half8 a[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const half8*>(As + (wm * (16 * MI) + i * 16 + (lane & 15)) * 128 + fo);
#pragma unroll
        for (int jh = 0; jh < 8 / NB; ++jh) {
            half8 b[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) b[j] = *reinterpret_cast<const half8*>(Ws + (wn * 128 + (jh * NB + j) * 16 + (lane & 15)) * 128 + fo);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[i][jh * NB + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j], a[i], acc[i][jh * NB + j], 0, 0, 0);
        }
    }
}

// a wave-uniform pointer pinned to scalar registers (the compiler otherwise keeps selected / loop-carried bases in VGPR pairs)
template <typename T> __device__ __forceinline__ T* gs_uniform(T* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}
// (variant 0 of round 3 -- activation loads requested under the MFMA phase, 2 spilled VGPRs, 3.8 TB/s against 4.3 -- was removed in round 5)

// variant 1: two workgroups per CU like variant 0, but the activation chunk k + 1 is requested at the TOP of iteration k (second
// register set) and is in flight for the whole iteration -- GroupNorm math, barrier, MFMA phase -- instead of one MFMA phase
// (0.45 us against an HBM latency of several us under load); the weight chunk k + 1 (L2) is requested under the MFMAs of chunk k and
// parked first thing in iteration k + 1, so it is never live across the GroupNorm math (register budget 256); the affine constants
// of the image sit in LDS.  Loads past the last chunk are clamped (re-read, unused): the loop body has no branches.
#ifndef GS1_NB
#define GS1_NB 2
#endif
constexpr int GS1_SMEM = GS_SMEM;                           // tiles 48 KiB + constants C * 8 B <= 16 KiB < epilogue staging 66 KiB
// BM = 128 pixels per workgroup, or 64 (round 4) where 128-pixel tiles would leave CUs idle: the 128^2 level at UNet batch 1 is 128 tiles
template <int BM>
__global__ __launch_bounds__(256, 2) void k_gn_skip_w1(const half_t* __restrict__ XA, const half_t* __restrict__ XB, int Ca, int C,
                                                       const float* __restrict__ stats, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const half_t* __restrict__ Wt,
                                                       const float* __restrict__ bias, half_t* __restrict__ H0, half_t* __restrict__ SK,
                                                       int HW) {
    extern __shared__ __align__(16) char gs_smem[];
    char* const As = gs_smem;
    constexpr int MI = BM / 32, RJ = BM / 32;                // 16-pixel row tiles per wave; 32-row load passes per chunk
    char* const Ws = gs_smem + BM * 128;
    float* const Gs = reinterpret_cast<float*>(gs_smem + BM * 128 + GS_BN * 128);   // [C/8][16] = (ga0..7, gb0..7) per channel octet
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave & 1, wn = wave >> 1;
    const long long m0 = (long long)blockIdx.x * BM;
    const int n = (int)(m0 / HW), cg = C / 32, Cb = C - Ca, KC = C / 64;
    const int slot = tid & 7, prow = tid >> 3;
    const int sw = (slot ^ (prow & 7)) << 4;                  // (rows prow + 32 j share prow & 7)
    half8 xa[2][RJ], wr[8];
    // addressing: wave-uniform 64-bit bases (scalar registers) + 32-bit per-lane byte offsets that do not depend on the chunk
    const unsigned offA = (unsigned)(prow * Ca + slot * 8) * 2u, offB = (unsigned)(prow * Cb + slot * 8) * 2u;   // row prow of XA / XB
    const unsigned offC = (unsigned)(prow * C + slot * 8) * 2u;                                                  // row prow of Wt / H0
    const char* const baseA = reinterpret_cast<const char*>(XA + (size_t)m0 * Ca);
    const char* const baseB = reinterpret_cast<const char*>(XB + (size_t)m0 * Cb);
    const char* const baseW = reinterpret_cast<const char*>(Wt);
    char* const baseH = reinterpret_cast<char*>(H0 + (size_t)m0 * C);
    auto issue_acts = [&](int kc, half8* dst) {
        const int cu = min(kc, KC - 1) * 64;                  // (uniform)
        const bool second = cu >= Ca;                         // (uniform over the workgroup: Ca % 64 == 0)
        const char* const src = gs_uniform(second ? baseB + (size_t)(cu - Ca) * 2 : baseA + (size_t)cu * 2);
        const unsigned off = second ? offB : offA, rs = (unsigned)(second ? Cb : Ca) * 64u;   // 32 rows in bytes
#pragma unroll
        for (int j = 0; j < RJ; ++j) dst[j] = *reinterpret_cast<const half8*>(src + (off + j * rs));
    };
    auto issue_w = [&](int kc) {
        const char* const src = gs_uniform(baseW + (size_t)(min(kc, KC - 1) * 64) * 2);
#pragma unroll
        for (int j = 0; j < 8; ++j) wr[j] = *reinterpret_cast<const half8*>(src + (offC + j * ((unsigned)C * 64u)));
    };
    issue_acts(0, xa[0]); issue_w(0);
    // the affine constants of this image, once per workgroup: ga = rstd gamma, gb = beta - mean ga (k_gn_apply's expressions)
    for (int c = tid; c < C; c += 256) {
        const float* st = stats + ((size_t)n * 32 + c / cg) * 2;
        const float ga = st[1] * gamma[c];
        Gs[(c >> 3) * 16 + (c & 7)] = ga;
        Gs[(c >> 3) * 16 + 8 + (c & 7)] = beta[c] - st[0] * ga;
    }
    float4_t acc[MI][8];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};
    auto gn_part = [&](int kc, const half8* cur) {           // h0 = silu(GN(x)) of chunk kc straight from its registers
        const int c0 = kc * 64 + slot * 8;
        const float4_t* gp = reinterpret_cast<const float4_t*>(Gs + (c0 >> 3) * 16);
        const float4_t a0 = gp[0], a1 = gp[1], b0 = gp[2], b1 = gp[3];
        char* const hb = gs_uniform(baseH + (size_t)(kc * 64) * 2);
#pragma unroll
        for (int j = 0; j < RJ; ++j) {
            half8 hv;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                hv[e] = (half_t)gn_elem<false, false>((float)cur[j][e], e < 4 ? a0[e & 3] : a1[e & 3], e < 4 ? b0[e & 3] : b1[e & 3], 1.f, 0.f, 1);
            *reinterpret_cast<half8*>(hb + (offC + j * ((unsigned)C * 64u))) = hv;
        }
    };
    auto step = [&](int kc, const half8* cur, half8* nxt) {
        __syncthreads();                                      // the previous chunk's fragment reads are done (kc == 0: Gs is complete)
#pragma unroll
        for (int j = 0; j < 8; ++j) *reinterpret_cast<half8*>(Ws + (prow + 32 * j) * 128 + sw) = wr[j];
#pragma unroll
        for (int j = 0; j < RJ; ++j) *reinterpret_cast<half8*>(As + (prow + 32 * j) * 128 + sw) = cur[j];
        issue_acts(kc + 1, nxt);
#ifndef GS1_GN_LATE
        gn_part(kc, cur);
#endif
        __syncthreads();
#ifndef GS1_W_LATE
        issue_w(kc + 1);
#endif
        gs_mfma_step<GS1_NB, MI>(acc, As, Ws, lane, wm, wn);
#ifdef GS1_GN_LATE
        // the GroupNorm math needs registers only: same basic block as the MFMAs, interleaved (matrix pipe and VALU overlap in-wave)
        gn_part(kc, cur);
#if GS1_GN_LATE > 1
#pragma unroll
        for (int g = 0; g < 64; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, GS1_GN_LATE, 0);      // a few VALU
        }
#endif
#endif
#ifdef GS1_W_LATE
        issue_w(kc + 1);
#endif
    };
    for (int kc = 0; kc < KC; kc += 2) {                      // (KC is even: C % 256 == 0) -- static register-set parities
        step(kc, xa[0], xa[1]);
        step(kc + 1, xa[1], xa[0]);
    }
    __syncthreads();
    gs_epilogue<MI>(acc, gs_smem, bias, SK, m0, tid);
}
thread_local int g_gs_variant = 1;     // tuning hook (pdhip_debug_set_gn_skip_variant)

bool gn_skip_eligible(int N, int HW, int Ca, int C, int Cout, int Cout_pad) {
    return Cout == GS_BN && Cout_pad == GS_BN && C % 64 == 0 && Ca % 64 == 0 && Ca > 0 && Ca <= C && ((C / 32) % 8) == 0 && C <= 2048 && HW % GS_BM == 0 && C % 128 == 0;
}

int gn_skip(const half_t* XA, const half_t* XB, int Ca, int C, const float* stats, const float* gamma, const float* beta, const half_t* Wt,
            const float* bias, half_t* H0, half_t* SK, int N, int HW, hipStream_t s) {
    PD_REQUIRE(gn_skip_eligible(N, HW, Ca, C, GS_BN, GS_BN) && (Ca == C || XB != nullptr), "gn_skip: shape not served by the fused kernel");
    static thread_local bool attr_set = false;
    if (!attr_set) {
        PD_HIP(hipFuncSetAttribute((const void*)k_gn_skip_w1<128>, hipFuncAttributeMaxDynamicSharedMemorySize, GS1_SMEM));
        PD_HIP(hipFuncSetAttribute((const void*)k_gn_skip_w1<64>, hipFuncAttributeMaxDynamicSharedMemorySize, GS1_SMEM));
        attr_set = true;
    }
    const long long tiles = (long long)N * HW / GS_BM;
    if (tiles < 256 && g_gs_variant != 2) k_gn_skip_w1<64><<<(int)(2 * tiles), 256, GS1_SMEM, s>>>(XA, XB, Ca, C, stats, gamma, beta, Wt, bias, H0, SK, HW);
    else k_gn_skip_w1<128><<<(int)tiles, 256, GS1_SMEM, s>>>(XA, XB, Ca, C, stats, gamma, beta, Wt, bias, H0, SK, HW);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}

}  // namespace pdnn
