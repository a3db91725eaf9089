// Rows U1 / D1: the small dense ops around the conv torso and the fused DDNM update.
//  * first conv 3->C (unet.py:483, f16) and the f32 output head conv C->6 (unet.py:613-617, stays f32);
//  * timestep embedding + time_embed MLP (nn.py:103-121, unet.py:472-476) and the per-ResBlock
//    emb_layers Linear (unet.py:199-205) as one batched GEMV over the concatenated weight rows (all f32);
//  * DDNM masked-projection step (models/DDNM/guided_diffusion/diffusion.py:529-552) as ONE elementwise
//    kernel (the reference runs ~15 torch kernels + randn_like + two host round trips per step), with
//    either injected noise (parity tests) or on-device Philox4x32-10 + Box-Muller noise.
#include "nn_common.h"
using namespace pdhip;
namespace pdnn {

// ------------------------------------------------------------------------------------------------
// conv_in: x f32 NCHW [N,3,H,W] -> (x.half()) conv3x3 -> Y f16 NHWC [N,H,W,Cout].
// K = 27 is far too short for its own kernel to matter: gather the 3x3x3 patch of every pixel into a 32-wide
// f16 row (k = tap*3 + c, k >= 27 zero) and run the MFMA GEMM as a 1x1 convolution with Cin = 32.
__global__ __launch_bounds__(256) void k_im2col_in(const float* __restrict__ x, half_t* __restrict__ A, int H, int W,
                                                   long long pixels) {
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < pixels; p += (long long)gridDim.x * blockDim.x) {
        const int xx = (int)(p % W), y = (int)((p / W) % H);
        const long long n = p / ((long long)W * H);
        half_t row[32];
#pragma unroll
        for (int k = 27; k < 32; ++k) row[k] = (half_t)0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int yy = y + ky - 1, xc = xx + kx - 1;
                const bool ok = yy >= 0 && yy < H && xc >= 0 && xc < W;
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    row[(ky * 3 + kx) * 3 + c] = ok ? (half_t)x[(((size_t)n * 3 + c) * H + yy) * W + xc] : (half_t)0.f;
            }
        half8* dst = reinterpret_cast<half8*>(A + (size_t)p * 32);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            half8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = row[q * 8 + e];
            dst[q] = v;
        }
    }
}

int conv_in_3x3(const float* x_nchw, const half_t* Wt, const float* bias, half_t* Y, int N, int H, int W, int Cout, int Cout_pad,
                half_t* im2col_ws, const half_t* zero_page, hipStream_t s, float* gn_part, int* gn_fused) {
    const long long pixels = (long long)N * H * W;
    k_im2col_in<<<(int)std::min<long long>((pixels + 255) / 256, 8192), 256, 0, s>>>(x_nchw, im2col_ws, H, W, pixels);
    PD_LAUNCH_CHECK();
    return conv_igemm(im2col_ws, Wt, bias, nullptr, Y, N, H, W, 32, Cout, Cout_pad, 1, zero_page, s, nullptr, 0, gn_part, gn_fused);
}

// ------------------------------------------------------------------------------------------------
// y[n][r] = b[r] + sum_k W[r][k] x[n][k], f32 (time-embedding MLP and every ResBlock's emb_layers in one call).
// HBM-bound on the weight stream: x (<= 8 x K floats per pass) is staged in LDS once per block, a wave owns
// GEMV_ROWS/4 rows and streams each row with 16-byte loads; batch entries are processed 8 at a time.
#define GEMV_ROWS 128
template <int KI>                                                        // KI > 0: K == KI * 256, the whole row slice is loaded up front
__global__ __launch_bounds__(256) void k_gemv_rows(const float* __restrict__ Wm, const float* __restrict__ b,
                                                   const float* __restrict__ x, float* __restrict__ y, int R, int K, int N,
                                                   int silu_out) {
    extern __shared__ __attribute__((aligned(16))) float s_x[];          // [8][K]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = blockIdx.x * GEMV_ROWS + wave * (GEMV_ROWS / 4);
    for (int n0 = 0; n0 < N; n0 += 8) {
        const int nb = min(8, N - n0);
        __syncthreads();
        for (int j = 0; j < 8; ++j)
            for (int k = threadIdx.x; k < K; k += 256) s_x[j * K + k] = j < nb ? x[(size_t)(n0 + j) * K + k] : 0.f;
        __syncthreads();
        for (int rr = 0; rr < GEMV_ROWS / 4; rr += 4) {          // 4 rows at a time, all their 16-byte loads in flight together
            const int r = r0 + rr;
            if (r >= R) break;
            float acc[4][8];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[q][j] = 0.f;
            if (KI > 0) {
                float4 w[4][KI > 0 ? KI : 1];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int it = 0; it < KI; ++it)
                        w[q][it] = *reinterpret_cast<const float4*>(Wm + (size_t)min(r + q, R - 1) * K + it * 256 + lane * 4);
#pragma unroll
                for (int it = 0; it < KI; ++it)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float4 xv = *reinterpret_cast<const float4*>(s_x + j * K + it * 256 + lane * 4);
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            acc[q][j] += w[q][it].x * xv.x + w[q][it].y * xv.y + w[q][it].z * xv.z + w[q][it].w * xv.w;
                    }
            } else {
                for (int k = lane; k < K; k += 64) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float w = Wm[(size_t)min(r + q, R - 1) * K + k];
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[q][j] += w * s_x[j * K + k];
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float a = acc[q][j];
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
                    if (lane == 0 && j < nb && r + q < R) {
                        a += b[r + q];
                        if (silu_out) a = a / (1.0f + expf(-a));
                        y[(size_t)(n0 + j) * R + r + q] = a;
                    }
                }
        }
    }
}

static int gemv_launch(const float* Wm, const float* b, const float* x, float* y, int R, int K, int N, int silu_out, hipStream_t s) {
    PD_REQUIRE((size_t)8 * K * sizeof(float) <= 64 * 1024, "gemv_rows: K too large (%d)", K);
    const size_t smem = (size_t)8 * K * sizeof(float);
    const int grid = cdiv(R, GEMV_ROWS);
    if (K == 1024) k_gemv_rows<4><<<grid, 256, smem, s>>>(Wm, b, x, y, R, K, N, silu_out);
    else if (K == 512) k_gemv_rows<2><<<grid, 256, smem, s>>>(Wm, b, x, y, R, K, N, silu_out);
    else if (K == 256) k_gemv_rows<1><<<grid, 256, smem, s>>>(Wm, b, x, y, R, K, N, silu_out);
    else k_gemv_rows<0><<<grid, 256, smem, s>>>(Wm, b, x, y, R, K, N, silu_out);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}

int gemv_rows(const float* Wm, const float* b, const float* x, float* y, int R, int K, int N, hipStream_t s) {
    return gemv_launch(Wm, b, x, y, R, K, N, 0, s);
}

__global__ void k_timestep_embedding(const float* __restrict__ t, int N, int mc, float* __restrict__ e) {
    const int half = mc / 2;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < N * half; idx += gridDim.x * blockDim.x) {
        const int n = idx / half, i = idx - n * half;
        const float freq = expf(-logf(10000.0f) * (float)i / (float)half);
        const float a = t[n] * freq;
        e[(size_t)n * mc + i] = cosf(a);
        e[(size_t)n * mc + half + i] = sinf(a);
    }
}

// emb_silu = silu( W2 silu(W0 temb + b0) + b2 )  -- every ResBlock applies SiLU to emb before its Linear.
int timestep_mlp(const float* t, int N, int mc, const float* w0, const float* b0, const float* w2, const float* b2,
                 float* emb_silu, float* tmp, hipStream_t s) {
    float* temb = tmp;                        // [N][mc]
    float* h1 = tmp + (size_t)N * mc;         // [N][4mc]
    k_timestep_embedding<<<cdiv((long long)N * mc / 2, 256), 256, 0, s>>>(t, N, mc, temb);
    int rc = gemv_launch(w0, b0, temb, h1, 4 * mc, mc, N, 1, s);
    if (rc) return rc;
    return gemv_launch(w2, b2, h1, emb_silu, 4 * mc, 4 * mc, N, 1, s);
}

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 counter RNG + Box-Muller: stateless N(0,1) stream (seed, stream id, element index).
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        c[0] = n0; c[1] = (uint32_t)p1; c[2] = n2; c[3] = (uint32_t)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ void normal4(unsigned long long seed, unsigned long long stream_id, unsigned long long quad, float out[4]) {
    uint32_t c[4] = {(uint32_t)quad, (uint32_t)(quad >> 32), (uint32_t)stream_id, (uint32_t)(stream_id >> 32)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float u0 = ((float)c[0] + 0.5f) * 2.3283064365386963e-10f, u1 = ((float)c[1] + 0.5f) * 2.3283064365386963e-10f;
    const float u2 = ((float)c[2] + 0.5f) * 2.3283064365386963e-10f, u3 = ((float)c[3] + 0.5f) * 2.3283064365386963e-10f;
    const float r0 = sqrtf(-2.0f * logf(u0)), r1 = sqrtf(-2.0f * logf(u2));
    out[0] = r0 * cosf(6.283185307179586f * u1); out[1] = r0 * sinf(6.283185307179586f * u1);
    out[2] = r1 * cosf(6.283185307179586f * u3); out[3] = r1 * sinf(6.283185307179586f * u3);
}

__global__ void k_philox_normal(float* __restrict__ out, long long n, unsigned long long seed, unsigned long long stream_id,
                                unsigned long long quad0) {
    const long long quads = (n + 3) / 4;
    for (long long qd = blockIdx.x * (long long)blockDim.x + threadIdx.x; qd < quads; qd += (long long)gridDim.x * blockDim.x) {
        float z[4];
        normal4(seed, stream_id, quad0 + (unsigned long long)qd, z);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (qd * 4 + e < n) out[qd * 4 + e] = z[e];
    }
}

int philox_normal(float* out, long long n, unsigned long long seed, unsigned long long stream_id, hipStream_t s,
                  unsigned long long quad0) {
    k_philox_normal<<<(int)std::min<long long>(((n + 3) / 4 + 255) / 256, 4096), 256, 0, s>>>(out, n, seed, stream_id, quad0);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}

// ------------------------------------------------------------------------------------------------
// D1.  y = mask * (2 img - 1);  step: x0 = (x - e sqrt(1-a_t))/sqrt(a_t); x0h = x0 - m (m x0 - y);
//      x' = sqrt(a_next) x0h + sigma_t (c1 eps + c2 e);  finish: clamp((x+1)/2, 0, 1).
__global__ void k_ddnm_prepare(const float* __restrict__ img, const float* __restrict__ mask, float* __restrict__ y, int HW,
                               long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / (3LL * HW);
        const int p = (int)(i % HW);
        y[i] = (2.0f * img[i] - 1.0f) * mask[n * HW + p];
    }
}
int ddnm_prepare(const float* masked_img, const float* mask, float* y, int N, int HW, hipStream_t s) {
    const long long total = (long long)N * 3 * HW;
    k_ddnm_prepare<<<(int)std::min<long long>((total + 255) / 256, 4096), 256, 0, s>>>(masked_img, mask, y, HW, total);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}

__global__ void k_ddnm_update(float* __restrict__ x, const float* __restrict__ et, int Cet, const float* __restrict__ y,
                              const float* __restrict__ mask, const float* __restrict__ eps, unsigned long long seed,
                              unsigned long long step, DdnmCoef co, int HW, long long total, unsigned long long quad0) {
    // one thread = 4 consecutive elements (HW % 4 == 0), so one Philox call feeds it
    for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q * 4 < total; q += (long long)gridDim.x * blockDim.x) {
        const long long i = q * 4;
        const long long n = i / (3LL * HW);
        const int c = (int)((i / HW) % 3);
        const int p = (int)(i % HW);
        float z[4];
        if (eps) { const float4 e4 = *reinterpret_cast<const float4*>(eps + i); z[0] = e4.x; z[1] = e4.y; z[2] = e4.z; z[3] = e4.w; }
        else normal4(seed, step, quad0 + (unsigned long long)q, z);
        const float4 x4 = *reinterpret_cast<const float4*>(x + i);
        const float4 e4 = *reinterpret_cast<const float4*>(et + ((size_t)n * Cet + c) * HW + p);
        const float4 y4 = *reinterpret_cast<const float4*>(y + i);
        const float4 m4 = *reinterpret_cast<const float4*>(mask + n * HW + p);
        const float xv[4] = {x4.x, x4.y, x4.z, x4.w}, ev[4] = {e4.x, e4.y, e4.z, e4.w};
        const float yv[4] = {y4.x, y4.y, y4.z, y4.w}, mv[4] = {m4.x, m4.y, m4.z, m4.w};
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float x0 = (xv[k] - ev[k] * co.sqrt_1m_at) / co.sqrt_at;
            const float x0h = x0 - mv[k] * (mv[k] * x0 - yv[k]);
            o[k] = co.sqrt_at_next * x0h + co.sigma_t * (co.c1 * z[k] + co.c2 * ev[k]);
        }
        *reinterpret_cast<float4*>(x + i) = make_float4(o[0], o[1], o[2], o[3]);
    }
}
int ddnm_update(float* x, const float* et, int Cet, const float* y, const float* mask, const float* eps,
                unsigned long long seed, unsigned long long step, DdnmCoef co, int N, int HW, hipStream_t s,
                unsigned long long quad0) {
    PD_REQUIRE(HW % 4 == 0, "ddnm_update: H*W must be a multiple of 4");
    const long long total = (long long)N * 3 * HW;
    k_ddnm_update<<<(int)std::min<long long>((total / 4 + 255) / 256, 4096), 256, 0, s>>>(x, et, Cet, y, mask, eps, seed, step, co,
                                                                                       HW, total, quad0);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}

__global__ void k_ddnm_finish(const float* __restrict__ x, float* __restrict__ out, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = fminf(fmaxf((x[i] + 1.0f) / 2.0f, 0.0f), 1.0f);
}
int ddnm_finish(const float* x, float* out, long long n, hipStream_t s) {
    k_ddnm_finish<<<(int)std::min<long long>((n + 255) / 256, 4096), 256, 0, s>>>(x, out, n);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}

// Calibration only (bench.py `roofline.calibration.copy16_gbs`): the plain 16-byte-per-lane streaming copy the HBM-bound
// passes (k_gn_apply, k_gn_skip_w1) are judged against -- MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy.  U 16-byte loads in
// flight per lane, one workgroup-stride sweep, nothing else.
// MODE 0: grid-stride sweep; 1: the same with nontemporal loads and stores; 2: every workgroup copies one contiguous slab (U x 4 KiB
// per trip)
typedef __attribute__((ext_vector_type(4))) float cp_f4;
template <int U, int MODE>
__global__ __launch_bounds__(256) void k_copy16(const cp_f4* __restrict__ src, cp_f4* __restrict__ dst, long long n16) {
    if (MODE == 2) {
        const long long per = (n16 + gridDim.x - 1) / gridDim.x, b0 = (long long)blockIdx.x * per, b1 = b0 + per < n16 ? b0 + per : n16;
        long long i = b0 + threadIdx.x;
        for (; i + (U - 1) * 256 < b1; i += U * 256) {
            cp_f4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = src[i + u * 256];
#pragma unroll
            for (int u = 0; u < U; ++u) dst[i + u * 256] = v[u];
        }
        for (; i < b1; i += 256) dst[i] = src[i];
        return;
    }
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        cp_f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = MODE == 1 ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (MODE == 1) __builtin_nontemporal_store(v[u], dst + i + u * stride); else dst[i + u * stride] = v[u]; }
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}
int copy16(const void* src, void* dst, long long bytes, int blocks, int unroll, hipStream_t s) {
    PD_REQUIRE(src && dst && bytes > 0 && bytes % 16 == 0, "copy16: bytes must be a positive multiple of 16");
    const long long n16 = bytes / 16;
    if (blocks <= 0) blocks = 256 * 8;
    const int mode = unroll >> 4, u = unroll & 15;         // unroll: low 4 bits = loads in flight per lane, bits 4.. = mode
#define PD_CP(U_, M_) k_copy16<U_, M_><<<blocks, 256, 0, s>>>((const cp_f4*)src, (cp_f4*)dst, n16)
#define PD_CPM(M_) (u == 8 ? PD_CP(8, M_) : u == 2 ? PD_CP(2, M_) : u == 1 ? PD_CP(1, M_) : PD_CP(4, M_))
    if (mode == 1) PD_CPM(1); else if (mode == 2) PD_CPM(2); else PD_CPM(0);
#undef PD_CPM
#undef PD_CP
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}

}  // namespace pdnn
