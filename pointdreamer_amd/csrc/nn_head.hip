// Row U1, output head: GroupNorm(32) -> SiLU -> conv3x3(C -> 3|6) in float32-equivalent arithmetic, one kernel.
// Replaces `self.out = normalization(ch), SiLU(), conv_nd(..., 3, padding=1)` (models/DDNM/guided_diffusion/unet.py:613-617,
// applied at :676 after h.type(x.dtype) -- the head is outside convert_to_fp16, so it runs in f32 on the f16 body output).
//
// The conv is restated as  z[p][tap*NO + o] = sum_c a[p][c] * w[o][tap][c]  for every INPUT pixel p of a tile (+1 halo),
// then  y[q][o] = bias[o] + sum_tap z[q + d(tap)][tap*NO + o].  The first part is a [pixels x C] x [C x 9*NO] GEMM whose A
// operand is produced in registers straight from the HBM stream (one 16-byte NHWC load = one MFMA A fragment: GN affine ->
// SiLU in f32), so the activated f32 tensor (537 MB at 8x256x256x256) is never written or re-read 9 times.
// f32 accuracy on the f16 matrix cores: a = a_hi + a_lo, w = w_hi + w_lo (f16 pairs, 22 significant bits) and
// a*w ~ a_hi*w_hi + a_lo*w_hi + a_hi*w_lo accumulated in f32 -- relative error ~2^-21 per product.
// Algorithmic bytes: 2 B/element of the input read once (+ ~33 % halo from L2) + 4*NO B/pixel written.
#include "nn_common.h"
using namespace pdhip;
namespace pdnn {

#define HD_TW 32
#define HD_TH 8
#define HD_TPW 4                             // tiles per workgroup (stacked in y) at large batches; fewer when the grid would not fill the chip
#define HD_IW (HD_TW + 2)
#define HD_IH (HD_TH + 2)
#define HD_NPIX (HD_IW * HD_IH)              // 340 input pixels per tile
#define HD_NCH ((HD_NPIX + 15) / 16)         // 22 chunks of 16 pixels (one MFMA M-tile each)
#define HD_ZS_OF(NO) ((NO) == 6 ? 60 : 28)   // z row stride (floats): 4*60 mod 64 = 4*28 mod 64 = 48 -> the 4 row groups of a C tile hit disjoint banks

// packed head weights: [2 (hi, lo)][64 rows n = tap*NO + o (zero above 9*NO)][C] f16, 16-byte chunk index XOR-swizzled by n
// so that the ds_read_b128 B-fragment reads (16 rows x one chunk column) spread over all banks
__device__ __host__ __forceinline__ int hd_swz(int n, int C) { return n & 15 & (C / 8 - 1); }

__global__ void k_head_pack(const float* __restrict__ w /*[NO][9*C], k = tap*C + c*/, int NO, int C, half_t* __restrict__ wz) {
    const int total = 64 * C;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int n = i / C, c = i - n * C;
        float v = 0.f;
        if (n < 9 * NO) { const int tap = n / NO, o = n - tap * NO; v = w[(size_t)o * 9 * C + (size_t)tap * C + c]; }
        const half_t hi = (half_t)v;
        const half_t lo = (half_t)(v - (float)hi);
        const int dst = n * C + (((c >> 3) ^ hd_swz(n, C)) << 3) + (c & 7);
        wz[dst] = hi;
        wz[64 * C + dst] = lo;
    }
}

int head_pack(const float* w, int NO, int C, half_t* wz, hipStream_t s) {
    PD_REQUIRE((NO == 3 || NO == 6) && (C == 32 || C == 64 || C == 128 || C == 256), "head_pack: out channels 3|6, C in {32,64,128,256}");
    k_head_pack<<<cdiv(64 * C, 256), 256, 0, s>>>(w, NO, C, wz);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}

__device__ __forceinline__ float hd_silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }   // (1 ulp reciprocal: the kernel is VALU-bound, an IEEE divide costs 8 more instructions per element)

template <int NO, int C>
__global__ __launch_bounds__(512) void k_head(const half_t* __restrict__ X, const float* __restrict__ stats,
                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                              const half_t* __restrict__ wz, const float* __restrict__ bias,
                                              float* __restrict__ y, int H, int W, int tpw) {
    constexpr int KK = C / 32;                       // MFMA k-steps over the channels
    constexpr int NT = (9 * NO + 15) / 16;           // 16-column tiles of z actually needed
    constexpr int NR = NT * 16;                      // weight rows staged (of the 64 packed ones)
    constexpr int HD_ZS = HD_ZS_OF(NO);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* Bs = reinterpret_cast<half_t*>(smem);                               // [2][NR][C]
    float* zs = reinterpret_cast<float*>(smem + (size_t)2 * NR * C * 2);        // [HD_NCH*16][HD_ZS]
    float* gab = zs + HD_NCH * 16 * HD_ZS;                                      // [2][C]: GN scale, shift of this image
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.z, x0 = blockIdx.x * HD_TW;

    for (int i = tid; i < 2 * NR * C / 8; i += 512) {    // rows [0, NR) of the hi and of the lo plane
        const int pl = i / (NR * C / 8), r = i - pl * (NR * C / 8);
        reinterpret_cast<half8*>(Bs)[i] = reinterpret_cast<const half8*>(wz)[pl * (64 * C / 8) + r];
    }
    for (int c = tid; c < C; c += 512) {
        const int grp = c / (C / 32);
        const float mean = stats[((size_t)n * 32 + grp) * 2], rstd = stats[((size_t)n * 32 + grp) * 2 + 1];
        const float ga = rstd * gamma[c];
        gab[c] = ga;
        gab[C + c] = beta[c] - mean * ga;
    }
    __syncthreads();

    const int q4 = lane >> 4, r16 = lane & 15;
    // a workgroup walks HD_TPW tiles down its column: the 64 KB of packed weights are staged once for all of them
    for (int it = 0; it < tpw; ++it) {
    const int y0 = (blockIdx.y * tpw + it) * HD_TH;
    if (y0 >= H) break;
    auto src_of = [&](int chunk, bool* inimg) -> const half_t* {
        const int idx = chunk * 16 + r16;
        const int iy = idx / HD_IW, ix = idx - iy * HD_IW;
        const int gy = y0 - 1 + iy, gx = x0 - 1 + ix;
        *inimg = idx < HD_NPIX && gy >= 0 && gy < H && gx >= 0 && gx < W;
        return X + (((size_t)n * H + (*inimg ? gy : 0)) * W + (*inimg ? gx : 0)) * C + 8 * q4;
    };
    half8 vn[KK];
    bool in_n = false;
    if (wave < HD_NCH) {
        const half_t* sp = src_of(wave, &in_n);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) vn[kk] = *reinterpret_cast<const half8*>(sp + kk * 32);
    }
    for (int chunk = wave; chunk < HD_NCH; chunk += 8) {
        half8 v[KK];
        const bool inimg = in_n;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) v[kk] = vn[kk];
        if (chunk + 8 < HD_NCH) {                     // prefetch the wave's next chunk while this one is transformed
            const half_t* sp = src_of(chunk + 8, &in_n);
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) vn[kk] = *reinterpret_cast<const half8*>(sp + kk * 32);
        }
        float4_t acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const int c0 = kk * 32 + 8 * q4;
            const float4 g0 = *reinterpret_cast<const float4*>(gab + c0), g1 = *reinterpret_cast<const float4*>(gab + c0 + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(gab + C + c0), b1 = *reinterpret_cast<const float4*>(gab + C + c0 + 4);
            const float ga[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            const float gb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            half8 ah, al;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float a = hd_silu((float)v[kk][e] * ga[e] + gb[e]);
                if (!inimg) a = 0.f;                  // the conv zero-pads the ACTIVATED tensor
                const half_t hi = (half_t)a;
                ah[e] = hi;
                al[e] = (half_t)(a - (float)hi);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int nrow = t * 16 + r16;
                const int off = nrow * C + (((kk * 4 + q4) ^ hd_swz(nrow, C)) << 3);
                const half8 bh = *reinterpret_cast<const half8*>(Bs + off);
                const half8 bl = *reinterpret_cast<const half8*>(Bs + NR * C + off);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int col = t * 16 + r16;
            if (col < 9 * NO) {
#pragma unroll
                for (int r = 0; r < 4; ++r) zs[(chunk * 16 + q4 * 4 + r) * HD_ZS + col] = acc[t][r];
            }
        }
    }
    __syncthreads();

    // gather: output pixel q sums its 9 neighbours' z columns (fixed tap order)
    for (int item = tid; item < HD_TW * HD_TH * NO; item += 512) {
        const int o = item / (HD_TW * HD_TH), q = item - o * (HD_TW * HD_TH);
        const int qy = q / HD_TW, qx = q - qy * HD_TW;
        const int gy = y0 + qy, gx = x0 + qx;
        if (gy >= H || gx >= W) continue;
        float a = bias[o];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int iy = qy + tap / 3, ix = qx + tap % 3;          // input-tile coordinates of (gy + ky - 1, gx + kx - 1)
            a += zs[(iy * HD_IW + ix) * HD_ZS + tap * NO + o];
        }
        y[(((size_t)n * NO + o) * H + gy) * W + gx] = a;
    }
    __syncthreads();                                       // zs is rewritten by the next tile
    }
}

size_t head_smem_bytes(int C, int NO) {
    const int NR = ((9 * NO + 15) / 16) * 16;
    return (size_t)2 * NR * C * 2 + (size_t)HD_NCH * 16 * HD_ZS_OF(NO) * 4 + (size_t)2 * C * 4;
}

template <int NO, int C>
static int head_launch(const half_t* X, const float* stats, const float* gamma, const float* beta, const half_t* wz,
                       const float* bias, float* y, int N, int H, int W, hipStream_t s) {
    auto kern = k_head<NO, C>;
    const size_t smem = head_smem_bytes(C, NO);
    PD_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    // the packed weights are staged once per workgroup: 4 tiles per workgroup at large batches, fewer while that leaves < 512 workgroups
    // (UNet batch 1: 64 workgroups of 4 tiles were a quarter of the chip, 56 us; one tile each: 256 workgroups)
    int tpw = HD_TPW;
    while (tpw > 1 && (long long)cdiv(W, HD_TW) * cdiv(cdiv(H, HD_TH), tpw) * N < 512) tpw >>= 1;
    dim3 grid(cdiv(W, HD_TW), cdiv(cdiv(H, HD_TH), tpw), N);
    kern<<<grid, 512, smem, s>>>(X, stats, gamma, beta, wz, bias, y, H, W, tpw);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}

int head_gn_silu_conv3x3(const half_t* X, const float* stats, const float* gamma, const float* beta, const half_t* wz,
                         const float* bias, float* y_nchw, int N, int H, int W, int C, int NO, hipStream_t s) {
    PD_REQUIRE(NO == 3 || NO == 6, "head: out channels must be 3 or 6");
#define HL(NO_, C_) head_launch<NO_, C_>(X, stats, gamma, beta, wz, bias, y_nchw, N, H, W, s)
    if (C == 256) return NO == 6 ? HL(6, 256) : HL(3, 256);
    if (C == 128) return NO == 6 ? HL(6, 128) : HL(3, 128);
    if (C == 64) return NO == 6 ? HL(6, 64) : HL(3, 64);
    if (C == 32) return NO == 6 ? HL(6, 32) : HL(3, 32);
#undef HL
    PD_REQUIRE(false, "head: final channel count must be 32, 64, 128 or 256 (got %d)", C);
    return PDHIP_OK;
}

}  // namespace pdnn
