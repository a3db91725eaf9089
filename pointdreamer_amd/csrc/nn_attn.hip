// Row U1: QKVAttentionLegacy (models/DDNM/guided_diffusion/unet.py:328-354) as one fused flash-style kernel.
// qkv [N,T,3C] f16 NHWC (per head h: q | k | v at channels h*3D .. h*3D+3D), out [N,T,C] f16.
// weight = softmax_s( (q*D^-1/4) . (k*D^-1/4) ) in f32, a = weight @ v.
// Workgroup = 4 waves = 64 queries of one (image, head); keys/values stream through LDS in chunks of 64;
// QK^T and PV on v_mfma_f32_16x16x32_f16 with f32 online softmax (row reductions by 16-lane shuffles);
// P goes C-layout -> A-layout through a per-wave LDS tile, V is stored transposed so the PV B-fragment is one
// ds_read_b128.  T <= 1024 and 0.5 % of the UNet's FLOPs: MFMA-bound in principle, latency-bound in practice.
#include "nn_common.h"
using namespace pdhip;
namespace pdnn {

#define KC 64          // keys per chunk
#define QT 64          // queries per workgroup
#define LDP 72         // padded leading dimension (halfs) of K / Vt / P tiles: 144 B rows -> conflict-free b128 reads

template <int D>
__global__ __launch_bounds__(256) void k_attention(const half_t* __restrict__ qkv, half_t* __restrict__ out, int T, int C,
                                                   float scale2) {
    __shared__ __attribute__((aligned(16))) half_t Ks[KC * LDP];          // [key][d]
    __shared__ __attribute__((aligned(16))) half_t Vt[D * LDP];           // [d][key]
    __shared__ __attribute__((aligned(16))) half_t Ps[4][16 * LDP];       // per wave [query][key]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int heads = C / D;
    const int qt = blockIdx.x, h = blockIdx.y, n = blockIdx.z;
    const size_t row_stride = (size_t)3 * C;
    const half_t* base = qkv + (size_t)n * T * row_stride + (size_t)h * 3 * D;
    constexpr int KS = D / 32;              // k-steps of QK^T
    constexpr int DT = D / 16;              // d-tiles of O

    // Q fragments (A operand): row = query lane&15, k = d chunk (lane>>4)*8 + 32*ks
    half8 qf[KS];
    {
        const int q = qt * QT + wave * 16 + (lane & 15);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            qf[ks] = *reinterpret_cast<const half8*>(base + (size_t)q * row_stride + ks * 32 + (lane >> 4) * 8);
    }
    float4_t o[DT];
#pragma unroll
    for (int j = 0; j < DT; ++j) o[j] = (float4_t){0.f, 0.f, 0.f, 0.f};
    float mrow[4], lrow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { mrow[r] = -INFINITY; lrow[r] = 0.f; }

    for (int k0 = 0; k0 < T; k0 += KC) {
        __syncthreads();                                           // previous chunk fully consumed
        // stage K chunk [64][D] and V chunk transposed [D][64]: thread -> (key, d-octet)
        for (int idx = tid; idx < KC * (D / 8); idx += 256) {
            const int key = idx / (D / 8), oc = idx - key * (D / 8);
            const half_t* src = base + (size_t)(k0 + key) * row_stride + D + oc * 8;
            const half8 kv = *reinterpret_cast<const half8*>(src);
            const half8 vv = *reinterpret_cast<const half8*>(src + D);
            *reinterpret_cast<half8*>(&Ks[key * LDP + oc * 8]) = kv;
#pragma unroll
            for (int e = 0; e < 8; ++e) Vt[(oc * 8 + e) * LDP + key] = vv[e];
        }
        __syncthreads();
        // S = Q K^T : 4 key tiles of 16
        float4_t s[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s[j] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const half8 kf = *reinterpret_cast<const half8*>(&Ks[(j * 16 + (lane & 15)) * LDP + ks * 32 + (lane >> 4) * 8]);
                s[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qf[ks], kf, s[j], 0, 0, 0);
            }
        }
        // online softmax; this lane holds rows (lane>>4)*4 + r, columns (lane&15) + 16 j
        float alpha[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < 4; ++j) { s[j][r] *= scale2; mx = fmaxf(mx, s[j][r]); }
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
            const float mnew = fmaxf(mrow[r], mx);
            alpha[r] = __expf(mrow[r] - mnew);
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) { s[j][r] = __expf(s[j][r] - mnew); sum += s[j][r]; }
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) sum += __shfl_xor(sum, off);
            lrow[r] = lrow[r] * alpha[r] + sum;
            mrow[r] = mnew;
        }
#pragma unroll
        for (int j = 0; j < DT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[j][r] *= alpha[r];
        // P: C-layout -> LDS -> A-layout
        half_t* P = Ps[wave];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) P[((lane >> 4) * 4 + r) * LDP + j * 16 + (lane & 15)] = (half_t)s[j][r];
        __builtin_amdgcn_s_waitcnt(0xc07f);                        // lgkmcnt(0): this wave's P writes are visible to itself
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {                           // 64 keys = 2 k-steps of 32
            const half8 pf = *reinterpret_cast<const half8*>(&P[(lane & 15) * LDP + ks * 32 + (lane >> 4) * 8]);
#pragma unroll
            for (int j = 0; j < DT; ++j) {
                const half8 vf = *reinterpret_cast<const half8*>(&Vt[(j * 16 + (lane & 15)) * LDP + ks * 32 + (lane >> 4) * 8]);
                o[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pf, vf, o[j], 0, 0, 0);
            }
        }
    }
    // normalise and store: row = query (lane>>4)*4 + r, col = d = j*16 + (lane&15)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int q = qt * QT + wave * 16 + (lane >> 4) * 4 + r;
        const float inv = 1.0f / lrow[r];
#pragma unroll
        for (int j = 0; j < DT; ++j)
            out[((size_t)n * T + q) * C + h * D + j * 16 + (lane & 15)] = (half_t)(o[j][r] * inv);
    }
    (void)heads;
}

int attention(const half_t* qkv, half_t* out, int N, int T, int C, int D, hipStream_t s) {
    PD_REQUIRE(D == 64 || D == 32, "attention: head dim must be 32 or 64 (got %d)", D);
    PD_REQUIRE(C % D == 0 && T % QT == 0, "attention: need C %% D == 0 and T %% 64 == 0 (T=%d C=%d)", T, C);
    dim3 g(T / QT, C / D, N);
    const float scale2 = 1.0f / sqrtf((float)D);
    if (D == 64) k_attention<64><<<g, 256, 0, s>>>(qkv, out, T, C, scale2);
    else k_attention<32><<<g, 256, 0, s>>>(qkv, out, T, C, scale2);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}

}  // namespace pdnn
