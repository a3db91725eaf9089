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

// ------------------------------------------------------------------------------------------------
// T >= 128: transposed formulation.  S^T = K Q^T puts one QUERY in every lane column (lane & 15) and 4 consecutive keys in its
// registers, so (i) the online-softmax statistics are per lane (2 shuffle steps over the 4 key groups instead of 4 over 16
// columns, and the rescale factor is a per-lane scalar), (ii) P^T is ALREADY the B operand of O^T = V^T P^T: the PV k-slots are
// permuted to the keys a lane holds (key order inside a sum is free), no LDS round trip for P, (iii) V^T rows (one d, keys
// contiguous) are what the A operand wants -- V is transposed once per attention block by k_transpose_v (8 MB at 32x32,
// ~5 us) instead of with scalar LDS writes in every workgroup.  32 queries per wave (2 column tiles), K / V^T chunks of 64 keys
// double-buffered in LDS by LDS-DMA (16-byte slot swizzle slot ^ ((row >> 1) & 7) on the 128-byte rows).
__global__ __launch_bounds__(256) void k_transpose_v(const half_t* __restrict__ qkv, half_t* __restrict__ vt, int T, int C, int D) {
    __shared__ half_t tile[64][66];
    const int n = blockIdx.z, h = blockIdx.y, t0 = blockIdx.x * 64;
    const half_t* src = qkv + ((size_t)n * T + t0) * 3 * C + (size_t)h * 3 * D + 2 * D;
    for (int i = threadIdx.x; i < 64 * D; i += 256) {
        const int t = i / D, d = i - t * D;
        tile[t][d] = src[(size_t)t * 3 * C + d];
    }
    __syncthreads();
    half_t* dst = vt + (((size_t)n * (C / D) + h) * D) * T + t0;
    for (int i = threadIdx.x; i < 64 * D; i += 256) {
        const int d = i / 64, t = i - d * 64;
        dst[(size_t)d * T + t] = tile[t][d];
    }
}

thread_local int g_attn_nbuf = 0;      // tuning hook: 2 / 3 LDS chunk buffers of k_attention_t64, 0 = automatic
thread_local int g_attn_qtn = 0;       // tuning hook: 1 / 2 = 64 / 128 queries per workgroup, 0 = automatic
thread_local int g_attn_vt = 0;        // tuning hook: 1 = transposed-V workspace form (k_transpose_v + plain reads), 0 = LDS transpose reads
#define PD_ATT_DSR128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define PD_ATT_DSR64T(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define PD_ATT_DSR64(dst, addr, off) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
typedef __attribute__((address_space(3))) void lds_void_a;
typedef const __attribute__((address_space(1))) void gbl_void_a;
__device__ __forceinline__ void glds16a(const void* g, void* l) { __builtin_amdgcn_global_load_lds((gbl_void_a*)g, (lds_void_a*)l, 16, 0, 0); }

// all-reduce over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48) without the LDS: v_permlane16_swap exchanges the odd
// rows of its first operand with the even rows of its second, v_permlane32_swap the upper half of the first with the lower half of the
// second -- fed the same value twice, the two results hold both partners of every lane.  (__shfl_xor compiles to ds_bpermute_b32: an
// LDS round trip and an lgkmcnt(0) wait on the softmax's critical path, four per query tile and chunk.)
__device__ __forceinline__ float xr_max(float x) {
    // (v_max_f32 through asm: fmaxf on the bit-cast halves would be preceded by a canonicalising v_max x, x each)
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    unsigned m;
    asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(a[0]), "v"(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(m, m, false, false);
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(b[0]), "v"(b[1]));
    return r;
}

// QTN: 16-query tiles per wave (2: 128 queries per workgroup; 1: 64 -- twice the workgroups for the small grids of batch 1-2, where
// 8 heads x 8 query blocks are 64 workgroups on 256 CUs)
// NBUF: LDS chunk buffers; 3 = K / V^T chunks requested TWO iterations ahead (the one-workgroup-per-CU grids of batch 1-2 have no
// other wave to cover an L2 round trip with)
// TR: V is staged ROW-major straight from the qkv tensor (like K) and its fragments are read with ds_read_b64_tr_b16, the gfx950 LDS
// transpose read: inside a 16-lane group lane i supplies the 8-byte address of (row i >> 2, columns 4 (i & 3) ..+3) of a 4 x 16 block
// and receives column i of it (tools/ub/ub_tr.py prints the mapping) -- exactly "keys 4 g4 .. 4 g4 + 3 of d = r16", the A operand of
// O^T = V^T P^T.  No k_transpose_v pass, no V^T workspace.  TR = false: the transposed-V workspace form (lab hook).
// Register budget: left alone the compiler takes 208 registers for QTN = 2 (two waves per SIMD); held to three waves per SIMD it fits
// 160 without spilling, and the third wave is what covers the per-chunk barrier and the LDS-DMA issue stalls of the other two.
template <int QTN, int NBUF, bool TR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((QTN == 2 || NBUF == 3) ? 3 : 4, (QTN == 2 || NBUF == 3) ? 3 : 4))) void k_attention_t64(const half_t* __restrict__ qkv, const half_t* __restrict__ vt,
                                                       half_t* __restrict__ out, int T, int C, float scale2) {
    constexpr int D = 64, KCH = 64, QPW = 16 * QTN, QPB = 4 * QPW;
    __shared__ __attribute__((aligned(16))) char lds[NBUF][2][KCH * 128];   // [buffer][K | Vt][64 rows x 128 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform: LDS-DMA destinations in SGPRs)
    const int r16 = lane & 15, g4 = lane >> 4;
    const int heads = C / D;
    const float scale2l = scale2 * 1.4426950408889634f;            // softmax in the log2 domain: exp(x) = 2^(x log2 e)
    // 1-D grid, XCD-aware: workgroup b runs on XCD b % 8 (private L2).  All T / 128 query blocks of one (image, head) read the same
    // K / V^T, so they are given CONSECUTIVE slots of ONE XCD -- with the query block as the fastest grid index the 8 query blocks
    // of a head at 32x32 landed on 8 different XCDs and every L2 fetched the same K / V^T (3.2x the algorithmic HBM reads,
    // profiles/r02_pmc_kernels.json).  N * heads is a multiple of 8 (8 or 16 heads).
    const int nqb = T / QPB;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int qb = slot % nqb, grp = (slot / nqb) * 8 + xcd;           // grp = n * heads + h
    const int n = grp / heads, h = grp - n * heads;
    const size_t row_stride = (size_t)3 * C;
    const half_t* base = qkv + (size_t)n * T * row_stride + (size_t)h * 3 * D;
    const half_t* vbase = TR ? nullptr : vt + (((size_t)n * heads + h) * D) * T;
    // Q fragments (B operand): column = query r16 of tile qt, k = d chunk ks*32 + 8 g4
    half8 qf[QTN][2];
#pragma unroll
    for (int qt = 0; qt < QTN; ++qt) {
        const int q = qb * QPB + wave * QPW + qt * 16 + r16;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[qt][ks] = *reinterpret_cast<const half8*>(base + (size_t)q * row_stride + ks * 32 + g4 * 8);
    }
    (void)scale2l;
    float4_t o[QTN][4];
    // the softmax denominators are a fifth accumulator tile of the PV product: V^T extended by a row of ones gives
    // l[q] = sum_k P[k][q] from the MATRIX pipe (every register of the tile holds the lane's query's sum) -- the kernel is bound by
    // VALU issue, the 16 adds + 2 cross-row exchanges per query tile and chunk were a fifth of it.  (The sum is then over the
    // f16-rounded weights, i.e. exactly the weights the numerator uses.)
    float mrun[QTN];
    float4_t ol[QTN];
#pragma unroll
    for (int qt = 0; qt < QTN; ++qt) {
        mrun[qt] = -INFINITY; ol[qt] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[qt][dt] = (float4_t){0.f, 0.f, 0.f, 0.f};
    }
    // loader: wave w stages rows 16 w .. 16 w + 15 of the K chunk and of the V^T chunk (2 + 2 pieces of 8 rows x 128 B)
    const int lrow = lane >> 3, lslot = lane & 7;
    // running source pointers of this lane's four pieces, advanced by one chunk per call (chunks are staged in order): the 64-bit
    // address arithmetic per piece (row * stride + ...) was ~40 VALU instructions per chunk in a loop that is issue-bound
    const half_t* kp[2];
    const half_t* vp[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int row = wave * 16 + p * 8 + lrow;                                       // key (K, V) / d (V^T workspace form)
        const int sl = lslot ^ ((row >> 1) & 7);                                        // source slot landing in physical slot lslot
        kp[p] = base + (size_t)row * row_stride + D + sl * 8;
        // TR: V row = key, 128 B = 64 d.  Slot swizzle 2 ((key >> 1) & 3): the 8 keys x 32 B a 32-lane half reads tile one bank row
        vp[p] = TR ? base + (size_t)row * row_stride + 2 * D + (lslot ^ (2 * ((row >> 1) & 3))) * 8 : vbase + (size_t)row * T + sl * 8;
    }
    const size_t kstep = (size_t)KCH * row_stride, vstep = TR ? kstep : (size_t)KCH;
    auto stage = [&](int buf, int) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            glds16a(kp[p], &lds[buf][0][(wave * 16 + p * 8) * 128]);
            glds16a(vp[p], &lds[buf][1][(wave * 16 + p * 8) * 128]);
            kp[p] += kstep; vp[p] += vstep;
        }
    };
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)&lds[0][0][0];
    uint32_t koff[2], voff[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) koff[ks] = r16 * 128 + (((ks * 4 + g4) ^ (r16 >> 1)) << 4);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < 2; ++j) voff[u][j] = r16 * 128 + (((4 * u + 2 * j + (g4 >> 1)) ^ (r16 >> 1)) << 4) + (g4 & 1) * 8;
    // TR: this lane supplies key 4 g4 + (r16 >> 2) (+ 32 u + 16 j: immediate), d = 16 dt + 4 (r16 & 3); the swizzle term of the key
    // is (2 g4 + (r16 >> 3)) & 3 for every (u, j), so one address per dt serves all four reads of that dt
    uint32_t vtr[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
        vtr[dt] = (4 * g4 + (r16 >> 2)) * 128 + ((2 * (dt ^ ((2 * g4 + (r16 >> 3)) & 3)) + ((r16 & 3) >> 1)) << 4) + (r16 & 1) * 8;
    // the Q fragments are complete BEFORE the first chunk is requested: left to the compiler, their s_waitcnt vmcnt(0) lands at
    // the first MFMA -- inside the loop, behind the prefetch, draining it in every iteration
#pragma unroll
    for (int qt = 0; qt < QTN; ++qt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) asm volatile("" : "+v"(qf[qt][ks]));
    const int nch = T / KCH;
    stage(0, 0);
    if (NBUF == 3 && nch > 1) stage(1, KCH);
    int buf = 0;
    for (int ch = 0; ch < nch; ++ch) {
        if (NBUF == 3 && ch + 1 < nch) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");    // chunk ch+1 (4 pieces per lane) stays in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // chunk `ch` landed for every wave; chunk ch-1 is fully consumed (its fragments fed MFMAs this wave has already issued).
        // Raw barrier: __syncthreads() would drain the LDS-DMA queue and with it the chunk that is meant to stay in flight.
        if (NBUF == 3) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
        else __syncthreads();
        // restage the buffer read in iteration ch-1 (one barrier per chunk)
        if (NBUF == 3) { if (ch + 2 < nch) stage(buf == 0 ? 2 : buf - 1, (ch + 2) * KCH); }
        else if (ch + 1 < nch) stage(buf ^ 1, (ch + 1) * KCH);
        // Fragment reads are inline-asm ds_reads: behind a plain C++ read of `lds` the compiler drains the LDS-DMA queue first
        // (s_waitcnt vmcnt(0) -- it cannot tell the chunk being prefetched from the one being read), which turned every chunk
        // into a synchronous L2 round trip.  LDS returns in issue order; the waits below are counted by hand.
        const uint32_t kb = lds0 + buf * (2 * KCH * 128), vb = kb + KCH * 128;
        // K fragments (A operand of S^T): row = key 16 kt + r16, slot ks*4 + g4 (swizzle (row >> 1) & 7 = r16 >> 1 for every kt)
        half8 kf[4][2];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) PD_ATT_DSR128(kf[kt][ks], kb + koff[ks], kt * 2048);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[0][0]), "+v"(kf[0][1]), "+v"(kf[1][0]), "+v"(kf[1][1]), "+v"(kf[2][0]), "+v"(kf[2][1]),
                     "+v"(kf[3][0]), "+v"(kf[3][1]));
        // V^T fragments (A operand of O^T): row d = 16 dt + r16; k-slots = keys 32u + 4 g4 + {0..3} and 32u + 16 + 4 g4 + {0..3}.
        // Requested now, needed after the softmax: their LDS latency sits under the QK^T MFMAs and the exponentials.
        half4 vlo[4][2], vhi[4][2];
#pragma unroll
        for (int dt = 0; dt < 3; ++dt)                             // (12 requests now, the last 4 below: the counter has 4 bits)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (TR) {
                    PD_ATT_DSR64T(vlo[dt][u], vb + vtr[dt], u * 4096);
                    PD_ATT_DSR64T(vhi[dt][u], vb + vtr[dt], u * 4096 + 2048);
                } else {
                    PD_ATT_DSR64(vlo[dt][u], vb + voff[u][0], dt * 2048);
                    PD_ATT_DSR64(vhi[dt][u], vb + voff[u][1], dt * 2048);
                }
            }
        half8 pf[QTN][2];                                          // [query tile][k-step of 32 keys]
        // The query tiles go through the softmax side by side in straight-line code (one rescale branch for all of them): a branch per
        // tile cut the block in two and left each tile's max -> exp2 -> convert chain to run alone.
        float4_t st[QTN][4];
        float mnew[QTN];
#pragma unroll
        for (int qt = 0; qt < QTN; ++qt)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                st[qt][kt] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) st[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kt][ks], qf[qt][ks], st[qt][kt], 0, 0, 0);
            }
        bool grew = false;
#pragma unroll
        for (int qt = 0; qt < QTN; ++qt) {
            // this lane: query r16 of the tile, keys 16 kt + 4 g4 + r
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[qt][kt][r]);
            mx = xr_max(mx);
            mnew[qt] = fmaxf(mrun[qt], mx * scale2l);                      // log2 domain; the scale is positive: max commutes with it
            grew |= mnew[qt] > mrun[qt];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) st[qt][kt][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[qt][kt][r], scale2l, -mnew[qt]));   // scale and shift in one FMA
        }
        // lazy rescale: the running maximum of a query stops moving after its first few key chunks -- when no lane's maximum grew
        // (wave-uniform test) alpha is exactly 1 for every lane and the multiplies of O and the exp2 are skipped
        if (__builtin_amdgcn_ballot_w64(grew) != 0ull) {
#pragma unroll
            for (int qt = 0; qt < QTN; ++qt) {
                const float alpha = __builtin_amdgcn_exp2f(mrun[qt] - mnew[qt]);
#pragma unroll
                for (int r = 0; r < 4; ++r) ol[qt][r] *= alpha;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[qt][dt][r] *= alpha;
            }
        }
#pragma unroll
        for (int qt = 0; qt < QTN; ++qt) {
            mrun[qt] = mnew[qt];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) { pf[qt][u][r] = (half_t)st[qt][2 * u][r]; pf[qt][u][4 + r] = (half_t)st[qt][2 * u + 1][r]; }
        }
        // O^T += V^T P^T (+ the row of ones)
        {
            const half8 ones = (half8){(half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f};
#pragma unroll
            for (int qt = 0; qt < QTN; ++qt)
#pragma unroll
                for (int u = 0; u < 2; ++u) ol[qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, pf[qt][u], ol[qt], 0, 0, 0);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            if (dt == 0) {
                asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(vlo[0][0]), "+v"(vhi[0][0]), "+v"(vlo[0][1]), "+v"(vhi[0][1]));
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (TR) {
                        PD_ATT_DSR64T(vlo[3][u], vb + vtr[3], u * 4096);
                        PD_ATT_DSR64T(vhi[3][u], vb + vtr[3], u * 4096 + 2048);
                    } else {
                        PD_ATT_DSR64(vlo[3][u], vb + voff[u][0], 3 * 2048);
                        PD_ATT_DSR64(vhi[3][u], vb + voff[u][1], 3 * 2048);
                    }
                }
            }
            if (dt == 1) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(vlo[1][0]), "+v"(vhi[1][0]), "+v"(vlo[1][1]), "+v"(vhi[1][1]));
            if (dt == 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(vlo[2][0]), "+v"(vhi[2][0]), "+v"(vlo[2][1]), "+v"(vhi[2][1]));
            if (dt == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vlo[3][0]), "+v"(vhi[3][0]), "+v"(vlo[3][1]), "+v"(vhi[3][1]));
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                half8 vf;
#pragma unroll
                for (int e = 0; e < 4; ++e) { vf[e] = vlo[dt][u][e]; vf[4 + e] = vhi[dt][u][e]; }
#pragma unroll
                for (int qt = 0; qt < QTN; ++qt) o[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[qt][u], o[qt][dt], 0, 0, 0);
            }
        }
        buf = (buf + 1 == NBUF) ? 0 : buf + 1;
    }
    // normalise and store: lane = query r16 of each tile, d = 16 dt + 4 g4 + r (8-byte packed stores)
#pragma unroll
    for (int qt = 0; qt < QTN; ++qt) {
        const int q = qb * QPB + wave * QPW + qt * 16 + r16;
        const float inv = 1.0f / ol[qt][0];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            half4 hv;
#pragma unroll
            for (int r = 0; r < 4; ++r) hv[r] = (half_t)(o[qt][dt][r] * inv);
            *reinterpret_cast<half4*>(out + ((size_t)n * T + q) * C + h * D + dt * 16 + g4 * 4) = hv;
        }
    }
}

// vt_ws: N*T*C halfs of workspace for the transposed V (only used when T >= 128 and D == 64); may be null -> the 64-query kernel
int attention(const half_t* qkv, half_t* out, int N, int T, int C, int D, hipStream_t s, half_t* vt_ws) {
    PD_REQUIRE(D == 64 || D == 32, "attention: head dim must be 32 or 64 (got %d)", D);
    PD_REQUIRE(C % D == 0 && T % QT == 0, "attention: need C %% D == 0 and T %% 64 == 0 (T=%d C=%d)", T, C);
    const float scale2 = 1.0f / sqrtf((float)D);
    if (vt_ws != nullptr && D == 64 && T % 128 == 0 && (N * (C / D)) % 8 == 0) {
        PD_REQUIRE((N * (C / D)) % 8 == 0, "attention: N * heads must be a multiple of 8");
        const bool small = g_attn_qtn == 1 || (g_attn_qtn != 2 && (T / 128) * (C / D) * N < 256);
        const int nbuf = g_attn_nbuf == 2 || g_attn_nbuf == 3 ? g_attn_nbuf : 3;
        const int gs = (small ? T / 64 : T / 128) * (C / D) * N;
#define ATT_LAUNCH(Q, B, R) k_attention_t64<Q, B, R><<<gs, 256, 0, s>>>(qkv, vt_ws, out, T, C, scale2)
#define ATT_BY_BUF(Q, R) do { if (nbuf == 3) ATT_LAUNCH(Q, 3, R); else ATT_LAUNCH(Q, 2, R); } while (0)
        if (g_attn_vt != 0) {
            k_transpose_v<<<dim3(T / 64, C / D, N), 256, 0, s>>>(qkv, vt_ws, T, C, D);
            if (small) ATT_BY_BUF(1, false); else ATT_BY_BUF(2, false);
        } else {
            if (small) ATT_BY_BUF(1, true); else ATT_BY_BUF(2, true);
        }
#undef ATT_BY_BUF
#undef ATT_LAUNCH
        PD_LAUNCH_CHECK();
        return PDHIP_OK;
    }
    dim3 g(T / QT, C / D, N);
    if (D == 64) k_attention<64><<<g, 256, 0, s>>>(qkv, out, T, C, scale2);
    else k_attention<32><<<g, 256, 0, s>>>(qkv, out, T, C, scale2);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}

}  // namespace pdnn
