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

__device__ __forceinline__ void sk_glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((sk_gbl_void*)gsrc, (sk_lds_void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ int sk_swz(int row) { return (row >> 1) & 7; }       // 128-byte rows: 16-byte slot ^= (row >> 1) & 7

template <int N> __device__ __forceinline__ void sk_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// grid: total_tiles * splits workgroups (1-D).  tickets[tile] must be zero at launch; the last arriver of a tile resets it.
template <int TAPS, int BM, int BN, int NST>
__global__ __launch_bounds__(256) void k_conv_sk(const half_t* __restrict__ X, const half_t* __restrict__ Wt, const float* __restrict__ bias,
                                                 const half_t* __restrict__ residual, half_t* __restrict__ Y, int N, int H, int W, int Cin,
                                                 int Cout, int n_tiles, int total_tiles, const half_t* __restrict__ zero_page, int splits,
                                                 float* __restrict__ slabs, unsigned* __restrict__ tickets, float* __restrict__ gn_part,
                                                 const half_t* __restrict__ X2, int Cin1) {
    constexpr int ROWB = 128, RPI = 8;                 // bytes per tile row (K-step 64), rows per 1 KiB wave-instruction
    constexpr int LPO = BM / 32, LPB = BN / 32;        // LDS-DMA pieces per wave per K-step: activation rows, weight rows
    constexpr int OPS = LPO + LPB;
    constexpr int TM = BM / 32, TN = BN / 32;          // accumulator tiles per wave (wave tile BM/2 x BN/2)
    constexpr int A_BYTES = BM * ROWB, STAGE_BYTES = (BM + BN) * ROWB;
    constexpr int CS_LD = BN + 8;
    constexpr int D = NST - 1;                         // prefetch distance
    constexpr int FLAG_OFF = NST * STAGE_BYTES;        // "I drew the last ticket" (ONE __shared__ object: cdna guide section 5 trap 4a)
    static_assert(NST >= 2 && NST <= 4 && OPS * 2 <= 63, "stage count / vmcnt range");
    static_assert(BM * CS_LD * 2 <= NST * STAGE_BYTES && 256 * 2 * 4 <= NST * STAGE_BYTES, "epilogue staging must fit the stages");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware work id: workgroup b runs on XCD b % 8; every XCD gets a contiguous run of (tile, split) ids, so the K-slices of a
    // tile and the n-tiles sharing an activation tile sit on one L2
    int wid;
    {
        const int total = total_tiles * splits;
        const int b = blockIdx.x, q = total >> 3, r = total & 7, xcd = b & 7, i = b >> 3;
        wid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
    }
    const int tile = wid / splits, split = wid - tile * splits;
    const int m0 = (tile / n_tiles) * BM, n0 = (tile % n_tiles) * BN;
    const long long M = (long long)N * H * W;
    const int K = TAPS * Cin;
    const int kc = Cin >> 6;                           // K-steps per tap
    const int KI = TAPS * kc;

    // ---- loader role (as k_conv_igemm): lane stages 16-byte slot (lane % 8) of row (lane / 8) of each of its pieces
    const int lrow = lane >> 3, lpos = lane & 7;
    int py[LPO], pxx[LPO];
    long long pbase[LPO], pbase2[LPO];
    const half_t* ap[LPO];
    int astep[LPO];
    const half_t* bp[LPB];
    const long long zoff = zero_page - X;
#pragma unroll
    for (int i = 0; i < LPO; ++i) {
        const int r = wave * (BM / 4) + i * RPI + lrow;
        const int c = lpos ^ sk_swz(r);
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
        const int r = wave * (BN / 4) + i * RPI + lrow;
        bp[i] = Wt + (size_t)(n0 + r) * K + (lpos ^ sk_swz(r)) * 8;
    }
    auto set_tap = [&](int tap) {
        const int dy = (TAPS == 1) ? 0 : tap / 3 - 1, dx = (TAPS == 1) ? 0 : tap - (tap / 3) * 3 - 1;
#pragma unroll
        for (int i = 0; i < LPO; ++i) {
            const int yy = py[i] + dy, xx = pxx[i] + dx;
            const bool ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W);
            const long long off = pbase[i] + ((long long)dy * W + dx) * Cin1;
            ap[i] = X + (ok ? off : zoff);
            astep[i] = ok ? 64 : 0;
        }
    };
    const int it0 = (int)((long long)KI * split / splits), it1 = (int)((long long)KI * (split + 1) / splits);
    int ntap = it0 / kc, nc = it0 - (it0 / kc) * kc;
    set_tap(ntap);
    if (TAPS == 1 && X2 != nullptr && nc >= (Cin1 >> 6)) {        // a split that starts inside the concat's second tensor
#pragma unroll
        for (int i = 0; i < LPO; ++i) ap[i] = X2 + pbase2[i] + (long long)(nc - (Cin1 >> 6)) * 64;
    } else {
#pragma unroll
        for (int i = 0; i < LPO; ++i) ap[i] += astep[i] * nc;
    }
#pragma unroll
    for (int i = 0; i < LPB; ++i) bp[i] += (size_t)it0 * 64;
    char* const wave_dst_a = smem + wave * ((BM / 4) * ROWB);
    char* const wave_dst_b = smem + A_BYTES + wave * ((BN / 4) * ROWB);
    auto issue = [&](int stage) {
#pragma unroll
        for (int p = 0; p < LPO; ++p) { sk_glds16(ap[p], wave_dst_a + stage * STAGE_BYTES + p * 1024); ap[p] += astep[p]; }
#pragma unroll
        for (int p = 0; p < LPB; ++p) { sk_glds16(bp[p], wave_dst_b + stage * STAGE_BYTES + p * 1024); bp[p] += 64; }
        if (TAPS == 1) {
            if (X2 != nullptr && ++nc == (Cin1 >> 6)) {           // the concat's first tensor is exhausted: continue in the second
#pragma unroll
                for (int i = 0; i < LPO; ++i) ap[i] = X2 + pbase2[i];
            }
        } else if (++nc == kc) {
            nc = 0;
            if (++ntap < TAPS) set_tap(ntap);
        }
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

#pragma unroll
    for (int d = 0; d < D; ++d)
        if (it0 + d < it1) issue(d);
    int cur = 0, nxt = D % NST;
    for (int it = it0; it < it1; ++it) {
        const int rem = it1 - it;                          // stages in flight at this point: min(D, rem) (NST 2: this one only)
        if (D >= 3 && rem >= 3) sk_wait_vm<2 * OPS>();
        else if (D >= 2 && rem >= 2) sk_wait_vm<OPS>();
        else sk_wait_vm<0>();
        __builtin_amdgcn_s_barrier();                      // stage `cur` landed for every wave; stage `nxt` is free again
        asm volatile("" ::: "memory");
        if (it + D < it1) issue(nxt);
        const char* As = smem + cur * STAGE_BYTES + (wm * (BM / 2)) * ROWB;
        const char* Bs = smem + cur * STAGE_BYTES + A_BYTES + (wn * (BN / 2)) * ROWB;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            half8 a[TM], b[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const half8*>(Bs + j * 16 * ROWB + frag_off[kk]);
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const half8*>(As + i * 16 * ROWB + frag_off[kk]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j], a[i], acc[i][j], 0, 0, 0);
        }
        cur = (cur + 1 == NST) ? 0 : cur + 1;
        nxt = (nxt + 1 == NST) ? 0 : nxt + 1;
    }
    __syncthreads();                                       // all fragment reads done before the stages are reused

    // ---- in-launch split-K combine
    if (splits > 1) {
        constexpr int F = TM * TN;                         // float4 pieces per thread
        constexpr int SLAB_BYTES = BM * BN * 4;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(slabs + (size_t)tile * splits * (BM * BN), 0, splits * SLAB_BYTES, 0x00020000);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rs,
                                                       split * SLAB_BYTES + ((i * TN + j) * 256 + tid) * 16, 0, /*sc1: write-through*/ 16);
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
        constexpr int U = 32 / F >= 1 ? 32 / F : 1;
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
                    v[u][f] = __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, sp * SLAB_BYTES + (f * 256 + tid) * 16, 0, /*sc1*/ 16));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float keep = sp0 + u < splits ? 1.0f : 0.0f;
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    float4_t& a = acc[f / TN][f % TN];
                    if (sp0 + u < splits) { a[0] += v[u][f][0]; a[1] += v[u][f][1]; a[2] += v[u][f][2]; a[3] += v[u][f][3]; }
                    (void)keep;
                }
            }
        }
    }

    // ---- epilogue: acc (+bias) -> f16 -> LDS [BM][CS_LD] -> coalesced 16-byte rows (+residual) + GroupNorm octet partials.
    // Transposed accumulators (weights x activations): lane holds pixel 16 i + (lane & 15), channels 16 j + 4 (lane >> 4) + 0..3
    half_t* Cs = reinterpret_cast<half_t*>(smem);
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
    __syncthreads();
    constexpr int CT = BN / 8;                             // column threads (one channel octet each)
    constexpr int RPP = 256 / CT;                          // rows per pass
    const int col8 = (tid % CT) * 8;
    float gs = 0.f, gq = 0.f;
#pragma unroll
    for (int p = 0; p < BM / RPP; ++p) {
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
            *reinterpret_cast<half8*>(Y + o) = v;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; gs += f; gq += f * f; }
        }
    }
    if (gn_part != nullptr) {                              // host guarantees the tile lies inside one image (H * W % BM == 0)
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);
        red[tid * 2] = gs; red[tid * 2 + 1] = gq;
        __syncthreads();
        if (tid < CT && n0 + tid * 8 < Cout) {
            float s1 = 0.f, q1 = 0.f;
            for (int r = 0; r < RPP; ++r) { s1 += red[(r * CT + tid) * 2]; q1 += red[(r * CT + tid) * 2 + 1]; }
            const int hw = H * W, chunks = hw / BM;
            const int img = m0 / hw, chunk = (m0 - img * hw) / BM;
            float* dst = gn_part + (((size_t)img * chunks + chunk) * (Cout >> 3) + (n0 >> 3) + tid) * 2;
            dst[0] = s1; dst[1] = q1;
        }
    }
}

template <int TAPS, int BM, int BN, int NST>
int launch_sk(int grid, hipStream_t s, const half_t* X, const half_t* Wt, const float* bias, const half_t* residual, half_t* Y, int N, int H,
              int W, int Cin, int Cout, int n_tiles, int total, const half_t* zero_page, int splits, float* slabs, unsigned* tickets,
              float* gnp, const half_t* X2, int Cin1) {
    auto kern = k_conv_sk<TAPS, BM, BN, NST>;
    const size_t smem = (size_t)NST * (BM + BN) * 128 + 16;
    if (smem > 65536) PD_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    kern<<<grid, 256, smem, s>>>(X, Wt, bias, residual, Y, N, H, W, Cin, Cout, n_tiles, total, zero_page, splits, slabs, tickets, gnp, X2, Cin1);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}

}  // namespace

thread_local int g_sk_mode = 1;        // tuning / test hook: 0 = never route to k_conv_sk, 1 = automatic, 2 = every eligible layer
thread_local int g_sk_tile = 0;        // tuning hook: 0 = automatic, 1 = 128x128, 2 = 128x64, 3 = 64x64, 4 = 64x32
thread_local int g_sk_splits = 0;      // tuning hook: >= 1 forces the split factor

// the plan for one layer; bm == 0: not a layer for this kernel
SkPlan conv_sk_plan(int N, int H, int W, int Cin, int Cout, int Cout_pad, int taps, bool two_source, size_t ws_floats) {
    SkPlan p{0, 0, 0, 0};
    if (g_sk_mode == 0 || Cin % 64 != 0 || Cout % 8 != 0 || Cout_pad % 128 != 0) return p;
    const long long M = (long long)N * H * W, hw = (long long)H * W;
    const int KI = taps * (Cin / 64);
    static const int BMs[4] = {128, 128, 64, 64}, BNs[4] = {128, 64, 64, 32};
    static const int SMAX[4] = {2, 4, 8, 16};          // slices whose slabs one workgroup re-reads: <= 128 KB per tile
    int pick = -1, ps = 1;
    for (int t = 0; t < 4; ++t) {
        if (g_sk_tile != 0 && g_sk_tile != t + 1) continue;
        const int bm = BMs[t], bn = BNs[t];
        if (hw % bm != 0) continue;                        // a tile never straddles two images (fused GroupNorm partials)
        const long long tiles = ((M + bm - 1) / bm) * (Cout_pad / bn);
        if (tiles > 640 && g_sk_tile == 0 && g_sk_mode != 2) return p;   // enough tiles already: the big-tile kernels own that regime
        int s = (int)std::min<long long>(std::min<long long>(511 / std::max<long long>(tiles, 1), SMAX[t]), std::max(KI / 2, 1));
        if (ws_floats == 0) s = 1;
        if (s < 1) s = 1;
        while (s > 1 && (size_t)tiles * s * bm * bn + 4096 > ws_floats) --s;
        if (g_sk_splits >= 1) s = (int)std::min<long long>(std::min(g_sk_splits, std::max(KI, 1)), s > 1 || ws_floats ? 64 : 1);
        if (g_sk_splits >= 1) while (s > 1 && (size_t)tiles * s * bm * bn + 4096 > ws_floats) --s;
        pick = t; ps = s;
        if (tiles * s >= 200 || g_sk_tile != 0) break;     // fills the chip: take the largest such tile
    }
    if (pick < 0) return p;
    (void)two_source;
    p.bm = BMs[pick]; p.bn = BNs[pick]; p.splits = ps; p.tile_id = pick + 1;
    return p;
}

int conv_sk(const SkPlan& pl, const half_t* X, const half_t* Wt, const float* bias, const half_t* residual, half_t* Y, int N, int H, int W,
            int Cin, int Cout, int Cout_pad, int taps, const half_t* zero_page, hipStream_t s, float* ws, size_t ws_floats, float* gn_part,
            int* gn_fused, const half_t* X2, int Cin1) {
    PD_REQUIRE(pl.bm > 0 && (taps == 1 || taps == 9), "conv_sk: no plan / bad taps");
    if (X2 == nullptr) Cin1 = Cin;
    PD_REQUIRE(X2 == nullptr || (taps == 1 && Cin1 % 64 == 0 && (Cin - Cin1) % 64 == 0 && Cin1 > 0 && Cin1 < Cin), "conv_sk: bad two-source split");
    const long long M = (long long)N * H * W;
    const int m_tiles = (int)((M + pl.bm - 1) / pl.bm), n_tiles = Cout_pad / pl.bn, total = m_tiles * n_tiles;
    PD_REQUIRE(pl.splits == 1 || (ws != nullptr && (size_t)total * pl.splits * pl.bm * pl.bn + 4096 <= ws_floats && total <= 4096),
               "conv_sk: split-K workspace too small");
    // workspace: [0, 4096) ticket words (zeroed at allocation, self-resetting), then the slabs
    unsigned* tickets = reinterpret_cast<unsigned*>(ws);
    float* slabs = ws ? ws + 4096 : nullptr;
    if (gn_fused) *gn_fused = gn_part ? (int)(((long long)H * W) / pl.bm) : 0;
    const int grid = total * pl.splits;
#define SK_ARGS grid, s, X, Wt, bias, residual, Y, N, H, W, Cin, Cout, n_tiles, total, zero_page, pl.splits, slabs, tickets, gn_part, X2, Cin1
#define SK_TILE(T) (pl.tile_id == 1 ? launch_sk<T, 128, 128, 2>(SK_ARGS) : pl.tile_id == 2 ? launch_sk<T, 128, 64, 3>(SK_ARGS) \
                    : pl.tile_id == 3 ? launch_sk<T, 64, 64, 4>(SK_ARGS) : launch_sk<T, 64, 32, 4>(SK_ARGS))
    return taps == 9 ? SK_TILE(9) : SK_TILE(1);
#undef SK_TILE
#undef SK_ARGS
}

}  // namespace pdnn
