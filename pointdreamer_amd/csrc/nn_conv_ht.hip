// Row U1, the 64^2 / 128^2 levels at UNet batch 1-2 (view-parallel: one view per GPU): 3x3 convolution on 256-pixel x 64-channel
// tiles with an LDS-resident activation halo -- the decomposition whose L2 -> LDS fill a CU can sustain when the layer has only
// 16 384 x 256 (128^2) or 4 096 x 512 (64^2) outputs to spread over 256 CUs (models/DDNM/guided_diffusion/unet.py:143-260, the ResBlock convs).
//
// Why a third conv kernel for these layers.  k_conv3x3_halo's 512 x 128 tile needs >= 256 tiles (batch >= 4 here); k_conv_sk's 128 x 128
// implicit-GEMM tile re-stages the activations once per filter tap: 32 KB of fill per 2.1 MFLOP = 64 FLOP per L2 -> LDS byte, and a CU takes
// ~30 B/clk -- the 128^2 layers ran at 0.5 PFLOP/s (DESIGN_HISTORY Appendix R5); k_conv_rr (weights in registers, K split over the waves) needs the
// whole tile's accumulators in every wave: 2-row bands at W = 128 are 3x halo overhead or 256 accumulator registers.  Here:
//   * tile = 256 pixels as whole image rows (2 rows of 128, 4 of 64) x 64 output channels, K unsplit: 256 tiles at 128^2 / 256 channels and at
//     64^2 / 512 channels / batch 2 -- one round of workgroups, no split-K combine;
//   * chunk-major K loop (k_conv3x3_halo's): per 32 input channels the (rows + 2) x (W + 2) halo (34 / 27 KB) and the nine 64 x 32 weight
//     slices (36 KB) are staged ONCE by LDS-DMA, double-buffered, and all nine taps read the halo at shifted addresses: 70 KB of fill per
//     9.4 MFLOP = 135 FLOP per byte;
//   * 512 threads: waves 4-7 only LOAD (an LDS-DMA issue holds its wave's instruction slot for ~100-180 cycles: seventeen of them in front
//     of a wave's own MFMAs idled the matrix pipe -- first form of this kernel, 27 us at 128^2 / 256 -> 256 against 22 now), waves 0-3 only
//     MULTIPLY: each 64 pixels x all 64 channels (16 accumulator fragments), so a tap costs a wave 4 + 4 ds_read_b128 for 16 MFMAs and no
//     cross-wave reduction exists;
//   * the multipliers' fragment addresses are loop constants (halo rows padded to a multiple of 8 pixels so that the bank swizzle of a pixel
//     does not depend on the filter row: the tap is an immediate offset), the reads of tap t + 1 go out one behind each of the first eight
//     MFMAs of tap t, and an MFMA waits only for the fragments it needs (counted lgkmcnt: LDS reads return in order);
//   * one barrier per chunk, inside tap 8: the wave is done reading the buffer, the next chunk is in, and tap 0 of the next chunk is
//     requested between the MFMAs of tap 8.
// Where the time goes (128^2, 256 -> 256, batch 1: 22 us; lab builds of profiles/r06_ht_bench.txt): 5 us outside the K loop; per chunk 1.25 us
// for the bare MFMA stream (144 x 16 cycles at the ~1.85 GHz the chip holds at its power cap), + 0.25 with the fragment reads, + 0.45 with the fill -- most of it
// present when every piece re-reads one L1-resident KiB, i.e. the DMA's own issue + LDS writes, not L2.  A register-staged loader (buffer
// loads two chunks ahead, ds_write behind the barrier) measured slower (26 us), an L2 prefetch by 4-byte LDS-DMA slower still (32 us).
// Same operand formats as the other conv kernels: NHWC f16 activations, weights [Cout_pad][9 * Cin] (k = tap * Cin + c), bias f32; output f16
// NHWC (+ residual, optionally read at half resolution) + GroupNorm octet partials per 256-pixel tile.
#include "nn_common.h"
#include <algorithm>
#include <type_traits>
using namespace pdhip;
namespace pdnn {

namespace {

typedef __attribute__((address_space(3))) void ht_lds_void;
typedef const __attribute__((address_space(1))) void ht_gbl_void;
__device__ __forceinline__ void ht_glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((ht_gbl_void*)gsrc, (ht_lds_void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ int ht_swz_b(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }   // weight slice: aligned 16-row windows
__device__ __forceinline__ int ht_swz_a(int hp) { return ((hp >> 2) & 1) << 1; }                     // halo: any 16-pixel window
#if defined(PD_LAB_HT_FILL) && (PD_LAB_HT_FILL == 7 || PD_LAB_HT_FILL == 8)     // (lab: no fragment reads)
#define HT_DSR128(dst, addr, off) asm volatile("; no read %0 %1" : "=v"(dst) : "v"(addr))
#else
#define HT_DSR128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#endif
#if defined(PD_LAB_HT_FILL) && (PD_LAB_HT_FILL == 6 || PD_LAB_HT_FILL == 8)     // (lab: no per-chunk barrier)
#define HT_BARRIER()
#else
#define HT_BARRIER() __builtin_amdgcn_s_barrier()
#endif
template <int B, int E, typename Fn>
__device__ __forceinline__ void ht_static_for(Fn&& fn) {
    if constexpr (B < E) { fn(std::integral_constant<int, B>{}); ht_static_for<B + 1, E>(fn); }
}

// WLOG: log2 of the image width (5, 6, 7).  grid: total tiles (1-D, XCD-aware), 512 threads: waves 0-3 multiply, waves 4-7 load.
// (An LDS-DMA issue holds its wave's instruction slot for ~100-180 cycles; 17 of them in front of a wave's own MFMAs left the matrix pipe idle
// for 2-3 000 cycles per chunk -- the first form of this kernel, 7 000 cycles per chunk.  On their own waves they overlap the MFMAs.)
template <int WLOG>
__global__ __launch_bounds__(512) void k_conv_ht(const half_t* __restrict__ X, const half_t* __restrict__ Wt, const float* __restrict__ bias,
                                                 const half_t* __restrict__ residual, half_t* __restrict__ Y, int N, int H, int Cin, int Cout,
                                                 int n_tiles, int total_tiles, const half_t* __restrict__ zero_page, float* __restrict__ gn_part,
                                                 int res_up, int S, float* __restrict__ slabs, unsigned* __restrict__ tickets) {
    constexpr int W = 1 << WLOG, BM = 256, BN = 64;
    constexpr int RT = BM / W, HW2 = (W + 2 + 7) & ~7, HP = (RT + 2) * HW2;   // tile rows, halo row length in LDS (W + 2 pixels used; a multiple of 8
                                                                      //   keeps the swizzle of a pixel the same one row below), halo pixels
    constexpr int NPA = (HP + 15) / 16;                               // 1 KiB halo pieces per chunk (16 pixels x 64 B)
    constexpr int NPW = 9 * 4;                                        // 1 KiB weight pieces per chunk (tap, 16 channels) x 64 B
    constexpr int NP = NPA + NPW, PW = (NP + 3) / 4;                  // pieces per chunk, per loader wave
    constexpr int BUF_BYTES = NP * 1024;
    constexpr int CS_LD = BN + 8;
    static_assert(2 * BUF_BYTES + 64 <= 160 * 1024 && BM * CS_LD * 2 <= 2 * BUF_BYTES, "LDS budget");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kg = lane >> 4;
    int tile, slab;                                                   // S K-slabs per tile (S = 1: K unsplit); the slabs of a tile are neighbours in one XCD
    {
        const int units = total_tiles * S, b = blockIdx.x, q = units >> 3, r = units & 7, xcd = b & 7, i = b >> 3;
        const int unit = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
        tile = unit / S; slab = unit - tile * S;
    }
    const int pt = tile / n_tiles, n0 = (tile - pt * n_tiles) * BN;   // pixel tile, output-channel tile (the n-tiles of a pixel tile share its halo in one L2)
    const int tpi = (H << WLOG) / BM;                                 // pixel tiles per image (= GroupNorm chunks)
    const int img = pt / tpi, tin = pt - img * tpi, ty0 = tin * RT;
    const int K = 9 * Cin, NCT = (Cin >> 5) / S, cb = slab * NCT;     // chunks of this slab, first chunk
    float4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    // Barrier B_c (c = 0 .. NCT): chunk c is in LDS and chunk c - 1 is consumed (the multipliers' reads returned), so between B_c and B_c+1
    // the loaders fill buffer (c + 1) & 1 while the multipliers read buffer c & 1.  Raw barriers: both branches execute NCT + 1 of them.
    if (wave >= 4) {
        // ---- loader wave lw: piece slot k = piece min(4 k + lw, NP - 1) (tail slots re-load the last piece).  Lane (lq = lane >> 2,
        // cq = lane & 3) supplies the 16 bytes landing in physical slot cq of row / pixel lq of the piece.  Byte offset of the source at
        // chunk 0 (~0u: zeros -- outside the image / past the halo); a chunk further = + 64 bytes for both operands.
        const int lw = wave - 4;
        uint32_t poff[PW];
        const int lq = lane >> 2, cq = lane & 3;
#pragma unroll
        for (int k = 0; k < PW; ++k) {
            const int q = min(k * 4 + lw, NP - 1);
            if (q < NPA) {
                const int hp = q * 16 + lq;
                const int hy = hp / HW2, hx = hp - hy * HW2;
                const int y = ty0 - 1 + hy, x = hx - 1;
                const bool ok = (hp < HP) & (y >= 0) & (y < H) & (x >= 0) & (x < W);
                const int spix = ((img * H + y) << WLOG) + x;
                poff[k] = ok ? (uint32_t)((spix * Cin + ((cq ^ ht_swz_a(hp)) << 3)) * 2) : ~0u;
            } else {
                const int wq = q - NPA, t = wq >> 2, row = (wq & 3) * 16 + lq;         // tap, output channel inside the tile
                poff[k] = (uint32_t)(((size_t)(n0 + row) * K + (size_t)t * Cin + ((cq ^ ht_swz_b(row)) << 3)) * 2);
            }
        }
        const char* const Xb = reinterpret_cast<const char*>(X);
        const char* const Wb = reinterpret_cast<const char*>(Wt);
        auto stage = [&](int buf, int chunk) {
            char* const dst = smem + buf * BUF_BYTES;
#pragma unroll
            for (int k = 0; k < PW; ++k) {
                const int q = min(k * 4 + lw, NP - 1);                // (wave-uniform)
                const void* src;
                if (q < NPA) src = poff[k] == ~0u ? (const void*)(zero_page + ((cq * 8) & 63)) : (const void*)(Xb + poff[k] + (uint32_t)(cb + chunk) * 64u);
                else src = (const void*)(Wb + poff[k] + (uint32_t)(cb + chunk) * 64u);
#if defined(PD_LAB_HT_FILL) && PD_LAB_HT_FILL == 2          // (lab builds, WRONG results: every piece reads the same 1 KiB -- DMA issue + LDS write alone)
                src = (const void*)(Xb + lane * 16);
#elif defined(PD_LAB_HT_FILL) && PD_LAB_HT_FILL == 3        // (every piece reads its own CONTIGUOUS 1 KiB: whole 128-byte lines)
                src = (const void*)(Xb + (size_t)((chunk * NP + q) * 1024 + lane * 16));
#endif
#if !defined(PD_LAB_HT_FILL) || (PD_LAB_HT_FILL != 1 && PD_LAB_HT_FILL < 6)   // (1, 6, 7, 8: no fill at all -- the multipliers alone)
                ht_glds16(src, dst + q * 1024);
#endif
            }
        };
        stage(0, 0);
        for (int c = 0; c <= NCT; ++c) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (c == 0) __builtin_amdgcn_s_barrier(); else HT_BARRIER();       // B_c
            if (c + 1 < NCT) stage((c + 1) & 1, c + 1);
        }
    } else {
        // ---- multiplier wave w owns pixels 64 w .. 64 w + 63 of the tile (all 64 channels).  Fragment addresses in buffer 0:
        //   activations (B operand): pixel p = 64 w + 16 i + r16, filter column tx -> halo pixel hp = (p / W) * HW2 + p % W + tx, 16-byte slot
        //     kg ^ swz_a(hp); the filter row ty is the immediate offset ty * HW2 * 64 (HW2 % 8 == 0: same swizzle)
        //   weights (A operand): row 16 j + r16, slot kg ^ swz_b(row); the tap is the immediate offset tap * 4096
        // -> no address arithmetic in the loop beyond the buffer parity.
        uint32_t a0[12], w0[4];
#pragma unroll
        for (int tx = 0; tx < 3; ++tx)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int p = wave * 64 + i * 16 + r16, hp = (p >> WLOG) * HW2 + (p & (W - 1)) + tx;
                a0[tx * 4 + i] = (uint32_t)((hp << 6) + ((kg ^ ht_swz_a(hp)) << 4));
            }
        const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
#pragma unroll
        for (int j = 0; j < 4; ++j) w0[j] = lds0 + (uint32_t)(NPA * 1024 + (j * 16 + r16) * 64 + ((kg ^ ht_swz_b(j * 16 + r16)) << 4));
#pragma unroll
        for (int k = 0; k < 12; ++k) a0[k] += lds0;
        half8 fa[2][4], fw[2][4];                                     // [tap parity][fragment]
        // One tap = 16 MFMAs (256 cycles) with the 8 fragment reads of the NEXT tap between them, one per two MFMAs; a tap starts by waiting
        // for its own reads (requested a tap earlier).  Tap 8 waits, passes barrier B_c+1 (this wave is done with the buffer; the next chunk is
        // in) and requests tap 0 of the next chunk between its MFMAs: the multipliers never wait for an LDS round trip.
        // (Macros, not lambdas: hipcc refuses the implicit captures of an asm operand in a generic lambda.)
#define HT_RW(T, S, j, BB) HT_DSR128(fw[S][j], w0[j] + (BB), ((T) % 9) * 4096);
#define HT_RA(T, S, i, BB) HT_DSR128(fa[S][i], a0[(((T) % 9) % 3) * 4 + i] + (BB), (((T) % 9) / 3) * HW2 * 64);
#define HT_WAIT0(S)                                                                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fw[S][0]), "+v"(fw[S][1]), "+v"(fw[S][2]), "+v"(fw[S][3]), "+v"(fa[S][0]), "+v"(fa[S][1]), "+v"(fa[S][2]), "+v"(fa[S][3]));
#define HT_WAITN(CNT, X) asm volatile("s_waitcnt lgkmcnt(" #CNT ")" : "+v"(X));
#define HT_WAITN2(CNT, X, Y) asm volatile("s_waitcnt lgkmcnt(" #CNT ")" : "+v"(X), "+v"(Y));
#define HT_MM(S, i, j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[S][j], fa[S][i], acc[i][j], 0, 0, 0);
#define HT_SB __builtin_amdgcn_sched_barrier(0);
        // Tap T multiplies out of register set S while the reads of tap T + 1 fill set S ^ 1: requested in the order A0 W0 W1 W2 W3 A1 A2 A3,
        // one behind each of the first eight MFMAs, so the youngest is 128 MFMA cycles old when the tap ends.  LDS reads return in order: MFMA
        // (i, j) waits only for the fragments it needs -- the counts below = reads of this tap that may still be out + reads of the next tap
        // requested so far.  (A wait for all eight at the top of a tap exposed an LDS round trip per tap: 3 700 cycles per chunk against 2 300.)
#define HT_TAP_BODY(T, S, BB)                                                                                          \
        HT_WAITN2(6, fa[S][0], fw[S][0]) HT_SB HT_MM(S, 0, 0) HT_RA((T) + 1, (S) ^ 1, 0, BB) HT_SB                     \
        HT_WAITN(6, fw[S][1]) HT_SB HT_MM(S, 0, 1) HT_RW((T) + 1, (S) ^ 1, 0, BB) HT_SB                               \
        HT_WAITN(6, fw[S][2]) HT_SB HT_MM(S, 0, 2) HT_RW((T) + 1, (S) ^ 1, 1, BB) HT_SB                               \
        HT_WAITN(6, fw[S][3]) HT_SB HT_MM(S, 0, 3) HT_RW((T) + 1, (S) ^ 1, 2, BB) HT_SB                               \
        HT_WAITN(6, fa[S][1]) HT_SB HT_MM(S, 1, 0) HT_RW((T) + 1, (S) ^ 1, 3, BB) HT_SB                               \
        HT_MM(S, 1, 1) HT_RA((T) + 1, (S) ^ 1, 1, BB) HT_SB HT_MM(S, 1, 2) HT_RA((T) + 1, (S) ^ 1, 2, BB) HT_SB       \
        HT_MM(S, 1, 3) HT_RA((T) + 1, (S) ^ 1, 3, BB) HT_SB                                                            \
        HT_WAITN(9, fa[S][2]) HT_SB HT_MM(S, 2, 0) HT_MM(S, 2, 1) HT_MM(S, 2, 2) HT_MM(S, 2, 3) HT_SB                 \
        HT_WAITN(8, fa[S][3]) HT_SB HT_MM(S, 3, 0) HT_MM(S, 3, 1) HT_MM(S, 3, 2) HT_MM(S, 3, 3) HT_SB
#define HT_TAP(T, P, CUR) { HT_TAP_BODY(T, ((T) + (P)) & 1, CUR) }
        // chunk of parity P (nine taps: the register sets swap roles from one chunk to the next, so the loop is unrolled by two).  Tap 8 first
        // waits for ALL its reads (this wave is done with the buffer), passes barrier B_c+1 (the next chunk is in) and requests tap 0 of the
        // next chunk between its MFMAs.
#define HT_CHUNK(P, CUR, NXT, MORE)                                                                                    \
        HT_TAP(0, P, CUR) HT_TAP(1, P, CUR) HT_TAP(2, P, CUR) HT_TAP(3, P, CUR) HT_TAP(4, P, CUR) HT_TAP(5, P, CUR) HT_TAP(6, P, CUR) HT_TAP(7, P, CUR) \
        HT_WAIT0(P) HT_SB                                                                                              \
        HT_BARRIER();                                                 /* B_c+1 */                                      \
        asm volatile("" ::: "memory");                                                                                 \
        HT_SB                                                                                                          \
        if (MORE) { HT_TAP_BODY(8, P, NXT) }                          /* ((8 + 1) % 9 = tap 0, in the other buffer) */ \
        else {                                                                                                         \
            HT_MM(P, 0, 0) HT_MM(P, 0, 1) HT_MM(P, 0, 2) HT_MM(P, 0, 3) HT_MM(P, 1, 0) HT_MM(P, 1, 1) HT_MM(P, 1, 2) HT_MM(P, 1, 3)   \
            HT_MM(P, 2, 0) HT_MM(P, 2, 1) HT_MM(P, 2, 2) HT_MM(P, 2, 3) HT_MM(P, 3, 0) HT_MM(P, 3, 1) HT_MM(P, 3, 2) HT_MM(P, 3, 3)   \
        }
        __builtin_amdgcn_s_barrier();                                 // B_0
        asm volatile("" ::: "memory");
        HT_RA(0, 0, 0, 0u) HT_RW(0, 0, 0, 0u) HT_RW(0, 0, 1, 0u) HT_RW(0, 0, 2, 0u) HT_RW(0, 0, 3, 0u) HT_RA(0, 0, 1, 0u) HT_RA(0, 0, 2, 0u) HT_RA(0, 0, 3, 0u)
        for (int c = 0; c < NCT; c += 2) {
            HT_CHUNK(0, 0u, (uint32_t)BUF_BYTES, c + 1 < NCT)
            if (c + 1 < NCT) { HT_CHUNK(1, (uint32_t)BUF_BYTES, 0u, c + 2 < NCT) }
        }
#undef HT_CHUNK
#undef HT_TAP
#undef HT_TAP_BODY
#undef HT_SB
#undef HT_MM
#undef HT_WAIT0
#undef HT_WAITN
#undef HT_WAITN2
#undef HT_RA
#undef HT_RW
    }
    __syncthreads();                                                  // every fragment read is done: the buffers become the output staging

    // ---- in-launch split-K combine (k_conv_sk's / k_conv_rr's protocol): every slab publishes its f32 accumulators write-through, drains, takes
    // a ticket; the last arriver sums ALL slices in slab order -- its own included, so the order never depends on who is last -- and goes on
    if (S > 1) {
        constexpr int SLICE_BYTES = 16 * 4096;                        // 16 fragments x 256 lanes x 16 bytes
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(slabs + (size_t)tile * S * (SLICE_BYTES / 4), 0, S * SLICE_BYTES, 0x00020000);
        typedef uint32_t ht_u4 __attribute__((ext_vector_type(4)));
        if (wave < 4) {
#pragma unroll
            for (int f = 0; f < 16; ++f)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ht_u4, acc[f >> 2][f & 3]), rs, slab * SLICE_BYTES + (f * 256 + tid) * 16, 0, /*sc1: write-through*/ 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        volatile int* flag = reinterpret_cast<volatile int*>(smem + 2 * BUF_BYTES);
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = old == (unsigned)(S - 1);
            if (last) __hip_atomic_store(tickets + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *flag = last;
        }
        __syncthreads();
        if (*flag == 0) return;
        if (wave < 4) {
#pragma unroll
            for (int f = 0; f < 16; ++f) acc[f >> 2][f & 3] = (float4_t){0.f, 0.f, 0.f, 0.f};
            for (int sp = 0; sp < S; ++sp) {
                float4_t v[16];
#pragma unroll
                for (int f = 0; f < 16; ++f)
                    v[f] = __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, sp * SLICE_BYTES + (f * 256 + tid) * 16, 0, /*sc1*/ 16));
#pragma unroll
                for (int f = 0; f < 16; ++f) { acc[f >> 2][f & 3][0] += v[f][0]; acc[f >> 2][f & 3][1] += v[f][1]; acc[f >> 2][f & 3][2] += v[f][2]; acc[f >> 2][f & 3][3] += v[f][3]; }
            }
        }
    }

    // ---- epilogue: acc + bias -> f16 -> LDS [pixel][CS_LD] -> 16-byte rows (+ residual) + GroupNorm octet partials of the tile.
    // Lane holds pixel 64 w + 16 i + r16, channels 16 j + 4 kg + 0..3
    half_t* Cs = reinterpret_cast<half_t*>(smem);
    if (wave < 4)                                          // (the loader waves hold no accumulators)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int nl = j * 16 + kg * 4;
        float4_t bv = (float4_t){0.f, 0.f, 0.f, 0.f};
        if (bias != nullptr && n0 + nl < Cout) bv = *reinterpret_cast<const float4_t*>(bias + n0 + nl);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ml = wave * 64 + i * 16 + r16;
            half4 h;
#pragma unroll
            for (int r = 0; r < 4; ++r) h[r] = (half_t)(acc[i][j][r] + bv[r]);
            *reinterpret_cast<half4*>(&Cs[ml * CS_LD + nl]) = h;
        }
    }
    __syncthreads();
    constexpr int CT = BN / 8, RPP = 512 / CT, PASSES = BM / RPP;     // 8 octet threads per row, 64 rows per pass, 4 passes
    const int col8 = (tid % CT) * 8;
    float gs = 0.f, gq = 0.f;
    const bool col_ok = n0 + col8 < Cout;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int row = p * RPP + tid / CT;
        if (col_ok) {
            half8 v = *reinterpret_cast<const half8*>(&Cs[row * CS_LD + col8]);
            const int yy = ty0 + (row >> WLOG), xx = row & (W - 1);
            const size_t o = ((size_t)((img * H + yy) << WLOG) + xx) * Cout + n0 + col8;
            if (residual != nullptr) {
                size_t ro = o;
                if (res_up) ro = ((size_t)(((img * (H >> 1) + (yy >> 1)) << (WLOG - 1)) + (xx >> 1))) * Cout + n0 + col8;
                const half8 rv = *reinterpret_cast<const half8*>(residual + ro);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] + (float)rv[e]);
            }
            *reinterpret_cast<half8*>(Y + o) = v;
            float s8 = 0.f, q8 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; s8 += f; q8 += f * f; }
            gs += s8; gq += q8;
        }
    }
    if (gn_part != nullptr) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);       // [thread][2]
        red[tid * 2] = gs; red[tid * 2 + 1] = gq;
        __syncthreads();
        // two-level, fixed-order column sums: 4 threads per octet take every 4th of the 64 row slots each, one thread adds the 4 sub-sums in order
        constexpr int SUB = 4;
        float* red2 = red + 2 * 512;
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
            float* dst = gn_part + (((size_t)img * tpi + tin) * (Cout >> 3) + (n0 >> 3) + tid) * 2;
            dst[0] = s1; dst[1] = q1;
        }
    }
}

template <int WLOG>
int launch_ht(const half_t* X, const half_t* Wt, const float* bias, const half_t* residual, half_t* Y, int N, int H, int Cin, int Cout, int Cout_pad,
              const half_t* zero_page, float* gn_part, int res_up, int S, float* ws, hipStream_t s) {
    constexpr int W = 1 << WLOG, RT = 256 / W, HP = (RT + 2) * ((W + 2 + 7) & ~7), NP = (HP + 15) / 16 + 36;
    constexpr int smem = 2 * NP * 1024 + 64;
    auto kern = k_conv_ht<WLOG>;
    PD_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    const int n_tiles = Cout_pad / 64, total = (int)(((long long)N * H * W / 256) * n_tiles);
    kern<<<total * S, 512, smem, s>>>(X, Wt, bias, residual, Y, N, H, Cin, Cout, n_tiles, total, zero_page, gn_part, res_up, S,
                                      ws ? ws + PD_SK_TICKET_FLOATS : nullptr, reinterpret_cast<unsigned*>(ws));
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}

}  // namespace

thread_local int g_ht_mode = 1;        // tuning / test hook (pdhip_debug_set_conv_ht): 0 = never, 1 = automatic, 2 = every eligible layer
thread_local int g_ht_slabs = 0;       //   K-slabs forced (0 = automatic)

// K-slabs of a layer: K is cut in two (four) when the tiles alone leave half (three quarters) of the CUs idle and every slab keeps >= 8 chunks
// (the combine -- 64 KB published write-through per workgroup, ticket, two slices re-read -- costs ~8 us: 64^2 / batch 1 at 256 input channels
// 19.6 us in two slabs against 17.1 unsplit, at 512 / 768 / 1 024: 26.7 / 34.3 / 41.7 against 28.7 / 44.8 / 54.7); needs the split-K workspace
// (tickets + tiles x S x 64 KB of f32 slices)
int conv_ht_slabs(int N, int H, int W, int Cin, int Cout_pad, size_t ws_floats) {
    const long long tiles = ((long long)N * H * W / 256) * (Cout_pad / 64);
    const int nct = Cin >> 5;
    auto fits = [&](int S) { return nct % S == 0 && tiles <= 4096 && (size_t)tiles * S * 16384 + PD_SK_TICKET_FLOATS <= ws_floats; };
    if (g_ht_slabs > 0) return fits(g_ht_slabs) ? g_ht_slabs : 1;
    int S = 1;
    while (S < 4 && tiles * S * 2 <= 256 && nct / (S * 2) >= 8 && fits(S * 2)) S *= 2;
    return S;
}

// does this layer run in k_conv_ht?  Eligible: 3x3, H == W in {32, 64, 128}, whole 256-pixel tiles, 32-channel chunks, single-source input.
// Automatic (profiles/r06_ht_bench.txt, against the route without it on the same box), at most two rounds of workgroups: the 128^2 level at
// batch 1 (22 / 37 / 52 us against 27 / 45 / 63 at Cin 256 / 512 / 768) and at batch 2 from 512 input channels (76 / 105 against 79 / 113; 256:
// 48 against 45); the 64^2 level once its tiles fill the chip (batch 2-4: 36 / 66 us against 45 / 84 at batch 2) and at batch 1 (half the
// chip) up to 768 input channels (17 / 29 / 45 against 20 / 32 / 46; 1 024: 56 against 52); never at 32^2 (k_conv_rr's).  Beyond two rounds
// the 512 x 128 halo tile owns the layer.
bool conv_ht_routes(int N, int H, int W, int Cin, int Cout, int Cout_pad, size_t ws_floats) {
    if (g_ht_mode == 0 || H != W || (W != 32 && W != 64 && W != 128) || ((long long)H * W) % 256 != 0 || Cin % 32 != 0 || Cout % 8 != 0 ||
        Cout_pad % 64 != 0 || (long long)N * H * W * Cin * 2 > 0x7ffffff0LL || (long long)Cout_pad * 9 * Cin * 2 > 0x7ffffff0LL) return false;
    if (g_ht_mode == 2) return true;
    const long long tiles = ((long long)N * H * W / 256) * (Cout_pad / 64);
    if (tiles > 512) return false;
    return W == 128 ? (N == 1 || Cin >= 512) : (W == 64 && (tiles >= 256 || Cin <= 768 || conv_ht_slabs(N, H, W, Cin, Cout_pad, ws_floats) > 1));
}

int conv_ht(const half_t* X, const half_t* Wt, const float* bias, const half_t* residual, half_t* Y, int N, int H, int W, int Cin, int Cout,
            int Cout_pad, const half_t* zero_page, hipStream_t s, float* gn_part, int* gn_chunks, int res_up, float* ws, size_t ws_floats) {
    PD_REQUIRE(X && Wt && Y && zero_page, "conv_ht: null argument");
    PD_REQUIRE(H == W && (W == 32 || W == 64 || W == 128) && Cin % 32 == 0 && Cout % 8 == 0 && Cout_pad % 64 == 0 && Cout_pad >= Cout,
               "conv_ht: unsupported geometry (H=%d W=%d Cin=%d Cout=%d)", H, W, Cin, Cout);
    PD_REQUIRE(res_up == 0 || (residual != nullptr && H % 2 == 0), "conv_ht: an up-sampled residual needs even H, W");
    PD_REQUIRE((long long)N * H * W * Cin * 2 <= 0x7ffffff0LL && (long long)Cout_pad * 9 * Cin * 2 <= 0x7ffffff0LL, "conv_ht: operand beyond the 2 GB buffer range");
    if (gn_chunks) *gn_chunks = gn_part ? H * W / 256 : 0;
    const int S = conv_ht_slabs(N, H, W, Cin, Cout_pad, ws ? ws_floats : 0);
    if (W == 128) return launch_ht<7>(X, Wt, bias, residual, Y, N, H, Cin, Cout, Cout_pad, zero_page, gn_part, res_up, S, ws, s);
    if (W == 64) return launch_ht<6>(X, Wt, bias, residual, Y, N, H, Cin, Cout, Cout_pad, zero_page, gn_part, res_up, S, ws, s);
    return launch_ht<5>(X, Wt, bias, residual, Y, N, H, Cin, Cout, Cout_pad, zero_page, gn_part, res_up, S, ws, s);
}

}  // namespace pdnn

extern "C" int pdhip_conv_ht_f16(const void* x, const void* w_packed, const float* bias, const void* residual, int res_up, void* y, int N, int H, int W, int Cin,
                                 int Cout, int Cout_pad, const void* zero_page, float* splitk_ws, long long splitk_ws_floats, float* gn_part, int* gn_chunks,
                                 void* stream) {
    return pdnn::conv_ht((const pdnn::half_t*)x, (const pdnn::half_t*)w_packed, bias, (const pdnn::half_t*)residual, (pdnn::half_t*)y, N, H, W, Cin, Cout, Cout_pad,
                         (const pdnn::half_t*)zero_page, as_stream(stream), gn_part, gn_chunks, res_up, splitk_ws, splitk_ws ? (size_t)splitk_ws_floats : 0);
}
// host-only: the routing decision for a layer (no launch, no GPU needed)
extern "C" int pdhip_conv_ht_plan(int N, int H, int W, int Cin, int Cout, int Cout_pad, long long splitk_ws_floats, int* routed, int* slabs) {
    const size_t wf = splitk_ws_floats > 0 ? (size_t)splitk_ws_floats : 0;
    if (routed) *routed = pdnn::conv_ht_routes(N, H, W, Cin, Cout, Cout_pad, wf) ? 1 : 0;
    if (slabs) *slabs = pdnn::conv_ht_slabs(N, H, W, Cin, Cout_pad, wf);
    return PDHIP_OK;
}
extern "C" int pdhip_debug_set_conv_ht(int mode, int slabs) {
    const int old = pdnn::g_ht_mode;
    pdnn::g_ht_mode = mode; pdnn::g_ht_slabs = slabs;
    return old;
}
