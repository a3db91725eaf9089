// Row P3b: hidden-point removal (Katz, Tal, Basri 2007) on the device -- replaces the Open3D call at
// pointdreamer/ours_utils.py:204-225 (`pcd.hidden_point_removal(eye, radius)`: spherical flip + qhull on the CPU).
//
//   flip:   q = p - eye,  p' = q + 2 (radius - |q|) q / |q|                       (float64, like open3d)
//   visible(i)  <=>  p'_i is a vertex of conv({p'_j} U {0})  <=>  0 is NOT in conv(S_i),
//                    S_i = {p'_j - p'_i : j != i} U {-p'_i}
// Instead of building the hull (qhull: serial, incremental), every point answers its own containment question with a
// boolean GJK iteration whose only heavy step is the support function  argmax_j  d . p'_j  -- an O(N) streaming
// reduction.  One wavefront owns 16 query points: all 64 lanes stream the flipped cloud once per round (coalesced f64
// SoA, L2-resident: 24 N bytes per view) and evaluate the 16 search directions against every point (48 f64 FMAs per
// 24 bytes), the 16 GJK states live in lanes 0-15.  Work: ~10 rounds x N^2 x 3 FMA per view (f64 vector rate bound).
// Two levels: ALL points are first tested against a COARSE support set -- the KC extreme points of the flipped cloud in KC
// Fibonacci-sphere directions (one streaming pass; its members are certain hull vertices and are never queried).  conv(subset)
// is inside conv(cloud), so "origin enclosed" there is already the final answer (hidden).  A point strictly inside conv(subset)
// is also never a support point of the full cloud, so the second level -- only for the queries the coarse hull cannot enclose --
// scans just the OUTSIDE set (~40 % of the cloud), compacted in cloud order so that the scan needs no index tie-break.  Short
// query lists (the pipeline's case: only depth-rejected points are queried) run 4 queries per 256-lane block with the scan split
// over the four waves.  30 k points x 8 views: 88 -> 15 ms for all points, 48 -> 2.4 ms behind the depth-test skip mask
// (KC = 1024 measured best of 512..8192).
// qhull's facet-merging tolerances are not reproduced (PARITY UNPINNED, open3d absent): points within ~1e-9 of a hull
// facet may be classified differently; tests bound the disagreement with scipy's qhull.
#include "common.h"
using namespace pdhip;

#define QPW 16                 // query points per wavefront
#define GJK_MAX_ROUNDS 64
#define GJK_COARSE_ROUNDS 32
#define HPR_KC 1024            // coarse support set size
#define HPR_COOP_WAVES 4       // waves that share one group of 4 queries in the short-list geometry (8: -6 %, 16: -13 %)
#define HPR_NARROW_BELOW 4096   // query lists shorter than this (per view) run 4 queries per wavefront instead of 16

#ifdef PD_HPR_STATS                                       // (lab builds only: round statistics of the two GJK passes)
__device__ unsigned long long g_hpr_stats[2][16];             // [pass][waves, wave rounds, queries, query rounds, unfinished, -, -, -, histogram of query rounds / 8]
extern "C" int pdhip_lab_hpr_stats(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hpr_stats), sizeof(g_hpr_stats)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_hpr_stats), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif
struct d3 { double x, y, z; };
__device__ __forceinline__ d3 operator-(d3 a, d3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ d3 neg(d3 a) { return {-a.x, -a.y, -a.z}; }
__device__ __forceinline__ double dot(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ d3 cross(d3 a, d3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

__global__ void k_hpr_flip(const float* __restrict__ pts, int N, const double* __restrict__ eyes, double radius,
                           double* __restrict__ flipped /*[V][3][N]*/) {
    const int v = blockIdx.y;
    const double ex = eyes[3 * v], ey = eyes[3 * v + 1], ez = eyes[3 * v + 2];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        double qx = (double)pts[3 * i] - ex, qy = (double)pts[3 * i + 1] - ey, qz = (double)pts[3 * i + 2] - ez;
        double n = sqrt(qx * qx + qy * qy + qz * qz);
        if (n < 1e-300) n = 1e-300;
        const double s = 1.0 + 2.0 * (radius - n) / n;
        double* f = flipped + (size_t)v * 3 * N;
        f[i] = qx * s; f[N + i] = qy * s; f[2 * (size_t)N + i] = qz * s;
    }
}

// queries that still need the hull test: all points, or only those a cheaper test (`skip`) has not already accepted
__global__ void k_hpr_collect(const uint8_t* __restrict__ skip, int N, int* __restrict__ count, int* __restrict__ list,
                              uint8_t* __restrict__ vis) {
    // one returning atomic per 256-thread block (a returning atomic on one address costs ~100 ns; per wave it serialised to 45 us)
    __shared__ int s_wcnt[4], s_base;
    const int v = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i0 = blockIdx.x * blockDim.x; i0 < N; i0 += gridDim.x * blockDim.x) {
        const int i = i0 + threadIdx.x;
        const bool in = i < N;
        const bool sk = in && skip != nullptr && skip[(size_t)v * N + i];
        const bool q = in && !sk;
        const unsigned long long bal = __ballot(q);
        if (lane == 0) s_wcnt[wave] = __popcll(bal);
        __syncthreads();
        if (threadIdx.x == 0) s_base = atomicAdd(&count[v], s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3]);
        __syncthreads();
        int base = s_base;
        for (int w = 0; w < wave; ++w) base += s_wcnt[w];
        if (q) list[(size_t)v * N + base + __popcll(bal & ((1ull << lane) - 1ull))] = i;
        else if (sk) vis[(size_t)v * N + i] = 1;
        __syncthreads();
    }
}

// Support set = ss[v][3][scap] (first `ns` entries valid) with original cloud indices sidx[v][scap].
// Q = queries per wavefront (16 when there are enough queries to fill the chip; short lists -- where the kernel time is the
// longest query's rounds x the scan length -- use Q = 4 with COOP: 4 queries per 256-lane block, 16x shorter scans).
// COARSE: every point of the cloud is a query (ns = scap = KC extreme points); writes outside[v][q] = origin not enclosed.
// !COARSE: queries from list / count, support set = the points outside the coarse hull (ns = scount[v]); writes vis.
template <bool COARSE, int Q, bool COOP, int NW>
__global__ __launch_bounds__(NW * 64, NW <= 4 ? 2 : 1) void k_hpr_gjk(const double* __restrict__ flipped, int N, const int* __restrict__ count,
                                                 const int* __restrict__ list, uint8_t* __restrict__ vis,
                                                 const double* __restrict__ ss, const int* __restrict__ sidx_all, int scap,
                                                 const int* __restrict__ scount, uint8_t* __restrict__ outside, int q_lo, int q_hi,
                                                    const uint8_t* __restrict__ skip) {
    __shared__ double s_dir[NW][Q][3];
    __shared__ int s_q[NW][Q];
    __shared__ double s_rv[2][NW][Q];                            // COOP: per-wave partial argmax of the round (double-buffered)
    __shared__ int s_ri[2][NW][Q];
    const int v = blockIdx.y;
    const double* qfx = flipped + (size_t)v * 3 * N;          // the queries' own coordinates
    const double* qfy = qfx + N;
    const double* qfz = qfy + N;
    const double* fx = ss + (size_t)v * 3 * scap;             // the support set
    const double* fy = fx + scap;
    const double* fz = fy + scap;
    const int* sidx = sidx_all + (size_t)v * scap;
    const int NS = COARSE ? scap : scount[v];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // COOP: the four waves of the block own the SAME Q queries, scan a quarter of the support set each and merge through LDS
    // (all four run the identical state machine on the merged result): a round is 4x shorter, which is what bounds the
    // kernel when the query list is short and a few queries need 40-60 rounds.
    const int q0 = COOP ? blockIdx.x * Q : (blockIdx.x * NW + wave) * Q;
    const int nq = COARSE ? N : count[v];
    if (q0 >= nq || nq < q_lo || nq >= q_hi) return;        // (q_lo, q_hi: which launch geometry serves this view's query count)
    // query k of the wave lives in lane k * ST (the lane the halving reduction below leaves its support point in)
    constexpr int ST = 64 / Q;
    const int kq = lane / ST;
    const bool slot = (lane % ST) == 0;
    bool owner = slot && q0 + kq < nq;
    const int q = owner ? (COARSE ? q0 + kq : list[(size_t)v * N + q0 + kq]) : -1;
    // coarse pass: a point of the coarse set itself is an extreme point of the cloud -- a certain hull vertex, marked 2 by
    // k_hpr_extremes, never queried (so no query of this pass is a member of its own support set)
    if (COARSE && owner && outside[(size_t)v * N + q] == 2) owner = false;
    // ... and a point the cheaper test already accepted needs no verdict of its own: it joins the second-level support set
    // unexamined (any superset of the outside set inside the cloud is a valid support set; depth-visible points are almost all
    // outside anyway), which removes a third of the coarse queries
    if (COARSE && owner && skip != nullptr && skip[(size_t)v * N + q]) { outside[(size_t)v * N + q] = 1; owner = false; }
    if (slot) s_q[wave][kq] = q;
    __builtin_amdgcn_wave_barrier();
    int qk[Q];
#pragma unroll
    for (int k = 0; k < Q; ++k) qk[k] = s_q[wave][k];
    // ---- per-query GJK state (meaningful in the slot lanes)
    d3 pi = {0, 0, 0}, sa = {0, 0, 0}, sb = {0, 0, 0}, sc = {0, 0, 0}, sd = {0, 0, 0}, dir = {0, 0, 1};
    int dim = 0;                   // simplex size; phases: 0 -> fetch c, 1 -> fetch b, >= 2 -> main loop
    int state = owner ? 0 : 2;     // 0 running, 1 visible (origin outside), 2 hidden / not a query
    if (owner) {
        pi = {qfx[q], qfy[q], qfz[q]};
        dir = pi;                  // start looking straight out along the point's own ray
    }
#ifdef PD_HPR_STATS
    int my_rounds = 0, wave_rounds = 0;
#endif
    for (int round = 0; round < (COARSE ? GJK_COARSE_ROUNDS : GJK_MAX_ROUNDS); ++round) {
        if (__ballot(state == 0) == 0ull) break;
#ifdef PD_HPR_STATS
        ++wave_rounds; if (state == 0) ++my_rounds;
#endif
        if (slot) { s_dir[wave][kq][0] = dir.x; s_dir[wave][kq][1] = dir.y; s_dir[wave][kq][2] = dir.z; }
        __builtin_amdgcn_wave_barrier();
        double dx[Q], dy[Q], dz[Q], best[Q];
        int bi[Q];                                                 // position in the support set
#pragma unroll
        for (int k = 0; k < Q; ++k) {
            dx[k] = s_dir[wave][k][0]; dy[k] = s_dir[wave][k][1]; dz[k] = s_dir[wave][k][2];
            best[k] = -1.0e300; bi[k] = 0x7fffffff;
        }
        // ---- support scan: every lane streams points j = lane, lane+64, ...  The support set is in ascending cloud order, a
        // lane keeps its FIRST maximum and the butterfly prefers the smaller position: equal values resolve to the smallest
        // cloud index without an index compare per pair (8 VALU per pair instead of 15).
        // (the next point is requested before the current one is evaluated: two waves per SIMD do not hide an L2 round trip)
        double nx = 0.0, ny = 0.0, nz = 0.0;
        int njo = -1;
        constexpr int JS = COOP ? NW * 64 : 64;
        const int j0 = COOP ? wave * 64 + lane : lane;
        if (j0 < NS) { nx = fx[j0]; ny = fy[j0]; nz = fz[j0]; if (!COARSE) njo = sidx[j0]; }
        for (int j = j0; j < NS; j += JS) {
            const double x = nx, y = ny, z = nz;
            const int jo = njo;
            const int jn = j + JS;
            if (jn < NS) { nx = fx[jn]; ny = fy[jn]; nz = fz[jn]; if (!COARSE) njo = sidx[jn]; }
            if (COARSE) {
#pragma unroll
                for (int k = 0; k < Q; ++k) {
                    const double val = dx[k] * x + dy[k] * y + dz[k] * z;
                    if (val > best[k]) { best[k] = val; bi[k] = j; }
                }
            } else {                                                 // jo = index in the cloud: S_i excludes the point itself
#pragma unroll
                for (int k = 0; k < Q; ++k) {
                    const double val = dx[k] * x + dy[k] * y + dz[k] * z;
                    if ((val > best[k]) & (jo != qk[k])) { best[k] = val; bi[k] = j; }
                }
            }
        }
        // ---- Q argmax reductions over the 64 lanes at once: every step swaps one half of the still-live queries with the
        // partner lane and keeps the other half (Q-1 exchanged items instead of 6 Q), then plain butterflies inside the ST
        // lanes that end up holding the same query.  Equal values: smaller position.
#define HPR_HALVE(I)                                                                                                  \
        if constexpr ((Q >> (I)) > 1) {                                                                               \
            constexpr int off = 32 >> (I), n = Q >> (I);                                                              \
            const bool hi = (lane & off) != 0;                                                                        \
            _Pragma("unroll") for (int k = 0; k < n / 2; ++k) {                                                       \
                const double send = hi ? best[k] : best[k + n / 2], keep = hi ? best[k + n / 2] : best[k];            \
                const int sendi = hi ? bi[k] : bi[k + n / 2], keepi = hi ? bi[k + n / 2] : bi[k];                     \
                const double ob = __shfl_xor(send, off);                                                              \
                const int oi = __shfl_xor(sendi, off);                                                                \
                const bool take = ob > keep || (ob == keep && oi < keepi);                                            \
                best[k] = take ? ob : keep; bi[k] = take ? oi : keepi;                                                \
            }                                                                                                         \
        }
        HPR_HALVE(0) HPR_HALVE(1) HPR_HALVE(2) HPR_HALVE(3)
#undef HPR_HALVE
#pragma unroll
        for (int off = ST / 2; off > 0; off >>= 1) {
            const double ob = __shfl_xor(best[0], off);
            const int oi = __shfl_xor(bi[0], off);
            if (ob > best[0] || (ob == best[0] && oi < bi[0])) { best[0] = ob; bi[0] = oi; }
        }
        if (COOP) {                                                  // merge the four quarters (ties: smaller position)
            const int pb = round & 1;
            if (slot) { s_rv[pb][wave][kq] = best[0]; s_ri[pb][wave][kq] = bi[0]; }
            __syncthreads();
            best[0] = s_rv[pb][0][kq]; bi[0] = s_ri[pb][0][kq];
#pragma unroll
            for (int w = 1; w < NW; ++w) {
                const double ob = s_rv[pb][w][kq];
                const int oi = s_ri[pb][w][kq];
                if (ob > best[0] || (ob == best[0] && oi < bi[0])) { best[0] = ob; bi[0] = oi; }
            }
        }
        const double myv = best[0];
        const int myi = bi[0];
        if (state == 0) {
            // support point of S_i in direction dir: best flipped point, or the origin of the flipped space (value 0)
            d3 a;
            if (myv > 0.0 && myi < NS) a = d3{fx[myi], fy[myi], fz[myi]} - pi;
            else a = neg(pi);
            if (dim == 0) {                       // first vertex
                sc = a; dir = neg(a); dim = 1;
            } else if (dim == 1) {                // second vertex, then the line case
                if (dot(a, dir) < 0.0) state = 1;
                else {
                    sb = a;
                    const d3 cb = sc - sb;
                    dir = cross(cross(cb, neg(sb)), cb);
                    if (dir.x == 0.0 && dir.y == 0.0 && dir.z == 0.0) {      // origin on the line: any perpendicular
                        dir = cross(cb, d3{1, 0, 0});
                        if (dir.x == 0.0 && dir.y == 0.0 && dir.z == 0.0) dir = cross(cb, d3{0, 0, -1});
                    }
                    dim = 2;
                }
            } else {
                if (dot(a, dir) < 0.0) state = 1;                            // could not pass the origin: outside
                else {
                    sa = a;
                    const d3 ao = neg(sa);
                    if (dim == 2) {                                          // triangle a, b, c
                        const d3 n = cross(sb - sa, sc - sa);
                        if (dot(cross(sb - sa, n), ao) > 0.0) { sc = sa; dir = cross(cross(sb - sa, ao), sb - sa); }
                        else if (dot(cross(n, sc - sa), ao) > 0.0) { sb = sa; dir = cross(cross(sc - sa, ao), sc - sa); }
                        else if (dot(n, ao) > 0.0) { sd = sc; sc = sb; sb = sa; dir = n; dim = 3; }
                        else { sd = sb; sb = sa; dir = neg(n); dim = 3; }
                    } else {                                                 // tetrahedron a, b, c, d
                        const d3 abc = cross(sb - sa, sc - sa), acd = cross(sc - sa, sd - sa), adb = cross(sd - sa, sb - sa);
                        if (dot(abc, ao) > 0.0) { sd = sc; sc = sb; sb = sa; dir = abc; }
                        else if (dot(acd, ao) > 0.0) { sb = sa; dir = acd; }
                        else if (dot(adb, ao) > 0.0) { sc = sd; sd = sb; sb = sa; dir = adb; }
                        else state = 2;                                      // origin enclosed: hidden
                    }
                    if (state == 0 && dir.x == 0.0 && dir.y == 0.0 && dir.z == 0.0) state = 2;   // degenerate: on the boundary
                }
            }
        }
    }
#ifdef PD_HPR_STATS
    if (lane == 0) { atomicAdd(&g_hpr_stats[COARSE ? 0 : 1][0], 1ull); atomicAdd(&g_hpr_stats[COARSE ? 0 : 1][1], (unsigned long long)wave_rounds); }
    if (owner) { atomicAdd(&g_hpr_stats[COARSE ? 0 : 1][2], 1ull); atomicAdd(&g_hpr_stats[COARSE ? 0 : 1][3], (unsigned long long)my_rounds);
                 if (state == 0) atomicAdd(&g_hpr_stats[COARSE ? 0 : 1][4], 1ull);
                 atomicAdd(&g_hpr_stats[COARSE ? 0 : 1][8 + min(my_rounds, 63) / 8], 1ull); }
#endif
    if (!COARSE) {
        if (owner) vis[(size_t)v * N + q] = (state == 1) ? 1 : 0;
    } else {
        if (owner) outside[(size_t)v * N + q] = state != 2;           // 0: enclosed by the coarse hull (hidden, and never a support point)
                                                                      // (entries marked 2 -- coarse-set members -- are left alone)
    }
}

// second-level inputs: the support set = all points outside the coarse hull (coordinates + original index), and the query list =
// those of them that no cheaper test accepted.  One workgroup per view walks the cloud in order (ballot ranks + a running
// offset): both lists come out in ascending cloud order, which is what lets the support scan drop its index tie-break.
// Members of the coarse set (outside == 2) are certain hull vertices: marked visible here, supports but never queries.
__global__ __launch_bounds__(1024) void k_hpr_build(const double* __restrict__ flipped, int N, const uint8_t* __restrict__ outside,
                                                    const uint8_t* __restrict__ skip, double* __restrict__ ss, int* __restrict__ sidx,
                                                    int* __restrict__ scount, int* __restrict__ count2, int* __restrict__ list2,
                                                    uint8_t* __restrict__ vis) {
    __shared__ int s_o[16], s_q[16];
    const int v = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double* f = flipped + (size_t)v * 3 * N;
    double* so = ss + (size_t)v * 3 * N;
    int base_o = 0, base_q = 0;
    for (int i0 = 0; i0 < N; i0 += 1024) {
        const int i = i0 + threadIdx.x;
        const uint8_t o = i < N ? outside[(size_t)v * N + i] : 0;
        const bool sk = i < N && skip != nullptr && skip[(size_t)v * N + i];
        const bool out = o != 0, qry = o == 1 && !sk;
        if (o == 2 && !sk) vis[(size_t)v * N + i] = 1;
        const unsigned long long bo = __ballot(out), bq = __ballot(qry);
        if (lane == 0) { s_o[wave] = __popcll(bo); s_q[wave] = __popcll(bq); }
        __syncthreads();
        int po = base_o, pq = base_q, to = 0, tq = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            po += w < wave ? s_o[w] : 0; pq += w < wave ? s_q[w] : 0;
            to += s_o[w]; tq += s_q[w];
        }
        if (out) {
            const int pos = po + __popcll(bo & ((1ull << lane) - 1ull));
            so[pos] = f[i]; so[N + pos] = f[N + i]; so[2 * (size_t)N + pos] = f[2 * (size_t)N + i];
            sidx[(size_t)v * N + pos] = i;
        }
        if (qry) list2[(size_t)v * N + pq + __popcll(bq & ((1ull << lane) - 1ull))] = i;
        base_o += to; base_q += tq;
        __syncthreads();
    }
    if (threadIdx.x == 0) { scount[v] = base_o; count2[v] = base_q; }
}

// coarse support set: the extreme point of the flipped cloud in each of KC Fibonacci-sphere directions (ties: smallest index)
__global__ __launch_bounds__(256) void k_hpr_extremes(const double* __restrict__ flipped, int N, int KC, double* __restrict__ cs,
                                                      int* __restrict__ cidx, uint8_t* __restrict__ mark) {
    const int v = blockIdx.y;
    const double* fx = flipped + (size_t)v * 3 * N;
    const double* fy = fx + N;
    const double* fz = fy + N;
    // a block owns QPW directions; its four waves scan a quarter of the cloud each and merge through LDS (KC / QPW blocks of
    // 4 waves per view fill the chip, one wave per QPW directions did not: 300 -> 100 us)
    __shared__ double s_b[4][QPW];
    __shared__ int s_i[4][QPW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k0 = blockIdx.x * QPW;
    if (k0 >= KC) return;
    double dx[QPW], dy[QPW], dz[QPW], best[QPW];
    int bi[QPW];
    // the block's QPW Fibonacci directions: one lane each (f64 sin / cos of a large argument are hundreds of instructions; every
    // lane computing all of them was most of this kernel's time), shared through LDS
    __shared__ double s_d[QPW][3];
    if (threadIdx.x < QPW) {
        const int kk = min(k0 + (int)threadIdx.x, KC - 1);
        const double z = 1.0 - (2.0 * kk + 1.0) / KC, r = sqrt(fmax(0.0, 1.0 - z * z)), phi = kk * 2.399963229728653;
        s_d[threadIdx.x][0] = r * cos(phi); s_d[threadIdx.x][1] = r * sin(phi); s_d[threadIdx.x][2] = z;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < QPW; ++k) {
        dx[k] = s_d[k][0]; dy[k] = s_d[k][1]; dz[k] = s_d[k][2];
        best[k] = -1.0e300; bi[k] = 0x7fffffff;
    }
    // four points per lane in flight: with two waves per SIMD a single dependent load per iteration leaves the L2 latency exposed
    for (int j = wave * 64 + lane; j < N; j += 1024) {
        double x[4], y[4], z[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ju = min(j + u * 256, N - 1);                   // (a clamped duplicate cannot change an argmax)
            x[u] = fx[ju]; y[u] = fy[ju]; z[u] = fz[ju];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ju = j + u * 256;
            if (ju < N) {
#pragma unroll
                for (int k = 0; k < QPW; ++k) {
                    const double val = dx[k] * x[u] + dy[k] * y[u] + dz[k] * z[u];
                    if (val > best[k]) { best[k] = val; bi[k] = ju; }
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < QPW; ++k) {
        double b = best[k];
        int id = bi[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ob = __shfl_xor(b, off);
            const int oi = __shfl_xor(id, off);
            if (ob > b || (ob == b && oi < id)) { b = ob; id = oi; }
        }
        if (lane == 0) { s_b[wave][k] = b; s_i[wave][k] = id; }
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int k = 0; k < QPW; ++k) {
        double b = s_b[0][k];
        int id = s_i[0][k];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const double ob = s_b[w][k];
            const int oi = s_i[w][k];
            if (ob > b || (ob == b && oi < id)) { b = ob; id = oi; }
        }
        if (lane == 0 && k0 + k < KC) {
            // The hull is that of the cloud AND the eye (the origin of the flipped space): in a direction where every point
            // has a negative projection the origin is the extreme element, not a cloud point -- the slot then holds the origin
            // (support value 0, which the GJK step already treats as "the origin").  Otherwise the point is a certain hull
            // vertex: member of the coarse set, marked 2, never queried.
            double* c = cs + (size_t)v * 3 * KC;
            const bool real = b > 0.0;
            c[k0 + k] = real ? fx[id] : 0.0; c[KC + k0 + k] = real ? fy[id] : 0.0; c[2 * (size_t)KC + k0 + k] = real ? fz[id] : 0.0;
            cidx[(size_t)v * KC + k0 + k] = real ? id : -1;
            if (real) mark[(size_t)v * N + id] = 2;
        }
    }
}

static size_t a256(size_t x) { return (x + 255) & ~(size_t)255; }
static size_t flipped_bytes(int V, int N) { return a256((size_t)V * 3 * (size_t)(N > 0 ? N : 1) * sizeof(double)); }
static size_t lists_bytes(int V, int N) { return a256((size_t)V * ((size_t)N + 64) * sizeof(int)); }
extern "C" size_t pdhip_hpr_ws_bytes(int V, int N) {
    return 2 * flipped_bytes(V, N) + 3 * lists_bytes(V, N) + a256((size_t)V * N) + a256((size_t)V * 3 * HPR_KC * sizeof(double)) +
           a256((size_t)V * HPR_KC * sizeof(int));
}

__global__ void k_hpr_fill_count(int* __restrict__ c, int V, int N) { if ((int)threadIdx.x < V) c[threadIdx.x] = N; }
__global__ void k_hpr_iota(int* __restrict__ idx, int N) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) idx[(size_t)blockIdx.y * N + i] = i;
}

extern "C" int pdhip_hidden_point_removal(const float* points, int N, const double* eyes_dev, int V, double radius,
                                          const uint8_t* skip, uint8_t* visibility, void* ws, void* stream) {
    PD_REQUIRE(V > 0 && N >= 0, "pdhip_hidden_point_removal: bad sizes");
    if (N == 0) return PDHIP_OK;
    PD_REQUIRE(points && eyes_dev && visibility && ws, "pdhip_hidden_point_removal: null pointer");
    PD_REQUIRE(V <= 64, "pdhip_hidden_point_removal: at most 64 views");
    hipStream_t s = as_stream(stream);
    char* p = reinterpret_cast<char*>(ws);
    double* flipped = reinterpret_cast<double*>(p); p += flipped_bytes(V, N);
    double* ss = reinterpret_cast<double*>(p); p += flipped_bytes(V, N);          // second-level support set (outside points)
    int* count = reinterpret_cast<int*>(p); int* list = count + 64; p += lists_bytes(V, N);
    int* count2 = reinterpret_cast<int*>(p); int* list2 = count2 + 64; p += lists_bytes(V, N);
    int* scount = reinterpret_cast<int*>(p); int* sidx = scount + 64; p += lists_bytes(V, N);
    uint8_t* outside = reinterpret_cast<uint8_t*>(p); p += a256((size_t)V * N);
    double* cs = reinterpret_cast<double*>(p); p += a256((size_t)V * 3 * HPR_KC * sizeof(double));
    int* cidx = reinterpret_cast<int*>(p);
    dim3 gf(min(cdiv(N, 256), 256), V);
    k_hpr_flip<<<gf, 256, 0, s>>>(points, N, eyes_dev, radius, flipped);
    PD_HIP(hipMemsetAsync(count, 0, 64 * sizeof(int), s));
    PD_HIP(hipMemsetAsync(count2, 0, 64 * sizeof(int), s));
    PD_HIP(hipMemsetAsync(scount, 0, 64 * sizeof(int), s));
    k_hpr_collect<<<gf, 256, 0, s>>>(skip, N, count, list, visibility);      // marks the skipped points visible; `list` = the queries
    dim3 gg(cdiv(N, 4 * QPW), V), gg4(cdiv(min(N, HPR_NARROW_BELOW), 4), V);
    constexpr int KC = HPR_KC;
    if (N > 4 * HPR_KC) {            // the coarse level pays off only when the cloud is much larger than the coarse set
        dim3 ge(cdiv(KC, QPW), V);
        PD_HIP(hipMemsetAsync(outside, 0, (size_t)V * N, s));
        k_hpr_extremes<<<ge, 256, 0, s>>>(flipped, N, KC, cs, cidx, outside);
        k_hpr_gjk<true, QPW, false, 4><<<gg, 256, 0, s>>>(flipped, N, nullptr, nullptr, nullptr, cs, cidx, KC, nullptr, outside, 0, 0x7fffffff, skip);
        k_hpr_build<<<V, 1024, 0, s>>>(flipped, N, outside, skip, ss, sidx, scount, count2, list2, visibility);
        k_hpr_gjk<false, QPW, false, 4><<<gg, 256, 0, s>>>(flipped, N, count2, list2, visibility, ss, sidx, N, scount, nullptr, HPR_NARROW_BELOW, 0x7fffffff, nullptr);
        k_hpr_gjk<false, 4, true, HPR_COOP_WAVES><<<gg4, HPR_COOP_WAVES * 64, 0, s>>>(flipped, N, count2, list2, visibility, ss, sidx, N, scount, nullptr, 0, HPR_NARROW_BELOW, nullptr);
    } else {                         // one level: support set = the whole cloud
        k_hpr_iota<<<gf, 256, 0, s>>>(sidx, N);
        k_hpr_fill_count<<<1, 64, 0, s>>>(scount, V, N);
        k_hpr_gjk<false, QPW, false, 4><<<gg, 256, 0, s>>>(flipped, N, count, list, visibility, flipped, sidx, N, scount, nullptr, HPR_NARROW_BELOW, 0x7fffffff, nullptr);
        k_hpr_gjk<false, 4, true, HPR_COOP_WAVES><<<gg4, HPR_COOP_WAVES * 64, 0, s>>>(flipped, N, count, list, visibility, flipped, sidx, N, scount, nullptr, 0, HPR_NARROW_BELOW, nullptr);
    }
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}
