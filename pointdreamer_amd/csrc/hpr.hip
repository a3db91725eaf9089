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
// Every verdict is CERTIFIED: "visible" by a separating direction d with  d.p'_i - max(max_j d.p'_j, 0) > rounding bound,
// "hidden" by a tetrahedron of cloud points (or the eye) whose four orientation determinants around p'_i pass Shewchuk's
// static filter.  A verdict the f64 filter cannot certify, a degenerate simplex and a query still running at the round cap
// go to k_hpr_exact: the same iteration in double-double arithmetic (2^-104) over the same support set, certified with
// double-double bounds; what even that cannot certify (exact coplanarity / duplicate points) is counted in the workspace
// counters (pdhip_hpr_read_counters) and reported hidden.  The result is therefore the vertex set of the exact hull of the
// f64 flipped points; qhull (open3d, scipy) differs from it only for points within its own merge tolerance (~1e-13 * radius)
// of a facet.  open3d itself is absent (PARITY UNPINNED); the oracle drives the same qhull through scipy.
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
// ---- double-double (unevaluated sum hi + lo, |lo| <= ulp(hi)/2): the arithmetic of the exact fallback.  This unit is
// compiled with -ffp-contract=off, so the error-free transformations below are not re-associated; products use explicit fma.
struct dd { double hi, lo; };
__device__ __forceinline__ dd dd_quick(double a, double b) { const double s = a + b; return {s, b - (s - a)}; }
__device__ __forceinline__ dd dd_two_sum(double a, double b) { const double s = a + b, bb = s - a; return {s, (a - (s - bb)) + (b - bb)}; }
__device__ __forceinline__ dd dd_two_prod(double a, double b) { const double p = a * b; return {p, fma(a, b, -p)}; }
__device__ __forceinline__ dd operator+(dd a, dd b) {
    dd s = dd_two_sum(a.hi, b.hi);
    const dd t = dd_two_sum(a.lo, b.lo);
    s = dd_quick(s.hi, s.lo + t.hi);
    return dd_quick(s.hi, s.lo + t.lo);
}
__device__ __forceinline__ dd operator-(dd a) { return {-a.hi, -a.lo}; }
__device__ __forceinline__ dd operator-(dd a, dd b) { return a + (-b); }
__device__ __forceinline__ dd operator*(dd a, dd b) {
    dd p = dd_two_prod(a.hi, b.hi);
    p.lo += a.hi * b.lo + a.lo * b.hi;
    return dd_quick(p.hi, p.lo);
}
__device__ __forceinline__ dd operator/(dd a, dd b) {                 // three quotient digits, each exact to f64: ~2^-104 relative
    const double q1 = a.hi / b.hi;
    dd r = a - b * dd{q1, 0.0};
    const double q2 = r.hi / b.hi;
    r = r - b * dd{q2, 0.0};
    const double q3 = r.hi / b.hi;
    return dd_quick(q1, q2) + dd{q3, 0.0};
}
__device__ __forceinline__ dd dd_from(double a) { return {a, 0.0}; }
__device__ __forceinline__ dd dd_diff(double a, double b) { return dd_two_sum(a, -b); }          // exact a - b
__device__ __forceinline__ double sgn_of(double a) { return a; }
__device__ __forceinline__ double sgn_of(dd a) { return a.hi != 0.0 ? a.hi : a.lo; }               // sign carrier
__device__ __forceinline__ double mag_of(double a) { return fabs(a); }
__device__ __forceinline__ double mag_of(dd a) { return fabs(a.hi) + fabs(a.lo); }
__device__ __forceinline__ bool is_zero(double a) { return a == 0.0; }
__device__ __forceinline__ bool is_zero(dd a) { return a.hi == 0.0 && a.lo == 0.0; }

template <typename T> struct v3 { T x, y, z; };
typedef v3<double> d3;
template <typename T> __device__ __forceinline__ v3<T> operator-(v3<T> a, v3<T> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename T> __device__ __forceinline__ v3<T> neg(v3<T> a) { return {-a.x, -a.y, -a.z}; }
template <typename T> __device__ __forceinline__ T dot(v3<T> a, v3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename T> __device__ __forceinline__ v3<T> cross(v3<T> a, v3<T> b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
template <typename T> __device__ __forceinline__ bool zero3(v3<T> a) { return is_zero(a.x) && is_zero(a.y) && is_zero(a.z); }

// |det[u; v; w]| certainly non-zero?  Returns the determinant's sign (+1 / -1) when |det| exceeds the rounding bound of its
// evaluation, else 0.  f64: u, v, w are ROUNDED differences of exact points and the bound is Shewchuk's static orient3d
// filter (7 + 56 eps) eps * permanent; double-double: exact differences, bound 2^-96 * permanent (>= 100x the dd error).
template <typename T> __device__ __forceinline__ int det_sign(v3<T> u, v3<T> v, v3<T> w) {
    const T m1 = v.y * w.z, m2 = v.z * w.y, m3 = v.z * w.x, m4 = v.x * w.z, m5 = v.x * w.y, m6 = v.y * w.x;
    const T det = u.x * (m1 - m2) + u.y * (m3 - m4) + u.z * (m5 - m6);
    const double perm = (mag_of(m1) + mag_of(m2)) * mag_of(u.x) + (mag_of(m3) + mag_of(m4)) * mag_of(u.y) + (mag_of(m5) + mag_of(m6)) * mag_of(u.z);
    const double bound = (sizeof(T) == sizeof(double) ? 7.771561172376103e-16 : 1.2621774483536189e-29) * perm;
    const double d = sgn_of(det);
    return mag_of(det) > bound ? (d > 0.0 ? 1 : -1) : 0;
}

// ---- the boolean GJK state of one query ("is the origin inside conv(S_i)", S_i = {p'_j - p'_i} U {-p'_i}); the simplex
// keeps the cloud index of every vertex (-1 = the eye, i.e. the origin of the flipped space) for the final certificate.
// state: 0 running, 1 visible (certified by the caller before the step), 2 hidden (certified here), 3 not certifiable
template <typename T> struct Gjk {
    v3<T> sa, sb, sc, sd, dir;
    int ia, ib, ic, id, dim, state;
};
// one step with the support point a (= p'_j - p'_i, index ai) found in direction g.dir; the caller has already handled
// "support does not pass the origin" (visible).  The four-vertex case certifies "enclosed" with det_sign.
template <typename T> __device__ __forceinline__ void gjk_step(Gjk<T>& g, v3<T> a, int ai) {
    const T zero = T{};
    if (zero3(a)) { g.state = 3; return; }          // support coincides with the query: duplicate point
    if (g.dim == 0) {                               // first vertex
        g.sc = a; g.ic = ai; g.dir = neg(a); g.dim = 1;
    } else if (g.dim == 1) {                        // second vertex, then the line case
        g.sb = a; g.ib = ai;
        const v3<T> cb = g.sc - g.sb;
        g.dir = cross(cross(cb, neg(g.sb)), cb);
        if (zero3(g.dir)) { g.state = 3; return; }  // origin on the line within rounding
        g.dim = 2;
    } else {
        g.sa = a; g.ia = ai;
        const v3<T> ao = neg(g.sa);
        if (g.dim == 2) {                           // triangle a, b, c
            const v3<T> n = cross(g.sb - g.sa, g.sc - g.sa);
            if (sgn_of(dot(cross(g.sb - g.sa, n), ao)) > 0.0) { g.sc = g.sa; g.ic = g.ia; g.dir = cross(cross(g.sb - g.sa, ao), g.sb - g.sa); }
            else if (sgn_of(dot(cross(n, g.sc - g.sa), ao)) > 0.0) { g.sb = g.sa; g.ib = g.ia; g.dir = cross(cross(g.sc - g.sa, ao), g.sc - g.sa); }
            else if (sgn_of(dot(n, ao)) > 0.0) { g.sd = g.sc; g.id = g.ic; g.sc = g.sb; g.ic = g.ib; g.sb = g.sa; g.ib = g.ia; g.dir = n; g.dim = 3; }
            else { g.sd = g.sb; g.id = g.ib; g.sb = g.sa; g.ib = g.ia; g.dir = neg(n); g.dim = 3; }
        } else {                                    // tetrahedron a, b, c, d
            const v3<T> abc = cross(g.sb - g.sa, g.sc - g.sa), acd = cross(g.sc - g.sa, g.sd - g.sa), adb = cross(g.sd - g.sa, g.sb - g.sa);
            if (sgn_of(dot(abc, ao)) > 0.0) { g.sd = g.sc; g.id = g.ic; g.sc = g.sb; g.ic = g.ib; g.sb = g.sa; g.ib = g.ia; g.dir = abc; }
            else if (sgn_of(dot(acd, ao)) > 0.0) { g.sb = g.sa; g.ib = g.ia; g.dir = acd; }
            else if (sgn_of(dot(adb, ao)) > 0.0) { g.sc = g.sd; g.ic = g.id; g.sd = g.sb; g.id = g.ib; g.sb = g.sa; g.ib = g.ia; g.dir = adb; }
            else {
                // origin enclosed (in this arithmetic).  Certificate: the barycentric numerators -det(B,C,D), det(A,C,D),
                // -det(A,B,D), det(A,B,C) of the origin all have one certain sign  <=>  p'_i strictly inside the tetrahedron
                const int s0 = -det_sign(g.sb, g.sc, g.sd), s1 = det_sign(g.sa, g.sc, g.sd), s2 = -det_sign(g.sa, g.sb, g.sd),
                          s3 = det_sign(g.sa, g.sb, g.sc);
                g.state = (s0 != 0 && s0 == s1 && s1 == s2 && s2 == s3) ? 2 : 3;
                return;
            }
        }
        if (zero3(g.dir)) g.state = 3;              // degenerate simplex: the origin is on its boundary within rounding
    }
    (void)zero;
}

__global__ void k_hpr_flip(const float* __restrict__ pts, int N, const double* __restrict__ eyes, double radius,
                           double* __restrict__ flipped /*[V][3][N]*/, unsigned long long* __restrict__ maxabs /*[V] f64 bits*/) {
    // open3d PointCloud::HiddenPointRemoval: p' = q + 2 (radius - |q|) q / |q| evaluated as q + ((2 (radius - n)) q) / n
    const int v = blockIdx.y;
    const double ex = eyes[3 * v], ey = eyes[3 * v + 1], ez = eyes[3 * v + 2];
    double m = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const double qx = (double)pts[3 * i] - ex, qy = (double)pts[3 * i + 1] - ey, qz = (double)pts[3 * i + 2] - ez;
        double n = sqrt(qx * qx + qy * qy + qz * qz);
        if (n == 0.0) n = 0.0001;
        const double k = 2.0 * (radius - n);
        double* f = flipped + (size_t)v * 3 * N;
        const double x = qx + (k * qx) / n, y = qy + (k * qy) / n, z = qz + (k * qz) / n;
        f[i] = x; f[N + i] = y; f[2 * (size_t)N + i] = z;
        m = fmax(m, fmax(fabs(x), fmax(fabs(y), fabs(z))));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) atomicMax(&maxabs[v], (unsigned long long)__double_as_longlong(m));   // non-negative f64: bit order = value order
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
                                                    const uint8_t* __restrict__ skip, const unsigned long long* __restrict__ maxabs,
                                                    int* __restrict__ unc_count, int* __restrict__ unc_list, int* __restrict__ unc_seed) {
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
    d3 pi = {0, 0, 0};
    Gjk<double> g;
    g.sa = g.sb = g.sc = g.sd = d3{0, 0, 0}; g.dir = d3{0, 0, 1};
    g.ia = g.ib = g.ic = g.id = -1;
    g.dim = 0;                     // simplex size; phases: 0 -> fetch c, 1 -> fetch b, >= 2 -> main loop
    g.state = owner ? 0 : 2;       // 0 running, 1 visible (certified), 2 hidden (certified) / not a query, 3 not certifiable in f64
    int& state = g.state;
    d3& dir = g.dir;
    if (owner) {
        pi = {qfx[q], qfy[q], qfz[q]};
        dir = pi;                  // start looking straight out along the point's own ray
    }
    // rounding bound of one support value (dx x + dy y) + dz z in f64: <= 3.01 u (|dx| + |dy| + |dz|) max|coordinate|
    const double rb = __longlong_as_double((long long)maxabs[v]) * (8.0 * 1.1102230246251565e-16);
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
            const bool real = myv > 0.0 && myi < NS;
            // "the support does not pass the origin"  <=>  p'_i is the strict maximiser of dir over the cloud and the eye.
            // Position form with the scan's own op order, so that the rounding bound rb applies to both sides.
            const double di = dir.x * pi.x + dir.y * pi.y + dir.z * pi.z;
            const double gap = di - (real ? myv : 0.0);
            const double tol = rb * (fabs(dir.x) + fabs(dir.y) + fabs(dir.z));
            if (gap > tol) state = 1;                                     // certified visible
            else if (gap > 0.0 && g.dim >= 1) state = 3;                  // visible in f64, but inside the rounding bound
            else {
                d3 a;
                int ai = -1;
                if (real) { a = d3{fx[myi], fy[myi], fz[myi]} - pi; ai = sidx[myi]; }
                else a = neg(pi);
                gjk_step(g, a, ai);
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
        if (owner) {
            vis[(size_t)v * N + q] = (state == 1) ? 1 : 0;
            if (state == 0 || state == 3)                             // round cap reached / not certifiable: exact fallback
            {
                    const int pos = atomicAdd(&unc_count[v], 1);
                    unc_list[(size_t)v * N + pos] = q;
                    int* sd = unc_seed + ((size_t)v * N + pos) * 4;          // the simplex it stopped at seeds the fallback
                    sd[0] = g.dim >= 1 ? g.ic : -2; sd[1] = g.dim >= 2 ? g.ib : -2; sd[2] = g.dim >= 3 ? g.id : -2; sd[3] = -2;
                }
        }
    } else {
        if (owner) outside[(size_t)v * N + q] = state != 2;           // 0: CERTIFIED enclosed by the coarse hull (hidden, and never a support
                                                                      // point); anything else -- also an uncertified verdict -- goes on to the
                                                                      // second level (entries marked 2 -- coarse-set members -- are left alone)
    }
}

// ---- distance GJK (Gilbert-Johnson-Keerthi with the closest-point sub-algorithm; region tests after Ericson, Real-Time
// Collision Detection 5.1.5 / 5.1.6): the squared distance of the simplex to the origin decreases strictly every round, so it
// terminates on a polytope -- unlike the boolean variant above, which can revisit a face when the origin projects outside it.
// Used by the double-double fallback only (a few dozen queries per view set); T needs + - * / and sgn_of.
template <typename T> struct Simplex { v3<T> w[4]; int idx[4]; int n; };
template <typename T> __device__ __forceinline__ v3<T> scale(v3<T> a, T k) { return {a.x * k, a.y * k, a.z * k}; }
template <typename T> __device__ __forceinline__ v3<T> add3(v3<T> a, v3<T> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename T> __device__ __forceinline__ bool le0(T a) { return sgn_of(a) <= 0.0; }
template <typename T> __device__ __forceinline__ bool ge0(T a) { return sgn_of(a) >= 0.0; }

// closest point of triangle (a, b, c) to the origin; keep[] = which of the three vertices support it
template <typename T> __device__ v3<T> closest_triangle(v3<T> a, v3<T> b, v3<T> c, bool keep[3]) {
    const v3<T> ab = b - a, ac = c - a, ap = neg(a);
    const T d1 = dot(ab, ap), d2 = dot(ac, ap);
    keep[0] = keep[1] = keep[2] = false;
    if (le0(d1) && le0(d2)) { keep[0] = true; return a; }
    const v3<T> bp = neg(b);
    const T d3 = dot(ab, bp), d4 = dot(ac, bp);
    if (ge0(d3) && le0(d4 - d3)) { keep[1] = true; return b; }
    const T vc = d1 * d4 - d3 * d2;
    if (le0(vc) && ge0(d1) && le0(d3)) { keep[0] = keep[1] = true; return add3(a, scale(ab, d1 / (d1 - d3))); }
    const v3<T> cp = neg(c);
    const T d5 = dot(ab, cp), d6 = dot(ac, cp);
    if (ge0(d6) && le0(d5 - d6)) { keep[2] = true; return c; }
    const T vb = d5 * d2 - d1 * d6;
    if (le0(vb) && ge0(d2) && le0(d6)) { keep[0] = keep[2] = true; return add3(a, scale(ac, d2 / (d2 - d6))); }
    const T va = d3 * d6 - d5 * d4;
    if (le0(va) && ge0(d4 - d3) && ge0(d5 - d6)) {
        keep[1] = keep[2] = true;
        return add3(b, scale(c - b, (d4 - d3) / ((d4 - d3) + (d5 - d6))));
    }
    keep[0] = keep[1] = keep[2] = true;
    const T den = (va + vb) + vc;
    return add3(a, add3(scale(ab, vb / den), scale(ac, vc / den)));
}

// closest point of the simplex to the origin; the simplex is reduced to the vertices that support it.
// Returns false when the origin is inside the tetrahedron (v is then meaningless, the simplex is left whole).
template <typename T> __device__ bool closest_simplex(Simplex<T>& S, v3<T>& v) {
    if (S.n == 1) { v = S.w[0]; return true; }
    if (S.n == 2) {
        const v3<T> a = S.w[0], b = S.w[1], ab = b - a;
        const T t = dot(neg(a), ab), den = dot(ab, ab);
        if (le0(t)) { S.n = 1; v = a; return true; }
        if (ge0(t - den)) { S.w[0] = b; S.idx[0] = S.idx[1]; S.n = 1; v = b; return true; }
        v = add3(a, scale(ab, t / den));
        return true;
    }
    bool keep[3];
    if (S.n == 3) {
        v = closest_triangle(S.w[0], S.w[1], S.w[2], keep);
        int m = 0;
        for (int i = 0; i < 3; ++i) if (keep[i]) { S.w[m] = S.w[i]; S.idx[m] = S.idx[i]; ++m; }
        S.n = m;
        return true;
    }
    // tetrahedron: the faces the origin lies outside of; the nearest of their closest points wins
    const int F[4][4] = {{0, 1, 2, 3}, {0, 2, 3, 1}, {0, 3, 1, 2}, {1, 3, 2, 0}};      // face (i, j, k), opposite vertex l
    bool any = false, bk[3] = {false, false, false};
    int bf = -1;
    v3<T> bv = S.w[0];
    T bd = T{};
    for (int f = 0; f < 4; ++f) {
        const v3<T> a = S.w[F[f][0]], b = S.w[F[f][1]], c = S.w[F[f][2]], d = S.w[F[f][3]];
        const v3<T> n = cross(b - a, c - a);
        const T sp = dot(neg(a), n), sd = dot(d - a, n);
        const double so = sgn_of(sp), sdd = sgn_of(sd);
        const bool outside = sdd == 0.0 ? true : ((so > 0.0 && sdd < 0.0) || (so < 0.0 && sdd > 0.0));
        if (!outside) continue;
        const v3<T> q = closest_triangle(a, b, c, keep);
        const T dq = dot(q, q);
        if (!any || sgn_of(dq - bd) < 0.0) { any = true; bd = dq; bv = q; bf = f; bk[0] = keep[0]; bk[1] = keep[1]; bk[2] = keep[2]; }
    }
    if (!any) return false;
    v3<T> nw[3];
    int ni[3], m = 0;
    for (int i = 0; i < 3; ++i) if (bk[i]) { nw[m] = S.w[F[bf][i]]; ni[m] = S.idx[F[bf][i]]; ++m; }
    for (int i = 0; i < m; ++i) { S.w[i] = nw[i]; S.idx[i] = ni[i]; }
    S.n = m;
    v = bv;
    return true;
}

// ---- exact fallback: one 512-lane block per query the f64 pass could not certify.  Same iteration, double-double state and
// double-double support values over the same support set (any superset of the hull vertices), duplicates of the query with a
// larger cloud index excluded from S_i (of coinciding points the smallest index is the hull vertex), up to 512 rounds.
__global__ __launch_bounds__(512) void k_hpr_exact(const double* __restrict__ flipped, int N, const int* __restrict__ unc_count,
                                                   const int* __restrict__ unc_list, const int* __restrict__ unc_seed, uint8_t* __restrict__ vis,
                                                   const double* __restrict__ ss, const int* __restrict__ sidx_all, int scap,
                                                   const int* __restrict__ scount, const unsigned long long* __restrict__ maxabs,
                                                   int* __restrict__ counters /*[V][4]: exact queries, unresolved, rounds, -*/) {
    const int v = blockIdx.y;
    const int nq = unc_count[v];
    __shared__ double s_hi[8], s_lo[8];
    __shared__ int s_i[8];
    const double* fx = ss + (size_t)v * 3 * scap;
    const double* fy = fx + scap;
    const double* fz = fy + scap;
    const int* sidx = sidx_all + (size_t)v * scap;
    const int NS = scount[v];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double rb = __longlong_as_double((long long)maxabs[v]) * 1.2621774483536189e-29;          // 2^-96 max|coordinate|
    for (int u = blockIdx.x; u < nq; u += gridDim.x) {
        const int q = unc_list[(size_t)v * N + u];
        const double px = flipped[(size_t)v * 3 * N + q], py = flipped[(size_t)v * 3 * N + N + q], pz = flipped[(size_t)v * 3 * N + 2 * (size_t)N + q];
        Simplex<dd> S;
        S.n = 0;
        v3<dd> dir = {dd_from(px), dd_from(py), dd_from(pz)};         // first direction: straight out along the point's own ray
        v3<dd> vclose = dir;
        int state = 0;                  // 0 running, 1 visible (certified), 2 hidden (certified), 3 not certifiable
        int rounds = 0;
        {   // seed: the vertices the f64 iteration stopped at (cloud indices, -1 = the eye) -- its closest point is the start
            const int* sd = unc_seed + ((size_t)v * N + u) * 4;
            const double* cx = flipped + (size_t)v * 3 * N;
            for (int k = 0; k < 3; ++k) {
                const int id = sd[k];
                if (id == -2 || id == q) continue;
                bool dup = false;
                for (int m = 0; m < S.n; ++m) dup = dup || S.idx[m] == id;
                if (dup) continue;
                v3<dd> a = id >= 0 ? v3<dd>{dd_diff(cx[id], px), dd_diff(cx[N + id], py), dd_diff(cx[2 * (size_t)N + id], pz)}
                                   : v3<dd>{dd_from(-px), dd_from(-py), dd_from(-pz)};
                if (zero3(a)) continue;
                for (int m = S.n; m > 0; --m) { S.w[m] = S.w[m - 1]; S.idx[m] = S.idx[m - 1]; }
                S.w[0] = a; S.idx[0] = id; ++S.n;
            }
            if (S.n > 0 && closest_simplex(S, vclose) && !zero3(vclose)) dir = neg(vclose);
            else { S.n = 0; vclose = dir; }
        }
        for (; rounds < 512 && state == 0; ++rounds) {
            // support scan in double-double: (dx x + dy y) + dz z with exact inputs x, y, z
            dd best = {-1.0e300, 0.0};
            int bi = 0x7fffffff;
            for (int j = threadIdx.x; j < NS; j += 512) {
                const double x = fx[j], y = fy[j], zz = fz[j];
                const int jo = sidx[j];
                if (jo == q || (jo > q && x == px && y == py && zz == pz)) continue;
                const dd val = (dir.x * dd_from(x) + dir.y * dd_from(y)) + dir.z * dd_from(zz);
                if (val.hi > best.hi || (val.hi == best.hi && val.lo > best.lo)) { best = val; bi = j; }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double oh = __shfl_xor(best.hi, off), ol = __shfl_xor(best.lo, off);
                const int oi = __shfl_xor(bi, off);
                if (oh > best.hi || (oh == best.hi && (ol > best.lo || (ol == best.lo && oi < bi)))) { best = dd{oh, ol}; bi = oi; }
            }
            __syncthreads();
            if (lane == 0) { s_hi[wave] = best.hi; s_lo[wave] = best.lo; s_i[wave] = bi; }
            __syncthreads();
            best = dd{s_hi[0], s_lo[0]}; bi = s_i[0];
#pragma unroll
            for (int w = 1; w < 8; ++w) {
                const double oh = s_hi[w], ol = s_lo[w];
                const int oi = s_i[w];
                if (oh > best.hi || (oh == best.hi && (ol > best.lo || (ol == best.lo && oi < bi)))) { best = dd{oh, ol}; bi = oi; }
            }
            // every lane runs the identical state machine on the merged result
            const bool real = sgn_of(best) > 0.0 && bi < NS;
            const dd di = (dir.x * dd_from(px) + dir.y * dd_from(py)) + dir.z * dd_from(pz);
            const dd gap = real ? di - best : di;
            const double tol = rb * (mag_of(dir.x) + mag_of(dir.y) + mag_of(dir.z));
            if (sgn_of(gap) > 0.0 && mag_of(gap) > tol) { state = 1; break; }          // separating direction, certified
            v3<dd> a;
            int ai = -1;
            if (real) { a = v3<dd>{dd_diff(fx[bi], px), dd_diff(fy[bi], py), dd_diff(fz[bi], pz)}; ai = sidx[bi]; }
            else a = v3<dd>{dd_from(-px), dd_from(-py), dd_from(-pz)};
            if (zero3(a)) { state = real ? 2 : 3; break; }     // coincides with a point of smaller index (larger ones are excluded): that one is the vertex
            bool seen = false;
            for (int k = 0; k < S.n; ++k) seen = seen || S.idx[k] == ai;
            if (S.n > 0) {
                // no progress: the support does not get closer to the origin than the closest point already found =>
                // vclose IS the closest point of conv(S_i): the origin is outside, but by less than the certificate can show
                const dd vv = dot(vclose, vclose), va = dot(vclose, a);
                if (seen || sgn_of(vv - va) <= 0.0 || mag_of(vv - va) <= 8.0e-25 * mag_of(vv)) { state = 3; break; }
            }
            // newest vertex first (Ericson's region tests are written around vertex a)
            for (int k = S.n; k > 0; --k) { S.w[k] = S.w[k - 1]; S.idx[k] = S.idx[k - 1]; }
            S.w[0] = a; S.idx[0] = ai; ++S.n;
            if (!closest_simplex(S, vclose)) {
                // origin inside the tetrahedron (in double-double): certify p'_i strictly inside with the four determinants
                const int s0 = -det_sign(S.w[1], S.w[2], S.w[3]), s1 = det_sign(S.w[0], S.w[2], S.w[3]),
                          s2 = -det_sign(S.w[0], S.w[1], S.w[3]), s3 = det_sign(S.w[0], S.w[1], S.w[2]);
                state = (s0 != 0 && s0 == s1 && s1 == s2 && s2 == s3) ? 2 : 3;
                break;
            }
            if (zero3(vclose)) { state = 3; break; }          // the origin lies ON a face / edge of the simplex: exactly degenerate input
            dir = neg(vclose);
        }
        struct { int state; } g = {state};
        if (threadIdx.x == 0) {
            vis[(size_t)v * N + q] = g.state == 1 ? 1 : 0;
            atomicAdd(&counters[4 * v], 1);
            if (g.state != 1 && g.state != 2) atomicAdd(&counters[4 * v + 1], 1);      // not certifiable even in double-double
            atomicAdd(&counters[4 * v + 2], rounds);
        }
        __syncthreads();
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
#define HPR_HEAD_BYTES 2048      // workspace head: counters int[64][4] (exact-fallback queries, unresolved, rounds, -) + maxabs u64[64]
extern "C" size_t pdhip_hpr_ws_bytes(int V, int N) {
    return HPR_HEAD_BYTES + 2 * flipped_bytes(V, N) + 8 * lists_bytes(V, N) + a256((size_t)V * N) + a256((size_t)V * 3 * HPR_KC * sizeof(double)) +
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
    int* counters = reinterpret_cast<int*>(p);
    unsigned long long* maxabs = reinterpret_cast<unsigned long long*>(p + 1024); p += HPR_HEAD_BYTES;
    double* flipped = reinterpret_cast<double*>(p); p += flipped_bytes(V, N);
    double* ss = reinterpret_cast<double*>(p); p += flipped_bytes(V, N);          // second-level support set (outside points)
    int* count = reinterpret_cast<int*>(p); int* list = count + 64; p += lists_bytes(V, N);
    int* count2 = reinterpret_cast<int*>(p); int* list2 = count2 + 64; p += lists_bytes(V, N);
    int* scount = reinterpret_cast<int*>(p); int* sidx = scount + 64; p += lists_bytes(V, N);
    int* ucount = reinterpret_cast<int*>(p); int* ulist = ucount + 64; p += lists_bytes(V, N);      // queries for the exact fallback
    int* useed = reinterpret_cast<int*>(p); p += 4 * lists_bytes(V, N);                               // ... and the simplex each stopped at
    uint8_t* outside = reinterpret_cast<uint8_t*>(p); p += a256((size_t)V * N);
    double* cs = reinterpret_cast<double*>(p); p += a256((size_t)V * 3 * HPR_KC * sizeof(double));
    int* cidx = reinterpret_cast<int*>(p);
    dim3 gf(min(cdiv(N, 256), 256), V);
    PD_HIP(hipMemsetAsync(ws, 0, HPR_HEAD_BYTES, s));
    k_hpr_flip<<<gf, 256, 0, s>>>(points, N, eyes_dev, radius, flipped, maxabs);
    PD_HIP(hipMemsetAsync(count, 0, 64 * sizeof(int), s));
    PD_HIP(hipMemsetAsync(count2, 0, 64 * sizeof(int), s));
    PD_HIP(hipMemsetAsync(scount, 0, 64 * sizeof(int), s));
    PD_HIP(hipMemsetAsync(ucount, 0, 64 * sizeof(int), s));
    k_hpr_collect<<<gf, 256, 0, s>>>(skip, N, count, list, visibility);      // marks the skipped points visible; `list` = the queries
    dim3 gg(cdiv(N, 4 * QPW), V), gg4(cdiv(min(N, HPR_NARROW_BELOW), 4), V), gx(32, V);
    constexpr int KC = HPR_KC;
    if (N > 4 * HPR_KC) {            // the coarse level pays off only when the cloud is much larger than the coarse set
        dim3 ge(cdiv(KC, QPW), V);
        PD_HIP(hipMemsetAsync(outside, 0, (size_t)V * N, s));
        k_hpr_extremes<<<ge, 256, 0, s>>>(flipped, N, KC, cs, cidx, outside);
        k_hpr_gjk<true, QPW, false, 4><<<gg, 256, 0, s>>>(flipped, N, nullptr, nullptr, nullptr, cs, cidx, KC, nullptr, outside, 0, 0x7fffffff, skip, maxabs, nullptr, nullptr, nullptr);
        k_hpr_build<<<V, 1024, 0, s>>>(flipped, N, outside, skip, ss, sidx, scount, count2, list2, visibility);
        k_hpr_gjk<false, QPW, false, 4><<<gg, 256, 0, s>>>(flipped, N, count2, list2, visibility, ss, sidx, N, scount, nullptr, HPR_NARROW_BELOW, 0x7fffffff, nullptr, maxabs, ucount, ulist, useed);
        k_hpr_gjk<false, 4, true, HPR_COOP_WAVES><<<gg4, HPR_COOP_WAVES * 64, 0, s>>>(flipped, N, count2, list2, visibility, ss, sidx, N, scount, nullptr, 0, HPR_NARROW_BELOW, nullptr, maxabs, ucount, ulist, useed);
        k_hpr_exact<<<gx, 512, 0, s>>>(flipped, N, ucount, ulist, useed, visibility, ss, sidx, N, scount, maxabs, counters);
    } else {                         // one level: support set = the whole cloud
        k_hpr_iota<<<gf, 256, 0, s>>>(sidx, N);
        k_hpr_fill_count<<<1, 64, 0, s>>>(scount, V, N);
        k_hpr_gjk<false, QPW, false, 4><<<gg, 256, 0, s>>>(flipped, N, count, list, visibility, flipped, sidx, N, scount, nullptr, HPR_NARROW_BELOW, 0x7fffffff, nullptr, maxabs, ucount, ulist, useed);
        k_hpr_gjk<false, 4, true, HPR_COOP_WAVES><<<gg4, HPR_COOP_WAVES * 64, 0, s>>>(flipped, N, count, list, visibility, flipped, sidx, N, scount, nullptr, 0, HPR_NARROW_BELOW, nullptr, maxabs, ucount, ulist, useed);
        k_hpr_exact<<<gx, 512, 0, s>>>(flipped, N, ucount, ulist, useed, visibility, flipped, sidx, N, scount, maxabs, counters);
    }
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}

// counters of the last pdhip_hidden_point_removal call that used `ws` (synchronises `stream`): out[0] = queries resolved by the
// double-double fallback, out[1] = queries not certifiable even there (reported hidden), out[2] = fallback rounds, summed over views
extern "C" int pdhip_hpr_read_counters(const void* ws, int V, long long* out /*[3]*/, void* stream) {
    PD_REQUIRE(ws && out && V > 0 && V <= 64, "pdhip_hpr_read_counters: bad arguments");
    int h[64 * 4];
    PD_HIP(hipMemcpyAsync(h, ws, sizeof(int) * 4 * V, hipMemcpyDeviceToHost, as_stream(stream)));
    PD_HIP(hipStreamSynchronize(as_stream(stream)));
    out[0] = out[1] = out[2] = 0;
    for (int v = 0; v < V; ++v) { out[0] += h[4 * v]; out[1] += h[4 * v + 1]; out[2] += h[4 * v + 2]; }
    return PDHIP_OK;
}
