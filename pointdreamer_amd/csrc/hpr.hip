// Row P3b: hidden-point removal (Katz, Tal, Basri 2007) on the device -- replaces the Open3D call at
// pointdreamer/ours_utils.py:204-225 (`pcd.hidden_point_removal(eye, radius)`: spherical flip + qhull on the CPU).
//
//   flip:   q = p - eye,  p' = q + 2 (radius - |q|) q / |q|                       (float64, like open3d)
//   visible(i)  <=>  p'_i is a vertex of conv({p'_j} U {0})  <=>  0 is NOT in conv(S_i),
//                    S_i = {p'_j - p'_i : j != i} U {-p'_i}
// Instead of building the hull (qhull: serial, incremental), every point answers its own containment question with a GJK
// iteration.  Every verdict is CERTIFIED on the f64 coordinates: "visible" by a separating direction d with
// d.p'_i - max(max_j d.p'_j, 0) > rounding bound, "hidden" by a tetrahedron of cloud points (or the eye) whose four
// orientation determinants around p'_i pass Shewchuk's static filter.  How a candidate direction / tetrahedron is FOUND is free,
// which is what the levels exploit:
//   level 0  k_hpr_grid + k_hpr_shield: a fixed tetrahedron (eye + the outermost points of three cells of a direction grid around
//            the query's own cell) per query; certified like everything else.  90 % of the hidden points of a cloud end here.
//   level 1  k_hpr_extremes + k_hpr_coarse: the rest against a coarse set of <= 1024 cloud points (the extreme points in
//            Fibonacci directions).  conv(subset) is inside conv(cloud): "enclosed" there is final.  Lane = query; the support
//            scans are a GEMM on the matrix cores (points x directions, coordinates split into two f16 so that one
//            v_mfma_f32_32x32x16_f16 carries the nine hi/lo products) followed by a column maximum -- approximate, but the
//            verdict is certified on the true coordinates.
//   level 2  k_hpr_fine_dist: what level 1 could not enclose, against the points outside the coarse hull (a point strictly
//            inside it is never a support point).  That set is Morton-sorted and cut into 64-point chunks with oriented boxes
//            (the flipped cloud is a thin shell around the eye); one wavefront per query runs the distance form of GJK on a
//            register-resident working set of the 256 points around it (closest-point sub-problem solved across the lanes)
//            and consults the whole set -- box-culled to a chunk or two -- only to certify "visible".
//   level 3  k_hpr_exact<double>, k_hpr_exact<dd>: the few queries whose f64 certificate fails or that stall: the distance
//            iteration with Ericson's full region tests and the duplicate-point rule, first in f64, then in
//            double-double arithmetic (2^-104) with 2^-96 bounds; what even that cannot certify (exact coplanarity /
//            duplicate points) is counted in the workspace counters (pdhip_hpr_read_counters) and reported hidden.
// The result is therefore the vertex set of the exact hull of the f64 flipped points; qhull (open3d, scipy) differs from it only
// for points within its own merge tolerance (~1e-13 * radius) of a facet.  open3d itself is absent (PARITY UNPINNED); the oracle
// drives the same qhull through scipy.  8 views x 30 k points behind the depth-test skip mask: 2.66 ms (round 2's first form:
// wave-cooperative f64 scans) -> 0.38 ms; all points queried: 17 -> 1.3 ms.
// The order of the sorted support set inside a Morton cell comes from atomics; the scans break ties by cloud index and every
// verdict is certified, so the visibility does not depend on it (the fallback counters may differ by a query between runs).
#include "common.h"
using namespace pdhip;

#ifndef GJK_MAX_ROUNDS
#define GJK_MAX_ROUNDS 64
#endif
#ifndef GJK_COARSE_ROUNDS
#define GJK_COARSE_ROUNDS 10     // level 1 gives up early: a wave pays every round for all 64 lanes, level 2 costs one wave per leftover (8: 0.69, 10: 0.60, 16: 0.66 ms)
#endif
#ifndef HPR_KC
#define HPR_KC 1024            // coarse support set size (directions tried; a multiple of 256)
#endif

#ifdef PD_HPR_STATS                                       // (lab builds only: round statistics of the two GJK passes)
__device__ unsigned long long g_hpr_stats[2][16];
__device__ unsigned long long g_hpr_t[8];
__device__ unsigned long long g_hpr_c[8];                        // level 1: core-clock cycles per phase, summed over wave rounds: B operands, scan, re-evaluation + join, step, rounds
__device__ __forceinline__ unsigned long long lab_clock() { unsigned long long t; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }                        // level 2: per-query wall time in 100 MHz ticks: sum, max, sum over coarse-set members, their count             // [pass][waves, wave rounds, queries, query rounds, unfinished, -, -, -, histogram of query rounds / 8]
extern "C" int pdhip_lab_hpr_stats(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hpr_stats), sizeof(g_hpr_stats)) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out + 32, HIP_SYMBOL(g_hpr_t), sizeof(g_hpr_t)) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out + 40, HIP_SYMBOL(g_hpr_c), sizeof(g_hpr_c)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_hpr_stats), z, sizeof(z)) != hipSuccess) return -1;
                 if (hipMemcpyToSymbol(HIP_SYMBOL(g_hpr_t), z, sizeof(g_hpr_t)) != hipSuccess) return -1;
                 if (hipMemcpyToSymbol(HIP_SYMBOL(g_hpr_c), z, sizeof(g_hpr_c)) != hipSuccess) return -1; }
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

// ---- wave-level reductions on the DPP network (no LDS round trip: a __shfl_xor butterfly is six dependent ds_bpermute pairs)
template <int CTRL, int ROWS> __device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, ROWS, 0xf, false); }
template <int CTRL, int ROWS> __device__ __forceinline__ double dpp_f64(double v) {
    return __hiloint2double(dpp_i32<CTRL, ROWS>(__double2hiint(v)), dpp_i32<CTRL, ROWS>(__double2loint(v)));
}
// quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_half_mirror 0x141, row_mirror 0x140: every lane of a 16-lane row holds the row's
// result; row_bcast15 (rows 1, 3) and row_bcast31 (rows 2, 3) then carry it to lane 63.  Lanes a step does not write keep their value.
__device__ __forceinline__ double wave_max_f64(double v) {
    v = fmax(v, dpp_f64<0xB1, 0xf>(v)); v = fmax(v, dpp_f64<0x4E, 0xf>(v)); v = fmax(v, dpp_f64<0x141, 0xf>(v)); v = fmax(v, dpp_f64<0x140, 0xf>(v));
    v = fmax(v, dpp_f64<0x142, 0xa>(v)); v = fmax(v, dpp_f64<0x143, 0xc>(v));
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
__device__ __forceinline__ float wave_max_f32(float v) {
    v = fmaxf(v, __int_as_float(dpp_i32<0xB1, 0xf>(__float_as_int(v)))); v = fmaxf(v, __int_as_float(dpp_i32<0x4E, 0xf>(__float_as_int(v))));
    v = fmaxf(v, __int_as_float(dpp_i32<0x141, 0xf>(__float_as_int(v)))); v = fmaxf(v, __int_as_float(dpp_i32<0x140, 0xf>(__float_as_int(v))));
    v = fmaxf(v, __int_as_float(dpp_i32<0x142, 0xa>(__float_as_int(v)))); v = fmaxf(v, __int_as_float(dpp_i32<0x143, 0xc>(__float_as_int(v))));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ int wave_min_i32(int v) {
    v = min(v, dpp_i32<0xB1, 0xf>(v)); v = min(v, dpp_i32<0x4E, 0xf>(v)); v = min(v, dpp_i32<0x141, 0xf>(v)); v = min(v, dpp_i32<0x140, 0xf>(v));
    v = min(v, dpp_i32<0x142, 0xa>(v)); v = min(v, dpp_i32<0x143, 0xc>(v));
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ double lane_f64(double v, int l) {        // l wave-uniform
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

__device__ __forceinline__ unsigned long long f64_key(double x) {            // order-preserving map to u64
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_f64(unsigned long long k) {
    return __longlong_as_double((long long)((k >> 63) ? (k & 0x7fffffffffffffffull) : ~k));
}
__global__ __launch_bounds__(1024) void k_hpr_flip(const float* __restrict__ pts, int N, const double* __restrict__ eyes, double radius,
                           double* __restrict__ flipped /*[V][3][N]*/, unsigned long long* __restrict__ maxabs /*[V] f64 bits*/,
                           unsigned long long* __restrict__ bbox /*[V][6] order keys of max(-x), max(-y), max(-z), max(x), max(y), max(z)*/,
                           const uint8_t* __restrict__ skip, int* __restrict__ count, int* __restrict__ list, uint8_t* __restrict__ vis,
                           int collect, float4* __restrict__ f32pts /*[V][N] (x, y, z, 0) rounded to f32, or null*/, int vps /*views per shape*/) {
    // open3d PointCloud::HiddenPointRemoval: p' = q + 2 (radius - |q|) q / |q| evaluated as q + ((2 (radius - n)) q) / n
    // Also here: the queries that still need the hull test -- all points, or only those a cheaper test (`skip`) has not already
    // accepted (those are marked visible) -- compacted into `list` with one returning atomic per 256-thread block and step.
    __shared__ int s_wcnt[16], s_base;
    const int v = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    pts += (size_t)(v / vps) * 3 * (size_t)N;            // several shapes in one call: view v looks at the cloud of shape v / vps
    const double ex = eyes[3 * v], ey = eyes[3 * v + 1], ez = eyes[3 * v + 2];
    double m = 0.0, b[6] = {-1.0e300, -1.0e300, -1.0e300, -1.0e300, -1.0e300, -1.0e300};
    for (int i0 = blockIdx.y * blockDim.x; i0 < N; i0 += gridDim.y * blockDim.x) {
        const int i = i0 + threadIdx.x;
        const bool in = i < N;
        if (in) {
            const double qx = (double)pts[3 * i] - ex, qy = (double)pts[3 * i + 1] - ey, qz = (double)pts[3 * i + 2] - ez;
            double n = sqrt(qx * qx + qy * qy + qz * qz);
            if (n == 0.0) n = 0.0001;
            const double k = 2.0 * (radius - n);
            double* f = flipped + (size_t)v * 3 * N;
            const double x = qx + (k * qx) / n, y = qy + (k * qy) / n, z = qz + (k * qz) / n;
            f[i] = x; f[N + i] = y; f[2 * (size_t)N + i] = z;
            if (f32pts != nullptr) f32pts[(size_t)v * N + i] = make_float4((float)x, (float)y, (float)z, 0.0f);
            m = fmax(m, fmax(fabs(x), fmax(fabs(y), fabs(z))));
            b[0] = fmax(b[0], -x); b[1] = fmax(b[1], -y); b[2] = fmax(b[2], -z); b[3] = fmax(b[3], x); b[4] = fmax(b[4], y); b[5] = fmax(b[5], z);
        }
        if (collect) {                                               // (one-level mode; with two levels k_hpr_shield builds the list)
            const bool sk = in && skip != nullptr && skip[(size_t)v * N + i];
            const bool q = in && !sk;
            const unsigned long long bal = __ballot(q);
            if (lane == 0) s_wcnt[wave] = __popcll(bal);
            __syncthreads();
            int mine = 0, tot = 0;
#pragma unroll
            for (int w = 0; w < 16; ++w) { const int c = s_wcnt[w]; mine += w < wave ? c : 0; tot += c; }
            if (threadIdx.x == 0) s_base = atomicAdd(&count[v], tot);
            __syncthreads();
            const int base = s_base + mine;
            if (q) list[(size_t)v * N + base + __popcll(bal & ((1ull << lane) - 1ull))] = i;
            else if (sk) vis[(size_t)v * N + i] = 1;
            __syncthreads();
        }
    }
    m = wave_max_f64(m);
#pragma unroll
    for (int k = 0; k < 6; ++k) b[k] = wave_max_f64(b[k]);
    // one set of atomics per block (they all land on the view's seven words: per wave they serialised to 45 us)
    __shared__ double s_r[16][7];
    if ((threadIdx.x & 63) == 0) {
        s_r[wave][0] = m;
#pragma unroll
        for (int k = 0; k < 6; ++k) s_r[wave][1 + k] = b[k];
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        double r = s_r[0][threadIdx.x];
        for (int w = 1; w < 16; ++w) r = fmax(r, s_r[w][threadIdx.x]);
        if (threadIdx.x == 0) atomicMax(&maxabs[v], (unsigned long long)__double_as_longlong(r));   // non-negative f64: bit order = value order
        else atomicMax(&bbox[6 * v + threadIdx.x - 1], f64_key(r));
    }
}


// ---- level 0, the shield: most of the points the depth test rejects are far behind the outer shell of the flipped cloud, and for
// those a GJK iteration is a waste -- a fixed tetrahedron (eye, A, B, C) with A, B, C the outermost points of three cells of a
// direction grid placed around the query's own cell almost always encloses them.  The grid is the gnomonic projection on the cube
// face the view's bounding box points at (96 x 96 cells over the box's extent); a cell keeps its point of largest radius.  The
// verdict is the same certificate as everywhere (four f64 determinants with the static filter, on cloud points), so nothing is
// assumed about the grid: a failed test just leaves the query to level 1.  90 % of the hidden points of a 30 k cloud end here.
#define HPR_GRID 96              // (hidden points certified on a 30 k cloud: 64 cells, wide triangles only 81 %; 64, tight then wide 88 %; 96: 90 %; 128: 89 %)
struct GridMap { int k, a, b, ok; double u0, us, w0, ws; };
__device__ __forceinline__ GridMap grid_map(const unsigned long long* __restrict__ bbox, int v) {
    double lo[3], hi[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { lo[k] = -key_f64(bbox[6 * v + k]); hi[k] = key_f64(bbox[6 * v + 3 + k]); }
    GridMap g;
    const double cx = fabs(lo[0] + hi[0]), cy = fabs(lo[1] + hi[1]), cz = fabs(lo[2] + hi[2]);
    g.k = (cx >= cy && cx >= cz) ? 0 : (cy >= cz ? 1 : 2);
    g.a = (g.k + 1) % 3; g.b = (g.k + 2) % 3;
    const double lk = g.k == 0 ? lo[0] : (g.k == 1 ? lo[1] : lo[2]), hk = g.k == 0 ? hi[0] : (g.k == 1 ? hi[1] : hi[2]);
    const double la = g.a == 0 ? lo[0] : (g.a == 1 ? lo[1] : lo[2]), ha = g.a == 0 ? hi[0] : (g.a == 1 ? hi[1] : hi[2]);
    const double lb = g.b == 0 ? lo[0] : (g.b == 1 ? lo[1] : lo[2]), hb = g.b == 0 ? hi[0] : (g.b == 1 ? hi[1] : hi[2]);
    g.ok = (lk > 0.0 && hk > 0.0) || (lk < 0.0 && hk < 0.0);      // the whole cloud on one side of the eye along that axis
    const double u1 = la / lk, u2 = la / hk, u3 = ha / lk, u4 = ha / hk, w1 = lb / lk, w2 = lb / hk, w3 = hb / lk, w4 = hb / hk;
    const double umin = fmin(fmin(u1, u2), fmin(u3, u4)), umax = fmax(fmax(u1, u2), fmax(u3, u4));
    const double wmin = fmin(fmin(w1, w2), fmin(w3, w4)), wmax = fmax(fmax(w1, w2), fmax(w3, w4));
    g.u0 = umin; g.us = umax > umin ? HPR_GRID / (umax - umin) : 0.0;
    g.w0 = wmin; g.ws = wmax > wmin ? HPR_GRID / (wmax - wmin) : 0.0;
    return g;
}
__device__ __forceinline__ void grid_cell(const GridMap& g, const double x, const double y, const double z, int& iu, int& iw) {
    const double pk = g.k == 0 ? x : (g.k == 1 ? y : z), pa = g.a == 0 ? x : (g.a == 1 ? y : z), pb = g.b == 0 ? x : (g.b == 1 ? y : z);
    iu = min(HPR_GRID - 1, max(0, (int)((pa / pk - g.u0) * g.us)));
    iw = min(HPR_GRID - 1, max(0, (int)((pb / pk - g.w0) * g.ws)));
}
__global__ void k_hpr_grid(const double* __restrict__ flipped, int N, const unsigned long long* __restrict__ bbox,
                           unsigned long long* __restrict__ grid /*[V][G*G], zeroed: (f32 bits of |p'|^2, index + 1)*/,
                           double* __restrict__ fdir /*[KC][4]: the Fibonacci directions of level 1, computed here on the side*/) {
    const int v = blockIdx.x;
    if (v == 0) {
        const int k = blockIdx.y * blockDim.x + threadIdx.x;
        if (k < HPR_KC) {
            const double zk = 1.0 - (2.0 * k + 1.0) / HPR_KC, rk = sqrt(fmax(0.0, 1.0 - zk * zk)), pk = k * 2.399963229728653;
            fdir[4 * k] = rk * cos(pk); fdir[4 * k + 1] = rk * sin(pk); fdir[4 * k + 2] = zk; fdir[4 * k + 3] = 0.0;
        }
    }
    const GridMap g = grid_map(bbox, v);
    if (!g.ok) return;
    const double* f = flipped + (size_t)v * 3 * N;
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < N; i += gridDim.y * blockDim.x) {
        const double x = f[i], y = f[N + i], z = f[2 * (size_t)N + i];
        int iu, iw;
        grid_cell(g, x, y, z, iu, iw);
        const float r2 = (float)(x * x + y * y + z * z);
        atomicMax(&grid[(size_t)v * HPR_GRID * HPR_GRID + iu * HPR_GRID + iw], ((unsigned long long)__float_as_uint(r2) << 32) | (unsigned int)(i + 1));
    }
}
// the shield test of every not-skipped point + the query list of level 1 (what the shield does not settle); skipped points: visible
__global__ __launch_bounds__(1024) void k_hpr_shield(const double* __restrict__ flipped, int N, const unsigned long long* __restrict__ bbox,
                                                     const unsigned long long* __restrict__ grid, const uint8_t* __restrict__ skip,
                                                     int* __restrict__ count, int* __restrict__ list, uint8_t* __restrict__ vis,
                                                     int* __restrict__ counters) {
    __shared__ int s_wcnt[16], s_base;
    const int v = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const GridMap g = grid_map(bbox, v);
    const double* f = flipped + (size_t)v * 3 * N;
    const unsigned long long* gv = grid + (size_t)v * HPR_GRID * HPR_GRID;
    for (int i0 = blockIdx.y * blockDim.x; i0 < N; i0 += gridDim.y * blockDim.x) {
        const int i = i0 + threadIdx.x;
        const bool in = i < N;
        const bool sk = in && skip != nullptr && skip[(size_t)v * N + i];
        bool q = in && !sk;
        if (q && g.ok) {
            const d3 p = {f[i], f[N + i], f[2 * (size_t)N + i]};
            int iu, iw;
            grid_cell(g, p.x, p.y, p.z, iu, iw);
            // placements of the three cells around the query's: tight triangles first (the flatter the triangle, the closer to the shell it
            // still covers), then wide ones (the triangle of ANY points of those three cells contains the whole cell)
            const int pat[8][3][2] = {{{-1, -1}, {1, -1}, {0, 2}}, {{-1, 1}, {1, 1}, {0, -2}}, {{-1, -1}, {-1, 1}, {2, 0}}, {{1, -1}, {1, 1}, {-2, 0}},
                                      {{-2, -2}, {2, -2}, {0, 3}}, {{-2, 2}, {2, 2}, {0, -3}}, {{-2, -2}, {-2, 2}, {3, 0}}, {{2, -2}, {2, 2}, {-3, 0}}};
#pragma unroll
            for (int t = 0; t < 8 && q; ++t) {
                int id[3];
                bool have = true;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int x = iu + pat[t][c][0], y = iw + pat[t][c][1];
                    id[c] = -1;
                    if (x >= 0 && x < HPR_GRID && y >= 0 && y < HPR_GRID) id[c] = (int)(unsigned int)(gv[x * HPR_GRID + y] & 0xffffffffull) - 1;
                    have = have && id[c] >= 0 && id[c] != i;
                }
                if (!have) continue;
                const d3 a = d3{f[id[0]], f[N + id[0]], f[2 * (size_t)N + id[0]]} - p, b = d3{f[id[1]], f[N + id[1]], f[2 * (size_t)N + id[1]]} - p,
                         c = d3{f[id[2]], f[N + id[2]], f[2 * (size_t)N + id[2]]} - p, d = neg(p);
                const int s0 = -det_sign(b, c, d), s1 = det_sign(a, c, d), s2 = -det_sign(a, b, d), s3 = det_sign(a, b, c);
                if (s0 != 0 && s0 == s1 && s1 == s2 && s2 == s3) q = false;           // certified strictly inside: hidden
            }
        }
        const unsigned long long bal = __ballot(q);
        if (lane == 0) s_wcnt[wave] = __popcll(bal);
        __syncthreads();
        int mine = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { const int c = s_wcnt[w]; mine += w < wave ? c : 0; tot += c; }
        if (threadIdx.x == 0) s_base = atomicAdd(&count[v], tot);
        __syncthreads();
        const int base = s_base + mine;
        if (q) list[(size_t)v * N + base + __popcll(bal & ((1ull << lane) - 1ull))] = i;
        else if (in) vis[(size_t)v * N + i] = sk ? 1 : 0;           // skipped: visible; certified behind the shield: hidden
        __syncthreads();
    }
    (void)counters;
}

// what a query lane does with the support point of its round: myv = the largest support value over the real points (real_pt: one
// was found; sp / sp_idx = that point and its cloud index).  The support of S_i is that point or the eye (value 0).
// "The support does not pass the origin"  <=>  p'_i is the strict maximiser of dir over the cloud and the eye: position form, every
// value evaluated as fma(dz, z, fma(dy, y, dx x)), so that the rounding bound rb covers both sides.
__device__ __forceinline__ void gjk_round(Gjk<double>& g, const d3 pi, const double myv, const bool real_pt, const d3 sp, const int sp_idx,
                                          const double rb) {
    const bool real = real_pt && myv > 0.0;
    const double di = fma(g.dir.z, pi.z, fma(g.dir.y, pi.y, g.dir.x * pi.x));
    const double gap = di - (real ? myv : 0.0);
    const double tol = rb * (fabs(g.dir.x) + fabs(g.dir.y) + fabs(g.dir.z));
    if (gap > tol) g.state = 1;                                       // certified visible
    else if (gap > 0.0 && g.dim >= 1) g.state = 3;                    // visible in f64, but inside the rounding bound
    else {
        // a support point that is already a vertex of the simplex: the origin lies on that face within rounding (exactly, the
        // support of a face's own normal would not pass the origin) -- the boolean iteration would only cycle from here
        const int ai = real ? sp_idx : -1;
        if ((g.dim >= 1 && ai == g.ic) || (g.dim >= 2 && ai == g.ib) || (g.dim >= 3 && ai == g.id)) g.state = 3;
        else gjk_step(g, real ? sp - pi : neg(pi), ai);
    }
}

// ---- level 1: is the query enclosed by the hull of the coarse set (<= KC points of the cloud, no repeats)?  One LANE per query.
// Only "enclosed" (certified by the four f64 determinants on the true coordinates) is a verdict of this level; everything else
// -- also a query still running at the round cap, also a member of the coarse set itself -- goes on to level 2.  That makes the
// support SCAN free to be approximate, and it is a small GEMM: values[point][query] = P[point][xyz] . D[xyz][query].  It runs on
// the matrix cores in f32 (v_mfma_f32_32x32x2_f32, K = x, y | z, 0): A = 32 coarse points, B = the directions of 32 of the
// wave's 64 queries, so that a lane receives 16 points' values for ONE query (its own column) and keeps a running maximum --
// no cross-lane work except joining the two half-waves at the end.  The VALU is left with the compare / select (3 per pair
// instead of 7 with the products) and the GJK step; the chosen support point is then re-evaluated in f64.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float tile_max(const f32x16 a) {          // (v_max3_f32: eight instructions)
    const float m0 = fmaxf(fmaxf(a[0], a[1]), a[2]), m1 = fmaxf(fmaxf(a[3], a[4]), a[5]), m2 = fmaxf(fmaxf(a[6], a[7]), a[8]),
                m3 = fmaxf(fmaxf(a[9], a[10]), a[11]), m4 = fmaxf(fmaxf(a[12], a[13]), a[14]);
    return fmaxf(fmaxf(fmaxf(m0, m1), m2), fmaxf(fmaxf(m3, m4), a[15]));
}
template <int COLS>          // query columns of 32 per wavefront: 2 = a query in every lane; 1 = queries in lanes 0-31 only, twice the waves (measured: 186 vs 158 us)
__global__ __launch_bounds__(256) void k_hpr_coarse(const double* __restrict__ flipped, int N, const int* __restrict__ count,
                                                    const int* __restrict__ list, const float4* __restrict__ csf /*[V][KC]*/,
                                                    const double* __restrict__ csd /*[V][KC][4]*/, const int* __restrict__ cidx /*[V][KC]*/,
                                                    const int* __restrict__ kcount, uint8_t* __restrict__ outside, uint8_t* __restrict__ vis,
                                                    const unsigned long long* __restrict__ maxabs, double* __restrict__ qdir /*[V][N][3]*/) {
    const int v = blockIdx.x;
    const int nq = count[v];
    const int lane = threadIdx.x & 63, l31 = lane & 31;
    const bool hi = lane >= 32;
    const int qi = COLS == 2 ? blockIdx.y * 256 + threadIdx.x : blockIdx.y * 128 + (threadIdx.x >> 6) * 32 + l31;
    if (blockIdx.y * (COLS == 2 ? 256 : 128) >= nq) return;           // (the whole block: before the barrier below)
    const int KS = min(kcount[v], HPR_KC), KT = (KS + 31) >> 5;       // (the set is padded with the eye (0, 0, 0) to whole tiles)
    // the view's coarse set lives in LDS for the block's lifetime (f32 records for the scans, f64 records + cloud indices for the
    // GJK step): every round ends with two dependent look-ups into it, which from L2 were a third of the round's latency
    // (f32 coordinates as three arrays with 33 slots per 32-point tile: as [x y z -] records at a 512-byte tile stride, the lanes
    // re-evaluating rows of DIFFERENT tiles all hit the same four banks -- a 64-way conflict on each of the 32 reads of a round,
    // 60 % of this kernel)
    __shared__ float s_cx[HPR_KC + HPR_KC / 32], s_cy[HPR_KC + HPR_KC / 32], s_cz[HPR_KC + HPR_KC / 32];
#define CSLOT(j) ((j) + ((j) >> 5))
    // the scan's matrix operands: every coordinate split into two f16 (hi + lo: 22 bits), the nine products x_h d_h, x_h d_l, x_l d_h
    // (x, y, z) laid along K = 16 of ONE v_mfma_f32_32x32x16_f16 (32 cycles) instead of two f32 MFMAs of 64 cycles each:
    //   A (lanes 0-31: k 0-7 | lanes 32-63: k 8-15) = [xh yh zh xh yh zh xl yl | zl 0 ...],  B = [dxh dyh dzh dxl dyl dzl dxh dyh | dzh 0 ...]
    __shared__ f16x8 s_alo[HPR_KC], s_ahi[HPR_KC];
    __shared__ double s_csd[HPR_KC][3];
    __shared__ int s_cidx[HPR_KC];
    {
        const int KL = min(HPR_KC, ((KT + 2) & ~1) * 32);          // (whole tiles, an even number of them, one more for the prefetch)
        for (int i = threadIdx.x; i < KL; i += 256) {
            const float4 rec = csf[(size_t)v * HPR_KC + i];
            s_cx[CSLOT(i)] = rec.x; s_cy[CSLOT(i)] = rec.y; s_cz[CSLOT(i)] = rec.z;
            const _Float16 xh = (_Float16)rec.x, yh = (_Float16)rec.y, zh = (_Float16)rec.z;
            const _Float16 xl = (_Float16)(rec.x - (float)xh), yl = (_Float16)(rec.y - (float)yh), zl = (_Float16)(rec.z - (float)zh);
            s_alo[i] = f16x8{xh, yh, zh, xh, yh, zh, xl, yl};
            s_ahi[i] = f16x8{zl, 0, 0, 0, 0, 0, 0, 0};
            const double* cd = csd + ((size_t)v * HPR_KC + i) * 4;
            s_csd[i][0] = cd[0]; s_csd[i][1] = cd[1]; s_csd[i][2] = cd[2];
            s_cidx[i] = cidx[(size_t)v * HPR_KC + i];
        }
        __syncthreads();
    }
    if ((COLS == 2 ? (qi & ~63) : (qi & ~31)) >= nq) return;
    const double* qf = flipped + (size_t)v * 3 * N;
    const double rb = __longlong_as_double((long long)maxabs[v]) * (8.0 * 1.1102230246251565e-16);
    const bool owner = qi < nq && (COLS == 2 || !hi);
    const int q = owner ? list[(size_t)v * N + qi] : 0;
    d3 pi = {0, 0, 0};
    Gjk<double> g;
    g.sa = g.sb = g.sc = g.sd = d3{0, 0, 0}; g.dir = d3{0, 0, 1};
    g.ia = g.ib = g.ic = g.id = -1;
    g.dim = 0;
    g.state = owner ? 0 : 2;
    if (owner) { pi = {qf[q], qf[N + q], qf[2 * (size_t)N + q]}; g.dir = pi; }
#ifdef PD_HPR_STATS
    int my_rounds = 0, wave_rounds = 0;
#endif
    for (int round = 0; round < GJK_COARSE_ROUNDS; ++round) {
        if (__ballot(g.state == 0) == 0ull) break;
#ifdef PD_HPR_STATS
        ++wave_rounds; if (g.state == 0) ++my_rounds;
#endif
#ifdef PD_HPR_STATS
        const unsigned long long c0 = lab_clock();
#endif
        // directions only matter up to scale: bring them into f32 range by their own magnitude (a power of two)
        const double mag = fmax(fabs(g.dir.x), fmax(fabs(g.dir.y), fabs(g.dir.z)));
        const double sc = mag > 0.0 ? __longlong_as_double((long long)((0x7feull - ((unsigned long long)__double_as_longlong(mag) >> 52)) << 52)) : 1.0;
        const float dx = (float)(g.dir.x * sc), dy = (float)(g.dir.y * sc), dz = (float)(g.dir.z * sc);
        // B operands (lane l: B[k = l >> 5][query column l & 31]) for the wave's two columns of 32 queries
        const float x0 = __shfl(dx, l31), y0 = __shfl(dy, l31), z0 = __shfl(dz, l31);
        const float x1 = COLS == 2 ? __shfl(dx, 32 + l31) : 0.0f, y1 = COLS == 2 ? __shfl(dy, 32 + l31) : 0.0f, z1 = COLS == 2 ? __shfl(dz, 32 + l31) : 0.0f;
        f16x8 bq0, bq1;                                            // (directions are scaled to [1, 2): well inside the f16 range)
        {
            const _Float16 xh = (_Float16)x0, yh = (_Float16)y0, zh = (_Float16)z0;
            const _Float16 xl = (_Float16)(x0 - (float)xh), yl = (_Float16)(y0 - (float)yh), zl = (_Float16)(z0 - (float)zh);
            bq0 = hi ? f16x8{zh, 0, 0, 0, 0, 0, 0, 0} : f16x8{xh, yh, zh, xl, yl, zl, xh, yh};
            const _Float16 uh = (_Float16)x1, vh = (_Float16)y1, wh = (_Float16)z1;
            const _Float16 ul = (_Float16)(x1 - (float)uh), vl = (_Float16)(y1 - (float)vh), wl = (_Float16)(z1 - (float)wh);
            bq1 = hi ? f16x8{wh, 0, 0, 0, 0, 0, 0, 0} : f16x8{uh, vh, wh, ul, vl, wl, uh, vh};
        }
#ifdef PD_HPR_STATS
        const unsigned long long c1 = lab_clock();
#endif
        float best0 = -3.0e38f, best1 = -3.0e38f;
        int code0 = -1, code1 = -1;                               // tile * 16 + accumulator entry
        // two tiles in flight: the matrix cores work on one while the VALU reduces the other (KT2 even: the pad tiles are the eye)
        const f32x16 zero = {0};
#define HPR_TILE_MFMA(ACC0, ACC1, PT)                                                                              \
        {                                                                                                          \
            ACC0 = __builtin_amdgcn_mfma_f32_32x32x16_f16((PT), bq0, zero, 0, 0, 0);                               \
            if (COLS == 2) ACC1 = __builtin_amdgcn_mfma_f32_32x32x16_f16((PT), bq1, zero, 0, 0, 0);                \
        }
#define HPR_TILE_REDUCE(ACC0, ACC1, T)                                                                             \
        {   /* per tile only WHICH tile holds the lane's maximum; the position inside it is found once, after the scan */ \
            const float m0 = tile_max(ACC0);                                                                       \
            if (m0 > best0) { best0 = m0; code0 = (T); }                                                           \
            if (COLS == 2) { const float m1 = tile_max(ACC1); if (m1 > best1) { best1 = m1; code1 = (T); } }       \
        }
        const int KT2 = (KT + 1) & ~1, TMAX = HPR_KC / 32 - 1;
        f32x16 A0, A1 = zero, B0, B1 = zero;
        const f16x8* s_aop = hi ? s_ahi : s_alo;                      // (lane l: A[point l & 31][k = 8 (l >> 5) ..])
#define HPR_TILE_LOAD(J) s_aop[J]
        f16x8 pa = HPR_TILE_LOAD(l31), pb = HPR_TILE_LOAD(32 + l31);
        HPR_TILE_MFMA(A0, A1, pa)
        for (int t = 0; t < KT2; t += 2) {
            pa = HPR_TILE_LOAD(min(t + 2, TMAX) * 32 + l31);
            HPR_TILE_MFMA(B0, B1, pb)
            HPR_TILE_REDUCE(A0, A1, t)
            pb = HPR_TILE_LOAD(min(t + 3, TMAX) * 32 + l31);
            HPR_TILE_MFMA(A0, A1, pa)
            HPR_TILE_REDUCE(B0, B1, t + 1)
        }
#undef HPR_TILE_MFMA
#undef HPR_TILE_REDUCE
#undef HPR_TILE_LOAD
#ifdef PD_HPR_STATS
        const unsigned long long c2 = lab_clock() + (best0 == 1.2345f ? 1 : 0) + (best1 == 1.2345f ? 1 : 0);
#endif
        // accumulator entry i of lane l is point row 8 (i / 4) + 4 (l >> 5) + (i % 4) of the tile: the lane re-evaluates its 16 rows of
        // the tile its maximum came from (in f32: close to the matrix core's split-f16 values, and this level only proposes), for both columns
        int j0 = -1, j1 = -1;
        {
            float r0 = -3.0e38f, r1 = -3.0e38f;
            const int base0 = max(code0, 0) * 32 + (hi ? 4 : 0), base1 = max(code1, 0) * 32 + (hi ? 4 : 0);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = 8 * (i / 4) + (i % 4);
                const int sl0 = CSLOT(base0 + row);
                const float w0 = fmaf(s_cz[sl0], z0, fmaf(s_cy[sl0], y0, s_cx[sl0] * x0));
                if (w0 > r0) { r0 = w0; j0 = base0 + row; }
                if (COLS == 2) {
                    const int sl1 = CSLOT(base1 + row);
                    const float w1 = fmaf(s_cz[sl1], z1, fmaf(s_cy[sl1], y1, s_cx[sl1] * x1));
                    if (w1 > r1) { r1 = w1; j1 = base1 + row; }
                }
            }
        }
        // join the two half-waves
        const float ob0 = __shfl_xor(best0, 32);
        const int oj0 = __shfl_xor(j0, 32);
        if (ob0 > best0) j0 = oj0;
        if (COLS == 2) {
            const float ob1 = __shfl_xor(best1, 32);
            const int oj1 = __shfl_xor(j1, 32);
            if (ob1 > best1) j1 = oj1;
        }
        int bi = (COLS == 2 && hi) ? j1 : j0;                                    // query lane l is column l & 31 of query tile l >> 5
        if (bi >= KS) bi = -1;                                    // a pad entry: the eye
#ifdef PD_HPR_STATS
        const unsigned long long c3 = lab_clock() + (bi == 123456789 ? 1 : 0);
#endif
        if (g.state == 0) {
            d3 sp = {0, 0, 0};
            int si = -1;
            double myv = 0.0;
            if (bi >= 0) {
                si = s_cidx[bi];
                sp = {s_csd[bi][0], s_csd[bi][1], s_csd[bi][2]};
                myv = fma(g.dir.z, sp.z, fma(g.dir.y, sp.y, g.dir.x * sp.x));
            }
            if (si == q) g.state = 3;                             // the query is itself a member of the coarse set: level 2 decides
            else gjk_round(g, pi, myv, bi >= 0, sp, si, rb);
        }
#ifdef PD_HPR_STATS
        { const unsigned long long c4 = lab_clock() + (g.state == 77 ? 1 : 0);
          if ((threadIdx.x & 63) == 0) { atomicAdd(&g_hpr_c[0], c1 - c0); atomicAdd(&g_hpr_c[1], c2 - c1); atomicAdd(&g_hpr_c[2], c3 - c2); atomicAdd(&g_hpr_c[3], c4 - c3); atomicAdd(&g_hpr_c[4], 1ull); } }
#endif
    }
#ifdef PD_HPR_STATS
    if ((threadIdx.x & 63) == 0) { atomicAdd(&g_hpr_stats[0][0], 1ull); atomicAdd(&g_hpr_stats[0][1], (unsigned long long)wave_rounds); }
    if (owner) { atomicAdd(&g_hpr_stats[0][2], 1ull); atomicAdd(&g_hpr_stats[0][3], (unsigned long long)my_rounds);
                 if (g.state == 0) atomicAdd(&g_hpr_stats[0][4], 1ull);
                 atomicAdd(&g_hpr_stats[0][8 + min(my_rounds, 63) / 8], 1ull); }
#endif
    if (owner) outside[(size_t)v * N + q] = g.state != 2;        // 0: CERTIFIED enclosed by the coarse hull: hidden, and never a support point
    if (owner && g.state == 2) vis[(size_t)v * N + q] = 0;       // (the verdict array is written in full: callers need not clear it)
    if (owner && g.state != 2) {                                  // level 2 starts where this level stopped looking
        double* qd = qdir + ((size_t)v * N + q) * 3;
        qd[0] = g.dir.x; qd[1] = g.dir.y; qd[2] = g.dir.z;
    }
}

// ---- level 2: the queries the coarse hull cannot enclose, against the points outside the coarse hull, which are sorted by a
// 3-D Morton cell of their position and cut into chunks of 64 with an oriented bounding box each (axes: the chunk's mean
// direction from the eye and two tangents -- the flipped cloud is a thin shell around the eye, its patches are flat in that
// frame).  A support scan in direction d only has to look into the chunks whose box bound  sum_k max((d.e_k) lo_k, (d.e_k) hi_k)
// reaches a value that some point of S_i certainly attains: the largest support value of the query's current simplex vertices,
// or, before there is a simplex, of its neighbour in the sorted order.  That is a handful of chunks out of hundreds.  The bounds
// are evaluated in f32 with a slack that covers their rounding (and the rounding of the stored axes); the support values
// themselves in f64.  Equal support values: the smaller cloud index, so the result does not depend on the order inside a cell.
#define HPR_BOX_FLOATS 16        // e0, e1, e2, lo, hi (3 each), pad
#define HPR_BOUND_SLACK 1.0e-5   // x |d|_1 max|coordinate|: f32 evaluation of the bound (<= 3e-6) with margin
struct Support { double val, x, y, z; int pos, idx;           // wave-uniform; pos < 0: no point of S_i reaches the threshold
                 double lx, ly, lz; int lidx; };            // lane-private: the best point THIS lane came across (lidx < 0: none)
// DUPX: also exclude the points that coincide with the query and have a larger cloud index (the distance iteration's rule)
template <bool DUPX, int BATCH>        // BATCH candidate chunks per trip (their loads are in flight together)
__device__ __forceinline__ Support support_scan(const double* __restrict__ fx, const double* __restrict__ fy, const double* __restrict__ fz,
                                                const int* __restrict__ sidx, const int NS, const float4* __restrict__ boxes,
                                                const double dx, const double dy, const double dz, const double th, const double slack,
                                                const int qk, const double px, const double py, const double pz,
                                                int* s_cand /*[256], this wave's*/, const int lane, unsigned long long* n_cand,
                                                const int skip_lo = 0, const int skip_hi = 0) {
    const int NCH = (NS + 63) >> 6;
    // the direction in f32, scaled into range by a power of two (the bound is homogeneous in d; th and slack are scaled alike)
    const double mag = fmax(fabs(dx), fmax(fabs(dy), fabs(dz)));
    const double sc = mag > 0.0 ? __longlong_as_double((long long)((0x3ffull + 0x3ffull - (((unsigned long long)__double_as_longlong(mag) >> 52) & 0x7ffull)) << 52)) : 1.0;
    const float dxf = (float)(dx * sc), dyf = (float)(dy * sc), dzf = (float)(dz * sc);
    double ths = th * sc;
    const double slacks = slack * sc;
    double best = -1.0e300, bx = 0.0, by = 0.0, bz = 0.0;
    int bidx = 0x7fffffff, bpos = -1;
    for (int g0 = 0; g0 < NCH; g0 += 256) {
        float bound[4];
        bool todo[4];
        float bm = -3.0e38f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int cc = g0 + u * 64 + lane;
            todo[u] = false; bound[u] = -3.0e38f;
            if (cc < NCH && !(cc >= skip_lo && cc < skip_hi)) {            // (skip: chunks the caller already holds)
                const float4* b = boxes + (size_t)cc * (HPR_BOX_FLOATS / 4);
                const float4 b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3];          // e0.xyz e1.x | e1.yz e2.xy | e2.z lo.xyz | hi.xyz -
                const float p0 = fmaf(dzf, b0.z, fmaf(dyf, b0.y, dxf * b0.x)), p1 = fmaf(dzf, b1.y, fmaf(dyf, b1.x, dxf * b0.w)),
                            p2 = fmaf(dzf, b2.x, fmaf(dyf, b1.w, dxf * b1.z));
                bound[u] = (fmaxf(p0 * b2.y, p0 * b3.x) + fmaxf(p1 * b2.z, p1 * b3.y)) + fmaxf(p2 * b2.w, p2 * b3.z);
                todo[u] = (double)bound[u] >= ths;
                bm = fmaxf(bm, bound[u]);
            }
        }
        // two tiers: first only the chunks whose bound is within 0.4 % of the largest (the flipped cloud is a shell: the support
        // point is almost always there), then whatever can still beat the value found -- a weak threshold (a direction far from the
        // working set's side of the cloud) would otherwise let half the chunks through
        bm = wave_max_f32(bm);
        double tier = bm > 0.0f ? fmax(ths, (double)bm * 0.996) : ths;
        for (int pass = 0; pass < 2; ++pass) {
            int nc = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool c = todo[u] && (double)bound[u] >= tier;
                const unsigned long long bal = __ballot(c);
                if (c) { s_cand[nc + __popcll(bal & ((1ull << lane) - 1ull))] = g0 + u * 64 + lane; todo[u] = false; }
                nc += __popcll(bal);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (n_cand) *n_cand += nc;
            for (int b0 = 0; b0 < nc; b0 += BATCH) {
                int jj[BATCH];
#pragma unroll
                for (int u = 0; u < BATCH; ++u) jj[u] = b0 + u < nc ? (s_cand[b0 + u] << 6) + lane : NS;
                double x[BATCH], y[BATCH], z[BATCH];
                int jo[BATCH];
#pragma unroll
                for (int u = 0; u < BATCH; ++u) {
                    const int j = min(jj[u], NS - 1);
                    x[u] = fx[j]; y[u] = fy[j]; z[u] = fz[j]; jo[u] = sidx[j];
                }
#pragma unroll
                for (int u = 0; u < BATCH; ++u) {
                    const double val = fma(dz, z[u], fma(dy, y[u], dx * x[u]));
                    bool ok = jj[u] < NS && jo[u] != qk;
                    if (DUPX) ok = ok && !(jo[u] > qk && x[u] == px && y[u] == py && z[u] == pz);
                    if (ok && (val > best || (val == best && jo[u] < bidx))) { best = val; bidx = jo[u]; bpos = jj[u]; bx = x[u]; by = y[u]; bz = z[u]; }
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (tier <= ths) break;                                // the first tier was everything
            const double wb = wave_max_f64(best);                  // what is found bounds what is still worth a look (ties included)
            if (wb > -1.0e299) ths = fmax(ths, wb * sc - slacks);
            tier = ths;
        }
    }
    Support r;
    r.lx = bx; r.ly = by; r.lz = bz; r.lidx = bpos >= 0 ? bidx : -1;
    r.val = wave_max_f64(best);
    r.idx = wave_min_i32(best == r.val ? bidx : 0x7fffffff);
    const unsigned long long who = __ballot(best == r.val && bidx == r.idx && bpos >= 0);
    r.pos = -1; r.x = r.y = r.z = 0.0;
    if (who != 0ull) {
        const int src = __builtin_amdgcn_readfirstlane(__builtin_ctzll(who));
        r.pos = __builtin_amdgcn_readlane(bpos, src);
        r.x = lane_f64(bx, src); r.y = lane_f64(by, src); r.z = lane_f64(bz, src);
    }
    return r;
}

// ---- level 2: one wavefront per query.  The kernel time is the longest query's chain of dependent rounds, so a round must not
// wait on memory: the wave keeps a WORKING SET in registers -- the 256 support points around the query in the sorted order, four
// per lane, plus every point a global scan has returned -- and the GJK iteration takes its support points from that set.  Any
// point of S_i that passes the origin is a legitimate next vertex, and "enclosed" is certified on real points as always; only
// "visible" and "no progress" need the whole support set: then one box-culled global scan looks for points with
// d.p' >= the working set's best (next to a hull vertex that is one or two chunks); what it returns joins the working set.
#define HPR_LOCAL 4
#ifndef HPR_SCAN_BATCH
#define HPR_SCAN_BATCH 2
#endif
// The iteration is GJK proper, the DISTANCE form (the search direction is minus the closest point of the simplex to the
// origin; the distance decreases every round, so it cannot cycle the way the boolean form does on nearly degenerate input).  The
// closest-point sub-problem is solved across the lanes: the closest point of conv{a, w0, w1, w2} lies on a face that contains
// the newest vertex a, so the eight candidate faces {a} U X, X a subset of the kept vertices, are one lane each -- project the
// origin on the face's affine hull through its 3x3 (padded) Gram system by Cramer's rule, keep the candidates whose barycentric
// coordinates are all positive (they are points of the simplex, and the true closest point is one of them), take the nearest.
// No branches on the simplex size.
// (The kernel is bound by VALU issue -- every lane executes a round's ~900 instructions for the one query.  Variants with one query
// per 16- or 32-lane group, i.e. 4x / 2x fewer instructions per query, ended at the same 105-110 us for the pipeline's 6 500
// queries: with 1 600 waves the chip runs out of waves to hide a round's dependent chain behind.)
__device__ __forceinline__ d3 sel3(bool c, d3 a, d3 b) { return d3{c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z}; }      // (by component: a struct select goes through memory)
__device__ __forceinline__ bool closest_with_newest(const d3 a, d3& W0, d3& W1, d3& W2, int& I0, int& I1, int& I2, int& n, const int ai,
                                                    d3& v, const int lane) {
    const d3 e0 = W0 - a, e1 = W1 - a, e2 = W2 - a;
    const int mask = lane & 7;
    const bool a0 = (mask & 1) && n >= 1, a1 = (mask & 2) && n >= 2, a2 = (mask & 4) && n >= 3;
    const bool mine = lane < 8 && (!(mask & 1) || n >= 1) && (!(mask & 2) || n >= 2) && (!(mask & 4) || n >= 3);
    const double G00 = a0 ? dot(e0, e0) : 1.0, G11 = a1 ? dot(e1, e1) : 1.0, G22 = a2 ? dot(e2, e2) : 1.0;
    const double G01 = (a0 && a1) ? dot(e0, e1) : 0.0, G02 = (a0 && a2) ? dot(e0, e2) : 0.0, G12 = (a1 && a2) ? dot(e1, e2) : 0.0;
    const double r0 = a0 ? -dot(a, e0) : 0.0, r1 = a1 ? -dot(a, e1) : 0.0, r2 = a2 ? -dot(a, e2) : 0.0;
    const double c00 = G11 * G22 - G12 * G12, c01 = G01 * G22 - G12 * G02, c02 = G01 * G12 - G11 * G02;
    const double det = (G00 * c00 - G01 * c01) + G02 * c02;
    const double n0 = (r0 * c00 - G01 * (r1 * G22 - G12 * r2)) + G02 * (r1 * G12 - G11 * r2);
    const double n1 = (G00 * (r1 * G22 - r2 * G12) - r0 * c01) + G02 * (G01 * r2 - r1 * G02);
    const double n2 = (G00 * (G11 * r2 - G12 * r1) - G01 * (G01 * r2 - r1 * G02)) + r0 * c02;
    const bool ok = mine && det > 0.0 && (!a0 || n0 > 0.0) && (!a1 || n1 > 0.0) && (!a2 || n2 > 0.0) && (n0 + n1) + n2 < det;
    const double rdet = 1.0 / det, l0 = n0 * rdet, l1 = n1 * rdet, l2 = n2 * rdet;       // (the search may be a rounding off; the verdicts are certified elsewhere)
    const d3 c = {(a.x + l0 * e0.x) + (l1 * e1.x + l2 * e2.x), (a.y + l0 * e0.y) + (l1 * e1.y + l2 * e2.y), (a.z + l0 * e0.z) + (l1 * e1.z + l2 * e2.z)};
    const double d2 = ok ? dot(c, c) : 1.0e300;
    const double best = -wave_max_f64(-d2);
    const int win = __builtin_amdgcn_readfirstlane(__builtin_ctzll(__ballot(d2 == best)));      // lane 0 ({a} alone) is always a candidate
    v = d3{lane_f64(c.x, win), lane_f64(c.y, win), lane_f64(c.z, win)};
    const int wm = win & 7;
    if (wm == 7) return false;                                   // the origin is inside the tetrahedron (simplex left whole)
    // new simplex: the newest vertex first, then the kept vertices the winning face uses
    const bool b0 = wm & 1, b1 = wm & 2, b2 = wm & 4;
    const d3 first = sel3(b0, W0, sel3(b1, W1, W2)), second = sel3(b0 && b1, W1, W2);       // (second is read only when two are kept)
    const int ifirst = b0 ? I0 : (b1 ? I1 : I2), isecond = (b0 && b1) ? I1 : I2;
    W0 = a; I0 = ai; W1 = first; I1 = ifirst; W2 = second; I2 = isecond;
    n = 1 + (b0 ? 1 : 0) + (b1 ? 1 : 0) + (b2 ? 1 : 0);
    return true;
}
// (lab: waves per SIMD the allocation must admit.  128 VGPRs = 4 waves; 5 waves = 96 VGPRs + 148 B of scratch: 122 instead of 105 us,
// 6 waves = 80 VGPRs + 280 B: 286 us)
#ifndef HPR_FINE_WPE
#define HPR_FINE_WPE 1
#endif
__global__ __launch_bounds__(256, HPR_FINE_WPE) void k_hpr_fine_dist(const double* __restrict__ flipped, int N, const int* __restrict__ count,
                                                       const int* __restrict__ list, uint8_t* __restrict__ vis, const double* __restrict__ ss,
                                                       const int* __restrict__ sidx_all, const int* __restrict__ scount,
                                                       const float4* __restrict__ boxes_all, const int* __restrict__ pos_of,
                                                       const int* __restrict__ mdir /*may be null*/, const double* __restrict__ fdir,
                                                       const double* __restrict__ qdir /*may be null*/,
                                                       const unsigned long long* __restrict__ maxabs, int* __restrict__ unc_count,
                                                       int* __restrict__ unc_list, int* __restrict__ unc_seed) {
    __shared__ int s_cand[4][256];
    const int v = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nq = count[v];
    const double* qf = flipped + (size_t)v * 3 * N;
    const double* fx = ss + (size_t)v * 3 * N;
    const double* fy = fx + N;
    const double* fz = fy + N;
    const int* sidx = sidx_all + (size_t)v * N;
    const int NS = scount[v];
    const float4* boxes = boxes_all + (size_t)v * ((N + 63) >> 6) * (HPR_BOX_FLOATS / 4);
    const double ma = __longlong_as_double((long long)maxabs[v]);
    const double rb = ma * (8.0 * 1.1102230246251565e-16);
    // (a fixed grid that strides over the list: a grid sized for N queries would be mostly empty blocks, and dispatching those costs
    // more than the work -- 58 k of 60 k in the pipeline's case)
    for (int qi = blockIdx.y * 4 + wave; qi < nq; qi += gridDim.y * 4) {
    const int q = list[(size_t)v * N + qi];
#ifdef PD_HPR_STATS
    const unsigned long long t_begin = wall_clock64();
#endif
    const d3 pi = {qf[q], qf[N + q], qf[2 * (size_t)N + q]};
    // working set (the query itself and the tail past NS are left out: index -2 never wins)
    const int pq = pos_of[(size_t)v * N + q];
    const int base = max(0, min((pq & ~63) - 64, ((NS + 63) & ~63) - 64 * HPR_LOCAL));
    double lx[HPR_LOCAL + 1], ly[HPR_LOCAL + 1], lz[HPR_LOCAL + 1];
    int li[HPR_LOCAL + 1];
#pragma unroll
    for (int u = 0; u < HPR_LOCAL; ++u) {
        const int j = base + u * 64 + lane;
        const int jc = min(j, NS - 1);
        lx[u] = fx[jc]; ly[u] = fy[jc]; lz[u] = fz[jc];
        const int id = sidx[jc];
        li[u] = (j < NS && id != q) ? id : -2;
    }
    lx[HPR_LOCAL] = ly[HPR_LOCAL] = lz[HPR_LOCAL] = 0.0; li[HPR_LOCAL] = -2;       // the slot for what the global scans return
    // (everything below is identical in every lane, except inside closest_with_newest)
    d3 W0 = {0, 0, 0}, W1 = {0, 0, 0}, W2 = {0, 0, 0}, vclose = pi, dir = pi;     // first direction: straight out along the point's own ray
    // ... except for a member of the coarse set (most of this level's queries): it was found as the extreme point of a known
    // direction, in which it is most likely separated from everything else -- one certified scan instead of an iteration
    const int mk = mdir != nullptr ? mdir[(size_t)v * N + q] : 0;
    if (mk > 0) dir = d3{fdir[4 * (mk - 1)], fdir[4 * (mk - 1) + 1], fdir[4 * (mk - 1) + 2]};
    else if (qdir != nullptr) {
        const double* qd = qdir + ((size_t)v * N + q) * 3;
        const d3 dq = {qd[0], qd[1], qd[2]};
        if (!zero3(dq)) dir = dq;
    }
    int I0 = -2, I1 = -2, I2 = -2, n = 0, state = 0;
#ifdef PD_HPR_STATS
    int my_rounds = 0;
    unsigned long long cand_chunks = 0, scans = 0;
#endif
    for (int round = 0; round < GJK_MAX_ROUNDS && state == 0; ++round) {
#ifdef PD_HPR_STATS
        ++my_rounds;
#endif
#ifdef PD_HPR_STATS
        const unsigned long long f0 = lab_clock();
#endif
        const double dx = dir.x, dy = dir.y, dz = dir.z;
        double best = -1.0e300, bx = 0.0, by = 0.0, bz = 0.0;
        int bidx = 0x7fffffff;
#pragma unroll
        for (int u = 0; u <= HPR_LOCAL; ++u) {
            const double val = fma(dz, lz[u], fma(dy, ly[u], dx * lx[u]));
            if (li[u] != -2 && (val > best || (val == best && li[u] < bidx))) { best = val; bidx = li[u]; bx = lx[u]; by = ly[u]; bz = lz[u]; }
        }
        double myv = wave_max_f64(best);
        int si = wave_min_i32(best == myv ? bidx : 0x7fffffff);
        const unsigned long long who = __ballot(best == myv && bidx == si && bidx != 0x7fffffff);
        bool have = who != 0ull;
        d3 sp = {0, 0, 0};
        if (have) {
            const int src = __builtin_amdgcn_readfirstlane(__builtin_ctzll(who));
            sp = d3{lane_f64(bx, src), lane_f64(by, src), lane_f64(bz, src)};
        }
        const double di = fma(dz, pi.z, fma(dy, pi.y, dx * pi.x));
        const double l1 = fabs(dx) + fabs(dy) + fabs(dz);
        const double vv = dot(vclose, vclose);
        // is the working set's answer enough?  Not if it does not pass the origin (only the whole set can certify "visible"), nor
        // if it is already a vertex of the simplex or brings the simplex no closer (only the whole set can say "no progress")
#ifdef PD_HPR_STATS
        const unsigned long long f1 = lab_clock() + (have ? 0 : 0);
#endif
        bool weak = true;
        if (have && myv > 0.0 && di - myv <= 0.0) {
            weak = (n >= 1 && si == I0) || (n >= 2 && si == I1) || (n >= 3 && si == I2);
            if (n > 0 && !weak) { const double va = dot(vclose, sp - pi); weak = vv - va <= 1.0e-11 * vv; }
        } else if (!(have && myv > 0.0) && di <= 0.0) weak = false;              // the eye passes the origin: it is the support
        if (weak) {
            // nothing below the working set's best, the eye's value or a simplex vertex's can be the support
            double th = have ? fmax(myv, 0.0) : 0.0;
            if (n >= 1) th = fmax(th, dot(dir, W0) + di);
            if (n >= 2) th = fmax(th, dot(dir, W1) + di);
            if (n >= 3) th = fmax(th, dot(dir, W2) + di);
            const double slk = (4.0 * rb + HPR_BOUND_SLACK * ma) * l1;
            th -= slk;
#ifdef PD_HPR_STATS
            ++scans;
            unsigned long long* ncp = &cand_chunks;
#else
            unsigned long long* ncp = nullptr;
#endif
            const Support r = support_scan<false, HPR_SCAN_BATCH>(fx, fy, fz, sidx, NS, boxes, dx, dy, dz, th, slk, q, 0.0, 0.0, 0.0, s_cand[wave], lane, ncp,
                                                     base >> 6, (base >> 6) + HPR_LOCAL);      // (the working set's own chunks are in myv already)
            if (r.pos >= 0 && (!have || r.val > myv || (r.val == myv && r.idx < si))) {
                myv = r.val; si = r.idx; sp = d3{r.x, r.y, r.z}; have = true;
            }
            // what the scan came across joins the working set: every lane keeps the best far point IT saw (the winner is among them) --
            // the next directions are close to this one, and the next violator is most likely one of these 64
            if (r.lidx >= 0) { lx[HPR_LOCAL] = r.lx; ly[HPR_LOCAL] = r.ly; lz[HPR_LOCAL] = r.lz; li[HPR_LOCAL] = r.lidx; }
        }
#ifdef PD_HPR_STATS
        const unsigned long long f2 = lab_clock() + (have ? 0 : 0);
#endif
        // the support point of S_i in this direction is now exact: that point, or the eye (value 0)
        const bool real = have && myv > 0.0;
        const double gap = di - (real ? myv : 0.0);
        if (gap > rb * l1) { state = 1; break; }                                 // certified visible
        if (gap > 0.0 && n >= 1) { state = 3; break; }                           // visible in f64, but inside the rounding bound
        const d3 a = sel3(real, sp - pi, neg(pi));
        const int ai = real ? si : -1;
        if (zero3(a)) { state = 3; break; }                                       // coincides with another point: the double-double stage has the rule
        if ((n >= 1 && ai == I0) || (n >= 2 && ai == I1) || (n >= 3 && ai == I2)) { state = 3; break; }
        if (n > 0 && vv - dot(vclose, a) <= 1.0e-11 * vv) { state = 3; break; }   // no progress: the origin is outside by less than f64 can show
        if (!closest_with_newest(a, W0, W1, W2, I0, I1, I2, n, ai, vclose, lane)) {
            // origin inside the tetrahedron (a, W0, W1, W2) in f64: certify p'_i strictly inside with the four determinants
            const int s0 = -det_sign(W0, W1, W2), s1 = det_sign(a, W1, W2), s2 = -det_sign(a, W0, W2), s3 = det_sign(a, W0, W1);
            state = (s0 != 0 && s0 == s1 && s1 == s2 && s2 == s3) ? 2 : 3;
            if (state == 3) { I2 = I1; I1 = I0; I0 = ai; n = 3; }               // (seed for the next stage: three of the four)
            break;
        }
        if (zero3(vclose)) { state = 3; break; }
        dir = neg(vclose);
#ifdef PD_HPR_STATS
        { const unsigned long long f3 = lab_clock() + (n == 77 ? 1 : 0);
          if (lane == 0) { atomicAdd(&g_hpr_c[5], f1 - f0); atomicAdd(&g_hpr_c[6], f2 - f1); atomicAdd(&g_hpr_c[7], f3 - f2); } }
#endif
    }
#ifdef PD_HPR_STATS
    if (lane == 0) { const unsigned long long dt = wall_clock64() - t_begin + (state == 77 ? 1 : 0);
                     atomicAdd(&g_hpr_t[0], dt); atomicMax(&g_hpr_t[1], dt); if (mk > 0) { atomicAdd(&g_hpr_t[2], dt); atomicAdd(&g_hpr_t[3], 1ull); }
                     atomicMax(&g_hpr_t[4], scans); if (scans > 3) atomicAdd(&g_hpr_t[5], 1ull); atomicMax(&g_hpr_t[6], cand_chunks); if (dt > 20000) atomicAdd(&g_hpr_t[7], 1ull); }
    if (lane == 0) { atomicAdd(&g_hpr_stats[1][0], 1ull); atomicAdd(&g_hpr_stats[1][1], (unsigned long long)my_rounds); atomicAdd(&g_hpr_stats[1][5], cand_chunks);
                     atomicAdd(&g_hpr_stats[1][6], scans);
                     atomicAdd(&g_hpr_stats[1][2], 1ull); atomicAdd(&g_hpr_stats[1][3], (unsigned long long)my_rounds);
                     if (state == 0) atomicAdd(&g_hpr_stats[1][4], 1ull);
                     atomicAdd(&g_hpr_stats[1][8 + min(my_rounds, 63) / 8], 1ull); }
#endif
    if (lane == 0) {
        vis[(size_t)v * N + q] = (state == 1) ? 1 : 0;
        if (state == 0 || state == 3) {                               // round cap reached / not certifiable in f64: the double-double stage
            const int pos = atomicAdd(&unc_count[v], 1);
            unc_list[(size_t)v * N + pos] = q;
            int* sd = unc_seed + ((size_t)v * N + pos) * 4;           // the simplex it stopped at seeds that stage
            sd[0] = n >= 1 ? I0 : -2; sd[1] = n >= 2 ? I1 : -2; sd[2] = n >= 3 ? I2 : -2; sd[3] = -2;
        }
    }
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

// ---- fallback: one 512-lane block per query the boolean iteration could not certify (almost always: it cycled until the round
// cap).  The distance iteration above over the same support set (any superset of the hull vertices), first in f64 with the f64
// certificates (T = double), and for what that cannot certify in double-double (T = dd: exact differences, 2^-104 arithmetic,
// 2^-96 bounds, up to 512 rounds).  Duplicates of the query with a larger cloud index are excluded from S_i (of coinciding
// points the smallest index is the hull vertex).
template <typename T> struct Num;
template <> struct Num<double> {
    static __device__ __forceinline__ double from(double a) { return a; }
    static __device__ __forceinline__ double diff(double a, double b) { return a - b; }
    static __device__ __forceinline__ bool gt(double a, double b) { return a > b; }
    static __device__ __forceinline__ bool eq(double a, double b) { return a == b; }
    static __device__ __forceinline__ double shfl(double a, int off) { return __shfl_xor(a, off); }
    static __device__ __forceinline__ double lowest() { return -1.0e300; }
    static constexpr double RB = 8.0 * 1.1102230246251565e-16;          // support value: <= 3.01 u |d|_1 max|coordinate| per side
    static constexpr double STALL = 1.0e-11;
    static constexpr int ROUNDS = 96;
};
template <> struct Num<dd> {
    static __device__ __forceinline__ dd from(double a) { return dd_from(a); }
    static __device__ __forceinline__ dd diff(double a, double b) { return dd_diff(a, b); }
    static __device__ __forceinline__ bool gt(dd a, dd b) { return a.hi > b.hi || (a.hi == b.hi && a.lo > b.lo); }
    static __device__ __forceinline__ bool eq(dd a, dd b) { return a.hi == b.hi && a.lo == b.lo; }
    static __device__ __forceinline__ dd shfl(dd a, int off) { return dd{__shfl_xor(a.hi, off), __shfl_xor(a.lo, off)}; }
    static __device__ __forceinline__ dd lowest() { return dd{-1.0e300, 0.0}; }
    static constexpr double RB = 1.2621774483536189e-29;                 // 2^-96
    static constexpr double STALL = 8.0e-25;
    static constexpr int ROUNDS = 512;
};
template <typename T>
__global__ __launch_bounds__(512) void k_hpr_exact(const double* __restrict__ flipped, int N, const int* __restrict__ unc_count,
                                                   const int* __restrict__ unc_list, const int* __restrict__ unc_seed, uint8_t* __restrict__ vis,
                                                   const double* __restrict__ ss, const int* __restrict__ sidx_all, int scap,
                                                   const int* __restrict__ scount, const float4* __restrict__ boxes_all,
                                                   const int* __restrict__ pos_of, const unsigned long long* __restrict__ maxabs,
                                                   int* __restrict__ next_count, int* __restrict__ next_list, int* __restrict__ next_seed,
                                                   int* __restrict__ counters /*[V][4]: dd queries, unresolved, dd rounds, f64-distance queries*/) {
    typedef Num<T> NT;
    constexpr bool WAVE = sizeof(T) == sizeof(double);      // f64 stage: one wavefront per query, box-culled scans (support_scan)
    __shared__ int s_cand[256];
    const float4* boxes = boxes_all + (size_t)blockIdx.y * ((scap + 63) >> 6) * (HPR_BOX_FLOATS / 4);
    const double ma = __longlong_as_double((long long)maxabs[blockIdx.y]);
    const int v = blockIdx.y;
    const int nq = unc_count[v];
    __shared__ T s_b[8];
    __shared__ int s_i[8];
    const double* fx = ss + (size_t)v * 3 * scap;
    const double* fy = fx + scap;
    const double* fz = fy + scap;
    const int* sidx = sidx_all + (size_t)v * scap;
    const int NS = scount[v];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double rb = __longlong_as_double((long long)maxabs[v]) * NT::RB;
    for (int u = blockIdx.x; u < nq; u += gridDim.x) {
        const int q = unc_list[(size_t)v * N + u];
        const double px = flipped[(size_t)v * 3 * N + q], py = flipped[(size_t)v * 3 * N + N + q], pz = flipped[(size_t)v * 3 * N + 2 * (size_t)N + q];
        double lx[HPR_LOCAL + 1], ly[HPR_LOCAL + 1], lz[HPR_LOCAL + 1];
        int li[HPR_LOCAL + 1], n_extra = 0, ws_chunk = 0;
        if constexpr (WAVE) {          // working set: the 256 support points around the query in the sorted order + what the global scans return
            const int pq = pos_of[(size_t)v * N + q];
            const int base = max(0, min((pq & ~63) - 64, ((NS + 63) & ~63) - 64 * HPR_LOCAL));
            ws_chunk = base >> 6;
#pragma unroll
            for (int u = 0; u < HPR_LOCAL; ++u) {
                const int j = base + u * 64 + lane;
                const int jc = min(j, NS - 1);
                lx[u] = fx[jc]; ly[u] = fy[jc]; lz[u] = fz[jc];
                const int id = sidx[jc];
                li[u] = (j < NS && id != q && !(id > q && lx[u] == px && ly[u] == py && lz[u] == pz)) ? id : -2;
            }
            lx[HPR_LOCAL] = ly[HPR_LOCAL] = lz[HPR_LOCAL] = 0.0; li[HPR_LOCAL] = -2;
        }
        Simplex<T> S;
        S.n = 0;
        v3<T> dir = {NT::from(px), NT::from(py), NT::from(pz)};         // first direction: straight out along the point's own ray
        v3<T> vclose = dir;
        int state = 0;                  // 0 running, 1 visible (certified), 2 hidden (certified), 3 not certifiable
        int rounds = 0;
        {   // seed: the vertices the previous pass stopped at (cloud indices, -1 = the eye) -- its closest point is the start
            const int* sd = unc_seed + ((size_t)v * N + u) * 4;
            const double* cx = flipped + (size_t)v * 3 * N;
            for (int k = 0; k < 3; ++k) {
                const int id = sd[k];
                if (id == -2 || id == q) continue;
                bool dup = false;
                for (int m = 0; m < S.n; ++m) dup = dup || S.idx[m] == id;
                if (dup) continue;
                v3<T> a = id >= 0 ? v3<T>{NT::diff(cx[id], px), NT::diff(cx[N + id], py), NT::diff(cx[2 * (size_t)N + id], pz)}
                                  : v3<T>{NT::from(-px), NT::from(-py), NT::from(-pz)};
                if (zero3(a)) continue;
                for (int m = S.n; m > 0; --m) { S.w[m] = S.w[m - 1]; S.idx[m] = S.idx[m - 1]; }
                S.w[0] = a; S.idx[0] = id; ++S.n;
            }
            if (S.n > 0 && closest_simplex(S, vclose) && !zero3(vclose)) dir = neg(vclose);
            else { S.n = 0; vclose = dir; }
        }
        for (; rounds < NT::ROUNDS && state == 0; ++rounds) {
            T best = NT::lowest();
            double spx = 0.0, spy = 0.0, spz = 0.0;           // the support point and its cloud index
            int spi = -1;
            bool have = false;
            if constexpr (WAVE) {
                // the working set first (see k_hpr_fine_local); the whole support set only when its answer is not enough
                const double dxx = sgn_of(dir.x), dyy = sgn_of(dir.y), dzz = sgn_of(dir.z);
                const double dpi = fma(dzz, pz, fma(dyy, py, dxx * px));
                const double l1 = fabs(dxx) + fabs(dyy) + fabs(dzz);
                double lb = -1.0e300, bx = 0.0, by = 0.0, bz = 0.0;
                int bidx = 0x7fffffff;
#pragma unroll
                for (int u = 0; u <= HPR_LOCAL; ++u) {
                    const double val = fma(dzz, lz[u], fma(dyy, ly[u], dxx * lx[u]));
                    if (li[u] != -2 && (val > lb || (val == lb && li[u] < bidx))) { lb = val; bidx = li[u]; bx = lx[u]; by = ly[u]; bz = lz[u]; }
                }
                double m = wave_max_f64(lb);
                int mi = wave_min_i32(lb == m ? bidx : 0x7fffffff);
                const unsigned long long who = __ballot(lb == m && bidx == mi && bidx != 0x7fffffff);
                have = who != 0ull;
                if (have) {
                    const int src = __builtin_amdgcn_readfirstlane(__builtin_ctzll(who));
                    spx = lane_f64(bx, src); spy = lane_f64(by, src); spz = lane_f64(bz, src); spi = mi;
                }
                bool weak = true;                                 // separated / repeated vertex / no progress: ask the whole set
                if (have && m > 0.0 && dpi - m <= 0.0) {
                    weak = false;
                    for (int k = 0; k < S.n; ++k) weak = weak || S.idx[k] == spi;
                    if (S.n > 0 && !weak) {
                        const v3<T> al = {NT::diff(spx, px), NT::diff(spy, py), NT::diff(spz, pz)};
                        const T vv = dot(vclose, vclose), va = dot(vclose, al);
                        weak = sgn_of(vv - va) <= 0.0 || mag_of(vv - va) <= NT::STALL * mag_of(vv);
                    }
                } else if (!(have && m > 0.0) && dpi <= 0.0) weak = false;       // the eye passes the origin: it is the support
                if (weak) {
                    // nothing below the working set's best, a simplex vertex's value or the eye's can be the support
                    double thr = have ? fmax(m, 0.0) : 0.0;
                    for (int k = 0; k < S.n; ++k) thr = fmax(thr, sgn_of(dot(dir, S.w[k])) + dpi);
                    const double slk = (4.0 * rb + HPR_BOUND_SLACK * ma) * l1;
                    thr -= slk;
#ifdef PD_HPR_STATS
                    unsigned long long ncand = 0;
                    const Support r = support_scan<true, 2>(fx, fy, fz, sidx, NS, boxes, dxx, dyy, dzz, thr, slk, q, px, py, pz, s_cand, lane, &ncand, ws_chunk, ws_chunk + HPR_LOCAL);
                    if (lane == 0) { atomicAdd(&g_hpr_stats[0][5], ncand); atomicAdd(&g_hpr_stats[0][6], 1ull); }
#else
                    const Support r = support_scan<true, 2>(fx, fy, fz, sidx, NS, boxes, dxx, dyy, dzz, thr, slk, q, px, py, pz, s_cand, lane, nullptr, ws_chunk, ws_chunk + HPR_LOCAL);
#endif
                    if (r.pos >= 0 && (!have || r.val > m || (r.val == m && r.idx < mi))) {
                        have = true; m = r.val; spx = r.x; spy = r.y; spz = r.z; spi = r.idx;
                        if (lane == n_extra) { lx[HPR_LOCAL] = r.x; ly[HPR_LOCAL] = r.y; lz[HPR_LOCAL] = r.z; li[HPR_LOCAL] = r.idx; }
                        n_extra = min(n_extra + 1, 64);
                    }
                }
                if (have) best = NT::from(m);
            } else {
                // support scan: (dx x + dy y) + dz z with exact inputs x, y, z
                int bi = 0x7fffffff;
                for (int j = threadIdx.x; j < NS; j += 512) {
                    const double x = fx[j], y = fy[j], zz = fz[j];
                    const int jo = sidx[j];
                    if (jo == q || (jo > q && x == px && y == py && zz == pz)) continue;
                    const T val = (dir.x * NT::from(x) + dir.y * NT::from(y)) + dir.z * NT::from(zz);
                    if (NT::gt(val, best)) { best = val; bi = j; }
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const T ob = NT::shfl(best, off);
                    const int oi = __shfl_xor(bi, off);
                    if (NT::gt(ob, best) || (NT::eq(ob, best) && oi < bi)) { best = ob; bi = oi; }
                }
                __syncthreads();
                if (lane == 0) { s_b[wave] = best; s_i[wave] = bi; }
                __syncthreads();
                best = s_b[0]; bi = s_i[0];
#pragma unroll
                for (int w = 1; w < 8; ++w) {
                    const T ob = s_b[w];
                    const int oi = s_i[w];
                    if (NT::gt(ob, best) || (NT::eq(ob, best) && oi < bi)) { best = ob; bi = oi; }
                }
                have = bi < NS;
                if (have) { spx = fx[bi]; spy = fy[bi]; spz = fz[bi]; spi = sidx[bi]; }
            }
            // every lane runs the identical state machine on the merged result
            const bool real = have && sgn_of(best) > 0.0;
            const T di = (dir.x * NT::from(px) + dir.y * NT::from(py)) + dir.z * NT::from(pz);
            const T gap = real ? di - best : di;
            const double tol = rb * (mag_of(dir.x) + mag_of(dir.y) + mag_of(dir.z));
            if (sgn_of(gap) > 0.0 && mag_of(gap) > tol) { state = 1; break; }          // separating direction, certified
            v3<T> a;
            int ai = -1;
            if (real) { a = v3<T>{NT::diff(spx, px), NT::diff(spy, py), NT::diff(spz, pz)}; ai = spi; }
            else a = v3<T>{NT::from(-px), NT::from(-py), NT::from(-pz)};
            if (zero3(a)) { state = real ? 2 : 3; break; }     // coincides with a point of smaller index (larger ones are excluded): that one is the vertex
            bool seen = false;
            for (int k = 0; k < S.n; ++k) seen = seen || S.idx[k] == ai;
            if (S.n > 0) {
                // no progress: the support does not get closer to the origin than the closest point already found =>
                // vclose IS the closest point of conv(S_i): the origin is outside, but by less than the certificate can show
                const T vv = dot(vclose, vclose), va = dot(vclose, a);
                if (seen || sgn_of(vv - va) <= 0.0 || mag_of(vv - va) <= NT::STALL * mag_of(vv)) { state = 3; break; }
            }
            // newest vertex first (Ericson's region tests are written around vertex a)
            for (int k = S.n; k > 0; --k) { S.w[k] = S.w[k - 1]; S.idx[k] = S.idx[k - 1]; }
            S.w[0] = a; S.idx[0] = ai; ++S.n;
            if (!closest_simplex(S, vclose)) {
                // origin inside the tetrahedron (in this arithmetic): certify p'_i strictly inside with the four determinants
                const int s0 = -det_sign(S.w[1], S.w[2], S.w[3]), s1 = det_sign(S.w[0], S.w[2], S.w[3]),
                          s2 = -det_sign(S.w[0], S.w[1], S.w[3]), s3 = det_sign(S.w[0], S.w[1], S.w[2]);
                state = (s0 != 0 && s0 == s1 && s1 == s2 && s2 == s3) ? 2 : 3;
                break;
            }
            if (zero3(vclose)) { state = 3; break; }          // the origin lies ON a face / edge of the simplex within this arithmetic
            dir = neg(vclose);
        }
#ifdef PD_HPR_STATS
        if (threadIdx.x == 0 && next_count != nullptr) atomicAdd(&g_hpr_stats[1][7], (unsigned long long)rounds);
#endif
        if (threadIdx.x == 0) {
            const bool settled = state == 1 || state == 2;
            if (next_count != nullptr) {                         // f64 stage: what it cannot certify goes on to double-double
                atomicAdd(&counters[4 * v + 3], 1);
                if (settled) vis[(size_t)v * N + q] = state == 1 ? 1 : 0;
                else {
                    const int pos = atomicAdd(&next_count[v], 1);
                    next_list[(size_t)v * N + pos] = q;
                    int* sd = next_seed + ((size_t)v * N + pos) * 4;
                    for (int k = 0; k < 4; ++k) sd[k] = k < 3 && k < S.n ? S.idx[k] : -2;
                }
            } else {
                vis[(size_t)v * N + q] = state == 1 ? 1 : 0;
                atomicAdd(&counters[4 * v], 1);
                if (!settled) atomicAdd(&counters[4 * v + 1], 1);      // not certifiable even in double-double
                atomicAdd(&counters[4 * v + 2], rounds);
            }
        }
        __syncthreads();
    }
}

// ---- level-2 inputs.  Support set = every point the coarse hull does not certainly enclose (coarse-set members, points the
// cheaper test accepted, queries level 1 could not settle); query list = those of them without a verdict yet.  The support set is
// counting-sorted by the Morton code of its cell in a 32^3 grid over the view's bounding box (order inside a cell is whatever the
// atomics give: the scans break ties by cloud index, so the results do not depend on it), then boxed per 64 points.
#define HPR_CELL_BITS 5
#define HPR_NCELL (1 << (3 * HPR_CELL_BITS))
__device__ __forceinline__ int spread3(int x) {                               // 5 bits -> every third bit
    x = (x | (x << 8)) & 0x0300f; x = (x | (x << 4)) & 0x030c3; x = (x | (x << 2)) & 0x09249;
    return x;
}
// outside == nullptr: one-level mode, every point is a support point and every not-skipped point a query
__global__ __launch_bounds__(1024) void k_hpr_bin(const double* __restrict__ flipped, int N, const uint8_t* __restrict__ outside, const uint8_t* __restrict__ skip,
                          const unsigned long long* __restrict__ bbox /*[V][6] keys of max(-x,-y,-z), max(x,y,z)*/, int* __restrict__ cellkey,
                          int* __restrict__ hist, int* __restrict__ count2, int* __restrict__ list2, uint8_t* __restrict__ vis) {
    __shared__ int s_wcnt[16], s_base;                            // (1024-lane blocks: the returning atomic below is ~0.2 us per block)
    const int v = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double* f = flipped + (size_t)v * 3 * N;
    double lo[3], sc[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        lo[k] = -key_f64(bbox[6 * v + k]);
        const double ext = key_f64(bbox[6 * v + 3 + k]) - lo[k];
        sc[k] = ext > 0.0 ? (double)(1 << HPR_CELL_BITS) / ext : 0.0;
    }
    for (int i0 = blockIdx.y * blockDim.x; i0 < N; i0 += gridDim.y * blockDim.x) {
        const int i = i0 + threadIdx.x;
        const bool in = i < N;
        const uint8_t o = in ? (outside ? outside[(size_t)v * N + i] : 1) : 0;
        const bool sk = in && skip != nullptr && skip[(size_t)v * N + i];
        const bool out = in && (o != 0 || sk), qry = o != 0 && !sk;
        if (in) {
            int key = -1;
            if (out) {
                int cxyz[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) cxyz[k] = min((1 << HPR_CELL_BITS) - 1, max(0, (int)((f[(size_t)k * N + i] - lo[k]) * sc[k])));
                key = spread3(cxyz[0]) | (spread3(cxyz[1]) << 1) | (spread3(cxyz[2]) << 2);
                atomicAdd(&hist[(size_t)v * HPR_NCELL + key], 1);
            }
            cellkey[(size_t)v * N + i] = key;
        }
        const unsigned long long bal = __ballot(qry);
        if (lane == 0) s_wcnt[wave] = __popcll(bal);
        __syncthreads();
        int mine = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { const int c = s_wcnt[w]; mine += w < wave ? c : 0; tot += c; }
        if (threadIdx.x == 0) s_base = atomicAdd(&count2[v], tot);
        __syncthreads();
        const int base = s_base + mine;
        if (qry) list2[(size_t)v * N + base + __popcll(bal & ((1ull << lane) - 1ull))] = i;
        __syncthreads();
    }
}
// exclusive prefix sum of a view's cell histogram, in place; scount = the number of support points
__global__ __launch_bounds__(1024) void k_hpr_cellscan(int* __restrict__ hist, int* __restrict__ scount) {
    __shared__ int s_w[16];
    const int v = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int PER = HPR_NCELL / 1024;
    int* h = hist + (size_t)v * HPR_NCELL + threadIdx.x * PER;
    int loc[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) { loc[k] = h[k]; sum += loc[k]; }
    int inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(inc, off); if (lane >= off) inc += o; }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    int base = inc - sum, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { base += w < wave ? s_w[w] : 0; total += s_w[w]; }
#pragma unroll
    for (int k = 0; k < PER; ++k) { h[k] = base; base += loc[k]; }
    if (threadIdx.x == 0) scount[v] = total;
}
__global__ void k_hpr_scatter(const double* __restrict__ flipped, int N, const int* __restrict__ cellkey, int* __restrict__ cellpos,
                              double* __restrict__ ss, int* __restrict__ sidx, int* __restrict__ pos_of) {
    const int v = blockIdx.x;
    const double* f = flipped + (size_t)v * 3 * N;
    double* so = ss + (size_t)v * 3 * N;
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < N; i += gridDim.y * blockDim.x) {
        const int key = cellkey[(size_t)v * N + i];
        if (key < 0) continue;
        const int pos = atomicAdd(&cellpos[(size_t)v * HPR_NCELL + key], 1);
        so[pos] = f[i]; so[N + pos] = f[N + i]; so[2 * (size_t)N + pos] = f[2 * (size_t)N + i];
        sidx[(size_t)v * N + pos] = i;
        pos_of[(size_t)v * N + i] = pos;
    }
}
// oriented bounding box of every 64 consecutive support points: one wavefront per chunk.  Axes (f32): e0 = the chunk's mean
// direction from the eye, e1, e2 = tangents; extents = min / max of the f64 projections on the axes AS STORED, rounded outwards.
__device__ __forceinline__ float f32_below(double a) {                      // largest f32 <= a
    float f = (float)a;
    if ((double)f > a) f = f > 0.0f ? __uint_as_float(__float_as_uint(f) - 1u) : (f < 0.0f ? __uint_as_float(__float_as_uint(f) + 1u) : -1.0e-45f);
    return f;
}
__global__ __launch_bounds__(256) void k_hpr_boxes(const double* __restrict__ ss, int N, const int* __restrict__ scount, float4* __restrict__ boxes) {
    const int v = blockIdx.y, lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int NS = scount[v];
    if ((c << 6) >= NS) return;
    const int j = min((c << 6) + lane, NS - 1);                   // (a clamped duplicate cannot change a min / max)
    const double* f = ss + (size_t)v * 3 * N;
    const double x = f[j], y = f[N + j], z = f[2 * (size_t)N + j];
    float sx = (float)x, sy = (float)y, sz = (float)z;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { sx += __shfl_xor(sx, off); sy += __shfl_xor(sy, off); sz += __shfl_xor(sz, off); }
    float e[3][3];
    const float n0 = sqrtf(sx * sx + sy * sy + sz * sz);
    if (n0 > 0.0f && n0 < 3.0e38f) { e[0][0] = sx / n0; e[0][1] = sy / n0; e[0][2] = sz / n0; }
    else { e[0][0] = 0.0f; e[0][1] = 0.0f; e[0][2] = 1.0f; }
    const float ax = fabsf(e[0][0]), ay = fabsf(e[0][1]), az = fabsf(e[0][2]);
    float hx = 0.0f, hy = 0.0f, hz = 0.0f;                        // the world axis least aligned with e0
    if (ax <= ay && ax <= az) hx = 1.0f; else if (ay <= az) hy = 1.0f; else hz = 1.0f;
    float tx = e[0][1] * hz - e[0][2] * hy, ty = e[0][2] * hx - e[0][0] * hz, tz = e[0][0] * hy - e[0][1] * hx;
    const float n1 = sqrtf(tx * tx + ty * ty + tz * tz);
    e[1][0] = tx / n1; e[1][1] = ty / n1; e[1][2] = tz / n1;
    e[2][0] = e[0][1] * e[1][2] - e[0][2] * e[1][1]; e[2][1] = e[0][2] * e[1][0] - e[0][0] * e[1][2]; e[2][2] = e[0][0] * e[1][1] - e[0][1] * e[1][0];
    float lo[3], hi[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double a = ((double)e[k][0] * x + (double)e[k][1] * y) + (double)e[k][2] * z;
        hi[k] = -f32_below(-wave_max_f64(a)); lo[k] = f32_below(-wave_max_f64(-a));
    }
    if (lane == 0) {
        float4* b = boxes + ((size_t)v * ((N + 63) >> 6) + c) * (HPR_BOX_FLOATS / 4);
        b[0] = make_float4(e[0][0], e[0][1], e[0][2], e[1][0]); b[1] = make_float4(e[1][1], e[1][2], e[2][0], e[2][1]);
        b[2] = make_float4(e[2][2], lo[0], lo[1], lo[2]); b[3] = make_float4(hi[0], hi[1], hi[2], 0.0f);
    }
}

// ---- coarse set: the extreme point of the flipped cloud in each of KC Fibonacci-sphere directions, without repeats.  Any set of
// cloud points serves level 1 (its only verdict is certified on the points themselves, and its members are queried like every
// other point), so this is the same approximate f32 GEMM + column maximum as the level-1 scan: A = 32 points, B = 64 of the
// directions, a wave keeps the best tile per lane and locates the row afterwards.  The cloud is cut into at most HPR_EXT_SLABS slabs
// of >= HPR_EXT_POINTS points (one wave per slab and 64 directions); every slab stores its (value, index) keys and k_hpr_extremes_fin
// takes the maximum over the slabs.  The points are read from an (x, y, z, 0) f32 copy of the flipped cloud that k_hpr_flip writes:
// the row search touches 32 scattered points per lane, and from the three f64 planes that was 96 separate cache lines per lane --
// 19 us of the texture addresser's time per shape (41 -> 24 us; slabs of 512 / 1024 / 2048 points: 36 / 26 / 24 us -- a wave's
// fixed costs, the row search first, outweigh the shorter scan).
#ifndef HPR_EXT_POINTS
#define HPR_EXT_POINTS 2048
#endif
#ifndef HPR_EXT_SLABS
#define HPR_EXT_SLABS 64
#endif
static int ext_slab_points(int N) {                                       // multiple of 128 (a trip of the scan)
    const int per = (N + HPR_EXT_SLABS - 1) / HPR_EXT_SLABS;
    return ((per > HPR_EXT_POINTS ? per : HPR_EXT_POINTS) + 127) / 128 * 128;
}
__device__ __forceinline__ unsigned int f32_key(float x) { const unsigned int b = __float_as_uint(x); return (b >> 31) ? ~b : (b | 0x80000000u); }
__global__ __launch_bounds__(256) void k_hpr_extremes(const float4* __restrict__ f32pts /*[V][N]*/, int N, const double* __restrict__ fdir /*[KC][4]*/,
                                                      int slab_points, unsigned long long* __restrict__ keys /*[slabs][V][KC]*/) {
    const int v = blockIdx.z, lane = threadIdx.x & 63, l31 = lane & 31;
    const bool hi = lane >= 32;
    const int grp = blockIdx.y * 4 + (threadIdx.x >> 6);                  // 64 directions
    const int p_lo = blockIdx.x * slab_points, p_hi = min(N, p_lo + slab_points);
    if (grp * 64 >= HPR_KC || p_lo >= N) return;
    const float4* fp = f32pts + (size_t)v * N;                            // (one 16-byte load per point: the f64 planes cost three
    const int k0 = grp * 64 + l31, k1 = k0 + 32;                           // scattered lines per point in the row search below)
    const float x0 = (float)fdir[4 * k0], y0 = (float)fdir[4 * k0 + 1], z0 = (float)fdir[4 * k0 + 2];
    const float x1 = (float)fdir[4 * k1], y1 = (float)fdir[4 * k1 + 1], z1 = (float)fdir[4 * k1 + 2];
    // split-f16 operands as in k_hpr_coarse: A (lanes 0-31 | 32-63) = [xh yh zh xh yh zh xl yl | zl 0 ...], B = [dxh dyh dzh dxl dyl dzl dxh dyh | dzh 0 ...]
    f16x8 bq0, bq1;
    {
        const _Float16 xh = (_Float16)x0, yh = (_Float16)y0, zh = (_Float16)z0;
        const _Float16 xl = (_Float16)(x0 - (float)xh), yl = (_Float16)(y0 - (float)yh), zl = (_Float16)(z0 - (float)zh);
        bq0 = hi ? f16x8{zh, 0, 0, 0, 0, 0, 0, 0} : f16x8{xh, yh, zh, xl, yl, zl, xh, yh};
        const _Float16 uh = (_Float16)x1, vh = (_Float16)y1, wh = (_Float16)z1;
        const _Float16 ul = (_Float16)(x1 - (float)uh), vl = (_Float16)(y1 - (float)vh), wl = (_Float16)(z1 - (float)wh);
        bq1 = hi ? f16x8{wh, 0, 0, 0, 0, 0, 0, 0} : f16x8{uh, vh, wh, ul, vl, wl, uh, vh};
    }
    const f32x16 zero = {0};
    float best0 = -3.0e38f, best1 = -3.0e38f;
    int t0 = 0, t1 = 0;
    // operands are requested two tiles-of-64 (one trip of 128 points) ahead
    float4 rp[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int h = 0; h < 2; ++h) rp[u][h] = fp[min(p_lo + 64 * u + 32 * h + l31, N - 1)];     // (a repeated point cannot change a maximum)
    for (int jo = p_lo; jo < p_hi; jo += 128) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {                                     // two tiles per trip
            const int j0 = jo + 64 * u;
            f16x8 ap[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float x = rp[u][h].x, y = rp[u][h].y, z = rp[u][h].z;
                const _Float16 xh = (_Float16)x, yh = (_Float16)y, zh = (_Float16)z;
                const _Float16 xl = (_Float16)(x - (float)xh), yl = (_Float16)(y - (float)yh), zl = (_Float16)(z - (float)zh);
                ap[h] = hi ? f16x8{zl, 0, 0, 0, 0, 0, 0, 0} : f16x8{xh, yh, zh, xh, yh, zh, xl, yl};
                rp[u][h] = fp[min(j0 + 128 + 32 * h + l31, N - 1)];
            }
            const f32x16 A0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ap[0], bq0, zero, 0, 0, 0), A1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ap[0], bq1, zero, 0, 0, 0);
            const f32x16 B0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ap[1], bq0, zero, 0, 0, 0), B1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ap[1], bq1, zero, 0, 0, 0);
            const float ma0 = tile_max(A0), ma1 = tile_max(A1), mb0 = tile_max(B0), mb1 = tile_max(B1);
            if (ma0 > best0) { best0 = ma0; t0 = j0; }
            if (ma1 > best1) { best1 = ma1; t1 = j0; }
            if (mb0 > best0) { best0 = mb0; t0 = j0 + 32; }
            if (mb1 > best1) { best1 = mb1; t1 = j0 + 32; }
        }
    }
    // the row inside the best tile: the lane's 16 rows in f32 (close to the matrix core's split-f16 values; the set only has to be good)
    int i0 = -1, i1 = -1;
    {
        float r0 = -3.0e38f, r1 = -3.0e38f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = 8 * (i / 4) + (i % 4) + (hi ? 4 : 0);
            const int ja = min(t0 + row, N - 1), jb = min(t1 + row, N - 1);
            const float4 pa = fp[ja], pb = fp[jb];
            const float w0 = fmaf(pa.z, z0, fmaf(pa.y, y0, pa.x * x0));
            const float w1 = fmaf(pb.z, z1, fmaf(pb.y, y1, pb.x * x1));
            if (w0 > r0) { r0 = w0; i0 = ja; }
            if (w1 > r1) { r1 = w1; i1 = jb; }
        }
    }
    // both half-waves hold a candidate for the same two directions: join them (ties: the smaller index), then one key per direction
    {
        const float ob0 = __shfl_xor(best0, 32), ob1 = __shfl_xor(best1, 32);
        const int oi0 = __shfl_xor(i0, 32), oi1 = __shfl_xor(i1, 32);
        if (ob0 > best0 || (ob0 == best0 && oi0 >= 0 && (i0 < 0 || oi0 < i0))) { best0 = ob0; i0 = oi0; }
        if (ob1 > best1 || (ob1 == best1 && oi1 >= 0 && (i1 < 0 || oi1 < i1))) { best1 = ob1; i1 = oi1; }
    }
    unsigned long long* kv = keys + ((size_t)blockIdx.x * gridDim.z + v) * HPR_KC + grp * 64;
    const float bq = hi ? best1 : best0;
    const int iq = hi ? i1 : i0;
    kv[lane] = iq >= 0 ? ((unsigned long long)f32_key(bq) << 32) | (unsigned int)(0x7fffffff - iq) : 0ull;
}
// The hull is that of the cloud AND the eye (the origin of the flipped space): in a direction where every point has a negative
// projection the eye is the extreme element and no point is taken (the GJK step supplies the eye itself, support value 0).
// A point joins the coarse set the first time a direction finds it (many directions share their extreme point).
__global__ void k_hpr_extremes_fin(const double* __restrict__ flipped, int N, const unsigned long long* __restrict__ keys, int slabs,
                                   float4* __restrict__ csf, double* __restrict__ csd /*[V][KC][4]: the same points in f64*/,
                                   int* __restrict__ cidx, int* __restrict__ kcount, int* __restrict__ claim /*[V][N], zeroed*/,
                                   int* __restrict__ mdir /*[V][N], zeroed: 1 + the direction that found the point*/) {
    const int v = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= HPR_KC) return;
    unsigned long long key = 0ull;                                        // maximum over the slabs: largest value, then smallest index
    for (int sl = 0; sl < slabs; sl += 8) {
        unsigned long long kk[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) kk[u] = keys[((size_t)min(sl + u, slabs - 1) * gridDim.y + v) * HPR_KC + k];
#pragma unroll
        for (int u = 0; u < 8; ++u) key = kk[u] > key ? kk[u] : key;
    }
    const unsigned int vb = (unsigned int)(key >> 32);
    const int id = 0x7fffffff - (int)(unsigned int)(key & 0xffffffffu);
    if (vb <= 0x80000000u || id < 0 || id >= N) return;                   // no point with a positive projection
    if (atomicExch(&claim[(size_t)v * N + id], 1) != 0) return;
    mdir[(size_t)v * N + id] = k + 1;
    const double* f = flipped + (size_t)v * 3 * N;
    const int pos = atomicAdd(&kcount[v], 1);
    csf[(size_t)v * HPR_KC + pos] = make_float4((float)f[id], (float)f[N + id], (float)f[2 * (size_t)N + id], 0.0f);
    double* cd = csd + ((size_t)v * HPR_KC + pos) * 4;
    cd[0] = f[id]; cd[1] = f[N + id]; cd[2] = f[2 * (size_t)N + id]; cd[3] = 0.0;
    cidx[(size_t)v * HPR_KC + pos] = id;
}

static size_t a256(size_t x) { return (x + 255) & ~(size_t)255; }
static size_t flipped_bytes(int V, int N) { return a256((size_t)V * 3 * (size_t)(N > 0 ? N : 1) * sizeof(double)); }
static size_t lists_bytes(int V, int N) { return a256((size_t)V * ((size_t)N + 64) * sizeof(int)); }
static size_t boxes_bytes(int V, int N) { return a256((size_t)V * (size_t)((N + 63) / 64 + 1) * HPR_BOX_FLOATS * sizeof(float)); }
static size_t hist_bytes(int V) { return a256((size_t)V * HPR_NCELL * sizeof(int)); }
// workspace head: counters int[64][4] (double-double queries, unresolved, double-double rounds, f64 distance-iteration queries),
// maxabs u64[64] at 1024, bounding-box keys u64[64][6] at 2048, six per-view counters int[64] at 5120
#define HPR_HEAD_BYTES 8192
extern "C" size_t pdhip_hpr_ws_bytes(int V, int N) {
    return HPR_HEAD_BYTES + 3 * flipped_bytes(V, N) + 16 * lists_bytes(V, N) + a256((size_t)V * N) + a256((size_t)V * HPR_KC * sizeof(float4)) + a256((size_t)V * HPR_KC * 4 * sizeof(double)) + a256((size_t)HPR_EXT_SLABS * V * HPR_KC * sizeof(unsigned long long)) + a256((size_t)V * (size_t)(N > 0 ? N : 1) * sizeof(float4)) +
           a256((size_t)V * HPR_KC * sizeof(int)) + boxes_bytes(V, N) + hist_bytes(V) + a256((size_t)HPR_KC * 4 * sizeof(double)) +
           a256((size_t)V * HPR_GRID * HPR_GRID * sizeof(unsigned long long));
}

static int hpr_impl(const float* points, int N, const double* eyes_dev, int V, int vps, double radius, const uint8_t* skip, uint8_t* visibility,
                    void* ws, void* stream);
extern "C" int pdhip_hidden_point_removal(const float* points, int N, const double* eyes_dev, int V, double radius,
                                          const uint8_t* skip, uint8_t* visibility, void* ws, void* stream) {
    return hpr_impl(points, N, eyes_dev, V, V > 0 ? V : 1, radius, skip, visibility, ws, stream);
}
// S shapes of N points each in ONE set of launches: points [S,N,3], eyes [S*V,3] (the shape's V eye positions, repeated per shape),
// skip / visibility [S*V,N]; view g = s * V + v looks at the cloud of shape s.  S * V <= 64; workspace pdhip_hpr_ws_bytes(S * V, N).
extern "C" int pdhip_hidden_point_removal_shapes(const float* points, int N, const double* eyes_dev, int V, int S, double radius,
                                                 const uint8_t* skip, uint8_t* visibility, void* ws, void* stream) {
    PD_REQUIRE(S >= 1 && V >= 1, "pdhip_hidden_point_removal_shapes: bad sizes");
    return hpr_impl(points, N, eyes_dev, S * V, V, radius, skip, visibility, ws, stream);
}
static int hpr_impl(const float* points, int N, const double* eyes_dev, int V, int vps, double radius, const uint8_t* skip, uint8_t* visibility,
                    void* ws, void* stream) {
    PD_REQUIRE(V > 0 && N >= 0, "pdhip_hidden_point_removal: bad sizes");
    if (N == 0) return PDHIP_OK;
    PD_REQUIRE(points && eyes_dev && visibility && ws, "pdhip_hidden_point_removal: null pointer");
    PD_REQUIRE(V <= 64, "pdhip_hidden_point_removal: at most 64 views");
    hipStream_t s = as_stream(stream);
    // everything that must start at zero sits at the front of the workspace: one fill
    char* p = reinterpret_cast<char*>(ws);
    int* counters = reinterpret_cast<int*>(p);
    unsigned long long* maxabs = reinterpret_cast<unsigned long long*>(p + 1024);
    unsigned long long* bbox = reinterpret_cast<unsigned long long*>(p + 2048);
    int* count = reinterpret_cast<int*>(p + 5120);                 // six per-view counters of 64 ints each
    int* count2 = count + 64; int* scount = count + 128; int* ucount = count + 192; int* u2count = count + 256; int* kcount = count + 320;
    p += HPR_HEAD_BYTES;
    int* hist = reinterpret_cast<int*>(p); p += hist_bytes(V);
    uint8_t* outside = reinterpret_cast<uint8_t*>(p); p += a256((size_t)V * N);
    float4* csf = reinterpret_cast<float4*>(p); p += a256((size_t)V * HPR_KC * sizeof(float4));      // (entries past the set's end: the eye)
    int* pos_of = reinterpret_cast<int*>(p); p += lists_bytes(V, N);   // (the extremes' claim flags until the scatter fills it)
    int* mdir = reinterpret_cast<int*>(p); p += lists_bytes(V, N);
    unsigned long long* sgrid = reinterpret_cast<unsigned long long*>(p); p += a256((size_t)V * HPR_GRID * HPR_GRID * sizeof(unsigned long long));
    const size_t zero_bytes = (size_t)(p - reinterpret_cast<char*>(ws));
    double* flipped = reinterpret_cast<double*>(p); p += flipped_bytes(V, N);
    double* ss = reinterpret_cast<double*>(p); p += flipped_bytes(V, N);          // level-2 support set, cell-sorted
    int* list = reinterpret_cast<int*>(p); p += lists_bytes(V, N);
    int* list2 = reinterpret_cast<int*>(p); p += lists_bytes(V, N);
    int* sidx = reinterpret_cast<int*>(p); p += lists_bytes(V, N);
    int* ulist = reinterpret_cast<int*>(p); p += lists_bytes(V, N);               // queries for the f64 distance iteration of level 3
    int* useed = reinterpret_cast<int*>(p); p += 4 * lists_bytes(V, N);           // ... and the simplex each stopped at
    int* u2list = reinterpret_cast<int*>(p); p += lists_bytes(V, N);              // queries for the double-double iteration
    int* u2seed = reinterpret_cast<int*>(p); p += 4 * lists_bytes(V, N);
    int* cellkey = reinterpret_cast<int*>(p); p += lists_bytes(V, N);
    double* csd = reinterpret_cast<double*>(p); p += a256((size_t)V * HPR_KC * 4 * sizeof(double));
    int* cidx = reinterpret_cast<int*>(p); p += a256((size_t)V * HPR_KC * sizeof(int));
    float4* boxes = reinterpret_cast<float4*>(p); p += boxes_bytes(V, N);
    double* fdir = reinterpret_cast<double*>(p); p += a256((size_t)HPR_KC * 4 * sizeof(double));
    double* qdir = reinterpret_cast<double*>(p); p += flipped_bytes(V, N);
    unsigned long long* ekeys = reinterpret_cast<unsigned long long*>(p); p += a256((size_t)HPR_EXT_SLABS * V * HPR_KC * sizeof(unsigned long long));   // [slab][V][KC]
    float4* f32pts = reinterpret_cast<float4*>(p); p += a256((size_t)V * (size_t)(N > 0 ? N : 1) * sizeof(float4));
    dim3 gf(V, min(cdiv(N, 256), 256));      // (the view is the FASTEST grid index in these kernels: workgroup b runs on XCD b % 8, so with 8 views an XCD's L2 holds ONE view's flipped cloud / grid / support set -- k_hpr_shield fetched 12x its input when every XCD touched every view)
    const bool two_level = N > 4 * HPR_KC;   // the coarse level pays off only when the cloud is much larger than the coarse set
    PD_HIP(hipMemsetAsync(ws, 0, zero_bytes, s));
    k_hpr_flip<<<dim3(V, min(cdiv(N, 1024), 64)), 1024, 0, s>>>
       (points, N, eyes_dev, radius, flipped, maxabs, bbox, skip, count, list, visibility, two_level ? 0 : 1, two_level ? f32pts : nullptr, vps);      // (one level: + marks the skipped points visible; `list` = the queries)
    constexpr int KC = HPR_KC;
    if (two_level) {
        k_hpr_grid<<<gf, 256, 0, s>>>(flipped, N, bbox, sgrid, fdir);
        k_hpr_shield<<<dim3(V, min(cdiv(N, 1024), 64)), 1024, 0, s>>>(flipped, N, bbox, sgrid, skip, count, list, visibility, counters);
        const int slab_points = ext_slab_points(N), slabs = cdiv(N, slab_points);
        k_hpr_extremes<<<dim3(slabs, KC / 256, V), 256, 0, s>>>(f32pts, N, fdir, slab_points, ekeys);
        k_hpr_extremes_fin<<<dim3(KC / 256, V), 256, 0, s>>>(flipped, N, ekeys, slabs, csf, csd, cidx, kcount, pos_of, mdir);
        k_hpr_coarse<2><<<dim3(V, cdiv(N, 256)), 256, 0, s>>>(flipped, N, count, list, csf, csd, cidx, kcount, outside, visibility, maxabs, qdir);
    }
    k_hpr_bin<<<dim3(V, min(cdiv(N, 1024), 64)), 1024, 0, s>>>(flipped, N, two_level ? outside : nullptr, skip, bbox, cellkey, hist, count2, list2, visibility);
    k_hpr_cellscan<<<V, 1024, 0, s>>>(hist, scount);
    k_hpr_scatter<<<gf, 256, 0, s>>>(flipped, N, cellkey, hist, ss, sidx, pos_of);
    k_hpr_boxes<<<dim3(cdiv(cdiv(N, 64), 4), V), 256, 0, s>>>(ss, N, scount, boxes);
    k_hpr_fine_dist<<<dim3(V, min(cdiv(N, 4), 512)), 256, 0, s>>>(flipped, N, count2, list2, visibility, ss, sidx, scount, boxes, pos_of, two_level ? mdir : nullptr, fdir, two_level ? qdir : nullptr, maxabs, ucount, ulist, useed);
    k_hpr_exact<double><<<dim3(64, V), 64, 0, s>>>(flipped, N, ucount, ulist, useed, visibility, ss, sidx, N, scount, boxes, pos_of, maxabs, u2count, u2list, u2seed, counters);
    k_hpr_exact<dd><<<dim3(32, V), 512, 0, s>>>(flipped, N, u2count, u2list, u2seed, visibility, ss, sidx, N, scount, boxes, pos_of, maxabs, nullptr, nullptr, nullptr, counters);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}

// counters of the last pdhip_hidden_point_removal call that used `ws` (synchronises `stream`), summed over views: out[0] = queries
// sent to the double-double iteration, out[1] = queries not certifiable even there (reported hidden), out[2] = double-double
// rounds, out[3] = queries sent to the f64 distance iteration (level 3's first stage)
extern "C" int pdhip_hpr_read_counters(const void* ws, int V, long long* out /*[4]*/, void* stream) {
    PD_REQUIRE(ws && out && V > 0 && V <= 64, "pdhip_hpr_read_counters: bad arguments");
    int h[64 * 4];
    PD_HIP(hipMemcpyAsync(h, ws, sizeof(int) * 4 * V, hipMemcpyDeviceToHost, as_stream(stream)));
    PD_HIP(hipStreamSynchronize(as_stream(stream)));
    out[0] = out[1] = out[2] = out[3] = 0;
    for (int v = 0; v < V; ++v) { out[0] += h[4 * v]; out[1] += h[4 * v + 1]; out[2] += h[4 * v + 2]; out[3] += h[4 * v + 3]; }
    return PDHIP_OK;
}
