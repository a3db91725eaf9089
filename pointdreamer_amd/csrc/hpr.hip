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
// Fibonacci-sphere directions (one streaming pass).  conv(subset) is inside conv(cloud), so "origin enclosed" there is already
// the final answer (hidden).  A point strictly inside conv(subset) is also never a support point of the full cloud, so the
// second level -- only for the queries the coarse hull cannot enclose -- scans just the OUTSIDE set (~40 % of the cloud).  30 k points x 8 views: 88 -> 27 ms for all points, 48 -> 12.5 ms behind the depth-test skip mask
// (KC = 1024 measured best of 512..8192).
// qhull's facet-merging tolerances are not reproduced (PARITY UNPINNED, open3d absent): points within ~1e-9 of a hull
// facet may be classified differently; tests bound the disagreement with scipy's qhull.
#include "common.h"
using namespace pdhip;

#define QPW 16                 // query points per wavefront
#define GJK_MAX_ROUNDS 64
#define GJK_COARSE_ROUNDS 32
#define HPR_KC 1024            // coarse support set size

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
// COARSE: every point of the cloud is a query (ns = scap = KC extreme points); writes outside[v][q] = origin not enclosed.
// !COARSE: queries from list / count, support set = the points outside the coarse hull (ns = scount[v]); writes vis.
template <bool COARSE>
__global__ __launch_bounds__(256) void k_hpr_gjk(const double* __restrict__ flipped, int N, const int* __restrict__ count,
                                                 const int* __restrict__ list, uint8_t* __restrict__ vis,
                                                 const double* __restrict__ ss, const int* __restrict__ sidx_all, int scap,
                                                 const int* __restrict__ scount, uint8_t* __restrict__ outside) {
    __shared__ double s_dir[4][QPW][3];
    __shared__ int s_q[4][QPW];
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
    const int q0 = (blockIdx.x * 4 + wave) * QPW;
    const int nq = COARSE ? N : count[v];
    if (q0 >= nq) return;
    const bool owner = lane < QPW && q0 + lane < nq;
    const int q = owner ? (COARSE ? q0 + lane : list[(size_t)v * N + q0 + lane]) : -1;
    if (lane < QPW) s_q[wave][lane] = q;
    __builtin_amdgcn_wave_barrier();
    int qk[QPW];
#pragma unroll
    for (int k = 0; k < QPW; ++k) qk[k] = s_q[wave][k];
    // ---- per-query GJK state (meaningful in lanes < QPW)
    d3 pi = {0, 0, 0}, sa = {0, 0, 0}, sb = {0, 0, 0}, sc = {0, 0, 0}, sd = {0, 0, 0}, dir = {0, 0, 1};
    int dim = 0;                   // simplex size; phases: 0 -> fetch c, 1 -> fetch b, >= 2 -> main loop
    int state = owner ? 0 : 2;     // 0 running, 1 visible (origin outside), 2 hidden / not a query
    if (owner) {
        pi = {qfx[q], qfy[q], qfz[q]};
        dir = pi;                  // start looking straight out along the point's own ray
    }
    for (int round = 0; round < (COARSE ? GJK_COARSE_ROUNDS : GJK_MAX_ROUNDS); ++round) {
        if (__ballot(state == 0) == 0ull) break;
        if (lane < QPW) { s_dir[wave][lane][0] = dir.x; s_dir[wave][lane][1] = dir.y; s_dir[wave][lane][2] = dir.z; }
        __builtin_amdgcn_wave_barrier();
        double dx[QPW], dy[QPW], dz[QPW], best[QPW];
        int bi[QPW], bo[QPW];                                        // position in the support set / original index
#pragma unroll
        for (int k = 0; k < QPW; ++k) {
            dx[k] = s_dir[wave][k][0]; dy[k] = s_dir[wave][k][1]; dz[k] = s_dir[wave][k][2];
            best[k] = -1.0e300; bi[k] = 0x7fffffff; bo[k] = 0x7fffffff;
        }
        // ---- support scan: every lane streams points j = lane, lane+64, ...
        for (int j = lane; j < NS; j += 64) {
            const double x = fx[j], y = fy[j], z = fz[j];
            const int jo = sidx[j];                                  // index in the cloud (self-exclusion, tie-break)
#pragma unroll
            for (int k = 0; k < QPW; ++k) {
                double val = dx[k] * x + dy[k] * y + dz[k] * z;
                if (jo == qk[k]) val = -1.0e300;                    // S_i excludes the point itself
                if (val > best[k] || (val == best[k] && jo < bo[k])) { best[k] = val; bi[k] = j; bo[k] = jo; }
            }
        }
        double myv = -1.0e300;
        int myi = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < QPW; ++k) {
            double b = best[k];
            int id = bi[k], io = bo[k];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double ob = __shfl_xor(b, off);
                const int oi = __shfl_xor(id, off), oo = __shfl_xor(io, off);
                if (ob > b || (ob == b && oo < io)) { b = ob; id = oi; io = oo; }
            }
            if (lane == k) { myv = b; myi = id; }
        }
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
    if (!COARSE) {
        if (owner) vis[(size_t)v * N + q] = (state == 1) ? 1 : 0;
    } else {
        if (owner) outside[(size_t)v * N + q] = state != 2;           // 0: enclosed by the coarse hull (hidden, and never a support point)
    }
}

// second-level inputs: the support set = all points outside the coarse hull (coordinates + original index), and the query list =
// those of them that no cheaper test accepted.  Wave-compacted with one atomic per wave (order across waves is a race; the
// support argmax breaks ties on the ORIGINAL index, so the result does not depend on it).
__global__ void k_hpr_build(const double* __restrict__ flipped, int N, const uint8_t* __restrict__ outside,
                            const uint8_t* __restrict__ skip, double* __restrict__ ss, int* __restrict__ sidx, int* __restrict__ scount,
                            int* __restrict__ count2, int* __restrict__ list2) {
    __shared__ int s_o[4], s_q[4], s_bo, s_bq;
    const int v = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double* f = flipped + (size_t)v * 3 * N;
    double* so = ss + (size_t)v * 3 * N;
    for (int i0 = blockIdx.x * blockDim.x; i0 < N; i0 += gridDim.x * blockDim.x) {
        const int i = i0 + threadIdx.x;
        const bool out = i < N && outside[(size_t)v * N + i];
        const bool qry = out && !(skip != nullptr && skip[(size_t)v * N + i]);
        const unsigned long long bo = __ballot(out), bq = __ballot(qry);
        if (lane == 0) { s_o[wave] = __popcll(bo); s_q[wave] = __popcll(bq); }
        __syncthreads();
        if (threadIdx.x == 0) {                                       // one returning atomic per block and counter
            s_bo = atomicAdd(&scount[v], s_o[0] + s_o[1] + s_o[2] + s_o[3]);
            s_bq = atomicAdd(&count2[v], s_q[0] + s_q[1] + s_q[2] + s_q[3]);
        }
        __syncthreads();
        int baseo = s_bo, baseq = s_bq;
        for (int w = 0; w < wave; ++w) { baseo += s_o[w]; baseq += s_q[w]; }
        if (out) {
            const int pos = baseo + __popcll(bo & ((1ull << lane) - 1ull));
            so[pos] = f[i]; so[N + pos] = f[N + i]; so[2 * (size_t)N + pos] = f[2 * (size_t)N + i];
            sidx[(size_t)v * N + pos] = i;
        }
        if (qry) list2[(size_t)v * N + baseq + __popcll(bq & ((1ull << lane) - 1ull))] = i;
        __syncthreads();
    }
}

// coarse support set: the extreme point of the flipped cloud in each of KC Fibonacci-sphere directions (ties: smallest index)
__global__ __launch_bounds__(256) void k_hpr_extremes(const double* __restrict__ flipped, int N, int KC, double* __restrict__ cs,
                                                      int* __restrict__ cidx) {
    const int v = blockIdx.y;
    const double* fx = flipped + (size_t)v * 3 * N;
    const double* fy = fx + N;
    const double* fz = fy + N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k0 = (blockIdx.x * 4 + wave) * QPW;
    if (k0 >= KC) return;
    double dx[QPW], dy[QPW], dz[QPW], best[QPW];
    int bi[QPW];
#pragma unroll
    for (int k = 0; k < QPW; ++k) {
        const int kk = min(k0 + k, KC - 1);
        const double z = 1.0 - (2.0 * kk + 1.0) / KC, r = sqrt(fmax(0.0, 1.0 - z * z)), phi = kk * 2.399963229728653;
        dx[k] = r * cos(phi); dy[k] = r * sin(phi); dz[k] = z;
        best[k] = -1.0e300; bi[k] = 0x7fffffff;
    }
    for (int j = lane; j < N; j += 64) {
        const double x = fx[j], y = fy[j], z = fz[j];
#pragma unroll
        for (int k = 0; k < QPW; ++k) {
            const double val = dx[k] * x + dy[k] * y + dz[k] * z;
            if (val > best[k]) { best[k] = val; bi[k] = j; }
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
        if (lane == 0 && k0 + k < KC) {
            double* c = cs + (size_t)v * 3 * KC;
            c[k0 + k] = fx[id]; c[KC + k0 + k] = fy[id]; c[2 * (size_t)KC + k0 + k] = fz[id];
            cidx[(size_t)v * KC + k0 + k] = id;
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
    dim3 gg(cdiv(N, 4 * QPW), V);
    if (N > 4 * HPR_KC) {            // the coarse level pays off only when the cloud is much larger than the coarse set
        dim3 ge(cdiv(HPR_KC, 4 * QPW), V);
        k_hpr_extremes<<<ge, 256, 0, s>>>(flipped, N, HPR_KC, cs, cidx);
        k_hpr_gjk<true><<<gg, 256, 0, s>>>(flipped, N, nullptr, nullptr, nullptr, cs, cidx, HPR_KC, nullptr, outside);
        k_hpr_build<<<gf, 256, 0, s>>>(flipped, N, outside, skip, ss, sidx, scount, count2, list2);
        k_hpr_gjk<false><<<gg, 256, 0, s>>>(flipped, N, count2, list2, visibility, ss, sidx, N, scount, nullptr);
    } else {                         // one level: support set = the whole cloud
        k_hpr_iota<<<gf, 256, 0, s>>>(sidx, N);
        k_hpr_fill_count<<<1, 64, 0, s>>>(scount, V, N);
        k_hpr_gjk<false><<<gg, 256, 0, s>>>(flipped, N, count, list, visibility, flipped, sidx, N, scount, nullptr);
    }
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}
