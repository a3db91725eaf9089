// Row I0, texture_gen_method = 'linear': scipy.interpolate.griddata(method='linear') at pointdreamer/ours_utils.py:610-643 --
// Delaunay triangulation of the known pixels (sites), piecewise-linear (barycentric) interpolation inside each triangle, NaN
// outside the convex hull of the sites.  The reference runs qhull + a point-location walk on the CPU, one view at a time.
//
// Here nothing is triangulated: every unknown pixel q finds ITS OWN Delaunay triangle.  Lift the sites to the paraboloid
// p' = (x, y, x^2 + y^2); the Delaunay triangle over q is the facet of the lower convex hull that the vertical line through q
// pierces.  Two phases, both a sequence of streaming arg-max scans over the site list (the same skeleton as the hidden-point
// removal, hpr.hip: 16 queries per wavefront share every pass over the sites, 64 lanes stream them):
//   phase 0  boolean 2-D GJK on {p - q}: a triangle of sites that contains q, or "outside the hull" (-> NaN, as scipy);
//   phase 1  pivoting: the site deepest inside the triangle's circumcircle (= lowest below the lifted plane) replaces the
//            vertex that keeps q inside; the plane's height over q falls strictly, so this ends at the empty-circumcircle
//            triangle, i.e. the Delaunay triangle.
// Every score is a linear form A x + B y + C (x^2 + y^2) with integer coefficients below 2^32 and integer coordinates below
// 2^11: all products and sums are integers below 2^46, EXACT in float64 -- orientation, in-circle and support decisions carry
// no rounding.  Pixels are a degenerate input for Delaunay (co-circular sites everywhere): where four or more sites share an
// empty circle several triangulations are valid and qhull's choice is a property of its merge order; this kernel returns a
// valid one (ties: first maximum in site order), tests check validity there and equality where the triangle is unique.
#include "common.h"
using namespace pdhip;

#define LQ 16                          // queries per wavefront
#define LIN_MAX_ROUNDS 192

namespace {

struct i2 { double x, y; };            // integer-valued
__device__ __forceinline__ double orient(i2 a, i2 b, i2 c) { return (b.x - a.x) * (c.y - a.y) - (b.y - a.y) * (c.x - a.x); }

// per image: sites (x, y, x^2 + y^2 as f32 -- exact below 2^24) in row-major order, queries (pixel index) likewise.  Three launches
// (round 5; one 1024-thread block per image walking its pixels behind 2 x 64 block barriers took 87 us for 8 views of 256^2): one
// wavefront per image row counts the row's sites, one block per image turns the row counts into row offsets, one wavefront per row
// writes its sites / queries at the row's offset.  rowstart[b][y] = sites before row y (rowstart[b][H] = all), rowq likewise for queries.
__device__ __forceinline__ bool lin_site(const void* __restrict__ mask, int mask_is_f32, size_t i) {
    return mask_is_f32 ? reinterpret_cast<const float*>(mask)[i] != 0.0f : reinterpret_cast<const uint8_t*>(mask)[i] != 0;
}
__global__ __launch_bounds__(256) void k_linear_rowcount(const void* __restrict__ mask, int mask_is_f32, int64_t mask_bstride, int H, int W,
                                                          int* __restrict__ rowstart /*[B][H+1]*/) {
    const int b = blockIdx.y, lane = threadIdx.x & 63, y = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (y >= H) return;
    int cs = 0;
    for (int x0 = 0; x0 < W; x0 += 64) {
        const int x = x0 + lane;
        cs += __popcll(__ballot(x < W && lin_site(mask, mask_is_f32, (size_t)b * mask_bstride + (size_t)y * W + x)));
    }
    if (lane == 0) rowstart[(size_t)b * (H + 1) + y] = cs;
}
// exclusive scan of the row counts of one image in place (H <= 2048: two rows per thread); rowq[y] = queries before row y = y W - sites
__global__ __launch_bounds__(1024) void k_linear_rowscan(int H, int W, int* __restrict__ rowstart, int* __restrict__ rowq, int* __restrict__ counts /*[B][2]*/,
                                                         int* __restrict__ fbcount /*[B][2]*/) {
    __shared__ int s_w[16];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int* rs = rowstart + (size_t)b * (H + 1);
    int* rq = rowq + (size_t)b * (H + 1);
    const int y0 = 2 * threadIdx.x;
    const int c0 = y0 < H ? rs[y0] : 0, c1 = y0 + 1 < H ? rs[y0 + 1] : 0;
    const int sum = c0 + c1;
    int inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(inc, off); if (lane >= off) inc += o; }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    int base = inc - sum, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { base += w < wave ? s_w[w] : 0; total += s_w[w]; }
    if (y0 < H) { rs[y0] = base; rq[y0] = y0 * W - base; }
    if (y0 + 1 < H) { rs[y0 + 1] = base + c0; rq[y0 + 1] = (y0 + 1) * W - (base + c0); }
    if (threadIdx.x == 0) {
        rs[H] = total; rq[H] = H * W - total;
        counts[2 * b] = total; counts[2 * b + 1] = H * W - total;
        fbcount[2 * b] = total; fbcount[2 * b + 1] = 0;
    }
}
__global__ __launch_bounds__(256) void k_linear_rowwrite(const void* __restrict__ mask, int mask_is_f32, int64_t mask_bstride, int H, int W,
                                                          const int* __restrict__ rowstart, const int* __restrict__ rowq,
                                                          float* __restrict__ sites /*[B][3][H*W]*/, int* __restrict__ qlist /*[B][H*W]*/, int make_qlist) {
    const int b = blockIdx.y, lane = threadIdx.x & 63, y = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (y >= H) return;
    const int n = H * W;
    float* sx = sites + (size_t)b * 3 * n;
    int ps = rowstart[(size_t)b * (H + 1) + y], pq = rowq[(size_t)b * (H + 1) + y];
    for (int x0 = 0; x0 < W; x0 += 64) {
        const int x = x0 + lane;
        const bool site = x < W && lin_site(mask, mask_is_f32, (size_t)b * mask_bstride + (size_t)y * W + x);
        const bool qry = x < W && !site;
        const unsigned long long bs = __ballot(site), bq = __ballot(qry);
        if (site) {
            const int pos = ps + __popcll(bs & ((1ull << lane) - 1ull));
            sx[pos] = (float)x; sx[n + pos] = (float)y; sx[2 * (size_t)n + pos] = (float)(x * x + y * y);
        }
        if (qry && make_qlist) qlist[(size_t)b * n + pq + __popcll(bq & ((1ull << lane) - 1ull))] = y * W + x;
        ps += __popcll(bs); pq += __popcll(bq);
    }
}

// sites copy themselves
__global__ void k_linear_copy_sites(const float* __restrict__ img, float* __restrict__ out, const void* __restrict__ mask,
                                    int mask_is_f32, int64_t mask_bstride, int C, int n, int32_t* __restrict__ tri) {
    const int b = blockIdx.y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const bool site = mask_is_f32 ? reinterpret_cast<const float*>(mask)[(size_t)b * mask_bstride + i] != 0.0f
                                      : reinterpret_cast<const uint8_t*>(mask)[(size_t)b * mask_bstride + i] != 0;
        if (!site) continue;
        for (int c = 0; c < C; ++c) out[((size_t)b * C + c) * n + i] = img[((size_t)b * C + c) * n + i];
        if (tri) { tri[((size_t)b * n + i) * 3] = tri[((size_t)b * n + i) * 3 + 1] = tri[((size_t)b * n + i) * 3 + 2] = -1; }
    }
}

__global__ __launch_bounds__(256, 2) void k_linear_tri(const float* __restrict__ img, float* __restrict__ out, int C, int H, int W,
                                                       const float* __restrict__ sites, const int* __restrict__ qlist,
                                                       const int* __restrict__ counts, int32_t* __restrict__ tri /*[B][H*W][3] or null*/,
                                                       int* __restrict__ unresolved, int* __restrict__ bcount, int* __restrict__ blist /*[B*H*W][3]*/) {
    __shared__ double s_co[4][LQ][3];
    const int b = blockIdx.y, n = H * W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int NS = counts[2 * b], NQ = counts[2 * b + 1];
    const int q0 = (blockIdx.x * 4 + wave) * LQ;
    if (q0 >= NQ) return;
    const float* sx = sites + (size_t)b * 3 * n;
    const float* sy = sx + n;
    const float* sr = sy + n;
    constexpr int ST = 64 / LQ;
    const int kq = lane / ST;
    const bool slot = (lane % ST) == 0;
    const bool owner = slot && q0 + kq < NQ;
    const int qi = owner ? qlist[(size_t)b * n + q0 + kq] : 0;
    const i2 q = {(double)(qi % W), (double)(qi / W)};
    // ---- per-query state (meaningful in the slot lanes).  phase 0: 2-D GJK on {p - q}; phase 1: Delaunay pivots.
    int phase = 0, dim = 0;
    int state = owner ? (NS > 0 ? 0 : 2) : 3;        // 0 running, 1 done (triangle ia, ib, ic), 2 outside the hull (NaN), 3 idle
    i2 pa = {0, 0}, pb = {0, 0}, pc = {0, 0};        // absolute site coordinates
    int ia = -1, ib = -1, ic = -1;
    double A = 1.0, B = 0.0, Cc = 0.0, K = 0.0;      // score(p) = A x + B y + Cc (x^2 + y^2); phase 1 violation <=> score + K > 0
    for (int round = 0; round < LIN_MAX_ROUNDS; ++round) {
        if (__ballot(state == 0) == 0ull) break;
        if (slot) { s_co[wave][kq][0] = A; s_co[wave][kq][1] = B; s_co[wave][kq][2] = Cc; }
        __builtin_amdgcn_wave_barrier();
        double ca[LQ], cb[LQ], cc[LQ], best[LQ];
        int bi[LQ];
#pragma unroll
        for (int k = 0; k < LQ; ++k) { ca[k] = s_co[wave][k][0]; cb[k] = s_co[wave][k][1]; cc[k] = s_co[wave][k][2]; best[k] = -1.0e300; bi[k] = 0x7fffffff; }
        // ---- scan: sites in ascending order, a lane keeps its FIRST maximum, the butterflies prefer the smaller index
        float nx = 0.f, ny = 0.f, nr = 0.f;
        if (lane < NS) { nx = sx[lane]; ny = sy[lane]; nr = sr[lane]; }
        for (int j = lane; j < NS; j += 64) {
            const double x = (double)nx, y = (double)ny, r2 = (double)nr;
            const int jn = j + 64;
            if (jn < NS) { nx = sx[jn]; ny = sy[jn]; nr = sr[jn]; }
#pragma unroll
            for (int k = 0; k < LQ; ++k) {
                const double val = (ca[k] * x + cb[k] * y) + cc[k] * r2;
                if (val > best[k]) { best[k] = val; bi[k] = j; }
            }
        }
#define LIN_HALVE(I)                                                                                                  \
        if constexpr ((LQ >> (I)) > 1) {                                                                              \
            constexpr int off = 32 >> (I), nn = LQ >> (I);                                                            \
            const bool hi = (lane & off) != 0;                                                                        \
            _Pragma("unroll") for (int k = 0; k < nn / 2; ++k) {                                                      \
                const double send = hi ? best[k] : best[k + nn / 2], keep = hi ? best[k + nn / 2] : best[k];          \
                const int sendi = hi ? bi[k] : bi[k + nn / 2], keepi = hi ? bi[k + nn / 2] : bi[k];                   \
                const double ob = __shfl_xor(send, off);                                                              \
                const int oi = __shfl_xor(sendi, off);                                                                \
                const bool take = ob > keep || (ob == keep && oi < keepi);                                            \
                best[k] = take ? ob : keep; bi[k] = take ? oi : keepi;                                                \
            }                                                                                                         \
        }
        LIN_HALVE(0) LIN_HALVE(1) LIN_HALVE(2) LIN_HALVE(3)
#undef LIN_HALVE
#pragma unroll
        for (int off = ST / 2; off > 0; off >>= 1) {
            const double ob = __shfl_xor(best[0], off);
            const int oi = __shfl_xor(bi[0], off);
            if (ob > best[0] || (ob == best[0] && oi < bi[0])) { best[0] = ob; bi[0] = oi; }
        }
        if (state != 0) continue;
        const int j = bi[0];
        const i2 p = {(double)sx[j], (double)sy[j]};
        if (phase == 0) {
            // support point of {p - q} in direction (A, B): value = best - (A qx + B qy)
            const double sv = best[0] - (A * q.x + B * q.y);
            if (sv < 0.0) { state = 2; continue; }                                  // q outside the hull of the sites
            if (sv == 0.0 && !(dim == 1 && (pa.x - q.x) * B == (pa.y - q.y) * A)) {  // q ON the hull boundary (supporting line of
                state = 4; continue;                                                 // direction (A, B)): k_linear_boundary
            }
            if (dim == 0) { pa = p; ia = j; A = -(p.x - q.x); B = -(p.y - q.y); dim = 1; continue; }
            if (dim == 1) {
                pb = p; ib = j;
                const double abx = pa.x - pb.x, aby = pa.y - pb.y;
                double px = -aby, py = abx;                                          // perpendicular to ab, turned towards q
                const double t = px * (q.x - pb.x) + py * (q.y - pb.y);
                if (t < 0.0) { px = -px; py = -py; }
                // t == 0: q lies strictly inside segment ab (sv > 0 put b beyond q), but a and b are far-apart EXTREME sites, not
                // Delaunay neighbours -- never interpolate along ab.  Keep either perpendicular: a site strictly on that side closes a
                // triangle with q on its edge ab (the containment tests are inclusive) and phase 1 pivots to the Delaunay triangle;
                // no site on that side (sv == 0 next round) makes line ab a supporting line of the hull: k_linear_boundary takes the
                // nearest site on either side of q along it -- which is also the answer when ALL sites are collinear.
                A = px; B = py; dim = 2; continue;
            }
            pc = p; ic = j;
            // triangle (c, b, a): which edge region of the newest vertex holds q?
            const double cbx = pb.x - pc.x, cby = pb.y - pc.y, cax = pa.x - pc.x, cay = pa.y - pc.y;
            const double cox = q.x - pc.x, coy = q.y - pc.y;
            double nbx = -cby, nby = cbx;                                            // perpendicular to cb, away from a
            if (nbx * cax + nby * cay > 0.0) { nbx = -nbx; nby = -nby; }
            double nax = -cay, nay = cax;                                            // perpendicular to ca, away from b
            if (nax * cbx + nay * cby > 0.0) { nax = -nax; nay = -nay; }
            if (nbx * cox + nby * coy > 0.0) { pa = pb; ia = ib; pb = pc; ib = ic; A = nbx; B = nby; continue; }
            if (nax * cox + nay * coy > 0.0) { pb = pc; ib = ic; A = nax; B = nay; continue; }
            if (orient(pa, pb, pc) == 0.0) { state = 2; continue; }                  // (degenerate: collinear sites only)
            if (orient(pa, pb, pc) < 0.0) { const i2 tp = pb; pb = pc; pc = tp; const int ti = ib; ib = ic; ic = ti; }
            phase = 1;
        } else {
            // deepest site inside the circumcircle of (a, b, c): violation <=> score + K > 0
            if (best[0] + K <= 0.0) { state = 1; continue; }
            // the site replaces the vertex that keeps q inside (fan of the new site over the old triangle)
            const i2 t0[3] = {p, pa, pa}, t1[3] = {pb, p, pb}, t2[3] = {pc, pc, p};
            const int n0[3] = {j, ia, ia}, n1[3] = {ib, j, ib}, n2[3] = {ic, ic, j};
            int pick = -1;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (pick >= 0) continue;
                if (orient(t0[k], t1[k], t2[k]) > 0.0 && orient(t0[k], t1[k], q) >= 0.0 && orient(t1[k], t2[k], q) >= 0.0 &&
                    orient(t2[k], t0[k], q) >= 0.0) pick = k;
            }
            if (pick < 0) { state = 1; atomicAdd(unresolved, 1); continue; }         // cannot happen for q inside (a, b, c)
            pa = t0[pick]; pb = t1[pick]; pc = t2[pick]; ia = n0[pick]; ib = n1[pick]; ic = n2[pick];
        }
        // lifted plane through a', b', c' (counter-clockwise): N = (b' - a') x (c' - a'); p' below it <=> N . (p' - a') < 0
        const double ar = pa.x * pa.x + pa.y * pa.y, br = pb.x * pb.x + pb.y * pb.y, cr = pc.x * pc.x + pc.y * pc.y;
        const double ux = pb.x - pa.x, uy = pb.y - pa.y, uz = br - ar, vx = pc.x - pa.x, vy = pc.y - pa.y, vz = cr - ar;
        const double Nx = uy * vz - uz * vy, Ny = uz * vx - ux * vz, Nz = ux * vy - uy * vx;
        A = -Nx; B = -Ny; Cc = -Nz; K = (Nx * pa.x + Ny * pa.y) + Nz * ar;
    }
    if (!owner) return;
    if (state == 0) { atomicAdd(unresolved, 1); state = phase == 1 ? 1 : 2; }   // round cap: phase 1 holds a triangle that contains q (valid, not Delaunay)
    if (state == 4) {                                     // on the hull boundary: finished by k_linear_boundary
        const int k = atomicAdd(bcount, 1);
        blist[3 * k] = b * n + qi; blist[3 * k + 1] = (int)A; blist[3 * k + 2] = (int)B;
        return;
    }
    float* o = out + (size_t)b * C * n + qi;
    const float* im = img + (size_t)b * C * n;
    if (tri) { int32_t* t = tri + ((size_t)b * n + qi) * 3; t[0] = state == 1 ? ia : -2; t[1] = state == 1 ? ib : -2; t[2] = state == 1 ? ic : -2; }
    if (state != 1) {
        for (int c = 0; c < C; ++c) o[(size_t)c * n] = __builtin_nanf("");
        return;
    }
    const int xa = (int)pa.y * W + (int)pa.x, xb = (int)pb.y * W + (int)pb.x, xc = (int)pc.y * W + (int)pc.x;
    if (ib == ic) {                                                                  // q on the segment (a, b)
        const double len = (pb.x - pa.x) * (pb.x - pa.x) + (pb.y - pa.y) * (pb.y - pa.y);
        const double t = ((q.x - pa.x) * (pb.x - pa.x) + (q.y - pa.y) * (pb.y - pa.y)) / len;
        for (int c = 0; c < C; ++c) o[(size_t)c * n] = (float)((1.0 - t) * (double)im[(size_t)c * n + xa] + t * (double)im[(size_t)c * n + xb]);
        return;
    }
    const double wa = orient(q, pb, pc), wb = orient(pa, q, pc), wc = orient(pa, pb, q), Wt = orient(pa, pb, pc);
    for (int c = 0; c < C; ++c) {
        const double v = (wa * (double)im[(size_t)c * n + xa] + wb * (double)im[(size_t)c * n + xb]) + wc * (double)im[(size_t)c * n + xc];
        o[(size_t)c * n] = (float)(v / Wt);
    }
}

// ---- local first pass (round 3).  The reference's images are dense in sites (background + splatted points), so the Delaunay triangle
// of an unknown pixel is a LOCAL object; the global kernel above streams all ~50 k sites of an image ~20 times per query batch
// (58 ms for 8 views).  Here one wavefront owns a 16 x 16 pixel tile: it gathers the sites of the tile's window (tile +- LW pixels,
// row-major order preserved: the tie rules stay those of the global scan) into LDS once, and runs the same two phases for the
// tile's queries against the window only.  A triangle found this way is accepted iff its circumcircle stays clear of every pixel
// column / row OUTSIDE the window (conservative f64 test, or the window side is the image border): then "no window site inside
// the circumcircle" is "no site inside" and the triangle is a Delaunay triangle of the whole image.  Everything else -- circle
// leaving the window, query outside / on the hull of the window's sites, window overflow -- goes to the global kernel through a
// fallback list.  Same exact integer predicates in f64.
// Two window sizes (round 5): a scan costs the window's site count, and most unknown pixels of a splatted view have known pixels a
// step or two away -- their triangle's circumcircle stays inside a margin of LW1 = 6 pixels.  The first launch (8 x 8 tiles, LW_ = LW1:
// 20 x 20 windows, a sixth of the sites of the 48 x 48 one) settles those and marks what it cannot accept in `todo`; the second launch
// (16 x 16 tiles, LW_ = 16, FROM_TODO) takes only the marked pixels and hands ITS rejects to the global kernel as before.  Measured on the
// bench shape (8 views of 256^2): margins 2 .. 12 on 8 x 8 first tiles 1 548 / 1 361 / 1 273 / 1 140 / 1 081 (6) / 1 123 / 1 228 /
// 1 210 / 1 349 us; 16 x 16 first tiles 1 342-1 431; one pass (round 3) 2 962.  Where the Delaunay
// triangle is unique both windows return it; on co-circular sites (several valid triangulations, see the header) the choice may
// depend on the window, as it does on qhull's merge order in the reference.
#define LW 16                          // window margin (pixels) of the last local pass
#ifndef LW1
#define LW1 6                          // ... of the first one
#endif
#define LT 16                          // tile edge of the last local pass
#ifndef LT1
#define LT1 8                          // ... of the first one (a 20 x 20 window at LW1 = 6)
#endif
template <int LW_, int LT_, bool FROM_TODO, bool MARK_TODO>
__global__ __launch_bounds__(256) void k_linear_local(const float* __restrict__ img, float* __restrict__ out, int C, int H, int W,
                                                      const void* __restrict__ mask, int mask_is_f32, int64_t mask_bstride,
                                                      const int* __restrict__ rowstart /*[B][H+1]*/, int32_t* __restrict__ tri,
                                                      int* __restrict__ fbq /*[B][H*W]*/, int* __restrict__ fbcount /*[B][2]: NS copy, count*/,
                                                      uint8_t* __restrict__ todo /*[B][H*W]: written (MARK_TODO) / read (FROM_TODO)*/) {
    constexpr int LCAP = (LT_ + 2 * LW_) * (LT_ + 2 * LW_);
    __shared__ unsigned int s_site[4][LCAP];             // x | y << 16, window sites in row-major order
    __shared__ unsigned short s_q[4][LT_ * LT_];            // tile queries (x - tx0 | (y - ty0) << 8)
    __shared__ double s_co[4][LQ][3];
    const int b = blockIdx.y, n = H * W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tiles_x = (W + LT_ - 1) / LT_, tiles_y = (H + LT_ - 1) / LT_;
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= tiles_x * tiles_y) return;
    const int tx0 = (tile % tiles_x) * LT_, ty0 = (tile / tiles_x) * LT_;
    auto is_site = [&](int y, int x) -> bool {
        const size_t i = (size_t)b * mask_bstride + (size_t)y * W + x;
        return mask_is_f32 ? reinterpret_cast<const float*>(mask)[i] != 0.0f : reinterpret_cast<const uint8_t*>(mask)[i] != 0;
    };
    // ---- the tile's queries
    int NQ = 0;
    for (int i0 = 0; i0 < LT_ * LT_; i0 += 64) {
        const int i = i0 + lane, ly = i / LT_, lx = i - ly * LT_;
        bool qry = ty0 + ly < H && tx0 + lx < W;
        if (FROM_TODO) qry = qry && todo[(size_t)b * n + (size_t)(ty0 + ly) * W + tx0 + lx] != 0;
        else qry = qry && !is_site(ty0 + ly, tx0 + lx);
        const unsigned long long bq = __ballot(qry);
        if (qry) s_q[wave][NQ + __popcll(bq & ((1ull << lane) - 1ull))] = (unsigned short)(lx | (ly << 8));
        NQ += __popcll(bq);
    }
    if (NQ == 0) return;
    // ---- the window's sites (row-major)
    const int wx0 = max(tx0 - LW_, 0), wx1 = min(tx0 + LT_ + LW_, W) - 1, wy0 = max(ty0 - LW_, 0), wy1 = min(ty0 + LT_ + LW_, H) - 1;
    const int ww = wx1 - wx0 + 1, wh = wy1 - wy0 + 1;
    int NS = 0;
    for (int i0 = 0; i0 < ww * wh; i0 += 64) {
        const int i = i0 + lane, ly = i / ww, lx = i - ly * ww;
        const bool st = i < ww * wh && is_site(wy0 + ly, wx0 + lx);
        const unsigned long long bs = __ballot(st);
        if (st) s_site[wave][NS + __popcll(bs & ((1ull << lane) - 1ull))] = (unsigned)(wx0 + lx) | ((unsigned)(wy0 + ly) << 16);
        NS += __popcll(bs);
    }
    __builtin_amdgcn_wave_barrier();
    constexpr int ST = 64 / LQ;
    const int kq = lane / ST;
    const bool slot = (lane % ST) == 0;
    for (int q0 = 0; q0 < NQ; q0 += LQ) {
        const bool owner = slot && q0 + kq < NQ;
        const unsigned qp = owner ? s_q[wave][q0 + kq] : 0;
        const int qxi = tx0 + (int)(qp & 255), qyi = ty0 + (int)(qp >> 8);
        const int qi = qyi * W + qxi;
        const i2 q = {(double)qxi, (double)qyi};
        int phase = 0, dim = 0;
        int state = owner ? (NS > 0 ? 0 : 2) : 3;        // 0 running, 1 done, 2 outside the window's hull, 3 idle, 4 on its boundary
        i2 pa = {0, 0}, pb = {0, 0}, pc = {0, 0};
        double A = 1.0, B = 0.0, Cc = 0.0, K = 0.0;
        for (int round = 0; round < LIN_MAX_ROUNDS; ++round) {
            if (__ballot(state == 0) == 0ull) break;
            if (slot) { s_co[wave][kq][0] = A; s_co[wave][kq][1] = B; s_co[wave][kq][2] = Cc; }
            __builtin_amdgcn_wave_barrier();
            double ca[LQ], cb[LQ], cc[LQ], best[LQ];
            int bi[LQ];
#pragma unroll
            for (int k = 0; k < LQ; ++k) { ca[k] = s_co[wave][k][0]; cb[k] = s_co[wave][k][1]; cc[k] = s_co[wave][k][2]; best[k] = -1.0e300; bi[k] = 0x7fffffff; }
            for (int j = lane; j < NS; j += 64) {
                const unsigned sp = s_site[wave][j];
                const double x = (double)(sp & 0xffffu), y = (double)(sp >> 16), r2 = x * x + y * y;
#pragma unroll
                for (int k = 0; k < LQ; ++k) {
                    const double val = (ca[k] * x + cb[k] * y) + cc[k] * r2;
                    if (val > best[k]) { best[k] = val; bi[k] = j; }
                }
            }
#define LIN_HALVE(I)                                                                                                  \
            if constexpr ((LQ >> (I)) > 1) {                                                                          \
                constexpr int off = 32 >> (I), nn = LQ >> (I);                                                        \
                const bool hi = (lane & off) != 0;                                                                    \
                _Pragma("unroll") for (int k = 0; k < nn / 2; ++k) {                                                  \
                    const double send = hi ? best[k] : best[k + nn / 2], keep = hi ? best[k + nn / 2] : best[k];      \
                    const int sendi = hi ? bi[k] : bi[k + nn / 2], keepi = hi ? bi[k + nn / 2] : bi[k];               \
                    const double ob = __shfl_xor(send, off);                                                          \
                    const int oi = __shfl_xor(sendi, off);                                                            \
                    const bool take = ob > keep || (ob == keep && oi < keepi);                                        \
                    best[k] = take ? ob : keep; bi[k] = take ? oi : keepi;                                            \
                }                                                                                                     \
            }
            LIN_HALVE(0) LIN_HALVE(1) LIN_HALVE(2) LIN_HALVE(3)
#undef LIN_HALVE
#pragma unroll
            for (int off = ST / 2; off > 0; off >>= 1) {
                const double ob = __shfl_xor(best[0], off);
                const int oi = __shfl_xor(bi[0], off);
                if (ob > best[0] || (ob == best[0] && oi < bi[0])) { best[0] = ob; bi[0] = oi; }
            }
            if (state != 0) continue;
            const unsigned sp = s_site[wave][bi[0]];
            const i2 p = {(double)(sp & 0xffffu), (double)(sp >> 16)};
            if (phase == 0) {
                const double sv = best[0] - (A * q.x + B * q.y);
                if (sv < 0.0) { state = 2; continue; }
                if (sv == 0.0 && !(dim == 1 && (pa.x - q.x) * B == (pa.y - q.y) * A)) { state = 4; continue; }
                if (dim == 0) { pa = p; A = -(p.x - q.x); B = -(p.y - q.y); dim = 1; continue; }
                if (dim == 1) {
                    pb = p;
                    const double abx = pa.x - pb.x, aby = pa.y - pb.y;
                    double px = -aby, py = abx;
                    const double t = px * (q.x - pb.x) + py * (q.y - pb.y);
                    if (t < 0.0) { px = -px; py = -py; }
                    A = px; B = py; dim = 2; continue;
                }
                pc = p;
                const double cbx = pb.x - pc.x, cby = pb.y - pc.y, cax = pa.x - pc.x, cay = pa.y - pc.y;
                const double cox = q.x - pc.x, coy = q.y - pc.y;
                double nbx = -cby, nby = cbx;
                if (nbx * cax + nby * cay > 0.0) { nbx = -nbx; nby = -nby; }
                double nax = -cay, nay = cax;
                if (nax * cbx + nay * cby > 0.0) { nax = -nax; nay = -nay; }
                if (nbx * cox + nby * coy > 0.0) { pa = pb; pb = pc; A = nbx; B = nby; continue; }
                if (nax * cox + nay * coy > 0.0) { pb = pc; A = nax; B = nay; continue; }
                if (orient(pa, pb, pc) == 0.0) { state = 2; continue; }
                if (orient(pa, pb, pc) < 0.0) { const i2 tp = pb; pb = pc; pc = tp; }
                phase = 1;
            } else {
                if (best[0] + K <= 0.0) { state = 1; continue; }
                const i2 t0[3] = {p, pa, pa}, t1[3] = {pb, p, pb}, t2[3] = {pc, pc, p};
                int pick = -1;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    if (pick >= 0) continue;
                    if (orient(t0[k], t1[k], t2[k]) > 0.0 && orient(t0[k], t1[k], q) >= 0.0 && orient(t1[k], t2[k], q) >= 0.0 &&
                        orient(t2[k], t0[k], q) >= 0.0) pick = k;
                }
                if (pick < 0) { state = 2; continue; }                                   // (cannot happen; the global kernel decides)
                pa = t0[pick]; pb = t1[pick]; pc = t2[pick];
            }
            const double ar = pa.x * pa.x + pa.y * pa.y, br = pb.x * pb.x + pb.y * pb.y, cr = pc.x * pc.x + pc.y * pc.y;
            const double ux = pb.x - pa.x, uy = pb.y - pa.y, uz = br - ar, vx = pc.x - pa.x, vy = pc.y - pa.y, vz = cr - ar;
            const double Nx = uy * vz - uz * vy, Ny = uz * vx - ux * vz, Nz = ux * vy - uy * vx;
            A = -Nx; B = -Ny; Cc = -Nz; K = (Nx * pa.x + Ny * pa.y) + Nz * ar;
        }
        if (!owner) continue;
        bool accept = state == 1;
        if (accept) {
            // circumcircle of (a, b, c) against the window: centre = -(Nx, Ny) / (2 Nz) of the lifted plane, conservative margins
            const double ar = pa.x * pa.x + pa.y * pa.y, br = pb.x * pb.x + pb.y * pb.y, cr = pc.x * pc.x + pc.y * pc.y;
            const double d = 2.0 * orient(pa, pb, pc);
            const double cx = (ar * (pb.y - pc.y) + br * (pc.y - pa.y) + cr * (pa.y - pb.y)) / d;
            const double cy = (ar * (pc.x - pb.x) + br * (pa.x - pc.x) + cr * (pb.x - pa.x)) / d;
            const double r = sqrt((pa.x - cx) * (pa.x - cx) + (pa.y - cy) * (pa.y - cy)) * (1.0 + 1e-9) + 1e-3;
            accept = (wx0 == 0 || cx - r > (double)(wx0 - 1)) && (wx1 == W - 1 || cx + r < (double)(wx1 + 1)) &&
                     (wy0 == 0 || cy - r > (double)(wy0 - 1)) && (wy1 == H - 1 || cy + r < (double)(wy1 + 1));
        }
        if (MARK_TODO) todo[(size_t)b * n + qi] = accept ? 0 : 1;
        if (!accept) {
            if (!MARK_TODO) {
                const int k = atomicAdd(fbcount + 2 * b + 1, 1);
                fbq[(size_t)b * n + k] = qi;
            }
            continue;
        }
        float* o = out + (size_t)b * C * n + qi;
        const float* im = img + (size_t)b * C * n;
        if (tri) {
            // site-list index (row-major rank) of a vertex: sites before its row + sites before it in the row (debug / test path)
            auto rank = [&](i2 v) -> int {
                int r = rowstart[(size_t)b * (H + 1) + (int)v.y];
                for (int x = 0; x < (int)v.x; ++x) r += is_site((int)v.y, x) ? 1 : 0;
                return r;
            };
            int32_t* t = tri + ((size_t)b * n + qi) * 3;
            t[0] = rank(pa); t[1] = rank(pb); t[2] = rank(pc);
        }
        const int xa = (int)pa.y * W + (int)pa.x, xb = (int)pb.y * W + (int)pb.x, xc = (int)pc.y * W + (int)pc.x;
        const double wa = orient(q, pb, pc), wb = orient(pa, q, pc), wc = orient(pa, pb, q), Wt = orient(pa, pb, pc);
        for (int c = 0; c < C; ++c) {
            const double v = (wa * (double)im[(size_t)c * n + xa] + wb * (double)im[(size_t)c * n + xb]) + wc * (double)im[(size_t)c * n + xc];
            o[(size_t)c * n] = (float)(v / Wt);
        }
    }
}

// q on the boundary of the sites' hull, supporting direction d (every site has d . (p - q) <= 0): the Delaunay triangulation
// has the hull edge between the two sites ON that line nearest to q on either side; q beyond the last one is outside (NaN).
// One wavefront per query (rare: the reference's images carry a full border of sites).
__global__ __launch_bounds__(64) void k_linear_boundary(const float* __restrict__ img, float* __restrict__ out, int C, int H, int W,
                                                        const float* __restrict__ sites, const int* __restrict__ counts,
                                                        const int* __restrict__ bcount, const int* __restrict__ blist,
                                                        int32_t* __restrict__ tri) {
    const int n = H * W, lane = threadIdx.x;
    for (int k = blockIdx.x; k < *bcount; k += gridDim.x) {
        const int gi = blist[3 * k], b = gi / n, qi = gi - b * n;
        const double dx = blist[3 * k + 1], dy = blist[3 * k + 2];
        const double qx = qi % W, qy = qi / W;
        const float* sx = sites + (size_t)b * 3 * n;
        const float* sy = sx + n;
        const int NS = counts[2 * b];
        double lo = -1.0e300, hi = 1.0e300;               // along e = (-dy, dx): largest negative / smallest positive offset
        int ilo = -1, ihi = -1;
        for (int j = lane; j < NS; j += 64) {
            const double x = sx[j], y = sy[j];
            if (dx * (x - qx) + dy * (y - qy) != 0.0) continue;
            const double t = -dy * (x - qx) + dx * (y - qy);
            if (t < 0.0 && t > lo) { lo = t; ilo = j; }
            if (t > 0.0 && t < hi) { hi = t; ihi = j; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ol = __shfl_xor(lo, off), oh = __shfl_xor(hi, off);
            const int il = __shfl_xor(ilo, off), ih = __shfl_xor(ihi, off);
            if (ol > lo) { lo = ol; ilo = il; }
            if (oh < hi) { hi = oh; ihi = ih; }
        }
        if (lane != 0) continue;
        float* o = out + (size_t)b * C * n + qi;
        const float* im = img + (size_t)b * C * n;
        const bool ok = ilo >= 0 && ihi >= 0;
        if (tri) { int32_t* t = tri + ((size_t)b * n + qi) * 3; t[0] = ok ? ilo : -2; t[1] = t[2] = ok ? ihi : -2; }
        if (!ok) { for (int c = 0; c < C; ++c) o[(size_t)c * n] = __builtin_nanf(""); continue; }
        const int xa = (int)sy[ilo] * W + (int)sx[ilo], xb = (int)sy[ihi] * W + (int)sx[ihi];
        const double t = -lo / (hi - lo);
        for (int c = 0; c < C; ++c) o[(size_t)c * n] = (float)((1.0 - t) * (double)im[(size_t)c * n + xa] + t * (double)im[(size_t)c * n + xb]);
    }
}

}  // namespace

extern "C" size_t pdhip_linear_fill_ws_bytes(int B, int H, int W) {
    return (size_t)B * 3 * H * W * sizeof(float) + (size_t)B * H * W * sizeof(int) + (size_t)(2 * B + 64) * sizeof(int) +
           (size_t)B * H * W * 3 * sizeof(int) + 2 * (size_t)B * (H + 1) * sizeof(int) + (size_t)2 * B * sizeof(int) + (size_t)B * H * W;
}
static thread_local int g_linear_local = 1;             // tuning / test hook: 0 = global scans only (the round-2 path), 2 = one local pass (the 48 x 48 window only: round 3), 1 = two
extern "C" int pdhip_debug_set_linear_local(int on) { int old = g_linear_local; g_linear_local = on; return old; }

/* tri (may be NULL): per pixel the three site indices (in the image's row-major site order) of the triangle used, -1 at
 * sites, -2 outside the hull -- for tests.  unresolved: device int, incremented for queries that hit the round cap. */
extern "C" int pdhip_linear_fill(const float* img, float* out, int B, int C, int H, int W, const void* mask, int mask_is_f32,
                                 int64_t mask_batch_stride, void* ws, int32_t* tri, void* stream) {
    PD_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && H <= 2048 && W <= 2048, "pdhip_linear_fill: bad sizes (H, W <= 2048)");
    PD_REQUIRE(img && out && mask && ws && img != out, "pdhip_linear_fill: bad pointers");
    hipStream_t s = as_stream(stream);
    const int n = H * W;
    float* sites = reinterpret_cast<float*>(ws);
    int* qlist = reinterpret_cast<int*>(sites + (size_t)B * 3 * n);
    int* counts = qlist + (size_t)B * n;
    int* unresolved = counts + 2 * B;
    int* bcount = unresolved + 1;
    int* blist = counts + 2 * B + 64;
    int* rowstart = blist + (size_t)B * n * 3;
    int* rowq = rowstart + (size_t)B * (H + 1);
    int* fbcount = rowq + (size_t)B * (H + 1);
    uint8_t* todo = reinterpret_cast<uint8_t*>(fbcount + 2 * (size_t)B);
    const bool local = g_linear_local != 0 && W < 65536 && H < 65536;
    PD_HIP(hipMemsetAsync(unresolved, 0, 2 * sizeof(int), s));
    k_linear_rowcount<<<dim3(cdiv(H, 4), B), 256, 0, s>>>(mask, mask_is_f32, mask_batch_stride, H, W, rowstart);
    k_linear_rowscan<<<B, 1024, 0, s>>>(H, W, rowstart, rowq, counts, fbcount);
    k_linear_rowwrite<<<dim3(cdiv(H, 4), B), 256, 0, s>>>(mask, mask_is_f32, mask_batch_stride, H, W, rowstart, rowq, sites, qlist, local ? 0 : 1);
    k_linear_copy_sites<<<dim3(min(cdiv(n, 256), 1024), B), 256, 0, s>>>(img, out, mask, mask_is_f32, mask_batch_stride, C, n, tri);
    if (local) {
        // local pass over 16 x 16 tiles (window sites in LDS); what it cannot certify lands in the fallback list (written over the unused
        // query list), which the global kernel then finishes with (NS, count) = fbcount
        const int tiles = cdiv(W, LT) * cdiv(H, LT), tiles1 = cdiv(W, LT1) * cdiv(H, LT1);
        if (g_linear_local == 2)
            k_linear_local<LW, LT, false, false><<<dim3(cdiv(tiles, 4), B), 256, 0, s>>>(img, out, C, H, W, mask, mask_is_f32, mask_batch_stride, rowstart, tri, qlist, fbcount, todo);
        else {
            // (every pixel the second pass reads was written by the first: sites and background are never queries, their bytes are never read)
            PD_HIP(hipMemsetAsync(todo, 0, (size_t)B * n, s));
            k_linear_local<LW1, LT1, false, true><<<dim3(cdiv(tiles1, 4), B), 256, 0, s>>>(img, out, C, H, W, mask, mask_is_f32, mask_batch_stride, rowstart, tri, qlist, fbcount, todo);
            k_linear_local<LW, LT, true, false><<<dim3(cdiv(tiles, 4), B), 256, 0, s>>>(img, out, C, H, W, mask, mask_is_f32, mask_batch_stride, rowstart, tri, qlist, fbcount, todo);
        }
        k_linear_tri<<<dim3(cdiv(n, 4 * LQ), B), 256, 0, s>>>(img, out, C, H, W, sites, qlist, fbcount, tri, unresolved, bcount, blist);
    } else
    k_linear_tri<<<dim3(cdiv(n, 4 * LQ), B), 256, 0, s>>>(img, out, C, H, W, sites, qlist, counts, tri, unresolved, bcount, blist);
    k_linear_boundary<<<256, 64, 0, s>>>(img, out, C, H, W, sites, counts, bcount, blist, tri);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}
extern "C" int pdhip_linear_fill_unresolved(const void* ws, int B, int H, int W, int* out, void* stream) {
    PD_REQUIRE(ws && out, "pdhip_linear_fill_unresolved: null pointer");
    const int* p = reinterpret_cast<const int*>(reinterpret_cast<const float*>(ws) + (size_t)B * 3 * H * W) + (size_t)B * H * W + 2 * B;
    PD_HIP(hipMemcpyAsync(out, p, sizeof(int), hipMemcpyDeviceToHost, as_stream(stream)));
    PD_HIP(hipStreamSynchronize(as_stream(stream)));
    return PDHIP_OK;
}
