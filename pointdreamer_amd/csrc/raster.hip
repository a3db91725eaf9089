// Row P2: triangle-mesh rasteriser for V views at R x R (replaces nvdiffrast.rasterize at
// pointdreamer/ours_utils.py:142-147).  Exact rules are the build's own (oracle/project.py header):
// 1/256-pixel snapped vertices, int64 edge functions at pixel centres, watertight tie rule,
// float64 depth interpolation, nearest z wins with ties to the smaller face id.
// Default path: LDS-tiled -- per-(view, face) setup once, 32x32-pixel tiles bin the faces whose bounding box touches them
// (ballot compaction into LDS) and every thread resolves its 4 pixels against the binned faces in registers: no global
// atomics, no z-key buffer, outputs written once.  Meshes with more than 65 536 faces fall back to one 64-bit atomicMin per
// covered pixel on a (z-order, face) key.  Both give identical images (min over the same keys).
// Compiled with -ffp-contract=off.
#include "common.h"
using namespace pdhip;

#define SUBPIX 256
#define FIX_CLAMP (1 << 24)

__device__ __forceinline__ long long snap_fix(float ndc, int R) {
    float v = (ndc * 0.5f + 0.5f) * (float)(R * SUBPIX);
    if (!(fabsf(v) <= 3.0e38f)) v = 0.f;           // NaN / inf -> 0
    v = rintf(v);
    v = fminf(fmaxf(v, (float)(-FIX_CLAMP)), (float)FIX_CLAMP);
    return (long long)v;
}

__device__ __forceinline__ long long floor_div(long long a, long long b) {   // b > 0
    long long q = a / b;
    return (a % b != 0 && a < 0) ? q - 1 : q;
}

__global__ void k_raster_init(uint64_t* zkey, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        zkey[i] = ~0ull;
}

// one thread per (view, face); bbox loop with 64-bit atomicMin.
__global__ void k_raster_faces(const float* __restrict__ pos, int Vn, const int32_t* __restrict__ faces, int F, int R,
                               uint64_t* __restrict__ zkey, int vps) {
    const int v = blockIdx.y;
    faces += (size_t)(v / vps) * 3 * (size_t)F;            // several shapes per call: view v draws the mesh of shape v / vps
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const float4* P = reinterpret_cast<const float4*>(pos) + (size_t)v * Vn;
    int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
    if ((unsigned)i0 >= (unsigned)Vn || (unsigned)i1 >= (unsigned)Vn || (unsigned)i2 >= (unsigned)Vn) return;
    float4 a = P[i0], b = P[i1], c = P[i2];
    long long x0 = snap_fix(a.x, R), y0 = snap_fix(a.y, R);
    long long x1 = snap_fix(b.x, R), y1 = snap_fix(b.y, R);
    long long x2 = snap_fix(c.x, R), y2 = snap_fix(c.y, R);
    double z0 = (double)a.z, z1 = (double)b.z, z2 = (double)c.z;
    long long area = (x1 - x0) * (y2 - y0) - (y1 - y0) * (x2 - x0);
    if (area == 0) return;
    if (area < 0) {
        long long t;
        t = x1; x1 = x2; x2 = t;
        t = y1; y1 = y2; y2 = t;
        double tz = z1; z1 = z2; z2 = tz;
        area = -area;
    }
    long long minx = min(x0, min(x1, x2)), maxx = max(x0, max(x1, x2));
    long long miny = min(y0, min(y1, y2)), maxy = max(y0, max(y1, y2));
    int jmin = (int)max(0ll, -floor_div(-(minx - 128), SUBPIX));
    int jmax = (int)min((long long)R - 1, floor_div(maxx - 128, SUBPIX));
    int imin = (int)max(0ll, -floor_div(-(miny - 128), SUBPIX));
    int imax = (int)min((long long)R - 1, floor_div(maxy - 128, SUBPIX));
    if (jmin > jmax || imin > imax) return;
    // edge e_k(p) = dx*(py-ay) - dy*(px-ax); weight of v0 <- edge v1->v2, v1 <- v2->v0, v2 <- v0->v1
    const long long dx0 = x2 - x1, dy0 = y2 - y1;
    const long long dx1 = x0 - x2, dy1 = y0 - y2;
    const long long dx2 = x1 - x0, dy2 = y1 - y0;
    const bool inc0 = (dy0 > 0) || (dy0 == 0 && dx0 > 0);
    const bool inc1 = (dy1 > 0) || (dy1 == 0 && dx1 > 0);
    const bool inc2 = (dy2 > 0) || (dy2 == 0 && dx2 > 0);
    const double darea = (double)area;
    uint64_t* zk = zkey + (size_t)v * R * R;
    for (int i = imin; i <= imax; ++i) {
        const long long py = (long long)i * SUBPIX + 128;
        for (int j = jmin; j <= jmax; ++j) {
            const long long px = (long long)j * SUBPIX + 128;
            long long E0 = dx0 * (py - y1) - dy0 * (px - x1);
            long long E1 = dx1 * (py - y2) - dy1 * (px - x2);
            long long E2 = dx2 * (py - y0) - dy2 * (px - x0);
            bool in = (E0 > 0 || (E0 == 0 && inc0)) && (E1 > 0 || (E1 == 0 && inc1)) && (E2 > 0 || (E2 == 0 && inc2));
            if (!in) continue;
            double zd = ((double)E0 * z0 + (double)E1 * z1) + (double)E2 * z2;
            float z = (float)(zd / darea);
            if (!(z >= -1.0f && z <= 1.0f)) continue;
            uint64_t key = ((uint64_t)f2ord(z) << 32) | (uint32_t)f;
            atomicMin(reinterpret_cast<unsigned long long*>(&zk[(size_t)i * R + j]), (unsigned long long)key);
        }
    }
}

// ---- LDS-tiled path
struct __attribute__((aligned(16))) FaceSetup {     // edge functions as exact doubles (integers below 2^52), winding made positive
    double e0, e1, e2;             // edge functions at the centre of pixel (0, 0), biased by the tie rule (see k_raster_setup)
    double ax0, ax1, ax2;          // step per +8 pixel columns
    double ay0, ay1, ay2;          // step per +8 pixel rows
    double z0, z1, z2, darea, rinv;   // rinv = 1 / darea (correctly rounded)
    short jmin, jmax, imin, imax;  // pixel bounding box, empty (jmin > jmax) for culled / degenerate faces
    int inc, pad;                  // bit k of inc = tie rule of edge k
};

__global__ void k_raster_setup(const float* __restrict__ pos, int Vn, const int32_t* __restrict__ faces, int F, int R,
                               FaceSetup* __restrict__ setup, short4* __restrict__ bbox, int vps) {
    const int v = blockIdx.y;
    faces += (size_t)(v / vps) * 3 * (size_t)F;            // several shapes per call: view v draws the mesh of shape v / vps
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    FaceSetup fs;
    fs.e0 = fs.e1 = fs.e2 = fs.ax0 = fs.ax1 = fs.ax2 = fs.ay0 = fs.ay1 = fs.ay2 = 0.0; fs.jmin = 1; fs.jmax = 0; fs.imin = 1; fs.imax = 0; fs.inc = 0; fs.pad = 0;
    fs.z0 = fs.z1 = fs.z2 = 0.0; fs.darea = 1.0; fs.rinv = 1.0;
    const float4* P = reinterpret_cast<const float4*>(pos) + (size_t)v * Vn;
    const int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
    if ((unsigned)i0 < (unsigned)Vn && (unsigned)i1 < (unsigned)Vn && (unsigned)i2 < (unsigned)Vn) {
        const float4 a = P[i0], b = P[i1], c = P[i2];
        long long x0 = snap_fix(a.x, R), y0 = snap_fix(a.y, R);
        long long x1 = snap_fix(b.x, R), y1 = snap_fix(b.y, R);
        long long x2 = snap_fix(c.x, R), y2 = snap_fix(c.y, R);
        double z0 = (double)a.z, z1 = (double)b.z, z2 = (double)c.z;
        long long area = (x1 - x0) * (y2 - y0) - (y1 - y0) * (x2 - x0);
        if (area != 0) {
            if (area < 0) {
                long long t;
                t = x1; x1 = x2; x2 = t;
                t = y1; y1 = y2; y2 = t;
                double tz = z1; z1 = z2; z2 = tz;
                area = -area;
            }
            const long long minx = min(x0, min(x1, x2)), maxx = max(x0, max(x1, x2));
            const long long miny = min(y0, min(y1, y2)), maxy = max(y0, max(y1, y2));
            // (clamped to [0, R] / [-1, R - 1] before the 16-bit store: an off-screen face stays an empty box)
            fs.jmin = (short)min((long long)R, max(0ll, -floor_div(-(minx - 128), SUBPIX)));
            fs.jmax = (short)max(-1ll, min((long long)R - 1, floor_div(maxx - 128, SUBPIX)));
            fs.imin = (short)min((long long)R, max(0ll, -floor_div(-(miny - 128), SUBPIX)));
            fs.imax = (short)max(-1ll, min((long long)R - 1, floor_div(maxy - 128, SUBPIX)));
            const long long dx0 = x2 - x1, dy0 = y2 - y1, dx1 = x0 - x2, dy1 = y0 - y2, dx2 = x1 - x0, dy2 = y1 - y0;
            fs.inc = (((dy0 > 0) || (dy0 == 0 && dx0 > 0)) ? 1 : 0) | (((dy1 > 0) || (dy1 == 0 && dx1 > 0)) ? 2 : 0) |
                     (((dy2 > 0) || (dy2 == 0 && dx2 > 0)) ? 4 : 0);
            // E_k(j, i) = E_k(0, 0) + j ax_k + i ay_k at pixel centres (256 j + 128, 256 i + 128): every term an integer below 2^52.
            // Stored: the steps per EIGHT pixels, and e_k = E_k(0, 0) + t_k - 1 (t_k = tie rule of edge k): a pixel is inside
            // edge k iff E_k > 0 or (E_k = 0 and t_k), i.e. iff the biased value is >= 0 -- a sign-bit test.
            fs.e0 = (double)(dx0 * (128 - y1) - dy0 * (128 - x1) + (fs.inc & 1) - 1);
            fs.e1 = (double)(dx1 * (128 - y2) - dy1 * (128 - x2) + ((fs.inc >> 1) & 1) - 1);
            fs.e2 = (double)(dx2 * (128 - y0) - dy2 * (128 - x0) + ((fs.inc >> 2) & 1) - 1);
            fs.ax0 = (double)(-dy0 * SUBPIX * 8); fs.ax1 = (double)(-dy1 * SUBPIX * 8); fs.ax2 = (double)(-dy2 * SUBPIX * 8);
            fs.ay0 = (double)(dx0 * SUBPIX * 8); fs.ay1 = (double)(dx1 * SUBPIX * 8); fs.ay2 = (double)(dx2 * SUBPIX * 8);
            fs.z0 = z0; fs.z1 = z1; fs.z2 = z2; fs.darea = (double)area; fs.rinv = 1.0 / (double)area;
            if (fs.imin > fs.imax) { fs.jmin = 1; fs.jmax = 0; }
        }
    }
    setup[(size_t)v * F + f] = fs;
    bbox[(size_t)v * F + f] = make_short4(fs.jmin, fs.jmax, fs.imin, fs.imax);
}

#ifndef RT
#define RT 64                      // tile edge (pixels); the tile's z-key buffer lives in LDS
#endif
#ifndef RNT
#define RNT 1024                   // threads per tile
#endif
#ifndef RWPE
#define RWPE 8                     // waves per SIMD the register allocation has to admit: TWO tiles per CU.  Unconstrained, the compiler
#endif                             // takes 106 SGPRs = 6 waves per SIMD = one 16-wave tile per CU (P1+P2 131 us; 12-wave tiles, two per CU:
                                   // 108-111; 16-wave tiles at 78 SGPRs, two per CU: 105)
#define RNW (RNT / 64)
// One workgroup per (view, 64x64 tile).  The faces whose bounding box touches the tile are compacted into an LDS list
// (ballot + per-wave counts, four faces per thread and step); every wave then takes faces off the list: the face's setup record
// is wave-uniform (scalar loads), its biased edge functions at the clipped bounding box's corner are three exact FMAs on the
// record, and the box is covered in 8x8 lane blocks, stepping the edge functions by additions, testing coverage on their sign
// bits and depth-testing with 64-bit LDS atomicMin on the same (z-order, face) key as the fallback path.
// Measured on 8 views x 9 800 faces x 512^2 (100 k face-tile pairs, 256 k lane blocks; 64 us): clear + binning + write-out 12 us,
// the record loads 13 us, the rest is VALU work whose slowest CU (two tiles: up to 2.6x the mean tile) sets the time.
// History: 98 us with one resident tile per CU (106 SGPRs) and the tile as the fastest grid index; 71 us with two tiles per CU
// and the view as the fastest index; 66 us with the sign-bit coverage test; 64 us with the reciprocal depth.  A staging pass that
// re-derived tile-local records from the snapped integer vertices, 256 faces at a time behind two barriers, was no faster.
__global__ __launch_bounds__(RNT, RWPE) void k_raster_tiles(const FaceSetup* __restrict__ setup, const short4* __restrict__ bbox, int F, int V,
                                                       int R, uint8_t* __restrict__ hard, int64_t* __restrict__ fid,
                                                       float* __restrict__ depth) {
    // (rows 72 words apart: at 64, the eight rows of an 8x8 lane block fall on the same 16 banks -- an 8-way conflict on every atomic)
    constexpr int RTS = RT + 8;
    __shared__ unsigned long long s_z[RT * RTS];
    __shared__ unsigned short s_list[RNT + 4 * RNT];              // (this path takes meshes of at most 65 536 faces)
    __shared__ int s_wcnt[RNW];
    // view = fastest index of the 1-D grid: workgroup b runs on XCD b % 8, so every XCD gets tiles of every screen position (with the
    // tile as the fastest index an XCD received one tile COLUMN of all views -- the empty border columns or the full centre ones)
    const int v = blockIdx.x % V, tile = blockIdx.x / V, tiles_x = (R + RT - 1) / RT;
    const int tx0 = (tile % tiles_x) * RT, ty0 = (tile / tiles_x) * RT;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const FaceSetup* S = setup + (size_t)v * F;
    const short4* B = bbox + (size_t)v * F;
    for (int k = t; k < RT * RTS; k += RNT) s_z[k] = ~0ull;
    int count = 0;                                                   // faces waiting in s_list (block-uniform)
    const double lx = (double)(lane & 7) * 0.125, ly = (double)(lane >> 3) * 0.125;

    auto resolve = [&](int n) {                                      // rasterise faces s_list[0 .. n) into s_z
        // (one L2 round trip per record, covered by the other seven waves of the SIMD: requesting the next record one face
        // ahead costs 32 more SGPRs and, with them, the second resident tile)
#ifdef PD_LAB_RASTER_NORESOLVE
        n = 0;
#endif
        for (int k = wave; k < n; k += RNW) {
            const int fidx = __builtin_amdgcn_readfirstlane((int)s_list[k]);
            const FaceSetup q = S[fidx];
#ifdef PD_LAB_RASTER_LOADONLY
            if (q.darea == 123.456) s_z[0] = 0;
            continue;
#endif
            const int j0 = max((int)q.jmin, tx0), j1 = min((int)q.jmax, tx0 + RT - 1);
            const int i0 = max((int)q.imin, ty0), i1 = min((int)q.imax, ty0 + RT - 1);
            const int nbx = (j1 - j0 + 8) >> 3, nby = (i1 - i0 + 8) >> 3;
            const int inc = q.inc;
            const double sx0 = q.ax0, sx1 = q.ax1, sx2 = q.ax2, sy0 = q.ay0, sy1 = q.ay1, sy2 = q.ay2;
            const double z0 = q.z0, z1 = q.z1, z2 = q.z2, darea = q.darea;
            const double px = fma((double)j0, 0.125, lx), py = fma((double)i0, 0.125, ly);     // in units of eight pixels (exact)
            double r0 = fma(py, sy0, fma(px, sx0, q.e0));            // exact: integers below 2^52
            double r1 = fma(py, sy1, fma(px, sx1, q.e1));
            double r2 = fma(py, sy2, fma(px, sx2, q.e2));
            const double b0 = (inc & 1) ? 0.0 : 1.0, b1 = (inc & 2) ? 0.0 : 1.0, b2 = (inc & 4) ? 0.0 : 1.0;   // E_k = biased + b_k
#ifndef PD_LAB_RASTER_DIV
            const double rinv = q.rinv;
#endif
            for (int by = 0; by < nby; ++by) {
                double E0 = r0, E1 = r1, E2 = r2;
                const int i = i0 + by * 8 + (lane >> 3);
                for (int bx = 0; bx < nbx; ++bx) {
                    const int j = j0 + bx * 8 + (lane & 7);
                    const bool in = j <= j1 && i <= i1 && (__double2hiint(E0) | __double2hiint(E1) | __double2hiint(E2)) >= 0;
                    if (in) {
                        const double zd = ((E0 + b0) * z0 + (E1 + b1) * z1) + (E2 + b2) * z2;
#ifdef PD_LAB_RASTER_DIV
                        const float z = (float)(zd / darea);
#else
                        // z = float(zd / darea) without the division: zd * (1 / darea) is within 2.5 ulp of the correctly rounded
                        // quotient, so both round to the same float unless a float rounding boundary (low 29 mantissa bits =
                        // 2^28) lies that close -- then, and below the float normal range, divide.
                        const double qa = zd * rinv;
                        const bool near = (unsigned)((__double2loint(qa) & 0x1fffffff) - (0x10000000 - 8)) <= 16u || !(fabs(qa) >= 0x1p-120);
                        float z = (float)qa;
                        if (__ballot(near) != 0ull) {                                  // (a real, wave-uniform branch: the empty asm keeps the compiler
                            asm volatile("" ::: "memory");                             // from turning it into a select that divides every time)
                            z = (float)(zd / darea);
                        }
#endif
                        if (z >= -1.0f && z <= 1.0f)
                            atomicMin(&s_z[(i - ty0) * RTS + (j - tx0)], ((unsigned long long)f2ord(z) << 32) | (uint32_t)fidx);
                    }
                    E0 += sx0; E1 += sx1; E2 += sx2;
                }
                r0 += sy0; r1 += sy1; r2 += sy2;
            }
        }
        __syncthreads();
    };

    // binning: four faces per thread and step (one block-wide compaction per 4096 faces instead of per 1024)
    for (int base = 0; base < F; base += 4 * RNT) {
        bool ov[4];
        int wtot = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int f = base + u * RNT + t;
            ov[u] = false;
            if (f < F) {
                const short4 bb = B[f];                                        // (jmin, jmax, imin, imax)
                ov[u] = bb.x <= bb.y && bb.y >= tx0 && bb.x < tx0 + RT && bb.w >= ty0 && bb.z < ty0 + RT;
            }
        }
        unsigned long long bal[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { bal[u] = __ballot(ov[u]); wtot += __popcll(bal[u]); }
        if (lane == 0) s_wcnt[wave] = wtot;
        __syncthreads();
        int off = count, tot = 0;
#pragma unroll
        for (int w = 0; w < RNW; ++w) {
            const int c = s_wcnt[w];
            off += w < wave ? c : 0;
            tot += c;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (ov[u]) s_list[off + __popcll(bal[u] & ((1ull << lane) - 1ull))] = (unsigned short)(base + u * RNT + t);
            off += __popcll(bal[u]);
        }
        count += tot;
        __syncthreads();
        if (count >= RNT) { resolve(count); count = 0; }
    }
    if (count > 0) resolve(count);
    __syncthreads();
    for (int k = t; k < RT * RT; k += RNT) {
        const int i = ty0 + k / RT, j = tx0 + k % RT;
        if (i >= R || j >= R) continue;
        const size_t o = ((size_t)v * R + i) * R + j;
        const unsigned long long key = s_z[(k / RT) * RTS + k % RT];
        const bool hit = key != ~0ull;
        hard[o] = hit ? 1 : 0;
        fid[o] = hit ? (int64_t)(uint32_t)(key & 0xffffffffu) : -1;
        depth[o] = hit ? ord2f((uint32_t)(key >> 32)) : 0.0f;
    }
}

__global__ void k_raster_resolve(const uint64_t* __restrict__ zkey, long long n, uint8_t* __restrict__ hard,
                                 int64_t* __restrict__ fid, float* __restrict__ depth) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        uint64_t k = zkey[i];
        bool hit = k != ~0ull;
        hard[i] = hit ? 1 : 0;
        fid[i] = hit ? (int64_t)(uint32_t)(k & 0xffffffffu) : -1;
        depth[i] = hit ? ord2f((uint32_t)(k >> 32)) : 0.0f;
    }
}

static thread_local int g_raster_path = 0;                    // 0 automatic, 1 force the global-atomic fallback
extern "C" int pdhip_debug_set_raster_path(int path) { int old = g_raster_path; g_raster_path = path; return old; }

/* workspace of the LDS-tiled path for a mesh of F faces: the face setups (128 B) + clipped boxes (8 B) of every view, or the z keys
 * of the atomic fallback, whichever is larger.  pdhip_raster_mesh() assumes the historical V*R*R*8 bytes, i.e. takes the tiled path
 * up to F <= R*R / 17 faces (15.4 k at R = 512); pdhip_raster_mesh_ws() with a workspace of this size takes it up to 65 536 faces. */
extern "C" size_t pdhip_raster_mesh_ws_bytes(int V, int F, int R) {
    const size_t zk = (size_t)V * R * R * sizeof(uint64_t), fs = (size_t)V * (size_t)std::min(F, 65536) * (sizeof(FaceSetup) + sizeof(short4));
    return std::max(zk, fs);
}
extern "C" int pdhip_raster_mesh_ws(const float* pos, int V, int Vn, const int32_t* faces, int F, int R, uint64_t* zkey_ws, size_t ws_bytes,
                                    uint8_t* hard_masks, int64_t* face_idxs, float* depths, void* stream);
extern "C" int pdhip_raster_mesh(const float* pos, int V, int Vn, const int32_t* faces, int F, int R,
                                 uint64_t* zkey_ws, uint8_t* hard_masks, int64_t* face_idxs, float* depths,
                                 void* stream) {
    return pdhip_raster_mesh_ws(pos, V, Vn, faces, F, R, zkey_ws, (size_t)V * R * R * sizeof(uint64_t), hard_masks, face_idxs, depths, stream);
}
static int raster_impl(const float* pos, int V, int vps, int Vn, const int32_t* faces, int F, int R, uint64_t* zkey_ws, size_t ws_bytes,
                       uint8_t* hard_masks, int64_t* face_idxs, float* depths, void* stream);
extern "C" int pdhip_raster_mesh_ws(const float* pos, int V, int Vn, const int32_t* faces, int F, int R, uint64_t* zkey_ws, size_t ws_bytes,
                                    uint8_t* hard_masks, int64_t* face_idxs, float* depths, void* stream) {
    return raster_impl(pos, V, V > 0 ? V : 1, Vn, faces, F, R, zkey_ws, ws_bytes, hard_masks, face_idxs, depths, stream);
}
// S meshes of Vn vertices / F faces each, V views per mesh, in ONE set of launches: pos [S*V,Vn,4] (pdhip_project_points_shapes),
// faces [S,F,3]; outputs [S*V,R,R]; workspace pdhip_raster_mesh_ws_bytes(S * V, F, R).
extern "C" int pdhip_raster_mesh_shapes(const float* pos, int V, int S, int Vn, const int32_t* faces, int F, int R, uint64_t* zkey_ws,
                                        size_t ws_bytes, uint8_t* hard_masks, int64_t* face_idxs, float* depths, void* stream) {
    PD_REQUIRE(S >= 1 && V >= 1, "pdhip_raster_mesh_shapes: bad sizes");
    return raster_impl(pos, S * V, V, Vn, faces, F, R, zkey_ws, ws_bytes, hard_masks, face_idxs, depths, stream);
}
static int raster_impl(const float* pos, int V, int vps, int Vn, const int32_t* faces, int F, int R, uint64_t* zkey_ws, size_t ws_bytes,
                       uint8_t* hard_masks, int64_t* face_idxs, float* depths, void* stream) {
    PD_REQUIRE(V > 0 && Vn > 0 && F >= 0 && R > 0 && R <= 16384, "pdhip_raster_mesh: bad sizes V=%d Vn=%d F=%d R=%d", V, Vn, F, R);
    PD_REQUIRE(pos && (F == 0 || faces) && zkey_ws && hard_masks && face_idxs && depths, "pdhip_raster_mesh: null pointer");
    hipStream_t s = as_stream(stream);
    long long n = (long long)V * R * R;
    PD_REQUIRE(ws_bytes >= (size_t)n * sizeof(uint64_t), "pdhip_raster_mesh: workspace smaller than V*R*R*8 bytes");
    // LDS-tiled path: the face setups (128 B + 8 B box per face and view) live in the workspace
    if (g_raster_path != 1 && F > 0 && F <= 65536 && R <= 32767 && (size_t)V * F * (sizeof(FaceSetup) + sizeof(short4)) <= ws_bytes) {
        FaceSetup* setup = reinterpret_cast<FaceSetup*>(zkey_ws);
        short4* bbox = reinterpret_cast<short4*>(setup + (size_t)V * F);
        k_raster_setup<<<dim3(cdiv(F, 256), V), 256, 0, s>>>(pos, Vn, faces, F, R, setup, bbox, vps);
        const int tiles = cdiv(R, RT) * cdiv(R, RT);
        k_raster_tiles<<<tiles * V, RNT, 0, s>>>(setup, bbox, F, V, R, hard_masks, face_idxs, depths);
        PD_LAUNCH_CHECK();
        return PDHIP_OK;
    }
    k_raster_init<<<min(cdiv(n, 256), 4096), 256, 0, s>>>(zkey_ws, n);
    if (F > 0) {
        dim3 g(cdiv(F, 64), V);
        k_raster_faces<<<g, 64, 0, s>>>(pos, Vn, faces, F, R, zkey_ws, vps);
    }
    k_raster_resolve<<<min(cdiv(n, 256), 4096), 256, 0, s>>>(zkey_ws, n, hard_masks, face_idxs, depths);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Barycentrics + attribute interpolation: the parts of nvdiffrast's rasterize()/interpolate() contract that the UV-atlas
// producer (models/get3d/extract_texture_map.py:57-63) and optimize_color (pointdreamer/ours_utils.py:1700-1705) use.
// bary[...,0:2] = weights (u, v) of the triangle's vertices 0 and 1 at the pixel centre (third weight 1-u-v), recomputed
// from the same snapped int64 edge functions that decided coverage; out = (u*a0 + v*a1) + ((1-u)-v)*a2 in float32.
__global__ void k_raster_bary(const float* __restrict__ pos, int Vn, const int32_t* __restrict__ faces, int R,
                              const int64_t* __restrict__ fid, float* __restrict__ bary, long long n) {
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const long long f = fid[idx];
        float u = 0.f, w = 0.f;
        if (f >= 0) {
            const int v = (int)(idx / ((long long)R * R));
            const int rem = (int)(idx - (long long)v * R * R);
            const int i = rem / R, j = rem - i * R;
            const float4* P = reinterpret_cast<const float4*>(pos) + (size_t)v * Vn;
            const float4 a = P[faces[3 * f]], b = P[faces[3 * f + 1]], c = P[faces[3 * f + 2]];
            const long long x0 = snap_fix(a.x, R), y0 = snap_fix(a.y, R), x1 = snap_fix(b.x, R), y1 = snap_fix(b.y, R);
            const long long x2 = snap_fix(c.x, R), y2 = snap_fix(c.y, R);
            const long long px = (long long)j * SUBPIX + 128, py = (long long)i * SUBPIX + 128;
            const long long area = (x1 - x0) * (y2 - y0) - (y1 - y0) * (x2 - x0);
            const long long E0 = (x2 - x1) * (py - y1) - (y2 - y1) * (px - x1);
            const long long E1 = (x0 - x2) * (py - y2) - (y0 - y2) * (px - x2);
            u = (float)((double)E0 / (double)area);
            w = (float)((double)E1 / (double)area);
        }
        bary[2 * idx] = u;
        bary[2 * idx + 1] = w;
    }
}

extern "C" int pdhip_raster_barycentrics(const float* pos, int V, int Vn, const int32_t* faces, int R,
                                         const int64_t* face_idxs, float* bary, void* stream) {
    PD_REQUIRE(V > 0 && Vn > 0 && R > 0 && pos && faces && face_idxs && bary, "pdhip_raster_barycentrics: bad arguments");
    const long long n = (long long)V * R * R;
    k_raster_bary<<<min(cdiv(n, 256), 4096), 256, 0, as_stream(stream)>>>(pos, Vn, faces, R, face_idxs, bary, n);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}

__global__ void k_interpolate(const float* __restrict__ attr, int C, const int32_t* __restrict__ tri,
                              const int64_t* __restrict__ fid, const float* __restrict__ bary, float* __restrict__ out,
                              long long n) {
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const long long f = fid[idx];
        if (f < 0) {
            for (int c = 0; c < C; ++c) out[idx * C + c] = 0.f;
            continue;
        }
        const float u = bary[2 * idx], v = bary[2 * idx + 1], w = (1.0f - u) - v;
        const float* a0 = attr + (size_t)tri[3 * f] * C;
        const float* a1 = attr + (size_t)tri[3 * f + 1] * C;
        const float* a2 = attr + (size_t)tri[3 * f + 2] * C;
        for (int c = 0; c < C; ++c) out[idx * C + c] = (u * a0[c] + v * a1[c]) + w * a2[c];
    }
}

extern "C" int pdhip_interpolate(const float* attr, int C, const int32_t* tri, const int64_t* face_idxs, const float* bary,
                                 long long pixels, float* out, void* stream) {
    PD_REQUIRE(attr && tri && face_idxs && bary && out && C > 0 && pixels > 0, "pdhip_interpolate: bad arguments");
    k_interpolate<<<min(cdiv(pixels, 256), 4096), 256, 0, as_stream(stream)>>>(attr, C, tri, face_idxs, bary, out, pixels);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}
