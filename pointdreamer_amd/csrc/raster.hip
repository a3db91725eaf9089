// Row P2: triangle-mesh rasteriser for V views at R x R (replaces nvdiffrast.rasterize at
// pointdreamer/ours_utils.py:142-147).  Exact rules are the build's own (oracle/project.py header):
// 1/256-pixel snapped vertices, int64 edge functions at pixel centres, watertight tie rule,
// float64 depth interpolation, nearest z wins with ties to the smaller face id.
// One 64-bit atomicMin per covered pixel on a (z-order, face) key; HBM/L2-atomic bound.
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
                               uint64_t* __restrict__ zkey) {
    const int v = blockIdx.y;
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

extern "C" int pdhip_raster_mesh(const float* pos, int V, int Vn, const int32_t* faces, int F, int R,
                                 uint64_t* zkey_ws, uint8_t* hard_masks, int64_t* face_idxs, float* depths,
                                 void* stream) {
    PD_REQUIRE(V > 0 && Vn > 0 && F >= 0 && R > 0 && R <= 16384, "pdhip_raster_mesh: bad sizes V=%d Vn=%d F=%d R=%d", V, Vn, F, R);
    PD_REQUIRE(pos && (F == 0 || faces) && zkey_ws && hard_masks && face_idxs && depths, "pdhip_raster_mesh: null pointer");
    hipStream_t s = as_stream(stream);
    long long n = (long long)V * R * R;
    k_raster_init<<<min(cdiv(n, 256), 4096), 256, 0, s>>>(zkey_ws, n);
    if (F > 0) {
        dim3 g(cdiv(F, 64), V);
        k_raster_faces<<<g, 64, 0, s>>>(pos, Vn, faces, F, R, zkey_ws);
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
