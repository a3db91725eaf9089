// Rows P1, P2b, P3 (+ demo.py:121-125 pixel coords): point / vertex projection into V views.
// Reference: pointdreamer/ours_utils.py:93-141 (P1), :153-202 (P3); demo.py:103-104 (P2b), :121-125.
// HBM-bound streaming kernels: coalesced xyz reads (12 B/pt), 8+4 B/pt writes per view.
// Compiled with -ffp-contract=off (arithmetic contract, see common.h).
#include "common.h"
using namespace pdhip;

// ---------------------------------------------------------------------------------------------
// P1 pass A: transform mesh vertices for every view, write pos = (x,y,z,1), reduce xy min/max.
// (vps = views per shape: view g = blockIdx.y belongs to shape g / vps and uses camera g % vps; one shape: vps = V)
__global__ void k_project_verts(const float* __restrict__ cams, const float* __restrict__ verts, int Vn,
                                float* __restrict__ pos, uint32_t* __restrict__ minmax, int single, int vps) {
    const int v = blockIdx.y;
    const Cam c = load_cam(cams + 16 * (v % vps));
    verts += (size_t)(v / vps) * 3 * (size_t)Vn;
    float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < Vn; i += gridDim.x * blockDim.x) {
        float x = verts[3 * i], y = verts[3 * i + 1], z = verts[3 * i + 2];
        float xn, yn, zn;
        cam_transform(c, x, y, z, xn, yn, zn);
        float4 o = make_float4(xn, yn, zn, 1.0f);
        reinterpret_cast<float4*>(pos)[(size_t)v * Vn + i] = o;
        mnx = fminf(mnx, xn); mxx = fmaxf(mxx, xn);
        mny = fminf(mny, yn); mxy = fmaxf(mxy, yn);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mnx = fminf(mnx, __shfl_xor(mnx, off)); mxx = fmaxf(mxx, __shfl_xor(mxx, off));
        mny = fminf(mny, __shfl_xor(mny, off)); mxy = fmaxf(mxy, __shfl_xor(mxy, off));
    }
    // one set of atomics per block (same-address atomics cost ~0.2 us each: per wave they were most of this kernel's 17 us)
    __shared__ float s_mm[16][4];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_mm[wave][0] = mnx; s_mm[wave][1] = mny; s_mm[wave][2] = mxx; s_mm[wave][3] = mxy; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int k = threadIdx.x, nw = (blockDim.x + 63) >> 6;
        float r = s_mm[0][k];
        for (int w = 1; w < nw; ++w) r = k < 2 ? fminf(r, s_mm[w][k]) : fmaxf(r, s_mm[w][k]);
        if (single) minmax[4 * v + k] = f2ord(r);          // one workgroup per view: its extrema are the view's (no init pass, no atomics)
        else if (k < 2) atomicMin(&minmax[4 * v + k], f2ord(r)); else atomicMax(&minmax[4 * v + k], f2ord(r));
    }
}

__global__ void k_init_minmax(uint32_t* minmax, int V) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 4 * V) minmax[i] = ((i & 3) < 2) ? 0xffffffffu : 0u;
}

// P1 pass B: rescale vertices (write back into pos.xy), transform + rescale points.
__global__ void k_project_finish(const float* __restrict__ cams, const uint32_t* __restrict__ minmax,
                                 const float* __restrict__ points, int N, int Vn, int rescale, float pad9,
                                 float* __restrict__ pos, float* __restrict__ vuv, float* __restrict__ uv_centers,
                                 float* __restrict__ uv_scales, float* __restrict__ puv, float* __restrict__ pdep, int vps) {
    const int v = blockIdx.y;
    points += (size_t)(v / vps) * 3 * (size_t)N;
    float cx = 0.f, cy = 0.f, sc = 2.f;
    if (rescale) {
        float mnx = ord2f(minmax[4 * v + 0]), mny = ord2f(minmax[4 * v + 1]);
        float mxx = ord2f(minmax[4 * v + 2]), mxy = ord2f(minmax[4 * v + 3]);
        cx = (mnx + mxx) / 2.0f;
        cy = (mny + mxy) / 2.0f;
        sc = fmaxf(mxx - mnx, mxy - mny);
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            uv_centers[2 * v] = cx; uv_centers[2 * v + 1] = cy; uv_scales[v] = sc;
        }
    }
    const Cam c = load_cam(cams + 16 * (v % vps));
    const int total = Vn + N;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        if (i < Vn) {
            float4* p4 = reinterpret_cast<float4*>(pos) + (size_t)v * Vn + i;
            float4 p = *p4;
            float u, w;
            if (rescale) {
                u = ((p.x - cx) / sc) * pad9 + 0.5f;
                w = ((p.y - cy) / sc) * pad9 + 0.5f;
                u = fminf(fmaxf(u, 0.f), 1.f);
                w = fminf(fmaxf(w, 0.f), 1.f);
                p.x = u * 2.0f - 1.0f;
                p.y = w * 2.0f - 1.0f;
                *p4 = p;
            } else {
                u = fminf(fmaxf((p.x + 1.0f) * 0.5f, 0.f), 1.f);
                w = fminf(fmaxf((p.y + 1.0f) * 0.5f, 0.f), 1.f);
            }
            vuv[2 * ((size_t)v * Vn + i)] = u;
            vuv[2 * ((size_t)v * Vn + i) + 1] = w;
        } else {
            int j = i - Vn;
            float xn, yn, zn;
            cam_transform(c, points[3 * j], points[3 * j + 1], points[3 * j + 2], xn, yn, zn);
            float u, w;
            if (rescale) {
                u = ((xn - cx) / sc) * pad9 + 0.5f;
                w = ((yn - cy) / sc) * pad9 + 0.5f;
            } else {
                u = (xn + 1.0f) * 0.5f;
                w = (yn + 1.0f) * 0.5f;
            }
            reinterpret_cast<float2*>(puv)[(size_t)v * N + j] = make_float2(u, w);
            pdep[(size_t)v * N + j] = zn;
        }
    }
}

static int project_impl(const float* cam_params, int V, int vps, const float* vertices, int Vn, const float* points, int N, int rescale,
                        double padding, float* pos, float* vertice_uvs, float* uv_centers, float* uv_scales, float* point_uvs,
                        float* point_depths, uint32_t* minmax_ws, void* stream);
extern "C" int pdhip_project_points(const float* cam_params, int V, const float* vertices, int Vn,
                                    const float* points, int N, int rescale, double padding, float* pos,
                                    float* vertice_uvs, float* uv_centers, float* uv_scales, float* point_uvs,
                                    float* point_depths, uint32_t* minmax_ws, void* stream) {
    return project_impl(cam_params, V, V > 0 ? V : 1, vertices, Vn, points, N, rescale, padding, pos, vertice_uvs, uv_centers, uv_scales, point_uvs,
                        point_depths, minmax_ws, stream);
}
// S shapes through the same V cameras in ONE set of launches: vertices [S,Vn,3], points [S,N,3]; every per-view output has S*V
// leading entries (view g = s * V + v), minmax_ws 4 * S * V words.
extern "C" int pdhip_project_points_shapes(const float* cam_params, int V, int S, const float* vertices, int Vn,
                                           const float* points, int N, int rescale, double padding, float* pos,
                                           float* vertice_uvs, float* uv_centers, float* uv_scales, float* point_uvs,
                                           float* point_depths, uint32_t* minmax_ws, void* stream) {
    PD_REQUIRE(S >= 1 && V >= 1, "pdhip_project_points_shapes: bad sizes");
    return project_impl(cam_params, S * V, V, vertices, Vn, points, N, rescale, padding, pos, vertice_uvs, uv_centers, uv_scales, point_uvs,
                        point_depths, minmax_ws, stream);
}
static int project_impl(const float* cam_params, int V, int vps, const float* vertices, int Vn, const float* points, int N, int rescale,
                        double padding, float* pos, float* vertice_uvs, float* uv_centers, float* uv_scales, float* point_uvs,
                        float* point_depths, uint32_t* minmax_ws, void* stream) {
    PD_REQUIRE(V > 0 && Vn > 0 && N >= 0, "pdhip_project_points: bad sizes V=%d Vn=%d N=%d", V, Vn, N);
    PD_REQUIRE(cam_params && vertices && pos && vertice_uvs && minmax_ws && (N == 0 || (points && point_uvs && point_depths)),
               "pdhip_project_points: null pointer");
    PD_REQUIRE(!rescale || (uv_centers && uv_scales), "pdhip_project_points: rescale needs uv_centers/uv_scales");
    hipStream_t s = as_stream(stream);
    if (Vn <= 65536) {                                     // one 1024-lane workgroup per view reduces its own extrema
        k_project_verts<<<dim3(1, V), 1024, 0, s>>>(cam_params, vertices, Vn, pos, minmax_ws, 1, vps);
    } else {
        k_init_minmax<<<cdiv(4 * V, 64), 64, 0, s>>>(minmax_ws, V);
        dim3 ga(min(cdiv(Vn, 256), 64), V);
        k_project_verts<<<ga, 256, 0, s>>>(cam_params, vertices, Vn, pos, minmax_ws, 0, vps);
    }
    dim3 gb(min(cdiv(Vn + N, 256), 256), V);
    const float pad9 = (float)(1.0 - 2.0 * padding);
    k_project_finish<<<gb, 256, 0, s>>>(cam_params, minmax_ws, points, N, Vn, rescale, pad9, pos, vertice_uvs,
                                        uv_centers, uv_scales, point_uvs, point_depths, vps);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// P3: depth-test gather.  visible = depth - mesh_depth[v,row,col] <= offset.
__global__ void k_point_visibility(int R, const float* __restrict__ uvs, const float* __restrict__ dep,
                                   const float* __restrict__ mesh, int N, float offset,
                                   uint8_t* __restrict__ vis, int64_t* __restrict__ pix) {
    const int v = blockIdx.y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        float2 uv = reinterpret_cast<const float2*>(uvs)[(size_t)v * N + i];
        int col = clip_to_int(uv.x * (float)R, R - 1);
        int row = clip_to_int(uv.y * (float)R, R - 1);
        float ref = mesh[((size_t)v * R + row) * R + col];
        float d = dep[(size_t)v * N + i];
        vis[(size_t)v * N + i] = ((d - ref) <= offset) ? 1 : 0;
        if (pix) {
            pix[2 * ((size_t)v * N + i)] = row;
            pix[2 * ((size_t)v * N + i) + 1] = col;
        }
    }
}

extern "C" int pdhip_point_visibility(int cam_res, const float* point_uvs, const float* point_depths,
                                      const float* mesh_depths, int V, int N, float offset, uint8_t* visibility,
                                      int64_t* point_pixels, void* stream) {
    PD_REQUIRE(V > 0 && N >= 0 && cam_res > 0, "pdhip_point_visibility: bad sizes");
    if (N == 0) return PDHIP_OK;
    PD_REQUIRE(point_uvs && point_depths && mesh_depths && visibility, "pdhip_point_visibility: null pointer");
    dim3 g(min(cdiv(N, 256), 1024), V);
    k_point_visibility<<<g, 256, 0, as_stream(stream)>>>(cam_res, point_uvs, point_depths, mesh_depths, N, offset,
                                                        visibility, point_pixels);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}

// P3 + demo.py:121-125 in one pass over the points (the pipeline's pair: the depth test at cam_res, the pixel coordinates at res)
__global__ void k_point_visibility_pixels(int R, const float* __restrict__ uvs, const float* __restrict__ dep,
                                          const float* __restrict__ mesh, int N, float offset, uint8_t* __restrict__ vis, int res,
                                          int64_t* __restrict__ pix) {
    const int v = blockIdx.y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const float2 uv = reinterpret_cast<const float2*>(uvs)[(size_t)v * N + i];
        const int col = clip_to_int(uv.x * (float)R, R - 1);
        const int row = clip_to_int(uv.y * (float)R, R - 1);
        const float ref = mesh[((size_t)v * R + row) * R + col];
        const float d = dep[(size_t)v * N + i];
        vis[(size_t)v * N + i] = ((d - ref) <= offset) ? 1 : 0;
        const float a = fminf(fmaxf(uv.x * (float)res, -1.0e9f), 1.0e9f);
        const float b = fminf(fmaxf(uv.y * (float)res, -1.0e9f), 1.0e9f);
        longlong2 o;
        o.x = min(max((int)b, 0), res - 1);                      // (row, col)
        o.y = min(max((int)a, 0), res - 1);
        reinterpret_cast<longlong2*>(pix)[(size_t)v * N + i] = o;
    }
}

extern "C" int pdhip_point_visibility_pixels(int cam_res, const float* point_uvs, const float* point_depths,
                                             const float* mesh_depths, int V, int N, float offset, uint8_t* visibility,
                                             int res, int64_t* point_pixels, void* stream) {
    PD_REQUIRE(V > 0 && N >= 0 && cam_res > 0 && res > 0, "pdhip_point_visibility_pixels: bad sizes");
    if (N == 0) return PDHIP_OK;
    PD_REQUIRE(point_uvs && point_depths && mesh_depths && visibility && point_pixels, "pdhip_point_visibility_pixels: null pointer");
    dim3 g(min(cdiv(N, 256), 1024), V);
    k_point_visibility_pixels<<<g, 256, 0, as_stream(stream)>>>(cam_res, point_uvs, point_depths, mesh_depths, N, offset,
                                                               visibility, res, point_pixels);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}

// demo.py:121-125: long(uv*res) (truncation toward zero), swap to (row,col), clip to [0,res-1].
__global__ void k_point_pixels(const float* __restrict__ uvs, long long total, int res, int64_t* __restrict__ pix) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        float2 uv = reinterpret_cast<const float2*>(uvs)[i];
        float a = fminf(fmaxf(uv.x * (float)res, -1.0e9f), 1.0e9f);
        float b = fminf(fmaxf(uv.y * (float)res, -1.0e9f), 1.0e9f);
        int col = min(max((int)a, 0), res - 1);
        int row = min(max((int)b, 0), res - 1);
        pix[2 * i] = row;
        pix[2 * i + 1] = col;
    }
}

extern "C" int pdhip_point_pixels(const float* point_uvs, int V, int N, int res, int64_t* point_pixels, void* stream) {
    PD_REQUIRE(V > 0 && N >= 0 && res > 0, "pdhip_point_pixels: bad sizes");
    if (N == 0) return PDHIP_OK;
    PD_REQUIRE(point_uvs && point_pixels, "pdhip_point_pixels: null pointer");
    long long total = (long long)V * N;
    k_point_pixels<<<min(cdiv(total, 256), 2048), 256, 0, as_stream(stream)>>>(point_uvs, total, res, point_pixels);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// P2b: bilinear (align_corners=False, no antialias) resize of a mask followed by != 0.
__device__ __forceinline__ void bilinear_taps(int n_in, int n_out, int d, int& i0, int& i1, bool& w0, bool& w1) {
    float scale = (float)n_in / (float)n_out;
    float src = scale * ((float)d + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    i0 = min((int)src, n_in - 1);
    i1 = min(i0 + 1, n_in - 1);
    float l1 = src - (float)i0;
    float l0 = 1.0f - l1;
    w0 = l0 > 0.f;
    w1 = l1 > 0.f;
}

__global__ void k_resize_mask(const uint8_t* __restrict__ in, int in_h, int in_w, uint8_t* __restrict__ out,
                              int out_h, int out_w) {
    const int b = blockIdx.y;
    const uint8_t* src = in + (size_t)b * in_h * in_w;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < out_h * out_w; idx += gridDim.x * blockDim.x) {
        int y = idx / out_w, x = idx - y * out_w;
        int r0, r1, c0, c1;
        bool wr0, wr1, wc0, wc1;
        bilinear_taps(in_h, out_h, y, r0, r1, wr0, wr1);
        bilinear_taps(in_w, out_w, x, c0, c1, wc0, wc1);
        bool o = (wr0 && wc0 && src[r0 * in_w + c0]) || (wr0 && wc1 && src[r0 * in_w + c1]) ||
                 (wr1 && wc0 && src[r1 * in_w + c0]) || (wr1 && wc1 && src[r1 * in_w + c1]);
        out[(size_t)b * out_h * out_w + idx] = o ? 1 : 0;
    }
}

extern "C" int pdhip_resize_mask(const uint8_t* in, int B, int in_h, int in_w, uint8_t* out, int out_h, int out_w,
                                 void* stream) {
    PD_REQUIRE(B > 0 && in_h > 0 && in_w > 0 && out_h > 0 && out_w > 0, "pdhip_resize_mask: bad sizes");
    PD_REQUIRE(in && out, "pdhip_resize_mask: null pointer");
    dim3 g(min(cdiv((long long)out_h * out_w, 256), 1024), B);
    k_resize_mask<<<g, 256, 0, as_stream(stream)>>>(in, in_h, in_w, out, out_h, out_w);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}
