// optimize_color (pointdreamer/ours_utils.py:1583-1785) -- refine the atlas against the inpainted
// views: `iterations` Adam steps (lr 5e-2, StepLR(15, 0.5)) on an L1 loss between bilinear texture lookups (float64,
// kaolin texture_mapping == grid_sample(align_corners=False, padding 'border', v flipped)) and the views resized to
// res x res, masked by the foreground and by the shrunk per-view visibility.  The reference runs ~25 torch autograd
// kernels per iteration over [8,3,1024,1024] float64 tensors; here one fused forward+backward kernel scatters the
// gradient with f64 atomics and one Adam kernel updates the 3 A^2 texels -- HBM-bound on the uv/target streams.
#include "common.h"
using namespace pdhip;

// pos.xy <- clip((((xy - c)/s) * pad9) * f_v + 0.5, 0, 1) * 2 - 1     (ours_utils.py:1688-1695)
__global__ void k_rescale_vertices(float* __restrict__ pos, int Vn, const float* __restrict__ uv_centers,
                                   const float* __restrict__ uv_scales, const float* __restrict__ factors, float pad9) {
    const int v = blockIdx.y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < Vn; i += gridDim.x * blockDim.x) {
        float4* p4 = reinterpret_cast<float4*>(pos) + (size_t)v * Vn + i;
        float4 p = *p4;
        float u = (((p.x - uv_centers[2 * v]) / uv_scales[v]) * pad9) * factors[v] + 0.5f;
        float w = (((p.y - uv_centers[2 * v + 1]) / uv_scales[v]) * pad9) * factors[v] + 0.5f;
        u = fminf(fmaxf(u, 0.f), 1.f);
        w = fminf(fmaxf(w, 0.f), 1.f);
        p.x = u * 2.0f - 1.0f;
        p.y = w * 2.0f - 1.0f;
        *p4 = p;
    }
}

extern "C" int pdhip_rescale_vertices(float* pos, int V, int Vn, const float* uv_centers, const float* uv_scales,
                                      const float* factors, double padding, void* stream) {
    PD_REQUIRE(pos && uv_centers && uv_scales && factors && V > 0 && Vn > 0, "pdhip_rescale_vertices: bad arguments");
    dim3 g(min(cdiv(Vn, 256), 256), V);
    k_rescale_vertices<<<g, 256, 0, as_stream(stream)>>>(pos, Vn, uv_centers, uv_scales, factors, (float)(1.0 - 2.0 * padding));
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}

// target[v,c,y,x] = bilinear_resize(inpainted[v,c])(y,x) * fg * shr ;  wmask[v,y,x] = fg & shr
// (uv_map / face_idxs are the UNflipped raster outputs; image row y reads raster row res-1-y: ours_utils.py:1710-1713)
__global__ void k_optcolor_target(const float* __restrict__ inp, int r, const float* __restrict__ uv_map,
                                  const int64_t* __restrict__ fid, int res, const uint8_t* __restrict__ shr, int A,
                                  float* __restrict__ target, uint8_t* __restrict__ wmask) {
    const int v = blockIdx.y;
    const float scale = (float)r / (float)res;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < res * res; idx += gridDim.x * blockDim.x) {
        const int y = idx / res, x = idx - y * res;
        const size_t src = ((size_t)v * res + (res - 1 - y)) * res + x;
        bool on = fid[src] >= 0;
        if (on && shr != nullptr) {
            const int tx = clip_to_int(uv_map[2 * src] * (float)A, A - 1), ty = clip_to_int(uv_map[2 * src + 1] * (float)A, A - 1);
            on = shr[((size_t)v * A + ty) * A + tx] != 0;
        }
        wmask[(size_t)v * res * res + idx] = on ? 1 : 0;
        float sy = scale * ((float)y + 0.5f) - 0.5f; if (sy < 0.f) sy = 0.f;
        float sx = scale * ((float)x + 0.5f) - 0.5f; if (sx < 0.f) sx = 0.f;
        const int y0 = min((int)sy, r - 1), x0 = min((int)sx, r - 1);
        const int y1 = min(y0 + 1, r - 1), x1 = min(x0 + 1, r - 1);
        const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.0f - ly, hx = 1.0f - lx;
        for (int c = 0; c < 3; ++c) {
            const float* im = inp + ((size_t)v * 3 + c) * r * r;
            const float val = hy * (hx * im[y0 * r + x0] + lx * im[y0 * r + x1]) + ly * (hx * im[y1 * r + x0] + lx * im[y1 * r + x1]);
            target[(((size_t)v * 3 + c) * res + y) * res + x] = on ? val : 0.f;
        }
    }
}

// ---- gather formulation.  The texture coordinates are fixed for the whole optimisation, so the transpose of the bilinear
// sampling operator is built ONCE as a CSR table: texel -> (pixel, weight) contributions, sorted by pixel index inside a texel
// (deterministic summation order -- the reference's grid_sample backward / index_put atomics are not).  An iteration is then
//   forward : per masked pixel, bilinear lookup (f64), clamp, sign of the L1 residual per channel -> 3 bytes
//   backward: per texel, sum of weight * sign over its contributions (f64, fixed order) -> Adam update, fused
// ~0.75 GB of streaming traffic per iteration at V = 8, res = 1024 instead of ~100 M f64 atomics (2.3 ms -> see DESIGN.md).

// the (up to 4) texels a pixel reads and their float64 weights (kaolin texture_mapping == grid_sample(align_corners=False,
// padding 'border', v flipped)); returns the number of valid corners
__device__ __forceinline__ int oc_corners(const float* __restrict__ uv_map, size_t src, int A, int* tex /*[4]*/, double* wt /*[4]*/) {
    const double u = (double)uv_map[2 * src], w = (double)uv_map[2 * src + 1];
    // grid_sample unnormalise: gx = 2u-1, gy = -(2w-1); ix = ((g+1)/2)*A - 0.5; border padding = clamp to [0, A-1]
    double ix = (((u * 2.0 - 1.0) + 1.0) / 2.0) * A - 0.5, iy = ((-(w * 2.0 - 1.0) + 1.0) / 2.0) * A - 0.5;
    ix = fmin(fmax(ix, 0.0), (double)(A - 1));
    iy = fmin(fmax(iy, 0.0), (double)(A - 1));
    const int x0 = (int)floor(ix), y0 = (int)floor(iy), x1 = x0 + 1, y1 = y0 + 1;
    const double fx = ix - x0, fy = iy - y0;
    const bool bx = x1 < A, by = y1 < A;
    int n = 0;
    tex[n] = y0 * A + x0; wt[n++] = (1.0 - fx) * (1.0 - fy);
    if (bx) { tex[n] = y0 * A + x1; wt[n++] = fx * (1.0 - fy); }
    if (by) { tex[n] = y1 * A + x0; wt[n++] = (1.0 - fx) * fy; }
    if (bx && by) { tex[n] = y1 * A + x1; wt[n++] = fx * fy; }
    return n;
}

__global__ void k_oc_count(const float* __restrict__ uv_map, const uint8_t* __restrict__ wmask, int V, int res, int A,
                           int* __restrict__ cnt) {
    const long long total = (long long)V * res * res;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < total; p += (long long)gridDim.x * blockDim.x) {
        if (!wmask[p]) continue;
        const int v = (int)(p / ((long long)res * res)), idx = (int)(p - (long long)v * res * res);
        const int y = idx / res, x = idx - y * res;
        int tex[4]; double wt[4];
        const int n = oc_corners(uv_map, ((size_t)v * res + (res - 1 - y)) * res + x, A, tex, wt);
        for (int k = 0; k < n; ++k) atomicAdd(&cnt[tex[k]], 1);
    }
}

// exclusive prefix sum of n ints in three steps (block sums -> scan of the block sums -> add): n <= 1024 * 1024 * 4
__global__ void k_oc_scan1(const int* __restrict__ in, int n, int* __restrict__ out, int* __restrict__ bsum) {
    __shared__ int sh[1024];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    const int v = i < n ? in[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int t = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    if (i < n) out[i] = sh[threadIdx.x] - v;
    if (threadIdx.x == 1023) bsum[blockIdx.x] = sh[1023];
}
__global__ void k_oc_scan2(int* __restrict__ bsum, int nb) {          // one block: exclusive scan of <= 4096 block sums
    __shared__ int sh[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) sh[i] = i < nb ? bsum[i] : 0;
    __syncthreads();
    if (threadIdx.x == 0) { int run = 0; for (int i = 0; i < nb; ++i) { const int t = sh[i]; sh[i] = run; run += t; } }
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += blockDim.x) bsum[i] = sh[i];
}
__global__ void k_oc_scan3(int* __restrict__ out, int n, const int* __restrict__ bsum) {
    const int i = blockIdx.x * 1024 + threadIdx.x;
    if (i < n) out[i] += bsum[blockIdx.x];
}

__global__ void k_oc_fill(const float* __restrict__ uv_map, const uint8_t* __restrict__ wmask, int V, int res, int A,
                          const int* __restrict__ off, int* __restrict__ cursor, int* __restrict__ e_pix, double* __restrict__ e_w) {
    const long long total = (long long)V * res * res;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < total; p += (long long)gridDim.x * blockDim.x) {
        if (!wmask[p]) continue;
        const int v = (int)(p / ((long long)res * res)), idx = (int)(p - (long long)v * res * res);
        const int y = idx / res, x = idx - y * res;
        int tex[4]; double wt[4];
        const int n = oc_corners(uv_map, ((size_t)v * res + (res - 1 - y)) * res + x, A, tex, wt);
        for (int k = 0; k < n; ++k) {
            const int slot = off[tex[k]] + atomicAdd(&cursor[tex[k]], 1);
            e_pix[slot] = (int)p; e_w[slot] = wt[k];
        }
    }
}

// contributions of one texel in ascending pixel order (the cursor order above is a race): insertion sort, lists are short.
// One wave owns 64 consecutive texels, whose lists are one contiguous CSR range: the range is staged through LDS (coalesced
// loads and stores), every lane sorts its own list there.  Ranges longer than OC_SORT_CAP fall back to sorting in global memory.
#define OC_SORT_CAP 3072
__global__ __launch_bounds__(64) void k_oc_sort(const int* __restrict__ off, const int* __restrict__ cnt, int ntex,
                                                int* __restrict__ e_pix, double* __restrict__ e_w) {
    __shared__ int s_p[OC_SORT_CAP];
    __shared__ double s_w[OC_SORT_CAP];
    const int lane = threadIdx.x;
    for (int t0 = blockIdx.x * 64; t0 < ntex; t0 += gridDim.x * 64) {
        const int t = t0 + lane;
        const int b = t < ntex ? off[t] : 0, n = t < ntex ? cnt[t] : 0;
        const int tl = min(t0 + 63, ntex - 1);
        const int R0 = off[t0], R1 = off[tl] + cnt[tl], len = R1 - R0;
        if (len <= OC_SORT_CAP) {
            for (int i = lane; i < len; i += 64) { s_p[i] = e_pix[R0 + i]; s_w[i] = e_w[R0 + i]; }
            __syncthreads();
            const int lb = b - R0;
            for (int i = 1; i < n; ++i) {
                const int kp = s_p[lb + i]; const double kw = s_w[lb + i];
                int j = i - 1;
                while (j >= 0 && s_p[lb + j] > kp) { s_p[lb + j + 1] = s_p[lb + j]; s_w[lb + j + 1] = s_w[lb + j]; --j; }
                s_p[lb + j + 1] = kp; s_w[lb + j + 1] = kw;
            }
            __syncthreads();
            for (int i = lane; i < len; i += 64) { e_pix[R0 + i] = s_p[i]; e_w[R0 + i] = s_w[i]; }
            __syncthreads();
        } else {
            for (int i = 1; i < n; ++i) {
                const int kp = e_pix[b + i]; const double kw = e_w[b + i];
                int j = i - 1;
                while (j >= 0 && e_pix[b + j] > kp) { e_pix[b + j + 1] = e_pix[b + j]; e_w[b + j + 1] = e_w[b + j]; --j; }
                e_pix[b + j + 1] = kp; e_w[b + j + 1] = kw;
            }
        }
    }
}

// forward: sign of the L1 residual per masked pixel and channel (0 where the clamp or the residual kills the gradient)
// interleaved copy of the atlas being optimised (x, y, z = the three planes): the forward pass fetches a corner with one 16-byte
// gather instead of three 4-byte ones; the backward pass keeps it in step with the planar parameter
__global__ void k_oc_pack(const float* __restrict__ atlas, int ntex, float4* __restrict__ at4) {
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < ntex; t += gridDim.x * blockDim.x)
        at4[t] = make_float4(atlas[t], atlas[(size_t)ntex + t], atlas[2 * (size_t)ntex + t], 0.f);
}

__global__ __launch_bounds__(256) void k_oc_forward(const float4* __restrict__ at4, int A, const float* __restrict__ uv_map, int V,
                                                    int res, const float* __restrict__ target, const uint8_t* __restrict__ wmask,
                                                    int8_t* __restrict__ sgn /*[V*res*res][4]*/, float* __restrict__ images) {
    const long long total = (long long)V * res * res;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < total; p += (long long)gridDim.x * blockDim.x) {
        const int v = (int)(p / ((long long)res * res)), idx = (int)(p - (long long)v * res * res);
        const int y = idx / res, x = idx - y * res;
        if (!wmask[p]) {
            if (images) for (int c = 0; c < 3; ++c) images[(((size_t)v * 3 + c) * res + y) * res + x] = 0.f;
            continue;
        }
        int tex[4]; double wt[4];
        const int n = oc_corners(uv_map, ((size_t)v * res + (res - 1 - y)) * res + x, A, tex, wt);
        char4 sg = make_char4(0, 0, 0, 0);
        float4 cor[4];
        for (int k = 0; k < n; ++k) cor[k] = at4[tex[k]];
        for (int c = 0; c < 3; ++c) {
            double val = 0.0;
            for (int k = 0; k < n; ++k) val += wt[k] * (double)(c == 0 ? cor[k].x : (c == 1 ? cor[k].y : cor[k].z));
            const bool pass = val >= 0.0 && val <= 1.0;                 // clamp backward is inclusive
            const double img = fmin(fmax(val, 0.0), 1.0);
            if (images) images[(((size_t)v * 3 + c) * res + y) * res + x] = (float)img;
            const double d = img - (double)target[(((size_t)v * 3 + c) * res + y) * res + x];
            const signed char sc = (!pass || d == 0.0) ? 0 : (d > 0.0 ? 1 : -1);
            if (c == 0) sg.x = sc; else if (c == 1) sg.y = sc; else sg.z = sc;
        }
        reinterpret_cast<char4*>(sgn)[p] = sg;
    }
}

// backward + torch.optim.Adam (betas 0.9/0.999, eps 1e-8, no weight decay; single-tensor update order), per texel.
// One wave owns 64 consecutive texels = one contiguous CSR range.  The range is streamed through LDS in chunks: 64 lanes load
// (pixel, weight) coalesced and gather the pixels' sign bytes with 64 independent requests in flight, then every lane adds the
// entries of its own list that fall into the chunk, in list order (same f64 summation order as a plain per-texel walk).
#define OC_BW_CHUNK 1024
__global__ __launch_bounds__(64) void k_oc_backward_adam(const int* __restrict__ off, const int* __restrict__ cnt,
                                                         const int* __restrict__ e_pix, const double* __restrict__ e_w,
                                                         const int8_t* __restrict__ sgn, double inv_count, float* __restrict__ param,
                                                         float* __restrict__ m, float* __restrict__ vv, int ntex, float step_size,
                                                         float bc2_sqrt, float4* __restrict__ at4) {
    __shared__ double sw[OC_BW_CHUNK];                         // one wave per workgroup: the barriers below are wave-local
    __shared__ char4 ss[OC_BW_CHUNK];
    const int lane = threadIdx.x;
    for (int t0 = blockIdx.x * 64; t0 < ntex; t0 += gridDim.x * 64) {
        const int t = t0 + lane;
        const int b = t < ntex ? off[t] : 0, n = t < ntex ? cnt[t] : 0;
        const int tl = min(t0 + 63, ntex - 1);
        const int R0 = off[t0], R1 = off[tl] + cnt[tl];
        double g0 = 0.0, g1 = 0.0, g2 = 0.0;
        int cur = b;
        const int e = b + n;
        for (int c0 = R0; c0 < R1; c0 += OC_BW_CHUNK) {
            const int len = min(OC_BW_CHUNK, R1 - c0);
            for (int i = lane; i < len; i += 64) {
                const int px = e_pix[c0 + i];
                sw[i] = e_w[c0 + i];
                ss[i] = reinterpret_cast<const char4*>(sgn)[px];
            }
            __syncthreads();
            const int stop = min(e, c0 + len);
            for (; cur < stop; ++cur) {
                const char4 sg = ss[cur - c0];
                const double wi = sw[cur - c0] * inv_count;                 // = w * (+-1/count), as the scatter form adds it
                g0 += wi * (double)sg.x; g1 += wi * (double)sg.y; g2 += wi * (double)sg.z;
            }
            __syncthreads();
        }
        if (t < ntex) {
            const double gs[3] = {g0, g1, g2};
            float np[3];
            for (int c = 0; c < 3; ++c) {
                const size_t i = (size_t)c * ntex + t;
                const float g = (float)gs[c];
                const float mi = m[i] + (g - m[i]) * (1.0f - 0.9f);
                const float vi = vv[i] * 0.999f + (g * g) * (1.0f - 0.999f);
                m[i] = mi; vv[i] = vi;
                const float denom = sqrtf(vi) / bc2_sqrt + 1e-8f;
                np[c] = param[i] + (-step_size) * (mi / denom);
                param[i] = np[c];
            }
            at4[t] = make_float4(np[0], np[1], np[2], 0.f);
        }
    }
}

static size_t a256(size_t x) { return (x + 255) & ~(size_t)255; }
extern "C" size_t pdhip_optimize_color_ws_bytes(int V, int res, int A) {
    const size_t px = (size_t)V * res * res, tx = (size_t)A * A;
    return a256(px * 3 * 4) /*target*/ + a256(px) /*wmask*/ + a256(px * 4) /*sgn*/ + 2 * a256(tx * 3 * 4) /*m, v*/ +
           3 * a256((tx + 4096) * 4) /*cnt, off, cursor*/ + a256(4096 * 4) /*block sums*/ + a256(px * 4 * 4) /*e_pix*/ + a256(px * 4 * 8) /*e_w*/ +
           a256(tx * 16) /*interleaved atlas*/;
}

extern "C" int pdhip_optimize_color(float* atlas /*[3,A,A] in/out*/, int A, const float* uv_map, const int64_t* face_idxs, int V,
                                    int res, const float* inpainted, int r, const uint8_t* shrinked, double lr, int iterations,
                                    float* final_images /*[V,3,res,res] or NULL*/, void* ws, void* stream) {
    PD_REQUIRE(atlas && uv_map && face_idxs && inpainted && ws && A > 0 && V > 0 && res > 0 && r > 0 && iterations >= 0,
               "pdhip_optimize_color: bad arguments");
    PD_REQUIRE((long long)A * A <= 4096LL * 1024 && (long long)V * res * res * 4 < 0x7fffffffLL, "pdhip_optimize_color: atlas / view size too large");
    hipStream_t s = as_stream(stream);
    const size_t px = (size_t)V * res * res, tx = (size_t)A * A;
    char* p = reinterpret_cast<char*>(ws);
    float* target = reinterpret_cast<float*>(p); p += a256(px * 3 * 4);
    uint8_t* wmask = reinterpret_cast<uint8_t*>(p); p += a256(px);
    int8_t* sgn = reinterpret_cast<int8_t*>(p); p += a256(px * 4);
    float* m = reinterpret_cast<float*>(p); p += a256(tx * 3 * 4);
    float* vv = reinterpret_cast<float*>(p); p += a256(tx * 3 * 4);
    int* cnt = reinterpret_cast<int*>(p); p += a256((tx + 4096) * 4);
    int* off = reinterpret_cast<int*>(p); p += a256((tx + 4096) * 4);
    int* cursor = reinterpret_cast<int*>(p); p += a256((tx + 4096) * 4);
    int* bsum = reinterpret_cast<int*>(p); p += a256(4096 * 4);
    int* e_pix = reinterpret_cast<int*>(p); p += a256(px * 4 * 4);
    double* e_w = reinterpret_cast<double*>(p); p += a256(px * 4 * 8);
    float4* at4 = reinterpret_cast<float4*>(p);
    const long long n = 3LL * A * A;
    PD_HIP(hipMemsetAsync(m, 0, n * 4, s));
    PD_HIP(hipMemsetAsync(vv, 0, n * 4, s));
    PD_HIP(hipMemsetAsync(cnt, 0, tx * 4, s));
    PD_HIP(hipMemsetAsync(cursor, 0, tx * 4, s));
    dim3 g(min(cdiv((long long)res * res, 256), 2048), V);
    k_optcolor_target<<<g, 256, 0, s>>>(inpainted, r, uv_map, face_idxs, res, shrinked, A, target, wmask);
    const int gp = min(cdiv((long long)px, 256), 8192);
    const int nb = cdiv((long long)tx, 1024);
    k_oc_count<<<gp, 256, 0, s>>>(uv_map, wmask, V, res, A, cnt);
    k_oc_scan1<<<nb, 1024, 0, s>>>(cnt, (int)tx, off, bsum);
    k_oc_scan2<<<1, 1024, 0, s>>>(bsum, nb);
    k_oc_scan3<<<nb, 1024, 0, s>>>(off, (int)tx, bsum);
    k_oc_fill<<<gp, 256, 0, s>>>(uv_map, wmask, V, res, A, off, cursor, e_pix, e_w);
    k_oc_sort<<<min(cdiv((long long)tx, 64), 16384), 64, 0, s>>>(off, cnt, (int)tx, e_pix, e_w);
    k_oc_pack<<<min(cdiv((long long)tx, 256), 4096), 256, 0, s>>>(atlas, (int)tx, at4);
    const double inv_count = 1.0 / ((double)V * 3.0 * res * res);
    for (int it = 0; it < iterations; ++it) {
        const bool last = it == iterations - 1;
        k_oc_forward<<<gp, 256, 0, s>>>(at4, A, uv_map, V, res, target, wmask, sgn, last ? final_images : nullptr);
        const int step = it + 1;
        const double cur_lr = lr * pow(0.5, (double)(it / 15));            // StepLR(step_size 15, gamma 0.5)
        const double bc1 = 1.0 - pow(0.9, step), bc2 = 1.0 - pow(0.999, step);
        k_oc_backward_adam<<<min(cdiv((long long)tx, 64), 16384), 64, 0, s>>>(off, cnt, e_pix, e_w, sgn, inv_count, atlas, m, vv, (int)tx, (float)(cur_lr / bc1),
                                              (float)sqrt(bc2), at4);
    }
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}
