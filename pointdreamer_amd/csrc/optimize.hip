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

// forward (bilinear lookup, clamp, L1) + backward (scatter sign/count to the 4 texels) for every masked pixel
__global__ __launch_bounds__(256) void k_optcolor_grad(const float* __restrict__ atlas, int A, const float* __restrict__ uv_map,
                                                       int res, const float* __restrict__ target,
                                                       const uint8_t* __restrict__ wmask, double inv_count,
                                                       double* __restrict__ grad, float* __restrict__ images) {
    const int v = blockIdx.y;
    const size_t plane = (size_t)A * A;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < res * res; idx += gridDim.x * blockDim.x) {
        const int y = idx / res, x = idx - y * res;
        const bool on = wmask[(size_t)v * res * res + idx];
        if (!on) {
            if (images) for (int c = 0; c < 3; ++c) images[(((size_t)v * 3 + c) * res + y) * res + x] = 0.f;
            continue;
        }
        const size_t src = ((size_t)v * res + (res - 1 - y)) * res + x;
        const double u = (double)uv_map[2 * src], w = (double)uv_map[2 * src + 1];
        // grid_sample unnormalise: gx = 2u-1, gy = -(2w-1); ix = ((g+1)/2)*A - 0.5; border padding = clamp to [0, A-1]
        double ix = (((u * 2.0 - 1.0) + 1.0) / 2.0) * A - 0.5, iy = ((-(w * 2.0 - 1.0) + 1.0) / 2.0) * A - 0.5;
        ix = fmin(fmax(ix, 0.0), (double)(A - 1));
        iy = fmin(fmax(iy, 0.0), (double)(A - 1));
        const int x0 = (int)floor(ix), y0 = (int)floor(iy), x1 = x0 + 1, y1 = y0 + 1;
        const double fx = ix - x0, fy = iy - y0;
        const double w00 = (1.0 - fx) * (1.0 - fy), w01 = fx * (1.0 - fy), w10 = (1.0 - fx) * fy, w11 = fx * fy;
        const bool bx = x1 < A, by = y1 < A;
        for (int c = 0; c < 3; ++c) {
            const float* at = atlas + (size_t)c * plane;
            double val = w00 * (double)at[(size_t)y0 * A + x0];
            if (bx) val += w01 * (double)at[(size_t)y0 * A + x1];
            if (by) val += w10 * (double)at[(size_t)y1 * A + x0];
            if (bx && by) val += w11 * (double)at[(size_t)y1 * A + x1];
            const bool pass = val >= 0.0 && val <= 1.0;                 // clamp backward is inclusive
            const double img = fmin(fmax(val, 0.0), 1.0);
            if (images) images[(((size_t)v * 3 + c) * res + y) * res + x] = (float)img;
            const double d = img - (double)target[(((size_t)v * 3 + c) * res + y) * res + x];
            if (!pass || d == 0.0) continue;
            const double g = (d > 0.0 ? inv_count : -inv_count);
            double* gr = grad + (size_t)c * plane;
            atomicAdd(&gr[(size_t)y0 * A + x0], w00 * g);
            if (bx) atomicAdd(&gr[(size_t)y0 * A + x1], w01 * g);
            if (by) atomicAdd(&gr[(size_t)y1 * A + x0], w10 * g);
            if (bx && by) atomicAdd(&gr[(size_t)y1 * A + x1], w11 * g);
        }
    }
}

// torch.optim.Adam (betas 0.9/0.999, eps 1e-8, no weight decay) single-tensor update order; clears the gradient
__global__ void k_optcolor_adam(float* __restrict__ param, double* __restrict__ grad, float* __restrict__ m,
                                float* __restrict__ vv, long long n, float step_size, float bc2_sqrt) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float g = (float)grad[i];
        grad[i] = 0.0;
        const float mi = m[i] + (g - m[i]) * (1.0f - 0.9f);
        const float vi = vv[i] * 0.999f + (g * g) * (1.0f - 0.999f);
        m[i] = mi; vv[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + 1e-8f;
        param[i] = param[i] + (-step_size) * (mi / denom);
    }
}

static size_t a256(size_t x) { return (x + 255) & ~(size_t)255; }
extern "C" size_t pdhip_optimize_color_ws_bytes(int V, int res, int A) {
    return a256((size_t)V * 3 * res * res * 4) + a256((size_t)V * res * res) + a256((size_t)3 * A * A * 8) + 2 * a256((size_t)3 * A * A * 4);
}

extern "C" int pdhip_optimize_color(float* atlas /*[3,A,A] in/out*/, int A, const float* uv_map, const int64_t* face_idxs, int V,
                                    int res, const float* inpainted, int r, const uint8_t* shrinked, double lr, int iterations,
                                    float* final_images /*[V,3,res,res] or NULL*/, void* ws, void* stream) {
    PD_REQUIRE(atlas && uv_map && face_idxs && inpainted && ws && A > 0 && V > 0 && res > 0 && r > 0 && iterations >= 0,
               "pdhip_optimize_color: bad arguments");
    hipStream_t s = as_stream(stream);
    char* p = reinterpret_cast<char*>(ws);
    float* target = reinterpret_cast<float*>(p); p += a256((size_t)V * 3 * res * res * 4);
    uint8_t* wmask = reinterpret_cast<uint8_t*>(p); p += a256((size_t)V * res * res);
    double* grad = reinterpret_cast<double*>(p); p += a256((size_t)3 * A * A * 8);
    float* m = reinterpret_cast<float*>(p); p += a256((size_t)3 * A * A * 4);
    float* vv = reinterpret_cast<float*>(p);
    const long long n = 3LL * A * A;
    PD_HIP(hipMemsetAsync(grad, 0, n * 8, s));
    PD_HIP(hipMemsetAsync(m, 0, n * 4, s));
    PD_HIP(hipMemsetAsync(vv, 0, n * 4, s));
    dim3 g(min(cdiv((long long)res * res, 256), 2048), V);
    k_optcolor_target<<<g, 256, 0, s>>>(inpainted, r, uv_map, face_idxs, res, shrinked, A, target, wmask);
    const double inv_count = 1.0 / ((double)V * 3.0 * res * res);
    for (int it = 0; it < iterations; ++it) {
        const bool last = it == iterations - 1;
        k_optcolor_grad<<<g, 256, 0, s>>>(atlas, A, uv_map, res, target, wmask, inv_count, grad, last ? final_images : nullptr);
        const int step = it + 1;
        const double cur_lr = lr * pow(0.5, (double)(it / 15));            // StepLR(step_size 15, gamma 0.5)
        const double bc1 = 1.0 - pow(0.9, step), bc2 = 1.0 - pow(0.999, step);
        k_optcolor_adam<<<min(cdiv(n, 256), 4096), 256, 0, s>>>(atlas, grad, m, vv, n, (float)(cur_lr / bc1), (float)sqrt(bc2));
    }
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}
