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

// ---- gather formulation on lists SORTED BY BASE TEXEL (round 4; round 3 kept a CSR table texel -> (pixel, corner) entries).
// The texture coordinates and the masks are fixed for the whole optimisation, so everything an iteration needs is laid out once:
//   * the masked pixels, counting-sorted by the texel their bilinear footprint starts at (`base` = top-left texel; inside a bucket
//     ascending pixel index: a deterministic f64 summation order -- the reference's grid_sample backward / index_put atomics are
//     not), as structure-of-arrays: fxy[i] = the bilinear fractions (exactly representable in f32: ix = u A - 0.5 with a 24-bit u
//     and A = 2^k leaves at most 24 significant bits below the binary point, out-of-range coordinates are clamped to integers),
//     tgt4[i] = (target r, g, b, base texel), sgn[i] = the three L1 signs of the current iteration in one byte, boff[t] = first
//     pixel of bucket t;
//   * forward = a coalesced stream over fxy / tgt4 (24 bytes per pixel) + four 16-byte gathers from the interleaved atlas, which
//     neighbouring threads share (sorted by base texel), writing ONE byte per pixel;
//   * backward + Adam = one wave per 64 consecutive texels.  Texel t receives corner 0 of bucket t, corner 1 of bucket t - 1,
//     corner 2 of bucket t - A and corner 3 of bucket t - A - 1: two CONTIGUOUS pixel ranges per wave ([t0 - 1, t0 + 63] and the
//     same one row up) streamed through LDS -- 9 bytes per pixel, each pixel read by exactly two waves.  No entry table (round 3:
//     4 bytes per corner + a 16-byte record gather per corner = 80 + 320 MB of requests per iteration at V = 8, res = 1024), no
//     active-texel records; texels without a contribution are skipped (zero gradient and zero Adam moments for ever: their update
//     is exactly + 0), so they and their 36 bytes of optimiser state per iteration are never touched.
// Per iteration at V = 8, res = 1024, A = 1024 (5.2 M masked pixels): ~130 MB forward + ~135 MB backward out of a ~190 MB working set
// (inside the 256 MB Infinity Cache; round 3: ~460 MB out of ~290 MB).  Timings: DESIGN.md.

// the top-left texel and the fractions of a pixel (kaolin texture_mapping == grid_sample(align_corners=False, padding 'border', v
// flipped)): gx = 2u-1, gy = -(2w-1); ix = ((g+1)/2)*A - 0.5; border padding = clamp to [0, A-1]
__device__ __forceinline__ void oc_base(const float* __restrict__ uv_map, size_t src, int A, int* tex, double* fx, double* fy) {
    const double u = (double)uv_map[2 * src], w = (double)uv_map[2 * src + 1];
    double ix = (((u * 2.0 - 1.0) + 1.0) / 2.0) * A - 0.5, iy = ((-(w * 2.0 - 1.0) + 1.0) / 2.0) * A - 0.5;
    ix = fmin(fmax(ix, 0.0), (double)(A - 1));
    iy = fmin(fmax(iy, 0.0), (double)(A - 1));
    const int x0 = (int)floor(ix), y0 = (int)floor(iy);
    *tex = y0 * A + x0; *fx = ix - x0; *fy = iy - y0;
}
// corner c (0 = (x0, y0), 1 = (x1, y0), 2 = (x0, y1), 3 = (x1, y1)) of a record: valid?, texel, float64 weight
__device__ __forceinline__ bool oc_corner(int tex, double fx, double fy, int A, int c, int* t, double* wt) {
    const int x0 = tex % A, y0 = tex / A;
    const bool bx = x0 + 1 < A, by = y0 + 1 < A;
    if (c == 0) { *t = tex; *wt = (1.0 - fx) * (1.0 - fy); return true; }
    if (c == 1) { *t = tex + 1; *wt = fx * (1.0 - fy); return bx; }
    if (c == 2) { *t = tex + A; *wt = (1.0 - fx) * fy; return by; }
    *t = tex + A + 1; *wt = fx * fy; return bx && by;
}

// exclusive prefix sum of n ints in three steps (block sums -> scan of the block sums -> add), four elements per thread:
// n <= 16384 * 4096 = 64 Mi (20 views of 1024^2 are 20 Mi pixels -- the 20-view camera distributions of demo.py).  total (may be
// null) receives the sum of all elements.  Integer sums: the result does not depend on the partition.
constexpr int OC_SCAN_EPT = 4, OC_SCAN_BLK = 1024 * OC_SCAN_EPT, OC_SCAN_MAXB = 16384;
__global__ void k_oc_scan1(const int* __restrict__ in, int n, int* __restrict__ out, int* __restrict__ bsum) {
    __shared__ int sh[1024];
    const long long i0 = (long long)blockIdx.x * OC_SCAN_BLK + threadIdx.x * OC_SCAN_EPT;
    int v[OC_SCAN_EPT], s4 = 0;
#pragma unroll
    for (int j = 0; j < OC_SCAN_EPT; ++j) { v[j] = i0 + j < n ? in[i0 + j] : 0; s4 += v[j]; }
    sh[threadIdx.x] = s4;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int t = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    int run = sh[threadIdx.x] - s4;
#pragma unroll
    for (int j = 0; j < OC_SCAN_EPT; ++j) { if (i0 + j < n) out[i0 + j] = run; run += v[j]; }
    if (threadIdx.x == 1023) bsum[blockIdx.x] = sh[1023];
}
__global__ void k_oc_scan2(int* __restrict__ bsum, int nb, int* __restrict__ total) {   // one block: exclusive scan of <= 16384 block sums
    __shared__ int sh[OC_SCAN_MAXB];
    for (int i = threadIdx.x; i < OC_SCAN_MAXB; i += blockDim.x) sh[i] = i < nb ? bsum[i] : 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < nb; ++i) { const int t = sh[i]; sh[i] = run; run += t; }
        if (total != nullptr) *total = run;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += blockDim.x) bsum[i] = sh[i];
}
__global__ void k_oc_scan3(int* __restrict__ out, int n, const int* __restrict__ bsum) {
    const long long i0 = (long long)blockIdx.x * OC_SCAN_BLK + threadIdx.x * OC_SCAN_EPT;
    const int add = bsum[blockIdx.x];
#pragma unroll
    for (int j = 0; j < OC_SCAN_EPT; ++j)
        if (i0 + j < n) out[i0 + j] += add;
}
#define OC_TRY(expr) do { int rc_ = (expr); if (rc_ != PDHIP_OK) return rc_; } while (0)
static int oc_scan(const int* in, int n, int* out, int* bsum, int* total, hipStream_t s) {
    const int nb = cdiv(n, OC_SCAN_BLK);
    PD_REQUIRE(nb <= OC_SCAN_MAXB, "optimize_color: scan of %d elements", n);
    k_oc_scan1<<<nb, 1024, 0, s>>>(in, n, out, bsum);
    k_oc_scan2<<<1, 1024, 0, s>>>(bsum, nb, total);
    k_oc_scan3<<<nb, 1024, 0, s>>>(out, n, bsum);
    return PDHIP_OK;
}


#define OC_FW_PX 4                                    // pixels per forward thread (sign bytes stored as one word)
// masked pixel -> its base texel's bucket count
__global__ void k_oc_count(const float* __restrict__ uv_map, const uint8_t* __restrict__ wmask, int V, int res, int A, int* __restrict__ cntb) {
    const long long total = (long long)V * res * res;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < total; p += (long long)gridDim.x * blockDim.x) {
        if (!wmask[p]) continue;
        const int v = (int)(p / ((long long)res * res)), idx = (int)(p - (long long)v * res * res);
        const int y = idx / res, x = idx - y * res;
        int tex; double fx, fy;
        oc_base(uv_map, ((size_t)v * res + (res - 1 - y)) * res + x, A, &tex, &fx, &fy);
        atomicAdd(&cntb[tex], 1);
    }
}
// ... and into the bucket (arrival order: a race, repaired by k_oc_sortb)
__global__ void k_oc_scatter(const float* __restrict__ uv_map, const uint8_t* __restrict__ wmask, int V, int res, int A,
                             const int* __restrict__ boff, int* __restrict__ cursor, int* __restrict__ pix) {
    const long long total = (long long)V * res * res;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < total; p += (long long)gridDim.x * blockDim.x) {
        if (!wmask[p]) continue;
        const int v = (int)(p / ((long long)res * res)), idx = (int)(p - (long long)v * res * res);
        const int y = idx / res, x = idx - y * res;
        int tex; double fx, fy;
        oc_base(uv_map, ((size_t)v * res + (res - 1 - y)) * res + x, A, &tex, &fx, &fy);
        pix[boff[tex] + atomicAdd(&cursor[tex], 1)] = (int)p;
    }
}
// pixels of one bucket in ascending order: insertion sort, buckets are short.  One wave owns 64 consecutive texels, whose buckets are
// one contiguous range staged through LDS; longer ranges are sorted in global memory.
#define OC_SORT_CAP 4096
__global__ __launch_bounds__(64) void k_oc_sortb(const int* __restrict__ boff, int ntex, int* __restrict__ pix) {
    __shared__ int s_e[OC_SORT_CAP];
    const int lane = threadIdx.x;
    for (int t0 = blockIdx.x * 64; t0 < ntex; t0 += gridDim.x * 64) {
        const int t = min(t0 + lane, ntex - 1);
        const int b = boff[t], n = t0 + lane < ntex ? boff[t + 1] - b : 0;
        const int R0 = boff[t0], R1 = boff[min(t0 + 64, ntex)], len = R1 - R0;
        if (len <= OC_SORT_CAP) {
            for (int i = lane; i < len; i += 64) s_e[i] = pix[R0 + i];
            __syncthreads();
            const int lb = b - R0;
            for (int i = 1; i < n; ++i) {
                const int k = s_e[lb + i];
                int j = i - 1;
                while (j >= 0 && s_e[lb + j] > k) { s_e[lb + j + 1] = s_e[lb + j]; --j; }
                s_e[lb + j + 1] = k;
            }
            __syncthreads();
            for (int i = lane; i < len; i += 64) pix[R0 + i] = s_e[i];
            __syncthreads();
        } else {
            for (int i = 1; i < n; ++i) {
                const int k = pix[b + i];
                int j = i - 1;
                while (j >= 0 && pix[b + j] > k) { pix[b + j + 1] = pix[b + j]; --j; }
                pix[b + j + 1] = k;
            }
        }
    }
}
// sorted position i -> fractions, targets + base texel
__global__ void k_oc_records(const int* __restrict__ pix, const int* __restrict__ npix, const float* __restrict__ uv_map,
                             const float* __restrict__ target, int res, int A, float2* __restrict__ fxy, float4* __restrict__ tgt4) {
    const int n = *npix;
    const size_t plane = (size_t)res * res;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int p = pix[i];
        const int v = (int)(p / plane), idx = (int)(p - (size_t)v * plane);
        const int y = idx / res, x = idx - y * res;
        int tex; double fx, fy;
        oc_base(uv_map, ((size_t)v * res + (res - 1 - y)) * res + x, A, &tex, &fx, &fy);
        fxy[i] = make_float2((float)fx, (float)fy);
        const size_t o = (size_t)v * 3 * plane + idx;
        tgt4[i] = make_float4(target[o], target[o + plane], target[o + 2 * plane], __int_as_float(tex));
    }
    // the forward pass reads whole groups of OC_FW_PX records: the tail group's padding must be finite and in range (base texel 0)
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid < OC_FW_PX) { fxy[n + gid] = make_float2(0.f, 0.f); tgt4[n + gid] = make_float4(0.f, 0.f, 0.f, __int_as_float(0)); }
}
// texels that receive a contribution (tools / tests: counters[1])
__global__ void k_oc_nactive(const int* __restrict__ boff, int A, int ntex, int* __restrict__ nact) {
    int c = 0;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < ntex; t += gridDim.x * blockDim.x) {
        const int x = t % A, y = t / A;
        bool on = boff[t + 1] > boff[t];
        if (x >= 1) on = on || boff[t] > boff[t - 1];
        if (y >= 1) on = on || boff[t - A + 1] > boff[t - A];
        if (x >= 1 && y >= 1) on = on || boff[t - A] > boff[t - A - 1];
        c += on ? 1 : 0;
    }
    __shared__ int s_c[256];
    s_c[threadIdx.x] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) s_c[threadIdx.x] += s_c[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0 && s_c[0]) atomicAdd(nact, s_c[0]);              // (one atomic per block; integer sum: any order)
}

// interleaved copy of the atlas being optimised (x, y, z = the three planes): the forward pass fetches a corner with one 16-byte
// gather instead of three 4-byte ones; the backward pass keeps it in step with the planar parameter
__global__ void k_oc_pack(const float* __restrict__ atlas, int ntex, float4* __restrict__ at4) {
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < ntex; t += gridDim.x * blockDim.x)
        at4[t] = make_float4(atlas[t], atlas[(size_t)ntex + t], atlas[2 * (size_t)ntex + t], 0.f);
}

// forward: bilinear lookup (f64), clamp, sign of the L1 residual per channel (0 where the clamp or the residual kills the gradient).
// sign byte: 2 bits per channel, code = sign + 1.  A thread takes OC_FW_PX consecutive pixels (sorted by base texel: its gathers and
// its neighbours' hit the same lines) and stores their sign bytes as one word.
__global__ __launch_bounds__(256) void k_oc_forward(const float4* __restrict__ at4, int A, const float2* __restrict__ fxy,
                                                    const float4* __restrict__ tgt4, const int* __restrict__ npix,
                                                    uint8_t* __restrict__ sgn, const int* __restrict__ pix_of, int res,
                                                    float* __restrict__ images) {
    const int n = *npix;
    for (int i0 = (blockIdx.x * blockDim.x + threadIdx.x) * OC_FW_PX; i0 < n; i0 += gridDim.x * blockDim.x * OC_FW_PX) {
        float2 f[OC_FW_PX]; float4 tg[OC_FW_PX]; int tex[OC_FW_PX][4]; double wt[OC_FW_PX][4]; bool ok[OC_FW_PX][4]; float4 cor[OC_FW_PX][4];
        // (the arrays are padded to a multiple of OC_FW_PX records: the vector loads below stay inside the allocation)
        {
            const float4 a = reinterpret_cast<const float4*>(fxy + i0)[0], b = reinterpret_cast<const float4*>(fxy + i0)[1];
            f[0] = make_float2(a.x, a.y); f[1] = make_float2(a.z, a.w); f[2] = make_float2(b.x, b.y); f[3] = make_float2(b.z, b.w);
        }
#pragma unroll
        for (int q = 0; q < OC_FW_PX; ++q) tg[q] = tgt4[i0 + q];
#pragma unroll
        for (int q = 0; q < OC_FW_PX; ++q) {
            const int tb = i0 + q < n ? __float_as_int(tg[q].w) : 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                ok[q][c] = oc_corner(tb, (double)f[q].x, (double)f[q].y, A, c, &tex[q][c], &wt[q][c]);
                cor[q][c] = at4[ok[q][c] ? tex[q][c] : tb];
            }
        }
        unsigned word = 0;
#pragma unroll
        for (int q = 0; q < OC_FW_PX; ++q) {
            const int i = i0 + q;
            unsigned code = 0;
            float im[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                double val = 0.0;
#pragma unroll
                for (int k = 0; k < 4; ++k) if (ok[q][k]) val += wt[q][k] * (double)(c == 0 ? cor[q][k].x : (c == 1 ? cor[q][k].y : cor[q][k].z));
                const bool pass = val >= 0.0 && val <= 1.0;                 // clamp backward is inclusive
                const double img = fmin(fmax(val, 0.0), 1.0);
                im[c] = (float)img;
                const double d = img - (double)(c == 0 ? tg[q].x : (c == 1 ? tg[q].y : tg[q].z));
                const unsigned sc = (!pass || d == 0.0) ? 1u : (d > 0.0 ? 2u : 0u);
                code |= sc << (2 * c);
            }
            word |= code << (8 * q);
            if (images && i < n) {                                          // last iteration: the final render (masked-out pixels stay 0)
                const int p = pix_of[i];
                const size_t plane = (size_t)res * res;
                const int v = (int)(p / plane);
                const size_t o = (size_t)v * 3 * plane + (p - (size_t)v * plane);
                images[o] = im[0]; images[o + plane] = im[1]; images[o + 2 * plane] = im[2];
            }
        }
        *reinterpret_cast<unsigned*>(sgn + i0) = word;
    }
}

// backward + torch.optim.Adam (betas 0.9/0.999, eps 1e-8, no weight decay; single-tensor update order).  One wave owns the 64
// consecutive texels [t0, t0 + 64).  Lane t sums, in this order: bucket t - A - 1 (corner 3), bucket t - A (corner 2) -- the TOP range,
// buckets [t0 - A - 1, t0 - A + 63] of the wave -- then bucket t - 1 (corner 1) and bucket t (corner 0) -- the CURRENT range
// [t0 - 1, t0 + 63]; a corner is skipped where the reference's border rule drops it (x0 + 1 == A, y0 + 1 == A).  Each range is one
// contiguous run of the sorted pixel arrays, streamed through LDS in chunks: 64 lanes load fractions and sign bytes coalesced, then
// every lane adds the pixels of its own two buckets that fall into the chunk, in sorted order.
#define OC_BW_CHUNK 512
__global__ __launch_bounds__(64) void k_oc_backward_adam(const int* __restrict__ boff, int A, int ntex, const float2* __restrict__ fxy,
                                                         const uint8_t* __restrict__ sgn, double inv_count, float* __restrict__ param,
                                                         float* __restrict__ m, float* __restrict__ vv, float step_size, float bc2_sqrt,
                                                         float4* __restrict__ at4) {
    __shared__ float2 sf[OC_BW_CHUNK];                         // one wave per workgroup: the barriers below are wave-local
    __shared__ uint8_t ss[OC_BW_CHUNK];
    const int lane = threadIdx.x;
    for (int t0 = blockIdx.x * 64; t0 < ntex; t0 += gridDim.x * 64) {
        const int t = t0 + lane;
        const bool in = t < ntex;
        const int tc = in ? t : ntex - 1;
        const int x = tc % A, y = tc / A;
        // lane spans [lo, hi) with a split: below the split the pixels contribute their odd corner (1 / 3: bucket one texel to the left)
        int lo[2], sp[2], hi[2];                                // [0] = top range, [1] = current range
        {
            const int bt = boff[tc], bt1 = boff[tc + 1];
            sp[1] = bt; hi[1] = in ? bt1 : bt; lo[1] = (in && x >= 1) ? boff[tc - 1] : bt;
            if (in && y >= 1) {
                const int u = tc - A;
                const int bu = boff[u], bu1 = boff[u + 1];
                sp[0] = bu; hi[0] = bu1; lo[0] = x >= 1 ? boff[u - 1] : bu;
            } else { lo[0] = sp[0] = hi[0] = 0; }
        }
        const bool any = hi[0] > lo[0] || hi[1] > lo[1];
        if (!__any(any)) continue;
        // the optimiser state of this lane's texel: requested now, used after the gradient is summed
        float pm[3] = {0.f, 0.f, 0.f}, pv[3] = {0.f, 0.f, 0.f}, pp[3] = {0.f, 0.f, 0.f};
        if (any) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const size_t i = (size_t)c * ntex + t;
                pm[c] = m[i]; pv[c] = vv[i]; pp[c] = param[i];
            }
        }
        double g0 = 0.0, g1 = 0.0, g2 = 0.0;
#pragma unroll
        for (int rg = 0; rg < 2; ++rg) {
            // the wave's range: from the first lane's lower bound to the last lane's upper bound (uniform)
            int R0, R1;
            if (rg == 0) {
                if (t0 + 63 < A) continue;                      // no lane has a row above
                const int ua = max(t0 - A - 1, 0), ub = min(t0 - A + 64, ntex);
                R0 = boff[ua]; R1 = boff[max(ub, 0)];
            } else {
                R0 = boff[max(t0 - 1, 0)]; R1 = boff[min(t0 + 64, ntex)];
            }
            int cur = lo[rg];
            const int e = hi[rg], split = sp[rg];
            const int c_lo = rg == 0 ? 3 : 1, c_hi = rg == 0 ? 2 : 0;
            for (int c0 = R0; c0 < R1; c0 += OC_BW_CHUNK) {
                const int len = min(OC_BW_CHUNK, R1 - c0);
                for (int i = lane; i < len; i += 64) { sf[i] = fxy[c0 + i]; ss[i] = sgn[c0 + i]; }
                __syncthreads();
                const int stop = min(e, c0 + len);
                for (cur = max(cur, min(c0, e)); cur < stop; ++cur) {
                    const float2 f = sf[cur - c0];
                    const unsigned code = ss[cur - c0];
                    const int c = cur < split ? c_lo : c_hi;
                    const double fx = (double)f.x, fy = (double)f.y;
                    const double wi = ((c & 1 ? fx : 1.0 - fx) * (c & 2 ? fy : 1.0 - fy)) * inv_count;   // = w * (+-1/count), as the scatter form adds it
                    g0 += wi * (double)((int)(code & 3u) - 1); g1 += wi * (double)((int)((code >> 2) & 3u) - 1); g2 += wi * (double)((int)((code >> 4) & 3u) - 1);
                }
                __syncthreads();
            }
        }
        if (any) {
            const double gs[3] = {g0, g1, g2};
            float np[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const size_t i = (size_t)c * ntex + t;
                const float g = (float)gs[c];
                const float mi = pm[c] + (g - pm[c]) * (1.0f - 0.9f);
                const float vi = pv[c] * 0.999f + (g * g) * (1.0f - 0.999f);
                m[i] = mi; vv[i] = vi;
                const float denom = sqrtf(vi) / bc2_sqrt + 1e-8f;
                np[c] = pp[c] + (-step_size) * (mi / denom);
                param[i] = np[c];
            }
            at4[t] = make_float4(np[0], np[1], np[2], 0.f);
        }
    }
}

static size_t a256(size_t x) { return (x + 255) & ~(size_t)255; }
extern "C" size_t pdhip_optimize_color_ws_bytes(int V, int res, int A) {
    const size_t px = (size_t)V * res * res, tx = (size_t)A * A;
    return a256(px * 3 * 4) /*target*/ + a256(px) /*wmask*/ + a256((px + 64) * 4) /*sorted pixel ids*/ + a256((px + 64) * 8) /*fractions*/ +
           a256((px + 64) * 16) /*targets + base texel*/ + a256(px + 64) /*sign bytes*/ + 2 * a256(tx * 3 * 4) /*m, v*/ +
           3 * a256((tx + 4096) * 4) /*bucket counts, offsets, cursor*/ + a256(16384 * 4) /*block sums*/ + a256(tx * 16) /*interleaved atlas*/ +
           256 /*counters*/;
}

extern "C" int pdhip_optimize_color(float* atlas /*[3,A,A] in/out*/, int A, const float* uv_map, const int64_t* face_idxs, int V,
                                    int res, const float* inpainted, int r, const uint8_t* shrinked, double lr, int iterations,
                                    float* final_images /*[V,3,res,res] or NULL*/, void* ws, void* stream) {
    PD_REQUIRE(atlas && uv_map && face_idxs && inpainted && ws && A > 0 && V > 0 && res > 0 && r > 0 && iterations >= 0,
               "pdhip_optimize_color: bad arguments");
    PD_REQUIRE((long long)A * A <= 4096LL * 1024 && (long long)V * res * res <= (1LL << 30),
               "pdhip_optimize_color: atlas / view size too large (A^2 <= 4 Mi texels, V res^2 <= 1 Gi pixels)");
    hipStream_t s = as_stream(stream);
    const size_t px = (size_t)V * res * res, tx = (size_t)A * A;
    char* p = reinterpret_cast<char*>(ws);
    float* target = reinterpret_cast<float*>(p); p += a256(px * 3 * 4);
    uint8_t* wmask = reinterpret_cast<uint8_t*>(p); p += a256(px);
    int* pix = reinterpret_cast<int*>(p); p += a256((px + 64) * 4);
    float2* fxy = reinterpret_cast<float2*>(p); p += a256((px + 64) * 8);
    float4* tgt4 = reinterpret_cast<float4*>(p); p += a256((px + 64) * 16);
    uint8_t* sgn = reinterpret_cast<uint8_t*>(p); p += a256(px + 64);
    float* m = reinterpret_cast<float*>(p); p += a256(tx * 3 * 4);
    float* vv = reinterpret_cast<float*>(p); p += a256(tx * 3 * 4);
    int* cntb = reinterpret_cast<int*>(p); p += a256((tx + 4096) * 4);
    int* boff = reinterpret_cast<int*>(p); p += a256((tx + 4096) * 4);
    int* cursor = reinterpret_cast<int*>(p); p += a256((tx + 4096) * 4);
    int* bsum = reinterpret_cast<int*>(p); p += a256(16384 * 4);
    float4* at4 = reinterpret_cast<float4*>(p); p += a256(tx * 16);
    int* npix = reinterpret_cast<int*>(p);                   // [0] masked pixels, [1] texels that receive a contribution
    int* nact = npix + 1;
    const long long n = 3LL * A * A;
    PD_HIP(hipMemsetAsync(m, 0, n * 4, s));
    PD_HIP(hipMemsetAsync(vv, 0, n * 4, s));
    PD_HIP(hipMemsetAsync(cntb, 0, tx * 4, s));
    PD_HIP(hipMemsetAsync(cursor, 0, tx * 4, s));
    PD_HIP(hipMemsetAsync(npix, 0, 8, s));
    if (final_images != nullptr) PD_HIP(hipMemsetAsync(final_images, 0, px * 3 * 4, s));
    dim3 g(min(cdiv((long long)res * res, 256), 2048), V);
    k_optcolor_target<<<g, 256, 0, s>>>(inpainted, r, uv_map, face_idxs, res, shrinked, A, target, wmask);
    const int gp = min(cdiv((long long)px, 256), 8192);
    const int gt = min(cdiv((long long)tx, 256), 4096);
    // masked pixels -> buckets by base texel (counting sort, ascending pixel index inside a bucket) -> per-pixel arrays in that order
    k_oc_count<<<gp, 256, 0, s>>>(uv_map, wmask, V, res, A, cntb);
    OC_TRY(oc_scan(cntb, (int)tx, boff, bsum, npix, s));
    PD_HIP(hipMemcpyAsync(boff + tx, npix, 4, hipMemcpyDeviceToDevice, s));
    k_oc_scatter<<<gp, 256, 0, s>>>(uv_map, wmask, V, res, A, boff, cursor, pix);
    const int gw = min(cdiv((long long)tx, 64), 16384);
    k_oc_sortb<<<gw, 64, 0, s>>>(boff, (int)tx, pix);
    k_oc_records<<<gp, 256, 0, s>>>(pix, npix, uv_map, target, res, A, fxy, tgt4);
    k_oc_nactive<<<gt, 256, 0, s>>>(boff, A, (int)tx, nact);                 // (256 blocks: 184 us for this counter -- 16 dependent reads per thread)
    k_oc_pack<<<gt, 256, 0, s>>>(atlas, (int)tx, at4);
    const double inv_count = 1.0 / ((double)V * 3.0 * res * res);
    const int gf = min(cdiv((long long)px, 256 * OC_FW_PX), 8192);
    for (int it = 0; it < iterations; ++it) {
        const bool last = it == iterations - 1;
        k_oc_forward<<<gf, 256, 0, s>>>(at4, A, fxy, tgt4, npix, sgn, pix, res, last ? final_images : nullptr);
        const int step = it + 1;
        const double cur_lr = lr * pow(0.5, (double)(it / 15));            // StepLR(step_size 15, gamma 0.5)
        const double bc1 = 1.0 - pow(0.9, step), bc2 = 1.0 - pow(0.999, step);
        k_oc_backward_adam<<<gw, 64, 0, s>>>(boff, A, (int)tx, fxy, sgn, inv_count, atlas, m, vv, (float)(cur_lr / bc1), (float)sqrt(bc2), at4);
    }
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}
