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

// ---- gather formulation on COMPACT lists (round 3).  The texture coordinates and the masks are fixed for the whole optimisation, so
// everything an iteration needs is laid out once:
//   * the masked pixels, in pixel order, as 16-byte records {fx, fy, tex, sgn}: the bilinear fractions (exactly representable in
//     f32: ix = u A - 0.5 with a 24-bit u and A = 2^k leaves at most 24 significant bits below the binary point, and out-of-range
//     coordinates are clamped to integers), the index of the top-left texel, and the sign bytes the forward pass writes;
//     their targets as float4 next to them.  The forward pass is a coalesced stream over these two arrays + four 16-byte gathers
//     from the interleaved atlas per pixel (no mask test, no uv arithmetic, no divergence);
//   * the transpose of the sampling operator as a CSR table texel -> entries, an entry = (compact pixel << 2 | corner) in 4 bytes,
//     sorted ascending inside a texel (deterministic f64 summation order -- the reference's grid_sample backward / index_put atomics
//     are not).  The backward pass gathers the pixel's record (one 16-byte request gives the sign AND the fractions the weight is
//     recomputed from, exactly) instead of streaming an 8-byte weight next to a 4-byte gather;
//   * the texels that receive any contribution: the others have zero gradient and zero Adam moments for ever, their update is
//     exactly + 0, so they -- and their 40 bytes of optimiser state per iteration -- are skipped.
// Round 2 (uv arithmetic + mask test per pixel and iteration, 12-byte entries, every texel updated): 129 + 154 us per iteration at
// V = 8, res = 1024 on the stage benchmark; this form: see DESIGN.md.

struct __align__(16) OcRec { float fx, fy; int tex; char4 sgn; };

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

__global__ void k_oc_flags(const uint8_t* __restrict__ b, long long n, int* __restrict__ f) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) f[i] = b[i] ? 1 : 0;
}
__global__ void k_oc_flags_pos(const int* __restrict__ c, int n, int* __restrict__ f) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) f[i] = c[i] > 0 ? 1 : 0;
}

// masked pixel p -> record cid_of[p]; counts the texel contributions on the way
__global__ void k_oc_records(const float* __restrict__ uv_map, const uint8_t* __restrict__ wmask, const int* __restrict__ cid_of,
                             const float* __restrict__ target, int V, int res, int A, OcRec* __restrict__ rec,
                             float4* __restrict__ tgt4, int* __restrict__ pix_of, int* __restrict__ cnt) {
    const long long total = (long long)V * res * res;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < total; p += (long long)gridDim.x * blockDim.x) {
        if (!wmask[p]) continue;
        const int v = (int)(p / ((long long)res * res)), idx = (int)(p - (long long)v * res * res);
        const int y = idx / res, x = idx - y * res;
        int tex; double fx, fy;
        oc_base(uv_map, ((size_t)v * res + (res - 1 - y)) * res + x, A, &tex, &fx, &fy);
        const int cid = cid_of[p];
        OcRec r; r.fx = (float)fx; r.fy = (float)fy; r.tex = tex; r.sgn = make_char4(0, 0, 0, 0);
        rec[cid] = r;
        const size_t plane = (size_t)res * res, o = (size_t)v * 3 * plane + (size_t)y * res + x;
        tgt4[cid] = make_float4(target[o], target[o + plane], target[o + 2 * plane], 0.f);
        pix_of[cid] = (int)p;
        for (int c = 0; c < 4; ++c) { int t; double wt; if (oc_corner(tex, fx, fy, A, c, &t, &wt)) atomicAdd(&cnt[t], 1); }
    }
}
__global__ void k_oc_fill(const OcRec* __restrict__ rec, const int* __restrict__ npix, int A, const int* __restrict__ off,
                          int* __restrict__ cursor, int* __restrict__ ent) {
    const int n = *npix;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const OcRec r = rec[i];
        for (int c = 0; c < 4; ++c) {
            int t; double wt;
            if (oc_corner(r.tex, (double)r.fx, (double)r.fy, A, c, &t, &wt)) ent[off[t] + atomicAdd(&cursor[t], 1)] = (i << 2) | c;
        }
    }
}
// active texel a -> (texel, first entry, entries) as one int4-sized record: the per-iteration kernels read them with one coalesced load
__global__ void k_oc_active(const int* __restrict__ cnt, const int* __restrict__ off, const int* __restrict__ apos, int ntex,
                            int4* __restrict__ act) {
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < ntex; t += gridDim.x * blockDim.x)
        if (cnt[t] > 0) act[apos[t]] = make_int4(t, off[t], cnt[t], 0);
}

// entries of one texel in ascending order (the cursor order above is a race): insertion sort, lists are short.  One wave owns 64
// consecutive ACTIVE texels, whose lists are one contiguous CSR range staged through LDS; longer ranges are sorted in global memory.
#define OC_SORT_CAP 4096
__global__ __launch_bounds__(64) void k_oc_sort(const int4* __restrict__ act, const int* __restrict__ nact, int* __restrict__ ent) {
    __shared__ int s_e[OC_SORT_CAP];
    const int lane = threadIdx.x, na = *nact;
    for (int a0 = blockIdx.x * 64; a0 < na; a0 += gridDim.x * 64) {
        const int a = a0 + lane;
        const int4 me = act[min(a, na - 1)], first = act[a0], lastr = act[min(a0 + 63, na - 1)];
        const int b = me.y, n = a < na ? me.z : 0;
        const int R0 = first.y, R1 = lastr.y + lastr.z, len = R1 - R0;
        if (len <= OC_SORT_CAP) {
            for (int i = lane; i < len; i += 64) s_e[i] = ent[R0 + i];
            __syncthreads();
            const int lb = b - R0;
            for (int i = 1; i < n; ++i) {
                const int k = s_e[lb + i];
                int j = i - 1;
                while (j >= 0 && s_e[lb + j] > k) { s_e[lb + j + 1] = s_e[lb + j]; --j; }
                s_e[lb + j + 1] = k;
            }
            __syncthreads();
            for (int i = lane; i < len; i += 64) ent[R0 + i] = s_e[i];
            __syncthreads();
        } else {
            for (int i = 1; i < n; ++i) {
                const int k = ent[b + i];
                int j = i - 1;
                while (j >= 0 && ent[b + j] > k) { ent[b + j + 1] = ent[b + j]; --j; }
                ent[b + j + 1] = k;
            }
        }
    }
}

// interleaved copy of the atlas being optimised (x, y, z = the three planes): the forward pass fetches a corner with one 16-byte
// gather instead of three 4-byte ones; the backward pass keeps it in step with the planar parameter
__global__ void k_oc_pack(const float* __restrict__ atlas, int ntex, float4* __restrict__ at4) {
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < ntex; t += gridDim.x * blockDim.x)
        at4[t] = make_float4(atlas[t], atlas[(size_t)ntex + t], atlas[2 * (size_t)ntex + t], 0.f);
}

// forward: bilinear lookup (f64), clamp, sign of the L1 residual per channel (0 where the clamp or the residual kills the gradient)
__global__ __launch_bounds__(256) void k_oc_forward(const float4* __restrict__ at4, int A, OcRec* __restrict__ rec,
                                                    const float4* __restrict__ tgt4, const int* __restrict__ npix,
                                                    const int* __restrict__ pix_of, int res, float* __restrict__ images) {
    const int n = *npix;
    const int stride = gridDim.x * blockDim.x;
    // two records per thread and trip: both records' streams and all eight corner gathers are requested before the first is used
    for (int i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += 2 * stride) {
        OcRec r[2]; float4 tg[2]; int tex[2][4]; double wt[2][4]; bool ok[2][4]; float4 cor[2][4];
#pragma unroll
        for (int q = 0; q < 2; ++q) { const int i = min(i0 + q * stride, n - 1); r[q] = rec[i]; tg[q] = tgt4[i]; }
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                ok[q][c] = oc_corner(r[q].tex, (double)r[q].fx, (double)r[q].fy, A, c, &tex[q][c], &wt[q][c]);
                cor[q][c] = at4[ok[q][c] ? tex[q][c] : r[q].tex];
            }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = i0 + q * stride;
            if (i >= n) break;
            char4 sg = make_char4(0, 0, 0, 0);
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
                const signed char sc = (!pass || d == 0.0) ? 0 : (d > 0.0 ? 1 : -1);
                if (c == 0) sg.x = sc; else if (c == 1) sg.y = sc; else sg.z = sc;
            }
            rec[i].sgn = sg;
            if (images) {                                                   // last iteration: the final render (masked-out pixels stay 0)
                const int p = pix_of[i];
                const size_t plane = (size_t)res * res;
                const int v = (int)(p / plane);
                const size_t o = (size_t)v * 3 * plane + (p - (size_t)v * plane);
                images[o] = im[0]; images[o + plane] = im[1]; images[o + 2 * plane] = im[2];
            }
        }
    }
}

// backward + torch.optim.Adam (betas 0.9/0.999, eps 1e-8, no weight decay; single-tensor update order), per ACTIVE texel.
// One wave owns 64 consecutive active texels = one contiguous CSR range.  The range is streamed through LDS in chunks: 64 lanes load
// entries coalesced and gather the pixels' records with 64 independent 16-byte requests in flight (sign + the fractions the f64
// weight is recomputed from), then every lane adds the entries of its own list that fall into the chunk, in list order.
#define OC_BW_CHUNK 512
#define OC_BW_U (OC_BW_CHUNK / 64)
__global__ __launch_bounds__(64) void k_oc_backward_adam(const int4* __restrict__ act, const int* __restrict__ nact,
                                                         const int* __restrict__ ent, const OcRec* __restrict__ rec, double inv_count,
                                                         float* __restrict__ param, float* __restrict__ m, float* __restrict__ vv,
                                                         int ntex, float step_size, float bc2_sqrt, float4* __restrict__ at4) {
    __shared__ double sw[OC_BW_CHUNK];                         // one wave per workgroup: the barriers below are wave-local
    __shared__ char4 ss[OC_BW_CHUNK];
    const int lane = threadIdx.x, na = *nact;
    for (int a0 = blockIdx.x * 64; a0 < na; a0 += gridDim.x * 64) {
        const int a = a0 + lane;
        const int4 me = act[min(a, na - 1)], first = act[a0], lastr = act[min(a0 + 63, na - 1)];
        const int t = me.x, b = me.y, n = a < na ? me.z : 0;
        const int R0 = first.y, R1 = lastr.y + lastr.z;
        // the optimiser state of this lane's texel: requested now, used after the gradient is summed
        float pm[3], pv[3], pp[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const size_t i = (size_t)c * ntex + t;
            pm[c] = m[i]; pv[c] = vv[i]; pp[c] = param[i];
        }
        double g0 = 0.0, g1 = 0.0, g2 = 0.0;
        int cur = b;
        const int e = b + n;
        for (int c0 = R0; c0 < R1; c0 += OC_BW_CHUNK) {
            const int len = min(OC_BW_CHUNK, R1 - c0);
            // a chunk = OC_BW_U entries per lane: all entry loads, then all record gathers in flight together (the staging loop used
            // to be a chain of dependent round trips, one entry per lane at a time)
            int en[OC_BW_U];
            OcRec rr[OC_BW_U];
#pragma unroll
            for (int u = 0; u < OC_BW_U; ++u) { const int i = u * 64 + lane; en[u] = i < len ? ent[c0 + i] : 0; }
#pragma unroll
            for (int u = 0; u < OC_BW_U; ++u) rr[u] = rec[en[u] >> 2];
#pragma unroll
            for (int u = 0; u < OC_BW_U; ++u) {
                const int i = u * 64 + lane;
                const double fx = (double)rr[u].fx, fy = (double)rr[u].fy;
                const int c = en[u] & 3;
                sw[i] = (c & 1 ? fx : 1.0 - fx) * (c & 2 ? fy : 1.0 - fy);
                ss[i] = rr[u].sgn;
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
        if (a < na) {
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
    return a256(px * 3 * 4) /*target*/ + a256(px) /*wmask*/ + 2 * a256(px * 4) /*flags -> cid_of, pix_of*/ + 2 * a256(px * 16) /*records, targets*/ +
           2 * a256(tx * 3 * 4) /*m, v*/ + 5 * a256((tx + 4096) * 4) /*cnt, off, cursor, apos, flags*/ + a256((tx + 4096) * 16) /*active texels*/ + a256(16384 * 4) /*block sums*/ +
           a256(px * 4 * 4) /*entries*/ + a256(tx * 16) /*interleaved atlas*/ + 256 /*counters*/;
}

extern "C" int pdhip_optimize_color(float* atlas /*[3,A,A] in/out*/, int A, const float* uv_map, const int64_t* face_idxs, int V,
                                    int res, const float* inpainted, int r, const uint8_t* shrinked, double lr, int iterations,
                                    float* final_images /*[V,3,res,res] or NULL*/, void* ws, void* stream) {
    PD_REQUIRE(atlas && uv_map && face_idxs && inpainted && ws && A > 0 && V > 0 && res > 0 && r > 0 && iterations >= 0,
               "pdhip_optimize_color: bad arguments");
    PD_REQUIRE((long long)A * A <= 4096LL * 1024 && (long long)V * res * res <= (long long)OC_SCAN_MAXB * OC_SCAN_BLK,
               "pdhip_optimize_color: atlas / view size too large (A^2 <= 4 Mi texels, V res^2 <= 64 Mi pixels)");
    hipStream_t s = as_stream(stream);
    const size_t px = (size_t)V * res * res, tx = (size_t)A * A;
    char* p = reinterpret_cast<char*>(ws);
    float* target = reinterpret_cast<float*>(p); p += a256(px * 3 * 4);
    uint8_t* wmask = reinterpret_cast<uint8_t*>(p); p += a256(px);
    int* cid_of = reinterpret_cast<int*>(p); p += a256(px * 4);
    int* pix_of = reinterpret_cast<int*>(p); p += a256(px * 4);
    OcRec* rec = reinterpret_cast<OcRec*>(p); p += a256(px * 16);
    float4* tgt4 = reinterpret_cast<float4*>(p); p += a256(px * 16);
    float* m = reinterpret_cast<float*>(p); p += a256(tx * 3 * 4);
    float* vv = reinterpret_cast<float*>(p); p += a256(tx * 3 * 4);
    int* cnt = reinterpret_cast<int*>(p); p += a256((tx + 4096) * 4);
    int* off = reinterpret_cast<int*>(p); p += a256((tx + 4096) * 4);
    int* cursor = reinterpret_cast<int*>(p); p += a256((tx + 4096) * 4);
    int* apos = reinterpret_cast<int*>(p); p += a256((tx + 4096) * 4);
    int* aflag = reinterpret_cast<int*>(p); p += a256((tx + 4096) * 4);
    int4* act = reinterpret_cast<int4*>(p); p += a256((tx + 4096) * 16);
    int* bsum = reinterpret_cast<int*>(p); p += a256(16384 * 4);
    int* ent = reinterpret_cast<int*>(p); p += a256(px * 4 * 4);
    float4* at4 = reinterpret_cast<float4*>(p); p += a256(tx * 16);
    int* npix = reinterpret_cast<int*>(p);                   // [0] masked pixels, [1] active texels
    int* nact = npix + 1;
    const long long n = 3LL * A * A;
    PD_HIP(hipMemsetAsync(m, 0, n * 4, s));
    PD_HIP(hipMemsetAsync(vv, 0, n * 4, s));
    PD_HIP(hipMemsetAsync(cnt, 0, tx * 4, s));
    PD_HIP(hipMemsetAsync(cursor, 0, tx * 4, s));
    if (final_images != nullptr) PD_HIP(hipMemsetAsync(final_images, 0, px * 3 * 4, s));
    dim3 g(min(cdiv((long long)res * res, 256), 2048), V);
    k_optcolor_target<<<g, 256, 0, s>>>(inpainted, r, uv_map, face_idxs, res, shrinked, A, target, wmask);
    const int gp = min(cdiv((long long)px, 256), 8192);
    const int gt = min(cdiv((long long)tx, 256), 4096);
    // masked pixels -> compact records (pixel order); texel contribution counts -> CSR offsets; active texels
    k_oc_flags<<<gp, 256, 0, s>>>(wmask, (long long)px, pix_of);
    OC_TRY(oc_scan(pix_of, (int)px, cid_of, bsum, npix, s));
    k_oc_records<<<gp, 256, 0, s>>>(uv_map, wmask, cid_of, target, V, res, A, rec, tgt4, pix_of, cnt);
    OC_TRY(oc_scan(cnt, (int)tx, off, bsum, nullptr, s));
    k_oc_flags_pos<<<gt, 256, 0, s>>>(cnt, (int)tx, aflag);
    OC_TRY(oc_scan(aflag, (int)tx, apos, bsum, nact, s));
    k_oc_active<<<gt, 256, 0, s>>>(cnt, off, apos, (int)tx, act);
    k_oc_fill<<<gp, 256, 0, s>>>(rec, npix, A, off, cursor, ent);
    const int gw = min(cdiv((long long)tx, 64), 16384);
    k_oc_sort<<<gw, 64, 0, s>>>(act, nact, ent);
    k_oc_pack<<<gt, 256, 0, s>>>(atlas, (int)tx, at4);
    const double inv_count = 1.0 / ((double)V * 3.0 * res * res);
    for (int it = 0; it < iterations; ++it) {
        const bool last = it == iterations - 1;
        k_oc_forward<<<gp, 256, 0, s>>>(at4, A, rec, tgt4, npix, pix_of, res, last ? final_images : nullptr);
        const int step = it + 1;
        const double cur_lr = lr * pow(0.5, (double)(it / 15));            // StepLR(step_size 15, gamma 0.5)
        const double bc1 = 1.0 - pow(0.9, step), bc2 = 1.0 - pow(0.999, step);
        k_oc_backward_adam<<<gw, 64, 0, s>>>(act, nact, ent, rec, inv_count, atlas, m, vv, (int)tx, (float)(cur_lr / bc1),
                                             (float)sqrt(bc2), at4);
    }
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}
