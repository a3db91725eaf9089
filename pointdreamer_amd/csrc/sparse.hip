// Rows P4, P5, P6: sparse per-view images and masks for all V views, no host round trip.
// Reference: pointdreamer/ours_utils.py:848-882 (get_sparse_images), :954-1044 (get_one_sparse_img),
// :456-495 (paint_pixels), :497-532 (get_forground_inner_edge_mask 'dilate').
// Build rules (oracle/sparse.py): duplicate writes -> largest write-order index wins (atomicMax on the
// write index, resolved in a compose pass); edge pixel -> nearest valid point by exact integer distance,
// ties -> smallest point index; degenerate view (no valid point / no foreground) -> all background.
// Compiled with -ffp-contract=off.
#include "common.h"
using namespace pdhip;

struct ViewParams {
    float scale;
    int after_res;
    int pad;
    int degenerate;
    int rescaled;   // mask_ratio > thresh branch taken (after_res may still equal res)
};

struct SparseWs {
    ViewParams* params;   // [V]
    uint32_t* counts;     // [V][2]
    uint8_t* mask_new;    // [V][r*r]
    uint32_t* winA;       // [V][r*r]
    uint32_t* winB;       // [V][r*r]
    int32_t* nn_idx;      // [V][r*r]
    int32_t* pp;          // [V][N] packed (row<<16|col) or -1
    uint32_t* minidx;     // [V][r*r] smallest valid point index landing on the pixel (0xffffffff: none)
    uint32_t* edge_cnt;   // [V] number of edge pixels
    int32_t* edge_list;   // [V][r*r] edge pixels (row * res + col), any order
};

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static SparseWs carve(void* ws, int V, int N, int res, size_t* total) {
    size_t off = 0;
    char* base = (char*)ws;
    size_t px = (size_t)V * res * res;
    SparseWs w;
    w.params = (ViewParams*)(base + off); off += align256(sizeof(ViewParams) * V);
    w.counts = (uint32_t*)(base + off); off += align256(8 * (size_t)V);
    w.mask_new = (uint8_t*)(base + off); off += align256(px);
    w.winA = (uint32_t*)(base + off); off += align256(4 * px);
    w.winB = (uint32_t*)(base + off); off += align256(4 * px);
    w.nn_idx = (int32_t*)(base + off); off += align256(4 * px);
    w.pp = (int32_t*)(base + off); off += align256(4 * (size_t)V * (N > 0 ? N : 1));
    w.minidx = (uint32_t*)(base + off); off += align256(4 * px);
    w.edge_cnt = (uint32_t*)(base + off); off += align256(4 * (size_t)V);
    w.edge_list = (int32_t*)(base + off); off += align256(4 * px);
    if (total) *total = off;
    return w;
}

extern "C" size_t pdhip_sparse_views_ws_bytes(int V, int N, int res) {
    size_t t = 0;
    carve(nullptr, V, N, res, &t);
    return t;
}

__global__ void k_sparse_counts(const uint8_t* __restrict__ hard, const uint8_t* __restrict__ valid, int N, int res,
                                float thresh_f, float one_minus_thresh_f, ViewParams* __restrict__ params,
                                uint32_t* __restrict__ counts) {
    const int v = blockIdx.x;
    __shared__ uint32_t s_fg[16], s_va[16];
    uint32_t fg = 0, va = 0;
    const size_t hb = (size_t)v * res * res;
    const int rr = res * res;
    if ((rr & 15) == 0 && ((reinterpret_cast<uintptr_t>(hard) + hb) & 15) == 0) {      // 16 mask bytes per load
        const uint4* h4 = reinterpret_cast<const uint4*>(hard + hb);
        for (int i = threadIdx.x; i < rr / 16; i += blockDim.x) {
            const uint4 q = h4[i];
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                fg += ((w[k] & 0xffu) != 0) + ((w[k] & 0xff00u) != 0) + ((w[k] & 0xff0000u) != 0) + ((w[k] & 0xff000000u) != 0);
        }
    } else {
        for (int i = threadIdx.x; i < rr; i += blockDim.x) fg += hard[hb + i] ? 1 : 0;
    }
    {   // the validation bytes, 16 per load where the view's row allows it (30 byte loads in a row per thread were most of this kernel)
        const uint8_t* vb = valid + (size_t)v * N;
        const int head = min(N, (int)((16 - (reinterpret_cast<uintptr_t>(vb) & 15)) & 15));
        const int groups = (N - head) / 16;
        const uint4* v4 = reinterpret_cast<const uint4*>(vb + head);
        for (int i = threadIdx.x; i < groups; i += blockDim.x) {
            const uint4 q = v4[i];
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                va += ((w[k] & 0xffu) != 0) + ((w[k] & 0xff00u) != 0) + ((w[k] & 0xff0000u) != 0) + ((w[k] & 0xff000000u) != 0);
        }
        for (int i = threadIdx.x; i < head; i += blockDim.x) va += vb[i] ? 1 : 0;
        for (int i = head + groups * 16 + threadIdx.x; i < N; i += blockDim.x) va += vb[i] ? 1 : 0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { fg += __shfl_xor(fg, off); va += __shfl_xor(va, off); }
    if ((threadIdx.x & 63) == 0) { s_fg[threadIdx.x >> 6] = fg; s_va[threadIdx.x >> 6] = va; }
    __syncthreads();
    if (threadIdx.x == 0) {
        fg = 0; va = 0;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) { fg += s_fg[k]; va += s_va[k]; }
        counts[2 * v] = fg; counts[2 * v + 1] = va;
        ViewParams p;
        p.scale = 1.0f; p.after_res = res; p.pad = 0; p.rescaled = 0;
        p.degenerate = (fg == 0 || va == 0) ? 1 : 0;
        if (!p.degenerate) {
            float fgf = (float)fg, vaf = (float)va;
            float ratio = 1.0f - vaf / fgf;
            if (ratio > thresh_f) {
                p.rescaled = 1;
                float wanted = vaf / one_minus_thresh_f;
                p.scale = wanted / fgf;
                int ar = (int)floorf((float)res * p.scale);
                if (((res - ar) % 2) == 1) ar += 1;
                p.after_res = ar;
                p.pad = (res - ar) / 2;
            }
        }
        params[v] = p;
    }
}

__device__ __forceinline__ void bilinear_taps_s(int n_in, int n_out, int d, int& i0, int& i1, bool& w0, bool& w1) {
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

// new foreground mask (Resize+Pad when the view is rescaled) and winner-buffer init
__global__ void k_sparse_mask(const uint8_t* __restrict__ hard, int res, const ViewParams* __restrict__ params,
                              uint8_t* __restrict__ mask_new, uint32_t* __restrict__ winA, uint32_t* __restrict__ winB,
                              uint32_t* __restrict__ minidx, uint32_t* __restrict__ edge_cnt) {
    const int v = blockIdx.y;
    if (blockIdx.x == 0 && threadIdx.x == 0) edge_cnt[v] = 0;
    const ViewParams p = params[v];
    const uint8_t* src = hard + (size_t)v * res * res;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < res * res; idx += gridDim.x * blockDim.x) {
        uint8_t m;
        if (!p.rescaled) {
            m = src[idx] ? 1 : 0;
        } else {
            int y = idx / res - p.pad, x = idx % res - p.pad;
            m = 0;
            if (y >= 0 && y < p.after_res && x >= 0 && x < p.after_res) {
                int r0, r1, c0, c1;
                bool wr0, wr1, wc0, wc1;
                bilinear_taps_s(res, p.after_res, y, r0, r1, wr0, wr1);
                bilinear_taps_s(res, p.after_res, x, c0, c1, wc0, wc1);
                m = ((wr0 && wc0 && src[r0 * res + c0]) || (wr0 && wc1 && src[r0 * res + c1]) ||
                     (wr1 && wc0 && src[r1 * res + c0]) || (wr1 && wc1 && src[r1 * res + c1])) ? 1 : 0;
            }
        }
        size_t o = (size_t)v * res * res + idx;
        mask_new[o] = m;
        winA[o] = 0;
        winB[o] = 0;
        minidx[o] = 0xffffffffu;
    }
}

// P5: splat valid points (write index -> atomicMax), remember the (rescaled) pixel of every point
__device__ __forceinline__ void sparse_splat_body(int bx, int nbx, const int64_t* __restrict__ pix, const uint8_t* __restrict__ valid, int N, int res,
                               int point_size, const ViewParams* __restrict__ params, uint32_t* __restrict__ winA,
                               int32_t* __restrict__ pp, uint32_t* __restrict__ minidx) {
    const int v = blockIdx.y;
    const ViewParams p = params[v];
    const int g = 2 * point_size - 1;
    for (int n = bx * blockDim.x + threadIdx.x; n < N; n += nbx * blockDim.x) {
        size_t o = (size_t)v * N + n;
        int packed = -1;
        if (valid[o] && !p.degenerate) {
            int row = (int)pix[2 * o], col = (int)pix[2 * o + 1];
            if (p.rescaled) {
                float ur = (float)row / (float)res, uc = (float)col / (float)res;
                ur = ur * 2.0f - 1.0f; uc = uc * 2.0f - 1.0f;
                ur = ur * p.scale; uc = uc * p.scale;
                ur = (ur + 1.0f) * 0.5f; uc = (uc + 1.0f) * 0.5f;
                row = clip_to_int(ur * (float)res, res - 1);
                col = clip_to_int(uc * (float)res, res - 1);
            }
            packed = (row << 16) | col;
            atomicMin(&minidx[(size_t)v * res * res + row * res + col], (uint32_t)n);
            for (int a = 0; a < g; ++a) {
                int rr = row + a - (point_size - 1);
                if (rr < 0 || rr >= res) continue;
                for (int b = 0; b < g; ++b) {
                    int cc = col + b - (point_size - 1);
                    if (cc < 0 || cc >= res) continue;
                    atomicMax(&winA[(size_t)v * res * res + rr * res + cc], (uint32_t)(n * g * g + a * g + b + 1));
                }
            }
        }
        pp[o] = packed;
    }
}

// P6 + edge colouring: one block per (view,row).  Every inner-edge pixel takes the colour of its nearest valid point
// (exact integer distance, ties -> smallest point index): one wave per edge pixel searches the per-pixel min-index image
// in a growing square window -- the window is exhaustive once its radius covers the best distance found.
// P6a: the edge pixels (foreground with a background 8-neighbour) of every view, compacted into one list per view (one returning
// atomic per image row).  P6b then gives every edge pixel a wavefront of its own: with one workgroup per row the rows tangent to
// the silhouette (up to ~100 edge pixels) were the whole kernel time (38 us).
#define EDGE_BAND 8             // image rows per workgroup: ONE returning atomic per band (same-address atomics cost ~0.1-0.2 us each)
__device__ __forceinline__ void sparse_edge_list_body(int band, int* s_wcnt, int& s_base, const uint8_t* __restrict__ mask_new, int res,
                                                      const ViewParams* __restrict__ params, uint32_t* __restrict__ edge_cnt,
                                                      int32_t* __restrict__ edge_list) {
    const int v = blockIdx.y, row0 = band * EDGE_BAND;
    const ViewParams p = params[v];
    if (p.degenerate) return;
    const uint8_t* m = mask_new + (size_t)v * res * res;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int npix = min(EDGE_BAND, res - row0) * res;
    int running = 0;                                              // edge pixels of the earlier steps of this band (block-uniform)
    // two sweeps over the band: count, reserve the band's slice of the list, write
    for (int pass = 0; pass < 2; ++pass) {
        int total = 0;
        for (int i0 = 0; i0 < npix; i0 += 1024) {
            const int i = i0 + (int)threadIdx.x;
            bool edge = false;
            int lin = 0;
            if (i < npix) {
                const int row = row0 + i / res, col = i - (i / res) * res;
                lin = row * res + col;
                if (m[lin]) {
                    for (int dy = -1; dy <= 1; ++dy) {
                        const int y = row + dy;
                        if (y < 0 || y >= res) continue;
                        for (int dx = -1; dx <= 1; ++dx) {
                            const int x = col + dx;
                            if (x < 0 || x >= res) continue;
                            edge |= !m[y * res + x];
                        }
                    }
                }
            }
            const unsigned long long bal = __ballot(edge);
            if (lane == 0) s_wcnt[wave] = __popcll(bal);
            __syncthreads();
            int mine = 0, tot = 0;
#pragma unroll
            for (int w = 0; w < 16; ++w) { const int c = s_wcnt[w]; mine += w < wave ? c : 0; tot += c; }
            if (pass == 1 && edge) edge_list[(size_t)v * res * res + s_base + running + mine + __popcll(bal & ((1ull << lane) - 1ull))] = lin;
            if (pass == 1) running += tot;
            total += tot;
            __syncthreads();
        }
        if (pass == 0) {
            if (threadIdx.x == 0) s_base = total > 0 ? (int)atomicAdd(&edge_cnt[v], (uint32_t)total) : 0;
            __syncthreads();
        }
    }
}

// P5 splat (over the points) and P6a edge list (over the pixels) do not depend on each other: one launch, the first `splat_blocks`
// workgroups of a view splat, the others take an 8-row band each (a launch boundary and the shorter kernel's time less per shape)
__global__ __launch_bounds__(1024) void k_sparse_splat_edge_list(int splat_blocks, const int64_t* __restrict__ pix, const uint8_t* __restrict__ valid,
                                                                 int N, int res, int point_size, const ViewParams* __restrict__ params,
                                                                 uint32_t* __restrict__ winA, int32_t* __restrict__ pp, uint32_t* __restrict__ minidx,
                                                                 const uint8_t* __restrict__ mask_new, uint32_t* __restrict__ edge_cnt,
                                                                 int32_t* __restrict__ edge_list) {
    __shared__ int s_wcnt[16], s_base;
    if ((int)blockIdx.x < splat_blocks) sparse_splat_body(blockIdx.x, splat_blocks, pix, valid, N, res, point_size, params, winA, pp, minidx);
    else sparse_edge_list_body(blockIdx.x - splat_blocks, s_wcnt, s_base, mask_new, res, params, edge_cnt, edge_list);
}

__global__ void k_sparse_edges(const int32_t* __restrict__ edge_list, const uint32_t* __restrict__ edge_cnt, int res, int edge_point_size,
                               const ViewParams* __restrict__ params, const uint32_t* __restrict__ minidx,
                               uint32_t* __restrict__ winB, int32_t* __restrict__ nn_idx) {
    const int v = blockIdx.y;
    const ViewParams p = params[v];
    if (p.degenerate) return;
    const int ne = (int)edge_cnt[v];
    const int ge = 2 * edge_point_size - 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint32_t* mi = minidx + (size_t)v * res * res;
    for (int e = blockIdx.x * nw + wave; e < ne; e += gridDim.x * nw) {
        const int lin0 = edge_list[(size_t)v * res * res + e];
        const int row = lin0 / res, col = lin0 - row * res;
        unsigned long long best = ~0ull;
        int rho = 8;
        while (true) {
            const int side = 2 * rho + 1;
            best = ~0ull;
            for (int i = lane; i < side * side; i += 64) {
                const int dr = i / side - rho, dc = i - (i / side) * side - rho;
                const int rr = row + dr, cc = col + dc;
                if (rr < 0 || rr >= res || cc < 0 || cc >= res) continue;
                const uint32_t q = mi[rr * res + cc];
                if (q == 0xffffffffu) continue;
                const unsigned long long key = ((unsigned long long)(unsigned)(dr * dr + dc * dc) << 32) | q;
                best = key < best ? key : best;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned long long o = __shfl_xor(best, off);
                best = o < best ? o : best;
            }
            if (rho >= res) break;                                         // window covers the whole image
            if (best != ~0ull) {
                const long long d2 = (long long)(best >> 32);
                if (d2 <= (long long)rho * rho) break;                      // nothing outside the window can be closer or tie
                int need = (int)ceil(sqrt((double)d2));
                rho = max(need, rho + 1);
            } else {
                rho *= 4;
            }
            if (rho > res) rho = res;
        }
        if (lane == 0) {
            const int lin = row * res + col;
            nn_idx[(size_t)v * res * res + lin] = (int)(best & 0xffffffffu);
            for (int a = 0; a < ge; ++a) {
                int rr = row + a - (edge_point_size - 1);
                if (rr < 0 || rr >= res) continue;
                for (int b = 0; b < ge; ++b) {
                    int cc = col + b - (edge_point_size - 1);
                    if (cc < 0 || cc >= res) continue;
                    atomicMax(&winB[(size_t)v * res * res + rr * res + cc], (uint32_t)(lin * ge * ge + a * ge + b + 1));
                }
            }
        }
    }
}

// compose + vertical flip + sparse*mask0
__global__ void k_sparse_compose(const float* __restrict__ colors, int res, int point_size, int edge_point_size,
                                 const ViewParams* __restrict__ params, const uint8_t* __restrict__ mask_new,
                                 const uint32_t* __restrict__ winA, const uint32_t* __restrict__ winB,
                                 const int32_t* __restrict__ nn_idx, float* __restrict__ sparse,
                                 float* __restrict__ mask0, float* __restrict__ mask2, float* __restrict__ scale_factors, int vps, int N) {
    const int v = blockIdx.y;
    colors += (size_t)(v / vps) * 3 * (size_t)N;           // several shapes per call: view v shows the cloud of shape v / vps
    const int g2 = (2 * point_size - 1) * (2 * point_size - 1);
    const int ge2 = (2 * edge_point_size - 1) * (2 * edge_point_size - 1);
    if (blockIdx.x == 0 && threadIdx.x == 0) scale_factors[v] = params[v].scale;
    const size_t plane = (size_t)res * res;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < res * res; idx += gridDim.x * blockDim.x) {
        int y = idx / res, x = idx - y * res;
        size_t src = (size_t)v * plane + (size_t)(res - 1 - y) * res + x;
        float fg = mask_new[src] ? 1.0f : 0.0f;
        uint32_t wb = winB[src], wa = winA[src];
        float c0 = 0.f, c1 = 0.f, c2 = 0.f, m2 = 1.0f - fg;
        int n = -1;
        if (wb) n = nn_idx[(size_t)v * plane + (wb - 1) / ge2];
        else if (wa) n = (int)((wa - 1) / g2);
        if (n >= 0) {
            c0 = colors[3 * n]; c1 = colors[3 * n + 1]; c2 = colors[3 * n + 2];
            m2 = 1.0f;
        }
        size_t o = (size_t)v * 3 * plane + idx;
        sparse[o] = c0 * fg; sparse[o + plane] = c1 * fg; sparse[o + 2 * plane] = c2 * fg;
        mask0[o] = fg; mask0[o + plane] = fg; mask0[o + 2 * plane] = fg;
        mask2[o] = m2; mask2[o + plane] = m2; mask2[o + 2 * plane] = m2;
    }
}

static int sparse_impl(const int64_t* point_pixels, const float* colors, const uint8_t* validation, const uint8_t* hard_masks, int V, int vps,
                       int N, int res, int point_size, int edge_point_size, double mask_ratio_thresh, float* sparse, float* mask0,
                       float* mask2, float* scale_factors, float* mask_ratios, void* ws, void* stream);
extern "C" int pdhip_sparse_views(const int64_t* point_pixels, const float* colors, const uint8_t* validation,
                                  const uint8_t* hard_masks, int V, int N, int res, int point_size,
                                  int edge_point_size, double mask_ratio_thresh, float* sparse, float* mask0,
                                  float* mask2, float* scale_factors, float* mask_ratios, void* ws, void* stream) {
    return sparse_impl(point_pixels, colors, validation, hard_masks, V, V > 0 ? V : 1, N, res, point_size, edge_point_size, mask_ratio_thresh, sparse,
                       mask0, mask2, scale_factors, mask_ratios, ws, stream);
}
// S clouds of N points each, V views per cloud, in ONE set of launches: colors [S,N,3]; every per-view array has S*V leading entries
// (view g = s * V + v); workspace pdhip_sparse_views_ws_bytes(S * V, N, res).
extern "C" int pdhip_sparse_views_shapes(const int64_t* point_pixels, const float* colors, const uint8_t* validation,
                                         const uint8_t* hard_masks, int V, int S, int N, int res, int point_size,
                                         int edge_point_size, double mask_ratio_thresh, float* sparse, float* mask0,
                                         float* mask2, float* scale_factors, float* mask_ratios, void* ws, void* stream) {
    PD_REQUIRE(S >= 1 && V >= 1, "pdhip_sparse_views_shapes: bad sizes");
    return sparse_impl(point_pixels, colors, validation, hard_masks, S * V, V, N, res, point_size, edge_point_size, mask_ratio_thresh, sparse,
                       mask0, mask2, scale_factors, mask_ratios, ws, stream);
}
static int sparse_impl(const int64_t* point_pixels, const float* colors, const uint8_t* validation, const uint8_t* hard_masks, int V, int vps,
                       int N, int res, int point_size, int edge_point_size, double mask_ratio_thresh, float* sparse, float* mask0,
                       float* mask2, float* scale_factors, float* mask_ratios, void* ws, void* stream) {
    (void)mask_ratios;
    PD_REQUIRE(V > 0 && N >= 0 && res > 0 && res <= 32768, "pdhip_sparse_views: bad sizes V=%d N=%d res=%d", V, N, res);
    PD_REQUIRE(point_size >= 1 && edge_point_size >= 1 && point_size <= 8 && edge_point_size <= 8,
               "pdhip_sparse_views: point sizes must be in [1,8]");
    PD_REQUIRE((long long)N * 225 < 0x7fffffffLL && (long long)res * res * 225 < 0x7fffffffLL,
               "pdhip_sparse_views: write-order index overflows 32 bits");
    PD_REQUIRE(hard_masks && sparse && mask0 && mask2 && scale_factors && ws && (N == 0 || (point_pixels && colors && validation)),
               "pdhip_sparse_views: null pointer");
    hipStream_t s = as_stream(stream);
    SparseWs w = carve(ws, V, N, res, nullptr);
    const float thr = (float)mask_ratio_thresh;
    const float omt = (float)(1.0 - mask_ratio_thresh);
    k_sparse_counts<<<V, 1024, 0, s>>>(hard_masks, validation, N, res, thr, omt, w.params, w.counts);
    dim3 gm(min(cdiv((long long)res * res, 256), 256), V);
    k_sparse_mask<<<gm, 256, 0, s>>>(hard_masks, res, w.params, w.mask_new, w.winA, w.winB, w.minidx, w.edge_cnt);
    if (N > 0) {
        const int splat_blocks = min(cdiv(N, 1024), 64);
        k_sparse_splat_edge_list<<<dim3(splat_blocks + cdiv(res, EDGE_BAND), V), 1024, 0, s>>>(
            splat_blocks, point_pixels, validation, N, res, point_size, w.params, w.winA, w.pp, w.minidx, w.mask_new, w.edge_cnt, w.edge_list);
        k_sparse_edges<<<dim3(256, V), 256, 0, s>>>(w.edge_list, w.edge_cnt, res, edge_point_size, w.params, w.minidx, w.winB, w.nn_idx);
    }
    k_sparse_compose<<<gm, 256, 0, s>>>(colors, res, point_size, edge_point_size, w.params, w.mask_new, w.winA, w.winB,
                                        w.nn_idx, sparse, mask0, mask2, scale_factors, vps, N);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}
