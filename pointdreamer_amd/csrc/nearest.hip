// Rows I0 and Uq5: exact nearest-site fill (replaces scipy.interpolate.griddata(method='nearest') at
// pointdreamer/ours_utils.py:610-643 and unproject.dilate_atlas, unproject.py:480-504).
// Separable exact Euclidean search: (1) per column, nearest site row for every row (ties -> smaller row);
// (2) per pixel, scan columns outward from its own, key = (dist2, row', col'), stop once dx*dx > best dist2.
// Tie rule: lexicographically smallest (row', col') among all sites at minimal distance (oracle/inpaint.py).
// Integer work; HBM-bound (mask read + near_row write/read + colour gather).
#include "common.h"
using namespace pdhip;

// Column pass, parallel over (column, 64-row segment).  Pass A records every segment's first / last site row; pass B
// scans its own segment down and up with the carries taken from the other segments' summaries.
#define SEG 16                 // (64: a 256^2 x 8 job is 128 wavefronts of 64 dependent iterations -- 27 us of latency)
template <bool F32MASK>
__device__ __forceinline__ bool is_site(const void* mask, size_t base, size_t idx) {
    if (F32MASK) return reinterpret_cast<const float*>(mask)[base + idx] != 0.0f;
    return reinterpret_cast<const uint8_t*>(mask)[base + idx] != 0;
}

template <bool F32MASK>
__global__ void k_nearest_seg_summary(const void* __restrict__ mask, int64_t mask_bstride, int H, int W, int nseg,
                                      int32_t* __restrict__ seg_first, int32_t* __restrict__ seg_last) {
    const int b = blockIdx.z, s = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= W) return;
    const size_t base = (size_t)b * mask_bstride;
    int first = -1, last = -1;
    const int r1 = min(H, (s + 1) * SEG);
    for (int r = s * SEG; r < r1; ++r)
        if (is_site<F32MASK>(mask, base, (size_t)r * W + c)) { if (first < 0) first = r; last = r; }
    seg_first[((size_t)b * nseg + s) * W + c] = first;
    seg_last[((size_t)b * nseg + s) * W + c] = last;
}

template <bool F32MASK>
__global__ void k_nearest_cols(const void* __restrict__ mask, int64_t mask_bstride, int H, int W, int nseg,
                               const int32_t* __restrict__ seg_first, const int32_t* __restrict__ seg_last,
                               int32_t* __restrict__ near_row) {
    const int b = blockIdx.z, s = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= W) return;
    int32_t* nr = near_row + (size_t)b * H * W;
    const size_t base = (size_t)b * mask_bstride;
    // carries: nearest site above this segment / below this segment.  Rows grow with the segment index, so the carries are a
    // maximum / minimum over the other segments' summaries: eight independent loads per step instead of a dependent chain
    int last = -1, next = 0x7fffffff;
    for (int t0 = s - 1; t0 >= 0; t0 -= 8) {
        int v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = seg_last[((size_t)b * nseg + max(t0 - u, 0)) * W + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) last = max(last, v[u]);
        if (!__any(last < 0)) break;
    }
    for (int t0 = s + 1; t0 < nseg; t0 += 8) {
        int v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = seg_first[((size_t)b * nseg + min(t0 + u, nseg - 1)) * W + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) next = min(next, v[u] >= 0 ? v[u] : 0x7fffffff);
        if (!__any(next == 0x7fffffff)) break;
    }
    if (next == 0x7fffffff) next = -1;
    const int r0 = s * SEG;
    // the segment's mask column in registers: every load and every store of the segment is independent of the others (the two
    // sweeps used to go through near_row in memory, a store -> load round trip per row)
    bool site[SEG];
#pragma unroll
    for (int k = 0; k < SEG; ++k) site[k] = r0 + k < H && is_site<F32MASK>(mask, base, (size_t)(r0 + k) * W + c);
    int up[SEG];
#pragma unroll
    for (int k = 0; k < SEG; ++k) {            // nearest site at or above
        if (site[k]) last = r0 + k;
        up[k] = last;
    }
#pragma unroll
    for (int k = SEG - 1; k >= 0; --k) {       // merge with nearest site at or below; ties -> up (smaller row)
        const int r = r0 + k;
        if (site[k]) next = r;
        int pick = up[k];
        if (next >= 0 && (up[k] < 0 || (next - r) < (r - up[k]))) pick = next;
        if (r < H) nr[(size_t)r * W + c] = pick;
    }
}

__global__ void k_nearest_rows(const int32_t* __restrict__ near_row, int H, int W, const float* __restrict__ img,
                               float* __restrict__ out, int C, int64_t bstride, int64_t cstride, int64_t pstride) {
    extern __shared__ int32_t s_nr[];          // near_row of this image row
    const int b = blockIdx.y, r = blockIdx.x;
    const int32_t* nr = near_row + ((size_t)b * H + r) * W;
    for (int c = threadIdx.x; c < W; c += blockDim.x) s_nr[c] = nr[c];
    __syncthreads();
    for (int c = threadIdx.x; c < W; c += blockDim.x) {
        unsigned long long best = ~0ull;
        long long best_d2 = 0x7fffffffffffLL;
        for (int dx = 0; dx < W; ++dx) {
            long long dx2 = (long long)dx * dx;
            if (dx2 > best_d2) break;
            int c1 = c - dx, c2 = c + dx;
            if (c1 < 0 && c2 >= W) break;
            if (c1 >= 0) {
                int rr = s_nr[c1];
                if (rr >= 0) {
                    long long dy = r - rr, d2 = dx2 + dy * dy;
                    unsigned long long key = ((unsigned long long)d2 << 32) | ((unsigned long long)rr << 16) | (unsigned)c1;
                    if (key < best) { best = key; best_d2 = d2; }
                }
            }
            if (dx > 0 && c2 < W) {
                int rr = s_nr[c2];
                if (rr >= 0) {
                    long long dy = r - rr, d2 = dx2 + dy * dy;
                    unsigned long long key = ((unsigned long long)d2 << 32) | ((unsigned long long)rr << 16) | (unsigned)c2;
                    if (key < best) { best = key; best_d2 = d2; }
                }
            }
        }
        const size_t dst = (size_t)b * bstride + ((size_t)r * W + c) * pstride;
        if (best == ~0ull) {                    // no site anywhere: keep the input
            for (int ch = 0; ch < C; ++ch) out[dst + ch * cstride] = img[dst + ch * cstride];
            continue;
        }
        int sr = (int)((best >> 16) & 0xffff), sc = (int)(best & 0xffff);
        const size_t src = (size_t)b * bstride + ((size_t)sr * W + sc) * pstride;
        for (int ch = 0; ch < C; ++ch) out[dst + ch * cstride] = img[src + ch * cstride];
    }
}

extern "C" size_t pdhip_nearest_fill_ws_ints(int B, int H, int W) {
    return (size_t)B * H * W + 2 * (size_t)B * ((H + SEG - 1) / SEG) * W;
}

extern "C" int pdhip_nearest_fill(const float* img, float* out, int B, int C, int H, int W, int64_t batch_stride,
                                  int64_t chan_stride, int64_t pix_stride, const void* mask, int mask_is_f32,
                                  int64_t mask_batch_stride, int32_t* ws, void* stream) {
    PD_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && H <= 65535 && W <= 65535, "pdhip_nearest_fill: bad sizes");
    PD_REQUIRE(img && out && mask && ws, "pdhip_nearest_fill: null pointer");
    PD_REQUIRE(img != out, "pdhip_nearest_fill: in-place operation is not supported");
    hipStream_t s = as_stream(stream);
    const int nseg = cdiv(H, SEG);
    int32_t* seg_first = ws + (size_t)B * H * W;
    int32_t* seg_last = seg_first + (size_t)B * nseg * W;
    dim3 g1(cdiv(W, 64), nseg, B);
    if (mask_is_f32) {
        k_nearest_seg_summary<true><<<g1, 64, 0, s>>>(mask, mask_batch_stride, H, W, nseg, seg_first, seg_last);
        k_nearest_cols<true><<<g1, 64, 0, s>>>(mask, mask_batch_stride, H, W, nseg, seg_first, seg_last, ws);
    } else {
        k_nearest_seg_summary<false><<<g1, 64, 0, s>>>(mask, mask_batch_stride, H, W, nseg, seg_first, seg_last);
        k_nearest_cols<false><<<g1, 64, 0, s>>>(mask, mask_batch_stride, H, W, nseg, seg_first, seg_last, ws);
    }
    dim3 g2(H, B);
    k_nearest_rows<<<g2, 256, W * sizeof(int32_t), s>>>(ws, H, W, img, out, C, batch_stride, chan_stride, pix_stride);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}
