// Rows Uq1-Uq4, N1-N3: UV-space unprojection with Non-Border-First view selection.
// Reference: pointdreamer/unproject.py:201-425 (unproject), :429-475 (shrink), utils/utils_2d.py:799-845.
// Dense [A,A] formulation: every atlas texel is one thread; the reference's compacted [P] lists are
// produced on demand by pdhip_compact_texels (row-major order == boolean-mask order).
// Compiled with -ffp-contract=off (arithmetic contract in oracle/unproject.py).
#include "common.h"
#include <algorithm>
using namespace pdhip;

#define MAXV 32
static thread_local int g_unproject_generic = 0;        // test / lab hook: 1 = the run-time-V kernels even at V = 8
extern "C" int pdhip_debug_set_unproject_generic(int on) { const int old = g_unproject_generic; g_unproject_generic = on; return old; }

// ------------------------------------------------------------------------------ Uq1 + Uq2
// (measured: four texels per thread with 16-byte position loads and 4-byte verdict stores is SLOWER, 34 vs 28 us -- the kernel is
// bound by the eight scattered depth-map reads per texel, which want more threads in flight, not fewer instructions)
__global__ void k_texel_visibility(const float* __restrict__ cams, int V, const float* __restrict__ gb_pos,
                                     const uint8_t* __restrict__ mask, int A, const float* __restrict__ uv_centers,
                                     const float* __restrict__ uv_scales, float pad9, const float* __restrict__ mesh,
                                     int R, float offset, uint8_t* __restrict__ vis, int S) {
    // S atlases (shapes) in one launch: texel idx of shape s = gidx / A^2; its views are g = s * V + v (cameras shared by the shapes)
    const size_t n = (size_t)A * A, total = n * (size_t)S;
    for (size_t gidx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; gidx < total; gidx += (size_t)gridDim.x * blockDim.x) {
        const size_t sh = gidx / n, idx = gidx - sh * n;
        const bool m = mask[gidx];
        float x = 0.f, y = 0.f, z = 0.f;
        if (m) { x = gb_pos[3 * gidx]; y = gb_pos[3 * gidx + 1]; z = gb_pos[3 * gidx + 2]; }
        for (int v = 0; v < V; ++v) {
            const size_t g = sh * V + v;
            uint8_t o = 0;
            if (m) {
                const Cam c = load_cam(cams + 16 * v);
                float xn, yn, zn;
                cam_transform(c, x, y, z, xn, yn, zn);
                float u = ((xn - uv_centers[2 * g]) / uv_scales[g]) * pad9 + 0.5f;
                float w = ((yn - uv_centers[2 * g + 1]) / uv_scales[g]) * pad9 + 0.5f;
                int col = clip_to_int(u * (float)R, R - 1);
                int row = clip_to_int(w * (float)R, R - 1);
                float ref = mesh[(g * R + row) * R + col];
                o = ((zn - ref) <= offset) ? 1 : 0;
            }
            vis[g * n + idx] = o;
        }
    }
}

// The same with the view count a compile-time constant (round 5): a texel's VC depth-map reads are independent, but with a run-time
// trip count the loop is gather -> wait -> store, VC times in a row; unrolled, all VC gathers are in flight together and the verdicts
// are stored after one wait.  Same expressions in the same order per view (the verdict is a comparison: bit-identical).
template <int VC>
__global__ __launch_bounds__(256) void k_texel_visibility_v(const float* __restrict__ cams, const float* __restrict__ gb_pos,
                                                            const uint8_t* __restrict__ mask, int A, const float* __restrict__ uv_centers,
                                                            const float* __restrict__ uv_scales, float pad9, const float* __restrict__ mesh,
                                                            int R, float offset, uint8_t* __restrict__ vis, int S) {
    const size_t n = (size_t)A * A, total = n * (size_t)S;
    for (size_t gidx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; gidx < total; gidx += (size_t)gridDim.x * blockDim.x) {
        const size_t sh = gidx / n, idx = gidx - sh * n;
        const bool m = mask[gidx];
        uint8_t o[VC];
#pragma unroll
        for (int v = 0; v < VC; ++v) o[v] = 0;
        if (m) {
            const float x = gb_pos[3 * gidx], y = gb_pos[3 * gidx + 1], z = gb_pos[3 * gidx + 2];
            float zn[VC], ref[VC];
#pragma unroll
            for (int v = 0; v < VC; ++v) {
                const size_t g = sh * VC + v;
                const Cam c = load_cam(cams + 16 * v);
                float xn, yn;
                cam_transform(c, x, y, z, xn, yn, zn[v]);
                float u = ((xn - uv_centers[2 * g]) / uv_scales[g]) * pad9 + 0.5f;
                float w = ((yn - uv_centers[2 * g + 1]) / uv_scales[g]) * pad9 + 0.5f;
                int col = clip_to_int(u * (float)R, R - 1);
                int row = clip_to_int(w * (float)R, R - 1);
                ref[v] = mesh[(g * R + row) * R + col];
            }
#pragma unroll
            for (int v = 0; v < VC; ++v) o[v] = ((zn[v] - ref[v]) <= offset) ? 1 : 0;
        }
#pragma unroll
        for (int v = 0; v < VC; ++v) vis[(sh * VC + v) * n + idx] = o[v];
    }
}

// S atlases with V views each in ONE launch: gb_pos [S,A,A,3], mask [S,A,A]; cam_params [V] shared by the shapes; uv_centers / uv_scales /
// mesh_depths / visibility have S*V leading entries (view g = s * V + v).  S = 1: pdhip_texel_visibility.
extern "C" int pdhip_texel_visibility_shapes(const float* cam_params, int V, int S, const float* gb_pos, const uint8_t* mask, int A,
                                             const float* uv_centers, const float* uv_scales, double padding,
                                             const float* mesh_depths, int R, float offset, uint8_t* visibility, void* stream) {
    PD_REQUIRE(V > 0 && V <= MAXV && A > 0 && R > 0 && S >= 1, "pdhip_texel_visibility: bad sizes V=%d A=%d R=%d S=%d", V, A, R, S);
    PD_REQUIRE(cam_params && gb_pos && mask && uv_centers && uv_scales && mesh_depths && visibility,
               "pdhip_texel_visibility: null pointer");
    const float pad9 = (float)(1.0 - 2.0 * padding);
    const int grid = min(cdiv((long long)A * A * S, 256), 4096 * min(S, 8));
    if (V == 8 && g_unproject_generic == 0)
        k_texel_visibility_v<8><<<grid, 256, 0, as_stream(stream)>>>(cam_params, gb_pos, mask, A, uv_centers, uv_scales, pad9, mesh_depths, R, offset, visibility, S);
    else
        k_texel_visibility<<<grid, 256, 0, as_stream(stream)>>>(cam_params, V, gb_pos, mask, A, uv_centers, uv_scales, pad9, mesh_depths, R, offset, visibility, S);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}
extern "C" int pdhip_texel_visibility(const float* cam_params, int V, const float* gb_pos, const uint8_t* mask, int A,
                                      const float* uv_centers, const float* uv_scales, double padding,
                                      const float* mesh_depths, int R, float offset, uint8_t* visibility, void* stream) {
    return pdhip_texel_visibility_shapes(cam_params, V, 1, gb_pos, mask, A, uv_centers, uv_scales, padding, mesh_depths, R, offset, visibility, stream);
}

// ------------------------------------------------------------------------------ N1-N3
__device__ __forceinline__ int px(const uint8_t* img, int A, int y, int x) {
    return (y >= 0 && y < A && x >= 0 && x < A) ? (img[(size_t)y * A + x] ? 1 : 0) : 0;
}
__device__ __forceinline__ bool scharr_edge(const uint8_t* img, int A, int y, int x) {
    int gx = 3 * (px(img, A, y - 1, x + 1) - px(img, A, y - 1, x - 1)) + 10 * (px(img, A, y, x + 1) - px(img, A, y, x - 1)) +
             3 * (px(img, A, y + 1, x + 1) - px(img, A, y + 1, x - 1));
    int gy = 3 * (px(img, A, y + 1, x - 1) - px(img, A, y - 1, x - 1)) + 10 * (px(img, A, y + 1, x) - px(img, A, y - 1, x)) +
             3 * (px(img, A, y + 1, x + 1) - px(img, A, y - 1, x + 1));
    return gx != 0 || gy != 0;
}

// N1: edges_v = scharr(vis_v) & ~scharr(chart mask)
__global__ void k_nbf_edges(const uint8_t* __restrict__ mask, const uint8_t* __restrict__ vis, int A,
                            uint8_t* __restrict__ edges, int vps) {
    const int v = blockIdx.y;
    const size_t n = (size_t)A * A;
    mask += (size_t)(v / vps) * n;                          // several atlases per call: view v belongs to shape v / vps
    const uint8_t* vv = vis + (size_t)v * n;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        int y = (int)(idx / A), x = (int)(idx - (size_t)y * A);
        bool e = scharr_edge(vv, A, y, x) && !scharr_edge(mask, A, y, x);
        edges[(size_t)v * n + idx] = e ? 1 : 0;
    }
}

// N2 horizontal: OR over [x-r, x+r] (reflect padding adds nothing to an OR window)
__global__ void k_nbf_dilate_h(const uint8_t* __restrict__ in, int A, int r, uint8_t* __restrict__ out) {
    const int v = blockIdx.y;
    const size_t n = (size_t)A * A;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        int y = (int)(idx / A), x = (int)(idx - (size_t)y * A);
        const uint8_t* row = in + (size_t)v * n + (size_t)y * A;
        int lo = max(0, x - r), hi = min(A - 1, x + r);
        uint8_t o = 0;
        for (int k = lo; k <= hi; ++k) o |= row[k];
        out[(size_t)v * n + idx] = o ? 1 : 0;
    }
}

// N2 vertical + N3: shrunk = vis & ~border
__global__ void k_nbf_dilate_v_shrink(const uint8_t* __restrict__ tmp, const uint8_t* __restrict__ vis, int A, int r,
                                      uint8_t* __restrict__ out) {
    const int v = blockIdx.y;
    const size_t n = (size_t)A * A;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        int y = (int)(idx / A), x = (int)(idx - (size_t)y * A);
        int lo = max(0, y - r), hi = min(A - 1, y + r);
        uint8_t b = 0;
        for (int k = lo; k <= hi; ++k) b |= tmp[(size_t)v * n + (size_t)k * A + x];
        out[(size_t)v * n + idx] = (vis[(size_t)v * n + idx] && !b) ? 1 : 0;
    }
}

// ---- bit-packed NBF (A % 64 == 0): 64 texels per 64-bit word.
// k_pack_bits: bytes -> bits by wave ballot.  k_nbf_bits: per (view, 32-row band) the Scharr edge test becomes ~20 bitwise ops
// per 64 texels (on binary images gx != 0  <=>  a12 != a10  or  a02 + a22 != a00 + a20, and likewise gy -- the +-10 term cannot be
// cancelled by the two +-3 terms), the (2r+1)-wide OR dilation is word shifts with carries from the neighbouring words, the
// vertical one an OR over 2r+1 LDS rows; the result is expanded back to the byte mask the ABI hands out.
// 8 MB of visibility in, 8 MB out, everything in between stays in 1 MB of L2-resident bitmaps.
__global__ void k_pack_bits(const uint8_t* __restrict__ in, long long nwords, unsigned long long* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long wave = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6, nw = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long w = wave; w < nwords; w += nw) {
        const unsigned long long b = __ballot(in[w * 64 + lane] != 0);
        if (lane == 0) out[w] = b;
    }
}
// the same for two byte images in one launch, 16 bytes per lane and load (16-byte aligned inputs, word counts multiples of 16):
// a lane turns its 16 bytes into 16 bits, four lanes make a word.  (One byte per lane and load was 2 x 6.6 us for 9 MB.)
__device__ __forceinline__ unsigned int nz_bits4(unsigned int w) {          // bit k = (byte k of w != 0)
    const unsigned int hi = (((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w) & 0x80808080u;
    return (((hi >> 7) * 0x00204081u) >> 21) & 0xfu;
}
__global__ __launch_bounds__(256) void k_pack_bits16(const uint8_t* __restrict__ in_a, long long groups_a, unsigned long long* __restrict__ out_a,
                                                    const uint8_t* __restrict__ in_b, long long groups_b, unsigned long long* __restrict__ out_b) {
    const long long t0 = blockIdx.x * (long long)blockDim.x + threadIdx.x, nt = (long long)gridDim.x * blockDim.x;
    for (long long g = t0; g < groups_a + groups_b; g += nt) {               // group = 16 bytes = a quarter word; whole waves stay in one image
        const bool second = g >= groups_a;
        const long long gi = second ? g - groups_a : g;
        const uint4 q = reinterpret_cast<const uint4*>(second ? in_b : in_a)[gi];
        unsigned long long m = (unsigned long long)(nz_bits4(q.x) | (nz_bits4(q.y) << 4) | (nz_bits4(q.z) << 8) | (nz_bits4(q.w) << 12));
        m <<= 16 * (int)(gi & 3);
        m |= __shfl_xor(m, 1);
        m |= __shfl_xor(m, 2);
        if ((gi & 3) == 0) (second ? out_b : out_a)[gi >> 2] = m;
    }
}

// ---- bit-packed transport of boolean texel maps (the view-parallel all-gather record carries the raw and the K shrunk
// visibility maps of a view as 64-texel words -- 128 KiB per 1024^2 map instead of 1 MiB).  Bit i of word w = (in[64 w + i] != 0),
// i.e. as a byte stream: bit b of byte j = element 8 j + b.  n must be a multiple of 64.
__global__ __launch_bounds__(256) void k_unpack_bits(const unsigned long long* __restrict__ in, long long nwords, uint8_t* __restrict__ out) {
    // thread = 16 output bytes (a quarter word): one 16-byte store per thread, whole 128-byte lines per 8 threads
    const long long t0 = blockIdx.x * (long long)blockDim.x + threadIdx.x, nt = (long long)gridDim.x * blockDim.x;
    for (long long g = t0; g < nwords * 4; g += nt) {
        const unsigned int m = (unsigned int)(in[g >> 2] >> (16 * (int)(g & 3))) & 0xffffu;
        uint4 q;
        unsigned int* qq = reinterpret_cast<unsigned int*>(&q);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned int b = (m >> (4 * k)) & 0xfu;
            qq[k] = (b & 1u) | ((b & 2u) << 7) | ((b & 4u) << 14) | ((b & 8u) << 21);
        }
        reinterpret_cast<uint4*>(out)[g] = q;
    }
}
extern "C" int pdhip_pack_bits(const uint8_t* in, long long n, uint64_t* out, void* stream) {
    PD_REQUIRE(in && out && n > 0 && n % 64 == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0, "pdhip_pack_bits: n must be a multiple of 64, input 16-byte aligned");
    const long long groups = n / 16;
    k_pack_bits16<<<(int)std::min<long long>((groups + 255) / 256, 4096), 256, 0, as_stream(stream)>>>(in, groups, reinterpret_cast<unsigned long long*>(out), nullptr, 0, nullptr);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}
extern "C" int pdhip_unpack_bits(const uint64_t* in, long long n, uint8_t* out, void* stream) {
    PD_REQUIRE(in && out && n > 0 && n % 64 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0, "pdhip_unpack_bits: n must be a multiple of 64, output 16-byte aligned");
    const long long nwords = n / 64;
    k_unpack_bits<<<(int)std::min<long long>((nwords * 4 + 255) / 256, 4096), 256, 0, as_stream(stream)>>>(reinterpret_cast<const unsigned long long*>(in), nwords, out);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}

#define NBF_RB 32
__device__ __forceinline__ unsigned long long nbf_word(const unsigned long long* __restrict__ bits, int A, int W64, int y, int w) {
    return (y >= 0 && y < A && w >= 0 && w < W64) ? bits[(size_t)y * W64 + w] : 0ull;
}
// Scharr response != 0 for the 64 texels of word w in row y (zero padding)
__device__ __forceinline__ unsigned long long nbf_edge_word(const unsigned long long* __restrict__ bits, int A, int W64, int y, int w) {
    unsigned long long up[3], mid[3], dn[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        up[k] = nbf_word(bits, A, W64, y - 1, w - 1 + k);
        mid[k] = nbf_word(bits, A, W64, y, w - 1 + k);
        dn[k] = nbf_word(bits, A, W64, y + 1, w - 1 + k);
    }
    // column x-1 / x+1 aligned to bit position x
    const unsigned long long a00 = (up[1] << 1) | (up[0] >> 63), a02 = (up[1] >> 1) | (up[2] << 63), a01 = up[1];
    const unsigned long long a10 = (mid[1] << 1) | (mid[0] >> 63), a12 = (mid[1] >> 1) | (mid[2] << 63);
    const unsigned long long a20 = (dn[1] << 1) | (dn[0] >> 63), a22 = (dn[1] >> 1) | (dn[2] << 63), a21 = dn[1];
    // gx = 3 (a02 - a00) + 10 (a12 - a10) + 3 (a22 - a20);  gy = 3 (a20 - a00) + 10 (a21 - a01) + 3 (a22 - a02)
    const unsigned long long gx = (a12 ^ a10) | ((a02 ^ a22) ^ (a00 ^ a20)) | ((a02 & a22) ^ (a00 & a20));
    const unsigned long long gy = (a21 ^ a01) | ((a20 ^ a22) ^ (a00 ^ a02)) | ((a20 & a22) ^ (a00 & a02));
    return gx | gy;
}

__global__ __launch_bounds__(256) void k_nbf_bits(const unsigned long long* __restrict__ visb, const unsigned long long* __restrict__ maskb,
                                                  int A, int r, uint8_t* __restrict__ out, int vps) {
    extern __shared__ unsigned long long s_nbf[];           // [rows][W64] edges, then [rows][W64] horizontally dilated
    const int W64 = A >> 6, v = blockIdx.y, y0 = blockIdx.x * NBF_RB;
    maskb += (size_t)(v / vps) * A * W64;                   // several atlases per call: view v belongs to shape v / vps
    const int rows = NBF_RB + 2 * r;
    unsigned long long* e = s_nbf;
    unsigned long long* h = s_nbf + (size_t)rows * W64;
    const unsigned long long* vb = visb + (size_t)v * A * W64;
    for (int i = threadIdx.x; i < rows * W64; i += blockDim.x) {
        const int row = i / W64, w = i - row * W64, y = y0 - r + row;
        e[i] = (y >= 0 && y < A) ? (nbf_edge_word(vb, A, W64, y, w) & ~nbf_edge_word(maskb, A, W64, y, w)) : 0ull;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < rows * W64; i += blockDim.x) {
        const int row = i / W64, w = i - row * W64;
        const unsigned long long c = e[i], lft = w > 0 ? e[i - 1] : 0ull, rgt = w + 1 < W64 ? e[i + 1] : 0ull;
        unsigned long long acc = c;
        for (int k = 1; k <= r; ++k) acc |= (c << k) | (lft >> (64 - k)) | (c >> k) | (rgt << (64 - k));
        h[i] = acc;
    }
    __syncthreads();
    // output: 16 texels (a quarter word) per thread and step -> one 16-byte store each, 1 KB per wavefront instruction (a byte per
    // lane made this kernel store-issue bound: 76 us for 8 MB)
    const int Q16 = A >> 4;                                       // 16-texel pieces per row
    for (int i = threadIdx.x; i < NBF_RB * Q16; i += blockDim.x) {
        const int yy = i / Q16, c = i - yy * Q16, w = c >> 2, y = y0 + yy;
        if (y >= A) break;
        unsigned long long border = 0ull;
        for (int k = 0; k <= 2 * r; ++k) border |= h[(size_t)(yy + k) * W64 + w];
        const unsigned int keep = (unsigned int)((vb[(size_t)y * W64 + w] & ~border) >> ((c & 3) * 16)) & 0xffffu;
        uint4 o;                                                  // bit j -> byte j: four bits at a time, (b * 0x00204081) & 0x01010101
        o.x = ((keep & 15u) * 0x00204081u) & 0x01010101u; o.y = (((keep >> 4) & 15u) * 0x00204081u) & 0x01010101u;
        o.z = (((keep >> 8) & 15u) * 0x00204081u) & 0x01010101u; o.w = (((keep >> 12) & 15u) * 0x00204081u) & 0x01010101u;
        *reinterpret_cast<uint4*>(out + ((size_t)v * A + y) * A + (size_t)c * 16) = o;
    }
}

static int nbf_impl(const uint8_t* mask, const uint8_t* visibility, int V, int vps, int A, const int32_t* kernels, int K, uint8_t* out,
                    uint8_t* ws, void* stream);
extern "C" int pdhip_nbf_shrink(const uint8_t* mask, const uint8_t* visibility, int V, int A, const int32_t* kernels,
                                int K, uint8_t* out, uint8_t* ws, void* stream) {
    return nbf_impl(mask, visibility, V, V > 0 ? V : 1, A, kernels, K, out, ws, stream);
}
// S atlases with V views each in ONE set of launches: mask [S,A,A], visibility [S*V,A,A], out [K][S*V][A][A] (level-major over ALL
// views), workspace 2 * S * V * A * A bytes.
extern "C" int pdhip_nbf_shrink_shapes(const uint8_t* mask, const uint8_t* visibility, int V, int S, int A, const int32_t* kernels,
                                       int K, uint8_t* out, uint8_t* ws, void* stream) {
    PD_REQUIRE(S >= 1 && V >= 1, "pdhip_nbf_shrink_shapes: bad sizes");
    return nbf_impl(mask, visibility, S * V, V, A, kernels, K, out, ws, stream);
}
static int nbf_impl(const uint8_t* mask, const uint8_t* visibility, int V, int vps, int A, const int32_t* kernels, int K, uint8_t* out,
                    uint8_t* ws, void* stream) {
    const int S = V / vps;
    PD_REQUIRE(V > 0 && A > 0 && K > 0 && kernels, "pdhip_nbf_shrink: bad sizes");
    PD_REQUIRE(mask && visibility && out, "pdhip_nbf_shrink: null pointer");
    hipStream_t s = as_stream(stream);
    const size_t n = (size_t)V * A * A;
    if (kernels[0] == 0) {                      // NBF off (unproject.py:436-437): single level == raw visibility
        PD_HIP(hipMemcpyAsync(out, visibility, n, hipMemcpyDeviceToDevice, s));
        return PDHIP_OK;
    }
    PD_REQUIRE(ws, "pdhip_nbf_shrink: workspace required");
    for (int k = 0; k < K; ++k)
        PD_REQUIRE(kernels[k] >= 1 && (kernels[k] & 1), "pdhip_nbf_shrink: kernel sizes must be odd and >= 1 (got %d)", kernels[k]);
    uint8_t* edges = ws;
    uint8_t* tmp = ws + n;
    dim3 g(min(cdiv((long long)A * A, 256), 2048), V);
    bool bits_ok = (A % 64 == 0) && A <= 4096 && V >= 2;            // bitmaps live in the byte workspace: (V + 1) A^2 / 8 <= 2 V A^2
    for (int k = 0; k < K; ++k) bits_ok = bits_ok && (kernels[k] - 1) / 2 < 64 && (size_t)(NBF_RB + kernels[k] - 1) * (A / 64) * 16 <= 64 * 1024;
    if (bits_ok) {
        unsigned long long* visb = reinterpret_cast<unsigned long long*>(ws);
        unsigned long long* maskb = visb + (size_t)V * A * (A / 64);
        const long long vw = (long long)V * A * (A / 64), mw = (long long)S * A * (A / 64);
        if ((((uintptr_t)visibility | (uintptr_t)mask) & 15) == 0 && vw % 16 == 0 && mw % 16 == 0) {
            k_pack_bits16<<<min(cdiv((vw + mw) * 4, 256), 4096), 256, 0, s>>>(visibility, vw * 4, visb, mask, mw * 4, maskb);
        } else {
            k_pack_bits<<<min(cdiv(vw * 64, 256), 4096), 256, 0, s>>>(visibility, vw, visb);
            k_pack_bits<<<min(cdiv(mw * 64, 256), 4096), 256, 0, s>>>(mask, mw, maskb);
        }
        for (int k = 0; k < K; ++k) {
            int same = -1;
            for (int j = 0; j < k; ++j) if (kernels[j] == kernels[k]) { same = j; break; }
            if (same >= 0) {
                PD_HIP(hipMemcpyAsync(out + (size_t)k * n, out + (size_t)same * n, n, hipMemcpyDeviceToDevice, s));
                continue;
            }
            const int r = (kernels[k] - 1) / 2;
            const size_t smem = (size_t)2 * (NBF_RB + 2 * r) * (A / 64) * sizeof(unsigned long long);
            k_nbf_bits<<<dim3(cdiv(A, NBF_RB), V), 256, smem, s>>>(visb, maskb, A, r, out + (size_t)k * n, vps);
        }
        PD_LAUNCH_CHECK();
        return PDHIP_OK;
    }
    k_nbf_edges<<<g, 256, 0, s>>>(mask, visibility, A, edges, vps);
    for (int k = 0; k < K; ++k) {
        int same = -1;                              // the reference's list repetition yields identical levels: copy them
        for (int j = 0; j < k; ++j) if (kernels[j] == kernels[k]) { same = j; break; }
        if (same >= 0) {
            PD_HIP(hipMemcpyAsync(out + (size_t)k * n, out + (size_t)same * n, n, hipMemcpyDeviceToDevice, s));
            continue;
        }
        int r = (kernels[k] - 1) / 2;
        k_nbf_dilate_h<<<g, 256, 0, s>>>(edges, A, r, tmp);
        k_nbf_dilate_v_shrink<<<g, 256, 0, s>>>(tmp, visibility, A, r, out + (size_t)k * n);
    }
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}

// ---- O1: the `others/shrink_per_view_edge/{v}.png` triptychs the reference writes next to N1-N3 (unproject.py:459-474):
// [ visibility, chart-background edges red, view edges blue | view edge mask | border mask of the last kernel ] side by side with
// 10 white columns between the panels (utils_2d.cat_images), rows reversed (`cat[:, ::-1, :]`), 8-bit RGB, HWC.
__global__ void k_nbf_bg_edges(const uint8_t* __restrict__ mask, int A, uint8_t* __restrict__ bg) {
    const size_t n = (size_t)A * A;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(idx / A), x = (int)(idx - (size_t)y * A);
        bg[idx] = scharr_edge(mask, A, y, x) ? 1 : 0;
    }
}
__global__ void k_nbf_triptych(const uint8_t* __restrict__ vis, const uint8_t* __restrict__ bg, const uint8_t* __restrict__ edges,
                               const uint8_t* __restrict__ dil_h, int A, int r, uint8_t* __restrict__ out) {
    const int v = blockIdx.y, Wd = 3 * A + 20;
    const size_t n = (size_t)A * A, total = (size_t)A * Wd;
    uint8_t* o = out + (size_t)v * total * 3;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int yo = (int)(idx / Wd), xo = (int)(idx - (size_t)yo * Wd);
        const int y = A - 1 - yo;
        uint8_t c0 = 255, c1 = 255, c2 = 255;
        if (xo < A) {
            const size_t t = (size_t)y * A + xo;
            const uint8_t g = vis[(size_t)v * n + t] ? 255 : 0;
            c0 = c1 = c2 = g;
            if (bg[t]) { c0 = 255; c1 = 0; c2 = 0; }
            if (edges[(size_t)v * n + t]) { c0 = 0; c1 = 0; c2 = 255; }
        } else if (xo >= A + 10 && xo < 2 * A + 10) {
            c0 = c1 = c2 = edges[(size_t)v * n + (size_t)y * A + (xo - A - 10)] ? 255 : 0;
        } else if (xo >= 2 * A + 20) {
            const int x = xo - 2 * A - 20;
            const int lo = max(0, y - r), hi = min(A - 1, y + r);
            uint8_t b = 0;
            for (int k = lo; k <= hi; ++k) b |= dil_h[(size_t)v * n + (size_t)k * A + x];
            c0 = c1 = c2 = b ? 255 : 0;
        }
        o[idx * 3] = c0; o[idx * 3 + 1] = c1; o[idx * 3 + 2] = c2;
    }
}
/* ws: (2 V + 1) A^2 bytes; out: [V][A][3A+20][3] u8 */
extern "C" int pdhip_nbf_triptych(const uint8_t* mask, const uint8_t* visibility, int V, int A, int kernel, uint8_t* out,
                                  uint8_t* ws, void* stream) {
    PD_REQUIRE(mask && visibility && out && ws && V > 0 && A > 0, "pdhip_nbf_triptych: bad arguments");
    PD_REQUIRE(kernel >= 1 && (kernel & 1), "pdhip_nbf_triptych: kernel size must be odd and >= 1 (got %d)", kernel);
    hipStream_t s = as_stream(stream);
    const size_t n = (size_t)V * A * A;
    uint8_t* edges = ws;
    uint8_t* tmp = ws + n;
    uint8_t* bg = ws + 2 * n;
    dim3 g(min(cdiv((long long)A * A, 256), 2048), V);
    const int r = (kernel - 1) / 2;
    k_nbf_edges<<<g, 256, 0, s>>>(mask, visibility, A, edges, V);
    k_nbf_bg_edges<<<g.x, 256, 0, s>>>(mask, A, bg);
    k_nbf_dilate_h<<<g, 256, 0, s>>>(edges, A, r, tmp);
    k_nbf_triptych<<<dim3(min(cdiv((long long)A * (3 * A + 20), 256), 4096), V), 256, 0, s>>>(visibility, bg, edges, tmp, A, r, out);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}

// ------------------------------------------------------------------------------ Uq3 + Uq4
__global__ void k_view_select_blend(const float* __restrict__ cams, int V, const float* __restrict__ gb_pos,
                                    const uint8_t* __restrict__ mask, const int64_t* __restrict__ face_id, int A,
                                    const float* __restrict__ f_normals, const float* __restrict__ base_dirs,
                                    const float* __restrict__ uv_centers, const float* __restrict__ uv_scales, float pad9,
                                    const float* __restrict__ scale_factors, const uint8_t* __restrict__ shrinked, int K,
                                    const uint8_t* __restrict__ visibility, int complete,
                                    const float* __restrict__ inpainted, int r, float* __restrict__ atlas,
                                    uint8_t* __restrict__ painted, int32_t* __restrict__ view_ids, int S, int F) {
    // S atlases (shapes) in one launch: the per-texel maps and the face normals ([S,F,3]) are stacked by shape, the per-view arrays hold
    // S*V entries (view g = sh * V + v), the shrunk levels are [K][S*V][A][A]; view ids stay LOCAL to the shape (0 .. V-1)
    const size_t n = (size_t)A * A, total = n * (size_t)S;
    for (size_t gidx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; gidx < total; gidx += (size_t)gridDim.x * blockDim.x) {
        const size_t sh = gidx / n, idx = gidx - sh * n;
        const size_t VT = (size_t)S * V;                  // views of the whole call
        float o0 = 0.f, o1 = 0.f, o2 = 0.f;
        uint8_t pt = 0;
        int vid = -1;
        if (mask[gidx]) {
            // candidate views: level 0, then looser levels only while nothing is visible (unproject.py:324-356)
            uint32_t cand = 0;
            for (int k = 0; k < K; ++k) {
                if (k > 0 && cand) break;
                for (int v = 0; v < V; ++v) cand |= (shrinked[((size_t)k * VT + sh * V + v) * n + idx] ? 1u : 0u) << v;
            }
            if (complete && !cand)
                for (int v = 0; v < V; ++v) cand |= (visibility[(sh * V + v) * n + idx] ? 1u : 0u) << v;
            // similarity of the face normal with every view direction, softmax in view order
            const int64_t f = face_id[gidx];
            const float* fnp = f_normals + 3 * (sh * (size_t)F + (size_t)f);
            const float n0 = fnp[0], n1 = fnp[1], n2 = fnp[2];
            float sim[MAXV];
            float mx = -INFINITY;
            for (int v = 0; v < V; ++v) {
                sim[v] = (n0 * base_dirs[3 * v] + n1 * base_dirs[3 * v + 1]) + n2 * base_dirs[3 * v + 2];
                mx = fmaxf(mx, sim[v]);
            }
            float sum = 0.f;
            for (int v = 0; v < V; ++v) {
                sim[v] = (float)exp((double)(sim[v] - mx));
                sum = sum + sim[v];
            }
            float best = -INFINITY;
            int bi = 0;
            for (int v = 0; v < V; ++v) {
                float w = ((cand >> v) & 1u) ? sim[v] / sum : -100.0f;
                if (w > best || (w != w && best == best)) { best = w; bi = v; }       // (torch.argmax: the first NaN is the maximum)
            }
            vid = bi;
            if (!complete && !cand) vid = -100;
            if (vid >= 0) {
                const Cam c = load_cam(cams + 16 * vid);
                const size_t g = sh * V + vid;
                float xn, yn, zn;
                cam_transform(c, gb_pos[3 * gidx], gb_pos[3 * gidx + 1], gb_pos[3 * gidx + 2], xn, yn, zn);
                float u = (((xn - uv_centers[2 * g]) / uv_scales[g]) * scale_factors[g]) * pad9 + 0.5f;
                float w = (((yn - uv_centers[2 * g + 1]) / uv_scales[g]) * scale_factors[g]) * pad9 + 0.5f;
                int col = clip_to_int(u * (float)r, r - 1);
                int row = clip_to_int(w * (float)r, r - 1);
                const float* img = inpainted + g * 3 * r * r + (size_t)(r - 1 - row) * r + col;
                o0 = img[0]; o1 = img[(size_t)r * r]; o2 = img[2 * (size_t)r * r];
                pt = 1;
            }
        }
        atlas[3 * gidx] = o0; atlas[3 * gidx + 1] = o1; atlas[3 * gidx + 2] = o2;
        painted[gidx] = pt;
        view_ids[gidx] = vid;
    }
}

// The same with the view count a compile-time constant (round 5).  (i) The VC candidate bytes of a level, the face id and the position
// are requested together (a run-time loop asked for them one at a time).  (ii) The selected view is the FIRST maximum of
// w_v = float(exp(double(sim_v - mx))) / sum over the candidates; x -> float(exp(double(x - mx))) and y -> y / sum are monotone
// non-decreasing, so that is the first maximum of sim_v itself unless an EARLIER candidate with a smaller sim rounds to the same
// weight.  If every earlier candidate is below the maximum by >= 1e-5 its weight is smaller by a factor >= 1 + 8e-6 -- a hundred f32
// ulps, no rounding of the exponential, of the sum or of the quotient can close that -- and the eight f64 exponentials (more than half
// of the generic kernel's time) are skipped.  Anything closer, and any non-finite similarity, takes the generic expressions.
template <int VC>
__global__ __launch_bounds__(256) void k_view_select_blend_v(const float* __restrict__ cams, const float* __restrict__ gb_pos,
                                    const uint8_t* __restrict__ mask, const int64_t* __restrict__ face_id, int A,
                                    const float* __restrict__ f_normals, const float* __restrict__ base_dirs,
                                    const float* __restrict__ uv_centers, const float* __restrict__ uv_scales, float pad9,
                                    const float* __restrict__ scale_factors, const uint8_t* __restrict__ shrinked, int K,
                                    const uint8_t* __restrict__ visibility, int complete,
                                    const float* __restrict__ inpainted, int r, float* __restrict__ atlas,
                                    uint8_t* __restrict__ painted, int32_t* __restrict__ view_ids, int S, int F) {
    const size_t n = (size_t)A * A, total = n * (size_t)S;
    for (size_t gidx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; gidx < total; gidx += (size_t)gridDim.x * blockDim.x) {
        const size_t sh = gidx / n, idx = gidx - sh * n;
        const size_t VT = (size_t)S * VC;
        float o0 = 0.f, o1 = 0.f, o2 = 0.f;
        uint8_t pt = 0;
        int vid = -1;
        if (mask[gidx]) {
            uint8_t sb[VC];
#pragma unroll
            for (int v = 0; v < VC; ++v) sb[v] = shrinked[(sh * VC + v) * n + idx];
            const int64_t f = face_id[gidx];
            const float px_ = gb_pos[3 * gidx], py_ = gb_pos[3 * gidx + 1], pz_ = gb_pos[3 * gidx + 2];
            uint32_t cand = 0;
#pragma unroll
            for (int v = 0; v < VC; ++v) cand |= (sb[v] ? 1u : 0u) << v;
            for (int k = 1; k < K && !cand; ++k) {
#pragma unroll
                for (int v = 0; v < VC; ++v) sb[v] = shrinked[((size_t)k * VT + sh * VC + v) * n + idx];
#pragma unroll
                for (int v = 0; v < VC; ++v) cand |= (sb[v] ? 1u : 0u) << v;
            }
            if (complete && !cand) {
#pragma unroll
                for (int v = 0; v < VC; ++v) sb[v] = visibility[(sh * VC + v) * n + idx];
#pragma unroll
                for (int v = 0; v < VC; ++v) cand |= (sb[v] ? 1u : 0u) << v;
            }
            const float* fnp = f_normals + 3 * (sh * (size_t)F + (size_t)f);
            const float n0 = fnp[0], n1 = fnp[1], n2 = fnp[2];
            float sim[VC];
            float mx = -INFINITY;
            bool finite = true;
#pragma unroll
            for (int v = 0; v < VC; ++v) {
                sim[v] = (n0 * base_dirs[3 * v] + n1 * base_dirs[3 * v + 1]) + n2 * base_dirs[3 * v + 2];
                mx = fmaxf(mx, sim[v]);
                finite = finite && fabsf(sim[v]) <= 3.0e38f;
            }
            // first maximum of sim over the candidates, and the largest sim of a candidate BEFORE it
            float s1 = -INFINITY, before = -INFINITY;
            int bi = 0;
#pragma unroll
            for (int v = 0; v < VC; ++v)
                if (((cand >> v) & 1u) && sim[v] > s1) { before = s1; s1 = sim[v]; bi = v; }
            //  the shortcut also needs the winner's weight in f32's NORMAL range: with unnormalised face normals (|n| >~ 50)
            // sim - mx can fall below -87 ... -104, exp underflows to a denormal or 0 and the reference's argmax then takes the FIRST
            // zero-weight candidate, not the largest similarity; exp(-80) / 8 is still normal, below that the generic expressions decide
            if (cand != 0u && !(finite && s1 - before >= 1.0e-5f && s1 - mx > -80.0f)) {
                // (generic expressions, same order)
                float ew[VC];
                float sum = 0.f;
#pragma unroll
                for (int v = 0; v < VC; ++v) {
                    ew[v] = (float)exp((double)(sim[v] - mx));
                    sum = sum + ew[v];
                }
                float best = -INFINITY;
                bi = 0;
#pragma unroll
                for (int v = 0; v < VC; ++v) {
                    const float w = ((cand >> v) & 1u) ? ew[v] / sum : -100.0f;
                    if (w > best || (w != w && best == best)) { best = w; bi = v; }   // (torch.argmax: the first NaN is the maximum)
                }
            }
            vid = bi;
            if (!complete && !cand) vid = -100;
            if (vid >= 0) {
                const Cam c = load_cam(cams + 16 * vid);
                const size_t g = sh * VC + vid;
                float xn, yn, zn;
                cam_transform(c, px_, py_, pz_, xn, yn, zn);
                float u = (((xn - uv_centers[2 * g]) / uv_scales[g]) * scale_factors[g]) * pad9 + 0.5f;
                float w = (((yn - uv_centers[2 * g + 1]) / uv_scales[g]) * scale_factors[g]) * pad9 + 0.5f;
                int col = clip_to_int(u * (float)r, r - 1);
                int row = clip_to_int(w * (float)r, r - 1);
                const float* img = inpainted + g * 3 * r * r + (size_t)(r - 1 - row) * r + col;
                o0 = img[0]; o1 = img[(size_t)r * r]; o2 = img[2 * (size_t)r * r];
                pt = 1;
            }
        }
        atlas[3 * gidx] = o0; atlas[3 * gidx + 1] = o1; atlas[3 * gidx + 2] = o2;
        painted[gidx] = pt;
        view_ids[gidx] = vid;
    }
}

// S atlases with V views each in ONE launch: gb_pos / mask / face_id / atlas / painted / view_ids stacked [S,A,A,..], f_normals [S,F,3],
// cam_params / base_dirs [V] shared by the shapes, uv_centers / uv_scales / scale_factors / visibility / inpainted with S*V leading entries,
// shrinked [K][S*V][A][A]; view ids are local to a shape.  S = 1 (F unused): pdhip_view_select_blend.
extern "C" int pdhip_view_select_blend_shapes(const float* cam_params, int V, int S, const float* gb_pos, const uint8_t* mask,
                                              const int64_t* face_id, int A, const float* f_normals, int F, const float* base_dirs,
                                              const float* uv_centers, const float* uv_scales, double padding,
                                              const float* scale_factors, const uint8_t* shrinked, int K,
                                              const uint8_t* visibility, int complete_unseen_by_projection,
                                              const float* inpainted, int r, float* atlas, uint8_t* painted,
                                              int32_t* view_ids, void* stream) {
    PD_REQUIRE(V > 0 && V <= MAXV && A > 0 && K > 0 && r > 0 && S >= 1 && (S == 1 || F > 0), "pdhip_view_select_blend: bad sizes V=%d A=%d K=%d r=%d S=%d", V, A, K, r, S);
    PD_REQUIRE(cam_params && gb_pos && mask && face_id && f_normals && base_dirs && uv_centers && uv_scales &&
                   scale_factors && shrinked && visibility && inpainted && atlas && painted && view_ids,
               "pdhip_view_select_blend: null pointer");
    const float pad9 = (float)(1.0 - 2.0 * padding);
    const int grid = min(cdiv((long long)A * A * S, 256), 4096 * min(S, 8));
    if (V == 8 && g_unproject_generic == 0)
        k_view_select_blend_v<8><<<grid, 256, 0, as_stream(stream)>>>(
            cam_params, gb_pos, mask, face_id, A, f_normals, base_dirs, uv_centers, uv_scales, pad9, scale_factors,
            shrinked, K, visibility, complete_unseen_by_projection, inpainted, r, atlas, painted, view_ids, S, S == 1 ? 0 : F);
    else
        k_view_select_blend<<<grid, 256, 0, as_stream(stream)>>>(
            cam_params, V, gb_pos, mask, face_id, A, f_normals, base_dirs, uv_centers, uv_scales, pad9, scale_factors,
            shrinked, K, visibility, complete_unseen_by_projection, inpainted, r, atlas, painted, view_ids, S, S == 1 ? 0 : F);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}
extern "C" int pdhip_view_select_blend(const float* cam_params, int V, const float* gb_pos, const uint8_t* mask,
                                       const int64_t* face_id, int A, const float* f_normals, const float* base_dirs,
                                       const float* uv_centers, const float* uv_scales, double padding,
                                       const float* scale_factors, const uint8_t* shrinked, int K,
                                       const uint8_t* visibility, int complete_unseen_by_projection,
                                       const float* inpainted, int r, float* atlas, uint8_t* painted,
                                       int32_t* view_ids, void* stream) {
    return pdhip_view_select_blend_shapes(cam_params, V, 1, gb_pos, mask, face_id, A, f_normals, 0, base_dirs, uv_centers, uv_scales, padding,
                                          scale_factors, shrinked, K, visibility, complete_unseen_by_projection, inpainted, r, atlas, painted,
                                          view_ids, stream);
}
// ------------------------------------------------------------------------------ compaction
__global__ void k_compact_count(const uint8_t* __restrict__ mask, int A, int32_t* __restrict__ row_cnt) {
    const int row = blockIdx.x;
    __shared__ int s[4];
    int c = 0;
    for (int x = threadIdx.x; x < A; x += blockDim.x) c += mask[(size_t)row * A + x] ? 1 : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) row_cnt[row] = s[0] + s[1] + s[2] + s[3];
}

__global__ void k_compact_scan(int32_t* __restrict__ row_cnt, int A, int32_t* __restrict__ count_dev) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int acc = 0;
        for (int i = 0; i < A; ++i) { int c = row_cnt[i]; row_cnt[i] = acc; acc += c; }
        row_cnt[A] = acc;
        *count_dev = acc;
    }
}

__global__ void k_compact_write(const float* __restrict__ gb_pos, const uint8_t* __restrict__ mask, int A,
                                const int32_t* __restrict__ view_ids, const int32_t* __restrict__ row_off,
                                float* __restrict__ points, int64_t* __restrict__ coords, int64_t* __restrict__ pvid) {
    const int row = blockIdx.x;                 // one wave per row: ranks via ballot prefix
    int base = row_off[row];
    const int lane = threadIdx.x;
    for (int x0 = 0; x0 < A; x0 += 64) {
        int x = x0 + lane;
        bool m = x < A && mask[(size_t)row * A + x];
        unsigned long long bal = __ballot(m);
        if (m) {
            int rank = base + __popcll(bal & ((1ull << lane) - 1ull));
            size_t idx = (size_t)row * A + x;
            if (points) { points[3 * rank] = gb_pos[3 * idx]; points[3 * rank + 1] = gb_pos[3 * idx + 1]; points[3 * rank + 2] = gb_pos[3 * idx + 2]; }
            if (coords) { coords[2 * rank] = row; coords[2 * rank + 1] = x; }
            if (pvid) pvid[rank] = view_ids[idx];
        }
        base += __popcll(bal);
    }
}

extern "C" int pdhip_compact_texels(const float* gb_pos, const uint8_t* mask, int A, const int32_t* view_ids,
                                    float* points, int64_t* coords, int64_t* point_view_ids, int32_t* count_dev,
                                    int32_t* ws, void* stream) {
    PD_REQUIRE(A > 0 && mask && ws && count_dev, "pdhip_compact_texels: bad arguments");
    PD_REQUIRE(!points || gb_pos, "pdhip_compact_texels: points requested without gb_pos");
    PD_REQUIRE(!point_view_ids || view_ids, "pdhip_compact_texels: point_view_ids requested without view_ids");
    hipStream_t s = as_stream(stream);
    k_compact_count<<<A, 256, 0, s>>>(mask, A, ws);
    k_compact_scan<<<1, 64, 0, s>>>(ws, A, count_dev);
    k_compact_write<<<A, 64, 0, s>>>(gb_pos, mask, A, view_ids, ws, points, coords, point_view_ids);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}
