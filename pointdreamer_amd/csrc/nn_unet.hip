// Rows U1 + D1: the guided-diffusion UNet engine and the DDNM sampler behind the C ABI.
// Graph construction mirrors UNetModel.__init__ (models/DDNM/guided_diffusion/unet.py:481-617) as configured
// by script_util.create_model (:130-185); weights are loaded under the reference's state-dict names
// (`input_blocks.N.M.in_layers.2.weight`, ...) and repacked on the device into the kernels' layouts:
//   conv3x3 / conv1x1 / conv1d(k=1) weights -> f16 [Cout_pad][taps*Cin] (k = tap*Cin + c), biases f16-rounded f32,
//   time_embed / emb_layers Linear and all GroupNorm affine and the `out` head stay f32 (fp16_util.py:15-22).
// Activations are NHWC f16 in a bump arena sized at create() for max_batch (288 GB HBM: no reuse games);
// every forward issues the same launches at the same addresses, asynchronously on the caller's stream.
#include "nn_common.h"
#include <map>
#include <string>
#include <vector>
#include <cmath>
#include <cstring>
using namespace pdhip;
using namespace pdnn;

namespace {

struct ConvW { half_t* w = nullptr; float* b = nullptr; int cin = 0, cout = 0, cout_pad = 0, taps = 0; bool have_w = false, have_b = false;
               half_t* wf = nullptr; };   // wf: fragment-major copy for the row-resident kernel (nn_conv_rr.hip), the 3x3 convs of the <= 64^2 levels; packed at load time
struct NormW { float* g = nullptr; float* b = nullptr; int c = 0; bool have_g = false, have_b = false; };
struct ResB { std::string name; int cin, cout, mode; NormW n1, n2; ConvW c1, c2, skip; long long emb_off; bool has_skip; bool have_ew = false, have_eb = false;
              half_t* c2s_w = nullptr; float* c2s_b = nullptr; bool c2s_ready = false; half_t* c2s_wf = nullptr; int hw = 0; };   // hw: resolution the block's convs run at   // conv2 with the skip 1x1 appended to its K loop (conv_sk_skip), built by pdhip_unet_load_tensor once its four sources are loaded
struct AttB { std::string name; int c; NormW n; ConvW qkv, proj; };
struct Block { int kind; int idx; };          // 0 conv_in, 1 res, 2 attn

// gn_part: octet partial sums fused into the producing conv's epilogue ([N][chunks][C/8][2]); for a channel concat the
// second source's partials are gn_partB (first Ca channels come from gn_part)
struct Act { half_t* p; int C, H, W; const float* gn_part = nullptr; int gn_chunks = 0; const float* gn_partB = nullptr; int Ca = 0; int gn_chunksB = 0;
             half_t* p2 = nullptr; };   // p2: virtual channel concat [p (Ca channels) | p2 (C - Ca channels)], never materialised

}  // namespace
// tuning / test hook (pdhip_debug_set_fuse_gn).  Default 0 = stand-alone GroupNorm-apply passes: MEASURED faster.  The in-conv
// transform removes 9.4 ms of gn_apply per batch-32 forward but costs the halo conv 23 % (tools/bench_conv.py --apply, lab builds:
// fetch of the staged pieces -4.5 %, the in-place ds_write_b128 -10 % -- a store occupies the SIMD's LDS path for 13 cycles in
// front of the fragment reads the MFMAs wait for -- arithmetic -8 %): 76.2 vs 71.4 ms per DDNM step at batch 32.
namespace pdnn { thread_local int g_fuse_gn = 0; }
// tuning / test hook (pdhip_debug_set_fold_resample): 1 (default) = the x branch of up / down ResBlocks is never materialised by a
// k_resample pass: AvgPool2d(2) of the raw input comes out of the GroupNorm-apply kernel that reads the same pixels anyway, the
// nearest x2 copy is replaced by index arithmetic in the residual read of the consuming conv (unsplit halo layers)
namespace pdnn { thread_local int g_fold_resample = 1; }
// tuning / test hook (pdhip_debug_set_fold_skip): 1 (default) = where a channel-changing ResBlock's conv2 runs in k_conv_sk (the small-M
// layers) the block's skip 1x1 conv is appended to conv2's K loop (one launch, one rounding) instead of a launch of its own + a residual read
namespace pdnn { thread_local int g_fold_skip = 1; }
// tuning / test hook (pdhip_debug_set_rr_gn): largest image width at which a ResBlock conv routed to k_conv_rr also applies the GroupNorm (+ FiLM) + SiLU
// in front of it while staging (no k_gn_apply launch, no normalised tensor; bit-identical).  Default 0 = never -- measured (profiles/r06_rr_gn_ab.txt):
// stand-alone the 8^2 / 1024-channel conv pays +0.9 us for it (the element map runs under the weight stream) against a ~6 us launch, the 16^2 / 32^2
// convs +9-10 us (every 16-32-channel tile redoes the map for its whole pixel tile: 100 VALU cycles per element at one wave per SIMD); inside the
// forward the batch-1 DDNM step is 4.256 ms with it at 8^2 against 4.254-4.268 without, 4.266 at 16^2: no gain, so the two-pass form stays
namespace pdnn { thread_local int g_rr_gn = 0; }
// tuning / test hook (pdhip_debug_set_fold_finalize): largest batch at which GroupNorm-apply reduces the conv epilogues' octet
// partials itself instead of reading the output of a k_gn_finalize_oct launch (0 = never); above that batch the fold is kept for the
// tensors whose producers left at most g_fold_finalize_chunks chunks per image (the 32^2 ... 8^2 levels: the re-reduction is a few
// loads per thread against a 5.8 us launch, 2 020 of them per 20 forwards at batch 32)
namespace pdnn { thread_local int g_fold_finalize = 8; thread_local int g_fold_finalize_chunks = 16; }
// tuning / test hook (pdhip_debug_set_fuse_skip): 0 = never, 1 (default) = a channel-changing ResBlock whose skip conv is C -> 256 over
// at least g_fuse_skip_min_tiles pixel tiles (default 1: measured a gain at every batch) runs GroupNorm-apply and the skip 1x1 as ONE
// pass over its input (k_gn_skip), 2 = always
namespace pdnn { thread_local int g_fuse_skip = 1; thread_local int g_fuse_skip_min_tiles = 1; }
namespace {
struct Prof { std::vector<hipEvent_t> ev; std::vector<uint8_t> cls; size_t used = 0; double flops[2] = {0, 0}; bool on = false;
              int period = 1; long long forwards = 0; bool armed = false; };   // cls 0: halo 3x3 conv, 1: attention

}  // namespace

struct pdhip_unet {
    int image_size, mc, nres, head, out_ch, max_batch, ted;
    std::vector<int> mult, att_ds;
    std::vector<ResB> res;
    std::vector<AttB> att;
    std::vector<std::vector<Block>> input, output;
    std::vector<Block> middle;
    int final_ch;
    // parameters
    half_t* w_in = nullptr; float* b_in = nullptr; bool have_win = false, have_bin = false;
    float *te_w0 = nullptr, *te_b0 = nullptr, *te_w2 = nullptr, *te_b2 = nullptr; bool have_te[4] = {false, false, false, false};
    float *emb_w = nullptr, *emb_b = nullptr; long long emb_rows = 0;
    NormW out_norm; float* out_w = nullptr; float* out_b = nullptr; bool have_ow = false, have_ob = false;
    half_t* zero_page = nullptr;
    // workspace
    char* arena = nullptr; size_t arena_bytes = 0, arena_off = 0; bool arena_overflow = false;
    float *stats = nullptr, *gn_ws = nullptr; size_t gn_ws_floats = 0;
    float* ap_table = nullptr;                   // GroupNorm (+ FiLM) folded to (A, B) per (image, channel) for the conv that applies it while staging
    float *emb_silu = nullptr, *emb_tmp = nullptr, *emb_all = nullptr;
    // DDNM sampler: the embeddings of all (<= 100) schedule steps, computed in ONE pass over the 180 MB of emb_layers weights
    // and kept until the weights change (every image of a sampling step shares t, so the per-step forward reads one row)
    float *steps_t = nullptr, *steps_silu = nullptr, *steps_tmp = nullptr, *steps_emb = nullptr; int steps_cached = 0;
    half_t* head_wz = nullptr;                   // output-head weights as f16 hi/lo pairs (nn_head.hip)
    half_t* head_wz3 = nullptr;                  // the same for output channels 0..2 only: the DDNM sampler discards the variance
                                                 // channels of a learn_sigma net (diffusion.py:527-528: et = et[:, :3]) -- half the head
    float *t_dev = nullptr;
    float* splitk_ws = nullptr; size_t splitk_floats = 0;
    // sampler state
    float *sx = nullptr, *sy = nullptr, *set_ = nullptr, *smask = nullptr;
    Prof prof;
    std::vector<void*> owned;
};

namespace {

template <typename T>
int dalloc(pdhip_unet* u, T** p, size_t count) {
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, std::max<size_t>(count * sizeof(T), 256));
    if (e != hipSuccess) { set_error("hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e)); return PDHIP_E_NOMEM; }
    u->owned.push_back(q);
    *p = reinterpret_cast<T*>(q);
    return PDHIP_OK;
}
#define PD_TRY(expr) do { int rc_ = (expr); if (rc_ != PDHIP_OK) return rc_; } while (0)

int alloc_conv(pdhip_unet* u, ConvW& c, int cin, int cout, int taps) {
    c.cin = cin; c.cout = cout; c.taps = taps; c.cout_pad = ((cout + 127) / 128) * 128;
    PD_TRY(dalloc(u, &c.w, (size_t)c.cout_pad * taps * cin));
    PD_HIP(hipMemset(c.w, 0, (size_t)c.cout_pad * taps * cin * sizeof(half_t)));
    PD_TRY(dalloc(u, &c.b, (size_t)cout));
    return PDHIP_OK;
}
int alloc_norm(pdhip_unet* u, NormW& n, int c) {
    n.c = c;
    PD_TRY(dalloc(u, &n.g, (size_t)c));
    PD_TRY(dalloc(u, &n.b, (size_t)c));
    return PDHIP_OK;
}

// ---- weight repack kernels
__global__ void k_pack_conv(const void* __restrict__ src, int src_f16, int Cout, int Cin, int taps, half_t* __restrict__ dst) {
    // src [Cout][Cin][taps] (PyTorch OIHW flattened) -> dst [Cout][taps*Cin], k = tap*Cin + c
    const long long total = (long long)Cout * Cin * taps;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int tap = (int)(i % taps);
        const int c = (int)((i / taps) % Cin);
        const int o = (int)(i / ((long long)taps * Cin));
        const float v = src_f16 ? (float)reinterpret_cast<const half_t*>(src)[i] : reinterpret_cast<const float*>(src)[i];
        dst[(size_t)o * taps * Cin + (size_t)tap * Cin + c] = (half_t)v;
    }
}
__global__ void k_pack_conv_in(const void* __restrict__ src, int src_f16, int Cout, half_t* __restrict__ dst) {
    // src [Cout][3][3][3] (OIHW) -> dst [Cout][32], k = (ky*3+kx)*3 + c  (columns 27..31 stay zero)
    const int total = Cout * 27;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int tap = i % 9, c = (i / 9) % 3, o = i / 27;
        const float v = src_f16 ? (float)reinterpret_cast<const half_t*>(src)[i] : reinterpret_cast<const float*>(src)[i];
        dst[(size_t)o * 32 + tap * 3 + c] = (half_t)v;
    }
}
__global__ void k_pack_conv_f32(const void* __restrict__ src, int src_f16, int Cout, int Cin, int taps, float* __restrict__ dst) {
    const long long total = (long long)Cout * Cin * taps;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int tap = (int)(i % taps);
        const int c = (int)((i / taps) % Cin);
        const int o = (int)(i / ((long long)taps * Cin));
        const float v = src_f16 ? (float)reinterpret_cast<const half_t*>(src)[i] : reinterpret_cast<const float*>(src)[i];
        dst[(size_t)o * taps * Cin + (size_t)tap * Cin + c] = v;
    }
}
__global__ void k_copy_f32(const void* __restrict__ src, int src_f16, long long n, float* __restrict__ dst, int round_f16) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float v = src_f16 ? (float)reinterpret_cast<const half_t*>(src)[i] : reinterpret_cast<const float*>(src)[i];
        if (round_f16) v = (float)(half_t)v;
        dst[i] = v;
    }
}
int grid_for(long long n) { return (int)std::min<long long>((n + 255) / 256, 4096); }

// ---- arena
half_t* arena_take(pdhip_unet* u, size_t halfs) {
    size_t bytes = (halfs * sizeof(half_t) + 255) & ~(size_t)255;
    if (u->arena == nullptr) { u->arena_off += bytes; return nullptr; }      // dry run (sizing)
    if (u->arena_off + bytes > u->arena_bytes) { u->arena_overflow = true; return reinterpret_cast<half_t*>(u->arena); }   // (reported by forward_impl)
    half_t* p = reinterpret_cast<half_t*>(u->arena + u->arena_off);
    u->arena_off += bytes;
    return p;
}

struct Ctx { pdhip_unet* u; int N; hipStream_t s; bool dry; const float* film_base; long long film_stride; };

int prof_begin(Ctx& c, double flops, int cls = 0) {
    Prof& p = c.u->prof;
    if (!p.on || !p.armed || c.dry) return PDHIP_OK;
    if (p.used + 2 > p.ev.size()) {
        for (int i = 0; i < 256; ++i) { hipEvent_t e; PD_HIP(hipEventCreate(&e)); p.ev.push_back(e); p.cls.push_back(0); }
    }
    PD_HIP(hipEventRecord(p.ev[p.used], c.s));
    p.cls[p.used] = (uint8_t)cls;
    p.flops[cls] += flops;
    return PDHIP_OK;
}
int prof_end(Ctx& c) {
    Prof& p = c.u->prof;
    if (!p.on || !p.armed || c.dry) return PDHIP_OK;
    PD_HIP(hipEventRecord(p.ev[p.used + 1], c.s));
    p.used += 2;
    return PDHIP_OK;
}

int run_conv(Ctx& c, const Act& x, const ConvW& w, const half_t* residual, Act* out, bool want_gn = false,
             const float* apply_table = nullptr, int res_up = 0, int in_up = 0) {
    // in_up: x is the half-resolution input, the conv runs on its nearest x2 (out has twice x's height / width)
    const int cH = x.H << in_up, cW = x.W << in_up;
    *out = Act{nullptr, w.cout, cH, cW};
    out->p = arena_take(c.u, (size_t)c.N * cH * cW * w.cout);
    // octet partials: [N][chunks][cout/8][2] floats, chunks <= HW/16 (split-K reduce) -- sized for the finest chunking
    float* part = want_gn ? reinterpret_cast<float*>(arena_take(c.u, (size_t)c.N * ((cH * cW + 15) / 16) * (w.cout / 8) * 2 * 2)) : nullptr;
    if (c.dry) return PDHIP_OK;
    PD_REQUIRE(w.have_w && w.have_b, "unet: conv weights not loaded");
    // profile hook (bench.py roofline): the dominant kernel only -- the halo-resident 3x3 conv
    const bool prof = x.p2 == nullptr && conv_uses_halo(c.N, cH, cW, w.cin, w.cout, w.cout_pad, w.taps, c.u->splitk_floats) &&
                      conv3x3_halo_splits(c.N, cH, cW, w.cin, w.cout, w.cout_pad, c.u->splitk_floats) == 1;
    if (prof) PD_TRY(prof_begin(c, 2.0 * c.N * cH * cW * (double)w.cout * 9.0 * w.cin));
    // small batch, <= 64^2: the row-resident kernel (register-resident fragment-major weights, halo staged once per channel chunk)
    if (w.wf != nullptr && w.taps == 9 && x.p2 == nullptr && apply_table == nullptr && in_up == 0 && !prof) {
        const RrPlan pl = conv_rr_plan(c.N, cH, cW, w.cin, w.cout, 9, 0, c.u->splitk_floats);
        if (pl.variant != 0) {
            const RrIn in{x.p, nullptr, w.cin, w.cin, 0, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0, 0, 1e-5f};
            int chunks = 0;
            PD_TRY(conv_rr(pl, in, nullptr, 9, w.wf, w.b, residual, res_up, out->p, c.N, cH, cW, w.cout, c.u->splitk_ws, c.u->splitk_floats, part, &chunks, c.s));
            if (part != nullptr && chunks > 0) { out->gn_part = part; out->gn_chunks = chunks; out->Ca = w.cout; }
            return PDHIP_OK;
        }
    }
    int fused = 0;
    int rc = conv_igemm(x.p, w.w, w.b, residual, out->p, c.N, cH, cW, w.cin, w.cout, w.cout_pad, w.taps, c.u->zero_page, c.s,
                        c.u->splitk_ws, c.u->splitk_floats, part, &fused, x.p2, x.p2 ? x.Ca : 0, apply_table, res_up, in_up);
    if (fused) { out->gn_part = part; out->gn_chunks = fused; out->Ca = w.cout; }
    if (prof) PD_TRY(prof_end(c));
    return rc;
}

// GroupNorm statistics of x: from the fused conv-epilogue partials when available, else a pass over the tensor
int run_gn_stats(Ctx& c, const Act& x) {
    if (x.gn_part != nullptr && ((x.C / 32) % 8) == 0)
        return gn_finalize_oct(x.gn_part, x.Ca, x.gn_chunks, x.gn_partB, x.C - x.Ca, x.gn_chunksB, c.N, x.H * x.W, 1e-5f, c.u->stats, c.s);
    PD_REQUIRE(x.p2 == nullptr, "unet: a virtual concat needs fused GroupNorm partials of both tensors");
    return gn_stats(x.p, c.N, x.H * x.W, x.C, 1e-5f, c.u->stats, c.u->gn_ws, c.u->gn_ws_floats, c.s);
}

int run_gn(Ctx& c, const Act& x, const NormW& n, const float* film, long long film_stride, int silu, int resample, Act* out,
           half_t* raw_pool = nullptr) {
    *out = Act{nullptr, x.C, 0, 0};
    out->H = resample == 1 ? x.H / 2 : (resample == 2 ? x.H * 2 : x.H);
    out->W = resample == 1 ? x.W / 2 : (resample == 2 ? x.W * 2 : x.W);
    out->p = arena_take(c.u, (size_t)c.N * out->H * out->W * x.C);
    if (c.dry) return PDHIP_OK;
    PD_REQUIRE(n.have_g && n.have_b, "unet: norm weights not loaded");
#ifdef PD_LAB_SKIP_GN                                       // (lab, timing only -- WRONG results: what the small-level GroupNorm-apply launches cost)
    if (resample == 0 && x.p2 == nullptr && x.W <= 128 && silu) { out->p = x.p; return PDHIP_OK; }
#endif
    // small batches: the apply kernel reduces the octet partials itself (no k_gn_finalize_oct launch); large batches: every
    // workgroup re-reducing its image's partials costs more than the 5 us launch it removes
    if (x.gn_part != nullptr && ((x.C / 32) % 8) == 0 && (x.C >> 3) <= 256 && g_fold_finalize > 0 &&
        (c.N <= g_fold_finalize || (x.gn_chunks <= g_fold_finalize_chunks && x.gn_chunksB <= g_fold_finalize_chunks))) {
        const GnPartsArg pa{x.gn_part, x.Ca, x.gn_chunks, x.gn_partB, x.C - x.Ca, x.gn_chunksB, 1e-5f};
        return gn_apply(x.p, nullptr, n.g, n.b, film, film_stride, c.N, x.H, x.W, x.C, silu, resample, out->p, 0, c.s, x.p2,
                        x.p2 ? x.Ca : 0, raw_pool, &pa);
    }
    PD_TRY(run_gn_stats(c, x));
    return gn_apply(x.p, c.u->stats, n.g, n.b, film, film_stride, c.N, x.H, x.W, x.C, silu, resample, out->p, 0, c.s, x.p2, x.p2 ? x.Ca : 0,
                    raw_pool);
}

// GroupNorm (+ FiLM) + SiLU of x folded into the conv that consumes it (the halo-resident kernel transforms its input tile in LDS,
// nn_conv_halo.hip APPLY): statistics -> (A, B) table -> conv on the RAW tensor.  The stand-alone apply pass and its output
// tensor disappear.  Taken for the 3x3 convs the halo kernel serves, single-source input, no resampling in between.
bool can_fuse_gn(const Ctx& c, const Act& x, const ConvW& w) {
    if (c.dry) return false;                     // the sizing pass takes the two-pass form: it needs the larger arena
    if (g_fuse_gn == 0 || x.p2 != nullptr || w.taps != 9 || (x.W == 256 && x.H % 4 != 0)) return false;
    return conv_uses_halo(c.N, x.H, x.W, w.cin, w.cout, w.cout_pad, w.taps, c.u->splitk_floats);
}
int run_gn_conv(Ctx& c, const Act& x, const NormW& n, const float* film, long long film_stride, const ConvW& w,
                const half_t* residual, Act* out, int res_up = 0) {
    if (!c.dry) {
        PD_REQUIRE(n.have_g && n.have_b, "unet: norm weights not loaded");
        PD_TRY(run_gn_stats(c, x));
        PD_TRY(gn_table(c.u->stats, n.g, n.b, film, film_stride, c.N, x.C, c.u->ap_table, c.s));
    }
    return run_conv(c, x, w, residual, out, true, c.dry ? nullptr : c.u->ap_table, res_up);
}

// GroupNorm (+ FiLM) + SiLU of x applied INSIDE the row-resident conv that consumes it (nn_conv_rr.hip; bit-identical to run_gn + run_conv on the same
// kernel): taken where it measured a gain -- see g_rr_gn.  Returns false (nothing launched) when the layer is not one of those.
bool rr_gn_ok(const Ctx& c, const Act& x, const ConvW& w, int H, int W) {
    if (c.dry || g_rr_gn <= 0 || W > g_rr_gn || w.wf == nullptr || w.taps != 9 || x.p2 != nullptr || x.gn_part == nullptr || x.gn_partB != nullptr ||
        ((x.C / 32) % 8) != 0 || w.cin > 1024 || x.C != w.cin) return false;
    const int pps = std::max(1, 256 / (x.C >> 3));
    if ((x.gn_chunks + pps - 1) / pps > PD_RR_MAX_PART_LOADS) return false;
    return conv_rr_plan(c.N, H, W, w.cin, w.cout, 9, 0, c.u->splitk_floats, true).variant != 0;
}
int run_rr_gn_conv(Ctx& c, const Act& x, const NormW& n, const float* film, long long film_stride, const ConvW& w, const half_t* residual, int res_up,
                   Act* out) {
    *out = Act{nullptr, w.cout, x.H, x.W};
    out->p = arena_take(c.u, (size_t)c.N * x.H * x.W * w.cout);
    float* part = reinterpret_cast<float*>(arena_take(c.u, (size_t)c.N * ((x.H * x.W + 15) / 16) * (w.cout / 8) * 2 * 2));
    PD_REQUIRE(n.have_g && n.have_b && w.have_w && w.have_b, "unet: norm / conv weights not loaded");
    const RrPlan pl = conv_rr_plan(c.N, x.H, x.W, w.cin, w.cout, 9, 0, c.u->splitk_floats, true);
    const RrIn in{x.p, nullptr, x.C, x.C, 2, n.g, n.b, film, film_stride, x.gn_part, nullptr, x.gn_chunks, 0, 1e-5f};
    int chunks = 0;
    PD_TRY(conv_rr(pl, in, nullptr, 9, w.wf, w.b, residual, res_up, out->p, c.N, x.H, x.W, w.cout, c.u->splitk_ws, c.u->splitk_floats, part, &chunks, c.s));
    if (chunks > 0) { out->gn_part = part; out->gn_chunks = chunks; out->Ca = w.cout; }
    return PDHIP_OK;
}

int run_res(Ctx& c, const Act& x, ResB& rb, Act* out) {
    Act h0, h1, h2, xr = x, sk;
    const float* film = c.dry ? nullptr : c.film_base + rb.emb_off;
    if (!c.dry) PD_REQUIRE(rb.have_ew && rb.have_eb, "unet: emb_layers of %s not loaded", rb.name.c_str());
    const int Ho = rb.mode == 1 ? x.H / 2 : (rb.mode == 2 ? x.H * 2 : x.H), Wo = rb.mode == 1 ? x.W / 2 : (rb.mode == 2 ? x.W * 2 : x.W);
    // x branch of an up / down block (unet.py:190-195, 237-242: h and x go through the same Upsample / Downsample)
    const bool fold = g_fold_resample != 0 && !c.dry && x.p2 == nullptr && !rb.has_skip;
    const bool fold_down = fold && rb.mode == 1;                     // AvgPool2d(2)(x): second output of the GroupNorm-apply pass
    const bool fold_up = fold && rb.mode == 2 &&
                         ((Wo <= 256 && (Wo != 256 || Ho % 4 == 0) &&      // nearest x2: index arithmetic in conv2's
                           conv_uses_halo(c.N, Ho, Wo, rb.c2.cin, rb.c2.cout, rb.c2.cout_pad, 9, c.u->splitk_floats) &&   // residual read
                           conv3x3_halo_splits(c.N, Ho, Wo, rb.c2.cin, rb.c2.cout, rb.c2.cout_pad, c.u->splitk_floats) == 1) ||
                          (g_fuse_gn == 0 && conv_routes_small(c.N, Ho, Wo, rb.c2.cin, rb.c2.cout, rb.c2.cout_pad, 9, c.u->splitk_floats)) ||   // (round 4 / 6: k_conv_sk's and k_conv_ht's epilogues too)
                          (g_fuse_gn == 0 && rb.c2.wf != nullptr && conv_rr_plan(c.N, Ho, Wo, rb.c2.cin, rb.c2.cout, 9, 0, c.u->splitk_floats).variant != 0));   // (round 6: k_conv_rr's)
    if (rb.mode != 0) {
        xr.H = Ho; xr.W = Wo;
        xr.p = arena_take(c.u, (size_t)c.N * Ho * Wo * x.C);     // (the sizing pass always reserves it: routing may differ later)
    }
    // nearest x2 of the normalised input: index arithmetic in conv1's halo staging, the GroupNorm pass stays at half resolution
    const bool fold_in_up = fold && rb.mode == 2 && Wo <= 256 && (Wo != 256 || Ho % 4 == 0) &&
                            conv_uses_halo(c.N, Ho, Wo, rb.c1.cin, rb.c1.cout, rb.c1.cout_pad, 9, c.u->splitk_floats);
    // channel-changing block: in_layers' GroupNorm-apply and the skip 1x1 read the same (virtual concat) input -- one pass (k_gn_skip)
    const bool fuse_skip = rb.has_skip && rb.mode == 0 && g_fuse_skip != 0 && rb.skip.taps == 1 && x.C == rb.skip.cin &&
                           gn_skip_eligible(c.N, x.H * x.W, x.p2 ? x.Ca : x.C, x.C, rb.skip.cout, rb.skip.cout_pad) &&
                           (g_fuse_skip == 2 || (long long)c.N * x.H * x.W / 128 >= g_fuse_skip_min_tiles) &&
                           (c.dry || (x.gn_part != nullptr && ((x.C / 32) % 8) == 0));
    if (fuse_skip) {
        h0 = Act{nullptr, x.C, x.H, x.W};
        h0.p = arena_take(c.u, (size_t)c.N * x.H * x.W * x.C);
        sk = Act{nullptr, rb.skip.cout, x.H, x.W};
        sk.p = arena_take(c.u, (size_t)c.N * x.H * x.W * rb.skip.cout);
        if (!c.dry) {
            PD_REQUIRE(rb.n1.have_g && rb.n1.have_b && rb.skip.have_w && rb.skip.have_b, "unet: norm / skip weights of %s not loaded", rb.name.c_str());
            PD_TRY(run_gn_stats(c, x));
            PD_TRY(gn_skip(x.p, x.p2, x.p2 ? x.Ca : x.C, x.C, c.u->stats, rb.n1.g, rb.n1.b, rb.skip.w, rb.skip.b, h0.p, sk.p, c.N, x.H * x.W, c.s));
        }
        PD_TRY(run_conv(c, h0, rb.c1, nullptr, &h1, true));
    } else if (rb.mode == 0 && can_fuse_gn(c, x, rb.c1)) {
        PD_TRY(run_gn_conv(c, x, rb.n1, nullptr, 0, rb.c1, nullptr, &h1));
    } else if (rb.mode == 0 && rr_gn_ok(c, x, rb.c1, x.H, x.W)) {
        PD_TRY(run_rr_gn_conv(c, x, rb.n1, nullptr, 0, rb.c1, nullptr, 0, &h1));
    } else if (fold_in_up) {
        PD_TRY(run_gn(c, x, rb.n1, nullptr, 0, 1, 0, &h0));
        PD_TRY(run_conv(c, h0, rb.c1, nullptr, &h1, true, nullptr, 0, 1));
    } else {
        PD_TRY(run_gn(c, x, rb.n1, nullptr, 0, 1, rb.mode, &h0, fold_down ? xr.p : nullptr));
        PD_TRY(run_conv(c, h0, rb.c1, nullptr, &h1, true));
    }
    if (rb.mode != 0 && !c.dry && !fold_down && !fold_up) PD_TRY(resample2x(x.p, c.N, x.H, x.W, x.C, rb.mode, xr.p, c.s));
    if (!fuse_skip) sk = fold_up ? x : xr;
    // small-M layers: the skip 1x1 rides in conv2's K loop (k_conv_sk<10, ..>): no launch of its own, no skip tensor, no residual read
    if (rb.has_skip && !fuse_skip && !c.dry && g_fold_skip != 0 && rb.mode == 0 && rb.c2s_w != nullptr && rb.skip.taps == 1 &&
        (x.p2 == nullptr || (x.Ca % 64 == 0 && (x.C - x.Ca) % 64 == 0)) &&
        !conv_uses_halo(c.N, h1.H, h1.W, rb.c2.cin, rb.c2.cout, rb.c2.cout_pad, 9, c.u->splitk_floats) && !can_fuse_gn(c, h1, rb.c2)) {
        const RrPlan rp = rb.c2s_wf != nullptr && (x.p2 == nullptr || (x.Ca % 128 == 0 && (x.C - x.Ca) % 128 == 0))
                              ? conv_rr_plan(c.N, h1.H, h1.W, rb.c2.cin, rb.cout, 9, rb.skip.cin, c.u->splitk_floats) : RrPlan{0, 0, 0, 0, 0, 0, 0};
        if (rp.variant != 0) {                           // round 6: conv2 + skip 1x1 as slabs of one row-resident launch
            PD_REQUIRE(rb.c2.have_w && rb.c2.have_b && rb.skip.have_w && rb.skip.have_b && rb.c2s_ready, "unet: conv2 / skip weights of %s not loaded", rb.name.c_str());
            PD_TRY(run_gn(c, h1, rb.n2, film, c.film_stride, 1, 0, &h2));
            *out = Act{nullptr, rb.cout, h1.H, h1.W};
            out->p = arena_take(c.u, (size_t)c.N * h1.H * h1.W * rb.cout);
            float* part = reinterpret_cast<float*>(arena_take(c.u, (size_t)c.N * ((h1.H * h1.W + 15) / 16) * (rb.cout / 8) * 2 * 2));
            const RrIn in{h2.p, nullptr, rb.c2.cin, rb.c2.cin, 0, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0, 0, 1e-5f};
            const RrIn sk_in{x.p, x.p2, x.C, x.p2 ? x.Ca : x.C, 0, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0, 0, 1e-5f};
            int chunks = 0;
            PD_TRY(conv_rr(rp, in, &sk_in, 9, rb.c2s_wf, rb.c2s_b, nullptr, 0, out->p, c.N, h1.H, h1.W, rb.cout, c.u->splitk_ws, c.u->splitk_floats, part, &chunks, c.s));
            if (chunks > 0) { out->gn_part = part; out->gn_chunks = chunks; out->Ca = rb.cout; }
            return PDHIP_OK;
        }
        const SkPlan pl = conv_sk_plan(c.N, h1.H, h1.W, rb.c2.cin, rb.c2.cout, rb.c2.cout_pad, 9, false, c.u->splitk_floats, rb.skip.cin);
        if (pl.bm > 0) {
            PD_REQUIRE(rb.c2.have_w && rb.c2.have_b && rb.skip.have_w && rb.skip.have_b, "unet: conv2 / skip weights of %s not loaded", rb.name.c_str());
            PD_REQUIRE(rb.c2s_ready, "unet: fused conv2 + skip weights of %s were not built at load time", rb.name.c_str());
            PD_TRY(run_gn(c, h1, rb.n2, film, c.film_stride, 1, 0, &h2));
            *out = Act{nullptr, rb.cout, h1.H, h1.W};
            out->p = arena_take(c.u, (size_t)c.N * h1.H * h1.W * rb.cout);
            float* part = reinterpret_cast<float*>(arena_take(c.u, (size_t)c.N * ((h1.H * h1.W + 15) / 16) * (rb.cout / 8) * 2 * 2));
            const long long hw = (long long)h1.H * h1.W;
            float* gnp = (hw % pl.bm == 0 || pl.bm == 2 * hw) ? part : nullptr;
            int fused = 0;
            PD_TRY(conv_sk_skip(pl, h2.p, rb.c2s_w, rb.c2s_b, out->p, c.N, h1.H, h1.W, rb.c2.cin, rb.cout, rb.c2.cout_pad, x.p, x.p2,
                                x.p2 ? x.Ca : x.C, x.C, c.u->zero_page, c.s, c.u->splitk_ws, c.u->splitk_floats, gnp, &fused));
            if (fused) { out->gn_part = part; out->gn_chunks = fused; out->Ca = rb.cout; }
            return PDHIP_OK;
        }
    }
    if (rb.has_skip && !fuse_skip) PD_TRY(run_conv(c, xr, rb.skip, nullptr, &sk));
    if (can_fuse_gn(c, h1, rb.c2)) return run_gn_conv(c, h1, rb.n2, film, c.film_stride, rb.c2, sk.p, out, fold_up ? 1 : 0);
    if (rr_gn_ok(c, h1, rb.c2, h1.H, h1.W)) return run_rr_gn_conv(c, h1, rb.n2, film, c.film_stride, rb.c2, sk.p, fold_up ? 1 : 0, out);
    PD_TRY(run_gn(c, h1, rb.n2, film, c.film_stride, 1, 0, &h2));
    return run_conv(c, h2, rb.c2, sk.p, out, true, nullptr, fold_up ? 1 : 0);
}

int run_att(Ctx& c, const Act& x, AttB& ab, Act* out) {
    Act xn, qkv, a;
    PD_TRY(run_gn(c, x, ab.n, nullptr, 0, 0, 0, &xn));
    PD_TRY(run_conv(c, xn, ab.qkv, nullptr, &qkv));
    a = x;
    a.p = arena_take(c.u, (size_t)c.N * x.H * x.W * x.C);
    half_t* vt = arena_take(c.u, (size_t)c.N * x.H * x.W * x.C);      // transposed V of the T >= 128 attention kernel
    if (!c.dry) {
        // profile hook (bench.py roofline.attention): QK^T + PV = 4 T^2 C flop per image (12.18 GFLOP per forward)
        const double T = (double)x.H * x.W;
        PD_TRY(prof_begin(c, 4.0 * c.N * T * T * x.C, 1));
        PD_TRY(attention(qkv.p, a.p, c.N, x.H * x.W, x.C, c.u->head, c.s, vt));
        PD_TRY(prof_end(c));
    }
    return run_conv(c, a, ab.proj, x.p, out, true);
}

int run_blocks(Ctx& c, std::vector<Block>& blocks, Act* h, const float* x_nchw) {
    for (Block& b : blocks) {
        Act o;
        if (b.kind == 0) {
            o.C = c.u->mult[0] * c.u->mc; o.H = o.W = c.u->image_size;
            o.p = arena_take(c.u, (size_t)c.N * o.H * o.W * o.C);
            half_t* col = arena_take(c.u, (size_t)c.N * o.H * o.W * 32);
            float* part = reinterpret_cast<float*>(arena_take(c.u, (size_t)c.N * ((o.H * o.W + 127) / 128) * (o.C / 8) * 2 * 2));
            if (!c.dry) {
                PD_REQUIRE(c.u->have_win && c.u->have_bin, "unet: input conv not loaded");
                int fused = 0;
                PD_TRY(conv_in_3x3(x_nchw, c.u->w_in, c.u->b_in, o.p, c.N, o.H, o.W, o.C, ((o.C + 127) / 128) * 128, col,
                                   c.u->zero_page, c.s, part, &fused));
                if (fused) { o.gn_part = part; o.gn_chunks = fused; o.Ca = o.C; }
            }
        } else if (b.kind == 1) {
            PD_TRY(run_res(c, *h, c.u->res[b.idx], &o));
        } else {
            PD_TRY(run_att(c, *h, c.u->att[b.idx], &o));
        }
        *h = o;
    }
    return PDHIP_OK;
}

// the whole forward; dry = sizing pass (no launches)
// shared_emb: one precomputed row of ResBlock embeddings [emb_rows] used by every image (t is then ignored), or nullptr
int forward_impl(pdhip_unet* u, const float* x, const float* t, int N, float* out, hipStream_t s, bool dry,
                 const float* shared_emb = nullptr, int head_ch = 0) {   // head_ch 3: only eps (out [N,3,H,W]); 0 = all out_ch
    Ctx c{u, N, s, dry, shared_emb ? shared_emb : u->emb_all, shared_emb ? 0 : u->emb_rows};
    u->arena_off = 0;
    u->arena_overflow = false;
    if (!dry && u->prof.on) u->prof.armed = (u->prof.forwards++ % u->prof.period) == 0;      // events around every period-th forward's launches
    if (!dry && shared_emb == nullptr) {
        PD_REQUIRE(u->have_te[0] && u->have_te[1] && u->have_te[2] && u->have_te[3], "unet: time_embed not loaded");
        PD_TRY(timestep_mlp(t, N, u->mc, u->te_w0, u->te_b0, u->te_w2, u->te_b2, u->emb_silu, u->emb_tmp, s));
        PD_TRY(gemv_rows(u->emb_w, u->emb_b, u->emb_silu, u->emb_all, (int)u->emb_rows, u->ted, N, s));
    }
    std::vector<Act> hs;
    Act h{nullptr, 0, 0, 0};
    for (auto& blk : u->input) {
        PD_TRY(run_blocks(c, blk, &h, x));
        hs.push_back(h);
    }
    PD_TRY(run_blocks(c, u->middle, &h, x));
    for (auto& blk : u->output) {
        Act sk = hs.back(); hs.pop_back();
        // torch.cat([h, skip], dim=1): when both tensors carry fused GroupNorm partials (and 64-channel granularity for the
        // 1x1 skip conv's K loop) the concat stays virtual -- its two consumers (GroupNorm apply, skip conv) read both tensors
        Act cat{nullptr, h.C + sk.C, h.H, h.W};
        const bool part_ok = h.gn_part && sk.gn_part && !h.gn_partB && !sk.gn_partB && !h.p2 && !sk.p2;
        if (part_ok) {
            cat.gn_part = h.gn_part; cat.gn_partB = sk.gn_part; cat.Ca = h.C; cat.gn_chunks = h.gn_chunks; cat.gn_chunksB = sk.gn_chunks;
        }
        const Block& first = blk.front();
        const bool virt = part_ok && h.C % 64 == 0 && sk.C % 64 == 0 && ((cat.C / 32) % 8) == 0 && first.kind == 1 &&
                          u->res[first.idx].mode == 0 && u->res[first.idx].has_skip;
        if (virt) {
            cat.p = h.p; cat.p2 = sk.p;
        } else {
            cat.p = arena_take(u, (size_t)N * h.H * h.W * (h.C + sk.C));
            if (!dry) PD_TRY(concat_channels(h.p, h.C, sk.p, sk.C, (long long)N * h.H * h.W, cat.p, s));
        }
        h = cat;
        PD_TRY(run_blocks(c, blk, &h, x));
    }
    if (!dry) {
        PD_REQUIRE(!u->arena_overflow, "unet: activation arena too small for this routing (sized at create())");
        PD_REQUIRE(u->out_norm.have_g && u->out_norm.have_b && u->have_ow && u->have_ob, "unet: output head not loaded");
        PD_TRY(run_gn_stats(c, h));
        const bool eps_only = head_ch == 3 && u->out_ch == 6;
        PD_TRY(head_gn_silu_conv3x3(h.p, u->stats, u->out_norm.g, u->out_norm.b, eps_only ? u->head_wz3 : u->head_wz, u->out_b, out, N,
                                    h.H, h.W, h.C, eps_only ? 3 : u->out_ch, s));
    }
    return PDHIP_OK;
}

}  // namespace

// =================================================================================================
extern "C" int pdhip_unet_create(int image_size, int model_channels, int num_res_blocks, const int* channel_mult, int n_mult,
                                 const int* attention_ds, int n_att, int num_head_channels, int out_channels, int max_batch,
                                 pdhip_unet** out) {
    PD_REQUIRE(out && channel_mult && n_mult > 0 && (n_att == 0 || attention_ds), "pdhip_unet_create: null argument");
    PD_REQUIRE(model_channels % 32 == 0 && image_size % (1 << (n_mult - 1)) == 0 && max_batch >= 1 && max_batch <= 64,
               "pdhip_unet_create: model_channels %% 32, image_size %% 2^(levels-1), 1 <= max_batch <= 64 required");
    PD_REQUIRE(num_head_channels == 32 || num_head_channels == 64, "pdhip_unet_create: num_head_channels must be 32 or 64");
    PD_REQUIRE(out_channels == 3 || out_channels == 6, "pdhip_unet_create: out_channels must be 3 or 6");
    pdhip_unet* u = new pdhip_unet();
    *out = nullptr;
    u->image_size = image_size; u->mc = model_channels; u->nres = num_res_blocks; u->head = num_head_channels;
    u->out_ch = out_channels; u->max_batch = max_batch; u->ted = model_channels * 4;
    u->mult.assign(channel_mult, channel_mult + n_mult);
    u->att_ds.assign(attention_ds, attention_ds + n_att);
    auto is_att = [&](int ds) { for (int a : u->att_ds) if (a == ds) return true; return false; };
    auto fail = [&](int rc) { for (void* p : u->owned) (void)hipFree(p); delete u; return rc; };
    long long emb_rows = 0;
    auto add_res = [&](const std::string& name, int cin, int cout, int mode, int hw) -> int {
        ResB rb; rb.name = name; rb.cin = cin; rb.cout = cout; rb.mode = mode; rb.has_skip = cin != cout; rb.hw = hw;
        rb.emb_off = emb_rows; emb_rows += 2 * cout;
        u->res.push_back(rb);
        return (int)u->res.size() - 1;
    };
    auto add_att = [&](const std::string& name, int c) -> int {
        AttB ab; ab.name = name; ab.c = c;
        u->att.push_back(ab);
        return (int)u->att.size() - 1;
    };
    // ---- graph (unet.py:481-617)
    int ch = u->mult[0] * u->mc;
    std::vector<int> chans{ch};
    u->input.push_back({Block{0, 0}});
    int ds = 1, n = 1;
    for (int level = 0; level < n_mult; ++level) {
        for (int r = 0; r < num_res_blocks; ++r) {
            std::vector<Block> layers;
            const int o = u->mult[level] * u->mc;
            layers.push_back(Block{1, add_res("input_blocks." + std::to_string(n) + ".0", ch, o, 0, image_size / ds)});
            ch = o;
            if (is_att(ds)) layers.push_back(Block{2, add_att("input_blocks." + std::to_string(n) + ".1", ch)});
            u->input.push_back(layers);
            chans.push_back(ch);
            ++n;
        }
        if (level != n_mult - 1) {
            u->input.push_back({Block{1, add_res("input_blocks." + std::to_string(n) + ".0", ch, ch, 1, image_size / (ds * 2))}});
            chans.push_back(ch);
            ds *= 2;
            ++n;
        }
    }
    u->middle.push_back(Block{1, add_res("middle_block.0", ch, ch, 0, image_size / ds)});
    u->middle.push_back(Block{2, add_att("middle_block.1", ch)});
    u->middle.push_back(Block{1, add_res("middle_block.2", ch, ch, 0, image_size / ds)});
    n = 0;
    for (int level = n_mult - 1; level >= 0; --level) {
        for (int i = 0; i < num_res_blocks + 1; ++i) {
            const int ich = chans.back(); chans.pop_back();
            const int o = u->mc * u->mult[level];
            std::vector<Block> layers;
            const std::string base = "output_blocks." + std::to_string(n) + ".";
            layers.push_back(Block{1, add_res(base + "0", ch + ich, o, 0, image_size / ds)});
            ch = o;
            int k = 1;
            if (is_att(ds)) { layers.push_back(Block{2, add_att(base + std::to_string(k), ch)}); ++k; }
            if (level && i == num_res_blocks) {
                layers.push_back(Block{1, add_res(base + std::to_string(k), ch, ch, 2, image_size / (ds / 2))});
                ds /= 2;
            }
            u->output.push_back(layers);
            ++n;
        }
    }
    u->final_ch = ch;
    u->emb_rows = emb_rows;
    // ---- parameters
    int rc = PDHIP_OK;
    auto chk = [&](int r) { if (rc == PDHIP_OK) rc = r; };
    for (ResB& rb : u->res) {
        chk(alloc_norm(u, rb.n1, rb.cin)); chk(alloc_norm(u, rb.n2, rb.cout));
        chk(alloc_conv(u, rb.c1, rb.cin, rb.cout, 9)); chk(alloc_conv(u, rb.c2, rb.cout, rb.cout, 9));
        if (rb.has_skip) chk(alloc_conv(u, rb.skip, rb.cin, rb.cout, 1));
        if (rb.has_skip && rb.mode == 0 && rb.cin % 64 == 0 && rb.cout % 64 == 0) {       // fused [conv2 | skip] weights (built at the first forward that routes there)
            chk(dalloc(u, &rb.c2s_w, (size_t)rb.c2.cout_pad * (9 * (size_t)rb.cout + rb.cin)));
            chk(dalloc(u, &rb.c2s_b, (size_t)rb.cout));
        }
        // fragment-major copies for the row-resident kernel (small-batch route of the <= 64^2 levels; +~1 GB at the 256^2 config: 288 GB HBM)
        if (rb.hw <= 64 && rb.cin % 128 == 0 && rb.cout % 128 == 0) {
            chk(dalloc(u, &rb.c1.wf, conv_rr_weight_halfs(rb.cin, 9, 0, rb.cout)));
            chk(dalloc(u, &rb.c2.wf, conv_rr_weight_halfs(rb.cout, 9, 0, rb.cout)));
            if (rb.c2s_w != nullptr) chk(dalloc(u, &rb.c2s_wf, conv_rr_weight_halfs(rb.cout, 9, rb.cin, rb.cout)));
        }
        if (rc) return fail(rc);
    }
    for (AttB& ab : u->att) {
        chk(alloc_norm(u, ab.n, ab.c)); chk(alloc_conv(u, ab.qkv, ab.c, 3 * ab.c, 1)); chk(alloc_conv(u, ab.proj, ab.c, ab.c, 1));
        if (rc) return fail(rc);
    }
    const int c0 = u->mult[0] * u->mc;
    chk(dalloc(u, &u->w_in, (size_t)(((c0 + 127) / 128) * 128) * 32)); chk(dalloc(u, &u->b_in, (size_t)c0));
    if (rc == PDHIP_OK && hipMemset(u->w_in, 0, (size_t)(((c0 + 127) / 128) * 128) * 32 * sizeof(half_t)) != hipSuccess) rc = PDHIP_E_HIP;
    chk(dalloc(u, &u->te_w0, (size_t)u->ted * u->mc)); chk(dalloc(u, &u->te_b0, (size_t)u->ted));
    chk(dalloc(u, &u->te_w2, (size_t)u->ted * u->ted)); chk(dalloc(u, &u->te_b2, (size_t)u->ted));
    chk(dalloc(u, &u->emb_w, (size_t)emb_rows * u->ted)); chk(dalloc(u, &u->emb_b, (size_t)emb_rows));
    chk(alloc_norm(u, u->out_norm, u->final_ch));
    chk(dalloc(u, &u->out_w, (size_t)out_channels * 9 * u->final_ch)); chk(dalloc(u, &u->out_b, (size_t)out_channels));
    chk(dalloc(u, &u->zero_page, (size_t)128));
    if (rc) return fail(rc);
    if (hipMemset(u->zero_page, 0, 256) != hipSuccess) { set_error("hipMemset failed"); return fail(PDHIP_E_HIP); }
    // ---- workspace (sized by a dry run of the forward at max_batch)
    const size_t S2 = (size_t)image_size * image_size;
    chk(forward_impl(u, nullptr, nullptr, max_batch, nullptr, nullptr, true));
    u->arena_bytes = u->arena_off + 4096;
    {
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, u->arena_bytes);
        if (e != hipSuccess) { set_error("pdhip_unet_create: activation arena of %zu bytes: %s", u->arena_bytes, hipGetErrorString(e)); return fail(PDHIP_E_NOMEM); }
        u->owned.push_back(q);
        u->arena = reinterpret_cast<char*>(q);
    }
    u->gn_ws_floats = (size_t)max_batch * ((S2 + 255) / 256) * 64;
    chk(dalloc(u, &u->stats, (size_t)max_batch * 64)); chk(dalloc(u, &u->gn_ws, u->gn_ws_floats));
    { int cmax = 0; for (const ResB& rb : u->res) cmax = std::max(cmax, std::max(rb.cin, rb.cout)); chk(dalloc(u, &u->ap_table, (size_t)max_batch * cmax * 2)); }
    chk(dalloc(u, &u->emb_silu, (size_t)max_batch * u->ted)); chk(dalloc(u, &u->emb_tmp, (size_t)max_batch * (u->mc + u->ted)));
    chk(dalloc(u, &u->emb_all, (size_t)max_batch * emb_rows));
    chk(dalloc(u, &u->steps_t, 128)); chk(dalloc(u, &u->steps_silu, (size_t)100 * u->ted));
    chk(dalloc(u, &u->steps_tmp, (size_t)100 * (u->mc + u->ted))); chk(dalloc(u, &u->steps_emb, (size_t)100 * emb_rows));
    { float* wz = nullptr; chk(dalloc(u, &wz, (size_t)64 * u->final_ch)); u->head_wz = reinterpret_cast<half_t*>(wz); }
    { float* wz = nullptr; chk(dalloc(u, &wz, (size_t)64 * u->final_ch)); u->head_wz3 = reinterpret_cast<half_t*>(wz); }
    chk(dalloc(u, &u->t_dev, (size_t)max_batch));
    u->splitk_floats = (size_t)16 * 384 * 128 * 128;               // 16 splits x (< 384 tiles of 128x128) f32
    chk(dalloc(u, &u->splitk_ws, u->splitk_floats));
    if (rc == PDHIP_OK && hipMemset(u->splitk_ws, 0, PD_SK_TICKET_FLOATS * sizeof(float)) != hipSuccess) rc = PDHIP_E_HIP;   // k_conv_sk's tickets
    chk(dalloc(u, &u->sx, (size_t)max_batch * 3 * S2)); chk(dalloc(u, &u->sy, (size_t)max_batch * 3 * S2));
    chk(dalloc(u, &u->set_, (size_t)max_batch * out_channels * S2)); chk(dalloc(u, &u->smask, (size_t)max_batch * S2));
    if (rc) return fail(rc);
    *out = u;
    return PDHIP_OK;
}

extern "C" void pdhip_unet_destroy(pdhip_unet* u) {
    if (!u) return;
    for (hipEvent_t e : u->prof.ev) (void)hipEventDestroy(e);
    for (void* p : u->owned) (void)hipFree(p);
    delete u;
}

extern "C" long long pdhip_unet_arena_bytes(const pdhip_unet* u) { return u ? (long long)u->arena_bytes : -1; }

// names still missing after loading (count); if buf != NULL the first names are written, '\n'-separated
extern "C" int pdhip_unet_missing_tensors(const pdhip_unet* u, char* buf, int buf_len) {
    if (!u) return -1;
    std::vector<std::string> miss;
    auto conv = [&](const std::string& n, const ConvW& c) { if (!c.have_w) miss.push_back(n + ".weight"); if (!c.have_b) miss.push_back(n + ".bias"); };
    auto norm = [&](const std::string& n, const NormW& c) { if (!c.have_g) miss.push_back(n + ".weight"); if (!c.have_b) miss.push_back(n + ".bias"); };
    if (!u->have_win) miss.push_back("input_blocks.0.0.weight");
    if (!u->have_bin) miss.push_back("input_blocks.0.0.bias");
    const char* te[4] = {"time_embed.0.weight", "time_embed.0.bias", "time_embed.2.weight", "time_embed.2.bias"};
    for (int i = 0; i < 4; ++i) if (!u->have_te[i]) miss.push_back(te[i]);
    for (const ResB& rb : u->res) {
        norm(rb.name + ".in_layers.0", rb.n1); conv(rb.name + ".in_layers.2", rb.c1);
        if (!rb.have_ew) miss.push_back(rb.name + ".emb_layers.1.weight");
        if (!rb.have_eb) miss.push_back(rb.name + ".emb_layers.1.bias");
        norm(rb.name + ".out_layers.0", rb.n2); conv(rb.name + ".out_layers.3", rb.c2);
        if (rb.has_skip) conv(rb.name + ".skip_connection", rb.skip);
    }
    for (const AttB& ab : u->att) { norm(ab.name + ".norm", ab.n); conv(ab.name + ".qkv", ab.qkv); conv(ab.name + ".proj_out", ab.proj); }
    norm("out.0", u->out_norm);
    if (!u->have_ow) miss.push_back("out.2.weight");
    if (!u->have_ob) miss.push_back("out.2.bias");
    if (buf && buf_len > 0) {
        std::string s;
        for (const std::string& m : miss) { if ((int)(s.size() + m.size() + 2) >= buf_len) break; s += m; s += '\n'; }
        strncpy(buf, s.c_str(), buf_len - 1); buf[buf_len - 1] = 0;
    }
    return (int)miss.size();
}

extern "C" int pdhip_unet_num_tensors(const pdhip_unet* u) {
    if (!u) return -1;
    int n = 2 + 4 + 4;                                   // conv_in, time_embed, out.0/out.2
    for (const ResB& rb : u->res) n += 10 + (rb.has_skip ? 2 : 0);
    n += 6 * (int)u->att.size();
    return n;
}

extern "C" int pdhip_unet_load_tensor(pdhip_unet* u, const char* name_c, const void* data, int is_f16, const int64_t* shape,
                                      int ndim, void* stream) {
    PD_REQUIRE(u && name_c && data && shape && ndim >= 1 && ndim <= 4, "pdhip_unet_load_tensor: bad arguments");
    u->steps_cached = 0;                         // any weight change invalidates the sampler's per-step embedding table
    hipStream_t s = as_stream(stream);
    const std::string name(name_c);
    long long numel = 1;
    for (int i = 0; i < ndim; ++i) numel *= shape[i];
    auto want = [&](long long n, const char* what) -> int {
        PD_REQUIRE(numel == n, "pdhip_unet_load_tensor: %s: %s expects %lld elements, got %lld", name_c, what, n, numel);
        return PDHIP_OK;
    };
    auto load_vec = [&](float* dst, long long n, bool* flag, int round16) -> int {
        PD_TRY(want(n, "vector"));
        k_copy_f32<<<grid_for(n), 256, 0, s>>>(data, is_f16, n, dst, round16);
        PD_LAUNCH_CHECK();
        *flag = true;
        return PDHIP_OK;
    };
    auto load_conv_w = [&](ConvW& c) -> int {
        PD_TRY(want((long long)c.cout * c.cin * c.taps, "conv weight"));
        PD_REQUIRE(shape[0] == c.cout && shape[1] == c.cin, "pdhip_unet_load_tensor: %s: shape mismatch", name_c);
        k_pack_conv<<<grid_for(numel), 256, 0, s>>>(data, is_f16, c.cout, c.cin, c.taps, c.w);
        PD_LAUNCH_CHECK();
        if (c.wf != nullptr) PD_TRY(conv_rr_pack(c.w, c.cin, c.taps, 0, c.cout, c.wf, s));      // (same stream: ordered behind the repack above)
        c.have_w = true;
        return PDHIP_OK;
    };
    auto ends = [&](const std::string& suf) { return name.size() >= suf.size() && name.compare(name.size() - suf.size(), suf.size(), suf) == 0; };
    auto starts = [&](const std::string& pre) { return name.compare(0, pre.size(), pre) == 0; };
    if (name == "input_blocks.0.0.weight") {
        const int c0 = u->mult[0] * u->mc;
        PD_TRY(want((long long)c0 * 27, "input conv weight"));
        k_pack_conv_in<<<grid_for(numel), 256, 0, s>>>(data, is_f16, c0, u->w_in);
        PD_LAUNCH_CHECK();
        u->have_win = true;
        return PDHIP_OK;
    }
    if (name == "input_blocks.0.0.bias") return load_vec(u->b_in, u->mult[0] * u->mc, &u->have_bin, 1);
    if (name == "time_embed.0.weight") return load_vec(u->te_w0, (long long)u->ted * u->mc, &u->have_te[0], 0);
    if (name == "time_embed.0.bias") return load_vec(u->te_b0, u->ted, &u->have_te[1], 0);
    if (name == "time_embed.2.weight") return load_vec(u->te_w2, (long long)u->ted * u->ted, &u->have_te[2], 0);
    if (name == "time_embed.2.bias") return load_vec(u->te_b2, u->ted, &u->have_te[3], 0);
    if (name == "out.0.weight") return load_vec(u->out_norm.g, u->final_ch, &u->out_norm.have_g, 0);
    if (name == "out.0.bias") return load_vec(u->out_norm.b, u->final_ch, &u->out_norm.have_b, 0);
    if (name == "out.2.bias") return load_vec(u->out_b, u->out_ch, &u->have_ob, 0);
    if (name == "out.2.weight") {
        PD_TRY(want((long long)u->out_ch * u->final_ch * 9, "out conv weight"));
        k_pack_conv_f32<<<grid_for(numel), 256, 0, s>>>(data, is_f16, u->out_ch, u->final_ch, 9, u->out_w);
        PD_LAUNCH_CHECK();
        PD_TRY(head_pack(u->out_w, u->out_ch, u->final_ch, u->head_wz, s));
        PD_TRY(head_pack(u->out_w, 3, u->final_ch, u->head_wz3, s));            // rows 0..2 of [out_ch][9 C]
        u->have_ow = true;
        return PDHIP_OK;
    }
    for (ResB& rb : u->res) {
        if (!starts(rb.name + ".")) continue;
        const std::string rest = name.substr(rb.name.size() + 1);
        if (rest == "in_layers.0.weight") return load_vec(rb.n1.g, rb.cin, &rb.n1.have_g, 0);
        if (rest == "in_layers.0.bias") return load_vec(rb.n1.b, rb.cin, &rb.n1.have_b, 0);
        if (rest == "in_layers.2.weight") return load_conv_w(rb.c1);
        if (rest == "in_layers.2.bias") return load_vec(rb.c1.b, rb.cout, &rb.c1.have_b, 1);
        if (rest == "emb_layers.1.weight") return load_vec(u->emb_w + rb.emb_off * u->ted, (long long)2 * rb.cout * u->ted, &rb.have_ew, 0);
        if (rest == "emb_layers.1.bias") return load_vec(u->emb_b + rb.emb_off, 2 * rb.cout, &rb.have_eb, 0);
        if (rest == "out_layers.0.weight") return load_vec(rb.n2.g, rb.cout, &rb.n2.have_g, 0);
        if (rest == "out_layers.0.bias") return load_vec(rb.n2.b, rb.cout, &rb.n2.have_b, 0);
        // The fused [conv2 | skip] copy of conv_sk_skip is (re)built HERE, on the loading stream, as soon as its four source tensors are
        // present (built lazily inside the first forward it could be merely RECORDED by a HIP-graph capture, and an eager
        // forward issued before the replay would have read uninitialised weights).
        auto refuse = [&](int rc) -> int {
            if (rc != PDHIP_OK || rb.c2s_w == nullptr) return rc;
            rb.c2s_ready = false;
            if (rb.c2.have_w && rb.c2.have_b && rb.skip.have_w && rb.skip.have_b && rb.skip.taps == 1) {
                PD_TRY(fuse_skip_weights(rb.c2.w, 9 * rb.c2.cin, rb.skip.w, rb.skip.cin, rb.c2.cout_pad, rb.c2.b, rb.skip.b, rb.cout, rb.c2s_w, rb.c2s_b, s));
                if (rb.c2s_wf != nullptr) PD_TRY(conv_rr_pack(rb.c2s_w, rb.c2.cin, 9, rb.skip.cin, rb.cout, rb.c2s_wf, s));
                rb.c2s_ready = true;
            }
            return PDHIP_OK;
        };
        if (rest == "out_layers.3.weight") return refuse(load_conv_w(rb.c2));
        if (rest == "out_layers.3.bias") return refuse(load_vec(rb.c2.b, rb.cout, &rb.c2.have_b, 1));
        if (rb.has_skip && rest == "skip_connection.weight") return refuse(load_conv_w(rb.skip));
        if (rb.has_skip && rest == "skip_connection.bias") return refuse(load_vec(rb.skip.b, rb.cout, &rb.skip.have_b, 1));
    }
    for (AttB& ab : u->att) {
        if (!starts(ab.name + ".")) continue;
        const std::string rest = name.substr(ab.name.size() + 1);
        if (rest == "norm.weight") return load_vec(ab.n.g, ab.c, &ab.n.have_g, 0);
        if (rest == "norm.bias") return load_vec(ab.n.b, ab.c, &ab.n.have_b, 0);
        if (rest == "qkv.weight") return load_conv_w(ab.qkv);
        if (rest == "qkv.bias") return load_vec(ab.qkv.b, 3 * ab.c, &ab.qkv.have_b, 1);
        if (rest == "proj_out.weight") return load_conv_w(ab.proj);
        if (rest == "proj_out.bias") return load_vec(ab.proj.b, ab.c, &ab.proj.have_b, 1);
    }
    (void)ends;
    set_error("pdhip_unet_load_tensor: unknown tensor name '%s'", name_c);
    return PDHIP_E_UNKNOWN_NAME;
}

extern "C" int pdhip_unet_forward(pdhip_unet* u, const float* x, const float* t, int N, float* out, void* stream) {
    PD_REQUIRE(u && x && t && out, "pdhip_unet_forward: null argument");
    PD_REQUIRE(N >= 1 && N <= u->max_batch, "pdhip_unet_forward: batch %d outside [1, %d]", N, u->max_batch);
    return forward_impl(u, x, t, N, out, as_stream(stream), false);
}

// ---- profiling of the dominant kernel (3x3 implicit-GEMM launches) with HIP events on the launch stream
extern "C" int pdhip_unet_profile(pdhip_unet* u, int enable) {
    PD_REQUIRE(u, "pdhip_unet_profile: null handle");
    u->prof.on = enable != 0;                      // enable = k > 1: only every k-th forward carries events (an event record is a
    u->prof.period = enable > 1 ? enable : 1;      // barrier packet: ~5 us of pipeline bubble each, 108 per forward)
    u->prof.forwards = 0;
    u->prof.armed = false;
    u->prof.used = 0;
    u->prof.flops[0] = u->prof.flops[1] = 0;
    return PDHIP_OK;
}
static int prof_read(pdhip_unet* u, int cls, double* total_ms, double* total_flops, long long* launches) {
    double ms = 0;
    long long n = 0;
    for (size_t i = 0; i + 1 < u->prof.used; i += 2) {
        if (u->prof.cls[i] != cls) continue;
        PD_HIP(hipEventSynchronize(u->prof.ev[i + 1]));
        float e = 0;
        PD_HIP(hipEventElapsedTime(&e, u->prof.ev[i], u->prof.ev[i + 1]));
        ms += e; ++n;
    }
    *total_ms = ms; *total_flops = u->prof.flops[cls]; *launches = n;
    return PDHIP_OK;
}
extern "C" int pdhip_unet_profile_read(pdhip_unet* u, double* total_ms, double* total_flops, long long* launches) {
    PD_REQUIRE(u && total_ms && total_flops && launches, "pdhip_unet_profile_read: null argument");
    return prof_read(u, 0, total_ms, total_flops, launches);
}
/* the same for the attention launches (k_attention_t64 + its V transpose / k_attention): 4 T^2 C flop per image */
extern "C" int pdhip_unet_profile_read_attention(pdhip_unet* u, double* total_ms, double* total_flops, long long* launches) {
    PD_REQUIRE(u && total_ms && total_flops && launches, "pdhip_unet_profile_read_attention: null argument");
    return prof_read(u, 1, total_ms, total_flops, launches);
}

// ---- D1: DDNM schedule (diffusion.py:46-113, 770-812) computed on the host exactly as the reference forms it
namespace {
struct Sched { std::vector<int> t; std::vector<DdnmCoef> co; std::vector<float> ab; };
Sched make_schedule() {
    // betas = np.linspace(1e-4, 0.02, 1000) (f64: start + i*step, last = stop) -> f32;
    // alpha_bar(t) = cumprod(1 - [0, beta])[t+1] on f32 inputs; torch's CPU cumprod accumulates in f64 and rounds
    // each prefix to f32 (the golden vectors pin exactly this; a CUDA scan differs in the last ulp).
    Sched s;
    s.ab.resize(1001);
    double acc = 1.0;
    s.ab[0] = 1.0f;                                                // index t+1 with t = -1
    const double b0 = 1e-4, b1 = 0.02, step = (b1 - b0) / 999.0;
    for (int i = 0; i < 1000; ++i) {
        const double beta = (i == 999) ? b1 : b0 + (double)i * step;
        acc = acc * (double)(1.0f - (float)beta);
        s.ab[i + 1] = (float)acc;
    }
    const double eta = 0.85;
    for (int k = 99; k >= 0; --k) {
        const int t = k * 10, tn = (k - 1) * 10 < 0 ? -1 : (k - 1) * 10;
        const float at = s.ab[t + 1], an = s.ab[tn + 1];
        DdnmCoef c;
        c.sqrt_1m_at = sqrtf(1.0f - at); c.sqrt_at = sqrtf(at); c.sqrt_at_next = sqrtf(an);
        c.sigma_t = sqrtf(1.0f - an * an);
        c.c1 = sqrtf(1.0f - an) * (float)eta;
        c.c2 = sqrtf(1.0f - an) * (float)std::sqrt(1.0 - eta * eta);
        s.t.push_back(t); s.co.push_back(c);
    }
    return s;
}
__global__ void k_fill(float* p, float v, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v; }
}  // namespace

extern "C" int pdhip_ddnm_schedule(float* at, float* at_next, int* t, int* t_next, float* coefs /*[100][6]*/) {
    Sched s = make_schedule();
    for (int k = 0; k < 100; ++k) {
        const int tt = s.t[k], tn = tt - 10 < 0 ? -1 : tt - 10;
        if (at) at[k] = s.ab[tt + 1];
        if (at_next) at_next[k] = s.ab[tn + 1];
        if (t) t[k] = tt;
        if (t_next) t_next[k] = tn;
        if (coefs) memcpy(coefs + 6 * k, &s.co[k], sizeof(DdnmCoef));
    }
    return PDHIP_OK;
}

extern "C" int pdhip_ddnm_step(float* x, const float* et, int et_channels, const float* y, const float* mask, const float* eps,
                               uint64_t seed, int step, int N, int HW, void* stream) {
    PD_REQUIRE(x && et && y && mask && step >= 0 && step < 100 && (et_channels == 3 || et_channels == 6), "pdhip_ddnm_step: bad arguments");
    static const Sched sched = make_schedule();
    return ddnm_update(x, et, et_channels, y, mask, eps, seed, (unsigned long long)step + 1, sched.co[step], N, HW, as_stream(stream));
}

extern "C" int pdhip_ddnm_prepare(const float* masked_imgs, const float* masks, float* y, int N, int HW, void* stream) {
    PD_REQUIRE(masked_imgs && masks && y, "pdhip_ddnm_prepare: null argument");
    return ddnm_prepare(masked_imgs, masks, y, N, HW, as_stream(stream));
}

// The whole of simplified_ddnm_inpainting for N images at once: no host round trip inside the loop.
// x_T / eps_tape (eps_tape[k] = noise of step k, [n_steps,N,3,S,S]) may be NULL -> Philox noise from `seed`.
// first_image_key: image n of the batch draws the Philox noise stream of key first_image_key + n (x_T: stream 0, step k: stream
// k + 1, counter = key * 3HW/4 + element / 4): the noise of a view does not depend on its position in the batch, on the batch
// it shares a sampler call with, or on how the views of a shape are sharded over ranks.
extern "C" int pdhip_ddnm_sample_keyed(pdhip_unet* u, const float* masked_imgs, const float* masks, int N, const float* x_T,
                                       const float* eps_tape, uint64_t seed, uint64_t first_image_key, int n_steps, float* out,
                                       void* stream) {
    PD_REQUIRE(u && masked_imgs && masks && out, "pdhip_ddnm_sample: null argument");
    PD_REQUIRE(N >= 1 && N <= u->max_batch && n_steps >= 1 && n_steps <= 100, "pdhip_ddnm_sample: bad N / n_steps");
    static const Sched sched = make_schedule();
    hipStream_t s = as_stream(stream);
    const int HW = u->image_size * u->image_size;
    const long long n3 = (long long)N * 3 * HW;
    PD_TRY(ddnm_prepare(masked_imgs, masks, u->sy, N, HW, s));
    const unsigned long long quad0 = (unsigned long long)first_image_key * (unsigned long long)(3LL * HW / 4);
    if (x_T) PD_HIP(hipMemcpyAsync(u->sx, x_T, n3 * sizeof(float), hipMemcpyDeviceToDevice, s));
    else PD_TRY(philox_normal(u->sx, n3, seed, 0, s, quad0));
    if (u->steps_cached != n_steps) {            // embeddings of every schedule step, one pass over the emb_layers weights
        PD_REQUIRE(u->have_te[0] && u->have_te[1] && u->have_te[2] && u->have_te[3], "unet: time_embed not loaded");
        float th[100];
        for (int k = 0; k < n_steps; ++k) th[k] = (float)sched.t[k];
        PD_HIP(hipMemcpyAsync(u->steps_t, th, n_steps * sizeof(float), hipMemcpyHostToDevice, s));
        PD_HIP(hipStreamSynchronize(s));         // `th` is a stack buffer
        PD_TRY(timestep_mlp(u->steps_t, n_steps, u->mc, u->te_w0, u->te_b0, u->te_w2, u->te_b2, u->steps_silu, u->steps_tmp, s));
        PD_TRY(gemv_rows(u->emb_w, u->emb_b, u->steps_silu, u->steps_emb, (int)u->emb_rows, u->ted, n_steps, s));
        u->steps_cached = n_steps;
    }
    for (int k = 0; k < n_steps; ++k) {
        PD_TRY(forward_impl(u, u->sx, nullptr, N, u->set_, s, false, u->steps_emb + (size_t)k * u->emb_rows, 3));
        PD_TRY(ddnm_update(u->sx, u->set_, 3, u->sy, masks, eps_tape ? eps_tape + (size_t)k * n3 : nullptr, seed,
                           (unsigned long long)k + 1, sched.co[k], N, HW, s, quad0));
    }
    return ddnm_finish(u->sx, out, n3, s);
}
extern "C" int pdhip_ddnm_sample(pdhip_unet* u, const float* masked_imgs, const float* masks, int N, const float* x_T,
                                 const float* eps_tape, uint64_t seed, int n_steps, float* out, void* stream) {
    return pdhip_ddnm_sample_keyed(u, masked_imgs, masks, N, x_T, eps_tape, seed, 0, n_steps, out, stream);
}

// ---- stand-alone operators (also the unit-test surface of the kernels)
extern "C" int pdhip_conv2d_nhwc_f16(const void* x, const void* w_packed, const float* bias, const void* residual, void* y,
                                     int N, int H, int W, int Cin, int Cout, int Cout_pad, int taps, const void* zero_page,
                                     void* stream) {
    PD_REQUIRE(x && w_packed && y && zero_page, "pdhip_conv2d_nhwc_f16: null argument");
    return conv_igemm((const half_t*)x, (const half_t*)w_packed, bias, (const half_t*)residual, (half_t*)y, N, H, W, Cin, Cout,
                      Cout_pad, taps, (const half_t*)zero_page, as_stream(stream), pdnn::g_dbg_splitk_ws, pdnn::g_dbg_splitk_floats);
}
/* tuning / test hook: split-K workspace (device floats) for pdhip_conv2d_nhwc_f16 and a forced split factor (0 = automatic) */
extern "C" int pdhip_debug_set_conv_splitk(void* ws, long long ws_floats, int splits) {
    int old = pdnn::g_force_splits;
    pdnn::g_dbg_splitk_ws = (float*)ws; pdnn::g_dbg_splitk_floats = ws ? (size_t)ws_floats : 0; pdnn::g_force_splits = splits;
    if (ws != nullptr && ws_floats >= PD_SK_TICKET_FLOATS) (void)hipMemset(ws, 0, PD_SK_TICKET_FLOATS * sizeof(float));   // k_conv_sk's tickets start at zero
    return old;
}
/* tuning / test hook of the small-M conv kernel (nn_conv_sk.hip): mode 0 = never, 1 = automatic (default), 2 = every eligible layer;
 * tile 0 = automatic, 1..4 = 128x128 / 128x64 / 64x64 / 64x32; splits 0 = automatic.  Returns the previous mode. */
extern "C" int pdhip_debug_set_conv_sk(int mode, int tile, int splits) {
    int old = pdnn::g_sk_mode;
    pdnn::g_sk_mode = mode; pdnn::g_sk_tile = tile; pdnn::g_sk_splits = splits;
    return old;
}
extern "C" int pdhip_debug_set_conv_sk_stages(int stages) { int old = pdnn::g_sk_stages; pdnn::g_sk_stages = stages; return old; }
extern "C" int pdhip_debug_set_conv_sk_kgroups(int kg) { int old = pdnn::g_sk_kg; pdnn::g_sk_kg = kg; return old; }
namespace pdnn { extern thread_local int g_attn_nbuf, g_attn_vt, g_attn_qtn; }
extern "C" int pdhip_debug_set_attn(int nbuf, int vt_form, int qtiles) {
    int old = pdnn::g_attn_nbuf | (pdnn::g_attn_vt << 8) | (pdnn::g_attn_qtn << 16);
    pdnn::g_attn_nbuf = nbuf; pdnn::g_attn_vt = vt_form; pdnn::g_attn_qtn = qtiles;
    return old;
}
extern "C" int pdhip_debug_set_conv_sk_order(int order) { int old = pdnn::g_sk_order; pdnn::g_sk_order = order; return old; }
/* tuning / test hook: 1 (default) = GroupNorm + SiLU applied inside the consuming halo conv, 0 = stand-alone passes */
/* tuning / test hook: 1 (default) = up / down ResBlocks never materialise their resampled x branch; 0 = k_resample passes */
extern "C" int pdhip_debug_set_rr_gn(int max_width) { int old = pdnn::g_rr_gn; pdnn::g_rr_gn = max_width; return old; }
extern "C" int pdhip_debug_set_fold_skip(int on) { int old = pdnn::g_fold_skip; pdnn::g_fold_skip = on; return old; }
extern "C" int pdhip_debug_set_fold_resample(int on) { int old = pdnn::g_fold_resample; pdnn::g_fold_resample = on; return old; }
extern "C" int pdhip_debug_set_fold_finalize_chunks(int chunks) { int old = pdnn::g_fold_finalize_chunks; pdnn::g_fold_finalize_chunks = chunks; return old; }
extern "C" int pdhip_debug_set_fold_finalize(int max_batch) { int old = pdnn::g_fold_finalize; pdnn::g_fold_finalize = max_batch; return old; }
namespace pdnn { extern thread_local int g_gs_variant; }
/* lab hook: 0 = two workgroups per CU, loads one MFMA phase ahead; 1 (default) = one workgroup per CU, activation chunks three K-steps ahead */
extern "C" int pdhip_debug_set_gn_skip_variant(int v) { int old = pdnn::g_gs_variant; pdnn::g_gs_variant = v; return old; }
extern "C" int pdhip_debug_set_fuse_skip(int mode, int min_tiles) {
    int old = pdnn::g_fuse_skip;
    pdnn::g_fuse_skip = mode;
    if (min_tiles > 0) pdnn::g_fuse_skip_min_tiles = min_tiles;
    return old;
}
/* stand-alone: h0 = silu(GroupNorm32(x)) and sk = conv1x1(x) (C -> 256) in one pass over x = [xa | xb] (xb NULL: single tensor, Ca == C);
 * stats [N][32][2] (mean, rstd) of x finished by the caller.  C % 256 == 0, Ca % 64 == 0, H * W % 128 == 0. */
extern "C" int pdhip_gn_silu_skip1x1_nhwc_f16(const void* xa, const void* xb, int Ca, int C, const float* stats, const float* gamma,
                                              const float* beta, const void* w_packed, const float* bias, void* h0, void* sk, int N, int H,
                                              int W, void* stream) {
    PD_REQUIRE(xa && stats && gamma && beta && w_packed && h0 && sk, "pdhip_gn_silu_skip1x1_nhwc_f16: null argument");
    return gn_skip((const half_t*)xa, (const half_t*)xb, Ca, C, stats, gamma, beta, (const half_t*)w_packed, bias, (half_t*)h0, (half_t*)sk,
                   N, H * W, as_stream(stream));
}
extern "C" int pdhip_debug_set_fuse_gn(int on) { int old = pdnn::g_fuse_gn; pdnn::g_fuse_gn = on; return old; }
/* stand-alone: y = conv3x3( silu( GroupNorm32(x) [* (1 + scale) + shift] ) ) (+ residual) with the transform applied inside the
 * conv (halo-resident kernel; W in {32, 64, 128, 256}, H * W % 512 == 0, Cin % 32 == 0).  ws: N*64 + N*64*ceil(HW/256) + N*Cin*2 floats. */
extern "C" int pdhip_gn_silu_conv3x3_nhwc_f16(const void* x, const float* gamma, const float* beta, const float* film,
                                              long long film_stride, const void* w_packed, const float* bias, const void* residual,
                                              void* y, int N, int H, int W, int Cin, int Cout, int Cout_pad, const void* zero_page,
                                              float* ws, long long ws_floats, void* stream) {
    PD_REQUIRE(x && gamma && beta && w_packed && y && zero_page && ws, "pdhip_gn_silu_conv3x3_nhwc_f16: null argument");
    const size_t gws = (size_t)N * 64 * ((H * W + 255) / 256);
    PD_REQUIRE((size_t)ws_floats >= (size_t)N * 64 + gws + (size_t)N * Cin * 2, "pdhip_gn_silu_conv3x3_nhwc_f16: workspace too small");
    hipStream_t s = as_stream(stream);
    float* stats = ws;
    float* table = ws + (size_t)N * 64 + gws;
    PD_TRY(gn_stats((const half_t*)x, N, H * W, Cin, 1e-5f, stats, ws + (size_t)N * 64, gws, s));
    PD_TRY(gn_table(stats, gamma, beta, film, film_stride, N, Cin, table, s));
    // (the debug split-K workspace is shared with pdhip_conv2d_nhwc_f16, whose k_conv_sk route keeps its tickets in the first words)
    float* hws = pdnn::g_dbg_splitk_ws != nullptr && pdnn::g_dbg_splitk_floats > PD_SK_TICKET_FLOATS ? pdnn::g_dbg_splitk_ws + PD_SK_TICKET_FLOATS : nullptr;
    return conv3x3_halo((const half_t*)x, (const half_t*)w_packed, bias, (const half_t*)residual, (half_t*)y, N, H, W, Cin, Cout,
                        Cout_pad, (const half_t*)zero_page, s, nullptr, nullptr, hws, hws ? pdnn::g_dbg_splitk_floats - PD_SK_TICKET_FLOATS : 0, table);
}
/* tuning hook: the APPLY conv on a ready-made (A, B) table [N][Cin/8][16] (tools/bench_conv.py --apply) */
extern "C" int pdhip_debug_conv3x3_apply(const void* x, const float* table, const void* w_packed, const float* bias, const void* residual,
                                         void* y, int N, int H, int W, int Cin, int Cout, int Cout_pad, const void* zero_page, void* stream) {
    PD_REQUIRE(x && table && w_packed && y && zero_page, "pdhip_debug_conv3x3_apply: null argument");
    return conv3x3_halo((const half_t*)x, (const half_t*)w_packed, bias, (const half_t*)residual, (half_t*)y, N, H, W, Cin, Cout,
                        Cout_pad, (const half_t*)zero_page, as_stream(stream), nullptr, nullptr, nullptr, 0, table);
}
extern "C" int pdhip_debug_set_conv_tile(int wmw) { int old = pdnn::g_force_wmw; pdnn::g_force_wmw = wmw; return old; }
extern "C" int pdhip_debug_set_conv_stages(int st) { int old = pdnn::g_force_stages; pdnn::g_force_stages = st; return old; }
/* tuning / test hook: force the conv K-step (32 or 64; 0 = automatic). Returns the previous value. */
extern "C" int pdhip_debug_set_conv_bk(int bk) { int old = pdnn::g_force_bk; pdnn::g_force_bk = bk; return old; }
extern "C" int pdhip_pack_conv_weight_f16(const float* w_oihw, int Cout, int Cin, int taps, void* w_packed, void* stream) {
    PD_REQUIRE(w_oihw && w_packed, "pdhip_pack_conv_weight_f16: null argument");
    k_pack_conv<<<grid_for((long long)Cout * Cin * taps), 256, 0, as_stream(stream)>>>(w_oihw, 0, Cout, Cin, taps, (half_t*)w_packed);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}
extern "C" int pdhip_groupnorm_nhwc_f16(const void* x, const float* gamma, const float* beta, const float* film, int N, int H,
                                        int W, int C, int silu, int resample, void* y, float* stats_ws /*[N*64]*/,
                                        float* ws, long long ws_floats, void* stream) {
    PD_REQUIRE(x && gamma && beta && y && stats_ws && ws, "pdhip_groupnorm_nhwc_f16: null argument");
    PD_TRY(gn_stats((const half_t*)x, N, H * W, C, 1e-5f, stats_ws, ws, (size_t)ws_floats, as_stream(stream)));
    return gn_apply((const half_t*)x, stats_ws, gamma, beta, film, 2LL * C, N, H, W, C, silu, resample, y, 0, as_stream(stream));
}
// output head on its own: GroupNorm(32) statistics + the fused GN -> SiLU -> conv3x3 kernel (f32-equivalent arithmetic)
extern "C" size_t pdhip_unet_head_ws_floats(int N, int H, int W, int C, int Cout) {
    return (size_t)N * 64 + (size_t)N * 64 * ((H * W + 255) / 256) + (size_t)Cout * 9 * C + (size_t)64 * C;
}
extern "C" int pdhip_unet_head_f32(const void* x, const float* gamma, const float* beta, const float* w_oihw, const float* bias,
                                   int N, int H, int W, int C, int Cout, float* y_nchw, float* ws, long long ws_floats, void* stream) {
    PD_REQUIRE(x && gamma && beta && w_oihw && bias && y_nchw && ws, "pdhip_unet_head_f32: null argument");
    PD_REQUIRE((size_t)ws_floats >= pdhip_unet_head_ws_floats(N, H, W, C, Cout), "pdhip_unet_head_f32: workspace too small");
    hipStream_t s = as_stream(stream);
    float* stats = ws;
    float* gws = stats + (size_t)N * 64;
    const size_t gws_floats = (size_t)N * 64 * ((H * W + 255) / 256);
    float* wp = gws + gws_floats;
    half_t* wz = reinterpret_cast<half_t*>(wp + (size_t)Cout * 9 * C);
    PD_TRY(gn_stats((const half_t*)x, N, H * W, C, 1e-5f, stats, gws, gws_floats, s));
    k_pack_conv_f32<<<grid_for((long long)Cout * 9 * C), 256, 0, s>>>(w_oihw, 0, Cout, C, 9, wp);
    PD_LAUNCH_CHECK();
    PD_TRY(head_pack(wp, Cout, C, wz, s));
    return head_gn_silu_conv3x3((const half_t*)x, stats, gamma, beta, wz, bias, y_nchw, N, H, W, C, Cout, s);
}
extern "C" int pdhip_attention_f16(const void* qkv, void* out, int N, int T, int C, int head_dim, void* vt_ws, void* stream) {
    PD_REQUIRE(qkv && out, "pdhip_attention_f16: null argument");
    return attention((const half_t*)qkv, (half_t*)out, N, T, C, head_dim, as_stream(stream), (half_t*)vt_ws);
}
extern "C" int pdhip_philox_normal(float* out, long long n, uint64_t seed, uint64_t stream_id, void* stream) {
    PD_REQUIRE(out && n > 0, "pdhip_philox_normal: bad arguments");
    return philox_normal(out, n, seed, stream_id, as_stream(stream));
}
extern "C" int pdhip_bench_copy16(const void* src, void* dst, long long bytes, int blocks, int unroll, void* stream) {
    return copy16(src, dst, bytes, blocks, unroll, as_stream(stream));
}
extern "C" int pdhip_debug_set_gn_iters(int iters) { const int old = g_gn_iters; g_gn_iters = iters; return old; }
