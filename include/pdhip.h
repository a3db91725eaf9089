/* pdhip.h -- C ABI of libpdhip.so: the MI355X (gfx950) implementation of PointDreamer's
 * project -> inpaint -> unproject texturing path.
 *
 * The reference (YuQiao0303/PointDreamer) exposes no FFI for this path; the boundary is the set of
 * Python call signatures demo.colorize_one_mesh uses (demo.py:93-95, 107-110, 127-129, 150-151,
 * 172-177, 203).  Each entry point below names the reference function (file:line) it replaces.
 * The Python shims in pointdreamer_amd/ keep the reference's names and tensor contracts and route
 * every call here through ctypes.
 *
 * Conventions
 *   - every pointer is a BORROWED device pointer (HBM) owned by the caller, contiguous, row-major;
 *     "bool" tensors are uint8 (torch.bool layout); index tensors are int64 unless stated;
 *   - every call enqueues asynchronously on `stream` (a hipStream_t; NULL = default stream) and never
 *     synchronises, except the few calls documented as returning a host value;
 *   - return value: 0 = OK, negative = error (PDHIP_E_*); pdhip_last_error() gives a thread-local
 *     message.  Nothing throws across the ABI;
 *   - the library holds no global mutable state; scratch memory is passed in by the caller
 *     (workspace pointers) or lives in an opaque handle created by *_create and freed by *_destroy.
 *     The pdhip_debug_set_* tuning / test hooks are THREAD-LOCAL switches (they steer the calling thread's
 *     subsequent launches only), so concurrent callers never see each other's settings.
 *   - camera parameters: 16 float32 per view = R(9, row-major world->camera) t(3) fx fy A B,
 *     NDC = (fx*xc/-zc, fy*yc/-zc, (A*zc+B)/-zc); see DESIGN.md "Arithmetic contract".
 */
#ifndef PDHIP_H
#define PDHIP_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PDHIP_OK 0
#define PDHIP_E_ARG (-1)      /* invalid argument */
#define PDHIP_E_HIP (-2)      /* HIP runtime error (message has the hipError string) */
#define PDHIP_E_STATE (-3)    /* handle in wrong state (e.g. weights missing) */
#define PDHIP_E_NOMEM (-4)
#define PDHIP_E_UNKNOWN_NAME (-5)   /* pdhip_unet_load_tensor: not a tensor of this architecture (strict=False callers skip it) */

int pdhip_version(void);
int pdhip_lab_build(void);   /* 0 for a product build; > 0: that many translation units carry a wrong-result PD_LAB_* timing switch (csrc/common.h) -- tests refuse such a library */
const char* pdhip_last_error(void);

/* ---- P1: ours_utils.get_rendered_hard_mask_and_face_idx_batch, transform + crop part
 *      (pointdreamer/ours_utils.py:93-141).  minmax_ws: 4*V uint32 scratch.
 *      uv_centers[V,2], uv_scales[V] are written only when rescale != 0. */
int pdhip_project_points(const float* cam_params, int V, const float* vertices, int Vn,
                         const float* points, int N, int rescale, double padding,
                         float* pos /*[V,Vn,4]*/, float* vertice_uvs /*[V,Vn,2]*/,
                         float* uv_centers, float* uv_scales,
                         float* point_uvs /*[V,N,2]*/, float* point_depths /*[V,N]*/,
                         uint32_t* minmax_ws, void* stream);

/* ---- P2: the nvdiffrast.rasterize call at ours_utils.py:142-147 (also extract_texture_map.py:57).
 *      zkey_ws: V*R*R uint64 scratch (z keys of the atomic path / face setups of the LDS-tiled path).  Outputs hard_masks[V,R,R] u8, face_idxs[V,R,R] i64 (-1 empty),
 *      depths[V,R,R] f32 (0 empty). */
int pdhip_raster_mesh(const float* pos /*[V,Vn,4]*/, int V, int Vn, const int32_t* faces /*[F,3]*/, int F,
                      int R, uint64_t* zkey_ws, uint8_t* hard_masks, int64_t* face_idxs, float* depths,
                      void* stream);
/* the same with an explicit workspace: pdhip_raster_mesh_ws_bytes(V, F, R) bytes keep meshes up to 65 536 faces on the LDS-tiled path
 * (with the historical V*R*R*8 bytes it is taken up to R*R/17 faces only: 15.4 k at R = 512) */
size_t pdhip_raster_mesh_ws_bytes(int V, int F, int R);
int pdhip_raster_mesh_ws(const float* pos, int V, int Vn, const int32_t* faces, int F, int R, uint64_t* zkey_ws, size_t ws_bytes,
                         uint8_t* hard_masks, int64_t* face_idxs, float* depths, void* stream);
/* tuning / test hook: 0 = automatic (LDS-tiled rasteriser up to 65 536 faces), 1 = force the global 64-bit atomicMin path;
 * both produce identical images.  Returns the previous value. */
int pdhip_debug_set_raster_path(int path);

/* ---- nvdiffrast contract pieces used by the UV-atlas producer (models/get3d/extract_texture_map.py:57-63) and by
 *      optimize_color (pointdreamer/ours_utils.py:1700-1705): barycentrics (u,v) of triangle vertices 0 and 1 at each covered
 *      pixel centre (rast[...,0:2]), and interpolate(attr, rast, tri) = u*a0 + v*a1 + (1-u-v)*a2 (zeros where empty). */
int pdhip_raster_barycentrics(const float* pos /*[V,Vn,4]*/, int V, int Vn, const int32_t* faces, int R,
                              const int64_t* face_idxs /*[V,R,R]*/, float* bary /*[V,R,R,2]*/, void* stream);
int pdhip_interpolate(const float* attr /*[Na,C]*/, int C, const int32_t* tri /*[F,3]*/, const int64_t* face_idxs,
                      const float* bary, long long pixels, float* out /*[pixels,C]*/, void* stream);

/* ---- SURVEY 8(e) configs[4], round 4: S independent shapes of EQUAL sizes (Vn vertices, F faces, N points, atlas A) through the same V
 *      cameras in ONE launch per stage.  Per-shape inputs are stacked ([S, ...]); every per-view array has S*V leading entries, view
 *      g = s * V + v; results equal the per-shape entry points bit for bit (tests/test_gpu_round4.py).  The per-view-independent stages
 *      (pdhip_resize_mask, pdhip_point_visibility*, pdhip_nearest_fill) already take any number of images: call them with S*V. */
int pdhip_project_points_shapes(const float* cam_params /*[V,16]*/, int V, int S, const float* vertices /*[S,Vn,3]*/, int Vn,
                                const float* points /*[S,N,3]*/, int N, int rescale, double padding, float* pos /*[S*V,Vn,4]*/,
                                float* vertice_uvs, float* uv_centers /*[S*V,2]*/, float* uv_scales /*[S*V]*/, float* point_uvs /*[S*V,N,2]*/,
                                float* point_depths /*[S*V,N]*/, uint32_t* minmax_ws /*[4*S*V]*/, void* stream);
int pdhip_raster_mesh_shapes(const float* pos /*[S*V,Vn,4]*/, int V, int S, int Vn, const int32_t* faces /*[S,F,3]*/, int F, int R,
                             uint64_t* zkey_ws, size_t ws_bytes /* pdhip_raster_mesh_ws_bytes(S*V, F, R) */, uint8_t* hard_masks,
                             int64_t* face_idxs, float* depths, void* stream);
int pdhip_hidden_point_removal_shapes(const float* points /*[S,N,3]*/, int N, const double* eyes /*[S*V,3]*/, int V, int S, double radius,
                                      const uint8_t* skip /*[S*V,N] or NULL*/, uint8_t* visibility /*[S*V,N]*/,
                                      void* ws /* pdhip_hpr_ws_bytes(S*V, N); S*V <= 64 */, void* stream);
int pdhip_sparse_views_shapes(const int64_t* point_pixels /*[S*V,N,2]*/, const float* colors /*[S,N,3]*/, const uint8_t* validation,
                              const uint8_t* hard_masks, int V, int S, int N, int res, int point_size, int edge_point_size,
                              double mask_ratio_thresh, float* sparse, float* mask0, float* mask2, float* scale_factors /*[S*V]*/,
                              float* mask_ratios, void* ws /* pdhip_sparse_views_ws_bytes(S*V, N, res) */, void* stream);
int pdhip_texel_visibility_shapes(const float* cam_params, int V, int S, const float* gb_pos /*[S,A,A,3]*/, const uint8_t* mask /*[S,A,A]*/,
                                  int A, const float* uv_centers, const float* uv_scales, double padding, const float* mesh_depths /*[S*V,R,R]*/,
                                  int R, float offset, uint8_t* visibility /*[S*V,A,A]*/, void* stream);
int pdhip_nbf_shrink_shapes(const uint8_t* mask /*[S,A,A]*/, const uint8_t* visibility /*[S*V,A,A]*/, int V, int S, int A,
                            const int32_t* kernels, int K, uint8_t* out /*[K][S*V][A][A]*/, uint8_t* ws /*2*S*V*A*A bytes*/, void* stream);
int pdhip_view_select_blend_shapes(const float* cam_params, int V, int S, const float* gb_pos, const uint8_t* mask, const int64_t* face_id,
                                   int A, const float* f_normals /*[S,F,3]*/, int F, const float* base_dirs /*[V,3]*/, const float* uv_centers,
                                   const float* uv_scales, double padding, const float* scale_factors, const uint8_t* shrinked /*[K][S*V][A][A]*/,
                                   int K, const uint8_t* visibility, int complete_unseen_by_projection, const float* inpainted /*[S*V,3,r,r]*/,
                                   int r, float* atlas /*[S,A,A,3]*/, uint8_t* painted /*[S,A,A]*/, int32_t* view_ids /*[S,A,A], local to the shape*/,
                                   void* stream);

/* ---- ours_utils.optimize_color (pointdreamer/ours_utils.py:1583-1785).
 *      pdhip_rescale_vertices: pos.xy <- clip((((xy-c)/s)*(1-2pad))*factor_v + 0.5, 0, 1)*2-1  (:1688-1695), in place.
 *      pdhip_optimize_color: `iterations` Adam steps (lr, StepLR(15, 0.5)) of the masked L1 texture loss, atlas[3,A,A] f32
 *      updated in place; uv_map[V,res,res,2] / face_idxs[V,res,res] are the UNflipped raster + interpolate outputs;
 *      shrinked[V,A,A] u8 may be NULL; final_images[V,3,res,res] f32 (render of the last iteration) may be NULL. */
int pdhip_rescale_vertices(float* pos /*[V,Vn,4]*/, int V, int Vn, const float* uv_centers, const float* uv_scales,
                           const float* factors /*[V]*/, double padding, void* stream);
size_t pdhip_optimize_color_ws_bytes(int V, int res, int A);
int pdhip_optimize_color(float* atlas, int A, const float* uv_map, const int64_t* face_idxs, int V, int res,
                         const float* inpainted /*[V,3,r,r]*/, int r, const uint8_t* shrinked, double lr, int iterations,
                         float* final_images, void* ws, void* stream);

/* ---- P2b: torchvision Resize(bilinear, no antialias) + .bool() on masks (demo.py:103-104,
 *      ours_utils.py:989-995): out is 1 iff any source pixel with non-zero bilinear weight is set. */
int pdhip_resize_mask(const uint8_t* in /*[B,in_h,in_w]*/, int B, int in_h, int in_w,
                      uint8_t* out /*[B,out_h,out_w]*/, int out_h, int out_w, void* stream);

/* ---- P3: ours_utils.get_point_validation_by_depth (ours_utils.py:153-202).
 *      point_pixels (row,col) may be NULL. */
int pdhip_point_visibility(int cam_res, const float* point_uvs /*[V,N,2]*/, const float* point_depths /*[V,N]*/,
                           const float* mesh_depths /*[V,R,R]*/, int V, int N, float offset,
                           uint8_t* visibility /*[V,N]*/, int64_t* point_pixels /*[V,N,2]*/, void* stream);

/* ---- P3b: ours_utils.get_point_validation_by_o3d (ours_utils.py:204-225), i.e. Open3D hidden_point_removal:
 *      spherical flip about each eye + convex-hull vertex test, all V views in one call, float64 on the device.
 *      eyes: V*3 float64 (device).  ws: pdhip_hpr_ws_bytes(V, N) bytes.  visibility[V,N] u8, written in full (it need
 *      not be cleared by the caller).
 *      skip (may be NULL): [V,N] u8; points already accepted by another test (demo.py:110 ORs the depth test with
 *      this one) are not queried and come back as 1, so the result is directly the OR-ed validation.
 *      Every verdict carries a floating-point certificate (separating direction / enclosing tetrahedron with error
 *      bounds) on the f64 coordinates; what the f64 certificates cannot decide is re-run as a distance iteration, first in
 *      f64, then in double-double arithmetic, so the result is the vertex set of the exact hull of the flipped points.
 *      pdhip_hpr_read_counters (synchronises) reports, for the last call on `ws`, summed over the views: out[0] queries
 *      sent to the double-double iteration, out[1] queries not certifiable even there (exactly degenerate input; reported
 *      hidden), out[2] double-double rounds, out[3] queries sent to the f64 distance iteration. */
size_t pdhip_hpr_ws_bytes(int V, int N);
int pdhip_hidden_point_removal(const float* points /*[N,3]*/, int N, const double* eyes /*[V,3]*/, int V, double radius,
                               const uint8_t* skip, uint8_t* visibility, void* ws, void* stream);
int pdhip_hpr_read_counters(const void* ws, int V, long long* out /*[4]*/, void* stream);

/* ---- demo.py:121-125: point_pixels = clip(long(uv*res)) as (row,col). */
int pdhip_point_pixels(const float* point_uvs /*[V,N,2]*/, int V, int N, int res,
                       int64_t* point_pixels /*[V,N,2]*/, void* stream);
/* both of the above in one pass over the points: the depth test at cam_res and the (row, col) pixels at res. */
int pdhip_point_visibility_pixels(int cam_res, const float* point_uvs, const float* point_depths, const float* mesh_depths,
                                  int V, int N, float offset, uint8_t* visibility /*[V,N]*/, int res,
                                  int64_t* point_pixels /*[V,N,2]*/, void* stream);

/* ---- P4-P6: ours_utils.get_sparse_images -> get_one_sparse_img -> paint_pixels /
 *      get_forground_inner_edge_mask (ours_utils.py:848-882, 954-1044, 456-532), all V views.
 *      ws: pdhip_sparse_views_ws_bytes(V,N,res) bytes.  Outputs are already flipped vertically and
 *      sparse is multiplied by mask0 (ours_utils.py:866).  Duplicate splats: largest write index wins. */
size_t pdhip_sparse_views_ws_bytes(int V, int N, int res);
int pdhip_sparse_views(const int64_t* point_pixels /*[V,N,2]*/, const float* colors /*[N,3]*/,
                       const uint8_t* validation /*[V,N]*/, const uint8_t* hard_masks /*[V,res,res]*/,
                       int V, int N, int res, int point_size, int edge_point_size, double mask_ratio_thresh,
                       float* sparse /*[V,3,res,res]*/, float* mask0, float* mask2,
                       float* scale_factors /*[V]*/, float* mask_ratios /*[V] or NULL*/,
                       void* ws, void* stream);

/* ---- I0 / Uq5: ours_utils.naive_inpainting(method='nearest') (ours_utils.py:610-643) and
 *      unproject.dilate_atlas (unproject.py:480-504): every pixel takes the value of its
 *      Euclidean-nearest site; ties -> lexicographically smallest (row,col).
 *      img/out: B images; element (b,c,y,x) at b*batch_stride + c*chan_stride + (y*W+x)*pix_stride.
 *      mask: per image H*W; mask_is_f32 != 0 -> float (site iff != 0), else uint8.
 *      mask_batch_stride in elements.  ws: pdhip_nearest_fill_ws_ints(B,H,W) int32. */
size_t pdhip_nearest_fill_ws_ints(int B, int H, int W);
int pdhip_nearest_fill(const float* img, float* out, int B, int C, int H, int W,
                       int64_t batch_stride, int64_t chan_stride, int64_t pix_stride,
                       const void* mask, int mask_is_f32, int64_t mask_batch_stride,
                       int32_t* ws, void* stream);

/* ---- I0, texture_gen_method='linear': ours_utils.naive_inpainting(method='linear') (ours_utils.py:610-643), i.e.
 *      scipy.interpolate.griddata(method='linear'): Delaunay triangulation of the sites + barycentric interpolation, NaN outside
 *      the sites' convex hull.  Every unknown pixel finds its own Delaunay triangle (lifted lower-hull facet over the pixel) with
 *      exact integer predicates evaluated in float64; where co-circular sites admit several triangulations a valid one is used.
 *      img/out [B,C,H,W] f32 contiguous, mask per image H*W (f32: site iff != 0, else uint8), H, W <= 2048.
 *      ws: pdhip_linear_fill_ws_bytes(B,H,W).  tri (may be NULL): [B,H,W,3] int32, the triangle's site indices (row-major site
 *      order), -1 at sites, -2 outside the hull.  pdhip_linear_fill_unresolved (synchronises): queries that hit the round cap (0). */
size_t pdhip_linear_fill_ws_bytes(int B, int H, int W);
int pdhip_linear_fill(const float* img, float* out, int B, int C, int H, int W, const void* mask, int mask_is_f32,
                      int64_t mask_batch_stride, void* ws, int32_t* tri, void* stream);
int pdhip_linear_fill_unresolved(const void* ws, int B, int H, int W, int* out, void* stream);
int pdhip_debug_set_linear_local(int on);   /* tuning / test hook: 1 (default) = two local window passes (8x8 tiles with 20x20 windows, then 16x16 tiles with 48x48 windows), global scans only for what they cannot certify; 2 = the 48x48 pass only; 0 = global scans only */

/* test / lab hook: 1 = Uq1-Uq4 run their run-time-view-count kernels even at V = 8 (default 0: the view loop is unrolled at V = 8 and
 * the view selection skips the softmax exponentials where the selected view provably does not depend on them; same results bit for
 * bit, tests/test_gpu_round5.py).  Returns the previous value; thread-local. */
int pdhip_debug_set_unproject_generic(int on);

/* ---- Uq1+Uq2: unproject.unproject texel transform + depth visibility (unproject.py:219-284).
 *      visibility[V,A,A] u8 (0 outside the chart mask). */
int pdhip_texel_visibility(const float* cam_params, int V, const float* gb_pos /*[A,A,3]*/,
                           const uint8_t* mask /*[A,A]*/, int A, const float* uv_centers /*[V,2]*/,
                           const float* uv_scales /*[V]*/, double padding,
                           const float* mesh_depths /*[V,R,R]*/, int R, float offset,
                           uint8_t* visibility, void* stream);

/* ---- N1-N3: unproject.get_shrinked_per_view_per_pixel_visibility_torch (unproject.py:429-475),
 *      utils_2d.detect_edges_in_gray_by_scharr_torch_batch (:799-827), dilate_torch_batch (:833-845).
 *      kernels: K host ints (odd, or kernels[0]==0 -> NBF off: out = visibility).
 *      out[K,V,A,A] u8.  ws: 2*V*A*A bytes. */
int pdhip_nbf_shrink(const uint8_t* mask /*[A,A]*/, const uint8_t* visibility /*[V,A,A]*/, int V, int A,
                     const int32_t* kernels, int K, uint8_t* out, uint8_t* ws, void* stream);
/* Bit-packed transport of boolean texel maps (no reference counterpart: the reference is single-GPU; SURVEY 8(e) prices the view-parallel
 * all-gather record with its visibility maps bit-packed): bit i of out word w = (in[64 w + i] != 0); n % 64 == 0; 16-byte aligned byte side. */
int pdhip_pack_bits(const uint8_t* in /*[n] 0 / non-0*/, long long n, uint64_t* out /*[n / 64]*/, void* stream);
int pdhip_unpack_bits(const uint64_t* in /*[n / 64]*/, long long n, uint8_t* out /*[n] 0 / 1*/, void* stream);
/* the reference's always-on debug triptychs `others/shrink_per_view_edge/{v}.png` (unproject.py:459-474): per view
 * [visibility + background edges (red) + view edges (blue) | view edge mask | border mask of `kernel`] with 10 white columns
 * between panels, rows reversed; out [V][A][3A+20][3] u8 (HWC), ws (2V+1)*A*A bytes. */
int pdhip_nbf_triptych(const uint8_t* mask, const uint8_t* visibility, int V, int A, int kernel, uint8_t* out, uint8_t* ws,
                       void* stream);

/* ---- Uq3+Uq4: view selection + colour gather (unproject.py:298-400).
 *      shrinked[K,V,A,A] (K = number of NBF levels actually consulted), visibility[V,A,A] raw.
 *      view_ids[A,A] int32: chosen view, -100 unseen, -1 outside the chart mask.
 *      atlas[A,A,3] f32, painted[A,A] u8. */
int pdhip_view_select_blend(const float* cam_params, int V, const float* gb_pos, const uint8_t* mask,
                            const int64_t* face_id /*[A,A]*/, int A, const float* f_normals /*[F,3]*/,
                            const float* base_dirs /*[V,3]*/, const float* uv_centers, const float* uv_scales,
                            double padding, const float* scale_factors /*[V]*/,
                            const uint8_t* shrinked, int K, const uint8_t* visibility,
                            int complete_unseen_by_projection,
                            const float* inpainted /*[V,3,r,r]*/, int r,
                            float* atlas, uint8_t* painted, int32_t* view_ids, void* stream);

/* ---- texel compaction in row-major order (unproject.py:223-233): points[P,3], coords[P,2] i64,
 *      optional gather of view_ids[A,A] -> point_view_ids[P] i64.  Writes P to *count_dev (int32).
 *      ws: (A+1) int32. */
int pdhip_compact_texels(const float* gb_pos, const uint8_t* mask, int A, const int32_t* view_ids,
                         float* points, int64_t* coords, int64_t* point_view_ids, int32_t* count_dev,
                         int32_t* ws, void* stream);

/* =============================== rows U1 + D1: guided-diffusion UNet + DDNM ===============================
 * Replaces models/DDNM/ddnm_inpainting.py:15-44 (Inpainter), guided_diffusion/diffusion.py:435-570
 * (get_model, simplified_ddnm_inpainting) and guided_diffusion/unet.py:396-664 (UNetModel, fp16 torso).
 * The handle owns the repacked weights and an activation arena sized for max_batch images. */
typedef struct pdhip_unet pdhip_unet;

/* script_util.create_model (script_util.py:130-185): channel_mult / attention_ds are the already-resolved
 * integer tuples (for 256x256: mult (1,1,2,2,4,4), attention_ds (8,16,32)). */
int pdhip_unet_create(int image_size, int model_channels, int num_res_blocks, const int* channel_mult, int n_mult,
                      const int* attention_ds, int n_att, int num_head_channels, int out_channels, int max_batch,
                      pdhip_unet** out);
void pdhip_unet_destroy(pdhip_unet* u);
long long pdhip_unet_arena_bytes(const pdhip_unet* u);
int pdhip_unet_num_tensors(const pdhip_unet* u);
/* model.load_state_dict (diffusion.py:453): one call per state-dict entry, reference key names, PyTorch layouts
 * (conv OIHW, linear [out,in]); data is a device pointer to float32 (is_f16 = 0) or float16 (1). */
int pdhip_unet_load_tensor(pdhip_unet* u, const char* name, const void* data, int is_f16, const int64_t* shape, int ndim,
                           void* stream);
int pdhip_unet_missing_tensors(const pdhip_unet* u, char* buf, int buf_len);
/* UNetModel.forward (unet.py:635-664): x[N,3,S,S] f32, t[N] f32 -> out[N,out_channels,S,S] f32.
 * A handle owns its activation arena and split-K workspace (ticket words + slabs): at most ONE forward / sampler call of a handle may
 * be in flight at a time (calls on one stream are ordered by the stream; two streams need two handles). */
int pdhip_unet_forward(pdhip_unet* u, const float* x, const float* t, int N, float* out, void* stream);
/* HIP-event timing of the dominant kernel (3x3 implicit-GEMM conv launches) on the launch stream.  enable: 0 off, 1 every
 * forward, k > 1 every k-th forward (an event record costs the stream a few us of pipeline bubble; 108 of them per forward). */
int pdhip_unet_profile(pdhip_unet* u, int enable);
int pdhip_unet_profile_read(pdhip_unet* u, double* total_ms, double* total_flops, long long* launches);
/* the same for the attention launches (QK^T + softmax + PV, 4 T^2 C flop per image) -- north_star's MFMA-utilisation evidence */
int pdhip_unet_profile_read_attention(pdhip_unet* u, double* total_ms, double* total_flops, long long* launches);

/* D1 schedule (diffusion.py:46-113, 770-812): 100 steps t = 990..0; host arrays (any may be NULL). coefs[100][6] =
 * sqrt(1-a_t), sqrt(a_t), sqrt(a_next), sigma_t, c1, c2. */
int pdhip_ddnm_schedule(float* at, float* at_next, int* t, int* t_next, float* coefs);
/* y = mask * (2*img - 1)   (diffusion.py:477-484). */
int pdhip_ddnm_prepare(const float* masked_imgs /*[N,3,HW]*/, const float* masks /*[N,HW]*/, float* y, int N, int HW, void* stream);
/* one fused update (diffusion.py:529-552), in place on x; eps NULL -> on-device Philox noise. */
int pdhip_ddnm_step(float* x, const float* et, int et_channels, const float* y, const float* mask, const float* eps,
                    uint64_t seed, int step, int N, int HW, void* stream);
/* simplified_ddnm_inpainting for N images at once (the reference loops views serially at batch 1):
 * masked_imgs[N,3,S,S] in [0,1], masks[N,S,S] (1 = keep), optional injected x_T[N,3,S,S] and eps_tape[n_steps,N,3,S,S];
 * out[N,3,S,S] in [0,1]. */
int pdhip_ddnm_sample(pdhip_unet* u, const float* masked_imgs, const float* masks, int N, const float* x_T,
                      const float* eps_tape, uint64_t seed, int n_steps, float* out, void* stream);
/* the same with an explicit noise key: image n draws the Philox stream of key first_image_key + n, so a view's noise does not
 * depend on its batch position, on the batch composition or on how views are sharded over ranks (the reference draws
 * independent torch.randn noise per view, diffusion.py:493,552).  pdhip_ddnm_sample == first_image_key 0. */
int pdhip_ddnm_sample_keyed(pdhip_unet* u, const float* masked_imgs, const float* masks, int N, const float* x_T,
                            const float* eps_tape, uint64_t seed, uint64_t first_image_key, int n_steps, float* out, void* stream);

/* stand-alone operators of the engine (NHWC f16); also the unit-test surface of the kernels */
/* tuning / test hook: force the conv K-step (32 or 64; 0 = automatic); returns the previous value */
int pdhip_debug_set_conv_bk(int bk);
int pdhip_debug_set_conv_tile(int geometry);    /* 2 = 128x128 tile / 4 waves, 4 = 256x128 / 8 waves, 8 = 256x256 / 8 waves (wave tile 128x64),
                                                 * 16 = 256x128 / 4 waves, 32 = halo-resident 3x3 kernel (512x128); 0 = automatic */
int pdhip_debug_set_conv_stages(int stages);   /* 2..4 LDS pipeline stages, 12 = 2 stages + hand-scheduled fragment loop; 0 = automatic */
int pdhip_debug_set_conv_splitk(void* ws, long long ws_floats, int splits); /* split-K workspace for pdhip_conv2d_nhwc_f16 + forced factor (0 = automatic) */
int pdhip_debug_set_conv_halo_strips(int mode); /* halo-resident 3x3 kernel on 256-wide images: 0 = automatic (column strips of 128), 1 = full-row tiles; returns the previous value */
int pdhip_pack_conv_weight_f16(const float* w_oihw, int Cout, int Cin, int taps, void* w_packed /*[Cout][taps*Cin] f16*/, void* stream);
int pdhip_conv2d_nhwc_f16(const void* x, const void* w_packed /*[Cout_pad][taps*Cin]*/, const float* bias, const void* residual,
                          void* y, int N, int H, int W, int Cin, int Cout, int Cout_pad, int taps, const void* zero_page,
                          void* stream);
int pdhip_groupnorm_nhwc_f16(const void* x, const float* gamma, const float* beta, const float* film /*[N][2C] or NULL*/, int N,
                             int H, int W, int C, int silu, int resample, void* y, float* stats_ws, float* ws,
                             long long ws_floats, void* stream);
/* y = conv3x3( silu( GroupNorm32(x) [* (1 + scale) + shift] ) ) (+ residual): the in_layers / out_layers chain of a ResBlock
 * (unet.py:183-252: normalization -> SiLU -> conv, FiLM scale-shift in between for out_layers) with the normalisation applied INSIDE
 * the halo-resident conv kernel while it stages its input tile -- no stand-alone GroupNorm pass over the tensor.
 * W in {32, 64, 128, 256}, H*W % 512 == 0, Cin % 32 == 0; ws: N*64 + N*64*ceil(H*W/256) + N*Cin*2 floats. */
int pdhip_gn_silu_conv3x3_nhwc_f16(const void* x, const float* gamma, const float* beta, const float* film /*[N][2Cin] or NULL*/,
                                   long long film_stride, const void* w_packed, const float* bias, const void* residual, void* y,
                                   int N, int H, int W, int Cin, int Cout, int Cout_pad, const void* zero_page, float* ws,
                                   long long ws_floats, void* stream);
int pdhip_debug_conv3x3_apply(const void* x, const float* table /*[N][Cin/8][16]*/, const void* w_packed, const float* bias, const void* residual,
                              void* y, int N, int H, int W, int Cin, int Cout, int Cout_pad, const void* zero_page, void* stream);   /* tuning hook */
int pdhip_debug_set_fold_skip(int on);   /* 1 (default): in the small-M layers a channel-changing ResBlock's skip 1x1 conv (unet.py:206-209, 255) is appended to conv2's K loop (k_conv_sk<10>: out = W2 im2col(h) + Wskip x + (b2 + bskip), one launch and one rounding); 0: its own launch + a residual read; returns the previous value */
int pdhip_debug_set_fold_resample(int on);   /* 1 (default): the resampled x branch of up / down ResBlocks is folded into its consumers; 0: k_resample passes */
int pdhip_debug_set_fold_finalize(int max_batch);   /* largest UNet batch at which GroupNorm-apply reduces the conv epilogues' statistics partials itself (no k_gn_finalize_oct launch); default 8, 0 = never */
int pdhip_debug_set_fold_finalize_chunks(int chunks);   /* above that batch the in-kernel statistics are kept for tensors whose producers left at most this many chunks per image (default 16; 0 = batch rule only); returns the previous value */
int pdhip_debug_set_conv_sk(int mode, int tile, int splits);   /* small-M conv kernel (in-launch split-K combine): mode 0 never / 1 automatic / 2 every eligible layer; tile 0 auto, 1..4 = 128x128, 128x64, 64x64, 64x32; splits 0 auto */
int pdhip_debug_set_conv_sk_stages(int stages);   /* lab hook: LDS stage count of the small-M conv kernel (2, 3, 4 where the tile allows; 0 = default) */
int pdhip_debug_set_conv_sk_kgroups(int kg);   /* lab hook: K-groups (4-wave groups working on alternate K-steps of one tile) per workgroup of the small-M conv kernel: 1, 2, 4 (64-row tiles), 8 = loader-specialised (4 compute + 4 loader waves), 12 = loader-specialised with EIGHT loader waves (the 128-row tiles; the default of the 128x128 tile since round 5); 0 = default */
int pdhip_debug_set_attn(int nbuf, int vt_form, int qtiles);   /* lab hook, T >= 128 attention kernel: qtiles = 16-query tiles per wave, 1 (64 queries per workgroup) or 2 (128), 0 automatic; nbuf = LDS chunk buffers, 2 (K / V requested one chunk ahead), 3 (two ahead), 0 automatic; vt_form 1 = V transposed into the workspace by a separate pass and read plainly, 0 = V staged row-major and read with the LDS transpose read */
int pdhip_debug_set_conv_sk_order(int order);   /* lab hook: tile order of the small-M conv kernel inside an XCD's run: 0 automatic, 1 pixel tiles fastest (weight slices shared through L2), 2 n-tiles fastest */
/* h0 = silu(GroupNorm32(x)) AND sk = conv1x1(x) (C -> 256 channels) in ONE pass over x: the in_layers normalisation and the
 * skip_connection of a channel-changing ResBlock (unet.py:197-209, 236-256) both read the block input, which for the decoder blocks is a
 * channel concat [xa (Ca channels) | xb (C - Ca)] that is never materialised (unet.py:657-659; xb NULL: x = xa, Ca == C).  stats [N][32][2] =
 * (mean, rstd) of x per GroupNorm group, finished (pdhip_groupnorm_nhwc_f16's stats_ws).  C % 256 == 0, Ca % 64 == 0, H*W % 128 == 0;
 * w_packed [256][C] f16, bias [256] f32 or NULL; h0 [N,H,W,C], sk [N,H,W,256] f16. */
int pdhip_gn_silu_skip1x1_nhwc_f16(const void* xa, const void* xb, int Ca, int C, const float* stats, const float* gamma, const float* beta,
                                   const void* w_packed, const float* bias, void* h0, void* sk, int N, int H, int W, void* stream);
int pdhip_debug_set_gn_iters(int iters);   /* lab hook: pixels per thread of the GroupNorm-apply kernel (1 .. 32; 0 = default); returns the previous value */
int pdhip_debug_set_gn_skip_variant(int v);   /* lab hook of the one-pass GroupNorm + skip kernel: 1 (default; 0 is taken as 1) = activation chunk k + 1 requested at the top of iteration k, 64-pixel tiles where 128-pixel tiles would leave CUs idle (< 256 tiles), 2 = as 1 with 128-pixel tiles always; returns the previous value */
int pdhip_debug_set_fuse_skip(int mode, int min_tiles);   /* ResBlock GroupNorm-apply + skip 1x1 as one pass: mode 0 never / 1 (default) layers of at least min_tiles 128-pixel tiles (default 1024; <= 0 keeps the value) / 2 every eligible layer; returns the previous mode */
int pdhip_debug_set_fuse_gn(int on);   /* 1: the UNet uses the fused form wherever the halo kernel serves a conv; 0 (default, measured faster): stand-alone passes */
/* ---- complete_unseen_by='neighbor' (pointdreamer/unproject.py:93-196, demo.py:180-200).
 * The mesh subdivision stays on the host (as in the reference); these are the per-texel / per-vertex kernels. */
/* flags[f] = 1 for every face f that owns a chart texel no view painted (demo.py:180-181). face_id [A*A] int64 (-1 = background). */
int pdhip_mark_unpainted_faces(const int64_t* face_id, const uint8_t* painted, int A, int F, uint8_t* flags /*[F]*/, void* stream);
/* texel (row, col) = long(clip(uv * A, 0, A-1))[(1, 0)], colours = atlas[texel], count = mask[texel] (unproject.py:130-139) */
int pdhip_vertex_texel_fetch(const float* vert_uvs /*[V,2]*/, int V, const float* atlas /*[A,A,3]*/, const uint8_t* mask /*[A,A]*/,
                             int A, int32_t* texel /*[V,2]*/, float* colors /*[V,3]*/, float* count /*[V]*/, void* stream);
/* one Jacobi round of unproject.py:160-166 over the `invalid` vertices (CSR neighbours, ascending); *colored = number of
 * invalid vertices that hold a colour afterwards (device int; the host drives the loop as the reference does) */
int pdhip_neighbor_diffuse_round(const int32_t* rowptr /*[V+1]*/, const int32_t* colidx, int V, const int32_t* invalid, int IV,
                                 float* colors, float* count, float* tmp /*[IV*4]*/, int* colored, void* stream);
/* atlas[texel[v]] = colors[v], mask[texel[v]] = 1 (unproject.py:187-188); several vertices on one texel: largest index wins */
int pdhip_scatter_vertex_colors(const int32_t* texel, const float* colors, int V, float* atlas, uint8_t* mask,
                                int32_t* owner_ws /*[A*A]*/, int A, void* stream);

/* ---- I/O edges in native code (host functions; no device work except pdhip_chw_f32_to_hwc_u8).
 * PLY (utils/other_utils.py:155-163): vertex element with x,y,z and red,green,blue, binary_little_endian or ascii. */
long long pdhip_io_ply_count(const char* path);                 /* number of vertices, -1 on error */
int pdhip_io_read_ply_xyzrgb(const char* path, float* xyz /*[n,3] host*/, uint8_t* rgb /*[n,3] host*/, long long n);
/* OBJ + MTL, byte-identical to savemeshtes2 (models/get3d/get3d_utils/utils_3d.py:27-64). Host arrays (coordinates as
 * double: `%f` of a float32 value is printed from its exact double). */
int pdhip_io_write_obj_mtl(const char* obj_path, const char* mtl_path, const char* texture_stem, const double* points, long long P,
                           const double* tcoords, long long T, const int64_t* faces, const int64_t* facetex, long long F);
/* 8-bit RGB / RGBA PNG (utils/utils_2d.py:351-399 save path); hwc: host [H,W,channels] */
int pdhip_io_write_png(const char* path, const uint8_t* hwc, int H, int W, int channels, int zlib_level);
/* device: out[h,w,c] = uint8(clip(img[c,h,w] * 255, 0, 255)) -- the conversion the reference does on the host before saving */
int pdhip_chw_f32_to_hwc_u8(const float* img, int C, int H, int W, uint8_t* out, void* stream);

/* Output head of the UNet on its own (models/DDNM/guided_diffusion/unet.py:613-617): GroupNorm(32) -> SiLU -> conv3x3 in
 * float32-equivalent arithmetic.  x: f16 NHWC [N,H,W,C] (C in {32,64,128,256}); w_oihw f32 [Cout][C][3][3], Cout 3 or 6;
 * y f32 NCHW [N,Cout,H,W]; ws: pdhip_unet_head_ws_floats() device floats. */
size_t pdhip_unet_head_ws_floats(int N, int H, int W, int C, int Cout);
int pdhip_unet_head_f32(const void* x, const float* gamma, const float* beta, const float* w_oihw, const float* bias,
                        int N, int H, int W, int C, int Cout, float* y_nchw, float* ws, long long ws_floats, void* stream);
/* Row-resident convolution of the UNet's 8^2 / 16^2 / 32^2 levels at small batch (csrc/nn_conv_rr.hip): 3x3 (taps 9) or 1x1 (taps 1) conv over
 * x = [x (Ca channels) | x2 (C - Ca), may be NULL] with the GroupNorm32 (+ FiLM) (+ SiLU) of the input applied while staging -- the pair
 * `GroupNorm32 -> [scale / shift] -> SiLU -> conv` of ResBlock.in_layers / out_layers (guided_diffusion/unet.py:185-260) as ONE launch; gn_mode 0 raw input,
 * 1 GroupNorm, 2 GroupNorm + SiLU; statistics from the producers' octet partials partA / partB ([N][chunks][channels / 8][2] = sum, sum of squares);
 * film rows (scale[C] | shift[C]) or NULL.  xs / xs2 (Cs channels, Cs1 in xs): the ResBlock's skip_connection 1x1 (unet.py:255) appended to the K loop,
 * raw input, or NULL.  wf: fragment-major weights from pdhip_conv_rr_pack_f16 (source: [Cout_pad][taps * C + Cs] as pdhip_pack_conv_weight_f16 lays them out).
 * ws: >= 4096 zeroed floats (tickets) + split-K slices; gn_part (may be NULL): octet partials of y, *gn_chunks chunks per image.  H == W in {8, 16, 32}. */
int pdhip_conv_rr_f16(const void* x, const void* x2, int C, int Ca, int gn_mode, const float* gamma, const float* beta, const float* film,
                      long long film_stride, const float* partA, int chunksA, const float* partB, int chunksB, const void* xs, const void* xs2, int Cs,
                      int Cs1, int taps, const void* wf, const float* bias, const void* residual, int res_up, void* y, int N, int H, int W, int Cout,
                      float* ws, long long ws_floats, float* gn_part, int* gn_chunks, void* stream);
long long pdhip_conv_rr_weight_halfs(int Cin, int taps, int Cs, int Cout);
int pdhip_conv_rr_pack_f16(const void* w_packed, int Cin, int taps, int Cs, int Cout, void* wf, void* stream);
/* test helpers: octet partials of a tensor as a conv epilogue leaves them; the two-pass reference of the fused input transform (k_gn_apply with in-kernel statistics) */
int pdhip_gn_octet_partials_f16(const void* x, int N, int HW, int C, int chunks, float* part, void* stream);
int pdhip_gn_apply_parts_f16(const void* x, const void* x2, int Ca, int C, const float* partA, int chunksA, const float* partB, int chunksB, const float* gamma,
                             const float* beta, const float* film, long long film_stride, int N, int H, int W, int silu, void* y, void* stream);
/* 3x3 conv of the UNet's 64^2 / 128^2 levels at small batch on 256-pixel x 64-channel halo tiles (csrc/nn_conv_ht.hip; nn.Conv2d(k=3, padding=1) of
 * ResBlock.in_layers / out_layers, guided_diffusion/unet.py:185-260): x [N,H,W,Cin] f16, w_packed [Cout_pad][9 * Cin] from pdhip_pack_conv_weight_f16 (Cout_pad % 64 == 0),
 * H == W in {32, 64, 128}; residual optional (res_up: read at half resolution); splitk_ws (may be NULL: K unsplit): >= 4096 + tiles * slabs * 16384 floats, the first 4096
 * zeroed once (self-resetting ticket counters) -- K is cut in 2 / 4 slabs combined inside the launch when the tiles leave half of the CUs idle; gn_part (may be NULL):
 * GroupNorm octet partials of y, *gn_chunks = H * W / 256 per image. */
int pdhip_conv_ht_f16(const void* x, const void* w_packed, const float* bias, const void* residual, int res_up, void* y, int N, int H, int W, int Cin, int Cout,
                      int Cout_pad, const void* zero_page, float* splitk_ws, long long splitk_ws_floats, float* gn_part, int* gn_chunks, void* stream);
int pdhip_conv_ht_plan(int N, int H, int W, int Cin, int Cout, int Cout_pad, long long splitk_ws_floats, int* routed, int* slabs);   /* host-only: would the engine's automatic routing (conv_route, csrc/nn_gemm.hip) send this 3x3 layer to the halo-tile kernel, and in how many K-slabs */
int pdhip_debug_set_conv_ht(int mode, int slabs);   /* 256 x 64 halo-tile conv: mode 0 never / 1 automatic (batch 1-4 of the 64^2, 128^2 levels) / 2 every eligible layer; slabs 0 automatic, 1 / 2 / 4 forced K-slabs; returns the previous mode */
int pdhip_debug_set_rr_gn(int max_width);   /* UNet engine: largest image width at which a conv routed to the row-resident kernel also applies the GroupNorm (+ FiLM) + SiLU in front of it (default 0 = never: no gain measured inside the forward; 8 / 16 / 32 = up to that width); returns the previous value */
int pdhip_debug_set_conv_rr(int mode, int variant, int slabs);   /* row-resident conv: mode 0 never / 1 automatic / 2 every eligible layer; variant 0 auto (1: 8^2, 2: 16^2 whole image, 3: 32^2 bands, 4: 16^2 half image, 5: 8^2 with 128-channel units); slabs 0 auto = K slices of the conv source; returns the previous mode */
int pdhip_attention_f16(const void* qkv /*[N,T,3C]*/, void* out /*[N,T,C]*/, int N, int T, int C, int head_dim,
                        void* vt_ws /*N*T*C halfs, non-NULL selects the 128-query MFMA kernel for T % 128 == 0, head_dim 64, N*heads % 8 == 0 (QKVAttentionLegacy, unet.py:341-373); the buffer is written only in the transposed-V lab form (pdhip_debug_set_attn); NULL: the 64-query kernel*/, void* stream);
int pdhip_philox_normal(float* out, long long n, uint64_t seed, uint64_t stream_id, void* stream);
/* Calibration only (no reference counterpart): a streaming device-to-device copy, 16 bytes per lane, `unroll` & 15 (1, 2, 4, 8; 0 = 4) loads in
 * flight per lane, `unroll` >> 4 the form (0 grid-stride sweep, 1 the same with nontemporal loads / stores, 2 one contiguous slab per workgroup), `blocks` workgroups of 256 threads (0 = 2048).  bench.py reports its rate (read + write bytes) as
 * roofline.calibration.copy16_gbs next to torch's copy_: the ceiling the HBM-bound GroupNorm passes are judged against. */
int pdhip_bench_copy16(const void* src, void* dst, long long bytes, int blocks, int unroll, void* stream);

#ifdef __cplusplus
}
#endif
#endif
