/*
 * gflow_hip.h -- C ABI of libgflow_hip.so, the MI355X (gfx950) native rasteriser,
 * loss and optimiser kernels behind GFlow's per-frame Gaussian-splatting fit.
 *
 * This is the drop-in boundary: the five operators GFlow calls on the external
 * CUDA extension `msplat` (gflow/utils/render.py:21-105, gflow/trainer.py:955)
 * plus the loss / Adam pieces of the iteration loop (gflow/trainer.py:387-558).
 * msplat's own binding layer is not visible in the reference tree, so each entry
 * point cites the REFERENCE CALL SITE it serves.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless named h_*; tensors are dense,
 *     row-major, float32 unless stated; `stream` is a hipStream_t (NULL = default);
 *   - no allocation and no host synchronisation inside: the caller supplies outputs and
 *     workspaces, all work is enqueued on `stream` (graph-capturable).  Process-wide state
 *     is limited to: the thread-local last HIP error (gfl_last_hip_error), the cached
 *     CU count of the device, the optional stage profiler (gfl_profile_*) and THREE
 *     environment switches, read once per process -- every switch the library has:
 *       GFL_EWA_MFMA=1        J Sigma J^T on the matrix cores (same results to rounding; gfl_ewa_on_mfma() reports it);
 *       GFL_RESERVED=0        every iteration bins on the exact three-launch path (no reserved tile regions): the sorted
 *                             lists, render and gradients are bit-identical either way;
 *       GFL_FWD_SPLIT_MIN=<n> list length from which the forward blend walks a queue's first tile on four CUs, sixteen splats
 *                             a step: scheduling only -- that walk forms a pixel's transmittance products in tree order, so T and
 *                             the render differ in the last bits on the tiles that take it (which tiles do depends on the
 *                             schedule as well), lists and contributor counts not at all;
 *     tests/test_gpu_switches.py flips each one in a process of its own and holds the results against the defaults.
 *     (Rounds 3-4 had fourteen; the measured-and-rejected variants behind the others now live in tools/experiments/.)
 *   - return value: GFL_OK or a negative gfl_status; HIP launch errors are
 *     returned as GFL_ERR_HIP and the hipError_t is kept in gfl_last_hip_error();
 *   - *_bwd functions OVERWRITE their gradient outputs (they zero what they
 *     accumulate into).
 */
#ifndef GFLOW_HIP_H
#define GFLOW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* gfl_stream_t; /* hipStream_t */

typedef enum gfl_status {
    GFL_OK = 0,
    GFL_ERR_INVALID = -1,   /* null pointer, negative size, unsupported channel count */
    GFL_ERR_WORKSPACE = -2, /* workspace smaller than gfl_*_workspace_bytes() */
    GFL_ERR_HIP = -3        /* a HIP call failed; see gfl_last_hip_error() */
} gfl_status;

/* Rasteriser constants.  They are internal to msplat and NOT observable from the
 * reference: every one is an assumption taken from the published
 * 3DGS/EWA formulation, kept here and mirrored by oracle/msplat_oracle.py.
 * A maintainer who can read the real msplat sources overrides any of them at build time
 * (make -C gflow_amd/csrc CONSTS="-DGFL_NEAREST=0.01f -DGFL_LOWPASS=0.0f") and sets the oracle's
 * copies to match; gfl_constants() reports what a built library uses and
 * tests/test_abi.py::test_library_and_oracle_use_the_same_constants compares the two.
 * GFL_TILE is structural (8x8 pixels per wave, four waves per tile) and stays 16. */
#define GFL_TILE 16
#ifndef GFL_NEAREST
#define GFL_NEAREST 0.2f
#endif
#ifndef GFL_EXTENT
#define GFL_EXTENT 1.3f
#endif
#ifndef GFL_FOV_CLAMP
#define GFL_FOV_CLAMP 1.3f
#endif
#ifndef GFL_LOWPASS
#define GFL_LOWPASS 0.3f
#endif
#ifndef GFL_EIG_FLOOR
#define GFL_EIG_FLOOR 0.1f
#endif
#ifndef GFL_RADIUS_SIGMA
#define GFL_RADIUS_SIGMA 3.0f
#endif
#ifndef GFL_ALPHA_MIN
#define GFL_ALPHA_MIN (1.0f / 255.0f)
#endif
#ifndef GFL_ALPHA_MAX
#define GFL_ALPHA_MAX 0.99f
#endif
#ifndef GFL_T_MIN
#define GFL_T_MIN 1e-4f
#endif
/* where inside pixel (px, py) the composite is SAMPLED: at (px + GFL_PIXEL_CENTER, py + GFL_PIXEL_CENTER) in the coordinates
 * project_point returns.  0 = pixel centres at integer coordinates (3DGS's ndc2Pix convention; what GFlow itself suggests:
 * splats are created AT integer pixels, complex_texture_sampling.py:39, and targets are read at uv.long(), trainer.py:473);
 * 0.5f = the gsplat convention.  If upstream msplat samples at +0.5 every render here is half a pixel off until this is
 * set.  Only the sampling moves: uv, radius and tiles_touched are coordinate quantities and stay. */
#ifndef GFL_PIXEL_CENTER
#define GFL_PIXEL_CENTER 0.0f
#endif
#define GFL_MAX_BLEND_CHANNELS 4 /* per launch; the host splits wider features */

/* 300 (round 5): gfl_fit_state.overflow is int32[4] (was [1] before 200's reserved regions), gfl_tile_sort_ordered reads a
 * trailer of GFL_SORT_ORDER_TRAILER ints behind order[T][4], gfl_fit_iterations' flags are GFL_ITER_RESERVED only
 * (GFL_ITER_PRE_DONE / _PRE_NEXT / _ODD, gfl_fit_next_preprocess_supported and gfl_bwd_rows_on are gone), the fit
 * workspace is smaller.  301: gfl_fit_iteration_snapshot.  302: the tile sorts no longer fill a table of list positions
 * (gfl_tile_sort_with_slots is gone, gfl_tile_sort_ordered / _reserved lost their rec / slot_inv / slot_pool arguments): the
 * per-splat launch finds its pair rows without one.  303: GFL_PIXEL_CENTER, gfl_constants_n.  304: gfl_fit_state.cu_count.
 * A binding checks gfl_version() >= GFL_VERSION of the header it was written for. */
#define GFL_VERSION 304
int gfl_version(void);
/* out[10] = TILE, NEAREST, EXTENT, FOV_CLAMP, LOWPASS, EIG_FLOOR, RADIUS_SIGMA, ALPHA_MIN, ALPHA_MAX, T_MIN of this build */
int gfl_constants(float* out10);
/* 303: the same list with what came later behind it -- [10] = PIXEL_CENTER.  Writes min(n, GFL_N_CONSTANTS) values, returns
 * GFL_N_CONSTANTS (so a binding written for a shorter list keeps working and one written for a longer list sees the gap). */
#define GFL_N_CONSTANTS 11
int gfl_constants_n(float* out, int n);
/* 1 when this process runs the J Sigma J^T contraction of gfl_fit_forward / gfl_render_fwd on the matrix cores
 * (GFL_EWA_MFMA=1 in the environment when the library first looked), 0 for the VALU form */
int gfl_ewa_on_mfma(void);
const char* gfl_status_string(int status);
int gfl_last_hip_error(void);
/* bytes of scratch any *_bwd that reduces camera gradients needs for N splats */
size_t gfl_reduce_workspace_bytes(int N);

/* ---- A4  msplat.project_point  (render.py:21-24,116-119; trainer.py:955) ------
 * uv[N,2], depth[N,1]; culled splats get uv=(0,0), depth=0 (render.py:29). */
int gfl_project_point_fwd(const float* xyz, const float* intr, const float* extr, int N, int W, int H,
                          float nearest, float extent, float* uv, float* depth, gfl_stream_t stream);
/* d_xyz[N,3], d_extr[12] from d_uv[N,2], d_depth[N,1]; `depth` is the forward output */
int gfl_project_point_bwd(const float* xyz, const float* intr, const float* extr, const float* depth,
                          const float* d_uv, const float* d_depth, int N, float* d_xyz, float* d_extr,
                          void* workspace, size_t workspace_bytes, gfl_stream_t stream);

/* ---- A5  msplat.compute_cov3d  (render.py:37-41) -------------------------------
 * rotate is WXYZ; visible is uint8[N]; cov3d[N,6] = xx,xy,xz,yy,yz,zz */
int gfl_cov3d_fwd(const float* scale, const float* rotate, const uint8_t* visible, int N, float* cov3d,
                  gfl_stream_t stream);
int gfl_cov3d_bwd(const float* scale, const float* rotate, const uint8_t* visible, const float* d_cov3d,
                  int N, float* d_scale, float* d_rotate, gfl_stream_t stream);

/* ---- A6  msplat.ewa_project  (render.py:44-49) ---------------------------------
 * conic[N,3] upper-triangular inverse 2-D covariance, radius[N] int32,
 * tiles_touched[N] int32 */
int gfl_ewa_fwd(const float* xyz, const float* cov3d, const float* intr, const float* extr, const float* uv,
                const uint8_t* visible, int N, int W, int H, float* conic, int32_t* radius,
                int32_t* tiles_touched, gfl_stream_t stream);
/* d_xyz[N,3], d_cov3d[N,6], d_extr[12] from d_conic[N,3]; `radius` is the forward output */
int gfl_ewa_bwd(const float* xyz, const float* cov3d, const float* intr, const float* extr,
                const int32_t* radius, const float* d_conic, int N, int W, int H, float* d_xyz,
                float* d_cov3d, float* d_extr, void* workspace, size_t workspace_bytes, gfl_stream_t stream);

/* ---- A7  msplat.sort_gaussian  (render.py:52-54) -------------------------------
 * Two phases so the host can size gaussian_ids_sorted exactly when it wants to:
 *   gfl_bin_count: tile_offsets[T+1] = exclusive scan of per-tile splat counts
 *                  (tile_offsets[T] = K, the number of splat-tile pairs);
 *   gfl_bin_sort : fills ids[0..min(K,K_cap)) ordered by (tile, depth, id) and
 *                  tile_range[T,2]; *overflow (device int32) is set to 1 when
 *                  K > K_cap (the lists are then truncated, never overrun).
 * `cutoff` may be NULL; if given (float[N]) a tile is only listed when the splat's
 * exact alpha>=1/255 disc (radius^2 = cutoff) reaches it -- output-preserving
 * tightening used by the fused path. T = ceil(W/16)*ceil(H/16). */
size_t gfl_bin_workspace_bytes(int N, int K_cap, int W, int H);
int gfl_bin_count(const float* uv, const int32_t* radius, const float* cutoff, int N, int W, int H,
                  int32_t* tile_offsets, gfl_stream_t stream);
int gfl_bin_sort(const float* uv, const float* depth, const int32_t* radius, const float* cutoff, int N, int W,
                 int H, const int32_t* tile_offsets, int K_cap, int32_t* ids, int32_t* tile_range,
                 int32_t* overflow, void* workspace, size_t workspace_bytes, gfl_stream_t stream);

/* ---- A8  msplat.alpha_blending  (render.py:58-64,68-74,84-90,99-105,148-154) ---
 * feature[N,C_total]; this launch composites channels [c0, c0+C), 1<=C<=4, into
 * out[C,H,W] (channel-planar, the caller offsets `out` for c0>0).
 * final_T[H,W] and n_contrib[H,W] (int32) are saved for the backward. */
int gfl_blend_fwd(const float* uv, const float* conic, const float* opacity, const float* feature,
                  int C_total, int c0, int C, const int32_t* ids, const int32_t* tile_range, float bg, int W,
                  int H, float* out, float* final_T, int32_t* n_contrib, gfl_stream_t stream);
/* Accumulates d_uv[N,2], d_conic[N,3], d_opacity[N], d_feature[N,C_total] (channels
 * c0..c0+C).  zero_first != 0 zeroes all four first (use 0 for the 2nd.. chunk). */
int gfl_blend_bwd(const float* uv, const float* conic, const float* opacity, const float* feature,
                  int C_total, int c0, int C, const int32_t* ids, const int32_t* tile_range, float bg, int W,
                  int H, const float* final_T, const int32_t* n_contrib, const float* d_out, int N,
                  float* d_uv, float* d_conic, float* d_opacity, float* d_feature, int zero_first,
                  gfl_stream_t stream);

/* ---- A17  compute_sh: real spherical harmonics of degree 0..3 -> colour ------------
 * Optional operator: GFlow never calls msplat.compute_sh (its colour is sigmoid(rgb),
 * trainer.py:68), so there is no reference call site; basis and constants follow the
 * 3D Gaussian Splatting code base.  shs[N][K][3], K = (degree+1)^2 in {1,4,9,16};
 * dirs[N][3] unit view directions; visible[N] or NULL; out[N][3] = sum_k Y_k(dir) shs[k]
 * (no +0.5, no clamp).  Backward: d_shs[N][K][3], d_dirs[N][3] (may be NULL). */
int gfl_sh_fwd(const float* shs, const float* dirs, const uint8_t* visible, int N, int K, float* out,
               gfl_stream_t stream);
int gfl_sh_bwd(const float* shs, const float* dirs, const uint8_t* visible, const float* d_out, int N, int K,
               float* d_shs, float* d_dirs, gfl_stream_t stream);

/* ---- A9  apply_float_colormap(depth,"turbo",non_zero=True)  (color.py:24-44) ----
 * Entirely on the device (the reference round-trips through the host every
 * iteration).  lut[256,3]; out[N,3]; workspace >= 16 bytes. */
int gfl_colormap_nonzero(const float* value, int N, const float* lut, float* out, void* workspace,
                         size_t workspace_bytes, gfl_stream_t stream);

/* ---- A10/A11  photometric + SSIM + depth loss, forward and backward ------------
 * (trainer.py:452-488; utils/pytorch_ssim.py:17-37)
 * render[4,H,W] = rgb + depth_map planes; gt_rgb[H,W,3]; gt_depth[H,W];
 * keep[H,W] uint8 or NULL (0 = pixel of the moving region, zeroed as trainer.py:453-455,484);
 * depth_ab[2] = (depth_a, depth_b) on the device.
 * Outputs: d_render[4,H,W] = d(lambda_rgb*(mse + 1 - ssim) + lambda_depth*depth)/d render;
 *          err_px[H,W] = per-pixel mse (the densification error map, trainer.py:459);
 *          sums[8] (device) = {sum mse_px, sum ssim_map, sum depth_term, d/d depth_a, d/d depth_b, 0,0,0}
 *          (un-normalised sums; the d/d terms already carry lambda_depth / (H*W)).
 * workspace: gfl_loss_workspace_bytes(W,H). */
size_t gfl_loss_workspace_bytes(int W, int H);
/* same kernels, but the scalar sums are left as per-block partial rows inside `workspace`
 * (p_ssim[n_ssim] floats: SSIM-map sums; p_grad[n_grad][4] floats: mse, depth term, d/d depth_a,
 * d/d depth_b) for a consumer that folds them itself (the fused iteration's camera kernel) */
int gfl_loss_fwd_bwd_partials(const float* render, const float* gt_rgb, const float* gt_depth, const uint8_t* keep,
                              const float* depth_ab, float lambda_rgb, float lambda_depth, int W, int H,
                              float* d_render, float* err_px, void* workspace, size_t workspace_bytes,
                              const float** p_ssim, int* n_ssim, const float** p_grad, int* n_grad,
                              gfl_stream_t stream);
/* The ground truth does not change during the iterations of a fit: gfl_loss_prepare_gt computes its
 * two SSIM statistics once (gt_stats[3][2][H][W]: conv(y), conv(y^2), same arithmetic as the full
 * kernel) and the _cached variant then filters three maps instead of five.  keep must be the mask
 * the later calls use. */
int gfl_loss_prepare_gt(const float* gt_rgb, const uint8_t* keep, int W, int H, float* gt_stats, gfl_stream_t stream);
int gfl_loss_fwd_bwd_partials_cached(const float* render, const float* gt_rgb, const float* gt_depth,
                                     const uint8_t* keep, const float* depth_ab, float lambda_rgb, float lambda_depth,
                                     int W, int H, float* d_render, float* err_px, void* workspace,
                                     size_t workspace_bytes, const float* gt_stats, const float** p_ssim, int* n_ssim,
                                     const float** p_grad, int* n_grad, gfl_stream_t stream);
int gfl_loss_fwd_bwd(const float* render, const float* gt_rgb, const float* gt_depth, const uint8_t* keep,
                     const float* depth_ab, float lambda_rgb, float lambda_depth, int W, int H,
                     float* d_render, float* err_px, float* sums, void* workspace, size_t workspace_bytes,
                     gfl_stream_t stream);

/* ---- A13  Adam (torch.optim.Adam defaults, trainer.py:153,554) ------------------
 * One fused step over n floats: p -= lr * m_hat / (sqrt(v_hat) + eps) with bias
 * correction for step `*d_step + 1` read on the device (so a captured graph can
 * replay); `lr_scale` multiplies lr (LinearLR factor computed on the device when
 * total_iters > 0: 1 + (lr_end_factor-1)*min(step,total)/total; trainer.py:384 uses
 * start 1.0, end 0.1).
 * row_zero_grad (uint8 per row of `row_len` floats, or NULL): rows flagged 1 see a
 * zero gradient (moments still decay), which is what trainer.py:543-551 does.
 * gfl_step_increment bumps the device step counter once all groups have stepped. */
int gfl_adam_step(float* param, const float* grad, float* m, float* v, int64_t n, int row_len,
                  const uint8_t* row_zero_grad, float lr, float beta1, float beta2, float eps,
                  const int32_t* d_step, float lr_end_factor, int total_iters, gfl_stream_t stream);
int gfl_step_increment(int32_t* d_step, gfl_stream_t stream);

/* ---- Fused fit iteration  (trainer.py:387-558 in one call) ----------------------
 * Everything the iteration loop does between "build input_group" (trainer.py:390)
 * and "scheduler.step()" (trainer.py:555): activations, render_multiple(["rgb","uv",
 * "depth","depth_map"]), the rgb/SSIM/depth/var/flow/still losses, backward, gradient
 * masking, Adam for the splats, the camera pose and the depth affine pair.
 * Plain structs of device pointers and sizes; the host owns every buffer.
 *   params/adam_m/adam_v [cap][16] f32 rows: x y z | sx sy sz | qw qx qy qz | opacity |
 *                         r g b | pad pad   (raw, pre-activation values, trainer.py:79-86)
 *   rec   [cap][12]: u v A B | C opacity r g | b depth cutoff radius   (outputs: uv =
 *                    rec[:,0:2], depth = rec[:,9], what render.py:21-49 returns)
 *   d_rec [cap][12] or NULL: dL/d(rec[:,0:10]) of the last backward -- nothing in the iteration reads it; written when given
 * Optional per-splat inputs (NULL = term absent): flow_target[cap][2] + flow_w[cap]
 * (weight = mask/(2 count), trainer.py:511-528), still_target[cap][3] + still_w[cap]
 * (weight = mask/count, trainer.py:505-509), row_flags[cap] (bit0: still splat -- xyz gradient
 * frozen, trainer.py:543-546; bit1: the row carries a still/moving label, i.e. it existed when the
 * labels were made: the scale term and nothing else looks at it).
 * workspace: gfl_fit_workspace_bytes(...) bytes, ZERO-INITIALISED ONCE by the host and then left
 * alone between calls: besides scratch it holds the tile scheduler's feedback (the work the
 * backward blend measured per tile, used to balance the next iteration's blend launches over the
 * CUs).  Garbage there cannot change a result, only the balance.  The tile grid is limited to
 * 16384 tiles (GFL_ERR_INVALID beyond). */
typedef struct gfl_fit_state {
    int32_t N, cap, W, H, K_cap;
    int32_t gt_cached;   /* 1: gfl_fit_prepare_targets ran for the current gt_rgb / keep (ignored with foot_flags) */
    float *params, *adam_m, *adam_v;
    float *rec, *d_rec;
    const float *flow_target, *flow_w, *still_target, *still_w;
    const uint8_t* row_flags;
    float *pose, *pose_m, *pose_v;             /* [7] qx qy qz qw tx ty tz (trainer.py:41) */
    float *depth_ab, *depth_ab_m, *depth_ab_v; /* [2] (trainer.py:145-146) */
    const float* intr;                          /* [4] */
    float* extr;                                /* [12] out: world->camera used this iteration */
    float* d_extr;                              /* [12] out: dL/d extr */
    int32_t* step;                              /* device step counter (Adam t-1, LinearLR epoch) */
    const float* gt_rgb;                        /* [H][W][3] */
    const float* gt_depth;                      /* [H][W] or NULL */
    uint8_t* keep;                              /* [H][W] or NULL (0 = masked-out pixel); IN/OUT of
                                                 * gfl_fit_forward when foot_flags is set */
    const uint8_t* move_mask;                   /* [H][W] or NULL: the frame's move mask (informational; the
                                                 * caller initialises keep = !move_mask for the camera-only stage) */
    const uint8_t* foot_flags;                  /* [cap] or NULL.  Non-NULL: every forward clears from keep the
                                                 * footprint of the flagged splats -- the mask GFlow gets from an
                                                 * extra render of the tentative moving splats with "grey > 0",
                                                 * accumulated over the iterations as trainer.py:426-451 does
                                                 * (move_mask = move_gs_mask | move_mask inside the loop) */
    float *render, *final_T;                    /* [4][H][W], [H][W] */
    int32_t* n_contrib;                         /* [H][W] */
    float *d_render, *err_px, *sums;            /* [4][H][W], [H][W], [8] as gfl_loss_fwd_bwd */
    int32_t *tile_offsets, *ids, *tile_range, *overflow; /* [T+1], [K_cap], [T][2], [4].  overflow[0] is sticky: 1 = a
                                                          * forward produced more than K_cap pairs (or wide-splat pair rows) and
                                                          * dropped some, 2 = GFL_ITER_RESERVED without reserved regions.
                                                          * It is only ever raised by the binning launches of a forward --
                                                          * never by a launch that also reads it --, so every update launch of
                                                          * one iteration sees the same value.  While it
                                                          * is set, gfl_fit_backward_step steps NOTHING (rows, moments, pose,
                                                          * depth affine and step counter stay) and adds 1 to overflow[1]: the
                                                          * caller grows the lists, clears the words and runs overflow[1]
                                                          * iterations again (gflow_amd/fused.py: settle_overflow).
                                                          * overflow[2] != 0: THIS iteration is void (a tile outgrew its
                                                          * reserved region: nothing is stepped, overflow[1] counts it, the
                                                          * next iteration is fine again -- with overflow[0] == 0 the caller
                                                          * just runs overflow[1] more iterations); overflow[3]: the same,
                                                          * between the binning launch and the tile sort.
                                                          * tile_offsets[0..T) is written by the exact binning path only;
                                                          * tile_offsets[T] = the number of pairs, always; after an iteration
                                                          * on reserved regions the lists in ids have gaps between them
                                                          * (tile_range says where each one is) */
    void* workspace;
    size_t workspace_bytes;                     /* >= gfl_fit_workspace_bytes() */
    int32_t cu_count;                           /* 304: compute units this state's launches may use -- 0 = the whole device.  The
                                                 * blend launches are persistent grids with one tile queue per CU: a caller that
                                                 * runs a state on a CU-masked stream (hipExtStreamCreateWithCUMask; several clips
                                                 * side by side on one device, each on its share of every XCD: gflow_amd/fit_video.py)
                                                 * says so here, and grids and queues are sized for that share.  Fixed for the life
                                                 * of the state's workspace (the queues in it are built for this many). */
    int32_t reserved_;
} gfl_fit_state;

typedef struct gfl_fit_hyper {
    float bg, nearest, extent;
    float lambda_rgb, lambda_depth, lambda_var, lambda_flow, lambda_still;
    float lambda_scale;        /* trainer.py:495-502: mean over the rows inside the image (still rows in the
                                * camera-only stage, moving rows in the joint stage, by row_flags) of |scale| / depth */
    float lr, lr_camera, beta1, beta2, eps, lr_end_factor;
    int32_t total_iters;       /* LinearLR total_iters, 0 = constant lr */
    int32_t freeze_rgb;        /* trainer.py:537-540 */
    int32_t freeze_all_splats; /* camera_only, trainer.py:548-551: every splat gradient is zeroed.  The rows and their Adam moments
                                * are then left untouched (not read, not written): what Adam does with zero gradients and
                                * the zero moments a fresh optimiser starts from (trainer.py:383) */
    int32_t step_camera;       /* 1: pose and depth affine are stepped; 0: neither (after densification replaced the
                                * optimiser, trainer.py:951).  While the camera cannot move -- 0, or 1 with lr_camera == 0 (the
                                * first frame and every joint stage) -- an iteration computes no pose gradient and has no camera
                                * launch: d_extr is set to zero, the pose moments stay as they are; loss sums, depth affine
                                * and step counter as always.  2 = as 1, and the gradient (d_extr, pose moments) is wanted
                                * although lr_camera is 0. */
} gfl_fit_hyper;

size_t gfl_fit_workspace_bytes(int cap, int K_cap, int W, int H);
/* render only: fills rec, ids, tile_range, render, final_T, n_contrib, extr */
int gfl_fit_forward(const gfl_fit_state* st, const gfl_fit_hyper* hp, gfl_stream_t stream);
/* loss + backward + optimiser step on the state gfl_fit_forward left behind */
int gfl_fit_backward_step(const gfl_fit_state* st, const gfl_fit_hyper* hp, gfl_stream_t stream);
int gfl_fit_iteration(const gfl_fit_state* st, const gfl_fit_hyper* hp, gfl_stream_t stream);
/* ``count`` iterations back to back (trainer.py:387-558 ``count`` times; what a caller captures into ONE graph).
 *
 * Reserved tile regions (round 4, the default): the last launch of every full iteration gives each tile a region of the
 * key array sized by what the tile holds now plus a margin (count + count / 4 + 32), and an iteration that FOLLOWS a full
 * iteration bins straight into those regions -- preprocess, column scan and scatter are one launch (one returning atomic
 * per (block of 512 splats, tile) reserves the block's part of the region), six launches per iteration instead of eight.
 * The sorted lists, the render and every gradient are those of the exact path; only the lists' positions in ids differ.
 * Iterations 2 .. count of a call always take it; the first does with
 *   GFL_ITER_RESERVED  the previous launches on this state were a full iteration (gfl_fit_iterations / gfl_fit_iteration /
 *                      gfl_fit_backward_step) and the splats have not been replaced since (checked on the device as far
 *                      as it can be: without reserved regions *overflow becomes 2).  The regions are a PREDICTION: a tile
 *                      that outgrows its region voids that one iteration (overflow[2], counted in overflow[1]; the regions
 *                      reserved at its end are sized by what the tiles wanted): nothing is stepped by it, but what its
 *                      forward left behind -- render, lists, loss sums -- is TRUNCATED; a caller that looks at those
 *                      (snapshot, log entry) checks overflow[2] first and runs the iteration again (gflow_amd/trainer.py).
 * gfl_fit_reserved_supported: 1 if this state's iterations can (tile grids of up to 4096 tiles; GFL_RESERVED=0 in the
 * environment switches it off). */
#define GFL_ITER_RESERVED 8
int gfl_fit_iterations(const gfl_fit_state* st, const gfl_fit_hyper* hp, int count, int flags, gfl_stream_t stream);
int gfl_fit_reserved_supported(const gfl_fit_state* st, const gfl_fit_hyper* hp);
/* ---- the fused rasteriser as a differentiable operator (no loss, no optimiser) ------------------------
 * render(gaussians, camera) -> {rgb, depth_map, uv, depth} = render_multiple(input_group, ["rgb", "uv", "depth",
 * "depth_map"]) of render.py:6-108 in ONE call (last row), and its backward.
 * Same state struct as the fit iteration, read differently:
 *   params [cap][16] hold the ACTIVATED attributes the reference passes to msplat (xyz | scale | unit quaternion
 *   wxyz | opacity | rgb | pad pad), st->extr is an INPUT (world->camera, 12 floats), pose / Adam / target / loss
 *   fields are ignored (may be NULL); hp supplies bg, nearest, extent only.
 * gfl_render_fwd fills rec (uv = rec[:,0:2], depth = rec[:,9]), render[4][H][W] (rgb planes + depth_map),
 * final_T, n_contrib, ids, tile_range.  gfl_render_bwd takes dL/d render [4][H][W] and, optionally (NULL = 0),
 * dL/d uv [N][2] and dL/d depth [N] and returns d_params [N][16] (same column layout as params, the two pad
 * columns zero) and d_extr[12]; it must follow the gfl_render_fwd whose state it differentiates. */
int gfl_render_fwd(const gfl_fit_state* st, const gfl_fit_hyper* hp, gfl_stream_t stream);
int gfl_render_bwd(const gfl_fit_state* st, const gfl_fit_hyper* hp, const float* d_render, const float* d_uv,
                   const float* d_depth, float* d_params, float* d_extr, gfl_stream_t stream);
/* The three snapshot images GFlow keeps every 10th iteration (trainer.py:573-582) in one call, entirely on the
 * device: rgb of the last forward, the turbo-coloured depth map and the centre blobs (render.py:76-106; both composites
 * of the same lists, the per-splat values are derived while the records are staged), each clamped, scaled by 255 and
 * truncated like render2img (render.py:158-166).  out_u8 [3 images][H][W][3] uint8; lut [256][3] = the turbo table.
 * For a forward that has already run (an evaluation render, a staged copy): the lists are walked again for
 * depth_map_color. */
size_t gfl_fit_snapshot_workspace_bytes(int N, int W, int H);
int gfl_fit_snapshot(const gfl_fit_state* st, const gfl_fit_hyper* hp, const float* lut, uint8_t* out_u8, void* workspace,
                     size_t workspace_bytes, gfl_stream_t stream);
/* One full iteration (gfl_fit_iteration, exact binning path) whose forward ALSO leaves those three images in out_u8 -- what
 * trainer.py:573-582 does every 10th iteration with three more renders.  rgb and depth_map_color come out of ONE walk of the
 * lists (the second composite has the first one's alphas and transmittances: three more sums per pixel), center from a small
 * kernel over the same lists; nothing is composed twice and nothing waits for the backward.  +~25 us on that iteration
 * (gfl_fit_snapshot after it: ~130 us).  lut [256][3] = the turbo table; out_u8 [3][H][W][3]. */
int gfl_fit_iteration_snapshot(const gfl_fit_state* st, const gfl_fit_hyper* hp, const float* lut, uint8_t* out_u8,
                               gfl_stream_t stream);
/* Copies from one engine to another (same image size, same row count N, capacities at least as large) everything
 * gfl_fit_snapshot reads of a forward: records, sorted ids, tile ranges, the rgb planes of the render and the forward's
 * tile queues -- one launch, ~9 MB at 480p / 60 000 splats.  The snapshot of iteration i can then be taken from the copy,
 * on another stream, BESIDE iteration i + 1 (gflow_amd/trainer.py: _snapshot_async) instead of between the two. */
int gfl_fit_snapshot_stage(const gfl_fit_state* src, const gfl_fit_state* dst, gfl_stream_t stream);
/* once per ground-truth image / keep mask: SSIM statistics of the target into the workspace; set
 * st->gt_cached = 1 afterwards (0 is always valid: everything is then recomputed per iteration) */
int gfl_fit_prepare_targets(const gfl_fit_state* st, gfl_stream_t stream);
/* where the last gfl_fit_forward's tile schedules live in the workspace (device pointers): queue c
 * holds d_counts[c] items d_lists[c * queue_capacity + k] = tile | priority << 28.  Every tile of the
 * frame appears in exactly one queue.  The backward blend's queues (weights: the units it counted per tile in the last
 * iteration) and, _fwd, the forward blend's (its own counts).  For tests and tools; nothing needs them to run. */
int gfl_fit_schedule_info(const gfl_fit_state* st, int* n_queues, int* queue_capacity, const int32_t** d_lists,
                          const int32_t** d_counts);
int gfl_fit_schedule_info_fwd(const gfl_fit_state* st, int* n_queues, int* queue_capacity, const int32_t** d_lists,
                              const int32_t** d_counts);
/* the per-tile sort of gfl_bin_sort alone (keys already scattered into segments) */
int gfl_tile_sort_only(const int32_t* tile_offsets, int T, int K_cap, void* keys, int32_t* ids,
                       int32_t* tile_range, gfl_stream_t stream);
/* same, with the order the workgroups take the tiles in given by the caller: order[T][4] = {tile, start, end, split} per
 * position (16-byte aligned; positions are assigned to the XCDs in contiguous runs, T / 8 each, like the tiles without it),
 * followed by a trailer of GFL_SORT_ORDER_TRAILER ints: trailer[0] = n_split (0 .. GFL_SORT_MAX_SPLIT), trailer[1 + j] =
 * the position of the j-th split list, whose order entry carries split = 1 + j (0 everywhere else).  A split list (64 ..
 * 2048 keys) is cut at a pivot key and sorted by two workgroups, half each; the result is the same sorted list.
 * gfl_fit_iteration's scatter launch writes all of it: every XCD's longest lists first (their workgroups then start with
 * the launch instead of in its second round), the lists of more than 768 keys split.  Any permutation that keeps a tile
 * inside its XCD's run, with or without split entries, is as good for the result. */
#define GFL_SORT_MAX_SPLIT 64
#define GFL_SORT_ORDER_TRAILER (GFL_SORT_MAX_SPLIT + 4)
int gfl_tile_sort_ordered(const int32_t* order, int W, int H, int K_cap, void* keys, int32_t* ids, int32_t* tile_range,
                          gfl_stream_t stream);
/* same for reserved tile regions: order[T][4] = {tile, start, capacity, split}; the list of `tile` is the first
 * min(fill[position], capacity) keys behind `start` (fill[T]: what the binning launch counted, by position in the order);
 * tile_counts[tile] = fill[position];
 * void_words (may be NULL): void_words[0] = void_words[1], void_words[1] = 0 (gfl_fit_state.overflow + 2). */
int gfl_tile_sort_reserved(const int32_t* order, const int32_t* fill, int32_t* tile_counts, int32_t* void_words, int W, int H,
                           int K_cap, void* keys, int32_t* ids, int32_t* tile_range, gfl_stream_t stream);

/* ---- optional per-stage timing of the fused iteration ---------------------------
 * HIP events are recorded on the launch stream around the stages whose bit is set in
 * stage_mask: 0 preprocess, 1 colscan, 2 scatter, 3 tile sort, 4 blend forward, 5 loss,
 * 6 blend backward, 7 preprocess-backward + Adam, 8 camera/affine Adam.
 * gfl_profile_read synchronises on the recorded events, fills total_ms[9] / counts[9]
 * and clears the records. */
/* device self-test of the wave64 reduce-scatter used by the blend backward: in[64][10] ->
 * out_scatter[10] (permlane-swap reduce-scatter) and out_dpp[10] (plain DPP sums) */
int gfl_selftest_reduce10(const float* in, float* out_scatter, float* out_dpp, gfl_stream_t stream);
/* device self-test of the EWA contraction Sigma2 = M Sigma M^T (gfl_math.hpp) both ways: m[n][6] = the two rows of
 * M = J W, cov[n][6] = xx xy xz yy yz zz -> out_valu[n][3], out_mfma[n][3] = (a, b, c) before the low-pass.  The
 * MFMA form (v_mfma_f32_4x4x1_16b_f32, sixteen splats per instruction) is what GFL_EWA_MFMA=1 puts into the fused
 * preprocess kernel; the VALU form is the default. */
int gfl_selftest_cov2d(const float* m, const float* cov, int n, float* out_valu, float* out_mfma, gfl_stream_t stream);
/* The blend kernels' culling test on its own: for n records (rows of 12 floats as in gfl_fit_state.rec; columns
 * u v A B | C o . . | . . cutoff .) the 4-bit mask of the four box x box pixel boxes at (x0, y0) the splat can reach
 * with alpha >= 1/255 (box = 8: the blocks of a tile, 4: the quarters of a block), and `truth`: the boxes that hold a
 * visible pixel by brute force with the kernels' own alpha test.  The mask must contain the truth. */
int gfl_selftest_block_mask(const float* rec, int n, int x0, int y0, int box, int32_t* mask, int32_t* truth, gfl_stream_t stream);

/* ---- moving-region hull (HOST function: plain pointers, no device work) ----------------------------------
 * The ring of the concave hull of n 2-D points (concaveman's algorithm, what the `concave_hull` package behind
 * gflow/utils/concave_hull.py:73-92 implements; gflow/trainer.py:604-609 masks the moving region with it).
 * points_xy[n][2]: sorted by x then y, no duplicates (numpy.unique(axis=0)); concavity 2, length_threshold 0 are the
 * package's defaults.  Writes up to cap_vertices ring vertices (not closed) to ring_xy and returns their number, or a
 * negative gfl_status (GFL_ERR_WORKSPACE: cap_vertices too small; 2 n + 8 always suffices). */
int gfl_concave_hull(const double* points_xy, int n, double concavity, double length_threshold, double* ring_xy,
                     int cap_vertices);

/* sizeof(gfl_fit_state), sizeof(gfl_fit_hyper): lets an FFI binding verify its struct mirrors */
int gfl_abi_sizes(int* sizeof_fit_state, int* sizeof_fit_hyper);

#define GFL_PROFILE_STAGES 9
int gfl_profile_enable(unsigned stage_mask);
int gfl_profile_read(double* total_ms, int* counts, int n_stages);

#ifdef __cplusplus
}
#endif
#endif /* GFLOW_HIP_H */
