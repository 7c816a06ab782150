/*
 * gisnav_amd.h -- C ABI of the MI355X-native PoseNode hot path (libgisnav_amd.so).
 *
 * GISNav has no formal plugin interface; PoseNode hard-codes three seams.
 * Each entry point below names the reference code it stands in for
 * (paths relative to the reference tree, ros/gisnav/gisnav/core/):
 *
 *   gn_create / gn_load_tensor   LightGlueMatcher("sift", params={n_layers:9, filter_threshold:.5,
 *                                depth_confidence:-1, width_confidence:-1}).to(device).eval()
 *                                                                       pose_node.py:109-121
 *   gn_match                     RootSIFT + self._matcher(desc_q, desc_r, laf_q, laf_r)
 *                                                                       pose_node.py:278-287
 *   gn_gather_points             kp[idx] gathers, MIN_MATCHES gate, _compute_3d_points
 *                                                 pose_node.py:289-303, _shared.py:95-102
 *   gn_pnp_ransac                cv2.solvePnPRansac(..., iterationsCount=10) + cv2.Rodrigues
 *                                                                       _shared.py:104-117
 *   gn_estimate                  the whole of PoseNode._pose lines 246-308 for a batch of pairs
 *                                (gn_set_active_kpts: padded size per call, for batches well below max_kpts)
 *   gn_pose_to_earth (+ gn_proj_to_affine, gn_wgs84_to_ecef)   the georeferencing after the pose
 *                                                 pose_node.py:333-381, _transformations.py:298-393
 * and, widening to the feeders of that path:
 *   gn_sift_detect_and_compute(_batch)   cv2.SIFT_create().detectAndCompute(img, None)
 *                                                 pose_node.py:122,230-232; twist_node.py:93,227-245
 *   gn_rotate_crop_center / gn_stereo_reference   StereoNode reference raster   stereo_node.py:229-262,292-335
 *   gn_vo_match / gn_vo_estimate cv2.BFMatcher.knnMatch(k=2) + ratio test + compute_pose   twist_node.py:95,248-289
 *
 * Conventions: plain C, no torch types.  Every data pointer is a DEVICE pointer unless the
 * parameter is marked "host".  `stream` is a hipStream_t passed as void* (NULL = default
 * stream).  All work is stream-ordered; no entry point synchronises except gn_debug_read and
 * gn_sift_detect_and_compute(_batch) (one sync at the end, to return the keypoint counts).
 * Return value: 0 on success, negative gn_status on failure; nothing throws.  One context
 * per GPU; a context is thread-compatible, not thread-safe (PoseNode calls from one executor
 * thread at a time, gisnav/__init__.py:140-154).  DIFFERENT contexts (gn_ctx and gn_loftr alike)
 * may be driven from different host threads at the same time: nothing the launch paths consult
 * is process-wide (the kernel-family selectors and gn_last_error(NULL) are per thread).
 */
#ifndef GISNAV_AMD_H
#define GISNAV_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gn_ctx gn_ctx;

enum gn_status {
  GN_OK = 0,
  GN_ERR_ARG = -1,       /* bad argument (NULL, out of range, B > max_batch, n > max_kpts) */
  GN_ERR_HIP = -2,       /* a HIP runtime call failed; see gn_last_error */
  GN_ERR_NAME = -3,      /* unknown tensor name */
  GN_ERR_SHAPE = -4,     /* tensor shape mismatch */
  GN_ERR_WEIGHTS = -5,   /* forward called before every required tensor was loaded */
  GN_ERR_ARCH = -6       /* device is not gfx950 */
};

enum gn_precision {
  GN_PREC_F32 = 0,       /* every contraction on f32 MFMA (parity mode) */
  GN_PREC_BF16_ATTN = 1, /* QK^T / PV on bf16 MFMA with f32 softmax+accumulate (what kornia's
                            Attention does in fp16 on CUDA when flash=True); projections,
                            FFN and the match head stay on the exact-f32 MFMA */
  GN_PREC_F32X3_BF16_ATTN = 2, /* as 1, but projections / FFN / match-head GEMMs run f32-ACCURATE on the bf16
                            matrix pipe: every f32 operand is split exactly into three bf16 terms and six
                            partial products are accumulated in f32 (error vs fp64 at or below the exact-f32
                            MFMA path's; see tests) */
  GN_PREC_F16X2_BF16_ATTN = 3, /* as 2 at half the matrix-pipe work: every f32 operand is split into two fp16
                            terms (round to nearest, 22 significant bits, subnormals honoured by the gfx950
                            matrix pipe) and three partial products are accumulated in f32; weight planes are
                            pre-scaled by a power of two.  Domain |activation| < 65504 (the reference's fp32
                            nn.Linear has no such limit): GUARDED -- every kernel that writes an activation in
                            this format raises a context word when a value leaves the range, a tripped call
                            reports ZERO matches instead of inf / NaN, and gn_set_guard(ctx, 2) re-runs it in
                            mode 2 (see gn_set_guard).  Error vs fp64 at the level of an f32 accumulation
                            (see tests) */
  GN_PREC_F16X2_F16_ATTN = 4  /* as 3, with QK^T / PV on the FP16 matrix instruction (v_mfma_f32_32x32x16_f16): q, k, v and
                            the probabilities are rounded to fp16 (11 significant bits) instead of bf16 (8), f32 softmax
                            and accumulation as before -- the arithmetic of the reference's own CUDA path (kornia casts
                            q, k, v to half for F.scaled_dot_product_attention when flash=True; pose_node.py:285-287 via
                            SURVEY.md:314), at the same matrix-pipe rate as bf16.  q / k / v share the guarded fp16 domain
                            of the activations (|value| < 65504, else the guard word is raised; the guard-2 re-run uses bf16
                            attention operands, which have f32's range) */
};

enum gn_kpt_format {
  GN_KPT_LAF = 0,        /* 6 floats/keypoint: kornia LAF (2x3 row-major) -- the B1 seam's argument */
  GN_KPT_XYSA = 1,       /* 4 floats/keypoint: x, y, size, angle_deg (cv2.KeyPoint fields,
                            KEYPOINT_DTYPE in _shared.py:26-35); the LAF of pose_node.py:267-276
                            is formed in registers */
  GN_KPT_RECORD = 2      /* 133 floats/keypoint = the RAW 532-byte wire record of KEYPOINT_DTYPE
                            (_shared.py:26-35: x, y, z, size, angle, descriptor[128]; all float32,
                            little endian) exactly as OrthoStereoImage.query_sift
                            carries it (pose_node.py:207-213): kpt_* point at [B][stride] records, the
                            descriptor is read from the record itself and the desc_* arguments are
                            ignored (may be NULL).  The message bytes go to the device as they are --
                            no host-side unpacking (np.frombuffer + column_stack).  SIFT contexts only. */
};
/* OR into kpt_format when the descriptors are ALREADY RootSIFT-normalised (the B1 seam: PoseNode
 * normalises at pose_node.py:278-284 before calling the matcher object); otherwise gn_match applies
 * RootSIFT itself to raw SIFT descriptors. */
#define GN_DESC_ROOTSIFT 0x100

/* Library / build info: "gisnav_amd <ver> gfx950 src:<source digest>". */
const char* gn_version(void);
/* Human-readable text for the last failure on this context (or global if ctx is NULL). */
const char* gn_last_error(const gn_ctx* ctx);

/* The fp16-range guard of GN_PREC_F16X2_BF16_ATTN (no effect in the other modes):
 *   0  off (no checks; an out-of-range activation silently becomes inf / NaN -- for kernel experiments only);
 *   1  (default) flag: stream-ordered, no host sync.  A call in which any activation reached |x| >= 65504 returns
 *      n_match = 0 for every pair (and therefore ok = 0 from gn_estimate); gn_get_guard_status tells the host.  With
 *      gn_set_substreams(n > 1) every sub-batch group has its own guard word: a trip zeroes the matches of the pairs of
 *      THAT group only, and gn_get_guard_status reports the OR over the groups of the last call;
 *   2  flag + fallback: gn_match / gn_estimate synchronise the stream once after the matcher, and a tripped call is
 *      re-run transparently with every operand split exactly into three bf16 terms (the arithmetic of mode
 *      GN_PREC_F32X3_BF16_ATTN: f32 range and accuracy, ~1.5x slower) before the call returns.  The Python mirrors of
 *      the reference's objects (LightGlueMatcher, PoseNode) use this: they synchronise after the matcher anyway. */
int gn_set_guard(gn_ctx* ctx, int mode);
/* Synchronises `stream`; *last_call_tripped = 1 if the most recent matcher run on this context left the fp16 range,
 * *trips_total = number of tripped calls observed so far (either pointer may be NULL). */
int gn_get_guard_status(gn_ctx* ctx, void* stream, int32_t* last_call_tripped, int64_t* trips_total);

/* The margin certificate of the correspondence indices.  kornia's LightGlueMatcher returns `match_indices` that PoseNode uses as exact
 * integers (ros/gisnav/gisnav/core/pose_node.py:285-297); the fast precision modes compute the assignment scores P with an arithmetic
 * error, so a decision whose margin is smaller than that error may come out differently from the exact-f32 arithmetic.  With the certificate
 * on, the match head keeps the runner-up of every row / column maximum of P and reports, per PAIR, whether every decision of that pair
 * is safe: (a) every row whose best score is >= log(filter_threshold) - eps leads its runner-up by more than 2 eps, (b) the same for every
 * column, (c) no row's best score lies within eps of log(filter_threshold).  eps is the stated bound on |P_mode - P_exact| for the
 * entries that decide (measured per weight set by tools/certify_eps.py / PoseEngine.calibrate_certify; pass < 0 to keep the current
 * value).  A pair that satisfies (a)-(c) has exactly the exact arithmetic's match list (proof in DESIGN.md).
 *   mode 0  off (default): no flags are written;
 *   mode 1  flags only, stream-ordered, no host synchronisation: gn_get_uncertain reads them;
 *   mode 3  as mode 2, but gn_estimate with sub-batch streams resolves call n's flags after call n + 1 has been enqueued (or in gn_flush): the caller keeps
 *           the inputs AND outputs of call n untouched until then (two alternating output sets); no host wait on an idle GPU;
 *   mode 2  certified results: gn_match / gn_estimate synchronise the stream once per call, read the flags and run every flagged pair
 *           again -- matcher, and for gn_estimate also gather + PnP -- on GN_PREC_F32's kernels (the context keeps f32 weights and f32
 *           workspaces in every mode).  A call whose activations left the fp16 range is flagged as a whole (flag value 2), so this mode
 *           also takes over gn_set_guard(2)'s fallback.  In a GN_PREC_F32 context nothing is re-run: the flags (for eps_f32) are counted.
 * Results of a pair do not depend on which other pairs were re-run. */
int gn_set_certify(gn_ctx* ctx, int mode, float eps, float eps_f32);
/* Arithmetic of the block tail (ffn.0 -> LayerNorm -> GELU -> ffn.3 with out_proj folded in; kornia `x + ffn(cat[x, msg])`) on bulk grids in the f16x2
 * modes: 3 (default) = every operand as two fp16 terms, three partial products, f32-accurate; 2 = the activations' fp16 HIGH term only (22-bit weights x
 * 11-bit activations, f32 accumulation: the arithmetic of the attention input projections, and the rounding the fp16 attention applies to q, k, v anyway) --
 * a third less matrix-pipe work.  The error of the assignment scores grows (gn_calibrate_certify measures it), so this setting is meant to run under the
 * margin certificate (gn_set_certify(2 / 3)), which makes the returned correspondence indices independent of the fast pass's arithmetic.  Small grids
 * (fewer than 256 tiles of 128 tokens) keep three products.
 * 0 = the level FOLLOWS THE CERTIFICATE (needs gn_set_certify(2 / 3) and a gn_calibrate_certify made under this setting, which measures eps for both
 * levels; three products otherwise): every matcher call also evaluates the other level's certificate on its scores, the context counts over windows
 * of >= 64 certified pairs how many pairs each level flags, and runs the next window on two products only when that would flag at most 1 pair in 64
 * more than three products (a flagged pair costs an exact-f32 re-run, ~4 fast passes; the two-product pass saves ~7 % of one).  Starts on three
 * products.  The returned indices do not depend on the level (both are certified against the same exact arithmetic); scores differ within eps.
 * gn_load_tensor discards the two levels' calibration (three products until gn_calibrate_certify / gn_set_ffn_level_eps is called again). */
int gn_set_ffn_products(gn_ctx* ctx, int products);
/* The level the next call runs on (2 / 3), the eps calibrated for each (< 0: not calibrated), and out4 = {certified calls that ran on two products, on three
 * products, level switches, 1 when the automatic setting is in effect} since gn_reset_certify_stats.  Any pointer may be NULL. */
int gn_get_ffn_level(gn_ctx* ctx, int32_t* level, float* eps2, float* eps3, int64_t* out4);
/* The two levels' eps and the level to start on, stated by the caller instead of measured (values a gn_calibrate_certify of the same weights and
 * batch size returned earlier: a process that restarts need not repeat the calibration's exact-f32 pass). */
int gn_set_ffn_level_eps(gn_ctx* ctx, float eps2, float eps3, int level);
/* On bulk grids the block-tail kernel also computes the next block's attention input projection (one launch and one pass over the residual rows
 * less).  Every context proves that fused form against the separate launches on its own weights before using it: at the first forward call after
 * a weight (re)load both forms run on pseudo-random rows and their outputs are compared bit for bit (~60 ms, once); a difference switches the
 * fusion off for the context.  Returns 1 = checked equal, 0 = differed (fusion off), -1 = not run yet / not applicable (small contexts, other modes). */
int gn_fused_projection_status(const gn_ctx* ctx);
/* NUMA node of HIP device `device` (its PCI function's /sys/bus/pci/devices/<bus id>/numa_node), or -1 when the platform does not say.  The host side pins
 * the staging threads of a rank to that node's cores (gisnav_amd.engine.RecordStager): with one rank per GPU, eight staging pools otherwise share whatever
 * cores the scheduler picks. */
int gn_device_numa_node(int device);
/* out8: calls certified, pairs certified, pairs flagged for margin, pairs flagged for fp16 range, pairs re-run in exact f32,
 * re-run (or, in an f32 context, original) pairs that are marginal even for eps_f32, current mode, reserved. */
int gn_get_certify_stats(gn_ctx* ctx, int64_t* out8);
int gn_reset_certify_stats(gn_ctx* ctx);
/* Measure eps for THIS context's weights and precision mode on a sample batch (arguments as gn_match): the batch is matched twice -- in the
 * context's arithmetic and on the exact-f32 kernels -- and eps = max(floor_eps, safety * max |P_mode - P_f32|) over the best score and the
 * runner-up of every valid row that comes within 1 of log(filter_threshold) in either arithmetic (all rows when the threshold is 0, or when no row
 * of the sample comes that close).
 * Synchronises; sets the context's eps and returns the measured maximum and eps through the two host pointers (either may be NULL).  Call
 * it once after loading a checkpoint, on representative pairs, IN A BATCH OF THE SIZE THE REAL CALLS HAVE (the kernel family -- and with it the arithmetic
 * whose error is being measured -- follows the grid size); safety >= 1 is the stated safety factor (the Python mirror uses 4). */
int gn_calibrate_certify(gn_ctx* ctx, int B, int kpt_format,
                         const float* desc_q, const float* kpt_q, const int32_t* n_q, int stride_q,
                         const float* desc_r, const float* kpt_r, const int32_t* n_r, int stride_r,
                         float safety, float floor_eps, float* measured_host, float* eps_host, void* stream);
/* Synchronises `stream`; host_flags[b] of the most recent matcher call: 0 certified, 1 a decision inside the margin, 2 fp16 range left. */
int gn_get_uncertain(gn_ctx* ctx, int B, int32_t* host_flags, void* stream);
/* 16 hex digits over every source file and compile flag the loaded binary was built from (gisnav_amd.build.source_digest() computes the
 * same value from the tree; gisnav_amd._lib.load refuses a library whose digest differs from its tree's). */
const char* gn_source_digest(void);

/* Create a context on HIP device `device` sized for batches of up to max_batch pairs with up
 * to max_kpts keypoints per side (rounded up to a multiple of 128 internally). */
int gn_create(int device, int max_batch, int max_kpts, int precision, gn_ctx** out);
/* The same for another local-feature type.  The reference instantiates LightGlueMatcher("sift") only (pose_node.py:109-121);
 * BASELINE.json configs[4] names the SuperPoint + LightGlue path, i.e. kornia's LightGlue(features = "superpoint"):
 * input_dim 256 = descriptor_dim (no input_proj), add_scale_ori False (posenc.Wr is [32][2] on the normalised (x, y)),
 * everything else -- 9 layers, shared to_qk, match head -- as for SIFT.  With GN_FEATURE_SUPERPOINT the descriptor arguments
 * of gn_match / gn_estimate are [B][stride][256] (used as they are: L2-normalised by the extractor), the keypoint records
 * keep their format (size / angle fields ignored), and the state dict has no input_proj.* entries. */
enum gn_feature { GN_FEATURE_SIFT = 0, GN_FEATURE_SUPERPOINT = 1 };
int gn_create_ex(int device, int max_batch, int max_kpts, int precision, int feature, gn_ctx** out);
/* kornia's LightGlueMatcher.forward(..., hw1, hw2): image sizes (w, h) used by normalize_keypoints for the query / reference
 * side; a non-positive value selects the keypoint extent (max_x, max_y) of that side, which is what PoseNode gets because it
 * passes hw1 = hw2 = None (pose_node.py:285-287).  Sticky host-side state, (0, 0, 0, 0) by default. */
int gn_set_image_size(gn_ctx* ctx, float w_q, float h_q, float w_r, float h_r);
/* Re-size the context for up to max_kpts keypoints per side: synchronises the device, replaces the max_kpts-dependent workspaces and keeps every
 * weight (and its pre-split planes / fragment layouts) where it is -- the reference accepts any keypoint count (cv2.SIFT_create() is unbounded,
 * pose_node.py:122), so the mirrors grow a context instead of failing, and a grow costs a few allocations, not a weight reload.  gn_kmax
 * changes; gn_set_active_kpts is reset to the new size; every other setting stays. */
int gn_resize(gn_ctx* ctx, int max_kpts);
void gn_destroy(gn_ctx* ctx);

/* Load one tensor of the kornia LightGlue("sift") state dict from HOST
 * memory, float32, row-major [out,in].  Both spellings `transformers.{i}.self_attn.*` and the
 * checkpoint's `self_attn.{i}.*` are accepted.  token_confidence.* and confidence_thresholds
 * are accepted and ignored (dead in PoseNode's live configuration, pose_node.py:113-116). */
int gn_load_tensor(gn_ctx* ctx, const char* name, const float* host_data, const int64_t* shape, int ndim);
/* Number of required tensors still missing (0 = ready). */
int gn_missing_tensors(const gn_ctx* ctx);

/* Matcher options (defaults = PoseNode's: 9 layers, threshold 0.5). */
int gn_set_num_layers(gn_ctx* ctx, int n_layers);
int gn_set_filter_threshold(gn_ctx* ctx, float th);

/* Batched LightGlueMatcher.forward on RAW (un-normalised) SIFT descriptors.
 *   desc_q  [B][stride_q][128] f32     kpt_q [B][stride_q][6 or 4] f32     n_q [B] int32
 *   desc_r  [B][stride_r][128] f32     kpt_r [B][stride_r][6 or 4] f32     n_r [B] int32
 * Outputs (device): idx [B][kmax][2] int64 (col0 query index, col1 reference index, rows in
 * ascending query index), score [B][kmax] f32, n_match [B] int32; kmax = gn_kmax(ctx).
 * A pair with n_q < 2 or n_r < 2 yields n_match = 0 (kornia's _no_match). */
int gn_match(gn_ctx* ctx, int B, int kpt_format,
             const float* desc_q, const float* kpt_q, const int32_t* n_q, int stride_q,
             const float* desc_r, const float* kpt_r, const int32_t* n_r, int stride_r,
             int64_t* idx, float* score, int32_t* n_match, void* stream);
int gn_kmax(const gn_ctx* ctx);

/* Gather matched points and lift the reference side to 3-D with the DEM.
 *   mkp_q [B][kmax][2] f32, obj [B][kmax][3] f32 = (x_r, y_r, dem[floor(y_r)][floor(x_r)])
 *   dem   [B][H][W] u8 or NULL (z = 0). */
int gn_gather_points(gn_ctx* ctx, int B, int kpt_format,
                     const float* kpt_q, int stride_q, const float* kpt_r, int stride_r,
                     const int64_t* idx, const int32_t* n_match,
                     const uint8_t* dem, int H, int W,
                     float* mkp_q, float* obj, void* stream);

/* Batched solvePnPRansac + Rodrigues.
 *   obj [B][kstride][3] f32, img [B][kstride][2] f32, n_pts [B] int32, K9: HOST 3x3 row-major f64.
 * Outputs (device): R [B][9] f64 row-major, t [B][3] f64, n_inliers [B] int32,
 * ok [B] u8 (0 when n_pts < min_pts or RANSAC found no model with > 4 inliers).  n_pts == 4 with min_pts <= 4 takes OpenCV's
 * `npoints == 4` branch: one P3P solve (Gao) on the first three points, the fourth picks the pose, all four are inliers, no refinement. */
int gn_pnp_ransac(gn_ctx* ctx, int B, const float* obj, const float* img, const int32_t* n_pts, int kstride,
                  const double* K9_host, int iterations_count, float reproj_error_px, double confidence,
                  int min_pts, double* R, double* t, int32_t* n_inliers, uint8_t* ok, void* stream);

/* PoseNode._pose lines 246-308 for B pairs: match -> gather -> MIN_MATCHES gate -> PnP. */
int gn_estimate(gn_ctx* ctx, int B, int kpt_format,
                const float* desc_q, const float* kpt_q, const int32_t* n_q, int stride_q,
                const float* desc_r, const float* kpt_r, const int32_t* n_r, int stride_r,
                const uint8_t* dem, int H, int W, const double* K9_host, int min_matches,
                double* R, double* t, int32_t* n_match, int32_t* n_inliers, uint8_t* ok, void* stream);

/* Throughput option for back-to-back gn_estimate calls (batch serving): with overlap enabled the PnP stage of a call
 * runs on an internal stream beside the matcher of the NEXT call (its inputs are double-buffered).  R / t /
 * n_inliers / ok of a call are then complete only after gn_flush(ctx, stream) has been issued behind it (it makes
 * `stream` wait for every outstanding PnP stage); n_match is complete in stream order as before.  Off by default:
 * without it everything is complete in stream order when gn_estimate returns. */
int gn_set_overlap(gn_ctx* ctx, int enable);
/* Throughput option for back-to-back calls: split the B pairs of every gn_estimate call into n groups (1..8) that run
 * the whole path on n internal streams, forked from the caller's stream; consecutive calls pipeline inside each
 * group's stream and one group's memory-bound kernels overlap another's matrix-bound ones.  By default the caller's stream
 * is made to wait for every group before gn_estimate returns: inputs and outputs obey plain stream order, as without the
 * option.  Measured on the 32-pair bench with the round-2 kernels: +7..10 % with n = 2 (bench.py times `value` that way), slower with
 * n >= 3 (smaller launches); off (n = 1) by default. */
int gn_set_substreams(gn_ctx* ctx, int n);
/* Opt-in companion of gn_set_substreams: leave the join to gn_flush, so that the groups of consecutive calls drift out of
 * phase.  Then ALL outputs of a call (n_match included) are complete only after gn_flush(ctx, stream), and the caller must
 * keep the INPUT buffers of a call alive (and unmodified) until it has flushed -- they are read off the caller's stream.
 * A call whose B or active padded size differs from the previous unflushed call joins first, by itself. */
int gn_set_deferred_join(gn_ctx* ctx, int enable);
/* The matcher pads every image to a multiple of 128 keypoints; by default that is max_kpts of gn_create.  When the caller
 * knows an upper bound of the keypoint counts of the coming calls (the SIFT entry points return them), this sets the padded
 * size those calls run at -- attention cost falls with its square, the GEMMs linearly.  Keypoints beyond it are ignored.
 * Results do not depend on the padded size (padding is masked out exactly).  Returns the padded size now in effect
 * (min(round_up(max_kpts_per_side, 128), padded max_kpts of the context)) or a negative status.  Output strides (gn_kmax)
 * do not change.  Host-side, STICKY state: it stays in effect for every later gn_match / gn_estimate on this context until
 * it is set again -- a caller that shrinks it for one batch restores it afterwards (gn_set_active_kpts(ctx, max_kpts));
 * the Python mirrors (PoseEngine.estimate_images, PoseNode.estimate) do. */
int gn_set_active_kpts(gn_ctx* ctx, int max_kpts_per_side);
int gn_flush(gn_ctx* ctx, void* stream);

/* ---- visual-odometry path of TwistNode ---------------------------- */
/* cv2.BFMatcher(crossCheck=False).knnMatch(desc_qry, desc_ref, k=2) + Lowe ratio test for B frame pairs --
 * ros/gisnav/gisnav/core/twist_node.py:95,248-267.  Descriptors [B][stride][128] f32 (cv2.SIFT output:
 * integer-valued, for which distances are bit-identical to OpenCV's; f32 rounding of d^2 otherwise).
 *   idx  [B][kmax][2] int64 (queryIdx, trainIdx) of the matches with m.distance < ratio * n.distance, in query
 *        order; dist [B][kmax] f32 = m.distance; n_good [B] (0 when a pair has fewer than 2 train descriptors)
 *   nn_idx / nn_dist: optional [B][kmax][2] best and second-best train index (int32, -1 = none) / distance
 *        per query keypoint (the raw knnMatch result), may be NULL. */
int gn_vo_match(gn_ctx* ctx, int B, const float* desc_q, const int32_t* n_q, int stride_q,
                const float* desc_r, const int32_t* n_r, int stride_r, double ratio,
                int64_t* idx, float* dist, int32_t* n_good, int32_t* nn_idx, float* nn_dist, void* stream);

/* TwistNode._pose lines 227-289 for B frame pairs: knnMatch -> ratio test -> MIN_MATCHES gate ->
 * compute_pose(camera_info, mkp_qry, mkp_ref, zeros) (planar PnP-RANSAC + Rodrigues); outputs as gn_estimate. */
int gn_vo_estimate(gn_ctx* ctx, int B, int kpt_format,
                   const float* desc_q, const float* kpt_q, const int32_t* n_q, int stride_q,
                   const float* desc_r, const float* kpt_r, const int32_t* n_r, int stride_r,
                   const double* K9_host, double ratio, int min_matches,
                   double* R, double* t, int32_t* n_match, int32_t* n_inliers, uint8_t* ok, void* stream);

/* ---- StereoNode reference-raster preparation ----------------------- */
/* StereoNode._rotate_and_crop_center(image, angle_degrees, shape) -- ros/gisnav/gisnav/core/stereo_node.py:292-335:
 * cv2.getRotationMatrix2D((W//2, H//2), angle, 1.0) + cv2.warpAffine(image, M, (W, H)) [INTER_LINEAR, constant 0
 * border] + centre crop, on a 2-channel u8 stack [H][W][2] (device).  out_stack [crop_h][crop_w][2] (device);
 * back9_host: optional HOST 3x3 f64 = the returned matrix (rotated-and-cropped frame -> original frame). */
int gn_rotate_crop_center(gn_ctx* ctx, const uint8_t* stack, int H, int W, double angle_degrees, int crop_h, int crop_w,
                          uint8_t* out_stack, double* back9_host, void* stream);
/* stereo_node.py:229-262 fused: cv2.cvtColor(BGR2GRAY) + np.dstack((gray, dem)) + _rotate_and_crop_center.
 * bgr [H][W][3] u8, dem [H][W] u8 (device) -> out_ref, out_dem [crop_h][crop_w] u8 (device). */
int gn_stereo_reference(gn_ctx* ctx, const uint8_t* bgr, const uint8_t* dem, int H, int W, double angle_degrees,
                        int crop_h, int crop_w, uint8_t* out_ref, uint8_t* out_dem, double* back9_host, void* stream);

/* ---- post-pose georeferencing (host-side scalar code, no context) ---- */
/* _transformations.py:298-323 proj_to_affine: "+proj=affine +xoff=.. +s11=.. ..." -> 3x4 row-major [s11 s12 s13 xoff; ...]. */
int gn_proj_to_affine(const char* proj_str, double* affine12);
/* _transformations.py:326-345 wgs84_to_ecef (pyproj latlong -> geocent on the WGS 84 datum), degrees / metres. */
int gn_wgs84_to_ecef(double lon_deg, double lat_deg, double alt_m, double* xyz3);
/* pose_node.py:333-381: r [9] row-major, t [3] of compute_pose + the OrthoStereoImage CRS affine ->
 * `earth`-frame camera position (ECEF metres) and orientation quaternion (x, y, z, w) of gisnav_camera_link_optical;
 * lonlatalt3 optional.  Returns GN_OK, or 1 if the camera centre is outside the expected range of the reference
 * raster (the node returns None, pose_node.py:339-341). */
int gn_pose_to_earth(const double* R9, const double* t3, const double* affine12, int ref_h, int ref_w,
                     double* position_ecef3, double* quat_xyzw4, double* lonlatalt3);

/* ---- SIFT feature extraction ------------------------------------------ */
/* cv2.SIFT_create().detectAndCompute(gray, None) with OpenCV's defaults -- the tile extractor of PoseNode
 * (pose_node.py:122,230-232) and the frame extractor of TwistNode (twist_node.py:93,227-245).
 *   gray [H][W] u8 (device).  Outputs (device): kpt_xysa [max_kpts][4] f32 = (x, y, size, angle_deg) -- the
 *   GN_KPT_XYSA keypoint format of gn_match / gn_estimate; response [max_kpts] f32 and octave [max_kpts] int32
 *   (packed as cv2.KeyPoint.octave) may be NULL; desc [max_kpts][128] f32 (integer-valued 0..255, as cv2 emits).
 *   n_out_host: HOST int32, number of keypoints, in OpenCV's order (sorted by x, y, size desc, angle, ...).
 *   More than max_kpts distinct keypoints: the call does NOT fail -- like cv2's `nfeatures` cap (KeyPointsFilter::retainBest,
 *   pose_node.py:108) the max_kpts keypoints of largest response are kept (equal responses: the earlier one in the order
 *   above), still listed in that order (cv2 itself leaves them in nth_element order and keeps every tie of the last
 *   response); gn_sift_last_totals reports how many there were, so the caller can grow its buffers.
 * Everything -- scale space, extrema, refinement, the sort / duplicate removal, descriptors -- runs on the device in
 * stream order; the call synchronises `stream` once, at the end, to hand the keypoint count to the host. */
int gn_sift_detect_and_compute(gn_ctx* ctx, const uint8_t* gray, int H, int W, int max_kpts,
                               float* kpt_xysa, float* response, int32_t* octave, float* desc, int32_t* n_out_host, void* stream);
/* The same for B equally sized images in ONE pass (every kernel launch covers all images): gray [B][H][W]; outputs
 * [B][max_kpts][...] with image b's keypoints in rows [0, n_out_host[b]); n_out_host: HOST int32 [B].  The reference
 * extracts one image per ROS message; batching is this build's addition for the frames -> pose pipeline, whose matcher
 * (gn_estimate) already takes batches of pairs in this keypoint format. */
int gn_sift_detect_and_compute_batch(gn_ctx* ctx, const uint8_t* gray, int B, int H, int W, int max_kpts,
                                     float* kpt_xysa, float* response, int32_t* octave, float* desc, int32_t* n_out_host, void* stream);

/* Distinct keypoints each image of the LAST gn_sift_detect_and_compute(_batch) call had before the max_kpts cap (HOST int32 [B]). */
int gn_sift_last_totals(gn_ctx* ctx, int B, int32_t* totals_host);

/* ---- SuperPoint extractor (BASELINE.json configs[4] / north_star "conv backbone"; not in the reference tree) ---------- */
/* Load one tensor of the SuperPoint network from HOST memory, float32, torch layout ([out][in][kh][kw] for conv weights), under
 * transformers' SuperPointForKeypointDetection key names: encoder.conv_blocks.{0..3}.conv_{a,b}.{weight,bias},
 * keypoint_decoder.conv_score_{a,b}.*, descriptor_decoder.conv_descriptor_{a,b}.*. */
int gn_sp_load_tensor(gn_ctx* ctx, const char* name, const float* host, const int64_t* shape, int ndim);
/* SuperPoint on B grayscale images: gray01 [B][H][W] f32 in [0, 1] (device), H and W multiples of 8.  Outputs (device):
 * kpt_xysa [B][max_kpts][4] = (x, y, 1, 0) keypoint records in the GN_KPT_XYSA format of gn_match / gn_estimate (for a
 * GN_FEATURE_SUPERPOINT context), score [B][max_kpts] (may be NULL), desc [B][max_kpts][256] L2-normalised; n_out_host: HOST
 * int32 [B] keypoints per image, sorted by descending score (raster order on equal scores).  Detector settings of the
 * published model: threshold 0.005, NMS radius 4, border 4, top max_kpts (<= 2048).  The convolutions are exact f32
 * (v_mfma_f32_32x32x2_f32).  Synchronises `stream` once per pass of 4 images to read the counts. */
int gn_sp_detect_and_describe(gn_ctx* ctx, const float* gray01, int B, int H, int W, int max_kpts,
                              float* kpt_xysa, float* score, float* desc, int32_t* n_out_host, void* stream);
/* Arithmetic of the SuperPoint convolutions: GN_SP_EXACT_F32 (v_mfma_f32_32x32x2_f32: every context can), GN_SP_SPLIT_FP16 (f32 operands
 * as two fp16 terms, three products, f32 accumulate: f32-accurate, the default of f16x2 contexts, with the fp16-range guard and exact
 * retry), GN_SP_FP16 (ONE fp16 product per block, f32 accumulate: the 16-bit-operand arithmetic BASELINE.json configs[4] names
 * ("bf16"; fp16 keeps three more mantissa bits) -- NOT f32-accurate: ~99 % of the keypoints of the exact path on the test images;
 * same guard and retry).  The two fp16 modes need a context of the f16x2 precision; elsewhere the call returns GN_ERR_ARG. */
enum { GN_SP_EXACT_F32 = 0, GN_SP_SPLIT_FP16 = 1, GN_SP_FP16 = 2 };
int gn_sp_set_arithmetic(gn_ctx* ctx, int mode);

/* ---- LoFTR (BASELINE.json north_star "LoFTR / SuperPoint+LightGlue", configs[1] "LoFTR matcher ... fp32"; not in the reference tree) ---- */
/* kornia.feature.LoFTR(pretrained = "outdoor").forward({"image0", "image1"}) -> keypoints0 / keypoints1 / confidence for ONE pair of equally
 * sized grayscale images (the detector-free matcher older GISNav releases used; at this tag only the word survives:
 * docs/vitepress/docs/glossary.md:186).  f32 throughout: ResNet-FPN backbone on the exact f32 matrix instruction, 4 x (self, cross)
 * LINEAR-attention encoder layers at 1/8 resolution, dual-softmax coarse matching (temperature 0.1, threshold 0.2, border 2, mutual
 * maxima), and -- when `fine` is set -- the 5x5-window fine level that refines keypoints1 to sub-pixel positions.  Specification:
 * the published architecture as restated in oracle/loftr.py (parity UNPINNED: kornia is not importable in the build image).
 * A context is sized for one image shape (H, W multiples of 8) and at most min(max_matches, H/8 * W/8, 131072) matches per call: a pair has
 * at most one match per coarse cell of image0, so max_matches >= H/8 * W/8 returns all of them, as kornia does (the Python mirror's default);
 * a smaller number keeps the first ones in ascending cell order.  The context is separate from gn_ctx and owns its weights. */
typedef struct gn_loftr gn_loftr;
int gn_loftr_create(int device, int H, int W, int max_matches, int fine, gn_loftr** out);
void gn_loftr_destroy(gn_loftr* ctx);
const char* gn_loftr_last_error(const gn_loftr* ctx);
/* One tensor of kornia's LoFTR state dict from HOST memory (float32, torch layout): backbone.* (convolutions [out][in][k][k], BatchNorm
 * weight / bias / running_mean / running_var), loftr_coarse.layers.{0..7}.* and loftr_fine.layers.{0,1}.* ({q,k,v}_proj, merge, mlp.0,
 * mlp.2 weights; norm1 / norm2 weight + bias), fine_preprocess.{down_proj,merge_feat}.{weight,bias}.  pos_encoding.pe and
 * num_batches_tracked entries are accepted and ignored. */
int gn_loftr_load_tensor(gn_loftr* ctx, const char* name, const float* host, const int64_t* shape, int ndim);
int gn_loftr_missing_tensors(const gn_loftr* ctx);
/* image0 / image1: DEVICE f32 [H][W] in [0, 1].  Outputs (device): kpts0 / kpts1 [max_matches][2] (x, y) pixels (kpts0 on the 1/8 grid;
 * kpts1 = coarse cell + fine offset, or the coarse cell without a fine level), conf [max_matches], ij (optional, may be NULL)
 * [max_matches][2] int32 coarse cell indices (i in image0, j in image1); matches in ascending i.  *n_host (HOST): number of matches;
 * the call synchronises `stream` once to return it. */
int gn_loftr_match(gn_loftr* ctx, const float* image0, const float* image1, float* kpts0, float* kpts1, float* conf, int32_t* ij,
                   int32_t* n_host, void* stream);
/* 1 (default): the forward's ~190 dependent launches are captured once into a hipGraph (all buffers belong to the context, the match count
 * stays on the device) and replayed with one graph launch per call; 0: plain stream launches (same kernels, same results). */
int gn_loftr_set_graph(gn_loftr* ctx, int enable);
/* Arithmetic of the convolutions and linear layers: 0 (default) exact f32 (v_mfma_f32_32x32x2_f32) -- what configs[1] names; 1 split fp16: every
 * f32 operand as two fp16 terms, three fp16 MFMA products, f32 accumulation (the matcher's f16x2 scheme: f32-accurate, 5 x the matrix-pipe
 * rate).  Mode 1 guards fp16's range: a forward in which any activation left it is repeated on the exact kernels before the call returns. */
int gn_loftr_set_arithmetic(gn_loftr* ctx, int mode);
/* test hook: internal tensor -> HOST after synchronising.  Names: "x1" "x2" "x3" "x3_out" "x1_out" (NHWC, 196 channels padded to 224),
 * "tok" ([2][Lp][256] coarse features after the transformer), "sim", "crow", "ccol", "ftok".  Returns the element count or a negative status. */
int64_t gn_loftr_debug_read(gn_loftr* ctx, const char* name, void* host_out, int64_t max_bytes, void* stream);

/* ---- test / profiling hooks (not part of the drop-in surface) --------------------------- */
/* Copy an internal workspace tensor to HOST memory after synchronising `stream`.
 * Names: "desc" "cos" "sin" "x" "qkv" "ctx" "msg" "h" "md" "ls" "sim" "rowmax" "rowlog"
 * "colmax" "collog" "m0" "m1" (ints are returned bit-cast in float slots). Returns the element
 * count copied, or a negative status.  "sim" exists only after the unfused match head (developer knob 16 = 0), gn_vo_match or a
 * phase-stamp knob allocated it: the matcher itself never materialises the similarity matrix (it returns 0 elements before that). */
int64_t gn_debug_read(gn_ctx* ctx, const char* name, void* host_out, int64_t max_bytes, void* stream);
/* Stand-alone kernels for unit tests: Y[M,N] = A[M,K] W[N,K]^T + bias (f32 MFMA path). */
int gn_debug_gemm(gn_ctx* ctx, int M, int N, int K, const float* A, const float* W, const float* bias,
                  float* Y, void* stream);
/* softmax(q k^T * scale) v for [BS][n][heads*64]-strided tensors; nkv [BS] valid key counts. */
int gn_debug_attention(gn_ctx* ctx, int BS, int npad, int cross, float qscale,
                       const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                       const int32_t* nkv, float* out, int ldo, void* stream);
/* Milliseconds spent in each stage of the last gn_estimate/gn_match when timing is enabled. */
int gn_set_stage_timing(gn_ctx* ctx, int enable);
int gn_get_stage_ms(gn_ctx* ctx, float* host_ms, int max_stages);
/* HIP-event timing of every launch of the dominant kernel (the f32 MFMA GEMM) on the caller's
 * stream: enable with room for max_launches launches (0 disables), then read
 * out3 = {launches recorded, total milliseconds, total algorithmic flops (2 M N K)}. */
/* Developer knobs: select a kernel variant (which = 0: GEMM) for A/B benchmarking; run a pure
 * v_mfma_f32_32x32x2_f32 issue-rate probe (blocks x 256 threads x iters x 8 MFMAs per wave).
 * Knobs added in round 2 (value 0 restores the round-1 path unless noted): 10 block-tail fusion level, 12 k_ffn_fused ablations /
 * phase stamps, 13 out_proj folded into the tail, 14 k_ffn_fused workgroup shape (64 / 32 tokens), 15 PnP phase stamps,
 * 16 fused match head (0 = similarity GEMM + five passes), 17 / 18 k_head_fused phase stamps / ablations, 19 k_qkv projections
 * (0 = tiled GEMM, 2 = force at any batch size), 20 k_qkv phase stamps, 21 SuperPoint split-fp16 convolutions, 23 largest number of key ranges the
 * attention of a small batch is split into (default 1 = never; 4 gives -5 % latency at batch 1 but rounds the probabilities per split), 25 start the guard word of sub-batch group value - 1 raised (tests).  Knob 1 (attention) values: 4 default,
 * 43 / 44 / 45 / 46 / 48 rejected variants kept for A/B timing, 51-55 timing probes with WRONG results, 56 the exact running maximum in every key
 * tile (the path a workgroup of the default kernel falls back to).  A bench line run with any knob set records it in `debug_variant`.
 * Process-wide knobs of late round 5 (every context; the shipped value is 0 unless noted): 41 largest 128 x 128 grid the exact-f32 GEMM leaves to 64-row
 * tiles (320; 0 = never), 42 LoFTR forms of rounds 3-4 (bit 0 staging without the register prefetch, bit 1 fine level over all max_matches windows
 * with interleaved sides, bit 2 the stem with [channel][tap] weights, bit 3 multiply the zero-padding channel steps, bit 4 no branch-free MFMA stream for the 128- / 256-channel layers; bits 8.. the overhead term of the
 * rows-per-wave cost model x 100), 43 exact-f32 attention of one or two pairs (bits 0-1: 0 = k_attn_f32_ks, 1 = k_attn_f32 always, 2 = k_attn_f32_ks
 * always; bit 2 its eight-wave form), 44 exact-f32 GEMM on 64 x 64 tiles (1 = for grids of at most 128 workgroups on 64 x 128 tiles, 0 = never,
 * 2 = the four-slot-ring kernel on 64 x 128 tiles everywhere).  All of them select between forms with identical results except 43 (f32 rounding). */
int gn_debug_set_variant(gn_ctx* ctx, int which, int value);
int gn_debug_mfma_probe(gn_ctx* ctx, int blocks, int iters, void* stream);
/* LDS-DMA addressing probe (80 KB of LDS per block filled by global_load_lds, verified by ds_read):
 * pattern [8][80][256] f32, out [81] u32 = mismatching words per 1 KB piece + number of blocks run. */
int gn_debug_lds_dma_probe(gn_ctx* ctx, const float* pattern, unsigned int* out, int blocks, int spin, void* stream);
/* EPnP minimal solver on n 5-point sets: pws [n][5][3] f64, us [n][5][2] f64 (normalised image
 * coordinates), out [n][64] f64 = R(9) t(3) candidate errors(3) candidate betas(12) eigenvalues(12) rho(6) L row0(10) ok(1). */
int gn_debug_epnp(gn_ctx* ctx, int n, const double* pws, const double* us, double* out, void* stream);
/* HIP-event timing of individual launches inside gn_match / gn_estimate (on the caller's stream):
 * record up to max_launches launches (0 = off, resets the log); kernel_class 0 = projection / FFN /
 * similarity GEMMs, 1 = attention; out3 = {launches, summed ms, summed algorithmic flops}. */
int gn_set_kernel_timing(gn_ctx* ctx, int max_launches);
int gn_get_kernel_stats(gn_ctx* ctx, int kernel_class, double* out3);
/* summed ALGORITHMIC HBM bytes of the same launches (every operand / result array once: the compulsory traffic). */
int gn_get_kernel_bytes(gn_ctx* ctx, int kernel_class, double* out1);
/* Per-kernel table of the launches recorded since gn_set_kernel_timing, as a JSON array in `json` (capacity bytes):
 * [{"name", "launches", "ms", "flops", "bytes"}, ...] with rocprofv3-style kernel names; returns the length or a negative status. */
int gn_get_kernel_table(gn_ctx* ctx, char* json, int capacity);

#ifdef __cplusplus
}
#endif
#endif /* GISNAV_AMD_H */
