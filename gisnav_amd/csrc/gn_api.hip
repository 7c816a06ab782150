// C ABI of libgisnav_amd.so: context, weight loading, and the stream-ordered schedule of the
// PoseNode hot path (ros/gisnav/gisnav/core/pose_node.py:246-308, core/_shared.py:89-125).
// See include/gisnav_amd.h for the contract of every entry point.
#include "gn_common.h"

#include <cstdio>
#include <cstring>
#include <cmath>
#include <cctype>
#include <algorithm>
#include <memory>

using namespace gn;

namespace {

thread_local std::string g_err;

struct Linear { float* w = nullptr; float* b = nullptr; int out = 0, in = 0; uint16_t* wp = nullptr; float acc_scale = 1.f;
                uint16_t* wf = nullptr; int frag_order = -1;       // frag_order: -1 none, else the k order of build_weight_fragments
                uint16_t* wf2 = nullptr;
                uint16_t* wfn = nullptr; };                         // ffn.3 only: NATURAL k order (gn_skinny.hip reads GELU rows from memory, not from registers)                         // ffn.0 only: order 2 (message half permuted) for the folded out_proj   // wf: the scaled fp16 pair in MFMA fragment order (gn_ffn.hip), same scale as wp
// wp: pre-split planes, [3][out][in] bf16 (f32x3) or [2][out][in] fp16 of w / acc_scale (f16x2; acc_scale a power of two)

struct Block {       // one SelfBlock or CrossBlock
  Linear proj_in;    // self: Wqkv re-ordered to [q|k|v][head][d] (768x256); cross: [to_qk ; to_v] (512x256)
  Linear proj_out;   // out_proj / to_out
  Linear ffn0;       // 512x512
  float* ln_g = nullptr; float* ln_b = nullptr;
  Linear ffn3;       // 256x512
  // out_proj / to_out COMPOSED into ffn.0 (round 4): ffn.0([x | out_proj(ctx)]) = [W1_x | W1_m Wo] [x | ctx] + (b1 + W1_m bo) -- the message tile and
  // its GEMM (14 % of the block tail's matrix work) disappear; built at the first forward call after the tensors were (re)loaded
  uint16_t* wfc = nullptr; float* b1c = nullptr; float wfc_scale = 1.f; bool comp_dirty = true;
};

enum Stage { ST_PREP = 0, ST_PROJ, ST_ATTN, ST_FFN, ST_HEAD, ST_GATHER, ST_PNP, ST_COUNT };

}  // namespace

struct gn_ctx {
  int device = 0, max_batch = 0, npad = 0, precision = 0;
  int x_planes_only = 1;   // f16x2 mode: between layers the residual stream x exists only as hm16 pairs (developer knob 11; 0 = also f32, residual read as f32)
  int qkv_stamps = 0;      // developer knob 20: k_qkv writes s_memtime phase stamps into the sim buffer
  int qkv_fused = 1;       // attention input projections by k_qkv (gn_qkv.hip) instead of the LDS-staged GEMM (developer knob 19)
  int sp_split = 1;        // gn_sp_set_arithmetic (developer knob 21): 0 exact f32, 1 split-fp16 operands (contexts of the f16x2 mode), 2 one fp16 product
  long long sp_split_trips = 0;
  int sp_stop = 0;
  int sp_ts_layer = 0;          // developer knob 35: the layer (1..11) whose k_sp_conv_s launch writes phase stamps into sp_ts (gn_debug_read("sp_ts"))
  long long* sp_ts = nullptr;   // [8192 workgroups][32] s_memtime stamps
  void* sp_allocs_dbg = nullptr;
  bool sp_enc_hm16 = false, sp_enc_fp16 = false;   // the last pass left the encoder output (sp_y) as hm16 records (gn_debug_read("sp_enc") converts)
  int head_fused = 1;      // match head: 1 = two fused sweeps that recompute the similarity tiles (no sim buffer); 0 = sim GEMM + five passes (developer knob 16)
  int head_stamps = 0;     // developer knob 17: k_head_fused writes s_memtime phase stamps into the sim buffer
  int pnp_stamps = 0;      // developer knob 15: k_pnp_* write s_memtime phase stamps into the sim buffer
  int ffn_compose = 1;     // with ffn_fused == 3 and ffn_fold: out_proj / to_out composed into ffn.0's weights (developer knob 28; 0 = the kernel computes the message)
  int ffn_fold = 1;        // with ffn_fused == 3: out_proj / to_out folded into the block-tail kernel (developer knob 13; 0 = separate GEMM launch)
  int ffn_fused = 3;       // f16x2 mode: 3 = the whole block tail in one launch (k_ffn_fused, gn_ffn.hip); 1 = ffn.0 + LayerNorm + GELU in one launch
                           // (k_gemm_p2ln) when the grid fills the chip, 2 = always; 0 = separate k_ln_gelu (developer knob 10)
  // f16x2 domain guard (gn_set_guard): device word raised by any hm16 writer whose value does not fit fp16
  // ovf_base: [0..7] one word per sub-batch group of the matcher (groups run concurrently on their own streams and each clears / reads its own
  // word), [8] the SuperPoint extractor's; ovf = the word of the group that is being enqueued (ovf_base + g); ovf_host: pinned mirror of all 16
  unsigned int* ovf_base = nullptr; unsigned int* ovf = nullptr; unsigned int* ovf_host = nullptr;
  int dbg_trip_group = 0;  // developer knob 25: g + 1 = the matcher starts group g's guard word RAISED instead of cleared (tests of the per-group words)
  bool in_group = false;   // gn_estimate is enqueuing one of its sub-batch groups (gn_match then leaves ovf_groups_last alone)
  int ovf_groups_last = 1; // number of group words the most recent matcher call used (gn_get_guard_status ORs exactly those)
  int guard = 1;           // 0 off, 1 flag (a tripped call reports zero matches), 2 flag + synchronous re-run in the f32x3 mode
  long long guard_trips = 0;   // calls that tripped (counted when observed: guard 2, or gn_get_guard_status)
  // margin certificate (gn_set_certify): the match head keeps the runner-up of every row / column maximum and flags, per PAIR, a decision that
  // lies within cert_eps of flipping; mode 2 reads the flags back once per call (one stream synchronisation) and re-runs the flagged pairs in the
  // exact-f32 arithmetic (GN_PREC_F32's kernels on this context's f32 weights and f32 workspaces), which also takes over the fp16-range fallback
  int certify = 0;             // 0 off, 1 flags only (gn_get_uncertain), 2 flags + f32 re-run of the flagged pairs
  float cert_eps = 4.0e-3f;    // stated bound on |P_mode - P_exact| for this context's arithmetic (tools/certify_eps.py measures it)
  float cert_eps_f32 = 1.0e-4f;   // the same for the exact-f32 kernels (GPU f32 against the torch-CPU f32 oracle: summation order only)
  float* max0b = nullptr; float* rpart_c = nullptr; int32_t* uncert = nullptr; int32_t* uncert_alt = nullptr; int32_t* uncert_host = nullptr;   // uncert_host: pinned [2 * max_batch] (flags | the other level's flags)
  bool cert_inner = false;     // a certificate re-run is being enqueued (no nested certification)
  void* cert_stage = nullptr; size_t cert_stage_bytes = 0;      // staging block of the gathered re-run (certify_rerun), grown on demand
  // mode 3 (deferred): gn_estimate leaves the flags of call n in a pinned slot behind an event and resolves them -- reads them, re-runs the flagged
  // pairs from the SAVED arguments -- after call n + 1 has been enqueued (or in gn_flush), so the host never waits for an idle GPU
  struct CertPending {
    bool active = false; hipEvent_t ev = nullptr; int32_t* flags = nullptr;   // flags: pinned [max_batch]
    int B = 0, kpt_format = 0, stride_q = 0, stride_r = 0, H = 0, W = 0, min_matches = 0, npad_run = 0, level = 0;
    const float *desc_q = nullptr, *kpt_q = nullptr, *desc_r = nullptr, *kpt_r = nullptr; const int32_t *n_q = nullptr, *n_r = nullptr; const uint8_t* dem = nullptr;
    double K9[9] = {0}; double *R = nullptr, *t = nullptr; int32_t *n_match = nullptr, *n_inliers = nullptr; uint8_t* ok = nullptr;
  } cert_pend[2];
  int cert_slot = 0;
  long long cert_calls = 0, cert_pairs = 0, cert_flag_margin = 0, cert_flag_range = 0, cert_rerun = 0, cert_f32_marginal = 0;
  int attn_f16 = 0;        // GN_PREC_F16X2_F16_ATTN: q | k rows, V^T panels and the probabilities are fp16 instead of bf16 (precision itself reads F16X2_BF16_ATTN)
  int precision_api = 0;   // the gn_precision value gn_create was called with
  int feature = 0;         // GN_FEATURE_SIFT / GN_FEATURE_SUPERPOINT (gn_create_ex)
  float size_q[2] = {0.f, 0.f}, size_r[2] = {0.f, 0.f};   // gn_set_image_size: (w, h) per side for the keypoint normalisation, 0 = keypoint extent
  int npad_run = 0;        // padded keypoint count the matcher runs at (<= npad, gn_set_active_kpts); buffers are laid out for it per call
  int n_layers = kMaxLayers;
  float threshold = 0.5f;
  std::string err;
  // weights
  Linear input_proj;
  float* wr = nullptr;
  Block self_blk[kMaxLayers], cross_blk[kMaxLayers];
  Linear final_proj[kMaxLayers], matchability[kMaxLayers];
  std::map<std::string, bool> loaded;
  std::vector<std::string> required;
  // workspace
  float *desc = nullptr, *cos_t = nullptr, *sin_t = nullptr, *extent = nullptr;
  int32_t* nvalid = nullptr;
  float *x = nullptr, *qkv = nullptr, *ctx = nullptr, *msg = nullptr, *h = nullptr, *md = nullptr, *ls = nullptr;
  // f16x2 mode: fp16 plane pairs [2][Tmax][C] of the activations that feed GEMMs (written by their producers)
  uint16_t *desc_p = nullptr, *x_p = nullptr, *ctx_p = nullptr, *msg_p = nullptr, *h_p = nullptr, *md_p = nullptr;
  size_t Tmax = 0;         // token slots allocated (max_batch * 2 * npad) = plane stride in rows
  int planes_mode = 0;     // 1: activations travel as fp16 planes and GEMMs run k_gemm_p2 (default in f16x2 mode)
  float* sim = nullptr;
  uint16_t *qkb = nullptr, *vtb = nullptr;   // bf16 q|k rows and V^T panels (GN_PREC_BF16_ATTN)
  float* attn_part = nullptr; unsigned int* attn_tickets = nullptr;   // split-keys attention of small batches: <= 256 partial results, their tickets
  int attn_split = 1;      // developer knob 23: largest number of key ranges the attention of a small batch is split into (default 1 = never: the
                           // split changes the rounding of the probabilities, and results would then depend on batch size / padding / sub-streams)
  int attn_variant = 4;    // 0: k_attn_bf16 (f32 inputs, in-kernel conversion), otherwise k_attn_bf16_v5 (4; 41 / 42 = timing ablations)
  int stop_after = 0;      // developer knob: return from run_matcher after this many GEMM/attention launches
  int launch_count = 0;
  int no_planes = 0;       // developer knob: ignore the pre-split weight planes (f32x3 splits B on the fly)
  int dbg_planes = 0;      // gn_debug_gemm: pre-split W into bf16 planes first (exercises the WP path)
  uint16_t* dbg_wp = nullptr; size_t dbg_wp_n = 0;
  int dbg_out = 0;         // gn_debug_gemm variant 7: 1 = hm16 output only (into scratch), 2 = f32 + hm16 (timing experiments)
  uint16_t* dbg_yp = nullptr; size_t dbg_yp_n = 0;
  int dbg_vt_skip = 0;     // timing probe: projection epilogues skip the V^T panel stores (wrong results)
  int dbg_reuse = 0;       // gn_debug_gemm: keep the operand planes of the previous call (micro-benchmarks time the GEMM alone)
  uint16_t* dbg_ap = nullptr; size_t dbg_ap_n = 0;   // gn_debug_gemm, variant 7: A planes for k_gemm_p2
  int gemm_variant = -1;   // -1: library default (f32 MFMA, LDS-DMA); 5: f32x3 (GN_PREC_F32X3_BF16_ATTN)
  float *rowmax = nullptr, *rowlog = nullptr, *colmax = nullptr, *collog = nullptr, *max0 = nullptr;
  int32_t *m0 = nullptr, *m1 = nullptr;
  float *cpart_m = nullptr, *cpart_s = nullptr;   // fused head: column partials [B][2 npad / 64][npad]
  int32_t* cpart_i = nullptr;
  float* rot4 = nullptr;                          // [16][T] float4 rotary table for k_qkv (re-laid-out after every prep)
  float *rpart_a = nullptr, *rpart_b = nullptr;   // fused head: row partials [B][8][npad]
  unsigned int* tickets = nullptr;                // fused head: [B][2] arrival counters, zero between calls
  // pipeline scratch for gn_estimate
  int64_t* e_idx = nullptr; float* e_score = nullptr; float* e_mkp = nullptr; float* e_obj = nullptr;
  // sub-batch streams (gn_set_substreams): the pairs of one call are split into groups that run the whole path on
  // internal streams, out of phase with each other (fork / join events around them on the caller's stream)
  int defer_join = 0, sub_last_B = 0, sub_last_np = 0;   // gn_set_deferred_join; shape of the last unjoined sub-stream call
  int sub_serial = 0;      // developer knob 26: the sub-batch groups run one after the other on ONE stream (a working set the size of the Infinity Cache) instead of concurrently
  int cu_mask_mode = 0;    // developer knob 29
  int use_lists = 1;       // knob 31.  0: every tile.  1 (default): k_qkv / k_attn_pw walk the call's lists of tiles with valid tokens, and the block tail k_ffn128
                           // chooses its form from the padding the SAME group's previous call had (tile_feedback): one workgroup per tile (padding-only tiles leave
                           // at once) while >= 90 % of the tiles hold tokens, else one workgroup per CU walking the list (3 % slower per tile on a batch without
                           // padding, 20 % faster on a ragged one; identical bits).  2: always walk.  3: never walk.
  int ncu = 256;           // compute units of ctx->device (grids of the walking kernels)
  int fused_proj_status = -1;   // self-check of the projection fused behind the block tail (selfcheck_fused_projection): -1 not run / not applicable to this
                                // context, 1 bitwise equal to the separate k_qkv launches, 0 differed -> qkv_in_tail switched off for this context
  bool fused_proj_pending = true;   // run the self-check at the next forward call (set by every weight (re)load)
  int qkv_in_tail = 1;     // knob 32.  1 (default): on bulk grids the block tail k_ffn128 also computes the NEXT block's attention input projection from the rows
                           // it has just produced (k_ffn128<., ., ., 1 / 2>: no k_qkv launch, no read-back of the residual stream); 0: separate k_qkv launches
  int skinny = 1;          // knob 33.  1 (default): calls of one or two pairs run the attention input projections and the block
                           // tail as CU-split small-grid kernels (gn_skinny.hip); 0: never; 2: whenever the shapes allow; + 4: not the projections; + 8: not the tail;
                           // >> 4: kernel variant (tools/skinny_ab.py)
  // gn_set_ffn_products(0) -- the level follows the certificate: with eps calibrated for both levels (cert_eps_lvl[0 / 1] = two / three products, < 0 = not
  // calibrated) every head launch also evaluates the OTHER level's certificate on the same scores (uncert_alt); over windows of >= 64 certified pairs the
  // context counts the pairs each level flags and runs the next window on two products only when that flags at most 1 pair in 64 more than three products
  // would (a re-run in exact f32 costs ~4 block-tail passes of the pair, the two-product pass saves ~7 % of one).  Starts on three products.
  int auto_level = 3; float cert_eps_lvl[2] = {-1.f, -1.f}; long long auto_pairs = 0, auto_wide = 0, auto_narrow = 0, auto_calls_lvl[2] = {0, 0}, auto_switches = 0;
  int ffn_products = 3;    // gn_set_ffn_products: fp16 partial products of the block tail's two GEMMs on bulk grids (k_ffn128, composed form): 3 = f32-accurate split,
                           // 2 = the activations' fp16 high term only (meant to run under the margin certificate)
  int qkv_products = 2;    // knob 27: fp16 partial products of the attention input projections (2 or 3), per context
  unsigned long long* tile_feedback = nullptr;   // pinned host [8]: (all tiles << 32 | valid tiles) written by k_tile_lists of sub-batch group g's last call
  int* lists = nullptr; long long lists_stride = 0;   // work lists (launch_tile_lists), lists_stride ints per pair
  int n_sub = 1; hipStream_t sub_s[8] = {}; hipEvent_t ev_fork = nullptr; hipEvent_t ev_join[8] = {}; bool sub_pending[8] = {};
  // overlapped pose stage (gn_set_overlap): PnP of call n runs on an internal stream beside the matcher of call n+1
  int overlap = 0; unsigned long long calls = 0;
  hipStream_t s_pnp = nullptr; hipEvent_t ev_gather[2] = {nullptr, nullptr}, ev_pnp[2] = {nullptr, nullptr}; bool pnp_pending[2] = {false, false};
  float* o_mkp[2] = {nullptr, nullptr}; float* o_obj[2] = {nullptr, nullptr}; int32_t* o_nmatch[2] = {nullptr, nullptr};
  // SIFT workspace (gn_sift_detect_and_compute), sized for the last image shape seen
  int sift_h = 0, sift_w = 0; std::vector<void*> sift_allocs; SiftPyramid sift_py; float* sift_tmp = nullptr;
  float* sift_dk = nullptr; std::vector<std::vector<float>> sift_kernels; std::vector<int> sift_koff;   // [0] = initial blur, [1..5] = layer blurs
  int4* sift_cand = nullptr; int* sift_counts = nullptr; SiftKeypoint* sift_kp = nullptr;
  int sift_max_cand = 0, sift_max_kp = 0, sift_raw_cap = 0, sift_batch = 0; long long sift_kp_stride = 0;
  std::vector<int32_t> sift_totals;   // distinct keypoints found per image by the last call (before the max_kpts cap)
  // SuperPoint extractor (gn_sp_*): 12 convolutions in network order, workspace sized for the last (chunk, H, W) seen
  struct SpConv { float* wf = nullptr; float* b = nullptr; int cout = 0, cin = 0, taps = 0, cout_pad = 0; bool have_w = false, have_b = false;
                  uint16_t* wfh = nullptr; float acc_scale = 1.f; };   // wfh: fp16 pairs in fragment order (contexts of the f16x2 mode)
  SpConv sp[12];
  std::vector<void*> sp_allocs; int sp_h = 0, sp_w = 0, sp_chunk = 0, sp_cap = 0, sp_max = 0;
  float *sp_x = nullptr, *sp_y = nullptr, *sp_z = nullptr, *sp_maps[6] = {};
  int *sp_cand = nullptr, *sp_counts = nullptr, *sp_index = nullptr;
  // visual-odometry matcher workspace (gn_vo_match)
  float* vo_norm2 = nullptr; int32_t* vo_nn_idx = nullptr; float* vo_nn_dist = nullptr; uint8_t* vo_good = nullptr;
  uint8_t* mask_ws = nullptr;
  float* pts_ws = nullptr;
  gn::HypResult* hyp_ws = nullptr;
  std::vector<void*> allocs;
  std::vector<void*> ws_allocs;     // the max_kpts-dependent workspaces (alloc_workspace): replaced by gn_resize, weights stay
  // stage timing
  bool timing = false;
  hipEvent_t ev[2 * 128];
  int ev_stage[128]; int n_ev = 0; bool ev_ready = false;
  float stage_ms[ST_COUNT] = {0};
  // per-launch timing of the dominant kernel (f32 MFMA GEMM) for the roofline report
  bool ktiming = false;
  std::vector<hipEvent_t> kev;      // pairs (start, stop)
  std::vector<double> kflops;       // algorithmic flops of each recorded launch
  std::vector<double> kbytes;       // algorithmic HBM bytes of each recorded launch (operands in, results out, each once)
  std::vector<int> kclass;          // 0 = projection/FFN/similarity GEMM, 1 = attention
  std::vector<const char*> kname;   // rocprof-style kernel name of each recorded launch (gn::g_last_kernel)
  size_t kused = 0;
};

namespace {

#define GN_HIP(call)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (call);                                                                       \
    if (e_ != hipSuccess) {                                                                       \
      char buf_[512];                                                                             \
      snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      if (ctx) ctx->err = buf_;                                                                   \
      g_err = buf_;                                                                               \
      return GN_ERR_HIP;                                                                          \
    }                                                                                             \
  } while (0)

int fail(gn_ctx* ctx, int code, const std::string& msg) {
  if (ctx) ctx->err = msg;
  g_err = msg;
  return code;
}

template <typename T>
int dalloc(gn_ctx* ctx, T** p, size_t count) {
  void* q = nullptr;
  GN_HIP(hipMalloc(&q, count * sizeof(T)));
  GN_HIP(hipMemset(q, 0, count * sizeof(T)));
  ctx->allocs.push_back(q);
  *p = reinterpret_cast<T*>(q);
  return GN_OK;
}

// Every buffer whose size depends on max_kpts (the per-token / per-pair workspaces), tracked apart from the weights so that gn_resize can
// replace them without touching a weight: ctx->ws_allocs.
template <typename T_>
int ws_alloc(gn_ctx* ctx, T_** p, size_t count) {
  void* q = nullptr;
  GN_HIP(hipMalloc(&q, count * sizeof(T_)));
  GN_HIP(hipMemset(q, 0, count * sizeof(T_)));
  ctx->ws_allocs.push_back(q);
  *p = reinterpret_cast<T_*>(q);
  return GN_OK;
}
// The [B][npad][npad] similarity buffer is no longer part of the matcher (fused match head): it exists only for TwistNode's
// brute-force matcher, the unfused developer path and the phase-stamp tools, and is allocated on their first use.
int ensure_sim(gn_ctx* ctx) {
  if (ctx->sim) return GN_OK;
  return ws_alloc(ctx, &ctx->sim, (size_t)ctx->max_batch * ctx->npad * ctx->npad);
}

std::string canonical(const std::string& k) {
  for (const char* kind : {"self_attn.", "cross_attn."}) {
    const size_t L = strlen(kind);
    if (k.compare(0, L, kind) == 0) {
      const size_t dot = k.find('.', L);
      if (dot == std::string::npos) return k;
      return "transformers." + k.substr(L, dot - L) + "." + std::string(kind) + k.substr(dot + 1);
    }
  }
  return k;
}

struct StageTimer {
  gn_ctx* c; hipStream_t s; int st;
  StageTimer(gn_ctx* c_, hipStream_t s_, int st_) : c(c_), s(s_), st(st_) {
    if (c->timing && c->n_ev < 128) { hipEventRecord(c->ev[2 * c->n_ev], s); }
  }
  ~StageTimer() {
    if (c->timing && c->n_ev < 128) { hipEventRecord(c->ev[2 * c->n_ev + 1], s); c->ev_stage[c->n_ev] = st; ++c->n_ev; }
  }
};

// f16x2 planes mode: the hm16 buffer that shadows an f32 activation buffer.  hm16 rows have the byte pitch of the
// f32 rows, so an element offset into the f32 buffer (whole rows) maps to twice that many fp16 elements.
bool planes_of(gn_ctx* c, const float* p, uint16_t** planes) {
  struct { const float* f; uint16_t* q; size_t cols; } tab[] = {
    {c->desc, c->desc_p, (size_t)kInDim}, {c->x, c->x_p, (size_t)kDim}, {c->ctx, c->ctx_p, (size_t)kDim},
    {c->msg, c->msg_p, (size_t)kDim}, {c->h, c->h_p, (size_t)2 * kDim}, {c->md, c->md_p, (size_t)kDim}};
  for (auto& e : tab) {
    if (e.f && p >= e.f && p < e.f + c->Tmax * e.cols && (size_t)(p - e.f) % e.cols == 0) {
      *planes = e.q + 2 * (p - e.f);
      return true;
    }
  }
  return false;
}

void timed_gemm(gn_ctx* c, int epi, const GemmArgs& g_in, int batch, hipStream_t s) {
  if (c->gemm_variant >= 0) gn::g_gemm_variant = c->gemm_variant;
  GemmArgs g = g_in;
  if (c->no_planes) g.Wp = nullptr;
  bool p2 = false;
  if (c->planes_mode) {
    uint16_t* q = nullptr;
    p2 = planes_of(c, g.A, &q);
    if (p2) {
      g.Ap = q;
      if (g.A2) { uint16_t* q2 = nullptr; p2 = planes_of(c, g.A2, &q2) && g.lda2 == g.lda; g.A2p = q2; }
    }
    if (p2 && !g.Wp) {   // similarity GEMM: the B operand is an activation too
      uint16_t* qw = nullptr;
      p2 = planes_of(c, g.W, &qw);
      g.Wp = qw; g.acc_scale = 1.f;
    }
    if (p2) g.ovf = c->guard ? c->ovf : nullptr;
    if (p2) {   // outputs that feed later GEMMs leave in hm16; f32 is kept only where something reads it
      uint16_t* qy = nullptr;
      if (g.Y == c->x) { planes_of(c, g.Y, &qy); g.Yp = qy; g.ldyp = g.ldy; if (g.drop_f32) g.Y = nullptr; }
      else if (g.Y == c->msg || g.Y == c->md) { planes_of(c, g.Y, &qy); g.Yp = qy; g.ldyp = g.ldy; g.Y = nullptr; }
      else if (epi == EPI_LN_GELU && g.Y == c->h) { planes_of(c, g.Y, &qy); g.Yp = qy; g.ldyp = g.ldy; g.Y = nullptr; }
    }
  }
  ++c->launch_count;
  if (c->stop_after && c->launch_count > c->stop_after) return;
  const bool rec = c->ktiming && c->kused < c->kflops.size();
  if (rec) hipEventRecord(c->kev[2 * c->kused], s);
  if (p2) launch_gemm_p2(epi, g, batch, s); else launch_gemm_f32(epi, g, batch, s);
  if (rec) {
    hipEventRecord(c->kev[2 * c->kused + 1], s);
    c->kflops[c->kused] = 2.0 * g.M * (double)g.N * g.K * batch;
    {   // compulsory bytes: A and W once (4 B per element as f32 or as an hm16 pair), every output array once, residual rows and
        // rotary tables once
      const double mn = (double)g.M * g.N * batch;
      double by = 4.0 * ((double)g.M * g.K * batch + (double)g.N * g.K * (g.strideW ? batch : 1));
      if (epi == EPI_ROTARY_BF16 || epi == EPI_SCALE_BF16) by += 2.0 * mn;
      else if (epi == EPI_LN_GELU) by += 4.0 * mn;
      else if (epi != EPI_LN_GELU) by += (g.Y ? 4.0 * mn : 0.0) + (g.Yp ? 4.0 * mn : 0.0);
      if (epi == EPI_RESIDUAL) by += 4.0 * mn;
      if (epi == EPI_ROTARY || epi == EPI_ROTARY_BF16) by += 2.0 * 4.0 * (double)g.M * kFreq;
      c->kbytes[c->kused] = by;
    }
    c->kclass[c->kused] = 0;
    c->kname[c->kused] = gn::g_last_kernel;
    ++c->kused;
  }
}

void gemm(gn_ctx* c, int epi, GemmArgs& g, hipStream_t s) {
  g.strideA = g.strideW = g.strideY = 0;
  timed_gemm(c, epi, g, 1, s);
}

// (re)build the three bf16 planes of a weight matrix (w = wh + wm + wl exactly) for the f32x3 GEMM
int build_planes(gn_ctx* ctx, Linear& L) {
  if (!L.w || L.out <= 0 || L.in <= 0) return GN_OK;
  const size_t n = (size_t)L.out * L.in;
  if (!L.wp) { int rc = dalloc(ctx, &L.wp, 3 * n); if (rc != GN_OK) return rc; }
  if (ctx->precision == GN_PREC_F16X2_BF16_ATTN) {
    // power-of-two scale that puts max |w| in [2^12, 2^13): both fp16 planes stay normal for every weight within
    // 2^-15 of the largest, and there is headroom to fp16's 65504
    std::vector<float> host(n);
    GN_HIP(hipMemcpy(host.data(), L.w, n * sizeof(float), hipMemcpyDeviceToHost));
    float mx = 0.f;
    for (float v : host) { const float av = fabsf(v); if (av > mx) mx = av; }
    int e = 0;
    if (mx > 0.f && std::isfinite(mx)) { frexpf(mx, &e); e = 13 - e; }   // mx = f * 2^(13 - e'), f in [0.5, 1)
    if (e > 60) e = 60;
    if (e < -60) e = -60;
    const float scale = ldexpf(1.0f, e);
    L.acc_scale = ldexpf(1.0f, -e);
    if (ctx->planes_mode) launch_split_hm16(L.w, L.wp, L.out, L.in, scale, 0);   // hm16 rows for k_gemm_p2
    else launch_split2_f16(L.w, L.wp, (long long)n, scale, 0);                   // [2][out][in] planes for k_gemm_f16x2
    if (ctx->planes_mode && L.frag_order >= 0 && L.out % 32 == 0 && L.in % 16 == 0) {   // fragment order for the fused block tail
      std::vector<uint16_t> frag(2 * n);
      build_weight_fragments(host.data(), L.out, L.in, scale, L.frag_order, frag.data());
      if (!L.wf) { int rc = dalloc(ctx, &L.wf, 2 * n); if (rc != GN_OK) return rc; }
      GN_HIP(hipMemcpy(L.wf, frag.data(), frag.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
      if (L.frag_order == 1) {   // ffn.3: also in natural k order for the small-grid tail (gn_skinny.hip)
        build_weight_fragments(host.data(), L.out, L.in, scale, 0, frag.data());
        if (!L.wfn) { int rc = dalloc(ctx, &L.wfn, 2 * n); if (rc != GN_OK) return rc; }
        GN_HIP(hipMemcpy(L.wfn, frag.data(), frag.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
      }
      if (L.out == 2 * kDim && L.in == 2 * kDim) {   // ffn.0: also the variant whose message half is fed from registers
        build_weight_fragments(host.data(), L.out, L.in, scale, 2, frag.data());
        if (!L.wf2) { int rc = dalloc(ctx, &L.wf2, 2 * n); if (rc != GN_OK) return rc; }
        GN_HIP(hipMemcpy(L.wf2, frag.data(), frag.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
      }
    }
  } else {
    L.acc_scale = 1.f;
    launch_split3_bf16(L.w, L.wp, (long long)n, 0);
  }
  GN_HIP(hipStreamSynchronize(0));
  return GN_OK;
}

GemmArgs gemm_args(const float* A, int lda, const Linear& L, float* Y, int ldy, int M) {
  GemmArgs g;
  memset(&g, 0, sizeof g);
  g.Wp = L.wp; g.wp_plane = (long long)L.out * L.in; g.acc_scale = L.acc_scale;
  g.A = A; g.lda = lda; g.A2 = nullptr; g.lda2 = 0; g.K1 = L.in;
  g.W = L.w; g.ldw = L.in; g.bias = L.b; g.Y = Y; g.ldy = ldy; g.M = M; g.N = L.out; g.K = L.in;
  return g;
}

// k_attn_bf16_v5 on a grid that leaves CUs idle (one to three pairs): split the keys of every (slot, head, query block) over up to four workgroups
static void attn_split(gn_ctx* c, AttnArgs& a) {
  const int blocks = a.npad / 128 * kHeads * a.BS;
  int S = 1;
  while (S < 4 && S < c->attn_split && blocks * S * 2 <= (c->attn_split >= 8 ? 512 : 256) && a.npad / 64 >= 4 * S) S *= 2;   // (knob 23 >= 8: up to 256 partial results = the buffer; 4: at most 128)
  // (sub-batch streams would share the partial-result buffer)
  if (c->attn_part && c->attn_tickets && c->attn_split && c->n_sub <= 1 && S > 1) { a.nsplit = S; a.part = c->attn_part; a.tickets = c->attn_tickets; }
}

void attention(gn_ctx* c, const AttnArgs& a, hipStream_t s) {
  if (c->precision != GN_PREC_F32) launch_attention_bf16(a, s); else launch_attention_f32(a, s);
}

// one attention launch of the matcher schedule, optionally bracketed by HIP events (kernel class 1);
// flops = QK^T + PV over full npad x npad score panels (the bench fills every slot)
void timed_attention(gn_ctx* c, const AttnArgs& a, bool bf16v2, hipStream_t s) {
  ++c->launch_count;
  if (c->stop_after && c->launch_count > c->stop_after) return;
  const bool rec = c->ktiming && c->kused < c->kflops.size();
  if (rec) hipEventRecord(c->kev[2 * c->kused], s);
  if (bf16v2) launch_attention_bf16_v2(a, s); else attention(c, a, s);
  if (rec) {
    hipEventRecord(c->kev[2 * c->kused + 1], s);
    c->kflops[c->kused] = 4.0 * a.BS * kHeads * (double)a.npad * a.npad * kHeadDim;
    c->kbytes[c->kused] = (double)a.BS * a.npad * kDim * (a.qb ? 2.0 + 2.0 + 2.0 : 12.0) + (double)a.BS * a.npad * kDim * 4.0;   // q, k, v in; context rows out
    c->kclass[c->kused] = 1;
    c->kname[c->kused] = gn::g_last_kernel;
    ++c->kused;
  }
}

// q | k | v (or qk | v) projection of one block straight into the attention kernel's bf16 layouts; false when the shape / mode
// needs the general GEMM
bool qkv_projection_applies(const gn_ctx* c, const Block& blk, int T, int np, int vt_perm) {
  // (small batches keep the tiled GEMM: T / 128 workgroups would leave most of the chip idle)
  return c->planes_mode && c->qkv_fused && blk.proj_in.wf && c->x_p && c->qkb && c->vtb && c->rot4 && T % 128 == 0 && np % 128 == 0 && (vt_perm & 1) &&
         (T / 128 >= 128 || c->qkv_fused == 2);
}
// The choice depends on the number of pairs of the call only (at most two), never on the padded length: gn_set_active_kpts must not change a result
// bit (test_active_kpts_padding_does_not_change_results), and the small-grid tail rounds differently from the bulk kernels.
bool skinny_applies(const gn_ctx* c, int T, int np) {
  return (c->skinny & 3) && c->planes_mode && c->x_planes_only && T % 32 == 0 && np > 0 && ((c->skinny & 3) == 2 || T / np <= 4);
}
bool qkv_projection(gn_ctx* c, const Block& blk, bool cross, int T, int np, int vt_perm, hipStream_t s) {
  const bool skinny = skinny_applies(c, T, np) && !(c->skinny & 4) && c->qkv_fused && blk.proj_in.wf && c->x_p && c->qkb && c->vtb && c->rot4 && np % 32 == 0 && vt_perm == 1 && !c->qkv_stamps;
  if (!skinny && !qkv_projection_applies(c, blk, T, np, vt_perm)) return false;
  QkvArgs q;
  q.xp = c->x_p; q.wf = blk.proj_in.wf; q.acc_scale = blk.proj_in.acc_scale; q.bias = blk.proj_in.b;
  q.rot4 = c->rot4; q.rot_stride = (long long)c->Tmax; q.qkb = c->qkb; q.ldyb = cross ? kDim : 2 * kDim; q.vt = c->vtb; q.npad = np;
  q.qscale = 0.125f; q.scale = 0.35355339059327373f; q.vt_perm = vt_perm; q.T = T;
  q.half_fmt = c->attn_f16; q.ovf = (c->attn_f16 && c->guard) ? c->ovf : nullptr;
  q.tiles = (c->use_lists && c->lists && !c->qkv_stamps) ? c->lists : nullptr;
  q.products = c->qkv_products == 3 ? 3 : 2; q.ncu = c->ncu;
  q.dbg_ts = (c->qkv_stamps && c->sim) ? reinterpret_cast<long long*>(c->sim) : nullptr;   // developer knob 20
  ++c->launch_count;
  if (c->stop_after && c->launch_count > c->stop_after) return true;
  const bool rec = c->ktiming && c->kused < c->kflops.size();
  if (rec) hipEventRecord(c->kev[2 * c->kused], s);
  if (skinny) launch_skinny_qkv(q, cross, s); else launch_qkv(q, cross, s);
  if (rec) {
    const double N = cross ? 2.0 * kDim : 3.0 * kDim;
    hipEventRecord(c->kev[2 * c->kused + 1], s);
    c->kflops[c->kused] = 2.0 * T * N * kDim;
    c->kbytes[c->kused] = 4.0 * T * kDim + 2.0 * T * N + 4.0 * N * kDim + (cross ? 0.0 : 2.0 * 4.0 * T * kFreq);   // x in (hm16), bf16 out, weights once, rotary tables
    c->kclass[c->kused] = 0;
    c->kname[c->kused] = gn::g_last_kernel;
    ++c->kused;
  }
  return true;
}

// [W1_x | W1_m Wo] and b1 + W1_m bo in double on the host, then the power-of-two scale and the fragment order of the other weights (natural k)
int build_composed(gn_ctx* ctx, Block& blk) {
  if (!blk.comp_dirty) return GN_OK;
  const Linear &F = blk.ffn0, &P = blk.proj_out;
  if (!ctx->planes_mode || !F.w || !F.b || !P.w || !P.b || F.out != 2 * kDim || F.in != 2 * kDim || P.out != kDim || P.in != kDim) return GN_OK;
  const int H = 2 * kDim, D = kDim;
  std::vector<float> w1((size_t)H * H), b1(H), wo((size_t)D * D), bo(D);
  GN_HIP(hipMemcpy(w1.data(), F.w, w1.size() * 4, hipMemcpyDeviceToHost));
  GN_HIP(hipMemcpy(b1.data(), F.b, b1.size() * 4, hipMemcpyDeviceToHost));
  GN_HIP(hipMemcpy(wo.data(), P.w, wo.size() * 4, hipMemcpyDeviceToHost));
  GN_HIP(hipMemcpy(bo.data(), P.b, bo.size() * 4, hipMemcpyDeviceToHost));
  std::vector<float> wc((size_t)H * H), bc(H);
  std::vector<double> row(D);
  float mx = 0.f;
  for (int h = 0; h < H; ++h) {
    const float* w1h = &w1[(size_t)h * H];
    for (int k = 0; k < D; ++k) { row[k] = 0.0; wc[(size_t)h * H + k] = w1h[k]; }
    double bb = b1[h];
    for (int f = 0; f < D; ++f) {
      const double wm = w1h[D + f];
      const float* wof = &wo[(size_t)f * D];
      for (int k = 0; k < D; ++k) row[k] += wm * wof[k];
      bb += wm * bo[f];
    }
    for (int k = 0; k < D; ++k) wc[(size_t)h * H + D + k] = (float)row[k];
    bc[h] = (float)bb;
    for (int k = 0; k < H; ++k) { const float av = fabsf(wc[(size_t)h * H + k]); if (av > mx) mx = av; }
  }
  int e = 0;
  if (mx > 0.f && std::isfinite(mx)) { frexpf(mx, &e); e = 13 - e; }
  if (e > 60) e = 60;
  if (e < -60) e = -60;
  std::vector<uint16_t> frag((size_t)2 * H * H);
  build_weight_fragments(wc.data(), H, H, ldexpf(1.0f, e), 0, frag.data());
  if (!blk.wfc) { int rc = dalloc(ctx, &blk.wfc, frag.size()); if (rc != GN_OK) return rc; }
  if (!blk.b1c) { int rc = dalloc(ctx, &blk.b1c, (size_t)H); if (rc != GN_OK) return rc; }
  GN_HIP(hipMemcpy(blk.wfc, frag.data(), frag.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
  GN_HIP(hipMemcpy(blk.b1c, bc.data(), bc.size() * sizeof(float), hipMemcpyHostToDevice));
  blk.wfc_scale = ldexpf(1.0f, -e);
  blk.comp_dirty = false;
  return GN_OK;
}

int ensure_composed(gn_ctx* ctx) {
  if (!ctx->planes_mode || !ctx->ffn_compose) return GN_OK;
  for (int i = 0; i < ctx->n_layers; ++i) {
    int rc = build_composed(ctx, ctx->self_blk[i]); if (rc != GN_OK) return rc;
    rc = build_composed(ctx, ctx->cross_blk[i]); if (rc != GN_OK) return rc;
  }
  return GN_OK;
}

// true when the block-tail kernel also computes msg = out_proj(ctx): the schedule then skips the out_proj GEMM launch
bool tail_folds_out_proj(const gn_ctx* c, const Block& blk, int T) {
  return c->planes_mode && c->x_planes_only && c->ffn_fused == 3 && c->ffn_fold && blk.ffn0.wf && blk.ffn0.wf2 && blk.ffn3.wf && blk.proj_out.wf &&
         T % 64 == 0 && c->precision != GN_PREC_F32 && c->attn_variant >= 1;   // needs the attention kernel's hm16 output rows (ctx_p)
}

// The block tail's form on this call, predicted from the same sub-batch group's previous call (k_tile_lists leaves (all tiles, valid tiles) in pinned
// host memory; read without synchronisation, so the word is from some earlier call -- workloads are stationary, and both forms give the same bits).
bool tail_should_walk(const gn_ctx* c) {
  if (!c->tile_feedback) return false;
  const int g = (int)(c->ovf - c->ovf_base);
  if (g < 0 || g >= 8) return false;
  const unsigned long long w = __atomic_load_n(&c->tile_feedback[g], __ATOMIC_RELAXED);
  const unsigned all = (unsigned)(w >> 32), valid = (unsigned)w;
  return all > 0 && (unsigned long long)valid * 10u < (unsigned long long)all * 9u;
}

// x += ffn3(gelu(ln(ffn0([x | msg]))))
// next / next_cross: the block whose attention input projection follows this tail (nullptr: none) -- when the tail runs as the composed k_ffn128 and that
// projection is a k_qkv<., true, 2> launch, it is computed inside the tail instead; returns true when it was (the caller then skips the projection)
// one recorded launch of the matcher schedule (kernel class 0): stop_after bookkeeping + optional HIP events around it
template <typename F> bool timed_launch(gn_ctx* c, hipStream_t s, double flops, double bytes, F&& launch) {
  ++c->launch_count;
  if (c->stop_after && c->launch_count > c->stop_after) return false;
  const bool rec = c->ktiming && c->kused < c->kflops.size();
  if (rec) hipEventRecord(c->kev[2 * c->kused], s);
  launch();
  if (rec) {
    hipEventRecord(c->kev[2 * c->kused + 1], s);
    c->kflops[c->kused] = flops; c->kbytes[c->kused] = bytes; c->kclass[c->kused] = 0; c->kname[c->kused] = gn::g_last_kernel;
    ++c->kused;
  }
  return true;
}

inline bool ffn_auto(const gn_ctx* c) { return c->ffn_products == 0 && c->cert_eps_lvl[0] >= 0.f && c->cert_eps_lvl[1] >= 0.f && c->certify >= 2; }
inline int ffn_level(const gn_ctx* c) { return c->ffn_products == 0 ? (ffn_auto(c) ? c->auto_level : 3) : c->ffn_products; }
inline float cert_eps_now(const gn_ctx* c) { return ffn_auto(c) ? c->cert_eps_lvl[c->auto_level - 2] : c->cert_eps; }

bool ffn(gn_ctx* c, const Block& blk, int T, hipStream_t s, bool keep_f32, const Block* next = nullptr, bool next_cross = false, int np = 0, int vt_perm = 0) {
  if (skinny_applies(c, T, np) && !(c->skinny & 8) && c->ffn_fused == 3 && c->ffn_compose && tail_folds_out_proj(c, blk, T) && blk.wfc && blk.b1c && !blk.comp_dirty && blk.ffn3.wfn &&
      c->h && c->ctx_p && gn::g_ffn_ablate == 0 && gn::g_ffn_shape == 0) {
    // small grid: the weight stream split across CUs -- two launches (gn_skinny.hip)
    SkinnyTailArgs a;
    a.xp = c->x_p; a.cp = c->ctx_p; a.w1 = blk.wfc; a.w1_scale = blk.wfc_scale; a.b1 = blk.b1c; a.h = c->h; a.ln_g = blk.ln_g; a.ln_b = blk.ln_b;
    a.w2 = blk.ffn3.wfn; a.w2_scale = blk.ffn3.acc_scale; a.b2 = blk.ffn3.b; a.xp_out = c->x_p; a.y = keep_f32 ? c->x : nullptr;
    a.ovf = c->guard ? c->ovf : nullptr; a.T = T;
    if (!timed_launch(c, s, 2.0 * T * 512.0 * 512.0, 4.0 * T * (256.0 + 256.0 + 512.0) + 4.0 * 512.0 * 512.0, [&] { launch_skinny_h(a, s, c->skinny >> 4); })) return false;
    timed_launch(c, s, 2.0 * T * 256.0 * 512.0, 4.0 * T * (512.0 + 256.0 + 256.0) + 4.0 * 256.0 * 512.0, [&] { launch_skinny_out(a, s, c->skinny >> 4); });
    return false;
  }
  if (c->planes_mode && c->x_planes_only && c->ffn_fused == 3 && blk.ffn0.wf && blk.ffn3.wf && T % 64 == 0) {   // the whole tail in one launch
    FfnArgs f;
    const bool fold = tail_folds_out_proj(c, blk, T);                     // out_proj computed inside the kernel from the attention output
    f.cp = fold ? c->ctx_p : nullptr; f.wos = blk.proj_out.wf; f.wo_scale = blk.proj_out.acc_scale; f.bo = blk.proj_out.b;
    f.xp = c->x_p; f.mp = c->msg_p; f.w1s = fold ? blk.ffn0.wf2 : blk.ffn0.wf; f.w1_scale = blk.ffn0.acc_scale; f.b1 = blk.ffn0.b; f.ln_g = blk.ln_g; f.ln_b = blk.ln_b;
    f.w2s = blk.ffn3.wf; f.w2_scale = blk.ffn3.acc_scale; f.b2 = blk.ffn3.b; f.yp = c->x_p; f.y = keep_f32 ? c->x : nullptr; f.T = T;
    const bool comp = fold && c->ffn_compose && blk.wfc && blk.b1c && !blk.comp_dirty;
    if (comp) { f.composed = 1; f.w1s = blk.wfc; f.w1_scale = blk.wfc_scale; f.b1 = blk.b1c; }
    f.ovf = c->guard ? c->ovf : nullptr;
    f.tiles = (c->use_lists && c->lists && !(gn::g_ffn_ablate & 8)) ? c->lists : nullptr;
    f.walk = c->use_lists == 2 || (c->use_lists == 1 && tail_should_walk(c));
    f.ncu = c->ncu;
    f.composed = comp ? 1 : 0;
    f.products = (comp && ffn_level(c) == 2) ? 2 : 3;
    const bool fuse_qkv = next != nullptr && c->qkv_in_tail && comp && ffn_selects_128(f) && gn::g_ffn_ablate == 0 && c->attn_f16 && c->qkv_products != 3 && !c->qkv_stamps &&
                          c->precision != GN_PREC_F32 && c->attn_variant >= 1 && qkv_projection_applies(c, *next, T, np, vt_perm) && !(vt_perm & 2);
    if (fuse_qkv) {
      f.qkv = next_cross ? 2 : 1;
      f.q_wf = next->proj_in.wf; f.q_acc_scale = next->proj_in.acc_scale; f.q_bias = next->proj_in.b;
      f.q_rot4 = c->rot4; f.q_rot_stride = (long long)c->Tmax; f.q_qkb = c->qkb; f.q_vt = c->vtb;
      f.q_qscale = 0.125f; f.q_scale = 0.35355339059327373f;
      f.npad = np;
    }
    if (c->use_lists) { f.nvalid = c->nvalid; f.npad = c->npad_run; }
    f.dbg_ts = (gn::g_ffn_ablate & 8) ? reinterpret_cast<long long*>(c->sim) : nullptr;   // developer: phase stamps land in the (idle) sim buffer
    ++c->launch_count;
    if (c->stop_after && c->launch_count > c->stop_after) return false;
    const bool rec = c->ktiming && c->kused < c->kflops.size();
    if (rec) hipEventRecord(c->kev[2 * c->kused], s);
    launch_ffn_fused(f, s);
    if (rec) {
      hipEventRecord(c->kev[2 * c->kused + 1], s);
      c->kflops[c->kused] = 2.0 * T * (512.0 * 512.0 + 256.0 * 512.0 + ((fold && !comp) ? 256.0 * 256.0 : 0.0));   // composed: the flops the kernel's own formulation needs (algorithmic: whatever the number of partial products)
      c->kbytes[c->kused] = 4.0 * T * (256.0 + 256.0 + 256.0 + 256.0) + 4.0 * (512.0 * 512.0 + 256.0 * 512.0);   // x, msg, residual rows in; x out; weights once
      if (fuse_qkv) {     // + the projection (k_qkv's figures without its read of the rows)
        const double N = next_cross ? 2.0 * kDim : 3.0 * kDim;
        c->kflops[c->kused] += 2.0 * T * N * kDim;
        c->kbytes[c->kused] += 2.0 * T * N + 4.0 * N * kDim + (next_cross ? 0.0 : 2.0 * 4.0 * T * kFreq);
      }
      c->kclass[c->kused] = 0;
      c->kname[c->kused] = gn::g_last_kernel;
      ++c->kused;
    }
    return fuse_qkv;
  }
  GemmArgs g = gemm_args(c->x, kDim, blk.ffn0, c->h, 2 * kDim, T);
  g.A2 = c->msg; g.lda2 = kDim; g.K1 = kDim;
  // LayerNorm + GELU in the GEMM's epilogue (the hidden tensor leaves once, as hm16) when there are enough 128-row tiles to fill the chip:
  // measured 9.64 vs 9.86 ms at 512 tiles (32 pairs), 5.68 vs 5.90 at 256, but 3.95 vs 3.73 ms at 128 tiles and 3.00 vs 2.40 ms at 16 (knob 10: 2 = always)
  if (c->planes_mode && T % 128 == 0 && (c->ffn_fused == 2 || (c->ffn_fused == 1 && T / 128 >= 256))) {
    g.ln_g = blk.ln_g; g.ln_b = blk.ln_b;
    gemm(c, EPI_LN_GELU, g, s);
  } else {
    gemm(c, EPI_BIAS, g, s);
    launch_ln_gelu(c->h, blk.ln_g, blk.ln_b, T, s, c->planes_mode ? c->h_p : nullptr, c->planes_mode && c->guard ? c->ovf : nullptr);
  }
  GemmArgs g3 = gemm_args(c->h, 2 * kDim, blk.ffn3, c->x, kDim, T);
  g3.resid = c->x; g3.ldr = kDim;
  if (c->planes_mode && c->x_planes_only) {   // the residual stream between layers lives in hm16 only (f32 x is written by the last block, for the head)
    uint16_t* xp = nullptr;
    if (planes_of(c, c->x, &xp)) { g3.residp = xp; g3.ldrp = kDim; g3.resid = nullptr; g3.drop_f32 = keep_f32 ? 0 : 1; }
  }
  gemm(c, EPI_RESIDUAL, g3, s);
  return false;
}

int run_matcher(gn_ctx* c, int B, int kpt_format,
                const float* desc_q, const float* kpt_q, const int32_t* n_q, int stride_q,
                const float* desc_r, const float* kpt_r, const int32_t* n_r, int stride_r,
                int64_t* idx, float* score, int32_t* n_match, hipStream_t s) {
  const int np = c->npad_run, T = B * 2 * np, BS = B * 2;
  const bool bf16v2 = c->precision != GN_PREC_F32 && c->attn_variant >= 1;
  gn::g_attn_variant = c->attn_variant;
  gn::g_attn_stamps = ((c->attn_variant == 73 || c->attn_variant >= 1000) && c->sim) ? reinterpret_cast<long long*>(c->sim) : nullptr;
  const int vt_perm = (bf16v2 ? 1 : 0) | (c->dbg_vt_skip ? 2 : 0);   // k_attn_bf16_v5 reads V^T with keys permuted inside 16-groups   // k_attn_bf16_v4 reads permuted V^T
  const bool attn_planes = c->planes_mode && bf16v2;   // k_attn_bf16_v5 writes the hm16 rows itself
  c->launch_count = 0;
  if (c->planes_mode && c->guard) hipMemsetAsync(c->ovf, (c->dbg_trip_group && c->ovf == c->ovf_base + (c->dbg_trip_group - 1)) ? 1 : 0, sizeof(unsigned int), s);
  {
    StageTimer tm(c, s, ST_PREP);
    PrepArgs p;
    p.desc_q = desc_q; p.kpt_q = kpt_q; p.n_q = n_q; p.stride_q = stride_q;
    p.desc_r = desc_r; p.kpt_r = kpt_r; p.n_r = n_r; p.stride_r = stride_r;
    p.kpt_format = kpt_format; p.B = B; p.npad = np; p.wr = c->wr;
    p.desc = c->desc; p.kxy = nullptr; p.cos_t = c->cos_t; p.sin_t = c->sin_t; p.nvalid = c->nvalid; p.extent = c->extent;
    p.size_q[0] = c->size_q[0]; p.size_q[1] = c->size_q[1]; p.size_r[0] = c->size_r[0]; p.size_r[1] = c->size_r[1];
    p.feature = c->feature; p.x = nullptr; p.xp = nullptr;
    const bool planes_only = c->planes_mode && c->x_planes_only && c->n_layers > 0;
    if (c->feature == 1) {   // 256-d descriptors ARE the initial residual stream (no input_proj)
      p.x = planes_only ? nullptr : c->x; p.xp = c->planes_mode ? c->x_p : nullptr;
      launch_prep(p, s);
      if (c->rot4 && c->qkv_fused) launch_rot_table(c->cos_t, c->sin_t, c->rot4, T, (long long)c->Tmax, s);
    } else {
      launch_prep(p, s);
      if (c->rot4 && c->qkv_fused) launch_rot_table(c->cos_t, c->sin_t, c->rot4, T, (long long)c->Tmax, s);
      if (c->planes_mode) launch_split_hm16(c->desc, c->desc_p, T, kInDim, 1.0f, s);
      GemmArgs g = gemm_args(c->desc, kInDim, c->input_proj, c->x, kDim, T);
      g.drop_f32 = planes_only ? 1 : 0;
      gemm(c, EPI_BIAS, g, s);
    }
  }
  // the call's lists of tiles / query blocks that hold valid tokens (k_prep has written the per-slot counts)
  if (c->use_lists && c->lists) {
    const int g = (int)(c->ovf - c->ovf_base);
    launch_tile_lists(c->nvalid, BS, np, c->lists, (c->tile_feedback && g >= 0 && g < 8) ? c->tile_feedback + g : nullptr, s);
  }
  bool qkv_done = false;      // the previous tail has already computed this block's attention input projection
  for (int i = 0; i < c->n_layers; ++i) {
    {  // SelfBlock on both sides at once
      const Block& blk = c->self_blk[i];
      if (!qkv_done) {
        StageTimer tm(c, s, ST_PROJ);
        GemmArgs g = gemm_args(c->x, kDim, blk.proj_in, c->qkv, 3 * kDim, T);
        g.cos_t = c->cos_t; g.sin_t = c->sin_t; g.rot_cols = 2 * kDim;
        if (bf16v2 && qkv_projection(c, blk, false, T, np, vt_perm, s)) {
        } else if (bf16v2) {
          g.Yb = c->qkb; g.ldyb = 2 * kDim; g.Vt = c->vtb; g.vt_start = 2 * kDim; g.q_cols = kDim; g.qscale = 0.125f; g.npad = np; g.vt_perm = vt_perm; g.half_fmt = c->attn_f16;
          gemm(c, EPI_ROTARY_BF16, g, s);
        } else {
          gemm(c, EPI_ROTARY, g, s);
        }
      }
      {
        StageTimer tm(c, s, ST_ATTN);
        AttnArgs a; a.ovf = nullptr;
        a.q = c->qkv; a.ldq = 3 * kDim; a.k = c->qkv + kDim; a.ldk = 3 * kDim; a.v = c->qkv + 2 * kDim; a.ldv = 3 * kDim;
        a.out = c->ctx; a.ldo = kDim; a.nvalid = c->nvalid; a.npad = np; a.cross = 0; a.qscale = 0.125f; a.BS = BS;
        a.outp = attn_planes ? c->ctx_p : nullptr; a.ovf = (attn_planes && c->guard) ? c->ovf : nullptr;
        a.qb = c->qkb; a.ldqb = 2 * kDim; a.kb = c->qkb + kDim; a.ldkb = 2 * kDim; a.vt = c->vtb; a.half_fmt = c->attn_f16; a.tiles = (c->use_lists && c->lists) ? c->lists : nullptr; a.ncu = c->ncu;
        if (bf16v2) attn_split(c, a);
        timed_attention(c, a, bf16v2, s);
        if (c->planes_mode && !attn_planes) launch_split_hm16(c->ctx, c->ctx_p, T, kDim, 1.0f, s);
      }
      if (!tail_folds_out_proj(c, blk, T)) {
        StageTimer tm(c, s, ST_PROJ);
        GemmArgs g = gemm_args(c->ctx, kDim, blk.proj_out, c->msg, kDim, T);
        gemm(c, EPI_BIAS, g, s);
      }
      { StageTimer tm(c, s, ST_FFN); qkv_done = ffn(c, blk, T, s, false, bf16v2 ? &c->cross_blk[i] : nullptr, true, np, vt_perm); }
    }
    {  // CrossBlock
      const Block& blk = c->cross_blk[i];
      if (!qkv_done) {
        StageTimer tm(c, s, ST_PROJ);
        GemmArgs g = gemm_args(c->x, kDim, blk.proj_in, c->qkv, 2 * kDim, T);
        g.scale = 0.35355339059327373f;  // (dim_head ** -0.5) ** 0.5 applied to both qk sides
        g.scale_cols = kDim;
        if (bf16v2 && qkv_projection(c, blk, true, T, np, vt_perm, s)) {
        } else if (bf16v2) {
          g.Yb = c->qkb; g.ldyb = kDim; g.Vt = c->vtb; g.vt_start = kDim; g.q_cols = 0; g.qscale = 1.0f; g.npad = np; g.vt_perm = vt_perm; g.half_fmt = c->attn_f16;
          gemm(c, EPI_SCALE_BF16, g, s);
        } else {
          gemm(c, EPI_SCALE_COLS, g, s);
        }
      }
      {
        StageTimer tm(c, s, ST_ATTN);
        AttnArgs a; a.ovf = nullptr;
        a.q = c->qkv; a.ldq = 2 * kDim; a.k = c->qkv; a.ldk = 2 * kDim; a.v = c->qkv + kDim; a.ldv = 2 * kDim;
        a.out = c->ctx; a.ldo = kDim; a.nvalid = c->nvalid; a.npad = np; a.cross = 1; a.qscale = 1.0f; a.BS = BS;
        a.outp = attn_planes ? c->ctx_p : nullptr; a.ovf = (attn_planes && c->guard) ? c->ovf : nullptr;
        a.qb = c->qkb; a.ldqb = kDim; a.kb = c->qkb; a.ldkb = kDim; a.vt = c->vtb; a.half_fmt = c->attn_f16; a.tiles = (c->use_lists && c->lists) ? c->lists : nullptr; a.ncu = c->ncu;
        if (bf16v2) attn_split(c, a);
        timed_attention(c, a, bf16v2, s);
        if (c->planes_mode && !attn_planes) launch_split_hm16(c->ctx, c->ctx_p, T, kDim, 1.0f, s);
      }
      if (!tail_folds_out_proj(c, blk, T)) {
        StageTimer tm(c, s, ST_PROJ);
        GemmArgs g = gemm_args(c->ctx, kDim, blk.proj_out, c->msg, kDim, T);
        gemm(c, EPI_BIAS, g, s);
      }
      { StageTimer tm(c, s, ST_FFN); qkv_done = ffn(c, blk, T, s, i == c->n_layers - 1, (bf16v2 && i + 1 < c->n_layers) ? &c->self_blk[i + 1] : nullptr, false, np, vt_perm); }   // the last block leaves f32 x for the match head
    }
  }
  {
    StageTimer tm(c, s, ST_HEAD);
    const int li = c->n_layers - 1;
    GemmArgs g = gemm_args(c->x, kDim, c->final_proj[li], c->md, kDim, T);
    g.scale = 0.25f; g.scale_cols = kDim;  // / d ** 0.25
    gemm(c, EPI_SCALE_COLS, g, s);
    launch_matchability(c->x, c->matchability[li].w, c->matchability[li].b, c->ls, T, s);
    HeadArgs hd;
    hd.sim = c->sim; hd.ls = c->ls; hd.nvalid = c->nvalid; hd.B = B; hd.npad = np; hd.threshold = c->threshold;
    hd.rowmax = c->rowmax; hd.rowlog = c->rowlog; hd.colmax = c->colmax; hd.collog = c->collog;
    hd.m0 = c->m0; hd.max0 = c->max0; hd.m1 = c->m1;
    hd.ovf = (c->planes_mode && c->guard) ? c->ovf : nullptr;
    hd.max0b = c->max0b; hd.rpart_c = c->rpart_c; hd.uncert = c->certify ? c->uncert : nullptr;
    hd.cert_eps = (c->precision == GN_PREC_F32) ? c->cert_eps_f32 : cert_eps_now(c);
    if (hd.uncert && c->precision != GN_PREC_F32 && ffn_auto(c)) { hd.uncert_alt = c->uncert_alt; hd.cert_eps_alt = c->cert_eps_lvl[3 - c->auto_level]; }
    hd.idx = idx; hd.score = score; hd.n_match = n_match; hd.kmax = c->npad;   // output stride: gn_kmax(), whatever the active size
    hd.md = c->planes_mode ? (const void*)c->md_p : (const void*)c->md; hd.md_f32 = c->planes_mode ? 0 : 1;
    hd.cpart_m = c->cpart_m; hd.cpart_s = c->cpart_s; hd.cpart_i = c->cpart_i; hd.rpart_a = c->rpart_a; hd.rpart_b = c->rpart_b; hd.tickets = c->tickets;
    hd.dbg_ts = (c->head_stamps && c->sim) ? reinterpret_cast<long long*>(c->sim) : nullptr;   // developer knob 17
    if (c->head_fused || !c->sim) {
      ++c->launch_count;
      if (c->stop_after && c->launch_count > c->stop_after) return GN_OK;
      const bool rec = c->ktiming && c->kused < c->kflops.size();
      if (rec) hipEventRecord(c->kev[2 * c->kused], s);
      launch_match_head_fused(hd, s);
      if (rec) {   // both sweeps as one entry
        hipEventRecord(c->kev[2 * c->kused + 1], s);
        c->kflops[c->kused] = 2.0 * 2.0 * B * (double)np * np * kDim;                                   // the similarity tiles are computed twice
        c->kbytes[c->kused] = 2.0 * (4.0 * T * kDim) + 4.0 * T * 4.0 + 8.0 * B * np * 3.0;               // descriptors once per sweep, per-row / per-column statistics, matches
        c->kclass[c->kused] = 0;
        c->kname[c->kused] = "k_head_fused (2 sweeps)";
        ++c->kused;
      }
      return GN_OK;
    }
    GemmArgs gs;
    memset(&gs, 0, sizeof gs);
    gs.A = c->md; gs.lda = kDim; gs.K1 = kDim; gs.W = c->md + (size_t)np * kDim; gs.ldw = kDim;
    gs.Y = c->sim; gs.ldy = np; gs.M = np; gs.N = np; gs.K = kDim; gs.acc_scale = 1.f;
    gs.strideA = gs.strideW = 2LL * np * kDim; gs.strideY = (long long)np * np;
    timed_gemm(c, EPI_PLAIN, gs, B, s);
    launch_match_head(hd, s);
  }
  return GN_OK;
}

int alloc_workspace(gn_ctx* ctx, int max_kpts) {
  ctx->npad = ((max_kpts + 127) / 128) * 128;
  ctx->npad_run = ctx->npad;
  const size_t np = ctx->npad, T = (size_t)ctx->max_batch * 2 * np, B = ctx->max_batch;
#define GN_ALLOC(field, count)                                     \
  do { int rc_ = ws_alloc(ctx, &ctx->field, (count)); if (rc_ != GN_OK) return rc_; } while (0)
  GN_ALLOC(desc, T * kInDim); GN_ALLOC(cos_t, T * kFreq); GN_ALLOC(sin_t, T * kFreq);
  GN_ALLOC(extent, B * 4); GN_ALLOC(nvalid, B * 2);
  GN_ALLOC(x, T * kDim); GN_ALLOC(qkv, T * 3 * kDim); GN_ALLOC(ctx, T * kDim); GN_ALLOC(msg, T * kDim);
  GN_ALLOC(h, T * 2 * kDim); GN_ALLOC(md, T * kDim); GN_ALLOC(ls, T);
  ctx->Tmax = T;
  if (ctx->precision == GN_PREC_F16X2_BF16_ATTN) {
    ctx->planes_mode = 1;
    GN_ALLOC(desc_p, 2 * T * kInDim); GN_ALLOC(x_p, 2 * T * kDim); GN_ALLOC(ctx_p, 2 * T * kDim);
    GN_ALLOC(msg_p, 2 * T * kDim); GN_ALLOC(h_p, 2 * T * 2 * kDim); GN_ALLOC(md_p, 2 * T * kDim);
    GN_ALLOC(rot4, T * 2 * kFreq);
  }
  if (ctx->precision != GN_PREC_F32) { GN_ALLOC(qkb, T * 2 * kDim); GN_ALLOC(vtb, T * kDim); GN_ALLOC(attn_part, (size_t)256 * 4 * 34 * 64); GN_ALLOC(attn_tickets, 256); }
  GN_ALLOC(rowmax, B * np); GN_ALLOC(rowlog, B * np); GN_ALLOC(colmax, B * np); GN_ALLOC(collog, B * np);
  GN_ALLOC(max0, B * np); GN_ALLOC(m0, B * np); GN_ALLOC(m1, B * np); GN_ALLOC(max0b, B * np); GN_ALLOC(rpart_c, B * 8 * np); GN_ALLOC(uncert, B); GN_ALLOC(uncert_alt, B);
  GN_ALLOC(cpart_m, B * (np / 32) * np); GN_ALLOC(cpart_s, B * (np / 32) * np); GN_ALLOC(cpart_i, B * (np / 32) * np); GN_ALLOC(rpart_a, B * 8 * np); GN_ALLOC(rpart_b, B * 8 * np); GN_ALLOC(tickets, B * 2);
  GN_ALLOC(e_idx, B * np * 2); GN_ALLOC(e_score, B * np); GN_ALLOC(e_mkp, B * np * 2); GN_ALLOC(e_obj, B * np * 3);
  GN_ALLOC(vo_norm2, T); GN_ALLOC(vo_nn_idx, B * np * 2); GN_ALLOC(vo_nn_dist, B * np * 2); GN_ALLOC(vo_good, B * np);
  GN_ALLOC(mask_ws, B * np * 16);
  GN_ALLOC(pts_ws, B * np * 5);
  GN_ALLOC(hyp_ws, B * 16);
  GN_ALLOC(ovf_base, 16);
  ctx->ovf = ctx->ovf_base;
  ctx->lists_stride = gn::kTileListBase + 2 * (np / 128) + 8 * ((np + 255) / 256);
  GN_ALLOC(lists, B * (size_t)ctx->lists_stride);
#undef GN_ALLOC
  return GN_OK;
}

void shift_workspaces(gn_ctx* c, long long b0, int sign);

// words of two device arrays that differ (16-byte granules; one atomic per wave that saw a difference)
__global__ void k_count_diff(const uint4* a, const uint4* b, size_t n, unsigned int* cnt) {
  bool d = false;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 x = a[i], y = b[i];
    d |= (x.x != y.x) | (x.y != y.y) | (x.z != y.z) | (x.w != y.w);
  }
  const unsigned long long bal = __ballot(d);
  if ((threadIdx.x & 63) == 0 && bal) atomicAdd(cnt, (unsigned int)__popcll(bal));
}

// the NEXT block's attention input projection fused behind k_ffn128 (knob 32, on by default) once miscomputed in a way that moved
// with unrelated edits (DESIGN_HISTORY; the epilogue is pinned scalar arithmetic since) -- so every context PROVES the fused form against the separate
// k_qkv launches on ITS weights before it is used: the first forward call after a weight (re)load runs, on pseudo-random token rows in the context's own
// workspaces, the block tail with the projection fused (self and cross form, one-tile and walking form) and tail + k_qkv separately, and compares the
// q | k rows and V^T panels bit for bit.  Any difference switches the fusion off for this context (gn_fused_projection_status).  ~60 ms, once.
int selfcheck_fused_projection(gn_ctx* c) {
  gn_ctx* ctx = c;
  c->fused_proj_pending = false;
  c->fused_proj_status = -1;
  const int np = c->npad;
  GN_HIP(hipDeviceSynchronize());        // (nothing of an earlier call may still be using the workspaces)
  if (!c->planes_mode || !c->qkv_in_tail || !c->attn_f16 || c->qkv_products == 3 || !c->ffn_compose || c->ffn_fused != 3 || !c->x_planes_only || c->n_layers < 1 ||
      !c->qkv_fused || !c->rot4 || !c->lists || !c->msg_p || !c->h_p || c->feature < 0 || gn::g_ffn_ablate != 0 || gn::g_ffn_shape != 0) return GN_OK;
  int B = (256 * 128 + 2 * np - 1) / (2 * np);          // the smallest batch whose grid selects k_ffn128 (>= 256 tiles of 128 tokens)
  if (B > c->max_batch) return GN_OK;                   // this context never runs the bulk kernels
  const int T = B * 2 * np;
  std::vector<float> h((size_t)T * kDim);
  unsigned long long st = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (float)((st >> 40) & 0xFFFF) * (1.0f / 32768.0f) - 1.0f; };   // U(-1, 1), 16 bits
  for (int pass = 0; pass < 2; ++pass) {
    for (auto& v : h) v = 1.5f * rnd();
    float* dst = pass == 0 ? c->x : c->ctx;
    GN_HIP(hipMemcpy(dst, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    launch_split_hm16(dst, pass == 0 ? c->x_p : c->ctx_p, T, kDim, 1.0f, 0);
  }
  {
    std::vector<float> cs((size_t)T * kFreq), sn((size_t)T * kFreq);
    for (size_t i = 0; i < cs.size(); ++i) { const float th = 3.14159265f * rnd(); cs[i] = cosf(th); sn[i] = sinf(th); }
    GN_HIP(hipMemcpy(c->cos_t, cs.data(), cs.size() * sizeof(float), hipMemcpyHostToDevice));
    GN_HIP(hipMemcpy(c->sin_t, sn.data(), sn.size() * sizeof(float), hipMemcpyHostToDevice));
    launch_rot_table(c->cos_t, c->sin_t, c->rot4, T, (long long)c->Tmax, 0);
    std::vector<int32_t> nv(2 * (size_t)B, np);
    GN_HIP(hipMemcpy(c->nvalid, nv.data(), nv.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    launch_tile_lists(c->nvalid, 2 * B, np, c->lists, nullptr, 0);
  }
  const size_t xbytes = (size_t)T * kDim * 4, qbytes = (size_t)T * 2 * kDim * 2, vbytes = (size_t)T * kDim * 2;
  uint16_t* const keep_q = c->h_p;                                      // (k_ffn128 never writes the hidden tensor to memory: h_p is free scratch, 2 KB per token)
  uint16_t* const keep_v = c->h_p + (size_t)T * 2 * kDim;
  GN_HIP(hipMemcpy(c->msg_p, c->x_p, xbytes, hipMemcpyDeviceToDevice));  // the rows every run starts from (the tail updates x_p in place)
  unsigned int* cnt = c->ovf_base + 12;                                 // (words 9..15 of the guard block are unused)
  GN_HIP(hipMemset(cnt, 0, sizeof(unsigned int)));
  const int lists0 = c->use_lists, fused0 = c->qkv_in_tail, guard0 = c->guard; const bool kt0 = c->ktiming; const int lc0 = c->launch_count, sa0 = c->stop_after;
  c->ktiming = false; c->stop_after = 0; c->guard = 0;
  bool applicable = true;
  const int nblk = c->n_layers > 1 ? 1 : 0;
  for (int cross = 0; cross < 2 && applicable; ++cross)
    for (int walk = 0; walk < 2 && applicable; ++walk) {
      const Block& tail = cross ? c->self_blk[0] : c->cross_blk[0];     // (any tail will do; the projection is the OTHER kind's)
      const Block* next = cross ? &c->cross_blk[0] : &c->self_blk[nblk];
      c->use_lists = walk ? 2 : 3;
      GN_HIP(hipMemcpyAsync(c->x_p, c->msg_p, xbytes, hipMemcpyDeviceToDevice, 0));
      GN_HIP(hipMemsetAsync(c->qkb, 0, qbytes, 0)); GN_HIP(hipMemsetAsync(c->vtb, 0, vbytes, 0));
      c->qkv_in_tail = 1;
      if (!ffn(c, tail, T, 0, false, next, cross != 0, np, 1)) { applicable = false; break; }
      GN_HIP(hipMemcpyAsync(keep_q, c->qkb, qbytes, hipMemcpyDeviceToDevice, 0));
      GN_HIP(hipMemcpyAsync(keep_v, c->vtb, vbytes, hipMemcpyDeviceToDevice, 0));
      GN_HIP(hipMemcpyAsync(c->x_p, c->msg_p, xbytes, hipMemcpyDeviceToDevice, 0));
      GN_HIP(hipMemsetAsync(c->qkb, 0, qbytes, 0)); GN_HIP(hipMemsetAsync(c->vtb, 0, vbytes, 0));
      c->qkv_in_tail = 0;
      ffn(c, tail, T, 0, false, next, cross != 0, np, 1);
      if (!qkv_projection(c, *next, cross != 0, T, np, 1, 0)) { applicable = false; break; }
      hipLaunchKernelGGL(k_count_diff, dim3(1024), dim3(256), 0, 0, reinterpret_cast<const uint4*>(keep_q), reinterpret_cast<const uint4*>(c->qkb), (cross ? qbytes / 2 : qbytes) / 16, cnt);
      hipLaunchKernelGGL(k_count_diff, dim3(1024), dim3(256), 0, 0, reinterpret_cast<const uint4*>(keep_v), reinterpret_cast<const uint4*>(c->vtb), vbytes / 16, cnt);
    }
  c->use_lists = lists0; c->qkv_in_tail = fused0; c->guard = guard0; c->ktiming = kt0; c->launch_count = lc0; c->stop_after = sa0;
  unsigned int diffs = 0;
  GN_HIP(hipStreamSynchronize(0));
  GN_HIP(hipMemcpy(&diffs, cnt, sizeof diffs, hipMemcpyDeviceToHost));
  // leave the workspaces as a fresh context has them
  GN_HIP(hipMemset(c->x_p, 0, xbytes)); GN_HIP(hipMemset(c->ctx_p, 0, xbytes)); GN_HIP(hipMemset(c->msg_p, 0, xbytes)); GN_HIP(hipMemset(c->h_p, 0, qbytes + vbytes));
  GN_HIP(hipMemset(c->qkb, 0, qbytes)); GN_HIP(hipMemset(c->vtb, 0, vbytes)); GN_HIP(hipMemset(c->x, 0, xbytes)); GN_HIP(hipMemset(c->ctx, 0, xbytes));
  GN_HIP(hipMemset(cnt, 0, sizeof(unsigned int)));
  if (!applicable) return GN_OK;
  c->fused_proj_status = diffs == 0 ? 1 : 0;
  if (diffs != 0) {
    c->qkv_in_tail = 0;
    fprintf(stderr, "[gisnav_amd] the projection fused behind the block tail differs from the separate k_qkv launches in %u 16-byte words on this context's weights: "
                    "fusion switched off for this context (results stay correct, ~4 %% slower)\n", diffs);
  }
  return GN_OK;
}

// The exact-f32 arithmetic inside a context of another precision, for the duration of a scope: GN_PREC_F32's kernels read the f32 weights (kept in
// every mode) and the f32 workspaces (allocated in every mode); nothing of the hm16 / 16-bit state is touched.

// gn_set_ffn_products(0): one certified call's flags (those of the level it ran on, and the other level's from the same scores) into the window
void ffn_level_update(gn_ctx* c, int B, const int32_t* flags, const int32_t* alt, int lvl) {      // lvl: the level the call ran on (a deferred call's: saved with it)
  if (!ffn_auto(c) || c->precision == GN_PREC_F32 || lvl < 2 || lvl > 3) return;
  ++c->auto_calls_lvl[lvl - 2];
  int nf = 0, na = 0;
  for (int b = 0; b < B; ++b) { nf += flags[b] == 1; na += alt[b] == 1; }
  c->auto_pairs += B; c->auto_wide += lvl == 2 ? nf : na; c->auto_narrow += lvl == 2 ? na : nf;
  if (c->auto_pairs < 64) return;
  const int next = (c->auto_wide - c->auto_narrow) * 64 <= c->auto_pairs ? 2 : 3;
  if (next != c->auto_level) ++c->auto_switches;
  c->auto_level = next;
  c->auto_pairs = c->auto_wide = c->auto_narrow = 0;
}

struct F32Scope {
  gn_ctx* c; int precision, planes_mode, gemm_variant, no_planes, attn_f16;
  explicit F32Scope(gn_ctx* c_) : c(c_), precision(c_->precision), planes_mode(c_->planes_mode), gemm_variant(c_->gemm_variant), no_planes(c_->no_planes), attn_f16(c_->attn_f16) {
    c->precision = GN_PREC_F32; c->planes_mode = 0; c->gemm_variant = 3; c->no_planes = 1; c->attn_f16 = 0;
  }
  ~F32Scope() { c->precision = precision; c->planes_mode = planes_mode; c->gemm_variant = gemm_variant; c->no_planes = no_planes; c->attn_f16 = attn_f16; }
};

// The per-pair arrays of a call (inputs, outputs), as pointers to its first pair, and what is needed to step from pair to pair
struct CertView { const float *desc_q, *kpt_q, *desc_r, *kpt_r; const int32_t *n_q, *n_r; const uint8_t* dem;
                  double *R, *t; int32_t *n_match, *n_inliers; uint8_t* ok; int64_t* idx; float* score; };
struct CertShape { int stride_q, stride_r, H, W, kw, in_dim; size_t km; };
CertView cert_view_at(const CertView& v, const CertShape& h, size_t b) {
  CertView o = v;
  if (v.desc_q) o.desc_q = v.desc_q + b * h.stride_q * h.in_dim;
  if (v.desc_r) o.desc_r = v.desc_r + b * h.stride_r * h.in_dim;
  o.kpt_q = v.kpt_q + b * h.stride_q * h.kw; o.kpt_r = v.kpt_r + b * h.stride_r * h.kw; o.n_q = v.n_q + b; o.n_r = v.n_r + b;
  if (v.dem) o.dem = v.dem + b * (size_t)h.H * h.W;
  if (v.R) o.R = v.R + b * 9;
  if (v.t) o.t = v.t + b * 3;
  o.n_match = v.n_match + b;
  if (v.n_inliers) o.n_inliers = v.n_inliers + b;
  if (v.ok) o.ok = v.ok + b;
  if (v.idx) o.idx = v.idx + b * h.km * 2;
  if (v.score) o.score = v.score + b * h.km;
  return o;
}

// gn_set_certify(2 / 3): the per-pair flags of a call (read here from the device -- synchronises s -- or handed in), then the flagged pairs again with
// the context switched to the exact-f32 arithmetic, through `run(view, n)` = matcher (+ gather + PnP) of n consecutive pairs.  One contiguous run of
// flagged pairs is re-run IN PLACE (pointers and per-pair workspaces moved to its first pair).  Scattered flagged pairs are GATHERED into a staging
// block first (device-to-device copies of their inputs), run as ONE batch on workspace slots 0 .. nf - 1, and their outputs scattered back: a
// batched f32 call costs ~0.8 ms per pair where one- and two-pair calls cost 1.7 / 1.2 (round 6: mid-margin weights 1.26 k -> 1.9 k certified pairs/s).
// Counts what it saw (gn_get_certify_stats).
template <typename F> int certify_rerun(gn_ctx* ctx, int B, hipStream_t s, const CertShape& h, const CertView& v, F&& run, const int32_t* flags_ready = nullptr, int flags_level = 0) {
  if (!flags_ready) {
    GN_HIP(hipMemcpyAsync(ctx->uncert_host, ctx->uncert, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (ffn_auto(ctx)) GN_HIP(hipMemcpyAsync(ctx->uncert_host + ctx->max_batch, ctx->uncert_alt, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    GN_HIP(hipStreamSynchronize(s));
    flags_ready = ctx->uncert_host;
  }
  ffn_level_update(ctx, B, flags_ready, flags_ready + ctx->max_batch, flags_level ? flags_level : ctx->auto_level);     // (both callers lay the other level's flags max_batch entries behind)
  ++ctx->cert_calls; ctx->cert_pairs += B;
  std::vector<int> flagged;
  for (int b = 0; b < B; ++b) {
    const int f = flags_ready[b];
    if (f == 1) ++ctx->cert_flag_margin; else if (f == 2) ++ctx->cert_flag_range;
    if (f != 0) flagged.push_back(b);
  }
  if (flagged.empty()) return GN_OK;
  if (ctx->precision == GN_PREC_F32) { ctx->cert_f32_marginal += (long long)flagged.size(); return GN_OK; }   // already the exact arithmetic: counted, nothing better to run
  const int nf = (int)flagged.size();
  const bool contiguous = flagged.back() - flagged.front() + 1 == nf;
  int rc = GN_OK;
  CertView st = v;          // staged view (compact form)
  if (!contiguous) {
    // staging block: [inputs of nf pairs | outputs of nf pairs], 256-byte aligned pieces; grown on demand, owned by the context
    const size_t a256 = 255;
    size_t off = 0;
    auto piece = [&](size_t bytes) { const size_t o = off; off = (off + bytes + a256) & ~a256; return o; };
    const size_t o_dq = v.desc_q ? piece((size_t)nf * h.stride_q * h.in_dim * 4) : 0, o_dr = v.desc_r ? piece((size_t)nf * h.stride_r * h.in_dim * 4) : 0;
    const size_t o_kq = piece((size_t)nf * h.stride_q * h.kw * 4), o_kr = piece((size_t)nf * h.stride_r * h.kw * 4), o_nq = piece((size_t)nf * 4), o_nr = piece((size_t)nf * 4);
    const size_t o_dem = v.dem ? piece((size_t)nf * h.H * h.W) : 0;
    const size_t o_R = v.R ? piece((size_t)nf * 72) : 0, o_t = v.t ? piece((size_t)nf * 24) : 0, o_nm = piece((size_t)nf * 4), o_ni = v.n_inliers ? piece((size_t)nf * 4) : 0, o_ok = v.ok ? piece((size_t)nf) : 0;
    const size_t o_idx = v.idx ? piece((size_t)nf * h.km * 16) : 0, o_sc = v.score ? piece((size_t)nf * h.km * 4) : 0;
    if (off > ctx->cert_stage_bytes) {
      if (ctx->cert_stage) { GN_HIP(hipStreamSynchronize(s)); GN_HIP(hipFree(ctx->cert_stage)); ctx->cert_stage = nullptr; ctx->cert_stage_bytes = 0; }
      GN_HIP(hipMalloc(&ctx->cert_stage, off));
      ctx->cert_stage_bytes = off;
    }
    char* const base = static_cast<char*>(ctx->cert_stage);
    st.desc_q = v.desc_q ? reinterpret_cast<const float*>(base + o_dq) : nullptr; st.desc_r = v.desc_r ? reinterpret_cast<const float*>(base + o_dr) : nullptr;
    st.kpt_q = reinterpret_cast<const float*>(base + o_kq); st.kpt_r = reinterpret_cast<const float*>(base + o_kr);
    st.n_q = reinterpret_cast<const int32_t*>(base + o_nq); st.n_r = reinterpret_cast<const int32_t*>(base + o_nr);
    st.dem = v.dem ? reinterpret_cast<const uint8_t*>(base + o_dem) : nullptr;
    st.R = v.R ? reinterpret_cast<double*>(base + o_R) : nullptr; st.t = v.t ? reinterpret_cast<double*>(base + o_t) : nullptr;
    st.n_match = reinterpret_cast<int32_t*>(base + o_nm); st.n_inliers = v.n_inliers ? reinterpret_cast<int32_t*>(base + o_ni) : nullptr;
    st.ok = v.ok ? reinterpret_cast<uint8_t*>(base + o_ok) : nullptr;
    st.idx = v.idx ? reinterpret_cast<int64_t*>(base + o_idx) : nullptr; st.score = v.score ? reinterpret_cast<float*>(base + o_sc) : nullptr;
    for (int k = 0; k < nf; ++k) {       // gather the inputs of flagged pair k into slot k
      const CertView src = cert_view_at(v, h, (size_t)flagged[k]), dst = cert_view_at(st, h, (size_t)k);
      auto cp = [&](const void* d, const void* q, size_t bytes) { return hipMemcpyAsync(const_cast<void*>(d), q, bytes, hipMemcpyDeviceToDevice, s); };
      if (v.desc_q) GN_HIP(cp(dst.desc_q, src.desc_q, (size_t)h.stride_q * h.in_dim * 4));
      if (v.desc_r) GN_HIP(cp(dst.desc_r, src.desc_r, (size_t)h.stride_r * h.in_dim * 4));
      GN_HIP(cp(dst.kpt_q, src.kpt_q, (size_t)h.stride_q * h.kw * 4)); GN_HIP(cp(dst.kpt_r, src.kpt_r, (size_t)h.stride_r * h.kw * 4));
      GN_HIP(cp(dst.n_q, src.n_q, 4)); GN_HIP(cp(dst.n_r, src.n_r, 4));
      if (v.dem) GN_HIP(cp(dst.dem, src.dem, (size_t)h.H * h.W));
    }
  }
  {
    F32Scope f32(ctx);
    const bool ig = ctx->in_group; unsigned int* const ovf = ctx->ovf;
    ctx->cert_inner = true; ctx->in_group = true; ctx->ovf = ctx->ovf_base;
    if (contiguous) {
      const int b0 = flagged.front();
      shift_workspaces(ctx, b0, +1);
      rc = run(cert_view_at(v, h, (size_t)b0), nf);
      shift_workspaces(ctx, b0, -1);
    } else {
      rc = run(st, nf);
    }
    ctx->cert_rerun += nf;
    ctx->cert_inner = false; ctx->in_group = ig; ctx->ovf = ovf;
  }
  if (rc != GN_OK) return rc;
  if (!contiguous) {
    for (int k = 0; k < nf; ++k) {       // scatter the outputs of slot k to flagged pair k
      const CertView dst = cert_view_at(v, h, (size_t)flagged[k]), src = cert_view_at(st, h, (size_t)k);
      auto cp = [&](void* d, const void* q, size_t bytes) { return hipMemcpyAsync(d, q, bytes, hipMemcpyDeviceToDevice, s); };
      if (v.R) GN_HIP(cp(dst.R, src.R, 72));
      if (v.t) GN_HIP(cp(dst.t, src.t, 24));
      GN_HIP(cp(dst.n_match, src.n_match, 4));
      if (v.n_inliers) GN_HIP(cp(dst.n_inliers, src.n_inliers, 4));
      if (v.ok) GN_HIP(cp(dst.ok, src.ok, 1));
      if (v.idx) GN_HIP(cp(dst.idx, src.idx, h.km * 16));
      if (v.score) GN_HIP(cp(dst.score, src.score, h.km * 4));
    }
  }
  // the re-run's own flags (stated for cert_eps_f32): how many of the pairs are marginal even in exact f32 -- counted, reported, not acted upon
  GN_HIP(hipMemcpyAsync(ctx->uncert_host, ctx->uncert, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  GN_HIP(hipStreamSynchronize(s));
  for (int k = 0; k < nf; ++k) if (ctx->uncert_host[contiguous ? flagged[k] : k] != 0) ++ctx->cert_f32_marginal;
  return GN_OK;
}

int check_fwd(gn_ctx* ctx, int B, int stride_q, int stride_r) {
  if (!ctx) return fail(nullptr, GN_ERR_ARG, "null context");
  if (ctx->npad <= 0) return fail(ctx, GN_ERR_ARG, "context has no workspaces (a gn_resize failed): call gn_resize again");
  if (B < 1 || B > ctx->max_batch) return fail(ctx, GN_ERR_ARG, "B out of range for this context");
  if (stride_q < 1 || stride_r < 1 || stride_q > ctx->npad || stride_r > ctx->npad)
    return fail(ctx, GN_ERR_ARG, "keypoint stride exceeds max_kpts of this context");
  if (gn_missing_tensors(ctx) != 0) return fail(ctx, GN_ERR_WEIGHTS, "weights not fully loaded");
  const int rc_c = ensure_composed(ctx);      // (host work on the first call after a (re)load only)
  if (rc_c != GN_OK) return rc_c;
  if (ctx->fused_proj_pending && !ctx->in_group && !ctx->cert_inner) return selfcheck_fused_projection(ctx);
  return GN_OK;
}

}  // namespace

extern "C" {

// gn_version / gn_source_digest: gn_build_id.hip (carries the digest of the sources this binary was built from)

const char* gn_last_error(const gn_ctx* ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

int gn_create(int device, int max_batch, int max_kpts, int precision, gn_ctx** out) {
  return gn_create_ex(device, max_batch, max_kpts, precision, GN_FEATURE_SIFT, out);
}

int gn_set_image_size(gn_ctx* ctx, float w_q, float h_q, float w_r, float h_r) {
  if (!ctx) return GN_ERR_ARG;
  ctx->size_q[0] = w_q; ctx->size_q[1] = h_q; ctx->size_r[0] = w_r; ctx->size_r[1] = h_r;
  return GN_OK;
}

int gn_create_ex(int device, int max_batch, int max_kpts, int precision, int feature, gn_ctx** out) {
  gn_ctx* ctx = nullptr;
  if (!out || max_batch < 1 || max_kpts < 2) return fail(nullptr, GN_ERR_ARG, "bad gn_create argument");
  if (precision != GN_PREC_F32 && precision != GN_PREC_BF16_ATTN && precision != GN_PREC_F32X3_BF16_ATTN &&
      precision != GN_PREC_F16X2_BF16_ATTN && precision != GN_PREC_F16X2_F16_ATTN)
    return fail(nullptr, GN_ERR_ARG, "bad precision");
  const int precision_api = precision;
  if (precision == GN_PREC_F16X2_F16_ATTN) precision = GN_PREC_F16X2_BF16_ATTN;   // the same projections / FFN / head; only the attention operand format differs (ctx->attn_f16)
  GN_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  GN_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, GN_ERR_ARCH, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
  ctx = new gn_ctx();
  if (feature != GN_FEATURE_SIFT && feature != GN_FEATURE_SUPERPOINT) { delete ctx; return fail(nullptr, GN_ERR_ARG, "bad feature type"); }
  ctx->device = device; ctx->max_batch = max_batch; ctx->precision = precision; ctx->feature = feature;
  ctx->precision_api = precision_api; ctx->attn_f16 = precision_api == GN_PREC_F16X2_F16_ATTN ? 1 : 0;
  ctx->gemm_variant = precision == GN_PREC_F16X2_BF16_ATTN ? 6 : precision == GN_PREC_F32X3_BF16_ATTN ? 5 : 3;
  { const int rc_ws = alloc_workspace(ctx, max_kpts); if (rc_ws != GN_OK) { gn_destroy(ctx); return rc_ws; } }
  // pinned host words: [0, 16) guard words, [16, 16 + 4096) the per-image counters the SIFT / SuperPoint calls read back (up to 1024 images per call)
  if (hipHostMalloc((void**)&ctx->ovf_host, (16 + 4096 + 16) * sizeof(unsigned int), hipHostMallocDefault) != hipSuccess) { gn_destroy(ctx); return fail(nullptr, GN_ERR_HIP, "hipHostMalloc failed"); }
  memset(ctx->ovf_host, 0, (16 + 4096 + 16) * sizeof(unsigned int));
  ctx->tile_feedback = reinterpret_cast<unsigned long long*>(ctx->ovf_host + 16 + 4096);   // [8] x 8 bytes behind the counters (8-byte aligned)
  if (hipHostMalloc((void**)&ctx->uncert_host, (size_t)2 * max_batch * sizeof(int32_t), hipHostMallocDefault) != hipSuccess) { gn_destroy(ctx); return fail(nullptr, GN_ERR_HIP, "hipHostMalloc failed"); }
  memset(ctx->uncert_host, 0, (size_t)2 * max_batch * sizeof(int32_t));
  for (auto& pd : ctx->cert_pend) {
    if (hipHostMalloc((void**)&pd.flags, (size_t)2 * max_batch * sizeof(int32_t), hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&pd.ev, hipEventDisableTiming) != hipSuccess) { gn_destroy(ctx); return fail(nullptr, GN_ERR_HIP, "certificate slots: allocation failed"); }
  }
  ctx->ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  for (int i = 0; i < 256; ++i) hipEventCreate(&ctx->ev[i]);
  ctx->ev_ready = true;
  // required tensor names
  auto req = [&](const std::string& n) { ctx->required.push_back(n + ".weight"); ctx->required.push_back(n + ".bias"); };
  if (feature == GN_FEATURE_SIFT) req("input_proj");
  ctx->required.push_back("posenc.Wr.weight");
  for (int i = 0; i < kMaxLayers; ++i) {
    const std::string ps = "transformers." + std::to_string(i) + ".self_attn.", pc = "transformers." + std::to_string(i) + ".cross_attn.";
    req(ps + "Wqkv"); req(ps + "out_proj"); req(ps + "ffn.0"); req(ps + "ffn.1"); req(ps + "ffn.3");
    req(pc + "to_qk"); req(pc + "to_v"); req(pc + "to_out"); req(pc + "ffn.0"); req(pc + "ffn.1"); req(pc + "ffn.3");
    req("log_assignment." + std::to_string(i) + ".final_proj");
    req("log_assignment." + std::to_string(i) + ".matchability");
  }
  *out = ctx;
  return GN_OK;
}

int gn_resize(gn_ctx* ctx, int max_kpts) {
  if (!ctx || max_kpts < 2) return fail(ctx, GN_ERR_ARG, "bad gn_resize argument");
  if (((max_kpts + 127) / 128) * 128 == ctx->npad) return GN_OK;
  GN_HIP(hipSetDevice(ctx->device));
  GN_HIP(hipDeviceSynchronize());                      // nothing may still be reading the old workspaces
  for (void* p : ctx->ws_allocs) hipFree(p);
  ctx->ws_allocs.clear();
  ctx->sim = nullptr;                                  // allocated on first use (ensure_sim), for the new size
  for (bool& b : ctx->sub_pending) b = false;
  // A failure part-way (out of memory while growing) must not leave the context pointing at freed workspaces: every entry point goes
  // through check_fwd, which refuses a context whose npad is 0 until a later gn_resize succeeds.
  auto broken = [&](int rc) { for (void* p : ctx->ws_allocs) hipFree(p); ctx->ws_allocs.clear(); ctx->npad = 0; ctx->npad_run = 0; return rc; };
  { const int rc = alloc_workspace(ctx, max_kpts); if (rc != GN_OK) return broken(rc); }
  if (ctx->s_pnp) {                                    // the overlapped pose stage's double buffers follow the padded size too
    const size_t B = ctx->max_batch, np = ctx->npad;
    for (int i = 0; i < 2; ++i) {
      ctx->pnp_pending[i] = false;
      int rc = ws_alloc(ctx, &ctx->o_mkp[i], B * np * 2); if (rc != GN_OK) return broken(rc);
      rc = ws_alloc(ctx, &ctx->o_obj[i], B * np * 3); if (rc != GN_OK) return broken(rc);
      rc = ws_alloc(ctx, &ctx->o_nmatch[i], B); if (rc != GN_OK) return broken(rc);
    }
  }
  return GN_OK;
}

void gn_destroy(gn_ctx* ctx) {
  if (!ctx) return;
  hipSetDevice(ctx->device);
  for (void* p : ctx->allocs) hipFree(p);
  for (void* p : ctx->ws_allocs) hipFree(p);
  if (ctx->ovf_host) hipHostFree(ctx->ovf_host);
  if (ctx->uncert_host) hipHostFree(ctx->uncert_host);
  if (ctx->cert_stage) hipFree(ctx->cert_stage);
  for (auto& pd : ctx->cert_pend) { if (pd.flags) hipHostFree(pd.flags); if (pd.ev) hipEventDestroy(pd.ev); }
  if (ctx->ev_ready) for (int i = 0; i < 256; ++i) hipEventDestroy(ctx->ev[i]);
  for (hipEvent_t e : ctx->kev) hipEventDestroy(e);
  for (void* p : ctx->sift_allocs) hipFree(p);
  for (void* p : ctx->sp_allocs) hipFree(p);
  if (ctx->sp_allocs_dbg) hipFree(ctx->sp_allocs_dbg);
  if (ctx->ev_fork) {
    for (int i = 0; i < 8; ++i) if (ctx->sub_s[i]) { hipStreamSynchronize(ctx->sub_s[i]); hipEventDestroy(ctx->ev_join[i]); hipStreamDestroy(ctx->sub_s[i]); }
    hipEventDestroy(ctx->ev_fork);
  }
  if (ctx->s_pnp) {
    hipStreamSynchronize(ctx->s_pnp);
    for (int i = 0; i < 2; ++i) { hipEventDestroy(ctx->ev_gather[i]); hipEventDestroy(ctx->ev_pnp[i]); }
    hipStreamDestroy(ctx->s_pnp);
  }
  delete ctx;
}

int gn_missing_tensors(const gn_ctx* ctx) {
  if (!ctx) return -1;
  int missing = 0;
  for (const auto& n : ctx->required) {
    // log_assignment.{i} for i other than the last configured layer is optional
    if (n.compare(0, 15, "log_assignment.") == 0) {
      const int i = atoi(n.c_str() + 15);
      if (i != ctx->n_layers - 1) continue;
    }
    if (n.compare(0, 13, "transformers.") == 0) {
      const int i = atoi(n.c_str() + 13);
      if (i >= ctx->n_layers) continue;
    }
    if (!ctx->loaded.count(n)) ++missing;
  }
  return missing;
}

int gn_set_num_layers(gn_ctx* ctx, int n_layers) {
  if (!ctx || n_layers < 1 || n_layers > kMaxLayers) return fail(ctx, GN_ERR_ARG, "n_layers must be in 1..9");
  ctx->n_layers = n_layers;
  return GN_OK;
}

int gn_set_filter_threshold(gn_ctx* ctx, float th) {
  if (!ctx) return GN_ERR_ARG;
  ctx->threshold = th;
  return GN_OK;
}

int gn_kmax(const gn_ctx* ctx) { return ctx ? ctx->npad : GN_ERR_ARG; }

int gn_load_tensor(gn_ctx* ctx, const char* name_c, const float* host, const int64_t* shape, int ndim) {
  if (!ctx || !name_c || !host || !shape || ndim < 1 || ndim > 2) return fail(ctx, GN_ERR_ARG, "bad gn_load_tensor argument");
  GN_HIP(hipSetDevice(ctx->device));
  const std::string name = canonical(name_c);
  if (name.compare(0, 17, "token_confidence.") == 0 || name == "confidence_thresholds") return GN_OK;  // dead in the live config
  const int64_t d0 = shape[0], d1 = ndim == 2 ? shape[1] : 1;
  auto upload = [&](float** dst, const float* src, size_t count) -> int {
    if (*dst == nullptr) { int rc = dalloc(ctx, dst, count); if (rc != GN_OK) return rc; }
    GN_HIP(hipMemcpy(*dst, src, count * sizeof(float), hipMemcpyHostToDevice));
    return GN_OK;
  };
  auto shape_err = [&]() { return fail(ctx, GN_ERR_SHAPE, "shape mismatch for " + name); };
  const bool is_w = name.size() > 7 && name.compare(name.size() - 7, 7, ".weight") == 0;
  const bool is_b = name.size() > 5 && name.compare(name.size() - 5, 5, ".bias") == 0;
  if (!is_w && !is_b) return fail(ctx, GN_ERR_NAME, "unknown tensor " + name);
  const std::string base = name.substr(0, name.size() - (is_w ? 7 : 5));

  auto load_linear = [&](Linear& L, int out, int in, int row_off, int rows_total) -> int {
    // loads rows [row_off, row_off + out) of a (rows_total x in) fused matrix
    if (is_w) {
      if (ndim != 2 || d0 != out || d1 != in) return shape_err();
      if (!L.w) { int rc = dalloc(ctx, &L.w, (size_t)rows_total * in); if (rc != GN_OK) return rc; }
      GN_HIP(hipMemcpy(L.w + (size_t)row_off * in, host, (size_t)out * in * sizeof(float), hipMemcpyHostToDevice));
      L.out = rows_total; L.in = in;
      { int rcp = build_planes(ctx, L); if (rcp != GN_OK) return rcp; }
    } else {
      if (d0 != out || d1 != 1) return shape_err();
      if (!L.b) { int rc = dalloc(ctx, &L.b, (size_t)rows_total); if (rc != GN_OK) return rc; }
      GN_HIP(hipMemcpy(L.b + row_off, host, (size_t)out * sizeof(float), hipMemcpyHostToDevice));
    }
    L.out = rows_total; L.in = in;
    return GN_OK;
  };

  int rc = GN_ERR_NAME;
  if (base == "input_proj") {
    if (ctx->feature != GN_FEATURE_SIFT) return fail(ctx, GN_ERR_NAME, "input_proj does not exist for 256-d features (input_dim == descriptor_dim)");
    rc = load_linear(ctx->input_proj, kDim, kInDim, 0, kDim);
  } else if (base == "posenc.Wr") {
    const int pin = ctx->feature == GN_FEATURE_SIFT ? 4 : 2;   // (x, y, scale, ori) or (x, y)
    if (!is_w || ndim != 2 || d0 != kFreq || d1 != pin) return shape_err();
    rc = upload(&ctx->wr, host, kFreq * pin);
  } else if (base.compare(0, 13, "transformers.") == 0) {
    const int i = atoi(base.c_str() + 13);
    if (i < 0 || i >= kMaxLayers) return fail(ctx, GN_ERR_NAME, "layer index out of range in " + name);
    const size_t p1 = base.find('.', 13);
    const std::string rest = base.substr(p1 + 1);  // e.g. self_attn.Wqkv
    const bool self = rest.compare(0, 10, "self_attn.") == 0;
    const bool cross = rest.compare(0, 11, "cross_attn.") == 0;
    if (!self && !cross) return fail(ctx, GN_ERR_NAME, "unknown tensor " + name);
    const std::string leaf = rest.substr(self ? 10 : 11);
    Block& blk = self ? ctx->self_blk[i] : ctx->cross_blk[i];
    if (self && leaf == "Wqkv") {
      // kornia splits the 768 outputs as unflatten(-1, (heads, 64, 3)): flat = h*192 + d*3 + s.
      // Re-order rows to [s][h][d] so q, k, v are contiguous 256-wide panels (numerically identical:
      // every output feature keeps its own weight row and bias).
      if (is_w) { if (ndim != 2 || d0 != 3 * kDim || d1 != kDim) return shape_err(); }
      else if (d0 != 3 * kDim || d1 != 1) return shape_err();
      const int in = is_w ? kDim : 1;
      std::vector<float> tmp((size_t)3 * kDim * in);
      for (int hh = 0; hh < kHeads; ++hh)
        for (int d = 0; d < kHeadDim; ++d)
          for (int s3 = 0; s3 < 3; ++s3) {
            const int src = hh * 192 + d * 3 + s3, dst = s3 * kDim + hh * kHeadDim + d;
            memcpy(&tmp[(size_t)dst * in], host + (size_t)src * in, in * sizeof(float));
          }
      Linear& L = blk.proj_in;
      L.frag_order = 0;
      if (is_w) { rc = upload(&L.w, tmp.data(), tmp.size()); } else { rc = upload(&L.b, tmp.data(), tmp.size()); }
      L.out = 3 * kDim; L.in = kDim;
      if (is_w && rc == GN_OK) rc = build_planes(ctx, L);
    } else if (self && leaf == "out_proj") { blk.comp_dirty = true; blk.proj_out.frag_order = 0; rc = load_linear(blk.proj_out, kDim, kDim, 0, kDim); }
    else if (cross && leaf == "to_qk") { blk.proj_in.frag_order = 0; rc = load_linear(blk.proj_in, kDim, kDim, 0, 2 * kDim); }
    else if (cross && leaf == "to_v") { blk.proj_in.frag_order = 0; rc = load_linear(blk.proj_in, kDim, kDim, kDim, 2 * kDim); }
    else if (cross && leaf == "to_out") { blk.comp_dirty = true; blk.proj_out.frag_order = 0; rc = load_linear(blk.proj_out, kDim, kDim, 0, kDim); }
    else if (leaf == "ffn.0") { blk.comp_dirty = true; blk.ffn0.frag_order = 0; rc = load_linear(blk.ffn0, 2 * kDim, 2 * kDim, 0, 2 * kDim); }
    else if (leaf == "ffn.3") { blk.ffn3.frag_order = 1; rc = load_linear(blk.ffn3, kDim, 2 * kDim, 0, kDim); }
    else if (leaf == "ffn.1") {
      if (d0 != 2 * kDim || d1 != 1) return shape_err();
      rc = upload(is_w ? &blk.ln_g : &blk.ln_b, host, 2 * kDim);
    } else return fail(ctx, GN_ERR_NAME, "unknown tensor " + name);
  } else if (base.compare(0, 15, "log_assignment.") == 0) {
    const int i = atoi(base.c_str() + 15);
    if (i < 0 || i >= kMaxLayers) return fail(ctx, GN_ERR_NAME, "layer index out of range in " + name);
    const std::string leaf = base.substr(base.find('.', 15) + 1);
    if (leaf == "final_proj") rc = load_linear(ctx->final_proj[i], kDim, kDim, 0, kDim);
    else if (leaf == "matchability") {
      if (is_w) { if (ndim != 2 || d0 != 1 || d1 != kDim) return shape_err(); rc = upload(&ctx->matchability[i].w, host, kDim); }
      else { if (d0 != 1) return shape_err(); rc = upload(&ctx->matchability[i].b, host, 1); }
    } else return fail(ctx, GN_ERR_NAME, "unknown tensor " + name);
  } else {
    return fail(ctx, GN_ERR_NAME, "unknown tensor " + name);
  }
  if (rc == GN_OK) {
    ctx->loaded[name] = true; ctx->fused_proj_pending = true;
    // a calibration belongs to the weights it was measured on: the automatic block-tail level goes back to "not calibrated" (three products, cert_eps)
    ctx->cert_eps_lvl[0] = ctx->cert_eps_lvl[1] = -1.f; ctx->auto_level = 3; ctx->auto_pairs = ctx->auto_wide = ctx->auto_narrow = 0;
  }
  return rc;
}

int gn_match(gn_ctx* ctx, int B, int kpt_format,
             const float* desc_q, const float* kpt_q, const int32_t* n_q, int stride_q,
             const float* desc_r, const float* kpt_r, const int32_t* n_r, int stride_r,
             int64_t* idx, float* score, int32_t* n_match, void* stream) {
  int rc = check_fwd(ctx, B, stride_q, stride_r);
  if (rc != GN_OK) return rc;
  const int kfmt = kpt_format & 0xff;
  if (kfmt != GN_KPT_LAF && kfmt != GN_KPT_XYSA && kfmt != GN_KPT_RECORD) return fail(ctx, GN_ERR_ARG, "bad kpt_format");
  if (kfmt == GN_KPT_RECORD && ctx->feature != GN_FEATURE_SIFT) return fail(ctx, GN_ERR_ARG, "GN_KPT_RECORD (KEYPOINT_DTYPE wire records) needs a SIFT context");
  if (((!desc_q || !desc_r) && kfmt != GN_KPT_RECORD) || !kpt_q || !n_q || !kpt_r || !n_r || !idx || !score || !n_match)
    return fail(ctx, GN_ERR_ARG, "null pointer passed to gn_match");
  GN_HIP(hipSetDevice(ctx->device));
  ctx->n_ev = 0;
  if (!ctx->in_group) ctx->ovf_groups_last = 1;
  rc = run_matcher(ctx, B, kpt_format, desc_q, kpt_q, n_q, stride_q, desc_r, kpt_r, n_r, stride_r, idx, score, n_match,
                   (hipStream_t)stream);
  if (rc != GN_OK) return rc;
  if (ctx->certify >= 2 && !ctx->in_group && !ctx->cert_inner) {
    // certified mode (3 = deferred exists for gn_estimate's sub-batch-stream path only; everywhere else it is 2): read this call's per-pair flags (one stream synchronisation -- the reference's call site synchronises right after the
    // matcher anyway, pose_node.py:296-297) and run the flagged pairs again on the exact-f32 kernels; covers the fp16-range fallback too
    const int kw = kfmt == GN_KPT_LAF ? 6 : kfmt == GN_KPT_RECORD ? kRecordFloats : 4;
    const int in_dim = ctx->feature == GN_FEATURE_SIFT ? kInDim : kDim;
    const size_t km = (size_t)ctx->npad;
    const CertShape shp{stride_q, stride_r, 0, 0, kw, in_dim, km};
    const CertView view{desc_q, kpt_q, desc_r, kpt_r, n_q, n_r, nullptr, nullptr, nullptr, n_match, nullptr, nullptr, idx, score};
    rc = certify_rerun(ctx, B, (hipStream_t)stream, shp, view, [&](const CertView& w, int n) {
      return run_matcher(ctx, n, kpt_format, w.desc_q, w.kpt_q, w.n_q, stride_q, w.desc_r, w.kpt_r, w.n_r, stride_r, w.idx, w.score, w.n_match, (hipStream_t)stream);
    });
    if (rc != GN_OK) return rc;
  } else if (ctx->planes_mode && ctx->guard == 2 && ctx->certify < 2) {
    // guarded mode: observe the domain word of THIS call (one stream synchronisation -- the reference's call site synchronises
    // right after the matcher anyway, pose_node.py:296-297) and, if an activation left the fp16 range, run the call again with
    // every operand split exactly into three bf16 terms (the f32x3 mode: f32 range, f32 accuracy), on the f32 workspaces
    GN_HIP(hipMemcpyAsync(ctx->ovf_host, ctx->ovf, sizeof(unsigned int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    GN_HIP(hipStreamSynchronize((hipStream_t)stream));
    if (*ctx->ovf_host != 0u) {
      ++ctx->guard_trips;
      const int pm = ctx->planes_mode, gv = ctx->gemm_variant, npl = ctx->no_planes, af = ctx->attn_f16;
      ctx->planes_mode = 0; ctx->gemm_variant = 5; ctx->no_planes = 1;   // weights stay as hm16 planes: f32x3 splits the f32 weights on the fly
      ctx->attn_f16 = 0;                                                   // and the attention operands go back to bf16 (f32's exponent range)
      rc = run_matcher(ctx, B, kpt_format, desc_q, kpt_q, n_q, stride_q, desc_r, kpt_r, n_r, stride_r, idx, score, n_match,
                       (hipStream_t)stream);
      ctx->planes_mode = pm; ctx->gemm_variant = gv; ctx->no_planes = npl; ctx->attn_f16 = af;
      if (rc != GN_OK) return rc;
    }
  }
  GN_HIP(hipGetLastError());
  return GN_OK;
}

int gn_set_guard(gn_ctx* ctx, int mode) {
  if (!ctx || mode < 0 || mode > 2) return GN_ERR_ARG;
  if (mode != ctx->guard && ctx->ovf_base) {   // a mode change starts from clean words: a trip raised under the old mode must not be seen by the new one
    GN_HIP(hipSetDevice(ctx->device));
    GN_HIP(hipDeviceSynchronize());
    GN_HIP(hipMemset(ctx->ovf_base, 0, 8 * sizeof(unsigned int)));
  }
  ctx->guard = mode;
  return GN_OK;
}

int gn_get_guard_status(gn_ctx* ctx, void* stream, int32_t* last_call_tripped, int64_t* trips_total) {
  if (!ctx) return GN_ERR_ARG;
  GN_HIP(hipSetDevice(ctx->device));
  // (with deferred joins the caller flushes first: the words of unjoined groups are still being written)
  GN_HIP(hipMemcpyAsync(ctx->ovf_host, ctx->ovf_base, 8 * sizeof(unsigned int), hipMemcpyDeviceToHost, (hipStream_t)stream));
  GN_HIP(hipStreamSynchronize((hipStream_t)stream));
  unsigned int any = 0u;
  for (int g = 0; g < ctx->ovf_groups_last && g < 8; ++g) any |= ctx->ovf_host[g];
  const int tripped = (ctx->planes_mode && ctx->guard && any != 0u) ? 1 : 0;
  if (tripped && ctx->guard == 1) ++ctx->guard_trips;
  if (last_call_tripped) *last_call_tripped = tripped;
  if (trips_total) *trips_total = ctx->guard_trips;
  return GN_OK;
}

int gn_device_numa_node(int device) {
  char bus[64] = {0};
  if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) return -1;
  for (char* c = bus; *c; ++c) *c = (char)tolower((unsigned char)*c);
  const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/numa_node";
  FILE* f = fopen(path.c_str(), "r");
  if (!f) return -1;
  int node = -1;
  if (fscanf(f, "%d", &node) != 1) node = -1;
  fclose(f);
  return node;
}

int gn_fused_projection_status(const gn_ctx* ctx) { return ctx ? ctx->fused_proj_status : GN_ERR_ARG; }

int gn_set_ffn_products(gn_ctx* ctx, int products) {
  if (!ctx || (products != 0 && products != 2 && products != 3)) return GN_ERR_ARG;
  if (products != ctx->ffn_products) { ctx->auto_level = 3; ctx->auto_pairs = ctx->auto_wide = ctx->auto_narrow = 0; }
  ctx->ffn_products = products;
  return GN_OK;
}

int gn_set_ffn_level_eps(gn_ctx* ctx, float eps2, float eps3, int level) {
  if (!ctx || !(eps2 >= 0.f) || !(eps3 >= 0.f) || (level != 2 && level != 3)) return GN_ERR_ARG;
  ctx->cert_eps_lvl[0] = std::max(eps2, eps3); ctx->cert_eps_lvl[1] = eps3;
  ctx->auto_level = level; ctx->auto_pairs = ctx->auto_wide = ctx->auto_narrow = 0;
  return GN_OK;
}

int gn_get_ffn_level(gn_ctx* ctx, int32_t* level, float* eps2, float* eps3, int64_t* out4) {
  if (!ctx) return GN_ERR_ARG;
  if (level) *level = ffn_level(ctx);
  if (eps2) *eps2 = ctx->cert_eps_lvl[0];
  if (eps3) *eps3 = ctx->cert_eps_lvl[1];
  if (out4) { out4[0] = ctx->auto_calls_lvl[0]; out4[1] = ctx->auto_calls_lvl[1]; out4[2] = ctx->auto_switches; out4[3] = ffn_auto(ctx) ? 1 : 0; }
  return GN_OK;
}

int gn_set_certify(gn_ctx* ctx, int mode, float eps, float eps_f32) {
  if (!ctx || mode < 0 || mode > 3) return GN_ERR_ARG;
  if (ctx->certify == 3 && mode != 3 && (ctx->cert_pend[0].active || ctx->cert_pend[1].active))
    return fail(ctx, GN_ERR_ARG, "gn_set_certify: deferred certificates are still open -- gn_flush first");
  ctx->certify = mode;
  if (eps >= 0.f) ctx->cert_eps = eps;
  if (eps_f32 >= 0.f) ctx->cert_eps_f32 = eps_f32;
  return GN_OK;
}

int gn_calibrate_certify(gn_ctx* ctx, int B, int kpt_format,
                         const float* desc_q, const float* kpt_q, const int32_t* n_q, int stride_q,
                         const float* desc_r, const float* kpt_r, const int32_t* n_r, int stride_r,
                         float safety, float floor_eps, float* measured_host, float* eps_host, void* stream) {
  int rc = check_fwd(ctx, B, stride_q, stride_r);
  if (rc != GN_OK) return rc;
  if (!(safety >= 1.f) || !(floor_eps >= 0.f)) return fail(ctx, GN_ERR_ARG, "gn_calibrate_certify: safety must be >= 1, floor_eps >= 0");
  if (ctx->precision == GN_PREC_F32) return fail(ctx, GN_ERR_ARG, "gn_calibrate_certify: this context already computes in exact f32");
  if (ctx->n_sub > 1 || ctx->overlap) { const int rcf = gn_flush(ctx, stream); if (rcf != GN_OK) return rcf; }
  GN_HIP(hipSetDevice(ctx->device));
  hipStream_t s = (hipStream_t)stream;
  const size_t np = (size_t)ctx->npad_run, n = (size_t)B * np;
  int32_t* nm = nullptr;
  GN_HIP(hipMalloc((void**)&nm, (size_t)B * sizeof(int32_t)));
  // passes: the context's arithmetic -- on BOTH block-tail levels under gn_set_ffn_products(0) -- then the exact-f32 kernels
  const int setting = ctx->ffn_products;
  const int n_lv = setting == 0 ? 2 : 1;
  const int lv[2] = {setting == 0 ? 2 : setting, 3};
  std::vector<float> best[3], second[3];
  std::vector<int32_t> nv(2 * (size_t)B);
  unsigned int tripped = 0u;
  for (int pass = 0; pass <= n_lv && rc == GN_OK; ++pass) {
    std::unique_ptr<F32Scope> f32;
    if (pass == n_lv) f32.reset(new F32Scope(ctx)); else ctx->ffn_products = lv[pass];
    const bool ig = ctx->in_group; ctx->in_group = true;       // (no nested certification, ovf_groups_last untouched)
    rc = run_matcher(ctx, B, kpt_format, desc_q, kpt_q, n_q, stride_q, desc_r, kpt_r, n_r, stride_r, ctx->e_idx, ctx->e_score, nm, s);
    ctx->in_group = ig; ctx->ffn_products = setting;
    if (rc != GN_OK) break;
    best[pass].resize(n); second[pass].resize(n);
    unsigned int trip = 0u;
    if (hipStreamSynchronize(s) != hipSuccess || hipMemcpy(best[pass].data(), ctx->max0, n * 4, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(second[pass].data(), ctx->max0b, n * 4, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(nv.data(), ctx->nvalid, nv.size() * 4, hipMemcpyDeviceToHost) != hipSuccess ||
        (pass < n_lv && ctx->planes_mode && ctx->guard && hipMemcpy(&trip, ctx->ovf, 4, hipMemcpyDeviceToHost) != hipSuccess)) rc = GN_ERR_HIP;
    tripped |= trip;
  }
  hipFree(nm);
  if (rc != GN_OK) return rc == GN_ERR_HIP ? fail(ctx, rc, "gn_calibrate_certify: a HIP call failed") : rc;
  if (tripped) return fail(ctx, GN_ERR_ARG, "gn_calibrate_certify: the sample left the fp16 range of this precision mode (nothing to calibrate: such calls are re-run as a whole)");
  // largest difference between the two arithmetics over the entries a decision looks at: every valid row's best score and runner-up,
  // restricted -- when there is a threshold -- to rows that come within 1 (in log units) of it in either arithmetic
  const float L = ctx->threshold > 0.f ? logf(ctx->threshold) : -INFINITY;
  float eps_of[2] = {0.f, 0.f}, mx_of[2] = {0.f, 0.f};
  for (int k = 0; k < n_lv; ++k) {
    double mx = 0.0, mx_all = 0.0; long long rows = 0, rows_all = 0;
    for (int b = 0; b < B; ++b) {
      const int n0 = nv[2 * b], n1 = nv[2 * b + 1];
      if (n0 < 2 || n1 < 2) continue;
      for (int i = 0; i < n0; ++i) {
        const size_t o = (size_t)b * np + i;
        const float bf = best[k][o], be = best[n_lv][o], sf = second[k][o], se = second[n_lv][o];
        const double d1 = std::fabs((double)bf - be), d2 = std::fabs((double)sf - se);
        const double d = std::max(std::isfinite(d1) ? d1 : (double)INFINITY, std::isfinite(d2) ? d2 : 0.0);
        ++rows_all; mx_all = std::max(mx_all, d);
        if (!(std::max(bf, be) >= L - 1.f)) continue;
        ++rows; mx = std::max(mx, d);
      }
    }
    // (a sample without any row near the threshold -- every decision far away -- still calibrates: over all rows, which only makes eps larger)
    if (rows == 0) { mx = mx_all; rows = rows_all; }
    if (!std::isfinite(mx)) return fail(ctx, GN_ERR_ARG, "gn_calibrate_certify: non-finite scores in the sample");
    if (rows == 0) return fail(ctx, GN_ERR_ARG, "gn_calibrate_certify: the sample holds no pair with at least two keypoints per side (nothing to measure)");
    mx_of[k] = (float)mx; eps_of[k] = std::max(floor_eps, safety * (float)mx);
  }
  if (setting == 0) {
    // the wider level's bound is never stated tighter than the narrower one's (a sample can happen to show the opposite)
    ctx->cert_eps_lvl[1] = eps_of[1]; ctx->cert_eps_lvl[0] = std::max(eps_of[0], eps_of[1]);
    ctx->auto_level = 3; ctx->auto_pairs = ctx->auto_wide = ctx->auto_narrow = 0;
  }
  const int rep = n_lv - 1;       // reported: the three-product level's values under the automatic setting (gn_get_ffn_level returns both eps)
  ctx->cert_eps = eps_of[rep];
  if (measured_host) *measured_host = mx_of[rep];
  if (eps_host) *eps_host = eps_of[rep];
  return GN_OK;
}

int gn_get_certify_stats(gn_ctx* ctx, int64_t* out8) {
  if (!ctx || !out8) return GN_ERR_ARG;
  out8[0] = ctx->cert_calls; out8[1] = ctx->cert_pairs; out8[2] = ctx->cert_flag_margin; out8[3] = ctx->cert_flag_range;
  out8[4] = ctx->cert_rerun; out8[5] = ctx->cert_f32_marginal; out8[6] = ctx->certify; out8[7] = 0;
  return GN_OK;
}

int gn_reset_certify_stats(gn_ctx* ctx) {
  if (!ctx) return GN_ERR_ARG;
  ctx->cert_calls = ctx->cert_pairs = ctx->cert_flag_margin = ctx->cert_flag_range = ctx->cert_rerun = ctx->cert_f32_marginal = 0;
  ctx->auto_calls_lvl[0] = ctx->auto_calls_lvl[1] = ctx->auto_switches = 0;
  return GN_OK;
}

int gn_get_uncertain(gn_ctx* ctx, int B, int32_t* host_flags, void* stream) {
  if (!ctx || !host_flags || B < 1 || B > ctx->max_batch) return GN_ERR_ARG;
  if (!ctx->certify) return fail(ctx, GN_ERR_ARG, "gn_get_uncertain: the certificate is off (gn_set_certify)");
  GN_HIP(hipSetDevice(ctx->device));
  GN_HIP(hipMemcpyAsync(ctx->uncert_host, ctx->uncert, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
  GN_HIP(hipStreamSynchronize((hipStream_t)stream));
  memcpy(host_flags, ctx->uncert_host, (size_t)B * sizeof(int32_t));
  return GN_OK;
}

int gn_gather_points(gn_ctx* ctx, int B, int kpt_format, const float* kpt_q, int stride_q, const float* kpt_r, int stride_r,
                     const int64_t* idx, const int32_t* n_match, const uint8_t* dem, int H, int W,
                     float* mkp_q, float* obj, void* stream) {
  if (!ctx || !kpt_q || !kpt_r || !idx || !n_match || !mkp_q || !obj || B < 1 || B > ctx->max_batch)
    return fail(ctx, GN_ERR_ARG, "bad gn_gather_points argument");
  if (dem && (H < 1 || W < 1)) return fail(ctx, GN_ERR_ARG, "bad DEM shape");
  GN_HIP(hipSetDevice(ctx->device));
  GatherArgs g;
  g.kpt_q = kpt_q; g.stride_q = stride_q; g.kpt_r = kpt_r; g.stride_r = stride_r; g.kpt_format = kpt_format & 0xff;
  g.idx = idx; g.n_match = n_match; g.kmax = ctx->npad; g.B = B; g.dem = dem; g.H = H; g.W = W; g.mkp_q = mkp_q; g.obj = obj;
  StageTimer tm(ctx, (hipStream_t)stream, ST_GATHER);
  launch_gather(g, (hipStream_t)stream);
  GN_HIP(hipGetLastError());
  return GN_OK;
}

int gn_pnp_ransac(gn_ctx* ctx, int B, const float* obj, const float* img, const int32_t* n_pts, int kstride,
                  const double* K9, int iterations_count, float reproj_error_px, double confidence, int min_pts,
                  double* R, double* t, int32_t* n_inliers, uint8_t* ok, void* stream) {
  if (!ctx || !obj || !img || !n_pts || !K9 || !R || !t || !n_inliers || !ok || B < 1 || B > ctx->max_batch)
    return fail(ctx, GN_ERR_ARG, "bad gn_pnp_ransac argument");
  if (kstride < 1 || kstride > ctx->npad) return fail(ctx, GN_ERR_ARG, "kstride exceeds max_kpts of this context");
  if (iterations_count < 1 || iterations_count > 16) return fail(ctx, GN_ERR_ARG, "iterations_count must be in 1..16 (PoseNode uses 10)");
  GN_HIP(hipSetDevice(ctx->device));
  PnpArgs a;
  a.obj = obj; a.img = img; a.n_pts = n_pts; a.kstride = kstride; a.B = B;
  a.fx = K9[0]; a.fy = K9[4]; a.cx = K9[2]; a.cy = K9[5];
  a.iterations = iterations_count; a.reproj = reproj_error_px; a.confidence = confidence; a.min_pts = min_pts;
  a.R = R; a.t = t; a.n_inliers = n_inliers; a.ok = ok; a.mask_ws = ctx->mask_ws; a.hyp = ctx->hyp_ws; a.pts_ws = ctx->pts_ws;
  a.dbg_ts = ctx->pnp_stamps ? reinterpret_cast<long long*>(ctx->sim) : nullptr;   // developer knob 15: phase stamps land in the (idle) sim buffer
  StageTimer tm(ctx, (hipStream_t)stream, ST_PNP);
  launch_pnp(a, (hipStream_t)stream);
  GN_HIP(hipGetLastError());
  return GN_OK;
}

namespace {
// every per-pair workspace pointer of the context moved by `b0` pairs (sign = +1) and back (sign = -1): kernels capture
// pointer values at launch, so a sub-batch group simply runs the ordinary path on its slice of the workspaces
void shift_workspaces(gn_ctx* c, long long b0, int sign) {
  const long long d = sign * b0, np = c->npad_run, T2 = 2 * np;
  auto mv = [&](auto*& p, long long per_pair) { if (p) p += d * per_pair; };
  mv(c->desc, T2 * kInDim); mv(c->cos_t, T2 * kFreq); mv(c->sin_t, T2 * kFreq); mv(c->rot4, T2 * 4); mv(c->extent, 4); mv(c->nvalid, 2);
  mv(c->x, T2 * kDim); mv(c->qkv, T2 * 3 * kDim); mv(c->ctx, T2 * kDim); mv(c->msg, T2 * kDim); mv(c->h, T2 * 2 * kDim);
  mv(c->md, T2 * kDim); mv(c->ls, T2); mv(c->sim, np * np);
  mv(c->desc_p, 2 * T2 * kInDim); mv(c->x_p, 2 * T2 * kDim); mv(c->ctx_p, 2 * T2 * kDim); mv(c->msg_p, 2 * T2 * kDim);
  mv(c->h_p, 2 * T2 * 2 * kDim); mv(c->md_p, 2 * T2 * kDim);
  mv(c->qkb, T2 * 2 * kDim); mv(c->vtb, T2 * kDim);
  mv(c->rowmax, np); mv(c->rowlog, np); mv(c->colmax, np); mv(c->collog, np); mv(c->max0, np); mv(c->m0, np); mv(c->m1, np);
  mv(c->cpart_m, (np / 32) * np); mv(c->cpart_s, (np / 32) * np); mv(c->cpart_i, (np / 32) * np); mv(c->rpart_a, 8 * np); mv(c->rpart_b, 8 * np); mv(c->tickets, 2);
  mv(c->max0b, np); mv(c->rpart_c, 8 * np); mv(c->uncert, 1); mv(c->uncert_alt, 1);
  // match lists and the PnP masks are strided by the context's padded maximum (gn_kmax), whatever the active size
  const long long km = c->npad;
  mv(c->e_idx, km * 2); mv(c->e_score, km); mv(c->e_mkp, km * 2); mv(c->e_obj, km * 3);
  mv(c->mask_ws, km * 16); mv(c->hyp_ws, 16); mv(c->pts_ws, km * 5);
  mv(c->lists, c->lists_stride);
}

int estimate_impl(gn_ctx* ctx, int B, int kpt_format,
                  const float* desc_q, const float* kpt_q, const int32_t* n_q, int stride_q,
                  const float* desc_r, const float* kpt_r, const int32_t* n_r, int stride_r,
                  const uint8_t* dem, int H, int W, const double* K9, int min_matches,
                  double* R, double* t, int32_t* n_match, int32_t* n_inliers, uint8_t* ok, void* stream);

// deferred certificate: wait for the flags of the pending call in `slot` (its kernels are behind at most one later call on the GPU), re-run
// its flagged pairs from the saved arguments on `s`
int cert_resolve(gn_ctx* ctx, int slot, hipStream_t s) {
  gn_ctx::CertPending& p = ctx->cert_pend[slot];
  if (!p.active) return GN_OK;
  p.active = false;
  GN_HIP(hipEventSynchronize(p.ev));
  const int kw = (p.kpt_format & 0xff) == GN_KPT_LAF ? 6 : (p.kpt_format & 0xff) == GN_KPT_RECORD ? kRecordFloats : 4;
  const int in_dim = ctx->feature == GN_FEATURE_SIFT ? kInDim : kDim;
  const int np_now = ctx->npad_run;
  ctx->npad_run = p.npad_run;
  const CertShape shp{p.stride_q, p.stride_r, p.H, p.W, kw, in_dim, (size_t)ctx->npad};
  const CertView view{p.desc_q, p.kpt_q, p.desc_r, p.kpt_r, p.n_q, p.n_r, p.dem, p.R, p.t, p.n_match, p.n_inliers, p.ok, nullptr, nullptr};
  const int rc = certify_rerun(ctx, p.B, s, shp, view, [&](const CertView& w, int n) {
    return estimate_impl(ctx, n, p.kpt_format, w.desc_q, w.kpt_q, w.n_q, p.stride_q, w.desc_r, w.kpt_r, w.n_r, p.stride_r, w.dem, p.H, p.W, p.K9, p.min_matches,
                         w.R, w.t, w.n_match, w.n_inliers, w.ok, s);
  }, p.flags, p.level);
  ctx->npad_run = np_now;
  return rc;
}
}  // namespace

int gn_estimate(gn_ctx* ctx, int B, int kpt_format,
                const float* desc_q, const float* kpt_q, const int32_t* n_q, int stride_q,
                const float* desc_r, const float* kpt_r, const int32_t* n_r, int stride_r,
                const uint8_t* dem, int H, int W, const double* K9, int min_matches,
                double* R, double* t, int32_t* n_match, int32_t* n_inliers, uint8_t* ok, void* stream) {
  if (!ctx) return fail(nullptr, GN_ERR_ARG, "null context");
  if (ctx->fused_proj_pending && ctx->npad > 0 && gn_missing_tensors(ctx) == 0) {      // (the sub-batch groups below skip it: their workspace pointers are shifted)
    GN_HIP(hipSetDevice(ctx->device));
    int rcs = ensure_composed(ctx);
    if (rcs == GN_OK) rcs = selfcheck_fused_projection(ctx);
    if (rcs != GN_OK) return rcs;
  }
  const int groups = std::min(ctx->n_sub, B);
  if (groups <= 1 || ctx->overlap)
    return estimate_impl(ctx, B, kpt_format, desc_q, kpt_q, n_q, stride_q, desc_r, kpt_r, n_r, stride_r, dem, H, W, K9, min_matches,
                         R, t, n_match, n_inliers, ok, stream);
  if (ctx->npad <= 0) return fail(ctx, GN_ERR_ARG, "context has no workspaces (a gn_resize failed): call gn_resize again");
  if (B < 1 || B > ctx->max_batch) return fail(ctx, GN_ERR_ARG, "B out of range for this context");
  GN_HIP(hipSetDevice(ctx->device));
  hipStream_t s = (hipStream_t)stream;
  // deferred joins still pending from a call of another shape: group g of this call would overlap the workspace slice of a
  // different group of that call on another stream -> join them first (the fork event below is recorded behind the joins)
  if (ctx->defer_join && (ctx->sub_last_B != B || ctx->sub_last_np != ctx->npad_run)) {
    const int rcj = gn_flush(ctx, stream);
    if (rcj != GN_OK) return rcj;
  }
  ctx->sub_last_B = B; ctx->sub_last_np = ctx->npad_run;
  ctx->ovf_groups_last = groups;
  GN_HIP(hipEventRecord(ctx->ev_fork, s));
  const int kw = (kpt_format & 0xff) == GN_KPT_LAF ? 6 : (kpt_format & 0xff) == GN_KPT_RECORD ? kRecordFloats : 4;
  const int in_dim = ctx->feature == GN_FEATURE_SIFT ? kInDim : kDim;
  int rc_all = GN_OK, b0 = 0;
  for (int g = 0; g < groups; ++g) {
    const int Bg = B / groups + (g < B % groups ? 1 : 0);
    hipStream_t sg = ctx->sub_serial ? ctx->sub_s[0] : ctx->sub_s[g];
    GN_HIP(hipStreamWaitEvent(sg, ctx->ev_fork, 0));
    shift_workspaces(ctx, b0, +1);
    ctx->in_group = true; ctx->ovf = ctx->ovf_base + g;            // this group's own guard word: cleared, raised and read on this group's stream only
    const int rc = estimate_impl(ctx, Bg, kpt_format,
                                 desc_q ? desc_q + (size_t)b0 * stride_q * in_dim : nullptr, kpt_q + (size_t)b0 * stride_q * kw, n_q + b0, stride_q,
                                 desc_r ? desc_r + (size_t)b0 * stride_r * in_dim : nullptr, kpt_r + (size_t)b0 * stride_r * kw, n_r + b0, stride_r,
                                 dem ? dem + (size_t)b0 * H * W : nullptr, H, W, K9, min_matches,
                                 R + (size_t)b0 * 9, t + (size_t)b0 * 3, n_match + b0, n_inliers + b0, ok + b0, sg);
    shift_workspaces(ctx, b0, -1);
    ctx->ovf = ctx->ovf_base; ctx->in_group = false;
    if (rc != GN_OK && rc_all == GN_OK) rc_all = rc;
    GN_HIP(hipEventRecord(ctx->ev_join[g], sg));
    ctx->sub_pending[g] = true;
    b0 += Bg;
  }
  // default: the caller's stream continues only after every group is done -- inputs may be released and outputs read in
  // stream order, like any other call.  gn_set_deferred_join(1) leaves the join to gn_flush (consecutive calls then pipeline
  // inside each group's stream; the caller keeps the INPUT buffers alive until it has flushed).
  if (ctx->certify == 3 && rc_all == GN_OK) {
    // deferred certificate: join the groups (stream order only), leave this call's flags in a pinned slot behind an event, and resolve the
    // PREVIOUS call now that this one is queued -- the host waits for call n while the GPU already works on call n + 1
    for (int g = 0; g < 8; ++g)
      if (ctx->sub_pending[g]) { GN_HIP(hipStreamWaitEvent(s, ctx->ev_join[g], 0)); ctx->sub_pending[g] = false; }
    const int slot = ctx->cert_slot;
    gn_ctx::CertPending& p = ctx->cert_pend[slot];
    if (p.active) { const int rcr = cert_resolve(ctx, slot, s); if (rcr != GN_OK) return rcr; }     // (cannot happen in the alternating order; kept for safety)
    GN_HIP(hipMemcpyAsync(p.flags, ctx->uncert, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (ffn_auto(ctx)) GN_HIP(hipMemcpyAsync(p.flags + ctx->max_batch, ctx->uncert_alt, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    else memset(p.flags + ctx->max_batch, 0, (size_t)B * sizeof(int32_t));
    GN_HIP(hipEventRecord(p.ev, s));
    p.active = true; p.level = ffn_auto(ctx) ? ctx->auto_level : -1; p.B = B; p.kpt_format = kpt_format; p.stride_q = stride_q; p.stride_r = stride_r; p.H = H; p.W = W; p.min_matches = min_matches;
    p.npad_run = ctx->npad_run; p.desc_q = desc_q; p.kpt_q = kpt_q; p.n_q = n_q; p.desc_r = desc_r; p.kpt_r = kpt_r; p.n_r = n_r; p.dem = dem;
    memcpy(p.K9, K9, sizeof p.K9); p.R = R; p.t = t; p.n_match = n_match; p.n_inliers = n_inliers; p.ok = ok;
    ctx->cert_slot ^= 1;
    return cert_resolve(ctx, slot ^ 1, s);
  }
  if (!ctx->defer_join || ctx->certify == 2) { const int rcj = gn_flush(ctx, stream); if (rcj != GN_OK && rc_all == GN_OK) rc_all = rcj; }
  if (ctx->certify == 2 && rc_all == GN_OK) {
    // the groups are joined: one read-back of the call's per-pair flags, then matcher + gather + PnP of the flagged pairs again in exact f32
    const CertShape shp{stride_q, stride_r, H, W, kw, in_dim, (size_t)ctx->npad};
    const CertView view{desc_q, kpt_q, desc_r, kpt_r, n_q, n_r, dem, R, t, n_match, n_inliers, ok, nullptr, nullptr};
    rc_all = certify_rerun(ctx, B, s, shp, view, [&](const CertView& w, int n) {
      return estimate_impl(ctx, n, kpt_format, w.desc_q, w.kpt_q, w.n_q, stride_q, w.desc_r, w.kpt_r, w.n_r, stride_r, w.dem, H, W, K9, min_matches,
                           w.R, w.t, w.n_match, w.n_inliers, w.ok, s);
    });
  }
  return rc_all;
}

int gn_set_deferred_join(gn_ctx* ctx, int enable) {
  if (!ctx) return GN_ERR_ARG;
  ctx->defer_join = enable ? 1 : 0;
  return GN_OK;
}

int gn_set_active_kpts(gn_ctx* ctx, int max_kpts_per_side) {
  if (!ctx || max_kpts_per_side < 1) return GN_ERR_ARG;
  ctx->npad_run = std::min(ctx->npad, ((max_kpts_per_side + 127) / 128) * 128);
  return ctx->npad_run;
}

int gn_set_substreams(gn_ctx* ctx, int n) {
  if (!ctx || n < 1 || n > 8) return GN_ERR_ARG;
  GN_HIP(hipSetDevice(ctx->device));
  if (n > 1 && !ctx->ev_fork) GN_HIP(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
  for (int i = 0; i < n && n > 1; ++i)
    if (!ctx->sub_s[i]) {
      if (ctx->cu_mask_mode != 0) {
        // developer experiment (knob 29): every sub-batch stream on its own share of the CUs.  bit b of the mask = CU b of the device's enumeration:
        // 1 = contiguous shares, 2 = interleaved (b % n), 3 = by XCD assuming b % 8 is the XCD (share g gets XCDs g, g + n, ...)
        uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int b = 0; b < 256; ++b) {
          const int share = ctx->cu_mask_mode == 1 ? b * n / 256 : ctx->cu_mask_mode == 2 ? b % n : (b % 8) % n;
          if (share == i) mask[b >> 5] |= 1u << (b & 31);
        }
        GN_HIP(hipExtStreamCreateWithCUMask(&ctx->sub_s[i], 8, mask));
      } else {
        GN_HIP(hipStreamCreateWithFlags(&ctx->sub_s[i], hipStreamNonBlocking));
      }
      GN_HIP(hipEventCreateWithFlags(&ctx->ev_join[i], hipEventDisableTiming));
    }
  ctx->n_sub = n;
  return GN_OK;
}

namespace {
int estimate_impl(gn_ctx* ctx, int B, int kpt_format,
                  const float* desc_q, const float* kpt_q, const int32_t* n_q, int stride_q,
                  const float* desc_r, const float* kpt_r, const int32_t* n_r, int stride_r,
                  const uint8_t* dem, int H, int W, const double* K9, int min_matches,
                  double* R, double* t, int32_t* n_match, int32_t* n_inliers, uint8_t* ok, void* stream) {
  if (!ctx->overlap) {
    int rc = gn_match(ctx, B, kpt_format, desc_q, kpt_q, n_q, stride_q, desc_r, kpt_r, n_r, stride_r,
                      ctx->e_idx, ctx->e_score, n_match, stream);
    if (rc != GN_OK) return rc;
    rc = gn_gather_points(ctx, B, kpt_format, kpt_q, stride_q, kpt_r, stride_r, ctx->e_idx, n_match, dem, H, W,
                          ctx->e_mkp, ctx->e_obj, stream);
    if (rc != GN_OK) return rc;
    return gn_pnp_ransac(ctx, B, ctx->e_obj, ctx->e_mkp, n_match, ctx->npad, K9, 10, 8.0f, 0.99, min_matches,
                         R, t, n_inliers, ok, stream);
  }
  // Overlapped pose stage: the latency-bound PnP kernels (a few hundred single-wave workgroups) of this call run on an
  // internal stream while the caller's stream is free to start the next call's matcher.  The PnP inputs are
  // double-buffered; R / t / n_inliers / ok of this call are complete once gn_flush() has been ordered behind it.
  hipStream_t s = (hipStream_t)stream;
  const int slot = (int)(ctx->calls++ & 1);
  GN_HIP(hipSetDevice(ctx->device));
  if (ctx->pnp_pending[slot]) GN_HIP(hipStreamWaitEvent(s, ctx->ev_pnp[slot], 0));   // the PnP that last read this slot
  int rc = gn_match(ctx, B, kpt_format, desc_q, kpt_q, n_q, stride_q, desc_r, kpt_r, n_r, stride_r,
                    ctx->e_idx, ctx->e_score, n_match, stream);
  if (rc != GN_OK) return rc;
  rc = gn_gather_points(ctx, B, kpt_format, kpt_q, stride_q, kpt_r, stride_r, ctx->e_idx, n_match, dem, H, W,
                        ctx->o_mkp[slot], ctx->o_obj[slot], stream);
  if (rc != GN_OK) return rc;
  GN_HIP(hipMemcpyAsync(ctx->o_nmatch[slot], n_match, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
  GN_HIP(hipEventRecord(ctx->ev_gather[slot], s));
  GN_HIP(hipStreamWaitEvent(ctx->s_pnp, ctx->ev_gather[slot], 0));
  rc = gn_pnp_ransac(ctx, B, ctx->o_obj[slot], ctx->o_mkp[slot], ctx->o_nmatch[slot], ctx->npad, K9, 10, 8.0f, 0.99, min_matches,
                     R, t, n_inliers, ok, ctx->s_pnp);
  if (rc != GN_OK) return rc;
  GN_HIP(hipEventRecord(ctx->ev_pnp[slot], ctx->s_pnp));
  ctx->pnp_pending[slot] = true;
  return GN_OK;
}

}  // namespace

int gn_set_overlap(gn_ctx* ctx, int enable) {
  if (!ctx) return GN_ERR_ARG;
  GN_HIP(hipSetDevice(ctx->device));
  if (enable && !ctx->s_pnp) {
    GN_HIP(hipStreamCreateWithFlags(&ctx->s_pnp, hipStreamNonBlocking));
    const size_t B = ctx->max_batch, np = ctx->npad;
    for (int i = 0; i < 2; ++i) {
      GN_HIP(hipEventCreateWithFlags(&ctx->ev_gather[i], hipEventDisableTiming));
      GN_HIP(hipEventCreateWithFlags(&ctx->ev_pnp[i], hipEventDisableTiming));
      int rc = ws_alloc(ctx, &ctx->o_mkp[i], B * np * 2); if (rc != GN_OK) return rc;
      rc = ws_alloc(ctx, &ctx->o_obj[i], B * np * 3); if (rc != GN_OK) return rc;
      rc = ws_alloc(ctx, &ctx->o_nmatch[i], B); if (rc != GN_OK) return rc;
    }
  }
  if (!enable && ctx->s_pnp) GN_HIP(hipStreamSynchronize(ctx->s_pnp));
  ctx->overlap = enable ? 1 : 0;
  return GN_OK;
}

int gn_flush(gn_ctx* ctx, void* stream) {
  if (!ctx) return GN_ERR_ARG;
  GN_HIP(hipSetDevice(ctx->device));
  for (int i = 0; i < 2; ++i)
    if (ctx->pnp_pending[i]) { GN_HIP(hipStreamWaitEvent((hipStream_t)stream, ctx->ev_pnp[i], 0)); ctx->pnp_pending[i] = false; }
  for (int g = 0; g < 8; ++g)
    if (ctx->sub_pending[g]) { GN_HIP(hipStreamWaitEvent((hipStream_t)stream, ctx->ev_join[g], 0)); ctx->sub_pending[g] = false; }
  if (!ctx->cert_inner)
    for (int k = 0; k < 2; ++k) {       // deferred certificates still open: the older one first
      const int rc = cert_resolve(ctx, ctx->cert_slot ^ k, (hipStream_t)stream);
      if (rc != GN_OK) return rc;
    }
  return GN_OK;
}

// TwistNode's matcher for B frame pairs: BFMatcher(L2).knnMatch(k=2) + ratio test (twist_node.py:248-267)
int gn_vo_match(gn_ctx* ctx, int B, const float* desc_q, const int32_t* n_q, int stride_q,
                const float* desc_r, const int32_t* n_r, int stride_r, double ratio,
                int64_t* idx, float* dist, int32_t* n_good, int32_t* nn_idx, float* nn_dist, void* stream) {
  if (!ctx) return fail(nullptr, GN_ERR_ARG, "null context");
  if (ctx->npad <= 0) return fail(ctx, GN_ERR_ARG, "context has no workspaces (a gn_resize failed): call gn_resize again");
  if (B < 1 || B > ctx->max_batch) return fail(ctx, GN_ERR_ARG, "B out of range for this context");
  if (stride_q < 1 || stride_r < 1 || stride_q > ctx->npad || stride_r > ctx->npad)
    return fail(ctx, GN_ERR_ARG, "keypoint stride exceeds max_kpts of this context");
  if (!desc_q || !n_q || !desc_r || !n_r || !idx || !dist || !n_good) return fail(ctx, GN_ERR_ARG, "null argument");
  GN_HIP(hipSetDevice(ctx->device));
  { const int rc = ensure_sim(ctx); if (rc != GN_OK) return rc; }   // q . r panel of the brute-force matcher (first call allocates it)
  hipStream_t s = (hipStream_t)stream;
  const int np = ctx->npad;
  VoArgs v;
  v.desc_q = desc_q; v.n_q = n_q; v.stride_q = stride_q; v.desc_r = desc_r; v.n_r = n_r; v.stride_r = stride_r;
  v.B = B; v.npad = np; v.ratio = ratio;
  v.desc = ctx->desc; v.norm2 = ctx->vo_norm2; v.nvalid = ctx->nvalid; v.sim = ctx->sim;
  v.nn_idx = ctx->vo_nn_idx; v.nn_dist = ctx->vo_nn_dist; v.good = ctx->vo_good;
  v.idx = idx; v.dist = dist; v.n_good = n_good; v.kmax = np;
  launch_vo_pack(v, s);
  // q.r for every (query, train) pair on the exact-f32 MFMA GEMM, whatever the context's precision mode
  GemmArgs gs;
  memset(&gs, 0, sizeof gs);
  gs.A = ctx->desc; gs.lda = kInDim; gs.K1 = kInDim; gs.W = ctx->desc + (size_t)np * kInDim; gs.ldw = kInDim;
  gs.Y = ctx->sim; gs.ldy = np; gs.M = np; gs.N = np; gs.K = kInDim; gs.acc_scale = 1.f;
  gs.strideA = gs.strideW = 2LL * np * kInDim; gs.strideY = (long long)np * np;
  const int saved = gn::g_gemm_variant;
  gn::g_gemm_variant = 3;
  launch_gemm_f32(EPI_PLAIN, gs, B, s);
  gn::g_gemm_variant = saved;
  launch_vo_knn2(v, s);
  if (nn_idx) GN_HIP(hipMemcpyAsync(nn_idx, ctx->vo_nn_idx, (size_t)B * np * 2 * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
  if (nn_dist) GN_HIP(hipMemcpyAsync(nn_dist, ctx->vo_nn_dist, (size_t)B * np * 2 * sizeof(float), hipMemcpyDeviceToDevice, s));
  GN_HIP(hipGetLastError());
  return GN_OK;
}

// TwistNode._pose lines 227-289 for B frame pairs: 2-NN match -> ratio test -> MIN_MATCHES gate -> compute_pose with
// a zero elevation raster (planar PnP)
int gn_vo_estimate(gn_ctx* ctx, int B, int kpt_format,
                   const float* desc_q, const float* kpt_q, const int32_t* n_q, int stride_q,
                   const float* desc_r, const float* kpt_r, const int32_t* n_r, int stride_r,
                   const double* K9, double ratio, int min_matches,
                   double* R, double* t, int32_t* n_match, int32_t* n_inliers, uint8_t* ok, void* stream) {
  if (!ctx) return fail(nullptr, GN_ERR_ARG, "null context");
  int rc = gn_vo_match(ctx, B, desc_q, n_q, stride_q, desc_r, n_r, stride_r, ratio, ctx->e_idx, ctx->e_score, n_match,
                       nullptr, nullptr, stream);
  if (rc != GN_OK) return rc;
  rc = gn_gather_points(ctx, B, kpt_format, kpt_q, stride_q, kpt_r, stride_r, ctx->e_idx, n_match, nullptr, 0, 0,
                        ctx->e_mkp, ctx->e_obj, stream);
  if (rc != GN_OK) return rc;
  return gn_pnp_ransac(ctx, B, ctx->e_obj, ctx->e_mkp, n_match, ctx->npad, K9, 10, 8.0f, 0.99, min_matches,
                       R, t, n_inliers, ok, stream);
}

namespace {
// cv::getRotationMatrix2D + the in-place inversion at the top of cv::warpAffine (f64, same operation order), and the
// matrix StereoNode._rotate_and_crop_center returns (inverse of the rotation, times the crop translation)
void rotate_crop_matrices(int H, int W, double angle_degrees, int crop_h, int crop_w, double Minv6[6], int* dx, int* dy, double back9[9]) {
  const int cxi = W / 2, cyi = H / 2;
  const double a = angle_degrees * (3.14159265358979323846 / 180.0);
  const double alpha = std::cos(a), beta = std::sin(a);
  const double cx = (double)(float)cxi, cy = (double)(float)cyi;
  const double m[6] = {alpha, beta, (1 - alpha) * cx - beta * cy, -beta, alpha, beta * cx + (1 - alpha) * cy};
  double M[6] = {m[0], m[1], m[2], m[3], m[4], m[5]};
  double D = M[0] * M[4] - M[1] * M[3];
  D = D != 0 ? 1. / D : 0;
  const double A11 = M[4] * D, A22 = M[0] * D;
  M[0] = A11; M[1] *= -D; M[3] *= -D; M[4] = A22;
  const double b1 = -M[0] * M[2] - M[1] * M[5];
  const double b2 = -M[3] * M[2] - M[4] * M[5];
  M[2] = b1; M[5] = b2;
  for (int i = 0; i < 6; ++i) Minv6[i] = M[i];
  *dx = cxi - crop_w / 2; *dy = cyi - crop_h / 2;
  if (back9) {   // inv([[m]; 0 0 1]) @ [[1 0 dx], [0 1 dy], [0 0 1]]  (adjugate form of the 3x3 inverse)
    const double det = m[0] * m[4] - m[1] * m[3];
    const double i00 = m[4] / det, i01 = -m[1] / det, i02 = (m[1] * m[5] - m[2] * m[4]) / det;
    const double i10 = -m[3] / det, i11 = m[0] / det, i12 = (m[2] * m[3] - m[0] * m[5]) / det;
    back9[0] = i00; back9[1] = i01; back9[2] = i00 * *dx + i01 * *dy + i02;
    back9[3] = i10; back9[4] = i11; back9[5] = i10 * *dx + i11 * *dy + i12;
    back9[6] = 0; back9[7] = 0; back9[8] = 1;
  }
}
}  // namespace

// StereoNode._rotate_and_crop_center on a 2-channel u8 stack (stereo_node.py:292-335)
int gn_rotate_crop_center(gn_ctx* ctx, const uint8_t* stack, int H, int W, double angle_degrees, int crop_h, int crop_w,
                          uint8_t* out_stack, double* back9_host, void* stream) {
  if (!ctx || !stack || !out_stack || H < 1 || W < 1 || crop_h < 1 || crop_w < 1 || crop_h > H || crop_w > W)
    return fail(ctx, GN_ERR_ARG, "bad gn_rotate_crop_center argument");
  GN_HIP(hipSetDevice(ctx->device));
  WarpArgs a;
  a.src0 = stack; a.src1 = nullptr; a.H = H; a.W = W; a.crop_h = crop_h; a.crop_w = crop_w; a.out0 = out_stack; a.out1 = nullptr;
  rotate_crop_matrices(H, W, angle_degrees, crop_h, crop_w, a.M, &a.dx, &a.dy, back9_host);
  launch_rotate_crop(a, false, (hipStream_t)stream);
  GN_HIP(hipGetLastError());
  return GN_OK;
}

// stereo_node.py:229-262 in one pass: BGR -> gray, stack with the DEM, rotate, crop -> reference raster + DEM raster
int gn_stereo_reference(gn_ctx* ctx, const uint8_t* bgr, const uint8_t* dem, int H, int W, double angle_degrees,
                        int crop_h, int crop_w, uint8_t* out_ref, uint8_t* out_dem, double* back9_host, void* stream) {
  if (!ctx || !bgr || !dem || !out_ref || !out_dem || H < 1 || W < 1 || crop_h < 1 || crop_w < 1 || crop_h > H || crop_w > W)
    return fail(ctx, GN_ERR_ARG, "bad gn_stereo_reference argument");
  GN_HIP(hipSetDevice(ctx->device));
  WarpArgs a;
  a.src0 = bgr; a.src1 = dem; a.H = H; a.W = W; a.crop_h = crop_h; a.crop_w = crop_w; a.out0 = out_ref; a.out1 = out_dem;
  rotate_crop_matrices(H, W, angle_degrees, crop_h, crop_w, a.M, &a.dx, &a.dy, back9_host);
  launch_rotate_crop(a, true, (hipStream_t)stream);
  GN_HIP(hipGetLastError());
  return GN_OK;
}

namespace {
int sift_prepare(gn_ctx* ctx, int B, int H, int W, int max_kp) {
  if (ctx->sift_h == H && ctx->sift_w == W && ctx->sift_max_kp >= max_kp && ctx->sift_batch >= B) return GN_OK;
  B = std::max(B, ctx->sift_h == H && ctx->sift_w == W ? ctx->sift_batch : 1);
  for (void* p : ctx->sift_allocs) hipFree(p);
  ctx->sift_allocs.clear();
  ctx->sift_h = ctx->sift_w = 0; ctx->sift_batch = 0;
  auto alloc = [&](size_t bytes) -> void* { void* p = nullptr; if (hipMalloc(&p, bytes) != hipSuccess) return nullptr; ctx->sift_allocs.push_back(p); return p; };
  const int bw = 2 * W, bh = 2 * H;
  int n_oct = (int)std::nearbyint(std::log((double)std::min(bw, bh)) / std::log(2.0) - 2) + 1;
  if (n_oct < 1) n_oct = 1;
  if (n_oct > kSiftMaxOctaves) n_oct = kSiftMaxOctaves;
  SiftPyramid& py = ctx->sift_py;
  py.n_oct = n_oct;
  int w = bw, h = bh;
  for (int o = 0; o < n_oct; ++o) {
    py.oct[o].w = w; py.oct[o].h = h;
    const size_t n = (size_t)w * h;
    py.oct[o].stride = (long long)(11 * n);            // per image: 6 Gaussian + 5 DoG levels, images back to back
    float* block = (float*)alloc((size_t)B * 11 * n * sizeof(float));
    if (!block) return fail(ctx, GN_ERR_HIP, "SIFT pyramid allocation failed");
    for (int i = 0; i < 6; ++i) py.oct[o].gauss[i] = block + i * n;
    for (int i = 0; i < 5; ++i) py.oct[o].dog[i] = block + (6 + i) * n;
    w /= 2; h /= 2;
    if (w < 1 || h < 1) { py.n_oct = o + 1; break; }
  }
  ctx->sift_tmp = (float*)alloc((size_t)bw * bh * sizeof(float));
  // blur kernels: sigma differences of createInitialImage / buildGaussianPyramid (f64 on the host, libm exp)
  const double sigma = 1.6;
  std::vector<double> sig(6);
  ctx->sift_kernels.assign(6, {});
  { const float sf = (float)sigma; sift_gaussian_kernel((double)sqrtf(std::max(sf * sf - 0.5f * 0.5f * 4, 0.01f)), ctx->sift_kernels[0]); }   // createInitialImage: float arithmetic
  const double k = std::pow(2.0, 1.0 / 3.0);
  for (int i = 1; i < 6; ++i) {
    const double sp = std::pow(k, (double)(i - 1)) * sigma, st = sp * k;
    sift_gaussian_kernel(std::sqrt(st * st - sp * sp), ctx->sift_kernels[i]);
  }
  size_t tot = 0; ctx->sift_koff.assign(6, 0);
  for (int i = 0; i < 6; ++i) { ctx->sift_koff[i] = (int)tot; tot += ctx->sift_kernels[i].size(); }
  ctx->sift_dk = (float*)alloc(tot * sizeof(float));
  for (int i = 0; i < 6; ++i)
    if (hipMemcpy(ctx->sift_dk + ctx->sift_koff[i], ctx->sift_kernels[i].data(), ctx->sift_kernels[i].size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
      return fail(ctx, GN_ERR_HIP, "SIFT kernel upload failed");
  ctx->sift_max_cand = std::max(65536, 16 * max_kp);
  ctx->sift_max_kp = std::max(max_kp, 1024);
  ctx->sift_cand = (int4*)alloc((size_t)B * ctx->sift_max_cand * sizeof(int4));
  ctx->sift_counts = (int*)alloc((size_t)B * 4 * sizeof(int));
  ctx->sift_raw_cap = 16384;                                   // raw keypoints (several orientations per extremum, duplicates) before the sort
  while (ctx->sift_raw_cap < 8 * ctx->sift_max_kp) ctx->sift_raw_cap <<= 1;
  ctx->sift_kp_stride = (long long)2 * ctx->sift_raw_cap + ctx->sift_max_kp;                        // per image: raw | sorted | final
  ctx->sift_kp = (SiftKeypoint*)alloc((size_t)B * ctx->sift_kp_stride * sizeof(SiftKeypoint));
  if (!ctx->sift_tmp || !ctx->sift_dk || !ctx->sift_cand || !ctx->sift_counts || !ctx->sift_kp)
    return fail(ctx, GN_ERR_HIP, "SIFT workspace allocation failed");
  ctx->sift_h = H; ctx->sift_w = W; ctx->sift_batch = B;
  return GN_OK;
}
}  // namespace

// cv2.SIFT_create().detectAndCompute(gray, None): pose_node.py:122,230-232, twist_node.py:93,227-245 -- for a batch of
// equally sized images in one pass (every launch covers all images; the reference extracts one image per message)
int gn_sift_detect_and_compute_batch(gn_ctx* ctx, const uint8_t* gray, int B, int H, int W, int max_kpts,
                                     float* kpt_xysa, float* response, int32_t* octave, float* desc, int32_t* n_out_host, void* stream) {
  if (!ctx || !gray || !kpt_xysa || !desc || !n_out_host || B < 1 || B > 4096 || H < 16 || W < 16 || max_kpts < 1)
    return fail(ctx, GN_ERR_ARG, "bad gn_sift_detect_and_compute argument");
  GN_HIP(hipSetDevice(ctx->device));
  int rc = sift_prepare(ctx, B, H, W, max_kpts);
  if (rc != GN_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  SiftPyramid& py = ctx->sift_py;
  auto blur = [&](int o_in, const float* in, int o_out, float* out, int ki, float* dog = nullptr, int in_step = 1, float* half_scratch = nullptr) {
    sift_blur(B, py.oct[o_in].stride, py.oct[o_out].stride, in, ctx->sift_tmp, out, py.oct[o_out].w, py.oct[o_out].h, ctx->sift_dk + ctx->sift_koff[ki],
              (int)ctx->sift_kernels[ki].size(), s, dog, in_step, py.oct[o_in].w, half_scratch);
  };
  // createInitialImage: 2x bilinear, blur to sigma 1.6; then the Gaussian and DoG pyramids
  sift_base_blur(gray, B, H, W, py.oct[0].gauss[5], ctx->sift_tmp, py.oct[0].gauss[0], py.oct[0].stride, ctx->sift_dk + ctx->sift_koff[0],
                 (int)ctx->sift_kernels[0].size(), ctx->sift_counts, s);   // (scratch = level 5, overwritten later); also zeroes the counters
  int ksize[6];
  for (int i = 0; i < 6; ++i) ksize[i] = (int)ctx->sift_kernels[i].size();
  const int o_tail = sift_tail_first(py, ksize);                     // octaves from here on: one single-workgroup launch per image
  for (int o = 0; o < o_tail; ++o) {
    const SiftOctave& oc = py.oct[o];
    // level 1 of octave o > 0 samples level 3 of the octave above directly (its level 0 is never stored)
    if (o > 0) blur(o - 1, py.oct[o - 1].gauss[3], o, oc.gauss[1], 1, oc.dog[0], 2, oc.gauss[0]);
    for (int i = o > 0 ? 2 : 1; i < 6; ++i) blur(o, oc.gauss[i - 1], o, oc.gauss[i], i, oc.dog[i - 1]);   // row + column pass + DoG level in one launch
  }
  sift_tail(py, B, o_tail, ctx->sift_dk, ctx->sift_koff.data(), ksize, s);
  const float threshold = (float)(int)std::floor(0.5 * 0.04 / 3 * 255);
  sift_find(py, B, threshold, ctx->sift_cand, ctx->sift_counts, ctx->sift_max_cand, s);
  // raw keypoints, their sorted copy and the final list of an image live back to back in one allocation
  const int max_raw = ctx->sift_raw_cap;
  sift_refine(py, B, ctx->sift_cand, ctx->sift_counts, ctx->sift_max_cand, ctx->sift_kp, ctx->sift_kp_stride, max_raw, s);
  // sort / de-duplicate / rescale on the device, then descriptors for the final keypoints; one sync at the very end
  const int max_out = std::min(max_kpts, ctx->sift_max_kp);
  sift_sort_dedup(B, ctx->sift_kp, ctx->sift_kp_stride, ctx->sift_counts, max_raw, max_out, kpt_xysa, response, octave, max_kpts, s);
  sift_descriptors(py, B, ctx->sift_kp + 2 * max_raw, ctx->sift_kp_stride, ctx->sift_counts, max_out, desc, max_kpts, s);
  std::vector<int> counts_v;
  int* counts = reinterpret_cast<int*>(ctx->ovf_host + 16);       // pinned (every per-message transfer of the library is: gisnav_amd/upload.py has the reason)
  if (B > 1024) { counts_v.assign((size_t)B * 4, 0); counts = counts_v.data(); }
  GN_HIP(hipMemcpyAsync(counts, ctx->sift_counts, (size_t)B * 4 * sizeof(int), hipMemcpyDeviceToHost, s));
  GN_HIP(hipStreamSynchronize(s));
  for (int b = 0; b < B; ++b) {
    if (counts[4 * b] > ctx->sift_max_cand) return fail(ctx, GN_ERR_ARG, "SIFT candidate buffer overflow (raise max_kpts)");
    if (counts[4 * b + 1] > max_raw) return fail(ctx, GN_ERR_ARG, "SIFT keypoint buffer overflow (raise max_kpts)");
    n_out_host[b] = counts[4 * b + 2];     // <= max_out: beyond it the strongest max_out by response were kept (k_sift_dedup_emit)
  }
  ctx->sift_totals.assign((size_t)B, 0);
  for (int b = 0; b < B; ++b) ctx->sift_totals[b] = counts[4 * b + 3];
  GN_HIP(hipGetLastError());
  return GN_OK;
}

int gn_sift_detect_and_compute(gn_ctx* ctx, const uint8_t* gray, int H, int W, int max_kpts,
                               float* kpt_xysa, float* response, int32_t* octave, float* desc, int32_t* n_out_host, void* stream) {
  return gn_sift_detect_and_compute_batch(ctx, gray, 1, H, W, max_kpts, kpt_xysa, response, octave, desc, n_out_host, stream);
}

int gn_sift_last_totals(gn_ctx* ctx, int B, int32_t* totals_host) {
  if (!ctx || !totals_host || B < 1 || (size_t)B > ctx->sift_totals.size()) return fail(ctx, GN_ERR_ARG, "gn_sift_last_totals: no SIFT call of that batch size yet");
  for (int b = 0; b < B; ++b) totals_host[b] = ctx->sift_totals[b];
  return GN_OK;
}

// ---- SuperPoint extractor -----------------------------------------------------------------------------------------------
namespace {
const char* const kSpNames[12] = {
    "encoder.conv_blocks.0.conv_a", "encoder.conv_blocks.0.conv_b", "encoder.conv_blocks.1.conv_a", "encoder.conv_blocks.1.conv_b",
    "encoder.conv_blocks.2.conv_a", "encoder.conv_blocks.2.conv_b", "encoder.conv_blocks.3.conv_a", "encoder.conv_blocks.3.conv_b",
    "keypoint_decoder.conv_score_a", "keypoint_decoder.conv_score_b", "descriptor_decoder.conv_descriptor_a", "descriptor_decoder.conv_descriptor_b"};
const int kSpShape[12][3] = {{64, 1, 9}, {64, 64, 9}, {64, 64, 9}, {64, 64, 9}, {128, 64, 9}, {128, 128, 9}, {128, 128, 9}, {128, 128, 9},
                             {256, 128, 9}, {65, 256, 1}, {256, 128, 9}, {256, 256, 1}};   // Cout, Cin, taps
}  // namespace

int gn_sp_load_tensor(gn_ctx* ctx, const char* name_c, const float* host, const int64_t* shape, int ndim) {
  if (!ctx || !name_c || !host || !shape) return fail(ctx, GN_ERR_ARG, "bad gn_sp_load_tensor argument");
  GN_HIP(hipSetDevice(ctx->device));
  const std::string name(name_c);
  for (int i = 0; i < 12; ++i) {
    const std::string base(kSpNames[i]);
    const int cout = kSpShape[i][0], cin = kSpShape[i][1], taps = kSpShape[i][2];
    const int k = taps == 9 ? 3 : 1;
    gn_ctx::SpConv& c = ctx->sp[i];
    c.cout = cout; c.cin = cin; c.taps = taps; c.cout_pad = ((cout + 63) / 64) * 64;
    if (name == base + ".weight") {
      if (ndim != 4 || shape[0] != cout || shape[1] != cin || shape[2] != k || shape[3] != k) return fail(ctx, GN_ERR_SHAPE, "shape mismatch for " + name);
      std::vector<float> frag;
      const float* src = host;
      if (i > 0) { frag.resize((size_t)c.cout_pad * taps * cin); sp_weight_fragments(host, cout, cin, taps, c.cout_pad, frag.data()); src = frag.data(); }
      const size_t n = i > 0 ? frag.size() : (size_t)cout * 9;       // the first layer (1 input channel) keeps [64][9] for the FMA kernel
      if (!c.wf) { void* q = nullptr; GN_HIP(hipMalloc(&q, n * sizeof(float))); ctx->sp_allocs.push_back(q); c.wf = (float*)q; }
      GN_HIP(hipMemcpy(c.wf, src, n * sizeof(float), hipMemcpyHostToDevice));
      if (i > 0 && ctx->precision == GN_PREC_F16X2_BF16_ATTN && cin % 16 == 0) {   // the split-fp16 convolution of this precision mode
        float mx = 0.f;
        for (size_t q = 0; q < (size_t)cout * cin * taps; ++q) mx = std::max(mx, fabsf(host[q]));
        int e = 0;
        if (mx > 0.f && std::isfinite(mx)) { frexpf(mx, &e); e = 13 - e; }      // max |w| * 2^e in [2^12, 2^13), as for the matcher's weights (build_planes)
        e = std::max(-60, std::min(60, e));
        std::vector<uint16_t> fh((size_t)2 * c.cout_pad * taps * cin);
        sp_weight_fragments_hm16(host, cout, cin, taps, c.cout_pad, ldexpf(1.0f, e), fh.data());
        if (!c.wfh) { void* q = nullptr; GN_HIP(hipMalloc(&q, fh.size() * sizeof(uint16_t))); ctx->sp_allocs.push_back(q); c.wfh = (uint16_t*)q; }
        GN_HIP(hipMemcpy(c.wfh, fh.data(), fh.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        c.acc_scale = ldexpf(1.0f, -e);
      }
      c.have_w = true;
      return GN_OK;
    }
    if (name == base + ".bias") {
      if (shape[0] != cout) return fail(ctx, GN_ERR_SHAPE, "shape mismatch for " + name);
      std::vector<float> pad((size_t)c.cout_pad, 0.f);
      memcpy(pad.data(), host, (size_t)cout * sizeof(float));
      if (!c.b) { void* q = nullptr; GN_HIP(hipMalloc(&q, pad.size() * sizeof(float))); ctx->sp_allocs.push_back(q); c.b = (float*)q; }
      GN_HIP(hipMemcpy(c.b, pad.data(), pad.size() * sizeof(float), hipMemcpyHostToDevice));
      c.have_b = true;
      return GN_OK;
    }
  }
  return fail(ctx, GN_ERR_NAME, "unknown SuperPoint tensor " + name);
}

int gn_sp_detect_and_describe(gn_ctx* ctx, const float* gray01, int B, int H, int W, int max_kpts,
                              float* kpt_xysa, float* score, float* desc, int32_t* n_out_host, void* stream) {
  if (!ctx || !gray01 || !kpt_xysa || !desc || !n_out_host || B < 1 || H < 16 || W < 16 || H % 8 || W % 8 || max_kpts < 1 || max_kpts > 2048)
    return fail(ctx, GN_ERR_ARG, "bad gn_sp_detect_and_describe argument (H, W multiples of 8; max_kpts <= 2048)");
  for (int i = 0; i < 12; ++i) if (!ctx->sp[i].have_w || !ctx->sp[i].have_b) return fail(ctx, GN_ERR_WEIGHTS, "SuperPoint weights not fully loaded");
  GN_HIP(hipSetDevice(ctx->device));
  hipStream_t s = (hipStream_t)stream;
  const int chunk = std::min(B, 4);                       // images per pass: the full-resolution 64-channel maps are 0.5 GB per 1080p image
  const int h = H / 8, w = W / 8;
  const int cap = std::max(16384, H * W / 16);            // candidates after NMS (radius 4): at most one per ~5 x 5 neighbourhood in practice
  if (ctx->sp_h != H || ctx->sp_w != W || ctx->sp_chunk < chunk || ctx->sp_max < max_kpts) {
    for (void* p : ctx->sp_allocs) { bool is_weight = false; for (int i = 0; i < 12; ++i) is_weight |= (p == ctx->sp[i].wf || p == ctx->sp[i].b || p == ctx->sp[i].wfh); if (!is_weight) hipFree(p); }
    std::vector<void*> keep;
    for (void* p : ctx->sp_allocs) { for (int i = 0; i < 12; ++i) if (p == ctx->sp[i].wf || p == ctx->sp[i].b || p == ctx->sp[i].wfh) { keep.push_back(p); break; } }
    ctx->sp_allocs = keep;
    auto alloc = [&](size_t bytes) -> void* { void* p = nullptr; if (hipMalloc(&p, bytes) != hipSuccess) return nullptr; ctx->sp_allocs.push_back(p); return p; };
    const size_t full = (size_t)chunk * H * W;
    ctx->sp_x = (float*)alloc(full * 64 * sizeof(float));
    ctx->sp_y = (float*)alloc(full * 64 * sizeof(float));
    ctx->sp_z = (float*)alloc((size_t)chunk * h * w * 256 * sizeof(float));
    bool ok = ctx->sp_x && ctx->sp_y && ctx->sp_z;
    for (int k = 0; k < 6; ++k) { ctx->sp_maps[k] = (float*)alloc(full * sizeof(float)); ok = ok && ctx->sp_maps[k]; }
    ctx->sp_cand = (int*)alloc((size_t)chunk * cap * 2 * sizeof(int));      // (raster index, score bits) pairs
    ctx->sp_counts = (int*)alloc((size_t)chunk * 4 * sizeof(int));
    ctx->sp_index = (int*)alloc((size_t)chunk * 2048 * sizeof(int));
    if (!ok || !ctx->sp_cand || !ctx->sp_counts || !ctx->sp_index) return fail(ctx, GN_ERR_HIP, "SuperPoint workspace allocation failed");
    ctx->sp_h = H; ctx->sp_w = W; ctx->sp_chunk = chunk; ctx->sp_cap = cap; ctx->sp_max = max_kpts;
  }
  float *X = ctx->sp_x, *Y = ctx->sp_y, *Z = ctx->sp_z;
  // contexts of the f16x2 mode run the convolutions on split-fp16 operands (5 x the matrix-pipe rate of the exact f32 instruction,
  // the same accuracy class); an activation that does not fit fp16 raises ovf[1] and the pass is repeated on the exact path
  bool split = ctx->sp_split && ctx->sp[1].wfh != nullptr;
  // GN_SP_FP16: the activations of the layers above 1 / 8 resolution (layers 0 .. 6: 95 % of the extractor's bytes) travel as fp16;
  // layer 6 writes f32 again, the 1 / 8-resolution layers and the heads read and write f32 as in the other modes
  bool hm16_io = false;
  auto conv = [&](int i, const float* in, float* out, int n, int hh, int ww, int relu, int pool = 0) {
    if (ctx->sp_stop > 0 && i > ctx->sp_stop) return;      // developer knob 39: the layers behind layer sp_stop are skipped (their outputs are garbage)
    const bool hm = split && ctx->sp[i].wfh != nullptr;
    const bool half_io = hm && ctx->sp_split == 2;
    // split mode (round 5): every activation between the layers travels as hm16 records (k_sp_conv_s); the two head outputs stay f32
    // GN_SP_FP16 with k_sp_conv_h16 (knob 24 = 2): fp16 activations all the way to the two head outputs (the single-product staging rounds the same
    // values to fp16 anyway); with the older kernels only layers 1 .. 6 read and layers 1 .. 5 write fp16
    const bool h_all = gn::g_sp_conv_h == 2;
    const int in_fmt = hm16_io ? 2 : (half_io && i >= 1 && (h_all || i <= 6) ? 1 : 0);
    const int out_fmt = hm16_io ? (i == 9 || i == 11 ? 0 : 2) : (half_io && i >= 1 && (h_all ? (i != 9 && i != 11) : i <= 5) ? 1 : 0);
    sp_conv(in, n, hh, ww, ctx->sp[i].cin, ctx->sp[i].wf, ctx->sp[i].b, out, ctx->sp[i].cout_pad, ctx->sp[i].taps, relu, s,
            hm ? ctx->sp[i].wfh : nullptr, ctx->sp[i].acc_scale, ctx->ovf_base + 8, pool, ctx->sp_split == 2, in_fmt, out_fmt,
            (ctx->sp_ts_layer == i && ctx->sp_ts != nullptr) ? ctx->sp_ts : nullptr);
  };
  std::vector<int> counts((size_t)chunk * 4);
  for (int b0 = 0; b0 < B; b0 += chunk) {
    const int n = std::min(chunk, B - b0);
    if (split) GN_HIP(hipMemsetAsync(ctx->ovf_base + 8, 0, sizeof(unsigned int), s));
    // (k_sp_conv_s addresses a frame's activations with 32-bit buffer offsets, 0x80000000 marking a pixel outside the image: frames whose
    // full-resolution 64-channel map reaches 2 GB -- beyond 8 Mpixel -- stay on the round-2 kernels)
    hm16_io = split && ctx->sp_split == 1 && gn::g_sp_conv_s != 0 && (size_t)H * W * 256 < 0x7fffffffull;
    for (int i = 1; i < 12; ++i) hm16_io = hm16_io && ctx->sp[i].wfh != nullptr;
    ctx->sp_enc_hm16 = hm16_io;
    ctx->sp_enc_fp16 = !hm16_io && split && ctx->sp_split == 2 && gn::g_sp_conv_h == 2 && ctx->sp[7].wfh != nullptr;
    // round 6: in the split-fp16 mode the first convolution is evaluated inside the second one's halo staging (k_sp_conv_s16<., ., true>): its
    // full-resolution 64-channel map -- 0.53 GB of records per 1080p frame -- is neither written nor read; same bits (knob 46 = 0: two launches)
    const bool fuse1 = hm16_io && gn::g_sp_fuse1 && gn::g_sp_conv_s == 2 && ctx->sp[1].cin == 64 && ctx->sp[1].cout_pad == 64 && ctx->sp[1].taps == 9 && ctx->sp_stop != 1 &&
                       !(ctx->sp_ts_layer == 1 && ctx->sp_ts != nullptr);
    if (fuse1) {
      sp_conv_fused1(gray01 + (size_t)b0 * H * W, ctx->sp[0].wf, ctx->sp[0].b, n, H, W, ctx->sp[1].b, Y, ctx->sp[1].cout_pad, s, ctx->sp[1].wfh, ctx->sp[1].acc_scale, ctx->ovf_base + 8);
    } else {
    sp_conv1(gray01 + (size_t)b0 * H * W, ctx->sp[0].wf, ctx->sp[0].b, X, n, H, W, s,
             hm16_io ? 2 : (split && ctx->sp_split == 2 && ctx->sp[1].wfh != nullptr ? 1 : 0), ctx->ovf_base + 8);
    // the three 2 x 2 max-pools are fused into the epilogues of the convolutions in front of them (the full-resolution 64-channel map of
    // block 0 alone is 0.5 GB per 1080p image: writing it and reading it back was a quarter of the extractor's HBM traffic)
    conv(1, X, Y, n, H, W, 1, 1);                                                                   // block 0 -> Y [H/2][W/2][64]
    }
    conv(2, Y, X, n, H / 2, W / 2, 1);     conv(3, X, Y, n, H / 2, W / 2, 1, 1);                    // block 1 -> Y [H/4][W/4][64]
    conv(4, Y, X, n, H / 4, W / 4, 1);     conv(5, X, Y, n, H / 4, W / 4, 1, 1);                    // block 2 -> Y [H/8][W/8][128]
    conv(6, Y, X, n, h, w, 1);             conv(7, X, Y, n, h, w, 1);                               // block 3 -> Y = encoder output
    conv(8, Y, X, n, h, w, 1);             conv(9, X, Z, n, h, w, 0);                               // detector head: Z = logits [h][w][128 (65 used)]
    sp_scores(Z, ctx->sp[9].cout_pad, ctx->sp_maps[0], n, h, w, s);
    sp_nms(ctx->sp_maps[0], n, H, W, 4, ctx->sp_maps[1], ctx->sp_maps[2], ctx->sp_maps[3], ctx->sp_maps[4], ctx->sp_maps[5], s);
    sp_select(ctx->sp_maps[5], n, H, W, 0.005f, 4, ctx->sp_cand, ctx->sp_counts, ctx->sp_cap, max_kpts,
              kpt_xysa + (size_t)b0 * max_kpts * 4, score ? score + (size_t)b0 * max_kpts : ctx->sp_maps[1], ctx->sp_index, max_kpts, s);
    conv(10, Y, X, n, h, w, 1);            conv(11, X, Z, n, h, w, 0);                          // descriptor head: Z = raw descriptor map [h][w][256]
    sp_describe(Z, n, h, w, kpt_xysa + (size_t)b0 * max_kpts * 4, ctx->sp_counts, max_kpts, max_kpts, desc + (size_t)b0 * max_kpts * 256, s);
    GN_HIP(hipMemcpyAsync(counts.data(), ctx->sp_counts, (size_t)n * 4 * sizeof(int), hipMemcpyDeviceToHost, s));
    unsigned int tripped = 0;
    if (split) GN_HIP(hipMemcpyAsync(&tripped, ctx->ovf_base + 8, sizeof(unsigned int), hipMemcpyDeviceToHost, s));
    GN_HIP(hipStreamSynchronize(s));
    if (split && tripped) { split = false; ++ctx->sp_split_trips; b0 -= chunk; continue; }   // repeat this pass with exact f32 convolutions
    for (int b = 0; b < n; ++b) {
      if (counts[4 * b] > ctx->sp_cap) return fail(ctx, GN_ERR_ARG, "SuperPoint candidate buffer overflow");
      n_out_host[b0 + b] = counts[4 * b + 1];
    }
  }
  GN_HIP(hipGetLastError());
  return GN_OK;
}

int64_t gn_debug_read(gn_ctx* ctx, const char* name, void* host_out, int64_t max_bytes, void* stream) {
  if (!ctx || !name || !host_out) return GN_ERR_ARG;
  hipSetDevice(ctx->device);
  const size_t np = ctx->npad, T = (size_t)ctx->max_batch * 2 * np, B = ctx->max_batch;
  struct Ent { const char* n; const void* p; size_t count; };
  const Ent tab[] = {
      {"desc", ctx->desc, T * kInDim}, {"cos", ctx->cos_t, T * kFreq}, {"sin", ctx->sin_t, T * kFreq},
      {"x", ctx->x, T * kDim}, {"qkv", ctx->qkv, T * 3 * kDim}, {"ctx", ctx->ctx, T * kDim}, {"msg", ctx->msg, T * kDim},
      {"h", ctx->h, T * 2 * kDim}, {"md", ctx->md, T * kDim}, {"ls", ctx->ls, T}, {"sim", ctx->sim, ctx->sim ? B * np * np : 0},
      {"rowmax", ctx->rowmax, B * np}, {"rowlog", ctx->rowlog, B * np}, {"colmax", ctx->colmax, B * np},
      {"collog", ctx->collog, B * np}, {"max0", ctx->max0, B * np}, {"max0b", ctx->max0b, B * np}, {"uncert", ctx->uncert, B}, {"m0", ctx->m0, B * np}, {"m1", ctx->m1, B * np},
      {"extent", ctx->extent, B * 4}, {"nvalid", ctx->nvalid, B * 2}, {"e_mkp", ctx->e_mkp, B * np * 2},
      {"e_obj", ctx->e_obj, B * np * 3}, {"e_score", ctx->e_score, B * np},
      {"hyp", ctx->hyp_ws, B * 16 * (sizeof(gn::HypResult) / 4)},
      {"sp_enc", ctx->sp_y, ctx->sp_y ? (size_t)ctx->sp_chunk * (ctx->sp_h / 8) * (ctx->sp_w / 8) * 128 : 0},          // SuperPoint: encoder output of the last pass, NHWC
      {"sp_scores", ctx->sp_maps[0], ctx->sp_maps[0] ? (size_t)ctx->sp_chunk * ctx->sp_h * ctx->sp_w : 0},             // softmax + depth-to-space scores
      {"sp_nms", ctx->sp_maps[5], ctx->sp_maps[5] ? (size_t)ctx->sp_chunk * ctx->sp_h * ctx->sp_w : 0},                // after simple_nms
      {"sp_counts", ctx->sp_counts, ctx->sp_counts ? (size_t)ctx->sp_chunk * 4 : 0},                                    // per image: candidates, keypoints, candidates seen by the select
      {"sp_cand", ctx->sp_cand, ctx->sp_cand ? (size_t)ctx->sp_chunk * ctx->sp_cap * 2 : 0},                            // (raster index, score bits) pairs, [image][cap]
      {"sp_x", ctx->sp_x, ctx->sp_x ? (size_t)ctx->sp_chunk * ctx->sp_h * ctx->sp_w * 64 : 0},                         // raw words of the two activation buffers
      {"sp_y", ctx->sp_y, ctx->sp_y ? (size_t)ctx->sp_chunk * ctx->sp_h * ctx->sp_w * 64 : 0},
      {"sp_ts", ctx->sp_ts, ctx->sp_ts ? (size_t)8192 * 32 * 2 : 0},                                                     // phase stamps (knob 35), int64 pairs of 4-byte words
      {"x_p", ctx->x_p, ctx->x_p ? T * kDim : 0}, {"msg_p", ctx->msg_p, ctx->msg_p ? T * kDim : 0},   // hm16 rows, raw (4 bytes per value)
      {"qkb", ctx->qkb, ctx->qkb ? T * kDim : 0}, {"rot4", ctx->rot4, ctx->rot4 ? T * 2 * kFreq : 0}, {"vtb", ctx->vtb, ctx->vtb ? T * kDim / 2 : 0}};
  for (const Ent& e : tab)
    if (strcmp(e.n, name) == 0) {
      size_t count = e.count;
      if ((int64_t)(count * 4) > max_bytes) count = (size_t)max_bytes / 4;
      if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return GN_ERR_HIP;
      if (hipMemcpy(host_out, e.p, count * 4, hipMemcpyDeviceToHost) != hipSuccess) return GN_ERR_HIP;
      if (strcmp(name, "sp_enc") == 0 && ctx->sp_enc_fp16) {
        // fp16 NHWC values (the first half of the bytes read) -> f32, back to front in place
        const uint16_t* hsrc = (const uint16_t*)host_out;
        float* dst = (float*)host_out;
        for (size_t g = count; g-- > 0;) dst[g] = (float)__builtin_bit_cast(_Float16, hsrc[g]);
      }
      if (strcmp(name, "sp_enc") == 0 && ctx->sp_enc_hm16) {
        // hm16 records (per 16 channels: 16 high halfs, 16 residual halfs) -> f32 values in channel order, in place
        uint16_t rec[32];
        for (size_t g = 0; g + 16 <= count; g += 16) {
          memcpy(rec, (const char*)host_out + g * 4, 64);
          float* dst = (float*)host_out + g;
          for (int c = 0; c < 16; ++c) dst[c] = (float)__builtin_bit_cast(_Float16, rec[c]) + (float)__builtin_bit_cast(_Float16, rec[16 + c]);
        }
      }
      return (int64_t)count;
    }
  if (strcmp(name, "e_idx") == 0) {
    size_t count = B * np * 2;
    if ((int64_t)(count * 8) > max_bytes) count = (size_t)max_bytes / 8;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return GN_ERR_HIP;
    if (hipMemcpy(host_out, ctx->e_idx, count * 8, hipMemcpyDeviceToHost) != hipSuccess) return GN_ERR_HIP;
    return (int64_t)count;
  }
  return fail(ctx, GN_ERR_NAME, std::string("unknown debug tensor ") + name);
}

int gn_debug_gemm(gn_ctx* ctx, int M, int N, int K, const float* A, const float* W, const float* bias, float* Y, void* stream) {
  if (!ctx || !A || !W || !Y || M % 128 || N % 128 || K % 32 || M < 128 || N < 128 || K < 32)
    return fail(ctx, GN_ERR_ARG, "gn_debug_gemm needs M,N multiples of 128 and K a multiple of 32");
  GN_HIP(hipSetDevice(ctx->device));
  GemmArgs g;
  memset(&g, 0, sizeof g);
  g.A = A; g.lda = K; g.K1 = K; g.W = W; g.ldw = K; g.bias = bias; g.Y = Y; g.ldy = N; g.M = M; g.N = N; g.K = K;
  g.acc_scale = 1.f;
  if (ctx->gemm_variant >= 0) gn::g_gemm_variant = ctx->gemm_variant;   // the kernel family of THIS context (the selector is a process-wide developer knob: another context's last launch may have left its own)
  if (ctx->gemm_variant == 7) {   // k_gemm_p2: both operands as fp16 planes (W scaled by 2^(dbg_planes - 1) if dbg_planes > 0)
    
    const size_t na = (size_t)M * K, nw = (size_t)N * K;
    if (ctx->dbg_ap_n < na) { ctx->dbg_ap = nullptr; int rc = dalloc(ctx, &ctx->dbg_ap, 2 * na); if (rc != GN_OK) return rc; ctx->dbg_ap_n = na; }
    if (ctx->dbg_wp_n < nw) { ctx->dbg_wp = nullptr; int rc = dalloc(ctx, &ctx->dbg_wp, 3 * nw); if (rc != GN_OK) return rc; ctx->dbg_wp_n = nw; }
    const int e = ctx->dbg_planes > 0 ? ctx->dbg_planes - 1 : 0;
    if (!ctx->dbg_reuse) {
      launch_split_hm16(A, ctx->dbg_ap, M, K, 1.0f, (hipStream_t)stream);
      launch_split_hm16(W, ctx->dbg_wp, N, K, ldexpf(1.0f, e), (hipStream_t)stream);
    }
    g.Ap = ctx->dbg_ap; g.Wp = ctx->dbg_wp; g.acc_scale = ldexpf(1.0f, -e);
    if (ctx->dbg_out) {
      const size_t ny = (size_t)M * N;
      if (ctx->dbg_yp_n < ny) { ctx->dbg_yp = nullptr; int rc = dalloc(ctx, &ctx->dbg_yp, 2 * ny); if (rc != GN_OK) return rc; ctx->dbg_yp_n = ny; }
      g.Yp = ctx->dbg_yp; g.ldyp = N;
      if (ctx->dbg_out == 1) g.Y = nullptr;
    }
    launch_gemm_p2(bias ? EPI_BIAS : EPI_PLAIN, g, 1, (hipStream_t)stream);
    GN_HIP(hipGetLastError());
    return GN_OK;
  }
  if (ctx->dbg_planes) {
    const size_t n = (size_t)N * K;
    if (ctx->dbg_wp_n < n) { ctx->dbg_wp = nullptr; int rc = dalloc(ctx, &ctx->dbg_wp, 3 * n); if (rc != GN_OK) return rc; ctx->dbg_wp_n = n; }
    if (ctx->gemm_variant == 6 || ctx->gemm_variant >= 60) {   // fp16 planes of W * 2^(dbg_planes - 1)
      launch_split2_f16(W, ctx->dbg_wp, (long long)n, ldexpf(1.0f, ctx->dbg_planes - 1), (hipStream_t)stream);
      g.acc_scale = ldexpf(1.0f, 1 - ctx->dbg_planes);
    } else {
      launch_split3_bf16(W, ctx->dbg_wp, (long long)n, (hipStream_t)stream);
    }
    g.Wp = ctx->dbg_wp; g.wp_plane = (long long)n;
  }
  launch_gemm_f32(bias ? EPI_BIAS : EPI_PLAIN, g, 1, (hipStream_t)stream);
  GN_HIP(hipGetLastError());
  return GN_OK;
}

int gn_debug_attention(gn_ctx* ctx, int BS, int npad, int cross, float qscale, const float* q, int ldq, const float* k, int ldk,
                       const float* v, int ldv, const int32_t* nkv, float* out, int ldo, void* stream) {
  if (!ctx || !q || !k || !v || !nkv || !out || npad % 128 || BS < 1 || (cross && (BS & 1)))
    return fail(ctx, GN_ERR_ARG, "bad gn_debug_attention argument");
  GN_HIP(hipSetDevice(ctx->device));
  AttnArgs a; a.ovf = nullptr;
  a.q = q; a.ldq = ldq; a.k = k; a.ldk = ldk; a.v = v; a.ldv = ldv; a.out = out; a.ldo = ldo;
  a.nvalid = nkv; a.npad = npad; a.cross = cross; a.qscale = qscale; a.BS = BS;
  a.qb = a.kb = a.vt = nullptr; a.ldqb = a.ldkb = 0; a.outp = nullptr; a.half_fmt = ctx->attn_f16; a.ncu = ctx->ncu;
  const long long cap = (long long)ctx->max_batch * 2 * ctx->npad;
  if (ctx->precision != GN_PREC_F32 && ctx->attn_variant >= 1 && ctx->qkb && ctx->vtb && (long long)BS * npad <= cap) {
    // the production kernel (k_attn_bf16_v5) on the layouts the projection epilogues would have written
    launch_pack_attn_bf16(a, ctx->qkb, ctx->vtb, (hipStream_t)stream);
    a.qb = ctx->qkb; a.kb = ctx->qkb + kDim; a.ldqb = a.ldkb = 2 * kDim; a.vt = ctx->vtb;
    gn::g_attn_variant = ctx->attn_variant;
    gn::g_attn_stamps = ((ctx->attn_variant == 73 || ctx->attn_variant >= 1000) && ctx->sim) ? reinterpret_cast<long long*>(ctx->sim) : nullptr;
    attn_split(ctx, a);
    launch_attention_bf16_v2(a, (hipStream_t)stream);
  } else {
    attention(ctx, a, (hipStream_t)stream);
  }
  GN_HIP(hipGetLastError());
  return GN_OK;
}

int gn_sp_set_arithmetic(gn_ctx* ctx, int mode) {
  if (!ctx || mode < GN_SP_EXACT_F32 || mode > GN_SP_FP16) return GN_ERR_ARG;
  if (mode != GN_SP_EXACT_F32 && ctx->precision != GN_PREC_F16X2_BF16_ATTN) return fail(ctx, GN_ERR_ARG, "the fp16 SuperPoint arithmetic needs a context of the f16x2 precision");
  ctx->sp_split = mode;
  return GN_OK;
}

int gn_debug_set_variant(gn_ctx* ctx, int which, int value) {
  if (!ctx) return GN_ERR_ARG;
  if ((which == 1 && (value == 73 || value >= 1000)) || (which == 12 && (value & 8)) || (which == 15 && value) || (which == 16 && !value) || (which == 17 && value) || (which == 20 && value)) {   // these developer paths use the similarity buffer
    GN_HIP(hipSetDevice(ctx->device));
    const int rc = ensure_sim(ctx); if (rc != GN_OK) return rc;
  }
  if (which == 0) { gn::g_gemm_variant = value; ctx->gemm_variant = value; }
  else if (which == 1) ctx->attn_variant = value;
  else if (which == 2) ctx->dbg_planes = value;
  else if (which == 3) ctx->no_planes = value;
  else if (which == 4) ctx->stop_after = value;
  else if (which == 5) ctx->planes_mode = (value && ctx->x_p) ? 1 : 0;
  else if (which == 6) ctx->dbg_reuse = value;
  else if (which == 7) ctx->dbg_out = value;
  else if (which == 10) ctx->ffn_fused = value;
  else if (which == 11) ctx->x_planes_only = value;
  else if (which == 8) gn::g_p2_wide = value;
  else if (which == 9) ctx->dbg_vt_skip = value;
  else if (which == 12) gn::g_ffn_ablate = value;
  else if (which == 13) ctx->ffn_fold = value;
  else if (which == 14) gn::g_ffn_shape = value;
  else if (which == 15) ctx->pnp_stamps = value;
  else if (which == 16) ctx->head_fused = value;
  else if (which == 17) ctx->head_stamps = value;
  else if (which == 18) gn::g_head_ablate = value;
  else if (which == 19) ctx->qkv_fused = value;
  else if (which == 20) ctx->qkv_stamps = value;
  else if (which == 21) ctx->sp_split = value;
  else if (which == 24) gn::g_sp_conv_h = value;
  else if (which == 34) gn::g_sp_conv_s = value;
  else if (which == 36) gn::g_sp_nms_fused = value;
  else if (which == 40) gn::g_sp_select_stream = value;
  else if (which == 41) gn::g_gemm_m64 = value;
  else if (which == 42) gn::g_lf_conv_knob = value;
  else if (which == 43) gn::g_attn_f32_ks = value;
  else if (which == 44) gn::g_gemm_r64 = value;
  else if (which == 46) gn::g_sp_fuse1 = value;
  else if (which == 47) { ctx->fused_proj_pending = value == 0; }   // 1: skip the fused-projection self-check of the next forward call (profiling passes: its launches are set-up, not steps)
  else if (which == 39) ctx->sp_stop = value;
  else if (which == 35) {
    ctx->sp_ts_layer = value;
    if (value > 0 && ctx->sp_ts == nullptr) { void* q = nullptr; if (hipMalloc(&q, (size_t)8192 * 32 * sizeof(long long)) == hipSuccess) { ctx->sp_ts = (long long*)q; ctx->sp_allocs_dbg = q; } }
    if (ctx->sp_ts != nullptr) hipMemset(ctx->sp_ts, 0, (size_t)8192 * 32 * sizeof(long long));
  }
  else if (which == 23) ctx->attn_split = value;
  else if (which == 25) ctx->dbg_trip_group = value;
  else if (which == 26) ctx->sub_serial = value;
  else if (which == 27) ctx->qkv_products = value == 3 ? 3 : 2;
  else if (which == 28) ctx->ffn_compose = value;
  else if (which == 31) ctx->use_lists = value;
  else if (which == 32) ctx->qkv_in_tail = value ? 1 : 0;
  else if (which == 33) ctx->skinny = value;
  else if (which == 29) {   // CU shares for the sub-batch streams: takes effect for streams created afterwards
    GN_HIP(hipSetDevice(ctx->device));
    for (int i = 0; i < 8; ++i) if (ctx->sub_s[i]) { hipStreamSynchronize(ctx->sub_s[i]); hipEventDestroy(ctx->ev_join[i]); hipStreamDestroy(ctx->sub_s[i]); ctx->sub_s[i] = nullptr; ctx->ev_join[i] = nullptr; }
    ctx->cu_mask_mode = value;
    if (ctx->n_sub > 1) { const int n = ctx->n_sub; ctx->n_sub = 1; return gn_set_substreams(ctx, n); }
  }
  else return GN_ERR_ARG;
  return GN_OK;
}

int gn_debug_epnp(gn_ctx* ctx, int n, const double* pws, const double* us, double* out, void* stream) {
  if (!ctx || n < 1 || !pws || !us || !out) return GN_ERR_ARG;
  GN_HIP(hipSetDevice(ctx->device));
  launch_epnp_debug(pws, us, out, n, (hipStream_t)stream);
  GN_HIP(hipGetLastError());
  return GN_OK;
}

int gn_debug_lds_dma_probe(gn_ctx* ctx, const float* pattern, unsigned int* out, int blocks, int spin, void* stream) {
  if (!ctx || !pattern || !out || blocks < 1) return GN_ERR_ARG;
  GN_HIP(hipSetDevice(ctx->device));
  launch_lds_dma_probe(pattern, out, blocks, spin, (hipStream_t)stream);
  GN_HIP(hipGetLastError());
  return GN_OK;
}

int gn_debug_mfma_probe(gn_ctx* ctx, int blocks, int iters, void* stream) {
  if (!ctx || blocks < 1 || iters == 0) return GN_ERR_ARG;
  GN_HIP(hipSetDevice(ctx->device));
  launch_mfma_probe(ctx->ls, blocks, iters, (hipStream_t)stream);
  GN_HIP(hipGetLastError());
  return GN_OK;
}

int gn_set_kernel_timing(gn_ctx* ctx, int max_launches) {
  if (!ctx || max_launches < 0) return GN_ERR_ARG;
  hipSetDevice(ctx->device);
  while ((int)ctx->kflops.size() < max_launches) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return GN_ERR_HIP;
    ctx->kev.push_back(a); ctx->kev.push_back(b); ctx->kflops.push_back(0.0); ctx->kbytes.push_back(0.0); ctx->kclass.push_back(0); ctx->kname.push_back("");
  }
  ctx->ktiming = max_launches > 0;
  ctx->kused = 0;
  return GN_OK;
}

int gn_get_kernel_stats(gn_ctx* ctx, int kernel_class, double* out3) {
  if (!ctx || !out3 || kernel_class < 0 || kernel_class > 1) return GN_ERR_ARG;
  hipSetDevice(ctx->device);
  double ms = 0.0, fl = 0.0, n = 0.0;
  for (size_t i = 0; i < ctx->kused; ++i) {
    if (hipEventSynchronize(ctx->kev[2 * i + 1]) != hipSuccess) return GN_ERR_HIP;
    float t = 0.f;
    if (hipEventElapsedTime(&t, ctx->kev[2 * i], ctx->kev[2 * i + 1]) != hipSuccess) return GN_ERR_HIP;
    if (ctx->kclass[i] != kernel_class) continue;
    ms += t; fl += ctx->kflops[i]; n += 1.0;
  }
  out3[0] = n; out3[1] = ms; out3[2] = fl;
  return GN_OK;
}

// Per-kernel table of the recorded launches as a JSON array (HIP-event time on the launch stream, algorithmic flops and bytes):
// [{"name": "k_ffn_fused<0, true>", "launches": n, "ms": total, "flops": total, "bytes": total}, ...].  Names are the ones
// rocprofv3's kernel_stats rows carry (namespace prefix and argument list stripped).  Returns the length written, or a negative status.
int gn_get_kernel_table(gn_ctx* ctx, char* json, int capacity) {
  if (!ctx || !json || capacity < 64) return GN_ERR_ARG;
  hipSetDevice(ctx->device);
  std::map<std::string, std::vector<double>> tab;   // name -> {launches, ms, flops, bytes}
  for (size_t i = 0; i < ctx->kused; ++i) {
    if (hipEventSynchronize(ctx->kev[2 * i + 1]) != hipSuccess) return GN_ERR_HIP;
    float t = 0.f;
    if (hipEventElapsedTime(&t, ctx->kev[2 * i], ctx->kev[2 * i + 1]) != hipSuccess) return GN_ERR_HIP;
    std::vector<double>& v = tab[ctx->kname[i] ? ctx->kname[i] : "?"];
    if (v.empty()) v.assign(4, 0.0);
    v[0] += 1.0; v[1] += t; v[2] += ctx->kflops[i]; v[3] += ctx->kbytes[i];
  }
  std::string out = "[";
  bool first = true;
  for (const auto& kv : tab) {
    char buf[512];
    snprintf(buf, sizeof buf, "%s{\"name\": \"%s\", \"launches\": %.0f, \"ms\": %.6f, \"flops\": %.6e, \"bytes\": %.6e}", first ? "" : ", ",
             kv.first.c_str(), kv.second[0], kv.second[1], kv.second[2], kv.second[3]);
    out += buf; first = false;
  }
  out += "]";
  if ((int)out.size() + 1 > capacity) return GN_ERR_ARG;
  memcpy(json, out.c_str(), out.size() + 1);
  return (int)out.size();
}

int gn_get_kernel_bytes(gn_ctx* ctx, int kernel_class, double* out1) {
  if (!ctx || !out1 || kernel_class < 0 || kernel_class > 1) return GN_ERR_ARG;
  double by = 0.0;
  for (size_t i = 0; i < ctx->kused; ++i)
    if (ctx->kclass[i] == kernel_class) by += ctx->kbytes[i];
  *out1 = by;
  return GN_OK;
}

int gn_set_stage_timing(gn_ctx* ctx, int enable) {
  if (!ctx) return GN_ERR_ARG;
  ctx->timing = enable != 0;
  ctx->n_ev = 0;
  return GN_OK;
}

int gn_get_stage_ms(gn_ctx* ctx, float* host_ms, int max_stages) {
  if (!ctx || !host_ms) return GN_ERR_ARG;
  hipSetDevice(ctx->device);
  for (int i = 0; i < ST_COUNT; ++i) ctx->stage_ms[i] = 0.f;
  if (ctx->n_ev > 0) {
    if (hipEventSynchronize(ctx->ev[2 * (ctx->n_ev - 1) + 1]) != hipSuccess) return GN_ERR_HIP;
    for (int i = 0; i < ctx->n_ev; ++i) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, ctx->ev[2 * i], ctx->ev[2 * i + 1]) == hipSuccess) ctx->stage_ms[ctx->ev_stage[i]] += ms;
    }
  }
  const int n = max_stages < ST_COUNT ? max_stages : (int)ST_COUNT;
  for (int i = 0; i < n; ++i) host_ms[i] = ctx->stage_ms[i];
  return n;
}

}  // extern "C"
