// Internal declarations shared by the gfx950 kernels and the C ABI (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <map>

#include "../../include/gisnav_amd.h"

namespace gn {

constexpr int kDim = 256;       // descriptor_dim
constexpr int kHeads = 4;
constexpr int kHeadDim = 64;
constexpr int kInDim = 128;     // SIFT descriptor length
constexpr int kFreq = 32;       // rotary frequencies per head (head_dim / 2)
constexpr int kMaxLayers = 9;
constexpr int kRecordFloats = 133;   // GN_KPT_RECORD: one 532-byte KEYPOINT_DTYPE record (ros/gisnav/gisnav/core/_shared.py:26-35) as floats

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

// Exact (erf) GELU, 0.5 y (1 + erf(y / sqrt 2)), with erf(t) = 1 - 2^(-q(t)) for t = min(|y| / sqrt 2, 4): q is the
// degree-8 weighted-minimax fit of -log2(erfc(t)) on [0, 4] (no constant term: erf(0) = 0 exactly).  In f32 the result is
// within 1 ulp-of-the-output of the f64 value over [-8, 8] (max abs error 4.7e-7 at |y| = 4.4, the same as rounding
// libm's erff), at half the instructions of ocml's two-branch erff -- k_ln_gelu is VALU-bound, not HBM-bound.
__device__ __forceinline__ float gelu_erf(float y) {
  const float t = fminf(fabsf(y) * 0.70710678118654752440f, 4.0f);
  float q = 4.6081331674940884e-05f;
  q = q * t + -0.00045161080197431147f;
  q = q * t + 0.0015096671413630247f;
  q = q * t + 0.0007409505778923631f;
  q = q * t + -0.028223754838109016f;
  q = q * t + 0.1484677642583847f;
  q = q * t + 0.918419361114502f;
  q = q * t + 1.6279083490371704f;
  const float e = 1.0f - __builtin_amdgcn_exp2f(-(q * t));
  return 0.5f * y * (1.0f + copysignf(e, y));
}

// hm16 row format of the f16x2 mode: a value x travels as two fp16 terms x = h + m.  A row of ld values is ld / 16
// groups of 64 bytes: the 16 high terms of columns 16g .. 16g+15 followed by their 16 residual terms.  Offset (in
// fp16 elements) of the HIGH term of (row, col); the residual term sits 16 elements further.
__host__ __device__ inline size_t hm16_off(size_t row, int ld, int col) {
  return row * (size_t)ld * 2 + (size_t)(col >> 4) * 32 + (col & 15);
}

// f16x2 domain guard: every kernel that writes an activation as hm16 tracks max |x| of what it writes and raises the context's
// overflow word when a value does not fit fp16 (|x| >= 65504 -> the high term would be inf).  The match head then reports
// zero matches for the call (never a silent inf / NaN) and gn_match can re-run it in the exact-split f32x3 mode (gn_set_guard).
__device__ __forceinline__ void ovf_track(float& amax, float a, float b) { amax = fmaxf(amax, fmaxf(fabsf(a), fabsf(b))); }
__device__ __forceinline__ void ovf_commit(unsigned int* flag, float amax) { if (flag != nullptr && !(amax < 65504.0f)) atomicOr(flag, 1u); }

// two f32 -> one dword of two 16-bit floats, round to nearest even: v_cvt_pk_bf16_f32 (F16 = false) or v_cvt_pk_f16_f32 (true)
template <bool F16> __device__ __forceinline__ unsigned int pack16(float lo, float hi) {
  typedef float f32x2_p __attribute__((ext_vector_type(2)));
  if constexpr (F16) {
    typedef _Float16 f16x2_p __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned int, __builtin_convertvector((f32x2_p){lo, hi}, f16x2_p));
  } else {
    typedef __bf16 bf16x2_p __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned int, __builtin_convertvector((f32x2_p){lo, hi}, bf16x2_p));
  }
}
// the same with the format chosen at run time (GEMM epilogues shared by both attention formats)
__device__ __forceinline__ unsigned int pack16_rt(float lo, float hi, int f16) { return f16 ? pack16<true>(lo, hi) : pack16<false>(lo, hi); }

// ---- GEMM -----------------------------------------------------------------------------------
enum GemmEpi { EPI_BIAS = 0, EPI_SCALE_COLS = 1, EPI_ROTARY = 2, EPI_RESIDUAL = 3, EPI_PLAIN = 4,
               EPI_ROTARY_BF16 = 5, EPI_SCALE_BF16 = 6,
               EPI_LN_GELU = 7,     // bias -> LayerNorm over the N = 512 outputs of a row -> erf GELU -> hm16 (k_gemm_p2ln only)
               EPI_RELU = 8 };      // (bias ->) max(., 0): the LoFTR encoder's MLP (k_gemm_f32_v3 / k_gemm_f16x2 only)

struct GemmArgs {
  const float* A;    int lda;      // A[M][K1] (k < K1)
  const float* A2;   int lda2;     // optional second source for k >= K1 (concat), else nullptr
  int K1;
  const float* W;    int ldw;      // W[N][K]
  const uint16_t* Wp; long long wp_plane;  // optional pre-split weights: bf16 planes [3][N][K] (f32x3) or scaled fp16 planes [2][N][K] (f16x2)
  float acc_scale;                 // f16x2: exact inverse of the power-of-two weight-plane scale (1 if none)
  const float* bias;               // [N] or nullptr
  float* Y;          int ldy;      // Y[M][N]
  int M, N, K;
  // batching over blockIdx.z
  long long strideA, strideW, strideY;
  // epilogue extras
  float scale; int scale_cols;     // EPI_SCALE_COLS: cols < scale_cols multiplied by scale (after bias)
  const float* cos_t; const float* sin_t; int rot_cols;  // EPI_ROTARY: [M][32] tables, cols < rot_cols rotated
  const float* resid; int ldr;     // EPI_RESIDUAL
  const uint16_t* residp; int ldrp;        // EPI_RESIDUAL, k_gemm_p2*: the residual rows as hm16 pairs (h + m) instead of f32 (resid may then be nullptr)
  int drop_f32;                            // k_gemm_p2*: when the output also leaves as hm16 (Yp), do not write the f32 copy
  const float* ln_g; const float* ln_b;   // EPI_LN_GELU: LayerNorm weight / bias over the N = 512 outputs (eps 1e-5)
  // EPI_*_BF16: columns < vt_start go to Yb (bf16 row-major, q columns < q_cols scaled by qscale),
  // columns >= vt_start go to Vt (bf16, [slot][head][64][npad] = V transposed per (pair, side, head))
  uint16_t* Yb; int ldyb; uint16_t* Vt; int vt_start; int q_cols; float qscale; int npad;
  int vt_perm;                     // 1: V^T tokens permuted within 16-groups for k_attn_bf16_v4 (see the epilogue)
  int half_fmt;                    // EPI_*_BF16: 0 = Yb / Vt hold bf16, 1 = fp16 (GN_PREC_F16X2_F16_ATTN; |value| >= 65504 raises ovf)
  // split-fp16 operands / outputs of k_gemm_p2 (gn_gemm_p2.hip) in the hm16 row format (see hm16_off): a row of
  // ld values occupies ld * 4 bytes, like the f32 row it shadows
  const uint16_t* Ap;                        // A (k < K1), row pitch lda values
  const uint16_t* A2p;                       // optional second source (k >= K1), same row pitch
  uint16_t* Yp; int ldyp;                    // optional hm16 output (Y may then be nullptr)
  unsigned int* ovf;                         // f16x2 domain guard word (see ovf_track), or nullptr
  // optional row limit read on the DEVICE (LoFTR's fine level: the number of matches stays on the device, the buffers hold max_matches windows):
  // a row tile whose first row r has (mlim_seg ? r % mlim_seg : r) >= mlim[0] * mlim_mul is skipped (its rows of Y keep what they held)
  const int* mlim; int mlim_mul; int mlim_seg;
};
void launch_gemm_f32(int epi, const GemmArgs& a, int batch, hipStream_t s);
void launch_gemm_p2(int epi, const GemmArgs& a, int batch, hipStream_t s);   // fp16-plane operands (Ap, Wp)
void launch_mfma_probe(float* out, int blocks, int iters, hipStream_t s);
void launch_lds_dma_probe(const float* pattern, unsigned int* out, int blocks, int spin, hipStream_t s);
extern int g_gemm_m64;
extern int g_gemm_r64;
extern int g_attn_f32_ks;     // developer knob 43 (gn_attention.hip launch_attention_f32)
extern int g_lf_conv_knob;   // developer knob 42 (gn_loftr.hip lf_conv)
extern thread_local const char* g_last_kernel;   // bench facility: rocprof-style name of the kernel the last launch_* call of THIS host thread dispatched
extern thread_local int g_gemm_variant;  // developer knob: kernel variant selector for A/B benchmarking (per host thread: every API entry sets it from its own context before it launches)
extern int g_p2_wide;       // developer knob: 1 = k_gemm_p2 uses the 256x256 kernel when the shape allows

// ---- fused block tail (gn_ffn.hip): x <- x + ffn.3(GELU(LayerNorm(ffn.0([x | msg])))) for hm16 rows ----------
struct FfnArgs {
  const uint16_t* xp;                      // [T][256] hm16 residual stream: k < 256 of ffn.0's input and the residual rows
  const uint16_t* mp;                      // [T][256] hm16 message: k >= 256 of ffn.0's input (used when cp == nullptr)
  int composed = 0;                        // 1: out_proj is composed into ffn.0 at load time (w1s = [W1_x | W1_m Wo] in natural k order, b1 = b1 + W1_m bo): cp holds the attention
                                           // output rows and is the k >= 256 half of ffn.0's input; wos / bo unused
  const uint16_t* cp;                      // [T][256] hm16 attention output: when set, message = out_proj(cp) is computed in the kernel ...
  const uint16_t* wos; float wo_scale; const float* bo;   // ... with the out_proj / to_out weight [256][256] in fragment order (natural k) and its bias
  const uint16_t* w1s; float w1_scale;     // ffn.0 weight [512][512] in MFMA fragment order (build_weight_fragments: natural k, or order 2 when cp is set), accumulator scale
  const float* b1; const float* ln_g; const float* ln_b;
  const uint16_t* w2s; float w2_scale;     // ffn.3 weight [256][512] in fragment order with the permuted k of a register-fed operand
  const float* b2;
  uint16_t* yp;                            // [T][256] hm16 output rows (may be xp: a workgroup only touches its own 64 rows)
  float* y;                                // optional f32 copy of the output, or nullptr
  int T;                                   // tokens, a multiple of 64 (of 32 for the 4-wave shape)
  unsigned int* ovf;                       // f16x2 domain guard word, or nullptr
  long long* dbg_ts;                       // developer: [blocks][8 waves][8] s_memtime stamps (ablation 8), or nullptr
  const int* tiles = nullptr;              // k_ffn128: work list of the call (launch_tile_lists), or nullptr = every 128-token tile of T
  int walk = 0;                            // 1: one workgroup per CU walks `tiles`; 0: one workgroup per tile of T, which leaves at once when `tiles` says (through
                                           // nvalid / npad) that its tile holds only padding
  const int32_t* nvalid = nullptr; int npad = 0;
  int ncu = 0;                             // compute units of the context's device (walking form: one workgroup per CU); 0 = ask the current device
  // k_ffn128 with the NEXT block's attention input projection fused behind the tail (round 5): the tile's new residual rows go from the epilogue's
  // registers into LDS as the projection's token operand (no 64 MB round trip through HBM, no second launch).  Same arithmetic, operand order and
  // outputs as k_qkv<., true, 2> (gn_qkv.hip: fp16 outputs, two partial products): bit-identical q | k rows and V^T panels.
  int products = 3;                        // k_ffn128, composed form: 3 = the f32-accurate split (three fp16 partial products per GEMM), 2 = activations' high term only (gn_set_ffn_products)
  int qkv = 0;                             // 0: none; 1: the self block's Wqkv + rotary; 2: the cross block's to_qk | to_v
  const uint16_t* q_wf = nullptr; float q_acc_scale = 1.f; const float* q_bias = nullptr;   // QkvArgs::wf / acc_scale / bias
  const float* q_rot4 = nullptr; long long q_rot_stride = 0;                               // QkvArgs::rot4 / rot_stride (qkv == 1)
  uint16_t* q_qkb = nullptr; uint16_t* q_vt = nullptr;                                     // fp16 q | k (ld 512) or qk (ld 256) rows; V^T panels [slot][4][64][npad]
  float q_qscale = 1.f, q_scale = 1.f;                                                     // QkvArgs::qscale (self) / scale (cross)
};
void launch_ffn_fused(const FfnArgs& a, hipStream_t s);
bool ffn_selects_128(const FfnArgs& a);   // true when launch_ffn_fused(a) dispatches k_ffn128 (the caller may then ask for the fused projection: a.qkv)
void launch_ffn128(const FfnArgs& a, int ablate, hipStream_t s);   // gn_ffn128.hip: 128 tokens per workgroup (a.cp set, a.T % 128 == 0); ablate: developer knob 12
extern int g_ffn_ablate;
extern int g_ffn_shape;
extern int g_sp_conv_h;
extern int g_head_ablate;
void build_weight_fragments(const float* w, int N, int K, float scale, int permute_k, uint16_t* out);   // host arrays; out: 2 * N * K halfs

// ---- attention input projections of the f16x2 mode (gn_qkv.hip): hm16 rows in, bf16 q | k rows and bf16 V^T panels out -------
struct QkvArgs {
  const uint16_t* xp;                 // [T][256] hm16 rows
  const uint16_t* wf; float acc_scale; const float* bias;   // Wqkv re-ordered to [q | k | v] (768 x 256) or [to_qk ; to_v] (512 x 256), fragment order (natural k)
  const float* rot4; long long rot_stride;                  // [16][rot_stride] float4 rotary table (launch_rot_table) (self block)
  uint16_t* qkb; int ldyb;            // bf16 rows: q | k (ldyb = 512) or qk (ldyb = 256)
  uint16_t* vt; int npad;             // V^T bf16 [slot][4][64][npad]
  float qscale;                       // self: multiplied into q after the rotation (dim_head^-0.5)
  float scale;                        // cross: multiplied into qk (dim_head^-0.25)
  int vt_perm;                        // bit 0: keys permuted inside 16-groups (k_attn_bf16_v5); bit 1: timing probe, skip the V^T stores
  int T;                              // tokens, a multiple of 128
  int half_fmt;                       // 0: qkb / vt hold bf16; 1: fp16 (GN_PREC_F16X2_F16_ATTN)
  unsigned int* ovf;                  // half_fmt 1: domain guard word raised when a q / k / v value does not fit fp16, or nullptr
  long long* dbg_ts;                  // developer: nullptr, or [blocks][8 waves][8] s_memtime stamps
  const int* tiles = nullptr;         // work list of the call (launch_tile_lists), or nullptr = every 128-token tile of T
  int products = 2;                   // fp16 partial products per output: 2 = x_h w_h + x_h w_m (the outputs are rounded to 16 bits anyway), 3 = + x_m w_h
                                      // (per context: gn_debug_set_variant(ctx, 27, .))
  int ncu = 0;                        // compute units of the context's device; 0 = ask the current device
};
void launch_qkv(const QkvArgs& a, bool cross, hipStream_t s);
// small grids (one or two pairs; gn_skinny.hip): 32 tokens x 128 features per workgroup, no LDS -- a.tiles / a.vt_perm / a.dbg_ts are not used
void launch_skinny_qkv(const QkvArgs& a, bool cross, hipStream_t s);
// the block tail of a small grid as two launches (gn_skinny.hip): h = composed ffn.0 over [x | ctx] (f32 rows); LayerNorm + GELU of the workgroup's
// rows into LDS, x += ffn.3(g) + b2 in place (hm16 rows, optional f32 copy)
struct SkinnyTailArgs {
  const uint16_t* xp = nullptr; const uint16_t* cp = nullptr;       // [T][256] hm16 residual stream and attention output
  const uint16_t* w1 = nullptr; float w1_scale = 1.f; const float* b1 = nullptr;   // composed ffn.0 [512][512], fragment order (natural k), and its bias
  float* h = nullptr;                                               // [T][512] f32 hidden rows (k_skinny_h out, k_skinny_out in)
  const float* ln_g = nullptr; const float* ln_b = nullptr;         // LayerNorm(512) weight / bias
  const uint16_t* w2 = nullptr; float w2_scale = 1.f; const float* b2 = nullptr;   // ffn.3 [256][512], fragment order with NATURAL k
  uint16_t* xp_out = nullptr; float* y = nullptr;                   // hm16 output rows (may be xp) and optional f32 copy
  unsigned int* ovf = nullptr; int T = 0;
};
void launch_skinny_h(const SkinnyTailArgs& a, hipStream_t s, int variant = 0);
void launch_skinny_out(const SkinnyTailArgs& a, hipStream_t s, int variant = 0);
int device_cu_count();                // multiProcessorCount of the CURRENT device (no caching across devices)

// ---- attention --------------------------------------------------------------------------------
struct AttnArgs {
  const uint16_t* qb; int ldqb;   // bf16 variants of q / k (row-major) and V^T ([BS][4][64][npad])
  const uint16_t* kb; int ldkb;
  const uint16_t* vt;
  const float* q; int ldq;   // rows = tokens, head h at column offset h*64 from the given pointer
  const float* k; int ldk;
  const float* v; int ldv;
  float* out; int ldo;
  uint16_t* outp;                        // optional hm16 output (row pitch ldo values) instead of `out` (k_attn_bf16_v5)
  const int32_t* nvalid;     // [BS] valid token count per (pair, side)
  int npad;                  // tokens per (pair, side) slot
  int cross;                 // 1: keys/values come from the other side of the same pair (bs ^ 1)
  float qscale;              // multiplied into q before QK^T
  int BS;                    // number of (pair, side) slots
  unsigned int* ovf;         // f16x2 domain guard word for the hm16 output, or nullptr
  int half_fmt = 0;          // k_attn16_v5: 0 = qb / kb / vt hold bf16 (v_mfma_f32_32x32x16_bf16), 1 = fp16 (v_mfma_f32_32x32x16_f16): the arithmetic of the
                             // reference's own CUDA path (kornia casts q, k, v to half for SDPA); probabilities are rounded to the same format
  int nsplit = 1;            // k_attn_bf16_v5 on small grids: key ranges per (slot, head, query block), merged by the last workgroup to finish
  float* part = nullptr;     // [slot][head][query block][split][4 waves][34][64] partial results
  unsigned int* tickets = nullptr;   // [slot][head][query block], zero between launches
  const int* tiles = nullptr;        // k_attn_pw: work list of the call (launch_tile_lists), or nullptr = every 256-query block of every (slot, head)
  int ncu = 0;                       // compute units of the context's device; 0 = ask the current device
};
// Work lists of one matcher call (round 4): the 128-token tiles and the (slot, head, 256-query block) items that hold at least one valid token, in
// ascending order.  Layout (ints): [0] number of tiles, [1] number of attention items, [kTileListBase ..) tile indices (token / 128), then at
// [kTileListBase + T / 128 ..) the items ((slot * 4 + head) * (npad / 256) + block).  The kernels that take a list run as many workgroups as CUs and
// walk it: tiles of padding cost nothing -- not even a workgroup dispatch (~40 ns each on this part, which is what skipping them inside the
// workgroup still paid).  Rows of padding are never written by those kernels; they keep whatever finite values they had (the workspaces are
// zero-initialised), and no valid token reads them (keys are masked, everything else is row-wise).
constexpr int kTileListBase = 16;
void launch_tile_lists(const int32_t* nvalid, int BS, int npad, int* lists, unsigned long long* feedback /* pinned host word or nullptr */, hipStream_t s);
void launch_attention_f32(const AttnArgs& a, hipStream_t s);
void launch_attention_bf16(const AttnArgs& a, hipStream_t s);
void launch_attention_bf16_v2(const AttnArgs& a, hipStream_t s);
bool launch_attention_pw(const AttnArgs& a, int ablate, hipStream_t s);   // k_attn_pw (gn_attention_pw.hip): bulk grids, npad % 256 == 0; false = not applicable
void launch_pack_attn_bf16(const AttnArgs& a, uint16_t* qkb, uint16_t* vtb, hipStream_t s);   // (a.half_fmt selects bf16 / fp16)   // f32 rows -> the bf16 layouts k_attn_bf16_v5 reads (test entry)
extern thread_local long long* g_attn_stamps;   // developer (knob 1 = 73): k_attn_pw writes s_memtime phase stamps here (the idle sim buffer)
extern thread_local int g_attn_variant;  // developer knob: 4 = k_attn_bf16_v5 (default), 41 / 42 = its timing-only ablations

// ---- elementwise / small kernels ----------------------------------------------------------------
struct PrepArgs {
  const float* desc_q; const float* kpt_q; const int32_t* n_q; int stride_q;
  const float* desc_r; const float* kpt_r; const int32_t* n_r; int stride_r;
  int kpt_format; int B; int npad;
  const float* wr;          // posenc.Wr.weight [32][4]
  float* desc;              // [B*2*npad][128] RootSIFT (zeros in padding)
  float* kxy;               // [B*2*npad][2] keypoint centres (for the gather)
  float* cos_t; float* sin_t;  // [B*2*npad][32]
  int32_t* nvalid;          // [B*2] (n_q[b], n_r[b]) interleaved
  float* extent;            // [B*2][2] (max_x, max_y)
  float size_q[2], size_r[2];   // image size (w, h) per side for normalize_keypoints; <= 0: the keypoint extent (hw = None in kornia's LightGlueMatcher)
  int feature;              // 0: SIFT (128-d descriptors, RootSIFT, Wr [32][4] on (x, y, scale, ori)); 1: SuperPoint-style (256-d descriptors used as they are, Wr [32][2] on (x, y))
  float* x; uint16_t* xp;   // feature 1: the descriptors go straight into the residual stream, f32 [T][256] and / or hm16 (either may be nullptr)
};
void launch_prep(const PrepArgs& a, hipStream_t s);
void launch_ln_gelu(float* h, const float* gamma, const float* beta, int rows, hipStream_t s,
                    uint16_t* hp = nullptr, unsigned int* ovf = nullptr);   // hp: write hm16 rows [rows][512] instead of h
void launch_matchability(const float* x, const float* w, const float* b, float* ls, int rows, hipStream_t s);

struct HeadArgs {
  const float* sim;         // [B][npad][npad]
  const float* ls;          // [B*2*npad] logsigmoid(matchability)
  const int32_t* nvalid;    // [B*2]
  int B; int npad; float threshold;
  float* rowmax; float* rowlog; float* colmax; float* collog;   // [B][npad] each
  int32_t* m0; float* max0; int32_t* m1;                        // [B][npad]
  int64_t* idx; float* score; int32_t* n_match; int kmax;       // outputs
  const unsigned int* ovf;  // f16x2 domain guard word: non-zero -> the call reports zero matches
  // fused head (launch_match_head_fused): the similarity matrix is recomputed tile by tile from the projected descriptors
  const void* md;           // [B*2*npad] rows of 256 values, 1 KB each: f32 (md_f32) or hm16
  int md_f32;
  float* cpart_m; float* cpart_s; int32_t* cpart_i;   // [B][npad / 32][npad] column partials: max / sum of exponentials (sweep 1), best score / row (sweep 2)
  float* rpart_a; float* rpart_b;                      // [B][8][npad] row partials per column split: (max, sum) or (best score, column as int bits)
  unsigned int* tickets;    // [B][2] arrival counters of the two sweeps, zero between calls
  long long* dbg_ts;        // developer: nullptr, or [2][B][npad / 128][8 splits][8] s_memtime phase stamps
  // margin certificate (gn_set_certify): the second sweep also keeps the RUNNER-UP of every row / column maximum; the last workgroup of a pair
  // raises uncert[b] when a decision of that pair lies within the stated arithmetic error of flipping (see k_head_fused)
  float* max0b;             // [B][npad] runner-up score of every row
  float* rpart_c;           // [B][8][npad] row partials per column split: runner-up
  int32_t* uncert;          // [B] out: 0 certified, 1 a decision inside the margin, 2 the fp16-range guard tripped; nullptr = no certificate
  float cert_eps;           // bound on |P_this mode - P_exact| the certificate is stated for
  int32_t* uncert_alt = nullptr;   // [B] out (optional): the same test for cert_eps_alt -- what the OTHER block-tail level's certificate would say about these
  float cert_eps_alt = -1.f;       // scores (gn_set_ffn_products(0): the context weighs the two levels' re-run fractions against each other); 0 / 1
};
void launch_match_head(const HeadArgs& a, hipStream_t s);
void launch_match_head_fused(const HeadArgs& a, hipStream_t s);

struct GatherArgs {
  const float* kpt_q; int stride_q; const float* kpt_r; int stride_r; int kpt_format;
  const int64_t* idx; const int32_t* n_match; int kmax; int B;
  const uint8_t* dem; int H; int W;
  float* mkp_q; float* obj;
};
void launch_gather(const GatherArgs& a, hipStream_t s);

struct HypResult { double R[9]; double t[3]; int good; int valid; };   // one RANSAC hypothesis

struct PnpArgs {
  const float* obj; const float* img; const int32_t* n_pts; int kstride; int B;
  double fx, fy, cx, cy;
  int iterations; float reproj; double confidence; int min_pts;
  double* R; double* t; int32_t* n_inliers; uint8_t* ok;
  uint8_t* mask_ws;          // [B][16][kstride] scratch: one inlier mask per concurrent hypothesis
  HypResult* hyp;            // [B][16]
  float* pts_ws;             // [B][kstride][5] scratch: a pair's compacted inliers when they exceed k_pnp_refine's LDS capacity (2048)
  long long* dbg_ts;         // developer: nullptr, or [B][16 + 1][16] s_memtime phase stamps (k_pnp_hyp waves, then k_pnp_refine)
};
void launch_pnp(const PnpArgs& a, hipStream_t s);
void launch_epnp_debug(const double* pws, const double* us, double* out, int n, hipStream_t s);

// ---- visual-odometry matcher (TwistNode): brute-force 2-NN + ratio test ------------------------------------
struct VoArgs {
  const float* desc_q; const int32_t* n_q; int stride_q;
  const float* desc_r; const int32_t* n_r; int stride_r;
  int B; int npad; double ratio;
  float* desc;              // [B*2*npad][128] packed descriptors (zeros in padding)
  float* norm2;             // [B*2*npad] squared norms
  int32_t* nvalid;          // [B*2]
  const float* sim;         // [B][npad][npad] q.r panel
  int32_t* nn_idx; float* nn_dist;   // [B][npad][2] best / second best train index and distance per query
  uint8_t* good;            // [B][npad] ratio test passed
  int64_t* idx; float* dist; int32_t* n_good; int kmax;   // compacted good matches (query index, train index), distance
};
void launch_vo_pack(const VoArgs& a, hipStream_t s);
void launch_vo_knn2(const VoArgs& a, hipStream_t s);

// ---- StereoNode reference raster: rotate about the centre + centre crop (2 u8 channels) --------------------
struct WarpArgs {
  const uint8_t* src0; const uint8_t* src1;   // fused: BGR [H][W][3] and DEM [H][W]; else src0 = stack [H][W][2], src1 unused
  int H, W;
  double M[6];                                // INVERTED affine map (dst -> src), as cv::warpAffine computes it
  int dx, dy, crop_h, crop_w;                 // crop window of the (virtual) rotated W x H image
  uint8_t* out0; uint8_t* out1;               // out1 != nullptr: two planes [crop_h][crop_w]; else interleaved [crop_h][crop_w][2]
};
void launch_rotate_crop(const WarpArgs& a, bool fused_gray, hipStream_t s);

// ---- SIFT (cv2.SIFT_create().detectAndCompute) -------------------------------------------------------------
constexpr int kSiftMaxOctaves = 12;
struct SiftOctave { float* gauss[6]; float* dog[5]; int w, h; long long stride; };   // stride: floats between consecutive images of a batch
struct SiftPyramid { SiftOctave oct[kSiftMaxOctaves]; int n_oct; };
struct SiftKeypoint { float x, y, size, angle, response; int octave; };
void sift_gaussian_kernel(double sigma, std::vector<float>& k);
void sift_base_blur(const uint8_t* gray, int B, int h, int w, float* scratch, float* tmp, float* out, long long out_stride, const float* dk, int n, int* counters, hipStream_t s);
// dog = out - in; in_step 2 reads every second pixel of a source image of row stride in_w (half_scratch: only used for non-stock kernel sizes)
void sift_blur(int B, long long stride_in, long long stride_out, const float* in, float* tmp, float* out, int w, int h, const float* dk, int n, hipStream_t s,
               float* dog = nullptr, int in_step = 1, int in_w = 0, float* half_scratch = nullptr);
int sift_tail_first(const SiftPyramid& py, const int* ksize);
void sift_tail(const SiftPyramid& py, int B, int o_first, const float* dk, const int* koff, const int* ksize, hipStream_t s);
void sift_find(const SiftPyramid& py, int B, float threshold, int4* cand, int* counters, int max_cand, hipStream_t s);
// counters: int[4] per image = {candidates, raw keypoints, final keypoints, -}; kp: per image [max_raw raw | max_raw sorted | final], kp_stride records apart
void sift_refine(const SiftPyramid& py, int B, const int4* cand, int* counters, int max_cand, SiftKeypoint* kp, long long kp_stride, int max_raw, hipStream_t s);
void sift_descriptors(const SiftPyramid& py, int B, const SiftKeypoint* kp_final, long long kp_stride, const int* counters, int max_n, float* desc, long long out_stride, hipStream_t s);
void sift_sort_dedup(int B, SiftKeypoint* kp, long long kp_stride, int* counters, int max_raw, int max_out,
                     float* kpt_xysa, float* response, int32_t* octave, long long out_stride, hipStream_t s);

// ---- SuperPoint extractor (gn_superpoint.hip) ------------------------------------------------------------------
void sp_weight_fragments(const float* w, int Cout, int Cin, int taps, int Cout_pad, float* out);   // host arrays
void sp_conv1(const float* in, const float* w, const float* bias, float* out, int B, int H, int W, hipStream_t s, int out_half = 0, unsigned int* ovf = nullptr);
void sp_conv_fused1(const float* gray, const float* w1, const float* b1, int B, int H, int W, const float* bias, float* out, int Cout_pad, hipStream_t s,
                    const uint16_t* wfh, float acc_scale, unsigned int* ovf);   // layers 0 + 1 in one launch (split-fp16 mode, hm16 records out, 2 x 2 max-pool fused)
extern int g_sp_fuse1;   // out_half: 0 f32, 1 fp16, 2 hm16 records (guarded by ovf)
extern int g_sp_conv_s, g_sp_nms_fused, g_sp_select_stream;
void sp_weight_fragments_hm16(const float* w, int Cout, int Cin, int taps, int Cout_pad, float scale, uint16_t* out);   // host arrays; out: 2 * Cout_pad * taps * Cin halfs
void sp_conv(const float* in, int B, int H, int W, int Cin, const float* wf, const float* bias, float* out, int Cout_pad, int taps, int relu, hipStream_t s,
             const uint16_t* wfh = nullptr, float acc_scale = 1.f, unsigned int* ovf = nullptr, int pool = 0,
             int single_product = 0, int in_half = 0, int out_half = 0, long long* dbg_ts = nullptr);   // single_product: one fp16 product per block instead of three (GN_SP_FP16);
                                                                                 // in_half / out_half: 1 = fp16 NHWC activations, 2 = hm16 records (k_sp_conv_s)   // wfh != nullptr: split-fp16 arithmetic; pool: fused 2 x 2 max-pool, out is [H/2][W/2]
void sp_pool(const float* in, float* out, int B, int H, int W, int C, hipStream_t s);
void sp_scores(const float* logits, int cp, float* scores, int B, int h, int w, hipStream_t s);
void sp_nms(const float* scores, int B, int H, int W, int r, float* pooled, float* tmp, float* mask, float* supp, float* aux, hipStream_t s);
void sp_select(const float* nms, int B, int H, int W, float thr, int border, int* cand, int* counts, int cap, int k,
               float* kpt_xysa, float* score, int* kp_index, long long out_stride, hipStream_t s);
void sp_describe(const float* dmap, int B, int h, int w, const float* kpt_xysa, const int* counts, long long out_stride, int max_k, float* desc, hipStream_t s);

// ---- bf16 helpers -------------------------------------------------------------------------------
void launch_cast_bf16(const float* in, uint16_t* out, long long n, hipStream_t s);
void launch_split3_bf16(const float* in, uint16_t* planes, long long n, hipStream_t s);  // planes[3][n]
void launch_split2_f16(const float* in, uint16_t* planes, long long n, float scale, hipStream_t s);  // planes[2][n] = fp16 split of in * scale
void launch_split_hm16(const float* in, uint16_t* out, long long rows, int cols, float scale, hipStream_t s);  // hm16 rows of in * scale
void launch_rot_table(const float* cos_t, const float* sin_t, float* rot4, int T, long long stride, hipStream_t s);   // [16][stride] float4 (cos, cos, sin, sin) for k_qkv

}  // namespace gn
