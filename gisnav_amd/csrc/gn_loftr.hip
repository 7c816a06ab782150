// LoFTR (Sun et al., CVPR 2021; kornia.feature.LoFTR, default "outdoor" configuration) on gfx950, f32 throughout -- the matcher
// BASELINE.json's north_star and configs[1] name ("Batch-1 640x480 pair, LoFTR matcher ... HIP conv + attention kernels, fp32").  The
// reference tree does not contain it at this tag (docs/vitepress/docs/glossary.md:186 is all that is left); the
// specification is the published architecture as restated in oracle/loftr.py (PARITY UNPINNED: kornia is not importable here).
//
//   backbone     ResNetFPN_8_2: 7x7/2 stem, three stages of two BasicBlocks (128 | 196 | 256 channels at 1/2 | 1/4 | 1/8), FPN head.
//                k_lf_conv<KS, S, ...>: implicit-GEMM convolution on the EXACT f32 matrix instruction (v_mfma_f32_32x32x2_f32), weights
//                in MFMA fragment order straight from L2, pixels from an LDS halo tile, eval-mode BatchNorm as a per-channel affine +
//                residual add + (leaky) ReLU in the epilogue; NHWC f32 activations (196 channels padded to 224).
//   coarse       sine position encoding, 4 x (self, cross) encoder layers with LINEAR attention (phi = elu + 1): the projections / merge /
//   transformer  MLP are the exact-f32 GEMM of gn_gemm.hip (k_gemm_f32_v3), K^T V (32 x 32 per head) is a two-stage token reduction,
//                the rest is row-wise (LayerNorm, ReLU, residual).  cross: feat0 <- layer(feat0, feat1), THEN feat1 <- layer(feat1, feat0').
//   coarse       conf = softmax_rows(S) * softmax_cols(S), S = <f0, f1> / (256 * 0.1) over hc*wc x hc*wc (4800 x 4800 at 640x480), threshold
//   matching     0.2, border 2, mutual maxima, matches in ascending query cell -- row / column statistics in three passes over S.
//   fine level   5x5 windows of the 1/2-resolution FPN map around every coarse match, merged with the coarse features, one (self, cross)
//                encoder pass per window pair, softmax correlation heat map -> expectation -> sub-pixel offset on the reference side.
#include "gn_common.h"

#include <cmath>
#include <cstring>

namespace gn {
namespace {

thread_local std::string g_lf_err;
constexpr int kLfDim = 256, kLfHeads = 8, kLfFine = 128, kLfWW = 25;

// ------------------------------------------------------------------------------------------------ stem: 7x7 stride 2, 1 -> 128
// thread -> (output pixel, 16-channel group); weights + affine in LDS.  Round 5: the weights sit TRANSPOSED there ([tap][channel]) and a thread reads
// its 16 channels of a tap as four 16-byte pieces (the [channel][tap] form read one float per multiply-add through 4-way bank conflicts: the
// kernel was LDS-bound at 181 us for 79 MB of output); every channel still accumulates its taps in the order 0 .. 48: the same bits.  OLD = the
// form of rounds 3-4 (developer knob 42, bit 2)
template <bool OLD>
__global__ __launch_bounds__(256) void k_lf_conv1(const float* in, const float* w, const float* scale, const float* shift, float* out, int H, int W) {
  __shared__ __attribute__((aligned(16))) float ws[128 * 49 + 256];
  for (int q = threadIdx.x; q < 128 * 49 + 256; q += 256) {
    float x;
    if (q < 128 * 49) x = OLD ? w[q] : w[128 * 49 + q];       // (the [tap][channel] copy lf_finalise put behind the [channel][tap] one)
    else x = q < 128 * 49 + 128 ? scale[q - 128 * 49] : shift[q - 128 * 49 - 128];
    ws[q] = x;
  }
  __syncthreads();
  const int Ho = H / 2, Wo = W / 2;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const int grp = (int)(idx & 7);
  const long long pix = idx >> 3;
  if (pix >= (long long)Ho * Wo) return;
  const int y = (int)(pix / Wo), x = (int)(pix - (long long)y * Wo);
  const float* img = in + (long long)blockIdx.z * H * W;
  float v[49];
#pragma unroll
  for (int t = 0; t < 49; ++t) {
    const int yy = 2 * y + t / 7 - 3, xx = 2 * x + t % 7 - 3;
    v[t] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? img[(long long)yy * W + xx] : 0.f;
  }
  float* o = out + ((long long)blockIdx.z * Ho * Wo + pix) * 128 + grp * 16;
  if (OLD) {
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      f32x4 r;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = grp * 16 + c4 * 4 + e;
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 49; ++t) acc = fmaf(ws[c * 49 + t], v[t], acc);
        r[e] = fmaxf(acc * ws[128 * 49 + c] + ws[128 * 49 + 128 + c], 0.f);
      }
      *reinterpret_cast<f32x4*>(o + c4 * 4) = r;
    }
  } else {
    // lane (pixel, grp) owns the channels 32 c4 + 4 grp .. + 3, c4 = 0 .. 3: the eight lanes of a pixel read 128 contiguous bytes of a tap's weights
    // and store 128 contiguous bytes of the pixel's row per instruction (with 16 channels in a row per lane every store instruction wrote 16 bytes
    // out of every 64)
    f32x4 acc[4];
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) acc[c4] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 49; ++t)
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) {
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(ws + t * 128 + c4 * 32 + grp * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[c4][e] = fmaf(w4[e], v[t], acc[c4][e]);
      }
    float* const op = out + ((long long)blockIdx.z * Ho * Wo + pix) * 128 + grp * 4;
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      const f32x4 sc = *reinterpret_cast<const f32x4*>(ws + 128 * 49 + c4 * 32 + grp * 4), sh = *reinterpret_cast<const f32x4*>(ws + 128 * 49 + 128 + c4 * 32 + grp * 4);
      f32x4 r;
#pragma unroll
      for (int e = 0; e < 4; ++e) r[e] = fmaxf(acc[c4][e] * sc[e] + sh[e], 0.f);
      *reinterpret_cast<f32x4*>(op + c4 * 32) = r;
    }
  }
}

// ------------------------------------------------------------------------------------------------ general convolution
struct LfConvArgs {
  const float* in; int Hin, Win, Cin;       // NHWC f32, Cin a multiple of 32
  const float* wf;                          // sp_weight_fragments order: [Cout/32][KS*KS][Cin/8][64 lanes][4]
  const float* scale; const float* shift;   // [Cout] eval-mode BatchNorm as an affine map (1, 0 for a plain convolution)
  const float* resid;                       // optional NHWC [Hout][Wout][Cout]: added before the activation
  float* out; int Hout, Wout, Cout;         // Cout a multiple of 32 (padded channels carry zero weights, scale 1, shift 0)
  int act;                                  // 0 none, 1 ReLU, 2 LeakyReLU(0.01)
  // split-fp16 arithmetic (k_lf_conv_h): weights as fp16 pairs in fragment order (sp_weight_fragments_hm16), the affine scale with the
  // inverse of the weights' power-of-two scale folded in, the fp16-range guard word
  const uint16_t* wfh; const float* scale_h; unsigned int* ovf;
  int cin_real;                             // input channels that exist (196 of Cin = 224 in LoFTR's middle layers): 8- (k_lf_conv) / 16-channel (k_lf_conv_h) steps behind them multiply zeros and are skipped
};

// grid (ceil(Wout / 32), ceil(Hout / (4 RPW)), N * ceil(Cout / 64)); 4 waves, wave w = output rows [RPW w, RPW w + RPW) of the tile,
// 32 output columns, two 32-channel tiles.  Halo tile: ((4 RPW - 1) S + KS) x (31 S + KS) input pixels x CH channels, 16-byte chunk
// c of pixel column lx at position c ^ sw(lx).
// PF (round 5): the halo tile of the NEXT channel slice is requested into registers behind the first tap of the current one and written to
// LDS at the slice boundary -- with one or two workgroups per CU nothing else hid the global latency of the staging pass (~7 us per slice
// beside ~17 us of MFMAs at RPW = 2; profiles/r05e_loftr_layers_exact_f32.txt); same values in the same LDS places: bitwise the same output
// FAST (late round 5): every output-channel group has both 32-channel tiles and every 32-channel slice is whole (Cout % 64 == 0, real Cin % 32 == 0:
// LoFTR's 128- and 256-channel layers) -- the MFMA stream of a tap is then ONE basic block: counters showed the matrix pipe busy 0.60-0.71 at 2.3 GHz, and
// the disassembly why: the run-time tests of `two` and of the channel-step limit sat between the MFMAs, so every 32-channel step began with its two
// ds_read_b128 and waited for them (the scheduler does not move loads across branches).  FAST requests the next step's fragments before the current
// step's MFMAs.  Same MFMAs in the same order: same bits.
template <int KS, int S, int RPW, int CH, bool PF, bool FAST = false>
__global__ __launch_bounds__(256, (PF && RPW <= 2 && S == 1) ? 2 : 1) void k_lf_conv(LfConvArgs a) {
  constexpr int TAPS = KS * KS, PAD = KS / 2;
  constexpr int TH = 4 * RPW, LH = (TH - 1) * S + KS, LW = 31 * S + KS, NCH = CH / 4;
  __shared__ __attribute__((aligned(16))) float tile[LH * LW * CH];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, ql = lane & 31;
  const int ogroups = (a.Cout + 63) / 64;
  const int img = blockIdx.z / ogroups, og = blockIdx.z % ogroups;
  const bool two = FAST || 64 * og + 32 < a.Cout;           // the last group of a 32 (mod 64) channel count has one tile
  const int x0 = blockIdx.x * 32, y0 = blockIdx.y * TH;
  const int gy0 = y0 * S - PAD, gx0 = x0 * S - PAD;
  const float* in = a.in + (long long)img * a.Hin * a.Win * a.Cin;
  const int csteps = a.Cin / 8, rsteps = FAST ? csteps : (a.cin_real + 7) / 8;
  const f32x4* wf = reinterpret_cast<const f32x4*>(a.wf) + lane;
  auto sw = [](int lx) { const int u = lx / S; return (u ^ (u >> 3)) & (NCH - 1); };

  f32x16 acc[2][RPW];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < RPW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // weight fragments of one tap (CH / 8 channel steps x two 32-channel tiles) travel L2 -> registers one tap AHEAD of the MFMAs that use
  // them: with one wave per SIMD nothing else hides the ~1 us of L2 latency (the first build fetched them right in front of each step's
  // MFMAs and ran at 45 TF)
  constexpr int NS = CH / 8;
  f32x4 wcur[NS][2], wnxt[NS][2];
  auto load_w = [&](f32x4 (&w)[NS][2], int c0_, int tap_) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (!FAST && c0_ / 8 + s >= rsteps) continue;
      w[s][0] = wf[(size_t)(((2 * og) * TAPS + tap_) * csteps + (c0_ / 8 + s)) * 64];
      if (FAST || two) w[s][1] = wf[(size_t)(((2 * og + 1) * TAPS + tap_) * csteps + (c0_ / 8 + s)) * 64];
    }
  };
  load_w(wcur, 0, 0);
  constexpr int NP = LH * LW * NCH, NQ = (NP + 255) / 256, SB = 8;
  f32x4 pre[PF ? NQ : 1];
  auto fetch = [&](int c0_) __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < NQ; ++e) {
      const int q = e * 256 + tid;
      const int pix = q / NCH, chunk = q - pix * NCH;
      const int ly = pix / LW, lx = pix - ly * LW;
      const int gy = gy0 + ly, gx = gx0 + lx;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      pre[e] = z;
      if (q < NP && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win)
        pre[e] = *reinterpret_cast<const f32x4*>(in + ((long long)gy * a.Win + gx) * a.Cin + c0_ + chunk * 4);
    }
  };
  auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < NQ; ++e) {
      const int q = e * 256 + tid;
      if (q >= NP) continue;
      const int pix = q / NCH, chunk = q - pix * NCH;
      const int lx = pix % LW;
      *reinterpret_cast<f32x4*>(tile + pix * CH + ((chunk ^ sw(lx)) * 4)) = pre[e];
    }
  };
  if (PF) fetch(0);
  for (int c0 = 0; c0 < a.Cin; c0 += CH) {
    __syncthreads();
    if (PF) commit();
#pragma unroll 1
    for (int q0 = 0; !PF && q0 < NQ; q0 += SB) {
      f32x4 v[SB];
#pragma unroll
      for (int e = 0; e < SB; ++e) {
        const int q = (q0 + e) * 256 + tid;
        const int pix = q / NCH, chunk = q - pix * NCH;
        const int ly = pix / LW, lx = pix - ly * LW;
        const int gy = gy0 + ly, gx = gx0 + lx;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        v[e] = z;
        if (q < NP && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win)
          v[e] = *reinterpret_cast<const f32x4*>(in + ((long long)gy * a.Win + gx) * a.Cin + c0 + chunk * 4);
      }
#pragma unroll
      for (int e = 0; e < SB; ++e) {
        const int q = (q0 + e) * 256 + tid;
        if (q >= NP) continue;
        const int pix = q / NCH, chunk = q - pix * NCH;
        const int lx = pix % LW;
        *reinterpret_cast<f32x4*>(tile + pix * CH + ((chunk ^ sw(lx)) * 4)) = v[e];
      }
    }
    __syncthreads();
#pragma unroll 1
    for (int tap = 0; tap < TAPS; ++tap) {
      const int ty = tap / KS, tx = tap - ty * KS;
      // next tap of this slice, or the first tap of the next slice (its request then also overlaps the staging of that slice)
      const bool last_tap = tap + 1 == TAPS;
      if (!last_tap || c0 + CH < a.Cin) load_w(wnxt, last_tap ? c0 + CH : c0, last_tap ? 0 : tap + 1);
      if (PF && tap == 0 && c0 + CH < a.Cin) fetch(c0 + CH);     // behind the next tap's weights: it has two taps of MFMAs to land in
      if (FAST) {
        f32x4 fb2[2][RPW];
        auto read_fb = [&](int buf, int s) __attribute__((always_inline)) {
#pragma unroll
          for (int j = 0; j < RPW; ++j) {
            const int ly = (RPW * wave + j) * S + ty, lx = ql * S + tx;
            fb2[buf][j] = *reinterpret_cast<const f32x4*>(tile + (ly * LW + lx) * CH + (((2 * s + hh) ^ sw(lx)) * 4));
          }
        };
        read_fb(0, 0);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          if (s + 1 < NS) read_fb((s + 1) & 1, s + 1);
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < RPW; ++j) {
              acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[s][0][e], fb2[s & 1][j][e], acc[0][j], 0, 0, 0);
              acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[s][1][e], fb2[s & 1][j][e], acc[1][j], 0, 0, 0);
            }
        }
      } else {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        if (c0 / 8 + s >= rsteps) continue;
        f32x4 fb[RPW];
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
          const int ly = (RPW * wave + j) * S + ty, lx = ql * S + tx;
          fb[j] = *reinterpret_cast<const f32x4*>(tile + (ly * LW + lx) * CH + (((2 * s + hh) ^ sw(lx)) * 4));
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int j = 0; j < RPW; ++j) {
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[s][0][e], fb[j][e], acc[0][j], 0, 0, 0);
            if (two) acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[s][1][e], fb[j][e], acc[1][j], 0, 0, 0);
          }
      }
      }
#pragma unroll
      for (int s = 0; s < NS; ++s) { wcur[s][0] = wnxt[s][0]; wcur[s][1] = wnxt[s][1]; }
    }
  }
  // epilogue: lane = pixel (row RPW wave + j, column ql); registers 4 g + c = output channels 64 og + 32 i + 8 g + 4 hh + c
  const int gx = x0 + ql;
  float* out = a.out + (long long)img * a.Hout * a.Wout * a.Cout;
  const float* res = a.resid ? a.resid + (long long)img * a.Hout * a.Wout * a.Cout : nullptr;
#pragma unroll
  for (int j = 0; j < RPW; ++j) {
    const int gy = y0 + RPW * wave + j;
    if (gy >= a.Hout || gx >= a.Wout) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (i == 1 && !two) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = 64 * og + 32 * i + 8 * g + 4 * hh;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + c), sh = *reinterpret_cast<const f32x4*>(a.shift + c);
        f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        v = v * sc + sh;
        const long long o = ((long long)gy * a.Wout + gx) * a.Cout + c;
        if (res) v += *reinterpret_cast<const f32x4*>(res + o);
        if (a.act == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        else if (a.act == 2) { v.x = v.x > 0.f ? v.x : 0.01f * v.x; v.y = v.y > 0.f ? v.y : 0.01f * v.y; v.z = v.z > 0.f ? v.z : 0.01f * v.z; v.w = v.w > 0.f ? v.w : 0.01f * v.w; }
        *reinterpret_cast<f32x4*>(out + o) = v;
      }
    }
  }
}

// The same convolution in the SPLIT-fp16 arithmetic of the matcher's f16x2 GEMMs (gn_gemm_p2.hip, gn_superpoint.hip HM = 1): every f32 operand
// as two fp16 terms (x = h + m, round to nearest, 22 significant bits), three v_mfma_f32_32x32x16_f16 per 16 input channels (W_m X_h, W_h X_m,
// W_h X_h, small terms first), f32 accumulation -- f32-ACCURATE (error at the level of an f32 accumulation) at 5 x the matrix-pipe rate of
// v_mfma_f32_32x32x2_f32.  Activations stay f32 in memory and are split while the halo tile is staged (hm16 pixel layout: per 16 channels 16
// high terms then 16 residual terms); weights are pre-split at load time with a power-of-two scale (folded into the epilogue's affine map)
// and, per tap, fetched ONCE per workgroup into a double-buffered LDS block (per-wave fetches would make the L2 -> CU ingest the bottleneck at
// this MFMA rate).  An activation that does not fit fp16 raises a.ovf: the caller repeats the forward on the exact kernels.
template <int KS, int S, int RPW, int CH, bool PF>      // PF: as in k_lf_conv
__global__ __launch_bounds__(256, (PF && RPW <= 2 && S == 1) ? 2 : 1) void k_lf_conv_h(LfConvArgs a) {
  typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
  typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
  constexpr int TAPS = KS * KS, PAD = KS / 2;
  constexpr int TH = 4 * RPW, LH = (TH - 1) * S + KS, LW = 31 * S + KS, NP16 = CH / 4;      // NP16: 16-byte pieces per pixel (CH * 4 B / 16)
  constexpr int NKS = CH / 16;                                                               // k-steps (16 channels) per slice
  constexpr int WB = 2 * NKS * 2 * 1024;                                                     // bytes of one tap's weight block: 2 tiles x k-steps x 2 terms x 1 KB
  __shared__ __attribute__((aligned(16))) unsigned char tb[LH * LW * CH * 4];
  __shared__ __attribute__((aligned(16))) unsigned char wbuf[2 * WB];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, ql = lane & 31;
  const int ogroups = (a.Cout + 63) / 64;
  const int img = blockIdx.z / ogroups, og = blockIdx.z % ogroups;
  const bool two = 64 * og + 32 < a.Cout;
  const int x0 = blockIdx.x * 32, y0 = blockIdx.y * TH;
  const int gy0 = y0 * S - PAD, gx0 = x0 * S - PAD;
  const float* in = a.in + (long long)img * a.Hin * a.Win * a.Cin;
  const int ksteps = a.Cin / 16, rsteps = (a.cin_real + 15) / 16;
  const uint4* wfh = reinterpret_cast<const uint4*>(a.wfh) + lane;
  auto sw = [](int lx) { const int u = lx / S; return (u ^ (u >> 3)) & (NP16 - 1); };

  f32x16 acc[2][RPW];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < RPW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float amax = 0.f;
  // this thread's share of a tap's weight block: WB / 16 pieces of 16 bytes over 256 threads
  constexpr int WPT = WB / 16 / 256;            // 2 (CH = 32) or 1 (CH = 16)
  auto wsrc = [&](int c0_, int tap_, int e) __attribute__((always_inline)) {
    const int blk = WPT * 4 * 0 + 4 * e + (tid >> 6);      // block index inside the tap: ((i * NKS + ks) * 2 + pl), 4 blocks per e
    const int i = blk / (2 * NKS), ks = (blk / 2) % NKS, pl = blk & 1;
    const int tile = 2 * og + ((i == 1 && !two) ? 0 : i);  // a missing second tile re-reads the first (never multiplied)
    return wfh[(size_t)((((size_t)tile * TAPS + tap_) * ksteps + (c0_ / 16 + ks)) * 2 + pl) * 64];
  };
  uint4 wnext[WPT];
#pragma unroll
  for (int e = 0; e < WPT; ++e) wnext[e] = wsrc(0, 0, e);
  int par = 0;
  constexpr int NQF = LH * LW * (CH / 4), NQ = (NQF + 255) / 256, SB = 8;
  f32x4 pre[PF ? NQ : 1];
  auto fetch = [&](int c0_) __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < NQ; ++e) {
      const int q = e * 256 + tid;
      const int pix = q / (CH / 4), chunk = q - pix * (CH / 4);
      const int ly = pix / LW, lx = pix - ly * LW;
      const int gy = gy0 + ly, gx = gx0 + lx;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      pre[e] = z;
      if (q < NQF && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win)
        pre[e] = *reinterpret_cast<const f32x4*>(in + ((long long)gy * a.Win + gx) * a.Cin + c0_ + chunk * 4);
    }
  };
  auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < NQ; ++e) {
      const int q = e * 256 + tid;
      if (q >= NQF) continue;
      const int pix = q / (CH / 4), chunk = q - pix * (CH / 4);
      const int lx = pix % LW;
      const h16x4 h4 = __builtin_convertvector(pre[e], h16x4);
      const h16x4 m4 = __builtin_convertvector(pre[e] - __builtin_convertvector(h4, f32x4), h16x4);
      ovf_track(amax, pre[e].x, pre[e].y); ovf_track(amax, pre[e].z, pre[e].w);
      const int piece = 4 * (chunk >> 2) + ((chunk & 3) >> 1), sub = (chunk & 1) * 8;
      *reinterpret_cast<h16x4*>(tb + pix * (CH * 4) + ((piece ^ sw(lx)) * 16) + sub) = h4;
      *reinterpret_cast<h16x4*>(tb + pix * (CH * 4) + (((piece + 2) ^ sw(lx)) * 16) + sub) = m4;
    }
  };
  if (PF) fetch(0);
  for (int c0 = 0; c0 < a.Cin; c0 += CH) {
    __syncthreads();
    if (PF) commit();
#pragma unroll 1
    for (int q0 = 0; !PF && q0 < NQ; q0 += SB) {
      f32x4 v[SB];
#pragma unroll
      for (int e = 0; e < SB; ++e) {
        const int q = (q0 + e) * 256 + tid;
        const int pix = q / (CH / 4), chunk = q - pix * (CH / 4);
        const int ly = pix / LW, lx = pix - ly * LW;
        const int gy = gy0 + ly, gx = gx0 + lx;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        v[e] = z;
        if (q < NQF && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win)
          v[e] = *reinterpret_cast<const f32x4*>(in + ((long long)gy * a.Win + gx) * a.Cin + c0 + chunk * 4);
      }
#pragma unroll
      for (int e = 0; e < SB; ++e) {
        const int q = (q0 + e) * 256 + tid;
        if (q >= NQF) continue;
        const int pix = q / (CH / 4), chunk = q - pix * (CH / 4);
        const int lx = pix % LW;
        // channels 4 chunk .. + 3 of the slice: k-step chunk >> 2, piece 4 (chunk >> 2) + 2 term + ((chunk & 3) >> 1), 8-byte half (chunk & 1)
        const h16x4 h4 = __builtin_convertvector(v[e], h16x4);
        const h16x4 m4 = __builtin_convertvector(v[e] - __builtin_convertvector(h4, f32x4), h16x4);
        ovf_track(amax, v[e].x, v[e].y); ovf_track(amax, v[e].z, v[e].w);
        const int piece = 4 * (chunk >> 2) + ((chunk & 3) >> 1), sub = (chunk & 1) * 8;
        *reinterpret_cast<h16x4*>(tb + pix * (CH * 4) + ((piece ^ sw(lx)) * 16) + sub) = h4;
        *reinterpret_cast<h16x4*>(tb + pix * (CH * 4) + (((piece + 2) ^ sw(lx)) * 16) + sub) = m4;
      }
    }
#pragma unroll 1
    for (int tap = 0; tap < TAPS; ++tap) {
      const int ty = tap / KS, tx = tap - ty * KS;
      unsigned char* const wb = wbuf + par * WB;
#pragma unroll
      for (int e = 0; e < WPT; ++e) *reinterpret_cast<uint4*>(wb + (4 * e + (tid >> 6)) * 1024 + lane * 16) = wnext[e];
      const bool last_tap = tap + 1 == TAPS;
      if (!last_tap || c0 + CH < a.Cin) {
#pragma unroll
        for (int e = 0; e < WPT; ++e) wnext[e] = wsrc(last_tap ? c0 + CH : c0, last_tap ? 0 : tap + 1, e);
      }
      if (PF && tap == 0 && c0 + CH < a.Cin) fetch(c0 + CH);
      __syncthreads();     // this tap's weights (and, at tap 0, the halo tile) are in LDS; the block written two taps ago is no longer read
#pragma unroll
      for (int s = 0; s < NKS; ++s) {
        if (c0 / 16 + s >= rsteps) continue;
        h16x8 fa[2][2], fb[RPW][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) fa[i][pl] = *reinterpret_cast<const h16x8*>(wb + ((i * NKS + s) * 2 + pl) * 1024 + lane * 16);
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
          const int ly = (RPW * wave + j) * S + ty, lx = ql * S + tx;
#pragma unroll
          for (int pl = 0; pl < 2; ++pl)
            fb[j][pl] = *reinterpret_cast<const h16x8*>(tb + (ly * LW + lx) * (CH * 4) + (((4 * s + 2 * pl + hh) ^ sw(lx)) * 16));
        }
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int j = 0; j < RPW; ++j) {
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][p == 0 ? 1 : 0], fb[j][p == 1 ? 1 : 0], acc[0][j], 0, 0, 0);
            if (two) acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[1][p == 0 ? 1 : 0], fb[j][p == 1 ? 1 : 0], acc[1][j], 0, 0, 0);
          }
      }
      par ^= 1;
    }
  }
  ovf_commit(a.ovf, amax);
  const int gx = x0 + ql;
  float* out = a.out + (long long)img * a.Hout * a.Wout * a.Cout;
  const float* res = a.resid ? a.resid + (long long)img * a.Hout * a.Wout * a.Cout : nullptr;
#pragma unroll
  for (int j = 0; j < RPW; ++j) {
    const int gy = y0 + RPW * wave + j;
    if (gy >= a.Hout || gx >= a.Wout) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (i == 1 && !two) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = 64 * og + 32 * i + 8 * g + 4 * hh;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale_h + c), sh = *reinterpret_cast<const f32x4*>(a.shift + c);
        f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        v = v * sc + sh;
        const long long o = ((long long)gy * a.Wout + gx) * a.Cout + c;
        if (res) v += *reinterpret_cast<const f32x4*>(res + o);
        if (a.act == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        else if (a.act == 2) { v.x = v.x > 0.f ? v.x : 0.01f * v.x; v.y = v.y > 0.f ? v.y : 0.01f * v.y; v.z = v.z > 0.f ? v.z : 0.01f * v.z; v.w = v.w > 0.f ? v.w : 0.01f * v.w; }
        *reinterpret_cast<f32x4*>(out + o) = v;
      }
    }
  }
}

// out = a + bilinear_2x(b), align_corners = True (F.interpolate(scale_factor = 2) of the FPN head); a / out [N][H][W][C], b [N][H/2][W/2][C]
__global__ __launch_bounds__(256) void k_lf_up2_add(const float* a, const float* b, float* out, int H, int W, int C, long long total4) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total4) return;
  const int c4 = (int)(idx % (C / 4));
  long long p = idx / (C / 4);
  const int x = (int)(p % W); p /= W;
  const int y = (int)(p % H); const int n = (int)(p / H);
  const int h2 = H / 2, w2 = W / 2;
  const float sy = (float)(h2 - 1) / (float)(H - 1), sx = (float)(w2 - 1) / (float)(W - 1);
  const float fy = sy * y, fx = sx * x;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < h2 - 1 ? 1 : 0), x1 = x0 + (x0 < w2 - 1 ? 1 : 0);
  const float ly = fy - y0, lx = fx - x0;
  const float* bb = b + (long long)n * h2 * w2 * C + c4 * 4;
  const f32x4 v00 = *reinterpret_cast<const f32x4*>(bb + ((long long)y0 * w2 + x0) * C), v01 = *reinterpret_cast<const f32x4*>(bb + ((long long)y0 * w2 + x1) * C);
  const f32x4 v10 = *reinterpret_cast<const f32x4*>(bb + ((long long)y1 * w2 + x0) * C), v11 = *reinterpret_cast<const f32x4*>(bb + ((long long)y1 * w2 + x1) * C);
  const f32x4 up = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
  const long long o = (((long long)n * H + y) * W + x) * C + c4 * 4;
  *reinterpret_cast<f32x4*>(out + o) = *reinterpret_cast<const f32x4*>(a + o) + up;
}

// ------------------------------------------------------------------------------------------------ token-side kernels
// x[img][l][256] = feat[img][l][256] + pe[l][256] for l < L, 0 for L <= l < Lp
__global__ __launch_bounds__(256) void k_lf_posenc(const float* feat, const float* pe, float* x, int L, int Lp) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;      // float4 index over [2][Lp][64]
  if (idx >= 2LL * Lp * 64) return;
  const int c4 = (int)(idx & 63);
  const long long t = idx >> 6;
  const int img = (int)(t / Lp), l = (int)(t - (long long)img * Lp);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (l < L) v = *reinterpret_cast<const f32x4*>(feat + ((long long)img * L + l) * 256 + c4 * 4) + *reinterpret_cast<const f32x4*>(pe + (long long)l * 256 + c4 * 4);
  *reinterpret_cast<f32x4*>(x + t * 256 + c4 * 4) = v;
}

__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x + 1.f : (expm1f(x) + 1.f); }   // F.elu(x) + 1

// K^T V and sum K of one (sequence, head) over a chunk of tokens.  k / v rows with row pitch ld, head h at column h * HD; grid
// (heads, nsplit, nseq); thread t -> (d = t / (HD / 4), v4 = (t % (HD / 4)) * 4), t < HD * HD / 4.  Partial results:
// part[seq][split][head][HD + 1][HD] (row HD = sum K).  HD = 32 (coarse: 256 / 8) or 16 (fine: 128 / 8).
template <int HD>
__global__ __launch_bounds__(256) void k_lf_kv_partial(const float* k, const float* v, int ld, long long seq_stride, int S, int chunk, float vdiv, float* part, int nsplit, int heads) {
  __shared__ float ks[64][HD + 1], vs[64][HD];
  constexpr int G = HD / 4;
  const int h = blockIdx.x, sp = blockIdx.y, seq = blockIdx.z, tid = threadIdx.x;
  const float* kb = k + seq * seq_stride + h * HD;
  const float* vb = v + seq * seq_stride + h * HD;
  const int s0 = sp * chunk, s1 = min(S, s0 + chunk);
  const bool active = tid < HD * G;
  const int d = tid / G, v4 = (tid % G) * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float ksum = 0.f;
  for (int t0 = s0; t0 < s1; t0 += 64) {
    __syncthreads();
    for (int q = tid; q < 64 * G; q += 256) {
      const int r = q / G, c = (q % G) * 4, s = t0 + r;
      f32x4 kk = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
      if (s < s1) {
        kk = *reinterpret_cast<const f32x4*>(kb + (long long)s * ld + c);
        vv = *reinterpret_cast<const f32x4*>(vb + (long long)s * ld + c);
        kk.x = elu1(kk.x); kk.y = elu1(kk.y); kk.z = elu1(kk.z); kk.w = elu1(kk.w);
        vv.x = vv.x / vdiv; vv.y = vv.y / vdiv; vv.z = vv.z / vdiv; vv.w = vv.w / vdiv;      // values / v_length
      }
      ks[r][c] = kk.x; ks[r][c + 1] = kk.y; ks[r][c + 2] = kk.z; ks[r][c + 3] = kk.w;
      *reinterpret_cast<f32x4*>(&vs[r][c]) = vv;
    }
    __syncthreads();
    const int nr = min(64, s1 - t0);
    if (active)
      for (int r = 0; r < nr; ++r) {
        const float kd = ks[r][d];
        acc += kd * *reinterpret_cast<const f32x4*>(&vs[r][v4]);
        if (v4 == 0) ksum += kd;
      }
  }
  if (!active) return;
  float* o = part + (((long long)seq * nsplit + sp) * heads + h) * (HD + 1) * HD;
  *reinterpret_cast<f32x4*>(o + d * HD + v4) = acc;
  if (v4 == 0) o[HD * HD + d] = ksum;
}
// kv[seq][head][HD + 1][HD] = sum over splits in order (deterministic); per = heads * (HD + 1) * HD
__global__ __launch_bounds__(256) void k_lf_kv_reduce(const float* part, float* kv, int nsplit, long long per, int nseq) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= per * nseq) return;
  const int seq = (int)(idx / per);
  const long long r = idx - seq * per;
  float s = 0.f;
  for (int sp = 0; sp < nsplit; ++sp) s += part[((long long)seq * nsplit + sp) * per + r];
  kv[idx] = s;
}
// out[seq][l][h][v] = (sum_d Q[d] KV[d][v]) * Z * S, Z = 1 / (sum_d Q[d] Ksum[d] + eps), Q = elu(q) + 1.  One block = 8 tokens of one
// sequence; thread t < heads * HD -> (head = t / HD, v = t % HD); the KV of sequence seq ^ 1 when `cross`.
template <int HD>
__global__ __launch_bounds__(256) void k_lf_attn_apply(const float* q, int ldq, long long q_seq_stride, const float* kv, int cross, float* out, int ldo, long long o_seq_stride,
                                                      int L, float slen, int heads) {
  extern __shared__ float sm[];                // KV [heads][HD + 1][HD] then Q tile [8][D]
  const int D = heads * HD, KVN = heads * (HD + 1) * HD;
  const int seq = blockIdx.y, tid = threadIdx.x;
  float* kvs = sm; float* qs = sm + KVN;
  const float* kvg = kv + (long long)(cross ? (seq ^ 1) : seq) * KVN;
  for (int i = tid; i < KVN; i += 256) kvs[i] = kvg[i];
  const int l0 = blockIdx.x * 8;
  for (int i = tid; i < 8 * D; i += 256) {
    const int r = i / D, c = i - r * D;
    qs[i] = (l0 + r < L) ? elu1(q[seq * q_seq_stride + (long long)(l0 + r) * ldq + c]) : 0.f;
  }
  __syncthreads();
  if (tid >= D) return;
  const int h = tid / HD, v = tid % HD;
  const float* kvh = kvs + h * (HD + 1) * HD;
  for (int r = 0; r < 8 && l0 + r < L; ++r) {
    const float* qr = qs + r * D + h * HD;
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) { num = fmaf(qr[d], kvh[d * HD + v], num); den = fmaf(qr[d], kvh[HD * HD + d], den); }
    out[seq * o_seq_stride + (long long)(l0 + r) * ldo + tid] = num * (1.0f / (den + 1e-6f)) * slen;
  }
}

// The fine level's linear attention in ONE launch: sequences of <= 32 tokens, 8 heads x 16 (d = 128).  One block per sequence; thread t -> (head
// t >> 4, v = t & 15).  K^T V (values / S), sum K and the application to the block's own queries, all from LDS; the source is the sequence
// itself or its partner seq ^ 1 (cross).  Same summation orders as k_lf_kv_partial<16> (one chunk) + k_lf_attn_apply<16>.
// partner: the sequence attended to is seq ^ 1 (cross = 1, partner = 0: interleaved pairs), seq + partner (partner != 0: side-major buffers, the
// other side's window of the same match) or seq itself; seq0: first sequence of the launch; nlim / per_side: sequences whose window index
// (seq % per_side) is at or behind nlim[0] (the match count, on the device) are skipped
__global__ __launch_bounds__(128) void k_lf_fine_attn(const float* qkv /*[nseq][S][384]*/, int S, int cross, float* out /*[nseq][S][128]*/, int seq0, int partner, const int* nlim, int per_side) {
  __shared__ float qs[32][128], ks[32][129], vs[32][128];
  const int seq = seq0 + blockIdx.x, src = partner != 0 ? seq + partner : cross ? (seq ^ 1) : seq, tid = threadIdx.x;
  if (nlim != nullptr && seq % per_side >= nlim[0]) return;
  const float vdiv = (float)S;
  for (int i = tid; i < S * 32; i += 128) {
    const int r = i >> 5, c = (i & 31) * 4;
    f32x4 q = *reinterpret_cast<const f32x4*>(qkv + ((long long)seq * S + r) * 384 + c);
    f32x4 k = *reinterpret_cast<const f32x4*>(qkv + ((long long)src * S + r) * 384 + 128 + c);
    f32x4 v = *reinterpret_cast<const f32x4*>(qkv + ((long long)src * S + r) * 384 + 256 + c);
    qs[r][c] = elu1(q.x); qs[r][c + 1] = elu1(q.y); qs[r][c + 2] = elu1(q.z); qs[r][c + 3] = elu1(q.w);
    ks[r][c] = elu1(k.x); ks[r][c + 1] = elu1(k.y); ks[r][c + 2] = elu1(k.z); ks[r][c + 3] = elu1(k.w);
    vs[r][c] = v.x / vdiv; vs[r][c + 1] = v.y / vdiv; vs[r][c + 2] = v.z / vdiv; vs[r][c + 3] = v.w / vdiv;
  }
  __syncthreads();
  const int h = tid >> 4, v = tid & 15;
  float kv[16], ksum[16];
#pragma unroll
  for (int d = 0; d < 16; ++d) { kv[d] = 0.f; ksum[d] = 0.f; }
  for (int r = 0; r < S; ++r) {
    const float vv = vs[r][h * 16 + v];
#pragma unroll
    for (int d = 0; d < 16; ++d) { const float kd = ks[r][h * 16 + d]; kv[d] += kd * vv; ksum[d] += kd; }
  }
  for (int r = 0; r < S; ++r) {
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int d = 0; d < 16; ++d) { const float qd = qs[r][h * 16 + d]; num = fmaf(qd, kv[d], num); den = fmaf(qd, ksum[d], den); }
    out[((long long)seq * S + r) * 128 + tid] = num * (1.0f / (den + 1e-6f)) * vdiv;
  }
}

__device__ inline float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// LayerNorm over rows of C (128 or 256; eps 1e-5, affine), optional residual: out = (resid ? resid : 0) + LN(in).  One wave per row.
// mode 0: every row; 1 / 2: only rows of even / odd sequences (sequence = row / seq_rows) -- the two halves of a 'cross' layer.
// mlim / mlim_mul / mlim_seg: GemmArgs' row limit (rows r with (seg ? r % seg : r) >= mlim[0] * mul are skipped)
__global__ __launch_bounds__(256) void k_lf_layernorm(const float* in, const float* g, const float* b, const float* resid, float* out, long long rows, int C, int seq_rows, int mode,
                                                     const int* mlim = nullptr, int mlim_mul = 0, int mlim_seg = 0) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  if (mlim != nullptr && (mlim_seg > 0 ? row % mlim_seg : row) >= (long long)mlim[0] * mlim_mul) return;
  if (mode != 0 && (int)((row / seq_rows) & 1) != mode - 1) return;
  // a lane owns C / 64 = 2 or 4 CONSECUTIVE values: one 8- or 16-byte access each for the row, the affine pair and the residual
  float v[4], gg[4], bb[4], rr[4] = {0.f, 0.f, 0.f, 0.f};
  const long long base = row * C;
  if (C == 256) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(in + base + lane * 4);
    const f32x4 tg = *reinterpret_cast<const f32x4*>(g + lane * 4), tb = *reinterpret_cast<const f32x4*>(b + lane * 4);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    gg[0] = tg.x; gg[1] = tg.y; gg[2] = tg.z; gg[3] = tg.w; bb[0] = tb.x; bb[1] = tb.y; bb[2] = tb.z; bb[3] = tb.w;
    if (resid) { const f32x4 tr = *reinterpret_cast<const f32x4*>(resid + base + lane * 4); rr[0] = tr.x; rr[1] = tr.y; rr[2] = tr.z; rr[3] = tr.w; }
  } else {   // C == 128
    const float2 t = *reinterpret_cast<const float2*>(in + base + lane * 2);
    const float2 tg = *reinterpret_cast<const float2*>(g + lane * 2), tb = *reinterpret_cast<const float2*>(b + lane * 2);
    v[0] = t.x; v[1] = t.y; v[2] = 0.f; v[3] = 0.f; gg[0] = tg.x; gg[1] = tg.y; gg[2] = gg[3] = 0.f; bb[0] = tb.x; bb[1] = tb.y; bb[2] = bb[3] = 0.f;
    if (resid) { const float2 tr = *reinterpret_cast<const float2*>(resid + base + lane * 2); rr[0] = tr.x; rr[1] = tr.y; }
  }
  const int per = C / 64;
  float s = 0.f;
  for (int e = 0; e < per; ++e) s += v[e];
  const float mean = wsum(s) / (float)C;
  float sq = 0.f;
  for (int e = 0; e < per; ++e) { v[e] -= mean; sq += v[e] * v[e]; }
  const float rstd = 1.0f / sqrtf(wsum(sq) / (float)C + 1e-5f);
  float y[4];
  for (int e = 0; e < 4; ++e) { y[e] = v[e] * rstd * gg[e] + bb[e]; if (resid) y[e] = rr[e] + y[e]; }
  if (C == 256) *reinterpret_cast<f32x4*>(out + base + lane * 4) = (f32x4){y[0], y[1], y[2], y[3]};
  else *reinterpret_cast<float2*>(out + base + lane * 2) = make_float2(y[0], y[1]);
}
__global__ __launch_bounds__(256) void k_lf_scale(const float* x, float* y, float div, long long n4) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
  v.x /= div; v.y /= div; v.z /= div; v.w /= div;
  reinterpret_cast<f32x4*>(y)[i] = v;
}

// split-fp16 mode: any value that left fp16's range inside a GEMM shows up as inf / NaN downstream -- raise the guard word
__global__ __launch_bounds__(256) void k_lf_check_finite(const float* x, long long n4, unsigned int* flag) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
  if (!(fabsf(v.x) < 3.0e38f && fabsf(v.y) < 3.0e38f && fabsf(v.z) < 3.0e38f && fabsf(v.w) < 3.0e38f)) atomicOr(flag, 1u);
}

// ------------------------------------------------------------------------------------------------ coarse matching
// sim[i][j] raw = <f0_i / 16, f1_j / 16>; s = raw / temperature.  conf = exp(s - cmax_j) / csum_j * exp(s - rmax_i) / rsum_i
// (F.softmax(sim, 1) * F.softmax(sim, 2): dim 1 = over i for a fixed column j, dim 2 = over j for a fixed row i).
__device__ __forceinline__ float lf_conf(float s, float rmax, float rsum, float cmax, float csum) {
  return (expf(s - cmax) / csum) * (expf(s - rmax) / rsum);
}
constexpr int kLfU = 5;       // loads in flight per thread in the passes over the similarity matrix
// one block per row i: max and sum of exponentials over j < L
__global__ __launch_bounds__(256) void k_lf_row_stats(const float* sim, int ld, int L, float temp, float* rmax, float* rsum) {
  __shared__ float red[8];
  const int i = blockIdx.x, tid = threadIdx.x;
  const float* row = sim + (long long)i * ld;
  // (late round 5, here and in the four kernels below: a thread's loads are requested kLfU at a time before the first is used -- same elements in the
  // same order per thread, so the same bits; with one load in flight per loop iteration these passes over the 92 MB matrix ran at 1.3-2.5 TB/s: k_lf_conf_colmax 74 -> 59 us, k_lf_col_stats 63 -> below 50,
  // the row passes -3 us each; the same treatment of k_lf_fine_attn's staging loop bought nothing -- that kernel is bound by its LDS reads)
  float m = -INFINITY;
  for (int j0 = tid; j0 < L; j0 += 256 * kLfU) {
    float x[kLfU];
#pragma unroll
    for (int u = 0; u < kLfU; ++u) { const int j = j0 + 256 * u; x[u] = j < L ? row[j] : 0.f; }
#pragma unroll
    for (int u = 0; u < kLfU; ++u) if (j0 + 256 * u < L) m = fmaxf(m, x[u] / temp);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float s = 0.f;
  for (int j0 = tid; j0 < L; j0 += 256 * kLfU) {
    float x[kLfU];
#pragma unroll
    for (int u = 0; u < kLfU; ++u) { const int j = j0 + 256 * u; x[u] = j < L ? row[j] : 0.f; }
#pragma unroll
    for (int u = 0; u < kLfU; ++u) if (j0 + 256 * u < L) s += expf(x[u] / temp - m);
  }
  s = wsum(s);
  if ((tid & 63) == 0) red[4 + (tid >> 6)] = s;
  __syncthreads();
  if (tid == 0) { rmax[i] = m; rsum[i] = (red[4] + red[5]) + (red[6] + red[7]); }
}
// thread per column j (coalesced across j), loop over rows in nsplit chunks: partial (max, sum) then merged by k_lf_col_merge
__global__ __launch_bounds__(256) void k_lf_col_stats(const float* sim, int ld, int L, float temp, float* pmax, float* psum, int rows_per) {
  const int j = blockIdx.x * 256 + threadIdx.x, sp = blockIdx.y;
  if (j >= L) return;
  const int i0 = sp * rows_per, i1 = min(L, i0 + rows_per);
  float m = -INFINITY, s = 0.f;
  for (int ib = i0; ib < i1; ib += 2 * kLfU) {
    float xv[2 * kLfU];
#pragma unroll
    for (int u = 0; u < 2 * kLfU; ++u) xv[u] = ib + u < i1 ? sim[(long long)(ib + u) * ld + j] : 0.f;
#pragma unroll
    for (int u = 0; u < 2 * kLfU; ++u) {
      if (ib + u >= i1) continue;
      const float x = xv[u] / temp;
      if (x > m) { s = s * expf(m - x) + 1.f; m = x; } else s += expf(x - m);
    }
  }
  pmax[(long long)sp * L + j] = m; psum[(long long)sp * L + j] = s;
}
__global__ __launch_bounds__(256) void k_lf_col_merge(const float* pmax, const float* psum, int nsplit, int L, float* cmax, float* csum) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= L) return;
  float m = -INFINITY;
  for (int sp = 0; sp < nsplit; ++sp) m = fmaxf(m, pmax[(long long)sp * L + j]);
  float s = 0.f;
  for (int sp = 0; sp < nsplit; ++sp) { const float pm = pmax[(long long)sp * L + j]; if (pm > -INFINITY) s += psum[(long long)sp * L + j] * expf(pm - m); }
  cmax[j] = m; csum[j] = s;
}
// max of conf over j per row (block per row) and over i per column (thread per column, split + merge)
__global__ __launch_bounds__(256) void k_lf_conf_rowmax(const float* sim, int ld, int L, float temp, const float* rmax, const float* rsum, const float* cmax, const float* csum, float* crow) {
  __shared__ float red[4];
  const int i = blockIdx.x, tid = threadIdx.x;
  const float* row = sim + (long long)i * ld;
  const float rm = rmax[i], rs = rsum[i];
  float m = 0.f;
  for (int j0 = tid; j0 < L; j0 += 256 * kLfU) {
    float x[kLfU], cm[kLfU], cs[kLfU];
#pragma unroll
    for (int u = 0; u < kLfU; ++u) { const int j = j0 + 256 * u; const bool in = j < L; x[u] = in ? row[j] : 0.f; cm[u] = in ? cmax[j] : 0.f; cs[u] = in ? csum[j] : 1.f; }
#pragma unroll
    for (int u = 0; u < kLfU; ++u) if (j0 + 256 * u < L) m = fmaxf(m, lf_conf(x[u] / temp, rm, rs, cm[u], cs[u]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  if (tid == 0) crow[i] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__global__ __launch_bounds__(256) void k_lf_conf_colmax(const float* sim, int ld, int L, float temp, const float* rmax, const float* rsum, const float* cmax, const float* csum, float* part, int rows_per) {
  const int j = blockIdx.x * 256 + threadIdx.x, sp = blockIdx.y;
  if (j >= L) return;
  const int i0 = sp * rows_per, i1 = min(L, i0 + rows_per);
  const float cm = cmax[j], cs = csum[j];
  float m = 0.f;
  for (int ib = i0; ib < i1; ib += 2 * kLfU) {
    float xv[2 * kLfU];
#pragma unroll
    for (int u = 0; u < 2 * kLfU; ++u) xv[u] = ib + u < i1 ? sim[(long long)(ib + u) * ld + j] : 0.f;
#pragma unroll
    for (int u = 0; u < 2 * kLfU; ++u) if (ib + u < i1) m = fmaxf(m, lf_conf(xv[u] / temp, rmax[ib + u], rsum[ib + u], cm, cs));
  }
  part[(long long)sp * L + j] = m;
}
__global__ __launch_bounds__(256) void k_lf_max_merge(const float* part, int nsplit, int L, float* out) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= L) return;
  float m = 0.f;
  for (int sp = 0; sp < nsplit; ++sp) m = fmaxf(m, part[(long long)sp * L + j]);
  out[j] = m;
}
// get_coarse_match for row i: the first j with conf > thr, both cells inside the border, conf == row max and conf == column max
__global__ __launch_bounds__(256) void k_lf_mutual(const float* sim, int ld, int L, int hc, int wc, float temp, float thr, int border, const float* rmax, const float* rsum,
                                                  const float* cmax, const float* csum, const float* crow, const float* ccol, int* jsel, float* csel) {
  __shared__ int best;
  const int i = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) best = 0x7fffffff;
  __syncthreads();
  const int yi = i / wc, xi = i - yi * wc;
  const bool in_i = yi >= border && yi < hc - border && xi >= border && xi < wc - border;
  const float* row = sim + (long long)i * ld;
  const float rm = rmax[i], rs = rsum[i], cr = crow[i];
  if (in_i)
    for (int j0 = tid; j0 < L; j0 += 256 * kLfU) {
      float x[kLfU], cm[kLfU], cs[kLfU], cc[kLfU];
#pragma unroll
      for (int u = 0; u < kLfU; ++u) {
        const int j = j0 + 256 * u; const bool in = j < L;
        x[u] = in ? row[j] : 0.f; cm[u] = in ? cmax[j] : 0.f; cs[u] = in ? csum[j] : 1.f; cc[u] = in ? ccol[j] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < kLfU; ++u) {
        const int j = j0 + 256 * u;
        if (j >= L) continue;
        const float c = lf_conf(x[u] / temp, rm, rs, cm[u], cs[u]);
        const int yj = j / wc, xj = j - yj * wc;
        if (c > thr && c == cr && c == cc[u] && yj >= border && yj < hc - border && xj >= border && xj < wc - border) atomicMin(&best, j);
      }
    }
  __syncthreads();
  if (tid == 0) {
    const int j = best;
    jsel[i] = j == 0x7fffffff ? -1 : j;
    csel[i] = j == 0x7fffffff ? 0.f : lf_conf(row[j] / temp, rm, rs, cmax[j], csum[j]);
  }
}
// ordered compaction (ascending i, like torch.where): one block
__global__ __launch_bounds__(1024) void k_lf_compact(const int* jsel, const float* csel, int L, int wc, int scale, int max_out, int* i_ids, int* j_ids, float* conf, float* k0, float* k1, int* n_out) {
  __shared__ int wcount[16], base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) base = 0;
  __syncthreads();
  for (int i0 = 0; i0 < L; i0 += 1024) {
    const int i = i0 + tid;
    const bool ok = i < L && jsel[i] >= 0;
    const unsigned long long bal = __ballot(ok);
    if (lane == 0) wcount[wave] = __popcll(bal);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wave; ++w) off += wcount[w];
    const int pos = off + __popcll(bal & ((1ull << lane) - 1ull));
    if (ok && pos < max_out) {
      const int j = jsel[i];
      i_ids[pos] = i; j_ids[pos] = j; conf[pos] = csel[i];
      k0[2 * pos] = (float)((i % wc) * scale); k0[2 * pos + 1] = (float)((i / wc) * scale);
      k1[2 * pos] = (float)((j % wc) * scale); k1[2 * pos + 1] = (float)((j / wc) * scale);
    }
    __syncthreads();
    if (tid == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += wcount[w]; base += t; }
    __syncthreads();
  }
  if (tid == 0) *n_out = min(base, max_out);
}

// ------------------------------------------------------------------------------------------------ fine level
// FinePreprocess: rows [m][ww][256] = [ unfold_5x5(feat_f, stride 4, pad 2)[cell(m)][ww][0..127] | down_proj(coarse feature of the match)[0..127] ]
// for side 0 (cells i_ids) in rows [0, M) and side 1 (cells j_ids) in rows [M, 2 M); merge_feat is then one GEMM over K = 256.
__global__ __launch_bounds__(256) void k_lf_fine_gather(const float* ff /*[2][Hf][Wf][128]*/, int Hf, int Wf, int wc, const int* i_ids, const int* j_ids, const int* n_match, int Mp,
                                                       const float* cwin /*[2 Mp][128]*/, float* rows /*[2 Mp * 25][256]*/) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;          // float4 index over [2 Mp][25][64]
  if (idx >= 2LL * Mp * kLfWW * 64) return;
  const int c4 = (int)(idx & 63);
  long long t = idx >> 6;
  const int ww = (int)(t % kLfWW); t /= kLfWW;
  const int m2 = (int)t, side = m2 >= Mp ? 1 : 0, m = m2 - side * Mp;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  const int M = *n_match;
  if (m * kLfWW + ww >= (M * kLfWW + 127) / 128 * 128) return;         // behind the last 128-row tile a GEMM of the fine level touches (GemmArgs::mlim)
  if (m < M) {
    if (c4 < 32) {
      const int cell = side ? j_ids[m] : i_ids[m];
      const int cy = cell / wc, cx = cell - cy * wc;
      const int y = cy * 4 - 2 + ww / 5, x = cx * 4 - 2 + ww % 5;          // F.unfold(kernel 5, stride 4, padding 2): zero outside
      if (y >= 0 && y < Hf && x >= 0 && x < Wf) v = *reinterpret_cast<const f32x4*>(ff + (((long long)side * Hf + y) * Wf + x) * 128 + c4 * 4);
    } else {
      v = *reinterpret_cast<const f32x4*>(cwin + (long long)m2 * 128 + (c4 - 32) * 4);
    }
  }
  *reinterpret_cast<f32x4*>(rows + ((long long)m2 * kLfWW + ww) * 256 + c4 * 4) = v;
}
// rows [2 Mp][256] = [coarse feature of cell i (side 0) / j (side 1)] for down_proj
__global__ __launch_bounds__(256) void k_lf_coarse_gather(const float* f0, const float* f1, const int* i_ids, const int* j_ids, const int* n_match, int Mp, float* rows) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= 2LL * Mp * 64) return;
  const int c4 = (int)(idx & 63), m2 = (int)(idx >> 6), side = m2 >= Mp ? 1 : 0, m = m2 - side * Mp;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (m < *n_match) v = *reinterpret_cast<const f32x4*>((side ? f1 + (long long)j_ids[m] * 256 : f0 + (long long)i_ids[m] * 256) + c4 * 4);
  *reinterpret_cast<f32x4*>(rows + (long long)m2 * 256 + c4 * 4) = v;
}
// FineMatching: one wave per match: sim_r = <f0[centre], f1[r]>, softmax(sim / sqrt(128)), expectation over the normalised 5x5 grid,
// k1_f = k1_c + expectation * 2 * 2
__global__ __launch_bounds__(256) void k_lf_fine_match(const float* f0 /*[Mp*25][128]*/, const float* f1, const int* n_match, const float* k1c, float* k1f) {
  const int lane = threadIdx.x & 63, m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= *n_match) return;
  const float* c = f0 + ((long long)m * kLfWW + 12) * 128;
  const float c0 = c[lane], c1 = c[lane + 64];
  float sim[kLfWW];
  float mx = -INFINITY;
  for (int r = 0; r < kLfWW; ++r) {
    const float* p = f1 + ((long long)m * kLfWW + r) * 128;
    sim[r] = wsum(c0 * p[lane] + c1 * p[lane + 64]) / sqrtf(128.f);
    mx = fmaxf(mx, sim[r]);
  }
  float den = 0.f, ex = 0.f, ey = 0.f;
  for (int r = 0; r < kLfWW; ++r) {
    const float e = expf(sim[r] - mx);
    den += e; ex += e * (-1.f + 0.5f * (float)(r % 5)); ey += e * (-1.f + 0.5f * (float)(r / 5));
  }
  if (lane == 0) { k1f[2 * m] = k1c[2 * m] + (ex / den) * 2.f * 2.f; k1f[2 * m + 1] = k1c[2 * m + 1] + (ey / den) * 2.f * 2.f; }
}

}  // namespace
}  // namespace gn

// ================================================================================================ context + C ABI
using namespace gn;

namespace {
struct LfLinear { float* w = nullptr; float* b = nullptr; int out = 0, in = 0; };
struct LfLayer { LfLinear qkv /* [q | k | v] stacked: 3d x d */, merge, mlp0, mlp2; float *n1g = nullptr, *n1b = nullptr, *n2g = nullptr, *n2b = nullptr; };
struct LfConv {
  std::vector<float> hw; int cout = 0, cin = 0, ks = 0;                         // host copy [cout][cin][ks*ks] until finalised
  std::vector<float> bn[4]; bool has_bn = false;                                // weight, bias, running_mean, running_var
  float* wf = nullptr; float* scale = nullptr; float* shift = nullptr; int cout_p = 0, cin_p = 0;
  uint16_t* wfh = nullptr; float* scale_h = nullptr;        // split-fp16 arithmetic: fp16 pairs in fragment order, affine scale x 1 / weight scale
};
int pad32(int c) { return (c + 31) / 32 * 32; }
}  // namespace

struct gn_loftr {
  int device = 0, H = 0, W = 0, hc = 0, wc = 0, L = 0, Lp = 0, fine = 1, max_matches = 0, Mp = 0;
  std::string err;
  std::map<std::string, LfConv> conv;          // by module name, e.g. "backbone.layer1.0.conv1"
  LfLayer coarse[8], finel[2];
  LfLinear down_proj, merge_feat;
  std::map<std::string, bool> loaded;
  bool finalised = false;
  std::vector<void*> allocs;
  // activations (NHWC) and workspaces
  float *img = nullptr, *x0 = nullptr, *x1 = nullptr, *x2 = nullptr, *x3 = nullptr, *t1 = nullptr, *t2 = nullptr, *t3 = nullptr;
  float *x3_out = nullptr, *x2_out = nullptr, *x1_out = nullptr, *fpn_a = nullptr, *fpn_b = nullptr;
  float *pe = nullptr, *tok = nullptr, *qkv = nullptr, *att = nullptr, *msg = nullptr, *hid = nullptr, *kvpart = nullptr, *kv = nullptr, *fs = nullptr, *sim = nullptr;
  float *rmax = nullptr, *rsum = nullptr, *cmax = nullptr, *csum = nullptr, *crow = nullptr, *ccol = nullptr, *cpart_a = nullptr, *cpart_b = nullptr, *csel = nullptr;
  int *jsel = nullptr, *i_ids = nullptr, *j_ids = nullptr, *n_dev = nullptr;
  float *k0c = nullptr, *k1c = nullptr, *mconf = nullptr;
  float *frows = nullptr, *fc = nullptr, *fwin = nullptr, *ftok = nullptr, *fqkv = nullptr, *fatt = nullptr, *fmsg = nullptr, *fhid = nullptr, *fkvpart = nullptr, *fkv = nullptr;
  int* n_host = nullptr;
  int use_graph = 1; bool graph_failed[2] = {false, false}; hipGraphExec_t graph_exec[2] = {nullptr, nullptr}; hipStream_t cap_stream = nullptr;   // gn_loftr_set_graph; one graph per arithmetic
  int arith = 0;                        // gn_loftr_set_arithmetic: 0 exact f32, 1 split fp16 (f32-accurate)
  unsigned int* ovf = nullptr; long long ovf_trips = 0;   // split mode: fp16-range guard word (device), calls that fell back to the exact kernels
};

namespace {
#define LF_HIP(call)                                                                               \
  do { hipError_t e_ = (call); if (e_ != hipSuccess) { char b_[512]; snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
       if (ctx) ctx->err = b_; g_lf_err = b_; return GN_ERR_HIP; } } while (0)
int lf_fail(gn_loftr* ctx, int code, const std::string& m) { if (ctx) ctx->err = m; g_lf_err = m; return code; }
template <typename T> int lf_alloc(gn_loftr* ctx, T** p, size_t n) {
  void* q = nullptr;
  LF_HIP(hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T)));
  ctx->allocs.push_back(q);              // owned by the context from here on (gn_loftr_destroy frees it even if the clear below fails)
  LF_HIP(hipMemset(q, 0, std::max<size_t>(n, 1) * sizeof(T)));
  *p = reinterpret_cast<T*>(q);
  return GN_OK;
}
int lf_upload(gn_loftr* ctx, float** dst, const float* src, size_t n) {
  if (!*dst) { int rc = lf_alloc(ctx, dst, n); if (rc != GN_OK) return rc; }
  LF_HIP(hipMemcpy(*dst, src, n * sizeof(float), hipMemcpyHostToDevice));
  return GN_OK;
}

const char* kConvNames[] = {
  "backbone.conv1",
  "backbone.layer1.0.conv1", "backbone.layer1.0.conv2", "backbone.layer1.1.conv1", "backbone.layer1.1.conv2",
  "backbone.layer2.0.conv1", "backbone.layer2.0.conv2", "backbone.layer2.0.downsample.0", "backbone.layer2.1.conv1", "backbone.layer2.1.conv2",
  "backbone.layer3.0.conv1", "backbone.layer3.0.conv2", "backbone.layer3.0.downsample.0", "backbone.layer3.1.conv1", "backbone.layer3.1.conv2",
  "backbone.layer3_outconv", "backbone.layer2_outconv", "backbone.layer2_outconv2.0", "backbone.layer2_outconv2.3",
  "backbone.layer1_outconv", "backbone.layer1_outconv2.0", "backbone.layer1_outconv2.3"};
// the BatchNorm that follows each convolution ("" = none)
const char* kConvBn[] = {
  "backbone.bn1",
  "backbone.layer1.0.bn1", "backbone.layer1.0.bn2", "backbone.layer1.1.bn1", "backbone.layer1.1.bn2",
  "backbone.layer2.0.bn1", "backbone.layer2.0.bn2", "backbone.layer2.0.downsample.1", "backbone.layer2.1.bn1", "backbone.layer2.1.bn2",
  "backbone.layer3.0.bn1", "backbone.layer3.0.bn2", "backbone.layer3.0.downsample.1", "backbone.layer3.1.bn1", "backbone.layer3.1.bn2",
  "", "", "backbone.layer2_outconv2.1", "", "", "backbone.layer1_outconv2.1", ""};
constexpr int kNumConv = 22;

// BatchNorm (eval) as scale / shift, padded channels, fragment-order weights
int lf_finalise(gn_loftr* ctx) {
  for (int ci = 0; ci < kNumConv; ++ci) {
    if (!ctx->fine && ci >= 16) continue;            // the FPN head below 1/8 resolution only feeds the fine level
    auto it = ctx->conv.find(kConvNames[ci]);
    if (it == ctx->conv.end() || it->second.hw.empty()) return lf_fail(ctx, GN_ERR_WEIGHTS, std::string("missing tensor ") + kConvNames[ci] + ".weight");
    LfConv& c = it->second;
    const int taps = c.ks * c.ks;
    c.cout_p = pad32(c.cout); c.cin_p = ci == 0 ? 1 : pad32(c.cin);
    std::vector<float> scale(c.cout_p, 1.f), shift(c.cout_p, 0.f);
    if (kConvBn[ci][0]) {
      auto bt = ctx->conv.find(kConvBn[ci]);
      if (bt == ctx->conv.end() || !bt->second.has_bn) return lf_fail(ctx, GN_ERR_WEIGHTS, std::string("missing BatchNorm ") + kConvBn[ci]);
      for (int k = 0; k < 4; ++k) if ((int)bt->second.bn[k].size() != c.cout) return lf_fail(ctx, GN_ERR_WEIGHTS, std::string("incomplete BatchNorm ") + kConvBn[ci]);
      for (int o = 0; o < c.cout; ++o) {   // y = (x - mean) / sqrt(var + eps) * gamma + beta
        const float inv = 1.0f / std::sqrt(bt->second.bn[3][o] + 1e-5f);
        scale[o] = bt->second.bn[0][o] * inv;
        shift[o] = bt->second.bn[1][o] - bt->second.bn[2][o] * scale[o];
      }
    }
    int rc = lf_upload(ctx, &c.scale, scale.data(), scale.size()); if (rc != GN_OK) return rc;
    rc = lf_upload(ctx, &c.shift, shift.data(), shift.size()); if (rc != GN_OK) return rc;
    if (ci == 0) {   // stem: [128][49] as it is, and behind it the same weights as [49][128] (what k_lf_conv1<false> keeps in LDS: a coalesced copy instead of a strided gather per workgroup)
      if (c.hw.size() != (size_t)128 * 49) return lf_fail(ctx, GN_ERR_SHAPE, "backbone.conv1.weight: expected [128][1][7][7]");
      std::vector<float> both(c.hw);
      both.resize(2 * c.hw.size());
      for (int o = 0; o < 128; ++o)
        for (int t = 0; t < 49; ++t) both[128 * 49 + t * 128 + o] = c.hw[(size_t)o * 49 + t];
      rc = lf_upload(ctx, &c.wf, both.data(), both.size()); if (rc != GN_OK) return rc;
      continue;
    }
    std::vector<float> wp((size_t)c.cout * c.cin_p * taps, 0.f);     // input channels padded with zeros
    for (int o = 0; o < c.cout; ++o)
      for (int i = 0; i < c.cin; ++i)
        memcpy(&wp[((size_t)o * c.cin_p + i) * taps], &c.hw[((size_t)o * c.cin + i) * taps], taps * sizeof(float));
    std::vector<float> frag((size_t)c.cout_p * taps * c.cin_p);
    sp_weight_fragments(wp.data(), c.cout, c.cin_p, taps, c.cout_p, frag.data());
    rc = lf_upload(ctx, &c.wf, frag.data(), frag.size()); if (rc != GN_OK) return rc;
    {   // split-fp16 arithmetic: power-of-two scale that puts max |w| in [2^12, 2^13) (both fp16 terms stay normal), inverse folded into the affine scale
      float mx = 0.f;
      for (float v : wp) mx = std::max(mx, std::fabs(v));
      int e = 0;
      if (mx > 0.f && std::isfinite(mx)) { std::frexp(mx, &e); e = 13 - e; }
      e = std::max(-60, std::min(60, e));
      const float wscale = std::ldexp(1.0f, e), inv = std::ldexp(1.0f, -e);
      std::vector<uint16_t> fh((size_t)2 * c.cout_p * taps * c.cin_p);
      sp_weight_fragments_hm16(wp.data(), c.cout, c.cin_p, taps, c.cout_p, wscale, fh.data());
      if (!c.wfh) { rc = lf_alloc(ctx, &c.wfh, fh.size()); if (rc != GN_OK) return rc; }
      LF_HIP(hipMemcpy(c.wfh, fh.data(), fh.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
      std::vector<float> sh2(scale);
      for (float& v : sh2) v *= inv;
      rc = lf_upload(ctx, &c.scale_h, sh2.data(), sh2.size()); if (rc != GN_OK) return rc;
    }
  }
  ctx->finalised = true;
  return GN_OK;
}

void lf_conv(gn_loftr* ctx, const char* name, const float* in, int N, int Hin, int Win, float* out, int stride, const float* resid, int act, hipStream_t s) {
  const LfConv& c = ctx->conv[name];
  LfConvArgs a;
  a.in = in; a.Hin = Hin; a.Win = Win; a.Cin = c.cin_p; a.wf = c.wf; a.scale = c.scale; a.shift = c.shift; a.resid = resid;
  a.out = out; a.Hout = Hin / stride; a.Wout = Win / stride; a.Cout = c.cout_p; a.act = act;
  a.wfh = c.wfh; a.scale_h = c.scale_h; a.ovf = ctx->ovf;
  a.cin_real = (gn::g_lf_conv_knob & 8) ? c.cin_p : c.cin;        // developer knob 42, bit 3: multiply the zero padding like rounds 3-4 did
  const bool hm = ctx->arith == 1 && c.wfh != nullptr;
  const int og = (c.cout_p + 63) / 64;
  // rows per wave (RPW): the workgroup covers 4 RPW output rows x 32 columns x 64 channels and one workgroup fits a CU, so a launch takes
  // ceil(workgroups / 256) rounds of RPW units each -- pick the RPW with the fewest units (layer1 at 240x320: 600 workgroups = 3 rounds of 4
  // against 1200 = 5 rounds of 2; the 1/8-resolution layers fill 96 CUs with RPW = 4 and 192 with 2)
  // (+ 0.75: the halo rows, the weight stream and the prologue a workgroup pays whatever its height -- without it RPW = 1 wins ties it loses on the GPU)
  const bool fast = !(gn::g_lf_conv_knob & 16) && c.cout_p % 64 == 0 && a.cin_real % 32 == 0 && a.cin_real == c.cin_p;     // k_lf_conv<.., FAST> (knob 42 bit 4: off)
  const bool pf = !(gn::g_lf_conv_knob & 1);                                   // developer knob 42, bit 0: the staging form of rounds 3-4
  const double ovh = (gn::g_lf_conv_knob >> 8) ? (gn::g_lf_conv_knob >> 8) * 0.01 : (pf ? 0.25 : 0.75);    // bits 8..: the per-workgroup overhead of the cost model, in 1/100 units
  auto units = [&](int rpw) { const long long wg = (long long)((a.Wout + 31) / 32) * ((a.Hout + 4 * rpw - 1) / (4 * rpw)) * N * og; return (double)((wg + 255) / 256) * (rpw + ovh); };
  const dim3 blk(256);
  auto grid = [&](int rpw) { return dim3((a.Wout + 31) / 32, (a.Hout + 4 * rpw - 1) / (4 * rpw), N * og); };
  if (stride == 1) {
    int rpw = 4;
    if (units(2) < units(rpw)) rpw = 2;
    if (units(1) < units(rpw)) rpw = 1;
#define LF_LAUNCH(KS_, S_, R_, C_) do { if (hm && pf) hipLaunchKernelGGL((k_lf_conv_h<KS_, S_, R_, C_, true>), grid(R_), blk, 0, s, a); \
                                        else if (hm) hipLaunchKernelGGL((k_lf_conv_h<KS_, S_, R_, C_, false>), grid(R_), blk, 0, s, a); \
                                        else if (pf && fast) hipLaunchKernelGGL((k_lf_conv<KS_, S_, R_, C_, true, true>), grid(R_), blk, 0, s, a); \
                                        else if (pf) hipLaunchKernelGGL((k_lf_conv<KS_, S_, R_, C_, true>), grid(R_), blk, 0, s, a); \
                                        else hipLaunchKernelGGL((k_lf_conv<KS_, S_, R_, C_, false>), grid(R_), blk, 0, s, a); } while (0)
    if (c.ks == 3) {
      if (rpw == 4) LF_LAUNCH(3, 1, 4, 32); else if (rpw == 2) LF_LAUNCH(3, 1, 2, 32); else LF_LAUNCH(3, 1, 1, 32);
    } else {
      if (rpw == 4) LF_LAUNCH(1, 1, 4, 32); else if (rpw == 2) LF_LAUNCH(1, 1, 2, 32); else LF_LAUNCH(1, 1, 1, 32);
    }
  } else {
    const int rpw = units(1) < units(2) ? 1 : 2;
    if (c.ks == 3) { if (rpw == 2) LF_LAUNCH(3, 2, 2, 16); else LF_LAUNCH(3, 2, 1, 16); }
    else { if (rpw == 2) LF_LAUNCH(1, 2, 2, 16); else LF_LAUNCH(1, 2, 1, 16); }
#undef LF_LAUNCH
  }
}

thread_local int g_lf_gemm_variant = 3;   // set by lf_forward from the context it runs (per host thread)
struct LfLimit { const int* n = nullptr; int mul = 0, seg = 0; };     // GemmArgs::mlim
void lf_gemm(const float* A, int lda, const float* A2, int lda2, int K1, const float* Wt, int ldw, const float* bias, float* Y, int ldy, int M, int N, int K, hipStream_t s, bool relu = false,
             LfLimit lim = LfLimit()) {
  GemmArgs g;
  memset(&g, 0, sizeof g);
  g.mlim = lim.n; g.mlim_mul = lim.mul; g.mlim_seg = lim.seg;
  g.A = A; g.lda = lda; g.A2 = A2; g.lda2 = lda2; g.K1 = A2 ? K1 : K; g.W = Wt; g.ldw = ldw; g.bias = bias; g.Y = Y; g.ldy = ldy; g.M = M; g.N = N; g.K = K; g.acc_scale = 1.f;
  const int saved = gn::g_gemm_variant;
  gn::g_gemm_variant = g_lf_gemm_variant;                   // 3 = the exact-f32 MFMA GEMM, whatever other contexts selected; 6 = every f32 operand split into two fp16 terms on the fly (gn_loftr_set_arithmetic)
  launch_gemm_f32(relu ? EPI_RELU : bias ? EPI_BIAS : EPI_PLAIN, g, 1, s);      // (relu: the MLP's first layer, max(x, 0) applied to the value the plain epilogue would store)
  gn::g_gemm_variant = saved;
}

// One LoFTREncoderLayer over the sequences of buffer x ([rows_pad][d], sequences `seq_rows` rows apart, Lseq valid tokens each):
//   x <- x + norm2(mlp([x, norm1(merge(attn(q(x), k(src), v(src))))])).
// cross = 0: every sequence attends to itself (mode 0 updates all).  cross = 1: sequence s attends to sequence s ^ 1; a 'cross' LAYER is
// two calls, mode 1 (even sequences = side 0 updated from side 1) then mode 2 (odd sequences from the UPDATED even ones) -- the
// sequential order of LocalFeatureTransformer.forward.  Projections run over the whole buffer (keeps M a multiple of 128).
// lim (fine level only): the windows behind the match count are skipped (LfLimit; the buffers then hold the sides one behind the other:
// side_seqs windows each, the cross partner of sequence q is q +- side_seqs)
void lf_encoder(const LfLayer& ly, float* x, int seq_rows, int Lseq, int rows_pad, int d, int cross, int mode,
                float* qkv, float* att, float* msg, float* hid, float* kvpart, float* kv, hipStream_t s, LfLimit lim = LfLimit(), int side_seqs = 0) {
  const int heads = kLfHeads, hd = d / heads, nall = rows_pad / seq_rows;
  if (side_seqs > 0 && cross) {
    // the fine level's cross halves on side-major buffers (round 5; the same split as the coarse level's below): only side a = mode - 1 is updated,
    // only side b is attended to -- q for a's rows, k | v for b's, merge / MLP / norms over a's rows: half of what the general path computes
    const int a = mode - 1, b = 1 - a, half = side_seqs * seq_rows;
    LfLimit hl = lim; hl.seg = 0;                                   // one side: rows [0, n * 25) are the valid ones
    float* xa = x + (size_t)a * half * d; const float* xb = x + (size_t)b * half * d;
    float* qa = qkv + (size_t)a * half * 3 * d; float* kvb = qkv + (size_t)b * half * 3 * d + d;
    float* atta = att + (size_t)a * half * d; float* msga = msg + (size_t)a * half * d; float* hida = hid + (size_t)a * half * 2 * d;
    lf_gemm(xa, d, nullptr, 0, 0, ly.qkv.w, d, nullptr, qa, 3 * d, half, d, d, s, false, hl);
    lf_gemm(xb, d, nullptr, 0, 0, ly.qkv.w + (size_t)d * d, d, nullptr, kvb, 3 * d, half, 2 * d, d, s, false, hl);
    hipLaunchKernelGGL(k_lf_fine_attn, dim3(side_seqs), dim3(128), 0, s, qkv, Lseq, 1, att, a * side_seqs, (b - a) * side_seqs, lim.n, side_seqs);
    lf_gemm(atta, d, nullptr, 0, 0, ly.merge.w, d, nullptr, msga, d, half, d, d, s, false, hl);
    hipLaunchKernelGGL(k_lf_layernorm, dim3((unsigned)((half + 3) / 4)), dim3(256), 0, s, msga, ly.n1g, ly.n1b, (const float*)nullptr, msga, (long long)half, d, seq_rows, 0, hl.n, hl.mul, hl.seg);
    lf_gemm(xa, d, msga, d, d, ly.mlp0.w, 2 * d, nullptr, hida, 2 * d, half, 2 * d, 2 * d, s, true, hl);
    lf_gemm(hida, 2 * d, nullptr, 0, 0, ly.mlp2.w, 2 * d, nullptr, atta, d, half, d, 2 * d, s, false, hl);
    hipLaunchKernelGGL(k_lf_layernorm, dim3((unsigned)((half + 3) / 4)), dim3(256), 0, s, atta, ly.n2g, ly.n2b, xa, xa, (long long)half, d, seq_rows, 0, hl.n, hl.mul, hl.seg);
    return;
  }
  if (cross && nall == 2 && hd == 32 && seq_rows % 128 == 0) {
    // the coarse level's cross halves: only sequence a = mode - 1 is updated, only sequence b = 1 - a is attended to -- project q for a's rows,
    // k | v for b's rows, and run merge / MLP / norms over a's rows alone (half the work of the general path below)
    const int a = mode - 1, b = 1 - a;
    float* xa = x + (size_t)a * seq_rows * d; const float* xb = x + (size_t)b * seq_rows * d;
    float* qa = qkv + (size_t)a * seq_rows * 3 * d; float* kvb = qkv + (size_t)b * seq_rows * 3 * d + d;
    float* atta = att + (size_t)a * seq_rows * d; float* msga = msg + (size_t)a * seq_rows * d; float* hida = hid + (size_t)a * seq_rows * 2 * d;
    lf_gemm(xa, d, nullptr, 0, 0, ly.qkv.w, d, nullptr, qa, 3 * d, seq_rows, d, d, s);
    lf_gemm(xb, d, nullptr, 0, 0, ly.qkv.w + (size_t)d * d, d, nullptr, kvb, 3 * d, seq_rows, 2 * d, d, s);
    const int chunk = 192, nsplit = (Lseq + chunk - 1) / chunk;
    const long long per = (long long)heads * 33 * 32;
    hipLaunchKernelGGL(k_lf_kv_partial<32>, dim3(heads, nsplit, 1), dim3(256), 0, s, kvb, kvb + d, 3 * d, 0LL, Lseq, chunk, (float)Lseq, kvpart, nsplit, heads);
    hipLaunchKernelGGL(k_lf_kv_reduce, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, s, kvpart, kv, nsplit, per, 1);
    const size_t smem = (size_t)(per + 8 * d) * sizeof(float);
    hipLaunchKernelGGL(k_lf_attn_apply<32>, dim3((Lseq + 7) / 8, 1), dim3(256), smem, s, qa, 3 * d, 0LL, kv, 0, atta, d, 0LL, Lseq, (float)Lseq, heads);
    lf_gemm(atta, d, nullptr, 0, 0, ly.merge.w, d, nullptr, msga, d, seq_rows, d, d, s);
    hipLaunchKernelGGL(k_lf_layernorm, dim3((unsigned)((seq_rows + 3) / 4)), dim3(256), 0, s, msga, ly.n1g, ly.n1b, (const float*)nullptr, msga, (long long)seq_rows, d, seq_rows, 0);
    lf_gemm(xa, d, msga, d, d, ly.mlp0.w, 2 * d, nullptr, hida, 2 * d, seq_rows, 2 * d, 2 * d, s, true);
    lf_gemm(hida, 2 * d, nullptr, 0, 0, ly.mlp2.w, 2 * d, nullptr, atta, d, seq_rows, d, 2 * d, s);
    hipLaunchKernelGGL(k_lf_layernorm, dim3((unsigned)((seq_rows + 3) / 4)), dim3(256), 0, s, atta, ly.n2g, ly.n2b, xa, xa, (long long)seq_rows, d, seq_rows, 0);
    return;
  }
  lf_gemm(x, d, nullptr, 0, 0, ly.qkv.w, d, nullptr, qkv, 3 * d, rows_pad, 3 * d, d, s, false, lim);
  const int chunk = Lseq <= 64 ? 64 : 192, nsplit = (Lseq + chunk - 1) / chunk;
  const long long per = (long long)heads * (hd + 1) * hd;
  const size_t smem = (size_t)(per + 8 * d) * sizeof(float);
  if (hd == 32) {
    hipLaunchKernelGGL(k_lf_kv_partial<32>, dim3(heads, nsplit, nall), dim3(256), 0, s, qkv + d, qkv + 2 * d, 3 * d, (long long)seq_rows * 3 * d, Lseq, chunk, (float)Lseq, kvpart, nsplit, heads);
    hipLaunchKernelGGL(k_lf_kv_reduce, dim3((unsigned)((per * nall + 255) / 256)), dim3(256), 0, s, kvpart, kv, nsplit, per, nall);
    hipLaunchKernelGGL(k_lf_attn_apply<32>, dim3((Lseq + 7) / 8, nall), dim3(256), smem, s, qkv, 3 * d, (long long)seq_rows * 3 * d, kv, cross, att, d, (long long)seq_rows * d, Lseq, (float)Lseq, heads);
  } else if (Lseq <= 32 && seq_rows == Lseq && d == 128) {
    hipLaunchKernelGGL(k_lf_fine_attn, dim3(nall), dim3(128), 0, s, qkv, Lseq, cross, att, 0, 0, lim.n, side_seqs > 0 ? side_seqs : nall);
  } else {
    hipLaunchKernelGGL(k_lf_kv_partial<16>, dim3(heads, nsplit, nall), dim3(256), 0, s, qkv + d, qkv + 2 * d, 3 * d, (long long)seq_rows * 3 * d, Lseq, chunk, (float)Lseq, kvpart, nsplit, heads);
    hipLaunchKernelGGL(k_lf_kv_reduce, dim3((unsigned)((per * nall + 255) / 256)), dim3(256), 0, s, kvpart, kv, nsplit, per, nall);
    hipLaunchKernelGGL(k_lf_attn_apply<16>, dim3((Lseq + 7) / 8, nall), dim3(256), smem, s, qkv, 3 * d, (long long)seq_rows * 3 * d, kv, cross, att, d, (long long)seq_rows * d, Lseq, (float)Lseq, heads);
  }
  lf_gemm(att, d, nullptr, 0, 0, ly.merge.w, d, nullptr, msg, d, rows_pad, d, d, s, false, lim);
  hipLaunchKernelGGL(k_lf_layernorm, dim3((unsigned)((rows_pad + 3) / 4)), dim3(256), 0, s, msg, ly.n1g, ly.n1b, (const float*)nullptr, msg, (long long)rows_pad, d, seq_rows, 0, lim.n, lim.mul, lim.seg);
  lf_gemm(x, d, msg, d, d, ly.mlp0.w, 2 * d, nullptr, hid, 2 * d, rows_pad, 2 * d, 2 * d, s, true, lim);
  lf_gemm(hid, 2 * d, nullptr, 0, 0, ly.mlp2.w, 2 * d, nullptr, att, d, rows_pad, d, 2 * d, s, false, lim);
  hipLaunchKernelGGL(k_lf_layernorm, dim3((unsigned)((rows_pad + 3) / 4)), dim3(256), 0, s, att, ly.n2g, ly.n2b, x, x, (long long)rows_pad, d, seq_rows, mode, lim.n, lim.mul, lim.seg);
}
}  // namespace

extern "C" {

const char* gn_loftr_last_error(const gn_loftr* ctx) { return ctx ? ctx->err.c_str() : g_lf_err.c_str(); }

int gn_loftr_create(int device, int H, int W, int max_matches, int fine, gn_loftr** out) {
  gn_loftr* ctx = nullptr;
  if (!out || H < 32 || W < 32 || (H % 8) || (W % 8) || max_matches < 1) return lf_fail(nullptr, GN_ERR_ARG, "gn_loftr_create: H, W multiples of 8 (>= 32), max_matches >= 1");
  LF_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  LF_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return lf_fail(nullptr, GN_ERR_ARCH, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
  ctx = new gn_loftr();
  ctx->device = device; ctx->H = H; ctx->W = W; ctx->hc = H / 8; ctx->wc = W / 8; ctx->L = ctx->hc * ctx->wc; ctx->Lp = (ctx->L + 127) / 128 * 128;
  ctx->fine = fine ? 1 : 0;
  // at most one match per coarse cell of image0, so L is "all of them" (what kornia returns).  The fine level's token rows (2 Mp x 25) are the
  // y dimension of its GEMM grids in units of 128: Mp <= 131072 keeps that below 65536 (a 4096 x 4096 image has 262144 cells: capped there)
  ctx->max_matches = std::min(std::min(max_matches, ctx->L), 131072);
  ctx->Mp = (ctx->max_matches + 127) / 128 * 128;
  const size_t h2 = H / 2, w2 = W / 2, h4 = H / 4, w4 = W / 4, hc = ctx->hc, wc = ctx->wc, Lp = ctx->Lp, L = ctx->L;
#define LF_A(field, n) do { int rc_ = lf_alloc(ctx, &ctx->field, (n)); if (rc_ != GN_OK) { gn_loftr_destroy(ctx); return rc_; } } while (0)
  LF_A(img, 2 * (size_t)H * W);
  LF_A(x0, 2 * h2 * w2 * 128); LF_A(x1, 2 * h2 * w2 * 128); LF_A(t1, 2 * h2 * w2 * 224);
  LF_A(x2, 2 * h4 * w4 * 224); LF_A(t2, 2 * h4 * w4 * 256); LF_A(x3, 2 * hc * wc * 256); LF_A(t3, 2 * hc * wc * 256);
  LF_A(x3_out, 2 * hc * wc * 256); LF_A(x2_out, 2 * h4 * w4 * 256); LF_A(fpn_a, 2 * h2 * w2 * 224); LF_A(fpn_b, 2 * h2 * w2 * 224); LF_A(x1_out, 2 * h2 * w2 * 128);
  LF_A(pe, L * 256); LF_A(tok, 2 * Lp * 256); LF_A(qkv, 2 * Lp * 768); LF_A(att, 2 * Lp * 256); LF_A(msg, 2 * Lp * 256); LF_A(hid, 2 * Lp * 512);
  LF_A(kvpart, 2 * ((L + 191) / 192) * 8 * 33 * 32); LF_A(kv, 2 * 8 * 33 * 32); LF_A(fs, 2 * Lp * 256); LF_A(sim, Lp * Lp);
  LF_A(rmax, Lp); LF_A(rsum, Lp); LF_A(cmax, Lp); LF_A(csum, Lp); LF_A(crow, Lp); LF_A(ccol, Lp); LF_A(cpart_a, 32 * Lp); LF_A(cpart_b, 32 * Lp); LF_A(csel, Lp);
  LF_A(ovf, 4); LF_A(jsel, Lp); LF_A(i_ids, ctx->Mp); LF_A(j_ids, ctx->Mp); LF_A(n_dev, 4); LF_A(k0c, 2 * (size_t)ctx->Mp); LF_A(k1c, 2 * (size_t)ctx->Mp); LF_A(mconf, ctx->Mp);
  if (ctx->fine) {
    const size_t R = 2 * (size_t)ctx->Mp * kLfWW;      // window tokens of both sides
    LF_A(fc, 2 * (size_t)ctx->Mp * 256); LF_A(fwin, 2 * (size_t)ctx->Mp * 128); LF_A(frows, R * 256); LF_A(ftok, R * 128); LF_A(fqkv, R * 384); LF_A(fatt, R * 128);
    LF_A(fmsg, R * 128); LF_A(fhid, R * 256); LF_A(fkvpart, 2 * (size_t)ctx->Mp * 8 * 17 * 16); LF_A(fkv, 2 * (size_t)ctx->Mp * 8 * 17 * 16);
  }
#undef LF_A
  if (hipHostMalloc((void**)&ctx->n_host, 4 * sizeof(int), hipHostMallocDefault) != hipSuccess) { gn_loftr_destroy(ctx); return lf_fail(nullptr, GN_ERR_HIP, "hipHostMalloc failed"); }
  {   // PositionEncodingSine, legacy divisor (temp_bug_fix = False): div_term_i = exp(-2 i); token l = y * wc + x, positions 1-based
    std::vector<float> pe((size_t)L * 256);
    for (int y = 0; y < (int)hc; ++y)
      for (int x = 0; x < (int)wc; ++x)
        for (int i = 0; i < 64; ++i) {
          const float div = std::exp((float)(2 * i) * -1.0f);
          float* p = &pe[((size_t)y * wc + x) * 256 + 4 * i];
          p[0] = std::sin((float)(x + 1) * div); p[1] = std::cos((float)(x + 1) * div);
          p[2] = std::sin((float)(y + 1) * div); p[3] = std::cos((float)(y + 1) * div);
        }
    if (hipMemcpy(ctx->pe, pe.data(), pe.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { gn_loftr_destroy(ctx); return lf_fail(nullptr, GN_ERR_HIP, "pe upload failed"); }
  }
  *out = ctx;
  return GN_OK;
}

void gn_loftr_destroy(gn_loftr* ctx) {
  if (!ctx) return;
  hipSetDevice(ctx->device);
  for (int i = 0; i < 2; ++i) if (ctx->graph_exec[i]) hipGraphExecDestroy(ctx->graph_exec[i]);
  if (ctx->cap_stream) hipStreamDestroy(ctx->cap_stream);
  for (void* p : ctx->allocs) hipFree(p);
  if (ctx->n_host) hipHostFree(ctx->n_host);
  delete ctx;
}

int gn_loftr_load_tensor(gn_loftr* ctx, const char* name_c, const float* host, const int64_t* shape, int ndim) {
  if (!ctx || !name_c || !host || !shape || ndim < 1 || ndim > 4) return lf_fail(ctx, GN_ERR_ARG, "bad gn_loftr_load_tensor argument");
  LF_HIP(hipSetDevice(ctx->device));
  const std::string name = name_c;
  if (name == "pos_encoding.pe" || name.find("num_batches_tracked") != std::string::npos) return GN_OK;
  const size_t dot = name.rfind('.');
  if (dot == std::string::npos) return lf_fail(ctx, GN_ERR_NAME, "unknown tensor " + name);
  const std::string base = name.substr(0, dot), leaf = name.substr(dot + 1);
  size_t count = 1;
  for (int i = 0; i < ndim; ++i) count *= (size_t)shape[i];
  int rc = GN_ERR_NAME;
  if (base.compare(0, 9, "backbone.") == 0) {
    LfConv& c = ctx->conv[base];
    if (ndim == 4 && leaf == "weight") {
      if (shape[2] != shape[3]) return lf_fail(ctx, GN_ERR_SHAPE, "square kernels only: " + name);
      c.cout = (int)shape[0]; c.cin = (int)shape[1]; c.ks = (int)shape[2];
      c.hw.assign(host, host + count);
      rc = GN_OK;
    } else if (ndim == 1) {
      const int k = leaf == "weight" ? 0 : leaf == "bias" ? 1 : leaf == "running_mean" ? 2 : leaf == "running_var" ? 3 : -1;
      if (k < 0) return lf_fail(ctx, GN_ERR_NAME, "unknown tensor " + name);
      c.bn[k].assign(host, host + count); c.has_bn = true;
      rc = GN_OK;
    }
    ctx->finalised = false;
  } else if (base.compare(0, 6, "loftr_") == 0) {
    const bool coarse = base.compare(0, 20, "loftr_coarse.layers.") == 0, finel = base.compare(0, 18, "loftr_fine.layers.") == 0;
    if (!coarse && !finel) return lf_fail(ctx, GN_ERR_NAME, "unknown tensor " + name);
    const size_t p0 = coarse ? 20 : 18;
    const int li = atoi(base.c_str() + p0);
    if (li < 0 || li >= (coarse ? 8 : 2)) return lf_fail(ctx, GN_ERR_NAME, "layer index out of range in " + name);
    LfLayer& ly = coarse ? ctx->coarse[li] : ctx->finel[li];
    const int d = coarse ? kLfDim : kLfFine;
    const std::string mod = base.substr(base.find('.', p0) + 1);
    auto lin_rows = [&](LfLinear& Lw, int rows_total, int row_off, int out, int in) -> int {
      if (ndim != 2 || shape[0] != out || shape[1] != in) return lf_fail(ctx, GN_ERR_SHAPE, "shape mismatch for " + name);
      if (!Lw.w) { int r = lf_alloc(ctx, &Lw.w, (size_t)rows_total * in); if (r != GN_OK) return r; }
      LF_HIP(hipMemcpy(Lw.w + (size_t)row_off * in, host, count * sizeof(float), hipMemcpyHostToDevice));
      Lw.out = rows_total; Lw.in = in;
      return GN_OK;
    };
    if (leaf == "weight" && mod == "q_proj") rc = lin_rows(ly.qkv, 3 * d, 0, d, d);
    else if (leaf == "weight" && mod == "k_proj") rc = lin_rows(ly.qkv, 3 * d, d, d, d);
    else if (leaf == "weight" && mod == "v_proj") rc = lin_rows(ly.qkv, 3 * d, 2 * d, d, d);
    else if (leaf == "weight" && mod == "merge") rc = lin_rows(ly.merge, d, 0, d, d);
    else if (leaf == "weight" && mod == "mlp.0") rc = lin_rows(ly.mlp0, 2 * d, 0, 2 * d, 2 * d);
    else if (leaf == "weight" && mod == "mlp.2") rc = lin_rows(ly.mlp2, d, 0, d, 2 * d);
    else if (mod == "norm1" || mod == "norm2") {
      if (ndim != 1 || shape[0] != d) return lf_fail(ctx, GN_ERR_SHAPE, "shape mismatch for " + name);
      float** dst = mod == "norm1" ? (leaf == "weight" ? &ly.n1g : &ly.n1b) : (leaf == "weight" ? &ly.n2g : &ly.n2b);
      rc = lf_upload(ctx, dst, host, d);
    }
  } else if (base == "fine_preprocess.down_proj" || base == "fine_preprocess.merge_feat") {
    LfLinear& Lw = base == "fine_preprocess.down_proj" ? ctx->down_proj : ctx->merge_feat;
    if (leaf == "weight") {
      if (ndim != 2 || shape[0] != kLfFine || shape[1] != 256) return lf_fail(ctx, GN_ERR_SHAPE, "shape mismatch for " + name);
      rc = lf_upload(ctx, &Lw.w, host, count); Lw.out = kLfFine; Lw.in = 256;
    } else if (leaf == "bias") {
      if (ndim != 1 || shape[0] != kLfFine) return lf_fail(ctx, GN_ERR_SHAPE, "shape mismatch for " + name);
      rc = lf_upload(ctx, &Lw.b, host, count);
    }
  }
  if (rc == GN_ERR_NAME) return lf_fail(ctx, GN_ERR_NAME, "unknown tensor " + name);
  if (rc == GN_OK) ctx->loaded[name] = true;
  return rc;
}

int gn_loftr_missing_tensors(const gn_loftr* ctx) {
  if (!ctx) return -1;
  int missing = 0;
  auto need = [&](const std::string& n) { if (!ctx->loaded.count(n)) ++missing; };
  for (int ci = 0; ci < kNumConv; ++ci) {
    if (!ctx->fine && ci >= 16) continue;              // the FPN head below 1/8 resolution only feeds the fine level
    need(std::string(kConvNames[ci]) + ".weight");
    if (kConvBn[ci][0]) for (const char* l : {".weight", ".bias", ".running_mean", ".running_var"}) need(std::string(kConvBn[ci]) + l);
  }
  for (int i = 0; i < 8 + (ctx->fine ? 2 : 0); ++i) {
    const std::string p = i < 8 ? "loftr_coarse.layers." + std::to_string(i) : "loftr_fine.layers." + std::to_string(i - 8);
    for (const char* l : {".q_proj.weight", ".k_proj.weight", ".v_proj.weight", ".merge.weight", ".mlp.0.weight", ".mlp.2.weight", ".norm1.weight", ".norm1.bias", ".norm2.weight", ".norm2.bias"}) need(p + l);
  }
  if (ctx->fine) for (const char* l : {"fine_preprocess.down_proj.weight", "fine_preprocess.down_proj.bias", "fine_preprocess.merge_feat.weight", "fine_preprocess.merge_feat.bias"}) need(l);
  return missing;
}

// LoFTR.forward on one pair of equally sized images.  image0 / image1: DEVICE f32 [H][W] in [0, 1].  Outputs (device): kpts0 / kpts1
// [max_matches][2] (x, y) pixels -- kpts0 on the 1/8 grid, kpts1 refined by the fine level when the context has one --, conf [max_matches],
// optional ij [max_matches][2] int32 coarse cell ids; *n_host (HOST) = number of matches (the call synchronises `stream` once to return it).
// the whole forward from ctx->img to the context's result buffers (k0c, k1c / fc, mconf, i_ids, j_ids, n_dev) on stream s: ~190 dependent
// launches with no host decision in between (the match count stays on the device), so it can be captured into one hipGraph
static int lf_forward(gn_loftr* ctx, hipStream_t s) {
  const int H = ctx->H, W = ctx->W, h2 = H / 2, w2 = W / 2, h4 = H / 4, w4 = W / 4, hc = ctx->hc, wc = ctx->wc, L = ctx->L, Lp = ctx->Lp;
  g_lf_gemm_variant = ctx->arith == 1 ? 6 : 3;
  LF_HIP(hipMemsetAsync(ctx->ovf, 0, sizeof(unsigned int), s));
  // ---- backbone (both images as a batch of 2)
  {
    const LfConv& c = ctx->conv["backbone.conv1"];
    const long long n = (long long)h2 * w2 * 8;
    if (gn::g_lf_conv_knob & 4) hipLaunchKernelGGL(k_lf_conv1<true>, dim3((unsigned)((n + 255) / 256), 1, 2), dim3(256), 0, s, ctx->img, c.wf, c.scale, c.shift, ctx->x0, H, W);
    else hipLaunchKernelGGL(k_lf_conv1<false>, dim3((unsigned)((n + 255) / 256), 1, 2), dim3(256), 0, s, ctx->img, c.wf, c.scale, c.shift, ctx->x0, H, W);
  }
  auto block = [&](const std::string& p, const float* x, int Hin, int Win, int stride, float* tmp, float* ds, float* out) {
    // y = relu(bn1(conv1(x))); y = bn2(conv2(y)); x' = stride != 1 ? bn(conv1x1(x)) : x; out = relu(x' + y)
    lf_conv(ctx, (p + ".conv1").c_str(), x, 2, Hin, Win, tmp, stride, nullptr, 1, s);
    const float* skip = x;
    if (stride != 1) { lf_conv(ctx, (p + ".downsample.0").c_str(), x, 2, Hin, Win, ds, stride, nullptr, 0, s); skip = ds; }
    lf_conv(ctx, (p + ".conv2").c_str(), tmp, 2, Hin / stride, Win / stride, out, 1, skip, 1, s);
  };
  block("backbone.layer1.0", ctx->x0, h2, w2, 1, ctx->t1, nullptr, ctx->x1);        // x0 -> x1
  block("backbone.layer1.1", ctx->x1, h2, w2, 1, ctx->t1, nullptr, ctx->x0);        // x1 -> x0 (= layer1 output, "x1" of the paper)
  float* const X1 = ctx->x0;
  block("backbone.layer2.0", X1, h2, w2, 2, ctx->t2, ctx->x2_out, ctx->x2);         // (x2_out is free until the FPN head)
  block("backbone.layer2.1", ctx->x2, h4, w4, 1, ctx->t2, nullptr, ctx->x2_out);
  LF_HIP(hipMemcpyAsync(ctx->x2, ctx->x2_out, (size_t)2 * h4 * w4 * 224 * sizeof(float), hipMemcpyDeviceToDevice, s));
  block("backbone.layer3.0", ctx->x2, h4, w4, 2, ctx->t3, ctx->x3_out, ctx->x3);
  block("backbone.layer3.1", ctx->x3, hc, wc, 1, ctx->t3, nullptr, ctx->x3_out);
  LF_HIP(hipMemcpyAsync(ctx->x3, ctx->x3_out, (size_t)2 * hc * wc * 256 * sizeof(float), hipMemcpyDeviceToDevice, s));
  lf_conv(ctx, "backbone.layer3_outconv", ctx->x3, 2, hc, wc, ctx->x3_out, 1, nullptr, 0, s);
  if (ctx->fine) {
    // x2_out = layer2_outconv2(layer2_outconv(x2) + up2(x3_out)); x1_out = layer1_outconv2(layer1_outconv(x1) + up2(x2_out))
    lf_conv(ctx, "backbone.layer2_outconv", ctx->x2, 2, h4, w4, ctx->t2, 1, nullptr, 0, s);
    { const long long n4 = 2LL * h4 * w4 * 64; hipLaunchKernelGGL(k_lf_up2_add, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, ctx->t2, ctx->x3_out, ctx->t2, h4, w4, 256, n4); }
    lf_conv(ctx, "backbone.layer2_outconv2.0", ctx->t2, 2, h4, w4, ctx->x2_out, 1, nullptr, 2, s);
    lf_conv(ctx, "backbone.layer2_outconv2.3", ctx->x2_out, 2, h4, w4, ctx->x2, 1, nullptr, 0, s);       // [.][224]: x2 is dead, reuse
    lf_conv(ctx, "backbone.layer1_outconv", X1, 2, h2, w2, ctx->fpn_a, 1, nullptr, 0, s);
    { const long long n4 = 2LL * h2 * w2 * 56; hipLaunchKernelGGL(k_lf_up2_add, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, ctx->fpn_a, ctx->x2, ctx->fpn_a, h2, w2, 224, n4); }
    lf_conv(ctx, "backbone.layer1_outconv2.0", ctx->fpn_a, 2, h2, w2, ctx->fpn_b, 1, nullptr, 2, s);
    lf_conv(ctx, "backbone.layer1_outconv2.3", ctx->fpn_b, 2, h2, w2, ctx->x1_out, 1, nullptr, 0, s);
  }
  // ---- coarse transformer
  hipLaunchKernelGGL(k_lf_posenc, dim3((unsigned)((2LL * Lp * 64 + 255) / 256)), dim3(256), 0, s, ctx->x3_out, ctx->pe, ctx->tok, L, Lp);
  for (int i = 0; i < 8; ++i) {
    const LfLayer& ly = ctx->coarse[i];
    if ((i & 1) == 0) {
      lf_encoder(ly, ctx->tok, Lp, L, 2 * Lp, kLfDim, 0, 0, ctx->qkv, ctx->att, ctx->msg, ctx->hid, ctx->kvpart, ctx->kv, s);
    } else {   // feat0 <- layer(feat0, feat1); then feat1 <- layer(feat1, feat0 UPDATED): two passes, each updating one sequence
      lf_encoder(ly, ctx->tok, Lp, L, 2 * Lp, kLfDim, 1, 1, ctx->qkv, ctx->att, ctx->msg, ctx->hid, ctx->kvpart, ctx->kv, s);
      lf_encoder(ly, ctx->tok, Lp, L, 2 * Lp, kLfDim, 1, 2, ctx->qkv, ctx->att, ctx->msg, ctx->hid, ctx->kvpart, ctx->kv, s);
    }
  }
  if (ctx->arith == 1) { const long long n4 = 2LL * Lp * 64; hipLaunchKernelGGL(k_lf_check_finite, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, ctx->tok, n4, ctx->ovf); }
  // ---- coarse matching
  const float temp = 0.1f;
  { const long long n4 = 2LL * Lp * 64; hipLaunchKernelGGL(k_lf_scale, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, ctx->tok, ctx->fs, 16.0f, n4); }
  lf_gemm(ctx->fs, 256, nullptr, 0, 0, ctx->fs + (size_t)Lp * 256, 256, nullptr, ctx->sim, Lp, Lp, Lp, 256, s);
  const int nsplit = 32, rows_per = (L + nsplit - 1) / nsplit;
  hipLaunchKernelGGL(k_lf_row_stats, dim3(L), dim3(256), 0, s, ctx->sim, Lp, L, temp, ctx->rmax, ctx->rsum);
  hipLaunchKernelGGL(k_lf_col_stats, dim3((L + 255) / 256, nsplit), dim3(256), 0, s, ctx->sim, Lp, L, temp, ctx->cpart_a, ctx->cpart_b, rows_per);
  hipLaunchKernelGGL(k_lf_col_merge, dim3((L + 255) / 256), dim3(256), 0, s, ctx->cpart_a, ctx->cpart_b, nsplit, L, ctx->cmax, ctx->csum);
  hipLaunchKernelGGL(k_lf_conf_rowmax, dim3(L), dim3(256), 0, s, ctx->sim, Lp, L, temp, ctx->rmax, ctx->rsum, ctx->cmax, ctx->csum, ctx->crow);
  hipLaunchKernelGGL(k_lf_conf_colmax, dim3((L + 255) / 256, nsplit), dim3(256), 0, s, ctx->sim, Lp, L, temp, ctx->rmax, ctx->rsum, ctx->cmax, ctx->csum, ctx->cpart_a, rows_per);
  hipLaunchKernelGGL(k_lf_max_merge, dim3((L + 255) / 256), dim3(256), 0, s, ctx->cpart_a, nsplit, L, ctx->ccol);
  hipLaunchKernelGGL(k_lf_mutual, dim3(L), dim3(256), 0, s, ctx->sim, Lp, L, hc, wc, temp, 0.2f, 2, ctx->rmax, ctx->rsum, ctx->cmax, ctx->csum, ctx->crow, ctx->ccol, ctx->jsel, ctx->csel);
  hipLaunchKernelGGL(k_lf_compact, dim3(1), dim3(1024), 0, s, ctx->jsel, ctx->csel, L, wc, 8, ctx->max_matches, ctx->i_ids, ctx->j_ids, ctx->mconf, ctx->k0c, ctx->k1c, ctx->n_dev);
  const int M = ctx->max_matches, Mp = ctx->Mp;
  if (ctx->fine) {
    // ---- fine level: windows of both sides as 2 Mp sequences of 25 tokens (side 0 first); cross pairs are (m, Mp + m)
    const int R = 2 * Mp * kLfWW;                       // token rows; 2 * Mp * 25 is a multiple of 128 (Mp is)
    hipLaunchKernelGGL(k_lf_coarse_gather, dim3((unsigned)((2LL * Mp * 64 + 255) / 256)), dim3(256), 0, s, ctx->tok, ctx->tok + (size_t)Lp * 256, ctx->i_ids, ctx->j_ids, ctx->n_dev, Mp, ctx->fc);
    // round 5 (developer knob 42, bit 1 = the form of rounds 3-4): the fine level works on the windows of the matches there ARE (the count stays on
    // the device: row tiles, rows and sequences behind it leave at once -- kornia's fine level runs on exactly the M matched windows) and keeps
    // the two sides one behind the other, so that a cross half projects, merges and normalises only the side it updates
    const bool lean = !(gn::g_lf_conv_knob & 2);
    LfLimit lim1, lim25;
    if (lean) { lim1.n = ctx->n_dev; lim1.mul = 1; lim1.seg = Mp; lim25.n = ctx->n_dev; lim25.mul = kLfWW; lim25.seg = Mp * kLfWW; }
    lf_gemm(ctx->fc, 256, nullptr, 0, 0, ctx->down_proj.w, 256, ctx->down_proj.b, ctx->fwin, 128, 2 * Mp, 128, 256, s, false, lim1);
    hipLaunchKernelGGL(k_lf_fine_gather, dim3((unsigned)((2LL * Mp * kLfWW * 64 + 255) / 256)), dim3(256), 0, s, ctx->x1_out, h2, w2, wc, ctx->i_ids, ctx->j_ids, ctx->n_dev, Mp, ctx->fwin, ctx->frows);
    lf_gemm(ctx->frows, 256, nullptr, 0, 0, ctx->merge_feat.w, 256, ctx->merge_feat.b, ctx->ftok, 128, R, 128, 256, s, false, lim25);
    if (lean) {
      // ftok [2][Mp][25][128] as it is: sequence q of side 0 pairs with q + Mp
      lf_encoder(ctx->finel[0], ctx->ftok, kLfWW, kLfWW, R, kLfFine, 0, 0, ctx->fqkv, ctx->fatt, ctx->fmsg, ctx->fhid, ctx->fkvpart, ctx->fkv, s, lim25, Mp);
      lf_encoder(ctx->finel[1], ctx->ftok, kLfWW, kLfWW, R, kLfFine, 1, 1, ctx->fqkv, ctx->fatt, ctx->fmsg, ctx->fhid, ctx->fkvpart, ctx->fkv, s, lim25, Mp);
      lf_encoder(ctx->finel[1], ctx->ftok, kLfWW, kLfWW, R, kLfFine, 1, 2, ctx->fqkv, ctx->fatt, ctx->fmsg, ctx->fhid, ctx->fkvpart, ctx->fkv, s, lim25, Mp);
    } else {
    // sequences: index q in [0, 2 Mp); its cross partner must be q ^ 1 for k_lf_attn_apply -> windows are stored INTERLEAVED? No: the
    // partner of window m of side 0 is window m of side 1, i.e. q + Mp.  The fine encoder therefore runs on a buffer re-ordered so that
    // sequence 2 m = side 0, 2 m + 1 = side 1 (fine_reorder below does it in place through frows).
    // (frows [2 Mp * 25][256] is free now: use its first half as the interleaved token buffer [2 Mp][25][128])
    float* ft = ctx->frows;
    for (int side = 0; side < 2; ++side)
      LF_HIP(hipMemcpy2DAsync(ft + (size_t)side * kLfWW * 128, 2 * (size_t)kLfWW * 128 * sizeof(float), ctx->ftok + (size_t)side * Mp * kLfWW * 128, (size_t)kLfWW * 128 * sizeof(float),
                              (size_t)kLfWW * 128 * sizeof(float), Mp, hipMemcpyDeviceToDevice, s));
    for (int i = 0; i < 2; ++i) {
      const LfLayer& ly = ctx->finel[i];
      if (i == 0) {
        lf_encoder(ly, ft, kLfWW, kLfWW, R, kLfFine, 0, 0, ctx->fqkv, ctx->fatt, ctx->fmsg, ctx->fhid, ctx->fkvpart, ctx->fkv, s);
      } else {
        // cross: even sequences (side 0) first, then odd sequences (side 1) against the UPDATED side 0
        lf_encoder(ly, ft, kLfWW, kLfWW, R, kLfFine, 1, 1, ctx->fqkv, ctx->fatt, ctx->fmsg, ctx->fhid, ctx->fkvpart, ctx->fkv, s);
        lf_encoder(ly, ft, kLfWW, kLfWW, R, kLfFine, 1, 2, ctx->fqkv, ctx->fatt, ctx->fmsg, ctx->fhid, ctx->fkvpart, ctx->fkv, s);
      }
    }
    // de-interleave into ftok: side 0 windows [Mp][25][128], side 1 behind them
    for (int side = 0; side < 2; ++side)
      LF_HIP(hipMemcpy2DAsync(ctx->ftok + (size_t)side * Mp * kLfWW * 128, (size_t)kLfWW * 128 * sizeof(float), ft + (size_t)side * kLfWW * 128, 2 * (size_t)kLfWW * 128 * sizeof(float),
                              (size_t)kLfWW * 128 * sizeof(float), Mp, hipMemcpyDeviceToDevice, s));
    }
    hipLaunchKernelGGL(k_lf_fine_match, dim3((M + 3) / 4), dim3(256), 0, s, ctx->ftok, ctx->ftok + (size_t)Mp * kLfWW * 128, ctx->n_dev, ctx->k1c, ctx->fc);
    if (ctx->arith == 1) { const long long n4 = (long long)M * 2 / 4; if (n4 > 0) hipLaunchKernelGGL(k_lf_check_finite, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, ctx->fc, n4, ctx->ovf); }
  }
  return GN_OK;
}

int gn_loftr_set_arithmetic(gn_loftr* ctx, int mode) {
  if (!ctx || mode < 0 || mode > 1) return GN_ERR_ARG;
  ctx->arith = mode;
  return GN_OK;
}

int gn_loftr_set_graph(gn_loftr* ctx, int enable) {
  if (!ctx) return GN_ERR_ARG;
  ctx->use_graph = enable ? 1 : 0;
  return GN_OK;
}

int gn_loftr_match(gn_loftr* ctx, const float* image0, const float* image1, float* kpts0, float* kpts1, float* conf, int32_t* ij, int32_t* n_host, void* stream) {
  if (!ctx || !image0 || !image1 || !kpts0 || !kpts1 || !conf || !n_host) return lf_fail(ctx, GN_ERR_ARG, "null pointer passed to gn_loftr_match");
  LF_HIP(hipSetDevice(ctx->device));
  if (gn_loftr_missing_tensors(ctx) != 0) return lf_fail(ctx, GN_ERR_WEIGHTS, "LoFTR weights not fully loaded");
  if (!ctx->finalised) { const int rc = lf_finalise(ctx); if (rc != GN_OK) return rc; }
  hipStream_t s = (hipStream_t)stream;
  const int H = ctx->H, W = ctx->W, M = ctx->max_matches;
  LF_HIP(hipMemcpyAsync(ctx->img, image0, (size_t)H * W * sizeof(float), hipMemcpyDeviceToDevice, s));
  LF_HIP(hipMemcpyAsync(ctx->img + (size_t)H * W, image1, (size_t)H * W * sizeof(float), hipMemcpyDeviceToDevice, s));
  bool ran = false;
  if (ctx->use_graph) {
    // every pointer inside the forward belongs to the context and the shapes are fixed at creation: capture the ~190 launches ONCE on an
    // internal stream and replay them as one graph launch (the host was the slower side between the small kernels of the fine level)
    hipGraphExec_t& gexec = ctx->graph_exec[ctx->arith];
    if (!gexec && !ctx->graph_failed[ctx->arith]) {
      hipGraph_t graph = nullptr;
      bool ok = true;
      if (!ctx->cap_stream) ok = hipStreamCreateWithFlags(&ctx->cap_stream, hipStreamNonBlocking) == hipSuccess;
      ok = ok && hipStreamBeginCapture(ctx->cap_stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
      if (ok) {
        const int rc = lf_forward(ctx, ctx->cap_stream);
        ok = hipStreamEndCapture(ctx->cap_stream, &graph) == hipSuccess && rc == GN_OK && graph != nullptr;
      }
      ok = ok && hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0) == hipSuccess;
      if (graph) hipGraphDestroy(graph);
      if (!ok) { ctx->graph_failed[ctx->arith] = true; gexec = nullptr; (void)hipGetLastError(); }
    }
    if (gexec) { LF_HIP(hipGraphLaunch(gexec, s)); ran = true; }
  }
  if (!ran) { const int rc = lf_forward(ctx, s); if (rc != GN_OK) return rc; }
  if (ctx->arith == 1) {   // split-fp16 arithmetic: a value outside fp16's range anywhere in the forward -> repeat it on the exact-f32 kernels
    LF_HIP(hipMemcpyAsync(ctx->n_host + 1, ctx->ovf, sizeof(unsigned int), hipMemcpyDeviceToHost, s));
    LF_HIP(hipStreamSynchronize(s));
    if (ctx->n_host[1] != 0) {
      ++ctx->ovf_trips;
      ctx->arith = 0;
      const int rc = lf_forward(ctx, s);
      ctx->arith = 1;
      if (rc != GN_OK) return rc;
    }
  }
  LF_HIP(hipMemcpyAsync(kpts0, ctx->k0c, (size_t)M * 2 * sizeof(float), hipMemcpyDeviceToDevice, s));
  LF_HIP(hipMemcpyAsync(conf, ctx->mconf, (size_t)M * sizeof(float), hipMemcpyDeviceToDevice, s));
  LF_HIP(hipMemcpyAsync(kpts1, ctx->fine ? ctx->fc : ctx->k1c, (size_t)M * 2 * sizeof(float), hipMemcpyDeviceToDevice, s));
  if (ij) {
    LF_HIP(hipMemcpy2DAsync(ij, 2 * sizeof(int), ctx->i_ids, sizeof(int), sizeof(int), M, hipMemcpyDeviceToDevice, s));
    LF_HIP(hipMemcpy2DAsync(ij + 1, 2 * sizeof(int), ctx->j_ids, sizeof(int), sizeof(int), M, hipMemcpyDeviceToDevice, s));
  }
  LF_HIP(hipMemcpyAsync(ctx->n_host, ctx->n_dev, sizeof(int), hipMemcpyDeviceToHost, s));
  LF_HIP(hipStreamSynchronize(s));
  *n_host = ctx->n_host[0];
  LF_HIP(hipGetLastError());
  return GN_OK;
}

// test hook: copy an internal tensor to HOST memory after synchronising.  Names: "x0" (stem, [2][H/2][W/2][128]), "x1" (layer1 output),
// "x2" (224-channel rows), "x3", "x3_out", "x1_out", "tok" ([2][Lp][256] coarse tokens after the transformer), "sim", "conf_row" / "conf_col" maxima.
int64_t gn_loftr_debug_read(gn_loftr* ctx, const char* name, void* host_out, int64_t max_bytes, void* stream) {
  if (!ctx || !name || !host_out) return GN_ERR_ARG;
  if (hipSetDevice(ctx->device) != hipSuccess || hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return GN_ERR_HIP;
  const size_t h2 = ctx->H / 2, w2 = ctx->W / 2, h4 = ctx->H / 4, w4 = ctx->W / 4, hc = ctx->hc, wc = ctx->wc, Lp = ctx->Lp;
  const std::string n = name;
  const float* p = nullptr; size_t count = 0;
  if (n == "x1") { p = ctx->x0; count = 2 * h2 * w2 * 128; }
  else if (n == "x2") { p = ctx->x2; count = 2 * h4 * w4 * 224; }
  else if (n == "x3") { p = ctx->x3; count = 2 * hc * wc * 256; }
  else if (n == "x3_out") { p = ctx->x3_out; count = 2 * hc * wc * 256; }
  else if (n == "x1_out") { p = ctx->x1_out; count = 2 * h2 * w2 * 128; }
  else if (n == "tok") { p = ctx->tok; count = 2 * Lp * 256; }
  else if (n == "sim") { p = ctx->sim; count = Lp * Lp; }
  else if (n == "crow") { p = ctx->crow; count = Lp; }
  else if (n == "ccol") { p = ctx->ccol; count = Lp; }
  else if (n == "ftok") { p = ctx->ftok; count = ctx->fine ? 2 * (size_t)ctx->Mp * kLfWW * 128 : 0; }
  else return GN_ERR_NAME;
  count = std::min<size_t>(count, (size_t)max_bytes / sizeof(float));
  if (hipMemcpy(host_out, p, count * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return GN_ERR_HIP;
  return (int64_t)count;
}

}  // extern "C"
