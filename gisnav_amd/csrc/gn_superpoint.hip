// SuperPoint extractor (DeTone et al.) -- the "conv backbone" BASELINE.json's north_star / configs[4] name for the deep-matcher path
// (the reference tree does not contain it; spec = the published architecture, pinned through
// oracle/superpoint.py to the `transformers` port).  Every kernel is hand-written for gfx950:
//
//   k_sp_conv1        first layer, 1 -> 64 channels, 3x3: plain FMAs (9 taps per output, VALU)
//   k_sp_conv<TAPS>   3x3 (TAPS = 9) or 1x1 (TAPS = 1) convolution + bias (+ ReLU), NHWC f32, as an implicit GEMM on the EXACT f32
//                     matrix instruction v_mfma_f32_32x32x2_f32 (bitwise an fmaf chain): weights are the A operand, fetched
//                     from L2 straight into registers in MFMA fragment order (re-laid-out once at load time, like gn_ffn.hip);
//                     output pixels are the B operand, read from an LDS halo tile of 18 x 34 pixels x 32 input channels
//   k_sp_conv_s       (round 5) the split-fp16 convolution on hm16 activation RECORDS: halo tile and weight fragments by LDS-DMA, unrolled taps,
//                     scalar VALU-lean epilogue through an LDS slab; bitwise k_sp_conv<., 1, ...>'s results.  Runs the 1 x 1 head layers
//   k_sp_conv_s16     (round 5) the same with 16-channel slices and four workgroups per CU: the 3 x 3 layers of the split mode (matrix pipe busy 0.72-0.81)
//   k_sp_conv_h16     (round 5) k_sp_conv_s16's form for GN_SP_FP16 (fp16 activations, one product per block)
//   k_sp_nms_fused    (round 5) simple_nms in one kernel (the five 9 x 9 pools of a 64 x 32 tile in LDS); k_sp_maxrow/col + k_sp_nms_step stay as knob 36 = 0
//   k_sp_pool         2x2 max-pool
//   k_sp_scores       65-way softmax per 8x8 cell, dustbin dropped, depth-to-space -> full-resolution score map
//   k_sp_maxrow/col   separable (2 r + 1)^2 max-pool passes of simple_nms (restated exactly: -inf padding)
//   k_sp_nms_step     the mask algebra of simple_nms between the pools
//   k_sp_candidates   threshold 0.005 + border 4 -> candidate list (atomic append)
//   k_sp_select       top-k by score: 4 x 8-bit radix select of the k-th largest score, compaction, rank sort (score descending,
//                     raster index ascending on ties) -- one workgroup per image
//   k_sp_describe     per keypoint: bilinear sample (align_corners) of the L2-normalised 1/8-resolution descriptor map, L2 normalise
//
// The convolutions of f32 contexts are f32-exact (no reduced-precision operand anywhere), so keypoints agree with an f32 CPU run except where
// two scores tie to the last bit; f16x2 contexts run the f32-ACCURATE split-fp16 kernels.
#include "gn_common.h"
#include "gn_ffn_util.h"

#include <cmath>

namespace gn {

namespace {
constexpr int TW = 32;                   // output tile of k_sp_conv: (4 RPW) rows x 32 columns, 64 output channels.  RPW = 3 rows per wave: halo tile
                                         // 61 KB + 16 KB of weights = two workgroups per CU (one stages its tile while the other is on the matrix
                                         // pipe); RPW = 2 for the layers whose 2 x 2 max-pool is fused into the epilogue (row pairs in one lane)
constexpr int CH = 32;                   // input channels staged per pass (128 B per pixel in LDS)

template <int OUT_FMT>   // 0: f32 NHWC; 1: the 64-channel map leaves as fp16 (GN_SP_FP16); 2: as hm16 records (16 high terms, 16 residual terms per 16 channels:
                         // what k_sp_conv_s stages without touching it; the thread's 16 channels are one record group, 64 contiguous bytes)
__global__ __launch_bounds__(256) void k_sp_conv1(const float* in, const float* w /*[64][9]*/, const float* bias, float* out, int H, int W, unsigned int* ovf) {
  // the 576 weights + 64 biases sit in LDS: a lane's channel group differs from its neighbours', so reading them from memory was
  // 144 vector loads per thread (the layer ran at 1.5 TB/s of output instead of the HBM rate).  Round 5: a thread owns FOUR horizontally
  // adjacent pixels of its 16-channel group -- one pixel per thread spent 160 LDS reads and 144 FMAs per 64 output bytes and was bound
  // by exactly that (0.78 ms per four 1080p frames for 2.1 GB of output); now a weight read feeds four FMAs.  Per output the FMA chain is
  // the same (bias, then the nine taps in order): same bits.
  __shared__ float ws[64 * 9 + 64];
  __shared__ __attribute__((aligned(16))) unsigned char slab[OUT_FMT == 2 ? 4 * 4096 : 16];
  for (int q = threadIdx.x; q < 64 * 9 + 64; q += 256) ws[q] = q < 576 ? w[q] : bias[q - 576];
  __syncthreads();
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const int grp = (int)(idx & 3);
  const long long quad = idx >> 2;
  const int wq = W >> 2;
  if (OUT_FMT != 2 && quad >= (long long)H * wq) return;      // (the hm16 form keeps whole waves: its rows leave through a wave-wide LDS exchange)
  const int y = (int)(quad / wq), x = 4 * (int)(quad - (long long)y * wq);
  const float* img = in + (long long)blockIdx.z * H * W;
  float v[3][6];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const int yy = y + r - 1, xx = x + c - 1;
      v[r][c] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? img[(long long)yy * W + xx] : 0.f;
    }
  typedef _Float16 h16x4_t __attribute__((ext_vector_type(4)));
  typedef _Float16 h16x8_t __attribute__((ext_vector_type(8)));
  float amax = 0.f;
  float r[4][16];
#pragma unroll
  for (int cc = 0; cc < 16; ++cc) {
    const int c = grp * 16 + cc;
    float wt[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wt[t] = ws[c * 9 + t];
    const float bs = ws[576 + c];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      float acc = bs;
#pragma unroll
      for (int t = 0; t < 9; ++t) acc = fmaf(wt[t], v[t / 3][p + t % 3], acc);
      r[p][cc] = fmaxf(acc, 0.f);
    }
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const long long pix = (long long)blockIdx.z * H * W + (long long)y * W + x + p;
    if (OUT_FMT == 1) {
      _Float16* oh = reinterpret_cast<_Float16*>(out) + pix * 64 + grp * 16;
#pragma unroll
      for (int c8 = 0; c8 < 2; ++c8) {
        h16x8_t h;
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = (_Float16)r[p][8 * c8 + e];
        *reinterpret_cast<h16x8_t*>(oh + 8 * c8) = h;
      }
    } else if (OUT_FMT == 2) {
      // the thread's record group of this pixel: 16 high terms, 16 residual terms = 64 contiguous bytes.  Stored directly, an instruction wrote 64
      // separate 16-byte pieces (16 of every 64 bytes, quads 1 KB apart); through the wave's LDS slab an instruction writes 1 KB in one piece
      // (the wave's 16 quads x 4 groups of pixel p are 16 records 1 KB apart: slab [quad][256 B], read back as [quad][16 lanes x 16 B])
      unsigned char* const sl = slab + (threadIdx.x >> 6) * 4096;
      const int lane = threadIdx.x & 63;
#pragma unroll
      for (int c8 = 0; c8 < 2; ++c8) {
        h16x8_t h, m;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float f = r[p][8 * c8 + e];
          amax = fmaxf(amax, f);          // (ReLU'd: no sign)
          h[e] = (_Float16)f;
          m[e] = (_Float16)(f - (float)h[e]);
        }
        *reinterpret_cast<h16x8_t*>(sl + (lane >> 2) * 256 + grp * 64 + 16 * c8) = h;
        *reinterpret_cast<h16x8_t*>(sl + (lane >> 2) * 256 + grp * 64 + 32 + 16 * c8) = m;
      }
      // read-back: instruction it covers quads 4 it .. 4 it + 3 of the wave: lane -> (quad 4 it + (lane >> 4), 16-byte chunk lane & 15) of pixel p's record
      const long long wq0 = quad - (lane >> 2);          // the wave's first quad (consecutive quads of one row, or the tail of the grid)
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int qq = 4 * it + (lane >> 4);
        const uint4 v4 = *reinterpret_cast<const uint4*>(sl + qq * 256 + (lane & 15) * 16);
        const long long qd = wq0 + qq;
        if (qd < (long long)H * wq) {
          const int yq = (int)(qd / wq), xq = 4 * (int)(qd - (long long)yq * wq);
          *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(out) + (((long long)blockIdx.z * H + yq) * W + xq + p) * 256 + (lane & 15) * 16) = v4;
        }
      }
    } else {
      float* o = out + pix * 64 + grp * 16;
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) *reinterpret_cast<f32x4*>(o + c4 * 4) = (f32x4){r[p][4 * c4], r[p][4 * c4 + 1], r[p][4 * c4 + 2], r[p][4 * c4 + 3]};
    }
  }
  if (OUT_FMT == 2) ovf_commit(ovf, amax);
}

struct ConvArgs {
  const float* in;  int H, W, Cin;          // NHWC f32 [B][H][W][Cin], Cin a multiple of 32
  const float* wf;                           // weights in fragment order (sp_weight_fragments): [Cout/32][TAPS][Cin/8][64 lanes][4]
  const float* bias;                         // [Cout_padded]
  float* out; int Cout;                      // NHWC f32 [B][H][W][Cout], Cout a multiple of 64 (padded with zero weights)
  int relu;
  const uint16_t* wfh; float acc_scale;      // HM variant: weights as fp16 pairs in fragment order (sp_weight_fragments_hm16), scaled by 1 / acc_scale
  unsigned int* ovf;                         // HM variant: raised when an input activation does not fit fp16 (the caller re-runs the exact path)
  long long* dbg_ts;                         // developer: s_memtime stamps of k_sp_conv_s's phases, [workgroup][32] (tools/sp_phases.py), or nullptr
  int in_half, out_half;                     // GN_SP_FP16 only: the layer reads / writes its NHWC activations as fp16 (half the HBM traffic of the full-resolution layers)
  // k_sp_conv_s16<., ., true> (round 6): the network's FIRST convolution (1 -> 64, 3 x 3 + ReLU) is evaluated while the halo tile is staged, from the
  // gray image itself -- `in` is then unused, Cin = 64
  const float* img = nullptr; const float* w1 = nullptr; const float* b1 = nullptr;   // [B][H][W] f32 image, conv1a weight [64][9], bias [64]
};

// grid (tiles_x, tiles_y, B * Cout/64)
// HM = false: exact f32 (v_mfma_f32_32x32x2_f32, 64 flops / clk / SIMD).  HM = true (contexts of the f16x2 precision mode): every f32
// operand as two fp16 terms, three v_mfma_f32_32x32x16_f16 per 16 input channels, f32 accumulation -- the arithmetic of the matcher's
// GEMMs (gn_gemm_p2.hip: error <= the f32 pipe's) at 5 x the matrix-pipe rate.  The activations stay f32 in memory: the halo tile is
// split while it is staged (hm16 row format: 16 high terms, 16 residual terms per 16 channels -- the same 128 bytes per pixel).
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4_sp __attribute__((ext_vector_type(4)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
// 16-byte chunk c of halo pixel (ly, lx) sits at position c ^ psw(lx): pixels lx, lx + 8, lx + 16, lx + 24 of a wave's fragment read
// must not share a chunk position (a plain lx & 7 made every ds_read_b128 a 4-way bank conflict: the LDS port, not the matrix
// pipe, set the pace of the full-resolution layers)
__device__ __forceinline__ int psw(int lx) { return (lx ^ (lx >> 3)) & 7; }
template <int TAPS, int HM, int RPW, bool POOL>   // HM: 0 exact f32, 1 split fp16 (3 products), 2 single fp16 product (the arithmetic BASELINE configs[4] names: 16-bit operands)
__global__ __launch_bounds__(256) void k_sp_conv(ConvArgs a) {
  constexpr int TH = 4 * RPW;
  constexpr int HALO = TAPS == 9 ? 1 : 0;
  constexpr int LW = TW + 2 * HALO, LH = TH + 2 * HALO;
  __shared__ __attribute__((aligned(16))) float tile[LH * LW * CH];
  __shared__ __attribute__((aligned(16))) unsigned char wbuf[HM ? 2 * 8192 : 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, ql = lane & 31;
  const int ogroups = a.Cout / 64;
  const int img = blockIdx.z / ogroups, og = blockIdx.z % ogroups;
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
  const float* in = a.in + (long long)img * a.H * a.W * a.Cin;
  const int csteps = a.Cin / 8;
  const f32x4* wf = reinterpret_cast<const f32x4*>(a.wf) + lane;
  const uint4* wfh = reinterpret_cast<const uint4*>(a.wfh) + lane;
  unsigned char* const tb = reinterpret_cast<unsigned char*>(tile);

  f32x16 acc[2][RPW];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < RPW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float amax = 0.f;

  for (int c0 = 0; c0 < a.Cin; c0 += CH) {
    __syncthreads();     // the previous pass is done with the tile
    // stage the halo tile of this 32-channel slice: thread -> (pixel, 16-byte chunk), zero outside the image.  The loads are issued in
    // batches of SB independent requests (one memory latency per batch; a load-convert-write loop paid one per ITEM: 19 latencies
    // per slice, twice the time of the slice's MFMAs)
    constexpr int NQ = (LH * LW * 8 + 255) / 256, SB = 10;
    if (HM == 2 && a.in_half) {
      // fp16 activations: a 16-byte piece is 8 channels and goes into the high-term position of the tile as it is (no conversion)
      constexpr int NQH = (LH * LW * 4 + 255) / 256;
      const uint16_t* inh = reinterpret_cast<const uint16_t*>(a.in) + (long long)img * a.H * a.W * a.Cin;
#pragma unroll
      for (int q0 = 0; q0 < NQH; q0 += SB) {
        uint4 v[SB];
#pragma unroll
        for (int e = 0; e < SB; ++e) {
          const int q = (q0 + e) * 256 + tid;
          const int pix = q >> 2, c8 = q & 3;
          const int ly = pix / LW, lx = pix - ly * LW;
          const int gy = y0 + ly - HALO, gx = x0 + lx - HALO;
          v[e] = make_uint4(0u, 0u, 0u, 0u);
          if (q0 + e < NQH && q < LH * LW * 4 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
            v[e] = *reinterpret_cast<const uint4*>(inh + ((long long)gy * a.W + gx) * a.Cin + c0 + c8 * 8);
        }
#pragma unroll
        for (int e = 0; e < SB; ++e) {
          const int q = (q0 + e) * 256 + tid;
          if (q0 + e >= NQH || q >= LH * LW * 4) continue;
          const int pix = q >> 2, c8 = q & 3;
          const int ly = pix / LW, lx = pix - ly * LW;
          const int piece = 4 * (c8 >> 1) + (c8 & 1);
          *reinterpret_cast<uint4*>(tb + pix * 128 + ((piece ^ psw(lx)) * 16)) = v[e];
        }
      }
    } else
#pragma unroll
    for (int q0 = 0; q0 < NQ; q0 += SB) {
      f32x4 v[SB];
#pragma unroll
      for (int e = 0; e < SB; ++e) {
        const int q = (q0 + e) * 256 + tid;
        const int pix = q >> 3, chunk = q & 7;
        const int ly = pix / LW, lx = pix - ly * LW;
        const int gy = y0 + ly - HALO, gx = x0 + lx - HALO;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        v[e] = z;
        if (q0 + e < NQ && q < LH * LW * 8 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
          v[e] = *reinterpret_cast<const f32x4*>(in + ((long long)gy * a.W + gx) * a.Cin + c0 + chunk * 4);
      }
#pragma unroll
      for (int e = 0; e < SB; ++e) {
        const int q = (q0 + e) * 256 + tid;
        if (q0 + e >= NQ || q >= LH * LW * 8) continue;
        const int pix = q >> 3, chunk = q & 7;
        const int ly = pix / LW, lx = pix - ly * LW;
        if (HM) {
          // channels 4 chunk .. 4 chunk + 3 of the slice: k-step chunk >> 2, 16-byte piece 4 (chunk >> 2) + 2 term + ((chunk & 3) >> 1), half (chunk & 1)
          const h16x4 h4 = __builtin_convertvector(v[e], h16x4);
          ovf_track(amax, v[e].x, v[e].y); ovf_track(amax, v[e].z, v[e].w);
          const int piece = 4 * (chunk >> 2) + ((chunk & 3) >> 1), sub = (chunk & 1) * 8;
          *reinterpret_cast<h16x4*>(tb + pix * 128 + ((piece ^ psw(lx)) * 16) + sub) = h4;
          if (HM == 1) {
            const h16x4 m4 = __builtin_convertvector(v[e] - __builtin_convertvector(h4, f32x4), h16x4);
            *reinterpret_cast<h16x4*>(tb + pix * 128 + (((piece + 2) ^ psw(lx)) * 16) + sub) = m4;
          }
        } else {
          *reinterpret_cast<f32x4*>(tile + pix * CH + ((chunk ^ psw(lx)) * 4)) = v[e];
        }
      }
    }
    __syncthreads();
    if (HM) {
      // The weight fragments of one tap of this slice (2 channel tiles x 2 k-steps x 2 terms = 8 KB) are fetched ONCE per workgroup into
      // LDS (two buffers, the next tap's in flight during this tap's MFMAs) and read from there by the four waves: fetched per wave
      // they were 590 KB of L2 -> CU traffic per workgroup, 4 x what the layer's input is, and the per-CU ingest rate set the pace.
      auto wsrc = [&](int tap, int e) __attribute__((always_inline)) {    // thread's e-th 16-byte piece: block 4 e + (tid >> 6) = ((i * 2 + s) * 2 + pl)
        const int blk = 4 * e + (tid >> 6), i = blk >> 2, ks = (blk >> 1) & 1, pl = blk & 1;
        return wfh[(size_t)((((2 * og + i) * TAPS + tap) * (a.Cin / 16) + (c0 / 16 + ks)) * 2 + pl) * 64];
      };
      uint4 wnext[2];
      wnext[0] = wsrc(0, 0); wnext[1] = wsrc(0, 1);
#pragma unroll 1
      for (int tap = 0; tap < TAPS; ++tap) {
        const int dy = TAPS == 9 ? tap / 3 : 0, dx = TAPS == 9 ? tap % 3 : 0;    // offsets into the halo tile (already shifted by HALO)
        unsigned char* const wb = wbuf + (tap & 1) * 8192;
        *reinterpret_cast<uint4*>(wb + (tid >> 6) * 1024 + lane * 16) = wnext[0];
        *reinterpret_cast<uint4*>(wb + (4 + (tid >> 6)) * 1024 + lane * 16) = wnext[1];
        if (tap + 1 < TAPS) { wnext[0] = wsrc(tap + 1, 0); wnext[1] = wsrc(tap + 1, 1); }
        __syncthreads();     // this tap's weights are in LDS (the buffer written two taps ago is no longer read: one barrier per tap in between)
#pragma unroll
        for (int s = 0; s < CH / 16; ++s) {
          h16x8 fa[2][2], fb[RPW][2];
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
              fa[i][pl] = *reinterpret_cast<const h16x8*>(wb + ((i * 2 + s) * 2 + pl) * 1024 + lane * 16);
#pragma unroll
          for (int j = 0; j < RPW; ++j) {
            const int ly = RPW * wave + j + dy, lx = ql + dx;
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
              fb[j][pl] = *reinterpret_cast<const h16x8*>(tb + (ly * LW + lx) * 128 + (((4 * s + 2 * pl + hh) ^ psw(lx)) * 16));
          }
          // products: W_m X_h, W_h X_m, W_h X_h (small terms first)
#pragma unroll
          for (int p = (HM == 2 ? 2 : 0); p < 3; ++p)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < RPW; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][p == 0 ? 1 : 0], fb[j][p == 1 ? 1 : 0], acc[i][j], 0, 0, 0);
        }
      }
    } else {
#pragma unroll 1
    for (int tap = 0; tap < TAPS; ++tap) {
      const int dy = TAPS == 9 ? tap / 3 : 0, dx = TAPS == 9 ? tap % 3 : 0;    // offsets into the halo tile (already shifted by HALO)
      {
#pragma unroll
        for (int s = 0; s < CH / 8; ++s) {
          f32x4 fa[2], fb[RPW];
#pragma unroll
          for (int i = 0; i < 2; ++i)
            fa[i] = wf[(size_t)(((2 * og + i) * TAPS + tap) * csteps + (c0 / 8 + s)) * 64];
#pragma unroll
          for (int j = 0; j < RPW; ++j) {
            const int ly = RPW * wave + j + dy, lx = ql + dx;
            fb[j] = *reinterpret_cast<const f32x4*>(tile + (ly * LW + lx) * CH + (((2 * s + hh) ^ psw(lx)) * 4));
          }
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < RPW; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
        }
      }
    }
    }
  }
  if (HM) ovf_commit(a.ovf, amax);
  // epilogue: lane = pixel (row RPW wave + j, column ql); registers 4 g + c = output channels 32 i + 8 g + 4 hh + c
  const int gx = x0 + ql;
  const float ascale = HM ? a.acc_scale : 1.f;
  if (POOL) {
    // fused 2 x 2 max-pool (RPW = 2: the two rows of a wave are one pooling row pair; the column pair is lanes ql, ql ^ 1):
    // the full-resolution map is never written -- out is [H / 2][W / 2][Cout]
    static_assert(!POOL || RPW == 2, "the fused pool pairs the two rows of a wave");
    float* out = a.out + (long long)img * (a.H / 2) * (a.W / 2) * a.Cout;
    const int gy = y0 + RPW * wave;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = 64 * og + 32 * i + 8 * g + 4 * hh;
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias + c);
        f32x4 m;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v0 = acc[i][0][4 * g + e], v1 = acc[i][RPW - 1][4 * g + e];
          if (HM) { v0 = v0 * ascale; v1 = v1 * ascale; }
          v0 += b4[e]; v1 += b4[e];
          if (a.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
          float mv = fmaxf(v0, v1);
          mv = fmaxf(mv, __shfl_xor(mv, 1));
          m[e] = mv;
        }
        if (HM == 2 && a.out_half) {
          ovf_track(amax, m.x, m.y); ovf_track(amax, m.z, m.w);
          if (!(ql & 1) && gy + 1 < a.H && gx + 1 < a.W)
            *reinterpret_cast<h16x4*>(reinterpret_cast<uint16_t*>(a.out) + (long long)img * (a.H / 2) * (a.W / 2) * a.Cout +
                                      ((long long)(gy >> 1) * (a.W / 2) + (gx >> 1)) * a.Cout + c) = __builtin_convertvector(m, h16x4);
        } else if (!(ql & 1) && gy + 1 < a.H && gx + 1 < a.W)
          *reinterpret_cast<f32x4*>(out + ((long long)(gy >> 1) * (a.W / 2) + (gx >> 1)) * a.Cout + c) = m;
      }
    if (HM == 2 && a.out_half) ovf_commit(a.ovf, amax);
    return;
  }
  float* out = a.out + (long long)img * a.H * a.W * a.Cout;
#pragma unroll
  for (int j = 0; j < RPW; ++j) {
    const int gy = y0 + RPW * wave + j;
    if (gy >= a.H || gx >= a.W) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = 64 * og + 32 * i + 8 * g + 4 * hh;
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias + c);
        f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        if (HM) v = v * ascale;
        v += b4;
        if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (HM == 2 && a.out_half) {
          ovf_track(amax, v.x, v.y); ovf_track(amax, v.z, v.w);
          *reinterpret_cast<h16x4*>(reinterpret_cast<uint16_t*>(a.out) + (long long)img * a.H * a.W * a.Cout + ((long long)gy * a.W + gx) * a.Cout + c) =
              __builtin_convertvector(v, h16x4);
        } else {
          *reinterpret_cast<f32x4*>(out + ((long long)gy * a.W + gx) * a.Cout + c) = v;
        }
      }
  }
  if (HM == 2 && a.out_half) ovf_commit(a.ovf, amax);
}

// GN_SP_FP16, 3 x 3 layers: the single-product arithmetic of k_sp_conv<9, 2, ...> with a tiling made for it.  With one MFMA per
// fragment pair that kernel reads 1 KB of LDS per MFMA and the LDS port sets its pace; here a wave owns FOUR pixel rows, the taps are
// walked column by column (dx), the three weight fragments of a kernel column stay in registers and a pixel fragment (tile row r,
// column ql + dx) is read ONCE for the up to three taps (dy = r - j) that use it: 12 LDS reads per 24 MFMAs.  The halo tile holds high
// terms only (64 bytes per pixel: chunk 2 s + hh of pixel (ly, lx) at position (2 s + hh) ^ ((lx >> 2) & 3), conflict-free for the lane
// groups of ds_read_b128), 18 x 34 pixels = 38 KB, + 24 KB of weight hand-over: two workgroups per CU.
constexpr int RPW4 = 4, TH4 = 4 * RPW4, LH4 = TH4 + 2, LW4 = TW + 2;
template <bool POOL>
__global__ __launch_bounds__(256, 2) void k_sp_conv_h(ConvArgs a) {   // two workgroups per CU: at most 256 registers
  __shared__ __attribute__((aligned(16))) unsigned char tb[LH4 * LW4 * 64];
  __shared__ __attribute__((aligned(16))) unsigned char wbuf[2 * 12288];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, ql = lane & 31;
  const int ogroups = a.Cout / 64;
  const int img = blockIdx.z / ogroups, og = blockIdx.z % ogroups;
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH4;
  const uint4* wfh = reinterpret_cast<const uint4*>(a.wfh) + lane;
  auto fsw = [](int lx) { return (lx >> 2) & 3; };

  f32x16 acc[2][RPW4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < RPW4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float amax = 0.f;

  for (int c0 = 0; c0 < a.Cin; c0 += CH) {
    __syncthreads();     // the previous slice is done with the tile
    // stage the halo tile of this 32-channel slice: thread -> (pixel, 16-byte chunk of 8 channels); zero outside the image
    constexpr int NQ = (LH4 * LW4 * 4 + 255) / 256, SB = 10;
    if (a.in_half) {
      const uint16_t* inh = reinterpret_cast<const uint16_t*>(a.in) + (long long)img * a.H * a.W * a.Cin;
#pragma unroll
      for (int q0 = 0; q0 < NQ; q0 += SB) {
        uint4 v[SB];
#pragma unroll
        for (int e = 0; e < SB; ++e) {
          const int q = (q0 + e) * 256 + tid;
          const int pix = q >> 2, c8 = q & 3;
          const int ly = pix / LW4, lx = pix - ly * LW4;
          const int gy = y0 + ly - 1, gx = x0 + lx - 1;
          v[e] = make_uint4(0u, 0u, 0u, 0u);
          if (q0 + e < NQ && q < LH4 * LW4 * 4 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
            v[e] = *reinterpret_cast<const uint4*>(inh + ((long long)gy * a.W + gx) * a.Cin + c0 + c8 * 8);
        }
#pragma unroll
        for (int e = 0; e < SB; ++e) {
          const int q = (q0 + e) * 256 + tid;
          if (q0 + e >= NQ || q >= LH4 * LW4 * 4) continue;
          const int pix = q >> 2, c8 = q & 3;
          const int lx = pix % LW4;
          *reinterpret_cast<uint4*>(tb + pix * 64 + ((c8 ^ fsw(lx)) * 16)) = v[e];
        }
      }
    } else {
      const float* in = a.in + (long long)img * a.H * a.W * a.Cin;
      constexpr int SBF = 5;
#pragma unroll
      for (int q0 = 0; q0 < NQ; q0 += SBF) {
        f32x4 v[SBF][2];
#pragma unroll
        for (int e = 0; e < SBF; ++e) {
          const int q = (q0 + e) * 256 + tid;
          const int pix = q >> 2, c8 = q & 3;
          const int ly = pix / LW4, lx = pix - ly * LW4;
          const int gy = y0 + ly - 1, gx = x0 + lx - 1;
          const f32x4 z = {0.f, 0.f, 0.f, 0.f};
          v[e][0] = z; v[e][1] = z;
          if (q0 + e < NQ && q < LH4 * LW4 * 4 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
            const float* p = in + ((long long)gy * a.W + gx) * a.Cin + c0 + c8 * 8;
            v[e][0] = *reinterpret_cast<const f32x4*>(p); v[e][1] = *reinterpret_cast<const f32x4*>(p + 4);
          }
        }
#pragma unroll
        for (int e = 0; e < SBF; ++e) {
          const int q = (q0 + e) * 256 + tid;
          if (q0 + e >= NQ || q >= LH4 * LW4 * 4) continue;
          const int pix = q >> 2, c8 = q & 3;
          const int lx = pix % LW4;
          ovf_track(amax, v[e][0].x, v[e][0].y); ovf_track(amax, v[e][0].z, v[e][0].w);
          ovf_track(amax, v[e][1].x, v[e][1].y); ovf_track(amax, v[e][1].z, v[e][1].w);
          const h16x4 lo = __builtin_convertvector(v[e][0], h16x4), hi = __builtin_convertvector(v[e][1], h16x4);
          *reinterpret_cast<h16x4*>(tb + pix * 64 + ((c8 ^ fsw(lx)) * 16)) = lo;
          *reinterpret_cast<h16x4*>(tb + pix * 64 + ((c8 ^ fsw(lx)) * 16) + 8) = hi;
        }
      }
    }
    __syncthreads();
    // weight fragments (high terms) of the three taps of kernel column dx: block tid >> 6 = i * 2 + k-step of tap dy * 3 + dx
    auto wsrc = [&](int dy, int dx) __attribute__((always_inline)) {
      const int b = tid >> 6, i = b >> 1, ks = b & 1;
      return wfh[(size_t)((((2 * og + i) * 9 + (dy * 3 + dx)) * (a.Cin / 16) + (c0 / 16 + ks)) * 2) * 64];
    };
    uint4 wn[3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) wn[dy] = wsrc(dy, 0);
#pragma unroll 1
    for (int dx = 0; dx < 3; ++dx) {
      unsigned char* const wb = wbuf + (dx & 1) * 12288;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) *reinterpret_cast<uint4*>(wb + dy * 4096 + (tid >> 6) * 1024 + lane * 16) = wn[dy];
      if (dx + 1 < 3) {
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) wn[dy] = wsrc(dy, dx + 1);
      }
      __syncthreads();     // (the buffer written two columns ago is no longer read: one barrier in between)
      const int lx = ql + dx;
#pragma unroll
      for (int s = 0; s < CH / 16; ++s) {
        h16x8 fa[3][2];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int i = 0; i < 2; ++i) fa[dy][i] = *reinterpret_cast<const h16x8*>(wb + dy * 4096 + (i * 2 + s) * 1024 + lane * 16);
#pragma unroll
        for (int r = 0; r < RPW4 + 2; ++r) {
          const h16x8 fb = *reinterpret_cast<const h16x8*>(tb + ((RPW4 * wave + r) * LW4 + lx) * 64 + (((2 * s + hh) ^ fsw(lx)) * 16));
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            const int j = r - dy;
            if (j >= 0 && j < RPW4) {
#pragma unroll
              for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[dy][i], fb, acc[i][j], 0, 0, 0);
            }
          }
        }
      }
    }
  }
  // epilogue: lane = pixel (row RPW4 wave + j, column ql); registers 4 g + c = output channels 32 i + 8 g + 4 hh + c
  const int gx = x0 + ql;
  const float ascale = a.acc_scale;
  if (POOL) {
#pragma unroll
    for (int jp = 0; jp < RPW4 / 2; ++jp) {
      const int gy = y0 + RPW4 * wave + 2 * jp;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = 64 * og + 32 * i + 8 * g + 4 * hh;
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias + c);
          f32x4 m;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v0 = acc[i][2 * jp][4 * g + e] * ascale + b4[e], v1 = acc[i][2 * jp + 1][4 * g + e] * ascale + b4[e];
            if (a.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
            float mv = fmaxf(v0, v1);
            mv = fmaxf(mv, __shfl_xor(mv, 1));
            m[e] = mv;
          }
          const bool wr = !(ql & 1) && gy + 1 < a.H && gx + 1 < a.W;
          const long long o = (long long)img * (a.H / 2) * (a.W / 2) * a.Cout + ((long long)(gy >> 1) * (a.W / 2) + (gx >> 1)) * a.Cout + c;
          if (a.out_half) {
            ovf_track(amax, m.x, m.y); ovf_track(amax, m.z, m.w);
            if (wr) *reinterpret_cast<h16x4*>(reinterpret_cast<uint16_t*>(a.out) + o) = __builtin_convertvector(m, h16x4);
          } else if (wr) {
            *reinterpret_cast<f32x4*>(a.out + o) = m;
          }
        }
    }
    ovf_commit(a.ovf, amax);
    return;
  }
#pragma unroll
  for (int j = 0; j < RPW4; ++j) {
    const int gy = y0 + RPW4 * wave + j;
    if (gy >= a.H || gx >= a.W) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = 64 * og + 32 * i + 8 * g + 4 * hh;
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias + c);
        f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        v = v * ascale + b4;
        if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        const long long o = (long long)img * a.H * a.W * a.Cout + ((long long)gy * a.W + gx) * a.Cout + c;
        if (a.out_half) {
          ovf_track(amax, v.x, v.y); ovf_track(amax, v.z, v.w);
          *reinterpret_cast<h16x4*>(reinterpret_cast<uint16_t*>(a.out) + o) = __builtin_convertvector(v, h16x4);
        } else {
          *reinterpret_cast<f32x4*>(a.out + o) = v;
        }
      }
  }
  ovf_commit(a.ovf, amax);
}

// Split-fp16 convolution on hm16 activations (round 5).  What the counters said about k_sp_conv<., 1, ...> (tools/sp_pmc.sh): matrix pipe busy
// 0.47-0.50 on every layer, 4.5 VALU instructions per MFMA -- the split of the f32 halo tile into fp16 terms while it is staged (~35
// instructions per 16-byte chunk, once per channel slice, per output-channel group and per halo overlap) and the fragment addresses of a
// run-time tap (dy, dx), none of it in the shadow of an MFMA (a wave's VALU work overlaps the matrix pipe only inside its own instruction
// stream).  Here the activations TRAVEL as hm16 records -- per pixel and 16 channels: 16 high terms, 16 residual terms (64 B; the same 4 bytes
// per value as f32, and exactly the terms the staging pass computed: h = fp16(v), m = fp16(v - h)) -- written once by the producer's
// epilogue, so that
//  * the halo tile of a 32-channel slice is the pixel records' bytes as they are: staged by LDS-DMA (`buffer_load_dwordx4 ... lds`: 1 KB = 8
//    pixels per instruction and wave, no register round trip, no VALU; the chunk swizzle is applied on the global side, pixels outside the
//    image are out-of-range offsets of the buffer descriptor and arrive as zeros: tools/probes/buf_lds.hip);
//  * the weight fragments of a tap (8 KB) are handed over the same way, one tap ahead;
//  * the nine taps are unrolled: every fragment read is `one of 12 base registers + immediate`;
//  * the fp16-range guard moves to the producer (the values it checked are the producer's outputs).
// Arithmetic: identical to k_sp_conv<., 1, ...> product for product (same terms, same order of the three products and of the k-steps), so the
// results are bitwise the same.  Tiles, accumulator layout and the fused 2 x 2 max-pool are k_sp_conv's.
//
// What the phase stamps of the first version said (tools/sp_phases.py; two workgroups per CU): with the co-resident workgroup in its MFMA
// stream, a VALU instruction of this one issues about once per 16-20 cycles (the dense MFMA stream of the other wave holds the SIMD's VALU
// port), so the ~420 VALU instructions of the per-lane staging offsets cost 6.8 k cycles and the ~900 of the epilogue 26 k, of a workgroup's
// 72 k -- against 21 k cycles of MFMAs.  Hence the VALU diet: a staging instruction covers 8 pixels of ONE halo row (row validity and row
// offset are scalar: SALU; five per-lane offset registers per tile; the two-pixel tail of a 34-pixel row is an EXEC-masked instruction), the
// epilogue is 16 instructions per four values (packed fma, max3 for the range guard, v_fma_mix for the residual term), the rows leave as
// whole 256-byte record segments through a per-wave LDS slab and buffer stores whose descriptor ends at the image border.
typedef __attribute__((address_space(3))) void* sp_lptr_t;
// Two wait states behind a 16-byte buffer store with an SGPR soffset, before anything may rewrite its data registers (hipcc's hazard recogniser
// pads that case only for an immediate soffset; cheap insurance: the stores are the last thing a row does).
// acc * scale + bias as ONE scalar v_fma_f32 per value, pinned so that nothing re-packs it: written on two-wide vectors hipcc emits
// `v_pk_fma_f32 D, V, s[n:n+1], V op_sel_hi:[1,0,1]` (the scale in an SGPR pair), and that instruction intermittently returned garbage for one
// quad of lanes (20..23 / 28..31) in this epilogue -- ReLU turned it into zeros: one word of four pixels in about one tile per 10^4, found with
// tools/sp_race.py.  The same class as gn_ffn128.hip's note on v_pk_mul_f32 with an SGPR scale and build.py's on k_qkv's v_pk_fma_f32.
__device__ __forceinline__ float sp_scale_bias(float a, float s, float b) { float r = __builtin_fmaf(a, s, b); asm volatile("" : "+v"(r)); return r; }
__device__ __forceinline__ void sp_store_guard() { asm volatile("s_nop 1" ::: "memory"); }
__device__ __forceinline__ float sp_resid_lo(unsigned int h, float y) { float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(y)); return r; }
__device__ __forceinline__ float sp_resid_hi(unsigned int h, float y) { float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(y)); return r; }
template <int TAPS, int RPW, bool POOL, bool OUTHM>   // OUTHM: the output leaves as hm16 records (else f32: the two head outputs)
__global__ __launch_bounds__(256, 2) void k_sp_conv_s(ConvArgs a) {
  constexpr int TH = 4 * RPW;
  constexpr int HALO = TAPS == 9 ? 1 : 0;
  constexpr int LW = TW + 2 * HALO, LH = TH + 2 * HALO;
  constexpr int NSEG = (LW + 7) / 8;                // staging instructions per halo row (8 pixels = 1 KB each)
  constexpr int TAIL = LW - 8 * (NSEG - 1);         // pixels of the last one
  constexpr int NROW = (LH + 3) / 4;                // halo rows per wave (row q * 4 + wave)
  constexpr int TILE_B = ((LH * LW * 128 + 1023) / 1024) * 1024;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[TILE_B + 2 * 8192];
  unsigned char* const tb = smem;
  unsigned char* const wbuf = smem + TILE_B;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, ql = lane & 31;
  const int ogroups = a.Cout / 64;
  const int img = blockIdx.z / ogroups, og = blockIdx.z % ogroups;
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
  const unsigned int rec = (unsigned int)a.Cin * 4u;                       // bytes of a pixel record
  unsigned char* const ibase = const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(a.in)) + (size_t)img * a.H * a.W * rec;
  const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc(ibase, 0, (unsigned int)((size_t)a.H * a.W * rec), 0x00020000);
  const __amdgpu_buffer_rsrc_t irs0 = __builtin_amdgcn_make_buffer_rsrc(ibase, 0, 0, 0x00020000);      // rows outside the image: every lane out of range = zeros
  // weights: block (((tile * TAPS + tap) * (Cin / 16) + kstep) * 2 + term) of 1 KB; this workgroup's tiles are 2 og, 2 og + 1
  const unsigned int ksteps = (unsigned int)a.Cin / 16u;
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(a.wfh) + (size_t)(2 * og) * TAPS * ksteps * 1024, 0, 2u * TAPS * ksteps * 2048u, 0x00020000);
  const unsigned int lds_tile = (unsigned int)(size_t)(sp_lptr_t)tb, lds_w = (unsigned int)(size_t)(sp_lptr_t)wbuf;
  const unsigned int lane16 = (unsigned int)lane * 16u;
  const int wg_lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  long long* const ts = (a.dbg_ts != nullptr && wg_lin < 8192 && tid == 0) ? a.dbg_ts + (size_t)wg_lin * 32 : nullptr;
  auto stamp = [&](int k) __attribute__((always_inline)) { if (ts != nullptr && k < 32) ts[k] = (long long)__builtin_amdgcn_s_memtime(); };
  stamp(0);
  // the bias of the lane's 32 output channels, requested first (read in the epilogue)
  f32x4 b4[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) b4[i][g] = *reinterpret_cast<const f32x4*>(a.bias + 64 * og + 32 * i + 8 * g + 4 * hh);

  // staging: instruction (row r, segment sg) covers halo pixels lx = 8 sg .. 8 sg + 7 of row r: lane -> (pixel lane >> 3, position lane & 7);
  // the piece that belongs at position pos of pixel lx is pos ^ psw(lx).  Per-lane part of the offset (column + piece), one register per segment;
  // the row part (and the channel slice) is the instruction's scalar offset
  unsigned int lanepart[NSEG];
#pragma unroll
  for (int sg = 0; sg < NSEG; ++sg) {
    const int lx = 8 * sg + (lane >> 3), gx = x0 + lx - HALO;
    lanepart[sg] = (gx >= 0 && gx < a.W) ? (unsigned int)gx * rec + (unsigned int)(((lane & 7) ^ psw(lx)) * 16) : 0x80000000u;
  }
  // the wave's two 1 KB weight blocks of a tap: block 4 e + wave = (i * 2 + s) * 2 + pl  ->  source block ((i * TAPS + tap) * ksteps + c0 / 16 + s) * 2 + pl
  auto weights_dma = [&](int tap, int c0, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int blk = 4 * e + wave, i = blk >> 2, ks = (blk >> 1) & 1, pl = blk & 1;
      const unsigned int src = ((((unsigned int)i * TAPS + tap) * ksteps + (unsigned int)(c0 / 16 + ks)) * 2u + pl) * 1024u;
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" ::"v"(lane16), "s"(wrs), "s"(lds_w + (unsigned int)(buf * 8192 + blk * 1024)), "s"(src) : "memory");
    }
  };
  auto dma_wait = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

  auto stage_issue = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < NROW; ++q) {
      const int r = q * 4 + wave;
      if (r < LH) {
        const int gy = y0 + r - HALO;
        const bool rv = gy >= 0 && gy < a.H;
        const __amdgpu_buffer_rsrc_t rs = rv ? irs : irs0;
        const unsigned int soff = rv ? (unsigned int)(gy * a.W) * rec + (unsigned int)(c0 * 4) : 0u;
        const unsigned int dst = lds_tile + (unsigned int)(r * LW * 128);
#pragma unroll
        for (int sg = 0; sg < NSEG; ++sg) {
          if (sg + 1 < NSEG || TAIL == 8)
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" ::"v"(lanepart[sg]), "s"(rs), "s"(dst + (unsigned int)(sg * 1024)), "s"(soff) : "memory");
          else     // the row's last TAIL pixels: the other lanes must not write (their slots are the next row's first pixels)
            asm volatile("s_mov_b32 m0, %2\n\ts_mov_b64 exec, %4\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds\n\ts_mov_b64 exec, -1" ::"v"(lanepart[sg]), "s"(rs), "s"(dst + (unsigned int)(sg * 1024)), "s"(soff), "n"((1ull << (8 * TAIL)) - 1ull) : "memory");
        }
      }
    }
    weights_dma(0, c0, 0);
  };
  stage_issue(0);
  stamp(1);
  // (behind the first requests, in the shadow of their latency)
  // fragment addresses: pixel (RPW wave + j + dy, ql + dx), piece 4 s + 2 pl + hh -> base[dx][2 s + pl] + (j + dy) * LW * 128
  unsigned int fbase[TAPS == 9 ? 3 : 1][4];
#pragma unroll
  for (int dx = 0; dx < (TAPS == 9 ? 3 : 1); ++dx)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int lx = ql + dx;
      fbase[dx][c] = (unsigned int)((RPW * wave * LW + lx) * 128 + (((2 * c + hh) ^ psw(lx)) * 16));
    }

  f32x16 acc[2][RPW];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < RPW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  __builtin_amdgcn_sched_barrier(0);
  for (int c0 = 0; c0 < a.Cin; c0 += CH) {
    if (c0 > 0) {
      __syncthreads();      // every wave is done with the previous slice's tile and weight buffers
      stage_issue(c0);
      stamp(12);
    }
    dma_wait();
    __syncthreads();
    stamp(c0 == 0 ? 2 : 13);
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const int dy = TAPS == 9 ? tap / 3 : 0, dx = TAPS == 9 ? tap % 3 : 0;
      if (tap + 1 < TAPS) weights_dma(tap + 1, c0, (tap + 1) & 1);      // (that buffer's last readers passed the barrier in front of this tap)
      const unsigned char* const wb = wbuf + (tap & 1) * 8192;
#pragma unroll
      for (int s = 0; s < CH / 16; ++s) {
        h16x8 fa[2][2], fb[RPW][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) fa[i][pl] = *reinterpret_cast<const h16x8*>(wb + ((i * 2 + s) * 2 + pl) * 1024 + lane16);
#pragma unroll
        for (int j = 0; j < RPW; ++j)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) fb[j][pl] = *reinterpret_cast<const h16x8*>(tb + fbase[dx][2 * s + pl] + (j + dy) * LW * 128);
        // products: W_m X_h, W_h X_m, W_h X_h (small terms first) -- the order of k_sp_conv
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < RPW; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][p == 0 ? 1 : 0], fb[j][p == 1 ? 1 : 0], acc[i][j], 0, 0, 0);
      }
      if (tap + 1 < TAPS) { dma_wait(); __syncthreads(); }
      stamp((c0 == 0 ? 3 : 14) + tap);       // slices beyond the second overwrite the second's stamps
    }
  }
  stamp(24);

  // epilogue: lane = pixel (row RPW wave + j, column ql); registers 4 g + c = output channels 32 i + 8 g + 4 hh + c.  hm16 output: channel
  // 32 i + 8 g + 4 hh + c of the workgroup's 64 sits in record group 2 i + (g >> 1) of the 256-byte segment, halfs 8 (g & 1) + 4 hh + c (high) and
  // 16 + that (residual).  Final bytes go into the wave's LDS slab (the tile is dead), 16-byte chunks come out: lane -> (pixel it * 4 + lane >> 4,
  // chunk lane & 15), one store instruction = four whole record segments; the store descriptor ends where the image row does.
  const float ascale = a.acc_scale;
  float amax = 0.f;
  constexpr bool out_hm = OUTHM;
  const float relu_lo = a.relu ? 0.f : -INFINITY;     // ReLU as one v_max_f32 per value, whatever the flag
  constexpr int PSTR = 256 + 16;                      // slab bytes per pixel (the pad spreads the lanes' pieces over the banks)
  constexpr int SLAB = 32 * PSTR;
  static_assert(4 * SLAB <= TILE_B, "the slabs alias the halo tile");
  unsigned char* const slab = smem + wave * SLAB;
  __syncthreads();                                    // every wave is done with the tile and the weight buffers
  stamp(26);
  auto put4 = [&](int p, int i, int g, f32x2v v0, f32x2v v1) __attribute__((always_inline)) {      // v0, v1: scaled + biased (+ ReLU'd) values of channels c .. c + 3
    const int cl = 32 * i + 8 * g + 4 * hh;
    if (out_hm) {
      amax = __builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fabsf(v0[0]), __builtin_fabsf(v0[1])));
      amax = __builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fabsf(v1[0]), __builtin_fabsf(v1[1])));
      const unsigned int h0 = pack16<true>(v0[0], v0[1]), h1 = pack16<true>(v1[0], v1[1]);
      const unsigned int m0 = pack16<true>(sp_resid_lo(h0, v0[0]), sp_resid_hi(h0, v0[1])), m1 = pack16<true>(sp_resid_lo(h1, v1[0]), sp_resid_hi(h1, v1[1]));
      unsigned char* o = slab + p * PSTR + (cl >> 4) * 64 + (cl & 15) * 2;
      *reinterpret_cast<uint2*>(o) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(o + 32) = make_uint2(m0, m1);
    } else {
      *reinterpret_cast<f32x4*>(slab + p * PSTR + cl * 4) = (f32x4){v0[0], v0[1], v1[0], v1[1]};
    }
  };
  const size_t orec = (size_t)a.Cout * 4;             // bytes of an output pixel record (f32 and hm16 alike)
  unsigned char* const obase = reinterpret_cast<unsigned char*>(a.out) + (size_t)og * 256;
  const unsigned int rd = (unsigned int)((lane >> 4) * PSTR + (lane & 15) * 16);                     // read-back address inside the slab
  const unsigned int so = (unsigned int)(lane >> 4) * (unsigned int)orec + (unsigned int)(lane & 15) * 16u;   // ... and inside the output row
  if (POOL) {
    static_assert(!POOL || RPW == 2, "the fused pool pairs the two rows of a wave");
    const int gy = y0 + RPW * wave;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x2v m[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const f32x2v bk = {b4[i][g][2 * k], b4[i][g][2 * k + 1]};
          // (ascale is a power of two: the fused multiply-add rounds like the multiplication followed by the addition)
          const f32x2v v0 = {sp_scale_bias(acc[i][0][4 * g + 2 * k], ascale, bk[0]), sp_scale_bias(acc[i][0][4 * g + 2 * k + 1], ascale, bk[1])};
          const f32x2v v1 = {sp_scale_bias(acc[i][RPW - 1][4 * g + 2 * k], ascale, bk[0]), sp_scale_bias(acc[i][RPW - 1][4 * g + 2 * k + 1], ascale, bk[1])};
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            float mv = fmaxf(fmaxf(v0[e], v1[e]), relu_lo);      // max(relu(a), relu(b)) = relu(max(a, b))
            mv = fmaxf(mv, __shfl_xor(mv, 1));
            m[k][e] = mv;
          }
        }
        if (!(ql & 1)) put4(ql >> 1, i, g, m[0], m[1]);
      }
    if (out_hm) ovf_commit(a.ovf, amax);
    const size_t prow = ((size_t)img * (a.H / 2) + (gy >> 1)) * (a.W / 2) + (x0 >> 1);
    const int npix = gy + 1 < a.H ? min(16, (a.W - x0) / 2) : 0;
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(obase + prow * orec, 0, npix > 0 ? (unsigned int)((npix - 1) * orec + 256) : 0u, 0x00020000);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const u32x4_sp v = *reinterpret_cast<const u32x4_sp*>(slab + rd + it * 4 * PSTR);
      __builtin_amdgcn_raw_buffer_store_b128(v, ors, so, (unsigned int)(it * 4) * (unsigned int)orec, 0);
      sp_store_guard();
    }
    stamp(25);
    return;
  }
#pragma unroll
  for (int j = 0; j < RPW; ++j) {
    const int gy = y0 + RPW * wave + j;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x2v v[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const f32x2v bk = {b4[i][g][2 * k], b4[i][g][2 * k + 1]};
          // (ascale is a power of two: the fused multiply-add rounds like multiplication, then addition)
          v[k][0] = fmaxf(sp_scale_bias(acc[i][j][4 * g + 2 * k], ascale, bk[0]), relu_lo);
          v[k][1] = fmaxf(sp_scale_bias(acc[i][j][4 * g + 2 * k + 1], ascale, bk[1]), relu_lo);
        }
        put4(ql, i, g, v[0], v[1]);
      }
    const size_t prow = ((size_t)img * a.H + gy) * a.W + x0;
    const int npix = gy < a.H ? min(32, a.W - x0) : 0;
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(obase + prow * orec, 0, npix > 0 ? (unsigned int)((npix - 1) * orec + 256) : 0u, 0x00020000);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const u32x4_sp v = *reinterpret_cast<const u32x4_sp*>(slab + rd + it * 4 * PSTR);
      __builtin_amdgcn_raw_buffer_store_b128(v, ors, so, (unsigned int)(it * 4) * (unsigned int)orec, 0);
      sp_store_guard();
    }
  }
  if (out_hm) ovf_commit(a.ovf, amax);
  stamp(25);
}

// The same convolution with 16-channel slices (3 x 3 layers): the halo tile of an 8 x 32 output tile is 22 KB and a tap's weight fragments 4 KB --
// 35 KB per workgroup, accumulators for two rows per wave -> up to four workgroups (16 waves) per CU, so that a staging workgroup leaves others on
// the matrix pipe (with two per CU the full-resolution layer ran at HBM time PLUS MFMA time).  The weight fragments run two taps ahead through
// three 4 KB buffers, across the slice boundaries (they do not depend on the tile), and are waited for with a counted vmcnt.  LDS pixel layout:
// 64 bytes = pieces 2 pl + hh at position piece ^ ((lx >> 2) & 3) (k_sp_conv_h's).  Same products, but the k order is (slice of 16, tap) instead
// of (slice of 32, tap, k-step): results equal k_sp_conv_s's to f32 rounding, not bitwise.
// FUSE1 (round 6): this is the network's SECOND convolution and the first one (1 -> 64 channels, 3 x 3, ReLU) is not a launch of its own
// any more: its 64-channel full-resolution map -- 0.53 GB of hm16 records per 1080p frame, written by k_sp_conv1 and read straight back here -- never
// exists.  A slice's halo tile (10 x 34 pixels x 16 channels) is COMPUTED into LDS from a 12 x 36 patch of the gray image: per pixel and channel the fma
// chain of k_sp_conv1 (bias, then the nine taps in order), ReLU, the fp16 split of its record format -- the same bits in the same LDS positions the
// LDS-DMA would have delivered, so everything downstream is bitwise what the two launches give.  Pixels outside the image are zero records (this
// convolution's zero padding), not first-layer values.
template <bool POOL, bool OUTHM, bool FUSE1 = false>
__global__ __launch_bounds__(256, 3) void k_sp_conv_s16(ConvArgs a) {
  constexpr int RPW = 2, TH = 8, LW = TW + 2, LH = TH + 2, TAPS = 9;
  constexpr int NSEG = 3, TAIL = 2;                 // 16 pixels per staging instruction; 34 = 16 + 16 + 2
  constexpr int NROW = (LH + 3) / 4;
  constexpr int TILE_B = ((LH * LW * 64 + 1023) / 1024) * 1024;
  constexpr int PSTR = 256 + 16, SLAB = (POOL ? 16 : 32) * PSTR;
  constexpr int PW = LW + 2, PH = LH + 2;           // FUSE1: image patch (the halo tile's own 3 x 3 neighbourhood)
  constexpr int FUSE_B = FUSE1 ? (PH * PW + 64 * 9 + 64) * 4 : 0;
  constexpr int SMEM0 = TILE_B + 3 * 4096 > 4 * SLAB ? TILE_B + 3 * 4096 : 4 * SLAB;
  constexpr int SMEM = SMEM0 + ((FUSE_B + 15) / 16) * 16;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM];
  float* const patch = reinterpret_cast<float*>(smem + SMEM0);      // [PH][PW], origin (y0 - 2, x0 - 2), zeros outside the image
  float* const ws1 = patch + PH * PW;                               // [64 channels][9 taps], then [64] biases
  unsigned char* const tb = smem;
  unsigned char* const wbuf = smem + TILE_B;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, ql = lane & 31;
  const int ogroups = a.Cout / 64;
  const int img = blockIdx.z / ogroups, og = blockIdx.z % ogroups;
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
  const unsigned int rec = (unsigned int)a.Cin * 4u;
  unsigned char* const ibase = const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(a.in)) + (size_t)img * a.H * a.W * rec;
  const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc(ibase, 0, (unsigned int)((size_t)a.H * a.W * rec), 0x00020000);
  const __amdgpu_buffer_rsrc_t irs0 = __builtin_amdgcn_make_buffer_rsrc(ibase, 0, 0, 0x00020000);
  const unsigned int ksteps = (unsigned int)a.Cin / 16u;
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(a.wfh) + (size_t)(2 * og) * TAPS * ksteps * 1024, 0, 2u * TAPS * ksteps * 2048u, 0x00020000);
  const unsigned int lds_tile = (unsigned int)(size_t)(sp_lptr_t)tb, lds_w = (unsigned int)(size_t)(sp_lptr_t)wbuf;
  const unsigned int lane16 = (unsigned int)lane * 16u;
  auto fsw = [](int lx) { return (lx >> 2) & 3; };
  // staging: instruction (row r, segment sg) covers halo pixels 16 sg .. 16 sg + 15 of row r: lane -> (pixel lane >> 2, position lane & 3)
  unsigned int lanepart[NSEG];
#pragma unroll
  for (int sg = 0; sg < NSEG; ++sg) {
    const int lx = 16 * sg + (lane >> 2), gx = x0 + lx - 1;
    lanepart[sg] = (gx >= 0 && gx < a.W) ? (unsigned int)gx * rec + (unsigned int)(((lane & 3) ^ fsw(lx)) * 16) : 0x80000000u;
  }
  // the wave's 1 KB weight block of a (tap, slice): block wave = i * 2 + pl  ->  source block ((i * TAPS + tap) * ksteps + c0 / 16) * 2 + pl
  const unsigned int wsrc0 = (((unsigned int)(wave >> 1) * TAPS) * ksteps * 2u + (unsigned int)(wave & 1)) * 1024u;
  auto weights_dma = [&](int tap, int c0, int buf) __attribute__((always_inline)) {
    const unsigned int src = wsrc0 + ((unsigned int)tap * ksteps + (unsigned int)(c0 >> 4)) * 2048u;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" ::"v"(lane16), "s"(wrs), "s"(lds_w + (unsigned int)(buf * 4096) + (unsigned int)wave * 1024u), "s"(src) : "memory");
  };
  auto stage_issue = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < NROW; ++q) {
      const int r = q * 4 + wave;
      if (r < LH) {
        const int gy = y0 + r - 1;
        const bool rv = gy >= 0 && gy < a.H;
        const __amdgpu_buffer_rsrc_t rs = rv ? irs : irs0;
        const unsigned int soff = rv ? (unsigned int)(gy * a.W) * rec + (unsigned int)(c0 * 4) : 0u;
        const unsigned int dst = lds_tile + (unsigned int)(r * LW * 64);
#pragma unroll
        for (int sg = 0; sg < NSEG; ++sg) {
          if (sg + 1 < NSEG)
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" ::"v"(lanepart[sg]), "s"(rs), "s"(dst + (unsigned int)(sg * 1024)), "s"(soff) : "memory");
          else     // the row's last two pixels: the other lanes must not write (their slots are the next row's first pixels)
            asm volatile("s_mov_b32 m0, %2\n\ts_mov_b64 exec, %4\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds\n\ts_mov_b64 exec, -1" ::"v"(lanepart[sg]), "s"(rs), "s"(dst + (unsigned int)(sg * 1024)), "s"(soff), "n"((1ull << (4 * TAIL)) - 1ull) : "memory");
        }
      }
    }
  };
  float amax1 = 0.f;
  // FUSE1: the halo tile of slice c0 computed from the image patch.  Thread -> the 8-channel half (tid & 1) of pixels (tid >> 1) + 128 k: the half's 72
  // weights and 8 biases are read once per slice and serve the thread's three pixels
  auto stage_compute = [&](int c0) __attribute__((always_inline)) {
    const int half = tid & 1;
    const float* const wsl = ws1 + (c0 + 8 * half) * 9;       // the half's 8 x 9 weights (contiguous), read four channels at a time: 36 floats = 9 x 16 bytes
    const float* const bsl = ws1 + 576 + c0 + 8 * half;
#pragma unroll 1
    for (int k = 0; k < 3; ++k) {
      const int p = (tid >> 1) + 128 * k;
      if (p < LH * LW) {
        const int r = p / LW, lx = p - r * LW;
        const int gy = y0 + r - 1, gx = x0 + lx - 1;
        unsigned char* const dst = tb + (r * LW + lx) * 64;
        const int ph = ((0 + half) ^ fsw(lx)) * 16, pm = ((2 + half) ^ fsw(lx)) * 16;
        if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
          float v[9];
#pragma unroll
          for (int t = 0; t < 9; ++t) v[t] = patch[(r + t / 3) * PW + lx + t % 3];
          h16x8 hv, mv;
#pragma unroll
          for (int g4 = 0; g4 < 2; ++g4) {
            f32x4 w4[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) w4[q] = *reinterpret_cast<const f32x4*>(wsl + 36 * g4 + 4 * q);
            const f32x4 b4v = *reinterpret_cast<const f32x4*>(bsl + 4 * g4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float acc1 = b4v[e];
#pragma unroll
              for (int t = 0; t < 9; ++t) acc1 = fmaf(w4[(9 * e + t) >> 2][(9 * e + t) & 3], v[t], acc1);
              const float f = fmaxf(acc1, 0.f);
              amax1 = fmaxf(amax1, f);
              hv[4 * g4 + e] = (_Float16)f;
              mv[4 * g4 + e] = (_Float16)(f - (float)hv[4 * g4 + e]);
            }
          }
          *reinterpret_cast<h16x8*>(dst + ph) = hv;
          *reinterpret_cast<h16x8*>(dst + pm) = mv;
        } else {
          const uint4 z = make_uint4(0u, 0u, 0u, 0u);
          *reinterpret_cast<uint4*>(dst + ph) = z;
          *reinterpret_cast<uint4*>(dst + pm) = z;
        }
      }
    }
  };
  if constexpr (FUSE1) {
    const float* const im = a.img + (size_t)img * a.H * a.W;
    for (int q = tid; q < PH * PW; q += 256) {
      const int py = q / PW, px = q - py * PW, gy = y0 - 2 + py, gx = x0 - 2 + px;
      patch[q] = (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? im[(size_t)gy * a.W + gx] : 0.f;
    }
    for (int q = tid; q < 64 * 9 + 64; q += 256) ws1[q] = q < 576 ? a.w1[q] : a.b1[q - 576];
    __syncthreads();
    stage_compute(0);
  } else {
    stage_issue(0);
  }
  weights_dma(0, 0, 0);
  weights_dma(1, 0, 1);
  // (behind the first requests, in the shadow of their latency)
  // fragment addresses: pixel (RPW wave + j + dy, ql + dx), piece 2 pl + hh -> fbase[dx][pl] + (j + dy) * LW * 64
  unsigned int fbase[3][2];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      const int lx = ql + dx;
      fbase[dx][pl] = (unsigned int)((RPW * wave * LW + lx) * 64 + (((2 * pl + hh) ^ fsw(lx)) * 16));
    }
  f32x16 acc[2][RPW];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < RPW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int c0 = 0; c0 < a.Cin; c0 += 16) {
    const bool more = c0 + 16 < a.Cin;
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const int dy = tap / 3, dx = tap % 3;
      // weights two taps ahead, into the buffer tap - 1 has released (its readers passed the barrier in front of this tap)
      const bool issue = tap + 2 < TAPS || more;
      if (tap + 2 < TAPS) weights_dma(tap + 2, c0, (tap + 2) % 3);
      else if (more) weights_dma(tap + 2 - TAPS, c0 + 16, (tap + 2) % 3);
      const unsigned char* const wb = wbuf + (tap % 3) * 4096;
      h16x8 fa[2][2], fb[RPW][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) fa[i][pl] = *reinterpret_cast<const h16x8*>(wb + (i * 2 + pl) * 1024 + lane16);
#pragma unroll
      for (int j = 0; j < RPW; ++j)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) fb[j][pl] = *reinterpret_cast<const h16x8*>(tb + fbase[dx][pl] + (j + dy) * LW * 64);
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < RPW; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][p == 0 ? 1 : 0], fb[j][p == 1 ? 1 : 0], acc[i][j], 0, 0, 0);
      if (tap + 1 < TAPS) {
        // the next tap's weights (requested a tap ago) must have landed; the request of this tap may stay in flight
        if (issue) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
    }
    if (more) {
      __syncthreads();      // every wave is done with this slice's tile
      if constexpr (FUSE1) stage_compute(c0 + 16); else stage_issue(c0 + 16);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }

  // epilogue: k_sp_conv_s's (bias requested here: no request of this wave is in flight any more)
  f32x4 b4[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) b4[i][g] = *reinterpret_cast<const f32x4*>(a.bias + 64 * og + 32 * i + 8 * g + 4 * hh);
  const float ascale = a.acc_scale;
  float amax = FUSE1 ? amax1 : 0.f;                    // (the first layer's values were rounded to fp16 too: they are in the same guard)
  constexpr bool out_hm = OUTHM;
  const float relu_lo = a.relu ? 0.f : -INFINITY;
  unsigned char* const slab = smem + wave * SLAB;
  __syncthreads();                                    // every wave is done with the tile and the weight buffers
  auto put4 = [&](int p, int i, int g, f32x2v v0, f32x2v v1) __attribute__((always_inline)) {
    const int cl = 32 * i + 8 * g + 4 * hh;
    if (out_hm) {
      amax = __builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fabsf(v0[0]), __builtin_fabsf(v0[1])));
      amax = __builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fabsf(v1[0]), __builtin_fabsf(v1[1])));
      const unsigned int h0 = pack16<true>(v0[0], v0[1]), h1 = pack16<true>(v1[0], v1[1]);
      const unsigned int m0 = pack16<true>(sp_resid_lo(h0, v0[0]), sp_resid_hi(h0, v0[1])), m1 = pack16<true>(sp_resid_lo(h1, v1[0]), sp_resid_hi(h1, v1[1]));
      unsigned char* o = slab + p * PSTR + (cl >> 4) * 64 + (cl & 15) * 2;
      *reinterpret_cast<uint2*>(o) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(o + 32) = make_uint2(m0, m1);
    } else {
      *reinterpret_cast<f32x4*>(slab + p * PSTR + cl * 4) = (f32x4){v0[0], v0[1], v1[0], v1[1]};
    }
  };
  const size_t orec = (size_t)a.Cout * 4;
  unsigned char* const obase = reinterpret_cast<unsigned char*>(a.out) + (size_t)og * 256;
  const unsigned int rd = (unsigned int)((lane >> 4) * PSTR + (lane & 15) * 16);
  const unsigned int so = (unsigned int)(lane >> 4) * (unsigned int)orec + (unsigned int)(lane & 15) * 16u;
  if (POOL) {
    const int gy = y0 + RPW * wave;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x2v m[2];
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const float bk = b4[i][g][2 * k + e];
            float mv = fmaxf(fmaxf(sp_scale_bias(acc[i][0][4 * g + 2 * k + e], ascale, bk), sp_scale_bias(acc[i][1][4 * g + 2 * k + e], ascale, bk)), relu_lo);
            mv = fmaxf(mv, __shfl_xor(mv, 1));
            m[k][e] = mv;
          }
        if (!(ql & 1)) put4(ql >> 1, i, g, m[0], m[1]);
      }
    if (out_hm) ovf_commit(a.ovf, amax);
    const size_t prow = ((size_t)img * (a.H / 2) + (gy >> 1)) * (a.W / 2) + (x0 >> 1);
    const int npix = gy + 1 < a.H ? min(16, (a.W - x0) / 2) : 0;
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(obase + prow * orec, 0, npix > 0 ? (unsigned int)((npix - 1) * orec + 256) : 0u, 0x00020000);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const u32x4_sp v = *reinterpret_cast<const u32x4_sp*>(slab + rd + it * 4 * PSTR);
      __builtin_amdgcn_raw_buffer_store_b128(v, ors, so, (unsigned int)(it * 4) * (unsigned int)orec, 0);
      sp_store_guard();
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < RPW; ++j) {
    const int gy = y0 + RPW * wave + j;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x2v v[2];
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
          for (int e = 0; e < 2; ++e) v[k][e] = fmaxf(sp_scale_bias(acc[i][j][4 * g + 2 * k + e], ascale, b4[i][g][2 * k + e]), relu_lo);
        put4(ql, i, g, v[0], v[1]);
      }
    const size_t prow = ((size_t)img * a.H + gy) * a.W + x0;
    const int npix = gy < a.H ? min(32, a.W - x0) : 0;
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(obase + prow * orec, 0, npix > 0 ? (unsigned int)((npix - 1) * orec + 256) : 0u, 0x00020000);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const u32x4_sp v = *reinterpret_cast<const u32x4_sp*>(slab + rd + it * 4 * PSTR);
      __builtin_amdgcn_raw_buffer_store_b128(v, ors, so, (unsigned int)(it * 4) * (unsigned int)orec, 0);
      sp_store_guard();
    }
  }
  if (out_hm) ovf_commit(a.ovf, amax);
}

// GN_SP_FP16 (one fp16 product per block, fp16 NHWC activations: the arithmetic BASELINE configs[4] names) in k_sp_conv_s16's form, for the 3 x 3 layers
// whose input already is fp16 (layers 1 .. 6: 85 % of the extractor's flops): a 32-channel slice of an fp16 pixel is the same 64 bytes as a
// 16-channel slice of an hm16 record, so the halo tile, the LDS-DMA staging, the three-buffer weight pipeline (high-term blocks only: k-steps s = 0, 1
// of the slice where k_sp_conv_s16 has the two terms) and the slab epilogue are that kernel's; per tap 8 MFMAs on 8 fragment reads.  Replaces
// k_sp_conv_h there (fragment reuse across the taps of a kernel column, but staging through registers and run-time tap addressing: 0.12 of the
// single-product issue ceiling).  A tolerance mode like k_sp_conv_h (another k order: f32 rounding).
template <bool POOL, bool OUTH>   // OUTH: fp16 NHWC output (else f32)
__global__ __launch_bounds__(256, 3) void k_sp_conv_h16(ConvArgs a) {
  constexpr int RPW = 2, TH = 8, LW = TW + 2, LH = TH + 2, TAPS = 9;
  constexpr int NSEG = 3, TAIL = 2;                 // 16 pixels per staging instruction; 34 = 16 + 16 + 2
  constexpr int NROW = (LH + 3) / 4;
  constexpr int TILE_B = ((LH * LW * 64 + 1023) / 1024) * 1024;
  constexpr int SEG = OUTH ? 128 : 256;               // bytes of the workgroup's 64 channels in an output pixel record
  constexpr int PSTR = SEG + 16, SLAB = (POOL ? 16 : 32) * PSTR;
  constexpr int SMEM = TILE_B + 3 * 4096 > 4 * SLAB ? TILE_B + 3 * 4096 : 4 * SLAB;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM];
  unsigned char* const tb = smem;
  unsigned char* const wbuf = smem + TILE_B;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, ql = lane & 31;
  const int ogroups = a.Cout / 64;
  const int img = blockIdx.z / ogroups, og = blockIdx.z % ogroups;
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
  const unsigned int rec = (unsigned int)a.Cin * 2u;                       // fp16 NHWC: a 32-channel slice of a pixel is the same 64 bytes
  unsigned char* const ibase = const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(a.in)) + (size_t)img * a.H * a.W * rec;
  const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc(ibase, 0, (unsigned int)((size_t)a.H * a.W * rec), 0x00020000);
  const __amdgpu_buffer_rsrc_t irs0 = __builtin_amdgcn_make_buffer_rsrc(ibase, 0, 0, 0x00020000);
  const unsigned int ksteps = (unsigned int)a.Cin / 16u;
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(a.wfh) + (size_t)(2 * og) * TAPS * ksteps * 1024, 0, 2u * TAPS * ksteps * 2048u, 0x00020000);
  const unsigned int lds_tile = (unsigned int)(size_t)(sp_lptr_t)tb, lds_w = (unsigned int)(size_t)(sp_lptr_t)wbuf;
  const unsigned int lane16 = (unsigned int)lane * 16u;
  auto fsw = [](int lx) { return (lx >> 2) & 3; };
  // staging: instruction (row r, segment sg) covers halo pixels 16 sg .. 16 sg + 15 of row r: lane -> (pixel lane >> 2, position lane & 3)
  unsigned int lanepart[NSEG];
#pragma unroll
  for (int sg = 0; sg < NSEG; ++sg) {
    const int lx = 16 * sg + (lane >> 2), gx = x0 + lx - 1;
    lanepart[sg] = (gx >= 0 && gx < a.W) ? (unsigned int)gx * rec + (unsigned int)(((lane & 3) ^ fsw(lx)) * 16) : 0x80000000u;
  }
  // the wave's 1 KB weight block of a (tap, slice): block wave = i * 2 + s  ->  HIGH-term source block ((i * TAPS + tap) * ksteps + c0 / 16 + s) * 2
  const unsigned int wsrc0 = (((unsigned int)(wave >> 1) * TAPS) * ksteps * 2u + 2u * (unsigned int)(wave & 1)) * 1024u;
  auto weights_dma = [&](int tap, int c0, int buf) __attribute__((always_inline)) {
    const unsigned int src = wsrc0 + ((unsigned int)tap * ksteps + (unsigned int)(c0 >> 4)) * 2048u;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" ::"v"(lane16), "s"(wrs), "s"(lds_w + (unsigned int)(buf * 4096) + (unsigned int)wave * 1024u), "s"(src) : "memory");
  };
  auto stage_issue = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < NROW; ++q) {
      const int r = q * 4 + wave;
      if (r < LH) {
        const int gy = y0 + r - 1;
        const bool rv = gy >= 0 && gy < a.H;
        const __amdgpu_buffer_rsrc_t rs = rv ? irs : irs0;
        const unsigned int soff = rv ? (unsigned int)(gy * a.W) * rec + (unsigned int)(c0 * 2) : 0u;
        const unsigned int dst = lds_tile + (unsigned int)(r * LW * 64);
#pragma unroll
        for (int sg = 0; sg < NSEG; ++sg) {
          if (sg + 1 < NSEG)
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" ::"v"(lanepart[sg]), "s"(rs), "s"(dst + (unsigned int)(sg * 1024)), "s"(soff) : "memory");
          else     // the row's last two pixels: the other lanes must not write (their slots are the next row's first pixels)
            asm volatile("s_mov_b32 m0, %2\n\ts_mov_b64 exec, %4\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds\n\ts_mov_b64 exec, -1" ::"v"(lanepart[sg]), "s"(rs), "s"(dst + (unsigned int)(sg * 1024)), "s"(soff), "n"((1ull << (4 * TAIL)) - 1ull) : "memory");
        }
      }
    }
  };
  stage_issue(0);
  weights_dma(0, 0, 0);
  weights_dma(1, 0, 1);
  // (behind the first requests, in the shadow of their latency)
  // fragment addresses: pixel (RPW wave + j + dy, ql + dx), piece 2 s + hh (k-step s of the slice) -> fbase[dx][s] + (j + dy) * LW * 64
  unsigned int fbase[3][2];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      const int lx = ql + dx;
      fbase[dx][pl] = (unsigned int)((RPW * wave * LW + lx) * 64 + (((2 * pl + hh) ^ fsw(lx)) * 16));
    }
  f32x16 acc[2][RPW];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < RPW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int c0 = 0; c0 < a.Cin; c0 += 32) {
    const bool more = c0 + 32 < a.Cin;
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const int dy = tap / 3, dx = tap % 3;
      // weights two taps ahead, into the buffer tap - 1 has released (its readers passed the barrier in front of this tap)
      const bool issue = tap + 2 < TAPS || more;
      if (tap + 2 < TAPS) weights_dma(tap + 2, c0, (tap + 2) % 3);
      else if (more) weights_dma(tap + 2 - TAPS, c0 + 32, (tap + 2) % 3);
      const unsigned char* const wb = wbuf + (tap % 3) * 4096;
      h16x8 fa[2][2], fb[RPW][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) fa[i][pl] = *reinterpret_cast<const h16x8*>(wb + (i * 2 + pl) * 1024 + lane16);
#pragma unroll
      for (int j = 0; j < RPW; ++j)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) fb[j][pl] = *reinterpret_cast<const h16x8*>(tb + fbase[dx][pl] + (j + dy) * LW * 64);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < RPW; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][ks], fb[j][ks], acc[i][j], 0, 0, 0);
      if (tap + 1 < TAPS) {
        // the next tap's weights (requested a tap ago) must have landed; the request of this tap may stay in flight
        if (issue) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
    }
    if (more) {
      __syncthreads();      // every wave is done with this slice's tile
      stage_issue(c0 + 32);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }

  // epilogue: k_sp_conv_s16's, with fp16 (or f32) records instead of hm16
  f32x4 b4[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) b4[i][g] = *reinterpret_cast<const f32x4*>(a.bias + 64 * og + 32 * i + 8 * g + 4 * hh);
  const float ascale = a.acc_scale;
  float amax = 0.f;
  constexpr bool out_h = OUTH;
  const float relu_lo = a.relu ? 0.f : -INFINITY;
  unsigned char* const slab = smem + wave * SLAB;
  __syncthreads();                                    // every wave is done with the tile and the weight buffers
  auto put4 = [&](int p, int i, int g, f32x2v v0, f32x2v v1) __attribute__((always_inline)) {
    const int cl = 32 * i + 8 * g + 4 * hh;
    if (out_h) {
      amax = __builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fabsf(v0[0]), __builtin_fabsf(v0[1])));
      amax = __builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fabsf(v1[0]), __builtin_fabsf(v1[1])));
      *reinterpret_cast<uint2*>(slab + p * PSTR + cl * 2) = make_uint2(pack16<true>(v0[0], v0[1]), pack16<true>(v1[0], v1[1]));
    } else {
      *reinterpret_cast<f32x4*>(slab + p * PSTR + cl * 4) = (f32x4){v0[0], v0[1], v1[0], v1[1]};
    }
  };
  const size_t orec = (size_t)a.Cout * (OUTH ? 2 : 4);
  unsigned char* const obase = reinterpret_cast<unsigned char*>(a.out) + (size_t)og * SEG;
  constexpr int CPP = SEG / 16, PPI = 64 / CPP;      // 16-byte chunks per pixel segment, pixels per read-back instruction
  const unsigned int rd = (unsigned int)((lane / CPP) * PSTR + (lane % CPP) * 16);
  const unsigned int so = (unsigned int)(lane / CPP) * (unsigned int)orec + (unsigned int)(lane % CPP) * 16u;
  if (POOL) {
    const int gy = y0 + RPW * wave;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x2v m[2];
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const float bk = b4[i][g][2 * k + e];
            float mv = fmaxf(fmaxf(sp_scale_bias(acc[i][0][4 * g + 2 * k + e], ascale, bk), sp_scale_bias(acc[i][1][4 * g + 2 * k + e], ascale, bk)), relu_lo);
            mv = fmaxf(mv, __shfl_xor(mv, 1));
            m[k][e] = mv;
          }
        if (!(ql & 1)) put4(ql >> 1, i, g, m[0], m[1]);
      }
    if (out_h) ovf_commit(a.ovf, amax);
    const size_t prow = ((size_t)img * (a.H / 2) + (gy >> 1)) * (a.W / 2) + (x0 >> 1);
    const int npix = gy + 1 < a.H ? min(16, (a.W - x0) / 2) : 0;
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(obase + prow * orec, 0, npix > 0 ? (unsigned int)((npix - 1) * orec + SEG) : 0u, 0x00020000);
#pragma unroll
    for (int it = 0; it < 16 / PPI; ++it) {
      const u32x4_sp v = *reinterpret_cast<const u32x4_sp*>(slab + rd + it * PPI * PSTR);
      __builtin_amdgcn_raw_buffer_store_b128(v, ors, so, (unsigned int)(it * PPI) * (unsigned int)orec, 0);
      sp_store_guard();
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < RPW; ++j) {
    const int gy = y0 + RPW * wave + j;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x2v v[2];
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
          for (int e = 0; e < 2; ++e) v[k][e] = fmaxf(sp_scale_bias(acc[i][j][4 * g + 2 * k + e], ascale, b4[i][g][2 * k + e]), relu_lo);
        put4(ql, i, g, v[0], v[1]);
      }
    const size_t prow = ((size_t)img * a.H + gy) * a.W + x0;
    const int npix = gy < a.H ? min(32, a.W - x0) : 0;
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(obase + prow * orec, 0, npix > 0 ? (unsigned int)((npix - 1) * orec + SEG) : 0u, 0x00020000);
#pragma unroll
    for (int it = 0; it < 32 / PPI; ++it) {
      const u32x4_sp v = *reinterpret_cast<const u32x4_sp*>(slab + rd + it * PPI * PSTR);
      __builtin_amdgcn_raw_buffer_store_b128(v, ors, so, (unsigned int)(it * PPI) * (unsigned int)orec, 0);
      sp_store_guard();
    }
  }
  if (out_h) ovf_commit(a.ovf, amax);
}

// 2x2 max-pool, NHWC; thread -> (output pixel, 4 channels)
__global__ __launch_bounds__(256) void k_sp_pool(const float* in, float* out, int H, int W, int C, long long total4) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total4) return;
  const int c4 = (int)(idx % (C / 4));
  long long p = idx / (C / 4);
  const int Wo = W / 2, Ho = H / 2;
  const int x = (int)(p % Wo); p /= Wo;
  const int y = (int)(p % Ho); const long long b = p / Ho;
  const float* s = in + ((b * H + 2 * y) * W + 2 * x) * C + c4 * 4;
  const f32x4 v00 = *reinterpret_cast<const f32x4*>(s), v01 = *reinterpret_cast<const f32x4*>(s + C);
  const f32x4 v10 = *reinterpret_cast<const f32x4*>(s + (long long)W * C), v11 = *reinterpret_cast<const f32x4*>(s + (long long)W * C + C);
  f32x4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) r[e] = fmaxf(fmaxf(v00[e], v01[e]), fmaxf(v10[e], v11[e]));
  *reinterpret_cast<f32x4*>(out + ((b * Ho + y) * Wo + x) * C + c4 * 4) = r;
}

// softmax over the 65 logits of a cell (channel pitch cp), dustbin dropped, 8x8 depth-to-space.  One wave per cell.
__global__ __launch_bounds__(256) void k_sp_scores(const float* logits, int cp, float* scores, int h, int w, long long cells) {
  const int lane = threadIdx.x & 63;
  const long long cell = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (cell >= cells) return;
  const float* l = logits + cell * cp;
  const float v = l[lane], d = l[64];
  float m = fmaxf(v, d);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  const float e = expf(v - m);
  float sum = e;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  sum += expf(d - m);
  long long c = cell;
  const int cx = (int)(c % w); c /= w;
  const int cy = (int)(c % h); const long long b = c / h;
  scores[(b * h * 8 + cy * 8 + (lane >> 3)) * (long long)(w * 8) + cx * 8 + (lane & 7)] = e / sum;
}

// F.max_pool2d(kernel 2 r + 1, stride 1, padding r) split into a row pass and a column pass (-inf padding = ignore)
__global__ __launch_bounds__(256) void k_sp_maxrow(const float* in, float* out, int H, int W, int r, long long total) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int x = (int)(idx % W);
  const float* row = in + (idx - x);
  float m = -INFINITY;
  for (int d = -r; d <= r; ++d) { const int xx = x + d; if (xx >= 0 && xx < W) m = fmaxf(m, row[xx]); }
  out[idx] = m;
}
__global__ __launch_bounds__(256) void k_sp_maxcol(const float* in, float* out, int H, int W, int r, long long total) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int x = (int)(idx % W);
  const long long p = idx / W;
  const int y = (int)(p % H);
  const float* col = in + (idx - (long long)y * W);
  float m = -INFINITY;
  for (int d = -r; d <= r; ++d) { const int yy = y + d; if (yy >= 0 && yy < H) m = fmaxf(m, col[(long long)yy * W]); }
  out[idx] = m;
  (void)x;
}
// simple_nms mask algebra.  mode 0: mask = (scores == pooled);  mode 1: supp = pooled(mask) > 0, tmp = supp ? 0 : scores (written to `aux`);
// mode 2: mask |= (aux == pooled(aux)) & !supp  (supp recomputed as aux == 0 && ... is not exact, so it is kept in `supp`)
__global__ __launch_bounds__(256) void k_sp_nms_step(int mode, const float* scores, const float* pooled, float* mask, float* supp, float* aux, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  if (mode == 0) mask[i] = scores[i] == pooled[i] ? 1.f : 0.f;
  else if (mode == 1) { const float sp = pooled[i] > 0.f ? 1.f : 0.f; supp[i] = sp; aux[i] = sp != 0.f ? 0.f : scores[i]; }
  else if (mode == 2) { if (aux[i] == pooled[i] && supp[i] == 0.f) mask[i] = 1.f; }
  else { /* mode 3: final scores = mask ? scores : 0, in place into aux */ aux[i] = mask[i] != 0.f ? scores[i] : 0.f; }
}

// simple_nms(scores, 4) in ONE kernel (round 5; the 16 launches above moved 50 full-resolution maps through HBM per call: 0.69 ms per four 1080p
// frames).  The five 9 x 9 max-pools of the algorithm reach 4 + 2 x 8 = 20 pixels: a workgroup computes a 64 x 32 output tile from the
// 104 x 72 region around it, every intermediate in LDS; values within 4 k pixels of the region's edge are wrong after k pools and are never
// read by the tile.  A pool is a row pass and a column pass; one work item = 8 consecutive outputs of a row (column) from 16 inputs held in
// registers (window maxima by doubling: 45 maximum operations per 8 outputs).  Pixels outside the image are -inf (F.max_pool2d's padding),
// carry no mask and never become maxima.  Comparisons, selections and their order are those of k_sp_nms_step: identical output.
constexpr int NMS_TX = 64, NMS_TY = 32, NMS_HL = 20, NMS_RW = NMS_TX + 2 * NMS_HL, NMS_RH = NMS_TY + 2 * NMS_HL;
constexpr int NMS_LP = 8, NMS_PITCH = NMS_RW + 2 * NMS_LP;      // 8 pad columns of -inf on either side (16-byte aligned row segments)
__device__ __forceinline__ void nms_window9(const float (&v)[16], float (&o)[8]) {
  float m2[15], m4[13], m8[9];
#pragma unroll
  for (int i = 0; i < 15; ++i) m2[i] = fmaxf(v[i], v[i + 1]);
#pragma unroll
  for (int i = 0; i < 13; ++i) m4[i] = fmaxf(m2[i], m2[i + 2]);
#pragma unroll
  for (int i = 0; i < 9; ++i) m8[i] = fmaxf(m4[i], m4[i + 4]);
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = fmaxf(m8[i], v[i + 8]);
}
__global__ __launch_bounds__(1024) void k_sp_nms_fused(const float* scores, float* out, int H, int W) {
  constexpr int NP = NMS_RH * NMS_PITCH;
  __shared__ __attribute__((aligned(16))) float S[NP], A[NP], M[NP], P[NP];
  __shared__ unsigned char U[NMS_RH * NMS_RW];
  const int tid = threadIdx.x;
  const int x0 = blockIdx.x * NMS_TX - NMS_HL, y0 = blockIdx.y * NMS_TY - NMS_HL;      // image coordinates of region pixel (0, 0)
  const float* img = scores + (long long)blockIdx.z * H * W;
  float* oimg = out + (long long)blockIdx.z * H * W;
  const float NINF = -INFINITY;
  // scores -> S (-inf outside the image and in the pad columns); pads of A and M
  // (four columns per load: x0 and W are multiples of 4, so a group of four is inside the image or outside it as a whole)
  for (int i = tid; i < NMS_RH * (NMS_PITCH / 4); i += 1024) {
    const int ry = i / (NMS_PITCH / 4), c4 = i - ry * (NMS_PITCH / 4), rx = 4 * c4 - NMS_LP;
    const int gy = y0 + ry, gx = x0 + rx;
    const bool pad = rx < 0 || rx >= NMS_RW;
    const bool in = !pad && gy >= 0 && gy < H && gx >= 0 && gx < W;
    const f32x4 ninf4 = {NINF, NINF, NINF, NINF};
    const int o = ry * NMS_PITCH + 4 * c4;
    *reinterpret_cast<f32x4*>(S + o) = in ? *reinterpret_cast<const f32x4*>(img + (long long)gy * W + gx) : ninf4;
    if (pad) { *reinterpret_cast<f32x4*>(A + o) = ninf4; *reinterpret_cast<f32x4*>(M + o) = ninf4; }
  }
  __syncthreads();
  const bool active = tid < NMS_RH * (NMS_RW / 8);       // 936 items per pass, both passes
  // row item: row rr, outputs 8 rs .. 8 rs + 7
  const int rr = tid / (NMS_RW / 8), rs = tid - rr * (NMS_RW / 8);
  auto row_pass = [&](const float* X) __attribute__((always_inline)) {
    if (active) {
      float v[16], o[8];
      const f32x4* src = reinterpret_cast<const f32x4*>(X + rr * NMS_PITCH + NMS_LP + 8 * rs - 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) { const f32x4 t = src[q]; v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w; }
      nms_window9(v, o);
      f32x4* dst = reinterpret_cast<f32x4*>(P + rr * NMS_PITCH + NMS_LP + 8 * rs);
      dst[0] = (f32x4){o[0], o[1], o[2], o[3]};
      dst[1] = (f32x4){o[4], o[5], o[6], o[7]};
    }
    __syncthreads();
  };
  // column item: column cc, outputs rows 8 cs .. 8 cs + 7 (consecutive threads = consecutive columns)
  const int cs = tid / NMS_RW, cc = tid - cs * NMS_RW;
  const int gxc = x0 + cc;
  auto col_pool = [&](float (&o)[8]) __attribute__((always_inline)) {
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int ry = 8 * cs - 4 + k;
      const int ryc = min(max(ry, 0), NMS_RH - 1);
      const float t = P[ryc * NMS_PITCH + NMS_LP + cc];
      v[k] = (ry >= 0 && ry < NMS_RH) ? t : NINF;
    }
    nms_window9(v, o);
  };
  auto in_image = [&](int k) __attribute__((always_inline)) { const int gy = y0 + 8 * cs + k; return gy >= 0 && gy < H && gxc >= 0 && gxc < W; };

  row_pass(S);
  if (active) {
    float o[8];
    col_pool(o);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = (8 * cs + k) * NMS_PITCH + NMS_LP + cc;
      M[i] = (in_image(k) && S[i] == o[k]) ? 1.f : 0.f;
    }
  }
  __syncthreads();
  for (int it = 0; it < 2; ++it) {
    row_pass(M);
    if (active) {
      float o[8];
      col_pool(o);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int i = (8 * cs + k) * NMS_PITCH + NMS_LP + cc;
        const bool supp = o[k] > 0.f;
        U[(8 * cs + k) * NMS_RW + cc] = supp ? 1 : 0;
        A[i] = in_image(k) ? (supp ? 0.f : S[i]) : NINF;
      }
    }
    __syncthreads();
    row_pass(A);
    if (active) {
      float o[8];
      col_pool(o);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int i = (8 * cs + k) * NMS_PITCH + NMS_LP + cc;
        if (in_image(k) && A[i] == o[k] && U[(8 * cs + k) * NMS_RW + cc] == 0) M[i] = 1.f;
      }
    }
    __syncthreads();
  }
  // the tile: mask ? scores : 0
  for (int i = tid; i < NMS_TX * NMS_TY; i += 1024) {
    const int ty = i / NMS_TX, tx = i - ty * NMS_TX;
    const int gy = blockIdx.y * NMS_TY + ty, gx = blockIdx.x * NMS_TX + tx;
    if (gy < H && gx < W) {
      const int j = (NMS_HL + ty) * NMS_PITCH + NMS_LP + NMS_HL + tx;
      oimg[(long long)gy * W + gx] = M[j] != 0.f ? S[j] : 0.f;
    }
  }
}

// candidates: score > threshold, y >= border, x >= border (transformers tests the far borders against 8 x the map size, i.e. never).  A list entry
// is (raster index, score bits): k_sp_select streams the list instead of gathering the scores behind the indices.  One atomic per WORKGROUP of
// 2048 pixels (untrained weights make almost every NMS survivor a candidate: one atomic per candidate, then per wave, on the image's single counter
// was 0.2 ms per call); the order of the list is immaterial (k_sp_select ranks by score and raster index).
__global__ __launch_bounds__(256) void k_sp_candidates(const float* nms, int H, int W, float thr, int border, int2* cand /*[B][cap]*/, int* counts /*[B][4]*/, int cap) {
  const int b = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long hw = (long long)H * W, base = (long long)blockIdx.x * 2048;
  __shared__ int wsum[4];
  __shared__ int wgbase;
  float sv[8];
  unsigned long long bal[8];
  int cnt = 0;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const long long i = base + it * 256 + tid;
    const bool in = i < hw;
    sv[it] = in ? nms[(long long)b * hw + i] : 0.f;
    const int y = (int)(i / W), x = (int)(i - (long long)y * W);
    bal[it] = __ballot(in && sv[it] > thr && y >= border && x >= border);
    cnt += __popcll(bal[it]);
  }
  if (lane == 0) wsum[wave] = cnt;
  __syncthreads();
  if (tid == 0) {
    const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    wgbase = total > 0 ? atomicAdd(&counts[4 * b], total) : 0;
  }
  __syncthreads();
  int slot0 = wgbase;
  for (int w = 0; w < wave; ++w) slot0 += wsum[w];
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    if ((bal[it] >> lane) & 1ull) {
      const int slot = slot0 + __popcll(bal[it] & ((1ull << lane) - 1ull));
      if (slot < cap) cand[(long long)b * cap + slot] = make_int2((int)(base + it * 256 + tid), __float_as_int(sv[it]));
    }
    slot0 += __popcll(bal[it]);
  }
}

// top-k by score, one workgroup per image: radix select of the k-th largest score over the candidates, survivors compacted into
// LDS and rank-sorted by (score descending, raster index ascending).  Outputs GN_KPT_XYSA keypoint records and scores.
// CACHED: the image's candidates fit the workgroup's registers (PT per thread, read ONCE with all loads in flight); the eight radix passes and the
// survivor pass then run from registers -- streamed, each pass was a chain of list reads by one workgroup (0.24 ms per four 1080p frames).  Histogram
// updates are aggregated per wave for the bin most lanes share (a flat score map puts every candidate of a pass into one or two bins: 64 serialised LDS
// atomics per instruction otherwise).
constexpr int kSelPT = 40;
template <bool CACHED>
__device__ __forceinline__ void sp_select_core(int W, const int2* cd, int n, int kk, int b, int* counts, float* kpt_xy, float* score_out, int* kp_index, long long out_stride,
                                               int* s_hist, unsigned int& s_prefix, int& s_need, int& s_cnt, int* s_idx, float* s_val) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int nu = (n + 1023) >> 10;               // list entries per thread (uniform)
  int2 e[CACHED ? kSelPT : 1];
  if constexpr (CACHED) {
#pragma unroll
    for (int u = 0; u < kSelPT; ++u) { const int i = tid + 1024 * u; e[u] = (u < nu && i < n) ? cd[i] : make_int2(-1, 0); }
  }
  auto item = [&](int u) __attribute__((always_inline)) -> int2 {
    if constexpr (CACHED) return e[u];
    else { const int i = tid + 1024 * u; return i < n ? cd[i] : make_int2(-1, 0); }
  };
  auto hist_add = [&](bool on, int bin) __attribute__((always_inline)) {
    const unsigned long long act = __ballot(on);
    if (act == 0ull) return;
    const int leader = __ffsll((long long)act) - 1;
    const int b0 = __shfl(bin, leader);
    const unsigned long long same = __ballot(on && bin == b0);
    if (lane == leader) atomicAdd(&s_hist[b0], __popcll(same));
    if (on && bin != b0) atomicAdd(&s_hist[bin], 1);
  };
  // pass(key, accept): radix select over the keys of the accepted entries; largest = true finds the need-th LARGEST key, else the need-th smallest
  auto radix = [&](int need0, bool largest, auto key, auto accept) __attribute__((always_inline)) {
    if (tid == 0) { s_prefix = 0u; s_need = need0; }
    for (int shift = 24; shift >= 0; shift -= 8) {
      if (tid < 256) s_hist[tid] = 0;
      __syncthreads();
      const unsigned int prefix = s_prefix;
      if constexpr (CACHED) {
#pragma unroll
        for (int u = 0; u < kSelPT; ++u)
          if (u < nu) { const int2 it = item(u); const unsigned int kv = key(it); hist_add(it.x >= 0 && accept(it) && (shift == 24 || ((kv ^ prefix) >> (shift + 8)) == 0u), (int)((kv >> shift) & 255u)); }
      } else {
        for (int u = 0; u < nu; ++u) { const int2 it = item(u); const unsigned int kv = key(it); hist_add(it.x >= 0 && accept(it) && (shift == 24 || ((kv ^ prefix) >> (shift + 8)) == 0u), (int)((kv >> shift) & 255u)); }
      }
      __syncthreads();
      // the bin that holds the need-th key, by one wave (thread 0 walking the 256 bins was 256 dependent LDS reads per pass: most of the kernel):
      // lane l owns bins 4 l .. 4 l + 3; running sums from the top (largest) or the bottom, the bin is where they first reach `need`
      if (tid < 64) {
        const int need = s_need;
        int c[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) c[q] = s_hist[4 * lane + q];
        const int t = c[0] + c[1] + c[2] + c[3];
        int run = t;      // inclusive scan over the lanes: from the top lane down (largest) or from lane 0 up
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const int o = __shfl(run, largest ? lane + d : lane - d);
          if (largest ? lane + d < 64 : lane - d >= 0) run += o;
        }
        const int before = run - t;      // keys in the bins beyond this lane's (above it / below it)
        int acc4[4];                     // running sum INCLUDING the bin, walking this lane's bins in the search direction
        int below = 0;                   // bins of this lane whose running sum is still short of need
        int sum = before;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int bq = largest ? 3 - q : q; sum += c[bq]; acc4[bq] = sum; below += sum < need ? 1 : 0; }
        int nshort = below;              // bins (in search order) before the one that reaches need
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) nshort += __shfl_xor(nshort, d);
        nshort = min(nshort, 255);
        const int bin = largest ? 255 - nshort : nshort;
        if (lane == (bin >> 2)) {
          const int incl = acc4[bin & 3];
          s_need = need - (incl - c[bin & 3]);
          s_prefix = prefix | ((unsigned int)bin << shift);
        }
      }
      __syncthreads();
    }
  };
  unsigned int tbits = 0u, ithr = 0xffffffffu;
  if (n > kk) {
    radix(kk, true, [](const int2& it) { return (unsigned int)it.y; }, [](const int2&) { return true; });
    tbits = s_prefix;
    const int eq_budget = s_need;
    __syncthreads();      // every thread has its copies before thread 0 re-uses the two words (a fast thread 0 used to zero s_prefix under slower waves: they
                          // selected with threshold 0 -- a whole image's keypoint list changed in ~1 of 10 runs once the passes got faster)
    // ties at the threshold are taken in raster order: a second radix select, over the raster index of the candidates whose score equals T, finds the
    // eq_budget-th SMALLEST index (a flat score map -- untrained weights -- makes almost every candidate a tie)
    radix(eq_budget, false, [](const int2& it) { return (unsigned int)it.x; }, [tbits](const int2& it) { return (unsigned int)it.y == tbits; });
    ithr = s_prefix;
  }
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  // survivors: score > T, plus the ties with a raster index up to the one found above
  auto keep = [&](const int2& it) __attribute__((always_inline)) {
    const unsigned int bits = (unsigned int)it.y;
    if (it.x >= 0 && (n <= kk || bits > tbits || (bits == tbits && (unsigned int)it.x <= ithr))) { const int p = atomicAdd(&s_cnt, 1); if (p < 2048) { s_idx[p] = it.x; s_val[p] = __int_as_float(it.y); } }
  };
  if constexpr (CACHED) {
#pragma unroll
    for (int u = 0; u < kSelPT; ++u) if (u < nu) keep(item(u));
  } else {
    for (int u = 0; u < nu; ++u) keep(item(u));
  }
  __syncthreads();
  const int m = min(s_cnt, min(kk, 2048));
  for (int i = tid; i < m; i += 1024) {
    const float v = s_val[i]; const int ci = s_idx[i];
    int rank = 0;
    // (four entries per step: two 16-byte LDS broadcasts instead of eight 4-byte ones; entries at and beyond m never outrank: score -1)
    for (int j = 0; j < m; j += 4) {
      const f32x4 vj = *reinterpret_cast<const f32x4*>(s_val + j);
      const int4 ij = *reinterpret_cast<const int4*>(s_idx + j);
      rank += (j + 0 < m && (vj.x > v || (vj.x == v && ij.x < ci))) ? 1 : 0;
      rank += (j + 1 < m && (vj.y > v || (vj.y == v && ij.y < ci))) ? 1 : 0;
      rank += (j + 2 < m && (vj.z > v || (vj.z == v && ij.z < ci))) ? 1 : 0;
      rank += (j + 3 < m && (vj.w > v || (vj.w == v && ij.w < ci))) ? 1 : 0;
    }
    const int y = ci / W, x = ci - y * W;
    float* kr = kpt_xy + ((long long)b * out_stride + rank) * 4;       // GN_KPT_XYSA record: x, y, size (unused: 1), angle (unused: 0)
    kr[0] = (float)x; kr[1] = (float)y; kr[2] = 1.f; kr[3] = 0.f;
    score_out[(long long)b * out_stride + rank] = v;
    kp_index[(long long)b * out_stride + rank] = ci;
  }
  if (tid == 0) { counts[4 * b + 1] = m; counts[4 * b + 2] = n; }
}
__global__ __launch_bounds__(1024) void k_sp_select(int H, int W, const int2* cand, int* counts, int cap, int k,
                                                      float* kpt_xy, float* score_out, int* kp_index, long long out_stride, int force_stream) {
  const int b = blockIdx.x;
  const int2* cd = cand + (long long)b * cap;
  const int n = min(counts[4 * b], cap);
  __shared__ int s_hist[256];
  __shared__ unsigned int s_prefix;
  __shared__ int s_need, s_cnt;
  __shared__ __attribute__((aligned(16))) int s_idx[2048];
  __shared__ __attribute__((aligned(16))) float s_val[2048];
  const int kk = min(k, 2048);
  if (n <= kSelPT * 1024 && !force_stream) sp_select_core<true>(W, cd, n, kk, b, counts, kpt_xy, score_out, kp_index, out_stride, s_hist, s_prefix, s_need, s_cnt, s_idx, s_val);
  else sp_select_core<false>(W, cd, n, kk, b, counts, kpt_xy, score_out, kp_index, out_stride, s_hist, s_prefix, s_need, s_cnt, s_idx, s_val);
}

// descriptors: one wave per keypoint; lane -> 4 of the 256 channels.  dmap [B][h][w][256] (conv_descriptor_b output, NOT yet normalised)
__global__ __launch_bounds__(256) void k_sp_describe(const float* dmap, int h, int w, const float* kpt_xy, const int* counts, long long out_stride, float* desc) {
  const int lane = threadIdx.x & 63, b = blockIdx.y;
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= counts[4 * b + 1]) return;
  const float kx = kpt_xy[((long long)b * out_stride + k) * 4], ky = kpt_xy[((long long)b * out_stride + k) * 4 + 1];
  // _sample_descriptors: kp - 4 + 0.5, / (w 8 - 4 - 0.5), * 2 - 1; grid_sample(align_corners=True): ((g + 1) / 2) (w - 1)
  const float gx = ((kx - 4.0f + 0.5f) / ((float)(w * 8) - 4.0f - 0.5f)) * 2.0f - 1.0f;
  const float gy = ((ky - 4.0f + 0.5f) / ((float)(h * 8) - 4.0f - 0.5f)) * 2.0f - 1.0f;
  const float ix = ((gx + 1.0f) / 2.0f) * (float)(w - 1), iy = ((gy + 1.0f) / 2.0f) * (float)(h - 1);
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  const int x0 = (int)fx0, y0 = (int)fy0;
  const float wx1 = ix - fx0, wy1 = iy - fy0, wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
  const float* base = dmap + (long long)b * h * w * 256;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int xx = x0 + (q & 1), yy = y0 + (q >> 1);
    const float wt = ((q & 1) ? wx1 : wx0) * ((q >> 1) ? wy1 : wy0);
    if (xx < 0 || xx >= w || yy < 0 || yy >= h) continue;     // zero padding
    const f32x4 v = *reinterpret_cast<const f32x4*>(base + ((long long)yy * w + xx) * 256 + lane * 4);
    float n2 = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n2 += __shfl_xor(n2, o);
    const float inv = 1.0f / fmaxf(sqrtf(n2), 1e-12f);        // F.normalize(p=2, dim=1, eps=1e-12) of the map
    acc += v * (inv * wt);
  }
  float n2 = acc.x * acc.x + acc.y * acc.y + acc.z * acc.z + acc.w * acc.w;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) n2 += __shfl_xor(n2, o);
  const float inv = 1.0f / fmaxf(sqrtf(n2), 1e-12f);
  *reinterpret_cast<f32x4*>(desc + ((long long)b * out_stride + k) * 256 + lane * 4) = acc * inv;
}
}  // namespace

// ------------------------------------------------------------------------------------------------ host side
// conv weight [Cout][Cin][kh][kw] (torch layout) -> fragment order [Cout_pad/32][TAPS][Cin/8][64 lanes][4]: lane (n = l & 31, hh = l >> 5)
// holds W[32 tile + n][tap][8 cs + 4 hh .. + 3]; rows >= Cout are zero.
void sp_weight_fragments(const float* w, int Cout, int Cin, int taps, int Cout_pad, float* out) {
  const int csteps = Cin / 8;
  for (int tile = 0; tile < Cout_pad / 32; ++tile)
    for (int tap = 0; tap < taps; ++tap)
      for (int cs = 0; cs < csteps; ++cs)
        for (int l = 0; l < 64; ++l)
          for (int e = 0; e < 4; ++e) {
            const int n = 32 * tile + (l & 31), c = 8 * cs + 4 * (l >> 5) + e;
            const float v = n < Cout ? w[((size_t)n * Cin + c) * taps + tap] : 0.f;
            out[((((size_t)tile * taps + tap) * csteps + cs) * 64 + l) * 4 + e] = v;
          }
}

// fp16 pairs (w * scale = h + m) in the fragment order of v_mfma_f32_32x32x16_f16: block (((tile * taps + tap) * (Cin / 16) + kstep) * 2 + term)
// of 1 KB, lane l -> output channel 32 tile + (l & 31), input channels 16 kstep + 8 (l >> 5) + e
void sp_weight_fragments_hm16(const float* w, int Cout, int Cin, int taps, int Cout_pad, float scale, uint16_t* out) {
  const int ksteps = Cin / 16;
  for (int tile = 0; tile < Cout_pad / 32; ++tile)
    for (int tap = 0; tap < taps; ++tap)
      for (int ks = 0; ks < ksteps; ++ks)
        for (int l = 0; l < 64; ++l)
          for (int e = 0; e < 8; ++e) {
            const int n = 32 * tile + (l & 31), c = 16 * ks + 8 * (l >> 5) + e;
            const float x = (n < Cout ? w[((size_t)n * Cin + c) * taps + tap] : 0.f) * scale;
            const _Float16 h = (_Float16)x;
            const _Float16 m = (_Float16)(x - (float)h);
            const size_t blk = (((size_t)tile * taps + tap) * ksteps + ks) * 2;
            out[(blk + 0) * 512 + l * 8 + e] = __builtin_bit_cast(uint16_t, h);
            out[(blk + 1) * 512 + l * 8 + e] = __builtin_bit_cast(uint16_t, m);
          }
}

int g_sp_conv_h = 2;   // developer knob 24: 2 = GN_SP_FP16 3 x 3 layers with fp16 input through k_sp_conv_h16 (round 5), 1 = all through k_sp_conv_h, 0 = through k_sp_conv<9, 2, ...> (the first single-product kernel)
int g_sp_conv_s = 2;   // developer knob 34: 0 = the split-fp16 mode on f32 activations through k_sp_conv<., 1, ...> (the round-2 kernel); 1 = hm16 activations, k_sp_conv_s
                       // for every layer (bitwise the round-2 results); 2 = k_sp_conv_s16 for the 3 x 3 layers (5 % faster, another k order: f32 rounding)
void sp_conv1(const float* in, const float* w, const float* bias, float* out, int B, int H, int W, hipStream_t s, int out_half, unsigned int* ovf) {
  const long long n = (long long)H * W;      // threads: (quad of pixels, 16-channel group) = H * W / 4 * 4
  const dim3 grid((unsigned)((n + 255) / 256), 1, B);
  if (out_half == 2) hipLaunchKernelGGL(k_sp_conv1<2>, grid, dim3(256), 0, s, in, w, bias, out, H, W, ovf);
  else if (out_half) hipLaunchKernelGGL(k_sp_conv1<1>, grid, dim3(256), 0, s, in, w, bias, out, H, W, ovf);
  else hipLaunchKernelGGL(k_sp_conv1<0>, grid, dim3(256), 0, s, in, w, bias, out, H, W, ovf);
}
void sp_conv(const float* in, int B, int H, int W, int Cin, const float* wf, const float* bias, float* out, int Cout_pad, int taps, int relu, hipStream_t s,
             const uint16_t* wfh, float acc_scale, unsigned int* ovf, int pool, int single_product, int in_half, int out_half, long long* dbg_ts) {
  ConvArgs a; a.in = in; a.H = H; a.W = W; a.Cin = Cin; a.wf = wf; a.bias = bias; a.out = out; a.Cout = Cout_pad; a.relu = relu;
  a.wfh = wfh; a.acc_scale = acc_scale; a.ovf = ovf; a.in_half = in_half; a.out_half = out_half; a.dbg_ts = dbg_ts;
  const int th = pool ? 8 : 12;
  const dim3 grid((W + TW - 1) / TW, (H + th - 1) / th, B * (Cout_pad / 64));
  const bool hm = wfh != nullptr;
  const bool single = hm && single_product;
  if (hm && !single && in_half == 2) {   // split fp16 on hm16 activations
    if (taps == 9 && g_sp_conv_s == 2) {   // 16-channel slices, 8 x 32 tiles
      const dim3 g8((W + TW - 1) / TW, (H + 7) / 8, B * (Cout_pad / 64));
      if (pool) hipLaunchKernelGGL((k_sp_conv_s16<true, true>), g8, dim3(256), 0, s, a);
      else if (out_half == 2) hipLaunchKernelGGL((k_sp_conv_s16<false, true>), g8, dim3(256), 0, s, a);
      else hipLaunchKernelGGL((k_sp_conv_s16<false, false>), g8, dim3(256), 0, s, a);
      return;
    }
    if (pool) hipLaunchKernelGGL((k_sp_conv_s<9, 2, true, true>), grid, dim3(256), 0, s, a);      // (the pooled layers are inner layers)
    else if (taps == 9 && out_half == 2) hipLaunchKernelGGL((k_sp_conv_s<9, 3, false, true>), grid, dim3(256), 0, s, a);
    else if (taps == 9) hipLaunchKernelGGL((k_sp_conv_s<9, 3, false, false>), grid, dim3(256), 0, s, a);
    else if (out_half == 2) hipLaunchKernelGGL((k_sp_conv_s<1, 3, false, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_sp_conv_s<1, 3, false, false>), grid, dim3(256), 0, s, a);
    return;
  }
  if (single && taps == 9 && g_sp_conv_h == 2 && in_half == 1) {   // fp16 activations in: k_sp_conv_s16's form
    const dim3 g8((W + TW - 1) / TW, (H + 7) / 8, B * (Cout_pad / 64));
    if (pool) hipLaunchKernelGGL((k_sp_conv_h16<true, true>), g8, dim3(256), 0, s, a);     // (the pooled layers write fp16)
    else if (out_half == 1) hipLaunchKernelGGL((k_sp_conv_h16<false, true>), g8, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_sp_conv_h16<false, false>), g8, dim3(256), 0, s, a);
    return;
  }
  if (single && taps == 9 && g_sp_conv_h) {
    const dim3 grid4((W + TW - 1) / TW, (H + TH4 - 1) / TH4, B * (Cout_pad / 64));
    if (pool) hipLaunchKernelGGL((k_sp_conv_h<true>), grid4, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_sp_conv_h<false>), grid4, dim3(256), 0, s, a);
    return;
  }
  if (pool) {   // 3 x 3 layers only (the SuperPoint blocks that end in a max-pool)
    if (single) hipLaunchKernelGGL((k_sp_conv<9, 2, 2, true>), grid, dim3(256), 0, s, a);
    else if (hm) hipLaunchKernelGGL((k_sp_conv<9, 1, 2, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_sp_conv<9, 0, 2, true>), grid, dim3(256), 0, s, a);
    return;
  }
  if (single) {
    if (taps == 9) hipLaunchKernelGGL((k_sp_conv<9, 2, 3, false>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_sp_conv<1, 2, 3, false>), grid, dim3(256), 0, s, a);
    return;
  }
  if (hm) {
    if (taps == 9) hipLaunchKernelGGL((k_sp_conv<9, 1, 3, false>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_sp_conv<1, 1, 3, false>), grid, dim3(256), 0, s, a);
    return;
  }
  if (taps == 9) hipLaunchKernelGGL((k_sp_conv<9, 0, 3, false>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((k_sp_conv<1, 0, 3, false>), grid, dim3(256), 0, s, a);
}
// layers 0 + 1 of the split-fp16 mode in ONE launch (k_sp_conv_s16<true, true, true>): gray image in, pooled hm16 records of layer 1 out
int g_sp_fuse1 = 1;    // developer knob 46: 0 = k_sp_conv1 + k_sp_conv_s16 as two launches (the round-5 form; bitwise the same results)
void sp_conv_fused1(const float* gray, const float* w1, const float* b1, int B, int H, int W, const float* bias, float* out, int Cout_pad, hipStream_t s,
                    const uint16_t* wfh, float acc_scale, unsigned int* ovf) {
  ConvArgs a; a.in = nullptr; a.H = H; a.W = W; a.Cin = 64; a.wf = nullptr; a.bias = bias; a.out = out; a.Cout = Cout_pad; a.relu = 1;
  a.wfh = wfh; a.acc_scale = acc_scale; a.ovf = ovf; a.in_half = 2; a.out_half = 2; a.dbg_ts = nullptr; a.img = gray; a.w1 = w1; a.b1 = b1;
  const dim3 g8((W + TW - 1) / TW, (H + 7) / 8, B * (Cout_pad / 64));
  hipLaunchKernelGGL((k_sp_conv_s16<true, true, true>), g8, dim3(256), 0, s, a);
}
void sp_pool(const float* in, float* out, int B, int H, int W, int C, hipStream_t s) {
  const long long total4 = (long long)B * (H / 2) * (W / 2) * (C / 4);
  hipLaunchKernelGGL(k_sp_pool, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, in, out, H, W, C, total4);
}
void sp_scores(const float* logits, int cp, float* scores, int B, int h, int w, hipStream_t s) {
  const long long cells = (long long)B * h * w;
  hipLaunchKernelGGL(k_sp_scores, dim3((unsigned)((cells + 3) / 4)), dim3(256), 0, s, logits, cp, scores, h, w, cells);
}
// simple_nms(scores, r) -> `aux` (the suppressed score map); scratch: pooled, tmp, mask, supp (each B*H*W floats)
int g_sp_select_stream = 0;   // developer knob 40: 1 = k_sp_select reads the candidate list from memory in every pass (the path of frames with more than 40 960 candidates)
int g_sp_nms_fused = 1;   // developer knob 36: 0 = simple_nms as 16 full-resolution launches (the round-1 form)
void sp_nms(const float* scores, int B, int H, int W, int r, float* pooled, float* tmp, float* mask, float* supp, float* aux, hipStream_t s) {
  if (g_sp_nms_fused && r == 4) {
    hipLaunchKernelGGL(k_sp_nms_fused, dim3((W + NMS_TX - 1) / NMS_TX, (H + NMS_TY - 1) / NMS_TY, B), dim3(1024), 0, s, scores, aux, H, W);
    return;
  }
  const long long total = (long long)B * H * W;
  const dim3 g((unsigned)((total + 255) / 256)), blk(256);
  auto pool = [&](const float* src) {
    hipLaunchKernelGGL(k_sp_maxrow, g, blk, 0, s, src, tmp, H, W, r, total);
    hipLaunchKernelGGL(k_sp_maxcol, g, blk, 0, s, tmp, pooled, H, W, r, total);
  };
  pool(scores);
  hipLaunchKernelGGL(k_sp_nms_step, g, blk, 0, s, 0, scores, pooled, mask, supp, aux, total);
  for (int it = 0; it < 2; ++it) {
    pool(mask);
    hipLaunchKernelGGL(k_sp_nms_step, g, blk, 0, s, 1, scores, pooled, mask, supp, aux, total);
    pool(aux);
    hipLaunchKernelGGL(k_sp_nms_step, g, blk, 0, s, 2, scores, pooled, mask, supp, aux, total);
  }
  hipLaunchKernelGGL(k_sp_nms_step, g, blk, 0, s, 3, scores, pooled, mask, supp, aux, total);
}
void sp_select(const float* nms, int B, int H, int W, float thr, int border, int* cand, int* counts, int cap, int k,
               float* kpt_xy, float* score, int* kp_index, long long out_stride, hipStream_t s) {
  hipMemsetAsync(counts, 0, (size_t)B * 4 * sizeof(int), s);
  hipLaunchKernelGGL(k_sp_candidates, dim3((unsigned)(((long long)H * W + 2047) / 2048), 1, B), dim3(256), 0, s, nms, H, W, thr, border, reinterpret_cast<int2*>(cand), counts, cap);
  hipLaunchKernelGGL(k_sp_select, dim3(B), dim3(1024), 0, s, H, W, reinterpret_cast<const int2*>(cand), counts, cap, k, kpt_xy, score, kp_index, out_stride, g_sp_select_stream);
}
void sp_describe(const float* dmap, int B, int h, int w, const float* kpt_xy, const int* counts, long long out_stride, int max_k, float* desc, hipStream_t s) {
  hipLaunchKernelGGL(k_sp_describe, dim3((max_k + 3) / 4, B), dim3(256), 0, s, dmap, h, w, kpt_xy, counts, out_stride, desc);
}

}  // namespace gn
