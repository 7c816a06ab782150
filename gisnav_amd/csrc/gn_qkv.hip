// Attention input projections of the f16x2 precision mode, one launch per block:
//
//     self block :  [q | k | v] = Wqkv x + b,  rotary(q, k),  q *= dim_head^-0.5           (kornia SelfBlock: Wqkv + apply_cached_rotary_emb)
//     cross block:  qk = (to_qk x + b) * dim_head^-0.25,  v = to_v x + b                   (kornia CrossBlock: to_qk / to_v)
//
// reached from ros/gisnav/gisnav/core/pose_node.py:285-287.  Outputs are what k_attn_bf16_v5 reads: q | k (or qk) as bf16 rows
// and V TRANSPOSED as bf16 [slot][head][d][npad] with the keys permuted inside 16-groups.  Replaces k_gemm_p2w<EPI_ROTARY_BF16> /
// <EPI_SCALE_BF16> (gn_gemm_p2.hip), which staged BOTH operands through LDS by LDS-DMA and ran at 0.29 of the matrix pipe's
// issue rate: the fill rate of LDS, not the MFMAs, set its pace.
//
// Same construction as the block tail (gn_ffn.hip): the weights come straight from L2 into registers in MFMA fragment order
// (build_weight_fragments, natural k), through a ring of k-steps; only the token tile (hm16 rows) is staged in LDS, once.
// 8 waves x 128 tokens per workgroup; wave w owns feature tile w of every pass (see below).  Arithmetic: hm16 scheme, three
// v_mfma_f32_32x32x16_f16 per block.
#include "gn_common.h"
#include <algorithm>

namespace gn {

namespace {
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ int swz(int row) { return (row ^ (row >> 3)) & 7; }
// pack16<F16> (gn_common.h): two f32 -> one dword of two bf16 / fp16 (round to nearest even): ONE v_cvt_pk_* (the integer rounding it
// replaces was 9 VALU instructions per pair and half of the q / k epilogues' instruction count)

constexpr int TM = 128;                 // tokens per workgroup
constexpr int KT = TM * 128;            // bytes of one 32-wide k-tile of the token tile (hm16: 128 B per row): 16 KB
constexpr int NJ = TM / 32;             // token tiles of 32

// What bounds these projections is how many bytes a CU has to pull out of L2 per token (the 64-token version of this kernel ran
// at 8.3 TB/s of L2 -> CU traffic and exactly as fast as the LDS-staged GEMM it replaced): 768 KB of weights per workgroup are
// the bulk of it, so a workgroup serves 128 tokens -- the token tile fills LDS (128 KB), and the feature panels are visited in
// PASSES (q, k, v / qk, v) so that the accumulators of one pass fit the registers of 8 waves; every weight byte is still fetched
// once per workgroup.  Results leave straight from the registers:
//   q / k passes: transposed GEMM (weights = A operand): a lane owns 4 consecutive features of one token = one 8-byte bf16 store
//                 and two rotary pairs;
//   v pass:       tokens are the A operand: a lane owns one feature and, per 16 tokens, exactly the 8 keys of one 16-byte group
//                 of the V^T layout (keys permuted inside 16-groups as k_attn_bf16_v5 reads them) = one 16-byte store.
// F16: the outputs are fp16 instead of bf16 (GN_PREC_F16X2_F16_ATTN, the reference's CUDA arithmetic); a value outside fp16's range raises a.ovf
// NP = 2 (developer knob 27, an accuracy experiment): the x_m . w_h product is dropped -- the outputs are rounded to 16 bits anyway
template <bool CROSS, bool F16 = false, int NP = 3>
__global__ __launch_bounds__(512) void k_qkv(QkvArgs a) {
  constexpr int NQK = CROSS ? kDim : 2 * kDim;      // q | k (or qk) features
  constexpr int NPASS = CROSS ? 2 : 3;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[8 * KT + (CROSS ? 0 : 32768)];   // token tile (128 KB) + the self block's rotary entries
  const int n_work = a.tiles != nullptr ? a.tiles[0] : a.T / TM;     // persistent form: see k_ffn128
#pragma unroll 1
  for (int work = blockIdx.x; work < n_work; work += gridDim.x) {
  // (the thread index is opaque per iteration: everything derived from it -- lane, wave, every address -- is then recomputed inside the body instead of
  // being hoisted out of the loop, where it would live across the whole tile and push the kernel over its register budget)
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hh = lane >> 5, ql = lane & 31;
  const int bm = (a.tiles != nullptr ? a.tiles[kTileListBase + work] : work) * TM;
  long long ts[8];
  auto stamp = [&](int k) __attribute__((always_inline)) { if (a.dbg_ts) ts[k] = (long long)__builtin_amdgcn_s_memtime(); };
  stamp(0);

  // ---- token tile: thread -> (row, 16-byte chunk) of every k-tile; chunk c of row r sits at position c ^ swz(r)
  {
    const int srow = tid >> 3, schunk = tid & 7;     // rows srow, srow + 64
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int row = srow + 64 * half;
      const unsigned char* xsrc = reinterpret_cast<const unsigned char*>(a.xp) + (size_t)(bm + row) * 1024 + schunk * 16;
      const int sdst = row * 128 + ((schunk ^ swz(row)) * 16);
      uint4 xt[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) xt[q] = *reinterpret_cast<const uint4*>(xsrc + q * 128);
#pragma unroll
      for (int q = 0; q < 8; ++q) *reinterpret_cast<uint4*>(smem + q * KT + sdst) = xt[q];
    }
  }
  // token fragments of k-step ks: lane (row 32 j + ql, hh), term pl -> chunk 4 (ks & 1) + 2 pl + hh of slot ks >> 1
  int brow[NJ], bsw[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) { brow[j] = (32 * j + ql) * 128; bsw[j] = swz(32 * j + ql); }
  f16x8 fb[2][NJ][2];
  auto read_b = [&](int buf, int ks) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
        fb[buf][j][pl] = *reinterpret_cast<const f16x8*>(smem + (ks >> 1) * KT + brow[j] + (((4 * (ks & 1) + 2 * pl + hh) ^ bsw[j]) * 16));
  };
  // weight fragments: block ((tile * 16 + kstep) * 2 + term) of 1 KB, lane l -> bytes [16 l, 16 l + 16); a ring of k-steps
  const uint4* const wf = reinterpret_cast<const uint4*>(a.wf) + lane;
  constexpr int RA = CROSS ? 8 : 6;     // k-steps in flight (2 KB per wave each); the rotary pass needs the registers
  f16x8 fa[RA][2];
  const float ascale = a.acc_scale;
  const int slot = bm / a.npad, i0 = bm - slot * a.npad;
  __syncthreads();
  stamp(1);

  // (cos 2 fg, cos 2 fg + 1, sin 2 fg, sin 2 fg + 1) of this lane's tokens: rot4[fg][token], consecutive lanes = consecutive tokens.
  // The k pass rotates the same (token, feature-in-head) positions as the q pass: the 16 table entries of a lane are fetched once and
  // wait out the k pass's k-loop in the 32 KB of LDS the token tile leaves free -- 128 KB less to pull per workgroup, and no memory
  // latency in front of the k epilogue (~5 k cycles).  (Requesting them under the q pass's k-loop as well was measured: the loop's
  // in-order waits then stall on the table, 10 k -> 18 k cycles.)  Waves w, w + 2, w + 4, w + 6 hold the SAME entries (32 w mod 64): they all write the copy of
  // set w & 1 ([set][entry][lane], 16 bytes each; identical bytes, so the race is benign) and a wave reads back what it wrote itself.
  f32x4 rot[NJ][4];
  f32x4* const rot_park = reinterpret_cast<f32x4*>(smem + 8 * KT) + (wave & 1) * 16 * 64 + lane;
#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
    const bool vpass = pass == NPASS - 1;
    const int tile = 8 * pass + wave;            // feature tile of this wave in this pass
    auto load_a = [&](int slot_, int ks) __attribute__((always_inline)) {
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) fa[slot_][pl] = __builtin_bit_cast(f16x8, wf[(size_t)((tile * 16 + ks) * 2 + pl) * 64]);
    };
#pragma unroll
    for (int q = 0; q < RA; ++q) load_a(q, q);
    f32x16 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    read_b(0, 0);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int rs = ks % RA, cb = ks & 1;
      if (ks + 1 < 16) read_b(cb ^ 1, ks + 1);
      // products: W_m X_h, W_h X_m, W_h X_h (small terms first); the accumulators are visited round-robin
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (NP == 2 && p == 1) continue;
          const f16x8 w = fa[rs][p == 0 ? 1 : 0], x = fb[cb][j][p == 1 ? 1 : 0];
          acc[j] = vpass ? __builtin_amdgcn_mfma_f32_32x32x16_f16(x, w, acc[j], 0, 0, 0)     // rows = tokens, columns = features
                         : __builtin_amdgcn_mfma_f32_32x32x16_f16(w, x, acc[j], 0, 0, 0);    // rows = features, columns = tokens
        }
      if (ks + RA < 16) load_a(rs, ks + RA);
      __builtin_amdgcn_sched_barrier(0);
    }
    stamp(2 + 2 * pass);
    float amax = 0.f;   // F16: max |value| this lane stores in this pass (scoped to the epilogue: the k-loop has no register to spare)
    if (!vpass) {
      // register r of tile j <-> feature 32 tile + (r & 3) + 8 (r >> 2) + 4 hh, token 32 j + ql.  Every table entry and bias this lane
      // needs is requested FIRST, back to back (the weight ring and the token fragments are dead: the registers are there): the
      // pass pays one memory latency, not one per token tile (measured: 25 k -> ? cycles per pass)
      f32x4 bias4[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) bias4[g] = *reinterpret_cast<const f32x4*>(a.bias + 32 * tile + 8 * g + 4 * hh);
      if (!CROSS && pass == 0) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int fg = ((32 * wave + 8 * g + 4 * hh) & 63) >> 2;
            rot[j][g] = *reinterpret_cast<const f32x4*>(a.rot4 + ((size_t)fg * a.rot_stride + (size_t)(bm + 32 * j + ql)) * 4);
          }
      }
      if (!CROSS && pass == 1) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) rot[j][g] = rot_park[(4 * j + g) * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const size_t row = (size_t)(bm + 32 * j + ql);
        uint2 pk[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v = {acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
          v = v * ascale;
          v += bias4[g];
          if (CROSS) {
            v *= a.scale;
          } else {
            f32x4 o;
            o.x = v.x * rot[j][g].x + (-v.y) * rot[j][g].z;
            o.y = v.y * rot[j][g].x + v.x * rot[j][g].z;
            o.z = v.z * rot[j][g].y + (-v.w) * rot[j][g].w;
            o.w = v.w * rot[j][g].y + v.z * rot[j][g].w;
            v = o;
            if (pass == 0) v *= a.qscale;
          }
          pk[g].x = pack16<F16>(v.x, v.y);
          pk[g].y = pack16<F16>(v.z, v.w);
          if (F16) { ovf_track(amax, v.x, v.y); ovf_track(amax, v.z, v.w); }
        }
        // the two half-waves hold interleaved groups of 4 features (hh = 0: 8 g .. 8 g + 3, hh = 1: 8 g + 4 .. 8 g + 7): they trade every
        // other group, so that a lane stores 8 consecutive features (16 bytes) -- half as many scattered stores
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          const uint2 give = hh ? pk[2 * gp] : pk[2 * gp + 1];
          uint2 got;
          got.x = __shfl_xor(give.x, 32); got.y = __shfl_xor(give.y, 32);
          const uint2 own = hh ? pk[2 * gp + 1] : pk[2 * gp];
          const uint4 out = hh ? make_uint4(got.x, got.y, own.x, own.y) : make_uint4(own.x, own.y, got.x, got.y);
          // hh = 0: features 16 gp + {0..3 own, 4..7 partner's group 2 gp};  hh = 1: features 16 gp + 8 + {0..3 partner's group 2 gp + 1, 4..7 own}
          *reinterpret_cast<uint4*>(a.qkb + row * a.ldyb + 32 * tile + 16 * gp + 8 * hh) = out;
        }
      }
      if (!CROSS && pass == 0) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) rot_park[(4 * j + g) * 64] = rot[j][g];
      }
    } else {
      // register r of tile j <-> token 32 j + (r & 3) + 8 (r >> 2) + 4 hh, feature 32 wave + ql of the V panel; registers 8 m .. 8 m + 7
      // are the keys 16 m + 4 hh + {0..3, 8..11}: group 2 m + hh of the permuted V^T layout
      const int d = 32 * wave + ql;
      const float bias = a.bias[NQK + d];
      uint16_t* const dst = a.vt + (((size_t)slot * kHeads + (d >> 6)) * kHeadDim + (d & 63)) * a.npad + i0;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          unsigned int w4[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lo = acc[j][8 * m + 2 * e] * ascale + bias;
            const float hi = acc[j][8 * m + 2 * e + 1] * ascale + bias;
            w4[e] = pack16<F16>(lo, hi);
            if (F16) ovf_track(amax, lo, hi);
          }
          if (!(a.vt_perm & 2)) *reinterpret_cast<uint4*>(dst + 32 * j + 8 * (2 * m + hh)) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
        }
    }
    if (F16) ovf_commit(a.ovf, amax);
    stamp(3 + 2 * pass);
  }
  if (a.dbg_ts && lane == 0) {   // developer: s_memtime phase stamps per wave
    for (int k = 0; k < 8; ++k) a.dbg_ts[((size_t)blockIdx.x * 8 + wave) * 8 + k] = ts[k];
  }
  __syncthreads();     // the token tile is free again
  }   // work
}
}  // namespace

// Partial products per block of the attention input projections.  q, k and v leave this kernel rounded to 16 bits (fp16: 11 significant bits,
// bf16: 8), so the x_m . w_h product -- a 2^-12 relative correction of every x -- is below their rounding: with TWO products (x_h w_h + x_h w_m)
// the low-margin index-mismatch count against the oracle is 2 / 7173 (three products: 5 / 7173), mid-margin 0 / 3954 both ways, the 32 bench
// pairs are index-identical, and the launches drop from 87 / 62 us to 72 / 53 us (tools/qkv2_study.py, profiles/r04_qkv2_study.json).
// Developer knob 27 = 3 restores the third product.  (The block tail and the match head keep all three: their outputs are f32-accurate values.)
int device_cu_count() {
  // per device id, filled on first use (hipGetDeviceProperties on every launch was slow; a process-wide static was wrong with several devices)
  static int cache[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int n = __atomic_load_n(&cache[dev], __ATOMIC_RELAXED);
  if (n > 0) return n;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  __atomic_store_n(&cache[dev], n, __ATOMIC_RELAXED);
  return n;
}
void launch_qkv(const QkvArgs& a, bool cross, hipStream_t s) {
  const int ncu = a.ncu > 0 ? a.ncu : device_cu_count();
  const dim3 grid(a.tiles != nullptr ? std::min(a.T / TM, ncu) : a.T / TM), block(512);
  const int sel = (a.half_fmt ? 4 : 0) | (cross ? 2 : 0) | (a.products == 2 ? 1 : 0);
  switch (sel) {
    case 7: hipLaunchKernelGGL((k_qkv<true, true, 2>), grid, block, 0, s, a); g_last_kernel = "k_qkv<true, true, 2>"; break;
    case 6: hipLaunchKernelGGL((k_qkv<true, true, 3>), grid, block, 0, s, a); g_last_kernel = "k_qkv<true, true, 3>"; break;
    case 5: hipLaunchKernelGGL((k_qkv<false, true, 2>), grid, block, 0, s, a); g_last_kernel = "k_qkv<false, true, 2>"; break;
    case 4: hipLaunchKernelGGL((k_qkv<false, true, 3>), grid, block, 0, s, a); g_last_kernel = "k_qkv<false, true, 3>"; break;
    case 3: hipLaunchKernelGGL((k_qkv<true, false, 2>), grid, block, 0, s, a); g_last_kernel = "k_qkv<true, false, 2>"; break;
    case 2: hipLaunchKernelGGL((k_qkv<true, false, 3>), grid, block, 0, s, a); g_last_kernel = "k_qkv<true, false, 3>"; break;
    case 1: hipLaunchKernelGGL((k_qkv<false, false, 2>), grid, block, 0, s, a); g_last_kernel = "k_qkv<false, false, 2>"; break;
    default: hipLaunchKernelGGL((k_qkv<false, false, 3>), grid, block, 0, s, a); g_last_kernel = "k_qkv<false, false, 3>"; break;
  }
}

}  // namespace gn
