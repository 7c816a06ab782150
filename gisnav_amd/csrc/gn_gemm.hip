// f32 projection / FFN / similarity GEMMs of the LightGlue matcher on the gfx950 f32 MFMA
// (v_mfma_f32_32x32x2_f32: exact f32 fma chain, 157 TF peak).
//
//   Y[M][N] = A[M][K] * W[N][K]^T (+ bias) (+ epilogue)          "NT" GEMM, K contiguous in both
//
// Stands in for the nn.Linear calls inside kornia's LightGlue (input_proj, Wqkv, out_proj, to_qk,
// to_v, to_out, ffn.0, ffn.3, final_proj) and the `einsum("bmd,bnd->bmn")` similarity of
// MatchAssignment -- all reached from ros/gisnav/gisnav/core/pose_node.py:285-287.
//
// Tiling: 128x128 block tile, 4 waves (2x2), each wave a 64x64 sub-tile = 2x2 MFMA tiles of 32x32,
// BK = 32 staged through LDS with a 36-float row stride (conflict-free ds_read_b128 / ds_write_b128).
// The MFMA's k index is free to permute as long as A and B agree, so each lane fetches FOUR
// consecutive k with one ds_read_b128 and feeds them to four back-to-back MFMAs: per 8-deep k chunk a
// wave issues 4 LDS reads for 16 MFMAs (1024 matrix-pipe cycles), leaving the LDS idle and the
// matrix pipe saturated from one wave per SIMD.
#include "gn_common.h"

namespace gn {

namespace {
constexpr int BM = 128, BN = 128, BK = 32, LS = 36;

__device__ __forceinline__ void load_tile(const float* asrc, size_t astr, const float* wsrc, size_t wstr,
                                          f32x4 (&ra)[4], f32x4 (&rb)[4]) {
  ra[0] = *reinterpret_cast<const f32x4*>(asrc);
  ra[1] = *reinterpret_cast<const f32x4*>(asrc + astr);
  ra[2] = *reinterpret_cast<const f32x4*>(asrc + 2 * astr);
  ra[3] = *reinterpret_cast<const f32x4*>(asrc + 3 * astr);
  rb[0] = *reinterpret_cast<const f32x4*>(wsrc);
  rb[1] = *reinterpret_cast<const f32x4*>(wsrc + wstr);
  rb[2] = *reinterpret_cast<const f32x4*>(wsrc + 2 * wstr);
  rb[3] = *reinterpret_cast<const f32x4*>(wsrc + 3 * wstr);
}

template <int EPI>
__global__ __launch_bounds__(256) void k_gemm_f32(GemmArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LS];
  float* As = smem;
  float* Bs = smem + BM * LS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int bm = blockIdx.y * BM, bn = blockIdx.x * BN;
  const float* A = a.A + (long long)blockIdx.z * a.strideA;
  const float* W = a.W + (long long)blockIdx.z * a.strideW;
  float* Y = a.Y + (long long)blockIdx.z * a.strideY;

  const int lrow = tid >> 3, lcol = (tid & 7) * 4;
  f32x4 ra[4], rb[4];
  // kernel arguments are copied to locals so that nothing takes the address of `a` (which would
  // spill the whole argument struct to scratch)
  const float* const A2 = a.A2;
  const int lda = a.lda, lda2 = a.lda2, ldw = a.ldw, K1 = a.K1, K = a.K;
  const float* const Arow = A + (size_t)(bm + lrow) * lda + lcol;
  const float* const A2row = A2 ? A2 + (size_t)(bm + lrow) * lda2 + lcol - K1 : nullptr;
  const float* const Wrow = W + (size_t)(bn + lrow) * ldw + lcol;

#define GN_LOAD_TILE(k0)                                                                          \
  {                                                                                               \
    const bool second = (A2 != nullptr) && ((k0) + lcol >= K1);                                   \
    const float* asrc = second ? A2row + (k0) : Arow + (k0);                                      \
    const size_t astr = second ? (size_t)32 * lda2 : (size_t)32 * lda;                            \
    load_tile(asrc, astr, Wrow + (k0), (size_t)32 * ldw, ra, rb);                                 \
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int arow = wr * 64 + (lane & 31), brow = wc * 64 + (lane & 31), khalf = (lane >> 5) * 4;

  GN_LOAD_TILE(0);
  for (int k0 = 0; k0 < K; k0 += BK) {
    *reinterpret_cast<f32x4*>(&As[(lrow + 0) * LS + lcol]) = ra[0];
    *reinterpret_cast<f32x4*>(&As[(lrow + 32) * LS + lcol]) = ra[1];
    *reinterpret_cast<f32x4*>(&As[(lrow + 64) * LS + lcol]) = ra[2];
    *reinterpret_cast<f32x4*>(&As[(lrow + 96) * LS + lcol]) = ra[3];
    *reinterpret_cast<f32x4*>(&Bs[(lrow + 0) * LS + lcol]) = rb[0];
    *reinterpret_cast<f32x4*>(&Bs[(lrow + 32) * LS + lcol]) = rb[1];
    *reinterpret_cast<f32x4*>(&Bs[(lrow + 64) * LS + lcol]) = rb[2];
    *reinterpret_cast<f32x4*>(&Bs[(lrow + 96) * LS + lcol]) = rb[3];
    __syncthreads();
    if (k0 + BK < K) GN_LOAD_TILE(k0 + BK);  // register prefetch of the next tile under the MFMAs
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      f32x4 af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = *reinterpret_cast<const f32x4*>(&As[(arow + 32 * i) * LS + kc * 8 + khalf]);
        bf[i] = *reinterpret_cast<const f32x4*>(&Bs[(brow + 32 * i) * LS + kc * 8 + khalf]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
  }

#undef GN_LOAD_TILE
  // Epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = bn + wc * 64 + j * 32 + (lane & 31);
    const float bias = (EPI != EPI_PLAIN && a.bias != nullptr) ? a.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = bm + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float v = acc[i][j][r] + bias;
        if (EPI == EPI_SCALE_COLS) {
          if (col < a.scale_cols) v *= a.scale;
        } else if (EPI == EPI_ROTARY) {
          // apply_cached_rotary_emb: t * cos + rotate_half(t) * sin on interleaved pairs (2i, 2i+1);
          // the pair partner lives in the neighbouring lane (col ^ 1).
          const float partner = __shfl_xor(v, 1);
          if (col < a.rot_cols) {
            const int f = (col & 63) >> 1;
            const float c = a.cos_t[(size_t)row * kFreq + f];
            const float s = a.sin_t[(size_t)row * kFreq + f];
            const float rot = (col & 1) ? partner : -partner;
            v = v * c + rot * s;
          }
        } else if (EPI == EPI_RESIDUAL) {
          v += a.resid[(size_t)row * a.ldr + col];
        }
        Y[(size_t)row * a.ldy + col] = v;
      }
    }
  }
}
}  // namespace

void launch_gemm_f32(int epi, const GemmArgs& a, int batch, hipStream_t s) {
  dim3 grid(a.N / BN, a.M / BM, batch), block(256);
  switch (epi) {
    case EPI_BIAS: hipLaunchKernelGGL(k_gemm_f32<EPI_BIAS>, grid, block, 0, s, a); break;
    case EPI_SCALE_COLS: hipLaunchKernelGGL(k_gemm_f32<EPI_SCALE_COLS>, grid, block, 0, s, a); break;
    case EPI_ROTARY: hipLaunchKernelGGL(k_gemm_f32<EPI_ROTARY>, grid, block, 0, s, a); break;
    case EPI_RESIDUAL: hipLaunchKernelGGL(k_gemm_f32<EPI_RESIDUAL>, grid, block, 0, s, a); break;
    default: hipLaunchKernelGGL(k_gemm_f32<EPI_PLAIN>, grid, block, 0, s, a); break;
  }
}

}  // namespace gn
