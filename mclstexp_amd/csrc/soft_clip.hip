// Elementwise middle of the baseline's soft-target contrastive loss (/root/reference/baselines/Bleep/models.py:34-43,
// 66-76, 228-234) between its GEMMs: with S = E_s E_i^T / T, Tg = softmax_rows(k A), the row / column log-sum-exp of S and the
// column sums of Tg given,
//
//     lsum_ij = (S_ij - lse_row_i) + (S_ij - lse_col_j)
//     loss    = -c sum_ij Tg_ij lsum_ij                              c = 1 / (2B)
//     dS_ij   = c (exp(S_ij - lse_row_i) + exp(S_ij - lse_col_j) tcol_j - 2 Tg_ij)
//     dA_ij   = -c lsum_ij                                           (d loss / d Tg, before the softmax backward)
//
// in ONE pass over the B x B matrices (the torch expression chain was a dozen elementwise launches), and afterwards the
// symmetrisation dsym = (dA + dA^T) / 2 of the softmax backward's output through LDS tiles.  Per-row loss partials are summed
// in a fixed order (strided lanes, wave butterfly, waves in order): deterministic.
#include "common.h"
#include <math.h>

namespace {

__global__ __launch_bounds__(256) void soft_clip_mid_kernel(const float* __restrict__ S, const float* __restrict__ Tg,
                                                            const float* __restrict__ lse_row, const float* __restrict__ lse_col,
                                                            const float* __restrict__ tcol, int B, float c, float* __restrict__ dS,
                                                            float* __restrict__ dA, float* __restrict__ loss_rows) {
  __shared__ float red[4];
  const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float lr = lse_row[i];
  const long long base = (long long)i * B;
  float acc = 0.0f;
  for (int j = tid; j < B; j += 256) {
    const float s = S[base + j], t = Tg[base + j];
    const float ls = s - lr, lc = s - lse_col[j];
    const float lsum = ls + lc;
    acc = fmaf(t, lsum, acc);
    dS[base + j] = (expf(ls) + expf(lc) * tcol[j] - 2.0f * t) * c;
    dA[base + j] = -c * lsum;
  }
  acc = wave_sum(acc);
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (tid == 0) loss_rows[i] = -c * ((red[0] + red[1]) + (red[2] + red[3]));
}

// out = (a + a^T) / 2, 32 x 32 tiles through LDS (both reads coalesced)
__global__ __launch_bounds__(256) void symmetrize_kernel(const float* __restrict__ a, int B, float* __restrict__ out) {
  __shared__ float t[32][33];
  const int bi = blockIdx.y * 32, bj = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;             // 32 x 8
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int i = bj + r, j = bi + tx;                                // tile (bj, bi) of a: read rows of the TRANSPOSED block
    t[r][tx] = (i < B && j < B) ? a[(long long)i * B + j] : 0.0f;
  }
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int i = bi + r, j = bj + tx;
    if (i < B && j < B) out[(long long)i * B + j] = 0.5f * (a[(long long)i * B + j] + t[tx][r]);
  }
}

}  // namespace

extern "C" int mcl_soft_clip_mid(const float* S, const float* Tg, const float* lse_row, const float* lse_col, const float* tcol,
                                 int32_t B, float c, float* dS, float* dA, float* loss_rows, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!S || !Tg || !lse_row || !lse_col || !tcol || !dS || !dA || !loss_rows || B <= 0) return MCL_EINVAL;
  hipLaunchKernelGGL(soft_clip_mid_kernel, dim3(B), dim3(256), 0, mcl_stream(stream), S, Tg, lse_row, lse_col, tcol, B, c, dS, dA,
                     loss_rows);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_symmetrize(const float* a, int32_t B, float* out, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!a || !out || a == out || B <= 0) return MCL_EINVAL;
  hipLaunchKernelGGL(symmetrize_kernel, dim3((B + 31) / 32, (B + 31) / 32), dim3(256), 0, mcl_stream(stream), a, B, out);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}
