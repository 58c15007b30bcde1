// K8: symmetric InfoNCE over a logits strip (model.py:242-247), closed form:
//   loss = 1/(2B) [ sum_i (LSE_j S_ij - S_ii) + sum_j (LSE_i S_ij - S_jj) ]
//   dS   = (softmax_rows(S) + softmax_cols(S) - 2I) / (2B)
// The logits reach +-100 (LayerNorm-ed, un-normalised embeddings), so every LSE is max-subtracted.
// Row LSEs: one wave per row with shuffle reductions.  Column LSEs: two-stage -- a 64-column x
// 4-row-group workgroup keeps an online (max, sum) per column with coalesced row reads, then merges
// the four partials through LDS.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void row_lse_kernel(const float* __restrict__ S, long long ld, int R, int C,
                                                      float* __restrict__ out) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= R) return;
  const float* r = S + (long long)row * ld;
  float mx = -INFINITY;
  for (int c = lane; c < C; c += 64) mx = fmaxf(mx, r[c]);
  mx = wave_max(mx);
  float sum = 0.0f;
  for (int c = lane; c < C; c += 64) sum += expf(r[c] - mx);
  sum = wave_sum(sum);
  if (lane == 0) out[row] = mx + logf(sum);
}

__global__ __launch_bounds__(256) void col_lse_kernel(const float* __restrict__ S, long long ld, int R, int C,
                                                      float* __restrict__ out) {
  __shared__ float sm[4][64], sl[4][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float m = -INFINITY, l = 0.0f;
  if (c < C) {
    // eight rows' loads in flight at a time (clamped row index, the surplus skipped), consumed in the same order: one load per
    // trip of the data-dependent loop below is one memory round trip per trip -- 32 in series at B = 128, 11 of the kernel's 12 us
    for (int r0 = grp; r0 < R; r0 += 32) {
      float v8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v8[u] = S[(long long)min(r0 + 4 * u, R - 1) * ld + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (r0 + 4 * u >= R) continue;
        const float v = v8[u];
        if (v > m) {
          l = l * expf(m - v) + 1.0f;  // exp(-inf)=0 on the first element
          m = v;
        } else {
          l += expf(v - m);
        }
      }
    }
  }
  sm[grp][lane] = m;
  sl[grp][lane] = l;
  __syncthreads();
  if (grp == 0 && c < C) {
    float M = fmaxf(fmaxf(sm[0][lane], sm[1][lane]), fmaxf(sm[2][lane], sm[3][lane]));
    float L = 0.0f;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      if (sl[g][lane] > 0.0f) L += sl[g][lane] * expf(sm[g][lane] - M);
    out[c] = M + logf(L);
  }
}

// single workgroup; deterministic tree over the (few thousand at most) target entries
__global__ __launch_bounds__(256) void loss_kernel(const float* __restrict__ S, long long ld,
                                                   const float* __restrict__ row_lse,
                                                   const float* __restrict__ col_lse, int di0, int dj0, int n_diag,
                                                   int use_rows, int use_cols, float* __restrict__ loss_sum) {
  __shared__ float sr[256], sc[256];
  float ar = 0.0f, ac = 0.0f;
  for (int t = threadIdx.x; t < n_diag; t += 256) {
    const float d = S[(long long)(di0 + t) * ld + (dj0 + t)];
    if (use_rows) ar += row_lse[di0 + t] - d;
    if (use_cols) ac += col_lse[dj0 + t] - d;
  }
  sr[threadIdx.x] = ar;
  sc[threadIdx.x] = ac;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      sr[threadIdx.x] += sr[threadIdx.x + s];
      sc[threadIdx.x] += sc[threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    loss_sum[0] += sr[0];
    loss_sum[1] += sc[0];
  }
}

// the same two sums as loss_kernel (same thread mapping and tree), finished in the launch: (sum_r + sum_c) / denom
__global__ __launch_bounds__(256) void loss_mean_kernel(const float* __restrict__ S, long long ld,
                                                        const float* __restrict__ row_lse,
                                                        const float* __restrict__ col_lse, int n, float denom,
                                                        float* __restrict__ loss_out) {
  __shared__ float sr[256], sc[256];
  float ar = 0.0f, ac = 0.0f;
  for (int t = threadIdx.x; t < n; t += 256) {
    const float d = S[(long long)t * ld + t];
    ar += row_lse[t] - d;
    ac += col_lse[t] - d;
  }
  sr[threadIdx.x] = ar;
  sc[threadIdx.x] = ac;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      sr[threadIdx.x] += sr[threadIdx.x + s];
      sc[threadIdx.x] += sc[threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) loss_out[0] = (sr[0] + sc[0]) / denom;
}

__global__ __launch_bounds__(256) void dlogits_kernel(const float* __restrict__ S, long long ldS,
                                                      const float* __restrict__ row_lse,
                                                      const float* __restrict__ col_lse, int R, int C, int row0,
                                                      int col0, float coef, float* __restrict__ dS, long long lddS) {
  const int i = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= C) return;
  const float s = S[(long long)i * ldS + j];
  float w = expf(s - row_lse[i]) + expf(s - col_lse[j]);
  if (row0 + i == col0 + j) w -= 2.0f;
  dS[(long long)i * lddS + j] = coef * w;
}

}  // namespace

extern "C" int mcl_infonce_lse(const float* S, int64_t ldS, int32_t R, int32_t C, float* row_lse, float* col_lse,
                               mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!S || R <= 0 || C <= 0 || (!row_lse && !col_lse)) return MCL_EINVAL;
  hipStream_t st = mcl_stream(stream);
  if (row_lse) hipLaunchKernelGGL(row_lse_kernel, dim3((R + 3) / 4), dim3(256), 0, st, S, ldS, R, C, row_lse);
  if (col_lse) hipLaunchKernelGGL(col_lse_kernel, dim3((C + 63) / 64), dim3(256), 0, st, S, ldS, R, C, col_lse);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_infonce_loss(const float* S, int64_t ldS, const float* row_lse, const float* col_lse, int32_t di0,
                                int32_t dj0, int32_t n_diag, int32_t use_rows, int32_t use_cols, float* loss_sum,
                                mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!S || !loss_sum || n_diag < 0 || di0 < 0 || dj0 < 0) return MCL_EINVAL;
  if ((use_rows && !row_lse) || (use_cols && !col_lse)) return MCL_EINVAL;
  hipLaunchKernelGGL(loss_kernel, dim3(1), dim3(256), 0, mcl_stream(stream), S, ldS, row_lse, col_lse, di0, dj0,
                     n_diag, use_rows, use_cols, loss_sum);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_infonce_loss_mean(const float* S, int64_t ldS, const float* row_lse, const float* col_lse, int32_t n,
                                     float denom, float* loss_out, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!S || !row_lse || !col_lse || !loss_out || n <= 0 || denom == 0.0f) return MCL_EINVAL;
  hipLaunchKernelGGL(loss_mean_kernel, dim3(1), dim3(256), 0, mcl_stream(stream), S, ldS, row_lse, col_lse, n, denom,
                     loss_out);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_infonce_dlogits(const float* S, int64_t ldS, const float* row_lse, const float* col_lse, int32_t R,
                                   int32_t C, int32_t row0, int32_t col0, float coef, float* dS, int64_t lddS,
                                   mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!S || !row_lse || !col_lse || !dS || R <= 0 || C <= 0) return MCL_EINVAL;
  hipLaunchKernelGGL(dlogits_kernel, dim3((C + 255) / 256, R), dim3(256), 0, mcl_stream(stream), S, ldS, row_lse,
                     col_lse, R, C, row0, col0, coef, dS, lddS);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}
