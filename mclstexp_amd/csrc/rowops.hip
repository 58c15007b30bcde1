// Row-wise HBM-bound kernels: LayerNorm fwd/bwd (K2), attention softmax fwd/bwd (part of K4),
// column sums (bias gradients).  One 64-lane wave owns one row; reductions are wave shuffles
// (no LDS, no barriers); a 256-thread workgroup processes 4 rows.  Column reductions run one
// thread per column so that consecutive lanes read consecutive addresses of each row.
#include "common.h"

namespace {

constexpr int ROWS_PER_BLOCK = 4;

__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, long long ldx,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y,
                                                            long long ldy, float* __restrict__ mean_out,
                                                            float* __restrict__ rstd_out, int rows, int cols,
                                                            float eps) {
  const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (long long)row * ldx;
  float s = 0.0f;
  for (int c = lane; c < cols; c += 64) s += xr[c];
  const float mean = wave_sum(s) / (float)cols;
  float q = 0.0f;
  for (int c = lane; c < cols; c += 64) {
    const float d = xr[c] - mean;
    q += d * d;
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)cols + eps);
  float* yr = y + (long long)row * ldy;
  for (int c = lane; c < cols; c += 64) yr[c] = (xr[c] - mean) * rstd * gamma[c] + beta[c];
  if (lane == 0) {
    mean_out[row] = mean;
    rstd_out[row] = rstd;
  }
}

__global__ __launch_bounds__(256) void layernorm_bwd_dx_kernel(const float* __restrict__ dy, long long lddy,
                                                               const float* __restrict__ x, long long ldx,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, const float* dx_add,
                                                               long long ldadd, float* dx, long long lddx, int rows,
                                                               int cols) {
  const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* dyr = dy + (long long)row * lddy;
  const float* xr = x + (long long)row * ldx;
  const float mu = mean[row], rs = rstd[row];
  float s1 = 0.0f, s2 = 0.0f;
  for (int c = lane; c < cols; c += 64) {
    const float g = dyr[c] * gamma[c];
    s1 += g;
    s2 += g * (xr[c] - mu) * rs;
  }
  s1 = wave_sum(s1) / (float)cols;
  s2 = wave_sum(s2) / (float)cols;
  float* dxr = dx + (long long)row * lddx;
  const float* addr = dx_add ? dx_add + (long long)row * ldadd : nullptr;
  for (int c = lane; c < cols; c += 64) {
    const float xh = (xr[c] - mu) * rs;
    const float v = rs * (dyr[c] * gamma[c] - s1 - xh * s2);
    dxr[c] = addr ? addr[c] + v : v;
  }
}

// dgamma[c] = sum_r dy[r,c]*xhat[r,c]; dbeta[c] = sum_r dy[r,c].  Block = 64 columns x 4 row-groups.
__global__ __launch_bounds__(256) void layernorm_bwd_dgb_kernel(const float* __restrict__ dy, long long lddy,
                                                                const float* __restrict__ x, long long ldx,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ rstd,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                int rows, int cols, int accumulate) {
  __shared__ float sg[4][64], sb[4][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float ag = 0.0f, ab = 0.0f;
  if (c < cols) {
    for (int r = grp; r < rows; r += 4) {
      const float d = dy[(long long)r * lddy + c];
      ag += d * (x[(long long)r * ldx + c] - mean[r]) * rstd[r];
      ab += d;
    }
  }
  sg[grp][lane] = ag;
  sb[grp][lane] = ab;
  __syncthreads();
  if (grp == 0 && c < cols) {
    const float g = (sg[0][lane] + sg[1][lane]) + (sg[2][lane] + sg[3][lane]);
    const float b = (sb[0][lane] + sb[1][lane]) + (sb[2][lane] + sb[3][lane]);
    dgamma[c] = accumulate ? dgamma[c] + g : g;          // += : straight into the parameters' .grad (no AccumulateGrad add)
    dbeta[c] = accumulate ? dbeta[c] + b : b;
  }
}

__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, long long ldx,
                                                     float* __restrict__ out, int rows, int cols, int accumulate) {
  __shared__ float sm[4][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float a = 0.0f;
  if (c < cols)
    for (int r = grp; r < rows; r += 4) a += x[(long long)r * ldx + c];
  sm[grp][lane] = a;
  __syncthreads();
  if (grp == 0 && c < cols) {
    const float v = (sm[0][lane] + sm[1][lane]) + (sm[2][lane] + sm[3][lane]);
    out[c] = accumulate ? out[c] + v : v;
  }
}

// Up to six column reductions over the same `rows` rows as one launch (a Transformer layer's bias gradients and LayerNorm
// parameter gradients: blockIdx.y picks the problem).  Kind "sum": out0[c] (+)= sum_r a[r, c].  Kind "LayerNorm" (x != NULL):
// out0[c] (+)= sum_r a[r, c] * xhat[r, c], out1[c] (+)= sum_r a[r, c].  Per problem the code and summation order of colsum_kernel /
// layernorm_bwd_dgb_kernel -> bit-identical to the separate launches.
constexpr int COLRED_MAX = 6;
struct ColredGroup {
  const float* a[COLRED_MAX]; long long lda[COLRED_MAX];
  const float* x[COLRED_MAX]; long long ldx[COLRED_MAX];
  const float* mean[COLRED_MAX]; const float* rstd[COLRED_MAX];
  float* out0[COLRED_MAX]; float* out1[COLRED_MAX];
  int cols[COLRED_MAX]; int accumulate[COLRED_MAX];
  int rows;
};
__global__ __launch_bounds__(256) void colred_group_kernel(const ColredGroup g) {
  __shared__ float sg[4][64], sb[4][64];
  const int q = blockIdx.y;
  const int cols = g.cols[q];
  if ((int)blockIdx.x * 64 >= cols) return;
  const float* __restrict__ a = g.a[q];
  const float* __restrict__ x = g.x[q];
  const long long lda = g.lda[q], ldx = g.ldx[q];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float ag = 0.0f, ab = 0.0f;
  if (c < cols) {
    if (x) {
      const float* __restrict__ mean = g.mean[q];
      const float* __restrict__ rstd = g.rstd[q];
      for (int r = grp; r < g.rows; r += 4) {
        const float d = a[(long long)r * lda + c];
        ag += d * (x[(long long)r * ldx + c] - mean[r]) * rstd[r];
        ab += d;
      }
    } else {
      for (int r = grp; r < g.rows; r += 4) ag += a[(long long)r * lda + c];
    }
  }
  sg[grp][lane] = ag;
  sb[grp][lane] = ab;
  __syncthreads();
  if (grp == 0 && c < cols) {
    const float v = (sg[0][lane] + sg[1][lane]) + (sg[2][lane] + sg[3][lane]);
    float* o0 = g.out0[q];
    o0[c] = g.accumulate[q] ? o0[c] + v : v;
    if (x) {
      const float w = (sb[0][lane] + sb[1][lane]) + (sb[2][lane] + sb[3][lane]);
      float* o1 = g.out1[q];
      o1[c] = g.accumulate[q] ? o1[c] + w : w;
    }
  }
}

// The same column reductions over MANY rows (the fp32 ViT: 6 400 - 25 216 token rows; the one-block-per-64-columns forms above
// are written for the spot branch's 128 rows and serialise everything else on a dozen workgroups): blockIdx.y owns RCHUNK rows
// and writes a partial per column; a second launch adds the partials in chunk order (deterministic).  HAS_X: the LayerNorm
// parameter gradients (two outputs), else plain column sums.
constexpr int RCHUNK = 128;
template <bool HAS_X>
__global__ __launch_bounds__(256) void colred_chunk_kernel(const float* __restrict__ dy, long long lddy, const float* __restrict__ x,
                                                           long long ldx, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, float* __restrict__ part, int rows,
                                                           int cols) {
  __shared__ float sg[4][64], sb[4][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int r0 = blockIdx.y * RCHUNK, r1 = min(rows, r0 + RCHUNK);
  float ag = 0.0f, ab = 0.0f;
  if (c < cols) {
    for (int r = r0 + grp; r < r1; r += 4) {
      const float d = dy[(long long)r * lddy + c];
      if (HAS_X) ag += d * (x[(long long)r * ldx + c] - mean[r]) * rstd[r];
      ab += d;
    }
  }
  sg[grp][lane] = ag;
  sb[grp][lane] = ab;
  __syncthreads();
  if (grp == 0 && c < cols) {
    const long long nch = gridDim.y;
    part[(long long)blockIdx.y * cols + c] = (sb[0][lane] + sb[1][lane]) + (sb[2][lane] + sb[3][lane]);
    if (HAS_X) part[(nch + blockIdx.y) * cols + c] = (sg[0][lane] + sg[1][lane]) + (sg[2][lane] + sg[3][lane]);
  }
}

// out_b[c] (+)= sum over chunks of part[chunk][c]; with out_g also the second plane
__global__ __launch_bounds__(256) void colred_merge_kernel(const float* __restrict__ part, int nch, int cols, float* __restrict__ out_b,
                                                           float* __restrict__ out_g, int accumulate) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  float b = 0.0f, g = 0.0f;
  for (int k = 0; k < nch; ++k) {
    b += part[(long long)k * cols + c];
    if (out_g) g += part[((long long)nch + k) * cols + c];
  }
  out_b[c] = accumulate ? out_b[c] + b : b;
  if (out_g) out_g[c] = accumulate ? out_g[c] + g : g;
}

__global__ __launch_bounds__(256) void softmax_rows_fwd_kernel(float* __restrict__ s, long long ld, int n_rows,
                                                               int cols, float scale) {
  const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n_rows) return;
  float* r = s + (long long)row * ld;
  float mx = -INFINITY;
  for (int c = lane; c < cols; c += 64) mx = fmaxf(mx, r[c] * scale);
  mx = wave_max(mx);
  float sum = 0.0f;
  for (int c = lane; c < cols; c += 64) {
    const float e = expf(r[c] * scale - mx);
    r[c] = e;
    sum += e;
  }
  const float inv = 1.0f / wave_sum(sum);
  for (int c = lane; c < cols; c += 64) r[c] *= inv;
}

__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float* __restrict__ p, float* __restrict__ dp,
                                                               long long ld, int n_rows, int cols, float scale) {
  const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n_rows) return;
  const float* pr = p + (long long)row * ld;
  float* dr = dp + (long long)row * ld;
  float dot = 0.0f;
  for (int c = lane; c < cols; c += 64) dot += pr[c] * dr[c];
  dot = wave_sum(dot);
  for (int c = lane; c < cols; c += 64) dr[c] = scale * pr[c] * (dr[c] - dot);
}

}  // namespace

extern "C" int mcl_layernorm_bwd_ws(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma, const float* mean,
                                    const float* rstd, const float* dx_add, int64_t ldadd, float* dx, int64_t lddx, float* dgamma,
                                    float* dbeta, int32_t accumulate_params, int32_t rows, int32_t cols, float* workspace,
                                    mcl_stream_t stream);
extern "C" int mcl_colsum_ws(const float* x, int64_t ldx, float* out, int32_t rows, int32_t cols, int32_t accumulate, float* workspace,
                             mcl_stream_t stream);

extern "C" int mcl_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float* y,
                                 int64_t ldy, float* mean, float* rstd, int32_t rows, int32_t cols, float eps,
                                 mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!x || !gamma || !beta || !y || !mean || !rstd || rows <= 0 || cols <= 0) return MCL_EINVAL;
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), dim3(256), 0,
                     mcl_stream(stream), x, ldx, gamma, beta, y, ldy, mean, rstd, rows, cols, eps);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_layernorm_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma,
                                 const float* mean, const float* rstd, const float* dx_add, int64_t ldadd,
                                 float* dx, int64_t lddx, float* dgamma, float* dbeta, int32_t accumulate_params,
                                 int32_t rows, int32_t cols, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || rows <= 0 || cols <= 0) return MCL_EINVAL;
  return mcl_layernorm_bwd_ws(dy, lddy, x, ldx, gamma, mean, rstd, dx_add, ldadd, dx, lddx, dgamma, dbeta, accumulate_params, rows,
                              cols, nullptr, stream);
}

extern "C" int64_t mcl_rowred_workspace_floats(int32_t rows, int32_t cols) {
  if (rows <= 0 || cols <= 0) return -1;
  return (int64_t)2 * ((rows + RCHUNK - 1) / RCHUNK) * cols;
}

// as mcl_layernorm_bwd; with a workspace (>= mcl_rowred_workspace_floats(rows, cols) floats) and more than 1024 rows the
// parameter gradients are reduced over row chunks in parallel (the workspace-free form walks all rows on cols / 64 workgroups)
extern "C" int mcl_layernorm_bwd_ws(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma,
                                    const float* mean, const float* rstd, const float* dx_add, int64_t ldadd, float* dx,
                                    int64_t lddx, float* dgamma, float* dbeta, int32_t accumulate_params, int32_t rows,
                                    int32_t cols, float* workspace, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!dy || !x || !gamma || !mean || !rstd || !dx || (!dgamma != !dbeta) || rows <= 0 || cols <= 0) return MCL_EINVAL;
  hipStream_t st = mcl_stream(stream);
  if (!dgamma) {
    // (dx only: the parameter gradients are part of a grouped column-reduction launch, mcl_colred_group)
  } else if (workspace && rows > 1024) {
    const int nch = (rows + RCHUNK - 1) / RCHUNK;
    hipLaunchKernelGGL(colred_chunk_kernel<true>, dim3((cols + 63) / 64, nch), dim3(256), 0, st, dy, (long long)lddy, x,
                       (long long)ldx, mean, rstd, workspace, rows, cols);
    hipLaunchKernelGGL(colred_merge_kernel, dim3((cols + 255) / 256), dim3(256), 0, st, (const float*)workspace, nch, cols, dbeta,
                       dgamma, accumulate_params);
  } else {
    hipLaunchKernelGGL(layernorm_bwd_dgb_kernel, dim3((cols + 63) / 64), dim3(256), 0, st, dy, lddy, x, ldx, mean, rstd,
                       dgamma, dbeta, rows, cols, accumulate_params);
  }
  hipLaunchKernelGGL(layernorm_bwd_dx_kernel, dim3((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), dim3(256), 0, st, dy,
                     lddy, x, ldx, gamma, mean, rstd, dx_add, ldadd, dx, lddx, rows, cols);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_colsum(const float* x, int64_t ldx, float* out, int32_t rows, int32_t cols, int32_t accumulate,
                          mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!x || !out || rows <= 0 || cols <= 0) return MCL_EINVAL;
  return mcl_colsum_ws(x, ldx, out, rows, cols, accumulate, nullptr, stream);
}

extern "C" int mcl_colsum_ws(const float* x, int64_t ldx, float* out, int32_t rows, int32_t cols, int32_t accumulate,
                             float* workspace, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!x || !out || rows <= 0 || cols <= 0) return MCL_EINVAL;
  hipStream_t st = mcl_stream(stream);
  if (workspace && rows > 1024) {
    const int nch = (rows + RCHUNK - 1) / RCHUNK;
    hipLaunchKernelGGL(colred_chunk_kernel<false>, dim3((cols + 63) / 64, nch), dim3(256), 0, st, x, (long long)ldx,
                       (const float*)nullptr, 0LL, (const float*)nullptr, (const float*)nullptr, workspace, rows, cols);
    hipLaunchKernelGGL(colred_merge_kernel, dim3((cols + 255) / 256), dim3(256), 0, st, (const float*)workspace, nch, cols, out,
                       (float*)nullptr, accumulate);
  } else {
    hipLaunchKernelGGL(colsum_kernel, dim3((cols + 63) / 64), dim3(256), 0, st, x, ldx, out, rows, cols, accumulate);
  }
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_colred_group(int32_t n, const float* const* a, const int64_t* lda, const float* const* x, const int64_t* ldx,
                                const float* const* mean, const float* const* rstd, float* const* out0, float* const* out1,
                                const int32_t* cols, const int32_t* accumulate, int32_t rows, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (n <= 0 || n > COLRED_MAX || !a || !lda || !x || !ldx || !mean || !rstd || !out0 || !out1 || !cols || !accumulate || rows <= 0 ||
      rows > 1024)
    return MCL_EINVAL;
  ColredGroup g;
  int maxc = 0;
  for (int i = 0; i < COLRED_MAX; ++i) {
    const int j = i < n ? i : 0;
    if (!a[j] || !out0[j] || cols[j] <= 0) return MCL_EINVAL;
    if (x[j] && (!mean[j] || !rstd[j] || !out1[j])) return MCL_EINVAL;
    g.a[i] = a[j]; g.lda[i] = lda[j]; g.x[i] = x[j]; g.ldx[i] = ldx[j]; g.mean[i] = mean[j]; g.rstd[i] = rstd[j];
    g.out0[i] = out0[j]; g.out1[i] = out1[j]; g.cols[i] = cols[j]; g.accumulate[i] = accumulate[j];
    if (i < n && cols[j] > maxc) maxc = cols[j];
  }
  g.rows = rows;
  hipLaunchKernelGGL(colred_group_kernel, dim3((maxc + 63) / 64, n), dim3(256), 0, mcl_stream(stream), g);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_softmax_rows_fwd(float* s, int64_t ld, int32_t n_rows, int32_t cols, float scale,
                                    mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!s || n_rows <= 0 || cols <= 0) return MCL_EINVAL;
  hipLaunchKernelGGL(softmax_rows_fwd_kernel, dim3((n_rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), dim3(256), 0,
                     mcl_stream(stream), s, ld, n_rows, cols, scale);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_softmax_rows_bwd(const float* p, float* dp, int64_t ld, int32_t n_rows, int32_t cols, float scale,
                                    mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!p || !dp || n_rows <= 0 || cols <= 0) return MCL_EINVAL;
  hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3((n_rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), dim3(256), 0,
                     mcl_stream(stream), p, dp, ld, n_rows, cols, scale);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}
