// K1: position-embedding gather+add (model.py:230-235) and its row-sparse, deterministic backward.
// HBM-bound row copies: one workgroup per spot, consecutive lanes on consecutive genes.
#include "common.h"

namespace {

__device__ __forceinline__ int trunc_index(float f, int n_rows, int* err) {
  long long i = (long long)f;  // .long(): truncation toward zero, model.py:230-231
  if (i < 0 || i >= n_rows) {
    if (err) *err = 1;
    i = i < 0 ? 0 : n_rows - 1;
  }
  return (int)i;
}

__global__ __launch_bounds__(256) void pos_embed_add_fwd_kernel(const float* __restrict__ expr, long long ld_expr,
                                                                const float* __restrict__ pos,
                                                                const float* __restrict__ xt,
                                                                const float* __restrict__ yt, long long ld_table,
                                                                int n_rows, float* __restrict__ out, long long ld_out,
                                                                int* __restrict__ ix, int* __restrict__ iy,
                                                                int* __restrict__ err, int G) {
  const int b = blockIdx.x;
  const int x = trunc_index(pos[2 * b + 0], n_rows, err);
  const int y = trunc_index(pos[2 * b + 1], n_rows, err);
  if (threadIdx.x == 0) {
    ix[b] = x;
    iy[b] = y;
  }
  const float* e = expr + (long long)b * ld_expr;
  const float* xr = xt + (long long)x * ld_table;
  const float* yr = yt + (long long)y * ld_table;
  float* o = out + (long long)b * ld_out;
  for (int c = threadIdx.x; c < G; c += 256) o[c] = e[c] + xr[c] + yr[c];  // same order as model.py:235
}

__global__ __launch_bounds__(256) void embed_rowgrad_kernel(const float* __restrict__ d, long long ldd,
                                                            const int* __restrict__ idx, int* __restrict__ owner,
                                                            float* __restrict__ rg, long long ldrg, int B, int G) {
  __shared__ int dup_before;
  const int b = blockIdx.x;
  const int me = idx[b];
  if (threadIdx.x == 0) dup_before = 0;
  __syncthreads();
  int found = 0;
  for (int j = threadIdx.x; j < b; j += 256) found |= (idx[j] == me);
  if (found) dup_before = 1;  // benign race: all writers store 1
  __syncthreads();
  if (dup_before) {
    if (threadIdx.x == 0) owner[b] = -1;
    return;
  }
  if (threadIdx.x == 0) owner[b] = me;
  for (int c = threadIdx.x; c < G; c += 256) {
    float acc = d[(long long)b * ldd + c];
    for (int j = b + 1; j < B; ++j)
      if (idx[j] == me) acc += d[(long long)j * ldd + c];
    rg[(long long)b * ldrg + c] = acc;
  }
}

__global__ __launch_bounds__(256) void embed_scatter_rows_kernel(const int* __restrict__ owner,
                                                                 const float* __restrict__ rg, long long ldrg,
                                                                 float* __restrict__ tg, long long ldt, int G,
                                                                 int accumulate) {
  const int b = blockIdx.x;
  const int row = owner[b];
  if (row < 0) return;
  float* t = tg + (long long)row * ldt;
  const float* s = rg + (long long)b * ldrg;
  for (int c = threadIdx.x; c < G; c += 256) t[c] = accumulate ? t[c] + s[c] : s[c];
}

__global__ void row_slot_update_kernel(int* __restrict__ slot, const int* __restrict__ owner, int B, int fill) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int row = owner[b];
  if (row >= 0) slot[row] = fill ? b : -1;
}

}  // namespace

extern "C" int mcl_pos_embed_add_fwd(const float* expr, int64_t ld_expr, const float* pos, const float* x_table,
                                     const float* y_table, int64_t ld_table, int32_t n_rows, float* out,
                                     int64_t ld_out, int32_t* ix, int32_t* iy, int32_t* err_flag, int32_t B,
                                     int32_t G, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!expr || !pos || !x_table || !y_table || !out || !ix || !iy || B <= 0 || G <= 0 || n_rows <= 0)
    return MCL_EINVAL;
  hipLaunchKernelGGL(pos_embed_add_fwd_kernel, dim3(B), dim3(256), 0, mcl_stream(stream), expr, ld_expr, pos, x_table,
                     y_table, ld_table, n_rows, out, ld_out, ix, iy, err_flag, G);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_embed_rowgrad(const float* d_out, int64_t ld_dout, const int32_t* idx, int32_t* owner_idx,
                                 float* row_grad, int64_t ld_rg, int32_t B, int32_t G, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!d_out || !idx || !owner_idx || !row_grad || B <= 0 || G <= 0) return MCL_EINVAL;
  hipLaunchKernelGGL(embed_rowgrad_kernel, dim3(B), dim3(256), 0, mcl_stream(stream), d_out, ld_dout, idx, owner_idx,
                     row_grad, ld_rg, B, G);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_embed_scatter_rows(const int32_t* owner_idx, const float* row_grad, int64_t ld_rg,
                                      float* table_grad, int64_t ld_table, int32_t B, int32_t G, int32_t accumulate,
                                      mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!owner_idx || !row_grad || !table_grad || B <= 0 || G <= 0) return MCL_EINVAL;
  hipLaunchKernelGGL(embed_scatter_rows_kernel, dim3(B), dim3(256), 0, mcl_stream(stream), owner_idx, row_grad, ld_rg,
                     table_grad, ld_table, G, accumulate);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_row_slot_update(int32_t* row_slot, const int32_t* owner_idx, int32_t B, int32_t fill,
                                   mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!row_slot || !owner_idx || B <= 0) return MCL_EINVAL;
  hipLaunchKernelGGL(row_slot_update_kernel, dim3((B + 255) / 256), dim3(256), 0, mcl_stream(stream), row_slot,
                     owner_idx, B, fill);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}
