// Fused multi-head self-attention core of the spot Transformer (/root/reference/model.py:49-57: q k^T * d^-1/2 -> softmax -> . v over
// the batch-as-sequence), head dimension 64, fp32 -- forward and backward without the (heads, B, B) probability tensor in HBM:
//
//   forward   out_h = softmax(q_h k_h^T * scale) v_h, and the row log-sum-exp (the only thing the backward needs beside qkv / out)
//   backward  dq_h = dS k_h,  dk_h = dS^T q_h,  dv_h = P^T dO_h,   dS = P .* (dO_h v_h^T - D) * scale,  D_i = sum_d dO[i][d] out[i][d]
//             with P recomputed from (q, k, lse)
//
// The unfused path (ops.py: two batched GEMMs + a softmax launch per direction, P and dP through HBM) is 3 + 5 launches per layer;
// here 1 + 2.  fp32 on the matrix cores: v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation).
//
// Forward / dq kernels: a workgroup (4 waves) owns 32 queries of one head and walks the keys in blocks of 128.
//   * scores: wave w computes the 32 x 32 tile of keys 32 w .. 32 w + 31.  Both MFMA operands are "row" operands (query rows, key
//     rows, 64 floats each); the contraction index is PERMUTED so that lane (row, half) holds 32 CONSECUTIVE floats of its row
//     (step t contracts elements t and 32 + t): operands are eight 16-byte loads per lane straight from HBM, no LDS staging;
//   * the 32 x 128 score block goes through LDS once (pitch 129: conflict-free as an MFMA operand), where eight threads per row do
//     the online softmax (running max / sum, flash-attention rescaling) -- or, backward, where dS is formed in registers first;
//   * P V (dS K): wave (n-tile, key half) multiplies the LDS block with value rows read straight from HBM (a lane reads one float
//     per contraction step, 32 lanes = one 128-byte row piece); the two key halves meet in LDS at the end.
// dk / dv kernel: the transposed problem -- a workgroup owns 32 keys and walks the queries; scores come out transposed
// (key rows as the A operand), so the per-query lse / D are per-LANE constants.
#include "common.h"
#include <math.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int AD = 64;          // head dimension
constexpr int LP = 129;         // LDS pitch (floats) of a 32 x 128 block

// lane (i = lane & 31, kk = lane >> 5) <- floats [32 kk, 32 kk + 32) of the 64-float head slice of row ``row``
__device__ __forceinline__ void load_row32(const float* __restrict__ base, long long ld, int row, int col0, int kk, float (&r)[32]) {
  const float4* p = reinterpret_cast<const float4*>(base + (long long)row * ld + col0 + 32 * kk);
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const float4 v = p[u];
    r[4 * u] = v.x; r[4 * u + 1] = v.y; r[4 * u + 2] = v.z; r[4 * u + 3] = v.w;
  }
}

// acc[r] = sum_k X[i_r][k] Y[n][k],  i_r = (r & 3) + 8 (r >> 2) + 4 (lane >> 5),  n = lane & 31
__device__ __forceinline__ f32x16 tile_xyT(const float (&x)[32], const float (&y)[32]) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
  for (int t = 0; t < 32; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x[t], y[t], acc, 0, 0, 0);
  return acc;
}

// acc[i][n] += sum_{c < 64} M[i][c0 + c] * Y[min(rowbase + c, nrows - 1)][ycol + n]       (M in LDS, pitch LP; Y in HBM)
__device__ __forceinline__ void tile_mY(f32x16& acc, const float* __restrict__ Ms, int c0, const float* __restrict__ Y, long long ldy,
                                        int rowbase, int nrows, int ycol, int lane) {
  const int i = lane & 31, kk = lane >> 5;
  float b[32];
#pragma unroll
  for (int t = 0; t < 32; ++t) {
    const int row = min(rowbase + 32 * kk + t, nrows - 1);
    b[t] = Y[(long long)row * ldy + ycol + i];
  }
#pragma unroll
  for (int t = 0; t < 32; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ms[i * LP + c0 + 32 * kk + t], b[t], acc, 0, 0, 0);
}

__device__ __forceinline__ int row_of(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// The two contraction halves of a (32 x 64) result meet in LDS; ``scale_rows`` (LDS, 32 floats, or nullptr) scales the rows;
// rows row0 + i < nrows are stored to dst[(row0 + i) * ldd + col]
__device__ __forceinline__ void reduce_store(f32x16 acc, float* __restrict__ red /* [32][LP] */, const float* __restrict__ scale_rows,
                                             float* __restrict__ dst, long long ldd, int row0, int nrows, int ntile, int half, int lane) {
  __syncthreads();
  if (half == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[row_of(r, lane) * LP + ntile * 32 + (lane & 31)] = acc[r];
  }
  __syncthreads();
  if (half == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = row_of(r, lane);
      float v = acc[r] + red[i * LP + ntile * 32 + (lane & 31)];
      if (scale_rows != nullptr) v *= scale_rows[i];
      if (row0 + i < nrows) dst[(long long)(row0 + i) * ldd + ntile * 32 + (lane & 31)] = v;
    }
  }
}

__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ qkv, long long ld, int Bn, int heads, float scale,
                                                       float* __restrict__ out, long long ldo, float* __restrict__ lse) {
  __shared__ float Ss[32 * LP];
  __shared__ float alpha_s[32], linv_s[32];
  {                                                        // sequence blockIdx.z of a batch of independent sequences
    const long long sq = blockIdx.z;
    qkv += sq * Bn * ld;
    out += sq * Bn * ldo;
    lse += sq * heads * Bn;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, kk = lane >> 5;
  const int q0 = blockIdx.x * 32, hd = blockIdx.y, inner = heads * AD;
  const int ntile = wave & 1, khalf = wave >> 1;
  const int srow = tid >> 3, sub = tid & 7;              // softmax role: 8 threads per query row

  float xq[32];
  load_row32(qkv, ld, min(q0 + i, Bn - 1), hd * AD, kk, xq);
  f32x16 oacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) oacc[r] = 0.0f;
  float m_run = -INFINITY, l_run = 0.0f;

  for (int k0 = 0; k0 < Bn; k0 += 128) {
    float yk[32];
    load_row32(qkv, ld, min(k0 + 32 * wave + i, Bn - 1), inner + hd * AD, kk, yk);
    const f32x16 s = tile_xyT(xq, yk);
#pragma unroll
    for (int r = 0; r < 16; ++r) Ss[row_of(r, lane) * LP + 32 * wave + i] = s[r] * scale;
    __syncthreads();
    {
      float* sr = Ss + srow * LP;
      float mx = -INFINITY;
      for (int c = sub; c < 128; c += 8)
        if (k0 + c < Bn) mx = fmaxf(mx, sr[c]);
      mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
      const float m_new = fmaxf(m_run, mx);               // finite: every block holds at least one valid key
      float ls = 0.0f;
      for (int c = sub; c < 128; c += 8) {
        const float p = k0 + c < Bn ? expf(sr[c] - m_new) : 0.0f;
        sr[c] = p;
        ls += p;
      }
      ls += __shfl_xor(ls, 1, 64);
      ls += __shfl_xor(ls, 2, 64);
      ls += __shfl_xor(ls, 4, 64);
      const float a = m_run == -INFINITY ? 0.0f : expf(m_run - m_new);
      l_run = a * l_run + ls;
      m_run = m_new;
      if (sub == 0) alpha_s[srow] = a;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[r] *= alpha_s[row_of(r, lane)];
    tile_mY(oacc, Ss, khalf * 64, qkv + 2 * inner + hd * AD, ld, k0 + khalf * 64, Bn, ntile * 32, lane);
    __syncthreads();
  }
  if (sub == 0) {
    linv_s[srow] = 1.0f / l_run;
    if (q0 + srow < Bn) lse[(long long)hd * Bn + q0 + srow] = m_run + logf(l_run);
  }
  reduce_store(oacc, Ss, linv_s, out + hd * AD, ldo, q0, Bn, ntile, khalf, lane);
}

// dq of 32 queries of one head; also D_i = sum_d dO[i][d] out[i][d] for them (dvec[head][query], read by the dk / dv kernel)
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const float* __restrict__ qkv, long long ld, int Bn, int heads, float scale,
                                                          const float* __restrict__ out, const float* __restrict__ dout, long long ldo,
                                                          const float* __restrict__ lse, float* __restrict__ dvec,
                                                          float* __restrict__ dqkv, long long ldq) {
  __shared__ float Ss[32 * LP];
  __shared__ float lse_s[32], d_s[32];
  {
    const long long sq = blockIdx.z;
    qkv += sq * Bn * ld;
    out += sq * Bn * ldo;
    dout += sq * Bn * ldo;
    lse += sq * heads * Bn;
    dvec += sq * heads * Bn;
    dqkv += sq * Bn * ldq;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, kk = lane >> 5;
  const int q0 = blockIdx.x * 32, hd = blockIdx.y, inner = heads * AD;
  const int ntile = wave & 1, khalf = wave >> 1;
  {
    const int srow = tid >> 3, sub = tid & 7, row = min(q0 + srow, Bn - 1);
    const float* po = out + (long long)row * ldo + hd * AD + 8 * sub;
    const float* pd = dout + (long long)row * ldo + hd * AD + 8 * sub;
    float d = 0.0f;
#pragma unroll
    for (int u = 0; u < 8; ++u) d = fmaf(po[u], pd[u], d);
    d += __shfl_xor(d, 1, 64);
    d += __shfl_xor(d, 2, 64);
    d += __shfl_xor(d, 4, 64);
    if (sub == 0) {
      d_s[srow] = d;
      lse_s[srow] = lse[(long long)hd * Bn + row];
      if (q0 + srow < Bn) dvec[(long long)hd * Bn + q0 + srow] = d;
    }
  }
  float xq[32], xdo[32];
  load_row32(qkv, ld, min(q0 + i, Bn - 1), hd * AD, kk, xq);
  load_row32(dout, ldo, min(q0 + i, Bn - 1), hd * AD, kk, xdo);
  f32x16 dq;
#pragma unroll
  for (int r = 0; r < 16; ++r) dq[r] = 0.0f;
  __syncthreads();

  for (int k0 = 0; k0 < Bn; k0 += 128) {
    const int key = k0 + 32 * wave + i;
    float yk[32], yv[32];
    load_row32(qkv, ld, min(key, Bn - 1), inner + hd * AD, kk, yk);
    load_row32(qkv, ld, min(key, Bn - 1), 2 * inner + hd * AD, kk, yv);
    const f32x16 s = tile_xyT(xq, yk);
    const f32x16 dp = tile_xyT(xdo, yv);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row_of(r, lane);
      const float p = key < Bn ? expf(s[r] * scale - lse_s[row]) : 0.0f;
      Ss[row * LP + 32 * wave + i] = p * (dp[r] - d_s[row]) * scale;
    }
    __syncthreads();
    tile_mY(dq, Ss, khalf * 64, qkv + inner + hd * AD, ld, k0 + khalf * 64, Bn, ntile * 32, lane);
    __syncthreads();
  }
  reduce_store(dq, Ss, nullptr, dqkv + hd * AD, ldq, q0, Bn, ntile, khalf, lane);
}

// dk, dv of 32 keys of one head (walks the queries in blocks of 128)
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const float* __restrict__ qkv, long long ld, int Bn, int heads, float scale,
                                                           const float* __restrict__ dout, long long ldo,
                                                           const float* __restrict__ lse, const float* __restrict__ dvec,
                                                           float* __restrict__ dqkv, long long ldq) {
  __shared__ float Ps[32 * LP];
  __shared__ float Ds[32 * LP];
  {
    const long long sq = blockIdx.z;
    qkv += sq * Bn * ld;
    dout += sq * Bn * ldo;
    lse += sq * heads * Bn;
    dvec += sq * heads * Bn;
    dqkv += sq * Bn * ldq;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, kk = lane >> 5;
  const int j0 = blockIdx.x * 32, hd = blockIdx.y, inner = heads * AD;
  const int ntile = wave & 1, qhalf = wave >> 1;
  float xk[32], xv[32];
  load_row32(qkv, ld, min(j0 + i, Bn - 1), inner + hd * AD, kk, xk);
  load_row32(qkv, ld, min(j0 + i, Bn - 1), 2 * inner + hd * AD, kk, xv);
  f32x16 dk, dv;
#pragma unroll
  for (int r = 0; r < 16; ++r) dk[r] = dv[r] = 0.0f;

  for (int q0 = 0; q0 < Bn; q0 += 128) {
    const int query = q0 + 32 * wave + i;
    const int qc = min(query, Bn - 1);
    float yq[32], ydo[32];
    load_row32(qkv, ld, qc, hd * AD, kk, yq);
    load_row32(dout, ldo, qc, hd * AD, kk, ydo);
    const float lse_n = lse[(long long)hd * Bn + qc], d_n = dvec[(long long)hd * Bn + qc];
    const f32x16 sT = tile_xyT(xk, yq);                    // sT[r] = score(query n, key j0 + row_of(r))
    const f32x16 dpT = tile_xyT(xv, ydo);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row_of(r, lane);
      const float p = query < Bn ? expf(sT[r] * scale - lse_n) : 0.0f;
      Ps[row * LP + 32 * wave + i] = p;
      Ds[row * LP + 32 * wave + i] = p * (dpT[r] - d_n) * scale;
    }
    __syncthreads();
    tile_mY(dv, Ps, qhalf * 64, dout + hd * AD, ldo, q0 + qhalf * 64, Bn, ntile * 32, lane);
    tile_mY(dk, Ds, qhalf * 64, qkv + hd * AD, ld, q0 + qhalf * 64, Bn, ntile * 32, lane);
    __syncthreads();
  }
  reduce_store(dk, Ps, nullptr, dqkv + inner + hd * AD, ldq, j0, Bn, ntile, qhalf, lane);
  reduce_store(dv, Ds, nullptr, dqkv + 2 * inner + hd * AD, ldq, j0, Bn, ntile, qhalf, lane);
}

inline bool attn_args_ok(const void* a, const void* b, int64_t ld, int64_t ldo, int32_t Bn, int32_t heads, int32_t dim_head) {
  return a && b && Bn > 0 && heads > 0 && dim_head == AD && (ld % 4) == 0 && (ldo % 4) == 0 && ld >= 3LL * heads * AD &&
         ldo >= (int64_t)heads * AD && !(reinterpret_cast<uintptr_t>(a) & 15u) && !(reinterpret_cast<uintptr_t>(b) & 15u);
}

}  // namespace

// out (B, heads*64) = attention core of qkv (B, 3*heads*64; q | k | v, head-major inside each), lse (heads, B) fp32.
// nseq > 1: nseq independent sequences of B tokens each, rows sequence-major (the fp32 ViT: one sequence per image,
// /root/reference/model.py:104-116); lse / dvec are (nseq, heads, B).
extern "C" int mcl_attention_batched_fwd(const float* qkv, int64_t ld, int32_t B, int32_t nseq, int32_t heads, int32_t dim_head,
                                         float scale, float* out, int64_t ldo, float* lse, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!qkv || !out || !lse || B <= 0 || heads <= 0 || nseq <= 0) return MCL_EINVAL;
  if (!attn_args_ok(qkv, out, ld, ldo, B, heads, dim_head) || nseq > 65535) return MCL_EUNSUPPORTED;
  hipLaunchKernelGGL(attn_fwd_kernel, dim3((B + 31) / 32, heads, nseq), dim3(256), 0, mcl_stream(stream), qkv, (long long)ld, B,
                     heads, scale, out, (long long)ldo, lse);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}
extern "C" int mcl_attention_fwd(const float* qkv, int64_t ld, int32_t B, int32_t heads, int32_t dim_head, float scale, float* out,
                                 int64_t ldo, float* lse, mcl_stream_t stream) {
  return mcl_attention_batched_fwd(qkv, ld, B, 1, heads, dim_head, scale, out, ldo, lse, stream);
}

// dqkv (B, 3*heads*64, row stride ldq) from dout, qkv, out and lse of the forward; dvec: (heads, B) fp32 scratch.
extern "C" int mcl_attention_batched_bwd(const float* qkv, int64_t ld, int32_t B, int32_t nseq, int32_t heads, int32_t dim_head,
                                         float scale, const float* out, const float* dout, int64_t ldo, const float* lse,
                                         float* dvec, float* dqkv, int64_t ldq, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!qkv || !out || !dout || !lse || !dvec || !dqkv || B <= 0 || heads <= 0 || nseq <= 0) return MCL_EINVAL;
  if (!attn_args_ok(qkv, out, ld, ldo, B, heads, dim_head) || (reinterpret_cast<uintptr_t>(dout) & 15u) ||
      ldq < 3LL * heads * AD || nseq > 65535)
    return MCL_EUNSUPPORTED;
  hipStream_t st = mcl_stream(stream);
  const dim3 grid((B + 31) / 32, heads, nseq);
  hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, dim3(256), 0, st, qkv, (long long)ld, B, heads, scale, out, dout, (long long)ldo, lse,
                     dvec, dqkv, (long long)ldq);
  hipLaunchKernelGGL(attn_bwd_dkv_kernel, grid, dim3(256), 0, st, qkv, (long long)ld, B, heads, scale, dout, (long long)ldo, lse, dvec,
                     dqkv, (long long)ldq);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}
extern "C" int mcl_attention_bwd(const float* qkv, int64_t ld, int32_t B, int32_t heads, int32_t dim_head, float scale,
                                 const float* out, const float* dout, int64_t ldo, const float* lse, float* dvec, float* dqkv,
                                 int64_t ldq, mcl_stream_t stream) {
  return mcl_attention_batched_bwd(qkv, ld, B, 1, heads, dim_head, scale, out, dout, ldo, lse, dvec, dqkv, ldq, stream);
}
