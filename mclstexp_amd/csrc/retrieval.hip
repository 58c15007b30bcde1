// Inference-time retrieval (/root/reference/evel_her2st.py:74-84,174-187, evel_cscc.py:74-84,
// 197-215, evel_visium.py:94-104,193-205): L2-normalise the embeddings, cosine similarity of every image query
// against every training spot (mcl_gemm, fp32 MFMA), per-query top-k (k = 200 / 600) and the
// inverse-squared-distance weighted average of the matched spots' embeddings and expression.
//
// All three kernels are memory/latency work, not GEMMs:
//   l2_normalize_rows_kernel   one wave per row, row read once
//   topk_rows_kernel           one workgroup per query row: exact MSD radix select on order-preserving integer keys
//                              (11-bit digits starting at the first bit in which the row's keys differ, so the
//                              narrow value range of cosine similarities does not waste a pass), LDS histograms with
//                              per-thread run-length aggregation, ballot-compacted collection, bitonic sort of the k
//                              winners; the row (<= 400 KB) stays in L2 between passes
//   knn_weighted_average_kernel one workgroup per query: distances by wave, fp64 accumulation of the k gathered rows
#include "common.h"

namespace {

constexpr int TOPK_THREADS = 512;
constexpr int TOPK_KMAX = 2048;
constexpr int RADIX_BITS = 11;
constexpr int RADIX_BINS = 1 << RADIX_BITS;

__global__ __launch_bounds__(256) void l2_normalize_rows_kernel(const float* __restrict__ x, long long ldx,
                                                                float* __restrict__ y, long long ldy, int rows,
                                                                int dim) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (long long)row * ldx;
  float ss = 0.f;
  for (int c = lane; c < dim; c += 64) {
    const float v = xr[c];
    ss = fmaf(v, v, ss);
  }
  ss = wave_sum(ss);
  const float denom = fmaxf(sqrtf(ss), 1e-12f);  // F.normalize: x / max(||x||_2, eps)
  float* yr = y + (long long)row * ldy;
  for (int c = lane; c < dim; c += 64) yr[c] = xr[c] / denom;
}

// larger float <=> larger key (NaN with the sign bit clear sorts above +inf, as torch.topk treats it)
__device__ __forceinline__ unsigned f2key(float f) {
  const unsigned u = __float_as_uint(f);
  return u ^ ((unsigned)((int)u >> 31) | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
  const unsigned u = (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k;
  return __uint_as_float(u);
}

__device__ __forceinline__ unsigned lanes_below(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// append (key, idx) of the flagged lanes to sel[] through ONE LDS atomic per wave
__device__ __forceinline__ void wave_append(bool flag, unsigned key, unsigned idx, unsigned long long* sel,
                                            unsigned* counter, unsigned cap) {
  const unsigned long long mask = __ballot(flag);
  if (mask == 0) return;
  const int lane = threadIdx.x & 63;
  const int leader = __builtin_ctzll(mask);
  unsigned base = 0;
  if (lane == leader) base = atomicAdd(counter, (unsigned)__builtin_popcountll(mask));
  base = __shfl(base, leader, 64);
  if (flag) {
    const unsigned pos = base + lanes_below(mask);
    if (pos < cap) sel[pos] = ((unsigned long long)key << 32) | (unsigned long long)(0xFFFFFFFFu - idx);
  }
}

// idx_in (optional): the rows are CANDIDATE LISTS in arbitrary order (mcl_gemm's threshold filter) and idx_in[row][i] is the
// element's original column: it is what gets returned and what orders equal values.  Exact ties AT the k-th value would have to
// be cut by original index, which the position-ordered compaction below cannot do on a scrambled list: such a row raises
// tie_flag[row] and the caller recomputes it on the materialised path.
__global__ __launch_bounds__(TOPK_THREADS) void topk_rows_kernel(const float* __restrict__ sim, long long ld, int n,
                                                                 int k, float* __restrict__ values,
                                                                 long long* __restrict__ indices,
                                                                 const int* __restrict__ idx_in, long long ld_idx,
                                                                 int* __restrict__ tie_flag) {
  __shared__ unsigned hist[RADIX_BINS];
  __shared__ unsigned long long sel[TOPK_KMAX];
  __shared__ unsigned red_min[TOPK_THREADS / 64], red_max[TOPK_THREADS / 64];
  __shared__ unsigned s_cnt, s_bin, s_above, s_inbin, s_run;
  __shared__ unsigned wave_tot[TOPK_THREADS / 64];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* row = sim + (long long)blockIdx.x * ld;
  const int* irow = idx_in ? idx_in + (long long)blockIdx.x * ld_idx : nullptr;

  // ---- pass 0: key range of the row
  unsigned kmin = 0xFFFFFFFFu, kmax = 0u;
  for (int i = tid; i < n; i += TOPK_THREADS) {
    const unsigned key = f2key(row[i]);
    kmin = min(kmin, key);
    kmax = max(kmax, key);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    kmin = min(kmin, (unsigned)__shfl_xor((int)kmin, o, 64));
    kmax = max(kmax, (unsigned)__shfl_xor((int)kmax, o, 64));
  }
  if (lane == 0) {
    red_min[wave] = kmin;
    red_max[wave] = kmax;
  }
  if (tid == 0) s_cnt = 0;
  __syncthreads();
#pragma unroll
  for (int w = 0; w < TOPK_THREADS / 64; ++w) {
    kmin = min(kmin, red_min[w]);
    kmax = max(kmax, red_max[w]);
  }

  // ---- MSD radix select: after the loop, keys with (key >> s) > pref are winners, the class (key >> s) == pref
  // holds `inbin` keys of which `need` are still wanted (inbin == need unless s == 0: exact ties)
  int s = (kmin == kmax) ? 0 : 32 - __builtin_clz(kmin ^ kmax);
  unsigned long long pref = (unsigned long long)kmax >> s;
  unsigned need = (unsigned)k, inbin = (unsigned)n;
  while (s > 0) {
    const int w = s < RADIX_BITS ? s : RADIX_BITS;
    const int s2 = s - w;
    const unsigned dmask = (1u << w) - 1u;
    for (int b = tid; b < RADIX_BINS; b += TOPK_THREADS) hist[b] = 0;
    __syncthreads();
    unsigned cur = 0xFFFFFFFFu, run = 0;
    for (int i = tid; i < n; i += TOPK_THREADS) {
      const unsigned key = f2key(row[i]);
      if (((unsigned long long)key >> s) == pref) {
        const unsigned d = (key >> s2) & dmask;
        if (d != cur) {
          if (run) atomicAdd(&hist[cur], run);
          cur = d;
          run = 0;
        }
        ++run;
      }
    }
    if (run) atomicAdd(&hist[cur], run);
    __syncthreads();
    if (wave == 0) {
      // lane L owns bins [32 L, 32 L + 32); suffix sums from the top bin down
      unsigned part = 0;
#pragma unroll 8
      for (int b = 0; b < RADIX_BINS / 64; ++b) part += hist[lane * (RADIX_BINS / 64) + b];
      unsigned suf = part;  // inclusive suffix sum over lanes >= L
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned v = (unsigned)__shfl_down((int)suf, o, 64);
        if (lane + o < 64) suf += v;
      }
      const unsigned above_lane = suf - part;  // keys in bins owned by higher lanes
      if (above_lane < need && need <= suf) {
        unsigned above = above_lane;
        for (int b = RADIX_BINS / 64 - 1; b >= 0; --b) {
          const unsigned c = hist[lane * (RADIX_BINS / 64) + b];
          if (above + c >= need) {
            s_bin = (unsigned)(lane * (RADIX_BINS / 64) + b);
            s_above = above;
            s_inbin = c;
            break;
          }
          above += c;
        }
      }
    }
    __syncthreads();
    need -= s_above;
    inbin = s_inbin;
    pref = (pref << w) | s_bin;
    s = s2;
    __syncthreads();
    if (inbin == need) break;
  }

  // ---- collection: winners, then the boundary class
  for (int i0 = 0; i0 < n; i0 += TOPK_THREADS) {
    const int i = i0 + tid;
    unsigned key = 0;
    bool win = false, tie = false;
    if (i < n) {
      key = f2key(row[i]);
      const unsigned long long hi = (unsigned long long)key >> s;
      win = hi > pref;
      tie = (hi == pref) && (inbin == need);
    }
    wave_append(win || tie, key, (unsigned)((irow && i < n) ? irow[i] : i), sel, &s_cnt, (unsigned)k);
  }
  if (inbin != need && tie_flag && tid == 0) tie_flag[blockIdx.x] = 1;
  if (inbin != need) {
    // exact ties at the k-th value: take the lowest indices (ordered compaction, chunk by chunk)
    __syncthreads();
    unsigned base = s_cnt, taken = 0;
    for (int i0 = 0; i0 < n && taken < need; i0 += TOPK_THREADS) {
      const int i = i0 + tid;
      unsigned key = 0;
      bool tie = false;
      if (i < n) {
        key = f2key(row[i]);
        tie = ((unsigned long long)key >> s) == pref;
      }
      const unsigned long long mask = __ballot(tie);
      if (lane == 0) wave_tot[wave] = (unsigned)__builtin_popcountll(mask);
      __syncthreads();
      unsigned before = 0, total = 0;
#pragma unroll
      for (int w = 0; w < TOPK_THREADS / 64; ++w) {
        const unsigned c = wave_tot[w];
        if (w < wave) before += c;
        total += c;
      }
      if (tie) {
        const unsigned r = taken + before + lanes_below(mask);
        if (r < need)
          sel[base + r] = ((unsigned long long)key << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)(irow ? irow[i] : i));
      }
      taken += total;
      __syncthreads();
    }
  }
  __syncthreads();

  // ---- bitonic sort (descending composite: value first, lower index first among equals)
  int n2 = 1;
  while (n2 < k) n2 <<= 1;
  for (int i = k + tid; i < n2; i += TOPK_THREADS) sel[i] = 0ull;
  __syncthreads();
  for (int size = 2; size <= n2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (n2 >> 1); t += TOPK_THREADS) {
        const int lo = ((t / stride) * stride << 1) + (t % stride);
        const int hi = lo + stride;
        const bool desc = ((lo & size) == 0);
        const unsigned long long a = sel[lo], b = sel[hi];
        if ((a < b) == desc) {
          sel[lo] = b;
          sel[hi] = a;
        }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < k; i += TOPK_THREADS) {
    const unsigned long long c = sel[i];
    values[(long long)blockIdx.x * k + i] = key2f((unsigned)(c >> 32));
    indices[(long long)blockIdx.x * k + i] = (long long)(0xFFFFFFFFu - (unsigned)c);
  }
}

constexpr int KNN_THREADS = 256;

__global__ __launch_bounds__(KNN_THREADS) void knn_weighted_average_kernel(
    const float* __restrict__ spot_key, long long ldk, const float* __restrict__ expression_key, long long lde,
    const float* __restrict__ query, long long ldq, const long long* __restrict__ indices, int k, int dim, int genes,
    int ord, float* __restrict__ emb_pred, float* __restrict__ expr_pred) {
  extern __shared__ unsigned char smem[];
  float* wgt = reinterpret_cast<float*>(smem);          // k
  int* nbr = reinterpret_cast<int*>(wgt + k);            // k
  __shared__ double red[KNN_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qi = blockIdx.x;
  const float* q = query + (long long)qi * ldq;

  for (int j = tid; j < k; j += KNN_THREADS) nbr[j] = (int)indices[(long long)qi * k + j];
  __syncthreads();
  // a_j = || key[idx_j] - q ||_ord on the un-normalised embeddings (fp32 as numpy computes it), w_j = 1 / a_j^2
  for (int j = wave; j < k; j += KNN_THREADS / 64) {
    const float* r = spot_key + (long long)nbr[j] * ldk;
    float acc = 0.f;
    for (int c = lane; c < dim; c += 64) {
      const float d = r[c] - q[c];
      acc = (ord == 1) ? acc + fabsf(d) : fmaf(d, d, acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) {
      const float a = (ord == 1) ? acc : sqrtf(acc);
      wgt[j] = 1.0f / (a * a);
    }
  }
  __syncthreads();
  double tot = 0.0;
  for (int j = tid; j < k; j += KNN_THREADS) tot += (double)wgt[j];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
  if (lane == 0) red[wave] = tot;
  __syncthreads();
  tot = 0.0;
#pragma unroll
  for (int w = 0; w < KNN_THREADS / 64; ++w) tot += red[w];

  // weighted column sums: thread owns a column, neighbours streamed (coalesced across the workgroup)
  for (int pass = 0; pass < 2; ++pass) {
    const float* src = pass ? expression_key : spot_key;
    const long long lds_ = pass ? lde : ldk;
    const int cols = pass ? genes : dim;
    float* dst = (pass ? expr_pred + (long long)qi * genes : emb_pred + (long long)qi * dim);
    if (src == nullptr || dst == nullptr) continue;
    for (int c = tid; c < cols; c += KNN_THREADS) {
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
      int j = 0;
      for (; j + 4 <= k; j += 4) {
        const float v0 = src[(long long)nbr[j] * lds_ + c];
        const float v1 = src[(long long)nbr[j + 1] * lds_ + c];
        const float v2 = src[(long long)nbr[j + 2] * lds_ + c];
        const float v3 = src[(long long)nbr[j + 3] * lds_ + c];
        a0 += (double)wgt[j] * (double)v0;
        a1 += (double)wgt[j + 1] * (double)v1;
        a2 += (double)wgt[j + 2] * (double)v2;
        a3 += (double)wgt[j + 3] * (double)v3;
      }
      for (; j < k; ++j) a0 += (double)wgt[j] * (double)src[(long long)nbr[j] * lds_ + c];
      dst[c] = (float)(((a0 + a1) + (a2 + a3)) / tot);
    }
  }
}

}  // namespace

extern "C" int mcl_l2_normalize_rows(const float* x, int64_t ldx, float* y, int64_t ldy, int rows, int dim,
                                     mcl_stream_t stream) {
  if (rows == 0) return MCL_OK;
  if (!x || !y || rows < 0 || dim <= 0 || ldx < dim || ldy < dim) return MCL_EINVAL;
  MCL_CLEAR_ERROR();
  hipLaunchKernelGGL(l2_normalize_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, mcl_stream(stream), x,
                     (long long)ldx, y, (long long)ldy, rows, dim);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_topk_rows_max_k(void) { return TOPK_KMAX; }

extern "C" int mcl_topk_rows(const float* sim, int64_t ld, int rows, int n, int k, float* values, int64_t* indices,
                             mcl_stream_t stream) {
  if (rows == 0) return MCL_OK;
  if (!sim || !values || !indices || rows < 0 || n <= 0 || k <= 0 || k > n || ld < n) return MCL_EINVAL;
  if (k > TOPK_KMAX) return MCL_EUNSUPPORTED;
  MCL_CLEAR_ERROR();
  hipLaunchKernelGGL(topk_rows_kernel, dim3(rows), dim3(TOPK_THREADS), 0, mcl_stream(stream), sim, (long long)ld, n, k,
                     values, reinterpret_cast<long long*>(indices), (const int*)nullptr, 0LL, (int*)nullptr);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

// top-k of candidate lists: row i holds n (value, original column) pairs in arbitrary order (unused slots: -inf); see the kernel
extern "C" int mcl_topk_rows_indexed(const float* cand_val, int64_t ld, const int32_t* cand_idx, int64_t ld_idx, int rows, int n,
                                     int k, float* values, int64_t* indices, int32_t* tie_flag, mcl_stream_t stream) {
  if (rows == 0) return MCL_OK;
  if (!cand_val || !cand_idx || !values || !indices || !tie_flag || rows < 0 || n <= 0 || k <= 0 || k > n || ld < n || ld_idx < n)
    return MCL_EINVAL;
  if (k > TOPK_KMAX) return MCL_EUNSUPPORTED;
  MCL_CLEAR_ERROR();
  hipLaunchKernelGGL(topk_rows_kernel, dim3(rows), dim3(TOPK_THREADS), 0, mcl_stream(stream), cand_val, (long long)ld, n, k,
                     values, reinterpret_cast<long long*>(indices), (const int*)cand_idx, (long long)ld_idx, (int*)tie_flag);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_knn_weighted_average(const float* spot_key, int64_t ldk, const float* expression_key, int64_t lde,
                                        const float* query, int64_t ldq, const int64_t* indices, int n_query, int k,
                                        int dim, int genes, int ord, float* emb_pred, float* expr_pred,
                                        mcl_stream_t stream) {
  if (n_query == 0) return MCL_OK;
  if (!spot_key || !query || !indices || n_query < 0 || k <= 0 || dim <= 0 || ldk < dim || ldq < dim)
    return MCL_EINVAL;
  if ((expr_pred != nullptr) && (!expression_key || genes <= 0 || lde < genes)) return MCL_EINVAL;
  if (ord != 1 && ord != 2) return MCL_EUNSUPPORTED;
  if ((size_t)k * 8 > 60000) return MCL_EUNSUPPORTED;
  MCL_CLEAR_ERROR();
  hipLaunchKernelGGL(knn_weighted_average_kernel, dim3(n_query), dim3(KNN_THREADS), (size_t)k * 8, mcl_stream(stream),
                     spot_key, (long long)ldk, expr_pred ? expression_key : nullptr, (long long)lde, query,
                     (long long)ldq, reinterpret_cast<const long long*>(indices), k, dim, genes, ord, emb_pred,
                     expr_pred);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}
