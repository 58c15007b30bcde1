// Train-mode BatchNorm(+ReLU) for the DenseNet backbone on channels-last activations, written for the
// concat-free dense block: every kernel addresses a (S = B*H*W rows) x (C channels) slice of a wider
// NHWC buffer through a row stride, so dense-layer inputs are read in place (no torch.cat copy), the
// per-channel batch statistics of a feature map are computed ONCE when it is produced, and the
// data-gradient is accumulated in place into the block's gradient buffer (no autograd add chain).
//
// All five kernels are HBM-streaming with the same thread mapping: a thread owns one 16-byte channel
// vector (8 bf16 / 4 fp32 channels) for the whole launch and walks rows, so per-channel parameters live
// in registers and per-channel reductions are register accumulators, merged once per workgroup through
// LDS and once per launch by a deterministic finalize kernel (no atomics).
//   stats:     mean/var over S per channel (shifted sums: d = x - x[0,c], immune to mean >> std)
//   act_fwd:   y = relu?(x*scale + shift), scale = gamma*rstd, shift = beta - mean*scale
//   act_bwd:   g = dy*[y>0]; dgamma = sum g*xhat; dbeta = sum g; dx (+)= gamma*rstd*(g - mean(g) - xhat*mean(g*xhat))
// nn.BatchNorm2d training semantics, torchvision DenseNet (/root/reference/model.py:75-76 via torchvision).
#include "common.h"

namespace {

constexpr int NTH = 256;

template <typename T> struct Vec;
template <> struct Vec<float> {
  static constexpr int V = 4;
  __device__ static void load(const float* p, float (&f)[4]) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
  }
  __device__ static void store(float* p, const float (&f)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
  }
};
typedef unsigned short bf16_t;
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ unsigned short f2bf_rne(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7F800000u) == 0x7F800000u) return (unsigned short)(u >> 16);  // inf/nan: truncate
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
template <> struct Vec<bf16_t> {
  static constexpr int V = 8;
  __device__ static void load(const bf16_t* p, float (&f)[8]) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = __uint_as_float(w[i] << 16);
      f[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
    }
  }
  __device__ static void store(bf16_t* p, const float (&f)[8]) {
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = (unsigned)f2bf_rne(f[2 * i]) | ((unsigned)f2bf_rne(f[2 * i + 1]) << 16);
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
  }
};

template <typename T>
__global__ __launch_bounds__(NTH) void stats_partial_kernel(const T* __restrict__ x, long long ld, long long S, int C,
                                                            T* __restrict__ copy_out, long long ld_out,
                                                            float* __restrict__ partial, int rows_per_block) {
  constexpr int V = Vec<T>::V;
  __shared__ float smem[NTH * 2 * V];
  const int cvn_all = C / V;
  const int tile0 = blockIdx.y * NTH;
  const int cvn = min(cvn_all - tile0, NTH);
  const int rpi = NTH / cvn;
  const int t = threadIdx.x;
  const bool active = t < rpi * cvn;
  const int cv = tile0 + t % cvn, rloc = t / cvn;
  float s1[V], s2[V], k[V];
#pragma unroll
  for (int i = 0; i < V; ++i) s1[i] = s2[i] = 0.0f;
  if (active) {
    Vec<T>::load(x + (long long)cv * V, k);  // shift = first row's value of each channel
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = min(S, r0 + rows_per_block);
    for (long long r = r0 + rloc; r < r1; r += rpi) {
      float f[V];
      Vec<T>::load(x + r * ld + (long long)cv * V, f);
      if (copy_out) Vec<T>::store(copy_out + r * ld_out + (long long)cv * V, f);
#pragma unroll
      for (int i = 0; i < V; ++i) {
        const float d = f[i] - k[i];
        s1[i] += d;
        s2[i] = fmaf(d, d, s2[i]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < V; ++i) {
    smem[i * NTH + t] = s1[i];
    smem[(V + i) * NTH + t] = s2[i];
  }
  __syncthreads();
  if (active && rloc == 0) {
    // partial layout [C][nblk] float2: the finalize kernel then reads one channel's partials contiguously
    const long long nblk = gridDim.x;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      float a = 0.0f, b = 0.0f;
      for (int q = 0; q < rpi; ++q) {
        a += smem[i * NTH + t + q * cvn];
        b += smem[(V + i) * NTH + t + q * cvn];
      }
      *reinterpret_cast<float2*>(partial + (((long long)cv * V + i) * nblk + blockIdx.x) * 2) = make_float2(a, b);
    }
  }
}

// Finalize kernels: ONE WAVE PER CHANNEL.  Partials are stored [C][nblk] (float2), so a channel's nblk
// partial pairs are contiguous: each lane sums a strided subset (independent loads, no serial chain), then a
// shuffle tree in double.  Fixed summation order -> deterministic.  4 channels per 256-thread workgroup.
__device__ __forceinline__ void reduce_partials(const float* __restrict__ partial, int nblk, int c, double& a,
                                                double& b) {
  const int lane = threadIdx.x & 63;
  const float2* p = reinterpret_cast<const float2*>(partial) + (long long)c * nblk;
  float fa = 0.0f, fb = 0.0f;
  double da = 0.0, db = 0.0;
  int cnt = 0;
  // eight loads in flight per trip (clamped index; the surplus not added), consumed in the same order and grouping as the
  // one-load-per-trip form: that form was a chain of nblk / 64 dependent round trips (8 us for the transitions' 1568 partials)
  for (int i0 = lane; i0 < nblk; i0 += 8 * 64) {
    float2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[min(i0 + 64 * u, nblk - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (i0 + 64 * u >= nblk) break;
      fa += v[u].x;
      fb += v[u].y;
      if (++cnt == 4) {  // keep fp32 chains short (<= 4 terms) before widening
        da += (double)fa; db += (double)fb; fa = fb = 0.0f; cnt = 0;
      }
    }
  }
  da += (double)fa;
  db += (double)fb;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    da += __shfl_xor(da, o, 64);
    db += __shfl_xor(db, o, 64);
  }
  a = da;
  b = db;
}

template <typename T>
__global__ __launch_bounds__(256) void stats_finalize_kernel(const T* __restrict__ x,
                                                             const float* __restrict__ partial, int nblk, int C,
                                                             long long S, float eps, float* __restrict__ mean,
                                                             float* __restrict__ var, float* __restrict__ rstd) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return;
  double a, b;
  reduce_partials(partial, nblk, c, a, b);
  if ((threadIdx.x & 63) != 0) return;
  float k;
  if (sizeof(T) == 2) k = bf2f(((const bf16_t*)x)[c]); else k = ((const float*)x)[c];
  const double n = (double)S;
  const double m = a / n;
  double v = b / n - m * m;
  if (v < 0.0) v = 0.0;
  mean[c] = (float)((double)k + m);
  var[c] = (float)v;
  rstd[c] = (float)(1.0 / sqrt(v + (double)eps));
}

template <typename T>
__global__ __launch_bounds__(NTH) void act_fwd_kernel(const T* __restrict__ x, long long ldx, long long S, int C,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      int relu, T* __restrict__ y, long long ldy, int rows_per_block) {
  constexpr int V = Vec<T>::V;
  const int cvn_all = C / V;
  const int tile0 = blockIdx.y * NTH;
  const int cvn = min(cvn_all - tile0, NTH);
  const int rpi = NTH / cvn;
  const int t = threadIdx.x;
  if (t >= rpi * cvn) return;
  const int cv = tile0 + t % cvn, rloc = t / cvn;
  float sc[V], sh[V];
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int c = cv * V + i;
    sc[i] = gamma[c] * rstd[c];
    sh[i] = fmaf(-mean[c], sc[i], beta[c]);
  }
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = min(S, r0 + rows_per_block);
  for (long long r = r0 + rloc; r < r1; r += rpi) {
    float f[V];
    Vec<T>::load(x + r * ldx + (long long)cv * V, f);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      float v = fmaf(f[i], sc[i], sh[i]);
      f[i] = relu ? fmaxf(v, 0.0f) : v;
    }
    Vec<T>::store(y + r * ldy + (long long)cv * V, f);
  }
}

// Transition prologue (torchvision _Transition: norm -> relu -> conv1x1 -> AvgPool2d(2, 2)).  The 1x1 convolution is
// linear and per pixel, so it commutes with the average pool: p = avgpool(relu(bn(x))) is formed HERE (S*C read,
// S/4*C written) and the convolution then runs on a quarter of the pixels.  Rows of y are pooled pixels.
template <typename T>
__global__ __launch_bounds__(NTH) void act_avgpool_fwd_kernel(const T* __restrict__ x, long long ldx, long long SP,
                                                              int H, int W, int C, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, T* __restrict__ y,
                                                              long long ldy, int rows_per_block) {
  constexpr int V = Vec<T>::V;
  const int cvn_all = C / V;
  const int tile0 = blockIdx.y * NTH;
  const int cvn = min(cvn_all - tile0, NTH);
  const int rpi = NTH / cvn;
  const int t = threadIdx.x;
  if (t >= rpi * cvn) return;
  const int cv = tile0 + t % cvn, rloc = t / cvn;
  float sc[V], sh[V];
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int c = cv * V + i;
    sc[i] = gamma[c] * rstd[c];
    sh[i] = fmaf(-mean[c], sc[i], beta[c]);
  }
  const int OW = W >> 1, OH = H >> 1;
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = min(SP, r0 + rows_per_block);
  for (long long r = r0 + rloc; r < r1; r += rpi) {
    const int ox = (int)(r % OW);
    const long long tt = r / OW;
    const int oy = (int)(tt % OH);
    const long long n = tt / OH;
    const long long base = ((n * H + 2 * oy) * W + 2 * ox);
    float a[V], b[V], c[V], d[V];
    Vec<T>::load(x + base * ldx + (long long)cv * V, a);
    Vec<T>::load(x + (base + 1) * ldx + (long long)cv * V, b);
    Vec<T>::load(x + (base + W) * ldx + (long long)cv * V, c);
    Vec<T>::load(x + (base + W + 1) * ldx + (long long)cv * V, d);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const float va = fmaxf(fmaf(a[i], sc[i], sh[i]), 0.0f), vb = fmaxf(fmaf(b[i], sc[i], sh[i]), 0.0f);
      const float vc = fmaxf(fmaf(c[i], sc[i], sh[i]), 0.0f), vd = fmaxf(fmaf(d[i], sc[i], sh[i]), 0.0f);
      a[i] = 0.25f * ((va + vb) + (vc + vd));
    }
    Vec<T>::store(y + r * ldy + (long long)cv * V, a);
  }
}

// POOL: dy is the gradient of the 2x2-average-pooled activation (rows = pooled pixels); the gradient reaching
// pixel r of the (n, H, W) map is dy[pooled(r)] / 4 (transition: norm -> relu -> avgpool, see bn_act_avgpool_fwd)
// Odd H / W (nn.AvgPool2d floors: the last row / column of the map belongs to no window): -1, the pixel receives no
// gradient through the pool (its BatchNorm-backward terms are still applied: the statistics cover every pixel).
__device__ __forceinline__ long long pooled_row(long long r, int H, int W) {
  const int xw = (int)(r % W);
  const long long t = r / W;
  const int yh = (int)(t % H);
  const long long n = t / H;
  if ((xw >> 1) >= (W >> 1) || (yh >> 1) >= (H >> 1)) return -1;
  return (n * (H >> 1) + (yh >> 1)) * (W >> 1) + (xw >> 1);
}

template <typename T, int MODE = 0>
__global__ __launch_bounds__(NTH) void act_bwd_reduce_kernel(const T* __restrict__ dy, long long lddy,
                                                             const T* __restrict__ x, long long ldx, long long S,
                                                             int C, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, int relu,
                                                             float* __restrict__ partial, int rows_per_block,
                                                             int H = 0, int W = 0) {
  constexpr bool POOL = MODE == 1;
  constexpr int V = Vec<T>::V;
  __shared__ float smem[NTH * 2 * V];
  const int cvn_all = C / V;
  const int tile0 = blockIdx.y * NTH;
  const int cvn = min(cvn_all - tile0, NTH);
  const int rpi = NTH / cvn;
  const int t = threadIdx.x;
  const bool active = t < rpi * cvn;
  const int cv = tile0 + t % cvn, rloc = t / cvn;
  float s1[V], s2[V];
#pragma unroll
  for (int i = 0; i < V; ++i) s1[i] = s2[i] = 0.0f;
  if (active) {
    float sc[V], sh[V], mu[V], rs[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int c = cv * V + i;
      mu[i] = mean[c];
      rs[i] = rstd[c];
      sc[i] = gamma[c] * rs[i];
      sh[i] = fmaf(-mu[i], sc[i], beta[c]);
    }
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = min(S, r0 + rows_per_block);
    for (long long r = r0 + rloc; r < r1; r += rpi) {
      float f[V], g[V];
      Vec<T>::load(x + r * ldx + (long long)cv * V, f);
      const long long pr = POOL ? pooled_row(r, H, W) : r;
      if (pr >= 0) Vec<T>::load(dy + pr * lddy + (long long)cv * V, g);
#pragma unroll
      for (int i = 0; i < V; ++i) {
        if (POOL) g[i] = pr >= 0 ? g[i] * 0.25f : 0.0f;
        const float gi = (relu && fmaf(f[i], sc[i], sh[i]) <= 0.0f) ? 0.0f : g[i];
        s1[i] += gi;
        s2[i] = fmaf(gi, (f[i] - mu[i]) * rs[i], s2[i]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < V; ++i) {
    smem[i * NTH + t] = s1[i];
    smem[(V + i) * NTH + t] = s2[i];
  }
  __syncthreads();
  if (active && rloc == 0) {
    // partial layout [C][nblk] float2: the finalize kernel then reads one channel's partials contiguously
    const long long nblk = gridDim.x;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      float a = 0.0f, b = 0.0f;
      for (int q = 0; q < rpi; ++q) {
        a += smem[i * NTH + t + q * cvn];
        b += smem[(V + i) * NTH + t + q * cvn];
      }
      *reinterpret_cast<float2*>(partial + (((long long)cv * V + i) * nblk + blockIdx.x) * 2) = make_float2(a, b);
    }
  }
}

// dbeta = sum g ; dgamma = sum g*xhat ; coef[c] = (sum g / S, sum g*xhat / S)
__global__ __launch_bounds__(256) void act_bwd_finalize_kernel(const float* __restrict__ partial, int nblk, int C,
                                                               long long S, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, float* __restrict__ coef,
                                                               int accumulate_params) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return;
  double a, b;
  reduce_partials(partial, nblk, c, a, b);
  if ((threadIdx.x & 63) != 0) return;
  if (accumulate_params) {  // straight into the parameters' .grad (flat optimizer bucket)
    dbeta[c] += (float)a;
    dgamma[c] += (float)b;
  } else {
    dbeta[c] = (float)a;
    dgamma[c] = (float)b;
  }
  coef[2 * c] = (float)(a / (double)S);
  coef[2 * c + 1] = (float)(b / (double)S);
}

template <typename T, int MODE = 0>
__global__ __launch_bounds__(NTH) void act_bwd_dx_kernel(const T* __restrict__ dy, long long lddy,
                                                         const T* __restrict__ x, long long ldx, long long S, int C,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta,
                                                         const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, int relu,
                                                         const float* __restrict__ coef, T* dx, long long lddx,
                                                         int accumulate, int rows_per_block, int H = 0, int W = 0) {
  constexpr bool POOL = MODE == 1;
  constexpr int V = Vec<T>::V;
  const int cvn_all = C / V;
  const int tile0 = blockIdx.y * NTH;
  const int cvn = min(cvn_all - tile0, NTH);
  const int rpi = NTH / cvn;
  const int t = threadIdx.x;
  if (t >= rpi * cvn) return;
  const int cv = tile0 + t % cvn, rloc = t / cvn;
  float sc[V], sh[V], mu[V], rs[V], c1[V], c2[V];
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int c = cv * V + i;
    mu[i] = mean[c];
    rs[i] = rstd[c];
    sc[i] = gamma[c] * rs[i];
    sh[i] = fmaf(-mu[i], sc[i], beta[c]);
    c1[i] = coef[2 * c];
    c2[i] = coef[2 * c + 1];
  }
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = min(S, r0 + rows_per_block);
  for (long long r = r0 + rloc; r < r1; r += rpi) {
    float f[V], g[V], o[V];
    Vec<T>::load(x + r * ldx + (long long)cv * V, f);
    const long long pr = POOL ? pooled_row(r, H, W) : r;
    if (pr >= 0) Vec<T>::load(dy + pr * lddy + (long long)cv * V, g);
    if (accumulate) Vec<T>::load(dx + r * lddx + (long long)cv * V, o);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      if (POOL) g[i] = pr >= 0 ? g[i] * 0.25f : 0.0f;
      const float gi = (relu && fmaf(f[i], sc[i], sh[i]) <= 0.0f) ? 0.0f : g[i];
      const float xh = (f[i] - mu[i]) * rs[i];
      const float d = sc[i] * (gi - c1[i] - xh * c2[i]);
      o[i] = accumulate ? o[i] + d : d;
    }
    Vec<T>::store(dx + r * lddx + (long long)cv * V, o);
  }
}

// dst[i] += (float)src[i] over n elements in storage order (both tensors dense with identical strides): hands
// a bf16 / fp32 weight gradient to the fp32 .grad view of the flat optimizer bucket in one small launch.
template <typename T>
__global__ __launch_bounds__(256) void accum_kernel(float* __restrict__ dst, const T* __restrict__ src, long long n) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float v;
    if (sizeof(T) == 2) v = bf2f(((const bf16_t*)src)[i]); else v = ((const float*)src)[i];
    dst[i] += v;
  }
}

inline int vec_of(int dtype) { return dtype == 1 ? 8 : 4; }
inline bool ok_layout(const void* p, long long ld, int C, int dtype) {
  const int V = vec_of(dtype);
  return p && (C % V == 0) && (ld % V == 0) && ((reinterpret_cast<uintptr_t>(p) & 15u) == 0);
}
// rows per workgroup so that the grid has at most 1024 row-blocks and each does >= 8 iterations
inline void plan(long long S, int C, int dtype, int* nblk, int* rows_per_block, int* tiles) {
  const int V = vec_of(dtype);
  const int cvn_all = C / V;
  const int cvn = cvn_all < NTH ? cvn_all : NTH;
  const int rpi = NTH / cvn;
  long long rpb = (long long)rpi * 8;
  long long nb = (S + rpb - 1) / rpb;
  if (nb > 1024) {
    nb = 1024;
    rpb = (S + nb - 1) / nb;
    rpb = (rpb + rpi - 1) / rpi * rpi;
    nb = (S + rpb - 1) / rpb;
  }
  *nblk = (int)nb;
  *rows_per_block = (int)rpb;
  *tiles = (cvn_all + NTH - 1) / NTH;
}

}  // namespace

extern "C" int64_t mcl_bn_workspace_floats(int64_t S, int32_t C, int32_t dtype) {
  if (S <= 0 || C <= 0 || (dtype != 0 && dtype != 1) || C % vec_of(dtype) != 0) return -1;
  int nblk, rpb, tiles;
  plan(S, C, dtype, &nblk, &rpb, &tiles);
  return (int64_t)nblk * 2 * C + 2 * C;
}

extern "C" int mcl_bn_stats(const void* x, int64_t ld, int64_t S, int32_t C, int32_t dtype, void* copy_out,
                            int64_t ld_out, float* workspace, float eps, float* mean, float* var, float* rstd,
                            mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!workspace || !mean || !var || !rstd || S <= 0 || C <= 0 || (dtype != 0 && dtype != 1)) return MCL_EINVAL;
  if (!ok_layout(x, ld, C, dtype) || (copy_out && !ok_layout(copy_out, ld_out, C, dtype))) return MCL_EUNSUPPORTED;
  int nblk, rpb, tiles;
  plan(S, C, dtype, &nblk, &rpb, &tiles);
  hipStream_t st = mcl_stream(stream);
  dim3 grid(nblk, tiles);
  if (dtype == 1) {
    hipLaunchKernelGGL(stats_partial_kernel<bf16_t>, grid, dim3(NTH), 0, st, (const bf16_t*)x, (long long)ld,
                       (long long)S, C, (bf16_t*)copy_out, (long long)ld_out, workspace, rpb);
    hipLaunchKernelGGL(stats_finalize_kernel<bf16_t>, dim3((C + 3) / 4), dim3(256), 0, st, (const bf16_t*)x,
                       workspace, nblk, C, (long long)S, eps, mean, var, rstd);
  } else {
    hipLaunchKernelGGL(stats_partial_kernel<float>, grid, dim3(NTH), 0, st, (const float*)x, (long long)ld,
                       (long long)S, C, (float*)copy_out, (long long)ld_out, workspace, rpb);
    hipLaunchKernelGGL(stats_finalize_kernel<float>, dim3((C + 3) / 4), dim3(256), 0, st, (const float*)x,
                       workspace, nblk, C, (long long)S, eps, mean, var, rstd);
  }
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_bn_act_fwd(const void* x, int64_t ldx, int64_t S, int32_t C, int32_t dtype, const float* gamma,
                              const float* beta, const float* mean, const float* rstd, int32_t relu, void* y,
                              int64_t ldy, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!gamma || !beta || !mean || !rstd || S <= 0 || C <= 0 || (dtype != 0 && dtype != 1)) return MCL_EINVAL;
  if (!ok_layout(x, ldx, C, dtype) || !ok_layout(y, ldy, C, dtype)) return MCL_EUNSUPPORTED;
  int nblk, rpb, tiles;
  plan(S, C, dtype, &nblk, &rpb, &tiles);
  dim3 grid(nblk, tiles);
  if (dtype == 1)
    hipLaunchKernelGGL(act_fwd_kernel<bf16_t>, grid, dim3(NTH), 0, mcl_stream(stream), (const bf16_t*)x,
                       (long long)ldx, (long long)S, C, gamma, beta, mean, rstd, relu, (bf16_t*)y, (long long)ldy, rpb);
  else
    hipLaunchKernelGGL(act_fwd_kernel<float>, grid, dim3(NTH), 0, mcl_stream(stream), (const float*)x,
                       (long long)ldx, (long long)S, C, gamma, beta, mean, rstd, relu, (float*)y, (long long)ldy, rpb);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_bn_act_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, int64_t S, int32_t C,
                              int32_t dtype, const float* gamma, const float* beta, const float* mean,
                              const float* rstd, int32_t relu, float* workspace, float* dgamma, float* dbeta,
                              int32_t accumulate_params, void* dx, int64_t lddx, int32_t accumulate,
                              mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!gamma || !beta || !mean || !rstd || !workspace || !dgamma || !dbeta || S <= 0 || C <= 0 ||
      (dtype != 0 && dtype != 1))
    return MCL_EINVAL;
  if (!ok_layout(dy, lddy, C, dtype) || !ok_layout(x, ldx, C, dtype) || !ok_layout(dx, lddx, C, dtype))
    return MCL_EUNSUPPORTED;
  int nblk, rpb, tiles;
  plan(S, C, dtype, &nblk, &rpb, &tiles);
  dim3 grid(nblk, tiles);
  hipStream_t st = mcl_stream(stream);
  float* coef = workspace + (long long)nblk * 2 * C;
  if (dtype == 1) {
    hipLaunchKernelGGL(act_bwd_reduce_kernel<bf16_t>, grid, dim3(NTH), 0, st, (const bf16_t*)dy, (long long)lddy,
                       (const bf16_t*)x, (long long)ldx, (long long)S, C, gamma, beta, mean, rstd, relu, workspace, rpb);
    hipLaunchKernelGGL(act_bwd_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, st, workspace, nblk, C,
                       (long long)S, dgamma, dbeta, coef, accumulate_params);
    hipLaunchKernelGGL(act_bwd_dx_kernel<bf16_t>, grid, dim3(NTH), 0, st, (const bf16_t*)dy, (long long)lddy,
                       (const bf16_t*)x, (long long)ldx, (long long)S, C, gamma, beta, mean, rstd, relu, coef,
                       (bf16_t*)dx, (long long)lddx, accumulate, rpb);
  } else {
    hipLaunchKernelGGL(act_bwd_reduce_kernel<float>, grid, dim3(NTH), 0, st, (const float*)dy, (long long)lddy,
                       (const float*)x, (long long)ldx, (long long)S, C, gamma, beta, mean, rstd, relu, workspace, rpb);
    hipLaunchKernelGGL(act_bwd_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, st, workspace, nblk, C,
                       (long long)S, dgamma, dbeta, coef, accumulate_params);
    hipLaunchKernelGGL(act_bwd_dx_kernel<float>, grid, dim3(NTH), 0, st, (const float*)dy, (long long)lddy,
                       (const float*)x, (long long)ldx, (long long)S, C, gamma, beta, mean, rstd, relu, coef,
                       (float*)dx, (long long)lddx, accumulate, rpb);
  }
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_bn_act_avgpool_fwd(const void* x, int64_t ldx, int32_t N, int32_t H, int32_t W, int32_t C,
                                      const float* gamma, const float* beta, const float* mean, const float* rstd,
                                      void* y, int64_t ldy, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!gamma || !beta || !mean || !rstd || N <= 0 || H <= 0 || W <= 0 || C <= 0) return MCL_EINVAL;
  if (H < 2 || W < 2 || !ok_layout(x, ldx, C, 1) || !ok_layout(y, ldy, C, 1)) return MCL_EUNSUPPORTED;
  const long long SP = (long long)N * (H / 2) * (W / 2);
  int nblk, rpb, tiles;
  plan(SP, C, 1, &nblk, &rpb, &tiles);
  hipLaunchKernelGGL(act_avgpool_fwd_kernel<bf16_t>, dim3(nblk, tiles), dim3(NTH), 0, mcl_stream(stream),
                     (const bf16_t*)x, (long long)ldx, SP, H, W, C, gamma, beta, mean, rstd, (bf16_t*)y,
                     (long long)ldy, rpb);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_bn_act_avgpool_bwd(const void* dp, int64_t lddp, const void* x, int64_t ldx, int32_t N, int32_t H,
                                      int32_t W, int32_t C, const float* gamma, const float* beta, const float* mean,
                                      const float* rstd, float* workspace, float* dgamma, float* dbeta,
                                      int32_t accumulate_params, void* dx, int64_t lddx, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!gamma || !beta || !mean || !rstd || !workspace || !dgamma || !dbeta || N <= 0 || H <= 0 || W <= 0 || C <= 0)
    return MCL_EINVAL;
  if (H < 2 || W < 2 || !ok_layout(dp, lddp, C, 1) || !ok_layout(x, ldx, C, 1) || !ok_layout(dx, lddx, C, 1))
    return MCL_EUNSUPPORTED;
  const long long S = (long long)N * H * W;
  int nblk, rpb, tiles;
  plan(S, C, 1, &nblk, &rpb, &tiles);
  dim3 grid(nblk, tiles);
  hipStream_t st = mcl_stream(stream);
  float* coef = workspace + (long long)nblk * 2 * C;
  hipLaunchKernelGGL((act_bwd_reduce_kernel<bf16_t, 1>), grid, dim3(NTH), 0, st, (const bf16_t*)dp,
                     (long long)lddp, (const bf16_t*)x, (long long)ldx, S, C, gamma, beta, mean, rstd, 1, workspace, rpb,
                     H, W);
  hipLaunchKernelGGL(act_bwd_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, st, workspace, nblk, C, S, dgamma,
                     dbeta, coef, accumulate_params);
  hipLaunchKernelGGL((act_bwd_dx_kernel<bf16_t, 1>), grid, dim3(NTH), 0, st, (const bf16_t*)dp, (long long)lddp,
                     (const bf16_t*)x, (long long)ldx, S, C, gamma, beta, mean, rstd, 1, coef, (bf16_t*)dx,
                     (long long)lddx, 0, rpb, H, W);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_accum_into_f32(float* dst, const void* src, int64_t n, int32_t src_dtype, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!dst || !src || n <= 0 || (src_dtype != 0 && src_dtype != 1)) return MCL_EINVAL;
  long long blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  if (src_dtype == 1)
    hipLaunchKernelGGL(accum_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, mcl_stream(stream), dst,
                       (const bf16_t*)src, (long long)n);
  else
    hipLaunchKernelGGL(accum_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, mcl_stream(stream), dst,
                       (const float*)src, (long long)n);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}
