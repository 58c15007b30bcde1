// The small pieces that kept the benched training step from being 100 % this library's kernels: the tail
// of the DenseNet feature extractor, norm5 -> adaptive_avg_pool2d((1, 1)) -> flatten (/root/reference/model.py:81-85; no ReLU
// between them), forward and backward; nn.BatchNorm2d's running-statistics bookkeeping for ALL 121 BatchNorm layers of the
// network in two launches; the input image's fp32 -> bf16 NHWC conversion; a zero fill.  All HBM / latency trivial.
#include "common.h"

namespace {

typedef unsigned short bf16_t;
__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xFFFF0000u); }
__device__ __forceinline__ unsigned f2bf_rne(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7F800000u) == 0x7F800000u) return u >> 16;   // inf / nan: truncate
  u += 0x7FFFu + ((u >> 16) & 1u);
  return u >> 16;
}
__device__ __forceinline__ unsigned pack_rne(float a, float b) { return f2bf_rne(a) | (f2bf_rne(b) << 16); }

// ---------------------------------------------------------------------------------------------------------------------
// norm + global average pool.  The pool of an affine map is the affine map of the pool:
//     out[b][c] = gamma*rstd*(mean_hw x[b][hw][c] - mean[c]) + beta            (fp32, never rounded to bf16 per pixel)
// One workgroup = one image x 512 channels: 64 chunk columns (8 channels, 16 bytes) x 4 pixel groups; the raw pooled mean
// xm[b][c] is kept for the backward.
__global__ __launch_bounds__(256) void bn_gap_fwd_kernel(const bf16_t* __restrict__ x, long long ldx, int HW, int C,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         float* __restrict__ out, float* __restrict__ xm) {
  __shared__ float red[4][64][8];
  const int b = blockIdx.y, col = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int c0 = (blockIdx.x * 64 + col) * 8;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
  if (c0 < C) {
    const bf16_t* p = x + (long long)b * HW * ldx + c0;
    for (int s = grp; s < HW; s += 4) {
      const uint4 v = *reinterpret_cast<const uint4*>(p + (long long)s * ldx);
      const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc[2 * u] += bf_lo(w[u]);
        acc[2 * u + 1] += bf_hi(w[u]);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[grp][col][e] = acc[e];
  __syncthreads();
  if (grp != 0 || c0 >= C) return;
  const float inv = 1.0f / (float)HW;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = c0 + e;
    const float m = ((red[0][col][e] + red[1][col][e]) + (red[2][col][e] + red[3][col][e])) * inv;
    const float sc = gamma[c] * rstd[c];
    out[(long long)b * C + c] = fmaf(m - mean[c], sc, beta[c]);
    if (xm) xm[(long long)b * C + c] = m;
  }
}

// backward, step 1: per channel over the B images (fixed order): dbeta = sum_b g, dgamma = sum_b g*xhat_m; the two means of
// the train-mode BatchNorm backward over S = B*HW positions are dbeta / S and dgamma / S (dy is constant g/HW per image).
__global__ __launch_bounds__(256) void bn_gap_bwd_sums_kernel(const float* __restrict__ g, const float* __restrict__ xm, int B,
                                                              int C, int HW, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int accumulate,
                                                              float* __restrict__ coef) {
  // 64 channels per workgroup x 4 image groups (b = grp, grp + 4, ..: eight independent load pairs in flight per thread),
  // merged in fixed order through LDS
  __shared__ double red[2][4][64];
  const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + col;
  const bool live = c < C;
  const float mu = live ? mean[c] : 0.0f, rs = live ? rstd[c] : 0.0f;
  double a = 0.0, q = 0.0;
  if (live) {
#pragma unroll 8
    for (int b = grp; b < B; b += 4) {
      const float gv = g[(long long)b * C + c];
      a += (double)gv;
      q += (double)gv * (double)((xm[(long long)b * C + c] - mu) * rs);
    }
  }
  red[0][grp][col] = a;
  red[1][grp][col] = q;
  __syncthreads();
  if (grp != 0 || !live) return;
  a = (red[0][0][col] + red[0][1][col]) + (red[0][2][col] + red[0][3][col]);
  q = (red[1][0][col] + red[1][1][col]) + (red[1][2][col] + red[1][3][col]);
  if (accumulate) {
    dbeta[c] += (float)a;
    dgamma[c] += (float)q;
  } else {
    dbeta[c] = (float)a;
    dgamma[c] = (float)q;
  }
  const double S = (double)B * (double)HW;
  coef[2 * c] = (float)(a / S);
  coef[2 * c + 1] = (float)(q / S);
}

// step 2: dx[b][hw][c] = gamma*rstd*(g[b][c]/HW - c1 - xhat*c2), one 16-byte chunk per thread and position
__global__ __launch_bounds__(256) void bn_gap_bwd_dx_kernel(const float* __restrict__ g, const bf16_t* __restrict__ x,
                                                            long long ldx, int B, int HW, int C,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ coef,
                                                            bf16_t* __restrict__ dx, long long lddx) {
  const int cpr = C >> 3;
  const long long n = (long long)B * HW * cpr;
  const float inv = 1.0f / (float)HW;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const long long row = i / cpr;
    const int c0 = (int)(i - row * cpr) * 8;
    const int b = (int)(row / HW);
    const uint4 v = *reinterpret_cast<const uint4*>(x + row * ldx + c0);
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    const float4 g0 = *reinterpret_cast<const float4*>(g + (long long)b * C + c0);
    const float4 g1 = *reinterpret_cast<const float4*>(g + (long long)b * C + c0 + 4);
    const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = c0 + e;
      const float xv = (e & 1) ? bf_hi(w[e >> 1]) : bf_lo(w[e >> 1]);
      const float rs = rstd[c];
      o[e] = gamma[c] * rs * (gv[e] * inv - coef[2 * c] - (xv - mean[c]) * rs * coef[2 * c + 1]);
    }
    *reinterpret_cast<uint4*>(dx + row * lddx + c0) =
        make_uint4(pack_rne(o[0], o[1]), pack_rne(o[2], o[3]), pack_rne(o[4], o[5]), pack_rne(o[6], o[7]));
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// nn.BatchNorm2d running statistics (momentum m, unbiased variance = var * n / (n - 1)) and num_batches_tracked for up to
// 64 BatchNorm layers per launch; the table of pointers travels BY VALUE in the kernel arguments (nothing to upload, nothing
// to keep alive; a captured launch replays with the pointers of its capture -- the graph's static buffers).
struct RunEntry {
  float* rm;
  float* rv;
  const float* mean;
  const float* var;
  long long* nbt;
  int C;
  float factor;
  float momentum;
  int pad_;
};
constexpr int RUN_MAX = 64;
struct RunBatch {
  RunEntry e[RUN_MAX];
};

__global__ __launch_bounds__(256) void bn_running_kernel(RunBatch t) {
  const RunEntry& e = t.e[blockIdx.x];
  for (int c = threadIdx.x; c < e.C; c += 256) {
    const float rm = e.rm[c], rv = e.rv[c];
    e.rm[c] = rm + e.momentum * (e.mean[c] - rm);                 // Tensor.lerp_(mean, momentum)
    e.rv[c] = rv + e.momentum * (e.var[c] * e.factor - rv);
  }
  if (threadIdx.x == 0 && e.nbt != nullptr) *e.nbt += 1;
}

// eval mode: rstd[c] = 1 / sqrt(running_var[c] + eps) for up to 64 BatchNorm layers per launch (same by-value pointer table)
struct RstdEntry {
  const float* var;
  float* out;
  int C;
  float eps;
};
struct RstdBatch {
  RstdEntry e[RUN_MAX];
};
__global__ __launch_bounds__(256) void bn_rstd_kernel(RstdBatch t) {
  const RstdEntry& e = t.e[blockIdx.x];
  for (int c = threadIdx.x; c < e.C; c += 256) e.out[c] = 1.0f / sqrtf(e.var[c] + e.eps);
}

// ---------------------------------------------------------------------------------------------------------------------
// image (B, C, H, W) fp32 with arbitrary element strides -> bf16 NHWC contiguous.
// Generic form: 8 consecutive NHWC elements per thread, one scalar load each (any strides).
__global__ __launch_bounds__(256) void image_to_bf16_nhwc_kernel(const float* __restrict__ x, long long sb, long long sc,
                                                                 long long sy, long long sx, int C, int H, int W,
                                                                 long long n, bf16_t* __restrict__ y) {
  const long long i8 = ((long long)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i8 >= n) return;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const long long i = i8 + e;
    if (i < n) {
      const long long pix = i / C;
      const int c = (int)(i - pix * C);
      const long long rowi = pix / W;
      const int xx = (int)(pix - rowi * W);
      const long long b = rowi / H;
      const int yy = (int)(rowi - b * H);
      v[e] = x[b * sb + c * sc + yy * sy + xx * sx];
    } else {
      v[e] = 0.0f;
    }
  }
  if (i8 + 8 <= n) {
    *reinterpret_cast<uint4*>(y + i8) = make_uint4(pack_rne(v[0], v[1]), pack_rne(v[2], v[3]), pack_rne(v[4], v[5]),
                                                   pack_rne(v[6], v[7]));
  } else {
    for (int e = 0; e < 8 && i8 + e < n; ++e) y[i8 + e] = (bf16_t)f2bf_rne(v[e]);
  }
}

// the input already IS NHWC-contiguous (a channels-last tensor: what bench.py holds): a plain cast, 2 x 16-byte loads and one
// 16-byte store per thread and trip
__global__ __launch_bounds__(256) void cast_f32_bf16_flat_kernel(const float* __restrict__ x, long long n8,
                                                                 bf16_t* __restrict__ y) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    const float4 a = reinterpret_cast<const float4*>(x)[2 * i], b = reinterpret_cast<const float4*>(x)[2 * i + 1];
    reinterpret_cast<uint4*>(y)[i] = make_uint4(pack_rne(a.x, a.y), pack_rne(a.z, a.w), pack_rne(b.x, b.y), pack_rne(b.z, b.w));
  }
}

// planar input (NCHW from a DataLoader: unit x stride), 3 channels: a thread converts 8 consecutive pixels of one image row
// -- three pairs of 16-byte loads (one per channel plane) interleaved into 48 contiguous output bytes
__global__ __launch_bounds__(256) void image_planar3_to_bf16_nhwc_kernel(const float* __restrict__ x, long long sb,
                                                                         long long sc, long long sy, int H, int W8,
                                                                         long long n_units, bf16_t* __restrict__ y) {
  for (long long u = (long long)blockIdx.x * 256 + threadIdx.x; u < n_units; u += (long long)gridDim.x * 256) {
    const long long rowi = u / W8;
    const int x8 = (int)(u - rowi * W8);
    const long long b = rowi / H;
    const int yy = (int)(rowi - b * H);
    const float* p = x + b * sb + yy * sy + x8 * 8;
    float v[3][8];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float4 a = *reinterpret_cast<const float4*>(p + c * sc), q = *reinterpret_cast<const float4*>(p + c * sc + 4);
      v[c][0] = a.x; v[c][1] = a.y; v[c][2] = a.z; v[c][3] = a.w; v[c][4] = q.x; v[c][5] = q.y; v[c][6] = q.z; v[c][7] = q.w;
    }
    float o[24];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int c = 0; c < 3; ++c) o[3 * i + c] = v[c][i];
    uint4* out = reinterpret_cast<uint4*>(y + u * 24);
#pragma unroll
    for (int k = 0; k < 3; ++k)
      out[k] = make_uint4(pack_rne(o[8 * k], o[8 * k + 1]), pack_rne(o[8 * k + 2], o[8 * k + 3]),
                          pack_rne(o[8 * k + 4], o[8 * k + 5]), pack_rne(o[8 * k + 6], o[8 * k + 7]));
  }
}

// n dwords of zeros; 16-byte stores once the pointer is aligned
__global__ __launch_bounds__(256) void fill_zero_kernel(unsigned* __restrict__ p, long long n) {
  const long long head = min(n, (long long)((16 - (reinterpret_cast<uintptr_t>(p) & 15u)) & 15u) >> 2);
  const long long n4 = (n - head) >> 2;
  uint4* q = reinterpret_cast<uint4*>(p + head);
  const long long tid = (long long)blockIdx.x * 256 + threadIdx.x, stride = (long long)gridDim.x * 256;
  for (long long i = tid; i < n4; i += stride) q[i] = make_uint4(0u, 0u, 0u, 0u);
  if (tid < head) p[tid] = 0u;
  const long long tail0 = head + (n4 << 2);
  if (tail0 + tid < n && tid < 4) p[tail0 + tid] = 0u;
}

__global__ __launch_bounds__(256) void scale2_kernel(const float* __restrict__ a, long long na, const float* __restrict__ b,
                                                     long long nb, const float* __restrict__ s, float* __restrict__ ya,
                                                     float* __restrict__ yb) {
  const float sv = s[0];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < na + nb; i += (long long)gridDim.x * 256) {
    if (i < na) ya[i] = a[i] * sv;
    else yb[i - na] = b[i - na] * sv;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// nn.Dropout(p) and nn.GELU() as own elementwise kernels for the dropout > 0 path of FeedForward / ProjectionHead
// (/root/reference/model.py:25-29,156,164; the reference never activates it: model.py:217 hard-codes 0.).  The mask comes
// from a counter-based generator (one 64-bit mix of (seed, element index) per element: reproducible for a given seed, no
// state), is kept as one byte per element for the backward, and kept elements are scaled by 1 / (1 - p).
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ __launch_bounds__(256) void dropout_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          unsigned char* __restrict__ mask, long long n, float p, float scale,
                                                          unsigned long long seed) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const unsigned long long r = mix64(seed ^ mix64((unsigned long long)i));
    const float u = (float)(r >> 40) * (1.0f / 16777216.0f);          // 24 random bits -> [0, 1)
    const unsigned char keep = u >= p ? 1 : 0;
    mask[i] = keep;
    y[i] = keep ? x[i] * scale : 0.0f;
  }
}
__global__ __launch_bounds__(256) void dropout_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ mask,
                                                          float* __restrict__ dx, long long n, float scale) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    dx[i] = mask[i] ? dy[i] * scale : 0.0f;
}
__global__ __launch_bounds__(256) void gelu_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                   float* __restrict__ out, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    out[i] = dy ? dy[i] * gelu_erf_grad(x[i]) : gelu_erf(x[i]);
}
__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ y, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] = a[i] + b[i];
}

// debug: the GPU's wall clock (s_memrealtime, 100 MHz) when the stream reaches this point -- an UNTRACED timeline of a replayed
// step graph (a kernel trace inflates launches and changes how the two hardware queues interleave)
__global__ void stamp_kernel(unsigned long long* __restrict__ buf, int idx) {
  if (threadIdx.x == 0) buf[idx] = wall_clock64();
}

}  // namespace

namespace {
inline unsigned ew_grid(long long n) {
  long long b = (n + 255) / 256;
  return (unsigned)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}
}  // namespace

extern "C" int mcl_dropout_fwd(const float* x, float* y, void* mask, int64_t n, float p, uint64_t seed, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!x || !y || !mask || n <= 0 || !(p >= 0.0f && p < 1.0f)) return MCL_EINVAL;
  hipLaunchKernelGGL(dropout_fwd_kernel, dim3(ew_grid(n)), dim3(256), 0, mcl_stream(stream), x, y, (unsigned char*)mask,
                     (long long)n, p, 1.0f / (1.0f - p), (unsigned long long)seed);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_dropout_bwd(const float* dy, const void* mask, float* dx, int64_t n, float p, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!dy || !dx || !mask || n <= 0 || !(p >= 0.0f && p < 1.0f)) return MCL_EINVAL;
  hipLaunchKernelGGL(dropout_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, mcl_stream(stream), dy, (const unsigned char*)mask,
                     dx, (long long)n, 1.0f / (1.0f - p));
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_gelu_f32(const float* x, const float* dy, float* out, int64_t n, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!x || !out || n <= 0) return MCL_EINVAL;
  hipLaunchKernelGGL(gelu_kernel, dim3(ew_grid(n)), dim3(256), 0, mcl_stream(stream), x, dy, out, (long long)n);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_add_f32(const float* a, const float* b, float* y, int64_t n, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!a || !b || !y || n <= 0) return MCL_EINVAL;
  hipLaunchKernelGGL(add_kernel, dim3(ew_grid(n)), dim3(256), 0, mcl_stream(stream), a, b, y, (long long)n);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_stamp(void* buf, int32_t idx, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!buf || idx < 0) return MCL_EINVAL;
  hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, mcl_stream(stream), (unsigned long long*)buf, idx);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_scale2_f32(const float* a, int64_t na, const float* b, int64_t nb, const float* s, float* ya, float* yb,
                              mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!a || !b || !s || !ya || !yb || na < 0 || nb < 0) return MCL_EINVAL;
  if (na + nb == 0) return MCL_OK;
  long long blocks = (na + nb + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(scale2_kernel, dim3((unsigned)blocks), dim3(256), 0, mcl_stream(stream), a, (long long)na, b,
                     (long long)nb, s, ya, yb);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_bn_gap_fwd(const void* x, int64_t ldx, int32_t B, int32_t HW, int32_t C, const float* gamma,
                              const float* beta, const float* mean, const float* rstd, float* out, float* xmean,
                              mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!x || !gamma || !beta || !mean || !rstd || !out || B <= 0 || HW <= 0 || C <= 0) return MCL_EINVAL;
  if ((C % 8) || (ldx % 8) || (reinterpret_cast<uintptr_t>(x) & 15u)) return MCL_EUNSUPPORTED;
  hipLaunchKernelGGL(bn_gap_fwd_kernel, dim3((C / 8 + 63) / 64, B), dim3(256), 0, mcl_stream(stream), (const bf16_t*)x,
                     (long long)ldx, HW, C, gamma, beta, mean, rstd, out, xmean);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_bn_gap_bwd(const float* g, const float* xmean, const void* x, int64_t ldx, int32_t B, int32_t HW,
                              int32_t C, const float* gamma, const float* mean, const float* rstd, float* coef /* 2C */,
                              float* dgamma, float* dbeta, int32_t accumulate_params, void* dx, int64_t lddx,
                              mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!g || !xmean || !x || !gamma || !mean || !rstd || !coef || !dgamma || !dbeta || !dx || B <= 0 || HW <= 0 || C <= 0)
    return MCL_EINVAL;
  if ((C % 8) || (ldx % 8) || (lddx % 8) || (reinterpret_cast<uintptr_t>(x) & 15u) ||
      (reinterpret_cast<uintptr_t>(dx) & 15u) || (reinterpret_cast<uintptr_t>(g) & 15u))
    return MCL_EUNSUPPORTED;
  hipStream_t st = mcl_stream(stream);
  hipLaunchKernelGGL(bn_gap_bwd_sums_kernel, dim3((C + 63) / 64), dim3(256), 0, st, g, xmean, B, C, HW, mean, rstd, dgamma,
                     dbeta, accumulate_params, coef);
  const long long n = (long long)B * HW * (C / 8);
  long long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(bn_gap_bwd_dx_kernel, dim3((unsigned)blocks), dim3(256), 0, st, g, (const bf16_t*)x, (long long)ldx, B,
                     HW, C, gamma, mean, rstd, (const float*)coef, (bf16_t*)dx, (long long)lddx);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_bn_running_update(int32_t n, float* const* running_mean, float* const* running_var,
                                     const float* const* mean, const float* const* var, int64_t* const* num_batches_tracked,
                                     const int32_t* C, const float* factor, const float* momentum, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (n < 0 || (n > 0 && (!running_mean || !running_var || !mean || !var || !C || !factor || !momentum))) return MCL_EINVAL;
  hipStream_t st = mcl_stream(stream);
  for (int i0 = 0; i0 < n; i0 += RUN_MAX) {
    RunBatch t;
    const int m = n - i0 < RUN_MAX ? n - i0 : RUN_MAX;
    for (int i = 0; i < m; ++i) {
      const int k = i0 + i;
      if (!running_mean[k] || !running_var[k] || !mean[k] || !var[k] || C[k] <= 0) return MCL_EINVAL;
      t.e[i] = RunEntry{running_mean[k], running_var[k], mean[k], var[k],
                        num_batches_tracked ? reinterpret_cast<long long*>(num_batches_tracked[k]) : nullptr, C[k],
                        factor[k], momentum[k], 0};
    }
    hipLaunchKernelGGL(bn_running_kernel, dim3(m), dim3(256), 0, st, t);
  }
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_bn_eval_rstd(int32_t n, const float* const* running_var, float* const* rstd_out, const int32_t* C,
                                const float* eps, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (n < 0 || (n > 0 && (!running_var || !rstd_out || !C || !eps))) return MCL_EINVAL;
  hipStream_t st = mcl_stream(stream);
  for (int i0 = 0; i0 < n; i0 += RUN_MAX) {
    RstdBatch t;
    const int m = n - i0 < RUN_MAX ? n - i0 : RUN_MAX;
    for (int i = 0; i < m; ++i) {
      const int k = i0 + i;
      if (!running_var[k] || !rstd_out[k] || C[k] <= 0) return MCL_EINVAL;
      t.e[i] = RstdEntry{running_var[k], rstd_out[k], C[k], eps[k]};
    }
    hipLaunchKernelGGL(bn_rstd_kernel, dim3(m), dim3(256), 0, st, t);
  }
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_image_to_bf16_nhwc(const float* x, int64_t sb, int64_t sc, int64_t sy, int64_t sx, int32_t B, int32_t C,
                                      int32_t H, int32_t W, void* y, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!x || !y || B <= 0 || C <= 0 || H <= 0 || W <= 0) return MCL_EINVAL;
  if (reinterpret_cast<uintptr_t>(y) & 15u) return MCL_EUNSUPPORTED;
  const long long n = (long long)B * C * H * W;
  hipStream_t st = mcl_stream(stream);
  const bool al16 = (reinterpret_cast<uintptr_t>(x) & 15u) == 0;
  if (sc == 1 && sx == C && sy == (int64_t)W * C && sb == (int64_t)H * W * C && (n % 8) == 0 && al16) {
    long long blocks = (n / 8 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(cast_f32_bf16_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, n / 8, (bf16_t*)y);
  } else if (C == 3 && sx == 1 && (W % 8) == 0 && (sy % 4) == 0 && (sc % 4) == 0 && (sb % 4) == 0 && al16) {
    const long long units = (long long)B * H * (W / 8);
    long long blocks = (units + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(image_planar3_to_bf16_nhwc_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, (long long)sb,
                       (long long)sc, (long long)sy, H, W / 8, units, (bf16_t*)y);
  } else {
    const long long blocks = (n / 8 + 256) / 256;
    hipLaunchKernelGGL(image_to_bf16_nhwc_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, (long long)sb, (long long)sc,
                       (long long)sy, (long long)sx, C, H, W, n, (bf16_t*)y);
  }
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

// A KERNEL, not hipMemsetAsync: a memset node in the middle of the captured step graph cost the step its two-lane
// execution (14.2 instead of 11.9 ms/step, same box; the serial kernel sum was unchanged) -- the step graph holds kernel
// nodes only.
extern "C" int mcl_fill_zero(void* p, int64_t bytes, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!p || bytes < 0) return MCL_EINVAL;
  if (bytes == 0) return MCL_OK;
  if ((reinterpret_cast<uintptr_t>(p) & 3u) || (bytes & 3)) return MCL_EUNSUPPORTED;
  long long blocks = (bytes / 16 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(fill_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, mcl_stream(stream), (unsigned*)p,
                     (long long)(bytes / 4));
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}
