// Row kernels of the ViT image encoder on bf16 activations (/root/reference/model.py:104-116, timm VisionTransformer):
// LayerNorm forward / backward, attention softmax forward / backward (in place on the (heads, N, N) score tensor),
// column sums (bias gradients), patch extraction for the patch-embedding GEMM.  All reductions over the token
// dimension go through per-workgroup partials + the fixed-order merge of csrc/wrw_fused.hip: deterministic.
#include "common.h"

namespace {

typedef unsigned short bf16_t;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf16(f, 0.0f) & 0xFFFFu); }

// ---- LayerNorm over D (multiple of 8, <= 1024): one wave per row, 16-byte loads
constexpr int LN_MAXV = 2;                    // 16-byte chunks per lane (D <= 64 * 8 * 2 = 1024)

__global__ __launch_bounds__(256) void ln_fwd_kernel(const bf16_t* __restrict__ x, long long ldx,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     bf16_t* __restrict__ y, long long ldy, float* __restrict__ mean,
                                                     float* __restrict__ rstd, long long rows, int D, float eps) {
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= rows) return;
  const int nch = D >> 3;
  float v[LN_MAXV][8];
  float s = 0.0f;
#pragma unroll
  for (int u = 0; u < LN_MAXV; ++u) {
    const int c = lane + 64 * u;
    if (c < nch) {
      const uint4 w = *reinterpret_cast<const uint4*>(x + r * ldx + c * 8);
      const unsigned ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[u][2 * i] = __uint_as_float(ww[i] << 16);
        v[u][2 * i + 1] = __uint_as_float(ww[i] & 0xFFFF0000u);
        s += v[u][2 * i] + v[u][2 * i + 1];
      }
    }
  }
  s = wave_sum(s);
  const float mu = s / (float)D;
  float q = 0.0f;
#pragma unroll
  for (int u = 0; u < LN_MAXV; ++u)
    if (lane + 64 * u < nch)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float d = v[u][i] - mu;
        q = fmaf(d, d, q);
      }
  q = wave_sum(q);
  const float rs = rsqrtf(q / (float)D + eps);
  if (lane == 0) {
    mean[r] = mu;
    rstd[r] = rs;
  }
#pragma unroll
  for (int u = 0; u < LN_MAXV; ++u) {
    const int c = lane + 64 * u;
    if (c < nch) {
      float o[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = fmaf((v[u][i] - mu) * rs, gamma[c * 8 + i], beta[c * 8 + i]);
      *reinterpret_cast<uint4*>(y + r * ldy + c * 8) =
          make_uint4(pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3]), pack_bf16(o[4], o[5]), pack_bf16(o[6], o[7]));
    }
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma ;  optional + dx_add (residual branch gradient)
__global__ __launch_bounds__(256) void ln_bwd_dx_kernel(const bf16_t* __restrict__ dy, long long lddy,
                                                        const bf16_t* __restrict__ x, long long ldx,
                                                        const float* __restrict__ gamma, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, const bf16_t* __restrict__ dx_add,
                                                        long long ldadd, bf16_t* __restrict__ dx, long long lddx,
                                                        long long rows, int D) {
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= rows) return;
  const int nch = D >> 3;
  const float mu = mean[r], rs = rstd[r];
  float g[LN_MAXV][8], xh[LN_MAXV][8];
  float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
  for (int u = 0; u < LN_MAXV; ++u) {
    const int c = lane + 64 * u;
    if (c < nch) {
      const uint4 wd = *reinterpret_cast<const uint4*>(dy + r * lddy + c * 8);
      const uint4 wx = *reinterpret_cast<const uint4*>(x + r * ldx + c * 8);
      const unsigned a[4] = {wd.x, wd.y, wd.z, wd.w}, b[4] = {wx.x, wx.y, wx.z, wx.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        g[u][2 * i] = __uint_as_float(a[i] << 16) * gamma[c * 8 + 2 * i];
        g[u][2 * i + 1] = __uint_as_float(a[i] & 0xFFFF0000u) * gamma[c * 8 + 2 * i + 1];
        xh[u][2 * i] = (__uint_as_float(b[i] << 16) - mu) * rs;
        xh[u][2 * i + 1] = (__uint_as_float(b[i] & 0xFFFF0000u) - mu) * rs;
        s1 += g[u][2 * i] + g[u][2 * i + 1];
        s2 = fmaf(g[u][2 * i], xh[u][2 * i], fmaf(g[u][2 * i + 1], xh[u][2 * i + 1], s2));
      }
    }
  }
  s1 = wave_sum(s1) / (float)D;
  s2 = wave_sum(s2) / (float)D;
#pragma unroll
  for (int u = 0; u < LN_MAXV; ++u) {
    const int c = lane + 64 * u;
    if (c < nch) {
      float o[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = rs * (g[u][i] - s1 - xh[u][i] * s2);
      if (dx_add) {
        const uint4 wa = *reinterpret_cast<const uint4*>(dx_add + r * ldadd + c * 8);
        const unsigned a[4] = {wa.x, wa.y, wa.z, wa.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          o[2 * i] += __uint_as_float(a[i] << 16);
          o[2 * i + 1] += __uint_as_float(a[i] & 0xFFFF0000u);
        }
      }
      *reinterpret_cast<uint4*>(dx + r * lddx + c * 8) =
          make_uint4(pack_bf16(o[0], o[1]), pack_bf16(o[2], o[3]), pack_bf16(o[4], o[5]), pack_bf16(o[6], o[7]));
    }
  }
}

// Column reductions over the rows: MODE 0: slab[0][c] = sum_r dy[r][c] (bias gradient);
// MODE 1: slab[0][c] = sum_r dy * xhat (dgamma), slab[1][c] = sum_r dy (dbeta).
// A thread owns 8 consecutive columns (16-byte loads); a workgroup covers RP = max(1, 256 / (D/8)) rows per pass and
// walks the rows with stride G * RP; every (workgroup, row lane) pair writes its own slab (merged in fixed order).
template <int MODE>
__global__ __launch_bounds__(256) void colred_kernel(const bf16_t* __restrict__ dy, long long lddy,
                                                     const bf16_t* __restrict__ x, long long ldx,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     long long rows, int D, float* __restrict__ part, long long slab,
                                                     int RP) {
  constexpr int U = 2;                        // chunk passes per row: D <= 2 * 256 * 8 = 4096
  const int nch = D >> 3;
  const int rl = nch < 256 ? threadIdx.x / nch : 0;
  const int c0 = nch < 256 ? threadIdx.x % nch : threadIdx.x;
  const bool lane_on = rl < RP;
  float a[U][8], b[U][8];
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int i = 0; i < 8; ++i) a[u][i] = b[u][i] = 0.0f;
  if (lane_on) {
    long long r = (long long)blockIdx.x * RP + rl;
    const long long stride = (long long)gridDim.x * RP;
    if (MODE == 0) {
      // bias gradient: four rows per trip with all eight loads requested before the first add (one or two loads in flight per
      // thread left the kernel at 2.4 TB/s)
      for (; r + 3 * stride < rows; r += 4 * stride) {
        uint4 w[4][U];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int c = c0 + 256 * u;
            w[q][u] = make_uint4(0, 0, 0, 0);
            if (c < nch) w[q][u] = *reinterpret_cast<const uint4*>(dy + (r + q * stride) * lddy + c * 8);
          }
#pragma unroll
        for (int q = 0; q < 4; ++q)               // (row order kept: the sums are the same bits as the one-row loop's)
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const unsigned dd[4] = {w[q][u].x, w[q][u].y, w[q][u].z, w[q][u].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              a[u][2 * i] += __uint_as_float(dd[i] << 16);
              a[u][2 * i + 1] += __uint_as_float(dd[i] & 0xFFFF0000u);
            }
          }
      }
    }
    for (; r < rows; r += stride) {
      float mu = 0.0f, rs = 0.0f;
      if (MODE == 1) {
        mu = mean[r];
        rs = rstd[r];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = c0 + 256 * u;
        if (c < nch) {
          const uint4 wd = *reinterpret_cast<const uint4*>(dy + r * lddy + c * 8);
          const unsigned dd[4] = {wd.x, wd.y, wd.z, wd.w};
          if (MODE == 1) {
            const uint4 wx = *reinterpret_cast<const uint4*>(x + r * ldx + c * 8);
            const unsigned xx[4] = {wx.x, wx.y, wx.z, wx.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float d0 = __uint_as_float(dd[i] << 16), d1 = __uint_as_float(dd[i] & 0xFFFF0000u);
              a[u][2 * i] = fmaf(d0, (__uint_as_float(xx[i] << 16) - mu) * rs, a[u][2 * i]);
              a[u][2 * i + 1] = fmaf(d1, (__uint_as_float(xx[i] & 0xFFFF0000u) - mu) * rs, a[u][2 * i + 1]);
              b[u][2 * i] += d0;
              b[u][2 * i + 1] += d1;
            }
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              a[u][2 * i] += __uint_as_float(dd[i] << 16);
              a[u][2 * i + 1] += __uint_as_float(dd[i] & 0xFFFF0000u);
            }
          }
        }
      }
    }
    float* p = part + ((long long)blockIdx.x * RP + rl) * slab;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + 256 * u;
      if (c < nch) {
        *reinterpret_cast<float4*>(p + c * 8) = make_float4(a[u][0], a[u][1], a[u][2], a[u][3]);
        *reinterpret_cast<float4*>(p + c * 8 + 4) = make_float4(a[u][4], a[u][5], a[u][6], a[u][7]);
        if (MODE == 1) {
          *reinterpret_cast<float4*>(p + D + c * 8) = make_float4(b[u][0], b[u][1], b[u][2], b[u][3]);
          *reinterpret_cast<float4*>(p + D + c * 8 + 4) = make_float4(b[u][4], b[u][5], b[u][6], b[u][7]);
        }
      }
    }
  }
}

// ---- attention softmax over rows of length n (<= 256), in place on bf16 scores with row stride ld: one wave per row.
// forward: p = softmax(s) (the 1/sqrt(d) scale is applied by the score GEMM's alpha)
__global__ __launch_bounds__(256) void softmax_fwd_kernel(bf16_t* __restrict__ s, long long ld, long long rows, int n) {
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= rows) return;
  bf16_t* p = s + r * ld;
  float v[4];
  float mx = -3.0e38f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < n ? bf2f(p[c]) : -3.0e38f;
    mx = fmaxf(mx, v[i]);
  }
  mx = wave_max(mx);
  float sum = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[i] = lane + 64 * i < n ? __expf(v[i] - mx) : 0.0f;
    sum += v[i];
  }
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    if (c < n) p[c] = f2bf(v[i] * inv);
    else if (c < ld) p[c] = 0;                 // padding columns: exact zeros (they are GEMM operands)
  }
}
// backward, in place on dP: dS = P * (dP - sum_c P dP) * scale
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const bf16_t* __restrict__ P, bf16_t* __restrict__ dP,
                                                          long long ld, long long rows, int n, float scale) {
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= rows) return;
  const bf16_t* p = P + r * ld;
  bf16_t* d = dP + r * ld;
  float pv[4], dv[4];
  float dot = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    pv[i] = c < n ? bf2f(p[c]) : 0.0f;
    dv[i] = c < n ? bf2f(d[c]) : 0.0f;
    dot = fmaf(pv[i], dv[i], dot);
  }
  dot = wave_sum(dot);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    if (c < n) d[c] = f2bf(pv[i] * (dv[i] - dot) * scale);
    else if (c < ld) d[c] = 0;
  }
}

// Vector variants (ld % 8 == 0, 16-byte aligned base): HALF a wave per row, one 16-byte chunk per lane (8 columns at
// 8*l, up to 256 columns), reductions over the 32 lanes of the half.  The scalar kernels above move 128 B per wave
// load and ran at 2 TB/s on the ViT-B/16 score tensor (605 k rows x 208); these move 1 KiB per wave load.
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float half_max(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ void unpack8(const uint4 u, float (&v)[8]) {
  const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = __uint_as_float(w[i] << 16);
    v[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
  }
}
__device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
  unsigned w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = pack_bf16(v[2 * i], v[2 * i + 1]);
  return make_uint4(w[0], w[1], w[2], w[3]);
}

__global__ __launch_bounds__(256) void softmax_fwd_vec_kernel(bf16_t* __restrict__ s, long long ld, long long rows, int n) {
  const long long r = (long long)blockIdx.x * 8 + (threadIdx.x >> 5);
  const int c0 = (threadIdx.x & 31) * 8;
  if (r >= rows) return;
  bf16_t* p = s + r * ld + c0;
  const bool live = c0 < ld;
  float v[8];
  if (live) unpack8(*reinterpret_cast<const uint4*>(p), v);
  float mx = -3.0e38f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (!live || c0 + i >= n) v[i] = -3.0e38f;
    mx = fmaxf(mx, v[i]);
  }
  mx = half_max(mx);
  float sum = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    v[i] = (live && c0 + i < n) ? __expf(v[i] - mx) : 0.0f;
    sum += v[i];
  }
  sum = half_sum(sum);
  const float inv = 1.0f / sum;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] *= inv;       // padding columns stay exact zeros
  if (live) *reinterpret_cast<uint4*>(p) = pack8(v);
}

__global__ __launch_bounds__(256) void softmax_bwd_vec_kernel(const bf16_t* __restrict__ P, bf16_t* __restrict__ dP,
                                                              long long ld, long long rows, int n, float scale) {
  const long long r = (long long)blockIdx.x * 8 + (threadIdx.x >> 5);
  const int c0 = (threadIdx.x & 31) * 8;
  if (r >= rows) return;
  const bool live = c0 < ld;
  float pv[8], dv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) pv[i] = dv[i] = 0.0f;
  if (live) {
    unpack8(*reinterpret_cast<const uint4*>(P + r * ld + c0), pv);
    unpack8(*reinterpret_cast<const uint4*>(dP + r * ld + c0), dv);
  }
  float dot = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (c0 + i >= n) pv[i] = dv[i] = 0.0f;
    dot = fmaf(pv[i], dv[i], dot);
  }
  dot = half_sum(dot);
#pragma unroll
  for (int i = 0; i < 8; ++i) dv[i] = pv[i] * (dv[i] - dot) * scale;
  if (live) *reinterpret_cast<uint4*>(dP + r * ld + c0) = pack8(dv);
}

// ---- patch extraction for the patch-embedding GEMM: out[(b*np + py*npw + px)][c*p*p + iy*p + ix] = img[b][c][py*p+iy][px*p+ix]
// (the flattening of timm's Conv2d(3, D, p, p) weight (D, 3, p, p)); img fp32 with arbitrary strides (NCHW or NHWC).
// lead = 1: every image's patches are preceded by ONE ZERO ROW (the class-token position of the (B, np + 1, K) token matrix: the
// patch-embedding weight gradient is then one reduction over all B (np + 1) rows).  OUT = bf16_t or float.
template <typename OUT>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, long long sb, long long sc,
                                                       long long sy, long long sx, int B, int H, int W, int p, int lead,
                                                       OUT* __restrict__ out) {
  const int npw = W / p, nph = H / p, K = 3 * p * p;
  const long long total = (long long)B * nph * npw * K;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
    const int k = (int)(q % K);
    const long long t = q / K;
    const int px = (int)(t % npw), py = (int)((t / npw) % nph);
    const long long b = t / ((long long)npw * nph);
    const int ix = k % p, iy = (k / p) % p, c = k / (p * p);
    const float v = img[b * sb + c * sc + (long long)(py * p + iy) * sy + (long long)(px * p + ix) * sx];
    OUT* o = out + (t + lead * (b + 1)) * K + k;
    if constexpr (sizeof(OUT) == 2) *o = f2bf(v);
    else *o = v;
  }
  if (lead) {
    const long long zt = (long long)B * K;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < zt; q += (long long)gridDim.x * 256) {
      const long long b = q / K;
      OUT* o = out + b * (long long)(nph * npw + 1) * K + (q - b * K);
      if constexpr (sizeof(OUT) == 2) *o = (OUT)0;
      else *o = 0.0f;
    }
  }
}

// the same for unit pixel stride and p % 8 == 0 (NCHW batches): a thread moves 8 consecutive pixels of one patch row -- two
// 16-byte loads, one 16-byte store, one index decode per 8 elements (the scalar form spends its time in 64-bit divisions)
__global__ __launch_bounds__(256) void patchify8_kernel(const float* __restrict__ img, long long sb, long long sc, long long sy,
                                                        int B, int H, int W, int p, int lead, bf16_t* __restrict__ out) {
  const int npw = W / p, nph = H / p, K8 = 3 * p * p / 8, p8 = p / 8;
  const long long total8 = (long long)B * nph * npw * K8;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total8; q += (long long)gridDim.x * 256) {
    const int k8 = (int)(q % K8);
    const int t = (int)(q / K8);
    const int px = t % npw, py = (t / npw) % nph, b = t / (npw * nph);
    const int ix8 = k8 % p8, iy = (k8 / p8) % p, c = k8 / (p8 * p);
    const float* src = img + (long long)b * sb + (long long)c * sc + (long long)(py * p + iy) * sy + (px * p + 8 * ix8);
    const float4 a = *reinterpret_cast<const float4*>(src), d = *reinterpret_cast<const float4*>(src + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, d.x, d.y, d.z, d.w};
    *reinterpret_cast<uint4*>(out + ((long long)t + lead * (b + 1)) * K8 * 8 + (long long)k8 * 8) = pack8(v);
  }
  if (lead) {
    const long long zt = (long long)B * K8;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < zt; q += (long long)gridDim.x * 256) {
      const long long b = q / K8;
      *reinterpret_cast<uint4*>(out + b * (long long)(nph * npw + 1) * K8 * 8 + (q - b * K8) * 8) = make_uint4(0u, 0u, 0u, 0u);
    }
  }
}

}  // namespace

extern "C" int mcl_ln_bf16_fwd(const void* x, int64_t ldx, const float* gamma, const float* beta, void* y, int64_t ldy,
                               float* mean, float* rstd, int64_t rows, int32_t D, float eps, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!x || !gamma || !beta || !y || !mean || !rstd || rows <= 0 || D <= 0) return MCL_EINVAL;
  if ((D % 8) || D > 1024 || (ldx % 8) || (ldy % 8) || (reinterpret_cast<uintptr_t>(x) & 15u) ||
      (reinterpret_cast<uintptr_t>(y) & 15u))
    return MCL_EUNSUPPORTED;
  hipLaunchKernelGGL(ln_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, mcl_stream(stream), (const bf16_t*)x,
                     (long long)ldx, gamma, beta, (bf16_t*)y, (long long)ldy, mean, rstd, (long long)rows, D, eps);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

namespace {
struct ColPlan {
  int G, RP;
};
inline ColPlan colplan(long long rows, int D) {
  ColPlan p;
  const int nch = D >> 3;
  p.RP = nch < 256 ? 256 / nch : 1;
  long long g = (rows + p.RP - 1) / p.RP;
  p.G = (int)(g < 1024 ? g : 1024);
  return p;
}
}  // namespace

extern "C" int64_t mcl_colred_workspace_floats(int64_t rows, int32_t D) {
  if (rows <= 0 || D <= 0) return -1;
  const ColPlan p = colplan(rows, D);
  return (int64_t)p.G * p.RP * (2 * (int64_t)D + 64);
}

// dx (+ dx_add) and, merged in fixed order, dgamma / dbeta (accumulate != 0: +=).  workspace: mcl_colred_workspace_floats.
extern "C" int mcl_ln_bf16_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, const float* gamma,
                               const float* mean, const float* rstd, const void* dx_add, int64_t ldadd, void* dx,
                               int64_t lddx, float* workspace, float* dgamma, float* dbeta, int32_t accumulate,
                               int64_t rows, int32_t D, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!dy || !x || !gamma || !mean || !rstd || !dx || !workspace || !dgamma || !dbeta || rows <= 0 || D <= 0)
    return MCL_EINVAL;
  if ((D % 8) || D > 1024 || (lddy % 8) || (ldx % 8) || (lddx % 8) || (dx_add && (ldadd % 8)) ||
      (reinterpret_cast<uintptr_t>(dy) & 15u) || (reinterpret_cast<uintptr_t>(x) & 15u) ||
      (reinterpret_cast<uintptr_t>(dx) & 15u) || (reinterpret_cast<uintptr_t>(workspace) & 15u))
    return MCL_EUNSUPPORTED;
  hipStream_t st = mcl_stream(stream);
  hipLaunchKernelGGL(ln_bwd_dx_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, (const bf16_t*)dy,
                     (long long)lddy, (const bf16_t*)x, (long long)ldx, gamma, mean, rstd, (const bf16_t*)dx_add,
                     (long long)ldadd, (bf16_t*)dx, (long long)lddx, (long long)rows, D);
  const ColPlan cp = colplan(rows, D);
  const long long slab = 2LL * D + 64;
  hipLaunchKernelGGL((colred_kernel<1>), dim3(cp.G), dim3(256), 0, st, (const bf16_t*)dy, (long long)lddy,
                     (const bf16_t*)x, (long long)ldx, mean, rstd, (long long)rows, D, workspace, slab, cp.RP);
  mcl_launch_wrw_merge_strided(workspace, cp.G * cp.RP, D, slab, dgamma, accumulate, st);
  mcl_launch_wrw_merge_strided(workspace + D, cp.G * cp.RP, D, slab, dbeta, accumulate, st);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

// out[c] (+)= sum_r x[r][c]   (bias gradients), bf16 rows; deterministic.  workspace: mcl_colred_workspace_floats.
extern "C" int mcl_colsum_bf16(const void* x, int64_t ldx, int64_t rows, int32_t D, float* workspace, float* out,
                               int32_t accumulate, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!x || !workspace || !out || rows <= 0 || D <= 0) return MCL_EINVAL;
  if ((D % 8) || D > 4096 || (ldx % 8) || (reinterpret_cast<uintptr_t>(x) & 15u) ||
      (reinterpret_cast<uintptr_t>(workspace) & 15u) || (reinterpret_cast<uintptr_t>(out) & 15u))
    return MCL_EUNSUPPORTED;
  hipStream_t st = mcl_stream(stream);
  const ColPlan cp = colplan(rows, D);
  const long long slab = 2LL * D + 64;
  hipLaunchKernelGGL((colred_kernel<0>), dim3(cp.G), dim3(256), 0, st, (const bf16_t*)x, (long long)ldx,
                     (const bf16_t*)nullptr, 0LL, (const float*)nullptr, (const float*)nullptr, (long long)rows, D,
                     workspace, slab, cp.RP);
  mcl_launch_wrw_merge_strided(workspace, cp.G * cp.RP, D, slab, out, accumulate, st);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_softmax_bf16_fwd(void* s, int64_t ld, int64_t rows, int32_t n, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!s || rows <= 0 || n <= 0) return MCL_EINVAL;
  if (n > 256 || ld < n) return MCL_EUNSUPPORTED;
  if (ld % 8 == 0 && (reinterpret_cast<uintptr_t>(s) & 15u) == 0)
    hipLaunchKernelGGL(softmax_fwd_vec_kernel, dim3((unsigned)((rows + 7) / 8)), dim3(256), 0, mcl_stream(stream),
                       (bf16_t*)s, (long long)ld, (long long)rows, n);
  else
    hipLaunchKernelGGL(softmax_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, mcl_stream(stream), (bf16_t*)s,
                       (long long)ld, (long long)rows, n);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_softmax_bf16_bwd(const void* P, void* dP, int64_t ld, int64_t rows, int32_t n, float scale,
                                    mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!P || !dP || rows <= 0 || n <= 0) return MCL_EINVAL;
  if (n > 256 || ld < n) return MCL_EUNSUPPORTED;
  if (ld % 8 == 0 && ((reinterpret_cast<uintptr_t>(P) | reinterpret_cast<uintptr_t>(dP)) & 15u) == 0)
    hipLaunchKernelGGL(softmax_bwd_vec_kernel, dim3((unsigned)((rows + 7) / 8)), dim3(256), 0, mcl_stream(stream),
                       (const bf16_t*)P, (bf16_t*)dP, (long long)ld, (long long)rows, n, scale);
  else
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, mcl_stream(stream),
                       (const bf16_t*)P, (bf16_t*)dP, (long long)ld, (long long)rows, n, scale);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

namespace {
int launch_patchify(const float* img, int64_t sb, int64_t sc, int64_t sy, int64_t sx, int32_t B, int32_t H, int32_t W, int32_t p,
                    void* out, int lead, int out_f32, mcl_stream_t stream) {
  if (!img || !out || B <= 0 || H <= 0 || W <= 0 || p <= 0) return MCL_EINVAL;
  if ((H % p) || (W % p)) return MCL_EUNSUPPORTED;
  const long long total = (long long)B * 3 * H * W;
  if (!out_f32 && sx == 1 && (p % 8) == 0 && (sy % 4) == 0 && (sc % 4) == 0 && (sb % 4) == 0 &&
      !(reinterpret_cast<uintptr_t>(img) & 15u) && !(reinterpret_cast<uintptr_t>(out) & 15u) &&
      (long long)B * (H / p) * (W / p) < (1ll << 31)) {
    long long blocks8 = (total / 8 + 255) / 256;
    if (blocks8 > 65535 * 4) blocks8 = 65535 * 4;
    hipLaunchKernelGGL(patchify8_kernel, dim3((unsigned)blocks8), dim3(256), 0, mcl_stream(stream), img, (long long)sb,
                       (long long)sc, (long long)sy, B, H, W, p, lead, (bf16_t*)out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MCL_OK : (int)e;
  }
  long long blocks = (total + 255) / 256;
  if (blocks > 65535 * 4) blocks = 65535 * 4;
  if (out_f32)
    hipLaunchKernelGGL(patchify_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, mcl_stream(stream), img, (long long)sb,
                       (long long)sc, (long long)sy, (long long)sx, B, H, W, p, lead, (float*)out);
  else
    hipLaunchKernelGGL(patchify_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, mcl_stream(stream), img, (long long)sb,
                       (long long)sc, (long long)sy, (long long)sx, B, H, W, p, lead, (bf16_t*)out);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MCL_OK : (int)e;
}
}  // namespace

extern "C" int mcl_vit_patchify(const float* img, int64_t sb, int64_t sc, int64_t sy, int64_t sx, int32_t B, int32_t H,
                                int32_t W, int32_t p, void* out_bf16, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  return launch_patchify(img, sb, sc, sy, sx, B, H, W, p, out_bf16, 0, 0, stream);
}

extern "C" int mcl_vit_patchify_tokens(const float* img, int64_t sb, int64_t sc, int64_t sy, int64_t sx, int32_t B, int32_t H,
                                       int32_t W, int32_t p, void* out, int32_t lead_zero_row, int32_t out_f32,
                                       mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  return launch_patchify(img, sb, sc, sy, sx, B, H, W, p, out, lead_zero_row ? 1 : 0, out_f32 ? 1 : 0, stream);
}
