// Generic convolution lowering for the shapes the specialised DenseNet kernels do not cover -- fp32 activations (the
// reference-numerics mode of the backbone: /root/reference/model.py:72-85 is pure fp32) and the ResNet encoders
// (/root/reference/model.py:88-148: 3x3 / 1x1 convolutions with stride 1 and 2 at arbitrary widths, the 7x7 stem):
//
//     cols[(n, oy, ox)][(ky, kx, c)] = x[n][oy*stride - pad + ky][ox*stride - pad + kx][c]      (0 outside the image)
//
// so that forward = cols . W^T, backward-data = col2im(dy . W), weight gradient = dy^T . cols run on this library's own GEMMs
// (mcl_gemm: fp32 MFMA; mcl_gemm_bf16).  The column order (ky, kx, c) is the storage order of a channels-last weight
// (C_out, kh, kw, C_in), which is therefore the GEMM operand as it lies in memory.  NHWC activations addressed through a
// row stride (a channel slice of a wider buffer is read in place).  Pure data movement, HBM-bound: 16-byte chunks along the
// channel dimension when C allows, a scalar form otherwise (the 3-channel stem).  col2im is a GATHER (every input pixel sums
// the <= kh*kw columns that reference it, in fixed order): deterministic, no atomics.
#include "common.h"

namespace {

template <typename T> struct El;
template <> struct El<float> {
  static constexpr int V = 4;
  __device__ static float to_f(float v) { return v; }
  __device__ static float from_f(float v) { return v; }
};
template <> struct El<unsigned short> {
  static constexpr int V = 8;
  __device__ static float to_f(unsigned short v) { return __uint_as_float(((unsigned)v) << 16); }
  __device__ static unsigned short from_f(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7F800000u) == 0x7F800000u) return (unsigned short)(u >> 16);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
  }
};

struct ConvGeo {
  int N, H, W, C, KH, KW, stride, pad, OH, OW;
};

// one 16-byte chunk of one column block per thread
template <typename T>
__global__ __launch_bounds__(256) void im2col_vec_kernel(const T* __restrict__ x, long long ldx, ConvGeo g,
                                                         T* __restrict__ cols) {
  constexpr int V = El<T>::V;
  const int cpr = g.C / V, taps = g.KH * g.KW;
  const long long total = (long long)g.N * g.OH * g.OW * taps * cpr;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % cpr);
    long long r = i / cpr;
    const int t = (int)(r % taps);
    r /= taps;                                                   // output pixel index (n, oy, ox)
    const int ox = (int)(r % g.OW);
    const long long r2 = r / g.OW;
    const int oy = (int)(r2 % g.OH), n = (int)(r2 / g.OH);
    const int iy = oy * g.stride - g.pad + t / g.KW, ix = ox * g.stride - g.pad + t % g.KW;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W)
      v = *reinterpret_cast<const uint4*>(x + (((long long)n * g.H + iy) * g.W + ix) * ldx + c * V);
    *reinterpret_cast<uint4*>(cols + (r * taps + t) * g.C + c * V) = v;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void im2col_scalar_kernel(const T* __restrict__ x, long long ldx, ConvGeo g,
                                                            T* __restrict__ cols) {
  const int taps = g.KH * g.KW;
  const long long total = (long long)g.N * g.OH * g.OW * taps * g.C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % g.C);
    long long r = i / g.C;
    const int t = (int)(r % taps);
    r /= taps;
    const int ox = (int)(r % g.OW);
    const long long r2 = r / g.OW;
    const int oy = (int)(r2 % g.OH), n = (int)(r2 / g.OH);
    const int iy = oy * g.stride - g.pad + t / g.KW, ix = ox * g.stride - g.pad + t % g.KW;
    T v = (T)0;
    if (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) v = x[(((long long)n * g.H + iy) * g.W + ix) * ldx + c];
    cols[i] = v;
  }
}

// dx[n][y][x][c] (+)= sum over taps (ky, kx) with (y + pad - ky) % stride == 0 and (x + pad - kx) % stride == 0 of
// dcols[(n, (y + pad - ky) / stride, (x + pad - kx) / stride)][(ky, kx, c)]; fp32 accumulation in fixed tap order
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void col2im_kernel(const T* __restrict__ dcols, ConvGeo g, T* __restrict__ dx,
                                                     long long lddx, int accumulate) {
  constexpr int V = VEC ? El<T>::V : 1;
  const int cpr = g.C / V, taps = g.KH * g.KW;
  const long long total = (long long)g.N * g.H * g.W * cpr;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % cpr);
    const long long p = i / cpr;
    const int xx = (int)(p % g.W);
    const long long p2 = p / g.W;
    const int yy = (int)(p2 % g.H), n = (int)(p2 / g.H);
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.0f;
    for (int ky = 0; ky < g.KH; ++ky) {
      const int ny = yy + g.pad - ky;
      if (ny < 0 || ny % g.stride) continue;
      const int oy = ny / g.stride;
      if (oy >= g.OH) continue;
      for (int kx = 0; kx < g.KW; ++kx) {
        const int nx = xx + g.pad - kx;
        if (nx < 0 || nx % g.stride) continue;
        const int ox = nx / g.stride;
        if (ox >= g.OW) continue;
        const T* src = dcols + ((((long long)n * g.OH + oy) * g.OW + ox) * taps + ky * g.KW + kx) * g.C + c * V;
        if (VEC) {
          const uint4 v = *reinterpret_cast<const uint4*>(src);
          const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
          for (int k = 0; k < V; ++k) acc[k] += El<T>::to_f(e[k]);
        } else {
          acc[0] += El<T>::to_f(src[0]);
        }
      }
    }
    T* dst = dx + p * lddx + c * V;
    if (VEC) {
      uint4 o;
      T* e = reinterpret_cast<T*>(&o);
      if (accumulate) {
        const uint4 old = *reinterpret_cast<const uint4*>(dst);
        const T* q = reinterpret_cast<const T*>(&old);
#pragma unroll
        for (int k = 0; k < V; ++k) e[k] = El<T>::from_f(El<T>::to_f(q[k]) + acc[k]);
      } else {
#pragma unroll
        for (int k = 0; k < V; ++k) e[k] = El<T>::from_f(acc[k]);
      }
      *reinterpret_cast<uint4*>(dst) = o;
    } else {
      dst[0] = El<T>::from_f(accumulate ? El<T>::to_f(dst[0]) + acc[0] : acc[0]);
    }
  }
}

inline bool geo(ConvGeo& g, int N, int H, int W, int C, int KH, int KW, int stride, int pad) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0) return false;
  g = ConvGeo{N, H, W, C, KH, KW, stride, pad, (H + 2 * pad - KH) / stride + 1, (W + 2 * pad - KW) / stride + 1};
  return g.OH > 0 && g.OW > 0;
}
inline unsigned grid_for(long long total) {
  long long b = (total + 255) / 256;
  return (unsigned)(b > 16384 ? 16384 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int mcl_im2col_nhwc(const void* x, int64_t ldx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t KH, int32_t KW,
                               int32_t stride, int32_t pad, int32_t dtype, void* cols, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  ConvGeo g;
  if (!x || !cols || !geo(g, N, H, W, C, KH, KW, stride, pad) || ldx < C || (dtype != 0 && dtype != 1)) return MCL_EINVAL;
  hipStream_t st = mcl_stream(stream);
  const int V = dtype ? 8 : 4;
  const bool vec = (C % V) == 0 && (ldx % V) == 0 && !(reinterpret_cast<uintptr_t>(x) & 15u) &&
                   !(reinterpret_cast<uintptr_t>(cols) & 15u);
  const long long total = (long long)N * g.OH * g.OW * KH * KW * (vec ? C / V : C);
  if (dtype == 0) {
    if (vec) hipLaunchKernelGGL(im2col_vec_kernel<float>, dim3(grid_for(total)), dim3(256), 0, st, (const float*)x, (long long)ldx, g, (float*)cols);
    else hipLaunchKernelGGL(im2col_scalar_kernel<float>, dim3(grid_for(total)), dim3(256), 0, st, (const float*)x, (long long)ldx, g, (float*)cols);
  } else {
    if (vec) hipLaunchKernelGGL(im2col_vec_kernel<unsigned short>, dim3(grid_for(total)), dim3(256), 0, st, (const unsigned short*)x, (long long)ldx, g, (unsigned short*)cols);
    else hipLaunchKernelGGL(im2col_scalar_kernel<unsigned short>, dim3(grid_for(total)), dim3(256), 0, st, (const unsigned short*)x, (long long)ldx, g, (unsigned short*)cols);
  }
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_col2im_nhwc(const void* dcols, int32_t N, int32_t H, int32_t W, int32_t C, int32_t KH, int32_t KW,
                               int32_t stride, int32_t pad, int32_t dtype, void* dx, int64_t lddx, int32_t accumulate,
                               mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  ConvGeo g;
  if (!dcols || !dx || !geo(g, N, H, W, C, KH, KW, stride, pad) || lddx < C || (dtype != 0 && dtype != 1)) return MCL_EINVAL;
  hipStream_t st = mcl_stream(stream);
  const int V = dtype ? 8 : 4;
  const bool vec = (C % V) == 0 && (lddx % V) == 0 && !(reinterpret_cast<uintptr_t>(dx) & 15u) &&
                   !(reinterpret_cast<uintptr_t>(dcols) & 15u);
  const long long total = (long long)N * H * W * (vec ? C / V : C);
  if (dtype == 0) {
    if (vec) hipLaunchKernelGGL((col2im_kernel<float, true>), dim3(grid_for(total)), dim3(256), 0, st, (const float*)dcols, g, (float*)dx, (long long)lddx, accumulate);
    else hipLaunchKernelGGL((col2im_kernel<float, false>), dim3(grid_for(total)), dim3(256), 0, st, (const float*)dcols, g, (float*)dx, (long long)lddx, accumulate);
  } else {
    if (vec) hipLaunchKernelGGL((col2im_kernel<unsigned short, true>), dim3(grid_for(total)), dim3(256), 0, st, (const unsigned short*)dcols, g, (unsigned short*)dx, (long long)lddx, accumulate);
    else hipLaunchKernelGGL((col2im_kernel<unsigned short, false>), dim3(grid_for(total)), dim3(256), 0, st, (const unsigned short*)dcols, g, (unsigned short*)dx, (long long)lddx, accumulate);
  }
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}
