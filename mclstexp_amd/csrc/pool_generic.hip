// Pooling / residual kernels for BOTH activation types (dtype 0 = fp32, 1 = bf16) on channels-last tensors: what the fp32
// ("reference numerics") mode of the DenseNet path and the ResNet encoders (/root/reference/model.py:88-148) need beside the
// convolutions: MaxPool2d(3, 2, 1), AvgPool2d(2, 2), adaptive_avg_pool2d((1, 1)) and the residual add + ReLU.  All
// HBM-streaming, one 16-byte channel chunk per thread and pixel, deterministic (the backward kernels gather).
// csrc/pool.hip holds the bf16-only forms with the fused BatchNorm prologue that the benched DenseNet step uses.
#include "common.h"

namespace {

template <typename T> struct El;
template <> struct El<float> {
  static constexpr int V = 4;
  __device__ static void load(const float* p, float (&f)[4]) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
  }
  __device__ static void store(float* p, const float (&f)[4]) { *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]); }
};
template <> struct El<unsigned short> {
  static constexpr int V = 8;
  __device__ static unsigned short rne(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7F800000u) == 0x7F800000u) return (unsigned short)(u >> 16);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
  }
  __device__ static void load(const unsigned short* p, float (&f)[8]) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = __uint_as_float(w[i] << 16);
      f[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
    }
  }
  __device__ static void store(unsigned short* p, const float (&f)[8]) {
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = (unsigned)rne(f[2 * i]) | ((unsigned)rne(f[2 * i + 1]) << 16);
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
  }
};

// MaxPool2d(3, 2, 1): idx = window position (ky*3 + kx) of the FIRST maximum in row-major window order (ATen's tie rule)
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                          unsigned char* __restrict__ idx, int N, int H, int W, int OH,
                                                          int OW, int CV) {
  constexpr int V = El<T>::V;
  const long long total = (long long)N * OH * OW * CV;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
    const int cv = (int)(q % CV);
    const long long op = q / CV;
    const int ox = (int)(op % OW), oy = (int)((op / OW) % OH), n = (int)(op / ((long long)OW * OH));
    float m[V];
    unsigned char am[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
      m[i] = -INFINITY;
      am[i] = 0;
    }
    // the nine taps: loaded unconditionally from clamped coordinates, all in flight together (a load inside the bounds branch is
    // followed by the compiler's vmcnt(0) at the join: nine dependent round trips per output); a tap outside the map is skipped
    float tap[9][V];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int iy = min(max(2 * oy - 1 + t / 3, 0), H - 1), ix = min(max(2 * ox - 1 + t % 3, 0), W - 1);
      El<T>::load(x + ((((long long)n * H + iy) * W + ix) * CV + cv) * V, tap[t]);
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int iy = 2 * oy - 1 + ky, ix = 2 * ox - 1 + kx;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
#pragma unroll
          for (int i = 0; i < V; ++i)
            if (tap[ky * 3 + kx][i] > m[i]) {
              m[i] = tap[ky * 3 + kx][i];
              am[i] = (unsigned char)(ky * 3 + kx);
            }
        }
      }
    El<T>::store(y + q * V, m);
#pragma unroll
    for (int i = 0; i < V; ++i) idx[q * V + i] = am[i];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const unsigned char* __restrict__ idx, const T* __restrict__ dy,
                                                          T* __restrict__ dx, int N, int H, int W, int OH, int OW, int CV) {
  constexpr int V = El<T>::V;
  const long long total = (long long)N * H * W * CV;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
    const int cv = (int)(q % CV);
    const long long p = q / CV;
    const int ix = (int)(p % W), iy = (int)((p / W) % H), n = (int)(p / ((long long)W * H));
    float acc[V];
#pragma unroll
    for (int i = 0; i < V; ++i) acc[i] = 0.0f;
    for (int oy = iy / 2; oy <= (iy + 1) / 2; ++oy) {
      if (oy >= OH) continue;
      for (int ox = ix / 2; ox <= (ix + 1) / 2; ++ox) {
        if (ox >= OW) continue;
        const long long oq = ((((long long)n * OH + oy) * OW + ox) * CV + cv) * V;
        const unsigned me = (unsigned)((iy - (2 * oy - 1)) * 3 + (ix - (2 * ox - 1)));
        float g[V];
        El<T>::load(dy + oq, g);
#pragma unroll
        for (int i = 0; i < V; ++i)
          if (idx[oq + i] == me) acc[i] += g[i];
      }
    }
    El<T>::store(dx + q * V, acc);
  }
}

// AvgPool2d(2, 2) with floor (odd maps drop the last row / column, like nn.AvgPool2d)
template <typename T>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W,
                                                          int CV) {
  constexpr int V = El<T>::V;
  const int OH = H / 2, OW = W / 2;
  const long long total = (long long)N * OH * OW * CV;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
    const int cv = (int)(q % CV);
    const long long op = q / CV;
    const int ox = (int)(op % OW), oy = (int)((op / OW) % OH), n = (int)(op / ((long long)OW * OH));
    const long long base = (((long long)n * H + 2 * oy) * W + 2 * ox) * CV + cv;
    float a[V], b[V], c[V], d[V], o[V];
    El<T>::load(x + base * V, a);
    El<T>::load(x + (base + CV) * V, b);
    El<T>::load(x + (base + (long long)W * CV) * V, c);
    El<T>::load(x + (base + (long long)W * CV + CV) * V, d);
#pragma unroll
    for (int i = 0; i < V; ++i) o[i] = 0.25f * ((a[i] + b[i]) + (c[i] + d[i]));
    El<T>::store(y + q * V, o);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int H, int W,
                                                          int CV) {
  constexpr int V = El<T>::V;
  const int OH = H / 2, OW = W / 2;
  const long long total = (long long)N * H * W * CV;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
    const int cv = (int)(q % CV);
    const long long p = q / CV;
    const int xx = (int)(p % W), yy = (int)((p / W) % H), n = (int)(p / ((long long)W * H));
    float g[V];
#pragma unroll
    for (int i = 0; i < V; ++i) g[i] = 0.0f;
    if (yy / 2 < OH && xx / 2 < OW) {
      El<T>::load(dy + ((((long long)n * OH + yy / 2) * OW + xx / 2) * CV + cv) * V, g);
#pragma unroll
      for (int i = 0; i < V; ++i) g[i] *= 0.25f;
    }
    El<T>::store(dx + q * V, g);
  }
}

// adaptive_avg_pool2d((1, 1)) + flatten: out[b][c] = mean over the HW positions (fp32 out); 64 chunk columns x 4 position
// groups per workgroup, fixed-order merge
template <typename T>
__global__ __launch_bounds__(256) void gap_fwd_kernel(const T* __restrict__ x, long long ldx, int HW, int C,
                                                      float* __restrict__ out) {
  constexpr int V = El<T>::V;
  __shared__ float red[4][64][V];
  const int b = blockIdx.y, col = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int c0 = (blockIdx.x * 64 + col) * V;
  float acc[V];
#pragma unroll
  for (int e = 0; e < V; ++e) acc[e] = 0.0f;
  if (c0 < C) {
    const T* p = x + (long long)b * HW * ldx + c0;
    for (int s = grp; s < HW; s += 4) {
      float v[V];
      El<T>::load(p + (long long)s * ldx, v);
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] += v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < V; ++e) red[grp][col][e] = acc[e];
  __syncthreads();
  if (grp != 0 || c0 >= C) return;
  const float inv = 1.0f / (float)HW;
#pragma unroll
  for (int e = 0; e < V; ++e)
    out[(long long)b * C + c0 + e] = ((red[0][col][e] + red[1][col][e]) + (red[2][col][e] + red[3][col][e])) * inv;
}

template <typename T>
__global__ __launch_bounds__(256) void gap_bwd_kernel(const float* __restrict__ g, int B, int HW, int C, T* __restrict__ dx) {
  constexpr int V = El<T>::V;
  const int cpr = C / V;
  const long long total = (long long)B * HW * cpr;
  const float inv = 1.0f / (float)HW;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % cpr);
    const long long row = i / cpr;
    const int b = (int)(row / HW);
    float o[V];
#pragma unroll
    for (int e = 0; e < V; ++e) o[e] = g[(long long)b * C + c * V + e] * inv;
    El<T>::store(dx + i * V, o);
  }
}

// y = relu(a + b)   |   dx = dy * [y > 0]  (the same gradient for both summands)
template <typename T>
__global__ __launch_bounds__(256) void add_relu_fwd_kernel(const T* __restrict__ a, const T* __restrict__ b, long long nv,
                                                           T* __restrict__ y) {
  constexpr int V = El<T>::V;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
    float u[V], w[V];
    El<T>::load(a + i * V, u);
    El<T>::load(b + i * V, w);
#pragma unroll
    for (int e = 0; e < V; ++e) u[e] = fmaxf(u[e] + w[e], 0.0f);
    El<T>::store(y + i * V, u);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void add_relu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, long long nv,
                                                           T* __restrict__ dx) {
  constexpr int V = El<T>::V;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
    float g[V], v[V];
    El<T>::load(dy + i * V, g);
    El<T>::load(y + i * V, v);
#pragma unroll
    for (int e = 0; e < V; ++e) g[e] = v[e] > 0.0f ? g[e] : 0.0f;
    El<T>::store(dx + i * V, g);
  }
}

// y = a + b (the gradient of a tensor with two consumers: the residual fork of a ResNet block)
template <typename T>
__global__ __launch_bounds__(256) void add_plain_kernel(const T* __restrict__ a, const T* __restrict__ b, long long nv,
                                                        T* __restrict__ y) {
  constexpr int V = El<T>::V;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
    float u[V], w[V];
    El<T>::load(a + i * V, u);
    El<T>::load(b + i * V, w);
#pragma unroll
    for (int e = 0; e < V; ++e) u[e] += w[e];
    El<T>::store(y + i * V, u);
  }
}

inline bool ok16(const void* p) { return p && (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline unsigned grid_for(long long total) {
  long long b = (total + 255) / 256;
  return (unsigned)(b > 16384 ? 16384 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int mcl_maxpool3s2_nhwc_fwd_any(const void* x, void* y, void* idx, int32_t N, int32_t H, int32_t W, int32_t C,
                                           int32_t dtype, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!ok16(x) || !ok16(y) || !idx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (dtype != 0 && dtype != 1)) return MCL_EINVAL;
  const int V = dtype ? 8 : 4;
  if (C % V) return MCL_EUNSUPPORTED;
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  hipStream_t st = mcl_stream(stream);
  const dim3 grid(grid_for((long long)N * OH * OW * (C / V)));
  if (dtype == 0)
    hipLaunchKernelGGL(maxpool_fwd_kernel<float>, grid, dim3(256), 0, st, (const float*)x, (float*)y, (unsigned char*)idx, N, H, W, OH, OW, C / V);
  else
    hipLaunchKernelGGL(maxpool_fwd_kernel<unsigned short>, grid, dim3(256), 0, st, (const unsigned short*)x, (unsigned short*)y, (unsigned char*)idx, N, H, W, OH, OW, C / V);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_maxpool3s2_nhwc_bwd_any(const void* idx, const void* dy, void* dx, int32_t N, int32_t H, int32_t W,
                                           int32_t C, int32_t dtype, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!idx || !ok16(dy) || !ok16(dx) || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (dtype != 0 && dtype != 1)) return MCL_EINVAL;
  const int V = dtype ? 8 : 4;
  if (C % V) return MCL_EUNSUPPORTED;
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  hipStream_t st = mcl_stream(stream);
  const dim3 grid(grid_for((long long)N * H * W * (C / V)));
  if (dtype == 0)
    hipLaunchKernelGGL(maxpool_bwd_kernel<float>, grid, dim3(256), 0, st, (const unsigned char*)idx, (const float*)dy, (float*)dx, N, H, W, OH, OW, C / V);
  else
    hipLaunchKernelGGL(maxpool_bwd_kernel<unsigned short>, grid, dim3(256), 0, st, (const unsigned char*)idx, (const unsigned short*)dy, (unsigned short*)dx, N, H, W, OH, OW, C / V);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_avgpool2_nhwc_any(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t backward,
                                     int32_t dtype, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!ok16(x) || !ok16(y) || N <= 0 || H < 2 || W < 2 || C <= 0 || (dtype != 0 && dtype != 1)) return MCL_EINVAL;
  const int V = dtype ? 8 : 4;
  if (C % V) return MCL_EUNSUPPORTED;
  hipStream_t st = mcl_stream(stream);
  if (!backward) {
    const dim3 grid(grid_for((long long)N * (H / 2) * (W / 2) * (C / V)));
    if (dtype == 0) hipLaunchKernelGGL(avgpool_fwd_kernel<float>, grid, dim3(256), 0, st, (const float*)x, (float*)y, N, H, W, C / V);
    else hipLaunchKernelGGL(avgpool_fwd_kernel<unsigned short>, grid, dim3(256), 0, st, (const unsigned short*)x, (unsigned short*)y, N, H, W, C / V);
  } else {   // x = dy (N, H/2, W/2, C), y = dx (N, H, W, C)
    const dim3 grid(grid_for((long long)N * H * W * (C / V)));
    if (dtype == 0) hipLaunchKernelGGL(avgpool_bwd_kernel<float>, grid, dim3(256), 0, st, (const float*)x, (float*)y, N, H, W, C / V);
    else hipLaunchKernelGGL(avgpool_bwd_kernel<unsigned short>, grid, dim3(256), 0, st, (const unsigned short*)x, (unsigned short*)y, N, H, W, C / V);
  }
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_gap_nhwc_fwd(const void* x, int64_t ldx, int32_t B, int32_t HW, int32_t C, int32_t dtype, float* out,
                                mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!ok16(x) || !out || B <= 0 || HW <= 0 || C <= 0 || ldx < C || (dtype != 0 && dtype != 1)) return MCL_EINVAL;
  const int V = dtype ? 8 : 4;
  if ((C % V) || (ldx % V)) return MCL_EUNSUPPORTED;
  hipStream_t st = mcl_stream(stream);
  const dim3 grid((C / V + 63) / 64, B);
  if (dtype == 0) hipLaunchKernelGGL(gap_fwd_kernel<float>, grid, dim3(256), 0, st, (const float*)x, (long long)ldx, HW, C, out);
  else hipLaunchKernelGGL(gap_fwd_kernel<unsigned short>, grid, dim3(256), 0, st, (const unsigned short*)x, (long long)ldx, HW, C, out);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_gap_nhwc_bwd(const float* g, int32_t B, int32_t HW, int32_t C, int32_t dtype, void* dx, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!g || !ok16(dx) || B <= 0 || HW <= 0 || C <= 0 || (dtype != 0 && dtype != 1)) return MCL_EINVAL;
  const int V = dtype ? 8 : 4;
  if (C % V) return MCL_EUNSUPPORTED;
  hipStream_t st = mcl_stream(stream);
  const dim3 grid(grid_for((long long)B * HW * (C / V)));
  if (dtype == 0) hipLaunchKernelGGL(gap_bwd_kernel<float>, grid, dim3(256), 0, st, g, B, HW, C, (float*)dx);
  else hipLaunchKernelGGL(gap_bwd_kernel<unsigned short>, grid, dim3(256), 0, st, g, B, HW, C, (unsigned short*)dx);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_add_relu(const void* a, const void* b, void* y, int64_t n, int32_t backward, int32_t dtype,
                            mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!ok16(a) || !ok16(b) || !ok16(y) || n <= 0 || (dtype != 0 && dtype != 1)) return MCL_EINVAL;
  const int V = dtype ? 8 : 4;
  if (n % V) return MCL_EUNSUPPORTED;
  hipStream_t st = mcl_stream(stream);
  const dim3 grid(grid_for(n / V));
  if (backward == 2) {  // y = a + b
    if (dtype == 0) hipLaunchKernelGGL(add_plain_kernel<float>, grid, dim3(256), 0, st, (const float*)a, (const float*)b, (long long)(n / V), (float*)y);
    else hipLaunchKernelGGL(add_plain_kernel<unsigned short>, grid, dim3(256), 0, st, (const unsigned short*)a, (const unsigned short*)b, (long long)(n / V), (unsigned short*)y);
  } else if (!backward) {     // y = relu(a + b)
    if (dtype == 0) hipLaunchKernelGGL(add_relu_fwd_kernel<float>, grid, dim3(256), 0, st, (const float*)a, (const float*)b, (long long)(n / V), (float*)y);
    else hipLaunchKernelGGL(add_relu_fwd_kernel<unsigned short>, grid, dim3(256), 0, st, (const unsigned short*)a, (const unsigned short*)b, (long long)(n / V), (unsigned short*)y);
  } else {             // a = dy, b = forward output, y = dx
    if (dtype == 0) hipLaunchKernelGGL(add_relu_bwd_kernel<float>, grid, dim3(256), 0, st, (const float*)a, (const float*)b, (long long)(n / V), (float*)y);
    else hipLaunchKernelGGL(add_relu_bwd_kernel<unsigned short>, grid, dim3(256), 0, st, (const unsigned short*)a, (const unsigned short*)b, (long long)(n / V), (unsigned short*)y);
  }
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}
