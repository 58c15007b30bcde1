// Pooling layers of the DenseNet stem / transitions on channels-last bf16 (torchvision: pool0 = MaxPool2d(3, 2, 1),
// transition.pool = AvgPool2d(2, 2); /root/reference/model.py:75-76 via torchvision).  ATen's NHWC pooling
// backward kernels run at ~1 TB/s on these shapes; these are plain HBM-streaming kernels, one thread per 16-byte
// channel chunk (8 channels) of one pixel.
#include "common.h"

namespace {

typedef unsigned short bf16_t;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void unpack8(uint4 v, float (&f)[8]) {
  const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(w[i] << 16);
    f[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
  }
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  unsigned w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x2 v = {f[2 * i], f[2 * i + 1]};
    w[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

// y[n, oy, ox, c] = mean of the 2x2 window (H, W even); fp32 accumulation, one rounding
__global__ __launch_bounds__(256) void avgpool2_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int N,
                                                           int H, int W, int C8) {
  const int OH = H / 2, OW = W / 2;
  const long long total = (long long)N * OH * OW * C8;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
    const int c8 = (int)(q % C8);
    const long long op = q / C8;
    const int ox = (int)(op % OW), oy = (int)((op / OW) % OH), n = (int)(op / ((long long)OW * OH));
    const long long base = (((long long)n * H + 2 * oy) * W + 2 * ox) * C8 + c8;
    float a[8], b[8], c[8], d[8], o[8];
    unpack8(*reinterpret_cast<const uint4*>(x + base * 8), a);
    unpack8(*reinterpret_cast<const uint4*>(x + (base + C8) * 8), b);
    unpack8(*reinterpret_cast<const uint4*>(x + (base + (long long)W * C8) * 8), c);
    unpack8(*reinterpret_cast<const uint4*>(x + (base + (long long)W * C8 + C8) * 8), d);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = 0.25f * ((a[i] + b[i]) + (c[i] + d[i]));
    *reinterpret_cast<uint4*>(y + q * 8) = pack8(o);
  }
}

// dx[n, y, x, c] = dy[n, y/2, x/2, c] / 4
__global__ __launch_bounds__(256) void avgpool2_bwd_kernel(const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx, int N,
                                                           int H, int W, int C8) {
  const int OH = H / 2, OW = W / 2;
  const long long total = (long long)N * H * W * C8;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
    const int c8 = (int)(q % C8);
    const long long p = q / C8;
    const int xx = (int)(p % W), yy = (int)((p / W) % H), n = (int)(p / ((long long)W * H));
    float g[8];
    unpack8(*reinterpret_cast<const uint4*>(dy + ((((long long)n * OH + yy / 2) * OW + xx / 2) * C8 + c8) * 8), g);
#pragma unroll
    for (int i = 0; i < 8; ++i) g[i] *= 0.25f;
    *reinterpret_cast<uint4*>(dx + q * 8) = pack8(g);
  }
}

// MaxPool2d(3, stride 2, pad 1): y = max over the window (padding = -inf); idx = window position (ky*3 + kx) of the
// FIRST maximum in row-major window order (ATen's tie rule: strict > while scanning), one byte per element
// BN: the pooled tensor is relu(x*scale + shift) (the DenseNet stem norm0 -> relu0 -> pool0 in one pass: the
// normalised full-resolution map is never written)
template <bool BN>
__global__ __launch_bounds__(256) void maxpool3s2_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                             unsigned char* __restrict__ idx, int N, int H, int W,
                                                             int OH, int OW, int C8, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ rstd) {
  const long long total = (long long)N * OH * OW * C8;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
    const int c8 = (int)(q % C8);
    const long long op = q / C8;
    const int ox = (int)(op % OW), oy = (int)((op / OW) % OH), n = (int)(op / ((long long)OW * OH));
    float m[8];
    unsigned am[8];
    float sc[8], sh[8];
    if (BN) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = c8 * 8 + i;
        sc[i] = gamma[c] * rstd[c];
        sh[i] = fmaf(-mean[c], sc[i], beta[c]);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      m[i] = -INFINITY;
      am[i] = 0;
    }
    // the nine taps: loaded unconditionally from clamped coordinates (all in flight together; a load inside the bounds branch
    // is followed by a vmcnt(0) at the join -- nine dependent round trips per output chunk), a tap outside the map is skipped
    uint4 tap[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int iy = min(max(2 * oy - 1 + t / 3, 0), H - 1), ix = min(max(2 * ox - 1 + t % 3, 0), W - 1);
      tap[t] = *reinterpret_cast<const uint4*>(x + ((((long long)n * H + iy) * W + ix) * C8 + c8) * 8);
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int iy = 2 * oy - 1 + ky, ix = 2 * ox - 1 + kx;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
          float v[8];
          unpack8(tap[ky * 3 + kx], v);
          if (BN) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = fmaxf(fmaf(v[i], sc[i], sh[i]), 0.0f);
          }
#pragma unroll
          for (int i = 0; i < 8; ++i)
            if (v[i] > m[i]) {      // strict: the first maximum wins; the first valid element beats -inf
              m[i] = v[i];
              am[i] = ky * 3 + kx;
            }
        }
      }
    *reinterpret_cast<uint4*>(y + q * 8) = pack8(m);
    uint2 pk;
    pk.x = am[0] | (am[1] << 8) | (am[2] << 16) | (am[3] << 24);
    pk.y = am[4] | (am[5] << 8) | (am[6] << 16) | (am[7] << 24);
    *reinterpret_cast<uint2*>(idx + q * 8) = pk;
  }
}

// dx[n, iy, ix, c] = sum of dy over the (<= 4) windows that contain the pixel and whose recorded arg-max is this
// pixel.  Gather form: deterministic, no atomics.
__global__ __launch_bounds__(256) void maxpool3s2_bwd_kernel(const unsigned char* __restrict__ idx,
                                                             const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx,
                                                             int N, int H, int W, int OH, int OW, int C8,
                                                             long long lddy /* elements per pooled pixel row of dy */) {
  const long long total = (long long)N * H * W * C8;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
    const int c8 = (int)(q % C8);
    const long long p = q / C8;
    const int ix = (int)(p % W), iy = (int)((p / W) % H), n = (int)(p / ((long long)W * H));
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.0f;
    // windows oy with 2*oy-1 <= iy <= 2*oy+1  <=>  iy/2 <= oy <= (iy+1)/2   (iy >= 0): one window per axis for an even
    // coordinate, two for an odd one.  All four candidate windows are loaded unconditionally from clamped addresses (a load
    // inside a branch is followed by a vmcnt(0) at the join: four dependent round trips per output chunk); a window that does
    // not exist gets the position code 255, which no recorded arg-max equals.  Same order of additions as the loop form.
    uint2 pk[4];
    uint4 g[4];
    unsigned me[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int ay = w >> 1, ax = w & 1;
      const int oy = iy / 2 + ay, ox = ix / 2 + ax;
      const bool valid = (ay == 0 || (iy & 1)) && (ax == 0 || (ix & 1)) && oy < OH && ox < OW;
      const int oyc = min(oy, OH - 1), oxc = min(ox, OW - 1);
      const long long op = ((long long)n * OH + oyc) * OW + oxc;
      pk[w] = *reinterpret_cast<const uint2*>(idx + (op * C8 + c8) * 8);
      g[w] = *reinterpret_cast<const uint4*>(dy + op * lddy + c8 * 8);
      me[w] = valid ? (unsigned)((iy - (2 * oy - 1)) * 3 + (ix - (2 * ox - 1))) : 255u;   // my position in that window
    }
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      float gv[8];
      unpack8(g[w], gv);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (((pk[w].x >> (8 * i)) & 0xFFu) == me[w]) acc[i] += gv[i];
        if (((pk[w].y >> (8 * i)) & 0xFFu) == me[w]) acc[4 + i] += gv[4 + i];
      }
    }
    *reinterpret_cast<uint4*>(dx + q * 8) = pack8(acc);
  }
}

// The same for even H and W, one thread per 2 x 2 block of input pixels (x 8 channels): the four pixels' windows are the SAME
// four pooled pixels (a + {0,1}, b + {0,1}), so every recorded arg-max byte and every dy chunk is loaded once per block instead
// of once per pixel that might own it (2.25 window loads per pixel -> 1: the one-pixel form moves 9 TB/s through the texture
// path for 0.3 GB of HBM traffic).  Per output the additions run in the window order of the form above: bit-identical.
__global__ __launch_bounds__(256) void maxpool3s2_bwd_quad_kernel(const unsigned char* __restrict__ idx,
                                                                  const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx,
                                                                  int N, int H, int W, int OH, int OW, int C8, long long lddy) {
  const int H2 = H >> 1, W2 = W >> 1;
  const long long total = (long long)N * H2 * W2 * C8;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
    const int c8 = (int)(q % C8);
    const long long p = q / C8;
    const int b = (int)(p % W2), a = (int)((p / W2) % H2), n = (int)(p / ((long long)W2 * H2));
    uint2 pk[4];
    uint4 g[4];
    bool valid[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {                          // unconditional loads from clamped addresses
      const int oy = a + (w >> 1), ox = b + (w & 1);
      valid[w] = oy < OH && ox < OW;
      const long long op = ((long long)n * OH + min(oy, OH - 1)) * OW + min(ox, OW - 1);
      pk[w] = *reinterpret_cast<const uint2*>(idx + (op * C8 + c8) * 8);
      g[w] = *reinterpret_cast<const uint4*>(dy + op * lddy + c8 * 8);
    }
    float gv[4][8];
#pragma unroll
    for (int w = 0; w < 4; ++w) unpack8(g[w], gv[w]);
#pragma unroll
    for (int o = 0; o < 4; ++o) {                          // output pixel (2a + oyy, 2b + oxx)
      const int oyy = o >> 1, oxx = o & 1;
      float acc[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = 0.0f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const int ay = w >> 1, ax = w & 1;
        if (ay > oyy || ax > oxx) continue;                // (an even coordinate lies in one window along its axis)
        const unsigned me = valid[w] ? (unsigned)((oyy - 2 * ay + 1) * 3 + (oxx - 2 * ax + 1)) : 255u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (((pk[w].x >> (8 * i)) & 0xFFu) == me) acc[i] += gv[w][i];
          if (((pk[w].y >> (8 * i)) & 0xFFu) == me) acc[4 + i] += gv[w][4 + i];
        }
      }
      const long long ip = ((long long)n * H + 2 * a + oyy) * W + 2 * b + oxx;
      *reinterpret_cast<uint4*>(dx + (ip * C8 + c8) * 8) = pack8(acc);
    }
  }
}

inline bool ok16(const void* p) { return p && (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline unsigned blocks_for(long long total) {
  long long b = (total + 255) / 256;
  return (unsigned)(b > 16384 ? 16384 : b);
}

}  // namespace

extern "C" int mcl_avgpool2_nhwc_bf16(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t backward,
                                      mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!ok16(x) || !ok16(y) || N <= 0 || H <= 0 || W <= 0 || C <= 0) return MCL_EINVAL;
  if ((H & 1) || (W & 1) || (C % 8)) return MCL_EUNSUPPORTED;
  const int C8 = C / 8;
  if (!backward) {
    const long long total = (long long)N * (H / 2) * (W / 2) * C8;
    hipLaunchKernelGGL(avgpool2_fwd_kernel, dim3(blocks_for(total)), dim3(256), 0, mcl_stream(stream), (const bf16_t*)x,
                       (bf16_t*)y, N, H, W, C8);
  } else {   // x = dy (N, H/2, W/2, C), y = dx (N, H, W, C)
    const long long total = (long long)N * H * W * C8;
    hipLaunchKernelGGL(avgpool2_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, mcl_stream(stream), (const bf16_t*)x,
                       (bf16_t*)y, N, H, W, C8);
  }
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_maxpool3s2_nhwc_bf16_fwd(const void* x, void* y, void* idx, int32_t N, int32_t H, int32_t W,
                                            int32_t C, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!ok16(x) || !ok16(y) || !idx || (reinterpret_cast<uintptr_t>(idx) & 7u) || N <= 0 || H <= 0 || W <= 0 || C <= 0)
    return MCL_EINVAL;
  if (C % 8) return MCL_EUNSUPPORTED;
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  const long long total = (long long)N * OH * OW * (C / 8);
  hipLaunchKernelGGL(maxpool3s2_fwd_kernel<false>, dim3(blocks_for(total)), dim3(256), 0, mcl_stream(stream),
                     (const bf16_t*)x, (bf16_t*)y, (unsigned char*)idx, N, H, W, OH, OW, C / 8, nullptr, nullptr, nullptr,
                     nullptr);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_bn_act_maxpool_fwd(const void* x, int32_t N, int32_t H, int32_t W, int32_t C, const float* gamma,
                                      const float* beta, const float* mean, const float* rstd, void* y, void* idx,
                                      mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!ok16(x) || !ok16(y) || !idx || (reinterpret_cast<uintptr_t>(idx) & 7u) || !gamma || !beta || !mean || !rstd ||
      N <= 0 || H <= 0 || W <= 0 || C <= 0)
    return MCL_EINVAL;
  if (C % 8) return MCL_EUNSUPPORTED;
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  const long long total = (long long)N * OH * OW * (C / 8);
  hipLaunchKernelGGL(maxpool3s2_fwd_kernel<true>, dim3(blocks_for(total)), dim3(256), 0, mcl_stream(stream),
                     (const bf16_t*)x, (bf16_t*)y, (unsigned char*)idx, N, H, W, OH, OW, C / 8, gamma, beta, mean, rstd);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_maxpool3s2_nhwc_bf16_bwd_ld(const void* idx, const void* dy, int64_t lddy, void* dx, int32_t N, int32_t H,
                                               int32_t W, int32_t C, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!idx || !ok16(dy) || !ok16(dx) || N <= 0 || H <= 0 || W <= 0 || C <= 0 || lddy < C) return MCL_EINVAL;
  if ((C % 8) || (lddy % 8)) return MCL_EUNSUPPORTED;
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  const long long total = (long long)N * H * W * (C / 8);
  if (!(H & 1) && !(W & 1)) {
    hipLaunchKernelGGL(maxpool3s2_bwd_quad_kernel, dim3(blocks_for(total / 4)), dim3(256), 0, mcl_stream(stream),
                       (const unsigned char*)idx, (const bf16_t*)dy, (bf16_t*)dx, N, H, W, OH, OW, C / 8, (long long)lddy);
    MCL_CHECK_LAUNCH();
    return MCL_OK;
  }
  hipLaunchKernelGGL(maxpool3s2_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, mcl_stream(stream),
                     (const unsigned char*)idx, (const bf16_t*)dy, (bf16_t*)dx, N, H, W, OH, OW, C / 8, (long long)lddy);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_maxpool3s2_nhwc_bf16_bwd(const void* idx, const void* dy, void* dx, int32_t N, int32_t H,
                                            int32_t W, int32_t C, mcl_stream_t stream) {
  return mcl_maxpool3s2_nhwc_bf16_bwd_ld(idx, dy, C, dx, N, H, W, C, stream);
}
