// The "glue" of the image encoders on own kernels (round 6, VERDICT r05 missing #5): what was left to ATen elementwise launches
// around the hand-written contraction / normalisation / attention kernels --
//   * the ViT's token assembly (/root/reference/model.py:104-116 via timm's VisionTransformer.forward_features: patch tokens,
//     class token, position embedding) and its `global_pool="avg"` token mean, forward and backward, bf16 and fp32;
//   * strided channels-last copies (a channel slice of a concat / gradient buffer handed to a kernel that wants dense rows) and
//     the 180-degree rotated, role-swapped weight of the "same"-convolution form of a backward-data pass (fp32 DenseNet mode,
//     /root/reference/model.py:72-85).
// All of them are single-pass, HBM-bound elementwise kernels; every reduction is a fixed-order loop (deterministic).
#include "common.h"

namespace {

typedef unsigned short bf16_t;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ float ldf(const bf16_t* p) { return __uint_as_float(((unsigned)*p) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  const f32x2 v = {f, 0.0f};
  return (bf16_t)(__builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t)) & 0xFFFFu);
}
__device__ __forceinline__ void stf(float* p, float v) { *p = v; }
__device__ __forceinline__ void stf(bf16_t* p, float v) { *p = f2bf(v); }

// ---- 2-D copy: dst[r][0..row_units) = src[r][0..row_units) in units of U bytes, row strides in bytes
template <typename U>
__global__ __launch_bounds__(256) void copy_rows_kernel(const unsigned char* __restrict__ src, long long lds_b,
                                                        unsigned char* __restrict__ dst, long long ldd_b, long long rows,
                                                        long long row_units) {
  const long long total = rows * row_units;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
    const long long r = q / row_units, c = q - r * row_units;
    reinterpret_cast<U*>(dst + r * ldd_b)[c] = reinterpret_cast<const U*>(src + r * lds_b)[c];
  }
}

// ---- wf[ci][ky][kx][co] = w[co][k-1-ky][k-1-kx][ci]   (w: (Co, k, k, Ci) storage = a channels-last Conv2d weight)
template <typename T>
__global__ __launch_bounds__(256) void weight_rot180_kernel(const T* __restrict__ w, T* __restrict__ wf, int Co, int k, int Ci) {
  const long long total = (long long)Co * k * k * Ci;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
    const int co = (int)(q % Co);
    long long t = q / Co;
    const int kx = (int)(t % k);
    t /= k;
    const int ky = (int)(t % k);
    const int ci = (int)(t / k);
    wf[q] = w[(((long long)co * k + (k - 1 - ky)) * k + (k - 1 - kx)) * Ci + ci];
  }
}

// ---- 4-D strided copy of an fp32 source: dst[i . sd] (+)= src[i . ss] over (n0, n1, n2, n3), dst fp32 or bf16 (element strides).
// Packs a Conv2d weight of any memory format into the (c, iy, ix) column order of the patch GEMM (with the bf16 cast), and adds a
// contiguous weight gradient into a .grad of the parameter's own strides.
template <typename OUT>
__global__ __launch_bounds__(256) void strided4_kernel(const float* __restrict__ src, int n1, int n2, int n3, long long a0,
                                                       long long a1, long long a2, long long a3, long long d0, long long d1,
                                                       long long d2, long long d3, long long total, OUT* __restrict__ dst,
                                                       int accumulate) {
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
    const int i3 = (int)(q % n3);
    long long t = q / n3;
    const int i2 = (int)(t % n2);
    t /= n2;
    const int i1 = (int)(t % n1);
    const long long i0 = t / n1;
    const float v = src[i0 * a0 + i1 * a1 + i2 * a2 + i3 * a3];
    OUT* o = dst + i0 * d0 + i1 * d1 + i2 * d2 + i3 * d3;
    stf(o, accumulate ? ldf(o) + v : v);
  }
}

// ---- ViT tokens.  x: (B, T, D), T = np + 1, row 0 of an image = the class token.
// row 0 <- cls + pos[0]  (and, ASSEMBLE: rows 1.. <- tok[b*np + t-1] + pos[t])
template <typename T, bool ASSEMBLE>
__global__ __launch_bounds__(256) void vit_assemble_kernel(const float* __restrict__ tok, const float* __restrict__ cls,
                                                           const float* __restrict__ pos, T* __restrict__ x, int B, int Tn,
                                                           int D) {
  const long long rows = ASSEMBLE ? (long long)B * Tn : B;
  const long long total = rows * D;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
    const int d = (int)(q % D);
    const long long r = q / D;
    const long long b = ASSEMBLE ? r / Tn : r;
    const int t = ASSEMBLE ? (int)(r - b * Tn) : 0;
    const float v = (t == 0 ? cls[d] : tok[(b * (Tn - 1) + (t - 1)) * D + d]) + pos[(long long)t * D + d];
    stf(x + (b * Tn + t) * D + d, v);
  }
}

// dtok[b*np + t-1] <- dx[b][t], t >= 1
template <typename T>
__global__ __launch_bounds__(256) void vit_tokens_extract_kernel(const T* __restrict__ dx, T* __restrict__ dtok, int B, int Tn,
                                                                 int D) {
  const long long total = (long long)B * (Tn - 1) * D;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
    const int d = (int)(q % D);
    const long long r = q / D;
    const long long b = r / (Tn - 1);
    const int t = (int)(r - b * (Tn - 1)) + 1;
    dtok[q] = dx[(b * Tn + t) * D + d];
  }
}

// feat[b][d] = mean_{t >= 1} x[b][t][d]   (fp32 accumulation in token order)
template <typename T>
__global__ __launch_bounds__(256) void vit_token_mean_fwd_kernel(const T* __restrict__ x, float* __restrict__ feat, int B, int Tn,
                                                                 int D) {
  const long long total = (long long)B * D;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
    const int d = (int)(q % D);
    const long long b = q / D;
    const T* p = x + (b * Tn + 1) * D + d;
    float s = 0.0f;
    for (int t = 1; t < Tn; ++t, p += D) s += ldf(p);
    feat[q] = s / (float)(Tn - 1);
  }
}

// dx[b][0] = 0 ; dx[b][t >= 1] = dfeat[b] / np
template <typename T>
__global__ __launch_bounds__(256) void vit_token_mean_bwd_kernel(const float* __restrict__ dfeat, T* __restrict__ dx, int B,
                                                                 int Tn, int D) {
  const long long total = (long long)B * Tn * D;
  const float np = (float)(Tn - 1);
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
    const int d = (int)(q % D);
    const long long r = q / D;
    const long long b = r / Tn;
    const int t = (int)(r - b * Tn);
    stf(dx + q, t == 0 ? 0.0f : dfeat[b * D + d] / np);
  }
}

// dpos[t][d] (+)= sum_b dx[b][t][d]  (batch order) ; dcls[d] (+)= the t = 0 row of that sum  (accumulate: bit 0 dpos, bit 1 dcls)
template <typename T>
__global__ __launch_bounds__(256) void vit_pos_grad_kernel(const T* __restrict__ dx, float* __restrict__ dpos,
                                                           float* __restrict__ dcls, int B, int Tn, int D, int accumulate) {
  const long long total = (long long)Tn * D;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
    float s = 0.0f;
    for (int b = 0; b < B; ++b) s += ldf(dx + (long long)b * total + q);
    dpos[q] = (accumulate & 1) ? dpos[q] + s : s;
    if (q < D && dcls) dcls[q] = (accumulate & 2) ? dcls[q] + s : s;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void vit_zero_cls_rows_kernel(T* __restrict__ dx, int B, int Tn, int D) {
  const long long total = (long long)B * D;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
    const long long b = q / D;
    stf(dx + b * Tn * D + (q - b * D), 0.0f);
  }
}

inline unsigned nblocks(long long total) {
  long long b = (total + 255) / 256;
  if (b < 1) b = 1;
  if (b > 65535 * 8) b = 65535 * 8;
  return (unsigned)b;
}

}  // namespace

extern "C" int mcl_copy_rows(const void* src, int64_t ld_src_bytes, void* dst, int64_t ld_dst_bytes, int64_t rows,
                             int64_t row_bytes, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!src || !dst || rows <= 0 || row_bytes <= 0 || ld_src_bytes < row_bytes || ld_dst_bytes < row_bytes) return MCL_EINVAL;
  const uintptr_t all = reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) | (uintptr_t)ld_src_bytes |
                        (uintptr_t)ld_dst_bytes | (uintptr_t)row_bytes;
  hipStream_t st = mcl_stream(stream);
  const unsigned char* s = (const unsigned char*)src;
  unsigned char* d = (unsigned char*)dst;
  if (!(all & 15u))
    hipLaunchKernelGGL(copy_rows_kernel<uint4>, dim3(nblocks(rows * (row_bytes / 16))), dim3(256), 0, st, s, (long long)ld_src_bytes,
                       d, (long long)ld_dst_bytes, (long long)rows, (long long)(row_bytes / 16));
  else if (!(all & 3u))
    hipLaunchKernelGGL(copy_rows_kernel<unsigned>, dim3(nblocks(rows * (row_bytes / 4))), dim3(256), 0, st, s,
                       (long long)ld_src_bytes, d, (long long)ld_dst_bytes, (long long)rows, (long long)(row_bytes / 4));
  else if (!(all & 1u))
    hipLaunchKernelGGL(copy_rows_kernel<unsigned short>, dim3(nblocks(rows * (row_bytes / 2))), dim3(256), 0, st, s,
                       (long long)ld_src_bytes, d, (long long)ld_dst_bytes, (long long)rows, (long long)(row_bytes / 2));
  else
    return MCL_EUNSUPPORTED;
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_weight_rot180(const void* w, void* wf, int32_t Co, int32_t k, int32_t Ci, int32_t dtype, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!w || !wf || Co <= 0 || k <= 0 || Ci <= 0) return MCL_EINVAL;
  const long long total = (long long)Co * k * k * Ci;
  if (dtype == 0)
    hipLaunchKernelGGL(weight_rot180_kernel<float>, dim3(nblocks(total)), dim3(256), 0, mcl_stream(stream), (const float*)w,
                       (float*)wf, Co, k, Ci);
  else if (dtype == 1)
    hipLaunchKernelGGL(weight_rot180_kernel<bf16_t>, dim3(nblocks(total)), dim3(256), 0, mcl_stream(stream), (const bf16_t*)w,
                       (bf16_t*)wf, Co, k, Ci);
  else
    return MCL_EUNSUPPORTED;
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_strided4_f32(const float* src, int32_t n0, int32_t n1, int32_t n2, int32_t n3, int64_t a0, int64_t a1, int64_t a2,
                                int64_t a3, void* dst, int64_t d0, int64_t d1, int64_t d2, int64_t d3, int32_t dst_dtype,
                                int32_t accumulate, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!src || !dst || n0 <= 0 || n1 <= 0 || n2 <= 0 || n3 <= 0) return MCL_EINVAL;
  const long long total = (long long)n0 * n1 * n2 * n3;
  if (dst_dtype == 0)
    hipLaunchKernelGGL(strided4_kernel<float>, dim3(nblocks(total)), dim3(256), 0, mcl_stream(stream), src, n1, n2, n3, (long long)a0,
                       (long long)a1, (long long)a2, (long long)a3, (long long)d0, (long long)d1, (long long)d2, (long long)d3, total,
                       (float*)dst, accumulate);
  else if (dst_dtype == 1)
    hipLaunchKernelGGL(strided4_kernel<bf16_t>, dim3(nblocks(total)), dim3(256), 0, mcl_stream(stream), src, n1, n2, n3,
                       (long long)a0, (long long)a1, (long long)a2, (long long)a3, (long long)d0, (long long)d1, (long long)d2,
                       (long long)d3, total, (bf16_t*)dst, accumulate);
  else
    return MCL_EUNSUPPORTED;
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_vit_cls_row(const float* cls, const float* pos, void* x, int32_t B, int32_t T, int32_t D, int32_t dtype,
                               mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!cls || !pos || !x || B <= 0 || T <= 1 || D <= 0) return MCL_EINVAL;
  const long long total = (long long)B * D;
  if (dtype == 0)
    hipLaunchKernelGGL((vit_assemble_kernel<float, false>), dim3(nblocks(total)), dim3(256), 0, mcl_stream(stream), nullptr, cls, pos,
                       (float*)x, B, T, D);
  else if (dtype == 1)
    hipLaunchKernelGGL((vit_assemble_kernel<bf16_t, false>), dim3(nblocks(total)), dim3(256), 0, mcl_stream(stream), nullptr, cls,
                       pos, (bf16_t*)x, B, T, D);
  else
    return MCL_EUNSUPPORTED;
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_vit_assemble_f32(const float* tok, const float* cls, const float* pos, float* x, int32_t B, int32_t T, int32_t D,
                                    mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!tok || !cls || !pos || !x || B <= 0 || T <= 1 || D <= 0) return MCL_EINVAL;
  hipLaunchKernelGGL((vit_assemble_kernel<float, true>), dim3(nblocks((long long)B * T * D)), dim3(256), 0, mcl_stream(stream), tok,
                     cls, pos, x, B, T, D);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_vit_tokens_extract(const void* dx, void* dtok, int32_t B, int32_t T, int32_t D, int32_t dtype,
                                      mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!dx || !dtok || B <= 0 || T <= 1 || D <= 0) return MCL_EINVAL;
  if (dtype == 0)
    hipLaunchKernelGGL(vit_tokens_extract_kernel<float>, dim3(nblocks((long long)B * (T - 1) * D)), dim3(256), 0, mcl_stream(stream),
                       (const float*)dx, (float*)dtok, B, T, D);
  else if (dtype == 1)
    hipLaunchKernelGGL(vit_tokens_extract_kernel<bf16_t>, dim3(nblocks((long long)B * (T - 1) * D)), dim3(256), 0,
                       mcl_stream(stream), (const bf16_t*)dx, (bf16_t*)dtok, B, T, D);
  else
    return MCL_EUNSUPPORTED;
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_vit_token_mean_fwd(const void* x, float* feat, int32_t B, int32_t T, int32_t D, int32_t dtype,
                                      mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!x || !feat || B <= 0 || T <= 1 || D <= 0) return MCL_EINVAL;
  if (dtype == 0)
    hipLaunchKernelGGL(vit_token_mean_fwd_kernel<float>, dim3(nblocks((long long)B * D)), dim3(256), 0, mcl_stream(stream),
                       (const float*)x, feat, B, T, D);
  else if (dtype == 1)
    hipLaunchKernelGGL(vit_token_mean_fwd_kernel<bf16_t>, dim3(nblocks((long long)B * D)), dim3(256), 0, mcl_stream(stream),
                       (const bf16_t*)x, feat, B, T, D);
  else
    return MCL_EUNSUPPORTED;
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_vit_token_mean_bwd(const float* dfeat, void* dx, int32_t B, int32_t T, int32_t D, int32_t dtype,
                                      mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!dfeat || !dx || B <= 0 || T <= 1 || D <= 0) return MCL_EINVAL;
  if (dtype == 0)
    hipLaunchKernelGGL(vit_token_mean_bwd_kernel<float>, dim3(nblocks((long long)B * T * D)), dim3(256), 0, mcl_stream(stream), dfeat,
                       (float*)dx, B, T, D);
  else if (dtype == 1)
    hipLaunchKernelGGL(vit_token_mean_bwd_kernel<bf16_t>, dim3(nblocks((long long)B * T * D)), dim3(256), 0, mcl_stream(stream),
                       dfeat, (bf16_t*)dx, B, T, D);
  else
    return MCL_EUNSUPPORTED;
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_vit_pos_grad(const void* dx, float* dpos, float* dcls, int32_t B, int32_t T, int32_t D, int32_t dtype,
                                int32_t accumulate, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!dx || !dpos || B <= 0 || T <= 1 || D <= 0) return MCL_EINVAL;
  if (dtype == 0)
    hipLaunchKernelGGL(vit_pos_grad_kernel<float>, dim3(nblocks((long long)T * D)), dim3(256), 0, mcl_stream(stream),
                       (const float*)dx, dpos, dcls, B, T, D, accumulate);
  else if (dtype == 1)
    hipLaunchKernelGGL(vit_pos_grad_kernel<bf16_t>, dim3(nblocks((long long)T * D)), dim3(256), 0, mcl_stream(stream),
                       (const bf16_t*)dx, dpos, dcls, B, T, D, accumulate);
  else
    return MCL_EUNSUPPORTED;
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_vit_zero_cls_rows(void* dx, int32_t B, int32_t T, int32_t D, int32_t dtype, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!dx || B <= 0 || T <= 1 || D <= 0) return MCL_EINVAL;
  if (dtype == 0)
    hipLaunchKernelGGL(vit_zero_cls_rows_kernel<float>, dim3(nblocks((long long)B * D)), dim3(256), 0, mcl_stream(stream), (float*)dx,
                       B, T, D);
  else if (dtype == 1)
    hipLaunchKernelGGL(vit_zero_cls_rows_kernel<bf16_t>, dim3(nblocks((long long)B * D)), dim3(256), 0, mcl_stream(stream),
                       (bf16_t*)dx, B, T, D);
  else
    return MCL_EUNSUPPORTED;
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}
