// Input pipeline on the GPU: the whole-slide image stays resident in HBM as uint8 HWC and the
// (B, 3, 2r, 2r) patch batch is gathered from it by one launch, instead of B PIL crops + ToTensor on the host and a
// 77 MB fp32 host-to-device copy per step (/root/reference/dataset.py:226-231, 330-336; train.py:34-35).
//   patch_gather_kernel   crop (zero outside the image, as PIL's crop pads), then -- per patch -- horizontal flip,
//                         vertical flip and a counter-clockwise rotation by k*90 degrees (TenxDataset.transform,
//                         dataset.py:315-324: TF.hflip / TF.vflip / TF.rotate(angle in {0, 90, 180, -90}), exact for
//                         square patches), division (by 255 = transforms.ToTensor; by 1 = Tenx raw values), written as the
//                         reference's fp32 NCHW tensor or directly as the bf16 NHWC tensor the backbone kernels read.
//   log_libsize_kernel    scprep.transform.log(scprep.normalize.library_size_normalize(counts)) (dataset.py:188-189):
//                         row / row-sum * 10^4, then log10(x + 1); one wave per spot.
// The reference's other train-time augmentations (ColorJitter, rotation by arbitrary angles on PIL images,
// dataset.py:63-68) are not built.
#include "common.h"

namespace {

typedef unsigned short bf16_t;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ bf16_t f2bf(float f) {
  const f32x2 v = {f, 0.0f};
  return (bf16_t)(__builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t)) & 0xFFFFu);
}

__global__ __launch_bounds__(256) void patch_gather_kernel(const unsigned char* __restrict__ img, int Hs, int Ws,
                                                           const int* __restrict__ centers, int N, int r,
                                                           const unsigned char* __restrict__ ops, float divisor,
                                                           float* __restrict__ out_nchw, bf16_t* __restrict__ out_nhwc) {
  const int P = 2 * r;
  const long long total = (long long)N * P * P;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
    const int x = (int)(q % P), y = (int)((q / P) % P), n = (int)(q / ((long long)P * P));
    const unsigned op = ops ? ops[n] : 0u;
    const int k = (op >> 2) & 3;
    // out = rot_k(Q), Q = vflip(hflip(patch)); PIL transposes: ROTATE_90: out[y][x] = in[x][P-1-y],
    // ROTATE_180: in[P-1-y][P-1-x], ROTATE_270: in[P-1-x][y]
    int qy, qx;
    if (k == 0) { qy = y; qx = x; }
    else if (k == 1) { qy = x; qx = P - 1 - y; }
    else if (k == 2) { qy = P - 1 - y; qx = P - 1 - x; }
    else { qy = P - 1 - x; qx = y; }
    const int sy = (op & 2u) ? P - 1 - qy : qy;          // vflip
    const int sx = (op & 1u) ? P - 1 - qx : qx;          // hflip
    const int row = centers[2 * n] - r + sy, col = centers[2 * n + 1] - r + sx;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
    if (row >= 0 && row < Hs && col >= 0 && col < Ws) {
      const unsigned char* p = img + ((long long)row * Ws + col) * 3;
      v0 = (float)p[0] / divisor;      // a true division, as ToTensor's .div(255)
      v1 = (float)p[1] / divisor;
      v2 = (float)p[2] / divisor;
    }
    if (out_nchw) {
      const long long plane = (long long)P * P;
      float* o = out_nchw + (long long)n * 3 * plane + (long long)y * P + x;
      o[0] = v0; o[plane] = v1; o[2 * plane] = v2;
    }
    if (out_nhwc) {
      bf16_t* o = out_nhwc + q * 3;
      o[0] = f2bf(v0); o[1] = f2bf(v1); o[2] = f2bf(v2);
    }
  }
}

__global__ __launch_bounds__(256) void log_libsize_kernel(const float* __restrict__ x, long long ldx,
                                                          float* __restrict__ y, long long ldy, int rows, int cols,
                                                          float rescale) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (long long)row * ldx;
  float s = 0.f;
  for (int c = lane; c < cols; c += 64) s += xr[c];
  s = wave_sum(s);
  const float f = s != 0.f ? rescale / s : 0.f;          // an empty spot stays all-zero
  float* yr = y + (long long)row * ldy;
  for (int c = lane; c < cols; c += 64) yr[c] = log10f(fmaf(xr[c], f, 1.0f));
}

}  // namespace

extern "C" int mcl_patch_gather(const void* image_u8, int32_t Hs, int32_t Ws, const int32_t* centers_rc, int32_t N,
                                int32_t r, const void* ops, float divisor, float* out_nchw_f32, void* out_nhwc_bf16,
                                mcl_stream_t stream) {
  if (N == 0) return MCL_OK;
  if (!image_u8 || !centers_rc || (!out_nchw_f32 && !out_nhwc_bf16) || Hs <= 0 || Ws <= 0 || N < 0 || r <= 0 || !(divisor > 0.f))
    return MCL_EINVAL;
  MCL_CLEAR_ERROR();
  const long long total = (long long)N * 4 * r * r;
  long long blocks = (total + 255) / 256;
  if (blocks > 65535) blocks = 65535;
  hipLaunchKernelGGL(patch_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, mcl_stream(stream),
                     (const unsigned char*)image_u8, Hs, Ws, centers_rc, N, r, (const unsigned char*)ops, divisor,
                     out_nchw_f32, (bf16_t*)out_nhwc_bf16);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_log_library_size_normalize(const float* counts, int64_t ldx, float* out, int64_t ldy, int32_t rows,
                                              int32_t cols, float rescale, mcl_stream_t stream) {
  if (rows == 0) return MCL_OK;
  if (!counts || !out || rows < 0 || cols <= 0 || ldx < cols || ldy < cols) return MCL_EINVAL;
  MCL_CLEAR_ERROR();
  hipLaunchKernelGGL(log_libsize_kernel, dim3((rows + 3) / 4), dim3(256), 0, mcl_stream(stream), counts,
                     (long long)ldx, out, (long long)ldy, rows, cols, rescale);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}
