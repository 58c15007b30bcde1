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
//   her2st_train_kernel  the HER2ST / cSCC TRAINING transform (dataset.py:63-68: ColorJitter(0.5, 0.5, 0.5),
//                         RandomHorizontalFlip, RandomRotation(180), ToTensor) for a whole batch with the random draws
//                         supplied per patch: one workgroup per patch keeps the cropped patch in LDS (150 KB at 224 x 224),
//                         applies PIL's ImageEnhance arithmetic bit for bit (Image.blend in fp32 with separate multiply
//                         and add roundings, the 16.16 fixed-point luma, the integer mean of Contrast) and reads it out
//                         through the flip and PIL's nearest-neighbour affine rotation (16.16 fixed point, Geometry.c).
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

// ---- PIL arithmetic, restated ------------------------------------------------------------------------------------
__device__ __forceinline__ int luma(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }

// Image.blend(in1 = degenerate, in2 = image, alpha): (UINT8)(in1 + alpha*(in2 - in1)) in fp32 with the product and the
// sum rounded separately (no FMA contraction: libImaging is plain C), truncated; clipped to [0, 255] when extrapolating.
__device__ __forceinline__ int blend1(int in1, int in2, float alpha, bool interp) {
  // hipcc contracts a*b + c into one FMA by default (and HIP's __fmul_rn / __fadd_rn are plain operators): two
  // roundings are part of the result here (rare ties otherwise land one grey level off PIL's)
#pragma clang fp contract(off)
  const float prod = alpha * (float)(in2 - in1);
  const float t = (float)in1 + prod;
  if (interp) return (int)t & 255;
  return t <= 0.0f ? 0 : (t >= 255.0f ? 255 : (int)t);
}

struct AugParams {        // one per patch (host fills it, see input_pipeline.her2st_train_patches)
  int order;              // the three colour adjustments in application order, 2 bits each: 0 brightness, 1 contrast, 2 saturation
  int hflip;              // RandomHorizontalFlip drew "flip"
  int rot_mode;           // 0: quarter turns (rot_k counter-clockwise), 1: affine nearest (fixed-point coefficients a[])
  int rot_k;
  float fb, fc, fs;       // brightness / contrast / saturation factors
  int a[6];               // libImaging affine_fixed coefficients (16.16)
  int pad[3];
};

__global__ __launch_bounds__(1024) void her2st_train_kernel(const unsigned char* __restrict__ img, int Hs, int Ws,
                                                            const int* __restrict__ centers, int r,
                                                            const AugParams* __restrict__ params, float divisor,
                                                            float* __restrict__ out_nchw, bf16_t* __restrict__ out_nhwc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pl[];      // [P*P][3] uint8
  __shared__ int red[16];
  const int n = blockIdx.x, tid = threadIdx.x, P = 2 * r, npx = P * P;
  const AugParams pr = params[n];
  const int r0 = centers[2 * n] - r, c0 = centers[2 * n + 1] - r;
  for (int q = tid; q < npx; q += 1024) {                  // Image.crop: zero outside the slide
    const int y = q / P, x = q - y * P, row = r0 + y, col = c0 + x;
    unsigned char v0 = 0, v1 = 0, v2 = 0;
    if (row >= 0 && row < Hs && col >= 0 && col < Ws) {
      const unsigned char* p = img + ((long long)row * Ws + col) * 3;
      v0 = p[0]; v1 = p[1]; v2 = p[2];
    }
    pl[3 * q] = v0; pl[3 * q + 1] = v1; pl[3 * q + 2] = v2;
  }
  __syncthreads();
  for (int step = 0; step < 3; ++step) {
    const int op = (pr.order >> (2 * step)) & 3;
    if (op == 0) {                                         // ImageEnhance.Brightness: degenerate = black
      const bool interp = pr.fb >= 0.0f && pr.fb <= 1.0f;
      for (int q = tid; q < 3 * npx; q += 1024) pl[q] = (unsigned char)blend1(0, pl[q], pr.fb, interp);
    } else if (op == 1) {                                  // ImageEnhance.Contrast: degenerate = int(mean(L) + 0.5)
      int s = 0;
      for (int q = tid; q < npx; q += 1024) s += luma(pl[3 * q], pl[3 * q + 1], pl[3 * q + 2]);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      if ((tid & 63) == 0) red[tid >> 6] = s;
      __syncthreads();
      int tot = 0;
#pragma unroll
      for (int w = 0; w < 16; ++w) tot += red[w];
      const int mean = (int)((2LL * tot + npx) / (2LL * npx));     // == int(tot / npx + 0.5), exactly
      const bool interp = pr.fc >= 0.0f && pr.fc <= 1.0f;
      __syncthreads();
      for (int q = tid; q < 3 * npx; q += 1024) pl[q] = (unsigned char)blend1(mean, pl[q], pr.fc, interp);
    } else {                                               // ImageEnhance.Color: degenerate = per-pixel luma
      const bool interp = pr.fs >= 0.0f && pr.fs <= 1.0f;
      for (int q = tid; q < npx; q += 1024) {
        const int a = pl[3 * q], b = pl[3 * q + 1], c = pl[3 * q + 2];
        const int L = luma(a, b, c);
        pl[3 * q] = (unsigned char)blend1(L, a, pr.fs, interp);
        pl[3 * q + 1] = (unsigned char)blend1(L, b, pr.fs, interp);
        pl[3 * q + 2] = (unsigned char)blend1(L, c, pr.fs, interp);
      }
    }
    __syncthreads();
  }
  // read-out: out = rotate(hflip(jittered)); ToTensor
  const long long plane = (long long)npx;
  for (int q = tid; q < npx; q += 1024) {
    const int y = q / P, x = q - y * P;
    int yin, xin;
    bool ok = true;
    if (pr.rot_mode == 0) {
      const int k = pr.rot_k & 3;
      if (k == 0) { yin = y; xin = x; }
      else if (k == 1) { yin = x; xin = P - 1 - y; }
      else if (k == 2) { yin = P - 1 - y; xin = P - 1 - x; }
      else { yin = P - 1 - x; xin = y; }
    } else {
      xin = (pr.a[2] + y * pr.a[1] + x * pr.a[0]) >> 16;
      yin = (pr.a[5] + y * pr.a[4] + x * pr.a[3]) >> 16;
      ok = xin >= 0 && xin < P && yin >= 0 && yin < P;
    }
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
    if (ok) {
      const int sx = pr.hflip ? P - 1 - xin : xin;
      const unsigned char* p = pl + 3 * (yin * P + sx);
      v0 = (float)p[0] / divisor; v1 = (float)p[1] / divisor; v2 = (float)p[2] / divisor;
    }
    if (out_nchw) {
      float* o = out_nchw + (long long)n * 3 * plane + q;
      o[0] = v0; o[plane] = v1; o[2 * plane] = v2;
    }
    if (out_nhwc) {
      bf16_t* o = out_nhwc + ((long long)n * plane + q) * 3;
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


// HER2ST / cSCC training transform for a batch (dataset.py:63-68) with the random draws given per patch: params = N
// records of 16 int32 words {order, hflip, rot_mode, rot_k, fb, fc, fs (float bits), a[6], pad[3]}.  2r <= 232 (the
// patch lives in LDS).
extern "C" int mcl_her2st_train_patches(const void* image_u8, int32_t Hs, int32_t Ws, const int32_t* centers_rc, int32_t N,
                                        int32_t r, const void* params, float divisor, float* out_nchw_f32,
                                        void* out_nhwc_bf16, mcl_stream_t stream) {
  if (N == 0) return MCL_OK;
  if (!image_u8 || !centers_rc || !params || (!out_nchw_f32 && !out_nhwc_bf16) || Hs <= 0 || Ws <= 0 || N < 0 || r <= 0 ||
      !(divisor > 0.f))
    return MCL_EINVAL;
  const size_t lds_bytes = ((size_t)4 * r * r * 3 + 15) & ~(size_t)15;
  if (lds_bytes > 160 * 1024 - 256) return MCL_EUNSUPPORTED;
  MCL_CLEAR_ERROR();
  static mcl_device_once attr_once;
  if (auto attr_guard = attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(her2st_train_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
  }
  hipLaunchKernelGGL(her2st_train_kernel, dim3(N), dim3(1024), lds_bytes, mcl_stream(stream),
                     (const unsigned char*)image_u8, Hs, Ws, centers_rc, r, (const AugParams*)params, divisor,
                     out_nchw_f32, (bf16_t*)out_nhwc_bf16);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}
