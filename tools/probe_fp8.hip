// Probe: operand layout and scale semantics of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 x e4m3) on gfx950.
//   hipcc --offload-arch=gfx950 -O2 tools/probe_fp8.hip -o /tmp/probe_fp8 && /tmp/probe_fp8
// D[i][j] = sum_k A[i][k] * B[j][k]  (A: 32 x 64, B: 32 x 64, both row-major e4m3 bytes).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <math.h>
#include <stdlib.h>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static float e4m3_to_f(uint8_t v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float f;
  if (e == 0) f = ldexpf((float)m, -9);
  else if (e == 15 && m == 7) f = NAN;
  else f = ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -f : f;
}

template <int LAYOUT, int SCALE_MODE>
__global__ void k(const uint8_t* A, const uint8_t* B, float* D, int sa_byte, int sb_byte) {
  const int lane = threadIdx.x;
  i32x8 a, b;
  uint8_t ab[32], bb[32];
  for (int i = 0; i < 32; ++i) {
    int kk;
    if (LAYOUT == 0) kk = 32 * (lane >> 5) + i;                     // 32 consecutive k per lane half
    else kk = 16 * (lane >> 5) + (i & 15) + 32 * (i >> 4);           // two 16-blocks per lane half
    ab[i] = A[(lane & 31) * 64 + kk];
    bb[i] = B[(lane & 31) * 64 + kk];
  }
  for (int i = 0; i < 8; ++i) {
    a[i] = ab[4 * i] | (ab[4 * i + 1] << 8) | (ab[4 * i + 2] << 16) | (ab[4 * i + 3] << 24);
    b[i] = bb[4 * i] | (bb[4 * i + 1] << 8) | (bb[4 * i + 2] << 16) | (bb[4 * i + 3] << 24);
  }
  f32x16 c = {0};
  if (SCALE_MODE == 0) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);
  else c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa_byte * 0x01010101, 0, sb_byte * 0x01010101);
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
    D[row * 32 + col] = c[r];
  }
}

// per-lane scales: lane l supplies scale for A row (l & 31), k-block (l >> 5)?  test with lane-dependent scales
__global__ void k_lane_scale(const uint8_t* A, const uint8_t* B, float* D, const int* sa, const int* sb) {
  const int lane = threadIdx.x;
  i32x8 a, b;
  uint8_t ab[32], bb[32];
  for (int i = 0; i < 32; ++i) {
    const int kk = 32 * (lane >> 5) + i;
    ab[i] = A[(lane & 31) * 64 + kk];
    bb[i] = B[(lane & 31) * 64 + kk];
  }
  for (int i = 0; i < 8; ++i) {
    a[i] = ab[4 * i] | (ab[4 * i + 1] << 8) | (ab[4 * i + 2] << 16) | (ab[4 * i + 3] << 24);
    b[i] = bb[4 * i] | (bb[4 * i + 1] << 8) | (bb[4 * i + 2] << 16) | (bb[4 * i + 3] << 24);
  }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa[lane], 0, sb[lane]);
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
    D[row * 32 + col] = c[r];
  }
}

int main() {
  uint8_t hA[32 * 64], hB[32 * 64];
  srand(1);
  for (int i = 0; i < 32 * 64; ++i) {
    do hA[i] = rand() & 0xFF; while ((hA[i] & 0x7F) == 0x7F || ((hA[i] >> 3) & 15) > 9);
    do hB[i] = rand() & 0xFF; while ((hB[i] & 0x7F) == 0x7F || ((hB[i] >> 3) & 15) > 9);
  }
  float ref[32 * 32];
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      double s = 0;
      for (int kk = 0; kk < 64; ++kk) s += (double)e4m3_to_f(hA[i * 64 + kk]) * e4m3_to_f(hB[j * 64 + kk]);
      ref[i * 32 + j] = (float)s;
    }
  uint8_t *dA, *dB;
  float* dD;
  int *dsa, *dsb;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, 4096); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
  hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  float out[1024];
  auto report = [&](const char* name, double expect_scale) {
    hipDeviceSynchronize();
    hipMemcpy(out, dD, 4096, hipMemcpyDeviceToHost);
    double worst = 0, wt = 0;
    for (int i = 0; i < 1024; ++i) {
      worst = fmax(worst, fabs(out[i] - ref[i] * expect_scale));
      wt = fmax(wt, fabs(out[(i % 32) * 32 + i / 32] - ref[i] * expect_scale));
    }
    printf("%-44s max|D - ref*%g| = %.4g   (transposed: %.4g)   D[0][1]=%g ref=%g\n", name, expect_scale, worst, wt, out[1], ref[1]);
  };
  hipLaunchKernelGGL((k<0, 0>), 1, 64, 0, 0, dA, dB, dD, 0, 0); report("layout 0 (32 consecutive k), scale arg 0", 1.0);
  hipLaunchKernelGGL((k<1, 0>), 1, 64, 0, 0, dA, dB, dD, 0, 0); report("layout 1 (two 16-blocks), scale arg 0", 1.0);
  hipLaunchKernelGGL((k<0, 1>), 1, 64, 0, 0, dA, dB, dD, 127, 127); report("layout 0, scales 127/127 (=1.0)", 1.0);
  hipLaunchKernelGGL((k<0, 1>), 1, 64, 0, 0, dA, dB, dD, 128, 127); report("layout 0, scales 128/127 (A x2)", 2.0);
  hipLaunchKernelGGL((k<0, 1>), 1, 64, 0, 0, dA, dB, dD, 127, 125); report("layout 0, scales 127/125 (B /4)", 0.25);
  // lane-dependent scales: A row i scaled by 2^(i%3), B row j by 2^-(j%2); same for both k halves
  int hsa[64], hsb[64];
  for (int l = 0; l < 64; ++l) {
    hsa[l] = (127 + (l & 31) % 3) * 0x01010101;
    hsb[l] = (127 - (l & 31) % 2) * 0x01010101;
  }
  hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice);
  hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_lane_scale, 1, 64, 0, 0, dA, dB, dD, dsa, dsb);
  hipDeviceSynchronize();
  hipMemcpy(out, dD, 4096, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j)
      worst = fmax(worst, fabs(out[i * 32 + j] - ref[i * 32 + j] * ldexp(1.0, i % 3) * ldexp(1.0, -(j % 2))));
  printf("per-lane row scales (lane l -> row l&31)      max err = %.4g\n", worst);
  return 0;
}
