// Probe: semantics of ds_read_b64_tr_b16 (gfx950) with per-lane addresses into a pitched tile.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned short* in, unsigned short* out) {
  __shared__ unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = in[i];
  __syncthreads();
  const int l = threadIdx.x;
  const int g = l >> 4, i = l & 15;
  // block g: rows 0..3 (pitch 160), cols 16g..16g+15 ; lane i supplies row i/4, col chunk (i%4)*4
  const unsigned short* p = &lds[g * 16 + (i >> 2) * 160 + (i & 3) * 4];
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)p);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)r[j];
}
int main() {
  unsigned short h[4096], o[256];
  for (int i = 0; i < 4096; ++i) h[i] = (unsigned short)((i / 160) * 1000 + (i % 160));  // row*1000 + col
  unsigned short *d, *e;
  hipMalloc(&d, sizeof(h)); hipMalloc(&e, sizeof(o));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e);
  hipMemcpy(o, e, sizeof(o), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, o[4*l], o[4*l+1], o[4*l+2], o[4*l+3]);
  return 0;
}
