// Weight gradient of the DenseNet growth 3x3 convolution, "row-walking" form for the large maps (image width 17..64: the
// 56 x 56 and 28 x 28 dense blocks of torchvision's DenseNet-121 that /root/reference/model.py:75-76 wraps), with
// norm2 + relu2 recomputed on the fly, deterministic:
//
//     dW2[co][ky][kx][ci] (+)= sum_{y, x} dy[y][x][co] * a2[y + ky - 1][x + kx - 1][ci],   a2 = relu(bn2(z))   (zero outside)
//
// A "TN" GEMM M = 32 (co), N = 9 x 128 (tap, ci), K = S pixels; both operands are pixel-major in HBM, so both MFMA
// fragments come from transposing LDS reads (ds_read_b64_tr_b16).  The kernel-row form of dense_conv.hip stages a 130-pixel z
// tile three times (once per kernel row, BatchNorm + ReLU each time), synchronises the workgroup twice per tile and reads
// two fresh fragments per MFMA (115 us per 56 x 56 layer = 10 % of the MFMA peak).  Here the contraction is re-associated
// around ONE a2 row:
//
//     dW2[:, ky, kx, :] += sum_x' dy[r + 1 - ky][x' + 1 - kx][:]^T  a2[r][x'][:]        for every image row r
//
//   * a wave owns (image, row chunk, input-channel quarter) and walks down the rows r of its chunk.  From one staged a2 row
//     (its 32 channels only, transformed once) every 16-pixel fragment feeds NINE MFMAs -- the nine taps -- against
//     fragments of the dy rows r+1, r, r-1 shifted by +1, 0, -1 pixels: 10 fragment reads per 9 MFMAs, nine independent
//     accumulators (144 registers) that live for the whole kernel;
//   * no workgroup synchronisation in the main loop: private LDS per wave (the a2 row and a three-slot ring of dy rows with
//     one zero pixel left and right, plain [pixel][64 B] rows: a fragment read touches 4 consecutive pixels = 256 contiguous
//     bytes, conflict-free at any shift); rows outside the image are read through zero-size buffer descriptors and
//     contribute zeros, so the MFMA loop has no masks and no dispatch over valid taps;
//   * the next row's z / dy chunks are requested before the row is multiplied;
//   * the two row streams of a workgroup meet in LDS at the end (ring memory is dead by then) and the workgroup writes ONE
//     32 x 1152 fp32 partial; a fixed-order merge launch (wrw_fused.hip) adds the <= 256 partials into dW2.
#include "common.h"
#include <stdlib.h>

namespace {

typedef unsigned short bf16_t;
typedef short v4s __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

constexpr int WR_NWAVE = 8;                       // two row streams x four channel quarters
constexpr int WR_MN = 32 * 9 * 128;               // elements of dW2
constexpr int WR_RED_BYTES = 4 * 9 * 16 * 64 * 4; // the second stream's accumulators in LDS at the end: 147,456 B

__device__ __forceinline__ unsigned wr_pack2(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

// 8 consecutive pixels (from pixel row p0 + 8 h of a [pixel][64 B] tile) of channel l & 31: two transposing reads, each a
// 4-pixel x 16-channel block per 16-lane group.  ``la`` = this lane's byte offset inside the first block.
__device__ __forceinline__ bf16x8 wr_frag(const unsigned char* p) {
  const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)p);
  const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(p + 256));
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <int NKS>
__global__ __launch_bounds__(64 * WR_NWAVE, 2) void conv3x3_wrw_rows_kernel(
    const bf16_t* __restrict__ dy, long long lddy, const bf16_t* __restrict__ z, int nimg, int H, int W,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
    const float* __restrict__ rstd, int rc, int nchunk, float* __restrict__ wpart) {
  constexpr int A2_BYTES = 16 * NKS * 64;          // staged a2 row: 16 NKS pixels x 32 channels
  constexpr int SLOT = (16 * NKS + 2) * 64;        // staged dy row: pixel index i = x + 1, x = -1 .. 16 NKS
  constexpr int WAVE_LDS = A2_BYTES + 3 * SLOT;
  constexpr int NZ = NKS, ND = NKS + 1;            // 16-byte chunks per lane and row: z quarter row, dy row
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const int q = wave & 3, stream = wave >> 2;
  unsigned char* a2buf = lds + wave * WAVE_LDS;
  unsigned char* ring = a2buf + A2_BYTES;

  // BatchNorm (scale, shift) of this lane's 8 staging channels 32 q + 8 (lane & 3) + j
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = 32 * q + 8 * (lane & 3) + j;
    sc[j] = gamma[c] * rstd[c];
    sh[j] = fmaf(-mean[c], sc[j], beta[c]);
  }

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

  // fragment lane offset: pixel row 8 h + ((lane & 15) >> 2), 32-byte half (lane >> 4) & 1, 8-byte piece lane & 3
  const int la = (8 * h + ((lane & 15) >> 2)) * 64 + 32 * ((lane >> 4) & 1) + 8 * (lane & 3);
  const unsigned zrow_bytes = (unsigned)W * 256u;
  const unsigned dyrow_bytes = (unsigned)(((long long)(W - 1) * lddy + 32) * 2);
  const unsigned lddy2 = (unsigned)(lddy * 2);
  const int nunit = nimg * nchunk;
  const int ustride = gridDim.x * 2;

  for (int u = blockIdx.x * 2 + stream; u < nunit; u += ustride) {
    const int chunk = u % nchunk, b = u / nchunk;
    const int j0 = chunk * rc, j1 = min(H, j0 + rc);
    const long long img = (long long)b * H;
    auto opaque_lane = [&]() { int ln = lane; asm volatile("" : "+v"(ln)); return ln; };

    // z row r, this wave's channel quarter: chunk tt covers pixels 16 tt + (ln >> 2), 16 bytes at (ln & 3) of the 64-byte
    // quarter row.  Rows past the unit read through a zero-size descriptor (zeros, no memory access).
    auto load_z = [&](int r, u32x4 (&v)[NZ]) {
      const int ln = opaque_lane();
      const __amdgpu_buffer_rsrc_t row = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<bf16_t*>(z) + (img + min(r, H - 1)) * W * 128, 0, r < j1 ? zrow_bytes : 0u, 0x00020000);
      const unsigned off = (unsigned)(ln >> 2) * 256u + (unsigned)q * 64u + (unsigned)(ln & 3) * 16u;
#pragma unroll
      for (int tt = 0; tt < NZ; ++tt) v[tt] = __builtin_amdgcn_raw_buffer_load_b128(row, off + tt * 4096u, 0, 0);
    };
    // a2 = relu(bn2(z)) of the row -> a2buf; pixels beyond the image width are zeros (their z reads returned zeros, but
    // relu(shift) need not be zero)
    auto write_a2 = [&](const u32x4 (&v)[NZ]) {
      const int ln = opaque_lane();
#pragma unroll
      for (int tt = 0; tt < NZ; ++tt) {
        const int x = 16 * tt + (ln >> 2);
        unsigned w[4] = {v[tt][0], v[tt][1], v[tt][2], v[tt][3]};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const float lo = fmaxf(fmaf(__uint_as_float(w[d] << 16), sc[2 * d], sh[2 * d]), 0.0f);
          const float hi = fmaxf(fmaf(__uint_as_float(w[d] & 0xFFFF0000u), sc[2 * d + 1], sh[2 * d + 1]), 0.0f);
          w[d] = x < W ? wr_pack2(lo, hi) : 0u;
        }
        *reinterpret_cast<uint4*>(a2buf + x * 64 + (ln & 3) * 16) = make_uint4(w[0], w[1], w[2], w[3]);
      }
    };
    // dy row y: chunk tt covers pixel index i = 16 tt + (ln >> 2) (x = i - 1).  Rows outside the image: zero-size descriptor.
    auto load_dy = [&](int y, u32x4 (&d)[ND]) {
      const int ln = opaque_lane();
      const bool inr = y >= 0 && y < H;
      const __amdgpu_buffer_rsrc_t row = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<bf16_t*>(dy) + (img + min(max(y, 0), H - 1)) * W * lddy, 0, inr ? dyrow_bytes : 0u, 0x00020000);
#pragma unroll
      for (int tt = 0; tt < ND; ++tt) {
        const int x = 16 * tt + (ln >> 2) - 1;
        d[tt] = __builtin_amdgcn_raw_buffer_load_b128(row, (unsigned)max(x, 0) * lddy2 + (unsigned)(ln & 3) * 16u, 0, 0);
      }
    };
    auto write_dy = [&](unsigned char* slot, const u32x4 (&d)[ND]) {
      const int ln = opaque_lane();
#pragma unroll
      for (int tt = 0; tt < ND; ++tt) {
        const int i = 16 * tt + (ln >> 2), x = i - 1;
        const bool ok = x >= 0 && x < W;
        if (i < 16 * NKS + 2)
          *reinterpret_cast<uint4*>(slot + i * 64 + (ln & 3) * 16) =
              ok ? make_uint4(d[tt][0], d[tt][1], d[tt][2], d[tt][3]) : make_uint4(0u, 0u, 0u, 0u);
      }
    };

    // ---- prologue: dy rows j0-1, j0, j0+1 -> ring slots A, B, C; a2 row j0 -> a2buf
    unsigned char *sA = ring, *sB = ring + SLOT, *sC = ring + 2 * SLOT;
    u32x4 zr[NZ], dr[ND];
    load_dy(j0 - 1, dr);
    load_z(j0, zr);
    write_dy(sA, dr);
    load_dy(j0, dr);
    write_a2(zr);
    write_dy(sB, dr);
    load_dy(j0 + 1, dr);
    write_dy(sC, dr);

    for (int r = j0; r < j1; ++r) {
      // the next row's operands are requested before this row is multiplied
      load_z(r + 1, zr);
      load_dy(r + 2, dr);
      __builtin_amdgcn_sched_barrier(0);
      // ---- 9 NKS MFMAs: tap (ky, kx) reads dy row r + 1 - ky (slots C, B, A) from pixel index 16 ks + 2 - kx
      {
        const int ln = opaque_lane();
        const int lo_ = (8 * (ln >> 5) + ((ln & 15) >> 2)) * 64 + 32 * ((ln >> 4) & 1) + 8 * (ln & 3);
        const unsigned char* pa = a2buf + lo_;
        const unsigned char* p0 = sC + lo_;
        const unsigned char* p1 = sB + lo_;
        const unsigned char* p2 = sA + lo_;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          const bf16x8 fb = wr_frag(pa + ks * 1024);
          bf16x8 fa[2];
          fa[0] = wr_frag(p0 + ks * 1024 + 2 * 64);
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            const int nt = t + 1;
            if (nt < 9) {
              const unsigned char* pn = (nt / 3 == 0) ? p0 : (nt / 3 == 1) ? p1 : p2;
              fa[nt & 1] = wr_frag(pn + ks * 1024 + (2 - nt % 3) * 64);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[t & 1], fb, acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      // ---- stage row r+1: a2 -> a2buf (this row's fragments are consumed), dy row r+2 -> the slot of row r-1; rotate
      write_a2(zr);
      write_dy(sA, dr);
      unsigned char* t_ = sA;
      sA = sB;
      sB = sC;
      sC = t_;
    }
  }

  // ---- the two row streams of the workgroup meet in LDS; stream 0 writes the workgroup's partial
  __syncthreads();                                  // all ring / a2 memory is dead
  float* red = reinterpret_cast<float*>(lds) + (size_t)q * (9 * 16 * 64);
  if (stream == 1) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(t * 16 + r) * 64 + lane] = acc[t][r];
  }
  __syncthreads();
  if (stream == 0) {
    float* out = wpart + (long long)blockIdx.x * WR_MN;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = (r & 3) + 8 * (r >> 2) + 4 * h;
        out[(co * 9 + t) * 128 + 32 * q + l31] = acc[t][r] + red[(t * 16 + r) * 64 + lane];
      }
  }
}

struct WrwRowsPlan {
  int nimg, rc, nchunk, grid, nks;
};

inline WrwRowsPlan wrw_rows_plan(long long S, int H, int W) {
  WrwRowsPlan p;
  p.nimg = (int)(S / ((long long)H * W));
  p.nks = W <= 32 ? 2 : 4;
  // two row streams per workgroup, one workgroup per CU (147 KB of LDS); chunks sized for 512 streams.  The launch is capped
  // at 160 workgroups: the kernel lives on the side stream, where a full-chip grid of CU-filling workgroups keeps the
  // critical chain's kernels off the CUs (bench.py: 12.93-12.97 ms/step at 256 workgroups, 12.78-12.84 at 128-160, 12.84-12.90
  // with the kernel-row form although this kernel alone is twice as fast)
  const int gcap = 160;
  long long rc = ((long long)H * p.nimg + 511) / 512;
  if (rc < 1) rc = 1;
  if (rc > H) rc = H;
  p.rc = (int)rc;
  p.nchunk = (H + p.rc - 1) / p.rc;
  const int nunit = p.nimg * p.nchunk;
  p.grid = (nunit + 1) / 2;
  if (p.grid > gcap) p.grid = gcap;
  if (p.grid > 256) p.grid = 256;
  return p;
}

}  // namespace

bool mcl_conv3x3_wrw_rows_applicable(long long S, int H, int W) {
  return W >= 17 && W <= 64 && S % ((long long)H * W) == 0;
}

long long mcl_conv3x3_wrw_rows_workspace_floats(long long S, int H, int W) {
  return (long long)wrw_rows_plan(S, H, W).grid * WR_MN;
}

int mcl_launch_conv3x3_wrw_rows(const void* dy, long long lddy, const void* z, long long S, int H, int W, const float* gamma,
                                const float* beta, const float* mean, const float* rstd, float* workspace, float* dW,
                                int accumulate_w, hipStream_t st) {
  const WrwRowsPlan p = wrw_rows_plan(S, H, W);
  const size_t ring_bytes = (size_t)WR_NWAVE * ((size_t)16 * p.nks * 64 + 3 * (size_t)(16 * p.nks + 2) * 64);
  const size_t lds_bytes = ring_bytes > (size_t)WR_RED_BYTES ? ring_bytes : (size_t)WR_RED_BYTES;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)conv3x3_wrw_rows_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)conv3x3_wrw_rows_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  if (p.nks == 2)
    hipLaunchKernelGGL((conv3x3_wrw_rows_kernel<2>), dim3(p.grid), dim3(64 * WR_NWAVE), lds_bytes, st, (const bf16_t*)dy, lddy,
                       (const bf16_t*)z, p.nimg, H, W, gamma, beta, mean, rstd, p.rc, p.nchunk, workspace);
  else
    hipLaunchKernelGGL((conv3x3_wrw_rows_kernel<4>), dim3(p.grid), dim3(64 * WR_NWAVE), lds_bytes, st, (const bf16_t*)dy, lddy,
                       (const bf16_t*)z, p.nimg, H, W, gamma, beta, mean, rstd, p.rc, p.nchunk, workspace);
  mcl_launch_wrw_merge(workspace, p.grid, (long long)WR_MN, dW, accumulate_w, st);
  return MCL_OK;
}
