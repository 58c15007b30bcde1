// Backward of a DenseNet layer's head  x -> norm1 -> relu1 -> conv1(1x1) -> z   with respect to x, fused:
//
//     da = dz W1                      (1x1 backward-data: M = S pixels, N = C_in, K = 128)
//     g  = da * [x*sc + sh > 0]       (ReLU mask recomputed from the layer input)
//     dgamma = sum_s g*xhat ,  dbeta = sum_s g ,  xhat = (x - mean)*rstd          (train-mode BatchNorm backward)
//     dx += gamma*rstd*(g - mean_s(g) - xhat*mean_s(g*xhat))                      (accumulated into the block's gradient buffer)
//
// The stock sequence materialises da (S x C_in), re-reads it together with x for the two reductions and reads it a
// third time with x and the gradient buffer for dx: 7 passes over the O(L^2) S x C_in data of a dense block.  Here
// da never exists: a "reduce" launch and a "dx" launch both recompute their 128 x 128 tile of dz W1 on the matrix
// cores (K = 128: 2 x 32 MFMAs per wave, ~1 % of the step's FLOPs) and touch HBM only for x (once each) and the
// gradient buffer (read-modify-write): 4 passes, and no MIOpen backward-data call with its zero-fill helper.
//
// Tile = 128 pixels x 128 input channels, 4 waves (2 x 2, each 64 x 64 as 2 x 2 v_mfma_f32_32x32x16_bf16), two
// workgroups per CU.  dz tile and the W1 column block are staged once (K = 128 needs no loop): dz rows XOR-swizzled
// for ds_read_b128 fragments, W1 rows read with ds_read_b64_tr_b16 (k-strided operand).  After the MFMAs the same LDS
// is reused to stage the x tile and (dx launch) the gradient-buffer tile with full 16-byte row chunks, because the
// accumulator layout (lane = channel) would otherwise touch HBM in 64-byte fragments; the per-channel BatchNorm
// constants live in registers (a lane keeps its channel for the whole tile).
#include "common.h"
#include <stdlib.h>

namespace {

typedef unsigned short bf16_t;
typedef short v4s __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

constexpr int TN = 128, TK = 128;   // input channels, bottleneck channels per tile (pixels per tile: template TMv)

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  const f32x2 v = {f, 0.0f};
  return (bf16_t)(__builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t)) & 0xFFFFu);
}

// MODE 0: reduce (partials of sum g, sum g*xhat)      MODE 1: dx accumulate
// MODE 2: ONE pass -- gbuf += gamma*rstd*g (the data-dependent term of dx) AND the partials of the two sums.  The layer's own
// two mean terms, per-channel constants times (1, xhat), are not known yet: they are applied ONE LAYER LATE, by the next
// layer's pass over the same channels (``coef`` = the previous pass's finalized terms, NULL for the first layer of a block's
// backward), and by bn1_fix_kernel for the 32 channels the next layer does not read
// A workgroup keeps ONE column tile (its W1 block and per-channel constants are loaded once) and walks row tiles.
// All global loads of a row tile (dz, x and -- dx launch -- the gradient-buffer chunks) are issued together at the
// top, so the x / gradient latency hides under the dz staging and the MFMAs; LDS holds W1 (32 KB) + one 32 KB tile
// that is first dz, then x, then (dx launch) the bf16 deltas: 64 KB, two workgroups per CU.
template <int MODE, int TMv>
__global__ __launch_bounds__(256, (TMv == 64 && MODE != 2 ? 3 : 2)) void bn1_bwd_kernel(const bf16_t* __restrict__ dz, const bf16_t* __restrict__ W1,
                                                         int K /* channels of this launch */, int ldw /* row length of W1 (>= K: a channel window) */,
                                                         const bf16_t* __restrict__ x, long long ldx, long long S,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta,
                                                         const float* __restrict__ mean,
                                                         const float* __restrict__ rstd,
                                                         const float* __restrict__ coef /* MODE 1: [C][2] */,
                                                         bf16_t* gbuf, long long ldg,
                                                         float2* __restrict__ partial /* MODE 0: [C][nrt] */, int nrt) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[TK * 256 + TMv * 256 + 2048];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  const int n0 = blockIdx.y * TN;
  unsigned char* wt = lds;                  // [128 k][128 n] bf16, 256-byte rows, chunk ^ f(k) for transposing reads
  unsigned char* dzt = lds + TK * 256;      // [128 rows][128 k] bf16, chunk ^ (row & 15); later the x tile / deltas
  bf16_t* xt = reinterpret_cast<bf16_t*>(dzt);                   // [128][128] bf16, plain rows
  float* red = reinterpret_cast<float*>(lds + TK * 256 + TMv * 256);     // [2 wm][128] float2

  const int cc = tid & 15, rr = tid >> 4;
  const bool cok = n0 + cc * 8 < K;
  // ---- the first row tile's global loads go out BEFORE the once-per-workgroup staging below (otherwise they would wait
  // behind its memory round trip: on the small maps a workgroup multiplies one to four tiles)
  constexpr int NL = TMv / 16, RI = TMv / 64;   // 16-byte chunks per thread and tile; 32-row blocks per wave
  uint4 dzr[NL], xr[NL], gr[NL];
  auto issue_loads = [&](int rt) {
    const long long row0 = (long long)rt * TMv;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const long long rg = row0 + rr + 16 * i;
      const bool ok = rg < S;
      dzr[i] = ok ? *reinterpret_cast<const uint4*>(dz + rg * TK + cc * 8) : make_uint4(0u, 0u, 0u, 0u);
      xr[i] = (ok && cok) ? *reinterpret_cast<const uint4*>(x + rg * ldx + n0 + cc * 8) : make_uint4(0u, 0u, 0u, 0u);
      if (MODE >= 1)
        gr[i] = (ok && cok) ? *reinterpret_cast<const uint4*>(gbuf + rg * ldg + n0 + cc * 8) : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  if ((int)blockIdx.x < nrt) issue_loads(blockIdx.x);
  // ---- once per workgroup: W1[:, n0:n0+128] and the per-channel constants of this lane's two channels
  {
    uint4 wv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = rr + 16 * i;
      wv[i] = cok ? *reinterpret_cast<const uint4*>(W1 + (long long)k * ldw + n0 + cc * 8) : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = rr + 16 * i;
      // swizzle for the transposing read: 4 consecutive k rows (k & 3) must land in 4 different 64-byte bank
      // windows -> XOR the chunk's bits 2-3 with (k & 3); bits 0-1 with ((k >> 2) & 3)
      const int f = ((k & 3) << 2) | ((k >> 2) & 3);
      *reinterpret_cast<uint4*>(wt + k * 256 + ((cc ^ f) << 4)) = wv[i];
    }
  }
  float mu[2], rs[2], sc[2], sh[2], c1[2], c2[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = n0 + wn * 64 + j * 32 + l31;
    const bool cvalid = c < K;
    mu[j] = cvalid ? mean[c] : 0.0f;
    rs[j] = cvalid ? rstd[c] : 0.0f;
    sc[j] = cvalid ? gamma[c] * rs[j] : 0.0f;
    sh[j] = cvalid ? fmaf(-mu[j], sc[j], beta[c]) : 0.0f;
    c1[j] = (MODE >= 1 && coef != nullptr && cvalid) ? coef[2 * c] : 0.0f;
    c2[j] = (MODE >= 1 && coef != nullptr && cvalid) ? coef[2 * c + 1] : 0.0f;
  }
  const int q = (lane & 15) >> 2, jj = lane & 3;
  const int g2 = 2 * ((lane >> 4) & 1) + (jj >> 1);

  for (int rt = blockIdx.x; rt < nrt; rt += gridDim.x) {
    const long long row0 = (long long)rt * TMv;
    // ---- all global loads of this row tile, issued together (the first tile's are in flight already)
    if (rt != (int)blockIdx.x) issue_loads(rt);
    __syncthreads();   // the previous row tile is done with the shared tile
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int r = rr + 16 * i;
      *reinterpret_cast<uint4*>(dzt + r * 256 + ((cc ^ (r & 15)) << 4)) = dzr[i];
    }
    __syncthreads();

    // ---- da tile = dz_tile (128 x 128k) . W1_tile (128k x 128n): wave (wm, wn) owns rows wm*64.., channels wn*64..
    f32x16 acc[RI][2];
#pragma unroll
    for (int i = 0; i < RI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      bf16x8 fa[RI], fb[2];
#pragma unroll
      for (int i = 0; i < RI; ++i) {
        const int r = wm * (TMv / 2) + i * 32 + l31;
        fa[i] = *reinterpret_cast<const bf16x8*>(dzt + r * 256 + (((2 * ks + h) ^ (r & 15)) << 4));
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        // B[k][n]: lane supplies k row 16*ks + 8*h + q (+4), 4 channels at n = wn*64 + j*32 + 16*((lane>>4)&1) + 4*jj
        const int chunk = (wn * 64 + j * 32) / 8 + g2;
        const int k_lo = 16 * ks + 8 * h + q, k_hi = k_lo + 4;
        const int f_lo = ((k_lo & 3) << 2) | ((k_lo >> 2) & 3), f_hi = ((k_hi & 3) << 2) | ((k_hi >> 2) & 3);
        const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (v4s __attribute__((address_space(3)))*)(wt + k_lo * 256 + ((chunk ^ f_lo) << 4) + (jj & 1) * 8));
        const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (v4s __attribute__((address_space(3)))*)(wt + k_hi * 256 + ((chunk ^ f_hi) << 4) + (jj & 1) * 8));
        fb[j][0] = lo[0]; fb[j][1] = lo[1]; fb[j][2] = lo[2]; fb[j][3] = lo[3];
        fb[j][4] = hi[0]; fb[j][5] = hi[1]; fb[j][6] = hi[2]; fb[j][7] = hi[3];
      }
#pragma unroll
      for (int i = 0; i < RI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();   // dz tile dead: the same LDS now takes the x tile (plain 256-byte rows)
#pragma unroll
    for (int i = 0; i < NL; ++i) *reinterpret_cast<uint4*>(xt + (rr + 16 * i) * TN + cc * 8) = xr[i];
    __syncthreads();

    // ---- epilogue: acc[i][j][r] is da at row wm*64 + i*32 + (r&3) + 8*(r>>2) + 4*h, channel wn*64 + j*32 + l31
    const long long nvalid = min((long long)TMv, S - row0);
    float s1[2], s2[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int cl = wn * 64 + j * 32 + l31;
      s1[j] = s2[j] = 0.0f;
      // dx = sc*(g - c1 - xhat*c2) = sc*g + (ka*x + kb);   sum g*xhat = rs*(sum g*x - mu*sum g)
      // MODE 1: c = (mean g, mean g*xhat) of THIS layer.  MODE 2: c = gamma*rstd*(mean g, mean g*xhat) of the PREVIOUS pass
      const float ka = MODE == 2 ? -c2[j] * rs[j] : -sc[j] * c2[j] * rs[j];
      const float kb = fmaf(-ka, mu[j], MODE == 2 ? -c1[j] : -sc[j] * c1[j]);
#pragma unroll
      for (int i = 0; i < RI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * (TMv / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const float xv = bf2f(xt[row * TN + cl]);
          const float gi = fmaf(xv, sc[j], sh[j]) > 0.0f ? acc[i][j][r] : 0.0f;
          if (MODE != 1) {
            if (row < nvalid) {
              s1[j] += gi;
              s2[j] = fmaf(gi, xv, s2[j]);
            }
          }
          // the x value of this element is dead: its slot takes the bf16 delta for the read-modify-write below
          if (MODE >= 1) xt[row * TN + cl] = f2bf(fmaf(sc[j], gi, fmaf(ka, xv, kb)));
        }
      if (MODE != 1) s2[j] = rs[j] * fmaf(-mu[j], s1[j], s2[j]);
    }
    if (MODE != 1) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        s1[j] += __shfl_xor(s1[j], 32, 64);
        s2[j] += __shfl_xor(s2[j], 32, 64);
        if (h == 0) {
          const int cl = wn * 64 + j * 32 + l31;
          red[(wm * 128 + cl) * 2] = s1[j];
          red[(wm * 128 + cl) * 2 + 1] = s2[j];
        }
      }
      __syncthreads();
      if (tid < 128 && n0 + tid < K)
        partial[(long long)(n0 + tid) * nrt + rt] = make_float2(red[tid * 2] + red[(128 + tid) * 2],
                                                                red[tid * 2 + 1] + red[(128 + tid) * 2 + 1]);
    }
    if (MODE >= 1) {
      if (MODE == 1) __syncthreads();
      if (cok) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
          const int r = rr + 16 * i;
          const long long rg = row0 + r;
          if (rg < S) {
            const uint4 dv = *reinterpret_cast<const uint4*>(xt + r * TN + cc * 8);
            const unsigned gw[4] = {gr[i].x, gr[i].y, gr[i].z, gr[i].w}, dw[4] = {dv.x, dv.y, dv.z, dv.w};
            unsigned o[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const float lo = __uint_as_float(gw[u] << 16) + __uint_as_float(dw[u] << 16);
              const float hi = __uint_as_float(gw[u] & 0xFFFF0000u) + __uint_as_float(dw[u] & 0xFFFF0000u);
              const f32x2 pv = {lo, hi};
              o[u] = __builtin_bit_cast(unsigned, __builtin_convertvector(pv, bf16x2_t));
            }
            *reinterpret_cast<uint4*>(gbuf + rg * ldg + n0 + cc * 8) = make_uint4(o[0], o[1], o[2], o[3]);
          }
        }
      }
    }
  }
}

// ---- the dx pass of TWO consecutive layers (A = l: W1A rows ldwA long; B = l - 1: W1B rows K long) over their common K input
// channels: 64-pixel row tiles, both 128 x 128 weight blocks resident (2 x 32 KB), ONE shared 16 KB tile that is dzA, then dzB,
// then x / the bf16 deltas: 80 KB, two workgroups per CU.  Same fragments, swizzles and rounding points as bn1_bwd_kernel<1, 64>;
// the two layers' deltas are added in fp32 and rounded to bf16 ONCE (the sequential form rounds each).
constexpr size_t PAIR_LDS = 2 * TK * 256 + 64 * 256;
__global__ __launch_bounds__(256, 2) void bn1_dx_pair_kernel(const bf16_t* __restrict__ dzA, const bf16_t* __restrict__ W1A, int ldwA,
                                                             const float* __restrict__ gammaA, const float* __restrict__ betaA,
                                                             const float* __restrict__ coefA, const bf16_t* __restrict__ dzB,
                                                             const bf16_t* __restrict__ W1B, const float* __restrict__ gammaB,
                                                             const float* __restrict__ betaB, const float* __restrict__ coefB,
                                                             int K, const bf16_t* __restrict__ x, long long ldx, long long S,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             bf16_t* gbuf, long long ldg, int nrt) {
  constexpr int TMv = 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  const int n0 = blockIdx.y * TN;
  unsigned char* wtA = lds;
  unsigned char* wtB = lds + TK * 256;
  unsigned char* dzt = lds + 2 * TK * 256;
  bf16_t* xt = reinterpret_cast<bf16_t*>(dzt);
  const int cc = tid & 15, rr = tid >> 4;
  const bool cok = n0 + cc * 8 < K;
  constexpr int NL = TMv / 16;
  uint4 dar[NL], dbr[NL], xr[NL], gr[NL];
  auto issue_loads = [&](int rt) {
    const long long row0 = (long long)rt * TMv;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const long long rg = row0 + rr + 16 * i;
      const bool ok = rg < S;
      dar[i] = ok ? *reinterpret_cast<const uint4*>(dzA + rg * TK + cc * 8) : make_uint4(0u, 0u, 0u, 0u);
      dbr[i] = ok ? *reinterpret_cast<const uint4*>(dzB + rg * TK + cc * 8) : make_uint4(0u, 0u, 0u, 0u);
      xr[i] = (ok && cok) ? *reinterpret_cast<const uint4*>(x + rg * ldx + n0 + cc * 8) : make_uint4(0u, 0u, 0u, 0u);
      gr[i] = (ok && cok) ? *reinterpret_cast<const uint4*>(gbuf + rg * ldg + n0 + cc * 8) : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  if ((int)blockIdx.x < nrt) issue_loads(blockIdx.x);
  {
    uint4 wa[8], wb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = rr + 16 * i;
      wa[i] = cok ? *reinterpret_cast<const uint4*>(W1A + (long long)k * ldwA + n0 + cc * 8) : make_uint4(0u, 0u, 0u, 0u);
      wb[i] = cok ? *reinterpret_cast<const uint4*>(W1B + (long long)k * K + n0 + cc * 8) : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = rr + 16 * i;
      const int f = ((k & 3) << 2) | ((k >> 2) & 3);
      *reinterpret_cast<uint4*>(wtA + k * 256 + ((cc ^ f) << 4)) = wa[i];
      *reinterpret_cast<uint4*>(wtB + k * 256 + ((cc ^ f) << 4)) = wb[i];
    }
  }
  // per-channel constants of this lane's two channels: the statistics are the channels' (shared), the affine / mean terms per layer
  float scA[2], shA[2], scB[2], shB[2], ka[2], kb[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = n0 + wn * 64 + j * 32 + l31;
    const bool cv = c < K;
    const float mu = cv ? mean[c] : 0.0f, rs = cv ? rstd[c] : 0.0f;
    scA[j] = cv ? gammaA[c] * rs : 0.0f;
    shA[j] = cv ? fmaf(-mu, scA[j], betaA[c]) : 0.0f;
    scB[j] = cv ? gammaB[c] * rs : 0.0f;
    shB[j] = cv ? fmaf(-mu, scB[j], betaB[c]) : 0.0f;
    const float c1A = cv ? coefA[2 * c] : 0.0f, c2A = cv ? coefA[2 * c + 1] : 0.0f;
    const float c1B = cv ? coefB[2 * c] : 0.0f, c2B = cv ? coefB[2 * c + 1] : 0.0f;
    // dx_L = sc_L*(g_L - c1_L - xhat*c2_L) = sc_L*g_L + (ka_L*x + kb_L), as bn1_bwd_kernel<1>; the two affine parts added
    const float kaA = -scA[j] * c2A * rs, kaB = -scB[j] * c2B * rs;
    const float kbA = fmaf(-kaA, mu, -scA[j] * c1A), kbB = fmaf(-kaB, mu, -scB[j] * c1B);
    ka[j] = kaA + kaB;
    kb[j] = kbA + kbB;
  }
  const int q = (lane & 15) >> 2, jj = lane & 3;
  const int g2 = 2 * ((lane >> 4) & 1) + (jj >> 1);
  auto product = [&](const unsigned char* wt, f32x16 (&acc)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int r = wm * (TMv / 2) + l31;
      const bf16x8 fa = *reinterpret_cast<const bf16x8*>(dzt + r * 256 + (((2 * ks + h) ^ (r & 15)) << 4));
      bf16x8 fb[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int chunk = (wn * 64 + j * 32) / 8 + g2;
        const int k_lo = 16 * ks + 8 * h + q, k_hi = k_lo + 4;
        const int f_lo = ((k_lo & 3) << 2) | ((k_lo >> 2) & 3), f_hi = ((k_hi & 3) << 2) | ((k_hi >> 2) & 3);
        const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (v4s __attribute__((address_space(3)))*)(wt + k_lo * 256 + ((chunk ^ f_lo) << 4) + (jj & 1) * 8));
        const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (v4s __attribute__((address_space(3)))*)(wt + k_hi * 256 + ((chunk ^ f_hi) << 4) + (jj & 1) * 8));
        fb[j][0] = lo[0]; fb[j][1] = lo[1]; fb[j][2] = lo[2]; fb[j][3] = lo[3];
        fb[j][4] = hi[0]; fb[j][5] = hi[1]; fb[j][6] = hi[2]; fb[j][7] = hi[3];
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb[j], acc[j], 0, 0, 0);
    }
  };
  for (int rt = blockIdx.x; rt < nrt; rt += gridDim.x) {
    const long long row0 = (long long)rt * TMv;
    if (rt != (int)blockIdx.x) issue_loads(rt);
    __syncthreads();   // the previous row tile is done with the shared tile (and, first trip, the weights are staged below)
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int r = rr + 16 * i;
      *reinterpret_cast<uint4*>(dzt + r * 256 + ((cc ^ (r & 15)) << 4)) = dar[i];
    }
    __syncthreads();
    f32x16 accA[2], accB[2];
    product(wtA, accA);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int r = rr + 16 * i;
      *reinterpret_cast<uint4*>(dzt + r * 256 + ((cc ^ (r & 15)) << 4)) = dbr[i];
    }
    __syncthreads();
    product(wtB, accB);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NL; ++i) *reinterpret_cast<uint4*>(xt + (rr + 16 * i) * TN + cc * 8) = xr[i];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int cl = wn * 64 + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * (TMv / 2) + (r & 3) + 8 * (r >> 2) + 4 * h;
        const float xv = bf2f(xt[row * TN + cl]);
        const float gA = fmaf(xv, scA[j], shA[j]) > 0.0f ? accA[j][r] : 0.0f;
        const float gB = fmaf(xv, scB[j], shB[j]) > 0.0f ? accB[j][r] : 0.0f;
        xt[row * TN + cl] = f2bf(fmaf(scA[j], gA, fmaf(scB[j], gB, fmaf(ka[j], xv, kb[j]))));
      }
    }
    __syncthreads();
    if (cok) {
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const int r = rr + 16 * i;
        const long long rg = row0 + r;
        if (rg < S) {
          const uint4 dv = *reinterpret_cast<const uint4*>(xt + r * TN + cc * 8);
          const unsigned gw[4] = {gr[i].x, gr[i].y, gr[i].z, gr[i].w}, dw[4] = {dv.x, dv.y, dv.z, dv.w};
          unsigned o[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float lo = __uint_as_float(gw[u] << 16) + __uint_as_float(dw[u] << 16);
            const float hi = __uint_as_float(gw[u] & 0xFFFF0000u) + __uint_as_float(dw[u] & 0xFFFF0000u);
            const f32x2 pv = {lo, hi};
            o[u] = __builtin_bit_cast(unsigned, __builtin_convertvector(pv, bf16x2_t));
          }
          *reinterpret_cast<uint4*>(gbuf + rg * ldg + n0 + cc * 8) = make_uint4(o[0], o[1], o[2], o[3]);
        }
      }
    }
  }
}

// one WORKGROUP per channel: dbeta = sum g, dgamma = sum g*xhat (fixed order, double), coef = the two means for the
// dx pass.  On the critical path between the reduce and dx launches of every layer.
__global__ __launch_bounds__(256) void bn1_bwd_finalize_kernel(const float2* __restrict__ partial, int nrt, int C,
                                                               long long S, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, float* __restrict__ coef,
                                                               int accumulate_params,
                                                               float* __restrict__ kacc = nullptr /* [C][2], single-pass form */,
                                                               const float* __restrict__ gamma = nullptr,
                                                               const float* __restrict__ rstd = nullptr) {
  __shared__ double red[2][4];
  const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float2* p = partial + (long long)c * nrt;
  double a = 0.0, b = 0.0;
  // batches of eight independent loads (clamped index, surplus zeroed after the load), added in ascending order: one
  // load per loop trip would be one memory round trip per trip
  constexpr int U = 8;
  for (int t0 = threadIdx.x; t0 < nrt; t0 += 256 * U) {
    float2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = p[min(t0 + 256 * u, nrt - 1)];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (t0 + 256 * u >= nrt) v[u] = make_float2(0.0f, 0.0f);
      a += (double)v[u].x;
      b += (double)v[u].y;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_xor(a, o, 64);
    b += __shfl_xor(b, o, 64);
  }
  if (lane == 0) {
    red[0][wave] = a;
    red[1][wave] = b;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  a = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
  b = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  if (accumulate_params) {
    dbeta[c] += (float)a;
    dgamma[c] += (float)b;
  } else {
    dbeta[c] = (float)a;
    dgamma[c] = (float)b;
  }
  if (kacc != nullptr) {
    // single-pass form: the layer's mean terms gamma*rstd*(mean g, mean g*xhat), for the next pass / bn1_fix_kernel
    const double scv = (double)gamma[c] * (double)rstd[c];
    kacc[2 * c] = (float)(scv * a / (double)S);
    kacc[2 * c + 1] = (float)(scv * b / (double)S);
    return;
  }
  coef[2 * c] = (float)(a / (double)S);
  coef[2 * c + 1] = (float)(b / (double)S);
}

// Single-pass form: the mean terms of the LAST pass over channels [c0, c0 + nc) of a dense block's concat buffer, for the
// channels no later pass covers (the 32 output channels of layer k once layer k+1's pass is done; the block input after
// layer 0's):     gbuf[s][c] -= K1[c] + K2[c]*xhat[s][c],   xhat = (x - mean)*rstd     (K from bn1_bwd_finalize_kernel)
// Elementwise over S x nc bf16, 16-byte chunks, both tensors strided (channel slices of the block's buffers).
__global__ __launch_bounds__(256) void bn1_fix_kernel(const bf16_t* __restrict__ x, long long ldx, bf16_t* __restrict__ gbuf,
                                                      long long ldg, long long S, int c0, int nc,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      const float* __restrict__ kacc) {
  const int cpr = nc >> 3;                                   // chunks per row
  const long long n = S * cpr;
  // a thread keeps its channel chunk for the whole launch when the grid stride is a multiple of the chunks per row (always
  // for the 32-channel ranges, cpr = 4): its 16 constants are computed once, after the first chunk's loads have gone out
  const long long i0 = (long long)blockIdx.x * 256 + threadIdx.x, istride = (long long)gridDim.x * 256;
  const bool fixed = (istride % cpr) == 0;
  uint4 xv = make_uint4(0u, 0u, 0u, 0u), gv = xv;
  if (i0 < n) {
    const long long row = i0 / cpr;
    const int c = c0 + (int)(i0 - row * cpr) * 8;
    xv = *reinterpret_cast<const uint4*>(x + row * ldx + c);
    gv = *reinterpret_cast<const uint4*>(gbuf + row * ldg + c);
  }
  float kav[8], kbv[8];
  auto consts = [&](int c) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float k1 = kacc[2 * (c + e)], k2 = kacc[2 * (c + e) + 1];
      kav[e] = k2 * rstd[c + e];
      kbv[e] = fmaf(-kav[e], mean[c + e], k1);
    }
  };
  if (i0 < n) consts(c0 + (int)(i0 % cpr) * 8);
  for (long long i = i0; i < n; i += istride) {
    const long long row = i / cpr;
    const int c = c0 + (int)(i - row * cpr) * 8;
    if (i != i0) {
      xv = *reinterpret_cast<const uint4*>(x + row * ldx + c);
      gv = *reinterpret_cast<const uint4*>(gbuf + row * ldg + c);
      if (!fixed) consts(c);
    }
    const unsigned xw[4] = {xv.x, xv.y, xv.z, xv.w};
    unsigned gw[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float o[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float ka = kav[2 * u + e], kb = kbv[2 * u + e];
        const float xf = e ? __uint_as_float(xw[u] & 0xFFFF0000u) : __uint_as_float(xw[u] << 16);
        const float gf = e ? __uint_as_float(gw[u] & 0xFFFF0000u) : __uint_as_float(gw[u] << 16);
        o[e] = gf - fmaf(ka, xf, kb);
      }
      const f32x2 pv = {o[0], o[1]};
      gw[u] = __builtin_bit_cast(unsigned, __builtin_convertvector(pv, bf16x2_t));
    }
    *reinterpret_cast<uint4*>(gbuf + row * ldg + c) = make_uint4(gw[0], gw[1], gw[2], gw[3]);
  }
}

// =====================================================================================================================
// Backward of the layer's tail  z -> norm2 -> relu2 -> conv2 (3x3, pad 1, 128 -> 32)  with respect to z:
//
//     da2[p][ci] = sum_{ky,kx,co} dy[p + (1-ky)*W + (1-kx)][co] * W2[co][ky][kx][ci]        (3x3 backward-data)
//     g2 = da2 * [z*sc2 + sh2 > 0]  ->  stored (bf16) ;  per-tile partials of sum g2, sum g2*zhat
//     (finalize: dgamma2, dbeta2, the two means)   dz = gamma2*rstd2*(g2 - mean(g2) - zhat*mean(g2*zhat))
//
// Implicit GEMM M = S pixels, N = 128, K = 9*32.  Same skeleton as the forward kernel (csrc/dense_conv.hip): a
// workgroup owns 128 consecutive pixels and stages the pixel range [p0 - W - 1, p0 + 128 + W + 1) of dy (only 64 B
// per pixel) into LDS, so the nine taps are row offsets and taps that leave the image select an all-zero row.  N
// is split over the four waves (32 channels each, whose 18 weight fragments stay in registers for every tile of
// the persistent workgroup), so no cross-wave reduction is needed.  dy is read straight from the block's gradient
// buffer (row stride lddy: no contiguous copy) and MIOpen's backward-data call, its zero-fill helper and the separate
// BatchNorm reduce pass disappear; the ReLU mask and the BatchNorm sums come out of the accumulators.
constexpr int C3I = 128, C3O = 32, T3B = 128;

__global__ __launch_bounds__(256, 1) void conv3x3_bwd_kernel(const bf16_t* __restrict__ dy, long long lddy, long long S,
                                                             int H, int W, const bf16_t* __restrict__ W2,
                                                             const bf16_t* __restrict__ z,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, bf16_t* __restrict__ g2,
                                                             float2* __restrict__ partial, int ntile,
                                                             // single-pass BatchNorm-1 backward: the mean
                                                             // terms of the PREVIOUS pass for these 32 channels are subtracted
                                                             // while dy is staged (xfix = the layer's output channels in the
                                                             // concat buffer), and the corrected dy of the tile's own pixels
                                                             // goes to dyc (S x 32) for the weight-gradient kernel: no
                                                             // separate bn1_fix launch on the critical chain.  NULL: plain dy.
                                                             const bf16_t* __restrict__ xfix, long long ldxf,
                                                             const float* __restrict__ fmean,
                                                             const float* __restrict__ frstd,
                                                             const float* __restrict__ fk, bf16_t* __restrict__ dyc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int nrow = T3B + 2 * W + 2;
  bf16_t* zt = reinterpret_cast<bf16_t*>(lds);                      // [128][128] bf16: z tile, then g2 in place
  unsigned char* slab = lds + T3B * 256;                            // [nrow][32] bf16, 64-byte rows, chunk ^ ((row>>2)&3)
  const int zero_off = T3B * 256 + nrow * 64;                       // 64 B of zeros
  const int Si = (int)S;

  // the first tile's z chunks go out before the weight gather below (144 two-byte loads per lane: otherwise the tile's loads
  // wait behind their round trip; on the 14 x 14 / 7 x 7 maps, where this kernel runs, a workgroup multiplies ONE tile)
  uint4 zr[8];
  auto load_z_tile = [&](int tile) {
    const int p0 = tile * T3B;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int p = p0 + (tid >> 4) + 16 * i;
      zr[i] = p < Si ? *reinterpret_cast<const uint4*>(z + (long long)p * C3I + (tid & 15) * 8) : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  if ((int)blockIdx.x < ntile) load_z_tile(blockIdx.x);
  // this wave's weight fragments: B[k = co][n = ci], ci = 32*wave + l31, k-step i -> tap = i >> 1, co = 16*(i&1) + 8h + j
  bf16x8 breg[18];
#pragma unroll
  for (int i = 0; i < 18; ++i) {
    const int tap = i >> 1, co0 = 16 * (i & 1) + 8 * h;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      breg[i][j] = (short)W2[((long long)(co0 + j) * 9 + tap) * C3I + 32 * wave + l31];
  }
  const int c = 32 * wave + l31;
  const float mu = mean[c], rs = rstd[c];
  const float sc = gamma[c] * rs, sh = fmaf(-mu, sc, beta[c]);
  // slab byte offset of this lane's fragment for k-step i in pixel block 0 (block mb adds 32 rows = 2048 B; 32 rows
  // leave (row >> 2) & 3 unchanged): row = l31 + (2-ky)*W + (2-kx), chunk = 2*(i&1) + h
  int abase[18];
#pragma unroll
  for (int i = 0; i < 18; ++i) {
    const int tap = i >> 1, ky = tap / 3, kx = tap % 3;
    const int row = l31 + (2 - ky) * W + (2 - kx);
    abase[i] = T3B * 256 + row * 64 + (((2 * (i & 1) + h) ^ ((row >> 2) & 3)) << 4);
  }
  if (tid < 4) *reinterpret_cast<uint4*>(lds + zero_off + tid * 16) = make_uint4(0u, 0u, 0u, 0u);
  // mean-term constants of this thread's dy chunk (q & 3 == tid & 3 for every q it stages): dy -= fma(ka, x, kb), the
  // arithmetic of bn1_fix_kernel, bit for bit
  float fka[8], fkb[8];
  if (xfix != nullptr) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int cf = (tid & 3) * 8 + e;
      const float k1 = fk[2 * cf], k2 = fk[2 * cf + 1];
      fka[e] = k2 * frstd[cf];
      fkb[e] = fmaf(-fka[e], fmean[cf], k1);
    }
  }

  for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const int p0 = tile * T3B;
    // ---- global loads of the tile: z chunks (8 per thread; the first tile's are in flight), dy slab chunks
    if (tile != (int)blockIdx.x) load_z_tile(tile);
    __syncthreads();   // previous tile done with LDS
    for (int q = tid; q < nrow * 4; q += 256) {
      const int j = q >> 2, ch = q & 3;
      const int p = p0 - (W + 1) + j;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (p >= 0 && p < Si) {
        v = *reinterpret_cast<const uint4*>(dy + (long long)p * lddy + ch * 8);
        if (xfix != nullptr) {
          const uint4 xv = *reinterpret_cast<const uint4*>(xfix + (long long)p * ldxf + ch * 8);
          const unsigned xw[4] = {xv.x, xv.y, xv.z, xv.w};
          unsigned gw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float o_lo = __uint_as_float(gw[u] << 16) - fmaf(fka[2 * u], __uint_as_float(xw[u] << 16), fkb[2 * u]);
            const float o_hi = __uint_as_float(gw[u] & 0xFFFF0000u) -
                               fmaf(fka[2 * u + 1], __uint_as_float(xw[u] & 0xFFFF0000u), fkb[2 * u + 1]);
            const f32x2 pv = {o_lo, o_hi};
            gw[u] = __builtin_bit_cast(unsigned, __builtin_convertvector(pv, bf16x2_t));
          }
          v = make_uint4(gw[0], gw[1], gw[2], gw[3]);
          // the tile's own pixels (slab rows W+1 .. W+128): hand the corrected dy to the weight-gradient kernel
          if (dyc != nullptr && j >= W + 1 && j < W + 1 + T3B) *reinterpret_cast<uint4*>(dyc + (long long)p * C3O + ch * 8) = v;
        }
      }
      *reinterpret_cast<uint4*>(slab + j * 64 + ((ch ^ ((j >> 2) & 3)) << 4)) = v;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
      *reinterpret_cast<uint4*>(zt + ((tid >> 4) + 16 * i) * C3I + (tid & 15) * 8) = zr[i];
    // tap validity of this lane's pixel in each of the 4 pixel blocks
    unsigned vmask[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      const int p = p0 + mb * 32 + l31;
      const int x = p % W, y = (p / W) % H;
      unsigned m = 0;
      if (p < Si) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int yy = y + 1 - ky, xx = x + 1 - kx;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) m |= 1u << (ky * 3 + kx);
          }
      }
      vmask[mb] = m;
    }
    f32x16 acc[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][r] = 0.0f;
    __syncthreads();

#pragma unroll
    for (int i = 0; i < 18; ++i) {
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        const int off = ((vmask[mb] >> (i >> 1)) & 1u) ? abase[i] + mb * 2048 : zero_off;
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(lds + off);
        acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, breg[i], acc[mb], 0, 0, 0);
      }
    }
    // ---- epilogue: acc[mb][r] = da2 at pixel mb*32 + (r&3) + 8*(r>>2) + 4*h, channel c
    const int nvalid = min(T3B, Si - p0);
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int px = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const float zv = bf2f(zt[px * C3I + c]);
        const float gi = fmaf(zv, sc, sh) > 0.0f ? acc[mb][r] : 0.0f;
        const bf16_t gb = f2bf(gi);
        if (px < nvalid) {
          const float gr_ = bf2f(gb);                 // the sums are those of the stored (rounded) g2
          s1 += gr_;
          s2 = fmaf(gr_, (zv - mu) * rs, s2);
        }
        zt[px * C3I + c] = gb;
      }
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (h == 0) partial[(long long)c * ntile + tile] = make_float2(s1, s2);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = (tid >> 4) + 16 * i;
      const int p = p0 + r;
      if (p < Si)
        *reinterpret_cast<uint4*>(g2 + (long long)p * C3I + (tid & 15) * 8) =
            *reinterpret_cast<const uint4*>(zt + r * C3I + (tid & 15) * 8);
    }
  }
}

// =====================================================================================================================
// Row-walking form of the kernel above for the large maps (image width 17..150: the 56 x 56 and 28 x 28 dense blocks), the
// backward-data counterpart of csrc/conv3x3_rows.hip.  The flat-tile kernel loads a tile, synchronises, multiplies, runs a
// 128-instruction-per-lane epilogue through LDS (lane = channel: z and g2 go through an LDS transpose as 2-byte
// elements) and stores -- one tile in flight per workgroup (96 us per 56 x 56 layer inside the step).
//
// Here the work is cut into THIN, independent waves: a wave owns (image, row chunk, 32-column strip, 32-channel quarter of
// the 128 input channels).  Its 18 weight fragments (9 taps x 2 halves of the 32 output channels) stay in registers; dy is
// only 64 B per pixel, so the wave keeps a private 4-row ring of its strip (34 pixels incl. explicit zero halo columns) in
// LDS and every output row is 18 MFMAs on 18 LDS fragment reads -- no masks (rows outside the image are skipped
// wave-uniformly), no workgroup barrier at all.  MFMA roles are swapped (A = weights, B = pixels) so that a lane owns ONE
// pixel and 16 channels: z arrives and g2 leaves as two 16-byte buffer accesses per lane (a v_permlane32_swap converts
// between the memory order and the accumulator order), the ReLU mask and the BatchNorm sums are per-lane arithmetic on
// per-lane running sums, reduced across lanes once per unit by DPP.  Loads run two rows ahead (dy) / one row ahead (z) in
// two alternating register sets; rows past the unit's range read through a zero-size buffer descriptor.
constexpr int BR_NWAVE = 8;                       // waves per workgroup (two per SIMD)
constexpr int BR_SLOT = 34 * 64;                  // one dy row of the strip: 34 pixels x 32 channels bf16
constexpr int BR_RING = 4 * BR_SLOT;              // rows jo-1, jo, jo+1 in use + the one being written
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned br_pack2(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

// sum over the 32 lanes of each half-wave by DPP; the total lands in lanes 16..31 / 48..63
__device__ __forceinline__ float br_half_wave_sum(float x) {
#define MCL_DPP_ADD(ctrl, rmask) x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, rmask, 0xF, false))
  MCL_DPP_ADD(0xB1, 0xF);     // quad_perm [1,0,3,2]
  MCL_DPP_ADD(0x4E, 0xF);     // quad_perm [2,3,0,1]
  MCL_DPP_ADD(0x141, 0xF);    // row_half_mirror
  MCL_DPP_ADD(0x140, 0xF);    // row_mirror
  MCL_DPP_ADD(0x142, 0xA);    // row_bcast15 into rows 1 and 3
#undef MCL_DPP_ADD
  return x;
}

// memory order <-> accumulator order of a lane's 16 channels (8 dwords): an involution (see conv3x3_rows.hip emit_row)
__device__ __forceinline__ void br_swap8(unsigned (&w)[8]) {
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const u32x2 r = __builtin_amdgcn_permlane32_swap(w[4 * g + d], w[4 * g + 2 + d], false, false);
      w[4 * g + d] = r[0];
      w[4 * g + 2 + d] = r[1];
    }
}

// the 18 MFMAs of one output row: taps of kernel row ky read dy row jo + 1 - ky (ring slot (jo + 1 - ky) & 3); fragment of
// tap (ky, kx), half ks: pixel i = l31 + 2 - kx of the slot, 16-byte chunk (2 ks + h) ^ ((i >> 2) & 3) -- the six byte offsets
// foff[kx][ks] inside a slot are lane constants
template <bool K0, bool K2>
__device__ __forceinline__ void br_row_mfma(const unsigned char* __restrict__ ring, int jo, const int (&foff)[3][2],
                                            const bf16x8 (&breg)[18], f32x16& acc) {
  // fragment list of the row: (ky, kx, ks) over the valid kernel rows, two k-steps (one tap) per step; the fragments of
  // step s+1 are requested before the MFMAs of step s issue (LDS latency would otherwise sit between every MFMA pair), and
  // even / odd k-steps go to two accumulators so that consecutive MFMAs do not wait on each other's result
  constexpr int KY0 = K0 ? 0 : 1, KY1 = K2 ? 3 : 2, NT = (KY1 - KY0) * 3;
  f32x16 acc2;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc2[r] = 0.0f;
  bf16x8 b[2][2];
  auto fetch = [&](int t, int buf) {
    const int ky = KY0 + t / 3, kx = t % 3;
    const unsigned char* slot = ring + ((jo + 1 - ky) & 3) * BR_SLOT;
    b[buf][0] = *reinterpret_cast<const bf16x8*>(slot + foff[kx][0]);
    b[buf][1] = *reinterpret_cast<const bf16x8*>(slot + foff[kx][1]);
  };
  fetch(0, 0);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int cur = t & 1, tap = (KY0 + t / 3) * 3 + t % 3;
    if (t + 1 < NT) fetch(t + 1, cur ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(breg[tap * 2], b[cur][0], acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(breg[tap * 2 + 1], b[cur][1], acc2, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
}

__global__ __launch_bounds__(64 * BR_NWAVE, 2) void conv3x3_bwd_rows_kernel(
    const bf16_t* __restrict__ dy, long long lddy, int nimg, int H, int W, const bf16_t* __restrict__ W2,
    const bf16_t* __restrict__ z, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ mean, const float* __restrict__ rstd, bf16_t* __restrict__ g2, float2* __restrict__ partial,
    int nsu, int rc, int nchunk, int nstrip) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const int gw = blockIdx.x * BR_NWAVE + wave;
  const int q = gw & 3;                                   // input-channel quarter: constant per wave (grid * 8 % 4 == 0)
  unsigned char* ring = lds + wave * BR_RING;

  // weight fragments: A[i = ci local][k = co]: lane (l31, h), fragment (tap, ks): W2[co = 16 ks + 8 h + j][tap][32 q + l31]
  bf16x8 breg[18];
#pragma unroll
  for (int f = 0; f < 18; ++f) {
    const int tap = f >> 1, co0 = 16 * (f & 1) + 8 * h;
#pragma unroll
    for (int j = 0; j < 8; ++j) breg[f][j] = (short)W2[((long long)(co0 + j) * 9 + tap) * 128 + 32 * q + l31];
  }
  // BatchNorm (scale, shift) per channel -> LDS; a lane re-reads the pairs of its 16 channels (accumulator order: r -> ci =
  // 32 q + (r & 3) + 8 (r >> 2) + 4 h, i.e. four runs of 4 consecutive channels) in every row epilogue: 32 VGPRs saved
  float2* coefs = reinterpret_cast<float2*>(lds + BR_NWAVE * BR_RING);
  if (tid < 128) {
    const float scv = gamma[tid] * rstd[tid];
    coefs[tid] = make_float2(scv, fmaf(-mean[tid], scv, beta[tid]));
  }
  __syncthreads();
  const unsigned dyrow_bytes = (unsigned)(((long long)(W - 1) * lddy + 32) * 2);
  const unsigned zrow_bytes = (unsigned)W * 256u;
  const unsigned lddy2 = (unsigned)(lddy * 2);
  const int su_stride = (gridDim.x * BR_NWAVE) >> 2;

  for (int su = gw >> 2; su < nsu; su += su_stride) {
    const int strip = su % nstrip, t = su / nstrip;
    const int chunk = t % nchunk, b = t / nchunk;
    const int x0 = strip * 32;
    const int j0 = chunk * rc, j1 = min(H, j0 + rc);
    const int jd1 = min(H, j1 + 1);                              // dy rows [max(j0 - 1, 0), jd1) are needed
    const long long img = (long long)b * H;
    auto opaque_lane = [&]() { int ln = lane; asm volatile("" : "+v"(ln)); return ln; };

    // dy row jr of the strip -> 3 chunks per lane: pixel i = (ln >> 2) + 16 tt, 16-byte chunk ln & 3.  Rows outside
    // [0, jd1) read through a zero-size descriptor (zeros, no memory access); x = -1 reads pixel 0 and is zeroed at the write.
    auto load_dy = [&](int jr, u32x4 (&d)[3]) {
      const int ln = opaque_lane();
      const bool inr = jr >= 0 && jr < jd1;
      const __amdgpu_buffer_rsrc_t row = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<bf16_t*>(dy) + (img + min(max(jr, 0), H - 1)) * W * lddy, 0, inr ? dyrow_bytes : 0u, 0x00020000);
#pragma unroll
      for (int tt = 0; tt < 3; ++tt) {
        const int x = x0 - 1 + (ln >> 2) + 16 * tt;
        d[tt] = __builtin_amdgcn_raw_buffer_load_b128(row, (unsigned)max(x, 0) * lddy2 + (unsigned)(ln & 3) * 16u, 0, 0);
      }
    };
    auto write_dy = [&](int jr, const u32x4 (&d)[3]) {
      const int ln = opaque_lane();
      unsigned char* slot = ring + (jr & 3) * BR_SLOT;
#pragma unroll
      for (int tt = 0; tt < 3; ++tt) {
        const int i = (ln >> 2) + 16 * tt, x = x0 - 1 + i;
        const bool ok = x >= 0 && x < W;
        if (i < 34)
          *reinterpret_cast<uint4*>(slot + i * 64 + ((((ln & 3) ^ ((i >> 2) & 3))) << 4)) =
              ok ? make_uint4(d[tt][0], d[tt][1], d[tt][2], d[tt][3]) : make_uint4(0u, 0u, 0u, 0u);
      }
    };
    // z row jo of this lane's pixel, quarter q, memory order: 16 bytes at 16 h and at 32 + 16 h of the 64-byte quarter row
    auto load_z = [&](int jo, u32x4 (&zr)[2]) {
      const int ln = opaque_lane();
      const __amdgpu_buffer_rsrc_t row = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<bf16_t*>(z) + (img + min(jo, H - 1)) * W * 128, 0, jo < j1 ? zrow_bytes : 0u, 0x00020000);
      const unsigned off = (unsigned)(x0 + (ln & 31)) * 256u + (unsigned)q * 64u + 16u * (unsigned)(ln >> 5);
      zr[0] = __builtin_amdgcn_raw_buffer_load_b128(row, off, 0, 0);
      zr[1] = __builtin_amdgcn_raw_buffer_load_b128(row, off + 32u, 0, 0);
    };

    float s1[16], s2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) s1[r] = s2[r] = 0.0f;
    int foff[3][2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int i = l31 + 2 - kx, sw = (i >> 2) & 3;
      foff[kx][0] = i * 64 + ((h ^ sw) << 4);
      foff[kx][1] = i * 64 + (((2 + h) ^ sw) << 4);
    }

    // ---- prologue: rows j0-1, j0, j0+1 into the ring; row j0+2 and z row j0 in flight
    u32x4 dA[3], dB[3], zA[2], zB[2];
    load_dy(j0 - 1, dA);
    load_dy(j0, dB);
    write_dy(j0 - 1, dA);
    load_dy(j0 + 1, dA);
    write_dy(j0, dB);
    write_dy(j0 + 1, dA);
    load_dy(j0 + 2, dB);
    load_z(j0, zB);

    int jo = j0;
    bool more = true;
    // X = sets loaded during the previous iteration (dy row jo + 2, z row jo); Y = the free sets (dy row jo + 3, z row jo + 1)
#define MCL_BR_STEP(DX, ZX, DY, ZY)                                                                             \
    {                                                                                                           \
      load_dy(jo + 3, DY);                                                                                      \
      load_z(jo + 1, ZY);                                                                                       \
      f32x16 acc;                                                                                               \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[r] = 0.0f;                                             \
      const bool k0 = jo + 1 < H, k2 = jo >= 1;                                                                 \
      if (k0 && k2) br_row_mfma<true, true>(ring, jo, foff, breg, acc);                                         \
      else if (k0) br_row_mfma<true, false>(ring, jo, foff, breg, acc);                                         \
      else if (k2) br_row_mfma<false, true>(ring, jo, foff, breg, acc);                                         \
      else br_row_mfma<false, false>(ring, jo, foff, breg, acc);                                                \
      /* epilogue: mask, round, sums, store */                                                                  \
      {                                                                                                         \
        const int ln = opaque_lane();                                                                           \
        const int px = x0 + (ln & 31);                                                                          \
        unsigned zw[8] = {ZX[0][0], ZX[0][1], ZX[0][2], ZX[0][3], ZX[1][0], ZX[1][1], ZX[1][2], ZX[1][3]};      \
        br_swap8(zw);                                                                                           \
        unsigned gw8[8];                                                                                        \
        const float4* cf = reinterpret_cast<const float4*>(coefs + 32 * q + 4 * (ln >> 5));                    \
        _Pragma("unroll") for (int d = 0; d < 8; ++d) {                                                         \
          const float4 c4 = cf[4 * (d >> 1) + (d & 1)];      /* (sc, sh) of channels 8 (d >> 1) + 4 h + 2 (d & 1) + {0, 1} */ \
          const float z0 = __uint_as_float(zw[d] << 16), z1 = __uint_as_float(zw[d] & 0xFFFF0000u);             \
          const float g0 = fmaf(z0, c4.x, c4.y) > 0.0f ? acc[2 * d] : 0.0f;                                     \
          const float g1 = fmaf(z1, c4.z, c4.w) > 0.0f ? acc[2 * d + 1] : 0.0f;                                 \
          gw8[d] = br_pack2(g0, g1);                                                                            \
          const float r0 = __uint_as_float(gw8[d] << 16), r1 = __uint_as_float(gw8[d] & 0xFFFF0000u);           \
          s1[2 * d] += r0;                                                                                      \
          s2[2 * d] = fmaf(r0, z0, s2[2 * d]);                                                                  \
          s1[2 * d + 1] += r1;                                                                                  \
          s2[2 * d + 1] = fmaf(r1, z1, s2[2 * d + 1]);                                                          \
        }                                                                                                       \
        br_swap8(gw8);                                                                                          \
        const __amdgpu_buffer_rsrc_t orow = __builtin_amdgcn_make_buffer_rsrc(g2 + (img + jo) * W * 128, 0,     \
                                                                              zrow_bytes, 0x00020000);          \
        const unsigned off = (unsigned)px * 256u + (unsigned)q * 64u + 16u * (unsigned)(ln >> 5);               \
        const u32x4 o0 = {gw8[0], gw8[1], gw8[2], gw8[3]}, o1 = {gw8[4], gw8[5], gw8[6], gw8[7]};               \
        __builtin_amdgcn_raw_buffer_store_b128(o0, orow, off, 0, 0);                                            \
        __builtin_amdgcn_raw_buffer_store_b128(o1, orow, off + 32u, 0, 0);                                      \
      }                                                                                                         \
      write_dy(jo + 2, DX);                                                                                     \
      ++jo;                                                                                                     \
      more = jo < j1;                                                                                           \
    }
    while (true) {
      MCL_BR_STEP(dB, zB, dA, zA)
      if (!more) break;
      MCL_BR_STEP(dA, zA, dB, zB)
      if (!more) break;
    }
#undef MCL_BR_STEP

    // ---- unit sums: sum_p g and sum_p g*zhat = rstd (sum g*z - mean sum g) per channel, lanes 31 / 63 write.  Lanes whose
    // pixel lies beyond the image width hold sums of values that were never stored (their z reads returned zeros, their g2
    // stores were dropped by the descriptor's range check): they contribute nothing.
    const float vmask = x0 + l31 < W ? 1.0f : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s1[r] = br_half_wave_sum(s1[r] * vmask);
      s2[r] = br_half_wave_sum(s2[r] * vmask);
    }
    if (l31 == 31) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = 32 * q + (r & 3) + 8 * (r >> 2) + 4 * h;
        partial[(long long)c * nsu + su] = make_float2(s1[r], rstd[c] * fmaf(-mean[c], s1[r], s2[r]));
      }
    }
  }
}

struct BwdRowsPlan {
  int nimg, nstrip, rc, nchunk, nsu, grid;
};

inline BwdRowsPlan bwd_rows_plan(long long S, int H, int W) {
  BwdRowsPlan p;
  p.nimg = (int)(S / ((long long)H * W));
  p.nstrip = (W + 31) / 32;
  // four channel-quarter waves per spatial unit; units sized to fill the 2048 wave slots about once
  long long rc = ((long long)H * p.nimg * p.nstrip * 4) / 2048;
  if (rc < 2) rc = 2;
  if (rc > H) rc = H;
  p.rc = (int)rc;
  p.nchunk = (H + p.rc - 1) / p.rc;
  p.nsu = p.nimg * p.nchunk * p.nstrip;
  p.grid = (4 * p.nsu + BR_NWAVE - 1) / BR_NWAVE;
  if (p.grid > 256) p.grid = 256;
  return p;
}

inline bool bwd_rows_applicable(long long S, int H, int W) {
  return W >= 17 && W <= 150 && S % ((long long)H * W) == 0;
}

// dz = gamma*rstd*(g2 - c1 - zhat*c2), elementwise over (S, 128) bf16
__global__ __launch_bounds__(256) void bn2_dz_kernel(const bf16_t* __restrict__ g2, const bf16_t* __restrict__ z,
                                                     long long n_chunks, const float* __restrict__ gamma,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ coef, bf16_t* __restrict__ dz) {
  const int cc = threadIdx.x & 15;
  const long long stride = (long long)gridDim.x * 256;
  const long long q0 = (long long)blockIdx.x * 256 + threadIdx.x;
  // the first chunk's data loads go out before the 40 coefficient loads (otherwise two memory round trips in series at the
  // head of a 5 us kernel)
  uint4 gv = make_uint4(0u, 0u, 0u, 0u), zv = gv;
  if (q0 < n_chunks) {
    gv = *reinterpret_cast<const uint4*>(g2 + q0 * 8);
    zv = *reinterpret_cast<const uint4*>(z + q0 * 8);
  }
  float mu[8], rs[8], sc[8], c1[8], c2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = cc * 8 + i;
    mu[i] = mean[c];
    rs[i] = rstd[c];
    sc[i] = gamma[c] * rs[i];
    c1[i] = coef[2 * c];
    c2[i] = coef[2 * c + 1];
  }
  for (long long q = q0; q < n_chunks; q += stride) {
    if (q != q0) {
      gv = *reinterpret_cast<const uint4*>(g2 + q * 8);
      zv = *reinterpret_cast<const uint4*>(z + q * 8);
    }
    const unsigned gw[4] = {gv.x, gv.y, gv.z, gv.w}, zw[4] = {zv.x, zv.y, zv.z, zv.w};
    unsigned o[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float g_lo = __uint_as_float(gw[u] << 16), g_hi = __uint_as_float(gw[u] & 0xFFFF0000u);
      const float z_lo = __uint_as_float(zw[u] << 16), z_hi = __uint_as_float(zw[u] & 0xFFFF0000u);
      const float d_lo = sc[2 * u] * (g_lo - c1[2 * u] - (z_lo - mu[2 * u]) * rs[2 * u] * c2[2 * u]);
      const float d_hi = sc[2 * u + 1] * (g_hi - c1[2 * u + 1] - (z_hi - mu[2 * u + 1]) * rs[2 * u + 1] * c2[2 * u + 1]);
      const f32x2 pv = {d_lo, d_hi};
      o[u] = __builtin_bit_cast(unsigned, __builtin_convertvector(pv, bf16x2_t));
    }
    *reinterpret_cast<uint4*>(dz + q * 8) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

}  // namespace

extern "C" int64_t mcl_dense_bn1_bwd_workspace_floats(int64_t S, int32_t C) {
  if (S <= 0 || C <= 0) return -1;
  return ((S + 63) / 64) * 2 * (int64_t)C + 2 * (int64_t)C;     // sized for the 64-row tiling
}

extern "C" int mcl_dense_bn1_bwd(const void* dz, const void* W1, int32_t C, const void* x, int64_t ldx, int64_t S,
                                 const float* gamma, const float* beta, const float* mean, const float* rstd,
                                 float* workspace, float* dgamma, float* dbeta, int32_t accumulate_params, void* gbuf,
                                 int64_t ldg, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!dz || !W1 || !x || !gamma || !beta || !mean || !rstd || !workspace || !dgamma || !dbeta || !gbuf || S <= 0 || C <= 0)
    return MCL_EINVAL;
  if ((C % 8) || (ldx % 8) || (ldg % 8) || (reinterpret_cast<uintptr_t>(dz) & 15u) ||
      (reinterpret_cast<uintptr_t>(W1) & 15u) || (reinterpret_cast<uintptr_t>(x) & 15u) ||
      (reinterpret_cast<uintptr_t>(gbuf) & 15u))
    return MCL_EUNSUPPORTED;
  // 64-row tiles: 50 KB of LDS and <= 168 VGPRs -> three workgroups per CU instead of two (the 128-row tiling of round 1 lost
  // in every block: 2302 -> 1835 us/step (dx), 1774 -> 1561 us/step (reduce) summed over the 58 layers).
  const int tmv = 64;
  const int nrt = (int)((S + tmv - 1) / tmv), nct = (C + TN - 1) / TN;
  float2* part = reinterpret_cast<float2*>(workspace);
  float* coef = workspace + (int64_t)nrt * 2 * C;
  hipStream_t st = mcl_stream(stream);
  // persistent over row tiles: two (128-row tiles) or three (64-row tiles) workgroups per CU, each keeps one column tile
  // 512 workgroups (two per CU) although LDS and registers allow three: alone the kernels are faster with three (r01:
  // 2302 -> 1835 us/step serial), but in the step they share the CUs with the weight-gradient kernels of the side stream,
  // which cannot become resident beside three: 14.02 / 14.09 ms/step at 768, 13.92 / 13.96 at 512.
  // Only on the 28 x 28 and larger maps: below, the grids do not fill the chip anyway and three per CU stay (no difference
  // in the step, 13.84-13.95 either way)
  const int gcap = S >= 50000 ? 512 : 768;
  int gx = (gcap + nct - 1) / nct;
  if (gx > nrt) gx = nrt;
  dim3 grid(gx, nct);
#define MCL_BN1(MODE, TMV, COEF, GB, LDG, PART)                                                                         \
  hipLaunchKernelGGL((bn1_bwd_kernel<MODE, TMV>), grid, dim3(256), 0, st, (const bf16_t*)dz, (const bf16_t*)W1, C, C,    \
                     (const bf16_t*)x, (long long)ldx, (long long)S, gamma, beta, mean, rstd, COEF, GB, LDG, PART, nrt)
  MCL_BN1(0, 64, (const float*)nullptr, (bf16_t*)nullptr, 0LL, part);
  hipLaunchKernelGGL(bn1_bwd_finalize_kernel, dim3(C), dim3(256), 0, st, (const float2*)part, nrt, C,
                     (long long)S, dgamma, dbeta, coef, accumulate_params);
  MCL_BN1(1, 64, (const float*)coef, (bf16_t*)gbuf, (long long)ldg, (float2*)nullptr);
#undef MCL_BN1
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

// Single-pass form for the latency-bound small maps.  dx = gamma*rstd*(g - mean g - xhat*mean(g*xhat)) is linear in the two
// means, and the statistics (mean, rstd) of a concat-buffer channel are the same for every layer of the block.  ONE kernel adds
// gamma*rstd*g into the gradient buffer, reduces the two sums, and -- on the same elements -- subtracts the mean terms of the
// PREVIOUS pass (``kprev``: [C_total][2] fp32, read when ``have_prev``, then overwritten by this layer's finalize).  So every
// layer's mean terms reach the channels the next layer reads one pass late (the buffer never carries more than one layer's
// un-subtracted mean component), and mcl_dense_bn1_fix applies them to the channels the next layer does not read.  Saves the
// separate reduce pass over (dz, x) of mcl_dense_bn1_bwd.
extern "C" int mcl_dense_bn1_dx_sums(const void* dz, const void* W1, int32_t C, const void* x, int64_t ldx, int64_t S,
                                     const float* gamma, const float* beta, const float* mean, const float* rstd,
                                     float* workspace, float* dgamma, float* dbeta, int32_t accumulate_params, float* kprev,
                                     int32_t have_prev, void* gbuf, int64_t ldg, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!dz || !W1 || !x || !gamma || !beta || !mean || !rstd || !workspace || !dgamma || !dbeta || !kprev || !gbuf || S <= 0 ||
      C <= 0)
    return MCL_EINVAL;
  if ((C % 8) || (ldx % 8) || (ldg % 8) || (reinterpret_cast<uintptr_t>(dz) & 15u) ||
      (reinterpret_cast<uintptr_t>(W1) & 15u) || (reinterpret_cast<uintptr_t>(x) & 15u) ||
      (reinterpret_cast<uintptr_t>(gbuf) & 15u))
    return MCL_EUNSUPPORTED;
  const int nrt = (int)((S + 63) / 64), nct = (C + TN - 1) / TN;
  float2* part = reinterpret_cast<float2*>(workspace);
  hipStream_t st = mcl_stream(stream);
  const int gcap = S >= 50000 ? 512 : 768;
  int gx = (gcap + nct - 1) / nct;
  if (gx > nrt) gx = nrt;
  hipLaunchKernelGGL((bn1_bwd_kernel<2, 64>), dim3(gx, nct), dim3(256), 0, st, (const bf16_t*)dz, (const bf16_t*)W1, C, C,
                     (const bf16_t*)x, (long long)ldx, (long long)S, gamma, beta, mean, rstd,
                     have_prev ? (const float*)kprev : (const float*)nullptr,
                     (bf16_t*)gbuf, (long long)ldg, part, nrt);
  hipLaunchKernelGGL(bn1_bwd_finalize_kernel, dim3(C), dim3(256), 0, st, (const float2*)part, nrt, C, (long long)S, dgamma,
                     dbeta, (float*)nullptr, accumulate_params, kprev, gamma, rstd);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_dense_bn1_fix(const void* x, int64_t ldx, void* gbuf, int64_t ldg, int64_t S, int32_t c0, int32_t nc,
                                 const float* mean, const float* rstd, const float* kacc, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!x || !gbuf || !mean || !rstd || !kacc || S <= 0 || c0 < 0 || nc <= 0) return MCL_EINVAL;
  if ((c0 % 8) || (nc % 8) || (ldx % 8) || (ldg % 8) || (reinterpret_cast<uintptr_t>(x) & 15u) ||
      (reinterpret_cast<uintptr_t>(gbuf) & 15u))
    return MCL_EUNSUPPORTED;
  const long long n = S * (nc >> 3);
  long long nb = (n + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(bn1_fix_kernel, dim3((int)nb), dim3(256), 0, mcl_stream(stream), (const bf16_t*)x, (long long)ldx,
                     (bf16_t*)gbuf, (long long)ldg, (long long)S, c0, nc, mean, rstd, kacc);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

// dx pass alone: gbuf += gamma*rstd*(g - coef[2c] - xhat*coef[2c+1]),  g = relu'(.)*(dz W1), with the two means per channel
// supplied by the caller (mcl_dense_bn1_wrw derives them from the weight-gradient Gram matrices: no reduce launch).
namespace {
int bn1_dx_impl(const void* dz, const void* W1, int32_t C, int32_t ldw, const void* x, int64_t ldx, int64_t S, const float* gamma,
                const float* beta, const float* mean, const float* rstd, const float* coef, void* gbuf, int64_t ldg,
                mcl_stream_t stream) {
  if (!dz || !W1 || !x || !gamma || !beta || !mean || !rstd || !coef || !gbuf || S <= 0 || C <= 0 || ldw < C) return MCL_EINVAL;
  if ((C % 8) || (ldw % 8) || (ldx % 8) || (ldg % 8) || (reinterpret_cast<uintptr_t>(dz) & 15u) ||
      (reinterpret_cast<uintptr_t>(W1) & 15u) || (reinterpret_cast<uintptr_t>(x) & 15u) ||
      (reinterpret_cast<uintptr_t>(gbuf) & 15u))
    return MCL_EUNSUPPORTED;
  const int nrt = (int)((S + 63) / 64), nct = (C + TN - 1) / TN;
  const int gcap = S >= 50000 ? 512 : 768;
  int gx = (gcap + nct - 1) / nct;
  if (gx > nrt) gx = nrt;
  hipLaunchKernelGGL((bn1_bwd_kernel<1, 64>), dim3(gx, nct), dim3(256), 0, mcl_stream(stream), (const bf16_t*)dz,
                     (const bf16_t*)W1, C, ldw, (const bf16_t*)x, (long long)ldx, (long long)S, gamma, beta, mean, rstd, coef,
                     (bf16_t*)gbuf, (long long)ldg, (float2*)nullptr, nrt);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}
}  // namespace

extern "C" int mcl_dense_bn1_dx(const void* dz, const void* W1, int32_t C, const void* x, int64_t ldx, int64_t S,
                                const float* gamma, const float* beta, const float* mean, const float* rstd,
                                const float* coef, void* gbuf, int64_t ldg, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  return bn1_dx_impl(dz, W1, C, C, x, ldx, S, gamma, beta, mean, rstd, coef, gbuf, ldg, stream);
}

// The same pass over a channel WINDOW [c0, c0 + nc) of the layer's input (round 6, the paired form below): every per-channel
// pointer (W1 column, x, gamma, beta, mean, rstd, coef (2 floats per channel), gbuf) is offset by c0 here.
extern "C" int mcl_dense_bn1_dx_window(const void* dz, const void* W1, int32_t C, int32_t c0, int32_t nc, const void* x,
                                       int64_t ldx, int64_t S, const float* gamma, const float* beta, const float* mean,
                                       const float* rstd, const float* coef, void* gbuf, int64_t ldg, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (c0 < 0 || nc <= 0 || c0 + nc > C || (c0 % 8)) return MCL_EINVAL;
  if (!dz || !W1 || !x || !gamma || !beta || !mean || !rstd || !coef || !gbuf) return MCL_EINVAL;
  return bn1_dx_impl(dz, (const bf16_t*)W1 + c0, nc, C, (const bf16_t*)x + c0, ldx, S, gamma + c0, beta + c0, mean + c0, rstd + c0,
                     coef + 2 * c0, (bf16_t*)gbuf + c0, ldg, stream);
}

// TWO layers' dx passes as ONE pass over the channels both read (round 6): layer A = l (its input has C + 32
// channels: W1A rows are ldwA long) and layer B = l - 1 (C channels).  gbuf[:, 0:C] += termA + termB with x and the gradient buffer
// read once and the buffer written once -- the traffic of one pass instead of two; both da = dz W1 products are recomputed on the
// matrix cores.  (Layer A's term on its last 32 input channels, which layer B's 3x3 backward consumes, is applied before by
// mcl_dense_bn1_dx_window.)
extern "C" int mcl_dense_bn1_dx_pair(const void* dzA, const void* W1A, int32_t ldwA, const float* gammaA, const float* betaA,
                                     const float* coefA, const void* dzB, const void* W1B, const float* gammaB,
                                     const float* betaB, const float* coefB, int32_t C, const void* x, int64_t ldx, int64_t S,
                                     const float* mean, const float* rstd, void* gbuf, int64_t ldg, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!dzA || !W1A || !gammaA || !betaA || !coefA || !dzB || !W1B || !gammaB || !betaB || !coefB || !x || !mean || !rstd || !gbuf ||
      S <= 0 || C <= 0 || ldwA < C)
    return MCL_EINVAL;
  if ((C % 8) || (ldwA % 8) || (ldx % 8) || (ldg % 8) || (reinterpret_cast<uintptr_t>(dzA) & 15u) ||
      (reinterpret_cast<uintptr_t>(dzB) & 15u) || (reinterpret_cast<uintptr_t>(W1A) & 15u) ||
      (reinterpret_cast<uintptr_t>(W1B) & 15u) || (reinterpret_cast<uintptr_t>(x) & 15u) ||
      (reinterpret_cast<uintptr_t>(gbuf) & 15u))
    return MCL_EUNSUPPORTED;
  static mcl_device_once attr_once;
  if (auto guard = attr_once.first())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bn1_dx_pair_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)PAIR_LDS);
  const int nrt = (int)((S + 63) / 64), nct = (C + TN - 1) / TN;
  const int gcap = 512;
  int gx = (gcap + nct - 1) / nct;
  if (gx > nrt) gx = nrt;
  hipLaunchKernelGGL(bn1_dx_pair_kernel, dim3(gx, nct), dim3(256), PAIR_LDS, mcl_stream(stream), (const bf16_t*)dzA,
                     (const bf16_t*)W1A, ldwA, gammaA, betaA, coefA, (const bf16_t*)dzB, (const bf16_t*)W1B, gammaB, betaB, coefB, C,
                     (const bf16_t*)x, (long long)ldx, (long long)S, mean, rstd, (bf16_t*)gbuf, (long long)ldg, nrt);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int64_t mcl_dense_conv3x3_bwd_workspace_floats(int64_t S) {
  if (S <= 0) return -1;
  // flat form: one partial per 128-pixel tile; row-walking form: one per spatial unit (<= image rows x strips / 2 <= S / 32)
  const int64_t flat = (S + T3B - 1) / T3B, rows = S / 32 + 8;
  return (flat > rows ? flat : rows) * 2 * (int64_t)C3I + 2 * (int64_t)C3I;
}

namespace {
int conv3x3_bwd_impl(const void* dy, int64_t lddy, int64_t S, int32_t H, int32_t W, const void* W2, const void* z,
                     const float* gamma, const float* beta, const float* mean, const float* rstd, float* workspace,
                     float* dgamma, float* dbeta, int32_t accumulate_params, void* g2, void* dz, const void* xfix,
                     int64_t ldxf, const float* fmean, const float* frstd, const float* fk, void* dyc, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!dy || !W2 || !z || !gamma || !beta || !mean || !rstd || !workspace || !dgamma || !dbeta || !g2 || !dz || S <= 0 ||
      H <= 0 || W <= 0)
    return MCL_EINVAL;
  if ((S % ((int64_t)H * W)) || S > 0x7fff0000LL || W > 150 || (lddy % 8) || lddy < C3O ||
      (reinterpret_cast<uintptr_t>(dy) & 15u) || (reinterpret_cast<uintptr_t>(z) & 15u) ||
      (reinterpret_cast<uintptr_t>(g2) & 15u) || (reinterpret_cast<uintptr_t>(dz) & 15u))
    return MCL_EUNSUPPORTED;
  if (xfix != nullptr) {
    if (!fmean || !frstd || !fk || !dyc) return MCL_EINVAL;
    if (bwd_rows_applicable(S, H, W) || (ldxf % 8) || (reinterpret_cast<uintptr_t>(xfix) & 15u) ||
        (reinterpret_cast<uintptr_t>(dyc) & 15u))
      return MCL_EUNSUPPORTED;                        // (the folded fix lives in the flat-tile kernel: maps narrower than 17)
  }
  hipStream_t st = mcl_stream(stream);
  float2* part = reinterpret_cast<float2*>(workspace);
  float* coef;
  if (bwd_rows_applicable(S, H, W)) {                 // the large maps: thin row-walking waves
    const BwdRowsPlan p = bwd_rows_plan(S, H, W);
    coef = workspace + (int64_t)p.nsu * 2 * C3I;
    static mcl_device_once attr_once;
    if (auto attr_guard = attr_once.first()) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_bwd_rows_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    hipLaunchKernelGGL(conv3x3_bwd_rows_kernel, dim3(p.grid), dim3(64 * BR_NWAVE), (size_t)BR_NWAVE * BR_RING + 1024, st,
                       (const bf16_t*)dy, (long long)lddy, p.nimg, H, W, (const bf16_t*)W2, (const bf16_t*)z, gamma, beta,
                       mean, rstd, (bf16_t*)g2, part, p.nsu, p.rc, p.nchunk, p.nstrip);
    hipLaunchKernelGGL(bn1_bwd_finalize_kernel, dim3(C3I), dim3(256), 0, st, (const float2*)part, p.nsu, C3I,
                       (long long)S, dgamma, dbeta, coef, accumulate_params);
  } else {
    const int ntile = (int)((S + T3B - 1) / T3B);
    coef = workspace + (int64_t)ntile * 2 * C3I;
    const size_t lds_bytes = (size_t)T3B * 256 + (size_t)(T3B + 2 * W + 2) * 64 + 64;
    const int gcap = S >= 50000 ? 512 : 768;
    hipLaunchKernelGGL(conv3x3_bwd_kernel, dim3(ntile < gcap ? ntile : gcap), dim3(256), lds_bytes, st, (const bf16_t*)dy,
                       (long long)lddy, (long long)S, H, W, (const bf16_t*)W2, (const bf16_t*)z, gamma, beta, mean, rstd,
                       (bf16_t*)g2, part, ntile, (const bf16_t*)xfix, (long long)ldxf, fmean, frstd, fk, (bf16_t*)dyc);
    hipLaunchKernelGGL(bn1_bwd_finalize_kernel, dim3(C3I), dim3(256), 0, st, (const float2*)part, ntile, C3I,
                       (long long)S, dgamma, dbeta, coef, accumulate_params);
  }
  const long long n_chunks = S * (C3I / 8);
  long long blocks = (n_chunks + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(bn2_dz_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)g2, (const bf16_t*)z,
                     n_chunks, gamma, mean, rstd, (const float*)coef, (bf16_t*)dz);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}
}  // namespace

extern "C" int mcl_dense_conv3x3_bwd(const void* dy, int64_t lddy, int64_t S, int32_t H, int32_t W, const void* W2,
                                     const void* z, const float* gamma, const float* beta, const float* mean,
                                     const float* rstd, float* workspace, float* dgamma, float* dbeta,
                                     int32_t accumulate_params, void* g2, void* dz, mcl_stream_t stream) {
  return conv3x3_bwd_impl(dy, lddy, S, H, W, W2, z, gamma, beta, mean, rstd, workspace, dgamma, dbeta, accumulate_params, g2,
                          dz, nullptr, 0, nullptr, nullptr, nullptr, nullptr, stream);
}

// The same with mcl_dense_bn1_fix folded into the dy staging (maps narrower than 17 pixels: the flat-tile kernel):
// dy'[s][c] = dy[s][c] - (K1[c] + K2[c]*xhat[s][c]) for the layer's 32 output channels (xfix / fmean / frstd / fk = the
// concat buffer, its statistics and the previous pass's mean terms [32][2], all offset to the layer's first output
// channel); dyc (S x 32 bf16, contiguous) receives dy' for mcl_dense_conv3x3_wrw_det.  Same arithmetic as the separate
// launch, bit for bit.
extern "C" int mcl_dense_conv3x3_bwd_fix(const void* dy, int64_t lddy, int64_t S, int32_t H, int32_t W, const void* W2,
                                         const void* z, const float* gamma, const float* beta, const float* mean,
                                         const float* rstd, float* workspace, float* dgamma, float* dbeta,
                                         int32_t accumulate_params, void* g2, void* dz, const void* xfix, int64_t ldxf,
                                         const float* fmean, const float* frstd, const float* fk, void* dyc,
                                         mcl_stream_t stream) {
  if (!xfix) return MCL_EINVAL;
  return conv3x3_bwd_impl(dy, lddy, S, H, W, W2, z, gamma, beta, mean, rstd, workspace, dgamma, dbeta, accumulate_params, g2,
                          dz, xfix, ldxf, fmean, frstd, fk, dyc, stream);
}
