// Weight gradient of the DenseNet bottleneck 1x1 convolution FUSED with the reduction pass of the train-mode
// BatchNorm backward in front of it -- deterministic (no atomics), one pass over the layer input.
//
// Layer head:  x --norm1--> y = gamma*xhat + beta --relu1--> a --conv1 (W1: 128 x C)--> z.   Given dz (S x 128):
//
//     dW1[m][c]  = sum_s dz[s][m] * a[s][c]                                  (1x1 weight gradient)
//     da[s][c]   = sum_m dz[s][m] * W1[m][c] ,   g = da * mask ,  mask = [y > 0]
//     dbeta[c]   = sum_s g[s][c] ,   dgamma[c] = sum_s g[s][c] * xhat[s][c]  (what the dx pass needs as means)
//
// Round 1 ran two kernels over (dz, x): the weight gradient (fp32 atomics into dW1) and a "reduce" launch that
// recomputed da = dz W1 on the matrix cores only to form the two sums.  Both sums are linear in dz, so they follow
// from the SAME two small Gram matrices the weight gradient is made of:
//
//     R[m][c]  = sum_s dz[s][m] * mask[s][c]            Qx[m][c] = sum_s dz[s][m] * (mask*x)[s][c]
//     Q        = rstd*(Qx - mean*R)                     ( = sum_s dz * mask * xhat )
//     dW1      = gamma*Q + beta*R                       ( a = mask*(gamma*xhat + beta) )
//     dbeta[c] = sum_m W1[m][c]*R[m][c]                 dgamma[c] = sum_m W1[m][c]*Q[m][c]
//
// mask*x is the raw bf16 input with masked elements zeroed (no re-rounding) and mask is exactly 0/1, so the operands
// are exact; the matrix cores do twice the (free: the kernel is HBM-bound, AI ~ 256 flop/B) work and the separate
// reduce launch + its finalize launch disappear.
//
// Kernel A (wrw_partial_kernel): "TN" GEMM with the huge dimension (S pixels, up to 401k) as K.  A workgroup owns one
// 128 x 128 output tile and one slab of pixels; operands are staged exactly as they lie in HBM (coalesced 16-byte
// chunks) and MFMA fragments come from the transposing LDS read ds_read_b64_tr_b16.  Its fp32 partial goes to a
// workspace with plain stores.  Workgroups that share a dz slab (the column tiles of one slab) are mapped to the same
// XCD so the slab is fetched from HBM once and re-read from that XCD's L2.
// Kernel B (wrw_merge_kernel): fixed-order sum of the slab partials -> dW1 (+=), dgamma / dbeta (+=) and the two
// means of the dx pass.  Bit-reproducible run to run.
//
// FUSED = false is the plain weight gradient dW = dz^T a (transition convolutions: a = pooled activation).
#include "common.h"

namespace {

typedef unsigned short bf16_t;
typedef short v4s __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 64;        // pixels per LDS tile
constexpr int BT = 128;       // channel-tile width of both operands
constexpr int PITCH = 160;    // elements per LDS row (128 + 32 pad = 320 B): the four 16-lane groups of a
                              // transposing read hit 64 distinct banks
constexpr int NCH = BK / 16;  // 16-byte chunks per thread and operand tile

// 8 consecutive-k bf16 of channel (cbase + lane&15 [+16 for odd 16-lane groups]) starting at row kbase
__device__ __forceinline__ bf16x8 frag(const bf16_t* tile, int kbase, int cbase, int lane) {
  const int i = lane & 15;
  const bf16_t* p = tile + (kbase + (i >> 2)) * PITCH + cbase + (i & 3) * 4;
  const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)p);
  const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(p + 4 * PITCH));
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

template <bool FUSED>
__global__ __launch_bounds__(256, 2) void wrw_partial_kernel(
    const bf16_t* __restrict__ dz, long long ldz, const bf16_t* __restrict__ x, long long ldx,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
    const float* __restrict__ rstd, const bf16_t* __restrict__ W1 /* [M][N] */, float* __restrict__ wpart /* [ks][M][N] */,
    float* __restrict__ spart /* [ks][2][N] */, long long S, int M, int N, long long rows_per_wg, int ks, int tn) {
  __shared__ __attribute__((aligned(16))) bf16_t lds[(FUSED ? 3 : 2) * BK * PITCH];
  __shared__ float red[2][2][BT];                     // [wm][sum][column]: cross-wave column sums of the epilogue
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware decode (block b runs on XCD b % 8): the tn column tiles of one pixel slab share an XCD (its L2 serves
  // the dz slab to all of them); consecutive slabs go round the 8 XCDs
  const int L = blockIdx.x, xcd = L & 7, qq = L >> 3;
  const int tx = qq % tn, z = (qq / tn) * 8 + xcd;
  if (z >= ks) return;
  const int n0 = tx * BT;
  const long long s_begin = (long long)z * rows_per_wg;
  const long long s_end = min(S, s_begin + rows_per_wg);

  bf16_t* tA = lds;                      // dz tile
  bf16_t* tX = lds + BK * PITCH;         // mask*x (FUSED) / a (plain)
  bf16_t* tM = lds + 2 * BK * PITCH;     // mask as bf16 1.0 / 0.0 (FUSED)

  const int col = (tid & 15) * 8, rr = tid >> 4;
  const bool cok_a = col < M, cok_x = n0 + col < N;
  float sc[8], sh[8];
  if (FUSED) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {       // mask = [x*sc + sh > 0]
      const int c = n0 + col + i;
      const bool ok = c < N;
      sc[i] = ok ? gamma[c] * rstd[c] : 0.0f;
      sh[i] = ok ? fmaf(-mean[c], sc[i], beta[c]) : -1.0f;
    }
  }

  f32x16 accq[2][2], accr[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        accq[i][j][r] = 0.0f;
        accr[i][j][r] = 0.0f;
      }

  uint4 ra[NCH], rx[NCH];
  auto gload = [&](long long s0) {
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      const long long s = s0 + rr + 16 * h;
      const bool ok = s < s_end;
      ra[h] = (ok && cok_a) ? *reinterpret_cast<const uint4*>(dz + s * ldz + col) : make_uint4(0u, 0u, 0u, 0u);
      rx[h] = (ok && cok_x) ? *reinterpret_cast<const uint4*>(x + s * ldx + n0 + col) : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      const int row = rr + 16 * h;
      *reinterpret_cast<uint4*>(tA + row * PITCH + col) = ra[h];
      if (!FUSED) {
        *reinterpret_cast<uint4*>(tX + row * PITCH + col) = rx[h];
      } else {
        const unsigned w[4] = {rx[h].x, rx[h].y, rx[h].z, rx[h].w};
        unsigned xm[4], mk[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float lo = __uint_as_float(w[i] << 16), hi = __uint_as_float(w[i] & 0xFFFF0000u);
          const bool plo = fmaf(lo, sc[2 * i], sh[2 * i]) > 0.0f, phi = fmaf(hi, sc[2 * i + 1], sh[2 * i + 1]) > 0.0f;
          const unsigned sel = (plo ? 0x0000FFFFu : 0u) | (phi ? 0xFFFF0000u : 0u);
          xm[i] = w[i] & sel;
          mk[i] = 0x3F803F80u & sel;
        }
        *reinterpret_cast<uint4*>(tX + row * PITCH + col) = make_uint4(xm[0], xm[1], xm[2], xm[3]);
        *reinterpret_cast<uint4*>(tM + row * PITCH + col) = make_uint4(mk[0], mk[1], mk[2], mk[3]);
      }
    }
  };

  const int nt = (int)((s_end - s_begin + BK - 1) / BK);
  const int half16 = 16 * ((lane >> 4) & 1);
  const int kg = 8 * (lane >> 5);
  if (nt > 0) gload(s_begin);
  for (int t = 0; t < nt; ++t) {
    __syncthreads();                       // previous tile's fragments are consumed
    lstore();
    __syncthreads();
    if (t + 1 < nt) gload(s_begin + (long long)(t + 1) * BK);      // in flight under this tile's MFMAs
#pragma unroll
    for (int kk = 0; kk < BK; kk += 16) {
      bf16x8 fa[2], fx[2], fm[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = frag(tA, kk + kg, wm * 64 + i * 32 + half16, lane);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        fx[j] = frag(tX, kk + kg, wn * 64 + j * 32 + half16, lane);
        if (FUSED) fm[j] = frag(tM, kk + kg, wn * 64 + j * 32 + half16, lane);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          accq[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fx[j], accq[i][j], 0, 0, 0);
          if (FUSED) accr[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fm[j], accr[i][j], 0, 0, 0);
        }
    }
  }

  // ---- epilogue: acc[i][j][r] is element m = wm*64 + i*32 + (r&3) + 8*(r>>2) + 4*(lane>>5), n = n0 + wn*64 + j*32 + (lane&31)
  float* wp = wpart + (long long)z * M * N;
  const int hh = lane >> 5, l31 = lane & 31;
  float t1[2] = {0.0f, 0.0f}, t2[2] = {0.0f, 0.0f};
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 64 + j * 32 + l31;
    const bool nok = n < N;
    float g = 0.0f, b = 0.0f, mu = 0.0f, rs = 0.0f;
    if (FUSED && nok) {
      g = gamma[n]; b = beta[n]; mu = mean[n]; rs = rstd[n];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (!nok || m >= M) continue;
        if (FUSED) {
          const float R = accr[i][j][r];
          const float Q = rs * fmaf(-mu, R, accq[i][j][r]);
          const float w = __uint_as_float(((unsigned)W1[(long long)m * N + n]) << 16);
          t1[j] = fmaf(w, R, t1[j]);
          t2[j] = fmaf(w, Q, t2[j]);
          wp[(long long)m * N + n] = fmaf(g, Q, b * R);
        } else {
          wp[(long long)m * N + n] = accq[i][j][r];
        }
      }
  }
  if (FUSED) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      t1[j] += __shfl_xor(t1[j], 32, 64);
      t2[j] += __shfl_xor(t2[j], 32, 64);
      if (hh == 0) {
        red[wm][0][wn * 64 + j * 32 + l31] = t1[j];
        red[wm][1][wn * 64 + j * 32 + l31] = t2[j];
      }
    }
    __syncthreads();
    if (tid < BT && n0 + tid < N) {
      float* sp = spart + (long long)z * 2 * N;
      sp[n0 + tid] = red[0][0][tid] + red[1][0][tid];
      sp[N + n0 + tid] = red[0][1][tid] + red[1][1][tid];
    }
  }
}

// dW[m][n] (+)= sum_z wpart[z][m][n] (blocks [0, nbw)): a block owns 32 float4 columns; its 8 thread groups each sum
// every 8th slab, then the 8 group sums are added in fixed order -- the same order every run.  Per channel (blocks
// [nbw, ...)): the two BatchNorm-backward sums in slab order (double), parameter gradients, and the means of the dx pass.
__global__ __launch_bounds__(256) void wrw_merge_kernel(const float* __restrict__ wpart, const float* __restrict__ spart,
                                                        int ks, long long MN, int N, long long S, float* __restrict__ dW,
                                                        int accumulate_w, float* __restrict__ dgamma,
                                                        float* __restrict__ dbeta, int accumulate_params,
                                                        float* __restrict__ coef, int nbw) {
  __shared__ float4 part[8][32];
  if ((int)blockIdx.x < nbw) {
    const int q = threadIdx.x & 31, zg = threadIdx.x >> 5;
    const long long e = ((long long)blockIdx.x * 32 + q) * 4;
    float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (e < MN) {                                          // MN is a multiple of 8 (M, N multiples of 8)
      for (int zz = zg; zz < ks; zz += 8) {
        const float4 v = *reinterpret_cast<const float4*>(wpart + (long long)zz * MN + e);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
      }
    }
    part[zg][q] = a;
    __syncthreads();
    if (zg != 0 || e >= MN) return;
#pragma unroll
    for (int g = 1; g < 8; ++g) {
      const float4 v = part[g][q];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    float4* o = reinterpret_cast<float4*>(dW + e);
    if (accumulate_w) {
      const float4 p = *o;
      a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
    }
    *o = a;
    return;
  }
  const int c = ((int)blockIdx.x - nbw) * 256 + threadIdx.x;
  if (c >= N) return;
  double a = 0.0, b = 0.0;
  for (int zz = 0; zz < ks; ++zz) {
    a += (double)spart[(long long)zz * 2 * N + c];
    b += (double)spart[(long long)zz * 2 * N + N + c];
  }
  if (accumulate_params) {
    dbeta[c] += (float)a;
    dgamma[c] += (float)b;
  } else {
    dbeta[c] = (float)a;
    dgamma[c] = (float)b;
  }
  coef[2 * c] = (float)(a / (double)S);
  coef[2 * c + 1] = (float)(b / (double)S);
}

struct Split {
  int tn, ks;
  long long rows;
};

inline Split plan(long long S, int N) {
  Split p;
  p.tn = (N + BT - 1) / BT;
  // ~2 workgroups per CU on the big maps; fewer on the small ones, where the 64 KB partial each workgroup writes
  // (and the merge re-reads) would otherwise outweigh the operands
  const long long target = S >= 200000 ? 512 : (S >= 50000 ? 384 : 256);
  long long ks = (target + p.tn - 1) / p.tn;
  const long long max_ks = (S + 4 * BK - 1) / (4 * BK);        // at least four LDS tiles per workgroup
  if (ks > max_ks) ks = max_ks;
  if (ks < 1) ks = 1;
  long long rows = (S + ks - 1) / ks;
  rows = (rows + BK - 1) / BK * BK;
  ks = (S + rows - 1) / rows;
  p.ks = (int)ks;
  p.rows = rows;
  return p;
}

}  // namespace

void mcl_launch_wrw_merge(const float* wpart, int ks, long long MN, float* dW, int accumulate_w, hipStream_t st) {
  const int nbw = (int)((MN / 4 + 31) / 32);
  hipLaunchKernelGGL(wrw_merge_kernel, dim3(nbw), dim3(256), 0, st, wpart, (const float*)nullptr, ks, MN, 0, 1LL, dW,
                     accumulate_w, (float*)nullptr, (float*)nullptr, 0, (float*)nullptr, nbw);
}

extern "C" int64_t mcl_wrw_workspace_floats(int64_t S, int32_t M, int32_t N) {
  if (S <= 0 || M <= 0 || N <= 0) return -1;
  const Split p = plan(S, N);
  return (int64_t)p.ks * ((int64_t)M * N + 2 * (int64_t)N) + 2 * (int64_t)N;
}

// Fused: dW1 (+)= dz^T relu(bn(x)); dgamma/dbeta (+)= BatchNorm backward sums; coef_out[2c], [2c+1] = the two means.
extern "C" int mcl_dense_bn1_wrw(const void* dz, const void* W1, int32_t C, const void* x, int64_t ldx, int64_t S,
                                 const float* gamma, const float* beta, const float* mean, const float* rstd,
                                 float* workspace, float* dW, int32_t accumulate_w, float* dgamma, float* dbeta,
                                 int32_t accumulate_params, float* coef_out, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!dz || !W1 || !x || !gamma || !beta || !mean || !rstd || !workspace || !dW || !dgamma || !dbeta || !coef_out ||
      S <= 0 || C <= 0)
    return MCL_EINVAL;
  if ((C % 8) || (ldx % 8) || (reinterpret_cast<uintptr_t>(dz) & 15u) || (reinterpret_cast<uintptr_t>(x) & 15u) ||
      (reinterpret_cast<uintptr_t>(dW) & 15u) || (reinterpret_cast<uintptr_t>(workspace) & 15u))
    return MCL_EUNSUPPORTED;
  const int M = 128, N = C;
  const Split p = plan(S, N);
  float* wpart = workspace;
  float* spart = workspace + (int64_t)p.ks * M * N;
  const int ks8 = (p.ks + 7) / 8 * 8;
  hipStream_t st = mcl_stream(stream);
  hipLaunchKernelGGL(wrw_partial_kernel<true>, dim3(ks8 * p.tn), dim3(256), 0, st, (const bf16_t*)dz, 128LL,
                     (const bf16_t*)x, (long long)ldx, gamma, beta, mean, rstd, (const bf16_t*)W1, wpart, spart,
                     (long long)S, M, N, p.rows, p.ks, p.tn);
  const long long MN = (long long)M * N;
  const int nbw = (int)((MN / 4 + 31) / 32), nbc = (N + 255) / 256;
  hipLaunchKernelGGL(wrw_merge_kernel, dim3(nbw + nbc), dim3(256), 0, st, (const float*)wpart, (const float*)spart,
                     p.ks, MN, N, (long long)S, dW, accumulate_w, dgamma, dbeta, accumulate_params, coef_out, nbw);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

// Plain: dW[M][N] (+)= dz[S][M]^T a[S][N]   (transition convolutions; deterministic replacement of the atomics kernel)
extern "C" int mcl_conv1x1_wrw_det(const void* dz, int64_t ldz, const void* a, int64_t lda, float* workspace, float* dW,
                                   int32_t accumulate_w, int64_t S, int32_t M, int32_t N, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!dz || !a || !dW || !workspace || S <= 0 || M <= 0 || N <= 0) return MCL_EINVAL;
  if ((M % 8) || (N % 8) || (ldz % 8) || (lda % 8) || (reinterpret_cast<uintptr_t>(dz) & 15u) ||
      (reinterpret_cast<uintptr_t>(a) & 15u) || (reinterpret_cast<uintptr_t>(dW) & 15u) ||
      (reinterpret_cast<uintptr_t>(workspace) & 15u))
    return MCL_EUNSUPPORTED;
  hipStream_t st = mcl_stream(stream);
  const Split p = plan(S, N);
  const int ks8 = (p.ks + 7) / 8 * 8;
  // M > 128 (transition convolutions: 128 / 256 / 512 output channels): one pass per 128 output channels, each with its
  // own partial set; the merge adds row block by row block
  for (int m0 = 0; m0 < M; m0 += BT) {
    const int mm = M - m0 < BT ? M - m0 : BT;
    float* wpart = workspace;
    hipLaunchKernelGGL(wrw_partial_kernel<false>, dim3(ks8 * p.tn), dim3(256), 0, st, (const bf16_t*)dz + m0,
                       (long long)ldz, (const bf16_t*)a, (long long)lda, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (const bf16_t*)nullptr, wpart, (float*)nullptr,
                       (long long)S, mm, N, p.rows, p.ks, p.tn);
    const long long MN = (long long)mm * N;
    const int nbw = (int)((MN / 4 + 31) / 32);
    hipLaunchKernelGGL(wrw_merge_kernel, dim3(nbw), dim3(256), 0, st, (const float*)wpart, (const float*)nullptr, p.ks,
                       MN, N, (long long)S, dW + (long long)m0 * N, accumulate_w, (float*)nullptr, (float*)nullptr, 0,
                       (float*)nullptr, nbw);
  }
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}
