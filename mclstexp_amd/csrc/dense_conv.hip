// DenseNet bottleneck 1x1 convolution with BatchNorm folded in on both sides (torchvision _DenseLayer:
// norm1 -> relu1 -> conv1, followed by norm2's batch statistics; /root/reference/model.py:75-76 via torchvision):
//
//     z[s][n] = sum_k relu(x[s][k]*scale[k] + shift[k]) * W[n][k]        x = concat buffer slice (S, K) bf16
//     (mean, var, rstd)[n] = batch statistics of the bf16-rounded z       -> norm2
//
// One HBM pass over the layer input instead of four: the stock sequence writes a = relu(bn1(x)), re-reads it in
// the convolution (MIOpen additionally zero-fills z for its split-K kernel) and re-reads z for the statistics.
// Here `a` never exists: BN+ReLU is applied to the 16-byte chunks on their way from HBM to LDS (each staging
// thread owns 8 fixed channels per K-stage), and the per-channel sums of z are reduced from the accumulators.
//
// MI355X mapping: a bf16 GEMM M = S (up to 401k rows), N = 128, K = C_in (64..1024) -- HBM-bound (AI ~ 100
// flop/B).  Workgroup = WM x 2 waves, 64*WM rows x 128 channels, each wave a 64 x 64 block as 2 x 2
// v_mfma_f32_32x32x16_bf16; K walked in 64-channel stages through double-buffered, XOR-swizzled LDS tiles
// (16-byte chunk ^ ((row >> 1) & 7): conflict-free ds_read_b128 fragments for 128-byte rows); the next stage's
// global loads are in flight while the current one is multiplied.  The output tile is staged through LDS and
// stored as whole 256-byte rows.  Statistics: per-tile (sum, M2) about a per-tile shift, merged by a one-wave-per
// -channel finalize with Chan's formula in double (deterministic, no atomics).
#include "common.h"

namespace {

typedef unsigned short bf16_t;
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

constexpr int BN = 128;  // bottleneck width (bn_size * growth_rate)
constexpr int BK = 64;   // input channels per stage (128-byte LDS rows)

__device__ __forceinline__ unsigned pack2(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xFFFF0000u); }
__device__ __forceinline__ float round_bf16(float v) { return bf_lo(pack2(v, 0.0f)); }

// a = relu(x*sc + sh) on one 16-byte chunk (8 channels)
__device__ __forceinline__ uint4 bn_relu_chunk(uint4 v, const float (&sc)[8], const float (&sh)[8]) {
  unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float lo = fmaxf(fmaf(bf_lo(w[i]), sc[2 * i], sh[2 * i]), 0.0f);
    const float hi = fmaxf(fmaf(bf_hi(w[i]), sc[2 * i + 1], sh[2 * i + 1]), 0.0f);
    w[i] = pack2(lo, hi);
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

template <int WM>
__global__ __launch_bounds__(128 * WM) void conv1x1_fwd_kernel(const bf16_t* __restrict__ x, long long ldx, long long S,
                                                               int K, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ rstd,
                                                               const bf16_t* __restrict__ W, bf16_t* __restrict__ z,
                                                               long long ldz, float2* __restrict__ partial, int nblk) {
  constexpr int BM = 64 * WM, NT = 128 * WM;
  constexpr int A_B = BM * 128, B_B = BN * 128, STAGE_B = A_B + B_B;
  constexpr int NB = (BN * 8) / NT;  // W chunks per thread and stage
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE_B + 2 * 1024 * 4];
  float* tab = reinterpret_cast<float*>(lds + 2 * STAGE_B);  // scale[K] then shift[K], K <= 1024

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  const long long row0 = (long long)blockIdx.x * BM;

  for (int k = tid; k < K; k += NT) {
    const float sc = gamma[k] * rstd[k];
    tab[k] = sc;
    tab[1024 + k] = fmaf(-mean[k], sc, beta[k]);
  }
  __syncthreads();

  // staging roles: A chunk column ca = tid & 7, rows (tid >> 3) + (NT/8)*i; B chunk column the same, rows (tid>>3) + (NT/8)*i
  const int cc = tid & 7, rr = tid >> 3;
  uint4 ra[4], rb[NB];
  auto load_stage = [&](int k0) {
    const int kc = k0 + cc * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long long r = row0 + rr + (NT / 8) * i;
      ra[i] = (r < S && kc < K) ? *reinterpret_cast<const uint4*>(x + r * ldx + kc) : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int n = rr + (NT / 8) * i;
      rb[i] = kc < K ? *reinterpret_cast<const uint4*>(W + (long long)n * K + kc) : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  auto store_stage = [&](int buf, int k0) {
    unsigned char* At = lds + buf * STAGE_B;
    unsigned char* Bt = At + A_B;
    const int kc = k0 + cc * 8;
    float sc[8], sh[8];
    if (kc < K) {
      const float4 s0 = *reinterpret_cast<const float4*>(tab + kc), s1 = *reinterpret_cast<const float4*>(tab + kc + 4);
      const float4 t0 = *reinterpret_cast<const float4*>(tab + 1024 + kc),
                   t1 = *reinterpret_cast<const float4*>(tab + 1024 + kc + 4);
      sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
      sh[0] = t0.x; sh[1] = t0.y; sh[2] = t0.z; sh[3] = t0.w; sh[4] = t1.x; sh[5] = t1.y; sh[6] = t1.z; sh[7] = t1.w;
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) sc[i] = sh[i] = 0.0f;   // K tail: relu(0*0 + 0) = 0
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = rr + (NT / 8) * i;
      const long long rg = row0 + r;
      uint4 v = bn_relu_chunk(ra[i], sc, sh);
      if (rg >= S) v = make_uint4(0u, 0u, 0u, 0u);
      *reinterpret_cast<uint4*>(At + r * 128 + ((cc ^ ((r >> 1) & 7)) << 4)) = v;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int n = rr + (NT / 8) * i;
      *reinterpret_cast<uint4*>(Bt + n * 128 + ((cc ^ ((n >> 1) & 7)) << 4)) = rb[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // fragment offsets: row (block base + l31), 16-byte chunk (2*kk + h) ^ ((l31 >> 1) & 7)
  const int sw = (l31 >> 1) & 7;
  int co[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) co[kk] = ((2 * kk + h) ^ sw) << 4;
  const int arow = (wm * 64 + l31) * 128, brow = (wn * 64 + l31) * 128;

  const int nst = (K + BK - 1) / BK;
  load_stage(0);
  store_stage(0, 0);
  __syncthreads();
  for (int st = 0; st < nst; ++st) {
    const int buf = st & 1;
    if (st + 1 < nst) load_stage((st + 1) * BK);
    const unsigned char* At = lds + buf * STAGE_B;
    const unsigned char* Bt = At + A_B;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(At + arow + i * 4096 + co[kk]);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(Bt + brow + j * 4096 + co[kk]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (st + 1 < nst) store_stage(buf ^ 1, (st + 1) * BK);
    __syncthreads();
  }

  // ---- epilogue.  acc[i][j][r]: row wm*64 + i*32 + (r&3) + 8*(r>>2) + 4*h, channel wn*64 + j*32 + l31
  float* red = reinterpret_cast<float*>(lds);                    // [WM][128] float2 (s1, s2)   (tiles are dead now)
  float* kshift = red + WM * 128 * 2;                            // [128]
  unsigned char* ot = lds + (WM * 128 * 2 + 128) * 4;           // output tile [BM][128] bf16, 256-byte rows
  const long long nvalid = min((long long)BM, S - row0);
  // round to bf16 (the statistics are those of the stored tensor)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = round_bf16(acc[i][j][r]);
  if (wm == 0 && h == 0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) kshift[wn * 64 + j * 32 + l31] = acc[0][j][0];   // tile row 0 (always < S)
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = wn * 64 + j * 32 + l31;
    const float ks = kshift[c];
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const float v = acc[i][j][r];
        if (row < nvalid) {
          const float dlt = v - ks;
          s1 += dlt;
          s2 = fmaf(dlt, dlt, s2);
        }
        reinterpret_cast<bf16_t*>(ot)[row * 128 + c] = (bf16_t)(__float_as_uint(v) >> 16);
      }
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (h == 0) {
      red[(wm * 128 + c) * 2] = s1;
      red[(wm * 128 + c) * 2 + 1] = s2;
    }
  }
  __syncthreads();
  if (tid < 128) {
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int w = 0; w < WM; ++w) {
      s1 += red[(w * 128 + tid) * 2];
      s2 += red[(w * 128 + tid) * 2 + 1];
    }
    const float n = (float)nvalid;
    // (sum, M2) of the tile: sum = s1 + n*k ; M2 = s2 - s1^2/n
    partial[(long long)tid * nblk + blockIdx.x] = make_float2(fmaf(n, kshift[tid], s1), s2 - s1 * s1 / n);
  }
  // whole 256-byte rows out: BM*16 chunks over NT threads
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int q = tid + NT * i;
    const int row = q >> 4, ch = q & 15;
    if (row < nvalid)
      *reinterpret_cast<uint4*>(z + (row0 + row) * ldz + ch * 8) = *reinterpret_cast<const uint4*>(ot + row * 256 + ch * 16);
  }
}

// one wave per channel: merge per-tile (sum, M2) pairs (Chan et al.), in double, fixed order
__global__ __launch_bounds__(256) void tile_stats_finalize_kernel(const float2* __restrict__ partial, int nblk, int C,
                                                                  long long S, int BM, float eps,
                                                                  float* __restrict__ mean, float* __restrict__ var,
                                                                  float* __restrict__ rstd) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;
  const float2* p = partial + (long long)c * nblk;
  double sum = 0.0, q = 0.0;   // q = sum_t (M2_t + sum_t^2 / n_t)
  for (int t = lane; t < nblk; t += 64) {
    const float2 v = p[t];
    const double nt = (double)min((long long)BM, S - (long long)t * BM);
    sum += (double)v.x;
    q += (double)v.y + (double)v.x * (double)v.x / nt;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sum += __shfl_xor(sum, o, 64);
    q += __shfl_xor(q, o, 64);
  }
  if (lane != 0) return;
  const double n = (double)S, m = sum / n;
  double v = (q - sum * sum / n) / n;
  if (v < 0.0) v = 0.0;
  mean[c] = (float)m;
  var[c] = (float)v;
  rstd[c] = (float)(1.0 / sqrt(v + (double)eps));
}

inline int pick_wm(long long S) { return S >= 131072 ? 4 : (S >= 32768 ? 2 : 1); }

}  // namespace

extern "C" int64_t mcl_dense_conv1x1_workspace_floats(int64_t S) {
  if (S <= 0) return -1;
  const int bm = 64 * pick_wm(S);
  return ((S + bm - 1) / bm) * BN * 2;
}

extern "C" int mcl_dense_conv1x1_fwd(const void* x, int64_t ldx, int64_t S, int32_t K, const float* gamma,
                                     const float* beta, const float* mean, const float* rstd, const void* W, void* z,
                                     int64_t ldz, float* workspace, float eps, float* zmean, float* zvar,
                                     float* zrstd, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!x || !gamma || !beta || !mean || !rstd || !W || !z || !workspace || !zmean || !zvar || !zrstd || S <= 0 || K <= 0)
    return MCL_EINVAL;
  if ((K % 8) || K > 1024 || (ldx % 8) || (ldz % 8) || ldz < BN || (reinterpret_cast<uintptr_t>(x) & 15u) ||
      (reinterpret_cast<uintptr_t>(W) & 15u) || (reinterpret_cast<uintptr_t>(z) & 15u))
    return MCL_EUNSUPPORTED;
  const int wm = pick_wm(S);
  const int bm = 64 * wm;
  const int nblk = (int)((S + bm - 1) / bm);
  hipStream_t st = mcl_stream(stream);
  float2* part = reinterpret_cast<float2*>(workspace);
#define MCL_LAUNCH(WMV)                                                                                              \
  hipLaunchKernelGGL(conv1x1_fwd_kernel<WMV>, dim3(nblk), dim3(128 * WMV), 0, st, (const bf16_t*)x, (long long)ldx, \
                     (long long)S, K, gamma, beta, mean, rstd, (const bf16_t*)W, (bf16_t*)z, (long long)ldz, part, nblk)
  if (wm == 4) MCL_LAUNCH(4);
  else if (wm == 2) MCL_LAUNCH(2);
  else MCL_LAUNCH(1);
#undef MCL_LAUNCH
  hipLaunchKernelGGL(tile_stats_finalize_kernel, dim3((BN + 3) / 4), dim3(256), 0, st, (const float2*)part, nblk, BN,
                     (long long)S, bm, eps, zmean, zvar, zrstd);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}
