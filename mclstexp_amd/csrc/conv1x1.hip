// 1x1-convolution weight gradient of the DenseNet bottleneck layers on channels-last bf16 activations:
//     dW[M][N] += sum_s dz[s][M]^T * a[s][N]          (M = out channels, N = in channels, s over B*H*W)
// A "TN" GEMM whose reduction dimension is the huge one (S up to 401k) and whose operands are both stored
// s-major, so neither matches the MFMA operand layout (8 consecutive k per lane).  gfx950's LDS transpose read
// (ds_read_b64_tr_b16) fixes that for free: tiles are staged exactly as they lie in HBM (16-byte rows chunks,
// coalesced) and each lane fetches 4 consecutive-s values of its own channel with one instruction.
//
// HBM-bound by construction (each operand element is read exactly once; AI ~ 128 flop/B << ridge): the grid
// splits S across ~768 workgroups which each stream their slab through a double-buffered LDS tile pair and
// add their fp32 128x128 partial straight into the parameter's .grad (the flat optimizer bucket) with
// hardware float atomics -- MIOpen's implicit-GEMM equivalent needs a zero-fill pass, the split-K kernel, a
// cast pass and (on the host side) a dtype cast + accumulate per weight.
// LDS rows are padded to 320 B so the four 16-lane groups of a transpose read hit 64 distinct banks.
#include "common.h"

namespace {

typedef unsigned short bf16_t;
typedef short v4s __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;        // rows (s) per LDS tile
constexpr int BT = 128;       // channel-tile width of both operands
constexpr int PITCH = 160;    // elements per LDS row (128 + 32 pad) = 320 B

__device__ __forceinline__ unsigned short f2bf_rne(float f) {
  unsigned u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

// One operand tile (BK x BT bf16) = 512 16-byte chunks; thread t owns chunks t and t+256 (same column chunk).
struct Stage {
  uint4 v[2];
};

template <bool PRO>
__device__ __forceinline__ void load_tile(Stage& st, const bf16_t* __restrict__ x, long long ld, long long s0,
                                          long long s_end, int c0, int C, int tid, const float* sc, const float* sh) {
  const int col = (tid & 15) * 8;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const long long s = s0 + (tid >> 4) + 16 * h;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (s < s_end && c0 + col < C) {
      v = *reinterpret_cast<const uint4*>(x + s * ld + c0 + col);
      if (PRO) {  // BatchNorm + ReLU applied on the fly: a = relu(x*sc + sh)
        unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float lo = __uint_as_float(w[i] << 16), hi = __uint_as_float(w[i] & 0xFFFF0000u);
          lo = fmaxf(fmaf(lo, sc[2 * i], sh[2 * i]), 0.0f);
          hi = fmaxf(fmaf(hi, sc[2 * i + 1], sh[2 * i + 1]), 0.0f);
          w[i] = (unsigned)f2bf_rne(lo) | ((unsigned)f2bf_rne(hi) << 16);
        }
        v = make_uint4(w[0], w[1], w[2], w[3]);
      }
    }
    st.v[h] = v;
  }
}

__device__ __forceinline__ void store_tile(const Stage& st, bf16_t* __restrict__ tile, int tid) {
  const int col = (tid & 15) * 8;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = (tid >> 4) + 16 * h;
    *reinterpret_cast<uint4*>(tile + row * PITCH + col) = st.v[h];
  }
}

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

template <bool PRO>
__global__ __launch_bounds__(256) void conv1x1_wrw_kernel(const bf16_t* __restrict__ dz, long long ldz,
                                                          const bf16_t* __restrict__ a, long long lda,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, float* __restrict__ dW,
                                                          long long lddw, long long S, int M, int N,
                                                          long long rows_per_wg) {
  __shared__ __attribute__((aligned(16))) bf16_t lds[2][2][BK * PITCH];   // [buffer][operand][tile]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int n0 = blockIdx.x * BT, m0 = blockIdx.y * BT;
  const long long s_begin = (long long)blockIdx.z * rows_per_wg;
  const long long s_end = min(S, s_begin + rows_per_wg);
  if (s_begin >= s_end) return;

  float sc[8], sh[8];
  if (PRO) {
    const int col = n0 + (tid & 15) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {   // BatchNorm folded to a = relu(x*sc + sh), recomputed from the layer input
      const bool ok = col + i < N;
      sc[i] = ok ? gamma[col + i] * rstd[col + i] : 0.0f;
      sh[i] = ok ? fmaf(-mean[col + i], sc[i], beta[col + i]) : 0.0f;
    }
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  Stage ra, rb;
  const int nt = (int)((s_end - s_begin + BK - 1) / BK);
  load_tile<false>(ra, dz, ldz, s_begin, s_end, m0, M, tid, nullptr, nullptr);
  load_tile<PRO>(rb, a, lda, s_begin, s_end, n0, N, tid, sc, sh);
  store_tile(ra, lds[0][0], tid);
  store_tile(rb, lds[0][1], tid);
  __syncthreads();

  const int half16 = 16 * ((lane >> 4) & 1);
  const int kg = 8 * (lane >> 5);
  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    if (t + 1 < nt) {
      const long long s0 = s_begin + (long long)(t + 1) * BK;
      load_tile<false>(ra, dz, ldz, s0, s_end, m0, M, tid, nullptr, nullptr);
      load_tile<PRO>(rb, a, lda, s0, s_end, n0, N, tid, sc, sh);
    }
    const bf16_t* tA = lds[cur][0];
    const bf16_t* tB = lds[cur][1];
#pragma unroll
    for (int kk = 0; kk < BK; kk += 16) {
      bf16x8 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = frag(tA, kk + kg, wm * 64 + i * 32 + half16, lane);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = frag(tB, kk + kg, wn * 64 + j * 32 + half16, lane);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (t + 1 < nt) {
      store_tile(ra, lds[cur ^ 1][0], tid);
      store_tile(rb, lds[cur ^ 1][1], tid);
    }
    __syncthreads();
  }

  // fp32 partial -> dW (+=) with hardware float atomics (no return value)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 64 + j * 32 + (lane & 31);
      if (n >= N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < M) unsafeAtomicAdd(dW + (long long)m * lddw + n, acc[i][j][r]);
      }
    }
}

}  // namespace

extern "C" int mcl_conv1x1_wrw_bf16(const void* dz, int64_t ldz, const void* a, int64_t lda, const float* gamma,
                                    const float* beta, const float* mean, const float* rstd, float* dW,
                                    int64_t lddw, int64_t S, int32_t M, int32_t N, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!dz || !a || !dW || S <= 0 || M <= 0 || N <= 0) return MCL_EINVAL;
  if ((M % 8) || (N % 8) || (ldz % 8) || (lda % 8) || (reinterpret_cast<uintptr_t>(dz) & 15u) ||
      (reinterpret_cast<uintptr_t>(a) & 15u))
    return MCL_EUNSUPPORTED;
  const bool pro = gamma || beta || mean || rstd;
  if (pro && !(gamma && beta && mean && rstd)) return MCL_EINVAL;
  const int tn = (N + BT - 1) / BT, tm = (M + BT - 1) / BT;
  // workgroup count: the 128x128 fp32 atomics per workgroup are throughput-bound (~0.3 T lane-atomics/s), so the
  // S split is kept as coarse as the streaming loop's latency hiding allows (measured: 384 / 256 beat 768)
  const long long target = S >= 200000 ? 384 : 256;
  long long ks = (target + tn * tm - 1) / (tn * tm);
  const long long max_ks = (S + 255) / 256;
  if (ks > max_ks) ks = max_ks;
  if (ks < 1) ks = 1;
  if (ks > 65535) ks = 65535;
  long long rows = (S + ks - 1) / ks;
  rows = (rows + BK - 1) / BK * BK;
  ks = (S + rows - 1) / rows;
  dim3 grid(tn, tm, (unsigned)ks);
  if (pro)
    hipLaunchKernelGGL(conv1x1_wrw_kernel<true>, grid, dim3(256), 0, mcl_stream(stream), (const bf16_t*)dz,
                       (long long)ldz, (const bf16_t*)a, (long long)lda, gamma, beta, mean, rstd, dW, (long long)lddw,
                       (long long)S, M, N, rows);
  else
    hipLaunchKernelGGL(conv1x1_wrw_kernel<false>, grid, dim3(256), 0, mcl_stream(stream), (const bf16_t*)dz,
                       (long long)ldz, (const bf16_t*)a, (long long)lda, gamma, beta, mean, rstd, dW, (long long)lddw,
                       (long long)S, M, N, rows);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}
