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
#include <stdlib.h>

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

// WM wave-rows of RW (64 or 32) rows each, two wave-columns of 64 channels: BM = WM*RW rows, 128*WM threads.  <2, 32> is
// the small-map form of <1, 64>: the same 64-row tile and LDS footprint but four waves instead of two per workgroup
// (two waves per SIMD instead of one at two workgroups per CU).
// DEPTH = K-stages requested ahead in registers.  1 on the large maps (several workgroups per CU cover each other's
// latency); on the 14 x 14 / 7 x 7 maps a CU holds one workgroup or none, the 64-channel stage multiplies in ~0.1 us and every
// stage exposed a whole memory round trip (1.3 us per stage measured: 21 us for K = 992 at S = 6272) -- four stages in flight
// there.
template <int WM, int RW = 64, int DEPTH = 1>
__global__ __launch_bounds__(128 * WM) void conv1x1_fwd_kernel(const bf16_t* __restrict__ x, long long ldx, long long S,
                                                               int K, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ rstd,
                                                               const bf16_t* __restrict__ W, bf16_t* __restrict__ z,
                                                               long long ldz, float2* __restrict__ partial, int nblk) {
  constexpr int BM = RW * WM, NT = 128 * WM;
  constexpr int RI = RW / 32;                // 32-row blocks per wave
  constexpr int NA = (BM * 8) / NT;          // A chunks per thread and stage
  constexpr int A_B = BM * 128, B_B = BN * 128, STAGE_B = A_B + B_B;
  constexpr int NB = (BN * 8) / NT;  // W chunks per thread and stage
  // DEPTH 0: ONE LDS stage (40 KB with the table: three workgroups per CU instead of two) and one register set; the
  // stage is overwritten between two barriers
  constexpr int NBUF = DEPTH == 0 ? 1 : 2, NSET = DEPTH == 0 ? 1 : DEPTH;
  __shared__ __attribute__((aligned(16))) unsigned char lds[NBUF * STAGE_B + 2 * 1024 * 4];
  float* tab = reinterpret_cast<float*>(lds + NBUF * STAGE_B);  // scale[K] then shift[K], K <= 1024

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  const long long row0 = (long long)blockIdx.x * BM;

  // (built AFTER the first stages' loads are in flight: its four dependent global loads per channel would otherwise sit in
  // front of them -- one memory round trip at the head of every workgroup)
  auto build_tab = [&]() {
    for (int k = tid; k < K; k += NT) {
      const float sc = gamma[k] * rstd[k];
      tab[k] = sc;
      tab[1024 + k] = fmaf(-mean[k], sc, beta[k]);
    }
    __syncthreads();
  };

  // staging roles: A chunk column ca = tid & 7, rows (tid >> 3) + (NT/8)*i; B chunk column the same, rows (tid>>3) + (NT/8)*i
  const int cc = tid & 7, rr = tid >> 3;
  // Buffer loads, never predicated: rows past S lie beyond the x descriptor and return zeros; past the last stage both
  // descriptors have size zero (zeros, no memory access).  (A branch or an exec mask around the loads makes the compiler
  // drain ALL outstanding loads -- s_waitcnt vmcnt(0) -- before it touches a register set again, which serialises the
  // stages in flight.)  Chunks past K inside a row read the neighbouring channels / the next weight row: store_stage zeroes
  // them.
  u32x4 rav[NSET][NA], rbv[NSET][NB];
  const unsigned xbytes = (unsigned)((((long long)S - 1) * ldx + K) * 2), wbytes = (unsigned)((long long)BN * K * 2);
  unsigned xoff[NA], woff[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const long long r = row0 + rr + (NT / 8) * i;
    xoff[i] = r < S ? (unsigned)((r * ldx + cc * 8) * 2) : 0xFFFFF000u;
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) woff[i] = (unsigned)(((rr + (NT / 8) * i) * K + cc * 8) * 2);
  auto load_stage = [&](u32x4 (&ra)[NA], u32x4 (&rb)[NB], int k0) {
    const bool live = k0 < K;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(x), 0, live ? xbytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(W), 0, live ? wbytes : 0u, 0x00020000);
#pragma unroll
    for (int i = 0; i < NA; ++i) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, xoff[i] + (unsigned)k0 * 2u, 0, 0);
#pragma unroll
    for (int i = 0; i < NB; ++i) rb[i] = __builtin_amdgcn_raw_buffer_load_b128(wr, woff[i] + (unsigned)k0 * 2u, 0, 0);
  };
  auto store_stage = [&](const u32x4 (&ra)[NA], const u32x4 (&rb)[NB], int buf, int k0) {
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
    for (int i = 0; i < NA; ++i) {
      const int r = rr + (NT / 8) * i;
      const long long rg = row0 + r;
      uint4 v = bn_relu_chunk(make_uint4(ra[i][0], ra[i][1], ra[i][2], ra[i][3]), sc, sh);
      if (rg >= S) v = make_uint4(0u, 0u, 0u, 0u);
      *reinterpret_cast<uint4*>(At + r * 128 + ((cc ^ ((r >> 1) & 7)) << 4)) = v;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int n = rr + (NT / 8) * i;
      *reinterpret_cast<uint4*>(Bt + n * 128 + ((cc ^ ((n >> 1) & 7)) << 4)) =
          kc < K ? make_uint4(rb[i][0], rb[i][1], rb[i][2], rb[i][3]) : make_uint4(0u, 0u, 0u, 0u);
    }
  };

  f32x16 acc[RI][2];
#pragma unroll
  for (int i = 0; i < RI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // fragment offsets: row (block base + l31), 16-byte chunk (2*kk + h) ^ ((l31 >> 1) & 7)
  const int sw = (l31 >> 1) & 7;
  int co[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) co[kk] = ((2 * kk + h) ^ sw) << 4;
  const int arow = (wm * RW + l31) * 128, brow = (wn * 64 + l31) * 128;

  const int nst = (K + BK - 1) / BK;
  auto multiply = [&](int buf) {
    const unsigned char* At = lds + buf * STAGE_B;
    const unsigned char* Bt = At + A_B;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8 fa[RI], fb[2];
#pragma unroll
      for (int i = 0; i < RI; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(At + arow + i * 4096 + co[kk]);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(Bt + brow + j * 4096 + co[kk]);
#pragma unroll
      for (int i = 0; i < RI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
  };
  if (DEPTH == 0) {
    load_stage(rav[0], rbv[0], 0);
    __builtin_amdgcn_sched_barrier(0);
    build_tab();
    store_stage(rav[0], rbv[0], 0, 0);
    __syncthreads();
    for (int st = 0; st < nst; ++st) {
      load_stage(rav[0], rbv[0], (st + 1) * BK);
      __builtin_amdgcn_sched_barrier(0);
      multiply(0);
      __syncthreads();                                   // everybody has read the stage
      if (st + 1 < nst) store_stage(rav[0], rbv[0], 0, (st + 1) * BK);
      __syncthreads();
    }
  } else if (DEPTH == 1) {
    load_stage(rav[0], rbv[0], 0);
    __builtin_amdgcn_sched_barrier(0);
    build_tab();
    store_stage(rav[0], rbv[0], 0, 0);
    __syncthreads();
    for (int st = 0; st < nst; ++st) {
      const int buf = st & 1;
      load_stage(rav[0], rbv[0], (st + 1) * BK);
      __builtin_amdgcn_sched_barrier(0);
      multiply(buf);
      if (st + 1 < nst) store_stage(rav[0], rbv[0], buf ^ 1, (st + 1) * BK);
      __syncthreads();
    }
  } else {
    // register set d holds stage st0 + d; once a stage is in LDS its set takes stage + DEPTH (loads past K are masked off
    // lane by lane: no branch, no memory access).  The loads are pinned at the top of the stage.
    static_assert(DEPTH <= 1 || (DEPTH % 2) == 0, "LDS parity must follow the register set");
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load_stage(rav[d], rbv[d], d * BK);
    __builtin_amdgcn_sched_barrier(0);
    build_tab();
    store_stage(rav[0], rbv[0], 0, 0);
    __syncthreads();
    for (int st0 = 0; st0 < nst; st0 += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const int st = st0 + d;
        if (st < nst) {
          load_stage(rav[d], rbv[d], (st + DEPTH) * BK);
          __builtin_amdgcn_sched_barrier(0);
          multiply(d & 1);
          if (st + 1 < nst) store_stage(rav[(d + 1) % DEPTH], rbv[(d + 1) % DEPTH], (d + 1) & 1, (st + 1) * BK);
          __syncthreads();
        }
      }
    }
  }

  // ---- epilogue.  acc[i][j][r]: row wm*RW + i*32 + (r&3) + 8*(r>>2) + 4*h, channel wn*64 + j*32 + l31
  float* red = reinterpret_cast<float*>(lds);                    // [WM][128] float2 (s1, s2)   (tiles are dead now)
  float* kshift = red + WM * 128 * 2;                            // [128]
  unsigned char* ot = lds + (WM * 128 * 2 + 128) * 4;           // output tile [BM][128] bf16, 256-byte rows
  const long long nvalid = min((long long)BM, S - row0);
  // round to bf16 (the statistics are those of the stored tensor)
#pragma unroll
  for (int i = 0; i < RI; ++i)
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
    for (int i = 0; i < RI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * RW + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
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
  for (int i = 0; i < (BM * 16) / NT; ++i) {
    const int q = tid + NT * i;
    const int row = q >> 4, ch = q & 15;
    if (row < nvalid)
      *reinterpret_cast<uint4*>(z + (row0 + row) * ldz + ch * 8) = *reinterpret_cast<const uint4*>(ot + row * 256 + ch * 16);
  }
}

// one WORKGROUP per channel: merge per-tile (sum, M2) pairs (Chan et al.), in double, fixed order.  (A wave per
// channel walks up to 3136 partials in 49 dependent steps; the kernel sits on the critical path twice per layer.)
__global__ __launch_bounds__(256) void tile_stats_finalize_kernel(const float2* __restrict__ partial, int nblk, int C,
                                                                  long long S, int BM, float eps,
                                                                  float* __restrict__ mean, float* __restrict__ var,
                                                                  float* __restrict__ rstd) {
  __shared__ double red[2][4];
  const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float2* p = partial + (long long)c * nblk;
  double sum = 0.0, q = 0.0;   // q = sum_t (M2_t + sum_t^2 / n_t)
  // A thread's partials t = tid, tid + 256, ... are fetched in batches of eight independent loads (clamped index, the
  // surplus zeroed after the load) and added in the same ascending order: one load per loop trip is one memory round
  // trip per trip -- 12 in series on the 56 x 56 maps, most of a 5 us kernel that sits on the forward chain twice per
  // layer.  Full tiles divide by BM (a power of two: multiplying by 1/BM is the same value); the ragged last tile, the
  // last element of its thread's sequence, keeps the true division.
  const int nfull = nblk - 1, tlast = nfull & 255;
  const float2 vlast = p[nfull];
  const double inv_bm = 1.0 / (double)BM;
  constexpr int U = 8;
  for (int t0 = threadIdx.x; t0 < nfull; t0 += 256 * U) {
    float2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = p[min(t0 + 256 * u, nfull)];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (t0 + 256 * u >= nfull) v[u] = make_float2(0.0f, 0.0f);
      sum += (double)v[u].x;
      q += (double)v[u].y + (double)v[u].x * (double)v[u].x * inv_bm;
    }
  }
  if ((int)threadIdx.x == tlast) {
    const double nt = (double)(S - (long long)nfull * BM);
    sum += (double)vlast.x;
    q += (double)vlast.y + (double)vlast.x * (double)vlast.x / nt;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sum += __shfl_xor(sum, o, 64);
    q += __shfl_xor(q, o, 64);
  }
  if (lane == 0) {
    red[0][wave] = sum;
    red[1][wave] = q;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  sum = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
  q = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  const double n = (double)S, m = sum / n;
  double v = (q - sum * sum / n) / n;
  if (v < 0.0) v = 0.0;
  mean[c] = (float)m;
  var[c] = (float)v;
  rstd[c] = (float)(1.0 / sqrt(v + (double)eps));
}

// 128-row tiles (WM = 2) also on the 56 x 56 maps: the 256-row tile (104 KB, ONE
// workgroup per CU, nothing overlaps its load / multiply / store phases) is faster alone but 0.1 ms/step slower in the step
// (13.07 / 13.09 vs 12.94 / 13.01 ms, interleaved A/B)
inline int pick_wm(long long S) { return S >= 32768 ? 2 : 1; }

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
  if (!x || !gamma || !beta || !mean || !rstd || !W || !z || !workspace || S <= 0 || K <= 0) return MCL_EINVAL;
  const bool want_stats = zmean || zvar || zrstd;   // all three or none (inference: BN2 uses running statistics)
  if (want_stats && (!zmean || !zvar || !zrstd)) return MCL_EINVAL;
  if ((K % 8) || K > 1024 || (ldx % 8) || (ldz % 8) || ldz < BN || S * ldx * 2 >= 0xFFFFF000LL /* 32-bit buffer offsets */ ||
      (reinterpret_cast<uintptr_t>(x) & 15u) ||
      (reinterpret_cast<uintptr_t>(W) & 15u) || (reinterpret_cast<uintptr_t>(z) & 15u))
    return MCL_EUNSUPPORTED;
  const int wm = pick_wm(S);
  const int bm = 64 * wm;      // wm == 1 runs as <2, 32>: the same 64-row tile on four waves
  const int nblk = (int)((S + bm - 1) / bm);
  hipStream_t st = mcl_stream(stream);
  float2* part = reinterpret_cast<float2*>(workspace);
#define MCL_LAUNCH(WMV, RWV, DPV)                                                                                    \
  hipLaunchKernelGGL((conv1x1_fwd_kernel<WMV, RWV, DPV>), dim3(nblk), dim3(128 * WMV), 0, st, (const bf16_t*)x,      \
                     (long long)ldx, (long long)S, K, gamma, beta, mean, rstd, (const bf16_t*)W, (bf16_t*)z,          \
                     (long long)ldz, part, nblk)
  // 56 x 56 / 28 x 28 maps: ONE LDS stage and one register set (40 KB, 166 registers: three workgroups per CU).  In-kernel
  // timestamps of the two-stage form showed a workgroup loading nothing for half of its life (prologue 2.7 us, K loop 5.7
  // at the HBM rate, statistics + store 3.4): a third resident workgroup fills more of that than the second stage hid.
  // r03 same box: 102 -> 88 us (C = 224), 57 -> 48 (C = 64) alone; 11.77 / 11.80 -> 11.65 / 11.70 ms/step.  The small maps run
  // the same 64-row tile on four waves with four K-stages requested ahead (the one-stage form measured 11.90 / 11.94 vs 11.88 /
  // 11.85 there; the two-wave form and shallower prefetch lost in round 3).
  if (wm == 2) MCL_LAUNCH(2, 64, 0);
  else MCL_LAUNCH(2, 32, 4);
#undef MCL_LAUNCH
  if (want_stats)
    hipLaunchKernelGGL(tile_stats_finalize_kernel, dim3(BN), dim3(256), 0, st, (const float2*)part, nblk, BN,
                       (long long)S, bm, eps, zmean, zvar, zrstd);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

// =====================================================================================================================
// DenseNet growth 3x3 convolution with BatchNorm folded in on both sides (torchvision _DenseLayer: norm2 -> relu2 ->
// conv2 (3x3, pad 1, 128 -> 32), followed by the statistics the next layers' norm1 need), writing straight into
// the block's concat buffer:
//
//     y[p][co] = sum_{ky,kx,ci} relu(z[p + (ky-1)*W + (kx-1)][ci]*scale[ci] + shift[ci]) * W2[co][ky][kx][ci]
//
// over flattened NHWC pixels p, taps that leave the image contributing zero.  Implicit GEMM M = S pixels, N = 32,
// K = 9*128.  One workgroup (4 waves, two per CU) owns 128 consecutive pixels: it stages the contiguous pixel
// range [p0 - W - 1, p0 + 128 + W + 1) of z ONCE into an XOR-swizzled LDS slab with BN+ReLU applied on the way
// (each staging thread owns 8 fixed channels), so `a2 = relu(bn2(z))` never exists in HBM and all nine taps are LDS
// row offsets.  N = 32 gives a B fragment no reuse across output columns, so the K dimension is split over the
// four waves instead (wave w owns k-steps 18w..18w+17 of 72 and keeps its 18 weight fragments in registers for the
// whole tile): every MFMA costs one ds_read_b128.  Invalid taps select the address of an all-zero LDS row.  The
// four K-partials meet in LDS; the tile's (sum, M2) per channel go to the same finalize as the 1x1 kernel.
constexpr int C3_IN = 128, C3_OUT = 32, T3 = 128;   // channels in / out, pixels per tile

// Stage slab rows [0, nrow) of 256 B (pixel p0 - (W+1) + j) with BN+ReLU, 16 rows per pass over the workgroup, in
// batches of NB independent global loads per thread (a load -> transform -> store loop would serialise one full
// memory latency per pass: the slab is 10-16 passes).
// independent global loads in flight per thread and batch (measured at cfg2: 8 beats 4 by 2 %, 16 loses 3 %)
#ifndef MCL_SLAB_NB
#define MCL_SLAB_NB 8
#endif
constexpr int SLAB_NB = MCL_SLAB_NB;
template <int NB>
__device__ __forceinline__ void stage_slab(unsigned char* lds, const bf16_t* __restrict__ z, int p0, long long S,
                                           int W, int nrow, int tid, const float (&sc)[8], const float (&sh)[8]) {
  const int cc = tid & 15;
  for (int jb = tid >> 4; jb < nrow; jb += 16 * NB) {
    uint4 v[NB];
    bool ok[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int j = jb + 16 * i;
      const int p = p0 - (W + 1) + j;
      ok[i] = j < nrow && p >= 0 && p < (int)S;
      v[i] = ok[i] ? *reinterpret_cast<const uint4*>(z + (long long)p * C3_IN + cc * 8) : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int j = jb + 16 * i;
      if (j < nrow)
        *reinterpret_cast<uint4*>(lds + j * 256 + ((cc ^ (j & 15)) << 4)) =
            ok[i] ? bn_relu_chunk(v[i], sc, sh) : make_uint4(0u, 0u, 0u, 0u);
    }
  }
}

__global__ __launch_bounds__(256, 2) void conv3x3_fwd_kernel(const bf16_t* __restrict__ z, long long S, int H, int W,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ rstd,
                                                             const bf16_t* __restrict__ W2, bf16_t* __restrict__ out,
                                                             long long ldo, float2* __restrict__ partial, int ntile) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int nrow = T3 + 2 * W + 2;                 // slab rows; slab row j <-> pixel p0 - (W+1) + j
  unsigned char* zero_row = lds + nrow * 256;      // 256 B of zeros
  const int Si = (int)S;

  // this wave's 18 weight fragments: k-step kidx = 18*wave + i -> tap = kidx >> 3, channels 16*(kidx & 7) + 8h ..
  bf16x8 breg[18];
#pragma unroll
  for (int i = 0; i < 18; ++i) {
    const int kidx = 18 * wave + i;
    breg[i] = *reinterpret_cast<const bf16x8*>(W2 + ((long long)l31 * 9 + (kidx >> 3)) * C3_IN + 16 * (kidx & 7) + 8 * h);
  }
  float sc[8], sh[8];
  {
    const int cc = tid & 15;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = cc * 8 + i;
      sc[i] = gamma[c] * rstd[c];
      sh[i] = fmaf(-mean[c], sc[i], beta[c]);
    }
  }
  // slab byte offset of this lane's fragment for k-step i in pixel block 0 (block mb adds 32 rows = 8192 B, which
  // leaves the swizzle term (row & 15) untouched); tap of each k-step for the validity test
  int abase[18], tapk[18];
#pragma unroll
  for (int i = 0; i < 18; ++i) {
    const int kidx = 18 * wave + i, tap = kidx >> 3, kk = kidx & 7;
    const int row = l31 + (tap / 3) * W + (tap % 3);
    abase[i] = row * 256 + (((2 * kk + h) ^ (row & 15)) << 4);
    tapk[i] = tap;
  }
  // persistent over tiles: the weight fragments and BN coefficients are loaded once per workgroup
  for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
  const int p0 = tile * T3;
  __syncthreads();   // the previous tile's epilogue scratch is dead

  // ---- stage the slab with BN+ReLU: thread owns chunk column tid & 15 (8 channels) of rows (tid >> 4) + 16*i
  stage_slab<SLAB_NB>(lds, z, p0, S, W, nrow, tid, sc, sh);
  if (tid < 16) *reinterpret_cast<uint4*>(zero_row + tid * 16) = make_uint4(0u, 0u, 0u, 0u);

  // per pixel block mb: slab row of tap (0,0) and the 9-bit tap validity of this lane's pixel
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
          const int yy = y + ky - 1, xx = x + kx - 1;
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

  const int zero_off = nrow * 256;
#pragma unroll
  for (int i = 0; i < 18; ++i) {
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      const int off = ((vmask[mb] >> tapk[i]) & 1u) ? abase[i] + mb * 8192 : zero_off;
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(lds + off);
      acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, breg[i], acc[mb], 0, 0, 0);
    }
  }
  __syncthreads();   // slab dead: reuse it for the K-partials  red[wave][pixel][co] fp32
  float* red = reinterpret_cast<float*>(lds);
#pragma unroll
  for (int mb = 0; mb < 4; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int px = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      red[(wave * T3 + px) * C3_OUT + l31] = acc[mb][r];
    }
  __syncthreads();
  // thread -> pixel tid >> 1, 16 channels (tid & 1) * 16 ..
  const int px = tid >> 1, c0 = (tid & 1) * 16;
  const long long p = (long long)p0 + px;
  float v[16];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float4 s = *reinterpret_cast<const float4*>(red + px * C3_OUT + c0 + 4 * q);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float4 t = *reinterpret_cast<const float4*>(red + (w * T3 + px) * C3_OUT + c0 + 4 * q);
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    v[4 * q] = s.x; v[4 * q + 1] = s.y; v[4 * q + 2] = s.z; v[4 * q + 3] = s.w;
  }
  unsigned pk[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) pk[q] = pack2(v[2 * q], v[2 * q + 1]);
  if (p < S) {
    *reinterpret_cast<uint4*>(out + p * ldo + c0) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    *reinterpret_cast<uint4*>(out + p * ldo + c0 + 8) = make_uint4(pk[4], pk[5], pk[6], pk[7]);
  }
  // ---- statistics of the bf16-rounded tile: rounded values -> LDS [px][32], then (channel, 16-pixel group) partials
  __syncthreads();
  float* rv = reinterpret_cast<float*>(lds);                 // [128][32]
  float* grp = rv + T3 * C3_OUT;                             // [8][32] float2
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    rv[px * C3_OUT + c0 + 2 * q] = bf_lo(pk[q]);
    rv[px * C3_OUT + c0 + 2 * q + 1] = bf_hi(pk[q]);
  }
  __syncthreads();
  {
    const int c = tid & 31, g = tid >> 5;
    const float ks = rv[c];                                  // shift: pixel 0 of the tile (always < S)
    const int nvalid = min(T3, Si - p0);
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int pp = g * 16 + q;
      if (pp < nvalid) {
        const float dlt = rv[pp * C3_OUT + c] - ks;
        s1 += dlt;
        s2 = fmaf(dlt, dlt, s2);
      }
    }
    grp[(g * 32 + c) * 2] = s1;
    grp[(g * 32 + c) * 2 + 1] = s2;
    __syncthreads();
    if (tid < 32) {
      float a = 0.0f, b = 0.0f;
#pragma unroll
      for (int gg = 0; gg < 8; ++gg) {
        a += grp[(gg * 32 + tid) * 2];
        b += grp[(gg * 32 + tid) * 2 + 1];
      }
      const float n = (float)nvalid;
      partial[(long long)tid * ntile + tile] = make_float2(fmaf(n, rv[tid], a), b - a * a / n);
    }
  }
  }   // tile loop
}

extern "C" int64_t mcl_dense_conv3x3_workspace_floats(int64_t S) {
  if (S <= 0) return -1;
  const int64_t flat = ((S + T3 - 1) / T3) * C3_OUT * 2, rows = mcl_conv3x3_rows_workspace_floats(S);
  return flat > rows ? flat : rows;
}

extern "C" int mcl_dense_conv3x3_fwd(const void* z, int64_t S, int32_t H, int32_t W, const float* gamma,
                                     const float* beta, const float* mean, const float* rstd, const void* W2, void* out,
                                     int64_t ldo, float* workspace, float eps, float* ymean, float* yvar, float* yrstd,
                                     mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!z || !gamma || !beta || !mean || !rstd || !W2 || !out || !workspace || S <= 0 || H <= 0 || W <= 0)
    return MCL_EINVAL;
  const bool want_stats = ymean || yvar || yrstd;   // all three or none
  if (want_stats && (!ymean || !yvar || !yrstd)) return MCL_EINVAL;
  if ((S % ((int64_t)H * W)) || S > 0x7fff0000LL || W > 150 || (ldo % 8) || ldo < C3_OUT || (reinterpret_cast<uintptr_t>(z) & 15u) ||
      (reinterpret_cast<uintptr_t>(W2) & 15u) || (reinterpret_cast<uintptr_t>(out) & 15u))
    return MCL_EUNSUPPORTED;
  if (mcl_conv3x3_rows_applicable(S, H, W)) {       // the large maps: row-walking form (csrc/conv3x3_rows.hip)
    const int rc = mcl_launch_conv3x3_fwd_rows(z, S, H, W, gamma, beta, mean, rstd, W2, out, ldo, workspace, eps, ymean,
                                               yvar, yrstd, mcl_stream(stream));
    if (rc != 0) return rc;
    MCL_CHECK_LAUNCH();
    return MCL_OK;
  }
  const int ntile = (int)((S + T3 - 1) / T3);
  // slab + zero row; the epilogue reuses it for 4 x 128 x 32 fp32 partials (64 KiB) / the statistics scratch
  size_t lds_bytes = (size_t)(T3 + 2 * W + 2) * 256 + 256;
  if (lds_bytes < 4 * T3 * C3_OUT * 4) lds_bytes = 4 * T3 * C3_OUT * 4;
  hipStream_t st = mcl_stream(stream);
  static mcl_device_once attr_once;
  if (auto attr_guard = attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_fwd_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024);
  }
  float2* part = reinterpret_cast<float2*>(workspace);
  hipLaunchKernelGGL(conv3x3_fwd_kernel, dim3(ntile < 512 ? ntile : 512), dim3(256), lds_bytes, st, (const bf16_t*)z, (long long)S, H, W,
                     gamma, beta, mean, rstd, (const bf16_t*)W2, (bf16_t*)out, (long long)ldo, part, ntile);
  if (want_stats)
    hipLaunchKernelGGL(tile_stats_finalize_kernel, dim3(C3_OUT), dim3(256), 0, st, (const float2*)part, ntile,
                       C3_OUT, (long long)S, T3, eps, ymean, yvar, yrstd);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

// =====================================================================================================================
// Weight gradient of the growth 3x3 convolution, with norm2+relu2 recomputed on the fly, deterministic:
//
//     dW2[co][ky][kx][ci] (+)= sum_p dy[p][co] * relu(bn2(z))[p + (ky-1)*W + (kx-1)][ci]        (taps inside the image)
//
// A "TN" GEMM M = 32, N = 9*128, K = S pixels, both operands pixel-major -> both MFMA fragments come from transposing
// LDS reads (ds_read_b64_tr_b16).  "One kernel row per workgroup":
//
// A workgroup owns ONE kernel row ky (3 taps, 384 of the 1152 columns) of a strided set of 128-pixel tiles:
//   * the z rows a tile needs for its three kx taps are the 130 consecutive pixels [p0 + (ky-1)W - 1, +130): no halo;
//   * border handling moves to the OTHER operand: dy is staged three times (8 KB each), each copy zeroed where the
//     pixel's (ky, kx) neighbour leaves the image -- the MFMA loop has no masks and no address selects at all;
//   * the next tile's raw rows are fetched into registers while the current tile is multiplied;
//   * the three ky workgroups of a pixel group run next to each other on one XCD (z is re-read from that L2) and write
//     disjoint column ranges of ONE 32 x 1152 partial: a third of the partial traffic per pixel group.
// Wave w owns input channels 32w..32w+31 of all three taps (48 accumulator registers).
namespace {

// z-tile swizzle: a transposing read takes 4 consecutive pixel rows x 32 bytes per 16 lanes, so consecutive rows must
// land in different 64-byte bank windows: chunk bits 2-3 ^= row & 3, bits 0-1 ^= (row >> 2) & 3 (a period-16 pattern:
// read addresses still advance by 4096 B per 16-pixel k-step).  The slab kernels' chunk ^ (row & 15) only swaps the two
// chunks of a pair between rows 2m and 2m+1: a 2-way conflict on every read (SQ_LDS_BANK_CONFLICT was 76 % of the
// kernel's busy cycles, profiles/r02_sq_counters_step_kernels_before_swizzle_fix.txt; 0 after).
__device__ __forceinline__ int w3k_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

constexpr int W3K_ROWS = T3 + 2;                         // staged z rows per tile
constexpr int W3K_LDS = W3K_ROWS * 256 + 3 * T3 * 64;    // z tile + three masked dy tiles = 57,856 B: two per CU

__global__ __launch_bounds__(256, 2) void conv3x3_wrw_ky_kernel(const bf16_t* __restrict__ dy, long long lddy,
                                                                const bf16_t* __restrict__ z, long long S, int H, int W,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, int ntile, int G,
                                                                float* __restrict__ wpart) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[W3K_LDS];
  typedef short v4s __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  // XCD-aware decode (block id % 8 = XCD): the three kernel rows of one pixel group sit in consecutive slots of one XCD
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int ky = slot % 3, g = (slot / 3) * 8 + xcd;
  if (g >= G) return;
  unsigned char* zt = lds;                               // [130][128 ch] bf16, 256-byte rows, chunk ^ (row & 15)
  unsigned char* dyt = lds + W3K_ROWS * 256;             // [3 kx][128][32] bf16, 64-byte rows, chunk ^ ((row >> 1) & 3)
  const int Si = (int)S;

  const int cc = tid & 15, rz = tid >> 4;                // z staging: chunk, first row (rows rz + 16 i, i < 9)
  const int rd = tid >> 1, cd = (tid & 1) * 2;           // dy staging: pixel row, first of two chunks
  float sc[8], sh[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = cc * 8 + i;
    sc[i] = gamma[c] * rstd[c];
    sh[i] = fmaf(-mean[c], sc[i], beta[c]);
  }
  f32x16 acc[3];
#pragma unroll
  for (int b = 0; b < 3; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.0f;

  // transposing-read lane geometry (as in conv3x3_wrw_kernel): lane i = lane & 15 supplies pixel row (i >> 2) [+4 for the
  // second read] of the 8-pixel k-group 8h, 4 channels at (i & 3)*4 of the 16-channel half (lane >> 4) & 1
  const int i15 = lane & 15, q = i15 >> 2, jj = i15 & 3;
  const int half16 = 16 * ((lane >> 4) & 1);
  const int byte = (jj & 1) * 8;
  int bz_lo[3], bz_hi[3];                                // z-tile read offsets per kx (+ 4096 per 16-pixel k-step)
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    const int chunk = wave * 4 + ((half16 + jj * 4) >> 3);
    const int r0 = 8 * h + q + kx, r1 = r0 + 4;
    bz_lo[kx] = r0 * 256 + ((chunk ^ w3k_swz(r0)) << 4) + byte;
    bz_hi[kx] = r1 * 256 + ((chunk ^ w3k_swz(r1)) << 4) + byte;
  }
  const int dchunk = (half16 + jj * 4) >> 3;

  // raw rows of one tile -> registers (zeros outside [0, S) and for tile >= ntile)
  auto fetch = [&](uint4 (&zv)[9], uint4 (&dv)[2], int tile) {
    const int p0 = tile * T3;
    const int qs = p0 + (ky - 1) * W - 1;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int j = rz + 16 * i;
      const int p = qs + j;
      zv[i] = (tile < ntile && j < W3K_ROWS && p >= 0 && p < Si)
                  ? *reinterpret_cast<const uint4*>(z + (long long)p * C3_IN + cc * 8)
                  : make_uint4(0u, 0u, 0u, 0u);
    }
    const int p = p0 + rd;
#pragma unroll
    for (int u = 0; u < 2; ++u)
      dv[u] = (tile < ntile && p < Si) ? *reinterpret_cast<const uint4*>(dy + (long long)p * lddy + (cd + u) * 8)
                                       : make_uint4(0u, 0u, 0u, 0u);
  };
  // registers -> LDS: BN+ReLU on z, three border-masked copies of dy
  auto stage = [&](const uint4 (&zv)[9], const uint4 (&dv)[2], int tile) {
    const int p0 = tile * T3;
    const int qs = p0 + (ky - 1) * W - 1;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int j = rz + 16 * i;
      if (j < W3K_ROWS) {
        const int p = qs + j;
        const bool ok = p >= 0 && p < Si;                // (rows outside the tensor: exact zeros, not relu(shift))
        *reinterpret_cast<uint4*>(zt + j * 256 + ((cc ^ w3k_swz(j)) << 4)) =
            ok ? bn_relu_chunk(zv[i], sc, sh) : make_uint4(0u, 0u, 0u, 0u);
      }
    }
    const int p = p0 + rd;
    const int x = p % W, y = (p / W) % H;
    const bool vy = (unsigned)(y + ky - 1) < (unsigned)H && p < Si;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const bool v = vy && (unsigned)(x + kx - 1) < (unsigned)W;
#pragma unroll
      for (int u = 0; u < 2; ++u)
        *reinterpret_cast<uint4*>(dyt + kx * (T3 * 64) + rd * 64 + (((cd + u) ^ ((rd >> 1) & 3)) << 4)) =
            v ? dv[u] : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  auto multiply = [&]() {
#pragma unroll 2
    for (int ks = 0; ks < 8; ++ks) {                     // 16 pixels per step
      const int r_lo = ks * 16 + 8 * h + q, r_hi = r_lo + 4;
      const int da_lo = r_lo * 64 + ((dchunk ^ ((r_lo >> 1) & 3)) << 4) + byte;
      const int da_hi = r_hi * 64 + ((dchunk ^ ((r_hi >> 1) & 3)) << 4) + byte;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const v4s alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (v4s __attribute__((address_space(3)))*)(dyt + kx * (T3 * 64) + da_lo));
        const v4s ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (v4s __attribute__((address_space(3)))*)(dyt + kx * (T3 * 64) + da_hi));
        const v4s blo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (v4s __attribute__((address_space(3)))*)(zt + bz_lo[kx] + ks * 4096));
        const v4s bhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (v4s __attribute__((address_space(3)))*)(zt + bz_hi[kx] + ks * 4096));
        bf16x8 fa, fb;
        fa[0] = alo[0]; fa[1] = alo[1]; fa[2] = alo[2]; fa[3] = alo[3];
        fa[4] = ahi[0]; fa[5] = ahi[1]; fa[6] = ahi[2]; fa[7] = ahi[3];
        fb[0] = blo[0]; fb[1] = blo[1]; fb[2] = blo[2]; fb[3] = blo[3];
        fb[4] = bhi[0]; fb[5] = bhi[1]; fb[6] = bhi[2]; fb[7] = bhi[3];
        acc[kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[kx], 0, 0, 0);
      }
    }
  };

  // A pixel group owns CONSECUTIVE tiles: kernel row ky of tile t+1 re-reads most of what row ky+1 read for tile t, and
  // both workgroups sit on this XCD's L2.  Global loads run TWO tiles ahead in two register sets: one tile of work
  // (~2 us) does not cover a miss to HBM, and the loop was latency-bound at ~6 us per tile with one set.
  const int per = (ntile + G - 1) / G;
  const int t_end = min(ntile, (g + 1) * per);
  uint4 zvA[9], dvA[2], zvB[9], dvB[2];
  int tile = g * per;
  fetch(zvA, dvA, tile < t_end ? tile : ntile);
  fetch(zvB, dvB, tile + 1 < t_end ? tile + 1 : ntile);
  for (; tile < t_end; tile += 2) {
    __syncthreads();                                     // the previous tile's MFMAs are done with the LDS tiles
    stage(zvA, dvA, tile);
    __syncthreads();
    fetch(zvA, dvA, tile + 2 < t_end ? tile + 2 : ntile);
    multiply();
    if (tile + 1 < t_end) {
      __syncthreads();
      stage(zvB, dvB, tile + 1);
      __syncthreads();
      fetch(zvB, dvB, tile + 3 < t_end ? tile + 3 : ntile);
      multiply();
    }
  }
  // acc[kx][r]: co = (r&3) + 8*(r>>2) + 4*h, column (3 ky + kx)*128 + 32 wave + l31 of the 1152-wide row of partial g
  float* dst = wpart + (long long)g * (C3_OUT * 9 * C3_IN);
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    const int n = (3 * ky + kx) * C3_IN + 32 * wave + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = (r & 3) + 8 * (r >> 2) + 4 * h;
      dst[co * (9 * C3_IN) + n] = acc[kx][r];
    }
  }
}

}  // namespace

// pixel groups (= partials) of the kernel-row form: three workgroups each, two workgroups per CU
static inline int wrw3k_groups(int ntile) {
  const int g = 88;                                            // 88 x 3 workgroups: 14.08 ms/step at 176, 13.99 at 88 (32-88 alike)
  return ntile < g ? ntile : g;
}

extern "C" int64_t mcl_dense_conv3x3_wrw_workspace_floats(int64_t S) {
  if (S <= 0) return -1;
  const int ntile = (int)((S + T3 - 1) / T3);
  // (the row-walking form needs one partial per workgroup: <= 256; the map shape is not known here)
  const int64_t g = wrw3k_groups(ntile);
  return (g > 256 ? g : 256) * (int64_t)(C3_OUT * 9 * C3_IN);
}

extern "C" int mcl_dense_conv3x3_wrw_det(const void* dy, int64_t lddy, const void* z, int64_t S, int32_t H, int32_t W,
                                         const float* gamma, const float* beta, const float* mean, const float* rstd,
                                         float* workspace, float* dW, int32_t accumulate_w, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!dy || !z || !gamma || !beta || !mean || !rstd || !dW || !workspace || S <= 0 || H <= 0 || W <= 0) return MCL_EINVAL;
  if ((S % ((int64_t)H * W)) || S > 0x7fff0000LL || W > 150 || (lddy % 8) || lddy < C3_OUT || (reinterpret_cast<uintptr_t>(dy) & 15u) ||
      (reinterpret_cast<uintptr_t>(z) & 15u) || (reinterpret_cast<uintptr_t>(dW) & 15u) ||
      (reinterpret_cast<uintptr_t>(workspace) & 15u))
    return MCL_EUNSUPPORTED;
  const int ntile = (int)((S + T3 - 1) / T3);
  hipStream_t st = mcl_stream(stream);
  if (mcl_conv3x3_wrw_rows_applicable(S, H, W)) {
    const int rc = mcl_launch_conv3x3_wrw_rows(dy, lddy, z, S, H, W, gamma, beta, mean, rstd, workspace, dW, accumulate_w, st);
    if (rc != MCL_OK) return rc;
    MCL_CHECK_LAUNCH();
    return MCL_OK;
  }
  // per pixel group ONE fp32 partial (32 x 1152) in the workspace, written in disjoint column ranges by its three kernel-row
  // workgroups; fixed-order merge launch (bit-reproducible)
  const int G = wrw3k_groups(ntile);
  const int nblk = ((G + 7) / 8) * 8 * 3;
  hipLaunchKernelGGL(conv3x3_wrw_ky_kernel, dim3(nblk), dim3(256), 0, st, (const bf16_t*)dy, (long long)lddy,
                     (const bf16_t*)z, (long long)S, H, W, gamma, beta, mean, rstd, ntile, G, workspace);
  mcl_launch_wrw_merge(workspace, G, (long long)C3_OUT * 9 * C3_IN, dW, accumulate_w, st);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

// =====================================================================================================================
// DenseNet stem convolution conv0: 7x7, stride 2, pad 3, 3 -> 64 channels (torchvision densenet121.features.conv0;
// /root/reference/model.py:75-76), with the batch statistics norm0 needs from its epilogue.
//
//     y[n, oy, ox, co] = sum_{ky, kx, c} x[n, 2 oy - 3 + ky, 2 ox - 3 + kx, c] * W[co][ky][kx][c]
//
// Implicit GEMM M = pixels, N = 64, K = 7 x 24 (per ky the 7*3 = 21 contiguous NHWC input values padded to 24: every
// 8-value K chunk then is a contiguous run of one input row, 4-byte aligned, and the 3 pad weights are zero).  A
// workgroup owns two output rows of one image (<= 256 pixels): the 9 input rows they touch are staged once into a
// zero-padded LDS slab (the input is read from HBM ~once: 38 MB at 128 x 224^2), A fragments are 4 ds_read_b32 from
// the slab, the 22 weight fragments live in registers for the whole launch.  HBM-bound on the 205 MB output; MIOpen's
// split-K igemm needs a zero-fill pass (71 us) + 199 us, and a separate statistics pass (40 us) follows it.
namespace {

constexpr int C0_OUT = 64, C0_K = 7, C0_KP = 24;        // output channels, kernel size, padded (kx, c) run
// (a kernel row is 7 * 3 chunks of 8 + one zero chunk = 22 chunks -> 11 k-steps of 16)

// Input slab of the stem kernels, filled by LDS-DMA (round 6): row r of the slab holds input row iy = 2 oy0 - 3 + r as it lies in
// memory, element d = 3 ix + c at position C0_P0 + d of a row of C0_PW(W) elements; positions before / after the data are zero
// (left / right padding of the convolution: zeroed once per workgroup), a row outside the image is written as zeros by the DMA
// itself (zero-size descriptor).  The window of output pixel ox starts at position C0_P0 - 9 + 6 ox = 7 + 6 ox: ODD -- so the
// forward reads its 24-element k-runs from the even position 6 + 6 ox and carries the weights shifted by one (k-run element j'
// holds weight j' - 1; j' = 0, 22, 23 are zero weights): 4-byte aligned A fragments from rows that were never touched by a VALU
// instruction.  (Rounds 1-5 staged the rows through registers at element offset 9 with 2-byte LDS stores: 24 per thread and tile,
// behind three loads that each waited for the previous one.)
constexpr int C0_P0 = 16;
__host__ __device__ inline int c0_pw(int W) { return ((C0_P0 + 3 * W + 24 + 7) / 8) * 8; }   // row pitch in elements (16-byte rows)
typedef unsigned u32x4_d __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void c0_dma16(u32x4_d rsrc, unsigned voff, unsigned dst) {     // 64 lanes x 16 B -> LDS dst + 16 lane
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(rsrc), "s"(dst)
               : "memory");
}
__device__ __forceinline__ u32x4_d c0_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  return u32x4_d{(unsigned)a, (unsigned)(a >> 32) & 0xFFFFu, bytes, 0x00020000u};
}
// Rows r = wave, wave + 4, ... of the 9-row slab of tile (n, oy0): each row W * 6 bytes = cpr 16-byte chunks, 64 per instruction
// (lanes past the row's last chunk are switched off: they must not write into the right padding).
__device__ __forceinline__ void c0_dma_slab(const bf16_t* __restrict__ x, int n, int oy0, int H, int W, unsigned slab_lds, int PW,
                                            int wave, int lane) {
  const int cpr = (W * 6) >> 4;
  for (int r = wave; r < 9; r += 4) {
    const int iy = 2 * oy0 - 3 + r;
    const bool inside = iy >= 0 && iy < H;
    const u32x4_d rs = c0_rsrc(x + ((long long)(n * H + (inside ? iy : 0)) * W) * 3, inside ? (unsigned)(W * 6) : 0u);
    const unsigned dst = slab_lds + (unsigned)(r * PW + C0_P0) * 2u;
    for (int c0 = 0; c0 < cpr; c0 += 64)
      if (c0 + lane < cpr) c0_dma16(rs, (unsigned)(c0 + lane) * 16u, dst + (unsigned)c0 * 16u);
  }
}

__global__ __launch_bounds__(256, 3) void conv0_fwd_kernel(const bf16_t* __restrict__ x, int N, int H, int W,
                                                           const bf16_t* __restrict__ Wt, bf16_t* __restrict__ y,
                                                           float2* __restrict__ partial, int ntile) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int OH = H >> 1, OW = W >> 1;
  const int PW = c0_pw(W);                               // slab row pitch in elements
  const int npix = 2 * OW;                               // pixels per tile
  bf16_t* slab = reinterpret_cast<bf16_t*>(lds);         // [9][PW]
  const int slab_bytes = 9 * PW * 2;
  bf16_t* ot = reinterpret_cast<bf16_t*>(lds + slab_bytes);                  // [256][64] bf16
  float* red = reinterpret_cast<float*>(lds + slab_bytes + 256 * 128);       // [4 waves][64][2]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;

  // weight fragments: B[k][n], lane supplies n = nt*32 + l31, k chunk q = 2*s + h -> (ky = q/3, j0 = 8*(q%3)).
  // The 18.8 KB weight goes through LDS first (coalesced dword loads, all in flight; the output staging area is free until the first
  // epilogue): read from global memory element by element, each of a lane's 176 two-byte loads sat behind a bounds branch and
  // waited for the previous one (vmcnt(0) at every join) -- 27 us at the head of every workgroup, a quarter of the launch; and
  // 150 unconditional two-byte loads in flight at once spill.
  {
    constexpr int WDW = (C0_OUT * C0_K * 21) / 2;         // 4704 dwords
    unsigned* wl = reinterpret_cast<unsigned*>(ot);
    const unsigned* w32 = reinterpret_cast<const unsigned*>(Wt);
    unsigned wv[(WDW + 255) / 256];
#pragma unroll
    for (int k = 0; k < (WDW + 255) / 256; ++k) wv[k] = w32[min(tid + 256 * k, WDW - 1)];
#pragma unroll
    for (int k = 0; k < (WDW + 255) / 256; ++k) wl[min(tid + 256 * k, WDW - 1)] = wv[k];
  }
  __syncthreads();
  bf16x8 breg[2][11];
  {
    const bf16_t* wls = reinterpret_cast<const bf16_t*>(ot);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int s = 0; s < 11; ++s) {
        const int q = 2 * s + h, ky = q / 3, j0 = (q % 3) * 8;
        const int co = nt * 32 + l31;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int j = j0 + i - 1;                       // (k-run element j' = j0 + i holds weight j' - 1: see C0_P0)
          const bool ok = q < 21 && j >= 0 && j < 21;
          const short w1 = (short)wls[ok ? (co * C0_K + ky) * 21 + j : 0];
          breg[nt][s][i] = ok ? w1 : (short)0;
        }
      }
  }
  const int nmt = (npix + 31) >> 5;
  const unsigned slab_lds = (unsigned)(size_t)((__attribute__((address_space(3))) void*)(lds));
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  // the slab's padding: zeroed once; the first tile's rows requested
  for (int i = tid; i < slab_bytes / 16; i += 256) reinterpret_cast<uint4*>(lds)[i] = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();
  if ((int)blockIdx.x < ntile)
    c0_dma_slab(x, (int)blockIdx.x / (OH >> 1), ((int)blockIdx.x % (OH >> 1)) * 2, H, W, slab_lds, PW, wave_u, lane);

  for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const int n = tile / (OH >> 1), oy0 = (tile % (OH >> 1)) * 2;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's rows of the tile have landed ...
    __syncthreads();                                     // ... everybody's have; the previous tile's output staging is done with LDS

    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    for (int mt = wave; mt < nmt; mt += 4) {
      int m = mt * 32 + l31;
      const bool mvalid = m < npix;
      if (!mvalid) m = npix - 1;
      const int rr = m / OW, ox = m - rr * OW;
      const unsigned* arow = reinterpret_cast<const unsigned*>(slab + (2 * rr) * PW + 6 + 6 * ox);   // 4-byte aligned
      f32x16 acc[2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.0f;
#pragma unroll
      for (int s = 0; s < 11; ++s) {
        const int q = 2 * s + h, ky = q / 3, j0 = (q % 3) * 8;
        const unsigned* ap = arow + ((ky * PW + j0) >> 1);
        u32x4 av;
        av[0] = ap[0]; av[1] = ap[1]; av[2] = ap[2]; av[3] = ap[3];
        if (q >= 21) av = (u32x4){0u, 0u, 0u, 0u};
        const bf16x8 a = __builtin_bit_cast(bf16x8, av);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, breg[nt][s], acc[nt], 0, 0, 0);
      }
      // acc[nt][r]: pixel mt*32 + (r&3) + 8*(r>>2) + 4*h, channel nt*32 + l31
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int px = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const float v = round_bf16(acc[nt][r]);
          if (px < npix) {
            s1[nt] += v;
            s2[nt] = fmaf(v, v, s2[nt]);
            ot[px * C0_OUT + nt * 32 + l31] = (bf16_t)(__float_as_uint(v) >> 16);
          }
        }
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      s1[nt] += __shfl_xor(s1[nt], 32, 64);
      s2[nt] += __shfl_xor(s2[nt], 32, 64);
      if (h == 0) {
        red[(wave * 64 + nt * 32 + l31) * 2] = s1[nt];
        red[(wave * 64 + nt * 32 + l31) * 2 + 1] = s2[nt];
      }
    }
    __syncthreads();
    // every wave is done with the slab: the next tile's rows are requested now and land while this tile's statistics and
    // output rows go out
    {
      const int nx = tile + (int)gridDim.x;
      if (nx < ntile) c0_dma_slab(x, nx / (OH >> 1), (nx % (OH >> 1)) * 2, H, W, slab_lds, PW, wave_u, lane);
    }
    if (tid < 64) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        a += red[(w * 64 + tid) * 2];
        b += red[(w * 64 + tid) * 2 + 1];
      }
      const float nn = (float)npix;
      partial[(long long)tid * ntile + tile] = make_float2(a, b - a * a / nn);       // (sum, M2) of the tile
    }
    // the tile's pixels are contiguous in y: npix rows of 128 bytes
    bf16_t* yo = y + ((long long)(n * OH + oy0) * OW) * C0_OUT;
    for (int c = tid; c < npix * 8; c += 256)
      reinterpret_cast<uint4*>(yo)[c] = reinterpret_cast<const uint4*>(ot)[c];
  }
}

// Weight gradient of conv0:  dW[co][ky][kx][c] (+)= sum_p dy[p][co] * x[n, 2 oy - 3 + ky, 2 ox - 3 + kx, c]  (the
// channels-last parameter's .grad).  GEMM M = 64 (co), N = 7 x 24 (k, as in the forward), K = pixels.  Same
// tiles and input slab as the forward; the dy tile lies in LDS as in memory ([pixel][co], 16-byte stores) and an A fragment
// (8 consecutive pixels of one channel) is two transposing reads (round 6; the [co][pixel] tile of rounds 1-5 took 56 two-byte
// LDS stores per thread and tile, and its staging loops waited for every 16-byte load alone: 148 -> 108 us at 128 x 224^2);
// a B fragment (8 consecutive pixels of one k column) is 8 strided 2-byte reads of the slab (12 bytes apart).  Wave w owns co tile w & 1 and k tiles 3 (w >> 1) .. + 2; accumulators persist over the
// workgroup's tiles.  Each output row of the tile is padded to a multiple of 16 pixels in the transposed dy tile (zero
// columns: a 16-pixel k-step never straddles output rows, and OW = 56 -- 112-pixel her2st patches -- works).
// The workgroup's fp32 partial (64 x 7 x 21) goes to its workspace slot (fixed-order merge launch: deterministic).
__global__ __launch_bounds__(256, 2) void conv0_wrw_kernel(const bf16_t* __restrict__ x, int N, int H, int W,
                                                           const bf16_t* __restrict__ dy, float* __restrict__ dW,
                                                           int ntile, float* __restrict__ wpart, int dbg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int OH = H >> 1, OW = W >> 1;
  const int PW = (W + 6) * 3 + 8;
  const int OWp = (OW + 15) & ~15;
  const int npix = 2 * OW;
  bf16_t* slab = reinterpret_cast<bf16_t*>(lds);
  const int slab_bytes = (9 * PW * 2 + 15) & ~15;
  // dy tile as it lies in memory: [2 * OWp padded pixels][64 co], 128-byte rows, 16-byte chunk c of row r at physical chunk
  // c ^ (((r >> 1) & 1) << 2) (the four rows of a transposing read then cover all 64 banks).  A fragments -- 8 consecutive pixels
  // of one channel -- are two ds_read_b64_tr_b16; the former [co][pixel] tile cost 56 two-byte LDS stores per thread and tile.
  unsigned char* dyt = lds + slab_bytes;
  typedef short v4s __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int mt = wave & 1, kt0 = (wave >> 1) * 3;
  int koff[3];
  bool kval[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int k = (kt0 + t) * 32 + l31;
    const int ky = k / C0_KP, j = k % C0_KP;
    kval[t] = k < 7 * C0_KP && j < 21;
    koff[t] = kval[t] ? ky * PW + j : 0;
  }
  f32x16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

  // the pad columns of the transposed dy tile and the pads of the input slab stay zero for every tile: zeroed once
  for (int i = tid; i < (slab_bytes + 2 * OWp * 128) / 16; i += 256) reinterpret_cast<uint4*>(lds)[i] = make_uint4(0u, 0u, 0u, 0u);
  // this lane's transposing-read offset inside the dy tile for tile row 0 (16-lane group: lane i supplies row i >> 2, 8-byte
  // piece i & 3 of the group's 16 channels)
  const int i16 = lane & 15, q4 = i16 >> 2;
  const int lch = (mt * 32 + 16 * ((lane >> 4) & 1)) / 8 + ((i16 & 3) >> 1);
  const unsigned a_off = (unsigned)(q4 * 128 + ((lch ^ (((q4 >> 1) & 1) << 2)) << 4) + (i16 & 1) * 8);
  // A tile's dy rows (npix * 8 16-byte chunks: 7 per thread at 224^2) and input rows (9 * cpr chunks: 3 per thread) are requested
  // together, unconditionally, into registers at the top of the tile.  (Keeping them in flight ACROSS the MFMA loop was tried:
  // the kernel then sits at its 256-register cap and the compiler parks each loaded chunk in an AGPR behind a vmcnt(0).)
  constexpr int NDY = 8, NX = 4;                         // chunks per thread (W <= 256)
  const int cpr = (W * 6) >> 4;
  u32x4 rdy[NDY];
  uint4 rx[NX];
  unsigned xvalid = 0;                                   // bit u: rx[u] is a row inside the image
  // (every load is unconditional, from a clamped address: a load inside a branch is followed by the compiler's vmcnt(0) at the
  //  join, which serialises the tile's loads -- what the former staging loops did)
  for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const int n = tile / (OH >> 1), oy0 = (tile % (OH >> 1)) * 2;
    // all of the tile's loads in flight at once (registers: not live in the loop)
    const bf16_t* dyo = dy + ((long long)(n * OH + oy0) * OW) * C0_OUT;
#pragma unroll
    for (int u = 0; u < NDY; ++u) rdy[u] = reinterpret_cast<const u32x4*>(dyo)[min(tid + 256 * u, npix * 8 - 1)];
    xvalid = 0;
#pragma unroll
    for (int u = 0; u < NX; ++u) {
      const int c = min(tid + 256 * u, 9 * cpr - 1);
      const int r = c / cpr, cc = c % cpr;
      const int iy = 2 * oy0 - 3 + r;
      xvalid |= (iy >= 0 && iy < H) ? (1u << u) : 0u;
      rx[u] = *reinterpret_cast<const uint4*>(x + ((long long)(n * H + min(max(iy, 0), H - 1)) * W) * 3 + cc * 8);
    }
    __syncthreads();                                     // the previous tile's loop is done with LDS
    // dy tile, transposed: chunk c = (pixel, 8 channels)
#pragma unroll
    for (int u = 0; u < NDY; ++u) {
      const int c = tid + 256 * u;
      if (c < npix * 8 && !(dbg & 1)) {
        const int px = c >> 3, c8 = c & 7;
        const int pc = px < OW ? px : px - OW + OWp;            // row in the padded [2][OWp] pixel layout
        *reinterpret_cast<u32x4*>(dyt + pc * 128 + ((c8 ^ (((pc >> 1) & 1) << 2)) << 4)) = rdy[u];
      }
    }
#pragma unroll
    for (int u = 0; u < NX; ++u) {
      const int c = tid + 256 * u;
      if (c < 9 * cpr && !(dbg & 2)) {
        const int r = c / cpr, cc = c % cpr;
        bf16_t* d = slab + r * PW + 9 + cc * 8;          // odd element offset: 2-byte stores
        const bool ok = (xvalid >> u) & 1u;              // (a row outside the image: zeros)
        const unsigned wv[4] = {ok ? rx[u].x : 0u, ok ? rx[u].y : 0u, ok ? rx[u].z : 0u, ok ? rx[u].w : 0u};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          d[2 * i] = (bf16_t)(wv[i] & 0xFFFFu);
          d[2 * i + 1] = (bf16_t)(wv[i] >> 16);
        }
      }
    }
    __syncthreads();

    for (int p0 = 0; p0 < 2 * OWp && !(dbg & 4); p0 += 16) {           // k-step: (padded) pixels p0 .. p0 + 15 of one output row
      const int rr = p0 / OWp, ox0 = p0 - rr * OWp;
      const unsigned char* ap = dyt + (p0 + 8 * h) * 128 + a_off;
      const v4s alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(ap));
      const v4s ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(ap + 4 * 128));
      bf16x8 a;
      a[0] = alo[0]; a[1] = alo[1]; a[2] = alo[2]; a[3] = alo[3];
      a[4] = ahi[0]; a[5] = ahi[1]; a[6] = ahi[2]; a[7] = ahi[3];
      const bf16_t* brow = slab + (2 * rr) * PW + 6 * (ox0 + 8 * h);
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const bf16_t* bp = brow + koff[t];
        bf16x8 b;
#pragma unroll
        for (int i = 0; i < 8; ++i) b[i] = (short)bp[6 * i];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
      }
    }
  }
  // acc[t][r]: co = mt*32 + (r&3) + 8*(r>>2) + 4*h, k = (kt0+t)*32 + l31 -> dW[co][ky][j]  ((64, 7, 7*3) fp32)
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    if (!kval[t]) continue;
    const int k = (kt0 + t) * 32 + l31;
    const int ky = k / C0_KP, j = k % C0_KP;
    float* dst = wpart + (long long)blockIdx.x * (C0_OUT * C0_K * 21);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      dst[(co * C0_K + ky) * 21 + j] = acc[t][r];
    }
  }
}

}  // namespace


extern "C" int64_t mcl_conv0_workspace_floats(int32_t N, int32_t H, int32_t W) {
  if (N <= 0 || H <= 0 || W <= 0) return -1;
  return (int64_t)N * (H / 4 + 1) * C0_OUT * 2;
}

extern "C" int mcl_conv0_fwd(const void* x, int32_t N, int32_t H, int32_t W, const void* Wt, void* y, float* workspace,
                             float eps, float* mean, float* var, float* rstd, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!x || !Wt || !y || !workspace || N <= 0 || H <= 0 || W <= 0) return MCL_EINVAL;
  const bool want_stats = mean || var || rstd;
  if (want_stats && (!mean || !var || !rstd)) return MCL_EINVAL;
  // two output rows per workgroup, whole 16-byte input chunks per row
  if ((H % 4) || (W % 8) || W > 256 || (reinterpret_cast<uintptr_t>(x) & 15u) || (reinterpret_cast<uintptr_t>(y) & 15u))
    return MCL_EUNSUPPORTED;
  const int OH = H / 2, OW = W / 2;
  const int ntile = N * (OH / 2);
  const int PW = c0_pw(W);
  const size_t lds_bytes = (size_t)(9 * PW * 2) + 256 * 128 + 4 * 64 * 2 * 4;
  static mcl_device_once attr_once;
  if (auto attr_guard = attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv0_fwd_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  }
  hipStream_t st = mcl_stream(stream);
  float2* part = reinterpret_cast<float2*>(workspace);
  // persistent: the 22 weight fragments are built once per workgroup; 768 = 3 per CU (47 KB of LDS each); measured
  // 108 us vs 124 us at 1024 and 212 us with one workgroup per tile (128 x 224^2)
  hipLaunchKernelGGL(conv0_fwd_kernel, dim3(ntile < 768 ? ntile : 768), dim3(256), lds_bytes, st, (const bf16_t*)x, N, H,
                     W, (const bf16_t*)Wt, (bf16_t*)y, part, ntile);
  if (want_stats)
    hipLaunchKernelGGL(tile_stats_finalize_kernel, dim3(C0_OUT), dim3(256), 0, st, (const float2*)part, ntile, C0_OUT,
                       (long long)N * OH * OW, 2 * OW, eps, mean, var, rstd);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

// Deterministic: per-workgroup partials in the workspace (mcl_conv0_wrw_workspace_floats floats) + a fixed-order merge
// launch; accumulate_w != 0 adds into dW.
// Workgroups of the stem weight gradient: each owns a 37.6 KB fp32 partial.  Three per CU are resident (42 KB of LDS each):
// the kernel's phases (transposing stores of the dy tile, the input slab, the MFMA loop) are separated by barriers, and a
// single workgroup per CU -- the former grid of 256 -- overlaps none of them.
inline int conv0_wrw_grid(int ntile) {
  const int cap = 768;
  return ntile < cap ? ntile : cap;
}
extern "C" int64_t mcl_conv0_wrw_workspace_floats(int32_t N, int32_t H, int32_t W) {
  if (N <= 0 || H <= 0 || W <= 0) return -1;
  return (int64_t)conv0_wrw_grid(N * (H / 4)) * (C0_OUT * C0_K * 21);
}

extern "C" int mcl_conv0_wrw(const void* x, int32_t N, int32_t H, int32_t W, const void* dy, float* workspace, float* dW,
                             int32_t accumulate_w, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!x || !dy || !dW || !workspace || N <= 0 || H <= 0 || W <= 0) return MCL_EINVAL;
  if ((H % 4) || (W % 8) || W > 256 || (reinterpret_cast<uintptr_t>(x) & 15u) || (reinterpret_cast<uintptr_t>(dy) & 15u) ||
      (reinterpret_cast<uintptr_t>(workspace) & 15u) || (reinterpret_cast<uintptr_t>(dW) & 15u))
    return MCL_EUNSUPPORTED;
  const int OH = H / 2, OW = W / 2, OWp = (OW + 15) & ~15;
  const int ntile = N * (OH / 2);
  const int PW = (W + 6) * 3 + 8;
  const size_t lds_bytes = (size_t)((9 * PW * 2 + 15) & ~15) + (size_t)2 * OWp * 128;
  static mcl_device_once attr_once;
  if (auto attr_guard = attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv0_wrw_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  }
  hipStream_t st = mcl_stream(stream);
  const int grid = conv0_wrw_grid(ntile);
  // timing ablations for tools/bench_stem.py (results invalid): 1 no dy tile, 2 no input slab, 4 no MFMA loop
  static const int dbg = getenv("MCL_CONV0_WRW_DBG") ? atoi(getenv("MCL_CONV0_WRW_DBG")) : 0;
  hipLaunchKernelGGL(conv0_wrw_kernel, dim3(grid), dim3(256), lds_bytes, st, (const bf16_t*)x, N, H, W,
                     (const bf16_t*)dy, dW, ntile, workspace, dbg);
  mcl_launch_wrw_merge(workspace, grid, (long long)C0_OUT * C0_K * 21, dW, accumulate_w, st);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}
