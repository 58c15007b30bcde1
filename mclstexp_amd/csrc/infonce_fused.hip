// K8 fused: symmetric InfoNCE (model.py:242-247) WITHOUT materialising the B x B logits.
//
//   S = A B^T / T                       A: (R, 256) bf16 "own" embeddings, B: (C, 256) bf16 "other" embeddings
//   lse[r]  = log sum_c exp(S[r,c])                                         (mcl_infonce_fused_lse)
//   dA[r,:] = coef * sum_c (exp(S-lse_a[r]) + exp(S-lse_b[c]) - 2[c == r+diag_off]) B[c,:]   (mcl_infonce_fused_grad)
//
// Both directions of the symmetric loss are the same kernel with the operands swapped (rows of S <-> columns of
// S), which is also exactly the data-parallel decomposition: a rank owns a row strip (E_spot_loc vs E_img_all)
// and a column strip (E_img_loc vs E_spot_all) -- dist.py.
//
// MI355X mapping (flash-attention-shaped, MFMA-bound at scale):
//   * workgroup = 4 waves = one wave per SIMD with the whole 512-register file; it owns 128 rows of A (each wave
//     32 rows, held in registers as the MFMA *B* operand for the whole kernel) and walks 128-column tiles of B;
//   * B tiles arrive by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip), double buffered: the next tile's
//     64 KB is in flight while the current one is multiplied;
//   * phase 1 computes the TRANSPOSED logits tile T[c][r] = B_tile A^T (v_mfma_f32_32x32x16_bf16): in the
//     accumulator layout a lane then holds ONE row r and 16 columns c per 32x32 block, so the row-wise softmax
//     statistics are plain per-lane register loops (no cross-lane traffic), and after exp + bf16 packing the very
//     same registers ARE the A operand of phase 2 (k = c) -- no LDS round trip for the probabilities;
//   * phase 2 accumulates dA[r][p] += w[r][c] B[c][p]; its B operand (k = c strided) comes from the same LDS tile
//     through the transposing read ds_read_b64_tr_b16, with the tile rows fetched in the k-order the accumulator
//     layout dictates;
//   * the LDS tile is XOR-swizzled (16-byte chunk ^ f(row), f = swap of the two low bit pairs of the row) so that
//     BOTH the ds_read_b128 of phase 1 (32 rows x one chunk) and the transposing reads of phase 2 (4 rows x 4
//     chunks) are bank-conflict free; the DMA writes LDS linearly, so the swizzle is applied to the per-lane
//     SOURCE address;
//   * the column range is split over workgroups (partial statistics / partial dA merged by a small second
//     kernel, fixed order -> deterministic, no atomics) so that >= 256 workgroups exist whenever the problem
//     allows; split index = blockIdx % nsplit keeps the workgroups of one XCD on one column range (L2 reuse).
// Algorithmic work: 2*R*C*256 flop per lse call, 4*R*C*256 per grad call (2 of them recompute the logits).
#include "common.h"

namespace {

typedef unsigned short bf16_t;
typedef short v4s __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

constexpr int P = 256;             // embedding width (projection_dim)
constexpr int TR = 128, TC = 128;  // rows per workgroup, columns per iteration
constexpr int ROWB = P * 2;        // bytes per LDS tile row
constexpr int TILE_B = TC * ROWB;  // 64 KB
constexpr int STAT_B = TC * 4;
constexpr int STAT_STRIDE = 16384;       // stat tile of buffer b at 2*TILE_B + b*16 KiB: a distinct address BIT per buffer,
                                       // so the compiler can tell the DMA target from the tile being read
constexpr int LDS_B = 2 * TILE_B + STAT_STRIDE + STAT_B;
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

#define MCL_LDSP(p) ((__attribute__((address_space(3))) void*)(p))
#define MCL_GLBP(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ int fsw(int c) { return ((c & 3) << 2) | ((c >> 2) & 3); }

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));  // v_cvt_pk_bf16_f32 (RNE)
}

// ---- one 128-column tile for one wave (32 rows).
// The tile is processed as four 32-column blocks through three kinds of work:
//   A(cb)  16 MFMAs: logits T[c][r] of block cb          (LDS: 16 ds_read_b128, prefetched 2 steps ahead)
//   E(cb)  16 elements/lane of VALU: exponentials -> bf16 weights (grad) or online (max, sum) (lse)
//   B(cb)  16 MFMAs: dA += w(cb) . B_tile(cb)             (LDS: 32 transposing reads, prefetched 2 steps ahead)
// With ONE wave per SIMD nothing else hides the VALU work, so it is interleaved by hand: a "stage" is a 16-step
// MFMA stream (A or B) and every step carries one element of the E work of ANOTHER block; sched_barrier(0) after
// each step pins that interleave (<= ~7 VALU + 1-2 LDS reads per 32-cycle MFMA slot).
//   grad:  A0 | A1+E0 | B0+E1 | A2 | B1+E2 | A3 | B2+E3 | B3          lse:  A0 | A1+E0 | A2+E1 | A3+E2 | E3
// FIX = the tile holds the diagonal or the ragged right edge (per-element fix-ups; at most two tiles per strip).
#define MCL_PIN() __builtin_amdgcn_sched_barrier(0)

struct TileCtx {
  const unsigned char* tile;   // LDS tile of this iteration
  const float* cls;            // LDS: -log2e * lse_b of the tile's 128 columns (grad only)
  float kscale, inv_t, nrl2;   // log2e/T ; 1/T ; -log2e * lse_a[row]
  int dcol, col0, C, h;
};

__device__ __forceinline__ bf16x8 ld_a(const TileCtx& x, const int (&a1)[8], int cb, int ks) {
  return *reinterpret_cast<const bf16x8*>(x.tile + a1[ks & 7] + cb * 16384 + (ks >> 3) * 256);
}

__device__ __forceinline__ bf16x8 ld_b(const TileCtx& x, const int (&a2)[4][2], int cb, int j) {
  // j = kk*8 + pb; k slots of lane half h = tile rows 16kk + 4h+{0..3} and 16kk + 8+4h+{0..3} of block cb
  const int kk = j >> 3, pb = j & 7;
  const unsigned char* p0 = x.tile + cb * 16384 + kk * 8192 + (pb >> 2) * 256;
  const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(p0 + a2[pb & 3][0]));
  const v4s hi =
      __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(p0 + 4096 + a2[pb & 3][1]));
  bf16x8 bv;
  bv[0] = lo[0]; bv[1] = lo[1]; bv[2] = lo[2]; bv[3] = lo[3];
  bv[4] = hi[0]; bv[5] = hi[1]; bv[6] = hi[2]; bv[7] = hi[3];
  return bv;
}

// E work, gradient flavour: element i of block CBE -> bf16 weight (packed pairwise into pk[])
template <int CBE, bool FIX>
struct EGrad {
  float cl[16];
  unsigned pk[8];
  float wprev;
  __device__ __forceinline__ void begin(const TileCtx& x) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 v = *reinterpret_cast<const float4*>(x.cls + CBE * 32 + 8 * g + 4 * x.h);
      cl[4 * g] = v.x; cl[4 * g + 1] = v.y; cl[4 * g + 2] = v.z; cl[4 * g + 3] = v.w;
    }
  }
  __device__ __forceinline__ void step(const TileCtx& x, const f32x16& T, int i) {
    const float t = T[i];
    float w = __builtin_amdgcn_exp2f(fmaf(t, x.kscale, x.nrl2)) + __builtin_amdgcn_exp2f(fmaf(t, x.kscale, cl[i]));
    if (FIX) {
      const int cl_ = CBE * 32 + (i & 3) + 8 * (i >> 2) + 4 * x.h;
      if (cl_ == x.dcol) w -= 2.0f;
      if (x.col0 + cl_ >= x.C) w = 0.0f;
    }
    // IR-level anchor: without it the whole (pure) element computation sinks below the per-step
    // sched_barriers to the end of the stage and nothing overlaps the MFMAs
    asm volatile("" : "+v"(w));
    if (i & 1) pk[i >> 1] = pack_bf16(wprev, w);
    else wprev = w;
  }
  __device__ __forceinline__ void finish(bf16x8 (&wf)[2]) {
    u32x4 p0 = {pk[0], pk[1], pk[2], pk[3]}, p1 = {pk[4], pk[5], pk[6], pk[7]};
    wf[0] = __builtin_bit_cast(bf16x8, p0);
    wf[1] = __builtin_bit_cast(bf16x8, p1);
  }
};

// E work, lse flavour: online (max, sum) of this lane's row over block CBE (steps 0-7: max, 8-15: exp-sum)
template <int CBE, bool FIX>
struct EStat {
  float mx, m_new, s0, s1;
  __device__ __forceinline__ float val(const TileCtx& x, const f32x16& T, int i, float& diag_v, bool& have_diag) {
    float t = T[i];
    if (FIX) {
      const int cl_ = CBE * 32 + (i & 3) + 8 * (i >> 2) + 4 * x.h;
      if (x.col0 + cl_ >= x.C) t = -3.0e38f;
      if (cl_ == x.dcol) {
        diag_v = t * x.inv_t;
        have_diag = true;
      }
    }
    return t;
  }
  __device__ __forceinline__ void step(const TileCtx& x, const f32x16& T, int i, float run_m, float& diag_v,
                                       bool& have_diag) {
    if (i < 8) {
      const float u = val(x, T, 2 * i, diag_v, have_diag), v = val(x, T, 2 * i + 1, diag_v, have_diag);
      mx = i == 0 ? fmaxf(u, v) : fmaxf(mx, fmaxf(u, v));
      asm volatile("" : "+v"(mx));   // IR-level anchor (see EGrad::step)
    } else {
      if (i == 8) {
        m_new = fmaxf(run_m, mx * x.kscale);
        s0 = s1 = 0.0f;
      }
      bool dummy_b = false;
      float dummy_f;
      const int e = 2 * (i - 8);
      s0 += __builtin_amdgcn_exp2f(fmaf(val(x, T, e, dummy_f, dummy_b), x.kscale, -m_new));
      s1 += __builtin_amdgcn_exp2f(fmaf(val(x, T, e + 1, dummy_f, dummy_b), x.kscale, -m_new));
      asm volatile("" : "+v"(s0), "+v"(s1));
    }
  }
  __device__ __forceinline__ void finish(float& run_m, float& run_l) {
    run_l = run_l * __builtin_amdgcn_exp2f(run_m - m_new) + (s0 + s1);
    run_m = m_new;
  }
};

// stage with the logits MFMAs of block CBA (CBA < 0: none) carrying the E work of block CBE (CBE < 0: none)
template <bool BWD, int CBA, int CBE, bool FIX>
__device__ __forceinline__ void stage_a(const TileCtx& x, const int (&a1)[8], const bf16x8 (&bfrag)[16], f32x16& TA,
                                        const f32x16& TE, bf16x8 (&wfE)[2], float& run_m, float& run_l, float& diag_v,
                                        bool& have_diag) {
  EGrad<(CBE < 0 ? 0 : CBE), FIX> eg;
  EStat<(CBE < 0 ? 0 : CBE), FIX> es;
  if (CBE >= 0 && BWD) eg.begin(x);
  bf16x8 f[16];
  if (CBA >= 0) {
    f[0] = ld_a(x, a1, CBA, 0);
    f[1] = ld_a(x, a1, CBA, 1);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if (CBA >= 0) {
      if (i + 2 < 16) f[i + 2] = ld_a(x, a1, CBA, i + 2);
      if (i == 0) {
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        TA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[0], bfrag[0], zero, 0, 0, 0);
      } else {
        TA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[i], bfrag[i], TA, 0, 0, 0);
      }
    }
    if (CBE >= 0) {
      if (BWD) eg.step(x, TE, i);
      else es.step(x, TE, i, run_m, diag_v, have_diag);
    }
    if (CBA >= 0) MCL_PIN();
  }
  if (CBE >= 0) {
    if (BWD) eg.finish(wfE);
    else es.finish(run_m, run_l);
  }
}

// stage with the gradient MFMAs of block CB (weights wf) carrying the E work of block CBE (CBE < 0: none)
template <int CB, int CBE, bool FIX>
__device__ __forceinline__ void stage_b(const TileCtx& x, const int (&a2)[4][2], const bf16x8 (&wf)[2],
                                        f32x16 (&acc)[8], const f32x16& TE, bf16x8 (&wfE)[2]) {
  EGrad<(CBE < 0 ? 0 : CBE), FIX> eg;
  if (CBE >= 0) eg.begin(x);
  bf16x8 g[16];
  g[0] = ld_b(x, a2, CB, 0);
  g[1] = ld_b(x, a2, CB, 1);
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    if (j + 2 < 16) g[j + 2] = ld_b(x, a2, CB, j + 2);
    acc[j & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j >> 3], g[j], acc[j & 7], 0, 0, 0);
    if (CBE >= 0) eg.step(x, TE, j);
    MCL_PIN();
  }
  if (CBE >= 0) eg.finish(wfE);
}

template <bool BWD, bool FIX>
__device__ __forceinline__ void tile_body(const TileCtx& x, const int (&a1)[8], const int (&a2)[4][2],
                                          const bf16x8 (&bfrag)[16], f32x16 (&acc)[8], float& run_m, float& run_l,
                                          float& diag_v, bool& have_diag) {
  f32x16 T0, T1, T2, T3;
  bf16x8 w0[2], w1[2], w2[2], w3[2];
  if (BWD) {
    stage_a<true, 0, -1, FIX>(x, a1, bfrag, T0, T0, w0, run_m, run_l, diag_v, have_diag);
    stage_a<true, 1, 0, FIX>(x, a1, bfrag, T1, T0, w0, run_m, run_l, diag_v, have_diag);
    stage_b<0, 1, FIX>(x, a2, w0, acc, T1, w1);
    stage_a<true, 2, -1, FIX>(x, a1, bfrag, T2, T2, w2, run_m, run_l, diag_v, have_diag);
    stage_b<1, 2, FIX>(x, a2, w1, acc, T2, w2);
    stage_a<true, 3, -1, FIX>(x, a1, bfrag, T3, T3, w3, run_m, run_l, diag_v, have_diag);
    stage_b<2, 3, FIX>(x, a2, w2, acc, T3, w3);
    stage_b<3, -1, FIX>(x, a2, w3, acc, T3, w3);
  } else {
    stage_a<false, 0, -1, FIX>(x, a1, bfrag, T0, T0, w0, run_m, run_l, diag_v, have_diag);
    stage_a<false, 1, 0, FIX>(x, a1, bfrag, T1, T0, w0, run_m, run_l, diag_v, have_diag);
    stage_a<false, 2, 1, FIX>(x, a1, bfrag, T2, T1, w0, run_m, run_l, diag_v, have_diag);
    stage_a<false, 3, 2, FIX>(x, a1, bfrag, T3, T2, w0, run_m, run_l, diag_v, have_diag);
    stage_a<false, -1, 3, FIX>(x, a1, bfrag, T0, T3, w0, run_m, run_l, diag_v, have_diag);
  }
}

template <bool BWD>
__global__ __launch_bounds__(256, 1) void strip_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                       int R, int C, int diag_off, float inv_t,
                                                       const float* __restrict__ lse_a,
                                                       const float* __restrict__ lse_b, int nsplit,
                                                       int tiles_per_split, float2* __restrict__ stat_out,
                                                       float* __restrict__ diag_out, float* __restrict__ dA,
                                                       float coef, const bf16_t* __restrict__ zero_row) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_B];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int split = blockIdx.x % nsplit, rtile = blockIdx.x / nsplit;
  const int row0 = rtile * TR;
  const int nct = (C + TC - 1) / TC;
  const int ct0 = split * tiles_per_split;
  const int nIt = min(nct, ct0 + tiles_per_split) - ct0;
  const float kscale = inv_t * LOG2E;

  // own rows as the MFMA B operand of phase 1: lane holds A[r][16*ks + 8*h .. +7]
  const int my_r = row0 + wave * 32 + l31;
  const int rr = min(my_r, R - 1);
  bf16x8 bfrag[16];
  {
    const bf16_t* arow = A + (size_t)rr * P + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) bfrag[ks] = *reinterpret_cast<const bf16x8*>(arow + 16 * ks);
  }
  float nrl2 = 0.0f;
  if (BWD) nrl2 = -lse_a[rr] * LOG2E;

  // LDS-DMA staging of one 128 x 256 bf16 tile (+ the tile's column LSEs): wave w issues pieces 16w .. 16w+15,
  // piece n = tile rows 2n, 2n+1 (1 KiB, lane-linear in LDS); the lane's SOURCE chunk carries the swizzle.
  // LDS-DMA is issued as inline asm ON PURPOSE: hipcc cannot prove that a DMA into the OTHER buffer does not
  // alias the column statistics it is about to read and would drain vmcnt(0) -- the whole tile prefetch -- in
  // the middle of the tile.  Hidden from its bookkeeping, the DMA is covered by the explicit vmcnt(0) + barrier
  // that ends every iteration (recipe: cdna_hip_programming.md 5.7, M0 = wave-uniform LDS destination).
  auto stage = [&](int buf, int ct) {
    const int col0 = ct * TC;
    const unsigned tile_lds = (unsigned)(size_t)MCL_LDSP(lds + buf * TILE_B);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int n = wave * 16 + t;
      const int row = 2 * n + h;
      const int logical = (l31 & 16) | ((l31 & 15) ^ fsw(row));
      // columns beyond C: the statistics pass masks them (clamped source); the gradient stages ZERO rows, so
      // whatever weight they get multiplies nothing
      const int gcol = col0 + row;
      const bf16_t* src = ((BWD && gcol >= C) ? zero_row : B + (size_t)min(gcol, C - 1) * P) + logical * 8;
      const unsigned dst = __builtin_amdgcn_readfirstlane(tile_lds + n * 1024);
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(src), "s"(dst)
                   : "memory");
    }
    if (BWD && wave < 2) {   // the tile's 128 column terms (-log2e * lse_b)
      const int c = min(col0 + wave * 64 + lane, C - 1);
      const float* gsrc = lse_b + c;
      const unsigned dst = __builtin_amdgcn_readfirstlane(
          (unsigned)(size_t)MCL_LDSP(lds + 2 * TILE_B + buf * STAT_STRIDE + wave * 256));
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(gsrc), "s"(dst)
                   : "memory");
    }
  };

  // phase-1 read offsets: row l31 of a 32-row block, logical chunk 2*ks + h  ->  ((2*ks) ^ (h ^ f(l31))) | hi bit
  int a1[8];
  {
    const int x0 = h ^ fsw(l31 & 15);
#pragma unroll
    for (int q = 0; q < 8; ++q) a1[q] = l31 * ROWB + (((2 * q) ^ x0) << 4);
  }
  // phase-2 (transposing) read offsets: lane i = lane&15 addresses row 4*h + (i>>2) [+8 for the hi half], 4 bf16
  // at column 32*pb + 16*((lane>>4)&1) + 4*(i&3)
  int a2[4][2];
  if (BWD) {
    const int q = (lane & 15) >> 2, jj = lane & 3;
    const int g2 = 2 * ((lane >> 4) & 1) + (jj >> 1);
    const int base = (4 * h + q) * ROWB + (jj & 1) * 8;
#pragma unroll
    for (int pbl = 0; pbl < 4; ++pbl)
#pragma unroll
      for (int hi = 0; hi < 2; ++hi) a2[pbl][hi] = base + ((((pbl ^ q) << 2) | (g2 ^ (h + 2 * hi))) << 4);
  }

  f32x16 acc[8];
  if (BWD) {
#pragma unroll
    for (int pb = 0; pb < 8; ++pb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[pb][r] = 0.0f;
  }
  float run_m = -1.0e30f, run_l = 0.0f, diag_v = 0.0f;
  bool have_diag = false;

  stage(0, ct0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // Tiles that hold this workgroup's positive pairs (the diagonal) need per-element fix-ups.  The tile loop is
  // split into [before | diagonal | after] so that each loop has ONE body: two bodies merging in one loop make
  // the 128 accumulator registers loop-carried phis that get copied between the VGPR and AGPR halves every tile.
  // The ragged right edge needs no fix-up in the gradient (its B rows are staged as zeros) and is handled by
  // the FIX body in the statistics pass.
  const int dlo = row0 + diag_off, dhi = row0 + TR - 1 + diag_off;   // global columns of the positive pairs
  int it_a = (dlo >= 0 ? dlo / TC : 0) - ct0, it_b = (dhi >= 0 ? dhi / TC + 1 : 0) - ct0;
  if (!BWD && (ct0 + nIt) * TC > C && it_b >= nIt - 1) it_b = nIt;   // ragged edge adjacent to the diagonal range
  it_a = max(0, min(it_a, nIt));
  it_b = max(it_a, min(it_b, nIt));
  const bool ragged_last = !BWD && (ct0 + nIt) * TC > C && it_b < nIt;   // statistics pass: last tile needs FIX too

#define MCL_TILE_ITER(FIXV)                                                                              \
  {                                                                                                      \
    const int buf = it & 1;                                                                              \
    if (it + 1 < nIt) stage(buf ^ 1, ct0 + it + 1);                                                      \
    TileCtx x;                                                                                           \
    x.tile = lds + buf * TILE_B;                                                                         \
    x.cls = reinterpret_cast<const float*>(lds + 2 * TILE_B + buf * STAT_STRIDE);                        \
    x.kscale = kscale; x.inv_t = inv_t; x.nrl2 = nrl2;                                                   \
    x.col0 = (ct0 + it) * TC;                                                                            \
    x.dcol = my_r + diag_off - x.col0; /* tile column of this row's positive pair */                     \
    x.C = C; x.h = h;                                                                                    \
    tile_body<BWD, FIXV>(x, a1, a2, bfrag, acc, run_m, run_l, diag_v, have_diag);                        \
    if (BWD) { /* keep the loop-carried accumulators in the AGPR half across the back edge */           \
      asm volatile("" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]));                       \
      asm volatile("" : "+a"(acc[4]), "+a"(acc[5]), "+a"(acc[6]), "+a"(acc[7]));                       \
    }                                                                                                    \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                     \
    __syncthreads();                                                                                     \
  }
  int it = 0;
  for (; it < it_a; ++it) MCL_TILE_ITER(false)
  for (; it < it_b; ++it) MCL_TILE_ITER(true)
  const int it_c = ragged_last ? nIt - 1 : nIt;
  for (; it < it_c; ++it) MCL_TILE_ITER(false)
  if (!BWD) {
    for (; it < nIt; ++it) MCL_TILE_ITER(true)
  }
#undef MCL_TILE_ITER

  if (!BWD) {
    // the two lane halves hold disjoint columns of the same row
    const float m_o = __shfl_xor(run_m, 32, 64), l_o = __shfl_xor(run_l, 32, 64);
    const float M = fmaxf(run_m, m_o);
    const float L = run_l * __builtin_amdgcn_exp2f(run_m - M) + l_o * __builtin_amdgcn_exp2f(m_o - M);
    if (h == 0 && my_r < R) stat_out[(size_t)split * R + my_r] = make_float2(M, L);
    if (have_diag && my_r < R && diag_out) diag_out[my_r] = diag_v;
  } else {
    float* out = dA + (size_t)split * R * P;
#pragma unroll
    for (int pb = 0; pb < 8; ++pb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < R) out[(size_t)row * P + pb * 32 + l31] = coef * acc[pb][r];
      }
  }
}

__global__ __launch_bounds__(256) void lse_merge_kernel(const float2* __restrict__ stat, int R, int nsplit,
                                                        float* __restrict__ lse) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= R) return;
  float M = -1.0e30f;
  for (int s = 0; s < nsplit; ++s) M = fmaxf(M, stat[(size_t)s * R + r].x);
  float L = 0.0f;
  for (int s = 0; s < nsplit; ++s) {
    const float2 v = stat[(size_t)s * R + r];
    L += v.y * __builtin_amdgcn_exp2f(v.x - M);
  }
  lse[r] = LN2 * (M + __log2f(L));
}

__global__ __launch_bounds__(256) void neg_log2e_kernel(const float* __restrict__ x, int n, float* __restrict__ y,
                                                        float* __restrict__ zero_row) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = -LOG2E * x[i];
  if (i < P / 2) zero_row[i] = 0.0f;   // one all-zero bf16 embedding row (512 B)
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ part, long long n, int nsplit,
                                                           float* __restrict__ out) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  float4 a = *reinterpret_cast<const float4*>(part + i);
  for (int s = 1; s < nsplit; ++s) {
    const float4 b = *reinterpret_cast<const float4*>(part + (long long)s * n + i);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  *reinterpret_cast<float4*>(out + i) = a;
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ x, long long ldx,
                                                        bf16_t* __restrict__ y, long long ldy, long long rows,
                                                        int cols) {
  const int cv = cols >> 3;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= rows * cv) return;
  const long long r = t / cv;
  const int c = (int)(t % cv) * 8;
  const float4 a = *reinterpret_cast<const float4*>(x + r * ldx + c);
  const float4 b = *reinterpret_cast<const float4*>(x + r * ldx + c + 4);
  u32x4 pk;
  pk[0] = pack_bf16(a.x, a.y); pk[1] = pack_bf16(a.z, a.w);
  pk[2] = pack_bf16(b.x, b.y); pk[3] = pack_bf16(b.z, b.w);
  *reinterpret_cast<u32x4*>(y + r * ldy + c) = pk;
}

struct Plan {
  int rt, nct, nsplit, tps;
};
inline Plan make_plan(int R, int C) {
  Plan p;
  p.rt = (R + TR - 1) / TR;
  p.nct = (C + TC - 1) / TC;
  int want = (256 + p.rt - 1) / p.rt;  // >= 256 workgroups (one per CU) when the column count allows
  if (want < 1) want = 1;
  p.nsplit = want < p.nct ? want : p.nct;
  p.tps = (p.nct + p.nsplit - 1) / p.nsplit;
  p.nsplit = (p.nct + p.tps - 1) / p.tps;
  return p;
}
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

extern "C" int64_t mcl_infonce_fused_workspace_bytes(int32_t R, int32_t C, int32_t dim) {
  if (R <= 0 || C <= 0 || dim != P) return -1;
  const Plan p = make_plan(R, C);
  const int64_t fwd = (int64_t)p.nsplit * R * (int64_t)sizeof(float2);
  // grad: [C floats: -log2e*lse_b, padded to 16 B] [one zero row, 512 B] [nsplit partial dA slabs when nsplit > 1]
  const int64_t bwd = (((int64_t)C * 4 + 15) / 16) * 16 + P * 2 +
                      (p.nsplit > 1 ? (int64_t)p.nsplit * R * P * (int64_t)sizeof(float) : 0);
  return fwd > bwd ? fwd : bwd;
}

extern "C" int mcl_infonce_fused_lse(const void* a, const void* b, int32_t R, int32_t C, int32_t dim,
                                     int32_t diag_off, float inv_temp, float* lse, float* diag, void* workspace,
                                     int64_t ws_bytes, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!a || !b || !lse || !workspace || R <= 0 || C <= 0 || !(inv_temp > 0.0f)) return MCL_EINVAL;
  if (dim != P || !aligned16(a) || !aligned16(b) || !aligned16(workspace)) return MCL_EUNSUPPORTED;
  const Plan p = make_plan(R, C);
  if (ws_bytes < (int64_t)p.nsplit * R * (int64_t)sizeof(float2)) return MCL_EWORKSPACE;
  hipStream_t st = mcl_stream(stream);
  hipLaunchKernelGGL(strip_kernel<false>, dim3(p.rt * p.nsplit), dim3(256), 0, st, (const bf16_t*)a,
                     (const bf16_t*)b, R, C, diag_off, inv_temp, (const float*)nullptr, (const float*)nullptr,
                     p.nsplit, p.tps, (float2*)workspace, diag, (float*)nullptr, 0.0f, (const bf16_t*)nullptr);
  hipLaunchKernelGGL(lse_merge_kernel, dim3((R + 255) / 256), dim3(256), 0, st, (const float2*)workspace, R,
                     p.nsplit, lse);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_infonce_fused_grad(const void* a, const void* b, int32_t R, int32_t C, int32_t dim,
                                      int32_t diag_off, float inv_temp, const float* lse_a, const float* lse_b,
                                      float coef, float* dA, void* workspace, int64_t ws_bytes,
                                      mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!a || !b || !lse_a || !lse_b || !dA || R <= 0 || C <= 0 || !(inv_temp > 0.0f)) return MCL_EINVAL;
  if (dim != P || !aligned16(a) || !aligned16(b) || !aligned16(dA)) return MCL_EUNSUPPORTED;
  const Plan p = make_plan(R, C);
  if (!workspace || !aligned16(workspace)) return MCL_EINVAL;
  const int64_t nb_bytes = (((int64_t)C * 4 + 15) / 16) * 16;
  const int64_t need = nb_bytes + P * 2 + (p.nsplit > 1 ? (int64_t)p.nsplit * R * P * (int64_t)sizeof(float) : 0);
  if (ws_bytes < need) return MCL_EWORKSPACE;
  float* nlse_b = (float*)workspace;
  float* zero_row = (float*)((char*)workspace + nb_bytes);
  float* slabs = (float*)((char*)workspace + nb_bytes + P * 2);
  float* target = p.nsplit > 1 ? slabs : dA;
  hipStream_t st = mcl_stream(stream);
  hipLaunchKernelGGL(neg_log2e_kernel, dim3((C + 255) / 256), dim3(256), 0, st, lse_b, C, nlse_b, zero_row);
  hipLaunchKernelGGL(strip_kernel<true>, dim3(p.rt * p.nsplit), dim3(256), 0, st, (const bf16_t*)a,
                     (const bf16_t*)b, R, C, diag_off, inv_temp, lse_a, (const float*)nlse_b, p.nsplit, p.tps,
                     (float2*)nullptr, (float*)nullptr, target, coef, (const bf16_t*)zero_row);
  if (p.nsplit > 1) {
    const long long n = (long long)R * P;
    hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st,
                       (const float*)slabs, n, p.nsplit, dA);
  }
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_cast_f32_to_bf16(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t rows, int32_t cols,
                                    mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!x || !y || rows <= 0 || cols <= 0) return MCL_EINVAL;
  if ((cols % 8) || (ldx % 4) || (ldy % 8) || !aligned16(x) || !aligned16(y)) return MCL_EUNSUPPORTED;
  const long long n = rows * (cols / 8);
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, mcl_stream(stream), x,
                     (long long)ldx, (bf16_t*)y, (long long)ldy, (long long)rows, cols);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}
