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
// are exact; the matrix cores do twice the work (at one workgroup per CU the kernel is issue-bound: 32 MFMAs per
// 64-pixel tile = 1024 cycles, measured ~1800 with the pipeline below; the HBM floor is half of that) and the separate
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
constexpr int ROWB = BT * 2;  // bytes per LDS tile row (raw rows as they lie in HBM: 128 bf16)
constexpr int TILE_B = BK * ROWB;          // 16 KB per operand tile
constexpr int STAGE_B = 2 * TILE_B;        // dz tile + x tile
// Tiles in LDS per workgroup (128 KB): one being multiplied, the next one landed (the software pipeline reads its first
// fragments early), one in flight, one being refilled.  Three stages stall on vmcnt(0) at every tile (r03: 183k vs 115k
// cycles for the 65 tiles of a 56 x 56 slab).
constexpr int NSTAGE = 4;

// Slab stride of the partial workspace in floats: (M + 2) x N (two extra rows: the BatchNorm-backward sums) plus 256
// bytes, so that the stride is never a power of two (same-offset reads of all slabs would share HBM channels).
__host__ __device__ inline long long slab_stride(long long MN, int N) { return MN + 2LL * N + 64; }

#define MCL_LDSP(p) ((__attribute__((address_space(3))) void*)(p))

// One LDS-DMA instruction (buffer form): every lane fetches 16 bytes at its own 32-bit byte offset from a per-workgroup
// descriptor (base = the slab's first row, num_records = the slab's bytes); the wave's 1 KiB lands lane-linear at the
// (wave-uniform) LDS address in M0 -- no VGPR round trip, no 64-bit address arithmetic per piece.  A 16-byte chunk beyond
// the slab's end is out of range: it lands in LDS as ZEROS and touches no memory (checked on MI355X), so ragged last
// tiles and the pipeline's run-out past the last tile need no branch.  Inline asm on purpose: hidden from the compiler's
// vmcnt bookkeeping, the DMA is covered by the explicit vmcnt(N) + barrier that opens every tile
// (cdna_hip_programming.md 5.7).
typedef unsigned u32x4_s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void blds16(u32x4_s rsrc, unsigned voff, unsigned dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(rsrc), "s"(dst)
               : "memory");
}
__device__ __forceinline__ u32x4_s raw_rsrc(const void* base, unsigned long long bytes) {
  const unsigned long long a = (unsigned long long)base;
  return u32x4_s{(unsigned)a, (unsigned)(a >> 32) & 0xFFFFu, (unsigned)(bytes < 0xFFFFFFFFull ? bytes : 0xFFFFFFFFull), 0x00020000u};
}

// Raw tiles keep 256-byte rows (what the DMA writes), so the transposing reads are de-conflicted by an XOR swizzle
// instead of row padding: 16-byte chunk c of row r lives at physical chunk c ^ ((r & 3) << 2).  A 32-lane group of a
// ds_read_b64_tr_b16 touches 4 consecutive rows x 4 chunks; the XOR spreads them over 16 distinct chunks = all 64 banks.
// The swizzle is applied on the SOURCE side of the DMA (lane l of a 4-row piece fetches logical chunk
// (l & 15) ^ ((l >> 4) << 2) and lands on physical chunk l & 15).
// Fragment: 8 consecutive-k bf16 of channel (cbase + (lane & 31)) starting at tile row kbase (kbase % 4 == 0).
// frag_off: the lane's byte offset inside a tile for tile row 0; frag_rd adds the (compile-time) row offset, so a tile's
// reads are one address register per operand block + immediates.
__device__ __forceinline__ unsigned frag_off(int kg, int cbase, int lane) {
  const int i = lane & 15, q = i >> 2;
  const int lchunk = (cbase + 16 * ((lane >> 4) & 1)) / 8 + ((i & 3) >> 1);
  return (unsigned)((kg + q) * ROWB + ((lchunk ^ (q << 2)) << 4) + (i & 1) * 8);
}
__device__ __forceinline__ bf16x8 frag_rd(unsigned addr, int kbase) {
  typedef v4s __attribute__((address_space(3))) * lp_t;
  const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(size_t)(addr + kbase * ROWB));
  const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(size_t)(addr + (kbase + 4) * ROWB));
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef short s16x2_t __attribute__((ext_vector_type(2)));

// Workgroup = 4 waves; wave w owns output columns [32w, 32w + 32) of the 128-column tile and ALL 128 rows (four
// 32 x 32 blocks, for Q and for R: 128 accumulator registers).  Its x fragment holds ONE channel per lane, so the
// BatchNorm+ReLU mask is two per-lane scalars and is applied in registers between the LDS read and the MFMA: the
// masked operands mask*x and mask never exist in LDS, and there is no separate staging phase.
// MODE 0: dW = dz^T a (plain).  MODE 1: Gram matrices Qx, R -> dW1 + BatchNorm-backward sums (see the header).
// MODE 2: dW = dz^T relu(bn(x)) with the prologue applied in registers (one GEMM; for the side stream).
template <int MODE>
__global__ __launch_bounds__(256, 1) void wrw_partial_kernel(
    const bf16_t* __restrict__ dz, long long ldz, const bf16_t* __restrict__ x, long long ldx,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
    const float* __restrict__ rstd, const bf16_t* __restrict__ W1 /* [M][N] */, float* __restrict__ wpart /* [ks][M][N] */,
    long long slab /* floats per slab */, long long S, int M, int N, long long rows_per_wg, int ks, int tn) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[NSTAGE * STAGE_B];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware decode (block b runs on XCD b % 8): the tn column tiles of one pixel slab share an XCD (its L2 serves
  // the dz slab to all of them); consecutive slabs go round the 8 XCDs
  const int L = blockIdx.x, xcd = L & 7, qq = L >> 3;
  const int tx = qq % tn, z = (qq / tn) * 8 + xcd;
  if (z >= ks) return;
  const int n0 = tx * BT;
  const long long s_begin = (long long)z * rows_per_wg;
  const long long s_end = min(S, s_begin + rows_per_wg);
  const int nt = (int)((s_end - s_begin + BK - 1) / BK);
  const unsigned lds_base = (unsigned)(size_t)MCL_LDSP(lds);

  // this lane's channel and its mask constants: mask = [x*sc + sh > 0]
  const int l31 = lane & 31, hh = lane >> 5;
  const int n_lane = n0 + wave * 32 + l31;
  float sc = 0.0f, sh = -1.0f;
  constexpr bool FUSED = MODE == 1;
  if (MODE != 0 && n_lane < N) {
    sc = gamma[n_lane] * rstd[n_lane];
    sh = fmaf(-mean[n_lane], sc, beta[n_lane]);
  }

  // DMA source geometry of this lane: piece p (tile rows 4p .. 4p+3) is issued by wave p & 3
  const int prow = lane >> 4;                                   // row inside a piece
  const int lchunk = (lane & 15) ^ (prow << 2);                 // logical 16-byte chunk this lane fetches
  const int ca = lchunk * 8 < M ? lchunk * 8 : 0;               // dz column (elements); chunks beyond M: any valid one
  const int cx = n0 + lchunk * 8 < N ? n0 + lchunk * 8 : n0;    // x column; chunks beyond N: any valid one (unused)
  // Piece u (u = 0..3) of tile t: tile rows 4p .. 4p+3, p = wave + 4u, of both operands.
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const long long nrow = s_end - s_begin;
  const u32x4_s rsrc_a = raw_rsrc(dz + s_begin * ldz, nrow > 0 ? (unsigned long long)nrow * ldz * 2 : 0ull);
  const u32x4_s rsrc_x = raw_rsrc(x + s_begin * ldx, nrow > 0 ? (unsigned long long)((nrow - 1) * ldx + N) * 2 : 0ull);
  unsigned offa[4], offx[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long long r = 4 * (wave + 4 * u) + prow;
    offa[u] = (unsigned)((r * ldz + ca) * 2);
    offx[u] = (unsigned)((r * ldx + cx) * 2);
  }
  const unsigned tstep_a = (unsigned)(BK * ldz * 2), tstep_x = (unsigned)(BK * ldx * 2);
  auto dma_piece = [&](int t, int stage, int u) {
    const unsigned dst = lds_base + stage * STAGE_B + (wave_u + 4 * u) * 1024;
    blds16(rsrc_a, offa[u] + (unsigned)t * tstep_a, dst);
    blds16(rsrc_x, offx[u] + (unsigned)t * tstep_x, dst + TILE_B);
  };
  auto dma_tile = [&](int t, int stage) {
#pragma unroll
    for (int u = 0; u < 4; ++u) dma_piece(t, stage, u);
  };

  f32x16 accq[4], accr[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      accq[i][r] = 0.0f;
      accr[i][r] = 0.0f;
    }

  const int kg = 8 * hh;
  // One 64-pixel tile into the accumulators; the whole loop body is one basic block (no ragged-tile or end-of-slab
  // branch: out-of-range DMA chunks are zeros), so the mask arithmetic is scheduled under the MFMAs.
  // B operands of one 16-pixel step from the raw x fragment.  MODE 1: mask*x and mask (bf16 1.0 / 0).  The mask bit is
  // the SIGN of u = fma(x, -sc, nsh), nsh = -sh (or +0 when sh is a zero): u < 0 or u = -0 exactly when
  // fma(x, sc, sh) > 0 -- fma is odd in (sc, sh), an exact cancellation gives +0 in both, and with nsh never -0 the sum
  // of two zeros is +0 -- so the 16-bit masks are an arithmetic shift of the packed sign bits instead of two compares and
  // two selects per element.
  const float nsc = -sc, nsh = sh == 0.0f ? 0.0f : -sh;
  auto prep = [&](const bf16x8 fx, bf16x8& o0, bf16x8& o1) {
    const u32x4 w = __builtin_bit_cast(u32x4, fx);
    if (MODE == 0) {
      o0 = fx;
    } else if (MODE == 2) {
      u32x4 av;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const float lo = fmaxf(fmaf(__uint_as_float(w[d] << 16), sc, sh), 0.0f);
        const float hi = fmaxf(fmaf(__uint_as_float(w[d] & 0xFFFF0000u), sc, sh), 0.0f);
        const f32x2 pv = {lo, hi};
        av[d] = __builtin_bit_cast(unsigned, __builtin_convertvector(pv, bf16x2_t));      // RNE, as the forward
      }
      o0 = __builtin_bit_cast(bf16x8, av);
    } else {
      u32x4 xm, mk;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const float ul = fmaf(__uint_as_float(w[d] << 16), nsc, nsh);
        const float uh = fmaf(__uint_as_float(w[d] & 0xFFFF0000u), nsc, nsh);
        const unsigned sg = __builtin_amdgcn_perm(__float_as_uint(uh), __float_as_uint(ul), 0x07060302u);   // [uh.hi16 | ul.hi16]
        const unsigned sel = __builtin_bit_cast(unsigned, __builtin_bit_cast(s16x2_t, sg) >> 15);         // 0xFFFF where the sign is set
        xm[d] = w[d] & sel;
        mk[d] = 0x3F803F80u & sel;
      }
      o0 = __builtin_bit_cast(bf16x8, xm);
      o1 = __builtin_bit_cast(bf16x8, mk);
    }
  };
  unsigned fo[5];                                        // lane offsets of the four dz row blocks and of the x block
#pragma unroll
  for (int i = 0; i < 4; ++i) fo[i] = frag_off(kg, i * 32, lane);
  fo[4] = TILE_B + frag_off(kg, wave * 32, lane);

  // ---- main loop: one software pipeline over the 16-pixel steps g = 4 t + kq of the whole slab ----------------------
  // A wave alone on its SIMD hides about five single-issue instructions per MFMA, and only if they sit BETWEEN the MFMAs
  // (MI355X_MICROARCH.md); at most 15 LDS reads are in flight per wave.  So under the 8 MFMAs of step g the wave issues
  // the 8 dz-fragment reads of step g+1, the 2 x-fragment reads of step g+2 and the mask arithmetic of step g+1 (hipcc
  // clusters instruction classes: the interleave is spelled out with scheduling groups), plus one piece pair of tile
  // t + NSTAGE - 1.  The pipeline runs across tile boundaries: the barrier that opens iteration t certifies tile t+1
  // (not t) as landed, so that the last steps of tile t can already read the first fragments of tile t+1.
  //   stages: tile t being multiplied, t+1 landed, t+2 .. t+NSTAGE-2 in flight, the stage of tile t-1 being refilled.
#pragma unroll
  for (int t = 0; t < NSTAGE - 1; ++t) dma_tile(t, t);
  static_assert(NSTAGE == 4, "vmcnt immediates: 8 DMA instructions per tile, completion in issue order");
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");      // tile 0 (tiles 1 and 2 behind it)
  __syncthreads();
  unsigned adc[5], adn[5];                               // operand block addresses in the current / next tile's stage
#pragma unroll
  for (int j = 0; j < 5; ++j) adc[j] = lds_base + fo[j];
  bf16x8 fa_c[4], fx_n, b0, b1;
#pragma unroll
  for (int i = 0; i < 4; ++i) fa_c[i] = frag_rd(adc[i], 0);
  prep(frag_rd(adc[4], 0), b0, b1);
  fx_n = frag_rd(adc[4], 16);
  int stage = 0, fill = NSTAGE - 1;                      // stage of tile t; stage tile t + NSTAGE - 1 is fetched into
  for (int t = 0; t < nt; ++t) {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // tile t+1 (tile t+2 behind it)
    __syncthreads();                                     // ... everybody's pieces; and the stage of tile t-1 is free
    stage = stage + 1 == NSTAGE ? 0 : stage + 1;
#pragma unroll
    for (int j = 0; j < 5; ++j) adn[j] = lds_base + stage * STAGE_B + fo[j];
#pragma unroll
    for (int kq = 0; kq < BK / 16; ++kq) {
      bf16x8 fa_n[4], n0v, n1v;
#pragma unroll
      for (int i = 0; i < 4; ++i) fa_n[i] = kq + 1 < BK / 16 ? frag_rd(adc[i], (kq + 1) * 16) : frag_rd(adn[i], 0);
      const bf16x8 fx_nn = kq + 2 < BK / 16 ? frag_rd(adc[4], (kq + 2) * 16) : frag_rd(adn[4], (kq + 2 - BK / 16) * 16);
      prep(fx_n, n0v, n1v);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        accq[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_c[i], b0, accq[i], 0, 0, 0);
        if (FUSED) accr[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_c[i], b1, accr[i], 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {                      // 10 LDS reads over the step: 2 2 1 1 1 1 1 1 (or 3 3 2 2)
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                          // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, FUSED ? 2 : 3, 0);              // LDS reads of the next steps
        if (MODE != 0) __builtin_amdgcn_sched_group_barrier(0x002, FUSED ? 4 : 8, 0);   // the next step's mask arithmetic
      }
#pragma unroll
      for (int j = 2; j < (FUSED ? 8 : 4); ++j) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, FUSED ? 1 : 2, 0);
        if (MODE != 0) __builtin_amdgcn_sched_group_barrier(0x002, FUSED ? 4 : 8, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      dma_piece(t + NSTAGE - 1, fill, kq);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) fa_c[i] = fa_n[i];
      b0 = n0v; b1 = n1v; fx_n = fx_nn;
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) adc[j] = adn[j];
    fill = fill + 1 == NSTAGE ? 0 : fill + 1;
  }

  // ---- epilogue: acc[i][r] is element m = i*32 + (r&3) + 8*(r>>2) + 4*hh, n = n_lane
  float* wp = wpart + (long long)z * slab;
  const bool nok = n_lane < N;
  float t1 = 0.0f, t2 = 0.0f;
  float g = 0.0f, b = 0.0f, mu = 0.0f, rs = 0.0f;
  if (FUSED && nok) {
    g = gamma[n_lane]; b = beta[n_lane]; mu = mean[n_lane]; rs = rstd[n_lane];
  }
  if (nok) {
    // the 64 weights of this lane's column go out as one batch of loads (clamped row: no branch between them); a load
    // per element inside the loop below is a chain of 64 dependent round trips (~0.5 us each) at one workgroup per CU
    constexpr int IB = 4;                                // 32-row blocks per batch of loads
#pragma unroll
    for (int i0 = 0; i0 < 4; i0 += IB) {
      unsigned short wv[FUSED ? IB * 16 : 1];
      if (FUSED) {
#pragma unroll
        for (int i = i0; i < i0 + IB; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            wv[(i - i0) * 16 + r] = W1[(long long)(m < M ? m : M - 1) * N + n_lane];
          }
      }
#pragma unroll
      for (int i = i0; i < i0 + IB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          if (m >= M) continue;
          if (FUSED) {
            const float R = accr[i][r];
            const float Q = rs * fmaf(-mu, R, accq[i][r]);
            const float w = __uint_as_float(((unsigned)wv[(i - i0) * 16 + r]) << 16);
            t1 = fmaf(w, R, t1);
            t2 = fmaf(w, Q, t2);
            wp[(long long)m * N + n_lane] = fmaf(g, Q, b * R);
          } else {
            wp[(long long)m * N + n_lane] = accq[i][r];
          }
        }
    }
  }
  if (FUSED) {
    t1 += __shfl_xor(t1, 32, 64);          // the two lane halves hold disjoint rows m of the same column
    t2 += __shfl_xor(t2, 32, 64);
    if (hh == 0 && nok) {                  // rows M, M+1 of the slab: merged by the same code as the weight gradient
      wp[(long long)M * N + n_lane] = t1;
      wp[(long long)(M + 1) * N + n_lane] = t2;
    }
  }
}

// dW[m][n] (+)= sum_z wpart[z][m][n] (blocks [0, nbw)): a block owns 8 float4 columns (128 contiguous bytes per slab);
// its 32 thread groups each sum every 32nd slab with four loads in flight, then the 32 group sums are added in fixed
// order -- the same order every run.  (A thread that walks the slabs one dependent load at a time is latency-bound:
// 512 slabs x ~1 us.)  Per channel (blocks [nbw, ...)): the two BatchNorm-backward sums in slab order (double),
// parameter gradients, and the means of the dx pass.
constexpr int MQ = 8, MZG = 256 / MQ;
// Elements [0, MN) of a slab are the weight-gradient partial; with sums != 0 elements [MN, MN + 2N) are the partial
// BatchNorm-backward sums (row M: sum g, row M+1: sum g*xhat), merged by the same code and turned into dbeta / dgamma
// (+=) and the two means of the dx pass.
__global__ __launch_bounds__(256) void wrw_merge_kernel(const float* __restrict__ wpart, int ks, long long MN, int N,
                                                        long long S, float* __restrict__ dW, int accumulate_w,
                                                        int sums, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                        int accumulate_params, float* __restrict__ coef,
                                                        long long stride) {
  __shared__ float4 part[MZG][MQ];
  const long long total = MN + (sums ? 2LL * N : 0LL);
  const int q = threadIdx.x % MQ, zg = threadIdx.x / MQ;
  const long long e = ((long long)blockIdx.x * MQ + q) * 4;
  // eight independent loads in flight per thread: with few blocks (N = 64: 260 of them) the merge is otherwise
  // latency-bound
  float4 acc8[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) acc8[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (e < total) {                                       // MN, N multiples of 8
    const float* p = wpart + e;
    int zz = zg;
    for (; zz + 7 * MZG < ks; zz += 8 * MZG) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(p + (long long)(zz + u * MZG) * stride);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        acc8[u].x += v[u].x; acc8[u].y += v[u].y; acc8[u].z += v[u].z; acc8[u].w += v[u].w;
      }
    }
    // the remaining < 8 rounds (ALL rounds for ks < 256): the loads unconditional, from a clamped slab index, 2 / 4 / 8 of them in
    // flight by the workgroup-uniform round count -- a load inside the per-thread bounds branch is followed by the compiler's
    // vmcnt(0), one dependent round trip per round in a 5 us kernel that runs 124 times per step
    const int rounds = (ks - zz + zg + MZG - 1) / MZG;    // = ceil((ks - first slab of group 0) / MZG): uniform
    float4 v[8];
#define MERGE_TAIL(NU)                                                                                                   \
  {                                                                                                                      \
    _Pragma("unroll") for (int u = 0; u < NU; ++u)                                                                       \
        v[u] = *reinterpret_cast<const float4*>(p + (long long)min(zz + u * MZG, ks - 1) * stride);                      \
    _Pragma("unroll") for (int u = 0; u < NU; ++u) if (zz + u * MZG < ks) {                                              \
      acc8[u].x += v[u].x; acc8[u].y += v[u].y; acc8[u].z += v[u].z; acc8[u].w += v[u].w;                                \
    }                                                                                                                    \
  }
    if (rounds <= 1) MERGE_TAIL(1) else if (rounds <= 2) MERGE_TAIL(2) else if (rounds <= 4) MERGE_TAIL(4) else MERGE_TAIL(8)
#undef MERGE_TAIL
  }
  float4 a;
  a.x = ((acc8[0].x + acc8[1].x) + (acc8[2].x + acc8[3].x)) + ((acc8[4].x + acc8[5].x) + (acc8[6].x + acc8[7].x));
  a.y = ((acc8[0].y + acc8[1].y) + (acc8[2].y + acc8[3].y)) + ((acc8[4].y + acc8[5].y) + (acc8[6].y + acc8[7].y));
  a.z = ((acc8[0].z + acc8[1].z) + (acc8[2].z + acc8[3].z)) + ((acc8[4].z + acc8[5].z) + (acc8[6].z + acc8[7].z));
  a.w = ((acc8[0].w + acc8[1].w) + (acc8[2].w + acc8[3].w)) + ((acc8[4].w + acc8[5].w) + (acc8[6].w + acc8[7].w));
  part[zg][q] = a;
  __syncthreads();
  if (zg != 0 || e >= total) return;
#pragma unroll
  for (int g = 1; g < MZG; ++g) {
    const float4 v = part[g][q];
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  if (e < MN) {
    float4* o = reinterpret_cast<float4*>(dW + e);
    if (accumulate_w) {
      const float4 pv = *o;
      a.x += pv.x; a.y += pv.y; a.z += pv.z; a.w += pv.w;
    }
    *o = a;
    return;
  }
  // four channels of one of the two sums
  const long long r = e - MN;
  const int which = r >= N ? 1 : 0, c = (int)(r - (long long)which * N);
  float* pg = which ? dgamma : dbeta;
  const float v[4] = {a.x, a.y, a.z, a.w};
  const float inv_s = 1.0f / (float)S;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (accumulate_params) pg[c + i] += v[i];
    else pg[c + i] = v[i];
    coef[2 * (c + i) + which] = v[i] * inv_s;
  }
}

struct Split {
  int tn, ks;
  long long rows;
};

inline Split plan(long long S, int N) {
  Split p;
  p.tn = (N + BT - 1) / BT;
  // 192 workgroups, one per CU on three quarters of the chip, not two per CU: alone the kernel is faster at 512 (two
  // waves per SIMD cover each other's issue gaps), but in the step it shares the chip with the other lane's kernels and
  // every workgroup costs a 64 KB partial that the merge re-reads (r03 same-box: 12.00 ms/step at 192, 12.18 at 256).
  // The small maps (blocks 3-4: the whole operand set is 8-56 MB) keep at least 4 tiles (2 on the 7 x 7 maps) per
  // workgroup.
  const long long target = 192;
  long long ks = (target + p.tn - 1) / p.tn;
  const long long min_rows = S < 10000 ? 2 * BK : 4 * BK;
  const long long max_ks = (S + min_rows - 1) / min_rows;
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
  const int nb = (int)((MN / 4 + MQ - 1) / MQ);
  hipLaunchKernelGGL(wrw_merge_kernel, dim3(nb), dim3(256), 0, st, wpart, ks, MN, 0, 1LL, dW, accumulate_w, 0,
                     (float*)nullptr, (float*)nullptr, 0, (float*)nullptr, MN);
}

namespace {
// Few slabs (split-K of the ViT weight gradients: 2 .. 16): a thread owns one float4 of the result, requests that float4 of every
// slab before the first add and sums in slab order.  The many-slab kernel above gives a 256-thread workgroup 32 floats x 32
// slab groups: with 4 slabs an eighth of its lanes load anything (1.8 TB/s on 60 MB merges).
constexpr int FEW = 16;
__global__ __launch_bounds__(256) void wrw_merge_few_kernel(const float* __restrict__ wpart, int ks, long long MN, long long stride,
                                                            float* __restrict__ dW, int accumulate_w) {
  const long long e = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (e >= MN) return;
  float4 v[FEW];
#pragma unroll
  for (int z = 0; z < FEW; ++z) {
    v[z] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (z < ks) v[z] = *reinterpret_cast<const float4*>(wpart + (long long)z * stride + e);
  }
  float4 a = v[0];
#pragma unroll
  for (int z = 1; z < FEW; ++z) {
    if (z < ks) { a.x += v[z].x; a.y += v[z].y; a.z += v[z].z; a.w += v[z].w; }
  }
  float4* o = reinterpret_cast<float4*>(dW + e);
  if (accumulate_w) {
    const float4 pv = *o;
    a.x += pv.x; a.y += pv.y; a.z += pv.z; a.w += pv.w;
  }
  *o = a;
}
}  // namespace

void mcl_launch_wrw_merge_strided(const float* wpart, int ks, long long MN, long long stride, float* dW, int accumulate_w,
                                  hipStream_t st) {
  if (ks <= FEW && (MN % 4) == 0) {
    hipLaunchKernelGGL(wrw_merge_few_kernel, dim3((unsigned)((MN / 4 + 255) / 256)), dim3(256), 0, st, wpart, ks, MN, stride, dW,
                       accumulate_w);
    return;
  }
  const int nb = (int)((MN / 4 + MQ - 1) / MQ);
  hipLaunchKernelGGL(wrw_merge_kernel, dim3(nb), dim3(256), 0, st, wpart, ks, MN, 0, 1LL, dW, accumulate_w, 0,
                     (float*)nullptr, (float*)nullptr, 0, (float*)nullptr, stride);
}

extern "C" int64_t mcl_wrw_workspace_floats(int64_t S, int32_t M, int32_t N) {
  if (S <= 0 || M <= 0 || N <= 0) return -1;
  const Split p = plan(S, N);
  return (int64_t)p.ks * slab_stride((int64_t)(M < BT ? M : BT) * N, N);
}

namespace {
// One 128-row block of output channels: partial kernel + fixed-order merge.
template <int MODE>
void launch_wrw(const bf16_t* dz, long long ldz, const bf16_t* x, long long ldx, const float* gamma, const float* beta,
                const float* mean, const float* rstd, const bf16_t* W1, float* workspace, float* dW, int accumulate_w,
                float* dgamma, float* dbeta, int accumulate_params, float* coef, long long S, int M, int N,
                hipStream_t st) {
  const Split p = plan(S, N);
  const int ks8 = (p.ks + 7) / 8 * 8;
  const long long MN = (long long)M * N, slab = slab_stride(MN, N);
  hipLaunchKernelGGL(wrw_partial_kernel<MODE>, dim3(ks8 * p.tn), dim3(256), 0, st, dz, ldz, x, ldx, gamma, beta, mean,
                     rstd, W1, workspace, slab, S, M, N, p.rows, p.ks, p.tn);
  const int sums = MODE == 1 ? 1 : 0;
  const long long total = MN + (sums ? 2LL * N : 0LL);
  const int nb = (int)((total / 4 + MQ - 1) / MQ);
  hipLaunchKernelGGL(wrw_merge_kernel, dim3(nb), dim3(256), 0, st, (const float*)workspace, p.ks, MN, N, S, dW,
                     accumulate_w, sums, dgamma, dbeta, accumulate_params, coef, slab);
}
}  // namespace

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
  launch_wrw<1>((const bf16_t*)dz, 128LL, (const bf16_t*)x, (long long)ldx, gamma, beta, mean, rstd, (const bf16_t*)W1,
                workspace, dW, accumulate_w, dgamma, dbeta, accumulate_params, coef_out, (long long)S, 128, C,
                mcl_stream(stream));
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

// dW[M][N] (+)= dz[S][M]^T a'[S][N], a' = a (gamma == NULL: transition convolutions) or relu(a*sc + sh) with the
// BatchNorm folded to sc = gamma*rstd, sh = beta - mean*sc (all four given: the bottleneck convolution on the side
// stream) -- the deterministic replacement of the atomics kernel mcl_conv1x1_wrw_bf16.
extern "C" int mcl_conv1x1_wrw_det(const void* dz, int64_t ldz, const void* a, int64_t lda, const float* gamma,
                                   const float* beta, const float* mean, const float* rstd, float* workspace, float* dW,
                                   int32_t accumulate_w, int64_t S, int32_t M, int32_t N, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!dz || !a || !dW || !workspace || S <= 0 || M <= 0 || N <= 0) return MCL_EINVAL;
  const bool pro = gamma || beta || mean || rstd;
  if (pro && !(gamma && beta && mean && rstd)) return MCL_EINVAL;
  if ((M % 8) || (N % 8) || (ldz % 8) || (lda % 8) || (reinterpret_cast<uintptr_t>(dz) & 15u) ||
      (reinterpret_cast<uintptr_t>(a) & 15u) || (reinterpret_cast<uintptr_t>(dW) & 15u) ||
      (reinterpret_cast<uintptr_t>(workspace) & 15u))
    return MCL_EUNSUPPORTED;
  hipStream_t st = mcl_stream(stream);
  // M > 128 (transition convolutions: 128 / 256 / 512 output channels): one pass per 128 output channels; the passes
  // share the workspace (stream order)
  for (int m0 = 0; m0 < M; m0 += BT) {
    const int mm = M - m0 < BT ? M - m0 : BT;
    if (pro)
      launch_wrw<2>((const bf16_t*)dz + m0, (long long)ldz, (const bf16_t*)a, (long long)lda, gamma, beta, mean, rstd,
                    (const bf16_t*)nullptr, workspace, dW + (long long)m0 * N, accumulate_w, (float*)nullptr,
                    (float*)nullptr, 0, (float*)nullptr, (long long)S, mm, N, st);
    else
      launch_wrw<0>((const bf16_t*)dz + m0, (long long)ldz, (const bf16_t*)a, (long long)lda, gamma, beta, mean, rstd,
                    (const bf16_t*)nullptr, workspace, dW + (long long)m0 * N, accumulate_w, (float*)nullptr,
                    (float*)nullptr, 0, (float*)nullptr, (long long)S, mm, N, st);
  }
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}
