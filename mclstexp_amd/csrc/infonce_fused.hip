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
constexpr int TC = 128;           // columns per iteration (rows per workgroup: 32 per wave)
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

// ---- one 128-column tile for one wave (32 rows), as four 32-column blocks and three kinds of work:
//   A(cb)  16 MFMAs: logits T[c][r] of block cb          (LDS: 16 ds_read_b128, prefetched 4 steps ahead)
//   E(cb)  16 elements/lane of VALU: exponentials -> bf16 weights (grad) or online (max, sum) (lse)
//   B(cb)  16 MFMAs: dA += w(cb) . B_tile(cb)             (LDS: 32 transposing reads, prefetched 3 steps ahead)
// With ONE wave per SIMD nothing else hides the VALU work, so it is interleaved by hand.  A "stage" is a 32-step
// MFMA stream over TWO blocks (alternating accumulators: no back-to-back dependent MFMAs); every step may carry
// one element of E work of another block and/or one LDS-DMA piece of the NEXT tile; sched_barrier(0) after each
// step pins that interleave.
//   grad:  [A0 A1 + DMA(next)] [A2 A3 + E0 E1] [B0 B1 + E2 E3] [B2 B3]
//   lse:   [A0 A1 + E2' E3' + DMA(next)] [A2 A3 + E0 E1]        (E2' E3' = blocks 2,3 of the PREVIOUS tile)
// FIX = per-element fix-ups: grad: the tile holds positive pairs (-2 on the diagonal); lse: ragged right edge.
#define MCL_PIN() __builtin_amdgcn_sched_barrier(0)

struct Ctx {
  float kscale, inv_t, nrl2;   // log2e/T ; 1/T ; -log2e * lse_a[row]
  int h, C;
};

struct Dma {
  const unsigned char* gbase;  // B as bytes
  const unsigned char* alt;    // row used for columns >= C (zeros for the gradient, row C-1 for the statistics)
  const float* stat;           // -log2e * lse_b (gradient only)
  unsigned base_l;             // (l31 & 16) | ((l31 & 15) ^ (h << 2)): this lane's chunk before the per-piece XOR
  int h, wave, lane, C, ppw;   // ppw = DMA pieces per wave and tile (64 / waves)
  long long ldb_bytes;         // row stride of B in bytes
  // full-tile fast path: the piece's row pair starts at a WAVE-UNIFORM address (SGPR pair, advanced by two rows per
  // piece) and the lane adds a 32-bit offset: h rows + its swizzled chunk -- 2 VALU + 2 SALU per piece instead of a
  // per-lane 64-bit row pointer with an edge select (~13 VALU + exec masking)
  unsigned hl;                 // h * ldb_bytes
  int wave_s;                  // wave index, scalar
  const unsigned char* cur;    // scalar: first row of the next piece
};

__device__ __forceinline__ void glds16(const void* src, unsigned dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(src), "s"(dst)
               : "memory");
}
// saddr form: wave-uniform 64-bit base in SGPRs + per-lane 32-bit byte offset
// (M0 is not saved / restored here: nothing else in these kernels uses it -- checked in the ISA -- and the two extra
// s_mov per piece are issue slots of a lone wave.)
__device__ __forceinline__ void glds16s(const void* sbase, unsigned voff, unsigned dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0"
               :
               : "s"(sbase), "v"(voff), "s"(dst)
               : "memory", "m0");
}
__device__ __forceinline__ void glds4(const void* src, unsigned dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(src), "s"(dst)
               : "memory");
}

// LDS-DMA piece t of this wave: tile rows 2n, 2n+1 with n = wave*ppw + t (1 KiB, lane-linear in LDS).  Lane l lands on
// row 2n + (l >> 5), physical chunk l & 31, so it FETCHES logical chunk (l & 31) ^ f(row): the swizzle lives in the
// source address.  f(row) = ((t & 1) << 3) | (h << 2) | ((t >> 1) & 3) for row = 32*wave + 2t + h.
// Issued as inline asm ON PURPOSE: hipcc cannot prove that a DMA into the OTHER buffer does not alias the LDS
// it is reading and would drain vmcnt(0) -- the whole prefetch -- mid-tile.  Hidden from its bookkeeping, the DMA
// is covered by the explicit vmcnt(0) + barrier that ends every iteration (cdna_hip_programming.md 5.7).
template <bool FAST>
__device__ __forceinline__ void dma_piece(Dma& d, int t, int col0n, unsigned dst_tile) {
  const unsigned chunk = d.base_l ^ (unsigned)(((t & 1) << 3) | ((t >> 1) & 3));
  if (FAST) {   // every row of the tile exists: pieces are issued in order t = 0, 1, ..
    glds16s(d.cur, (chunk << 4) + d.hl, dst_tile + (unsigned)(d.wave_s * d.ppw + t) * 1024u);
    d.cur += 2 * d.ldb_bytes;
    return;
  }
  const int n = d.wave * d.ppw + t;          // piece = tile rows 2n, 2n+1 (ppw is 16 or 8: (2n + h) & 15 == 2t + h)
  const int gcol = col0n + 2 * n + d.h;
  // row pointer without a per-piece 64-bit multiply: (col0n + 2*wave*ppw + h) * ld is per lane and tile, the piece
  // adds the wave-uniform 2*t*ld
  const unsigned char* rowp =
      gcol >= d.C ? d.alt : d.gbase + (long long)(col0n + 2 * d.wave * d.ppw + d.h) * d.ldb_bytes + (long long)(2 * t) * d.ldb_bytes;
  glds16(rowp + (chunk << 4), __builtin_amdgcn_readfirstlane(dst_tile + n * 1024));
}
__device__ __forceinline__ void dma_stat(const Dma& d, int col0n, unsigned dst_stat) {
  if (d.wave < 2) {
    const int c = min(col0n + d.wave * 64 + d.lane, d.C - 1);
    glds4(d.stat + c, __builtin_amdgcn_readfirstlane(dst_stat + d.wave * 256));
  }
}

__device__ __forceinline__ bf16x8 ld_a(const unsigned char* lds, const int (&a1)[8], int cb, int ks) {
  return *reinterpret_cast<const bf16x8*>(lds + a1[ks & 7] + cb * 16384 + (ks >> 3) * 256);
}

__device__ __forceinline__ bf16x8 ld_b(const unsigned char* lds, const int (&a2)[8], int cb, int j) {
  // j = kk*8 + pb; k slots of lane half h = tile rows 16kk + 4h+{0..3} and 16kk + 8+4h+{0..3} of block cb
  const int kk = j >> 3, pb = j & 7;
  const unsigned char* p0 = lds + cb * 16384 + kk * 8192 + (pb >> 2) * 256;
  const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(p0 + a2[2 * (pb & 3)]));
  const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (v4s __attribute__((address_space(3)))*)(p0 + 4096 + a2[2 * (pb & 3) + 1]));
  bf16x8 bv;
  bv[0] = lo[0]; bv[1] = lo[1]; bv[2] = lo[2]; bv[3] = lo[3];
  bv[4] = hi[0]; bv[5] = hi[1]; bv[6] = hi[2]; bv[7] = hi[3];
  return bv;
}

// E work, gradient flavour, for the 32 elements (two 32-column blocks ECB, ECB+1) that ride on a 32-step stage.
// The per-element chain fma -> exp -> add -> pack is SKEWED over three consecutive steps (step s: fma of element
// s, exp of element s-1, add/pack of element s-2), so the VALU instructions inside one step are independent of
// each other: with one wave per SIMD a dependent chain would expose every VALU latency.  The column terms cl[]
// (-log2e*lse_b, from the LDS statistics tile) are prefetched >= 8 steps before their first use.
template <bool FIX>
struct EGradPair {
  float cl0[16], cl1[16];
  unsigned pk0[8], pk1[8];
  float xa1, xa2, ea1, ea2, wprev;
  __device__ __forceinline__ void load_cl(float (&cl)[16], const unsigned char* lds, int cls_off, int cb, int h) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 v = *reinterpret_cast<const float4*>(lds + cls_off + (cb * 32 + 8 * g + 4 * h) * 4);
      cl[4 * g] = v.x; cl[4 * g + 1] = v.y; cl[4 * g + 2] = v.z; cl[4 * g + 3] = v.w;
    }
  }
  // s in [0, 34): steps 32, 33 are the two drain steps after the stage's MFMA loop
  __device__ __forceinline__ void step(int s, int ecb, const Ctx& c, const f32x16& TE0, const f32x16& TE1, int col0,
                                       int dcol, bf16x8 (&wE0)[2], bf16x8 (&wE1)[2]) {
    if (s >= 2) {               // add + pack of element s-2
      const int e = s - 2, i = e & 15, cb = ecb + (e >> 4);
      float w = ea1 + ea2;
      if (FIX) {
        const int cl_ = cb * 32 + (i & 3) + 8 * (i >> 2) + 4 * c.h;
        if (cl_ == dcol) w -= 2.0f;
        if (col0 + cl_ >= c.C) w = 0.0f;
      }
      if (i & 1) {
        if (e < 16) pk0[i >> 1] = pack_bf16(wprev, w);
        else pk1[i >> 1] = pack_bf16(wprev, w);
      } else {
        wprev = w;
      }
      if (e == 15) {
        u32x4 p0 = {pk0[0], pk0[1], pk0[2], pk0[3]}, p1 = {pk0[4], pk0[5], pk0[6], pk0[7]};
        wE0[0] = __builtin_bit_cast(bf16x8, p0);
        wE0[1] = __builtin_bit_cast(bf16x8, p1);
      }
      if (e == 31) {
        u32x4 p0 = {pk1[0], pk1[1], pk1[2], pk1[3]}, p1 = {pk1[4], pk1[5], pk1[6], pk1[7]};
        wE1[0] = __builtin_bit_cast(bf16x8, p0);
        wE1[1] = __builtin_bit_cast(bf16x8, p1);
      }
    }
    if (s >= 1 && s <= 32) {    // exponentials of element s-1
      ea1 = __builtin_amdgcn_exp2f(xa1);
      ea2 = __builtin_amdgcn_exp2f(xa2);
    }
    if (s < 32) {               // exponent arguments of element s
      const int i = s & 15;
      const float t = s < 16 ? TE0[i] : TE1[i];
      xa1 = fmaf(t, c.kscale, c.nrl2);
      xa2 = fmaf(t, c.kscale, s < 16 ? cl0[i] : cl1[i]);
    }
    // IR-level anchor: without it the (pure) element computations sink below the per-step sched_barriers to the
    // end of the stage and nothing overlaps the MFMAs
    asm volatile("" : "+v"(xa1), "+v"(xa2), "+v"(ea1), "+v"(ea2), "+v"(wprev));
  }
};

// E work, lse flavour: online (max, sum) of this lane's row over one block (steps 0-7: max, 8-15: exp-sum)
template <bool FIX>
struct EStat {
  float mx, m_new, s0, s1;
  __device__ __forceinline__ float val(const Ctx& c, const f32x16& T, int cb, int i, int col0) {
    float t = T[i];
    if (FIX) {
      if (col0 + cb * 32 + (i & 3) + 8 * (i >> 2) + 4 * c.h >= c.C) t = -3.0e38f;
    }
    return t;
  }
  __device__ __forceinline__ void step(const Ctx& c, const f32x16& T, int cb, int i, int col0, float run_m) {
    if (i < 8) {
      const float u = val(c, T, cb, 2 * i, col0), v = val(c, T, cb, 2 * i + 1, col0);
      mx = i == 0 ? fmaxf(u, v) : fmaxf(mx, fmaxf(u, v));
      asm volatile("" : "+v"(mx));   // IR-level anchor (see EGrad::step)
    } else {
      if (i == 8) {
        m_new = fmaxf(run_m, mx * c.kscale);
        s0 = s1 = 0.0f;
      }
      const int e = 2 * (i - 8);
      s0 += __builtin_amdgcn_exp2f(fmaf(val(c, T, cb, e, col0), c.kscale, -m_new));
      s1 += __builtin_amdgcn_exp2f(fmaf(val(c, T, cb, e + 1, col0), c.kscale, -m_new));
      asm volatile("" : "+v"(s0), "+v"(s1));
    }
  }
  __device__ __forceinline__ void finish(float& run_m, float& run_l) {
    run_l = run_l * __builtin_amdgcn_exp2f(run_m - m_new) + (s0 + s1);
    run_m = m_new;
  }
};

// 32 logits MFMAs of blocks CB0, CB0+1 (alternating) [+ E work of blocks ECB, ECB+1] [+ the next tile's DMA].
// NDMA: DMA pieces of the next tile issued by this wave inside the stage (0 = none).
// PF0 / PF1: blocks whose column terms are prefetched into eg.cl0 (at step 16) / eg.cl1 (at step 17), or -1.
template <bool BWD, int CB0, int ECB, bool FIX, int NDMA, int PF0, int PF1, bool FASTDMA>
__device__ __forceinline__ void stage_aa(const unsigned char* lds, const int (&a1)[8], const bf16x8 (&bfrag)[16],
                                         f32x16& TA0, f32x16& TA1, const f32x16& TE0, const f32x16& TE1, int e_col0,
                                         int e_dcol, int cls_off, EGradPair<FIX>& eg, bf16x8 (&wE0)[2],
                                         bf16x8 (&wE1)[2], float& run_m, float& run_l, const Ctx& c, Dma& d,
                                         int col0n, unsigned dst_tile, unsigned dst_stat) {
  EStat<FIX> s0, s1;
  bf16x8 f[32];
#pragma unroll
  for (int s = 0; s < 4; ++s) f[s] = ld_a(lds, a1, CB0 + (s & 1), s >> 1);
#pragma unroll
  for (int s = 0; s < 32; ++s) {
    if (s + 4 < 32) f[s + 4] = ld_a(lds, a1, CB0 + ((s + 4) & 1), (s + 4) >> 1);
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if ((s & 1) == 0) TA0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[s], bfrag[s >> 1], s < 2 ? zero : TA0, 0, 0, 0);
    else TA1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[s], bfrag[s >> 1], s < 2 ? zero : TA1, 0, 0, 0);
    if (ECB >= 0) {
      if (BWD) {
        eg.step(s, ECB, c, TE0, TE1, e_col0, e_dcol, wE0, wE1);
      } else {
        const int i = s & 15;
        if (s < 16) {
          s0.step(c, TE0, ECB, i, e_col0, run_m);
          if (i == 15) s0.finish(run_m, run_l);
        } else {
          s1.step(c, TE1, ECB + 1, i, e_col0, run_m);
          if (i == 15) s1.finish(run_m, run_l);
        }
      }
    }
    if (BWD && PF0 >= 0 && s == 16) eg.load_cl(eg.cl0, lds, cls_off, PF0, c.h);
    if (BWD && PF1 >= 0 && s == 17) eg.load_cl(eg.cl1, lds, cls_off, PF1, c.h);
    if (NDMA > 0) {   // the next tile's DMA: this wave's NDMA pieces, one per even step
      if ((s & 1) == 0 && (s >> 1) < NDMA) dma_piece<FASTDMA>(d, s >> 1, col0n, dst_tile);
      if (BWD && s == 1) dma_stat(d, col0n, dst_stat);
    }
    MCL_PIN();
  }
  if (BWD && ECB >= 0) {
    eg.step(32, ECB, c, TE0, TE1, e_col0, e_dcol, wE0, wE1);
    eg.step(33, ECB, c, TE0, TE1, e_col0, e_dcol, wE0, wE1);
  }
}

// 32 gradient MFMAs of blocks CB0, CB0+1 (weights w0, w1) [+ E work of blocks ECB, ECB+1]; PF1: block whose
// column terms are prefetched into eg.cl1 at step 0 (cl1 is free since the previous stage's last step), or -1.
template <int CB0, int ECB, bool FIX, int PF1>
__device__ __forceinline__ void stage_bb(const unsigned char* lds, const int (&a2)[8], const bf16x8 (&w0)[2],
                                         const bf16x8 (&w1)[2], f32x16 (&acc)[8], const f32x16& TE0,
                                         const f32x16& TE1, int e_col0, int e_dcol, int cls_off, EGradPair<FIX>& eg,
                                         bf16x8 (&wE0)[2], bf16x8 (&wE1)[2], const Ctx& c) {
  bf16x8 g[32];
#pragma unroll
  for (int s = 0; s < 3; ++s) g[s] = ld_b(lds, a2, CB0 + (s >> 4), s & 15);
#pragma unroll
  for (int s = 0; s < 32; ++s) {
    if (s + 3 < 32) g[s + 3] = ld_b(lds, a2, CB0 + ((s + 3) >> 4), (s + 3) & 15);
    const int j = s & 15;
    acc[j & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s < 16 ? w0[j >> 3] : w1[j >> 3], g[s], acc[j & 7], 0, 0, 0);
    if (ECB >= 0) eg.step(s, ECB, c, TE0, TE1, e_col0, e_dcol, wE0, wE1);
    if (PF1 >= 0 && s == 0) eg.load_cl(eg.cl1, lds, cls_off, PF1, c.h);
    MCL_PIN();
  }
  if (ECB >= 0) {
    eg.step(32, ECB, c, TE0, TE1, e_col0, e_dcol, wE0, wE1);
    eg.step(33, ECB, c, TE0, TE1, e_col0, e_dcol, wE0, wE1);
  }
}

// NW waves per workgroup: 4 (one per SIMD, 512 registers each) for the gradient, whose 128 accumulator registers
// per wave leave no room for a second wave; 8 (two per SIMD, 256 registers each) for the statistics pass, which is
// instruction-issue bound with one wave per SIMD (a lone wave issues one instruction per ~4 cycles).
template <bool BWD, int NW>
__global__ __launch_bounds__(64 * NW, NW / 4) void strip_kernel(const bf16_t* __restrict__ A, long long lda,
                                                                const bf16_t* __restrict__ B, long long ldb,
                                                       int R, int C, int diag_off, float inv_t,
                                                       const float* __restrict__ lse_a,
                                                       const float* __restrict__ nlse_b, int nsplit,
                                                       int tiles_per_split, float2* __restrict__ stat_out,
                                                       float* __restrict__ dA, float coef,
                                                       const bf16_t* __restrict__ alt_row) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_B];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int split = blockIdx.x % nsplit, rtile = blockIdx.x / nsplit;
  constexpr int TRW = 32 * NW;   // rows per workgroup
  constexpr int PPW = 64 / NW;   // DMA pieces per wave and tile
  const int row0 = rtile * TRW;
  const int nct = (C + TC - 1) / TC;
  const int ct0 = split * tiles_per_split;
  const int nIt = min(nct, ct0 + tiles_per_split) - ct0;
  const unsigned lds_base = (unsigned)(size_t)MCL_LDSP(lds);

  // own rows as the MFMA B operand of phase A: lane holds A[r][16*ks + 8*h .. +7]
  const int my_r = row0 + wave * 32 + l31;
  const int rr = min(my_r, R - 1);
  bf16x8 bfrag[16];
  {
    const bf16_t* arow = A + (size_t)rr * lda + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) bfrag[ks] = *reinterpret_cast<const bf16x8*>(arow + 16 * ks);
  }
  Ctx c;
  c.inv_t = inv_t;
  c.kscale = inv_t * LOG2E;
  c.nrl2 = BWD ? -lse_a[rr] * LOG2E : 0.0f;
  c.h = h;
  c.C = C;
  Dma d;
  d.gbase = reinterpret_cast<const unsigned char*>(B);
  d.alt = reinterpret_cast<const unsigned char*>(alt_row);
  d.stat = nlse_b;
  d.base_l = (unsigned)((l31 & 16) | ((l31 & 15) ^ (h << 2)));
  d.h = h; d.wave = wave; d.lane = lane; d.C = C; d.ppw = PPW;
  d.ldb_bytes = ldb * 2;
  d.hl = (unsigned)(h * (ldb * 2));
  d.wave_s = __builtin_amdgcn_readfirstlane(wave);
  d.cur = d.gbase;

  // LDS byte offsets (relative to lds[]) of this lane's reads in the CURRENT buffer; flipped by ^TILE_B per tile.
  // phase A: row l31 of a 32-row block, logical chunk 2*ks + h  ->  physical ((2*ks) ^ (h ^ f(l31))) | (ks >> 3) << 4
  int a1[8];
  {
    const int x0 = h ^ fsw(l31 & 15);
#pragma unroll
    for (int q = 0; q < 8; ++q) a1[q] = l31 * ROWB + (((2 * q) ^ x0) << 4);
  }
  // phase B (transposing reads): lane i = lane&15 addresses row 4*h + (i>>2) [+8 for the hi half], 4 bf16 at
  // column 32*pb + 16*((lane>>4)&1) + 4*(i&3);  a2[2*(pb&3) + hi]
  int a2[8];
  if (BWD) {
    const int q = (lane & 15) >> 2, jj = lane & 3;
    const int g2 = 2 * ((lane >> 4) & 1) + (jj >> 1);
    const int base = (4 * h + q) * ROWB + (jj & 1) * 8;
#pragma unroll
    for (int pbl = 0; pbl < 4; ++pbl)
#pragma unroll
      for (int hi = 0; hi < 2; ++hi) a2[2 * pbl + hi] = base + ((((pbl ^ q) << 2) | (g2 ^ (h + 2 * hi))) << 4);
  }
  int cls_off = 2 * TILE_B;
  int buf = 0;

  f32x16 acc[8];
  if (BWD) {
#pragma unroll
    for (int pb = 0; pb < 8; ++pb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[pb][r] = 0.0f;
  }
  float run_m = -1.0e30f, run_l = 0.0f;
  // lse: blocks 2,3 of the PREVIOUS tile are reduced under the first MFMAs of the next one; "tile -1" is all -inf
  f32x16 Tp2, Tp3;
#pragma unroll
  for (int r = 0; r < 16; ++r) Tp2[r] = Tp3[r] = -3.0e38f;

  // first tile: plain (not interleaved) DMA
#pragma unroll
  for (int t = 0; t < PPW; ++t) dma_piece<false>(d, t, ct0 * TC, lds_base);
  if (BWD) dma_stat(d, ct0 * TC, lds_base + 2 * TILE_B);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // Gradient: tiles that hold this workgroup's positive pairs need the -2 fix-up.  The tile loop is split into
  // [before | diagonal | after] so that each loop has ONE body: two bodies merging in one loop turn the 128
  // accumulator registers into loop-carried phis that get copied between the VGPR and AGPR halves every tile.
  // (Ragged right edge of the gradient: those B rows are staged as zeros, no fix-up.)
  // Statistics: only the ragged last tile needs masking: [full tiles | ragged tile].
  int it_a, it_b;
  if (BWD) {
    const int dlo = row0 + diag_off, dhi = row0 + TRW - 1 + diag_off;   // global columns of the positive pairs
    it_a = (dlo >= 0 ? dlo / TC : 0) - ct0;
    it_b = (dhi >= 0 ? dhi / TC + 1 : 0) - ct0;
    it_a = max(0, min(it_a, nIt));
    it_b = max(it_a, min(it_b, nIt));
  } else {
    it_a = ((ct0 + nIt) * TC > C) ? nIt - 1 : nIt;
    it_b = nIt;
  }

  bf16x8 w0[2], w1[2], w2[2], w3[2];
#define MCL_TILE_ITER(FIXV, FD)                                                                                    \
  {                                                                                                               \
    const int col0 = (ct0 + it) * TC;                                                                             \
    const int col0n = (ct0 + min(it + 1, nIt - 1)) * TC; /* last iteration re-fetches its own tile: no branch */  \
    const unsigned dst_tile = lds_base + (buf ^ 1) * TILE_B;                                                      \
    const unsigned dst_stat = lds_base + 2 * TILE_B + (buf ^ 1) * STAT_STRIDE;                                    \
    const int dcol = my_r + diag_off - col0; /* tile column of this row's positive pair */                        \
    /* launder the lane constant of the DMA source address: otherwise the per-piece offsets are hoisted out of */ \
    /* the tile loop, spilled, and every reload's compiler-inserted vmcnt(0) drains the hand-placed DMA        */ \
    Dma dl = d;                                                                                                   \
    asm volatile("" : "+v"(dl.base_l), "+v"(dl.h), "+v"(dl.wave), "+v"(dl.hl));                                   \
    if (FD) dl.cur = d.gbase + (long long)(col0n + 2 * d.wave_s * PPW) * d.ldb_bytes;                             \
    f32x16 T0, T1;                                                                                                \
    EGradPair<FIXV> eg;                                                                                           \
    if (BWD) {                                                                                                    \
      f32x16 T2, T3;                                                                                              \
      stage_aa<true, 0, -1, FIXV, PPW, 0, 1, FD>(lds, a1, bfrag, T0, T1, T0, T1, col0, dcol, cls_off, eg, w0, w1,    \
                                              run_m, run_l, c, dl, col0n, dst_tile, dst_stat);                     \
      stage_aa<true, 2, 0, FIXV, 0, 2, -1, FD>(lds, a1, bfrag, T2, T3, T0, T1, col0, dcol, cls_off, eg, w0, w1,   \
                                               run_m, run_l, c, dl, col0n, dst_tile, dst_stat);                    \
      stage_bb<0, 2, FIXV, 3>(lds, a2, w0, w1, acc, T2, T3, col0, dcol, cls_off, eg, w2, w3, c);                  \
      stage_bb<2, -1, FIXV, -1>(lds, a2, w2, w3, acc, T2, T3, col0, dcol, cls_off, eg, w2, w3, c);                \
      /* keep the loop-carried accumulators in the AGPR half across the back edge */                            \
      asm volatile("" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]));                                \
      asm volatile("" : "+a"(acc[4]), "+a"(acc[5]), "+a"(acc[6]), "+a"(acc[7]));                                \
    } else {                                                                                                      \
      stage_aa<false, 0, 2, FIXV, PPW, -1, -1, FD>(lds, a1, bfrag, T0, T1, Tp2, Tp3, col0 - TC, dcol, cls_off, eg,   \
                                                w0, w1, run_m, run_l, c, dl, col0n, dst_tile, dst_stat);           \
      stage_aa<false, 2, 0, FIXV, 0, -1, -1, FD>(lds, a1, bfrag, Tp2, Tp3, T0, T1, col0, dcol, cls_off, eg, w0,   \
                                                 w1, run_m, run_l, c, dl, col0n, dst_tile, dst_stat);              \
    }                                                                                                             \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                              \
    __syncthreads();                                                                                            \
    buf ^= 1;                                                                                                     \
    cls_off ^= STAT_STRIDE;                                                                                       \
    _Pragma("unroll") for (int q = 0; q < 8; ++q) {                                                               \
      a1[q] ^= TILE_B;                                                                                            \
      if (BWD) a2[q] ^= TILE_B;                                                                                   \
    }                                                                                                             \
  }
  // Iterations whose PREFETCH targets a ragged tile (the last two of the split that owns the right edge, when C is
  // not a multiple of the tile) take the generic per-lane DMA addressing, on the fix-up body (valid for any tile);
  // all others use the scalar-base fast DMA.
  const int it_e = ((ct0 + nIt) * TC > C) ? max(0, nIt - 2) : nIt;
  int it = 0;
  if (BWD) {
    for (; it < min(it_a, it_e); ++it) MCL_TILE_ITER(false, true)
    for (; it < min(it_b, it_e); ++it) MCL_TILE_ITER(true, true)
    for (; it < it_e; ++it) MCL_TILE_ITER(false, true)
  } else {
    for (; it < it_e; ++it) MCL_TILE_ITER(false, true)
  }
  for (; it < nIt; ++it) MCL_TILE_ITER(true, false)
#undef MCL_TILE_ITER

  if (!BWD) {
    // blocks 2,3 of the last tile
    const int col0 = (ct0 + nIt - 1) * TC;
    EStat<true> e2, e3;
#pragma unroll
    for (int i = 0; i < 16; ++i) e2.step(c, Tp2, 2, i, col0, run_m);
    e2.finish(run_m, run_l);
#pragma unroll
    for (int i = 0; i < 16; ++i) e3.step(c, Tp3, 3, i, col0, run_m);
    e3.finish(run_m, run_l);
    // the two lane halves hold disjoint columns of the same row
    const float m_o = __shfl_xor(run_m, 32, 64), l_o = __shfl_xor(run_l, 32, 64);
    const float M = fmaxf(run_m, m_o);
    const float L = run_l * __builtin_amdgcn_exp2f(run_m - M) + l_o * __builtin_amdgcn_exp2f(m_o - M);
    if (h == 0 && my_r < R) stat_out[(size_t)split * R + my_r] = make_float2(M, L);
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

// diag[r] = inv_t * a[r] . b[r + diag_off]  (the positive-pair logit), one wave per row
__global__ __launch_bounds__(256) void rowdot_kernel(const bf16_t* __restrict__ a, long long lda,
                                                     const bf16_t* __restrict__ b, long long ldb, int R, int C,
                                                     int diag_off, float inv_t, float* __restrict__ diag) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= R) return;
  const int cidx = r + diag_off;
  if (cidx < 0 || cidx >= C) return;
  const uint2 va = *reinterpret_cast<const uint2*>(a + (size_t)r * lda + lane * 4);
  const uint2 vb = *reinterpret_cast<const uint2*>(b + (size_t)cidx * ldb + lane * 4);
  float s = __uint_as_float(va.x << 16) * __uint_as_float(vb.x << 16);
  s = fmaf(__uint_as_float(va.x & 0xFFFF0000u), __uint_as_float(vb.x & 0xFFFF0000u), s);
  s = fmaf(__uint_as_float(va.y << 16), __uint_as_float(vb.y << 16), s);
  s = fmaf(__uint_as_float(va.y & 0xFFFF0000u), __uint_as_float(vb.y & 0xFFFF0000u), s);
  s = wave_sum(s);
  if (lane == 0) diag[r] = s * inv_t;
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
constexpr int NW_LSE = 8, NW_GRAD = 4;
inline Plan make_plan(int R, int C, int nw) {
  Plan p;
  p.rt = (R + 32 * nw - 1) / (32 * nw);
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
  const Plan pf = make_plan(R, C, NW_LSE), p = make_plan(R, C, NW_GRAD);
  const int64_t fwd = (int64_t)pf.nsplit * R * (int64_t)sizeof(float2);
  // grad: [C floats: -log2e*lse_b, padded to 16 B] [one zero row, 512 B] [nsplit partial dA slabs when nsplit > 1]
  const int64_t bwd = (((int64_t)C * 4 + 15) / 16) * 16 + P * 2 +
                      (p.nsplit > 1 ? (int64_t)p.nsplit * R * P * (int64_t)sizeof(float) : 0);
  return fwd > bwd ? fwd : bwd;
}

extern "C" int mcl_infonce_fused_lse(const void* a, int64_t lda, const void* b, int64_t ldb, int32_t R, int32_t C,
                                     int32_t dim, int32_t diag_off, float inv_temp, float* lse, float* diag, void* workspace,
                                     int64_t ws_bytes, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!a || !b || !lse || !workspace || R <= 0 || C <= 0 || !(inv_temp > 0.0f)) return MCL_EINVAL;
  if (dim != P || !aligned16(a) || !aligned16(b) || !aligned16(workspace) || lda < P || ldb < P || (lda % 8) || (ldb % 8))
    return MCL_EUNSUPPORTED;
  const Plan p = make_plan(R, C, NW_LSE);
  if (ws_bytes < (int64_t)p.nsplit * R * (int64_t)sizeof(float2)) return MCL_EWORKSPACE;
  hipStream_t st = mcl_stream(stream);
  const bf16_t* last_row = (const bf16_t*)b + (size_t)(C - 1) * ldb;
  hipLaunchKernelGGL((strip_kernel<false, NW_LSE>), dim3(p.rt * p.nsplit), dim3(64 * NW_LSE), 0, st, (const bf16_t*)a,
                     (long long)lda, (const bf16_t*)b, (long long)ldb, R, C, diag_off, inv_temp, (const float*)nullptr, (const float*)nullptr,
                     p.nsplit, p.tps, (float2*)workspace, (float*)nullptr, 0.0f, last_row);
  if (diag)
    hipLaunchKernelGGL(rowdot_kernel, dim3((R + 3) / 4), dim3(256), 0, st, (const bf16_t*)a, (long long)lda,
                       (const bf16_t*)b, (long long)ldb, R, C, diag_off, inv_temp, diag);
  hipLaunchKernelGGL(lse_merge_kernel, dim3((R + 255) / 256), dim3(256), 0, st, (const float2*)workspace, R,
                     p.nsplit, lse);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_infonce_fused_grad(const void* a, int64_t lda, const void* b, int64_t ldb, int32_t R, int32_t C,
                                      int32_t dim, int32_t diag_off, float inv_temp, const float* lse_a, const float* lse_b,
                                      float coef, float* dA, void* workspace, int64_t ws_bytes,
                                      mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!a || !b || !lse_a || !lse_b || !dA || R <= 0 || C <= 0 || !(inv_temp > 0.0f)) return MCL_EINVAL;
  if (dim != P || !aligned16(a) || !aligned16(b) || !aligned16(dA) || lda < P || ldb < P || (lda % 8) || (ldb % 8))
    return MCL_EUNSUPPORTED;
  const Plan p = make_plan(R, C, NW_GRAD);
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
  hipLaunchKernelGGL((strip_kernel<true, NW_GRAD>), dim3(p.rt * p.nsplit), dim3(64 * NW_GRAD), 0, st, (const bf16_t*)a,
                     (long long)lda, (const bf16_t*)b, (long long)ldb, R, C, diag_off, inv_temp, lse_a, (const float*)nlse_b, p.nsplit, p.tps,
                     (float2*)nullptr, target, coef, (const bf16_t*)zero_row);
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
