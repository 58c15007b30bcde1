// Strided batched fp32 GEMM on the gfx950 matrix cores with a fused epilogue.
//   C = epi(alpha * A(MxK) * B(KxN));  see include/mclstexp_hip.h for the contract.
//
// MI355X mapping: 256-thread workgroup = 4 waves in a 2x2 arrangement over a BM x BN = 64 x 64
// output tile; each wave owns a 32x32 block computed with v_mfma_f32_32x32x2_f32 (exact fp32
// products and accumulation: bit-identical to an fmaf chain, which is what lets the path meet the
// reference's fp32 numerics) or, in bf16 mode, v_mfma_f32_32x32x16_bf16 on operands rounded at
// staging time.  K is walked in BK = 32 slices staged through LDS in [k][m] / [k][n] layout
// (row stride 65 words: conflict-free both for the scalar transposing stores and for the
// 32-lane MFMA operand reads); the next slice is prefetched into registers while the current one
// is multiplied.  All four operand layouts are handled by two staging variants per operand
// (contiguous along K, or along M/N), vectorised to 16-byte loads when base and leading dimension
// allow and falling back to dword loads for the odd gene counts (785, 171, 685, 3467).
#include "common.h"
#include <cstddef>
#include <cstring>
#include <stdlib.h>

namespace {

constexpr int BM = 64, BN = 64, BK = 32, NT = 256;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

struct GemmP {
  int M, N, K;
  const float* A; long long sAm, sAk, sAb;
  const float* B; long long sBk, sBn, sBb;
  float* C; long long ldc, sCb;
  float alpha; int flags;
  const float* bias;
  const float* resid; long long ldr, sRb;
  float* pre_out; long long ldp;
  const float* aux; long long ldaux;
  int batch, ksplit, kchunk;     // split-K: slice s covers k in [s*kchunk, min(K, (s+1)*kchunk)), kchunk % BK == 0
  float* ws;                     // split-K partials [ksplit][batch][M][N]
  unsigned* cnt;                 // split-K, one launch: arrival counter per (batch, tile) -- NULL: the two-launch form
  // threshold filter instead of a C store (retrieval): alpha * acc >= flt_thr[row] appends (value, col) to the row's list
  const float* flt_thr; int* flt_cnt; float* flt_val; int* flt_idx; int flt_cap;
};

__device__ __forceinline__ unsigned short f2bf(float f) {  // round-to-nearest-even
  unsigned u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

// Stage one (rows x BK) operand slice into regs: the operand is addressed as X[r*sr + k*sk] with
// r in [r0, r0+64) and k in [k0, k0+BK).  KC: contiguous along k (sk == 1) else along r (sr == 1).
// Each thread carries 8 floats.  VEC: 16-byte loads allowed (host: bases and leading dimensions 16-byte aligned AND the extent
// along the contiguous dimension a multiple of 4, so a chunk is inside the operand or outside it as a whole).
// Every load is UNCONDITIONAL, from an address clamped into the operand; what lies outside is zeroed when the registers are
// written to LDS (``ok``: one bit per element).  A load inside a bounds branch is followed by the compiler's vmcnt(0) at the
// join: the four loads of a slice then waited for one another, and the prefetch of the next slice ended before the MFMAs began.
template <bool KC, bool VEC>
__device__ __forceinline__ unsigned load_slice(float (&reg)[8], const float* __restrict__ X, long long sr, long long sk,
                                               int r0, int k0, int R, int K, int tid) {
  unsigned ok = 0;
  if (KC) {
    // 64 rows x 32 k: 8 float4 per row -> thread t: row = t/8 + 32*h, kq = (t%8)*4
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = r0 + (tid >> 3) + 32 * h;
      const int k = k0 + (tid & 7) * 4;
      const int rc = min(r, R - 1);
      if (VEC) {
        const float4 v = *reinterpret_cast<const float4*>(X + (long long)rc * sr + min(k, K - 4));
        reg[h * 4 + 0] = v.x; reg[h * 4 + 1] = v.y; reg[h * 4 + 2] = v.z; reg[h * 4 + 3] = v.w;
        ok |= (r < R && k < K) ? (0xFu << (4 * h)) : 0u;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          reg[h * 4 + i] = X[(long long)rc * sr + min(k + i, K - 1)];
          ok |= (r < R && k + i < K) ? (1u << (4 * h + i)) : 0u;
        }
      }
    }
  } else {
    // contiguous along r: 32 k-rows x 64 r: 16 float4 per k -> thread t: k = t/16 + 16*h, rq = (t%16)*4
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = k0 + (tid >> 4) + 16 * h;
      const int r = r0 + (tid & 15) * 4;
      const int kc = min(k, K - 1);
      if (VEC) {
        const float4 v = *reinterpret_cast<const float4*>(X + (long long)kc * sk + min(r, R - 4));
        reg[h * 4 + 0] = v.x; reg[h * 4 + 1] = v.y; reg[h * 4 + 2] = v.z; reg[h * 4 + 3] = v.w;
        ok |= (k < K && r < R) ? (0xFu << (4 * h)) : 0u;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          reg[h * 4 + i] = X[(long long)kc * sk + min(r + i, R - 1)];
          ok |= (k < K && r + i < R) ? (1u << (4 * h + i)) : 0u;
        }
      }
    }
  }
  return ok;
}

// Write the staged registers into the LDS slice T[k][r] (row stride LDT words); elements outside the operand as zeros.
template <bool KC, int LDT>
__device__ __forceinline__ void store_slice(const float (&reg)[8], unsigned ok, float* __restrict__ T, int tid) {
  if (KC) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = (tid >> 3) + 32 * h, k = (tid & 7) * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) T[(k + i) * LDT + r] = ((ok >> (4 * h + i)) & 1u) ? reg[h * 4 + i] : 0.0f;
    }
  } else {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = (tid >> 4) + 16 * h, r = (tid & 15) * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) T[k * LDT + r + i] = ((ok >> (4 * h + i)) & 1u) ? reg[h * 4 + i] : 0.0f;
    }
  }
}

// The same with the loads inside their bounds branches (the form of rounds 1-5), for launches with many workgroups per CU: there
// other workgroups cover the serialised loads, and the clamps, masks and selects of the form above cost the VALU-bound big-tile
// kernels 7-9 % (fp32 ViT-B/16 170.7 vs 156.1 ms/step).  Stage one (rows x BK) operand slice: X[r*sr + k*sk] with
// r in [r0, r0+64) and k in [k0, k0+BK).  KC: contiguous along k (sk == 1) else along r (sr == 1).
// Each thread carries 8 floats.  VEC: 16-byte loads allowed.
template <bool KC, bool VEC>
__device__ __forceinline__ void load_slice_branchy(float (&reg)[8], const float* __restrict__ X, long long sr, long long sk,
                                           int r0, int k0, int R, int K, int tid) {
  if (KC) {
    // 64 rows x 32 k: 8 float4 per row -> thread t: row = t/8 + 32*h, kq = (t%8)*4
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = r0 + (tid >> 3) + 32 * h;
      const int k = k0 + (tid & 7) * 4;
      const float* p = X + (long long)r * sr + k;
      if (VEC && r < R && k + 3 < K) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        reg[h * 4 + 0] = v.x; reg[h * 4 + 1] = v.y; reg[h * 4 + 2] = v.z; reg[h * 4 + 3] = v.w;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) reg[h * 4 + i] = (r < R && k + i < K) ? p[i] : 0.0f;
      }
    }
  } else {
    // contiguous along r: 32 k-rows x 64 r: 16 float4 per k -> thread t: k = t/16 + 16*h, rq = (t%16)*4
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = k0 + (tid >> 4) + 16 * h;
      const int r = r0 + (tid & 15) * 4;
      const float* p = X + (long long)k * sk + r;
      if (VEC && k < K && r + 3 < R) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        reg[h * 4 + 0] = v.x; reg[h * 4 + 1] = v.y; reg[h * 4 + 2] = v.z; reg[h * 4 + 3] = v.w;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) reg[h * 4 + i] = (k < K && r + i < R) ? p[i] : 0.0f;
      }
    }
  }
}

// Write the staged registers into the LDS slice T[k][r] (row stride LDT words).
template <bool KC, int LDT>
__device__ __forceinline__ void store_slice_plain(const float (&reg)[8], float* __restrict__ T, int tid) {
  if (KC) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = (tid >> 3) + 32 * h, k = (tid & 7) * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) T[(k + i) * LDT + r] = reg[h * 4 + i];
    }
  } else {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = (tid >> 4) + 16 * h, r = (tid & 15) * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) T[k * LDT + r + i] = reg[h * 4 + i];
    }
  }
}

// TM = 1: the 64 x 64 tile (each wave one 32 x 32 block).  TM = 2: a 128 x 128 tile for problems with many tiles (the fp32
// "reference numerics" image encoders: M = 25 216 token rows, im2col rows of the generic convolutions) -- each wave a 64 x 64 block as
// 2 x 2 MFMA tiles: two A and two B operand reads feed four products, and a tile's operand traffic per flop halves (the 64 x 64
// form asks L2 for 16 KB per 262 kflop: ~10 TB/s with every CU busy).
// (the body is a device function of the block coordinates: mcl_gemm_group runs several problems' tiles in one launch)
template <bool AKC, bool BKC, bool VEC, bool BF16, int TM, int PF = 1, bool CLAMP = true>
__device__ __forceinline__ void gemm_tile(const GemmP& p, const int bx, const int by, const int bzz, float* __restrict__ As,
                                          float* __restrict__ Bs) {
  constexpr int BMT = BM * TM, BNT = BN * TM, LD = BMT + 1;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = by * BMT, n0 = bx * BNT;
  const int bz = bzz % p.batch, ksl = bzz / p.batch;
  const int kbeg = ksl * p.kchunk, kend = min(p.K, kbeg + p.kchunk);
  const float* __restrict__ A = p.A + (long long)bz * p.sAb;
  const float* __restrict__ B = p.B + (long long)bz * p.sBb;

  f32x16 acc[TM][TM];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;

  // PF-deep register pipeline of operand slices (PF = 1 everywhere: two / four slices in flight were measured on the spot path's
  // skinny problems, before and after the loads were made unconditional -- spot branch 0.800-0.822 / 0.790 ms against 0.772 at
  // PF = 1, 136-212 registers against 80 -- and did not pay).
  float ra[PF][TM][8], rb[PF][TM][8];
  unsigned oa[PF][TM], ob[PF][TM];
  const int nk = (kend - kbeg + BK - 1) / BK;
#pragma unroll
  for (int u = 0; u < PF; ++u)
    if (u < nk) {
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        if (CLAMP) {
          oa[u][t] = load_slice<AKC, VEC>(ra[u][t], A, p.sAm, p.sAk, m0 + 64 * t, kbeg + u * BK, p.M, kend, tid);
          ob[u][t] = load_slice<BKC, VEC>(rb[u][t], B, p.sBn, p.sBk, n0 + 64 * t, kbeg + u * BK, p.N, kend, tid);
        } else {
          load_slice_branchy<AKC, VEC>(ra[u][t], A, p.sAm, p.sAk, m0 + 64 * t, kbeg + u * BK, p.M, kend, tid);
          load_slice_branchy<BKC, VEC>(rb[u][t], B, p.sBn, p.sBk, n0 + 64 * t, kbeg + u * BK, p.N, kend, tid);
        }
      }
    }

  const int arow = wm * 32 * TM + (lane & 31);
  const int brow = wn * 32 * TM + (lane & 31);
  const int khalf = lane >> 5;

  for (int kt0 = 0; kt0 < nk; kt0 += PF)
#pragma unroll
  for (int u = 0; u < PF; ++u) {
    const int kt = kt0 + u;
    if (kt >= nk) break;
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      if (CLAMP) {
        store_slice<AKC, LD>(ra[u][t], oa[u][t], As + 64 * t, tid);
        store_slice<BKC, LD>(rb[u][t], ob[u][t], Bs + 64 * t, tid);
      } else {
        store_slice_plain<AKC, LD>(ra[u][t], As + 64 * t, tid);
        store_slice_plain<BKC, LD>(rb[u][t], Bs + 64 * t, tid);
      }
    }
    __syncthreads();
    if (kt + PF < nk) {
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        if (CLAMP) {
          oa[u][t] = load_slice<AKC, VEC>(ra[u][t], A, p.sAm, p.sAk, m0 + 64 * t, kbeg + (kt + PF) * BK, p.M, kend, tid);
          ob[u][t] = load_slice<BKC, VEC>(rb[u][t], B, p.sBn, p.sBk, n0 + 64 * t, kbeg + (kt + PF) * BK, p.N, kend, tid);
        } else {
          load_slice_branchy<AKC, VEC>(ra[u][t], A, p.sAm, p.sAk, m0 + 64 * t, kbeg + (kt + PF) * BK, p.M, kend, tid);
          load_slice_branchy<BKC, VEC>(rb[u][t], B, p.sBn, p.sBk, n0 + 64 * t, kbeg + (kt + PF) * BK, p.N, kend, tid);
        }
      }
    }
    if (!BF16) {
#pragma unroll
      for (int ks = 0; ks < BK; ks += 2) {
        float a[TM], b[TM];
#pragma unroll
        for (int t = 0; t < TM; ++t) {
          a[t] = As[(ks + khalf) * LD + arow + 32 * t];
          b[t] = Bs[(ks + khalf) * LD + brow + 32 * t];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    } else {
      // 32x32x16 bf16: lane l holds A[i = l&31][k = 8*(l>>5) .. +7] (8 consecutive k)
#pragma unroll
      for (int ks = 0; ks < BK; ks += 16) {
        bf16x8 a[TM], b[TM];
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            a[t][i] = (short)f2bf(As[(ks + khalf * 8 + i) * LD + arow + 32 * t]);
            b[t][i] = (short)f2bf(Bs[(ks + khalf * 8 + i) * LD + brow + 32 * t]);
          }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  if (p.flt_thr) {
    // Threshold filter instead of a C store (retrieval): what passes its row's threshold goes to the row's candidate list.
    // Positions are reserved in two levels -- an LDS counter per tile row hands out the slot inside the tile, then ONE global
    // atomic per tile row reserves the tile's range in the list -- so that a workgroup waits for one round trip of global
    // atomics, not for a chain of them (one returning atomic per passing element: 4.3 ms against 2.2 for the plain product).
    int* s_cnt = reinterpret_cast<int*>(As);                       // [BMT] (the operand tiles are dead)
    int* s_base = s_cnt + BMT;
    for (int i = tid; i < BMT; i += NT) s_cnt[i] = 0;
    __syncthreads();
    int lpos[TM][TM][16];
#pragma unroll
    for (int ti = 0; ti < TM; ++ti)
#pragma unroll
      for (int tj = 0; tj < TM; ++tj) {
        const int col = n0 + (wn * TM + tj) * 32 + (lane & 31);
        const int rl0 = (wm * TM + ti) * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rl = rl0 + (r & 3) + 8 * (r >> 2), row = m0 + rl;
          const bool pass = row < p.M && col < p.N && p.alpha * acc[ti][tj][r] >= p.flt_thr[row];
          lpos[ti][tj][r] = pass ? atomicAdd(s_cnt + rl, 1) : -1;
        }
      }
    __syncthreads();
    for (int i = tid; i < BMT; i += NT) {
      const int c = s_cnt[i];
      s_base[i] = (c > 0 && m0 + i < p.M) ? atomicAdd(p.flt_cnt + m0 + i, c) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int ti = 0; ti < TM; ++ti)
#pragma unroll
      for (int tj = 0; tj < TM; ++tj) {
        const int col = n0 + (wn * TM + tj) * 32 + (lane & 31);
        const int rl0 = (wm * TM + ti) * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (lpos[ti][tj][r] < 0) continue;
          const int rl = rl0 + (r & 3) + 8 * (r >> 2);
          const int pos = s_base[rl] + lpos[ti][tj][r];
          if (pos < p.flt_cap) {
            p.flt_val[(long long)(m0 + rl) * p.flt_cap + pos] = p.alpha * acc[ti][tj][r];
            p.flt_idx[(long long)(m0 + rl) * p.flt_cap + pos] = col;
          }
        }
      }
    return;
  }
  if (CLAMP && TM == 1 && p.ksplit > 1 && p.cnt) {
    // Split-K in ONE launch: the partial goes out with write-through stores, the workgroup drains them and takes a ticket on its
    // tile's counter; the LAST slice to arrive adds all slices in slice order (the order of gemm_splitk_epilogue_kernel: the
    // result is the same bit for bit, whoever is last) and runs the epilogue below.  Nobody waits for anybody; the counter is
    // left at zero.  (A second launch costs the spot path ~8 us per product: 25 of them per step.)
    const int col = n0 + wn * 32 + (lane & 31);
    const int rbase = m0 + wm * 32 + 4 * (lane >> 5);
    const long long MN = (long long)p.M * p.N;
    unsigned* W = reinterpret_cast<unsigned*>(p.ws) + ((long long)ksl * p.batch + bz) * MN;
    if (col < p.N) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rbase + (r & 3) + 8 * (r >> 2);
        if (row < p.M)
          __hip_atomic_store(W + (long long)row * p.N + col, __float_as_uint(acc[0][0][r]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* s_flag = reinterpret_cast<int*>(As);
    if (tid == 0) {
      unsigned* c = p.cnt + ((long long)bz * gridDim.y + by) * gridDim.x + bx;
      const unsigned ticket = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *s_flag = (ticket == (unsigned)p.ksplit - 1u);
      if (*s_flag) __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!*s_flag) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.0f;
    if (col < p.N) {
      const unsigned* W0 = reinterpret_cast<const unsigned*>(p.ws) + (long long)bz * MN + col;
      for (int s0 = 0; s0 < p.ksplit; s0 += 4) {        // four slices' elements in flight
        unsigned v[4][16];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const unsigned* Ws_ = W0 + (long long)min(s0 + i, p.ksplit - 1) * p.batch * MN;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = min(rbase + (r & 3) + 8 * (r >> 2), p.M - 1);
            v[i][r] = __hip_atomic_load(Ws_ + (long long)row * p.N, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (s0 + i < p.ksplit) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][0][r] += __uint_as_float(v[i][r]);
          }
      }
    }
  }
  const bool merged = CLAMP && TM == 1 && p.ksplit > 1 && p.cnt;
  // epilogue: acc[i][j][r] -> row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31 of the wave's (i, j) 32x32 block
#pragma unroll
  for (int ti = 0; ti < TM; ++ti)
#pragma unroll
    for (int tj = 0; tj < TM; ++tj) {
      const int col = n0 + (wn * TM + tj) * 32 + (lane & 31);
      const int rbase = m0 + (wm * TM + ti) * 32 + 4 * (lane >> 5);
      if (col >= p.N) continue;
      if (p.ksplit > 1 && !merged) {   // raw partial of this K slice; the epilogue runs in gemm_splitk_epilogue_kernel
        float* __restrict__ W = p.ws + ((long long)ksl * p.batch + bz) * p.M * p.N;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rbase + (r & 3) + 8 * (r >> 2);
          if (row < p.M) W[(long long)row * p.N + col] = acc[ti][tj][r];
        }
        continue;
      }
      float* __restrict__ C = p.C + (long long)bz * p.sCb;
      const float* __restrict__ R = p.resid ? p.resid + (long long)bz * p.sRb : nullptr;
      const float bias = p.bias ? p.bias[col] : 0.0f;
      // The epilogue's operands (GELU' argument, residual, the C that is accumulated into) for all 16 rows of the lane, loaded
      // up front from row-clamped addresses: inside the per-row bounds branch every one of them was followed by the compiler's
      // vmcnt(0) -- up to 48 dependent round trips per tile, more than the product itself on the spot path's skinny problems.
      // (the skinny-launch instances only: 48 registers -- the many-tile instances keep their occupancy and the per-row form)
      if (CLAMP) {
        float av[16], rv[16], cv[16];
        if (p.flags & MCL_EPI_GELU_BWD) {
#pragma unroll
          for (int r = 0; r < 16; ++r) av[r] = p.aux[(long long)min(rbase + (r & 3) + 8 * (r >> 2), p.M - 1) * p.ldaux + col];
        }
        if (R) {
#pragma unroll
          for (int r = 0; r < 16; ++r) rv[r] = R[(long long)min(rbase + (r & 3) + 8 * (r >> 2), p.M - 1) * p.ldr + col];
        }
        if (p.flags & MCL_EPI_ACCUM) {
#pragma unroll
          for (int r = 0; r < 16; ++r) cv[r] = C[(long long)min(rbase + (r & 3) + 8 * (r >> 2), p.M - 1) * p.ldc + col];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rbase + (r & 3) + 8 * (r >> 2);
          if (row >= p.M) continue;
          float v = p.alpha * acc[ti][tj][r] + bias;
          if (p.pre_out) p.pre_out[(long long)row * p.ldp + col] = v;
          if (p.flags & MCL_EPI_GELU) v = gelu_erf(v);
          if (p.flags & MCL_EPI_GELU_BWD) v *= gelu_erf_grad(av[r]);
          if (R) v += rv[r];
          if (p.flags & MCL_EPI_ACCUM) v += cv[r];
          C[(long long)row * p.ldc + col] = v;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rbase + (r & 3) + 8 * (r >> 2);
          if (row >= p.M) continue;
          float v = p.alpha * acc[ti][tj][r] + bias;
          if (p.pre_out) p.pre_out[(long long)row * p.ldp + col] = v;
          if (p.flags & MCL_EPI_GELU) v = gelu_erf(v);
          if (p.flags & MCL_EPI_GELU_BWD) v *= gelu_erf_grad(p.aux[(long long)row * p.ldaux + col]);
          if (R) v += R[(long long)row * p.ldr + col];
          if (p.flags & MCL_EPI_ACCUM) v += C[(long long)row * p.ldc + col];
          C[(long long)row * p.ldc + col] = v;
        }
      }
    }
}

template <bool AKC, bool BKC, bool VEC, bool BF16, int TM, int PF = 1, bool CLAMP = true>
__global__ __launch_bounds__(NT) void gemm_kernel(const GemmP p) {
  constexpr int LD = BM * TM + 1;
  __shared__ float As[BK * LD];
  __shared__ float Bs[BK * LD];
  gemm_tile<AKC, BKC, VEC, BF16, TM, PF, CLAMP>(p, blockIdx.x, blockIdx.y, blockIdx.z, As, Bs);
}

// Up to four independent fp32 problems as ONE launch (mcl_gemm_group): the weight gradients and the data gradient of a layer's
// backward are a few dozen 64 x 64 tiles each -- launched one by one they leave most of the chip idle three times over.  Block b
// belongs to the problem whose tile range [first[i], first[i + 1]) holds it; each problem keeps its own operand layout, epilogue
// and tile order, so a grouped launch is bit-identical to the separate ones.
constexpr int GROUP_MAX = 4;
struct GemmGroup {
  GemmP p[GROUP_MAX];
  int first[GROUP_MAX + 1];
  int layout[GROUP_MAX];       // bit 0: A contiguous along k, bit 1: B contiguous along k, bit 2: 16-byte loads allowed
  int n;
};

template <bool VEC>
__device__ __forceinline__ void group_dispatch(const GemmP& p, int lay, int bx, int by, float* As, float* Bs) {
  switch (lay & 3) {
    case 3: gemm_tile<true, true, VEC, false, 1>(p, bx, by, 0, As, Bs); break;
    case 1: gemm_tile<true, false, VEC, false, 1>(p, bx, by, 0, As, Bs); break;
    case 2: gemm_tile<false, true, VEC, false, 1>(p, bx, by, 0, As, Bs); break;
    default: gemm_tile<false, false, VEC, false, 1>(p, bx, by, 0, As, Bs); break;
  }
}

__global__ __launch_bounds__(NT) void gemm_group_kernel(const GemmGroup g) {
  constexpr int LD = BM + 1;
  __shared__ float As[BK * LD];
  __shared__ float Bs[BK * LD];
  int i = 0;
  while (i + 1 < g.n && (int)blockIdx.x >= g.first[i + 1]) ++i;
  const GemmP& p = g.p[i];
  const int t = (int)blockIdx.x - g.first[i];
  const int tn = (p.N + BN - 1) / BN;
  if (g.layout[i] & 4) group_dispatch<true>(p, g.layout[i], t % tn, t / tn, As, Bs);
  else                 group_dispatch<false>(p, g.layout[i], t % tn, t / tn, As, Bs);
}

// Split-K second pass: fixed-order sum of the K slices + the same epilogue as the one-pass kernel (deterministic).
__global__ __launch_bounds__(256) void gemm_splitk_epilogue_kernel(const GemmP p) {
  const long long MN = (long long)p.M * p.N, total = MN * p.batch;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
    const int bz = (int)(q / MN);
    const long long e = q - (long long)bz * MN;
    const int row = (int)(e / p.N), col = (int)(e - (long long)row * p.N);
    // (all of an element's loads issued together -- slices eight at a time from clamped indices, added in slice order -- and the
    //  epilogue's operands before the first use: a load per loop trip or per branch is a round trip of its own)
    float* C = p.C + (long long)bz * p.sCb + (long long)row * p.ldc + col;
    float xa = 0.0f, xr = 0.0f, xc = 0.0f;
    if (p.flags & MCL_EPI_GELU_BWD) xa = p.aux[(long long)row * p.ldaux + col];
    if (p.resid) xr = p.resid[(long long)bz * p.sRb + (long long)row * p.ldr + col];
    if (p.flags & MCL_EPI_ACCUM) xc = *C;
    const float bias = p.bias ? p.bias[col] : 0.0f;
    float a = 0.0f;
    // (batches of 2 / 4 / 8 by the slice count: a clamped surplus load is a real load -- with two slices of a 9 MB weight
    //  gradient each, eight-wide batches read four times the bytes)
#define MCL_SK_BATCH(NB)                                                                                              \
  for (int s0 = 0; s0 < p.ksplit; s0 += NB) {                                                                         \
    float w[NB];                                                                                                      \
    _Pragma("unroll") for (int i = 0; i < NB; ++i)                                                                    \
        w[i] = p.ws[((long long)min(s0 + i, p.ksplit - 1) * p.batch + bz) * MN + e];                                  \
    _Pragma("unroll") for (int i = 0; i < NB; ++i) if (s0 + i < p.ksplit) a += w[i];                                  \
  }
    if (p.ksplit <= 2) MCL_SK_BATCH(2) else if (p.ksplit <= 4) MCL_SK_BATCH(4) else if (p.ksplit % 8 == 0 || p.ksplit > 16)
      MCL_SK_BATCH(8) else MCL_SK_BATCH(4)
#undef MCL_SK_BATCH
    float v = p.alpha * a + bias;
    if (p.pre_out) p.pre_out[(long long)row * p.ldp + col] = v;
    if (p.flags & MCL_EPI_GELU) v = gelu_erf(v);
    if (p.flags & MCL_EPI_GELU_BWD) v *= gelu_erf_grad(xa);
    if (p.resid) v += xr;
    if (p.flags & MCL_EPI_ACCUM) v += xc;
    *C = v;
  }
}

// 128 x 128 tiles?
bool mcl_gemm_wide_tiles(const GemmP& p) {
  const int batch = p.batch;
  // 128 x 128 tiles once they alone fill the chip twice over (and no split-K: that is the skinny-problem form)
  const long long big_tiles = (long long)((p.N + 2 * BN - 1) / (2 * BN)) * ((p.M + 2 * BM - 1) / (2 * BM)) * batch;
  // ... or a long reduction cut into K slices over a mid-sized output (mcl_gemm_auto_ksplit's second rule)
  const long long tiles64 = (long long)((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM) * batch;
  // (only where a 128-wide tile is not mostly padding: the generic convolutions' N = 32 / 128 outputs stay on 64 x 64 tiles --
  //  fp32 DenseNet step 105 vs 115 ms with 128 x 128 tiles there)
  const bool wide = p.N >= 640 && p.M >= 256 && ((p.N + 127) / 128) * 128 - p.N <= p.N / 8;
  return wide && ((p.ksplit == 1 && big_tiles >= 512) || (p.ksplit > 1 && tiles64 >= 128 && p.K >= 8192 && big_tiles * p.ksplit >= 256));
}

// fp32 launches of at most four 64 x 64-tile workgroups per CU: the clamped + masked operand loads
bool gemm_skinny(const GemmP& p, bool bf16) {
  if (bf16 || mcl_gemm_wide_tiles(p)) return false;
  return (long long)((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM) * p.batch * p.ksplit <= 1024;
}

template <bool AKC, bool BKC, bool VEC>
void launch2(const GemmP& p, int batch, bool bf16, hipStream_t st) {
  if (mcl_gemm_wide_tiles(p)) {
    dim3 grid((p.N + 2 * BN - 1) / (2 * BN), (p.M + 2 * BM - 1) / (2 * BM), batch * p.ksplit), block(NT);
    if (bf16) hipLaunchKernelGGL((gemm_kernel<AKC, BKC, VEC, true, 2, 1, false>), grid, block, 0, st, p);
    else      hipLaunchKernelGGL((gemm_kernel<AKC, BKC, VEC, false, 2, 1, false>), grid, block, 0, st, p);
    return;
  }
  dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM, batch * p.ksplit), block(NT);
  // the unconditional (clamped + masked) operand loads for launches of at most four workgroups per CU -- the skinny problems of the
  // spot path, whose few waves cannot cover serialised loads; the many-tile problems keep the cheaper branchy staging
  const bool skinny = gemm_skinny(p, bf16);
  if (bf16) {
    hipLaunchKernelGGL((gemm_kernel<AKC, BKC, VEC, true, 1, 1, false>), grid, block, 0, st, p);
  } else if (skinny) {
    hipLaunchKernelGGL((gemm_kernel<AKC, BKC, VEC, false, 1, 1, true>), grid, block, 0, st, p);
  } else {
    hipLaunchKernelGGL((gemm_kernel<AKC, BKC, VEC, false, 1, 1, false>), grid, block, 0, st, p);
  }
}

template <bool AKC, bool BKC>
void launch1(const GemmP& p, int batch, bool vec, bool bf16, hipStream_t st) {
  if (vec) launch2<AKC, BKC, true>(p, batch, bf16, st);
  else     launch2<AKC, BKC, false>(p, batch, bf16, st);
}

inline bool aligned16(const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15u) == 0; }

}  // namespace

// The struct travels by pointer and grows between ABI versions: never read past what the caller declared.
static constexpr uint32_t kGemmArgsMin = (uint32_t)(offsetof(mcl_gemm_args, workspace) + sizeof(float*));
extern "C" uint32_t mcl_gemm_args_size(void) { return (uint32_t)sizeof(mcl_gemm_args); }
extern "C" uint32_t mcl_gemm_args_min_size(void) { return kGemmArgsMin; }

// Argument check + the kernel-side descriptor of one problem.  `vec`: 16-byte operand loads allowed.
// (the clamped staging needs a 16-byte chunk to lie inside an operand or outside it as a whole: the extent along the contiguous
//  dimension -- and the start of every split-K slice, a multiple of BK -- a multiple of 4)
static bool gemm_vec_strict(const mcl_gemm_args* a, bool akc, bool bkc) {
  return ((akc ? a->K : a->M) % 4 == 0) && ((bkc ? a->K : a->N) % 4 == 0);
}

static int gemm_prepare(const mcl_gemm_args* caller_args, mcl_gemm_args* a, GemmP& p, bool& akc, bool& bkc, bool& vec) {
  if (!caller_args) return MCL_EINVAL;
  const uint32_t sz = caller_args->struct_size;
  if (sz < kGemmArgsMin) return MCL_EINVAL;
  memset(a, 0, sizeof(*a));
  memcpy(a, caller_args, sz < sizeof(*a) ? sz : sizeof(*a));   // fields beyond the caller's size stay zero / NULL
  if (!a->A || !a->B || (!a->C && !a->flt_thr)) return MCL_EINVAL;
  if (a->flt_thr && (!a->flt_cnt || !a->flt_val || !a->flt_idx || a->flt_cap <= 0 || a->batch != 1 || a->ksplit > 1 || a->bias ||
                     a->resid || a->pre_out || a->flags))
    return MCL_EINVAL;
  if (a->M <= 0 || a->N <= 0 || a->K <= 0 || a->batch <= 0) return MCL_EINVAL;
  akc = (a->sAk == 1);
  bkc = (a->sBk == 1);
  const bool amc = (a->sAm == 1), bnc = (a->sBn == 1);
  if (!(akc || amc) || !(bkc || bnc)) return MCL_EINVAL;
  if ((a->pre_out || (a->flags & MCL_EPI_GELU_BWD)) && a->batch != 1) return MCL_EINVAL;
  if ((a->flags & MCL_EPI_GELU_BWD) && !a->aux) return MCL_EINVAL;
  if (a->compute != MCL_COMPUTE_F32 && a->compute != MCL_COMPUTE_BF16) return MCL_EUNSUPPORTED;
  if (a->batch > 65535) return MCL_EUNSUPPORTED;
  const int ksplit = a->ksplit > 1 ? a->ksplit : 1;
  if (ksplit > 1 && (!a->workspace || (long long)a->batch * ksplit > 65535 || ksplit > 64)) return MCL_EINVAL;
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.A = a->A; p.sAm = a->sAm; p.sAk = a->sAk; p.sAb = a->sAb;
  p.B = a->B; p.sBk = a->sBk; p.sBn = a->sBn; p.sBb = a->sBb;
  p.C = a->C; p.ldc = a->ldc; p.sCb = a->sCb;
  p.alpha = a->alpha; p.flags = a->flags; p.bias = a->bias;
  p.resid = a->resid; p.ldr = a->ldr; p.sRb = a->sRb;
  p.pre_out = a->pre_out; p.ldp = a->ldp; p.aux = a->aux; p.ldaux = a->ldaux;
  p.batch = a->batch; p.ksplit = ksplit; p.ws = a->workspace;
  p.cnt = ksplit > 1 ? a->counters : nullptr;       // (mcl_gemm drops it again for a launch that is not "skinny")
  p.flt_thr = a->flt_thr; p.flt_cnt = a->flt_cnt; p.flt_val = a->flt_val; p.flt_idx = a->flt_idx; p.flt_cap = a->flt_cap;
  p.kchunk = ksplit > 1 ? (((a->K + ksplit - 1) / ksplit + BK - 1) / BK) * BK : a->K;
  // (the k-contiguous reading is preferred when a dimension of extent-1 stride is ambiguous)
  const long long lda = akc ? a->sAm : a->sAk, ldb = bkc ? a->sBn : a->sBk;
  vec = aligned16(a->A) && aligned16(a->B) && (lda % 4 == 0) && (ldb % 4 == 0) && (a->sAb % 4 == 0) && (a->sBb % 4 == 0);
  return MCL_OK;
}

// mcl_gemm_group: n <= 4 problems (fp32 compute, batch 1, no split-K, no filter epilogue) as one launch; `args` is an array of
// structs `args[0].struct_size` bytes apart.
extern "C" int mcl_gemm_group(const mcl_gemm_args* args, int32_t n, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!args || n <= 0 || n > GROUP_MAX) return MCL_EINVAL;
  const uint32_t stride = args->struct_size;
  if (stride < kGemmArgsMin) return MCL_EINVAL;
  GemmGroup g;
  memset(&g, 0, sizeof(g));
  g.n = n;
  int tiles = 0;
  for (int i = 0; i < n; ++i) {
    const mcl_gemm_args* ai = reinterpret_cast<const mcl_gemm_args*>(reinterpret_cast<const char*>(args) + (size_t)i * stride);
    if (ai->struct_size != stride) return MCL_EINVAL;
    mcl_gemm_args local;
    bool akc, bkc, vec;
    const int rc = gemm_prepare(ai, &local, g.p[i], akc, bkc, vec);
    if (rc != MCL_OK) return rc;
    if (local.compute != MCL_COMPUTE_F32 || local.batch != 1 || local.ksplit > 1 || local.flt_thr) return MCL_EUNSUPPORTED;
    g.layout[i] = (akc ? 1 : 0) | (bkc ? 2 : 0) | (vec && gemm_vec_strict(&local, akc, bkc) ? 4 : 0);
    g.first[i] = tiles;
    tiles += ((local.M + BM - 1) / BM) * ((local.N + BN - 1) / BN);
  }
  for (int i = n; i <= GROUP_MAX; ++i) g.first[i] = tiles;
  hipLaunchKernelGGL(gemm_group_kernel, dim3(tiles), dim3(NT), 0, mcl_stream(stream), g);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_gemm(const mcl_gemm_args* caller_args, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  mcl_gemm_args local;
  GemmP p;
  bool akc, bkc, vec;
  const int rc = gemm_prepare(caller_args, &local, p, akc, bkc, vec);
  if (rc != MCL_OK) return rc;
  const mcl_gemm_args* a = &local;
  const int ksplit = p.ksplit;
  const bool AKC = akc, BKC = bkc;
  hipStream_t st = mcl_stream(stream);
  const bool bf16 = a->compute == MCL_COMPUTE_BF16;
  if (gemm_skinny(p, bf16)) vec = vec && gemm_vec_strict(a, akc, bkc);
  else p.cnt = nullptr;
  if (AKC && BKC) launch1<true, true>(p, a->batch, vec, bf16, st);
  else if (AKC && !BKC) launch1<true, false>(p, a->batch, vec, bf16, st);
  else if (!AKC && BKC) launch1<false, true>(p, a->batch, vec, bf16, st);
  else launch1<false, false>(p, a->batch, vec, bf16, st);
  if (ksplit > 1 && !(p.cnt && !mcl_gemm_wide_tiles(p))) {       // (the one-launch split-K exists for the 64 x 64 tiles)
    const long long total = (long long)a->M * a->N * a->batch;
    const unsigned nb = (unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(gemm_splitk_epilogue_kernel, dim3(nb), dim3(256), 0, st, p);
  }
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

// K slices that bring a skinny problem (the spot path: M = batch of 128 spots) to >= ~256 workgroups: a 64 x 64 tile
// over K = 1000 is a chain of 500 dependent fp32 MFMAs (13 us) however few tiles there are.  1 = do not split.
extern "C" int32_t mcl_gemm_auto_ksplit(int32_t M, int32_t N, int32_t K, int32_t batch) {
  if (M <= 0 || N <= 0 || K <= 0 || batch <= 0) return 1;
  const long long tiles = (long long)((M + BM - 1) / BM) * ((N + BN - 1) / BN) * batch;
  if (tiles >= 128 && tiles < 1024 && K >= 2048) {
    // a mid-sized output over a LONG reduction (weight gradients of the fp32 image encoders: 768 x 768 ... 3072 outputs over 6 400 -
    // 25 216 token rows): a few hundred workgroups would each walk thousands of k
    long long ks;
    if (K >= 8192) {      // 128 x 128 tiles (launch2): slices for ~384 of them
      const long long big = (long long)((M + 2 * BM - 1) / (2 * BM)) * ((N + 2 * BN - 1) / (2 * BN)) * batch;
      ks = (384 + big - 1) / big;
      if (ks > K / 512) ks = K / 512;
    } else {              // 64 x 64 tiles: slices for ~1024 of them (measured on ViT-B/32, 6 400 rows: 40.5 vs 43.3 ms/step)
      ks = 1024 / tiles;
      if (ks > K / 1024) ks = K / 1024;
    }
    if (ks > 8) ks = 8;
    if (ks * batch > 65535) ks = 65535 / batch;
    return ks < 1 ? 1 : (int32_t)ks;
  }
  if (tiles >= 128 || K < 256) return 1;
  if (K >= 16384) {
    // a handful of output tiles over a reduction of 10^4 - 10^6 (weight gradients of the generic fp32 convolutions: C_out x C_in k^2
    // over every pixel of the batch): up to 64 slices -- eight of them left 8 - 288 workgroups walking 50 000 pixels each
    long long ks = 1024 / tiles;
    if (ks > K / 1024) ks = K / 1024;
    if (ks > 64) ks = 64;
    if (ks * batch > 65535) ks = 65535 / batch;
    return ks < 1 ? 1 : (int32_t)ks;
  }
  long long ks = 256 / tiles;
  if (ks > K / 128) ks = K / 128;
  if (ks > 8) ks = 8;
  if (ks * batch > 65535) ks = 65535 / batch;
  return ks < 1 ? 1 : (int32_t)ks;
}

extern "C" int64_t mcl_gemm_workspace_floats(int32_t M, int32_t N, int32_t batch, int32_t ksplit) {
  if (M <= 0 || N <= 0 || batch <= 0) return -1;
  return ksplit > 1 ? (int64_t)ksplit * batch * M * N : 0;
}
