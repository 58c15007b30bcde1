// bf16 GEMM for the ViT image encoder (/root/reference/model.py:104-116: timm vit_base_patch{32,16}_224) and its
// backward: every dense contraction of a Transformer block -- QKV / projection / MLP linears (forward, data gradient,
// weight gradient) and the attention products Q K^T, P V, P^T dO, dO V^T, dS K, dS^T Q -- is this one kernel.
//
//     C[b][M][N] = epilogue( alpha * sum_k A[b](m, k) * B[b](k, n) )          fp32 accumulate on v_mfma_f32_32x32x16_bf16
//
// Operand storage is described per operand, so no transposed copy is ever made:
//     A_KMAJOR = 0:  A(m, k) = A[m*lda + k]   (k contiguous; activations [tokens][features])
//     A_KMAJOR = 1:  A(m, k) = A[k*lda + m]   (reduction-major; dY for a weight gradient, P for dV = P^T dO)
//     B_KMAJOR = 0:  B(k, n) = B[n*ldb + k]   (k contiguous; nn.Linear weights [out][in] in the forward, K in Q K^T)
//     B_KMAJOR = 1:  B(k, n) = B[k*ldb + n]   (reduction-major; the weight in a data gradient, V in P V)
//
// MI355X mapping: 256 x 256 output tile per workgroup (8 waves, 2 x 4, each 128 x 64 = eight 32 x 32 MFMA blocks, 128
// accumulator registers), K in steps of 64: 128 flop per byte staged -- a 128 x 128 tile (64 flop/B) needs ~39 TB/s of
// L2 -> LDS traffic at MFMA peak and measured 335 TF/s.  Operand tiles (two 16 KB sub-tiles of 128 rows / columns each)
// arrive by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip) into a double-buffered 2 x 64 KB LDS stage exactly as
// they lie in HBM -- k-contiguous tiles as 128-byte rows read back with
// ds_read_b128, reduction-major tiles as 256-byte rows read back with the transposing ds_read_b64_tr_b16 -- with the
// bank-conflict-avoiding XOR swizzle applied on the DMA's SOURCE address.  Ragged M / N / K: out-of-range rows and
// chunks are fetched from clamped (valid) addresses and the reduction tail is zeroed in the A fragment, so arbitrary
// sizes (197 tokens) need no padding copies -- only 16-byte aligned rows (ld % 8 == 0).
// Workgroup -> tile mapping is XCD-aware: the column tiles of one row panel run back to back on one XCD, so the A panel
// is fetched once into that XCD's L2.  Split-K (weight gradients: K = 50 k tokens, few output tiles) writes fp32 slabs
// that mcl_launch_wrw_merge adds in fixed order: deterministic, no atomics.
// Epilogue (through LDS, so that HBM sees 16-byte row chunks): + bias[n], exact-erf GELU (optionally also storing the
// pre-activation), * gelu'(aux), + residual; bf16 or fp32 output.
// Round 4: problems made of interior 256 x 256 tiles only (every ViT linear at batch 256: 50432 = 197 x 256 tokens) take
// gemm_bf16_stag_kernel -- same tile, same fragments, same k order (bit-identical results), but the two wave groups run one
// barrier apart over a four-slot half-K ring filled three ahead: +6...19 % on the ViT shapes (below).
#include "common.h"
#include <stdlib.h>

namespace {

typedef unsigned short bf16_t;
typedef short v4s __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 64;
constexpr int SUB_B = 16384;                  // bytes per 128-row (or 128-column) operand sub-tile

#define MCL_LDSP(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ void glds16(const void* src, unsigned dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(src), "s"(dst)
               : "memory");
}

// saddr form: wave-uniform 64-bit base in SGPRs + per-lane 32-bit byte offset (M0 is used by nothing else here)
__device__ __forceinline__ void glds16s(const void* sbase, unsigned voff, unsigned dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0"
               :
               : "s"(sbase), "v"(voff), "s"(dst)
               : "memory", "m0");
}

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xFFFF0000u); }

// ---- k-contiguous tile: [128 rows][64 k] bf16 = 128-byte rows, 16-byte chunk c of row r at physical chunk
// c ^ ((r >> 1) & 7): 16 consecutive rows of one logical chunk land on 16 distinct 16-byte slots of the 256-byte bank row.
// Fragment of a 32-row block for the k-step kk (16 deep): lane l reads row (l & 31), k = kk + 8*(l >> 5) .. + 7.
__device__ __forceinline__ bf16x8 frag_kc(const unsigned char* tile, int row, int kk, int h) {
  const int c = (kk >> 3) + h;
  return *reinterpret_cast<const bf16x8*>(tile + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
}
// ---- k-contiguous HALF tile of the staggered ring: [128 rows][32 k] bf16 = 64-byte rows, chunk c (0..3) of row r at physical
// chunk c ^ ((r >> 2) & 3): a ds_read_b128 lane group (16 lanes, rows {0-3, 12-15, 20-27} of one logical chunk) lands on the 16
// distinct 16-byte slots of the 256-byte bank row (slot = 4 (r & 3) + physical chunk).
__device__ __forceinline__ bf16x8 frag_kc32(const unsigned char* tile, int row, int kk, int h) {
  const int c = (kk >> 3) + h;
  return *reinterpret_cast<const bf16x8*>(tile + row * 64 + ((c ^ ((row >> 2) & 3)) << 4));
}
// ---- reduction-major tile: [64 k][128 cols] bf16 = 256-byte rows, chunk c of row r at c ^ ((r & 3) << 2); fragment =
// 8 consecutive k of column cbase + (lane & 31) through the transposing read (see csrc/wrw_fused.hip frag_sw)
__device__ __forceinline__ bf16x8 frag_km(const unsigned char* tile, int kbase, int cbase, int lane) {
  const int i = lane & 15, q = i >> 2;
  const int lchunk = (cbase + 16 * ((lane >> 4) & 1)) / 8 + ((i & 3) >> 1);
  const unsigned char* p = tile + (kbase + q) * 256 + ((lchunk ^ (q << 2)) << 4) + (i & 1) * 8;
  const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)p);
  const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(p + 4 * 256));
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

struct GemmB {
  const bf16_t* A; long long lda, sAb, sAb2;
  const bf16_t* B; long long ldb, sBb, sBb2;
  void* C; long long ldc, sCb, sCb2;            // bf16 (or fp32 when out_f32)
  int M, N, K, batch, batch2;                   // batch index bi -> (bi / batch2, bi % batch2): strides s?b, s?b2
  float alpha;
  const float* bias;                            // [N] or null
  const bf16_t* resid; long long ldr, sRb;      // + resid[m][n] (bf16) or null
  const bf16_t* aux; long long ldaux;           // GELU_BWD: pre-activation [m][n]
  bf16_t* pre_out; long long ldp;               // GELU: also store the pre-activation (bf16) or null
  int gelu, gelu_bwd, out_f32;
  int tm, tn, ksplit;                           // tiles in M, N; K slices (slab s at C + s * slab_stride floats)
  long long k_per_split, slab_stride;
  int dbg;                                      // experiment switches of the pipe kernel (MCL_GEMM_DBG; 0 in production)
};

// GELU (exact-erf form, nn.GELU's default) and its derivative for bf16 outputs.  Phi(x) = 0.5 erfc(-x / sqrt 2) through Abramowitz &
// Stegun 7.1.26 (erfc(z) = t (a1 + t (a2 + t (a3 + t (a4 + t a5)))) exp(-z^2), t = 1 / (1 + p z), |error| <= 1.5e-7 on erf): one
// v_rcp_f32, one v_exp_f32 and 9 multiply-adds instead of libdevice's erff (~40 instructions with range branches) -- the epilogue of
// the ViT's fc1 (50 432 x 3072 outputs) spent more time in erff than the tile in its K loop.  The error is 2^-14 of the bf16
// rounding that follows; the fp32 paths (csrc/gemm.hip, the spot encoder) keep erff.  exp(-z^2) = exp(-x^2 / 2) is shared with
// the density term of the derivative.
__device__ __forceinline__ void gelu_parts(float x, float& Phi, float& ex) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));   // (v_rcp_f32, 1 ulp: __frcp_rn expands to the 10-instruction IEEE division)
  ex = __expf(-z * z);
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float half_erfc = 0.5f * poly * ex;
  Phi = x < 0.0f ? half_erfc : 1.0f - half_erfc;
}
__device__ __forceinline__ float gelu_f(float x) {
  float Phi, ex;
  gelu_parts(x, Phi, ex);
  return x * Phi;
}
__device__ __forceinline__ float gelu_g(float x) {
  float Phi, ex;
  gelu_parts(x, Phi, ex);
  return fmaf(x * 0.39894228040143267794f, ex, Phi);
}

// ---- epilogue of a wave's (32 NI) x 64 accumulator tile, through LDS in halves of 64 rows (64 x 64 fp32 = 16 KB per wave: the
// operand stages, free after the K loop's last barrier; a wave reads back only what it wrote, so no workgroup barrier).
// Loads, stores and LDS-DMA pieces retire in issue order on one counter, so a load waited for after a store drains that store:
// the bias is loaded once before the first store, and the residual / gelu' operand rows of iteration i + 1 are requested before
// the stores of iteration i (round 4; the per-iteration loads used to wait out the previous iteration's store acknowledgements).
template <int NI, bool HAS_X>
__device__ __forceinline__ void gemm_epilogue_impl(const GemmB& g, f32x16 (&acc)[NI][2], unsigned char* lds, int wave, int lane,
                                                   int mw0, int nw, int ks, long long c_off, int b1) {
  constexpr int EP = 64;
  const int h = lane >> 5, l31 = lane & 31;
  float* et = reinterpret_cast<float*>(lds) + wave * (64 * EP);
  const int cch = lane & 7;                          // 8 chunks of 8 columns per 64-column row
  const int n = nw + cch * 8;
  float bv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bv[e] = (g.bias && n + e < g.N) ? g.bias[n + e] : 0.0f;
  constexpr bool has_x = HAS_X;                      // residual / gelu' operand rows (compile-time: the plain form carries no
                                                     // prefetch registers and no per-iteration moves)
  auto load_x = [&](int m, u32x4& xa, u32x4& xr) {
    if (m >= g.M || n >= g.N) return;
    if (g.gelu_bwd) xa = *reinterpret_cast<const u32x4*>(g.aux + (long long)m * g.ldaux + n);
    if (g.resid) xr = *reinterpret_cast<const u32x4*>(g.resid + (long long)b1 * g.sRb + (long long)m * g.ldr + n);
  };
#pragma unroll 1
  for (int half = 0; half < NI / 2; ++half) {
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = ii * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          et[row * EP + j * 32 + l31] = (half == 0 ? acc[ii][j][r] : acc[NI - 2 + ii][j][r]) * g.alpha;
        }
    const int mw = mw0 + half * 64;
    u32x4 xa = {0, 0, 0, 0}, xr = {0, 0, 0, 0}, xa_n = {0, 0, 0, 0}, xr_n = {0, 0, 0, 0};
    if (has_x) load_x(mw + (lane >> 3), xa, xr);
#pragma unroll 2
    for (int rr = lane >> 3; rr < 64; rr += 8) {
      const int m = mw + rr;
      if (has_x && rr + 8 < 64) load_x(m + 8, xa_n, xr_n);
      if (m < g.M && n < g.N) {
        const float4 v0 = *reinterpret_cast<const float4*>(et + rr * EP + cch * 8);
        const float4 v1 = *reinterpret_cast<const float4*>(et + rr * EP + cch * 8 + 4);
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        if (g.bias) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += bv[e];
        }
        if (g.out_f32) {                               // split-K slab / fp32 result: no activation
          float* o = reinterpret_cast<float*>(g.C) + (long long)ks * g.slab_stride + c_off + (long long)m * g.ldc + n;
          if (n + 8 <= g.N) {
            *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
          } else {
            for (int e = 0; e < 8 && n + e < g.N; ++e) o[e] = v[e];
          }
        } else {
          if (g.gelu == 2) {                           // GELU, and pre_out <- gelu'(pre-activation): one exp / rcp serves both
            float gr[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              float Phi, ex;
              gelu_parts(v[e], Phi, ex);
              gr[e] = fmaf(v[e] * 0.39894228040143267794f, ex, Phi);
              v[e] *= Phi;
            }
            *reinterpret_cast<u32x4*>(g.pre_out + (long long)m * g.ldp + n) =
                u32x4{pack_bf16(gr[0], gr[1]), pack_bf16(gr[2], gr[3]), pack_bf16(gr[4], gr[5]), pack_bf16(gr[6], gr[7])};
          } else {
            if (g.pre_out) {
              *reinterpret_cast<u32x4*>(g.pre_out + (long long)m * g.ldp + n) =
                  u32x4{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
            }
            if (g.gelu) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = gelu_f(v[e]);
            }
          }
          if (g.gelu_bwd == 2) {                       // aux holds the stored derivative
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[2 * e] *= bf_lo(xa[e]);
              v[2 * e + 1] *= bf_hi(xa[e]);
            }
          } else if (g.gelu_bwd) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[2 * e] *= gelu_g(bf_lo(xa[e]));
              v[2 * e + 1] *= gelu_g(bf_hi(xa[e]));
            }
          }
          if (g.resid) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[2 * e] += bf_lo(xr[e]);
              v[2 * e + 1] += bf_hi(xr[e]);
            }
          }
          // 16-byte store; columns beyond N inside the chunk fall into the row's padding (ldc >= round_up(N, 8) is required)
          *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(g.C) + c_off + (long long)m * g.ldc + n) =
              u32x4{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
        }
      }
      if (has_x) { xa = xa_n; xr = xr_n; }
    }
  }
}
template <int NI>
__device__ __forceinline__ void gemm_epilogue(const GemmB& g, f32x16 (&acc)[NI][2], unsigned char* lds, int wave, int lane,
                                              int mw0, int nw, int ks, long long c_off, int b1) {
  if (g.gelu_bwd || g.resid) gemm_epilogue_impl<NI, true>(g, acc, lds, wave, lane, mw0, nw, ks, c_off, b1);
  else gemm_epilogue_impl<NI, false>(g, acc, lds, wave, lane, mw0, nw, ks, c_off, b1);
}

// SUBS = 2: 256 x 256 tile, 8 waves (2 x 4, 128 x 64 each) -- the large linears.  SUBS = 1: 128 x 128 tile, 4 waves
// (2 x 2, 64 x 64 each), two workgroups per CU -- the batched attention products (197 x 197 x 64 per image and head:
// a 256 x 256 tile would be 41 % padding and one workgroup per CU).
template <bool A_KMAJOR, bool B_KMAJOR, int SUBS>
__global__ __launch_bounds__(256 * SUBS) void gemm_bf16_kernel(GemmB g) {
  constexpr int BM = 128 * SUBS, BN = 128 * SUBS;
  constexpr int TILE_B = SUBS * SUB_B, STAGE_B = 2 * TILE_B;
  constexpr int NI = 2 * SUBS;                 // 32-row blocks per wave
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];      // 2 stages x (A tile + B tile)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = SUBS == 2 ? wave >> 2 : wave >> 1, wn = SUBS == 2 ? wave & 3 : wave & 1;
  const int h = lane >> 5, l31 = lane & 31;
  int b, ks, tx, ty;
  if (g.tm >= 8) {
    // XCD-aware decode (block id % 8 = XCD): row panel = (q / tn) * 8 + xcd, column tile = q % tn
    const int per_batch = ((g.tm + 7) / 8) * 8 * g.tn * g.ksplit;
    const int bid = blockIdx.x % per_batch;
    b = blockIdx.x / per_batch;
    const int xcd = bid & 7, q = bid >> 3;
    ks = q % g.ksplit;
    const int q2 = q / g.ksplit;
    tx = q2 % g.tn;
    ty = (q2 / g.tn) * 8 + xcd;
    if (ty >= g.tm) return;
  } else {                                     // few row panels (batched small problems): plain decode, no padding
    const int per_batch = g.tm * g.tn * g.ksplit;
    const int bid = blockIdx.x % per_batch;
    b = blockIdx.x / per_batch;
    ks = bid % g.ksplit;
    const int q2 = bid / g.ksplit;
    tx = q2 % g.tn;
    ty = q2 / g.tn;
  }
  const int m0 = ty * BM, n0 = tx * BN;
  const long long k_begin = (long long)ks * g.k_per_split;
  const long long k_end = min((long long)g.K, k_begin + g.k_per_split);
  const int nt = (int)((k_end - k_begin + BK - 1) / BK);
  const int b1 = b / g.batch2, b2 = b % g.batch2;
  const bf16_t* A = g.A + (long long)b1 * g.sAb + (long long)b2 * g.sAb2;
  const bf16_t* B = g.B + (long long)b1 * g.sBb + (long long)b2 * g.sBb2;
  const long long c_off = (long long)b1 * g.sCb + (long long)b2 * g.sCb2;
  const unsigned lds_base = (unsigned)(size_t)MCL_LDSP(lds);

  // ---- DMA geometry: 32 pieces of 1 KB per operand tile (16 per sub-tile); wave w issues pieces w, w+8, w+16, w+24.
  // Out-of-range rows / chunks are clamped to valid addresses (finite garbage); the reduction tail is zeroed in the A
  // fragment, M / N tails are never stored.
  auto dma_tile = [&](int t, int stage) {
    const long long k0 = k_begin + (long long)t * BK;
    const unsigned dA = lds_base + stage * STAGE_B, dB = dA + TILE_B;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = wave + 4 * SUBS * u, sub = p >> 4, pp = p & 15;
      if (!A_KMAJOR) {                                             // 8 rows (m) x 128 B
        const int row = 8 * pp + (lane >> 3);
        const int lc = (lane & 7) ^ ((row >> 1) & 7);
        long long m = m0 + sub * 128 + row;  m = m < g.M ? m : g.M - 1;
        long long k = k0 + lc * 8;  k = k < g.K ? k : 0;
        glds16(A + m * g.lda + k, __builtin_amdgcn_readfirstlane(dA + p * 1024));
      } else {                                                     // 4 rows (k) x 256 B
        const int row = 4 * pp + (lane >> 4);
        const int lc = (lane & 15) ^ (((lane >> 4) & 3) << 2);
        long long k = k0 + row;  k = k < g.K ? k : g.K - 1;
        long long m = m0 + sub * 128 + lc * 8;  m = m < g.M ? m : 0;
        glds16(A + k * g.lda + m, __builtin_amdgcn_readfirstlane(dA + p * 1024));
      }
      if (!B_KMAJOR) {
        const int row = 8 * pp + (lane >> 3);
        const int lc = (lane & 7) ^ ((row >> 1) & 7);
        long long n = n0 + sub * 128 + row;  n = n < g.N ? n : g.N - 1;
        long long k = k0 + lc * 8;  k = k < g.K ? k : 0;
        glds16(B + n * g.ldb + k, __builtin_amdgcn_readfirstlane(dB + p * 1024));
      } else {
        const int row = 4 * pp + (lane >> 4);
        const int lc = (lane & 15) ^ (((lane >> 4) & 3) << 2);
        long long k = k0 + row;  k = k < g.K ? k : g.K - 1;
        long long n = n0 + sub * 128 + lc * 8;  n = n < g.N ? n : 0;
        glds16(B + k * g.ldb + n, __builtin_amdgcn_readfirstlane(dB + p * 1024));
      }
    }
  };

  // Interior tiles (every row / column of the tile exists, K range a multiple of BK): the pieces of a wave differ only
  // by a wave-uniform row offset, so the source address is an SGPR base (advanced per K-tile) + ONE per-lane 32-bit
  // offset per operand -- ~5 instructions per piece instead of ~20 (per-lane 64-bit pointers with clamps).
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const bool interior = m0 + BM <= g.M && n0 + BN <= g.N && (k_end - k_begin) % BK == 0 &&
                        g.lda * 16 < (1ll << 31) && g.ldb * 16 < (1ll << 31);
  unsigned voffA, voffB;
  {
    const int pp1 = wave_s & 1;                                    // parity of every piece index of this wave
    if (!A_KMAJOR) voffA = (unsigned)((lane >> 3) * g.lda * 2) + (unsigned)((((lane & 7) ^ ((4 * pp1 + (lane >> 4)) & 7))) << 4);
    else voffA = (unsigned)((lane >> 4) * g.lda * 2) + (unsigned)(((lane & 15) ^ (((lane >> 4) & 3) << 2)) << 4);
    if (!B_KMAJOR) voffB = (unsigned)((lane >> 3) * g.ldb * 2) + (unsigned)((((lane & 7) ^ ((4 * pp1 + (lane >> 4)) & 7))) << 4);
    else voffB = (unsigned)((lane >> 4) * g.ldb * 2) + (unsigned)(((lane & 15) ^ (((lane >> 4) & 3) << 2)) << 4);
  }
  auto dma_tile_fast = [&](int t, int stage) {
    const long long k0 = k_begin + (long long)t * BK;
    const unsigned dA = lds_base + stage * STAGE_B, dB = dA + TILE_B;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = wave_s + 4 * SUBS * u, sub = p >> 4, pp = p & 15;
      const unsigned char* pa = reinterpret_cast<const unsigned char*>(
          !A_KMAJOR ? A + (long long)(m0 + sub * 128 + 8 * pp) * g.lda + k0
                    : A + (k0 + 4 * pp) * g.lda + (m0 + sub * 128));
      const unsigned char* pb = reinterpret_cast<const unsigned char*>(
          !B_KMAJOR ? B + (long long)(n0 + sub * 128 + 8 * pp) * g.ldb + k0
                    : B + (k0 + 4 * pp) * g.ldb + (n0 + sub * 128));
      glds16s(pa, voffA, dA + p * 1024);
      glds16s(pb, voffB, dB + p * 1024);
    }
  };
  auto dma = [&](int t, int stage) {
    if (interior) dma_tile_fast(t, stage);
    else dma_tile(t, stage);
  };

  f32x16 acc[NI][2];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  if (nt > 0) dma(0, 0);
  if (nt > 1) dma(1, 1);
  for (int t = 0; t < nt; ++t) {
    if (t == 0 && nt > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t >= 1 && t + 1 < nt) dma(t + 1, (t + 1) & 1);
    // this wave's 128-row sub-tile of A / 128-column sub-tile of B and its offsets inside them
    const unsigned char* tA = lds + (t & 1) * STAGE_B + (SUBS == 2 ? wm * SUB_B : 0);
    const unsigned char* tB = lds + (t & 1) * STAGE_B + TILE_B + (SUBS == 2 ? (wn >> 1) * SUB_B : 0);
    const int ra = SUBS == 2 ? 0 : wm * 64;
    const int cb = SUBS == 2 ? (wn & 1) * 64 : wn * 64;
    const int kvalid = (int)min((long long)BK, k_end - (k_begin + (long long)t * BK));
#pragma unroll
    for (int kk = 0; kk < BK; kk += 16) {
      bf16x8 fa[NI], fb[2];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        if (!A_KMAJOR) fa[i] = frag_kc(tA, ra + i * 32 + l31, kk, h);
        else fa[i] = frag_km(tA, kk + 8 * h, ra + i * 32, lane);
      }
      if (kvalid < BK) {                             // reduction tail: zero A beyond K (B's clamped reads are finite)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (kk + 8 * h + e >= kvalid) fa[i][e] = 0;
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (!B_KMAJOR) fb[j] = frag_kc(tB, cb + j * 32 + l31, kk, h);
        else fb[j] = frag_km(tB, kk + 8 * h, cb + j * 32, lane);
      }
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
  }
  __syncthreads();                                   // every wave is done with the operand stages: reuse them

  gemm_epilogue<NI>(g, acc, lds, wave, lane, m0 + wm * (64 * SUBS), n0 + wn * 64, ks, c_off, b1);
}

// ---- staggered form of the 256 x 256 tile for problems made of interior tiles only (M, N multiples of 256, every K range a
// multiple of 64): two wave groups (waves 0-3 / 4-7: one wave of each per SIMD) run ONE BARRIER APART, so that one group's MFMA
// segment coincides with the other group's fragment reads + DMA issue.  LDS = ring of four half-K slots (32 k: 16 KB of A + 16 KB
// of B) filled three ahead with counted waits.  Per half-tile x and wave:
//     R(x):  vmcnt -> own pieces of x+1 landed | 12 fragment reads of x | 4 DMA pieces of x+3 | s_barrier
//     M(x):  lgkmcnt(0) | 16 MFMAs | s_barrier      (s_setprio around the MFMAs: measured 0-5 % slower; finer phases of 8 MFMAs: no gain)
// Group 1 executes one extra barrier up front, group 0 one at the end.  Hazards by barrier count (group 0: R(x) ends at barrier
// 2x, M(x) at 2x+1; group 1 one later): the slot of x-1 is last read in group 1's R(x-1), which ends at 2x-1, and is refilled
// (x+3) in R(x) segments, all after 2x-1; every wave's wait for x+1 sits in its R(x), before barrier 2x+1, and x+1 is first
// read after 2x+1.  Same fragments, same k order as the lockstep kernel: bit-identical results.
template <bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(512) void gemm_bf16_stag_kernel(GemmB g) {
  constexpr int SUBS = 2;
  constexpr int BM = 128 * SUBS, BN = 128 * SUBS;
  constexpr int TILE_B = SUBS * SUB_B, STAGE_B = 2 * TILE_B;
  constexpr int NI = 2 * SUBS;                 // 32-row blocks per wave
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];      // 2 stages x (A tile + B tile)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = SUBS == 2 ? wave >> 2 : wave >> 1, wn = SUBS == 2 ? wave & 3 : wave & 1;
  const int h = lane >> 5, l31 = lane & 31;
  int b, ks, tx, ty;
  if (g.tm >= 8) {
    // XCD-aware decode (block id % 8 = XCD): row panel = (q / tn) * 8 + xcd, column tile = q % tn
    const int per_batch = ((g.tm + 7) / 8) * 8 * g.tn * g.ksplit;
    const int bid = blockIdx.x % per_batch;
    b = blockIdx.x / per_batch;
    const int xcd = bid & 7, q = bid >> 3;
    ks = q % g.ksplit;
    const int q2 = q / g.ksplit;
    tx = q2 % g.tn;
    ty = (q2 / g.tn) * 8 + xcd;
    if (ty >= g.tm) return;
  } else {                                     // few row panels (batched small problems): plain decode, no padding
    const int per_batch = g.tm * g.tn * g.ksplit;
    const int bid = blockIdx.x % per_batch;
    b = blockIdx.x / per_batch;
    ks = bid % g.ksplit;
    const int q2 = bid / g.ksplit;
    tx = q2 % g.tn;
    ty = q2 / g.tn;
  }
  const int m0 = ty * BM, n0 = tx * BN;
  const long long k_begin = (long long)ks * g.k_per_split;
  const long long k_end = min((long long)g.K, k_begin + g.k_per_split);
  const int nt = (int)((k_end - k_begin + BK - 1) / BK);
  const int b1 = b / g.batch2, b2 = b % g.batch2;
  const bf16_t* A = g.A + (long long)b1 * g.sAb + (long long)b2 * g.sAb2;
  const bf16_t* B = g.B + (long long)b1 * g.sBb + (long long)b2 * g.sBb2;
  const long long c_off = (long long)b1 * g.sCb + (long long)b2 * g.sCb2;
  const unsigned lds_base = (unsigned)(size_t)MCL_LDSP(lds);


  constexpr int HK = 32, HSUB = 8192, HOP = 2 * HSUB, SLOT = 2 * HOP;   // half-K, bytes per half sub-tile / operand / slot
  const int nh = (int)((k_end - k_begin) / HK);
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const int grp = wave_s >> 2;
  // DMA: a half sub-tile is 8 pieces of 1 KB; wave w moves piece (w & 7) of sub-tile 0 and of sub-tile 1 of both operands
  const int pp = wave_s;
  unsigned vA[2], vB[2];
  {
    const unsigned a0 = !A_KMAJOR ? (unsigned)((lane >> 2) * g.lda * 2) + (unsigned)((((lane & 3) ^ ((lane >> 4) & 3))) << 4)
                                  : (unsigned)((lane >> 4) * g.lda * 2) + (unsigned)(((lane & 15) ^ (((lane >> 4) & 3) << 2)) << 4);
    const unsigned b0 = !B_KMAJOR ? (unsigned)((lane >> 2) * g.ldb * 2) + (unsigned)((((lane & 3) ^ ((lane >> 4) & 3))) << 4)
                                  : (unsigned)((lane >> 4) * g.ldb * 2) + (unsigned)(((lane & 15) ^ (((lane >> 4) & 3) << 2)) << 4);
    vA[0] = a0; vA[1] = a0 + (unsigned)(!A_KMAJOR ? 256ll * g.lda : 256);
    vB[0] = b0; vB[1] = b0 + (unsigned)(!B_KMAJOR ? 256ll * g.ldb : 256);
  }
  const unsigned char* baseA = reinterpret_cast<const unsigned char*>(
      !A_KMAJOR ? A + (long long)(m0 + 16 * pp) * g.lda + k_begin : A + (k_begin + 4 * pp) * g.lda + m0);
  const unsigned char* baseB = reinterpret_cast<const unsigned char*>(
      !B_KMAJOR ? B + (long long)(n0 + 16 * pp) * g.ldb + k_begin : B + (k_begin + 4 * pp) * g.ldb + n0);
  const long long stepA = !A_KMAJOR ? 2ll * HK : 2ll * HK * g.lda, stepB = !B_KMAJOR ? 2ll * HK : 2ll * HK * g.ldb;
  auto dma_half = [&](int ht) {                       // 4 DMA instructions per wave
    const unsigned char* pa = baseA + ht * stepA;
    const unsigned char* pb = baseB + ht * stepB;
    const unsigned d = lds_base + (ht & 3) * SLOT + pp * 1024;
    glds16s(pa, vA[0], d);
    glds16s(pb, vB[0], d + HOP);
    glds16s(pa, vA[1], d + HSUB);
    glds16s(pb, vB[1], d + HOP + HSUB);
  };

  f32x16 acc[NI][2];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  for (int i = 0; i < 3 && i < nh; ++i) dma_half(i);
  if (nh >= 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (nh == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();         // the stagger: group 1 runs one barrier behind group 0
  const int cb = (wn & 1) * 64;
  for (int ht = 0; ht < nh; ++ht) {
    // ---- R segment
    if (ht + 1 < nh) {                                // own pieces of ht + 1 landed (younger ones may stay in flight)
      if (ht + 2 < nh) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned char* tA = lds + (ht & 3) * SLOT + wm * HSUB;
    const unsigned char* tB = lds + (ht & 3) * SLOT + HOP + (wn >> 1) * HSUB;
    bf16x8 fa[2][NI], fb[2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int kk = 16 * q;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (!B_KMAJOR) fb[q][j] = frag_kc32(tB, cb + j * 32 + l31, kk, h);
        else fb[q][j] = frag_km(tB, kk + 8 * h, cb + j * 32, lane);
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        if (!A_KMAJOR) fa[q][i] = frag_kc32(tA, i * 32 + l31, kk, h);
        else fa[q][i] = frag_km(tA, kk + 8 * h, i * 32, lane);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // into the slot of ht - 1.  Its last readers are the OTHER wave group's fragment reads of R(ht - 1), ISSUED before the barrier
    // this group has just passed but only waited for (lgkmcnt(0)) in that group's M segment, after the barrier: formally those
    // reads may still be outstanding here.  What keeps them ahead of the refill is latency, not a counted wait -- an LDS read
    // returns in ~100 cycles, the LDS-DMA write lands after a global-memory round trip (> 1000) -- and tools/stress_gemm_stag.py
    // (random problems, bit-compared with the lockstep kernel: 400 / 0 mismatches, kept in the profile scripts) screens it.
    // Moving the lgkmcnt(0) in front of the R-segment barrier closes the window formally and was measured 2-3 % slower.
    if (ht + 3 < nh) dma_half(ht + 3);
    __builtin_amdgcn_sched_barrier(0);                // (the 4 pieces spread among the MFMAs below instead: measured 2-5 % slower)
    __builtin_amdgcn_s_barrier();
    // ---- M segment
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[q][i], fb[q][j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();         // group 0 catches up: every wave has executed 2 nh + 2 barriers
  __syncthreads();                                   // every wave is done with the operand stages: reuse them

  gemm_epilogue<NI>(g, acc, lds, wave, lane, m0 + wm * (64 * SUBS), n0 + wn * 64, ks, c_off, b1);
}


// ================================================================================================================================
// Round 6: gemm_bf16_pipe_kernel -- PERSISTENT workgroups, ONE software pipeline across all their tiles, epilogue without LDS.
//
// What the round-4 ablation of the staggered kernel said about a K = 768 tile (33 us): the two barrier-separated segments per
// half-tile are bound by the longer one (fragment reads + DMA issue ~ 900 cycles beside 512 cycles of MFMA), the prologue
// (three half-tiles of DMA latency with nothing to compute) and the epilogue (LDS transposition + every CU storing its 128 KB
// at the same moment while no MFMA runs) are paid per tile, and at K = 768 a tile is only 24 half-tiles long.  This kernel:
//   * one workgroup per CU walks its tiles (virtual block ids w, w + G, ...: the XCD-aware order of the other kernels, the XCD
//     of a workgroup never changes) and the four-slot half-K ring RUNS ON ACROSS TILES: the DMA cursor is three half-tiles ahead
//     of the MFMAs whatever tile they belong to, so only the first tile of a workgroup has a prologue;
//   * inside a half-tile every wave interleaves its own 16 MFMAs with the 12 fragment reads of the NEXT 16 (two register sets)
//     and its 4 DMA pieces -- one barrier per half-tile instead of two, no wave-group stagger: the MFMA pipe is the pole;
//   * the MFMA operands are swapped (D^T = B^T A^T): a lane then holds ONE output row and 4 x 4 consecutive columns, so the
//     result leaves through v_permlane32_swap + 16-byte stores straight from the registers -- no LDS (the ring keeps filling for
//     the next tile during the epilogue), no barrier, and the stores drain behind the next tile's loop (counted vmcnt).
// Hazards (un-staggered, by barrier count; g = position of a half-tile in the workgroup's stream, slot g & 3):
//   RAW  slot g+1 is read (fragments F(g+1, 0)) after barrier B(g); every wave waited for its own pieces of g+1 before B(g).
//   WAR  slot g-1 is refilled (pieces of g+3) in iteration g; its last reads, F(g-1, 1), were issued before B(g-1) and retired
//        by the lgkmcnt(0) every wave executes before B(g-1).
// Interior tiles only (M, N multiples of 256; every K range a multiple of 64 and >= 128); everything else keeps the other kernels.
struct PipeTile {
  const unsigned char* baseA;   // this wave's piece of half-tile 0 (bytes)
  const unsigned char* baseB;
  int nh;                       // half-tiles (32 k)
  int m0, n0, ks, b1;
  long long c_off;
};

__device__ __forceinline__ unsigned swap_lo_hi(unsigned& x, unsigned& y) {       // x.hi <-> y.lo  (v_permlane32_swap_b32)
  const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
  x = r[0];
  y = r[1];
  return 0;
}


// Epilogue forms of the pipe kernel (compile-time: one form's code per kernel -- with run-time branches the eight unrolled blocks
// grew to 13 000 instructions, more than the instruction cache holds)
enum { PE_PLAIN = 0, PE_GELU2 = 1, PE_GELU1 = 2, PE_GBWD2 = 3, PE_GBWD1 = 4, PE_RESID = 5, PE_F32 = 6 };

// Straight from the registers.  Block (i, j) in the transposed MFMA layout: this lane = row m = i*32 + l31, register r = column
// (r & 3) + 8 (r >> 2) + 4 h of the 32-column block j: four groups of 4 consecutive columns.
// bf16 results leave through a wave-private 4 KB LDS area `stage` (beyond the ring): a 32-row block of the wave's 64 columns is
// written as 8-byte pieces (this lane's 4 x 4 columns of its row), read back as 16-byte chunks with 8 lanes per row, and stored as
// WHOLE 128-byte lines -- the register-direct form (v_permlane32_swap + two 16-byte stores per block: 32-byte row segments) wrote
// 28 % more bytes to HBM than the tensor holds (profiles/r06_gemm_pipe_experiments.txt 7).  LDS operations of one wave execute in
// order: no wait between the writes and the reads, no barrier (nobody else touches the area).
template <int EPI, int WN>
__device__ __forceinline__ void pipe_epilogue(const GemmB& g, f32x16 (&acc)[4][2], const PipeTile& cur, int wm, int wn, int h,
                                              int l31, unsigned char* __restrict__ stage) {
  const int mw = cur.m0 + wm * 128, nw = cur.n0 + wn * 64;
  constexpr bool has_x = EPI == PE_GBWD2 || EPI == PE_GBWD1 || EPI == PE_RESID;
  // one 64-bit row base per lane and tensor; a block adds a wave-uniform offset
  const long long row = mw + l31;
  const bf16_t* xrow = (EPI == PE_GBWD2 || EPI == PE_GBWD1) ? g.aux + row * g.ldaux + nw + 8 * h
                       : (EPI == PE_RESID ? g.resid + (long long)cur.b1 * g.sRb + row * g.ldr + nw + 8 * h : nullptr);
  const long long ldx = (EPI == PE_GBWD2 || EPI == PE_GBWD1) ? g.ldaux : g.ldr;
  bf16_t* crow = reinterpret_cast<bf16_t*>(g.C) + cur.c_off + row * g.ldc + nw + 8 * h;
  float* frow = reinterpret_cast<float*>(g.C) + (long long)cur.ks * g.slab_stride + cur.c_off + row * g.ldc + nw + 4 * h;
  const bool second = (EPI == PE_GELU2) || (EPI == PE_GELU1 && g.pre_out);
  bf16_t* prow = second ? g.pre_out + row * g.ldp + nw + 8 * h : nullptr;
  const int lane = l31 + 32 * h;
  // staging geometry: row l31 of the 32-row block, 16-byte chunk c (8 columns) stored at chunk c ^ (row & 7)
  auto stage_put = [&](int j, const unsigned (&P)[4][2]) {
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int col = j * 32 + gq * 8 + 4 * h;                  // first of this piece's 4 columns
      const int chunk = (col >> 3) ^ (l31 & 7);
      *reinterpret_cast<uint2*>(stage + l31 * 128 + (chunk << 4) + (col & 7) * 2) = make_uint2(P[gq][0], P[gq][1]);
    }
  };
  auto stage_flush = [&](bf16_t* rowbase, long long ld) {      // rowbase: this wave's (row 0, column 0) of the 32-row block
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int r = (lane >> 3) + 8 * t, c = lane & 7;
      const u32x4 v = *reinterpret_cast<const u32x4*>(stage + r * 128 + ((c ^ (r & 7)) << 4));
      if (!(g.dbg & 1) || v[0] == 0x12345678u) *reinterpret_cast<u32x4*>(rowbase + (long long)r * ld + c * 8) = v;
    }
  };
  bf16_t* cblk = reinterpret_cast<bf16_t*>(g.C) + cur.c_off + (long long)mw * g.ldc + nw;
  bf16_t* pblk = second ? g.pre_out + (long long)mw * g.ldp + nw : nullptr;
  (void)crow; (void)prow;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // the residual / gelu' rows of the block row's two 32-column halves (16 registers)
    u32x4 xq[2][2];
    if (has_x) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr)
          xq[j][pr] = *reinterpret_cast<const u32x4*>(xrow + (long long)(i * 32) * ldx + j * 32 + 16 * pr);
    }
    unsigned Pp[2][4][2];                            // packed second result of the two halves (the first goes to the staging area at once)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      __builtin_amdgcn_sched_barrier(0);             // one block at a time: keeps the epilogue's live set small
      float4 bq[4];
#pragma unroll
      for (int gq = 0; gq < 4; ++gq)
        bq[gq] = g.bias ? *reinterpret_cast<const float4*>(g.bias + nw + j * 32 + 8 * gq + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
      float vv[16];
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        vv[4 * gq + 0] = fmaf(acc[i][j][4 * gq + 0], g.alpha, bq[gq].x);
        vv[4 * gq + 1] = fmaf(acc[i][j][4 * gq + 1], g.alpha, bq[gq].y);
        vv[4 * gq + 2] = fmaf(acc[i][j][4 * gq + 2], g.alpha, bq[gq].z);
        vv[4 * gq + 3] = fmaf(acc[i][j][4 * gq + 3], g.alpha, bq[gq].w);
      }
      if (EPI == PE_F32) {
        float* o = frow + (long long)(i * 32) * g.ldc + j * 32;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
          *reinterpret_cast<float4*>(o + 8 * gq) = make_float4(vv[4 * gq], vv[4 * gq + 1], vv[4 * gq + 2], vv[4 * gq + 3]);
        continue;
      }
      if (has_x) {                                 // the x operand into this lane's own layout, applied at once
        unsigned xo[4][2];
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          unsigned u0 = xq[j][pr][0], u1 = xq[j][pr][1], w0 = xq[j][pr][2], w1 = xq[j][pr][3];
          swap_lo_hi(u0, w0);
          swap_lo_hi(u1, w1);
          xo[2 * pr][0] = u0; xo[2 * pr][1] = u1; xo[2 * pr + 1][0] = w0; xo[2 * pr + 1][1] = w1;
        }
        if (EPI == PE_GBWD2) {                     // aux holds the stored derivative
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            vv[4 * gq + 0] *= bf_lo(xo[gq][0]); vv[4 * gq + 1] *= bf_hi(xo[gq][0]);
            vv[4 * gq + 2] *= bf_lo(xo[gq][1]); vv[4 * gq + 3] *= bf_hi(xo[gq][1]);
          }
        } else if (EPI == PE_GBWD1) {
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            vv[4 * gq + 0] *= gelu_g(bf_lo(xo[gq][0])); vv[4 * gq + 1] *= gelu_g(bf_hi(xo[gq][0]));
            vv[4 * gq + 2] *= gelu_g(bf_lo(xo[gq][1])); vv[4 * gq + 3] *= gelu_g(bf_hi(xo[gq][1]));
          }
        } else {
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            vv[4 * gq + 0] += bf_lo(xo[gq][0]); vv[4 * gq + 1] += bf_hi(xo[gq][0]);
            vv[4 * gq + 2] += bf_lo(xo[gq][1]); vv[4 * gq + 3] += bf_hi(xo[gq][1]);
          }
        }
      }
      if (EPI == PE_GELU2) {                        // second output: gelu'(pre-activation); one exp / rcp serves both
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          float gr[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float Phi, ex;
            gelu_parts(vv[4 * gq + e], Phi, ex);
            gr[e] = fmaf(vv[4 * gq + e] * 0.39894228040143267794f, ex, Phi);
            vv[4 * gq + e] *= Phi;
          }
          Pp[j][gq][0] = pack_bf16(gr[0], gr[1]);
          Pp[j][gq][1] = pack_bf16(gr[2], gr[3]);
          __builtin_amdgcn_sched_barrier(0);          // four elements at a time: the 16-element form spilled
        }
      } else if (EPI == PE_GELU1) {                 // second output (optional): the pre-activation
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          Pp[j][gq][0] = pack_bf16(vv[4 * gq], vv[4 * gq + 1]);
          Pp[j][gq][1] = pack_bf16(vv[4 * gq + 2], vv[4 * gq + 3]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) vv[r] = gelu_f(vv[r]);
      }
      unsigned Pc[4][2];
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        Pc[gq][0] = pack_bf16(vv[4 * gq], vv[4 * gq + 1]);
        Pc[gq][1] = pack_bf16(vv[4 * gq + 2], vv[4 * gq + 3]);
      }
      stage_put(j, Pc);
    }
    if (EPI == PE_F32) continue;
    __builtin_amdgcn_sched_barrier(0);
    stage_flush(cblk + (long long)(i * 32) * g.ldc, g.ldc);
    if (second) {
      stage_put(0, Pp[0]);
      stage_put(1, Pp[1]);
      stage_flush(pblk + (long long)(i * 32) * g.ldp, g.ldp);
    }
  }
}

// WN = wave columns of the workgroup: 4 -> 8 waves, 256 x 256 tile, four 32 KB slots, one workgroup per CU (what is built);
//                                      2 -> 4 waves, 256 x 128 tile, three 24 KB slots: two independent workgroups per CU.  The second
// form was built to let one workgroup's epilogue run beside the other's MFMA loop and measured SLOWER on every ViT shape -- with
// the workgroups started half a tile apart, at one workgroup per CU, and with the tile's stores trickled out one per half-tile
// of the next tile from parked registers (profiles/r06_gemm_pipe_experiments.txt); the code path stays generic, only WN = 4 is
// instantiated.
template <bool A_KMAJOR, bool B_KMAJOR, int EPI, int WN, int DBG = 0>
__global__ __launch_bounds__(128 * WN, 2) void gemm_bf16_pipe_kernel(GemmB g, int total_virtual) {
  constexpr int NW = 2 * WN;                      // waves
  constexpr int BM = 256, BN = 64 * WN;
  constexpr int HK = 32, HSUB = 8192;             // half-K; bytes per half sub-tile (128 rows or columns x 32 k)
  constexpr int BSUBS = WN / 2;                   // 128-column sub-tiles of B
  constexpr int HOP = 2 * HSUB;                   // offset of B inside a slot (A: two half sub-tiles)
  constexpr int SLOT = HOP + BSUBS * HSUB;        // 32 KB / 24 KB
  constexpr int NSLOT = WN == 4 ? 4 : 3;
  constexpr int D = NSLOT - 1;                    // half-tiles the DMA cursor runs ahead
  constexpr int PPS = 8 / NW;                     // pieces of a half sub-tile per wave (1 / 2)
  constexpr int PA = 2 * PPS, PB = BSUBS * PPS;   // DMA instructions per wave and half-tile: A, B
  constexpr int P = PA + PB;                      // 4 / 6
  static_assert(P % 2 == 0, "pieces split evenly between the two k-steps");
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int h = lane >> 5, l31 = lane & 31;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const unsigned lds_base = (unsigned)(size_t)MCL_LDSP(lds);
  const int G = gridDim.x;

  // ---- tile decode (wave-uniform).  v = virtual block id in the XCD-aware order of the non-persistent kernels.
  auto decode = [&](int v, PipeTile& t) -> bool {
    int b, ks, tx, ty;
    if (g.tm >= 8) {
      const int per_batch = ((g.tm + 7) / 8) * 8 * g.tn * g.ksplit;
      const int bid = v % per_batch;
      b = v / per_batch;
      const int xcd = bid & 7, q = bid >> 3;
      ks = q % g.ksplit;
      const int q2 = q / g.ksplit;
      tx = q2 % g.tn;
      ty = (q2 / g.tn) * 8 + xcd;
      if (ty >= g.tm) return false;
    } else {
      const int per_batch = g.tm * g.tn * g.ksplit;
      const int bid = v % per_batch;
      b = v / per_batch;
      ks = bid % g.ksplit;
      const int q2 = bid / g.ksplit;
      tx = q2 % g.tn;
      ty = q2 / g.tn;
    }
    const int m0 = ty * BM, n0 = tx * BN;
    const long long k_begin = (long long)ks * g.k_per_split;
    const long long k_end = min((long long)g.K, k_begin + g.k_per_split);
    const int b1 = b / g.batch2, b2 = b % g.batch2;
    const bf16_t* A = g.A + (long long)b1 * g.sAb + (long long)b2 * g.sAb2;
    const bf16_t* B = g.B + (long long)b1 * g.sBb + (long long)b2 * g.sBb2;
    t.baseA = reinterpret_cast<const unsigned char*>(
        !A_KMAJOR ? A + (long long)(m0 + 16 * wave_s) * g.lda + k_begin : A + (k_begin + 4 * wave_s) * g.lda + m0);
    t.baseB = reinterpret_cast<const unsigned char*>(
        !B_KMAJOR ? B + (long long)(n0 + 16 * wave_s) * g.ldb + k_begin : B + (k_begin + 4 * wave_s) * g.ldb + n0);
    t.nh = (int)((k_end - k_begin) / HK);
    t.m0 = m0; t.n0 = n0; t.ks = ks; t.b1 = b1;
    t.c_off = (long long)b1 * g.sCb + (long long)b2 * g.sCb2;
    return true;
  };
  int v = (int)blockIdx.x - G;
  auto next_tile = [&](PipeTile& t) -> bool {
    for (v += G; v < total_virtual; v += G)
      if (decode(v, t)) return true;
    return false;
  };

  // ---- DMA geometry: a half sub-tile (128 rows x 32 k, or 32 k x 128 columns) is 8 pieces of 1 KB; wave w moves pieces
  // w, w + NW, ... of every half sub-tile of both operands.  The per-lane source offsets (swizzle on the source side, as in
  // gemm_bf16_stag_kernel) are computed once; piece (sub s, index w + NW e) adds a constant to the wave's base.
  unsigned vA[2][PPS], vB[BSUBS][PPS];
  {
    const unsigned a0 = !A_KMAJOR ? (unsigned)((lane >> 2) * g.lda * 2) + (unsigned)((((lane & 3) ^ ((lane >> 4) & 3))) << 4)
                                  : (unsigned)((lane >> 4) * g.lda * 2) + (unsigned)(((lane & 15) ^ (((lane >> 4) & 3) << 2)) << 4);
    const unsigned b0 = !B_KMAJOR ? (unsigned)((lane >> 2) * g.ldb * 2) + (unsigned)((((lane & 3) ^ ((lane >> 4) & 3))) << 4)
                                  : (unsigned)((lane >> 4) * g.ldb * 2) + (unsigned)(((lane & 15) ^ (((lane >> 4) & 3) << 2)) << 4);
#pragma unroll
    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
      for (int e = 0; e < PPS; ++e)        // sub-tile: +128 rows (k-contiguous) / +128 columns; piece + NW e: +16 NW e rows / +4 NW e k-rows
        vA[sb][e] = a0 + (unsigned)(!A_KMAJOR ? (256ll * sb + 32ll * NW * e) * g.lda : 256ll * sb + 8ll * NW * e * g.lda);
#pragma unroll
    for (int sb = 0; sb < BSUBS; ++sb)
#pragma unroll
      for (int e = 0; e < PPS; ++e)
        vB[sb][e] = b0 + (unsigned)(!B_KMAJOR ? (256ll * sb + 32ll * NW * e) * g.ldb : 256ll * sb + 8ll * NW * e * g.ldb);
  }
  const long long stepA = !A_KMAJOR ? 2ll * HK : 2ll * HK * g.lda, stepB = !B_KMAJOR ? 2ll * HK : 2ll * HK * g.ldb;

  PipeTile cur, nxt;
  if (!next_tile(cur)) return;
  bool nxt_valid = false, pf_in_nxt = false, pf_active = true;
  int pf_x = 0;                  // half-tile of the prefetch cursor inside its tile
  unsigned spf = 0, sc = 0;      // ring slots of the prefetch cursor and of the compute cursor
  const unsigned char* pfA = cur.baseA;
  const unsigned char* pfB = cur.baseB;
  // part 0 / 1 = the two halves of the wave's P instructions: (A sub-tile `part`: PPS pieces) + half of B's
  auto dma_part = [&](int part) {
    if (!pf_active) return;
    const unsigned d = lds_base + spf * SLOT + wave_s * 1024;
    if (!(DBG & 8)) {
#pragma unroll
      for (int e = 0; e < PPS; ++e) glds16s(pfA, vA[part][e], d + part * HSUB + e * NW * 1024);
      if (WN == 4) {
        glds16s(pfB, vB[part % BSUBS][0], d + HOP + (part % BSUBS) * HSUB);
      } else {
        glds16s(pfB, vB[0][part % PPS], d + HOP + (part % PPS) * NW * 1024);
      }
    }
    if (part == 1) {
      spf = spf + 1 == NSLOT ? 0 : spf + 1;
      ++pf_x;
      pfA += stepA;
      pfB += stepB;
      if (pf_x == (pf_in_nxt ? nxt.nh : cur.nh)) {       // (nh >= 4 > the cursor's lead: it is never two tiles ahead)
        nxt_valid = next_tile(nxt);
        pf_in_nxt = true;
        pf_x = 0;
        pf_active = nxt_valid;
        pfA = nxt.baseA;
        pfB = nxt.baseB;
      }
    }
  };

  f32x16 acc[4][2];
  bf16x8 fa0[4], fb0[2], fa1[4], fb1[2];       // fragments of k-step 0 / k-step 1 of a half-tile
  const int tA_off = wm * HSUB, tB_off = HOP + (wn >> 1) * HSUB, cb = (wn & 1) * 64;
  auto read_a = [&](unsigned slot, int q, int i) -> bf16x8 {
    const unsigned char* tA = lds + slot * SLOT + tA_off;
    if (!A_KMAJOR) return frag_kc32(tA, i * 32 + l31, 16 * q, h);
    return frag_km(tA, 16 * q + 8 * h, i * 32, lane);
  };
  auto read_b = [&](unsigned slot, int q, int j) -> bf16x8 {
    const unsigned char* tB = lds + slot * SLOT + tB_off;
    if (!B_KMAJOR) return frag_kc32(tB, cb + j * 32 + l31, 16 * q, h);
    return frag_km(tB, 16 * q + 8 * h, cb + j * 32, lane);
  };
#define MCL_SB() __builtin_amdgcn_sched_barrier(0)
#define MCL_MFMA(I, J, FA, FB) \
  do { if (!(DBG & 4)) acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FB[J], FA[I], acc[I][J], 0, 0, 0); \
       else acc[I][J][0] += __builtin_bit_cast(float, FA[I][0] ^ FB[J][0]); } while (0)
  // counted waits on the vector-memory counter (immediates): n = instructions that may stay in flight
  auto wait_vm = [&](int n) {
    switch (n) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;
      case 22: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break;
      case 35: asm volatile("s_waitcnt vmcnt(35)" ::: "memory"); break;
      case 38: asm volatile("s_waitcnt vmcnt(38)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
  };

  // ---- prologue of the workgroup's stream: D half-tiles in flight, the first one landed, its k-step 0 in registers
  for (int i = 0; i < D; ++i) { dma_part(0); dma_part(1); }
  wait_vm((D - 1) * P);
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int j = 0; j < 2; ++j) fb0[j] = read_b(0, 0, j);
#pragma unroll
  for (int i = 0; i < 4; ++i) fa0[i] = read_a(0, 0, i);

  bool first_tile = true;
  // stores per wave and tile of this epilogue form: 16 (one bf16 result) or 32 (two results / fp32)
  const int S = (EPI == PE_GELU2 || EPI == PE_F32 || (EPI == PE_GELU1 && g.pre_out != nullptr)) ? 32 : 16;
  // Steady state at barrier B(g): in issue order [g+1: P] ... [g+D-1: P] [g+D: P/2] -- wait for g+1.  For D-1 half-tiles after
  // an epilogue its S stores sit in that window too (they were issued after the pieces of the last position + D - 1).
  constexpr int VM_STEADY = (D - 2) * P + P / 2;
  for (;;) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const int nh = cur.nh;
    for (int x = 0; x < nh; ++x) {
      const unsigned s0 = sc, s1 = sc + 1 == NSLOT ? 0 : sc + 1;
      // ---- k-step 0: MFMAs on (fa0, fb0) | reads of k-step 1 of this half-tile | first half of the DMA pieces
      MCL_SB();
      MCL_MFMA(0, 0, fa0, fb0); fb1[0] = read_b(s0, 1, 0); MCL_SB();
      MCL_MFMA(0, 1, fa0, fb0); fb1[1] = read_b(s0, 1, 1); MCL_SB();
      MCL_MFMA(1, 0, fa0, fb0); fa1[0] = read_a(s0, 1, 0); MCL_SB();
      MCL_MFMA(1, 1, fa0, fb0); fa1[1] = read_a(s0, 1, 1); MCL_SB();
      MCL_MFMA(2, 0, fa0, fb0); fa1[2] = read_a(s0, 1, 2); MCL_SB();
      MCL_MFMA(2, 1, fa0, fb0); fa1[3] = read_a(s0, 1, 3); MCL_SB();
      MCL_MFMA(3, 0, fa0, fb0); MCL_SB();
      dma_part(0);
      MCL_SB();
      MCL_MFMA(3, 1, fa0, fb0); MCL_SB();
      // own pieces of g+1 landed (counted: the younger pieces -- and, right after an epilogue, its stores, which sit between
      // them in issue order -- stay in flight); own fragment reads of slot g retired; then the barrier
      if (!pf_active) wait_vm(0);
      else if (!first_tile && x < D - 1) wait_vm(VM_STEADY + S);
      else wait_vm(VM_STEADY);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (!(DBG & 16)) __builtin_amdgcn_s_barrier();
      // ---- k-step 1: MFMAs on (fa1, fb1) | reads of k-step 0 of the NEXT half-tile (the next tile's first one at a tile end)
      MCL_SB();
      MCL_MFMA(0, 0, fa1, fb1); fb0[0] = read_b(s1, 0, 0); MCL_SB();
      MCL_MFMA(0, 1, fa1, fb1); fb0[1] = read_b(s1, 0, 1); MCL_SB();
      MCL_MFMA(1, 0, fa1, fb1); fa0[0] = read_a(s1, 0, 0); MCL_SB();
      MCL_MFMA(1, 1, fa1, fb1); fa0[1] = read_a(s1, 0, 1); MCL_SB();
      MCL_MFMA(2, 0, fa1, fb1); fa0[2] = read_a(s1, 0, 2); MCL_SB();
      MCL_MFMA(2, 1, fa1, fb1); fa0[3] = read_a(s1, 0, 3); MCL_SB();
      MCL_MFMA(3, 0, fa1, fb1); MCL_SB();
      dma_part(1);
      MCL_SB();
      MCL_MFMA(3, 1, fa1, fb1); MCL_SB();
      sc = s1;
    }

    if (!(g.dbg & 2)) pipe_epilogue<EPI, WN>(g, acc, cur, wm, wn, h, l31, lds + NSLOT * SLOT + wave * 4096);
    if (!nxt_valid) break;
    cur = nxt;
    nxt_valid = false;
    pf_in_nxt = false;
    first_tile = false;
    // the next tile's first fragments again (slot `sc`, already read once in the last k-step above): re-reading them here
    // instead of keeping 24 registers live across the epilogue keeps the epilogue out of scratch
#pragma unroll
    for (int j = 0; j < 2; ++j) fb0[j] = read_b(sc, 0, j);
#pragma unroll
    for (int i = 0; i < 4; ++i) fa0[i] = read_a(sc, 0, i);
  }
#undef MCL_SB
#undef MCL_MFMA
}

}  // namespace

// flags: bit 0 A reduction-major, bit 1 B reduction-major, bit 2 GELU, bit 3 multiply by gelu'(aux), bit 4 fp32 output,
// bit 5 (with bit 2 and pre_out): pre_out receives gelu'(pre-activation) instead of the pre-activation, bit 6: multiply by aux itself
// (aux = that stored derivative): the backward's epilogue then needs no exp / rcp per element
extern "C" int64_t mcl_gemm_bf16_workspace_floats(int32_t M, int64_t ldc, int32_t ksplit) {
  if (M <= 0 || ldc <= 0 || ksplit <= 1) return 0;
  return (int64_t)ksplit * ((int64_t)M * ldc + 64);
}

extern "C" int mcl_gemm_bf16(const void* A, int64_t lda, int64_t sAb, const void* B, int64_t ldb, int64_t sBb, void* C,
                             int64_t ldc, int64_t sCb, int32_t M, int32_t N, int32_t K, int32_t batch, int32_t batch2,
                             int64_t sAb2, int64_t sBb2, int64_t sCb2, float alpha,
                             int32_t flags, const float* bias, const void* resid, int64_t ldr, int64_t sRb, const void* aux,
                             int64_t ldaux, void* pre_out, int64_t ldp, int32_t ksplit, float* workspace,
                             int32_t accumulate, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || batch <= 0 || batch2 <= 0 || (batch % batch2)) return MCL_EINVAL;
  const bool akm = flags & 1, bkm = flags & 2, gelu = flags & 4, gbwd = flags & (8 | 64), f32 = flags & 16;
  const bool gelu_grad_out = flags & 32, aux_is_grad = flags & 64;
  if (gelu_grad_out && (!gelu || !pre_out)) return MCL_EINVAL;
  if ((lda % 8) || (ldb % 8) || (reinterpret_cast<uintptr_t>(A) & 15u) || (reinterpret_cast<uintptr_t>(B) & 15u) ||
      (reinterpret_cast<uintptr_t>(C) & 15u) || (sAb % 8) || (sBb % 8) || (sAb2 % 8) || (sBb2 % 8) || (sCb % 4) || (sCb2 % 4))
    return MCL_EUNSUPPORTED;
  if (!f32 && ((ldc % 8) || ldc < (N + 7) / 8 * 8)) return MCL_EUNSUPPORTED;
  if (f32 && (ldc % 4)) return MCL_EUNSUPPORTED;
  if (gbwd && (!aux || (ldaux % 8))) return MCL_EINVAL;
  if (resid && (ldr % 8)) return MCL_EUNSUPPORTED;
  if (pre_out && (ldp % 8)) return MCL_EUNSUPPORTED;
  if (ksplit < 1) ksplit = 1;
  if (ksplit > 1 && (!f32 || !workspace || batch != 1 || bias || resid || gelu || gbwd || pre_out)) return MCL_EINVAL;
  GemmB g;
  g.A = (const bf16_t*)A; g.lda = lda; g.sAb = sAb; g.sAb2 = sAb2;
  g.B = (const bf16_t*)B; g.ldb = ldb; g.sBb = sBb; g.sBb2 = sBb2;
  g.C = C; g.ldc = ldc; g.sCb = sCb; g.sCb2 = sCb2;
  g.M = M; g.N = N; g.K = K; g.batch = batch; g.batch2 = batch2;
  g.alpha = alpha; g.bias = bias;
  g.resid = (const bf16_t*)resid; g.ldr = ldr; g.sRb = sRb;
  g.aux = (const bf16_t*)aux; g.ldaux = ldaux;
  g.pre_out = (bf16_t*)pre_out; g.ldp = ldp;
  g.dbg = 0;
  g.gelu = gelu ? (gelu_grad_out ? 2 : 1) : 0; g.gelu_bwd = gbwd ? (aux_is_grad ? 2 : 1) : 0; g.out_f32 = f32;
  // small problems (batched attention products) take the 128 x 128 tile, two workgroups per CU
  const int subs = (M <= 512 && N <= 512) ? 1 : 2;
  const int BMh = 128 * subs;
  g.tm = (M + BMh - 1) / BMh; g.tn = (N + BMh - 1) / BMh; g.ksplit = ksplit;
  long long kps = (K + ksplit - 1) / ksplit;
  kps = (kps + BK - 1) / BK * BK;
  g.k_per_split = kps;
  const bool via_slabs = ksplit > 1;       // requested: partial slabs + fixed-order merge, which is what honours `accumulate`
  g.ksplit = (int)((K + kps - 1) / kps);   // (a short reduction may collapse to ONE slab: still merged, so that += holds)
  g.slab_stride = 0;
  hipStream_t st = mcl_stream(stream);
  float* final_c = (float*)C;
  if (via_slabs) {
    g.slab_stride = (long long)M * ldc + 64;          // (+64: never a power-of-two stride, see csrc/wrw_fused.hip)
    g.C = workspace;
  }
  const int per_batch = (g.tm >= 8 ? ((g.tm + 7) / 8) * 8 : g.tm) * g.tn * g.ksplit;
  const dim3 grid((unsigned)(per_batch * batch));
  const size_t lds_bytes = (size_t)4 * subs * SUB_B;
  static mcl_device_once attr_once;
  if (auto attr_guard = attr_once.first()) {
#define MCL_ATTR(AK, BKM, SB) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel<AK, BKM, SB>), \
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, 4 * SB * SUB_B)
    MCL_ATTR(false, false, 1); MCL_ATTR(false, true, 1); MCL_ATTR(true, false, 1); MCL_ATTR(true, true, 1);
    MCL_ATTR(false, false, 2); MCL_ATTR(false, true, 2); MCL_ATTR(true, false, 2); MCL_ATTR(true, true, 2);
#undef MCL_ATTR
#define MCL_ATTR_S(AK, BKM) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_stag_kernel<AK, BKM>), \
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * SUB_B)
    MCL_ATTR_S(false, false); MCL_ATTR_S(false, true); MCL_ATTR_S(true, false); MCL_ATTR_S(true, true);
#undef MCL_ATTR_S
  }
  // the staggered kernel where every tile is interior (MCL_GEMM_STAG=0: the lockstep kernel; bit-identical results; read per
  // launch -- tests flip it inside one process)
  const char* e_stag = getenv("MCL_GEMM_STAG");
  const bool stag = subs == 2 && !(e_stag && e_stag[0] == '0') && M % 256 == 0 && N % 256 == 0 && K % 64 == 0 && kps % 64 == 0 &&
                    lda * 512 < (1ll << 31) && ldb * 512 < (1ll << 31);
  // the persistent pipelined kernel (round 6) wherever the staggered one applies and every K range holds >= 4 half-tiles.
  // MCL_GEMM_PIPE (read per launch -- tests and A/B runs flip it inside one process): "0" never, anything else always, unset =
  // where it measured faster than the staggered kernel (profiles/r06_gemm_pipe_experiments.txt 8): every bf16-output epilogue
  // (qkv / fc1 / fc2 -1 ... -3 %, the gelu'-multiplying data gradient -19 %); the fp32 split-K slabs of the weight gradients stay
  // on the staggered kernel (+9 % there).
  const char* e_pipe = getenv("MCL_GEMM_PIPE");
  const long long last_range = (long long)K - (long long)(g.ksplit - 1) * kps;
  const bool pipe_ok = stag && last_range >= 128 && kps >= 128 && !(gbwd && resid) && !(gelu && (gbwd || resid));
  const bool pipe = pipe_ok && (e_pipe ? e_pipe[0] != '0' : !f32);
  if (pipe) {
    const int total_virtual = per_batch * batch;
    const int cus = mcl_cu_count();
    const int slots = (cus / 8) * 8;
    const int G = total_virtual < slots ? total_virtual : slots;
    const int epi = f32 ? PE_F32 : gelu ? (gelu_grad_out ? PE_GELU2 : PE_GELU1) : gbwd ? (aux_is_grad ? PE_GBWD2 : PE_GBWD1)
                    : resid ? PE_RESID : PE_PLAIN;
    const char* e_dbg = getenv("MCL_GEMM_DBG");           // experiment switches: 1 no stores, 2 no epilogue, 4 / 8 / 16 loop ablations
    g.dbg = e_dbg ? atoi(e_dbg) : 0;
    const int lay = (akm ? 2 : 0) | (bkm ? 1 : 0);
    using KernelT = void (*)(GemmB, int);
#define MCL_PK(AK, BKM) {gemm_bf16_pipe_kernel<AK, BKM, PE_PLAIN, 4>, gemm_bf16_pipe_kernel<AK, BKM, PE_GELU2, 4>,   \
                         gemm_bf16_pipe_kernel<AK, BKM, PE_GELU1, 4>, gemm_bf16_pipe_kernel<AK, BKM, PE_GBWD2, 4>,   \
                         gemm_bf16_pipe_kernel<AK, BKM, PE_GBWD1, 4>, gemm_bf16_pipe_kernel<AK, BKM, PE_RESID, 4>,   \
                         gemm_bf16_pipe_kernel<AK, BKM, PE_F32, 4>}
    static const KernelT table[4][7] = {MCL_PK(false, false), MCL_PK(false, true), MCL_PK(true, false), MCL_PK(true, true)};
#undef MCL_PK
    static const KernelT dbg_table[4] = {gemm_bf16_pipe_kernel<false, false, PE_PLAIN, 4, 4>, gemm_bf16_pipe_kernel<false, false, PE_PLAIN, 4, 8>,
                                         gemm_bf16_pipe_kernel<false, false, PE_PLAIN, 4, 12>, gemm_bf16_pipe_kernel<false, false, PE_PLAIN, 4, 16>};
    static mcl_device_once pipe_once;
    if (auto guard = pipe_once.first()) {
      for (int a = 0; a < 4; ++a)
        for (int e = 0; e < 7; ++e)
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(table[a][e]), hipFuncAttributeMaxDynamicSharedMemorySize, 5 * 32768);
      for (int e = 0; e < 4; ++e)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dbg_table[e]), hipFuncAttributeMaxDynamicSharedMemorySize, 5 * 32768);
    }
    KernelT kern = table[lay][epi];
    if (g.dbg & 28) {                                     // loop ablations (NT, plain): 4 no MFMA, 8 no DMA, 16 no barrier
      if (lay != 0 || epi != PE_PLAIN) return MCL_EUNSUPPORTED;
      const int d = g.dbg & 28;
      kern = d == 4 ? dbg_table[0] : d == 8 ? dbg_table[1] : d == 12 ? dbg_table[2] : dbg_table[3];
    }
    // LDS: the four-slot ring (128 KB) + 4 KB of epilogue staging per wave = all 160 KB of the CU
    hipLaunchKernelGGL(kern, dim3(G), dim3(512), (size_t)(5 * 32768), st, g, total_virtual);
    if (via_slabs) {
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return (int)e;
      mcl_launch_wrw_merge_strided(workspace, g.ksplit, (long long)M * ldc, g.slab_stride, final_c, accumulate, st);
    }
    MCL_CHECK_LAUNCH();
    return MCL_OK;
  }
#define MCL_LAUNCH(AK, BKM)                                                                                          \
  do {                                                                                                               \
    if (stag) hipLaunchKernelGGL((gemm_bf16_stag_kernel<AK, BKM>), grid, dim3(512), lds_bytes, st, g);              \
    else if (subs == 2) hipLaunchKernelGGL((gemm_bf16_kernel<AK, BKM, 2>), grid, dim3(512), lds_bytes, st, g);           \
    else hipLaunchKernelGGL((gemm_bf16_kernel<AK, BKM, 1>), grid, dim3(256), lds_bytes, st, g);                     \
  } while (0)
  if (!akm && !bkm) MCL_LAUNCH(false, false);
  else if (!akm && bkm) MCL_LAUNCH(false, true);
  else if (akm && !bkm) MCL_LAUNCH(true, false);
  else MCL_LAUNCH(true, true);
#undef MCL_LAUNCH
  if (via_slabs) {
    // slabs are [M][ldc] fp32 at stride slab_stride; merge adds them in fixed order (+= when accumulate)
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    mcl_launch_wrw_merge_strided(workspace, g.ksplit, (long long)M * ldc, g.slab_stride, final_c, accumulate, st);
  }
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}
