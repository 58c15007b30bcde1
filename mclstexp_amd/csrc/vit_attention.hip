// Fused multi-head self-attention core of the ViT image encoder (/root/reference/model.py:104-116: timm vit_base_patch{16,32}_224,
// blocks[i].attn: softmax(q k^T d^-1/2) v per image and head), bf16 operands, fp32 accumulation, head dimension 64, T <= 224 tokens
// (197 for patch 16, 50 for patch 32) -- forward and backward WITHOUT the (B heads, T, T) probability tensor in HBM.
//
//   forward    o_h = softmax(q_h k_h^T * scale) v_h  and the row log-sum-exp (all the backward needs beside qkv, o, do)
//   backward   dq_h = dS k_h,  dk_h = dS^T q_h,  dv_h = P^T do_h,   dS = P .* (do_h v_h^T - D) * scale,  D_i = sum_d do[i][d] o[i][d],
//              P recomputed from (q, k, lse)
//
// The unfused path (vit_fused.py round 2-3: batched GEMMs + softmax launches) moved P and dP through HBM five times forward and six
// times backward per layer (~2.7 GB of the layer's traffic, 13.6 ms of the ViT-B/16 step in 9 launches per layer); here 1 + 2
// launches whose traffic is qkv, o, do and dqkv once or twice.
//
// One workgroup (8 waves) per (image, head): the whole problem (197 x 64 per operand) lives in LDS.  A wave owns a strip of 32
// queries (forward, dq) or 32 keys (dk / dv).  The score tile is computed TRANSPOSED with respect to the strip -- S^T = K Q^T for a
// query strip -- so that in the MFMA result layout a LANE is one query and the REGISTERS run over keys: the row softmax is a
// per-lane loop plus one exchange with lane ^ 32, and P^T (as it sits in the result registers, rounded to bf16) IS the B operand of
// the next product O^T = V^T P^T once the contraction index is permuted to the register order: MFMA step s2 of key block kb
// contracts, for lane half h, the keys kb*32 + 16 s2 + 4 h + {0..3, 8..11} -- so the A operand (V^T, rows = head dimension) is two
// 8-byte LDS reads from a tile stored [d][key].  No probability ever leaves the registers.  The dk / dv kernel is the mirrored
// problem (lane = key, registers = queries; per-query lse and D are broadcast LDS reads).
//
// LDS tiles: row-major operand tiles [rows][64] bf16 (128-byte rows, 16-byte chunk c of row r at c ^ ((r >> 1) & 7): conflict-free
// ds_read_b128 fragments, the layout of csrc/gemm_bf16.hip) and transposed tiles [64][260] bf16 (row pitch 130 dwords = 2 mod 64:
// the 32 lanes of a ds_read_b64 group, one row each, cover the 64 banks exactly once).
#include "common.h"
#include <math.h>

namespace {

typedef unsigned short bf16_t;
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int DH = 64;          // head dimension
constexpr int TP = 260;         // pitch (elements) of a transposed tile
constexpr int TT_B = DH * TP * 2;  // bytes of a transposed tile

__device__ __forceinline__ unsigned pack2(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xFFFF0000u); }

// ---- fill a row-major tile [nrows32][64] from src rows (row stride ld elements, column offset applied by the caller); rows >= T zero
__device__ __forceinline__ void fill_rows(unsigned char* tile, const bf16_t* __restrict__ src, long long ld, int T, int nrows, int tid,
                                          int nthreads) {
  for (int i = tid; i < nrows * 8; i += nthreads) {
    const int row = i >> 3, c = i & 7;
    u32x4 v = {0, 0, 0, 0};
    if (row < T) v = *reinterpret_cast<const u32x4*>(src + (long long)row * ld + 8 * c);
    *reinterpret_cast<u32x4*>(tile + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)) = v;
  }
}
// ---- fill a transposed tile [64][TP]: element (d, row) = src[row][d]; columns >= T zero (up to ncols)
__device__ __forceinline__ void fill_transposed(unsigned char* tile, const bf16_t* __restrict__ src, long long ld, int T, int ncols,
                                                int tid, int nthreads) {
  bf16_t* t = reinterpret_cast<bf16_t*>(tile);
  for (int i = tid; i < ncols * 8; i += nthreads) {
    const int row = i >> 3, c = i & 7;
    u32x4 v = {0, 0, 0, 0};
    if (row < T) v = *reinterpret_cast<const u32x4*>(src + (long long)row * ld + 8 * c);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      t[(8 * c + 2 * e) * TP + row] = (bf16_t)(v[e] & 0xFFFFu);
      t[(8 * c + 2 * e + 1) * TP + row] = (bf16_t)(v[e] >> 16);
    }
  }
}
// fragment of a row-major tile: rows rbase + (lane & 31), 8 consecutive head-dimension elements 16 ks + 8 (lane >> 5) ..
__device__ __forceinline__ bf16x8 frag_rows(const unsigned char* tile, int row, int ks, int h) {
  const int c = 2 * ks + h;
  return *reinterpret_cast<const bf16x8*>(tile + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
}
// the same fragment straight from global memory (rows of the strip itself: read once per strip); rows >= T clamp to T - 1
__device__ __forceinline__ bf16x8 frag_global(const bf16_t* __restrict__ src, long long ld, int row, int T, int ks, int h) {
  const int r = row < T ? row : T - 1;
  return *reinterpret_cast<const bf16x8*>(src + (long long)r * ld + 16 * ks + 8 * h);
}
// A fragment of a transposed tile for the permuted contraction: row d, columns base + {0..3, 8..11}
__device__ __forceinline__ bf16x8 frag_transposed(const unsigned char* tile, int d, int base) {
  const bf16x4 lo = *reinterpret_cast<const bf16x4*>(tile + (d * TP + base) * 2);
  const bf16x4 hi = *reinterpret_cast<const bf16x4*>(tile + (d * TP + base + 8) * 2);
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}
// registers 8 s2 .. 8 s2 + 7 of a result block, rounded to bf16: the B operand of the next product (slot order = register order)
__device__ __forceinline__ bf16x8 pack_regs(const f32x16& p, int s2) {
  u32x4 w;
#pragma unroll
  for (int e = 0; e < 4; ++e) w[e] = pack2(p[8 * s2 + 2 * e], p[8 * s2 + 2 * e + 1]);
  return __builtin_bit_cast(bf16x8, w);
}
// row offset inside a 32-row block of result register r for lane half h
__device__ __forceinline__ int reg_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// transposed result (rows = head dimension in registers, lane = token) -> out[token][col0 + d], through a wave-private LDS staging
// area of 32 x 64 bf16 (4 KB): whole 128-byte rows leave the CU
__device__ __forceinline__ void store_strip(const f32x16 (&acc)[2], unsigned char* stage, bf16_t* __restrict__ out, long long ld,
                                            int tok0, int T, int lane) {
  const int h = lane >> 5, l31 = lane & 31;
  bf16_t* st = reinterpret_cast<bf16_t*>(stage);
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {          // 4 consecutive d per register group
      const u32x2 w = {pack2(acc[db][4 * g], acc[db][4 * g + 1]), pack2(acc[db][4 * g + 2], acc[db][4 * g + 3])};
      *reinterpret_cast<u32x2*>(st + l31 * 64 + db * 32 + 8 * g + 4 * h) = w;
    }
  // (a wave reads back only what it wrote; LDS operations of a wave complete in order)
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = it * 8 + (lane >> 3), c = lane & 7;
    const u32x4 v = *reinterpret_cast<const u32x4*>(st + row * 64 + 8 * c);
    if (tok0 + row < T) *reinterpret_cast<u32x4*>(out + (long long)(tok0 + row) * ld + 8 * c) = v;
  }
}

// ================================================================ forward
__global__ __launch_bounds__(512) void vit_attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o, float* __restrict__ lse,
                                                           int T, int heads, float scale) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
  const int bh = blockIdx.x, b = bh / heads, head = bh % heads;
  const int D = heads * DH, nb = (T + 31) / 32;
  const long long ld = 3ll * D;
  const bf16_t* q = qkv + (long long)b * T * ld + head * DH;
  const bf16_t* k = q + D;
  const bf16_t* v = q + 2 * D;
  unsigned char* Ks = lds;                       // [nb*32][64]
  unsigned char* Vt = lds + nb * 32 * 128;       // [64][TP]
  fill_rows(Ks, k, ld, T, nb * 32, tid, 512);
  fill_transposed(Vt, v, ld, T, nb * 32, tid, 512);
  __syncthreads();
  f32x16 oT[2];
  const int strip = wave;
  const bool active = strip < nb;
  if (active) {
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = frag_global(q, ld, strip * 32 + l31, T, ks, h);
    f32x16 s[7];
#pragma unroll
    for (int kb = 0; kb < 7; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = 0.0f;
      if (kb < nb) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Ks, kb * 32 + l31, ks, h), qf[ks], s[kb], 0, 0, 0);
      }
    }
    // row softmax: this lane = query strip*32 + l31, registers = keys; the other 16 keys of every block sit in lane ^ 32
    float m = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 7; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool ok = kb < nb && kb * 32 + reg_row(r, h) < T;
        s[kb][r] = ok ? s[kb][r] * scale : -INFINITY;
        m = fmaxf(m, s[kb][r]);
      }
    m = fmaxf(m, __shfl_xor(m, 32));
    float sum = 0.0f;
#pragma unroll
    for (int kb = 0; kb < 7; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[kb][r] = __expf(s[kb][r] - m);
        sum += s[kb][r];
      }
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
    if (h == 0 && strip * 32 + l31 < T) lse[(long long)bh * T + strip * 32 + l31] = m + __logf(sum);
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) oT[db][r] = 0.0f;
#pragma unroll
    for (int kb = 0; kb < 7; ++kb) {
      if (kb < nb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] *= inv;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 pf = pack_regs(s[kb], s2);
          const int base = kb * 32 + 16 * s2 + 4 * h;
#pragma unroll
          for (int db = 0; db < 2; ++db)
            oT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_transposed(Vt, db * 32 + l31, base), pf, oT[db], 0, 0, 0);
        }
      }
    }
  }
  __syncthreads();                               // every wave is done with the tiles: the staging rows reuse them
  if (active) store_strip(oT, lds + wave * 4096, o + (long long)b * T * D + head * DH, D, strip * 32, T, lane);
}

// ================================================================ backward, query strips: dq (and D = rowsum(do .* o) for the dk/dv kernel)
__global__ __launch_bounds__(512) void vit_attn_bwd_dq_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                              const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                              float* __restrict__ Dsum, bf16_t* __restrict__ dqkv, int T, int heads,
                                                              float scale) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
  const int bh = blockIdx.x, b = bh / heads, head = bh % heads;
  const int D = heads * DH, nb = (T + 31) / 32;
  const long long ld = 3ll * D;
  const bf16_t* q = qkv + (long long)b * T * ld + head * DH;
  const bf16_t* k = q + D;
  const bf16_t* v = q + 2 * D;
  const bf16_t* dO = dout + (long long)b * T * D + head * DH;
  const bf16_t* O = o + (long long)b * T * D + head * DH;
  unsigned char* Ks = lds;
  unsigned char* Vs = Ks + nb * 32 * 128;
  unsigned char* Kt = Vs + nb * 32 * 128;
  fill_rows(Ks, k, ld, T, nb * 32, tid, 512);
  fill_rows(Vs, v, ld, T, nb * 32, tid, 512);
  fill_transposed(Kt, k, ld, T, nb * 32, tid, 512);
  __syncthreads();
  f32x16 dqT[2];
  const int strip = wave;
  const bool active = strip < nb;
  if (active) {
    const int qi = strip * 32 + l31;
    bf16x8 qf[4], dof[4];
    float dsum = 0.0f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[ks] = frag_global(q, ld, qi, T, ks, h);
      dof[ks] = frag_global(dO, D, qi, T, ks, h);
      const u32x4 a = __builtin_bit_cast(u32x4, dof[ks]);
      const u32x4 c = __builtin_bit_cast(u32x4, frag_global(O, D, qi, T, ks, h));
#pragma unroll
      for (int e = 0; e < 4; ++e) dsum += bf_lo(a[e]) * bf_lo(c[e]) + bf_hi(a[e]) * bf_hi(c[e]);
    }
    dsum += __shfl_xor(dsum, 32);
    const float l = lse[(long long)bh * T + (qi < T ? qi : T - 1)];
    if (h == 0 && qi < T) Dsum[(long long)bh * T + qi] = dsum;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) dqT[db][r] = 0.0f;
    for (int kb = 0; kb < nb; ++kb) {
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.0f; dp[r] = 0.0f; }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Ks, kb * 32 + l31, ks, h), qf[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Vs, kb * 32 + l31, ks, h), dof[ks], dp, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool ok = kb * 32 + reg_row(r, h) < T;
        const float p = ok ? __expf(s[r] * scale - l) : 0.0f;
        s[r] = p * (dp[r] - dsum) * scale;              // dS^T
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 df = pack_regs(s, s2);
        const int base = kb * 32 + 16 * s2 + 4 * h;
#pragma unroll
        for (int db = 0; db < 2; ++db)
          dqT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_transposed(Kt, db * 32 + l31, base), df, dqT[db], 0, 0, 0);
      }
    }
  }
  __syncthreads();
  if (active) store_strip(dqT, lds + wave * 4096, dqkv + (long long)b * T * ld + head * DH, ld, strip * 32, T, lane);
}

// ================================================================ backward, key strips: dk, dv
__global__ __launch_bounds__(512) void vit_attn_bwd_dkv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                               const float* __restrict__ lse, const float* __restrict__ Dsum,
                                                               bf16_t* __restrict__ dqkv, int T, int heads, float scale) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
  const int bh = blockIdx.x, b = bh / heads, head = bh % heads;
  const int D = heads * DH, nb = (T + 31) / 32;
  const long long ld = 3ll * D;
  const bf16_t* q = qkv + (long long)b * T * ld + head * DH;
  const bf16_t* k = q + D;
  const bf16_t* v = q + 2 * D;
  const bf16_t* dO = dout + (long long)b * T * D + head * DH;
  unsigned char* Qs = lds;
  unsigned char* dOs = Qs + nb * 32 * 128;
  unsigned char* Qt = dOs + nb * 32 * 128;
  unsigned char* dOt = Qt + TT_B;
  float* lseS = reinterpret_cast<float*>(dOt + TT_B);   // [nb*32] lse, then [nb*32] D
  float* DS = lseS + nb * 32;
  fill_rows(Qs, q, ld, T, nb * 32, tid, 512);
  fill_rows(dOs, dO, D, T, nb * 32, tid, 512);
  fill_transposed(Qt, q, ld, T, nb * 32, tid, 512);
  fill_transposed(dOt, dO, D, T, nb * 32, tid, 512);
  for (int i = tid; i < nb * 32; i += 512) {
    lseS[i] = i < T ? lse[(long long)bh * T + i] : 0.0f;
    DS[i] = i < T ? Dsum[(long long)bh * T + i] : 0.0f;
  }
  __syncthreads();
  f32x16 dkT[2], dvT[2];
  const int strip = wave;
  const bool active = strip < nb;
  if (active) {
    const int ki = strip * 32 + l31;
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kf[ks] = frag_global(k, ld, ki, T, ks, h);
      vf[ks] = frag_global(v, ld, ki, T, ks, h);
    }
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dkT[db][r] = 0.0f; dvT[db][r] = 0.0f; }
    for (int qb = 0; qb < nb; ++qb) {
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.0f; dp[r] = 0.0f; }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Qs, qb * 32 + l31, ks, h), kf[ks], s, 0, 0, 0);     // S[q][key]
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(dOs, qb * 32 + l31, ks, h), vf[ks], dp, 0, 0, 0);  // dP[q][key]
      }
      f32x16 ds;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qi = qb * 32 + reg_row(r, h);
        const float p = qi < T ? __expf(s[r] * scale - lseS[qi]) : 0.0f;
        s[r] = p;
        ds[r] = p * (dp[r] - DS[qi]) * scale;
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pf = pack_regs(s, s2), df = pack_regs(ds, s2);
        const int base = qb * 32 + 16 * s2 + 4 * h;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          dvT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_transposed(dOt, db * 32 + l31, base), pf, dvT[db], 0, 0, 0);
          dkT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_transposed(Qt, db * 32 + l31, base), df, dkT[db], 0, 0, 0);
        }
      }
    }
  }
  __syncthreads();
  if (active) {
    bf16_t* dbase = dqkv + (long long)b * T * ld + head * DH;
    store_strip(dkT, lds + wave * 4096, dbase + D, ld, strip * 32, T, lane);
    store_strip(dvT, lds + wave * 4096, dbase + 2 * D, ld, strip * 32, T, lane);
  }
}

inline size_t rows_bytes(int T) { return (size_t)((T + 31) / 32) * 32 * 128; }

}  // namespace

extern "C" int mcl_vit_attn_fwd(const void* qkv, void* o, float* lse, int32_t B, int32_t T, int32_t heads, float scale,
                                mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!qkv || !o || !lse || B <= 0 || T <= 0 || heads <= 0) return MCL_EINVAL;
  if (T > 224 || (reinterpret_cast<uintptr_t>(qkv) & 15u) || (reinterpret_cast<uintptr_t>(o) & 15u)) return MCL_EUNSUPPORTED;
  size_t lds_bytes = rows_bytes(T) + TT_B;
  if (lds_bytes < (size_t)8 * 4096) lds_bytes = (size_t)8 * 4096;
  static mcl_device_once attr_once;
  if (auto attr_guard = attr_once.first())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(vit_attn_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(vit_attn_fwd_kernel, dim3((unsigned)(B * heads)), dim3(512), lds_bytes, mcl_stream(stream),
                     (const bf16_t*)qkv, (bf16_t*)o, lse, T, heads, scale);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_vit_attn_bwd(const void* qkv, const void* o, const void* dout, const float* lse, float* dsum, void* dqkv,
                                int32_t B, int32_t T, int32_t heads, float scale, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!qkv || !o || !dout || !lse || !dsum || !dqkv || B <= 0 || T <= 0 || heads <= 0) return MCL_EINVAL;
  if (T > 224 || (reinterpret_cast<uintptr_t>(qkv) & 15u) || (reinterpret_cast<uintptr_t>(o) & 15u) ||
      (reinterpret_cast<uintptr_t>(dout) & 15u) || (reinterpret_cast<uintptr_t>(dqkv) & 15u))
    return MCL_EUNSUPPORTED;
  static mcl_device_once attr_once;
  if (auto attr_guard = attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(vit_attn_bwd_dq_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(vit_attn_bwd_dkv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  hipStream_t st = mcl_stream(stream);
  size_t lds_dq = 2 * rows_bytes(T) + TT_B;
  if (lds_dq < (size_t)8 * 4096) lds_dq = (size_t)8 * 4096;
  hipLaunchKernelGGL(vit_attn_bwd_dq_kernel, dim3((unsigned)(B * heads)), dim3(512), lds_dq, st, (const bf16_t*)qkv, (const bf16_t*)o,
                     (const bf16_t*)dout, lse, dsum, (bf16_t*)dqkv, T, heads, scale);
  const size_t lds_dkv = 2 * rows_bytes(T) + 2 * TT_B + (size_t)((T + 31) / 32) * 32 * 8;
  hipLaunchKernelGGL(vit_attn_bwd_dkv_kernel, dim3((unsigned)(B * heads)), dim3(512), lds_dkv, st, (const bf16_t*)qkv,
                     (const bf16_t*)dout, lse, dsum, (bf16_t*)dqkv, T, heads, scale);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}
