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
#include <stdlib.h>

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

// ---- tile fills in two phases so that EVERY global load of a workgroup's prologue is in flight before the first LDS write waits
// for one: thread t owns the 16-byte chunks t, t + 512, t + 1024, t + 1536 of a [rows][64] source (rows <= 224: 1792 chunks).
constexpr int FILL_IT = 4;
struct Chunks { u32x4 v[FILL_IT]; };
__device__ __forceinline__ void load_chunks(Chunks& ch, const bf16_t* __restrict__ src, long long ld, int T, int tid) {
#pragma unroll
  for (int j = 0; j < FILL_IT; ++j) {
    const int i = tid + 512 * j, row = i >> 3, c = i & 7;
    ch.v[j] = u32x4{0, 0, 0, 0};
    if (row < T) ch.v[j] = *reinterpret_cast<const u32x4*>(src + (long long)row * ld + 8 * c);
  }
}
// row-major tile [nrows][64]; rows >= T zero
__device__ __forceinline__ void store_rows(unsigned char* tile, const Chunks& ch, int nrows, int tid) {
#pragma unroll
  for (int j = 0; j < FILL_IT; ++j) {
    const int i = tid + 512 * j, row = i >> 3, c = i & 7;
    if (row < nrows) *reinterpret_cast<u32x4*>(tile + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)) = ch.v[j];
  }
}
// transposed tile [64][TP]: element (d, row) = src[row][d]; columns T .. ncols - 1 zero.  Lanes l and l ^ 8 hold the same chunk of
// two adjacent rows (row = chunk index >> 3): they exchange their dwords so that each writes FOUR dwords (row pair packed) for half
// of the chunk's eight d -- instead of eight 2-byte stores per lane, which collide on banks and on half-dwords.
__device__ __forceinline__ void store_transposed(unsigned char* tile, const Chunks& ch, int ncols, int tid) {
  unsigned* t32 = reinterpret_cast<unsigned*>(tile);
#pragma unroll
  for (int j = 0; j < FILL_IT; ++j) {
    const int i = tid + 512 * j, row = i >> 3, c = i & 7, p = row & 1;
    unsigned mine[4], other[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      mine[e] = ch.v[j][e];
      other[e] = __shfl_xor(mine[e], 8);
    }
    if (row < ncols) {                                // (ncols is a multiple of 32: both rows of a pair exist together)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const unsigned ev = p ? other[2 * p + u] : mine[2 * p + u];      // the even row's dword (d = 8c + 4p + 2u, + 1)
        const unsigned od = p ? mine[2 * p + u] : other[2 * p + u];
        const int d = 8 * c + 4 * p + 2 * u, r2 = (row & ~1) >> 1;
        t32[(d * TP) / 2 + r2] = (ev & 0xFFFFu) | (od << 16);
        t32[((d + 1) * TP) / 2 + r2] = (ev >> 16) | (od & 0xFFFF0000u);
      }
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
// Persistent over (image, head) items: the next item's K / V chunks and query fragments are requested into registers right after
// the barrier that publishes the current item's tiles, and land under its MFMAs and softmax (one workgroup per CU: nothing else
// would cover the prologue's memory latency).
__global__ __launch_bounds__(512) void vit_attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o, float* __restrict__ lse,
                                                           int T, int heads, float scale, int items) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
  const int D = heads * DH, nb = (T + 31) / 32;
  const long long ld = 3ll * D;
  unsigned char* Ks = lds;                       // [nb*32][64]
  unsigned char* Vt = lds + nb * 32 * 128;       // [64][TP]
  const int strip = wave;
  const bool active = strip < nb;
  auto q_of = [&](int bh) { return qkv + (long long)(bh / heads) * T * ld + (bh % heads) * DH; };
  bf16x8 qf[4], qn[4];                           // this wave's query fragments of the current / next item
  Chunks ck, cv, nk, nv;
  int bh = blockIdx.x;
  if (bh >= items) return;
  {
    const bf16_t* q = q_of(bh);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = frag_global(q, ld, strip * 32 + l31, T, ks, h);
    load_chunks(ck, q + D, ld, T, tid);
    load_chunks(cv, q + 2 * D, ld, T, tid);
  }
  for (; bh < items; bh += gridDim.x) {
    const int b = bh / heads, head = bh % heads;
    store_rows(Ks, ck, nb * 32, tid);
    store_transposed(Vt, cv, nb * 32, tid);
    __syncthreads();
    const int next = bh + gridDim.x;
    if (next < items) {                          // in flight under this item's compute
      const bf16_t* q = q_of(next);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) qn[ks] = frag_global(q, ld, strip * 32 + l31, T, ks, h);
      load_chunks(nk, q + D, ld, T, tid);
      load_chunks(nv, q + 2 * D, ld, T, tid);
    }
    f32x16 oT[2];
    if (active) {
      // Two passes over the key blocks instead of holding the 224-key score row (112 registers) beside the prefetch registers:
      // pass 1 = scores -> row maximum, pass 2 = scores again -> exp, row sum and P V.  The 28 extra MFMAs are free here (the
      // kernel is bound by memory latency and the softmax's vector instructions, not by the matrix pipe).
      auto scores = [&](int kb) {
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Ks, kb * 32 + l31, ks, h), qf[ks], s, 0, 0, 0);
        return s;
      };
      // this lane = query strip*32 + l31, registers = keys; the other 16 keys of every block sit in lane ^ 32
      float m = -INFINITY;
      for (int kb = 0; kb < nb; ++kb) {
        const f32x16 s = scores(kb);
        const int kmax = T - kb * 32;            // keys of this block that exist: only the last block is partial
        if (kmax >= 32) {
#pragma unroll
          for (int r = 0; r < 16; ++r) m = fmaxf(m, s[r]);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) m = fmaxf(m, reg_row(r, h) < kmax ? s[r] : -INFINITY);
        }
      }
      const float sl2 = scale * 1.44269504088896340736f;                       // exp(x) = exp2(x log2 e): folded into the scale
      m = fmaxf(m, __shfl_xor(m, 32)) * scale;   // (scale > 0)
      const float ml2 = m * 1.44269504088896340736f;
      float sum = 0.0f;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oT[db][r] = 0.0f;
      for (int kb = 0; kb < nb; ++kb) {
        f32x16 s = scores(kb);
        const int kmax = T - kb * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], sl2, -ml2));   // unnormalised (<= 1): o is divided instead
        if (kmax < 32) {
#pragma unroll
          for (int r = 0; r < 16; ++r) s[r] = reg_row(r, h) < kmax ? s[r] : 0.0f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += s[r];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 pf = pack_regs(s, s2);
          const int base = kb * 32 + 16 * s2 + 4 * h;
#pragma unroll
          for (int db = 0; db < 2; ++db)
            oT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_transposed(Vt, db * 32 + l31, base), pf, oT[db], 0, 0, 0);
        }
      }
      sum += __shfl_xor(sum, 32);
      const float inv = 1.0f / sum;
      if (h == 0 && strip * 32 + l31 < T) lse[(long long)bh * T + strip * 32 + l31] = m + __logf(sum);
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oT[db][r] *= inv;   // (lane = query: one factor per lane)
    }
    __syncthreads();                             // every wave is done with the tiles: the staging rows reuse them
    if (active) store_strip(oT, lds + wave * 4096, o + (long long)b * T * D + head * DH, D, strip * 32, T, lane);
    __syncthreads();                             // staging rows read back: the next item's tiles may be written
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = qn[ks];
    ck = nk; cv = nv;
  }
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
  const int strip = wave;
  const bool active = strip < nb;
  const int qi = strip * 32 + l31;
  bf16x8 qf[4], dof[4], of[4];                   // this wave's strip operands: requested first
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    qf[ks] = frag_global(q, ld, qi, T, ks, h);
    dof[ks] = frag_global(dO, D, qi, T, ks, h);
    of[ks] = frag_global(O, D, qi, T, ks, h);
  }
  const float l = lse[(long long)bh * T + (qi < T ? qi : T - 1)];
  {
    Chunks ck, cv;
    load_chunks(ck, k, ld, T, tid);
    load_chunks(cv, v, ld, T, tid);
    store_rows(Ks, ck, nb * 32, tid);
    store_transposed(Kt, ck, nb * 32, tid);
    store_rows(Vs, cv, nb * 32, tid);
  }
  __syncthreads();
  f32x16 dqT[2];
  if (active) {
    float dsum = 0.0f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const u32x4 a = __builtin_bit_cast(u32x4, dof[ks]);
      const u32x4 c = __builtin_bit_cast(u32x4, of[ks]);
#pragma unroll
      for (int e = 0; e < 4; ++e) dsum += bf_lo(a[e]) * bf_lo(c[e]) + bf_hi(a[e]) * bf_hi(c[e]);
    }
    dsum += __shfl_xor(dsum, 32);
    if (h == 0 && qi < T) Dsum[(long long)bh * T + qi] = dsum;
    const float sl2 = scale * 1.44269504088896340736f, ll2 = l * 1.44269504088896340736f;   // exp(x) = exp2(x log2 e)
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
      const int kmax = T - kb * 32;                       // only the last key block is partial
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], sl2, -ll2)) * ((dp[r] - dsum) * scale);   // dS^T = P (dP - D) scale
      if (kmax < 32) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = reg_row(r, h) < kmax ? s[r] : 0.0f;
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
  const int strip = wave;
  const bool active = strip < nb;
  const int ki = strip * 32 + l31;
  bf16x8 kf[4], vf[4];                           // this wave's key / value fragments: requested first
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    kf[ks] = frag_global(k, ld, ki, T, ks, h);
    vf[ks] = frag_global(v, ld, ki, T, ks, h);
  }
  {
    Chunks cq, cd;
    load_chunks(cq, q, ld, T, tid);
    load_chunks(cd, dO, D, T, tid);
    const float lv = tid < T ? lse[(long long)bh * T + tid] : 0.0f, dv = tid < T ? Dsum[(long long)bh * T + tid] : 0.0f;
    store_rows(Qs, cq, nb * 32, tid);
    store_transposed(Qt, cq, nb * 32, tid);
    store_rows(dOs, cd, nb * 32, tid);
    store_transposed(dOt, cd, nb * 32, tid);
    if (tid < nb * 32) { lseS[tid] = lv * 1.44269504088896340736f; DS[tid] = dv; }   // lse in the exp2 domain
  }
  __syncthreads();
  f32x16 dkT[2], dvT[2];
  if (active) {
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dkT[db][r] = 0.0f; dvT[db][r] = 0.0f; }
    const float sl2 = scale * 1.44269504088896340736f;
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
      const int qmax = T - qb * 32;                        // only the last query block is partial
#pragma unroll
      for (int g = 0; g < 4; ++g) {                        // registers 4g .. 4g + 3 = queries qb*32 + 8g + 4h + 0..3
        const float4 lq = *reinterpret_cast<const float4*>(lseS + qb * 32 + 8 * g + 4 * h);
        const float4 dq4 = *reinterpret_cast<const float4*>(DS + qb * 32 + 8 * g + 4 * h);
        const float lv[4] = {lq.x, lq.y, lq.z, lq.w}, dv4[4] = {dq4.x, dq4.y, dq4.z, dq4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          const float p = __builtin_amdgcn_exp2f(fmaf(s[r], sl2, -lv[e]));
          s[r] = p;
          ds[r] = p * ((dp[r] - dv4[e]) * scale);
        }
      }
      if (qmax < 32) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool ok = reg_row(r, h) < qmax;
          s[r] = ok ? s[r] : 0.0f;
          ds[r] = ok ? ds[r] : 0.0f;
        }
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
  const char* e_p = getenv("MCL_VIT_ATTN_PERSIST");     // 0: one workgroup per item (A/B)
  const int items = B * heads;
  const int grid = (items < 2 * mcl_cu_count() || (e_p && e_p[0] == '0')) ? items : mcl_cu_count();   // persistent from two rounds up
  hipLaunchKernelGGL(vit_attn_fwd_kernel, dim3((unsigned)grid), dim3(512), lds_bytes, mcl_stream(stream),
                     (const bf16_t*)qkv, (bf16_t*)o, lse, T, heads, scale, items);
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
