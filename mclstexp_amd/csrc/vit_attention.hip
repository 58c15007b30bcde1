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
// contracts, for lane half h, the keys kb*32 + 16 s2 + 4 h + {0..3, 8..11} -- and the A operand (V^T: rows = head dimension) for
// exactly those keys is two transposing LDS reads (ds_read_b64_tr_b16) of the row-major V tile.  No probability ever leaves the
// registers and no transposed copy of any operand exists.  The dk / dv kernel is the mirrored problem (lane = key, registers =
// queries; per-query lse and D are broadcast LDS reads).
//
// LDS tiles: row-major [rows][64] bf16 (128-byte rows) with ONE swizzle (swz below) that is conflict-free both for the row fragments
// (ds_read_b128) and for the transposing fragments: 56-58 KB per workgroup.
#include "common.h"
#include <math.h>
#include <stdlib.h>

namespace {

typedef unsigned short bf16_t;
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short v4s_t __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int DH = 64;          // head dimension

// 16-byte chunk c of tile row r lives at physical chunk c ^ swz(r): bits 1..3 of r, bit-REVERSED.  Row fragments (ds_read_b128: a
// lane group = rows {0-3, 12-15, 20-27} of one logical chunk) see eight distinct values per row parity -> 16 distinct 16-byte
// slots; transposing fragments (ds_read_b64_tr_b16: a 32-lane group = 4 consecutive rows x 64 bytes) see rows r, r + 1 on the two
// 128-byte halves of the bank row and rows r + 2, r + 3 moved by four chunks (bit 1 of r -> chunk bit 2) -> all 64 banks once.
__device__ __forceinline__ int swz(int row) { return ((row >> 1) & 1) << 2 | ((row >> 2) & 1) << 1 | ((row >> 3) & 1); }

__device__ __forceinline__ unsigned pack2(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xFFFF0000u); }

// ---- tile fills in two phases so that EVERY global load of a workgroup's prologue is in flight before the first LDS write waits
// for one: thread t of NT owns the 16-byte chunks t, t + NT, t + 2 NT, t + 3 NT of a [rows][64] source (rows <= NT / 2: a workgroup
// has one wave per 32-row strip -- 512 threads up to 224 tokens, 256 up to 128, 128 up to 64).
constexpr int FILL_IT = 4;
struct Chunks { u32x4 v[FILL_IT]; };
template <int NT>
__device__ __forceinline__ void load_chunks(Chunks& ch, const bf16_t* __restrict__ src, long long ld, int T, int tid) {
#pragma unroll
  for (int j = 0; j < FILL_IT; ++j) {
    const int i = tid + NT * j, row = i >> 3, c = i & 7;
    ch.v[j] = u32x4{0, 0, 0, 0};
    if (row < T) ch.v[j] = *reinterpret_cast<const u32x4*>(src + (long long)row * ld + 8 * c);
  }
}
// row-major tile [nrows][64]; rows >= T zero
template <int NT>
__device__ __forceinline__ void store_rows(unsigned char* tile, const Chunks& ch, int nrows, int tid) {
#pragma unroll
  for (int j = 0; j < FILL_IT; ++j) {
    const int i = tid + NT * j, row = i >> 3, c = i & 7;
    if (row < nrows) *reinterpret_cast<u32x4*>(tile + row * 128 + ((c ^ swz(row)) << 4)) = ch.v[j];
  }
}
// fragment of a row-major tile: rows rbase + (lane & 31), 8 consecutive head-dimension elements 16 ks + 8 (lane >> 5) ..
__device__ __forceinline__ bf16x8 frag_rows(const unsigned char* tile, int row, int ks, int h) {
  const int c = 2 * ks + h;
  return *reinterpret_cast<const bf16x8*>(tile + row * 128 + ((c ^ swz(row)) << 4));
}
// the same fragment straight from global memory (rows of the strip itself: read once per strip); rows >= T clamp to T - 1
__device__ __forceinline__ bf16x8 frag_global(const bf16_t* __restrict__ src, long long ld, int row, int T, int ks, int h) {
  const int r = row < T ? row : T - 1;
  return *reinterpret_cast<const bf16x8*>(src + (long long)r * ld + 16 * ks + 8 * h);
}
// TRANSPOSED fragment of a row-major tile [row][64] for the permuted contraction: for lane (column = col0 + (lane & 31), half h) the
// eight tile rows row0 + {0..3, 8..11} of that column (row0 = block base + 16 s2 + 4 h, a multiple of 4) -- two transposing reads
// (ds_read_b64_tr_b16: a 16-lane group reads a 4-row x 16-column block, lane i supplying the 8-byte piece (i & 3) of row i >> 2
// and receiving the four rows of column i).  No transposed copy of the tile exists.
__device__ __forceinline__ bf16x8 frag_tr(const unsigned char* tile, int col0, int row0, int lane) {
  const int i = lane & 15, q = i >> 2;
  const int chunk = (col0 + 16 * ((lane >> 4) & 1)) / 8 + ((i & 3) >> 1);
  const int r_lo = row0 + q, r_hi = row0 + 8 + q;
  const unsigned char* p_lo = tile + r_lo * 128 + ((chunk ^ swz(r_lo)) << 4) + (i & 1) * 8;
  const unsigned char* p_hi = tile + r_hi * 128 + ((chunk ^ swz(r_hi)) << 4) + (i & 1) * 8;
  const v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_t __attribute__((address_space(3)))*)p_lo);
  const v4s_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_t __attribute__((address_space(3)))*)p_hi);
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
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

// the same through a 2 KB wave-private staging area, one 32-d half at a time (64-byte row pieces leave the CU)
__device__ __forceinline__ void store_strip_halves(const f32x16 (&acc)[2], unsigned char* stage, bf16_t* __restrict__ out, long long ld,
                                                   int tok0, int T, int lane) {
  const int h = lane >> 5, l31 = lane & 31;
  bf16_t* st = reinterpret_cast<bf16_t*>(stage);
#pragma unroll
  for (int db = 0; db < 2; ++db) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const u32x2 w = {pack2(acc[db][4 * g], acc[db][4 * g + 1]), pack2(acc[db][4 * g + 2], acc[db][4 * g + 3])};
      *reinterpret_cast<u32x2*>(st + l31 * 32 + 8 * g + 4 * h) = w;
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int row = it * 16 + (lane >> 2), c = lane & 3;
      const u32x4 v = *reinterpret_cast<const u32x4*>(st + row * 32 + 8 * c);
      if (tok0 + row < T) *reinterpret_cast<u32x4*>(out + (long long)(tok0 + row) * ld + db * 32 + 8 * c) = v;
    }
  }
}

// ================================================================ forward
// Persistent over (image, head) items: the next item's K / V chunks and query fragments are requested into registers right after
// the barrier that publishes the current item's tiles, and land under its MFMAs and softmax (one workgroup per CU: nothing else
// would cover the prologue's memory latency).
// PREFETCH = false: no next-item registers -> under 128 VGPRs, four waves per SIMD, two workgroups per CU (56 KB of LDS each)
// cover each other's prologue instead.
template <bool PREFETCH, int NT>
__global__ __launch_bounds__(NT, PREFETCH ? 2 : 4) void vit_attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o,
                                                                             float* __restrict__ lse, int T, int heads, float scale,
                                                                             int items) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
  const int D = heads * DH, nb = (T + 31) / 32;
  const long long ld = 3ll * D;
  unsigned char* Ks = lds;                       // [nb*32][64]
  unsigned char* Vs = lds + nb * 32 * 128;       // [nb*32][64]
  const int strip = wave;
  const bool active = strip < nb;
  auto q_of = [&](int bh) { return qkv + (long long)(bh / heads) * T * ld + (bh % heads) * DH; };
  bf16x8 qf[4], qn[4];                           // this wave's query fragments of the current / next item
  Chunks ck, cv, nk, nv;
  int bh = blockIdx.x;
  if (bh >= items) return;
  auto load_item = [&](int item, bf16x8 (&qq)[4], Chunks& kk, Chunks& vv) {
    const bf16_t* q = q_of(item);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qq[ks] = frag_global(q, ld, strip * 32 + l31, T, ks, h);
    load_chunks<NT>(kk, q + D, ld, T, tid);
    load_chunks<NT>(vv, q + 2 * D, ld, T, tid);
  };
  if (PREFETCH) load_item(bh, qf, ck, cv);
  for (; bh < items; bh += gridDim.x) {
    const int b = bh / heads, head = bh % heads;
    if (!PREFETCH) load_item(bh, qf, ck, cv);    // (nothing carried across items: the registers are free during the compute)
    store_rows<NT>(Ks, ck, nb * 32, tid);
    store_rows<NT>(Vs, cv, nb * 32, tid);
    __syncthreads();
    const int next = bh + gridDim.x;
    if (PREFETCH && next < items) load_item(next, qn, nk, nv);   // in flight under this item's compute
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
            oT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(Vs, db * 32, base, lane), pf, oT[db], 0, 0, 0);
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
    if (PREFETCH) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) qf[ks] = qn[ks];
      ck = nk; cv = nv;
    }
  }
}

// ================================================================ backward, query strips: dq (and D = rowsum(do .* o) for the dk/dv kernel)
template <int NT>
__global__ __launch_bounds__(NT, 4) void vit_attn_bwd_dq_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
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
    load_chunks<NT>(ck, k, ld, T, tid);
    load_chunks<NT>(cv, v, ld, T, tid);
    store_rows<NT>(Ks, ck, nb * 32, tid);
    store_rows<NT>(Vs, cv, nb * 32, tid);
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
          dqT[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(Ks, db * 32, base, lane), df, dqT[db], 0, 0, 0);
      }
    }
  }
  __syncthreads();
  if (active) store_strip(dqT, lds + wave * 4096, dqkv + (long long)b * T * ld + head * DH, ld, strip * 32, T, lane);
}

// ================================================================ backward, key strips: dk, dv
template <int NT>
__global__ __launch_bounds__(NT, 4) void vit_attn_bwd_dkv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
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
  float* lseS = reinterpret_cast<float*>(dOs + nb * 32 * 128);   // [nb*32] lse, then [nb*32] D
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
    load_chunks<NT>(cq, q, ld, T, tid);
    load_chunks<NT>(cd, dO, D, T, tid);
    const float lv = tid < T ? lse[(long long)bh * T + tid] : 0.0f, dv = tid < T ? Dsum[(long long)bh * T + tid] : 0.0f;
    store_rows<NT>(Qs, cq, nb * 32, tid);
    store_rows<NT>(dOs, cd, nb * 32, tid);
    if (tid < nb * 32) { lseS[tid] = lv * 1.44269504088896340736f; DS[tid] = dv; }   // lse in the exp2 domain
  }
  __syncthreads();
  // Two passes over the query blocks -- dv (needs P only), then dk (needs P and dP) -- so that at most one 32-register result, one
  // or two score blocks and the strip's fragments are live: under 128 registers, four waves per SIMD, two workgroups per CU (the
  // single-pass form needed 194 registers and ran one).  The scores of a block are computed twice; the matrix pipe has the room.
  const float sl2 = scale * 1.44269504088896340736f;
  auto probs = [&](f32x16& s, int qb) {                    // s: scores S[q][key] -> P, queries beyond T zero
#pragma unroll
    for (int g = 0; g < 4; ++g) {                          // registers 4g .. 4g + 3 = queries qb*32 + 8g + 4h + 0..3
      const float4 lq = *reinterpret_cast<const float4*>(lseS + qb * 32 + 8 * g + 4 * h);
      s[4 * g] = __builtin_amdgcn_exp2f(fmaf(s[4 * g], sl2, -lq.x));
      s[4 * g + 1] = __builtin_amdgcn_exp2f(fmaf(s[4 * g + 1], sl2, -lq.y));
      s[4 * g + 2] = __builtin_amdgcn_exp2f(fmaf(s[4 * g + 2], sl2, -lq.z));
      s[4 * g + 3] = __builtin_amdgcn_exp2f(fmaf(s[4 * g + 3], sl2, -lq.w));
    }
    const int qmax = T - qb * 32;                          // only the last query block is partial
    if (qmax < 32) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = reg_row(r, h) < qmax ? s[r] : 0.0f;
    }
  };
  bf16_t* dbase = dqkv + (long long)b * T * ld + head * DH;
  unsigned char* stage = reinterpret_cast<unsigned char*>(DS + nb * 32) + wave * 2048;   // wave-private, behind the tiles
  if (active) {
    f32x16 acc[2];
    // ---- pass 1: dv^T[d][key] = sum_q dO^T[d][q] P[q][key]
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[db][r] = 0.0f;
    for (int qb = 0; qb < nb; ++qb) {
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Qs, qb * 32 + l31, ks, h), kf[ks], s, 0, 0, 0);     // S[q][key]
      probs(s, qb);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pf = pack_regs(s, s2);
        const int base = qb * 32 + 16 * s2 + 4 * h;
#pragma unroll
        for (int db = 0; db < 2; ++db)
          acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(dOs, db * 32, base, lane), pf, acc[db], 0, 0, 0);
      }
    }
    store_strip_halves(acc, stage, dbase + 2 * D, ld, strip * 32, T, lane);
    // ---- pass 2: dk^T[d][key] = sum_q Q^T[d][q] dS[q][key],  dS = P (dP - D) scale
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[db][r] = 0.0f;
    for (int qb = 0; qb < nb; ++qb) {
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.0f; dp[r] = 0.0f; }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Qs, qb * 32 + l31, ks, h), kf[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(dOs, qb * 32 + l31, ks, h), vf[ks], dp, 0, 0, 0);  // dP[q][key]
      }
      probs(s, qb);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 dq4 = *reinterpret_cast<const float4*>(DS + qb * 32 + 8 * g + 4 * h);
        s[4 * g] *= (dp[4 * g] - dq4.x) * scale;
        s[4 * g + 1] *= (dp[4 * g + 1] - dq4.y) * scale;
        s[4 * g + 2] *= (dp[4 * g + 2] - dq4.z) * scale;
        s[4 * g + 3] *= (dp[4 * g + 3] - dq4.w) * scale;
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 df = pack_regs(s, s2);
        const int base = qb * 32 + 16 * s2 + 4 * h;
#pragma unroll
        for (int db = 0; db < 2; ++db)
          acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(Qs, db * 32, base, lane), df, acc[db], 0, 0, 0);
      }
    }
    store_strip_halves(acc, stage, dbase + D, ld, strip * 32, T, lane);
  }
}

inline size_t rows_bytes(int T) { return (size_t)((T + 31) / 32) * 32 * 128; }

}  // namespace

extern "C" int mcl_vit_attn_fwd(const void* qkv, void* o, float* lse, int32_t B, int32_t T, int32_t heads, float scale,
                                mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!qkv || !o || !lse || B <= 0 || T <= 0 || heads <= 0) return MCL_EINVAL;
  if (T > 224 || (reinterpret_cast<uintptr_t>(qkv) & 15u) || (reinterpret_cast<uintptr_t>(o) & 15u)) return MCL_EUNSUPPORTED;
  size_t lds_bytes = 2 * rows_bytes(T);
  if (lds_bytes < (size_t)8 * 4096) lds_bytes = (size_t)8 * 4096;
  static mcl_device_once attr_once;
  if (auto attr_guard = attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(vit_attn_fwd_kernel<true, 512>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(vit_attn_fwd_kernel<false, 512>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  // Long sequences (T = 197): one workgroup per item without next-item registers -- four waves per SIMD, two workgroups per CU cover
  // each other's prologue (136-139 us per layer at B = 256 against 146-148 persistent).  Short ones (T = 50): the persistent walk
  // with the next item prefetched into registers (41 against 43.6 us).
  // One wave per 32-token strip: 128 / 256 / 512 threads for up to 64 / 128 / 224 tokens.
  const int items = B * heads, nb = (T + 31) / 32;
  const bool persist = T <= 128;
  const int wg_per_cu = nb <= 2 ? 4 : (nb <= 4 ? 2 : 1);           // persistent grid: eight waves per CU whatever the workgroup size
  const int pgrid = mcl_cu_count() * wg_per_cu;
  const int grid = (items < 2 * pgrid || !persist) ? items : pgrid;
  hipStream_t st = mcl_stream(stream);
#define MCL_FWD(PF, NTV)                                                                                              \
  hipLaunchKernelGGL((vit_attn_fwd_kernel<PF, NTV>), dim3((unsigned)grid), dim3(NTV), lds_bytes, st, (const bf16_t*)qkv, \
                     (bf16_t*)o, lse, T, heads, scale, items)
  if (grid < items) {
    if (nb <= 2) MCL_FWD(true, 128);
    else if (nb <= 4) MCL_FWD(true, 256);
    else MCL_FWD(true, 512);
  } else {
    if (nb <= 2) MCL_FWD(false, 128);
    else if (nb <= 4) MCL_FWD(false, 256);
    else MCL_FWD(false, 512);
  }
#undef MCL_FWD
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
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(vit_attn_bwd_dq_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(vit_attn_bwd_dkv_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  hipStream_t st = mcl_stream(stream);
  const int nb = (T + 31) / 32;
  size_t lds_dq = 2 * rows_bytes(T);
  if (lds_dq < (size_t)8 * 4096) lds_dq = (size_t)8 * 4096;
  const size_t lds_dkv = 2 * rows_bytes(T) + (size_t)nb * 32 * 8 + (size_t)8 * 2048;   // tiles, lse / D, staging rows (74 KB at T = 197: two workgroups per CU)
#define MCL_BWD(NTV)                                                                                                   \
  do {                                                                                                                 \
    hipLaunchKernelGGL((vit_attn_bwd_dq_kernel<NTV>), dim3((unsigned)(B * heads)), dim3(NTV), lds_dq, st, (const bf16_t*)qkv, \
                       (const bf16_t*)o, (const bf16_t*)dout, lse, dsum, (bf16_t*)dqkv, T, heads, scale);             \
    hipLaunchKernelGGL((vit_attn_bwd_dkv_kernel<NTV>), dim3((unsigned)(B * heads)), dim3(NTV), lds_dkv, st,           \
                       (const bf16_t*)qkv, (const bf16_t*)dout, lse, dsum, (bf16_t*)dqkv, T, heads, scale);           \
  } while (0)
  if (nb <= 2) MCL_BWD(128);
  else if (nb <= 4) MCL_BWD(256);
  else MCL_BWD(512);
#undef MCL_BWD
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}
