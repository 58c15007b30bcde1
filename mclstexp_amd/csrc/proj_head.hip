// ProjectionHead (reference model.py:151-168) as ONE forward launch and ONE row-local backward launch, exact fp32.
//
//   forward   p = x Wp^T + bp ; a = gelu(p) ; z = a Wf^T + bf + p ; e = LayerNorm(z)           (dropout p = 0, model.py:164)
//   backward  dz = LayerNorm'(de) ; dp = (dz Wf) * gelu'(p) + dz ; d gamma, d beta, d bf, d bp  (row-local part + column sums;
//             the three products over the batch rows -- dWf = dz^T a, dWp = dp^T x, dx = dp Wp -- are one mcl_gemm_group launch)
//
// MI355X mapping.  P = 256 projection columns are fixed; a workgroup (4 waves) owns a block of 16 batch rows and ALL 256 columns
// of it, because the second linear layer and the LayerNorm need whole rows.  Each wave owns 64 columns as four
// v_mfma_f32_16x16x4_f32 tiles (exact fp32 products, fp32 accumulation -- what the 1e-4 contract of the path needs).
// At M = 128 batch rows that is 8 workgroups: far too few to stream the first layer's 1 - 3.5 MB weight, and a wave's chain of
// dependent fp32 MFMAs over D = 1000 would alone take 7 us.  So the first product is cut into K slices over the grid
// (blockIdx.x): every workgroup writes its 16 x 256 partial with write-through (sc1) stores, drains them, and takes a ticket on
// its row block's counter; the LAST arriver of a row block adds the slices in slice order (deterministic whoever is last),
// and carries on alone with bias + GELU, the 256 x 256 second layer (K = 256: 3.4 us of MFMA), the skip and the LayerNorm.
// No workgroup ever waits for another one: no co-residency requirement, no spin, nothing that can time out.
// The counter is left at zero by the last arriver (the next launch on the same counter finds it clean).
//
// LDS: weight slices as [column][k] rows of 36 words (K-contiguous operands: 16-byte stores, conflict-free 16-byte operand reads --
// lanes 0-15 of an MFMA operand read hit banks 36 n mod 64 = every multiple of 4 once) or [k][column] rows of 260 words (the
// backward's dz Wf, contraction along Wf's rows); 16 x 256 activation tiles as rows of 260 words (4 row mod 64: same argument).
#include "common.h"

namespace {

constexpr int P = 256, RB = 16, BK = 32, NT = 256;
constexpr int LDW = 36;          // weight slice row: 32 k + 4 pad
constexpr int LDT = 260;         // 16 x 256 activation tile row
constexpr int MAX_KS = 16;

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// One BK-slice of a K-contiguous 256-row weight (row stride ld, k in [k0, k0 + 32) clipped to kend) -> 8 float4 per thread.
// Loads are unconditional, from addresses clamped into the row (a load inside a bounds branch is followed by the compiler's
// vmcnt(0): eight serial round trips per slice); the returned mask (4 bits per chunk) zeroes what lies past kend at the LDS store.
// VEC: 16-byte loads (host: aligned bases / strides and D % 4 == 0, so a chunk is inside [0, kend) or outside as a whole).
template <bool VEC>
__device__ __forceinline__ unsigned load_w_kc(float4 (&r)[8], const float* __restrict__ W, long long ld, int k0, int kend, int tid) {
  const int k = k0 + (tid & 7) * 4;
  unsigned ok = 0;
#pragma unroll
  for (int h = 0; h < 8; ++h) {
    const float* q = W + (long long)((tid >> 3) + 32 * h) * ld;
    if (VEC) {
      r[h] = *reinterpret_cast<const float4*>(q + min(k, kend - 4));
      ok |= (k < kend ? 0xFu : 0u) << (4 * h);
    } else {
      r[h].x = q[min(k + 0, kend - 1)];
      r[h].y = q[min(k + 1, kend - 1)];
      r[h].z = q[min(k + 2, kend - 1)];
      r[h].w = q[min(k + 3, kend - 1)];
      ok |= ((k + 0 < kend ? 1u : 0u) | (k + 1 < kend ? 2u : 0u) | (k + 2 < kend ? 4u : 0u) | (k + 3 < kend ? 8u : 0u)) << (4 * h);
    }
  }
  return ok;
}
__device__ __forceinline__ float4 mask4(float4 v, unsigned m) {
  return make_float4((m & 1u) ? v.x : 0.0f, (m & 2u) ? v.y : 0.0f, (m & 4u) ? v.z : 0.0f, (m & 8u) ? v.w : 0.0f);
}
__device__ __forceinline__ void store_w_kc(const float4 (&r)[8], unsigned ok, float* __restrict__ Ws, int tid) {
#pragma unroll
  for (int h = 0; h < 8; ++h)
    *reinterpret_cast<float4*>(Ws + ((tid >> 3) + 32 * h) * LDW + (tid & 7) * 4) = mask4(r[h], (ok >> (4 * h)) & 0xFu);
}

// 16 rows x 32 k of a K-contiguous activation: threads 0..127 carry one float4 each (bits 0-3 of the mask: valid elements)
template <bool VEC>
__device__ __forceinline__ float4 load_x(const float* __restrict__ X, long long ld, int m0, int M, int k0, int kend, int tid,
                                         unsigned& ok) {
  const int r = m0 + ((tid & 127) >> 3), k = k0 + (tid & 7) * 4;
  const float* q = X + (long long)min(r, M - 1) * ld;
  float4 v;
  if (VEC) {
    v = *reinterpret_cast<const float4*>(q + min(k, kend - 4));
    ok = (r < M && k < kend) ? 0xFu : 0u;
  } else {
    v.x = q[min(k + 0, kend - 1)];
    v.y = q[min(k + 1, kend - 1)];
    v.z = q[min(k + 2, kend - 1)];
    v.w = q[min(k + 3, kend - 1)];
    ok = r < M ? ((k + 0 < kend ? 1u : 0u) | (k + 1 < kend ? 2u : 0u) | (k + 2 < kend ? 4u : 0u) | (k + 3 < kend ? 8u : 0u)) : 0u;
  }
  return v;
}

// acc[t] += A(16 x 32) * W(32 x 64 of this wave): A rows of lda words starting at As (k offset applied by the caller),
// W slice in [column][k] form.  Lane l: row / column l & 15, k = 4 (l >> 4) + j within each 16-k chunk at MFMA j.
__device__ __forceinline__ void mma_slice_kc(f32x4 (&acc)[4], const float* __restrict__ As, int lda, const float* __restrict__ Ws,
                                             int wave, int lane) {
#pragma unroll
  for (int kc = 0; kc < 2; ++kc) {
    const int ko = kc * 16 + 4 * (lane >> 4);
    const float4 a = *reinterpret_cast<const float4*>(As + (lane & 15) * lda + ko);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float4 b = *reinterpret_cast<const float4*>(Ws + (wave * 64 + t * 16 + (lane & 15)) * LDW + ko);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc[t], 0, 0, 0);
    }
  }
}

// accumulator element (t, v) of lane l of wave w: row 4 (l >> 4) + v, column 64 w + 16 t + (l & 15)
__device__ __forceinline__ void acc_to_tile(const f32x4 (&acc)[4], float* __restrict__ T, int wave, int lane) {
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int v = 0; v < 4; ++v) T[(4 * (lane >> 4) + v) * LDT + wave * 64 + t * 16 + (lane & 15)] = acc[t][v];
}

__device__ __forceinline__ float sum16(float v) {        // over the 16 lanes that share a row
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  return v;
}

struct HeadF {
  const float* x; long long ldx; int M, D;
  const float* wp; long long ldwp; const float* bp;
  const float* wf; long long ldwf; const float* bf;
  const float* gamma; const float* beta; float eps;
  float *e, *p, *a, *z, *mean, *rstd;
  float* ws; unsigned* cnt;
  int ks, kchunk;
};

constexpr int FWD_LDS_FLOATS = P * LDW + RB * LDW + 2 * RB * LDT + 4;     // weight slice + x slice + two activation tiles + flag

// Everything here is a chain of dependent round trips to L2 / memory (~1 us each) with microseconds of arithmetic between them:
// the weight slices travel through a PD-deep register pipeline (all of a short K slice's loads are in flight at once), the
// second layer's first PD slices are requested before the partials are read back, and the partials of four K slices at a time.
constexpr int PD = 4;
typedef unsigned long long u64;

template <bool VEC>
__global__ __launch_bounds__(NT) void proj_head_fwd_kernel(const HeadF h) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Ws = lds;                         // [256][36]
  float* Xs = Ws + P * LDW;                // [16][36]
  float* T0 = Xs + RB * LDW;               // [16][260]: partial / p
  float* T1 = T0 + RB * LDT;               // [16][260]: gelu(p), then z
  int& s_last = *reinterpret_cast<int*>(T1 + RB * LDT);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sl = blockIdx.x, rb = blockIdx.y, m0 = rb * RB;
  const int kbeg = sl * h.kchunk, kend = min(h.D, kbeg + h.kchunk);

  // ---- phase 1: this K slice's partial of p = x Wp^T
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
  float4 rw[PD][8], rx[PD];
  unsigned ow[PD], ox[PD];
  const int nk = (kend - kbeg + BK - 1) / BK;
#pragma unroll
  for (int u = 0; u < PD; ++u)
    if (u < nk) {                          // (workgroup-uniform: a scalar branch)
      ow[u] = load_w_kc<VEC>(rw[u], h.wp, h.ldwp, kbeg + u * BK, kend, tid);
      rx[u] = load_x<VEC>(h.x, h.ldx, m0, h.M, kbeg + u * BK, kend, tid, ox[u]);
    }
  for (int kt0 = 0; kt0 < nk; kt0 += PD) {
#pragma unroll
    for (int u = 0; u < PD; ++u) {
      const int kt = kt0 + u;
      if (kt < nk) {
        store_w_kc(rw[u], ow[u], Ws, tid);
        if (tid < 128) *reinterpret_cast<float4*>(Xs + (tid >> 3) * LDW + (tid & 7) * 4) = mask4(rx[u], ox[u]);
        __syncthreads();
        if (kt + PD < nk) {
          ow[u] = load_w_kc<VEC>(rw[u], h.wp, h.ldwp, kbeg + (kt + PD) * BK, kend, tid);
          rx[u] = load_x<VEC>(h.x, h.ldx, m0, h.M, kbeg + (kt + PD) * BK, kend, tid, ox[u]);
        }
        mma_slice_kc(acc, Xs, LDW, Ws, wave, lane);
        __syncthreads();
      }
    }
  }

  acc_to_tile(acc, T0, wave, lane);
  __syncthreads();
  // the 16 x 256 partial out, write-through: 8-byte stores, 512 contiguous bytes per wave instruction
  u64* mine = reinterpret_cast<u64*>(h.ws + ((long long)rb * h.ks + sl) * (RB * P));
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int f = tid + NT * j, row = f >> 7, c2 = f & 127;
    const float2 v = *reinterpret_cast<const float2*>(T0 + row * LDT + c2 * 2);
    __hip_atomic_store(mine + row * (P / 2) + c2, ((u64)__float_as_uint(v.y) << 32) | (u64)__float_as_uint(v.x), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  }
  drain();
  __syncthreads();
  if (tid == 0) {
    const unsigned ticket = __hip_atomic_fetch_add(h.cnt + rb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (ticket == (unsigned)h.ks - 1u);
  }
  __syncthreads();
  if (!s_last) return;

  // ---- the row block's last arriver.  The second layer's first PD weight slices: requested now.
#pragma unroll
  for (int u = 0; u < PD; ++u) ow[u] = load_w_kc<VEC>(rw[u], h.wf, h.ldwf, u * BK, P, tid);
  // p = sum of the slices in slice order + bias
  {
    const u64* part = reinterpret_cast<const u64*>(h.ws + (long long)rb * h.ks * (RB * P));
    float2 sum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) sum[j] = make_float2(0.0f, 0.0f);
    for (int s0 = 0; s0 < h.ks; s0 += 4) {
      u64 v[4][8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {        // (past the last slice: the last slice again, not added)
        const u64* q = part + (long long)min(s0 + i, h.ks - 1) * (RB * P / 2);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int f = tid + NT * j;
          v[i][j] = __hip_atomic_load(q + (f >> 7) * (P / 2) + (f & 127), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (s0 + i < h.ks) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            sum[j].x += __uint_as_float((unsigned)v[i][j]);
            sum[j].y += __uint_as_float((unsigned)(v[i][j] >> 32));
          }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int f = tid + NT * j, row = f >> 7, c2 = f & 127;
      const float2 b = *reinterpret_cast<const float2*>(h.bp + c2 * 2);
      const float2 s = make_float2(sum[j].x + b.x, sum[j].y + b.y);
      const float2 g = make_float2(gelu_erf(s.x), gelu_erf(s.y));
      *reinterpret_cast<float2*>(T0 + row * LDT + c2 * 2) = s;
      *reinterpret_cast<float2*>(T1 + row * LDT + c2 * 2) = g;
      if (m0 + row < h.M) {
        *reinterpret_cast<float2*>(h.p + (long long)(m0 + row) * P + c2 * 2) = s;
        *reinterpret_cast<float2*>(h.a + (long long)(m0 + row) * P + c2 * 2) = g;
      }
    }
  }
  if (tid == 0) __hip_atomic_store(h.cnt + rb, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  // ---- phase 2: z = gelu(p) Wf^T + bf + p   (K = 256)
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
  for (int kt0 = 0; kt0 < P / BK; kt0 += PD) {
#pragma unroll
    for (int u = 0; u < PD; ++u) {
      const int kt = kt0 + u;
      store_w_kc(rw[u], ow[u], Ws, tid);
      __syncthreads();                     // (first pass: also publishes T0 / T1)
      if (kt + PD < P / BK) ow[u] = load_w_kc<VEC>(rw[u], h.wf, h.ldwf, (kt + PD) * BK, P, tid);
      mma_slice_kc(acc, T1 + kt * BK, LDT, Ws, wave, lane);
      __syncthreads();
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int col = wave * 64 + t * 16 + (lane & 15);
    const float b = h.bf[col];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int row = 4 * (lane >> 4) + v;
      T1[row * LDT + col] = acc[t][v] + b + T0[row * LDT + col];       // (T1 = gelu(p) is dead: the K loop ended with a barrier)
    }
  }
  __syncthreads();

  // ---- LayerNorm over the 256 columns of each row: 16 threads per row, float4 c4 = sub + 16 q
  const int row = tid >> 4, sub = tid & 15;
  float4 zv[4];
  float s = 0.0f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    zv[q] = *reinterpret_cast<const float4*>(T1 + row * LDT + (sub + 16 * q) * 4);
    s += (zv[q].x + zv[q].y) + (zv[q].z + zv[q].w);
  }
  const float mean = sum16(s) * (1.0f / P);
  float q2 = 0.0f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float dx = zv[q].x - mean, dy = zv[q].y - mean, dz = zv[q].z - mean, dw = zv[q].w - mean;
    q2 += (dx * dx + dy * dy) + (dz * dz + dw * dw);
  }
  const float rstd = rsqrtf(sum16(q2) * (1.0f / P) + h.eps);
  if (m0 + row < h.M) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = (sub + 16 * q) * 4;
      const float4 g = *reinterpret_cast<const float4*>(h.gamma + c);
      const float4 b = *reinterpret_cast<const float4*>(h.beta + c);
      float4 o;
      o.x = (zv[q].x - mean) * rstd * g.x + b.x;
      o.y = (zv[q].y - mean) * rstd * g.y + b.y;
      o.z = (zv[q].z - mean) * rstd * g.z + b.z;
      o.w = (zv[q].w - mean) * rstd * g.w + b.w;
      *reinterpret_cast<float4*>(h.e + (long long)(m0 + row) * P + c) = o;
      *reinterpret_cast<float4*>(h.z + (long long)(m0 + row) * P + c) = zv[q];
    }
    if (sub == 0) {
      h.mean[m0 + row] = mean;
      h.rstd[m0 + row] = rstd;
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward, row-local part
struct HeadB {
  const float* de; long long ldde; int M;
  const float *z, *mean, *rstd, *gamma, *p;
  const float* wf; long long ldwf;
  float *dz, *dp;
  float* out[4];            // d gamma, d beta, d bf, d bp  [256]
  int accumulate;           // bit i: out[i] += (the parameter's .grad) instead of =
  float* ws;                // [row blocks][4][256] column-sum partials
  unsigned* cnt;
  int nrb;
};

constexpr int BWD_LDS_FLOATS = BK * LDT + 3 * RB * LDT + 4;   // weight slice [32][260] + dz tile + p tile + scratch tile + flag

__global__ __launch_bounds__(NT) void proj_head_bwd_rows_kernel(const HeadB h) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Ws = lds;                         // [32 contraction rows][260]; before the product: two scratch tiles (de xhat, de)
  float* Dz = Ws + BK * LDT;               // [16][260]
  float* Pt = Dz + RB * LDT;               // [16][260]: p, then dp
  float* Sc = Pt + RB * LDT;               // [16][260]: de
  float* Gx = Ws;                          // [16][260]: de * xhat (dead before the first weight slice is stored)
  int& s_last = *reinterpret_cast<int*>(Sc + RB * LDT);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rb = blockIdx.x, m0 = rb * RB;

  // the first PD weight slices (32 rows of Wf each, 256 contiguous columns per row; 8 float4 per thread and slice): requested
  // before anything else -- the LayerNorm backward below runs under their latency
  float4 rw[PD][8];
  const bool vec = ((reinterpret_cast<uintptr_t>(h.wf) & 15u) == 0) && (h.ldwf % 4 == 0);
  auto load_w = [&](float4 (&r)[8], int n0) {
#pragma unroll
    for (int hh = 0; hh < 8; ++hh) {
      const float* q = h.wf + (long long)(n0 + (tid >> 6) + 4 * hh) * h.ldwf + (tid & 63) * 4;
      if (vec) r[hh] = *reinterpret_cast<const float4*>(q);
      else     r[hh] = make_float4(q[0], q[1], q[2], q[3]);
    }
  };
#pragma unroll
  for (int u = 0; u < PD; ++u) load_w(rw[u], u * BK);

  // ---- LayerNorm backward, row-local: 16 threads per row
  {
    const int row = tid >> 4, sub = tid & 15;
    const bool live = m0 + row < h.M;
    const float mu = live ? h.mean[m0 + row] : 0.0f, rs = live ? h.rstd[m0 + row] : 0.0f;
    float4 g4[4], xh[4], d4[4];
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = (sub + 16 * q) * 4;
      float4 d = make_float4(0.0f, 0.0f, 0.0f, 0.0f), zz = d, pp = d;
      if (live) {
        const float* dr = h.de + (long long)(m0 + row) * h.ldde + c;
        d = make_float4(dr[0], dr[1], dr[2], dr[3]);
        zz = *reinterpret_cast<const float4*>(h.z + (long long)(m0 + row) * P + c);
        pp = *reinterpret_cast<const float4*>(h.p + (long long)(m0 + row) * P + c);
      }
      const float4 ga = *reinterpret_cast<const float4*>(h.gamma + c);
      d4[q] = d;
      xh[q] = make_float4((zz.x - mu) * rs, (zz.y - mu) * rs, (zz.z - mu) * rs, (zz.w - mu) * rs);
      g4[q] = make_float4(d.x * ga.x, d.y * ga.y, d.z * ga.z, d.w * ga.w);
      s1 += (g4[q].x + g4[q].y) + (g4[q].z + g4[q].w);
      s2 += (g4[q].x * xh[q].x + g4[q].y * xh[q].y) + (g4[q].z * xh[q].z + g4[q].w * xh[q].w);
      *reinterpret_cast<float4*>(Pt + row * LDT + c) = pp;
      *reinterpret_cast<float4*>(Sc + row * LDT + c) = d;
      *reinterpret_cast<float4*>(Gx + row * LDT + c) = make_float4(d.x * xh[q].x, d.y * xh[q].y, d.z * xh[q].z, d.w * xh[q].w);
    }
    s1 = sum16(s1) * (1.0f / P);
    s2 = sum16(s2) * (1.0f / P);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = (sub + 16 * q) * 4;
      float4 v;
      v.x = rs * (g4[q].x - s1 - xh[q].x * s2);
      v.y = rs * (g4[q].y - s1 - xh[q].y * s2);
      v.z = rs * (g4[q].z - s1 - xh[q].z * s2);
      v.w = rs * (g4[q].w - s1 - xh[q].w * s2);
      *reinterpret_cast<float4*>(Dz + row * LDT + c) = v;          // (rows past M: rs = 0 -> zeros)
      if (live) *reinterpret_cast<float4*>(h.dz + (long long)(m0 + row) * P + c) = v;
    }
  }
  __syncthreads();
  // column sums over this block's 16 rows (thread = column): d gamma, d beta, d bf
  float cs[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    cs[0] += Gx[r * LDT + tid];
    cs[1] += Sc[r * LDT + tid];
    cs[2] += Dz[r * LDT + tid];
  }
  __syncthreads();                         // Gx (= Ws) is free for the weight slices now

  // ---- dp = (dz Wf) * gelu'(p) + dz : contraction over Wf's 256 rows, 32 per slice, slice in [row][column] form
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
  for (int kt0 = 0; kt0 < P / BK; kt0 += PD) {
#pragma unroll
    for (int u = 0; u < PD; ++u) {
      const int kt = kt0 + u;
#pragma unroll
      for (int hh = 0; hh < 8; ++hh) *reinterpret_cast<float4*>(Ws + ((tid >> 6) + 4 * hh) * LDT + (tid & 63) * 4) = rw[u][hh];
      __syncthreads();
      if (kt + PD < P / BK) load_w(rw[u], (kt + PD) * BK);
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) {
        const int ko = kc * 16 + 4 * (lane >> 4);
        const float4 a = *reinterpret_cast<const float4*>(Dz + (lane & 15) * LDT + kt * BK + ko);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float* b = Ws + ko * LDT + wave * 64 + t * 16 + (lane & 15);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[0], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[LDT], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[2 * LDT], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[3 * LDT], acc[t], 0, 0, 0);
        }
      }
      __syncthreads();
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int col = wave * 64 + t * 16 + (lane & 15);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int row = 4 * (lane >> 4) + v;
      Pt[row * LDT + col] = acc[t][v] * gelu_erf_grad(Pt[row * LDT + col]) + Dz[row * LDT + col];   // (each element: one owner)
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int f = tid + NT * j, row = f >> 6, c4 = f & 63;
    if (m0 + row < h.M)
      *reinterpret_cast<float4*>(h.dp + (long long)(m0 + row) * P + c4 * 4) = *reinterpret_cast<const float4*>(Pt + row * LDT + c4 * 4);
  }
#pragma unroll
  for (int r = 0; r < RB; ++r) cs[3] += (m0 + r < h.M) ? Pt[r * LDT + tid] : 0.0f;

  // ---- column sums: this block's partials out (write-through), ticket; the last arriver adds the blocks in block order
  float* mine = h.ws + (long long)rb * 4 * P;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    __hip_atomic_store(reinterpret_cast<unsigned*>(mine + i * P + tid), __float_as_uint(cs[i]), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  drain();
  __syncthreads();
  if (tid == 0) {
    const unsigned ticket = __hip_atomic_fetch_add(h.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (ticket == (unsigned)h.nrb - 1u);
  }
  __syncthreads();
  if (!s_last) return;
  float tot[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int b0 = 0; b0 < h.nrb; b0 += 8) {          // eight blocks' partials in flight at a time
    unsigned v[8][4];
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        v[b][i] = __hip_atomic_load(reinterpret_cast<const unsigned*>(h.ws + ((long long)min(b0 + b, h.nrb - 1) * 4 + i) * P + tid),
                                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int b = 0; b < 8; ++b)
      if (b0 + b < h.nrb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) tot[i] += __uint_as_float(v[b][i]);
      }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (h.out[i]) h.out[i][tid] = ((h.accumulate >> i) & 1) ? h.out[i][tid] + tot[i] : tot[i];
  if (tid == 0) __hip_atomic_store(h.cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

// K slices of the first product: enough workgroups for ~one per CU, slices of >= 64 k (a multiple of 32), at most 16.
extern "C" int32_t mcl_proj_head_ksplit(int32_t M, int32_t D) {
  if (M <= 0 || D <= 0) return 1;
  const int nrb = (M + RB - 1) / RB;
  int ks = 256 / nrb;
  if (ks > (D + 63) / 64) ks = (D + 63) / 64;
  if (ks > MAX_KS) ks = MAX_KS;
  return ks < 1 ? 1 : ks;
}

extern "C" int64_t mcl_proj_head_ws_floats(int32_t M, int32_t ksplit) {
  if (M <= 0 || ksplit <= 0) return -1;
  const long long nrb = (M + RB - 1) / RB;
  const long long fwd = nrb * ksplit * RB * P, bwd = nrb * 4 * P;
  return fwd > bwd ? fwd : bwd;
}

extern "C" int mcl_proj_head_fwd(const float* x, int64_t ldx, int32_t M, int32_t D, const float* wp, int64_t ldwp, const float* bp,
                                 const float* wf, int64_t ldwf, const float* bf, const float* gamma, const float* beta, float eps,
                                 float* e, float* p, float* a, float* z, float* mean, float* rstd, float* ws, uint32_t* counters,
                                 int32_t ksplit, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!x || !wp || !bp || !wf || !bf || !gamma || !beta || !e || !p || !a || !z || !mean || !rstd || !ws || !counters) return MCL_EINVAL;
  if (M <= 0 || D <= 0 || ksplit <= 0 || ksplit > MAX_KS || ldx < D || ldwp < D || ldwf < P) return MCL_EINVAL;
  if (!al16(e) || !al16(p) || !al16(a) || !al16(z) || !al16(ws) || !al16(bp) || !al16(gamma) || !al16(beta)) return MCL_EINVAL;
  HeadF h;
  h.x = x; h.ldx = ldx; h.M = M; h.D = D;
  h.wp = wp; h.ldwp = ldwp; h.bp = bp; h.wf = wf; h.ldwf = ldwf; h.bf = bf;
  h.gamma = gamma; h.beta = beta; h.eps = eps;
  h.e = e; h.p = p; h.a = a; h.z = z; h.mean = mean; h.rstd = rstd; h.ws = ws; h.cnt = counters;
  h.kchunk = (((D + ksplit - 1) / ksplit + BK - 1) / BK) * BK;
  h.ks = (D + h.kchunk - 1) / h.kchunk;            // (slices that would start past D are not launched)
  const int nrb = (M + RB - 1) / RB;
  const bool vec = al16(x) && al16(wp) && al16(wf) && ldx % 4 == 0 && ldwp % 4 == 0 && ldwf % 4 == 0 && D % 4 == 0;
  const size_t lds = FWD_LDS_FLOATS * sizeof(float);
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(proj_head_fwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(proj_head_fwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    once = true;
  }
  hipStream_t st = mcl_stream(stream);
  if (vec) hipLaunchKernelGGL(proj_head_fwd_kernel<true>, dim3(h.ks, nrb), dim3(NT), lds, st, h);
  else     hipLaunchKernelGGL(proj_head_fwd_kernel<false>, dim3(h.ks, nrb), dim3(NT), lds, st, h);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_proj_head_bwd_rows(const float* de, int64_t ldde, int32_t M, const float* z, const float* mean, const float* rstd,
                                      const float* gamma, const float* p, const float* wf, int64_t ldwf, float* dz, float* dp,
                                      float* dgamma, float* dbeta, float* dbf, float* dbp, int32_t accumulate_mask, float* ws,
                                      uint32_t* counter, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!de || !z || !mean || !rstd || !gamma || !p || !wf || !dz || !dp || !ws || !counter) return MCL_EINVAL;
  if (M <= 0 || ldde < P || ldwf < P) return MCL_EINVAL;
  if (!al16(z) || !al16(p) || !al16(dz) || !al16(dp) || !al16(gamma)) return MCL_EINVAL;
  HeadB h;
  h.de = de; h.ldde = ldde; h.M = M; h.z = z; h.mean = mean; h.rstd = rstd; h.gamma = gamma; h.p = p;
  h.wf = wf; h.ldwf = ldwf; h.dz = dz; h.dp = dp;
  h.out[0] = dgamma; h.out[1] = dbeta; h.out[2] = dbf; h.out[3] = dbp;
  h.accumulate = accumulate_mask; h.ws = ws; h.cnt = counter;
  h.nrb = (M + RB - 1) / RB;
  const size_t lds = BWD_LDS_FLOATS * sizeof(float);
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(proj_head_bwd_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    once = true;
  }
  hipLaunchKernelGGL(proj_head_bwd_rows_kernel, dim3(h.nrb), dim3(NT), lds, mcl_stream(stream), h);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}
