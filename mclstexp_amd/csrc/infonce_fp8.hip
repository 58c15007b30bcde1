// fp8 (OCP e4m3) similarity contraction for the symmetric InfoNCE (BASELINE configs[4]: "fp8 MFMA similarity GEMM";
// the contraction is /root/reference/model.py:242, cos_smi = spot_embeddings @ image_embeddings.T / temperature).
//
// Quantisation: one power-of-two scale per embedding row, x ~= q * 2^e with q in e4m3 (|q| <= 448, round to nearest
// even).  Two consequences of the scale being a power of two:
//   * the hardware applies it for free: v_mfma_scale_f32_32x32x64_f8f6f4 takes an E8M0 scale (= e + 127) per operand
//     row and 32-element k block, so the MFMA accumulates the DEQUANTISED products in fp32 -- no epilogue scaling;
//   * q * 2^e is exactly representable in bf16 (4 significand bits), so the existing bf16 gradient kernel
//     (csrc/infonce_fused.hip) run on the dequantised copy sees bit-for-bit the operands of the fp8 logits: the row /
//     column LSEs computed here normalise exactly the probabilities it forms.
// The fp8 MFMA (K = 64 per instruction) runs at twice the bf16 rate; the LSE pass itself is then bound by the
// exponentials (one v_exp_f32 per logit), see DESIGN.md 4.1.
//
// Kernels: quant_rows_kernel (fp32 -> e4m3 + E8M0 byte [+ bf16 dequantised copy]), dequant_rows_kernel,
// fp8_lse_kernel (flash-style row LSE of S = A B^T / T, logits never in HBM), rowdot (positive-pair logits).
#include "common.h"

namespace {

typedef unsigned short bf16_t;
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

constexpr int P = 256;              // embedding width
constexpr int TC = 128;             // columns per tile
constexpr int TR = 128;             // rows per workgroup (32 per wave)
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

// exact value of an e4m3 byte
__device__ __forceinline__ float e4m3_to_f(unsigned v) {
  const unsigned s = v >> 7, e = (v >> 3) & 15u, m = v & 7u;
  float f = e == 0 ? (float)m * 0.001953125f /* 2^-9 */ : __builtin_ldexpf(1.0f + (float)m * 0.125f, (int)e - 7);
  return s ? -f : f;
}

// One wave per row: amax -> exponent e = ceil(log2(amax / 448)) (so |x| 2^-e <= 448), q = e4m3_rne(x 2^-e).
__global__ __launch_bounds__(256) void quant_rows_kernel(const float* __restrict__ x, long long ldx, int rows,
                                                         unsigned char* __restrict__ q, long long ldq,
                                                         unsigned char* __restrict__ scale, long long lds_,
                                                         bf16_t* __restrict__ deq, long long ldd) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= rows) return;
  const float4 v = *reinterpret_cast<const float4*>(x + (long long)r * ldx + lane * 4);
  float amax = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
  amax = wave_max(amax);
  int e = 0;
  if (amax > 0.0f) {
    int ex;
    const float mant = frexpf(amax / 448.0f, &ex);       // amax/448 = mant * 2^ex, mant in [0.5, 1)
    e = mant == 0.5f ? ex - 1 : ex;                      // smallest e with amax <= 448 * 2^e
  }
  e = max(-126, min(126, e));
  const float inv = __builtin_ldexpf(1.0f, -e);
  const float a0 = __builtin_amdgcn_fmed3f(v.x * inv, 448.0f, -448.0f), a1 = __builtin_amdgcn_fmed3f(v.y * inv, 448.0f, -448.0f);
  const float a2 = __builtin_amdgcn_fmed3f(v.z * inv, 448.0f, -448.0f), a3 = __builtin_amdgcn_fmed3f(v.w * inv, 448.0f, -448.0f);
  unsigned w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(a0, a1, w, false);    // bytes 0, 1
  w = __builtin_amdgcn_cvt_pk_fp8_f32(a2, a3, w, true);     // bytes 2, 3
  *reinterpret_cast<unsigned*>(q + (long long)r * ldq + lane * 4) = w;
  if (lane == 0) scale[(long long)r * lds_] = (unsigned char)(e + 127);
  if (deq) {
    const float sc = __builtin_ldexpf(1.0f, e);
    const unsigned lo = pack_bf16(e4m3_to_f(w & 255u) * sc, e4m3_to_f((w >> 8) & 255u) * sc);
    const unsigned hi = pack_bf16(e4m3_to_f((w >> 16) & 255u) * sc, e4m3_to_f(w >> 24) * sc);
    *reinterpret_cast<uint2*>(deq + (long long)r * ldd + lane * 4) = make_uint2(lo, hi);
  }
}

__global__ __launch_bounds__(256) void dequant_rows_kernel(const unsigned char* __restrict__ q, long long ldq,
                                                           const unsigned char* __restrict__ scale, long long lds_,
                                                           int rows, bf16_t* __restrict__ deq, long long ldd) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= rows) return;
  const unsigned w = *reinterpret_cast<const unsigned*>(q + (long long)r * ldq + lane * 4);
  const float sc = __builtin_ldexpf(1.0f, (int)scale[(long long)r * lds_] - 127);
  const unsigned lo = pack_bf16(e4m3_to_f(w & 255u) * sc, e4m3_to_f((w >> 8) & 255u) * sc);
  const unsigned hi = pack_bf16(e4m3_to_f((w >> 16) & 255u) * sc, e4m3_to_f(w >> 24) * sc);
  *reinterpret_cast<uint2*>(deq + (long long)r * ldd + lane * 4) = make_uint2(lo, hi);
}

// Row LSE of S = A B^T / T on the fp8 operands.  Workgroup = 4 waves x 32 own rows (the MFMA "B" operand, in
// registers for the whole kernel: 4 k-steps x 32 bytes per lane); 128-column tiles of B (32 KB of e4m3) are staged
// through a double-buffered LDS tile (16-byte chunks XOR-swizzled by row so that the 32-row ds_read_b128 pattern of
// the MFMA "A" fragments is bank-conflict free).  The logits tile is computed TRANSPOSED (T[c][r]): a lane then
// holds ONE row r and 16 columns per 32 x 32 block, so the online softmax statistics are per-lane register loops.
__global__ __launch_bounds__(256, 2) void fp8_lse_kernel(const unsigned char* __restrict__ A, long long lda,
                                                         const unsigned char* __restrict__ sA, long long ldsa,
                                                         const unsigned char* __restrict__ B, long long ldb,
                                                         const unsigned char* __restrict__ sB, long long ldsb, int R,
                                                         int C, float inv_t, int nsplit, int tiles_per_split,
                                                         float2* __restrict__ stat_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_dyn[];      // 2 x 32 KB tiles + 2 x 128 scale words
  unsigned char(*tile)[TC * P] = reinterpret_cast<unsigned char(*)[TC * P]>(lds_dyn);
  int(*scl)[TC] = reinterpret_cast<int(*)[TC]>(lds_dyn + 2 * TC * P);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int split = blockIdx.x % nsplit, rtile = blockIdx.x / nsplit;
  const int row0 = rtile * TR;
  const int nct = (C + TC - 1) / TC;
  const int ct0 = split * tiles_per_split;
  const int nIt = min(nct, ct0 + tiles_per_split) - ct0;
  const int my_r = row0 + wave * 32 + l31;
  const int rr = min(my_r, R - 1);
  // own row: k-step ks covers bytes [64 ks + 32 h, +32)
  i32x8 own[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const uint4* p = reinterpret_cast<const uint4*>(A + (long long)rr * lda + 64 * ks + 32 * h);
    const uint4 u0 = p[0], u1 = p[1];
    own[ks] = i32x8{(int)u0.x, (int)u0.y, (int)u0.z, (int)u0.w, (int)u1.x, (int)u1.y, (int)u1.z, (int)u1.w};
  }
  const int own_scale = (int)sA[(long long)rr * ldsa];
  const float kscale = inv_t * LOG2E;

  // staging: 128 rows x 16 chunks of 16 bytes = 2048 chunks, 8 per thread
  uint4 st[8];
  int sreg = 127;
  auto gload = [&](int ct) {
    const int col0 = ct * TC;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = tid + 256 * i, row = c >> 4, ch = c & 15;
      const int gc = min(col0 + row, C - 1);                 // ragged right edge: repeat the last column (masked below)
      st[i] = *reinterpret_cast<const uint4*>(B + (long long)gc * ldb + ch * 16);
    }
    if (tid < TC) sreg = (int)sB[(long long)min(col0 + tid, C - 1) * ldsb];
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = tid + 256 * i, row = c >> 4, ch = c & 15;
      *reinterpret_cast<uint4*>(&tile[buf][row * P + ((ch ^ (row & 15)) << 4)]) = st[i];
    }
    if (tid < TC) scl[buf][tid] = sreg;
  };

  float run_m = -1.0e30f, run_l = 0.0f;
  if (nIt > 0) {
    gload(ct0);
    lstore(0);
  }
  __syncthreads();
  for (int it = 0; it < nIt; ++it) {
    const int buf = it & 1;
    if (it + 1 < nIt) gload(ct0 + it + 1);
    const int col0 = (ct0 + it) * TC;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const int row = cb * 32 + l31;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
      const int sa = scl[buf][row];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int ch = 4 * ks + 2 * h;
        const uint4 u0 = *reinterpret_cast<const uint4*>(&tile[buf][row * P + ((ch ^ (row & 15)) << 4)]);
        const uint4 u1 = *reinterpret_cast<const uint4*>(&tile[buf][row * P + (((ch + 1) ^ (row & 15)) << 4)]);
        const i32x8 fa = {(int)u0.x, (int)u0.y, (int)u0.z, (int)u0.w, (int)u1.x, (int)u1.y, (int)u1.z, (int)u1.w};
        // D[i = tile column][j = own row] += sum_k (B[i][k] 2^(sa-127)) (A[j][k] 2^(own_scale-127))
        acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa, own[ks], acc, 0, 0, 0, sa, 0, own_scale);
      }
      // online softmax over the 16 columns this lane holds: c = col0 + cb*32 + (i&3) + 8*(i>>2) + 4*h
      float mx = -3.0e38f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (col0 + cb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h >= C) acc[i] = -3.0e38f;
        mx = fmaxf(mx, acc[i]);
      }
      const float m_new = fmaxf(run_m, mx * kscale);
      float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        s0 += __builtin_amdgcn_exp2f(fmaf(acc[i], kscale, -m_new));
        s1 += __builtin_amdgcn_exp2f(fmaf(acc[i + 1], kscale, -m_new));
      }
      run_l = run_l * __builtin_amdgcn_exp2f(run_m - m_new) + (s0 + s1);
      run_m = m_new;
    }
    if (it + 1 < nIt) lstore(buf ^ 1);
    __syncthreads();
  }
  const float m_o = __shfl_xor(run_m, 32, 64), l_o = __shfl_xor(run_l, 32, 64);
  const float M = fmaxf(run_m, m_o);
  const float L = run_l * __builtin_amdgcn_exp2f(run_m - M) + l_o * __builtin_amdgcn_exp2f(m_o - M);
  if (h == 0 && my_r < R) stat_out[(size_t)split * R + my_r] = make_float2(M, L);
}

__global__ __launch_bounds__(256) void lse_merge8_kernel(const float2* __restrict__ stat, int R, int nsplit,
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

// diag[r] = inv_t * a[r] . b[r + diag_off] on bf16 rows (the dequantised copies: exact fp8 products), one wave per row
__global__ __launch_bounds__(256) void rowdot16_kernel(const bf16_t* __restrict__ a, long long lda,
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

struct Plan8 {
  int rt, nct, nsplit, tps;
};
inline Plan8 plan8(int R, int C) {
  Plan8 p;
  p.rt = (R + TR - 1) / TR;
  p.nct = (C + TC - 1) / TC;
  int want = (512 + p.rt - 1) / p.rt;      // ~2 workgroups per CU when the column count allows
  if (want < 1) want = 1;
  p.nsplit = want < p.nct ? want : p.nct;
  p.tps = (p.nct + p.nsplit - 1) / p.nsplit;
  p.nsplit = (p.nct + p.tps - 1) / p.tps;
  return p;
}
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

extern "C" int mcl_quant_e4m3_rows(const float* x, int64_t ldx, int32_t rows, int32_t cols, void* q, int64_t ldq,
                                   void* scale, int64_t ld_scale, void* deq_bf16, int64_t ldd, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!x || !q || !scale || rows <= 0) return MCL_EINVAL;
  if (cols != P || (ldx % 4) || (ldq % 4) || (ldd % 4) || !al16(x) || (reinterpret_cast<uintptr_t>(q) & 3u) ||
      (deq_bf16 && (reinterpret_cast<uintptr_t>(deq_bf16) & 7u)))
    return MCL_EUNSUPPORTED;
  hipLaunchKernelGGL(quant_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, mcl_stream(stream), x, (long long)ldx, rows,
                     (unsigned char*)q, (long long)ldq, (unsigned char*)scale, (long long)ld_scale, (bf16_t*)deq_bf16,
                     (long long)ldd);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_dequant_e4m3_rows(const void* q, int64_t ldq, const void* scale, int64_t ld_scale, int32_t rows,
                                     int32_t cols, void* deq_bf16, int64_t ldd, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!q || !scale || !deq_bf16 || rows <= 0) return MCL_EINVAL;
  if (cols != P || (ldq % 4) || (ldd % 4) || (reinterpret_cast<uintptr_t>(q) & 3u) ||
      (reinterpret_cast<uintptr_t>(deq_bf16) & 7u))
    return MCL_EUNSUPPORTED;
  hipLaunchKernelGGL(dequant_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, mcl_stream(stream),
                     (const unsigned char*)q, (long long)ldq, (const unsigned char*)scale, (long long)ld_scale, rows,
                     (bf16_t*)deq_bf16, (long long)ldd);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int64_t mcl_infonce_fp8_workspace_bytes(int32_t R, int32_t C) {
  if (R <= 0 || C <= 0) return -1;
  return (int64_t)plan8(R, C).nsplit * R * (int64_t)sizeof(float2);
}

extern "C" int mcl_infonce_fp8_lse(const void* a8, int64_t lda, const void* scale_a, int64_t ld_sa, const void* b8,
                                   int64_t ldb, const void* scale_b, int64_t ld_sb, int32_t R, int32_t C, int32_t dim,
                                   float inv_temp, float* lse, void* workspace, int64_t ws_bytes, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!a8 || !b8 || !scale_a || !scale_b || !lse || !workspace || R <= 0 || C <= 0 || !(inv_temp > 0.0f)) return MCL_EINVAL;
  if (dim != P || !al16(a8) || !al16(b8) || (lda % 16) || (ldb % 16) || lda < P || ldb < P) return MCL_EUNSUPPORTED;
  const Plan8 p = plan8(R, C);
  if (ws_bytes < (int64_t)p.nsplit * R * (int64_t)sizeof(float2)) return MCL_EWORKSPACE;
  hipStream_t st = mcl_stream(stream);
  constexpr size_t lds_bytes = 2 * TC * P + 2 * TC * sizeof(int);
  static mcl_device_once attr_once;
  if (auto attr_guard = attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fp8_lse_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds_bytes);
  }
  hipLaunchKernelGGL(fp8_lse_kernel, dim3(p.rt * p.nsplit), dim3(256), lds_bytes, st, (const unsigned char*)a8, (long long)lda,
                     (const unsigned char*)scale_a, (long long)ld_sa, (const unsigned char*)b8, (long long)ldb,
                     (const unsigned char*)scale_b, (long long)ld_sb, R, C, inv_temp, p.nsplit, p.tps,
                     (float2*)workspace);
  hipLaunchKernelGGL(lse_merge8_kernel, dim3((R + 255) / 256), dim3(256), 0, st, (const float2*)workspace, R, p.nsplit,
                     lse);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_infonce_rowdot_bf16(const void* a, int64_t lda, const void* b, int64_t ldb, int32_t R, int32_t C,
                                       int32_t dim, int32_t diag_off, float inv_temp, float* diag, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!a || !b || !diag || R <= 0 || C <= 0) return MCL_EINVAL;
  if (dim != P || (lda % 4) || (ldb % 4) || (reinterpret_cast<uintptr_t>(a) & 7u) || (reinterpret_cast<uintptr_t>(b) & 7u))
    return MCL_EUNSUPPORTED;
  hipLaunchKernelGGL(rowdot16_kernel, dim3((R + 3) / 4), dim3(256), 0, mcl_stream(stream), (const bf16_t*)a,
                     (long long)lda, (const bf16_t*)b, (long long)ldb, R, C, diag_off, inv_temp, diag);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}
