// A whole torchvision _DenseBlock FORWARD on the 7 x 7 maps as ONE persistent launch (/root/reference/model.py:75-76 via
// torchvision: denseblock4 = 16 x [norm1 -> relu1 -> conv1 (1x1, C_in -> 128) -> norm2 -> relu2 -> conv2 (3x3, 128 -> 32) ->
// concat]; train-mode BatchNorm: batch statistics).
//
// Why: on this map a layer is ~13 MFLOP per image, yet the per-layer launch sequence (1x1 + statistics finalize + 3x3 +
// statistics finalize) costs 42 us -- four dependent launches whose K loops are latency chains on 98 workgroups.  Here ONE
// workgroup owns ONE image for the whole block: its 49 pixels are the M dimension of every product, the 3x3 convolution
// needs no halo from anybody else, and the only thing the images of a batch ever exchange is what BatchNorm's batch
// statistics force: per layer two all-to-all seams of (sum, M2) pairs -- 128 channels after the 1x1, 32 after the 3x3.
//
// MI355X mapping.  grid = B workgroups of 4 waves, all co-resident (B <= CUs, checked by the launcher).  Per layer:
//   1x1:  z[49 x 128] = relu(bn1(x))[49 x C_in] * W1^T.  x is re-read from global memory (the image's rows are written and
//         read by this CU only: L2 hits) in 128-channel chunks, transformed on the way into a double-buffered LDS tile; wave w
//         owns output channels 32w..32w+31 for both 32-row pixel tiles; its weight fragments come STRAIGHT from global
//         memory into the MFMA operand registers (W1 rows are k-contiguous): the contraction index inside a chunk is permuted
//         so that a lane's eight fragments of a chunk are 128 contiguous bytes.  Next chunk's x and W loads are in flight
//         while the current chunk multiplies.
//   seam: per-image (sum, M2) of the bf16-rounded z -> write-through (sc1) 8-byte stores -> every storing wave drains ->
//         one flag per image; every workgroup polls the B flags (one wave, relaxed sc1 loads, s_sleep), then reads all B
//         records with sc1 loads and merges them in double in a fixed order (Chan) -- every workgroup computes the same
//         mean / var / rstd bit for bit; image 0 stores them for the backward.  No fence, no atomic read-modify-write.
//   3x3:  relu(bn2(z)) is written into a zero-bordered 9 x 9 LDS tile, the nine taps are row offsets; the 72 k-steps are
//         split over the four waves (18 weight fragments each, prefetched before the seam), partials meet in LDS.
//   seam: (sum, M2) of the 32 new channels, as above; the new statistics stay in LDS for the later layers.
// Every spin is bounded: a timeout sets *err and the kernel runs to its end without waiting again (results invalid).
#include "common.h"
#include <type_traits>

namespace {

typedef unsigned short bf16_t;
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned long long u64;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int DB_MAXL = 24;
constexpr int HW = 49, MAPW = 7, PW = 9;          // pixels per image, map width, padded width
constexpr int ZERO_ROW = PW * PW;                 // a2 tile row of zeros (pixels 49..63 of the second MFMA tile)

struct DBLayer {
  const float* g1;
  const float* b1;
  const bf16_t* w1;     // PACKED conv1 weight of this layer (mcl_dense_block_pack_w1): MFMA-fragment order, 128*cin elements
  const float* g2;
  const float* b2;
  const bf16_t* w2;     // [32][3][3][128]
  bf16_t* z;            // [B][49][128]
  float* m2;
  float* v2;
  float* r2;
};
struct DBArgs {
  bf16_t* buf;          // [B][49][Ct]; channels [0, C0) hold the block input
  int Ct, B, C0, L;
  float eps1, eps2;
  float* mean;          // [Ct] statistics of the concat buffer: [0, C0) given, the rest written (image 0)
  float* var;
  float* rstd;
  u64* xch;             // [2L][B][128] (sum, M2) records
  unsigned* flags;      // [2L][B], zero before the launch
  int* err;
  unsigned max_spins;   // bound of every seam poll loop
  unsigned long long* dbg;   // optional in-kernel phase stamps [B][L][8] (wall clock, 10 ns); nullptr in production
  DBLayer ly[DB_MAXL];
};

__device__ __forceinline__ unsigned pack2(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xFFFF0000u); }
__device__ __forceinline__ float round_bf16(float v) { return bf_lo(pack2(v, 0.0f)); }

__device__ __forceinline__ uint4 bn_relu_chunk(uint4 v, const float* sc, const float* sh) {
  unsigned w[4] = {v.x, v.y, v.z, v.w};
  const float4 s0 = *reinterpret_cast<const float4*>(sc), s1 = *reinterpret_cast<const float4*>(sc + 4);
  const float4 t0 = *reinterpret_cast<const float4*>(sh), t1 = *reinterpret_cast<const float4*>(sh + 4);
  const float s[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
  const float t[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float lo = fmaxf(fmaf(bf_lo(w[i]), s[2 * i], t[2 * i]), 0.0f);
    const float hi = fmaxf(fmaf(bf_hi(w[i]), s[2 * i + 1], t[2 * i + 1]), 0.0f);
    w[i] = pack2(lo, hi);
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

__device__ __forceinline__ void store_rec(u64* p, float sum, float m2) {
  const u64 v = ((u64)__float_as_uint(m2) << 32) | (u64)__float_as_uint(sum);
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // write-through (sc1) 8-byte store
}
__device__ __forceinline__ u64 load_rec(const u64* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1 load: bypasses this CU's L1
}

// One wave polls the B flags of a seam; bounded.  `dead` (LDS) is sticky: after a timeout nobody waits again.
__device__ __forceinline__ void seam_wait(const unsigned* flags, int B, int tid, int* dead, int* err, unsigned max_spins) {
  if (tid < 64) {
    if (*dead == 0) {
      bool ok = false;
      for (unsigned spins = 0; spins < max_spins; ++spins) {
        ok = true;
        for (int i = tid; i < B; i += 64) ok &= __hip_atomic_load(flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
        if (__all(ok)) break;
        __builtin_amdgcn_s_sleep(8);
      }
      if (!__all(ok) && tid == 0) {
        *dead = 1;
        atomicExch(err, 1);
      }
    }
  }
  __syncthreads();
}

#define DB_STAMP(k)                                                                                   \
  do {                                                                                                 \
    if (a.dbg && tid == 0) a.dbg[((long long)img * a.L + l) * 8 + (k)] = wall_clock64();               \
  } while (0)

// One chunk of W (32 ... 128 channels = W/16 k-steps) times the transformed x tile A ([64 rows][256 B], swizzled), AND the
// BatchNorm+ReLU transform of the NEXT chunk's raw pieces into the next LDS tile -- one basic block, so that the VALU work
// fills the issue slots the MFMA stream leaves (one wave per SIMD: nothing else would).  The two row tiles alternate so that
// consecutive MFMAs never share an accumulator.  W is a template value: straight-line code.
template <int W, bool STAGE>
__device__ __forceinline__ void mul_chunk(const unsigned char* A, const u32x4 (&wv)[8], f32x16 (&acc)[2], int h, int l31,
                                          unsigned char* An, const u32x4 (&xv)[4], const float* tsc, const float* tsh, int cc,
                                          int rr) {
  constexpr int NK = W / 16;
  const int hb = h * NK;                                        // first 16-byte chunk of this lane half
  constexpr int NH = NK >= 4 ? NK / 2 : NK;                     // fragments are read half a chunk at a time (32 registers)
  uint4 o[3];
#pragma unroll
  for (int k0 = 0; k0 < NK; k0 += NH) {
    bf16x8 fa[NH][2];
#pragma unroll
    for (int kk = 0; kk < NH; ++kk) {
      const int co = ((hb + k0 + kk) ^ (l31 & 15)) << 4;
      fa[kk][0] = *reinterpret_cast<const bf16x8*>(A + l31 * 256 + co);
      fa[kk][1] = *reinterpret_cast<const bf16x8*>(A + (l31 + 32) * 256 + co);
    }
    if (STAGE && k0 == 0) {
#pragma unroll
      for (int i = 0; i < 3; ++i) o[i] = bn_relu_chunk(make_uint4(xv[i][0], xv[i][1], xv[i][2], xv[i][3]), tsc, tsh);
    }
#pragma unroll
    for (int kk = 0; kk < NH; ++kk) {
      const bf16x8 fb = __builtin_bit_cast(bf16x8, wv[k0 + kk]);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk][0], fb, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk][1], fb, acc[1], 0, 0, 0);
    }
  }
  if (STAGE) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int r = rr + 16 * i;
      *reinterpret_cast<uint4*>(An + r * 256 + ((cc ^ (r & 15)) << 4)) = o[i];
    }
    if (rr == 0)                                                // rows 48..63: only row 48 exists
      *reinterpret_cast<uint4*>(An + 48 * 256 + (cc << 4)) =
          bn_relu_chunk(make_uint4(xv[3][0], xv[3][1], xv[3][2], xv[3][3]), tsc, tsh);
  }
}

__global__ __launch_bounds__(256, 1) void dense_block_fwd_kernel(DBArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float* s_mean = reinterpret_cast<float*>(lds);                    // [1024] statistics of the concat channels
  float* s_rstd = s_mean + 1024;                                    // [1024]
  float* tab = s_rstd + 1024;                                       // BN1: scale[1024], shift[1024]
  unsigned char* At = reinterpret_cast<unsigned char*>(tab + 2048); // 3 x [64][256 B] transformed x chunks (48 KB); 3x3 partials
  unsigned char* zt = At + 3 * 16384;                               // [64][256 B] bf16 z tile; later fp32 scratch of the 3x3 tail
  unsigned char* a2t = zt + 16384;                                  // [82][256 B] zero-bordered relu(bn2(z))
  float* tab2 = reinterpret_cast<float*>(a2t + 82 * 256);           // BN2: scale[128], shift[128]
  double* dred = reinterpret_cast<double*>(tab2 + 256);             // 1024 doubles (seam reductions)
  int* dead = reinterpret_cast<int*>(dred + 1024);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int img = blockIdx.x, B = a.B, Ct = a.Ct;
  bf16_t* xb = a.buf + (long long)img * HW * Ct;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(xb, 0, (unsigned)(HW * Ct * 2), 0x00020000);

  for (int i = tid; i < 82 * 16; i += 256) reinterpret_cast<uint4*>(a2t)[i] = make_uint4(0u, 0u, 0u, 0u);
  for (int k = tid; k < a.C0; k += 256) {
    s_mean[k] = a.mean[k];
    s_rstd[k] = a.rstd[k];
  }
  if (tid == 0) *dead = 0;
  __syncthreads();

  // staging role of a thread inside a chunk: 16-byte piece column cc (8 channels), rows rr + 16*i
  const int cc = tid & 15, rr = tid >> 4;
  unsigned xoff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = rr + 16 * i;
    xoff[i] = r < HW ? (unsigned)((r * Ct + cc * 8) * 2) : 0xFFFFF000u;       // rows past the image: out of range -> zeros
  }
  // this lane's pixel in the padded 9 x 9 tile, per MFMA row tile (tap (0, 0) position); rows >= 49 read the zero row
  int prow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = i * 32 + l31;
    prow[i] = p < HW ? (p / MAPW) * PW + (p % MAPW) : -1;
  }

  f32x16 acc[2];
  u32x4 xr[3][4], wr[3][8];

  // BN1 coefficient table of a layer over the channels [0, K) (K <= 1024, a multiple of 4): one shot
  auto build_tab = [&](const float* g1, const float* b1, int K) {
    const int k = tid * 4;
    if (k < K) {
      const float4 g = *reinterpret_cast<const float4*>(g1 + k), b = *reinterpret_cast<const float4*>(b1 + k);
      const float4 m = *reinterpret_cast<const float4*>(s_mean + k), r = *reinterpret_cast<const float4*>(s_rstd + k);
      const float4 sc = make_float4(g.x * r.x, g.y * r.y, g.z * r.z, g.w * r.w);
      *reinterpret_cast<float4*>(tab + k) = sc;
      *reinterpret_cast<float4*>(tab + 1024 + k) =
          make_float4(fmaf(-m.x, sc.x, b.x), fmaf(-m.y, sc.y, b.y), fmaf(-m.z, sc.z, b.z), fmaf(-m.w, sc.w, b.w));
    }
  };

  // acc = relu(bn1(x[:, 0:K])) W1[:, 0:K]^T over 128-channel chunks (the last one 32 / 64 / 96 wide); W1 rows are cin_w long
  auto kloop = [&](const bf16_t* w1, int cin_w, int K) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    const int nchunk = (K + 127) >> 7;
    // packed weight: chunk (c0, w) starts at byte c0 * 256; inside it unit (wave, kk) is 1 KiB = 64 lanes x 16 bytes, a wave's
    // w/16 units are contiguous: every load instruction reads 1 KiB of consecutive memory (8 cache lines, not 64)
    const __amdgpu_buffer_rsrc_t wrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(w1), 0, (unsigned)(128 * cin_w * 2), 0x00020000);
    (void)cin_w;
    auto load_chunk = [&](u32x4 (&xv)[4], u32x4 (&wv)[8], int c) {
      const int c0 = c << 7;
      const int w = min(128, K - c0);                         // chunk width: 32, 64, 96 or 128 channels (<= 0: no chunk)
      const bool xlive = cc * 8 < w;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#ifdef DB_EXP_NOX
        xv[i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, (unsigned)c};
#else
        xv[i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, xlive ? xoff[i] + (unsigned)c0 * 2u : 0xFFFFF000u, 0, 0);
#endif
      // lane (n, h), k-step kk < w/16: channels c0 + h*w/2 + 8*kk .. +8 of row n = wave*32 + l31, at its packed position
      const unsigned wb = (unsigned)c0 * 256u + (unsigned)(wave * (w >> 4)) * 1024u + (unsigned)lane * 16u;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk)
#ifdef DB_EXP_NOW
        wv[kk] = u32x4{0x3c003c00u, 0x3c003c00u, wb, (unsigned)kk};
#else
        wv[kk] = __builtin_amdgcn_raw_buffer_load_b128(wrs, (w > 0 && kk * 16 < w) ? wb + kk * 1024u : 0xFFFFF000u, 0, 0);
#endif
    };
    // (debug: cycle stamps of one workgroup's chunk loop, K = 960: [chunk][multiplied + staged, loads issued, barrier passed])
    unsigned long long* cst = (a.dbg && img == 0 && tid == 0 && K == 960) ? a.dbg + (long long)B * a.L * 8 : nullptr;
    // chunk c lives in LDS tile c % 3 and register set c % 3.  Interval c: multiply chunk c while chunk c + 1 is transformed
    // into its tile (loaded two intervals ago), then request chunk c + 3 into the set chunk c has just freed; ONE barrier.
    auto stage_first = [&](const u32x4 (&xv)[4]) {
      const int kc = cc * 8;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i < 3 || rr == 0) {
          const int r = rr + 16 * i;
          const uint4 v = bn_relu_chunk(make_uint4(xv[i][0], xv[i][1], xv[i][2], xv[i][3]), tab + kc, tab + 1024 + kc);
          *reinterpret_cast<uint4*>(At + r * 256 + ((cc ^ (r & 15)) << 4)) = v;
        }
      }
    };
    auto interval = [&](auto SET, int c) {
      constexpr int S = decltype(SET)::value, SN = (S + 1) % 3;
      const unsigned char* A = At + S * 16384;
      unsigned char* An = At + SN * 16384;
      const int w = min(128, K - (c << 7));
      const int kcn = min(((c + 1) << 7) + cc * 8, 1016);       // next chunk's table entries (past K: any finite ones)
      if (w == 128) mul_chunk<128, true>(A, wr[S], acc, h, l31, An, xr[SN], tab + kcn, tab + 1024 + kcn, cc, rr);
      else if (w == 96) mul_chunk<96, false>(A, wr[S], acc, h, l31, An, xr[SN], tab, tab, cc, rr);
      else if (w == 64) mul_chunk<64, false>(A, wr[S], acc, h, l31, An, xr[SN], tab, tab, cc, rr);
      else if (w == 32) mul_chunk<32, false>(A, wr[S], acc, h, l31, An, xr[SN], tab, tab, cc, rr);
      if (cst) cst[c * 4 + 0] = clock64();
      load_chunk(xr[S], wr[S], c + 3);
      if (cst) cst[c * 4 + 1] = clock64();
      __syncthreads();
      if (cst) cst[c * 4 + 2] = clock64();
    };
    if (cst) cst[3] = clock64();
    load_chunk(xr[0], wr[0], 0);
    load_chunk(xr[1], wr[1], 1);
    load_chunk(xr[2], wr[2], 2);
    stage_first(xr[0]);
    __syncthreads();
    if (cst) cst[7] = clock64();
    // always three intervals per trip (a chunk past the range loads nothing and multiplies nothing): every path through the
    // loop issues the same loads, so the compiler's vmcnt bookkeeping stays exact and the prefetched chunks stay in flight
    for (int c = 0; c < nchunk; c += 3) {
      interval(std::integral_constant<int, 0>{}, c);
      interval(std::integral_constant<int, 1>{}, c + 1);
      interval(std::integral_constant<int, 2>{}, c + 2);
    }
  };

  // ---------------------------------------------------------------- layer 0: the whole 1x1 (every input channel is known)
  build_tab(a.ly[0].g1, a.ly[0].b1, a.C0);
  __syncthreads();
  kloop(a.ly[0].w1, a.C0, a.C0);

  for (int l = 0; l < a.L; ++l) {
    const DBLayer ly = a.ly[l];
    const int cin = a.C0 + 32 * l;
    DB_STAMP(0);
    if (l > 0) {
      // ---- the last 32 input channels: produced by layer l-1, their statistics arrive through its second seam
      u32x4 wsl[8];
      {
        const __amdgpu_buffer_rsrc_t wrs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(ly.w1), 0, (unsigned)(128 * cin * 2), 0x00020000);
        const unsigned wb = (unsigned)(cin - 32) * 256u + (unsigned)(wave * 2) * 1024u + (unsigned)lane * 16u;   // packed slice chunk
        wsl[0] = __builtin_amdgcn_raw_buffer_load_b128(wrs, wb, 0, 0);
        wsl[1] = __builtin_amdgcn_raw_buffer_load_b128(wrs, wb + 1024u, 0, 0);
      }
      float gsl = 0.0f, bsl = 0.0f;
      if (tid < 32) {
        gsl = ly.g1[cin - 32 + tid];
        bsl = ly.b1[cin - 32 + tid];
      }
      seam_wait(a.flags + (2 * l - 1) * B, B, tid, dead, a.err, a.max_spins);
      DB_STAMP(6);
      {
        const int cp = tid & 15, part = tid >> 4;               // 16 channel pairs x 16 image groups
        const int per = (B + 15) >> 4, i0 = part * per, i1 = min(B, i0 + per);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<void*>(a.xch + (long long)(2 * l - 1) * B * 128), 0, (unsigned)(B * 1024), 0x00020000);
        double s0 = 0.0, q0 = 0.0, s1 = 0.0, q1 = 0.0;
        for (int i = i0; i < i1; i += 16) {
          u32x4 v[16];
#pragma unroll
          for (int u = 0; u < 16; ++u)
            v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, i + u < i1 ? (unsigned)((i + u) * 1024 + cp * 16) : 0xFFFFF000u, 0, 16);
#pragma unroll
          for (int u = 0; u < 16; ++u) {                         // (records past the range read as zeros: they add nothing)
            const double sa = (double)__uint_as_float(v[u][0]), sb = (double)__uint_as_float(v[u][2]);
            s0 += sa;
            q0 += (double)__uint_as_float(v[u][1]) + sa * sa * (1.0 / HW);
            s1 += sb;
            q1 += (double)__uint_as_float(v[u][3]) + sb * sb * (1.0 / HW);
          }
        }
        double* d = dred + (part * 32 + 2 * cp) * 2;
        d[0] = s0; d[1] = q0; d[2] = s1; d[3] = q1;
      }
      __syncthreads();
      if (tid < 32) {
        double sum = 0.0, q = 0.0;
#pragma unroll
        for (int p = 0; p < 16; ++p) {
          sum += dred[(p * 32 + tid) * 2];
          q += dred[(p * 32 + tid) * 2 + 1];
        }
        const double n = (double)B * HW, m = sum / n;
        double v = (q - sum * sum / n) / n;
        if (v < 0.0) v = 0.0;
        const float mf = (float)m, rf = (float)(1.0 / sqrt(v + (double)a.eps1));
        const int k = cin - 32 + tid;
        s_mean[k] = mf;
        s_rstd[k] = rf;
        const float sc = gsl * rf;
        tab[k] = sc;
        tab[1024 + k] = fmaf(-mf, sc, bsl);
        if (img == 0) {
          a.mean[k] = mf;
          a.var[k] = (float)v;
          a.rstd[k] = rf;
        }
      }
      __syncthreads();
      {
        // the slice straight from the rounded values the 3x3 tail left in LDS (rv[64][32] floats): no global round trip
        const float* rv = reinterpret_cast<const float*>(zt);
        const int px = tid >> 2, c0 = (tid & 3) * 8, k = cin - 32 + c0;
        unsigned w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float lo = fmaxf(fmaf(rv[px * 32 + c0 + 2 * q], tab[k + 2 * q], tab[1024 + k + 2 * q]), 0.0f);
          const float hi = fmaxf(fmaf(rv[px * 32 + c0 + 2 * q + 1], tab[k + 2 * q + 1], tab[1024 + k + 2 * q + 1]), 0.0f);
          w[q] = pack2(lo, hi);
        }
        *reinterpret_cast<uint4*>(At + px * 256 + (((c0 >> 3) ^ (px & 15)) << 4)) = make_uint4(w[0], w[1], w[2], w[3]);
      }
      __syncthreads();
      mul_chunk<32, false>(At, wsl, acc, h, l31, At, xr[0], tab, tab, cc, rr);
      DB_STAMP(7);
    }
    DB_STAMP(1);

    // ---- z epilogue.  acc[i][r]: pixel i*32 + (r&3) + 8*(r>>2) + 4*h, channel wave*32 + l31
    {
      const int c = wave * 32 + l31;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = round_bf16(acc[i][r]);       // the statistics are those of the stored tensor
      const float ks = __shfl(acc[0][0], l31, 64);                          // pixel 0 (h == 0 lanes hold rows 0..3)
      float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const float v = acc[i][r];
          if (row < HW) {
            const float d = v - ks;
            s1 += d;
            s2 = fmaf(d, d, s2);
          }
          // z tile [row][channel] bf16, 16-byte chunk (c >> 3) swizzled by the row
          reinterpret_cast<bf16_t*>(zt + row * 256 + (((c >> 3) ^ (row & 15)) << 4))[c & 7] = (bf16_t)(__float_as_uint(v) >> 16);
        }
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (h == 0) {
        const float n = (float)HW;
        store_rec(a.xch + ((long long)(2 * l) * B + img) * 128 + c, fmaf(n, ks, s1), s2 - s1 * s1 / n);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every storing wave drains its write-through stores
    __syncthreads();
    if (tid == 0) __hip_atomic_store(a.flags + (2 * l) * B + img, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    DB_STAMP(2);
    // this wave's 18 weight fragments of the 3x3 and the BatchNorm-2 affine: requested now, in flight during the seam
    bf16x8 breg[18];
#pragma unroll
    for (int i = 0; i < 18; ++i) {
      const int kidx = 18 * wave + i;
      breg[i] = *reinterpret_cast<const bf16x8*>(ly.w2 + ((long long)l31 * 9 + (kidx >> 3)) * 128 + 16 * (kidx & 7) + 8 * h);
    }
    float g2v = 0.0f, b2v = 0.0f;
    if (tid < 128) {
      g2v = ly.g2[tid];
      b2v = ly.b2[tid];
    }
    // z to global memory (the backward's copy), off the seam's critical path: whole 256-byte rows
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = rr + 16 * i;
      if (r < HW)
        *reinterpret_cast<uint4*>(ly.z + ((long long)img * HW + r) * 128 + cc * 8) =
            *reinterpret_cast<const uint4*>(zt + r * 256 + ((cc ^ (r & 15)) << 4));
    }

    // ---------------------------------------------------------------- seam 1: batch statistics of z
    seam_wait(a.flags + (2 * l) * B, B, tid, dead, a.err, a.max_spins);
    DB_STAMP(3);
    {
      const int cp = tid & 63, part = tid >> 6;                  // 64 channel pairs x 4 image groups
      const int per = (B + 3) >> 2, i0 = part * per, i1 = min(B, i0 + per);
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          reinterpret_cast<void*>(a.xch + (long long)(2 * l) * B * 128), 0, (unsigned)(B * 1024), 0x00020000);
      double s0 = 0.0, q0 = 0.0, s1 = 0.0, q1 = 0.0;
      for (int i = i0; i < i1; i += 16) {
        u32x4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u)
          v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, i + u < i1 ? (unsigned)((i + u) * 1024 + cp * 16) : 0xFFFFF000u, 0, 16);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const double sa = (double)__uint_as_float(v[u][0]), sb = (double)__uint_as_float(v[u][2]);
          s0 += sa;
          q0 += (double)__uint_as_float(v[u][1]) + sa * sa * (1.0 / HW);
          s1 += sb;
          q1 += (double)__uint_as_float(v[u][3]) + sb * sb * (1.0 / HW);
        }
      }
      double* d = dred + (part * 128 + 2 * cp) * 2;
      d[0] = s0; d[1] = q0; d[2] = s1; d[3] = q1;
    }
    __syncthreads();
    // the NEXT layer's packed 1x1 weight: touch every cache line now (one dword per 128-byte line, results discarded), so
    // that the chunk loop after the 3x3 finds it in this XCD's L2 instead of paying HBM latency at its head.  Issued here --
    // after the seam's records have been read -- so that neither the flag poll nor the record loads queue behind it.
    // Shared duty: block b runs on XCD b % 8 (observed placement -- used for speed only), so the 16 workgroups of an XCD
    // touch one sixteenth of the lines each (<= 124 lines: one load per thread of two waves) and fill their common L2.
    unsigned warm = 0u;
    if (l + 1 < a.L) {
      const __amdgpu_buffer_rsrc_t wns = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.ly[l + 1].w1), 0,
                                                                           (unsigned)(cin * 256), 0x00020000);
      const unsigned line = (unsigned)((img >> 3) & 15) + 16u * (unsigned)tid;
      warm = __builtin_amdgcn_raw_buffer_load_b32(wns, tid < 128 ? line << 7 : 0xFFFFF000u, 0, 0);   // (past the end: no access)
    }
    if (tid < 128) {
      double sum = 0.0, q = 0.0;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        sum += dred[(p * 128 + tid) * 2];
        q += dred[(p * 128 + tid) * 2 + 1];
      }
      const double n = (double)B * HW, m = sum / n;
      double v = (q - sum * sum / n) / n;
      if (v < 0.0) v = 0.0;
      const float mf = (float)m, rf = (float)(1.0 / sqrt(v + (double)a.eps2));
      const float sc = g2v * rf;
      tab2[tid] = sc;
      tab2[128 + tid] = fmaf(-mf, sc, b2v);
      if (img == 0) {
        ly.m2[tid] = mf;
        ly.v2[tid] = (float)v;
        ly.r2[tid] = rf;
      }
    }
    __syncthreads();
    DB_STAMP(4);
    // ---------------------------------------------------------------- relu(bn2(z)) into the zero-bordered 9 x 9 tile
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = rr + 16 * i;
      if (r < HW) {
        const uint4 v = *reinterpret_cast<const uint4*>(zt + r * 256 + ((cc ^ (r & 15)) << 4));
        const int pr = (r / MAPW + 1) * PW + (r % MAPW) + 1;
        *reinterpret_cast<uint4*>(a2t + pr * 256 + ((cc ^ (pr & 15)) << 4)) = bn_relu_chunk(v, tab2 + cc * 8, tab2 + 128 + cc * 8);
      }
    }
    __syncthreads();
    // ---------------------------------------------------------------- 3x3: 72 k-steps split over the four waves
    f32x16 acc3[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc3[i][r] = 0.0f;
#pragma unroll
    for (int i = 0; i < 18; ++i) {
      const int kidx = 18 * wave + i, tap = kidx >> 3, kk = kidx & 7;
      const int toff = (tap / 3) * PW + (tap % 3);
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const int row = prow[mb] >= 0 ? prow[mb] + toff : ZERO_ROW;
        const bf16x8 fa = *reinterpret_cast<const bf16x8*>(a2t + row * 256 + (((2 * kk + h) ^ (row & 15)) << 4));
        acc3[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, breg[i], acc3[mb], 0, 0, 0);
      }
    }
    float* red = reinterpret_cast<float*>(At);                  // [4][64][32] fp32 K-partials (the x chunks are dead)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int px = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        red[(wave * 64 + px) * 32 + l31] = acc3[mb][r];
      }
    __syncthreads();
    float* rv = reinterpret_cast<float*>(zt);                   // [64][32] rounded y (the z tile is dead)
    float* grp = rv + 64 * 32;                                  // [8][32] float2
    unsigned pk[4];
    const int px = tid >> 2, c0 = (tid & 3) * 8;
    {
      float v[8];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float4 s = *reinterpret_cast<const float4*>(red + px * 32 + c0 + 4 * q);
#pragma unroll
        for (int w = 1; w < 4; ++w) {
          const float4 t = *reinterpret_cast<const float4*>(red + (w * 64 + px) * 32 + c0 + 4 * q);
          s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        v[4 * q] = s.x; v[4 * q + 1] = s.y; v[4 * q + 2] = s.z; v[4 * q + 3] = s.w;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        pk[q] = pack2(v[2 * q], v[2 * q + 1]);
        rv[px * 32 + c0 + 2 * q] = bf_lo(pk[q]);
        rv[px * 32 + c0 + 2 * q + 1] = bf_hi(pk[q]);
      }
    }
    __syncthreads();
    {
      const int c = tid & 31, g = tid >> 5;
      const float ks = rv[c];
      float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int pp = g * 8 + q;
        if (pp < HW) {
          const float d = rv[pp * 32 + c] - ks;
          s1 += d;
          s2 = fmaf(d, d, s2);
        }
      }
      grp[(g * 32 + c) * 2] = s1;
      grp[(g * 32 + c) * 2 + 1] = s2;
    }
    __syncthreads();
    if (tid < 32) {
      float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        s1 += grp[(g * 32 + tid) * 2];
        s2 += grp[(g * 32 + tid) * 2 + 1];
      }
      const float n = (float)HW;
      store_rec(a.xch + ((long long)(2 * l + 1) * B + img) * 128 + tid, fmaf(n, rv[tid], s1), s2 - s1 * s1 / n);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (tid == 0) __hip_atomic_store(a.flags + (2 * l + 1) * B + img, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    DB_STAMP(5);
    // the new 32 channels into the concat buffer (the later layers of this workgroup and the backward read them)
    if (px < HW)
      *reinterpret_cast<uint4*>(xb + (long long)px * Ct + cin + c0) = make_uint4(pk[0], pk[1], pk[2], pk[3]);

    // ---------------------------------------------------------------- the NEXT layer's 1x1 over every channel that is already
    // final ([0, cin): they do not depend on this layer), while the second seam's records travel
    if (l + 1 < a.L) {
      if (warm == 0x7fc0dead && tid == 255) atomicExch(a.err, 2);   // (consumes the warm-up load: practically never true)
      build_tab(a.ly[l + 1].g1, a.ly[l + 1].b1, cin);
      __syncthreads();
      kloop(a.ly[l + 1].w1, cin + 32, cin);
    }
  }

  // ---------------------------------------------------------------- the last layer's new channels: their statistics (image 0
  // stores them for the block's consumer and the backward)
  if (img != 0) return;
  {
    const int l = a.L;                                            // (seam index 2L - 1)
    const int cin = a.C0 + 32 * l;
    seam_wait(a.flags + (2 * l - 1) * B, B, tid, dead, a.err, a.max_spins);
    const int cp = tid & 15, part = tid >> 4;
    const int per = (B + 15) >> 4, i0 = part * per, i1 = min(B, i0 + per);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void*>(a.xch + (long long)(2 * l - 1) * B * 128), 0, (unsigned)(B * 1024), 0x00020000);
    double s0 = 0.0, q0 = 0.0, s1 = 0.0, q1 = 0.0;
    for (int i = i0; i < i1; i += 16) {
      u32x4 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u)
        v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, i + u < i1 ? (unsigned)((i + u) * 1024 + cp * 16) : 0xFFFFF000u, 0, 16);
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const double sa = (double)__uint_as_float(v[u][0]), sb = (double)__uint_as_float(v[u][2]);
        s0 += sa;
        q0 += (double)__uint_as_float(v[u][1]) + sa * sa * (1.0 / HW);
        s1 += sb;
        q1 += (double)__uint_as_float(v[u][3]) + sb * sb * (1.0 / HW);
      }
    }
    double* d = dred + (part * 32 + 2 * cp) * 2;
    d[0] = s0; d[1] = q0; d[2] = s1; d[3] = q1;
    __syncthreads();
    if (tid < 32) {
      double sum = 0.0, q = 0.0;
#pragma unroll
      for (int p = 0; p < 16; ++p) {
        sum += dred[(p * 32 + tid) * 2];
        q += dred[(p * 32 + tid) * 2 + 1];
      }
      const double n = (double)B * HW, m = sum / n;
      double v = (q - sum * sum / n) / n;
      if (v < 0.0) v = 0.0;
      const int k = cin - 32 + tid;
      a.mean[k] = (float)m;
      a.var[k] = (float)v;
      a.rstd[k] = (float)(1.0 / sqrt(v + (double)a.eps1));
    }
  }
}

// conv1 weights [128][cin] (k-contiguous rows) -> the fragment order the persistent kernel streams: see kloop().  Layer l
// covers the channels [0, K_old) in 128-channel chunks (the last one 32 / 64 / 96 wide), K_old = cin for l = 0 and cin - 32
// otherwise, then -- for l > 0 -- the 32-channel slice produced by layer l - 1.  One thread per 16-byte piece.
struct PackArgs {
  const bf16_t* w1[DB_MAXL];
  bf16_t* out[DB_MAXL];
  int C0, L;
};
__global__ __launch_bounds__(256) void pack_w1_kernel(PackArgs p) {
  const int l = blockIdx.y;
  const int cin = p.C0 + 32 * l;
  const int kold = l == 0 ? cin : cin - 32;
  const int npiece = 128 * cin / 8;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= npiece) return;
  const int o = q * 16;                                   // byte offset inside the packed layer
  const int cpos = o >> 8;                                // "channel position": chunk (c0, w) covers bytes [c0*256, (c0+w)*256)
  int c0, w;
  if (cpos < (kold & ~127)) {
    c0 = cpos & ~127;
    w = 128;
  } else if (cpos < kold) {
    c0 = kold & ~127;
    w = kold & 127;
  } else {
    c0 = kold;
    w = 32;
  }
  const int rel = o - c0 * 256, unit = rel >> 10, lane = (rel & 1023) >> 4;
  const int nk = w >> 4, wave = unit / nk, kk = unit % nk;
  const int n = wave * 32 + (lane & 31), h = lane >> 5;
  const int k = c0 + h * (w >> 1) + kk * 8;
  *reinterpret_cast<uint4*>(p.out[l] + (long long)q * 8) = *reinterpret_cast<const uint4*>(p.w1[l] + (long long)n * cin + k);
}

__global__ void zero_words_kernel(unsigned* p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0u;
}

constexpr size_t DB_LDS = 16384 /* s_mean, s_rstd, tab */ + 3 * 16384 + 16384 + 82 * 256 + 1024 + 1024 * 8 + 16;

// =====================================================================================================================
// The same dense block BACKWARD (7 x 7 maps) as ONE persistent launch: what DenseBlockFn.backward issues per layer as
// mcl_dense_conv3x3_bwd_fix (+ finalize + bn2_dz) and mcl_dense_bn1_dx_sums (+ finalize) -- five dependent launches, ~74 us per
// layer inside the step -- with the arithmetic of those kernels (csrc/dense_bwd.hip), rounding points included:
//
//   dy' = bf16(dy - (K1 + K2*xhat))            the layer's 32 channels of the gradient buffer, minus the mean terms the
//                                              previous pass (layer l+1) left pending for them        -> dyc (weight gradient)
//   da2 = conv2^T(dy') ; g2 = bf16(da2 * [bn2(z) > 0]) ; (sum g2, sum g2*zhat) over the BATCH  [seam A]
//   dz  = bf16(gamma2*rstd2*(g2 - mean g2 - zhat*mean(g2 zhat)))                               -> dz  (weight gradients)
//   da  = dz W1 ; g = da * [bn1(x) > 0] ; G[:, :cin] += bf16(gamma1*rstd1*g - (K1 + K2*xhat)[previous pass])
//   (sum g, sum g*xhat) per input channel over the BATCH  [seam B, two hops] -> dgamma1, dbeta1, K of THIS pass
//
// One workgroup owns one image; its 49 x C_total slice of the block's gradient buffer lives in LDS for the whole block
// (100 KB) and only the block-input channels are written back.  Seam A is the forward kernel's seam (128 records of
// (sum, sum) pairs per image, everybody merges all of them).  Seam B would be C_in pairs per image -- 1 MB to read per
// workgroup and layer -- so it is reduced in two hops: every workgroup publishes its C_in pairs; workgroup i merges channels
// [8i, 8i + 8) of all images (8 KB), adds dgamma1 / dbeta1 into the parameter gradients and publishes that slice's mean
// terms; everybody waits for the slices (the terms themselves are read where they are applied, with L1-bypassing loads).
// The two weight gradients stay what they are (side-stream kernels on dz / dyc), launched after this kernel.
struct DBBLayer {
  const float* g1;
  const float* b1;
  const bf16_t* w1t;    // conv1 weight packed for da = dz W1: [C_in/32][8 k-steps][64 lanes][8]
  const float* g2;
  const float* b2;
  const bf16_t* w2t;    // conv2 weight packed for the backward-data product: [4 waves][18 k-steps][64 lanes][8]
  const bf16_t* z;      // [B][49][128]
  const float* m2;
  const float* r2;
  bf16_t* dz;           // out [B][49][128]
  bf16_t* dyc;          // out [B][49][32]
  float* dg1;           // parameter gradients, accumulated into
  float* db1;
  float* dg2;
  float* db2;
};
struct DBBArgs {
  const bf16_t* buf;    // [B][49][Ct] the block's concat buffer (forward activations)
  bf16_t* gbuf;         // [B][49][Ct] gradient buffer: read whole, channels [0, C0) written back
  int Ct, B, C0, L;
  const float* mean;    // [Ct] statistics of the concat channels
  const float* rstd;
  u64* xa;              // seam A records [L][B][128]
  u64* xb;              // seam B hop-1 records [L][B][1024]
  u64* kacc;            // [L][1024] (K1, K2) of each pass
  unsigned* fa;         // flags [L][B]
  unsigned* fb1;        // [L][B]
  unsigned* fb2;        // [L][128] slices (8 channels each)
  int* err;
  unsigned max_spins;
  unsigned long long* dbg;
  DBBLayer ly[DB_MAXL];
};

__device__ __forceinline__ float bf2f_(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f2bf_(float f) { return (bf16_t)(pack2(f, 0.0f) & 0xFFFFu); }

#define DBB_STAMP(k)                                                                                     \
  do {                                                                                                   \
    if (a.dbg && tid == 0) a.dbg[((long long)img * a.L + l) * 8 + (k)] = wall_clock64();                 \
  } while (0)

__global__ __launch_bounds__(256, 1) void dense_block_bwd_kernel(DBBArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  bf16_t* Gt = reinterpret_cast<bf16_t*>(lds);                        // [49][1024] gradient rows of this image (row stride 2048 B)
  bf16_t* zt = reinterpret_cast<bf16_t*>(lds + 49 * 2048);            // [49][128] z, plain rows
  unsigned char* g2t = lds + 49 * 2048 + 49 * 256;                    // [64][256 B] g2, then dz in place (chunk ^ (row & 15))
  unsigned char* dyt = g2t + 64 * 256;                                // [82][64 B] zero-bordered dy' (chunk ^ ((row >> 2) & 3))
  unsigned char* xs = dyt + 82 * 64;                                  // [4 waves][49][64 B] x chunk of the wave's channel tile
  float* tdz = reinterpret_cast<float*>(xs + 4 * 49 * 64);            // [5][128]: sc2, c1, mu2, rs2, c2
  int* dead = reinterpret_cast<int*>(tdz + 5 * 128);
  double* dred = reinterpret_cast<double*>(xs);                       // seam A merge (xs is idle then): 4 x 128 x 2 doubles
  float* hred = reinterpret_cast<float*>(g2t);                        // hop-1 merge (the dz tile is in registers by then): [B][16]
  double* hsum = reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(dead) + 64);   // 16 doubles
  double* hpart = reinterpret_cast<double*>(zt);                      // [16][16] group sums (z is dead by then)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int img = blockIdx.x, B = a.B, Ct = a.Ct;
  const double Sd = (double)B * HW;
  const bf16_t* xb_ = a.buf + (long long)img * HW * Ct;
  bf16_t* gb_ = a.gbuf + (long long)img * HW * Ct;

  // ---- the image's gradient rows into LDS; zero borders of the dy' tile
  for (int q = tid; q < HW * (Ct >> 3); q += 256) {
    const int p = q / (Ct >> 3), c8 = q - p * (Ct >> 3);
    *reinterpret_cast<uint4*>(Gt + p * 1024 + c8 * 8) = *reinterpret_cast<const uint4*>(gb_ + (long long)p * Ct + c8 * 8);
  }
  for (int i = tid; i < 82 * 4; i += 256) reinterpret_cast<uint4*>(dyt)[i] = make_uint4(0u, 0u, 0u, 0u);
  // rows 49..63 of the dz tile (the second MFMA row tile's padding) stay zero for the whole kernel: their products are zeros,
  // so the epilogue of the da product needs no validity test on the sums (only its stores are predicated)
  for (int i = tid; i < 15 * 16; i += 256) reinterpret_cast<uint4*>(g2t + 49 * 256)[i] = make_uint4(0u, 0u, 0u, 0u);
  if (tid == 0) *dead = 0;
  int prow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = i * 32 + l31;
    prow[i] = p < HW ? (p / MAPW) * PW + (p % MAPW) : -1;
  }
  __syncthreads();

  for (int l = a.L - 1; l >= 0; --l) {
    const DBBLayer ly = a.ly[l];
    const int cin = a.C0 + 32 * l;
    const bool have_prev = l + 1 < a.L;                               // the pass of layer l+1 left mean terms pending
    const u64* kprev = a.kacc + (long long)(l + 1) * 1024;
    DBB_STAMP(0);
    // ------------------------------------------------------------ a. dy' into the padded tile (+ dyc), z into LDS
    if (tid < HW * 4) {
      const int p = tid >> 2, ch = tid & 3, c = cin + ch * 8;
      uint4 v = *reinterpret_cast<const uint4*>(Gt + p * 1024 + c);
      if (have_prev) {
        const uint4 xv = *reinterpret_cast<const uint4*>(xb_ + (long long)p * Ct + c);
        const __amdgpu_buffer_rsrc_t krs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<u64*>(kprev), 0, 8192u, 0x00020000);
        u32x4 kv[4];                                                  // (K1, K2) of 8 channels: L1-bypassing loads
#pragma unroll
        for (int u = 0; u < 4; ++u) kv[u] = __builtin_amdgcn_raw_buffer_load_b128(krs, (unsigned)(c * 8 + u * 16), 0, 16);
        const float4 m0 = *reinterpret_cast<const float4*>(a.mean + c), m1 = *reinterpret_cast<const float4*>(a.mean + c + 4);
        const float4 r0 = *reinterpret_cast<const float4*>(a.rstd + c), r1 = *reinterpret_cast<const float4*>(a.rstd + c + 4);
        const float mv[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
        const float rv[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
        const unsigned xw[4] = {xv.x, xv.y, xv.z, xv.w};
        unsigned gw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float o[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const float k1 = __uint_as_float(kv[u][2 * e]), k2 = __uint_as_float(kv[u][2 * e + 1]);
            const float ka = k2 * rv[2 * u + e], kb = fmaf(-ka, mv[2 * u + e], k1);
            const float xf = e ? bf_hi(xw[u]) : bf_lo(xw[u]);
            const float gf = e ? bf_hi(gw[u]) : bf_lo(gw[u]);
            o[e] = gf - fmaf(ka, xf, kb);
          }
          gw[u] = pack2(o[0], o[1]);
        }
        v = make_uint4(gw[0], gw[1], gw[2], gw[3]);
      }
      const int pr = (p / MAPW + 1) * PW + (p % MAPW) + 1;
      *reinterpret_cast<uint4*>(dyt + pr * 64 + ((ch ^ ((pr >> 2) & 3)) << 4)) = v;
      *reinterpret_cast<uint4*>(ly.dyc + ((long long)img * HW + p) * 32 + ch * 8) = v;
    }
    for (int q = tid; q < HW * 16; q += 256)
      *reinterpret_cast<uint4*>(zt + q * 8) = *reinterpret_cast<const uint4*>(ly.z + (long long)img * HW * 128 + q * 8);
    // this wave's weight fragments of the backward-data product
    bf16x8 breg[18];
#pragma unroll
    for (int i = 0; i < 18; ++i)
      breg[i] = *reinterpret_cast<const bf16x8*>(ly.w2t + ((long long)(wave * 18 + i) * 64 + lane) * 8);
    const int c2ch = 32 * wave + l31;                                 // this lane's bottleneck channel
    const float mu2 = ly.m2[c2ch], rs2 = ly.r2[c2ch];
    const float sc2 = ly.g2[c2ch] * rs2, sh2 = fmaf(-mu2, sc2, ly.b2[c2ch]);
    __syncthreads();
    // ------------------------------------------------------------ b. da2 = conv2^T(dy'): K = 9 taps x 32 channels = 18 k-steps
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
#pragma unroll
    for (int i = 0; i < 18; ++i) {
      const int tap = i >> 1, ky = tap / 3, kx = tap % 3;
      const int toff = (2 - ky) * PW + (2 - kx);
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const int row = prow[mb] >= 0 ? prow[mb] + toff : ZERO_ROW;
        const bf16x8 fa = *reinterpret_cast<const bf16x8*>(dyt + row * 64 + (((2 * (i & 1) + h) ^ ((row >> 2) & 3)) << 4));
        acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, breg[i], acc[mb], 0, 0, 0);
      }
    }
    // ------------------------------------------------------------ c. g2, its batch sums (seam A), dz
    {
      float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int px = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (px < HW) {
            const float zv = bf2f_(zt[px * 128 + c2ch]);
            const float gi = fmaf(zv, sc2, sh2) > 0.0f ? acc[mb][r] : 0.0f;
            const bf16_t gb = f2bf_(gi);
            const float gr = bf2f_(gb);                                // the sums are those of the stored (rounded) g2
            s1 += gr;
            s2 = fmaf(gr, (zv - mu2) * rs2, s2);
            reinterpret_cast<bf16_t*>(g2t + px * 256 + (((c2ch >> 3) ^ (px & 15)) << 4))[c2ch & 7] = gb;
          }
        }
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (h == 0) store_rec(a.xa + ((long long)l * B + img) * 128 + c2ch, s1, s2);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(a.fa + l * B + img, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    DBB_STAMP(1);
    seam_wait(a.fa + l * B, B, tid, dead, a.err, a.max_spins);
    DBB_STAMP(2);
    {
      const int cp = tid & 63, part = tid >> 6;                      // 64 channel pairs x 4 image groups
      const int per = (B + 3) >> 2, i0 = part * per, i1 = min(B, i0 + per);
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          reinterpret_cast<void*>(a.xa + (long long)l * B * 128), 0, (unsigned)(B * 1024), 0x00020000);
      double s0 = 0.0, q0 = 0.0, s1 = 0.0, q1 = 0.0;
      for (int i = i0; i < i1; i += 16) {
        u32x4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u)
          v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, i + u < i1 ? (unsigned)((i + u) * 1024 + cp * 16) : 0xFFFFF000u, 0, 16);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          s0 += (double)__uint_as_float(v[u][0]);
          q0 += (double)__uint_as_float(v[u][1]);
          s1 += (double)__uint_as_float(v[u][2]);
          q1 += (double)__uint_as_float(v[u][3]);
        }
      }
      double* d = dred + (part * 128 + 2 * cp) * 2;
      d[0] = s0; d[1] = q0; d[2] = s1; d[3] = q1;
    }
    __syncthreads();
    if (tid < 128) {
      double sa = 0.0, sb = 0.0;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        sa += dred[(p * 128 + tid) * 2];
        sb += dred[(p * 128 + tid) * 2 + 1];
      }
      const float mu = ly.m2[tid], rs = ly.r2[tid];
      tdz[tid] = ly.g2[tid] * rs;
      tdz[128 + tid] = (float)(sa / Sd);
      tdz[256 + tid] = mu;
      tdz[384 + tid] = rs;
      tdz[512 + tid] = (float)(sb / Sd);
      if (img == 0) {
        ly.db2[tid] += (float)sa;
        ly.dg2[tid] += (float)sb;
      }
    }
    __syncthreads();
    // dz = gamma2*rstd2*(g2 - c1 - (z - mu)*rstd2*c2), in place over g2 (the A operand of the next product) and to global memory
    for (int q = tid; q < HW * 16; q += 256) {
      const int p = q >> 4, cc = q & 15;
      unsigned char* gp = g2t + p * 256 + ((cc ^ (p & 15)) << 4);
      const uint4 gv = *reinterpret_cast<const uint4*>(gp);
      const uint4 zv = *reinterpret_cast<const uint4*>(zt + p * 128 + cc * 8);
      const unsigned gw[4] = {gv.x, gv.y, gv.z, gv.w}, zw[4] = {zv.x, zv.y, zv.z, zv.w};
      float tv[5][8];
#pragma unroll
      for (int t = 0; t < 5; ++t) {
        const float4 lo = *reinterpret_cast<const float4*>(tdz + t * 128 + cc * 8), hi = *reinterpret_cast<const float4*>(tdz + t * 128 + cc * 8 + 4);
        tv[t][0] = lo.x; tv[t][1] = lo.y; tv[t][2] = lo.z; tv[t][3] = lo.w;
        tv[t][4] = hi.x; tv[t][5] = hi.y; tv[t][6] = hi.z; tv[t][7] = hi.w;
      }
      unsigned o[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e0 = 2 * u, e1 = 2 * u + 1;
        const float d_lo = tv[0][e0] * (bf_lo(gw[u]) - tv[1][e0] - (bf_lo(zw[u]) - tv[2][e0]) * tv[3][e0] * tv[4][e0]);
        const float d_hi = tv[0][e1] * (bf_hi(gw[u]) - tv[1][e1] - (bf_hi(zw[u]) - tv[2][e1]) * tv[3][e1] * tv[4][e1]);
        o[u] = pack2(d_lo, d_hi);
      }
      const uint4 ov = make_uint4(o[0], o[1], o[2], o[3]);
      *reinterpret_cast<uint4*>(gp) = ov;
      *reinterpret_cast<uint4*>(ly.dz + ((long long)img * HW + p) * 128 + cc * 8) = ov;
    }
    __syncthreads();
    DBB_STAMP(3);
    // ------------------------------------------------------------ d/e. da = dz W1 per 32-channel tile, mask, sums, G update
    {
      bf16x8 fdz[8][2];                                               // the whole dz tile as A fragments (K = 128: 8 k-steps)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int r = l31 + 32 * mt;
          fdz[ks][mt] = *reinterpret_cast<const bf16x8*>(g2t + r * 256 + (((2 * ks + h) ^ (r & 15)) << 4));
        }
      unsigned char* xw_ = xs + wave * (49 * 64);
      const int ntile = cin >> 5;
      const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(xb_), 0, (unsigned)(HW * Ct * 2), 0x00020000);
      const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(ly.w1t), 0, (unsigned)(cin * 256), 0x00020000);
      // piece q = lane + 64*i of the wave's x chunk: pixel q >> 2, 16-byte piece q & 3
      unsigned xo[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int q = lane + 64 * i;
        xo[i] = q < HW * 4 ? (unsigned)(((q >> 2) * Ct + (q & 3) * 8) * 2) : 0xFFFFF000u;
      }
      u32x4 xr[4], wr[8];
      float cmu = 0.0f, crs = 0.0f, cg = 0.0f, cb = 0.0f;
      u64 ck = 0;
      const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u64*>(kprev), 0, have_prev ? 8192u : 0u, 0x00020000);
      auto load_tile = [&](int nt) {
        const bool live = nt < ntile;
        const int nn = live ? nt * 32 + l31 : 0;                      // per-channel constants of this lane for that tile
        cmu = a.mean[nn];
        crs = a.rstd[nn];
        cg = ly.g1[nn];
        cb = ly.b1[nn];
        {
          const u32x2 kk = __builtin_amdgcn_raw_buffer_load_b64(krs, (unsigned)nn * 8u, 0, 16);   // (no previous pass: size 0 -> zeros)
          ck = ((u64)kk[1] << 32) | (u64)kk[0];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
          xr[i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, live ? xo[i] + (unsigned)nt * 64u : 0xFFFFF000u, 0, 0);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
          wr[ks] = __builtin_amdgcn_raw_buffer_load_b128(wrs, live ? (unsigned)((nt * 8 + ks) * 1024 + lane * 16) : 0xFFFFF000u, 0, 0);
      };
      load_tile(wave);
      for (int nt = wave; nt < ntile; nt += 4) {
        const int n = nt * 32 + l31;
        // the tile's operands leave the prefetch registers: x chunk -> the wave's private LDS rows, W fragments -> wf
        bf16x8 wf[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) wf[ks] = __builtin_bit_cast(bf16x8, wr[ks]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int q = lane + 64 * i;
          if (q < HW * 4) *reinterpret_cast<uint4*>(xw_ + q * 16) = make_uint4(xr[i][0], xr[i][1], xr[i][2], xr[i][3]);
        }
        // per-channel constants of this lane (prefetched with the tile)
        const float mu = cmu, rs = crs;
        const float sc = cg * rs, sh = fmaf(-mu, sc, cb);
        const float c1 = __uint_as_float((unsigned)ck), c2 = __uint_as_float((unsigned)(ck >> 32));
        const float ka = -c2 * rs, kb = fmaf(-ka, mu, -c1);
        load_tile(nt + 4);                                             // next tile's operands in flight during this one
        f32x16 da[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) da[i][r] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          da[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fdz[ks][0], wf[ks], da[0], 0, 0, 0);
          da[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fdz[ks][1], wf[ks], da[1], 0, 0, 0);
        }
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int px = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const bool valid = mt == 0 || px < HW;
            const int pc = valid ? px : HW - 1;                       // (a padding row reads a real row: finite values, da = 0)
            const float xv = bf2f_(reinterpret_cast<const bf16_t*>(xw_ + pc * 64)[l31]);
            const float gi = fmaf(xv, sc, sh) > 0.0f ? da[mt][r] : 0.0f;
            s1 += gi;
            s2 = fmaf(gi, xv, s2);
            const float delta = bf2f_(f2bf_(fmaf(sc, gi, fmaf(ka, xv, kb))));
            bf16_t* gp = Gt + pc * 1024 + n;
            const bf16_t gnew = f2bf_(bf2f_(*gp) + delta);
            if (valid) *gp = gnew;
          }
        s2 = rs * fmaf(-mu, s1, s2);
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if (h == 0) store_rec(a.xb + ((long long)l * B + img) * 1024 + n, s1, s2);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(a.fb1 + l * B + img, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    DBB_STAMP(4);
    // ------------------------------------------------------------ seam B, hop 1: slice s = channels [8s, 8s+8), owner s % B
    const int nslice = cin >> 3;
    if (img < nslice) seam_wait(a.fb1 + l * B, B, tid, dead, a.err, a.max_spins);
    DBB_STAMP(5);
    for (int s = img; s < nslice; s += B) {
      __syncthreads();
      float pdb = 0.0f, pdg = 0.0f, pg1 = 0.0f, prs = 0.0f;
      if (tid < 8) {                                                   // this slice's parameter-gradient words, ahead of the merge
        const int c = s * 8 + tid;
        pdb = ly.db1[c];
        pdg = ly.dg1[c];
        pg1 = ly.g1[c];
        prs = a.rstd[c];
      }
      {
        // (ONE descriptor for the workgroup: a per-thread base would make every load a 64-trip scalarisation loop)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<void*>(a.xb + (long long)l * B * 1024), 0, (unsigned)(B * 8192), 0x00020000);
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, tid < B ? (unsigned)(tid * 8192 + s * 64 + u * 16) : 0xFFFFF000u, 0, 16);
        if (tid < B) {
#pragma unroll
          for (int u = 0; u < 4; ++u) *reinterpret_cast<uint4*>(hred + tid * 16 + u * 4) = make_uint4(v[u][0], v[u][1], v[u][2], v[u][3]);
        }
      }
      __syncthreads();
      {
        // fixed-order two-level merge over the images: 16 groups of ceil(B/16) images, then the 16 group sums
        const int v = tid & 15, grp = tid >> 4, per = (B + 15) >> 4;
        const int j0 = grp * per, j1 = min(B, j0 + per);
        double sum = 0.0;
        for (int j = j0; j < j1; ++j) sum += (double)hred[j * 16 + v];
        hpart[grp * 16 + v] = sum;
      }
      __syncthreads();
      if (tid < 16) {
        double sum = 0.0;
#pragma unroll
        for (int g = 0; g < 16; ++g) sum += hpart[g * 16 + tid];
        hsum[tid] = sum;
      }
      __syncthreads();
      if (tid < 8) {
        const int c = s * 8 + tid;
        const double sa = hsum[2 * tid], sb = hsum[2 * tid + 1];
        ly.db1[c] = pdb + (float)sa;
        ly.dg1[c] = pdg + (float)sb;
        const double scv = (double)pg1 * (double)prs;
        store_rec(a.kacc + (long long)l * 1024 + c, (float)(scv * sa / Sd), (float)(scv * sb / Sd));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
      if (tid == 0) __hip_atomic_store(a.fb2 + l * 128 + s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ------------------------------------------------------------ hop 2: every slice of this pass is published
    seam_wait(a.fb2 + l * 128, nslice, tid, dead, a.err, a.max_spins);
    DBB_STAMP(6);
  }

  // ---- the block input's channels: the mean terms of layer 0's pass, then back to the gradient buffer
  {
    const u64* k0 = a.kacc;
    const int C0 = a.C0;
    for (int q = tid; q < HW * (C0 >> 3); q += 256) {
      const int p = q / (C0 >> 3), c = (q - p * (C0 >> 3)) * 8;
      const uint4 xv = *reinterpret_cast<const uint4*>(xb_ + (long long)p * Ct + c);
      const uint4 gv = *reinterpret_cast<const uint4*>(Gt + p * 1024 + c);
      const unsigned xw[4] = {xv.x, xv.y, xv.z, xv.w};
      unsigned gw[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float o[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int cf = c + 2 * u + e;
          const u64 kk = load_rec(k0 + cf);
          const float k1 = __uint_as_float((unsigned)kk), k2 = __uint_as_float((unsigned)(kk >> 32));
          const float ka = k2 * a.rstd[cf], kb = fmaf(-ka, a.mean[cf], k1);
          const float xf = e ? bf_hi(xw[u]) : bf_lo(xw[u]);
          const float gf = e ? bf_hi(gw[u]) : bf_lo(gw[u]);
          o[e] = gf - fmaf(ka, xf, kb);
        }
        gw[u] = pack2(o[0], o[1]);
      }
      *reinterpret_cast<uint4*>(gb_ + (long long)p * Ct + c) = make_uint4(gw[0], gw[1], gw[2], gw[3]);
    }
  }
}

// conv1 / conv2 weights -> the fragment orders of the backward kernel (see DBBLayer).  One thread per 16-byte piece.
struct PackBArgs {
  const bf16_t* w1[DB_MAXL];
  const bf16_t* w2[DB_MAXL];
  bf16_t* w1t[DB_MAXL];
  bf16_t* w2t[DB_MAXL];
  int C0, L;
};
__global__ __launch_bounds__(256) void pack_bwd_kernel(PackBArgs p) {
  const int l = blockIdx.y;
  const int cin = p.C0 + 32 * l;
  const int n1 = (cin >> 5) * 8 * 64, n2 = 4 * 18 * 64;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q < n1) {
    const int lane = q & 63, ks = (q >> 6) & 7, nt = q >> 9;
    const int n = nt * 32 + (lane & 31), k0 = 16 * ks + 8 * (lane >> 5);
    unsigned short v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = p.w1[l][(long long)(k0 + j) * cin + n];
    *reinterpret_cast<uint4*>(p.w1t[l] + (long long)q * 8) =
        make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
  } else if (q < n1 + n2) {
    const int r = q - n1, lane = r & 63, i = (r >> 6) % 18, wave = (r >> 6) / 18;
    const int tap = i >> 1, co0 = 16 * (i & 1) + 8 * (lane >> 5), ci = 32 * wave + (lane & 31);
    unsigned short v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = p.w2[l][((long long)(co0 + j) * 9 + tap) * 128 + ci];
    *reinterpret_cast<uint4*>(p.w2t[l] + (long long)r * 8) =
        make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
  }
}

constexpr size_t DBB_LDS = 49 * 2048 + 49 * 256 + 64 * 256 + 82 * 64 + 4 * 49 * 64 + 5 * 128 * 4 + 64 + 128;

constexpr unsigned DB_DEFAULT_SPINS = 1u << 19;   // x (B flag loads + s_sleep 8) ~ 0.5 s

// Every workgroup of a persistent launch must be resident at once (the seams are all-to-all): one image per CU at most.  Asked of
// the runtime on every call -- a host-side query, no device work; no process-global cache (the library keeps no state).
template <typename K>
static bool db_fits(K kernel, size_t lds, int B) {
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kernel), 256, lds) != hipSuccess) return false;
  return per_cu >= 1 && B <= mcl_cu_count();
}

}  // namespace

extern "C" int mcl_dense_block_pack_w1(const void* const* w1_ptrs, void* const* out_ptrs, int32_t L, int32_t C0,
                                       mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!w1_ptrs || !out_ptrs || L <= 0 || L > DB_MAXL || C0 <= 0 || (C0 % 32)) return MCL_EINVAL;
  PackArgs p;
  p.C0 = C0;
  p.L = L;
  for (int l = 0; l < L; ++l) {
    if (!w1_ptrs[l] || !out_ptrs[l] || (reinterpret_cast<uintptr_t>(w1_ptrs[l]) & 15u) ||
        (reinterpret_cast<uintptr_t>(out_ptrs[l]) & 15u))
      return MCL_EINVAL;
    p.w1[l] = (const bf16_t*)w1_ptrs[l];
    p.out[l] = (bf16_t*)out_ptrs[l];
  }
  const int cmax = C0 + 32 * (L - 1);
  hipLaunchKernelGGL(pack_w1_kernel, dim3((128 * cmax / 8 + 255) / 256, L), dim3(256), 0, mcl_stream(stream), p);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int64_t mcl_dense_block_fwd_workspace_bytes(int32_t B, int32_t L) {
  if (B <= 0 || L <= 0 || L > DB_MAXL) return -1;
  return (int64_t)2 * L * B * 4 + 256 + (int64_t)2 * L * B * 128 * 8;
}

extern "C" int mcl_dense_block_fwd(void* buf, int32_t B, int32_t H, int32_t W, int32_t Ct, int32_t C0, int32_t L,
                                   const void* const* layer_ptrs, float eps1, float eps2, float* mean, float* var, float* rstd,
                                   void* workspace, int32_t* err_flag, uint32_t max_spins, void* stamps, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!buf || !layer_ptrs || !mean || !var || !rstd || !workspace || !err_flag || B <= 0 || L <= 0) return MCL_EINVAL;
  if (H != MAPW || W != MAPW || L > DB_MAXL || (C0 % 32) || C0 <= 0 || Ct != C0 + 32 * L || Ct > 1024 ||
      (reinterpret_cast<uintptr_t>(buf) & 15u) || (reinterpret_cast<uintptr_t>(workspace) & 255u))
    return MCL_EUNSUPPORTED;
  hipStream_t st = mcl_stream(stream);
  static mcl_device_once attr_once;
  if (auto attr_guard = attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dense_block_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)DB_LDS);
  }
  if (!db_fits(dense_block_fwd_kernel, DB_LDS, B)) return MCL_EUNSUPPORTED;
  DBArgs a;
  a.buf = reinterpret_cast<bf16_t*>(buf);
  a.Ct = Ct;
  a.B = B;
  a.C0 = C0;
  a.L = L;
  a.eps1 = eps1;
  a.eps2 = eps2;
  a.mean = mean;
  a.var = var;
  a.rstd = rstd;
  a.flags = reinterpret_cast<unsigned*>(workspace);
  const size_t flag_bytes = ((size_t)2 * L * B * 4 + 255) & ~(size_t)255;
  a.xch = reinterpret_cast<u64*>(reinterpret_cast<unsigned char*>(workspace) + flag_bytes);
  a.err = err_flag;
  a.max_spins = max_spins ? max_spins : DB_DEFAULT_SPINS;
  a.dbg = reinterpret_cast<unsigned long long*>(stamps);
  for (int l = 0; l < L; ++l) {
    const void* const* p = layer_ptrs + 10 * l;
    for (int k = 0; k < 10; ++k)
      if (!p[k]) return MCL_EINVAL;
    DBLayer& y = a.ly[l];
    y.g1 = (const float*)p[0];
    y.b1 = (const float*)p[1];
    y.w1 = (const bf16_t*)p[2];
    y.g2 = (const float*)p[3];
    y.b2 = (const float*)p[4];
    y.w2 = (const bf16_t*)p[5];
    y.z = (bf16_t*)p[6];
    y.m2 = (float*)p[7];
    y.v2 = (float*)p[8];
    y.r2 = (float*)p[9];
    if ((reinterpret_cast<uintptr_t>(y.w1) & 15u) || (reinterpret_cast<uintptr_t>(y.w2) & 15u) ||
        (reinterpret_cast<uintptr_t>(y.z) & 15u))
      return MCL_EUNSUPPORTED;
  }
  const int nflag = 2 * L * B;
  hipLaunchKernelGGL(zero_words_kernel, dim3((nflag + 255) / 256), dim3(256), 0, st, a.flags, nflag);
  hipLaunchKernelGGL(dense_block_fwd_kernel, dim3(B), dim3(256), DB_LDS, st, a);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_dense_block_pack_bwd(const void* const* w1_ptrs, const void* const* w2_ptrs, void* const* w1t_ptrs,
                                        void* const* w2t_ptrs, int32_t L, int32_t C0, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!w1_ptrs || !w2_ptrs || !w1t_ptrs || !w2t_ptrs || L <= 0 || L > DB_MAXL || C0 <= 0 || (C0 % 32)) return MCL_EINVAL;
  PackBArgs p;
  p.C0 = C0;
  p.L = L;
  for (int l = 0; l < L; ++l) {
    if (!w1_ptrs[l] || !w2_ptrs[l] || !w1t_ptrs[l] || !w2t_ptrs[l] || (reinterpret_cast<uintptr_t>(w1t_ptrs[l]) & 15u) ||
        (reinterpret_cast<uintptr_t>(w2t_ptrs[l]) & 15u))
      return MCL_EINVAL;
    p.w1[l] = (const bf16_t*)w1_ptrs[l];
    p.w2[l] = (const bf16_t*)w2_ptrs[l];
    p.w1t[l] = (bf16_t*)w1t_ptrs[l];
    p.w2t[l] = (bf16_t*)w2t_ptrs[l];
  }
  const int cmax = C0 + 32 * (L - 1);
  const int npiece = (cmax >> 5) * 8 * 64 + 4 * 18 * 64;
  hipLaunchKernelGGL(pack_bwd_kernel, dim3((npiece + 255) / 256, L), dim3(256), 0, mcl_stream(stream), p);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int64_t mcl_dense_block_bwd_workspace_bytes(int32_t B, int32_t L) {
  if (B <= 0 || L <= 0 || L > DB_MAXL) return -1;
  const int64_t flags = ((int64_t)(2 * L * B + L * 128) * 4 + 255) & ~(int64_t)255;
  return flags + (int64_t)L * B * 128 * 8 + (int64_t)L * B * 1024 * 8 + (int64_t)L * 1024 * 8;
}

extern "C" int mcl_dense_block_bwd(const void* buf, void* gbuf, int32_t B, int32_t H, int32_t W, int32_t Ct, int32_t C0,
                                   int32_t L, const void* const* layer_ptrs, const float* mean, const float* rstd,
                                   void* workspace, int32_t* err_flag, uint32_t max_spins, void* stamps, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!buf || !gbuf || !layer_ptrs || !mean || !rstd || !workspace || !err_flag || B <= 0 || L <= 0) return MCL_EINVAL;
  if (H != MAPW || W != MAPW || L > DB_MAXL || (C0 % 32) || C0 <= 0 || Ct != C0 + 32 * L || Ct > 1024 ||
      (reinterpret_cast<uintptr_t>(buf) & 15u) || (reinterpret_cast<uintptr_t>(gbuf) & 15u) ||
      (reinterpret_cast<uintptr_t>(workspace) & 255u))
    return MCL_EUNSUPPORTED;
  hipStream_t st = mcl_stream(stream);
  static mcl_device_once attr_once;
  if (auto attr_guard = attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dense_block_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)DBB_LDS);
  }
  if (!db_fits(dense_block_bwd_kernel, DBB_LDS, B)) return MCL_EUNSUPPORTED;
  DBBArgs a;
  a.buf = (const bf16_t*)buf;
  a.gbuf = (bf16_t*)gbuf;
  a.Ct = Ct;
  a.B = B;
  a.C0 = C0;
  a.L = L;
  a.mean = mean;
  a.rstd = rstd;
  unsigned char* w = reinterpret_cast<unsigned char*>(workspace);
  const int nflag = 2 * L * B + L * 128;
  a.fa = reinterpret_cast<unsigned*>(w);
  a.fb1 = a.fa + L * B;
  a.fb2 = a.fb1 + L * B;
  w += ((size_t)nflag * 4 + 255) & ~(size_t)255;
  a.xa = reinterpret_cast<u64*>(w);
  w += (size_t)L * B * 128 * 8;
  a.xb = reinterpret_cast<u64*>(w);
  w += (size_t)L * B * 1024 * 8;
  a.kacc = reinterpret_cast<u64*>(w);
  a.err = err_flag;
  a.max_spins = max_spins ? max_spins : DB_DEFAULT_SPINS;
  a.dbg = reinterpret_cast<unsigned long long*>(stamps);
  for (int l = 0; l < L; ++l) {
    const void* const* p = layer_ptrs + 15 * l;
    for (int k = 0; k < 15; ++k)
      if (!p[k]) return MCL_EINVAL;
    DBBLayer& y = a.ly[l];
    y.g1 = (const float*)p[0];
    y.b1 = (const float*)p[1];
    y.w1t = (const bf16_t*)p[2];
    y.g2 = (const float*)p[3];
    y.b2 = (const float*)p[4];
    y.w2t = (const bf16_t*)p[5];
    y.z = (const bf16_t*)p[6];
    y.m2 = (const float*)p[7];
    y.r2 = (const float*)p[8];
    y.dz = (bf16_t*)p[9];
    y.dyc = (bf16_t*)p[10];
    y.dg1 = (float*)p[11];
    y.db1 = (float*)p[12];
    y.dg2 = (float*)p[13];
    y.db2 = (float*)p[14];
    if ((reinterpret_cast<uintptr_t>(y.w1t) & 15u) || (reinterpret_cast<uintptr_t>(y.w2t) & 15u) ||
        (reinterpret_cast<uintptr_t>(y.z) & 15u) || (reinterpret_cast<uintptr_t>(y.dz) & 15u) ||
        (reinterpret_cast<uintptr_t>(y.dyc) & 15u))
      return MCL_EUNSUPPORTED;
  }
  hipLaunchKernelGGL(zero_words_kernel, dim3((nflag + 255) / 256), dim3(256), 0, st, a.fa, nflag);
  hipLaunchKernelGGL(dense_block_bwd_kernel, dim3(B), dim3(256), DBB_LDS, st, a);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}
