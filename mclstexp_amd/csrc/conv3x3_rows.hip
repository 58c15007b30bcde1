// DenseNet growth 3x3 convolution, "row-walking" form for the large maps (image width 17..150: the 56 x 56 and 28 x 28
// dense blocks of torchvision's DenseNet-121 that /root/reference/model.py:75-76 wraps):
//
//     y[jo][x][co] = sum_{ky,kx,ci} a2[jo + ky - 1][x + kx - 1][ci] * W2[co][ky][kx][ci],  a2 = relu(bn2(z))  (zero outside)
//
// Why a second form next to the flat-pixel-tile kernel of dense_conv.hip: with N = 32 output channels an MFMA tile has no
// operand reuse, so the flat kernel pays one fresh 1 KB LDS operand per MFMA, stages a (2W+2)-pixel halo per 128-pixel
// tile (1.9x at W = 56), masks every tap per lane and reduces four K-split partials through 64 KB of LDS -- all
// synchronously (73.8 us per 56 x 56 layer inside the step; this form: 43 us).  Here the contraction is re-associated by
// kernel ROW:
//
//   * a wave owns a 32-column strip of one image and walks DOWN its rows.  For input row j it forms the three partial
//     products  W2[:, ky] (*) a2[j]  (ky = 0, 1, 2) from ONE staged row: every pixel fragment read from LDS feeds three
//     MFMAs (the three kernel rows), which land in three accumulators that belong to output rows j+1, j, j-1.  When input
//     row j is done, output row j-1 is complete and leaves through the epilogue; the accumulators rotate.  No vertical
//     halo is ever staged, no K split, no cross-wave reduction;
//   * the staged row carries explicit zero pixels at x = -1 and x = W (and beyond the strip's last valid column), and rows
//     outside the image are simply not visited (wave-uniform dispatch over the valid kernel rows), so the MFMA loop is
//     mask-free;
//   * waves never synchronise with each other after the prologue: each owns a private 34-pixel x 256 B LDS row slab
//     (XOR-swizzled, conflict-free ds_read_b128 / ds_write_b128).  HBM -> registers through buffer loads whose descriptor
//     covers exactly one image row (out-of-image pixels come back as zeros, no branch), TWO rows ahead: while row j is
//     multiplied, row j+1 is transformed (BatchNorm + ReLU, ~300 VALU instructions) in the shadow of the MFMAs, pinned
//     between them by scheduling barriers, and row j+2 is in flight -- 18 KB per wave outstanding;
//   * the whole weight (72 KB) sits in LDS once per workgroup in its HBM order, rows padded by 16 B (conflict-free
//     fragment reads, coalesced fill);
//   * MFMA roles are swapped (A = weights, B = pixels) so that a lane ends up with 16 output channels of ONE pixel: the
//     bf16 row leaves as two 16-byte buffer stores per lane after a v_permlane32_swap (cdna guide T21);
//   * the batch statistics the next layers' norm1 need are accumulated ON THE MATRIX CORES (RowStats below): the emitted
//     row is transposed through a 2 KB LDS scratch with ds_read_b64_tr_b16 and multiplied with itself / with ones.
//
// Work unit = (image, row chunk, strip); chunks are sized so that the units fill the chip's 2048 wave slots once.
// Measured anatomy at S = 401 408 (B = 128, 56 x 56), kernel alone 43 us: the 504 MFMAs per unit are 33.8 GFLOP incl. the
// 32/28 strip padding = 21 us at the ~1.6 PF/s this chip sustains on bf16 MFMA streams; a build with the
// MFMAs removed runs at the HBM rate (25 us: z is read 1.29x, rows j0-1 and j1 of neighbouring chunks); the pure MFMA
// loop with everything else removed takes 28 us.  What is left is the overlap of the two inside a wave and between the
// two waves of a SIMD (LDS and registers allow no third wave).
#include "common.h"
#include <stdlib.h>

namespace {

typedef unsigned short bf16_t;
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

constexpr int CI = 128, CO = 32;
constexpr int NWAVE = 8;                        // waves per workgroup (two per SIMD), one workgroup per CU
constexpr int WROW = 2304 + 16;                 // LDS bytes per output channel's 1152 weights (+16: conflict-free fragment reads)
constexpr int WF_BYTES = CO * WROW;             // the whole (32, 3, 3, 128) bf16 weight, in its HBM order, rows padded
constexpr int SLAB_ROWS = 34;                   // 32 strip pixels + one halo pixel each side
constexpr int SLAB_BYTES = SLAB_ROWS * 256;
constexpr int FRAG_DEPTH = 1;                   // fragment sets requested ahead of the MFMAs (2 and 3 measured no faster)

__device__ __forceinline__ unsigned pack2(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xFFFF0000u); }

__device__ __forceinline__ uint4 bn_relu_chunk(uint4 v, const float (&sc)[8], const float (&sh)[8]) {
  unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float lo = fmaxf(fmaf(bf_lo(w[i]), sc[2 * i], sh[2 * i]), 0.0f);
    const float hi = fmaxf(fmaf(bf_hi(w[i]), sc[2 * i + 1], sh[2 * i + 1]), 0.0f);
    w[i] = pack2(lo, hi);
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

// a = relu(x*sc + sh) on one dword (two bf16 channels)
__device__ __forceinline__ unsigned bn_relu_pair(unsigned w, float sc0, float sc1, float sh0, float sh1) {
  return pack2(fmaxf(fmaf(bf_lo(w), sc0, sh0), 0.0f), fmaxf(fmaf(bf_hi(w), sc1, sh1), 0.0f));
}

// One staged input row against the kernel rows selected at compile time.  wf = this lane's weight row (+ 16 h): fragment
// (ky, kx, kk) lies at byte (ky*3 + kx)*256 + kk*32 of it; pixel fragment (kx, kk) of lane (l31, h) = slab row l31 + kx, chunk
// (2 kk + h) ^ (row & 15) (pbase[0] carries the lane id).
// Software-pipelined by hand, D (kx, kk) steps ahead: the fragments of step i+D are requested before the MFMAs of step i
// issue, and scheduling barriers per step keep the compiler from hoisting all 96 fragment reads to the top (384 VGPRs).
// The BatchNorm+ReLU transform of the NEXT input row (raw bf16 in v[], requested from HBM before this call) is spread over
// steps 12..23, three dwords per step: its ~300 VALU instructions issue in the shadow of the MFMAs instead of after them.
template <bool V0, bool V1, bool V2>
__device__ __forceinline__ void row_mfma(const unsigned char* __restrict__ wf, const unsigned char* __restrict__ slab,
                                         const int (&pbase)[3], f32x16& aN, f32x16& aC, f32x16& aP, u32x4 (&v)[9],
                                         const float4* __restrict__ coef4 /* this lane's 4 x (sc0, sc1, sh0, sh1) */) {
  constexpr int D = FRAG_DEPTH, NB = D + 1;    // fragment sets in flight ahead of the MFMAs
  bf16x8 p[NB], w0[NB], w1[NB], w2[NB];
  // the 24 swizzled fragment addresses are recomputed per row (one v_xor each): left to itself the compiler hoists them
  // out of the row loop as 24 loop-invariant VGPRs and spills them
  int pb[3];
  {
    int ln = pbase[0];                          // = lane id, made opaque: everything derived from it is recomputed here
    asm volatile("" : "+v"(ln));
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int i = (ln & 31) + kx;
      pb[kx] = i * 256 + ((((i & 15) ^ (ln >> 5))) << 4);
    }
  }
  auto fetch = [&](int it, int buf) {
    const int kx = it >> 3, kk = it & 7;
    p[buf] = *reinterpret_cast<const bf16x8*>(slab + (pb[kx] ^ (kk << 5)));
    if (V0) w0[buf] = *reinterpret_cast<const bf16x8*>(wf + (0 * 3 + kx) * 256 + kk * 32);
    if (V1) w1[buf] = *reinterpret_cast<const bf16x8*>(wf + (1 * 3 + kx) * 256 + kk * 32);
    if (V2) w2[buf] = *reinterpret_cast<const bf16x8*>(wf + (2 * 3 + kx) * 256 + kk * 32);

  };
#pragma unroll
  for (int it = 0; it < D; ++it) fetch(it, it % NB);
#pragma unroll
  for (int it = 0; it < 24; ++it) {
    const int cur = it % NB;
    // an in-order wave hides ~7 single-issue instructions behind each 32-cycle MFMA: the step's fragment reads and the three
    // transform dwords are placed BETWEEN its MFMAs (pinned by scheduling barriers), not after them
    const bool tr = it >= 12;           // dword k of the 9 chunks in steps 12 + 3k .. 14 + 3k: one coefficient quad live at a time
    const int k = tr ? (it - 12) / 3 : 0, t0 = tr ? 3 * ((it - 12) % 3) : 0;
    float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tr) c = coef4[16 * k];
    const int nx = it + D, nb = nx % NB, nkx = nx >> 3, nkk = nx & 7;
    __builtin_amdgcn_sched_barrier(0);
    if (V0) aN = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0[cur], p[cur], aN, 0, 0, 0);
    if (nx < 24) {
      p[nb] = *reinterpret_cast<const bf16x8*>(slab + (pb[nkx] ^ (nkk << 5)));
      if (V0) w0[nb] = *reinterpret_cast<const bf16x8*>(wf + (0 * 3 + nkx) * 256 + nkk * 32);
    }
    if (tr) v[t0][k] = bn_relu_pair(v[t0][k], c.x, c.y, c.z, c.w);
    __builtin_amdgcn_sched_barrier(0);
    if (V1) aC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1[cur], p[cur], aC, 0, 0, 0);
    if (nx < 24) {
      if (V1) w1[nb] = *reinterpret_cast<const bf16x8*>(wf + (1 * 3 + nkx) * 256 + nkk * 32);
    }
    if (tr) v[t0 + 1][k] = bn_relu_pair(v[t0 + 1][k], c.x, c.y, c.z, c.w);
    __builtin_amdgcn_sched_barrier(0);
    if (V2) aP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2[cur], p[cur], aP, 0, 0, 0);
    if (nx < 24) {
      if (V2) w2[nb] = *reinterpret_cast<const bf16x8*>(wf + (2 * 3 + nkx) * 256 + nkk * 32);
    }
    if (tr) v[t0 + 2][k] = bn_relu_pair(v[t0 + 2][k], c.x, c.y, c.z, c.w);
    __builtin_amdgcn_sched_barrier(0);
  }
}

__device__ __forceinline__ void row_mfma_dispatch(int m, const unsigned char* wf, const unsigned char* slab,
                                                  const int (&pbase)[3], f32x16& aN, f32x16& aC, f32x16& aP, u32x4 (&v)[9],
                                                  const float4* coef4) {
  switch (m) {                                  // wave-uniform
    case 7: row_mfma<true, true, true>(wf, slab, pbase, aN, aC, aP, v, coef4); break;
    case 3: row_mfma<true, true, false>(wf, slab, pbase, aN, aC, aP, v, coef4); break;
    case 6: row_mfma<false, true, true>(wf, slab, pbase, aN, aC, aP, v, coef4); break;
    case 1: row_mfma<true, false, false>(wf, slab, pbase, aN, aC, aP, v, coef4); break;
    case 4: row_mfma<false, false, true>(wf, slab, pbase, aN, aC, aP, v, coef4); break;
    case 2: row_mfma<false, true, false>(wf, slab, pbase, aN, aC, aP, v, coef4); break;
    default: break;
  }
}

// sum over the 32 lanes of each half-wave by DPP (5 VALU instructions, no LDS); the total lands in lanes 16..31 / 48..63
__device__ __forceinline__ float half_wave_sum(float x) {
#define MCL_DPP_ADD(ctrl, rmask)                                                                              \
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, rmask, 0xF, false))
  MCL_DPP_ADD(0xB1, 0xF);     // quad_perm [1,0,3,2]
  MCL_DPP_ADD(0x4E, 0xF);     // quad_perm [2,3,0,1]
  MCL_DPP_ADD(0x141, 0xF);    // row_half_mirror
  MCL_DPP_ADD(0x140, 0xF);    // row_mirror: every lane of a 16-lane row holds the row sum
  MCL_DPP_ADD(0x142, 0xA);    // row_bcast15 into rows 1 and 3: + the sum of the row below
#undef MCL_DPP_ADD
  return x;
}

// Completed output row: acc[r] = y[pixel = lane & 31][co = (r & 3) + 8 (r >> 2) + 4 h].  bf16 rounding, running sums of the
// rounded values (the statistics the consumers' norm1 need), two 16-byte stores per lane.
// ``orow`` = buffer descriptor of the output image row (exactly its valid bytes): a lane whose pixel lies beyond the image
// width addresses past the descriptor's range and the hardware drops its stores -- no branch.
typedef short v4s __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Batch statistics of the output on the matrix cores.  A lane of the accumulator layout owns ONE pixel, so per-channel
// sums over pixels would be 32 per-lane running sums (no registers left) or a cross-lane reduction per row.  Instead the
// bf16 row just emitted (32 pixels x 32 channels) goes through a 2 KB per-wave LDS scratch as [pixel][channel] and comes
// back through the transposing read ds_read_b64_tr_b16 as MFMA fragments "8 consecutive pixels of one channel"; with
// F_cb = the fragment of channel block cb (16 channels) two 16x16x32 MFMAs per block accumulate
//     gram_cb += F_cb^T F_cb   (diagonal = sum over pixels of y^2, exact: bf16 x bf16 products in fp32)
//     sum_cb  += F_cb^T 1      (every column = sum over pixels of y)
// -- 16 accumulator registers, 4 short MFMAs and 6 LDS instructions per row, no VALU reduction.
constexpr int SROW = 64;                        // scratch bytes per pixel row (32 channels bf16)
struct RowStats {
  f32x4 gram[2], sum[2];
};

// Completed output row: acc[r] = y[pixel = lane & 31][co = (r & 3) + 8 (r >> 2) + 4 h].  bf16 rounding, two 16-byte stores
// per lane.  ``orow`` = buffer descriptor of the output image row (exactly its valid bytes): a lane whose pixel lies beyond
// the image width addresses past the descriptor's range and the hardware drops its stores -- no branch.
template <bool STATS>
__device__ __forceinline__ void emit_row(const f32x16& a, __amdgpu_buffer_rsrc_t orow, int ln, int x0, int W, unsigned ldo2,
                                         unsigned char* __restrict__ scratch, RowStats& st) {
  const int px = x0 + (ln & 31);
  const unsigned ooff = (unsigned)px * ldo2 + 16u * (unsigned)(ln >> 5);
  unsigned pk[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) pk[q] = pack2(a[2 * q], a[2 * q + 1]);
  // piece P_q = (pk[2q], pk[2q+1]) = channels 8q + 4h + {0..3}.  Swap pairs (P0, P1) and (P2, P3) between the half-waves: the
  // lower half then holds channels 0-7 / 16-23 of its pixel, the upper half 8-15 / 24-31, each as 16 contiguous bytes.
#pragma unroll
  for (int g = 0; g < 2; ++g) {
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const u32x2 r = __builtin_amdgcn_permlane32_swap(pk[4 * g + d], pk[4 * g + 2 + d], false, false);
      pk[4 * g + d] = r[0];
      pk[4 * g + 2 + d] = r[1];
    }
  }
  const u32x4 lo4 = {pk[0], pk[1], pk[2], pk[3]}, hi4 = {pk[4], pk[5], pk[6], pk[7]};
  __builtin_amdgcn_raw_buffer_store_b128(lo4, orow, ooff, 0, 0);
  __builtin_amdgcn_raw_buffer_store_b128(hi4, orow, ooff + 32u, 0, 0);
  if (STATS) {
    // scratch[pixel][channel] bf16, 64-byte rows, 16-byte chunk c of pixel row r stored at chunk c ^ ((r >> 1) & 3) (the stores of
    // 8 consecutive lanes then tile all 32 banks); pixels beyond the image width contribute zeros
    const bool valid = px < W;
    unsigned char* wp = scratch + (ln & 31) * SROW;
    const int wsw = ((ln & 31) >> 1) & 3, hh = ln >> 5;
    *reinterpret_cast<uint4*>(wp + ((hh ^ wsw) << 4)) = valid ? make_uint4(pk[0], pk[1], pk[2], pk[3]) : make_uint4(0u, 0u, 0u, 0u);
    *reinterpret_cast<uint4*>(wp + (((2 + hh) ^ wsw) << 4)) = valid ? make_uint4(pk[4], pk[5], pk[6], pk[7]) : make_uint4(0u, 0u, 0u, 0u);
    // fragment of channel block cb for the 16x16x32 MFMA: lane (i = l & 15, g = l >> 4) gets pixels 8g .. 8g+7 of channel
    // 16 cb + i.  ds_read_b64_tr_b16: within a 16-lane group lane i supplies the address of row (i >> 2), 8-byte piece
    // (i & 3) of a 4 x 16 block and receives column i of it.
    const int i = ln & 15, g = ln >> 4;
    const int r0 = 8 * g + (i >> 2);                      // pixel row of the first read; the second reads row r0 + 4
    const bf16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int ch = 2 * cb + ((i & 3) >> 1);             // 16-byte chunk of this lane's 8-byte piece
      const unsigned char* p0 = scratch + r0 * SROW + ((ch ^ ((r0 >> 1) & 3)) << 4) + (i & 1) * 8;
      const unsigned char* p1 = scratch + (r0 + 4) * SROW + ((ch ^ (((r0 + 4) >> 1) & 3)) << 4) + (i & 1) * 8;
      const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)p0);
      const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)p1);
      const bf16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      st.gram[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f, f, st.gram[cb], 0, 0, 0);
      st.sum[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f, ones, st.sum[cb], 0, 0, 0);
    }
  }
}

__global__ __launch_bounds__(64 * NWAVE, 2) void conv3x3_fwd_rows_kernel(
    const bf16_t* __restrict__ z, int nimg, int H, int W, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ mean, const float* __restrict__ rstd, const bf16_t* __restrict__ W2,
    bf16_t* __restrict__ out, long long ldo, float2* __restrict__ partial, int nunits, int rc, int nchunk, int nstrip) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform: unit / row indices live in SGPRs
  const int h = lane >> 5, l31 = lane & 31;

  // ---- the weight (32, 1152) bf16 -> LDS in its HBM order (coalesced copy), each channel's row padded by 16 B: a fragment
  // read (lane = channel, stride 2320 B) then touches 16 different 16-byte slots per 16-lane group
#pragma unroll 3
  for (int g = tid; g < CO * 144; g += 64 * NWAVE) {
    const int co = g / 144, c = g - co * 144;
    *reinterpret_cast<uint4*>(lds + co * WROW + c * 16) = *reinterpret_cast<const uint4*>(W2 + (long long)g * 8);
  }
  // BatchNorm coefficients -> LDS as quads (scale[2k], scale[2k+1], shift[2k], shift[2k+1]) per channel pair: the staging
  // lane of chunk column cc reads its quad k = 0..3 when it transforms dword k (4 VGPRs live instead of 16)
  float* coef = reinterpret_cast<float*>(lds + WF_BYTES + NWAVE * SLAB_BYTES);
  if (tid < CI) {
    const float scv = gamma[tid] * rstd[tid];
    const int quad = ((tid >> 1) & 3) * 16 + (tid >> 3);        // [k = pair within the chunk][cc = chunk column]: lanes read 16 B apart
    coef[quad * 4 + (tid & 1)] = scv;
    coef[quad * 4 + 2 + (tid & 1)] = fmaf(-mean[tid], scv, beta[tid]);
  }
  __syncthreads();                                   // the only workgroup-wide barrier

  const unsigned char* wf = lds + l31 * WROW + h * 16;
  unsigned char* slab = lds + WF_BYTES + wave * SLAB_BYTES;
  unsigned char* scratch = lds + WF_BYTES + NWAVE * SLAB_BYTES + CI * 8 + wave * (32 * SROW);     // statistics transpose buffer
  const int pbase[3] = {lane, 0, 0};               // row_mfma derives its fragment offsets from the lane id per call
  const unsigned row_bytes = (unsigned)W * 256u;                         // one image row of z
  const unsigned orow_bytes = (unsigned)(((long long)(W - 1) * ldo + CO) * 2);
  const unsigned ldo2 = (unsigned)(ldo * 2);

  const int total_waves = gridDim.x * NWAVE;
  for (int u = blockIdx.x * NWAVE + wave; u < nunits; u += total_waves) {
    const int strip = u % nstrip, t = u / nstrip;
    const int chunk = t % nchunk, b = t / nchunk;
    const int x0 = strip * 32;
    const int j0 = chunk * rc, j1 = min(H, j0 + rc);
    const int jin0 = max(0, j0 - 1), jin1 = min(H, j1 + 1);
    const long long img = (long long)b * H;
    // lane stages slab rows i = sr + 4 tt (pixel x = x0 - 1 + i), chunk column cc.  The row is read through a buffer
    // descriptor that covers exactly the image row: x = -1 wraps to a huge unsigned offset, x >= W lies past the end --
    // the hardware returns zeros for both, no branch, no address clamp.
    // (lane-derived offsets are recomputed from an opaque copy of the lane id at each use: kept live across the row loop they
    // get spilled, and a scratch reload costs an s_waitcnt vmcnt(0) in the middle of the load / MFMA pipeline)
    auto opaque_lane = [&]() { int ln = lane; asm volatile("" : "+v"(ln)); return ln; };

    // two staging register sets: while row j is multiplied, row j+1 (requested one row earlier) is transformed in the MFMA
    // shadow and row j+2 is in flight from HBM -- 18 KB per wave outstanding, what the HBM latency needs at 8 waves per CU
    u32x4 vA[9], vB[9];
    // (always issued: past the unit's last input row the descriptor has zero size and the hardware returns zeros without a
    // memory access -- a branch around the loads would make their destination registers phi values, which the compiler
    // resolves with copies behind s_waitcnt vmcnt(0))
    auto load_row = [&](int j, u32x4 (&v)[9]) {
      const int ln = opaque_lane();
      // vector offsets only (the scalar soffset operand is not range-checked) and never negative: a wrapped 32-bit offset plus
      // an immediate must not depend on how wide the hardware adds them.  Only (x0 = 0, first pixel row, tt = 0) has x = -1:
      // it reads pixel 0 instead and write_row stores zeros for it anyway.
      const int xs = x0 - 1 + (ln >> 4);
      const unsigned off0 = (unsigned)(max(xs, 0) * 256 + (ln & 15) * 16);
      unsigned off1 = (unsigned)((xs + 4) * 256 + (ln & 15) * 16);
      asm volatile("" : "+v"(off1));
      const __amdgpu_buffer_rsrc_t zrow = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<bf16_t*>(z) + (img + min(j, jin1 - 1)) * W * CI, 0, j < jin1 ? row_bytes : 0u, 0x00020000);
      v[0] = __builtin_amdgcn_raw_buffer_load_b128(zrow, off0, 0, 0);
#pragma unroll
      for (int tt = 1; tt < 9; ++tt) v[tt] = __builtin_amdgcn_raw_buffer_load_b128(zrow, off1 + (tt - 1) * 1024u, 0, 0);
    };
    const float4* coef4 = reinterpret_cast<const float4*>(coef) + (lane & 15);        // quad k at coef4[16 k]
    auto transform_row = [&](u32x4 (&v)[9]) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float4 c = coef4[16 * k];
#pragma unroll
        for (int tt = 0; tt < 9; ++tt) v[tt][k] = bn_relu_pair(v[tt][k], c.x, c.y, c.z, c.w);
      }
    };
    // v[] holds a2 = relu(bn(z)) already.  Slab row i = sr + 4 tt, swizzled chunk cc ^ (i & 15) = (cc ^ sr) ^ 4 (tt & 3): four
    // lane-constant bases + tt * 1024.  The conv's zero padding applies to a2, not to z: out-of-image pixels store zeros.
    auto write_row = [&](u32x4 (&v)[9]) {
      const int ln = opaque_lane();
      const int sr = ln >> 4, cc = ln & 15, wbase = sr * 256 + ((cc ^ sr) << 4);
#pragma unroll
      for (int tt = 0; tt < 9; ++tt) {
        const int i = sr + 4 * tt, x = x0 - 1 + i;
        const bool ok = x >= 0 && x < W;
        if (i < SLAB_ROWS)
          *reinterpret_cast<uint4*>(slab + (wbase ^ ((tt & 3) << 6)) + tt * 1024) =
              ok ? make_uint4(v[tt][0], v[tt][1], v[tt][2], v[tt][3]) : make_uint4(0u, 0u, 0u, 0u);
      }
    };

    RowStats st;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) st.gram[cb][r] = st.sum[cb][r] = 0.0f;
    f32x16 aP, aC, aN;
#pragma unroll
    for (int r = 0; r < 16; ++r) aP[r] = aC[r] = aN[r] = 0.0f;
    load_row(jin0, vB);
    load_row(jin0 + 1, vA);
    transform_row(vB);
    write_row(vB);
    int j = jin0;
    bool more = true;
    // P = output row j-1 (completes at input row j), C = row j, N = row j+1 (first contribution); X = staging set holding input
    // row j+1 (raw; transformed inside row_mfma), Y = the free set that receives row j+2
#define MCL_STEP(X, Y)                                                                                         \
    {                                                                                                          \
      more = j + 1 < jin1;                                                                                     \
      load_row(j + 2, Y);                                                                                      \
      const int m = (j + 1 < j1 ? 1 : 0) | ((j >= j0 && j < j1) ? 2 : 0) | (j - 1 >= j0 ? 4 : 0);              \
      row_mfma_dispatch(m, wf, slab, pbase, aN, aC, aP, X, coef4);                                             \
      if ((m & 4))                                                                           \
        emit_row<true>(aP, __builtin_amdgcn_make_buffer_rsrc(out + (img + j - 1) * W * ldo, 0, orow_bytes, 0x00020000), \
                       opaque_lane(), x0, W, ldo2, scratch, st);                                               \
      if (j == H - 1 && (m & 2))                                                                               \
        emit_row<true>(aC, __builtin_amdgcn_make_buffer_rsrc(out + (img + j) * W * ldo, 0, orow_bytes, 0x00020000),    \
                       opaque_lane(), x0, W, ldo2, scratch, st);                                               \
      if (more) write_row(X);                                                                \
      aP = aC;                                                                                                 \
      aC = aN;                                                                                                 \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) aN[r] = 0.0f;                                             \
      ++j;                                                                                                     \
    }
    while (true) {
      MCL_STEP(vA, vB)
      if (!more) break;
      MCL_STEP(vB, vA)
      if (!more) break;
    }
#undef MCL_STEP

    // ---- unit statistics: lane i + 16 (i >> 2) of the 16x16 accumulator layout (column l & 15, rows 4 (l >> 4) + r) holds
    // the diagonal element gram[i][i] and sum[i][.] in register r = i & 3
    if (partial != nullptr) {
      const int i = lane & 15;
      if ((lane >> 4) == (i >> 2)) {
        const int r = i & 3;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          const float q = r == 0 ? st.gram[cb][0] : r == 1 ? st.gram[cb][1] : r == 2 ? st.gram[cb][2] : st.gram[cb][3];
          const float sm = r == 0 ? st.sum[cb][0] : r == 1 ? st.sum[cb][1] : r == 2 ? st.sum[cb][2] : st.sum[cb][3];
          partial[(long long)u * CO + 16 * cb + i] = make_float2(sm, q);
        }
      }
    }

  }
}

// one workgroup per channel: total sum and sum of squares over the unit partials (partial[unit][channel]), in double,
// fixed order
__global__ __launch_bounds__(256) void sums_finalize_kernel(const float2* __restrict__ partial, int nunits, long long S,
                                                            float eps, float* __restrict__ mean, float* __restrict__ var,
                                                            float* __restrict__ rstd) {
  __shared__ double red[2][4];
  const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float2* p = partial + c;
  double sum = 0.0, q = 0.0;
  // batches of eight independent loads (clamped index, surplus zeroed after the load), added in ascending order
  constexpr int U = 8;
  for (int t0 = threadIdx.x; t0 < nunits; t0 += 256 * U) {
    float2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = p[(long long)min(t0 + 256 * u, nunits - 1) * CO];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (t0 + 256 * u >= nunits) v[u] = make_float2(0.0f, 0.0f);
      sum += (double)v[u].x;
      q += (double)v[u].y;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sum += __shfl_xor(sum, o, 64);
    q += __shfl_xor(q, o, 64);
  }
  if (lane == 0) {
    red[0][wave] = sum;
    red[1][wave] = q;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  sum = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
  q = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  const double n = (double)S, m = sum / n;
  double vv = (q - sum * sum / n) / n;
  if (vv < 0.0) vv = 0.0;
  mean[c] = (float)m;
  var[c] = (float)vv;
  rstd[c] = (float)(1.0 / sqrt(vv + (double)eps));
}

struct RowsPlan {
  int nimg, nstrip, rc, nchunk, nunits, grid;
};

inline RowsPlan rows_plan(long long S, int H, int W) {
  RowsPlan p;
  p.nimg = (int)(S / ((long long)H * W));
  p.nstrip = (W + 31) / 32;
  // row chunks sized so that the units fill the 256 CUs x 8 wave slots about once; at least 2 output rows per unit
  // (every unit re-stages two extra input rows)
  long long rc = ((long long)H * p.nimg * p.nstrip) / 2048;
  if (rc < 2) rc = 2;
  if (rc > H) rc = H;
  p.rc = (int)rc;
  p.nchunk = (H + p.rc - 1) / p.rc;
  p.nunits = p.nimg * p.nchunk * p.nstrip;
  p.grid = (p.nunits + NWAVE - 1) / NWAVE;
  if (p.grid > 256) p.grid = 256;
  return p;
}

}  // namespace

bool mcl_conv3x3_rows_applicable(long long S, int H, int W) {
  // (maps narrower than 17 pixels keep the flat-tile kernels: the row form measured 12.16-12.28 vs 11.90 ms/step there)
  return W >= 17 && W <= 150 && H >= 1 && S % ((long long)H * W) == 0;
}

long long mcl_conv3x3_rows_workspace_floats(long long S) {
  // units <= image rows x strips / 2 <= S / 16; 32 channels x float2 each
  return (S / 16 + 8) * CO * 2;
}

int mcl_launch_conv3x3_fwd_rows(const void* z, long long S, int H, int W, const float* gamma, const float* beta,
                                const float* mean, const float* rstd, const void* W2, void* out, long long ldo,
                                float* workspace, float eps, float* ymean, float* yvar, float* yrstd, hipStream_t st) {
  const RowsPlan p = rows_plan(S, H, W);
  const size_t lds_bytes = WF_BYTES + NWAVE * SLAB_BYTES + CI * 8 + NWAVE * 32 * SROW;
  static mcl_device_once attr_once;
  if (auto attr_guard = attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_fwd_rows_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  float2* part = reinterpret_cast<float2*>(workspace);
  hipLaunchKernelGGL(conv3x3_fwd_rows_kernel, dim3(p.grid), dim3(64 * NWAVE), lds_bytes, st, (const bf16_t*)z, p.nimg, H, W,
                     gamma, beta, mean, rstd, (const bf16_t*)W2, (bf16_t*)out, ldo, ymean != nullptr ? part : (float2*)nullptr,
                     p.nunits, p.rc, p.nchunk, p.nstrip);
  if (ymean != nullptr)
    hipLaunchKernelGGL(sums_finalize_kernel, dim3(CO), dim3(256), 0, st, (const float2*)part, p.nunits, S, eps, ymean, yvar,
                       yrstd);
  return 0;
}
