// Experiment (DESIGN 4.2 / VERDICT r03 #1): what does an order-independent statistics hand-off cost on MI355X?
//
// A producer grid of NT workgroups ("tiles") each contributes one value per (channel, quantity) -- 128 channels x 2
// quantities, like the conv1x1 forward epilogue.  Compared:
//   mode 0  float2 partial store per tile (what the shipped kernels do; a finalize LAUNCH merges them)
//   mode 1  exact fixed-point accumulation: the double is split into three 40-bit limbs of a 120-bit integer and every
//           non-zero limb is added with a no-return 64-bit AGENT-scope atomic (integer addition: order-independent)
//   mode 2  the same limbs, WORKGROUP-scope atomics (executed in the XCD's own L2) into a per-XCD replica selected by
//           HW_REG_XCC_ID; the consumer adds the eight replicas
//   mode 3  agent-scope atomics into replica blockIdx % 8 (contention / 8 without relying on the XCC id)
// and a consumer grid whose every workgroup rebuilds (mean, rstd) of the 128 channels from the accumulators, against
// the finalize kernel it would replace.  Results are checked against exact host integer sums.
//
//   hipcc --offload-arch=gfx950 -O3 tools/exp_stat_atomics.hip -o tools/bin/exp_stat_atomics && tools/bin/exp_stat_atomics
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int NCH = 128, NQ = 2, NLIMB = 3, NREP = 8;
constexpr int FRAC = 80;                       // unit 2^-80; 120 bits: |sum| < 2^39
constexpr long long MASK40 = (1LL << 40) - 1;

struct Limbs { long long l[3]; };

__host__ __device__ inline Limbs split_fixed(double v) {
  // v = m * 2^e with |m| < 2^53 an integer
  int e;
  const double fr = frexp(v, &e);              // v = fr * 2^e, 0.5 <= |fr| < 1
  long long m = (long long)ldexp(fr, 53);      // exact
  int sh = e - 53 + FRAC;                      // X = m << sh
  __int128 X;
  if (v == 0.0) X = 0;
  else if (sh >= 0) X = (sh > 66) ? ((__int128)(m < 0 ? -1 : 1) << 119) : ((__int128)m << sh);
  else X = (sh < -63) ? (m < 0 ? -1 : 0) : (__int128)(m >> (-sh));    // floor
  Limbs r;
  r.l[0] = (long long)(X & MASK40);
  r.l[1] = (long long)((X >> 40) & MASK40);
  r.l[2] = (long long)(X >> 80);
  return r;
}

__host__ __device__ inline double join_fixed(long long l0, long long l1, long long l2) {
  l1 += l0 >> 40; l0 &= MASK40;
  l2 += l1 >> 40; l1 &= MASK40;
  return ldexp(((double)l2 * 1099511627776.0 + (double)l1) * 1099511627776.0 + (double)l0, -FRAC);
}

__device__ __forceinline__ double tile_value(int tile, int ch, int q) {
  // deterministic pseudo-data: quantity 0 signed, quantity 1 positive, a wide range of magnitudes over channels
  unsigned h = (unsigned)tile * 2654435761u ^ (unsigned)(ch * 97 + q) * 40503u;
  h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
  const double u = (double)(h & 0xFFFFFF) / 16777216.0;             // [0, 1)
  const double mag = ldexp(1.0, (ch % 40) - 30);                    // 2^-30 .. 2^9
  return q == 0 ? (u - 0.5) * mag : (float)(u * mag);
}

__device__ __forceinline__ int xcc_id() {
  int v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 7;
}

// stream: optional background read so that the epilogue runs beside memory traffic (bytes per workgroup)
template <int MODE>
__global__ __launch_bounds__(256) void producer(float2* __restrict__ partial, long long* __restrict__ acc, int nt,
                                                const float4* __restrict__ bg, int bg_vec_per_thread, float* sink) {
  const int tid = threadIdx.x, tile = blockIdx.x;
  float4 s = make_float4(0, 0, 0, 0);
  if (bg_vec_per_thread > 0) {
    const float4* p = bg + ((long long)tile * 256 * bg_vec_per_thread) + tid;
    for (int i = 0; i < bg_vec_per_thread; ++i) { const float4 v = p[(long long)i * 256]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    if (s.x == 1.2345f) sink[0] = s.x + s.y + s.z + s.w;
  }
  if (tid >= NCH) return;
  const double v0 = tile_value(tile, tid, 0), v1 = tile_value(tile, tid, 1);
  if (MODE == 0) {
    partial[(long long)tid * nt + tile] = make_float2((float)v0, (float)v1);
    return;
  }
  int rep = 0;
  if (MODE == 2) rep = xcc_id();
  if (MODE == 3) rep = tile & 7;
  long long* a = acc + ((long long)rep * NCH + tid) * 8;              // 64 B per (replica, channel): q0 limbs 0..2, q1 limbs 4..6
  const Limbs x0 = split_fixed(v0), x1 = split_fixed(v1);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (MODE == 2) {
      if (x0.l[i]) __hip_atomic_fetch_add(a + i, x0.l[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (x1.l[i]) __hip_atomic_fetch_add(a + 4 + i, x1.l[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
      if (x0.l[i]) __hip_atomic_fetch_add(a + i, x0.l[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (x1.l[i]) __hip_atomic_fetch_add(a + 4 + i, x1.l[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// consumer prologue: every workgroup rebuilds the 128 (mean, rstd) pairs from NR replicas (1 or 8)
template <int NR>
__global__ __launch_bounds__(256) void consumer(const long long* __restrict__ acc, double n, float* __restrict__ out,
                                                double* __restrict__ raw) {
  __shared__ float tab[2 * NCH];
  const int tid = threadIdx.x;
  if (tid < NCH) {
    long long l[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const longlong4* p = reinterpret_cast<const longlong4*>(acc + ((long long)r * NCH + tid) * 8);
      const longlong4 a = p[0], b = p[1];
      l[0] += a.x; l[1] += a.y; l[2] += a.z; l[4] += b.x; l[5] += b.y; l[6] += b.z;
    }
    const double s = join_fixed(l[0], l[1], l[2]), q = join_fixed(l[4], l[5], l[6]);
    const double m = s / n;
    double v = q / n - m * m;
    if (v < 0) v = 0;
    tab[tid] = (float)m;
    tab[NCH + tid] = (float)(1.0 / sqrt(v + 1e-5));
    if (blockIdx.x == 0 && raw) { raw[tid] = s; raw[NCH + tid] = q; }
  }
  __syncthreads();
  if (tid < 2 * NCH && out) out[(long long)blockIdx.x * 2 * NCH + tid] = tab[tid];   // (keeps the prologue alive)
}

// the finalize launch being replaced (one workgroup per channel over nt float2 partials, double, fixed order)
__global__ __launch_bounds__(256) void finalize(const float2* __restrict__ partial, int nt, double n, float* __restrict__ out) {
  __shared__ double red[2][4];
  const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float2* p = partial + (long long)c * nt;
  double s = 0, q = 0;
  for (int t = threadIdx.x; t < nt; t += 256) { s += (double)p[t].x; q += (double)p[t].y; }
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
  if (lane == 0) { red[0][wave] = s; red[1][wave] = q; }
  __syncthreads();
  if (threadIdx.x) return;
  s = red[0][0] + red[0][1] + red[0][2] + red[0][3];
  q = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  const double m = s / n;
  double v = q / n - m * m;
  if (v < 0) v = 0;
  out[c] = (float)m;
  out[NCH + c] = (float)(1.0 / sqrt(v + 1e-5));
}

static double host_tile_value(int tile, int ch, int q) {
  unsigned h = (unsigned)tile * 2654435761u ^ (unsigned)(ch * 97 + q) * 40503u;
  h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
  const double u = (double)(h & 0xFFFFFF) / 16777216.0;
  const double mag = ldexp(1.0, (ch % 40) - 30);
  return q == 0 ? (u - 0.5) * mag : (double)(float)(u * mag);
}

template <typename F>
static float time_us(F&& f, int iters, hipStream_t st) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 5; ++i) f();
  CK(hipEventRecord(a, st));
  for (int i = 0; i < iters; ++i) f();
  CK(hipEventRecord(b, st));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1000.0f / iters;
}

int main() {
  hipStream_t st;
  CK(hipStreamCreate(&st));
  const int tiles[] = {98, 392, 784, 3136};
  const size_t acc_bytes = (size_t)NREP * NCH * 8 * sizeof(long long);
  long long* acc; float2* partial; float* out; double* raw; float4* bg; float* sink;
  CK(hipMalloc(&acc, acc_bytes));
  CK(hipMalloc(&partial, (size_t)NCH * 4096 * sizeof(float2)));
  CK(hipMalloc(&out, (size_t)4096 * 2 * NCH * sizeof(float)));
  CK(hipMalloc(&raw, 2 * NCH * sizeof(double)));
  const int bgv = 16;                                        // 16 x 16 B x 256 threads = 64 KB per workgroup
  CK(hipMalloc(&bg, (size_t)4096 * 256 * bgv * sizeof(float4)));
  CK(hipMemset(bg, 0, (size_t)4096 * 256 * bgv * sizeof(float4)));
  CK(hipMalloc(&sink, 16));

  for (int nt : tiles) {
    // exact host reference of the limb sums (integers)
    std::vector<double> want(2 * NCH);
    for (int ch = 0; ch < NCH; ++ch)
      for (int q = 0; q < 2; ++q) {
        __int128 X = 0;
        for (int t = 0; t < nt; ++t) {
          const Limbs x = split_fixed(host_tile_value(t, ch, q));
          X += ((__int128)x.l[2] << 80) + ((__int128)x.l[1] << 40) + x.l[0];
        }
        const long long l0 = (long long)(X & MASK40), l1 = (long long)((X >> 40) & MASK40), l2 = (long long)(X >> 80);
        want[q * NCH + ch] = join_fixed(l0, l1, l2);
      }
    // correctness of modes 1..3 (one accumulation from zero, consumer with the matching replica count)
    for (int mode = 1; mode <= 3; ++mode) {
      CK(hipMemsetAsync(acc, 0, acc_bytes, st));
      if (mode == 1) hipLaunchKernelGGL(producer<1>, dim3(nt), dim3(256), 0, st, partial, acc, nt, bg, 0, sink);
      if (mode == 2) hipLaunchKernelGGL(producer<2>, dim3(nt), dim3(256), 0, st, partial, acc, nt, bg, 0, sink);
      if (mode == 3) hipLaunchKernelGGL(producer<3>, dim3(nt), dim3(256), 0, st, partial, acc, nt, bg, 0, sink);
      if (mode == 1) hipLaunchKernelGGL(consumer<1>, dim3(1), dim3(256), 0, st, acc, (double)nt, out, raw);
      else hipLaunchKernelGGL(consumer<8>, dim3(1), dim3(256), 0, st, acc, (double)nt, out, raw);
      CK(hipStreamSynchronize(st));
      std::vector<double> got(2 * NCH);
      CK(hipMemcpy(got.data(), raw, 2 * NCH * sizeof(double), hipMemcpyDeviceToHost));
      int bad = 0;
      for (int i = 0; i < 2 * NCH; ++i) bad += got[i] != want[i];
      printf("tiles %4d mode %d: %s (%d of %d sums differ from the exact integer sum)\n", nt, mode, bad ? "MISMATCH" : "exact", bad, 2 * NCH);
    }
    // repeatability of mode 2 under load (XCC id must be stable and the L2 atomics must not lose updates): 20 rounds
    {
      int bad_rounds = 0;
      for (int r = 0; r < 20; ++r) {
        CK(hipMemsetAsync(acc, 0, acc_bytes, st));
        hipLaunchKernelGGL(producer<2>, dim3(nt), dim3(256), 0, st, partial, acc, nt, bg, bgv, sink);
        hipLaunchKernelGGL(consumer<8>, dim3(1), dim3(256), 0, st, acc, (double)nt, out, raw);
        CK(hipStreamSynchronize(st));
        std::vector<double> got(2 * NCH);
        CK(hipMemcpy(got.data(), raw, 2 * NCH * sizeof(double), hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < 2 * NCH; ++i) bad += got[i] != want[i];
        bad_rounds += bad != 0;
      }
      printf("tiles %4d mode 2 under load, 20 rounds: %d rounds with a mismatch\n", nt, bad_rounds);
    }
    // timing: producer alone (epilogue only, then with a 64 KB streaming read per workgroup in front of it)
    for (int bgn : {0, bgv}) {
      const float t0 = time_us([&] { hipLaunchKernelGGL(producer<0>, dim3(nt), dim3(256), 0, st, partial, acc, nt, bg, bgn, sink); }, 200, st);
      const float t1 = time_us([&] { hipLaunchKernelGGL(producer<1>, dim3(nt), dim3(256), 0, st, partial, acc, nt, bg, bgn, sink); }, 200, st);
      const float t2 = time_us([&] { hipLaunchKernelGGL(producer<2>, dim3(nt), dim3(256), 0, st, partial, acc, nt, bg, bgn, sink); }, 200, st);
      const float t3 = time_us([&] { hipLaunchKernelGGL(producer<3>, dim3(nt), dim3(256), 0, st, partial, acc, nt, bg, bgn, sink); }, 200, st);
      printf("tiles %4d bg %2d KB  producer us/launch: float2-store %.2f | agent atomics %.2f | per-XCD L2 atomics %.2f | agent atomics 8 replicas %.2f\n",
             nt, bgn * 4, t0, t1, t2, t3);
      // producer -> dependent consumer pairs (what a layer's chain pays)
      const float p0 = time_us([&] {
        hipLaunchKernelGGL(producer<0>, dim3(nt), dim3(256), 0, st, partial, acc, nt, bg, bgn, sink);
        hipLaunchKernelGGL(finalize, dim3(NCH), dim3(256), 0, st, partial, nt, (double)nt, out);
        hipLaunchKernelGGL(consumer<1>, dim3(nt), dim3(256), 0, st, acc, (double)nt, out, (double*)nullptr);
      }, 200, st);
      const float p1 = time_us([&] {
        hipLaunchKernelGGL(producer<1>, dim3(nt), dim3(256), 0, st, partial, acc, nt, bg, bgn, sink);
        hipLaunchKernelGGL(consumer<1>, dim3(nt), dim3(256), 0, st, acc, (double)nt, out, (double*)nullptr);
      }, 200, st);
      const float p2 = time_us([&] {
        hipLaunchKernelGGL(producer<2>, dim3(nt), dim3(256), 0, st, partial, acc, nt, bg, bgn, sink);
        hipLaunchKernelGGL(consumer<8>, dim3(nt), dim3(256), 0, st, acc, (double)nt, out, (double*)nullptr);
      }, 200, st);
      const float p3 = time_us([&] {
        hipLaunchKernelGGL(producer<3>, dim3(nt), dim3(256), 0, st, partial, acc, nt, bg, bgn, sink);
        hipLaunchKernelGGL(consumer<8>, dim3(nt), dim3(256), 0, st, acc, (double)nt, out, (double*)nullptr);
      }, 200, st);
      printf("tiles %4d bg %2d KB  chain us: store+finalize+consumer %.2f | agent+consumer %.2f | per-XCD+consumer<8> %.2f | 8 replicas+consumer<8> %.2f\n",
             nt, bgn * 4, p0, p1, p2, p3);
    }
  }
  return 0;
}
