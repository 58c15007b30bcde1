// K9: torch.optim.Adam(lr, betas, eps, weight_decay) (L2-coupled), train.py:118-120, as one fused
// streaming pass.  Pure HBM traffic: 28 B/element (read p,g,m,v; write p,m,v); the embedding-table
// form reads no dense gradient (24 B/element) and adds the row-sparse data gradient through a
// row->slot map.  16-byte accesses per lane, grid-stride, ~8 workgroups per CU.
#include "common.h"
#include <stdlib.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));   // native vector type (the non-temporal builtins reject HIP's float4 class)
__device__ __forceinline__ float4 nt_load4(const float* q) {
  const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(q));
  return make_float4(t[0], t[1], t[2], t[3]);
}
__device__ __forceinline__ void nt_store4(float* q, float4 a) {
  const f32x4 t = {a.x, a.y, a.z, a.w};
  __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(q));
}

struct AdamC {
  float lr_over_bc1, beta1, beta2, omb1, omb2, eps, wd, rsqrt_bc2;  // omb = 1 - beta, rounded from double
};

// The operation sequence is spelled out instruction by instruction (contraction off, every fused multiply-add explicit):
// the lazy table kernels below replay missed steps of a row in registers and must reproduce, bit for bit, what the dense
// pass would have computed step by step -- whatever the surrounding code the compiler inlines this into.
#pragma clang fp contract(off)
__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, const AdamC& c) {
  g = fmaf(c.wd, p, g);                         // grad.add(param, alpha=wd)
  m = fmaf(c.omb1, g - m, m);                   // exp_avg.lerp_(grad, 1-beta1)
  v = fmaf(c.beta2, v, (c.omb2 * g) * g);       // exp_avg_sq.mul_(b2).addcmul_(g, g, value=1-b2)
  const float denom = fmaf(sqrtf(v), c.rsqrt_bc2, c.eps);
  p = fmaf(-c.lr_over_bc1, m / denom, p);
}

__device__ __forceinline__ unsigned bf16_rne(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7F800000u) == 0x7F800000u) return u >> 16;
  u += 0x7FFFu + ((u >> 16) & 1u);
  return u >> 16;
}

// SHADOW: also store the updated parameter rounded to bf16 (the flat low-precision copy the backbone kernels read: the
// separate 63 MB -> 31 MB cast pass after the step disappears)
template <bool SHADOW>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long long n,
                                                   AdamC c_host, const AdamC* __restrict__ c_dev,
                                                   unsigned short* __restrict__ shadow) {
  const AdamC c = c_dev ? *c_dev : c_host;      // device-resident constants: the launch can be replayed from a HIP graph
  const long long n4 = n >> 2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  float4* p4 = reinterpret_cast<float4*>(p);
  const float4* g4 = reinterpret_cast<const float4*>(g);
  float4* m4 = reinterpret_cast<float4*>(m);
  float4* v4 = reinterpret_cast<float4*>(v);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
    adam1(pp.x, gg.x, mm.x, vv.x, c);
    adam1(pp.y, gg.y, mm.y, vv.y, c);
    adam1(pp.z, gg.z, mm.z, vv.z, c);
    adam1(pp.w, gg.w, mm.w, vv.w, c);
    p4[i] = pp; m4[i] = mm; v4[i] = vv;
    if (SHADOW)
      reinterpret_cast<uint2*>(shadow)[i] = make_uint2(bf16_rne(pp.x) | (bf16_rne(pp.y) << 16),
                                                       bf16_rne(pp.z) | (bf16_rne(pp.w) << 16));
  }
  const long long tail0 = n4 << 2;
  for (long long i = tail0 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    adam1(p[i], g[i], m[i], v[i], c);
    if (SHADOW) shadow[i] = (unsigned short)bf16_rne(p[i]);
  }
}

// unaligned fallback (base pointers not 16-byte aligned)
__global__ __launch_bounds__(256) void adam_kernel_scalar(float* __restrict__ p, const float* __restrict__ g,
                                                          float* __restrict__ m, float* __restrict__ v, long long n,
                                                          AdamC c_host, const AdamC* __restrict__ c_dev) {
  const AdamC c = c_dev ? *c_dev : c_host;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    adam1(p[i], g[i], m[i], v[i], c);
}

// One workgroup walks whole rows (grid-stride over rows); the row's slot is wave-uniform.  The vector form handles
// RPI rows per iteration so that every thread has 3*RPI independent 16-byte loads in flight (one row at a time
// leaves a thread with three and the pass latency-bound: 5.1 TB/s), and streams with non-temporal loads/stores:
// the 3 x 262 MB of table state are touched once per step and would only evict everything else from L2/MALL.
template <bool VEC>
__global__ __launch_bounds__(256) void adam_table_kernel(float* __restrict__ p, float* __restrict__ m,
                                                         float* __restrict__ v, int n_rows, int cols,
                                                         const int* __restrict__ row_slot,
                                                         const float* __restrict__ rg, long long ldrg, AdamC c_host,
                                                         const AdamC* __restrict__ c_dev) {
  const AdamC c = c_dev ? *c_dev : c_host;
  if (VEC) {
    constexpr int RPI = 4;
    const int c4 = cols >> 2;
    for (int r0 = blockIdx.x * RPI; r0 < n_rows; r0 += gridDim.x * RPI) {
      for (int i = threadIdx.x; i < c4; i += 256) {
        float4 pp[RPI], mm[RPI], vv[RPI], gg[RPI];
        bool ok[RPI];
#pragma unroll
        for (int k = 0; k < RPI; ++k) {
          const int r = r0 + k;
          ok[k] = r < n_rows;
          const long long base = (long long)(ok[k] ? r : r0) * cols;
          pp[k] = nt_load4(p + base + 4 * i);
          mm[k] = nt_load4(m + base + 4 * i);
          vv[k] = nt_load4(v + base + 4 * i);
          const int slot = ok[k] ? row_slot[r] : -1;
          gg[k] = slot >= 0 ? reinterpret_cast<const float4*>(rg + (long long)slot * ldrg)[i]
                            : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < RPI; ++k) {
          if (!ok[k]) continue;
          adam1(pp[k].x, gg[k].x, mm[k].x, vv[k].x, c);
          adam1(pp[k].y, gg[k].y, mm[k].y, vv[k].y, c);
          adam1(pp[k].z, gg[k].z, mm[k].z, vv[k].z, c);
          adam1(pp[k].w, gg[k].w, mm[k].w, vv[k].w, c);
          const long long base = (long long)(r0 + k) * cols;
          nt_store4(p + base + 4 * i, pp[k]);
          nt_store4(m + base + 4 * i, mm[k]);
          nt_store4(v + base + 4 * i, vv[k]);
        }
      }
    }
  } else {
    for (int r = blockIdx.x; r < n_rows; r += gridDim.x) {
      const int slot = row_slot[r];
      const long long base = (long long)r * cols;
      const float* g = slot >= 0 ? rg + (long long)slot * ldrg : nullptr;
      for (int i = threadIdx.x; i < cols; i += 256) adam1(p[base + i], g ? g[i] : 0.0f, m[base + i], v[base + i], c);
    }
  }
}

// One thread: advances the device-resident step counter and derives this step's constants from it (double arithmetic,
// rounded once, exactly as make_consts does on the host).  With the counter on the device the whole optimizer step can
// be captured in a HIP graph and replayed: nothing step-dependent is baked into the launches.
// The hyper-parameters (lr, beta1, beta2, eps, weight_decay) are READ FROM DEVICE MEMORY too: a by-value kernel argument
// is frozen into a captured graph, so an LR schedule or a manual param_groups edit would be ignored by every replay.
__global__ void adam_consts_kernel(long long* __restrict__ step, AdamC* __restrict__ out,
                                   const double* __restrict__ hyper, AdamC* __restrict__ hist, int hmask) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4];
  const long long t = *step + 1;
  *step = t;
  const double bc1 = 1.0 - pow(b1, (double)t), bc2 = 1.0 - pow(b2, (double)t);
  AdamC c;
  c.lr_over_bc1 = (float)(lr / bc1);
  c.beta1 = (float)b1;
  c.beta2 = (float)b2;
  c.omb1 = (float)(1.0 - b1);
  c.omb2 = (float)(1.0 - b2);
  c.eps = (float)eps;
  c.wd = (float)wd;
  c.rsqrt_bc2 = (float)(1.0 / sqrt(bc2));
  *out = c;
  // the lazy position-table update replays a row's missed steps later with exactly the constants those steps used (an LR
  // schedule included): they are kept in a ring indexed by the step number
  if (hist) hist[t & hmask] = c;
}

// ---------------------------------------------------------------------------------------------------------------------
// Lazy-exact position-table Adam (hard part 1(b); reference: train.py:118-120 over model.py:204-205).
// torch.optim.Adam with L2 weight decay moves EVERY row of the two (65536, G) tables on every step (g = wd * p where the
// data gradient is zero): 1.57 GB of state traffic per step for 64-128 touched rows.  But a row without a data gradient
// follows a recurrence in its own (p, m, v) and the step's constants only, so it need not be advanced until somebody
// looks at it: every row carries a "valid through step" stamp; a row is brought up to date -- the missed steps replayed in
// registers, the same fp32 operation sequence (adam1) with the recorded per-step constants, hence bit-identical to the
// dense pass -- when it is gathered by the forward, when it receives a data gradient, or when the table is materialised
// (state_dict / checkpoint / a direct module call).  One workgroup per row; a row that is current exits after one load.
struct LazyTable {
  float* p;
  float* m;
  float* v;
  int* row_step;
  const int* owner;     // per workgroup: the row to process (< 0: nothing); nullptr: from `pos`, or row = workgroup index
  const float* rg;      // per workgroup: the row's data gradient for THIS step (nullptr: catch-up only)
};

template <bool VEC>
__global__ __launch_bounds__(256) void adam_table_lazy_kernel(LazyTable t0, LazyTable t1, int per_table, int n_rows,
                                                              int cols, const float* __restrict__ pos, long long ldrg,
                                                              const long long* __restrict__ step,
                                                              const AdamC* __restrict__ hist, int hmask) {
  const int which = blockIdx.x >= (unsigned)per_table;
  const int b = blockIdx.x - (which ? per_table : 0);
  const LazyTable T = which ? t1 : t0;
  int row;
  if (pos) {
    // forward catch-up: the row this spot gathers; the first spot of the batch that names a row owns it
    __shared__ int dup_before;
    long long i = (long long)pos[2 * b + which];          // .long(): truncation toward zero (model.py:230-231)
    i = i < 0 ? 0 : (i >= n_rows ? n_rows - 1 : i);       // (out-of-range positions are flagged by the gather kernel)
    row = (int)i;
    if (threadIdx.x == 0) dup_before = 0;
    __syncthreads();
    int found = 0;
    for (int j = threadIdx.x; j < b; j += 256) {
      long long q = (long long)pos[2 * j + which];
      q = q < 0 ? 0 : (q >= n_rows ? n_rows - 1 : q);
      found |= ((int)q == row);
    }
    if (found) dup_before = 1;
    __syncthreads();
    if (dup_before) return;
  } else {
    row = T.owner ? T.owner[b] : b;
    if (row < 0) return;
  }
  row = __builtin_amdgcn_readfirstlane(row);
  const int t = __builtin_amdgcn_readfirstlane((int)*step);
  const int rs = __builtin_amdgcn_readfirstlane(T.row_step[row]);
  const bool has_g = T.rg != nullptr;
  const int last_zero = has_g ? t - 1 : t;                // steps (rs, last_zero] are replayed with a zero data gradient
  if (!has_g && rs >= last_zero) return;
  __syncthreads();                                        // every wave has read the stamp before it is rewritten
  const long long base = (long long)row * cols;
  const float* g = has_g ? T.rg + (long long)b * ldrg : nullptr;
  if (VEC) {
    const int c4 = cols >> 2;
    for (int i = threadIdx.x; i < c4; i += 256) {
      float4 pp = reinterpret_cast<const float4*>(T.p + base)[i];
      float4 mm = reinterpret_cast<const float4*>(T.m + base)[i];
      float4 vv = reinterpret_cast<const float4*>(T.v + base)[i];
      for (int s = rs + 1; s <= last_zero; ++s) {
        const AdamC c = hist[s & hmask];
        adam1(pp.x, 0.0f, mm.x, vv.x, c);
        adam1(pp.y, 0.0f, mm.y, vv.y, c);
        adam1(pp.z, 0.0f, mm.z, vv.z, c);
        adam1(pp.w, 0.0f, mm.w, vv.w, c);
      }
      if (has_g) {
        const AdamC c = hist[t & hmask];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        adam1(pp.x, gg.x, mm.x, vv.x, c);
        adam1(pp.y, gg.y, mm.y, vv.y, c);
        adam1(pp.z, gg.z, mm.z, vv.z, c);
        adam1(pp.w, gg.w, mm.w, vv.w, c);
      }
      reinterpret_cast<float4*>(T.p + base)[i] = pp;
      reinterpret_cast<float4*>(T.m + base)[i] = mm;
      reinterpret_cast<float4*>(T.v + base)[i] = vv;
    }
  } else {
    for (int i = threadIdx.x; i < cols; i += 256) {
      float pp = T.p[base + i], mm = T.m[base + i], vv = T.v[base + i];
      for (int s = rs + 1; s <= last_zero; ++s) adam1(pp, 0.0f, mm, vv, hist[s & hmask]);
      if (has_g) adam1(pp, g[i], mm, vv, hist[t & hmask]);
      T.p[base + i] = pp;
      T.m[base + i] = mm;
      T.v[base + i] = vv;
    }
  }
  if (threadIdx.x == 0) T.row_step[row] = t;
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; }

// Scalars arrive as doubles (Python floats on the host side, as in torch.optim) and are rounded once.
inline AdamC make_consts(double lr, double b1, double b2, double eps, double wd, double bc1, double bc2) {
  AdamC c;
  c.lr_over_bc1 = (float)(lr / bc1);
  c.beta1 = (float)b1;
  c.beta2 = (float)b2;
  c.omb1 = (float)(1.0 - b1);
  c.omb2 = (float)(1.0 - b2);
  c.eps = (float)eps;
  c.wd = (float)wd;
  c.rsqrt_bc2 = (float)(1.0 / sqrt(bc2));
  return c;
}

}  // namespace

namespace {
int launch_adam(float* p, const float* g, float* m, float* v, int64_t n, const AdamC& c, const AdamC* c_dev,
                hipStream_t st, unsigned short* shadow = nullptr) {
  long long blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  if (shadow != nullptr) {
    if (!(al16(p) && al16(g) && al16(m) && al16(v)) || (reinterpret_cast<uintptr_t>(shadow) & 7u)) return MCL_EUNSUPPORTED;
    hipLaunchKernelGGL(adam_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, p, g, m, v, (long long)n, c, c_dev,
                       shadow);
  } else if (al16(p) && al16(g) && al16(m) && al16(v))
    hipLaunchKernelGGL(adam_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, p, g, m, v, (long long)n, c, c_dev,
                       (unsigned short*)nullptr);
  else
    hipLaunchKernelGGL(adam_kernel_scalar, dim3((unsigned)blocks), dim3(256), 0, st, p, g, m, v, (long long)n, c, c_dev);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

int launch_adam_table(float* p, float* m, float* v, int32_t n_rows, int32_t cols, const int32_t* row_slot,
                      const float* row_grad, int64_t ld_rg, const AdamC& c, const AdamC* c_dev, hipStream_t st) {
  // one 4-row group per workgroup (no grid-stride loop): the hardware dispatcher balances 16 K short workgroups
  // better than 4 K persistent ones -- measured in bench.py on one box: 6.00 vs 5.33 TB/s
  // (round 4: a persistent, smaller grid -- 4096 / 2048 / 1024 / 512 workgroups, so that the update leaves HBM bandwidth to the
  // 7 x 7 block's backward it runs beside -- did not move the step: 11.59-11.67 ms at every size, profiles/r04_table_grid_ab.txt)
  const int blocks = (n_rows + 3) / 4;
  const bool vec = (cols % 4 == 0) && (ld_rg % 4 == 0) && al16(p) && al16(m) && al16(v) && al16(row_grad);
  if (vec)
    hipLaunchKernelGGL((adam_table_kernel<true>), dim3(blocks), dim3(256), 0, st, p, m, v, n_rows, cols, row_slot,
                       row_grad, (long long)ld_rg, c, c_dev);
  else
    hipLaunchKernelGGL((adam_table_kernel<false>), dim3(blocks), dim3(256), 0, st, p, m, v, n_rows, cols, row_slot,
                       row_grad, (long long)ld_rg, c, c_dev);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}
}  // namespace

extern "C" int mcl_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1,
                             double beta2, double eps, double weight_decay, double bc1, double bc2,
                             mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!p || !g || !m || !v || n <= 0 || bc1 <= 0. || bc2 <= 0.) return MCL_EINVAL;
  return launch_adam(p, g, m, v, n, make_consts(lr, beta1, beta2, eps, weight_decay, bc1, bc2), nullptr,
                     mcl_stream(stream));
}

extern "C" int mcl_adam_table_step(float* p, float* m, float* v, int32_t n_rows, int32_t cols,
                                   const int32_t* row_slot, const float* row_grad, int64_t ld_rg, double lr,
                                   double beta1, double beta2, double eps, double weight_decay, double bc1,
                                   double bc2, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!p || !m || !v || !row_slot || !row_grad || n_rows <= 0 || cols <= 0 || bc1 <= 0. || bc2 <= 0.)
    return MCL_EINVAL;
  return launch_adam_table(p, m, v, n_rows, cols, row_slot, row_grad, ld_rg,
                           make_consts(lr, beta1, beta2, eps, weight_decay, bc1, bc2), nullptr, mcl_stream(stream));
}

// ---- graph-replayable form: the step counter and the derived constants live on the device
extern "C" int mcl_adam_consts_update(int64_t* step, float* consts /* 8 floats */, const double* hyper /* 5 doubles */,
                                      mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!step || !consts || !hyper) return MCL_EINVAL;
  hipLaunchKernelGGL(adam_consts_kernel, dim3(1), dim3(64), 0, mcl_stream(stream), (long long*)step,
                     reinterpret_cast<AdamC*>(consts), hyper, (AdamC*)nullptr, 0);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_adam_consts_update_hist(int64_t* step, float* consts, const double* hyper, float* hist,
                                           int32_t hist_len, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!step || !consts || !hyper || !hist || hist_len < 2 || (hist_len & (hist_len - 1))) return MCL_EINVAL;
  hipLaunchKernelGGL(adam_consts_kernel, dim3(1), dim3(64), 0, mcl_stream(stream), (long long*)step,
                     reinterpret_cast<AdamC*>(consts), hyper, reinterpret_cast<AdamC*>(hist), hist_len - 1);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_adam_table_lazy(float* p0, float* m0, float* v0, int32_t* row_step0, float* p1, float* m1, float* v1,
                                   int32_t* row_step1, int32_t n_rows, int32_t cols, const float* pos,
                                   const int32_t* owner0, const int32_t* owner1, int32_t n_owner, const float* row_grad0,
                                   const float* row_grad1, int64_t ld_rg, const int64_t* step, const float* hist,
                                   int32_t hist_len, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!p0 || !m0 || !v0 || !row_step0 || n_rows <= 0 || cols <= 0 || n_owner <= 0 || !step || !hist || hist_len < 2 ||
      (hist_len & (hist_len - 1)))
    return MCL_EINVAL;
  const bool two = p1 != nullptr;
  if (two && (!m1 || !v1 || !row_step1)) return MCL_EINVAL;
  if (row_grad0 && !owner0) return MCL_EINVAL;                                                  // a gradient needs its owner list
  if (row_grad0 && (pos || (two && (!row_grad1 || !owner1)))) return MCL_EINVAL;
  if (!pos && !owner0 && n_owner != n_rows) return MCL_EINVAL;                                  // materialise: every row
  LazyTable t0{p0, m0, v0, row_step0, owner0, row_grad0};
  LazyTable t1{p1, m1, v1, row_step1, owner1, row_grad1};
  const bool vec = (cols % 4 == 0) && (ld_rg % 4 == 0) && al16(p0) && al16(m0) && al16(v0) && al16(row_grad0) &&
                   (!two || (al16(p1) && al16(m1) && al16(v1) && al16(row_grad1)));
  const unsigned grid = (unsigned)n_owner * (two ? 2u : 1u);
  if (vec)
    hipLaunchKernelGGL((adam_table_lazy_kernel<true>), dim3(grid), dim3(256), 0, mcl_stream(stream), t0, t1, n_owner,
                       n_rows, cols, pos, (long long)ld_rg, (const long long*)step, reinterpret_cast<const AdamC*>(hist),
                       hist_len - 1);
  else
    hipLaunchKernelGGL((adam_table_lazy_kernel<false>), dim3(grid), dim3(256), 0, mcl_stream(stream), t0, t1, n_owner,
                       n_rows, cols, pos, (long long)ld_rg, (const long long*)step, reinterpret_cast<const AdamC*>(hist),
                       hist_len - 1);
  MCL_CHECK_LAUNCH();
  return MCL_OK;
}

extern "C" int mcl_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* consts,
                                 mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!p || !g || !m || !v || !consts || n <= 0) return MCL_EINVAL;
  return launch_adam(p, g, m, v, n, AdamC{}, reinterpret_cast<const AdamC*>(consts), mcl_stream(stream));
}

extern "C" int mcl_adam_step_dev_shadow(float* p, const float* g, float* m, float* v, int64_t n, const float* consts,
                                        void* shadow_bf16, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!p || !g || !m || !v || !consts || !shadow_bf16 || n <= 0) return MCL_EINVAL;
  return launch_adam(p, g, m, v, n, AdamC{}, reinterpret_cast<const AdamC*>(consts), mcl_stream(stream),
                     reinterpret_cast<unsigned short*>(shadow_bf16));
}

extern "C" int mcl_adam_table_step_dev(float* p, float* m, float* v, int32_t n_rows, int32_t cols,
                                       const int32_t* row_slot, const float* row_grad, int64_t ld_rg,
                                       const float* consts, mcl_stream_t stream) {
  MCL_CLEAR_ERROR();
  if (!p || !m || !v || !row_slot || !row_grad || !consts || n_rows <= 0 || cols <= 0) return MCL_EINVAL;
  return launch_adam_table(p, m, v, n_rows, cols, row_slot, row_grad, ld_rg, AdamC{},
                           reinterpret_cast<const AdamC*>(consts), mcl_stream(stream));
}
