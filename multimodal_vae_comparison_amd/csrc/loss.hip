// Reconstruction losses and ELBO assembly for gfx950.  HBM-bound streaming kernels: 16-byte loads,
// one wave-shuffle + LDS reduction per row, no atomics.
#include "common.hpp"

#define BCE_ETA 1e-6f

__device__ __forceinline__ float bce_term(float x, float t) {
  // torch.nn.functional.binary_cross_entropy clamps each log at -100 (ReconLoss.bce, objectives.py:405)
  // (v_log_f32: ~1e-7 absolute error per term, 1e-7 relative on a row sum; logf costs ~20 VALU instructions more per call)
  float lx = fmaxf(__logf(x), -100.0f), l1x = fmaxf(__logf(1.0f - x), -100.0f);
  return -(t * lx + (1.0f - t) * l1x);
}

// one 256-thread block per sample row
// d(row)/d(logit) for x_hat = clamp(sigmoid(logit)): g (x_hat - t) where the clamp is inactive
__device__ __forceinline__ float bce_dlogit(float x, float t, float g) {
  return (x > BCE_ETA && x < 1.0f - BCE_ETA) ? g * (x - t) : 0.f;
}
// SEEDED: the upstream gradient of every row is the known constant `seed` (the ELBO weight of this term), so the
// logit gradient is written by the same pass over x_hat and target -- backward of the loss costs no launch.
// trows: the target has that many rows and output row b is paired with target row b % trows (a K-sample decoder output
// against the target repeated K times, BaseObjective.reshape_for_loss, objectives.py:118-120 -- never materialised)
template <bool SEEDED>
__global__ __launch_bounds__(256) void bce_rowsum_kernel(const float* __restrict__ xh, const float* __restrict__ tg,
                                                         float* __restrict__ row, int F, float seed,
                                                         float* __restrict__ dl, int trows) {
  __shared__ float red[4];
  MMVAE_TRACE_STAMP(28);
  const size_t base = (size_t)blockIdx.x * F;
  const size_t tbase = (size_t)(blockIdx.x % trows) * F;
  float acc = 0.f;
  if ((F & 3) == 0) {
    const float4* x4 = reinterpret_cast<const float4*>(xh + base);
    const float4* t4 = reinterpret_cast<const float4*>(tg + tbase);
    float4* d4 = reinterpret_cast<float4*>(dl + base);
    const int n4 = F / 4;
    int i = threadIdx.x;
    for (; i + 3 * 256 < n4; i += 4 * 256) {       // 8 independent 16-byte loads in flight per thread
      float4 x[4], t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        x[u] = x4[i + u * 256];
        t[u] = t4[i + u * 256];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc += bce_term(x[u].x, t[u].x) + bce_term(x[u].y, t[u].y) + bce_term(x[u].z, t[u].z) + bce_term(x[u].w, t[u].w);
        if (SEEDED)
          d4[i + u * 256] = make_float4(bce_dlogit(x[u].x, t[u].x, seed), bce_dlogit(x[u].y, t[u].y, seed),
                                        bce_dlogit(x[u].z, t[u].z, seed), bce_dlogit(x[u].w, t[u].w, seed));
      }
    }
    for (; i < n4; i += 256) {
      float4 x = x4[i], t = t4[i];
      acc += bce_term(x.x, t.x) + bce_term(x.y, t.y) + bce_term(x.z, t.z) + bce_term(x.w, t.w);
      if (SEEDED)
        d4[i] = make_float4(bce_dlogit(x.x, t.x, seed), bce_dlogit(x.y, t.y, seed), bce_dlogit(x.z, t.z, seed),
                            bce_dlogit(x.w, t.w, seed));
    }
  } else {
    for (int i = threadIdx.x; i < F; i += 256) {
      const float x = xh[base + i], t = tg[tbase + i];
      acc += bce_term(x, t);
      if (SEEDED) dl[base + i] = bce_dlogit(x, t, seed);
    }
  }
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) row[blockIdx.x] = acc;
}

__global__ __launch_bounds__(256) void bce_elem_kernel(const float* __restrict__ xh, const float* __restrict__ tg,
                                                       float* __restrict__ out, long n) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long stride = (long)gridDim.x * 256;
  for (; i < n; i += stride) out[i] = bce_term(xh[i], tg[i]);
}

// d/dlogit of bce(clamp(sigmoid(logit))) = (xh - t) where the clamp is inactive, else 0
// (rows on grid.x, column chunks on grid.y: a K-sample batch has M K B rows, more than grid.y's 65535)
__global__ __launch_bounds__(256) void bce_bwd_kernel(const float* __restrict__ xh, const float* __restrict__ tg,
                                                      const float* __restrict__ grow, float* __restrict__ dl, int F,
                                                      int trows) {
  const size_t base = (size_t)blockIdx.x * F;
  const size_t tbase = (size_t)(blockIdx.x % trows) * F;
  const float g = grow[blockIdx.x];
  const int i = (blockIdx.y * 256 + threadIdx.x) * 4;
  if (i + 3 < F && (F & 3) == 0) {
    float4 x = *reinterpret_cast<const float4*>(xh + base + i);
    float4 t = *reinterpret_cast<const float4*>(tg + tbase + i);
    float4 o;
    o.x = (x.x > BCE_ETA && x.x < 1.0f - BCE_ETA) ? g * (x.x - t.x) : 0.f;
    o.y = (x.y > BCE_ETA && x.y < 1.0f - BCE_ETA) ? g * (x.y - t.y) : 0.f;
    o.z = (x.z > BCE_ETA && x.z < 1.0f - BCE_ETA) ? g * (x.z - t.z) : 0.f;
    o.w = (x.w > BCE_ETA && x.w < 1.0f - BCE_ETA) ? g * (x.w - t.w) : 0.f;
    *reinterpret_cast<float4*>(dl + base + i) = o;
  } else {
    for (int j = i; j < F && j < i + 4; ++j) {
      float x = xh[base + j];
      dl[base + j] = (x > BCE_ETA && x < 1.0f - BCE_ETA) ? g * (x - tg[tbase + j]) : 0.f;
    }
  }
}

// true gradient wrt x_hat (torch: grad * (x - t) / max((1 - x) * x, 1e-12))
__global__ __launch_bounds__(256) void bce_rowsum_bwd_kernel(const float* __restrict__ xh, const float* __restrict__ tg,
                                                             const float* __restrict__ grow, float* __restrict__ dx,
                                                             int F) {
  const size_t base = (size_t)blockIdx.x * F;
  const float g = grow[blockIdx.x];
  for (int j = blockIdx.y * 256 + threadIdx.x; j < F; j += gridDim.y * 256) {
    const float x = xh[base + j];
    dx[base + j] = g * (x - tg[base + j]) / fmaxf((1.0f - x) * x, 1e-12f);
  }
}
__global__ __launch_bounds__(256) void sigmoid_clamp_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                                float* __restrict__ dl, long n) {
  const long gs = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += gs) {
    const float v = y[i];
    dl[i] = (v > BCE_ETA && v < 1.0f - BCE_ETA) ? dy[i] * v * (1.0f - v) : 0.f;
  }
}
extern "C" int mmvae_bce_rowsum_bwd(const float* x_hat, const float* target, const float* g_row, float* dxhat, int B,
                                    int F, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(x_hat && target && g_row && dxhat && B > 0 && F > 0);
  int bx = (F + 255) / 256;
  if (bx > 16) bx = 16;
  hipLaunchKernelGGL(bce_rowsum_bwd_kernel, dim3(B, bx), dim3(256), 0, (hipStream_t)stream, x_hat, target, g_row, dxhat,
                     F);
  return mmvae_launch_status();
}
extern "C" int mmvae_sigmoid_clamp_bwd(const float* dy, const float* y, float* dl, long n, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(dy && y && dl && n > 0);
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(sigmoid_clamp_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dy, y, dl, n);
  return mmvae_launch_status();
}

extern "C" int mmvae_bce_rowsum_fwd(const float* x_hat, const float* target, float* row_loss, int B, int F,
                                    int target_rows, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(x_hat && target && row_loss && B > 0 && F > 0 && target_rows > 0);
  hipLaunchKernelGGL(bce_rowsum_kernel<false>, dim3(B), dim3(256), 0, (hipStream_t)stream, x_hat, target, row_loss, F,
                     0.f, nullptr, target_rows);
  return mmvae_launch_status();
}
extern "C" int mmvae_bce_rowsum_seeded(const float* x_hat, const float* target, float* row_loss, float seed,
                                       float* dlogit, int B, int F, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(x_hat && target && row_loss && dlogit && B > 0 && F > 0);
  if ((F & 3) == 0 && (((uintptr_t)dlogit) & 15) != 0) return MMVAE_ERR_ARG;
  hipLaunchKernelGGL(bce_rowsum_kernel<true>, dim3(B), dim3(256), 0, (hipStream_t)stream, x_hat, target, row_loss, F,
                     seed, dlogit, B);
  return mmvae_launch_status();
}
extern "C" int mmvae_bce_elem_fwd(const float* x_hat, const float* target, float* loss, long n,
                                  mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(x_hat && target && loss && n > 0);
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(bce_elem_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x_hat, target, loss,
                     n);
  return mmvae_launch_status();
}
extern "C" int mmvae_bce_sigmoid_clamp_bwd(const float* x_hat, const float* target, const float* g_row,
                                           float* dlogit, int B, int F, int target_rows, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(x_hat && target && g_row && dlogit && B > 0 && F > 0 && target_rows > 0);
  hipLaunchKernelGGL(bce_bwd_kernel, dim3(B, (F + 1023) / 1024), dim3(256), 0, (hipStream_t)stream, x_hat, target,
                     g_row, dlogit, F, target_rows);
  return mmvae_launch_status();
}

// ---------------------------------------------------------------------------------------------
// lprob (objectives.py:409-424): -log p(t) under Normal / Laplace(loc, scale); elements in fp32 as torch.distributions
// computes them, row sums accumulated in fp64 (the reference casts the elements to double before summing), NaN
// elements count as 0 and carry no gradient.  scale <= 0 selects the reference's masked-modality quirk scale := loc.
// ---------------------------------------------------------------------------------------------
#define HALF_LOG_2PI 0.9189385332046727f
__device__ __forceinline__ float lprob_logp(float x, float t, float s, int laplace) {
  if (laplace) return -logf(2.0f * s) - fabsf(t - x) / s;
  const float d = t - x;
  return -(d * d) / (2.0f * (s * s)) - logf(s) - HALF_LOG_2PI;
}
__global__ __launch_bounds__(256) void lprob_rowsum_kernel(const float* __restrict__ loc, const float* __restrict__ tg,
                                                           float* __restrict__ row, int F, int trows, float scale,
                                                           int laplace, int lap_rows, int perm_c) {
  __shared__ double red[4];
  if (lap_rows > 0) laplace = (laplace >> (blockIdx.x / lap_rows)) & 1;
  const size_t base = (size_t)blockIdx.x * F;
  const size_t tbase = (size_t)(blockIdx.x % trows) * F;
  const int hw = perm_c > 0 ? F / perm_c : 0;
  double acc = 0.0;
  for (int i = threadIdx.x; i < F; i += 256) {
    const float x = loc[base + (perm_c > 0 ? (i % perm_c) * hw + i / perm_c : i)];
    const float lp = lprob_logp(x, tg[tbase + i], scale > 0.f ? scale : x, laplace);
    if (lp == lp) acc -= (double)lp;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) row[blockIdx.x] = (float)(red[0] + red[1] + red[2] + red[3]);
}
__global__ __launch_bounds__(256) void lprob_bwd_kernel(const float* __restrict__ loc, const float* __restrict__ tg,
                                                        const float* __restrict__ grow, float* __restrict__ dl, int F,
                                                        int trows, float scale, int laplace, int lap_rows, int perm_c,
                                                        int logit_grad) {
  // rows on grid.x (M K B of them with K samples: more than grid.y's 65535), column chunks on grid.y
  if (lap_rows > 0) laplace = (laplace >> (blockIdx.x / lap_rows)) & 1;
  const size_t base = (size_t)blockIdx.x * F;
  const size_t tbase = (size_t)(blockIdx.x % trows) * F;
  const float g = grow[blockIdx.x];
  const int i = blockIdx.y * 256 + threadIdx.x;     // index into loc's memory: coalesced loads / stores there
  if (i >= F) return;
  // the target element paired with loc element i = (c, p) of a (perm_c, F / perm_c) plane set is j = p * perm_c + c
  const int j = perm_c > 0 ? (i % (F / perm_c)) * perm_c + i / (F / perm_c) : i;
  const float x = loc[base + i], t = tg[tbase + j], d = t - x;
  const bool own = !(scale > 0.f);
  const float s = own ? x : scale;
  const float lp = lprob_logp(x, t, s, laplace);
  float v;   // d(-log p)/d loc
  if (laplace) {
    const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    v = -sg / s;
    if (own) v += 1.0f / s - fabsf(d) / (s * s);
  } else {
    v = -d / (s * s);
    if (own) v += 1.0f / s - (d * d) / (s * s * s);
  }
  if (logit_grad) v *= x * (1.0f - x);
  dl[base + i] = (lp == lp && v == v) ? g * v : 0.f;
}
extern "C" int mmvae_lprob_rowsum_fwd(const float* loc, const float* target, float* row_loss, int B, int F,
                                      int target_rows, float scale, int laplace, int lap_block_rows, int perm_c,
                                      mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(loc && target && row_loss && B > 0 && F > 0 && target_rows > 0 && lap_block_rows >= 0);
  MMVAE_CHECK_ARG(perm_c >= 0 && (perm_c == 0 || F % perm_c == 0));
  hipLaunchKernelGGL(lprob_rowsum_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, loc, target, row_loss, F,
                     target_rows, scale, laplace, lap_block_rows, perm_c);
  return mmvae_launch_status();
}
extern "C" int mmvae_lprob_rowsum_bwd(const float* loc, const float* target, const float* g_row, float* dloc, int B,
                                      int F, int target_rows, float scale, int laplace, int lap_block_rows, int perm_c,
                                      int logit_grad, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(loc && target && g_row && dloc && B > 0 && F > 0 && target_rows > 0 && lap_block_rows >= 0);
  MMVAE_CHECK_ARG(perm_c >= 0 && (perm_c == 0 || F % perm_c == 0));
  hipLaunchKernelGGL(lprob_bwd_kernel, dim3(B, (F + 255) / 256), dim3(256), 0, (hipStream_t)stream, loc, target, g_row,
                     dloc, F, target_rows, scale, laplace, lap_block_rows, perm_c, logit_grad);
  return mmvae_launch_status();
}

// ---------------------------------------------------------------------------------------------
// optimal_sigma (objectives.py:503-509, sigma-VAE): log_sigma = softclip(log sqrt(mean_all (t-x)^2), -6), ONE scalar
// per call; loss[b,f] = detach(((t-x)/sigma)^2) + log_sigma + log sqrt(2 pi).  The only gradient path is log_sigma.
// stats = {mean square, log_sigma, raw log sigma}.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sqerr_partial_kernel(const float* __restrict__ loc, const float* __restrict__ tg,
                                                            float* __restrict__ ws, long n) {
  __shared__ float red[4];
  float acc = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float d = tg[i] - loc[i];
    acc += d * d;
  }
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) ws[blockIdx.x] = acc;
}
__device__ __forceinline__ void optsig_stats(const float* __restrict__ ws, int nparts, long n, float* red, float* msq,
                                             float* ls_raw, float* log_sigma) {
  float a = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) a += ws[i];
  a = block_sum_256(a, red);
  const float m = a / (float)n;
  const float r = 0.5f * logf(m);                       // log sqrt(mean)
  const float y = r + 6.0f;
  *msq = m;
  *ls_raw = r;
  *log_sigma = -6.0f + (y > 20.0f ? y : log1pf(expf(y)));   // utils.softclip: min + softplus(x - min)
}
__global__ __launch_bounds__(256) void optsig_rows_kernel(const float* __restrict__ loc, const float* __restrict__ tg,
                                                          const float* __restrict__ ws, int nparts,
                                                          float* __restrict__ row, float* __restrict__ stats, int B,
                                                          int F) {
  __shared__ float red[4];
  float msq, ls_raw, ls;
  optsig_stats(ws, nparts, (long)B * F, red, &msq, &ls_raw, &ls);
  const float inv = expf(-ls);
  const size_t base = (size_t)blockIdx.x * F;
  float acc = 0.f;
  for (int i = threadIdx.x; i < F; i += 256) {
    const float q = (tg[base + i] - loc[base + i]) * inv;
    acc += q * q;
  }
  __syncthreads();
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) {
    row[blockIdx.x] = acc + (float)F * (ls + HALF_LOG_2PI);
    if (blockIdx.x == 0) {
      stats[0] = msq;
      stats[1] = ls;
      stats[2] = ls_raw;
    }
  }
}
__global__ __launch_bounds__(256) void optsig_bwd_kernel(const float* __restrict__ loc, const float* __restrict__ tg,
                                                         const float* __restrict__ grow, const float* __restrict__ stats,
                                                         float* __restrict__ dl, int B, int F) {
  __shared__ float red[4];
  float gs = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) gs += grow[b];
  gs = block_sum_256(gs, red);
  const long n = (long)B * F;
  // d loss / d log_sigma = F * sum_b g_b;  d log_sigma / d raw = sigmoid(raw + 6);  d raw / d x_i = -(t_i - x_i) / (n * msq)
  const float coef = -(gs * (float)F) * dev_sigmoid(stats[2] + 6.0f) / ((float)n * stats[0]);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dl[i] = coef * (tg[i] - loc[i]);
}
static inline int optsig_parts(long n) {
  long p = (n + 4095) / 4096;
  return (int)(p < 1 ? 1 : (p > 1024 ? 1024 : p));
}
extern "C" size_t mmvae_optimal_sigma_ws_floats(int B, int F) { return (size_t)optsig_parts((long)B * F); }
extern "C" int mmvae_optimal_sigma_fwd(const float* loc, const float* target, float* row_loss, float* stats, float* ws,
                                       int B, int F, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(loc && target && row_loss && stats && ws && B > 0 && F > 0);
  const long n = (long)B * F;
  const int parts = optsig_parts(n);
  hipLaunchKernelGGL(sqerr_partial_kernel, dim3(parts), dim3(256), 0, (hipStream_t)stream, loc, target, ws, n);
  hipLaunchKernelGGL(optsig_rows_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, loc, target, ws, parts, row_loss,
                     stats, B, F);
  return mmvae_launch_status();
}
extern "C" int mmvae_optimal_sigma_bwd(const float* loc, const float* target, const float* g_row, const float* stats,
                                       float* dloc, int B, int F, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(loc && target && g_row && stats && dloc && B > 0 && F > 0);
  const long n = (long)B * F;
  long blocks = (n + 1023) / 1024;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(optsig_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, loc, target, g_row,
                     stats, dloc, B, F);
  return mmvae_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Element-wise forms behind the loss-plugin contract ReconLoss.<name>(output, target, bs) -> (bs, -1)
// (objectives.py:389-509); the mixers' objective() uses the row-sum kernels above.
//   lprob (:409-424): -log p(t) per element, fp32 as torch.distributions computes it, THEN cast to double, NaN -> 0
//   optimal_sigma (:503-509): detach(((t - x) / sigma)^2) + log_sigma + log sqrt(2 pi), one log_sigma per call
//   l1 (:427-442), mse (:444-459): |x - t|, (x - t)^2
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lprob_elem_kernel(const float* __restrict__ loc, const float* __restrict__ tg,
                                                         double* __restrict__ out, long n, long tn, float scale,
                                                         int laplace) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float x = loc[i];
    const float lp = lprob_logp(x, tg[i % tn], scale > 0.f ? scale : x, laplace);
    out[i] = lp == lp ? -(double)lp : 0.0;
  }
}
__global__ __launch_bounds__(256) void lprob_elem_bwd_kernel(const float* __restrict__ loc, const float* __restrict__ tg,
                                                             const double* __restrict__ g, float* __restrict__ dl,
                                                             long n, long tn, float scale, int laplace) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float x = loc[i], t = tg[i % tn], d = t - x;
    const bool own = !(scale > 0.f);
    const float s = own ? x : scale;
    const float lp = lprob_logp(x, t, s, laplace);
    float v;
    if (laplace) {
      const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
      v = -sg / s;
      if (own) v += 1.0f / s - fabsf(d) / (s * s);
    } else {
      v = -d / (s * s);
      if (own) v += 1.0f / s - (d * d) / (s * s * s);
    }
    dl[i] = (lp == lp && v == v) ? (float)(g[i] * (double)v) : 0.f;
  }
}
extern "C" int mmvae_lprob_elem_fwd(const float* loc, const float* target, double* out, long n, long target_n,
                                    float scale, int laplace, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(loc && target && out && n > 0 && target_n > 0 && n % target_n == 0);
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(lprob_elem_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, loc, target, out, n,
                     target_n, scale, laplace);
  return mmvae_launch_status();
}
extern "C" int mmvae_lprob_elem_bwd(const float* loc, const float* target, const double* g, float* dloc, long n,
                                    long target_n, float scale, int laplace, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(loc && target && g && dloc && n > 0 && target_n > 0 && n % target_n == 0);
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(lprob_elem_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, loc, target, g,
                     dloc, n, target_n, scale, laplace);
  return mmvae_launch_status();
}

__global__ __launch_bounds__(256) void optsig_elem_kernel(const float* __restrict__ loc, const float* __restrict__ tg,
                                                          const float* __restrict__ ws, int nparts,
                                                          float* __restrict__ out, float* __restrict__ stats, long n) {
  __shared__ float red[4];
  float msq, ls_raw, ls;
  optsig_stats(ws, nparts, n, red, &msq, &ls_raw, &ls);
  const float inv = expf(-ls), c = ls + HALF_LOG_2PI;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float q = (tg[i] - loc[i]) * inv;
    out[i] = q * q + c;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    stats[0] = msq;
    stats[1] = ls;
    stats[2] = ls_raw;
  }
}
__global__ __launch_bounds__(256) void sum_partial_kernel(const float* __restrict__ g, float* __restrict__ ws, long n) {
  __shared__ float red[4];
  float acc = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) acc += g[i];
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) ws[blockIdx.x] = acc;
}
// the only gradient path is log_sigma: d out_i / d log_sigma = 1 for every element
__global__ __launch_bounds__(256) void optsig_elem_bwd_kernel(const float* __restrict__ loc, const float* __restrict__ tg,
                                                              const float* __restrict__ ws, int nparts,
                                                              const float* __restrict__ stats, float* __restrict__ dl,
                                                              long n) {
  __shared__ float red[4];
  float gs = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) gs += ws[i];
  gs = block_sum_256(gs, red);
  const float coef = -gs * dev_sigmoid(stats[2] + 6.0f) / ((float)n * stats[0]);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dl[i] = coef * (tg[i] - loc[i]);
}
extern "C" int mmvae_optimal_sigma_elem_fwd(const float* loc, const float* target, float* out, float* stats, float* ws,
                                            long n, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(loc && target && out && stats && ws && n > 0);
  const int parts = optsig_parts(n);
  long blocks = (n + 1023) / 1024;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(sqerr_partial_kernel, dim3(parts), dim3(256), 0, (hipStream_t)stream, loc, target, ws, n);
  hipLaunchKernelGGL(optsig_elem_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, loc, target, ws,
                     parts, out, stats, n);
  return mmvae_launch_status();
}
extern "C" int mmvae_optimal_sigma_elem_bwd(const float* loc, const float* target, const float* g, const float* stats,
                                            float* ws, float* dloc, long n, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(loc && target && g && stats && ws && dloc && n > 0);
  const int parts = optsig_parts(n);
  long blocks = (n + 1023) / 1024;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(sum_partial_kernel, dim3(parts), dim3(256), 0, (hipStream_t)stream, g, ws, n);
  hipLaunchKernelGGL(optsig_elem_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, loc, target, ws,
                     parts, stats, dloc, n);
  return mmvae_launch_status();
}

// l1 / mse: kind 0 = |x - t|, 1 = (x - t)^2.  Row sums (the objective) and elements (the plugin contract).
__device__ __forceinline__ float pw_term(float x, float t, int kind) {
  const float d = x - t;
  return kind ? d * d : fabsf(d);
}
__device__ __forceinline__ float pw_grad(float x, float t, int kind) {
  const float d = x - t;
  return kind ? 2.0f * d : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));      // torch: sign(0) = 0
}
__global__ __launch_bounds__(256) void pw_rowsum_kernel(const float* __restrict__ x, const float* __restrict__ tg,
                                                        float* __restrict__ row, int F, int trows, int kind) {
  __shared__ float red[4];
  const size_t base = (size_t)blockIdx.x * F, tbase = (size_t)(blockIdx.x % trows) * F;
  float acc = 0.f;
  for (int i = threadIdx.x; i < F; i += 256) acc += pw_term(x[base + i], tg[tbase + i], kind);
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) row[blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void pw_rowsum_bwd_kernel(const float* __restrict__ x, const float* __restrict__ tg,
                                                            const float* __restrict__ grow, float* __restrict__ dx,
                                                            int F, int trows, int kind) {
  const size_t base = (size_t)blockIdx.x * F, tbase = (size_t)(blockIdx.x % trows) * F;
  const float g = grow[blockIdx.x];
  for (int i = blockIdx.y * 256 + threadIdx.x; i < F; i += gridDim.y * 256)
    dx[base + i] = g * pw_grad(x[base + i], tg[tbase + i], kind);
}
__global__ __launch_bounds__(256) void pw_elem_kernel(const float* __restrict__ x, const float* __restrict__ tg,
                                                      const float* __restrict__ g, float* __restrict__ out, long n,
                                                      int kind) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    out[i] = g ? g[i] * pw_grad(x[i], tg[i], kind) : pw_term(x[i], tg[i], kind);
}
extern "C" int mmvae_pointwise_rowsum_fwd(const float* x, const float* target, float* row_loss, int B, int F,
                                          int target_rows, int kind, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(x && target && row_loss && B > 0 && F > 0 && target_rows > 0 && (kind == 0 || kind == 1));
  hipLaunchKernelGGL(pw_rowsum_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, target, row_loss, F, target_rows,
                     kind);
  return mmvae_launch_status();
}
extern "C" int mmvae_pointwise_rowsum_bwd(const float* x, const float* target, const float* g_row, float* dx, int B,
                                          int F, int target_rows, int kind, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(x && target && g_row && dx && B > 0 && F > 0 && target_rows > 0 && (kind == 0 || kind == 1));
  int by = (F + 255) / 256;
  if (by > 16) by = 16;
  hipLaunchKernelGGL(pw_rowsum_bwd_kernel, dim3(B, by), dim3(256), 0, (hipStream_t)stream, x, target, g_row, dx, F,
                     target_rows, kind);
  return mmvae_launch_status();
}
/* g == NULL: out = loss elements; g != NULL: out = g * d loss / d x */
extern "C" int mmvae_pointwise_elem(const float* x, const float* target, const float* g, float* out, long n, int kind,
                                    mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(x && target && out && n > 0 && (kind == 0 || kind == 1));
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(pw_elem_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, target, g, out, n,
                     kind);
  return mmvae_launch_status();
}

// ---------------------------------------------------------------------------------------------
// category_ce: softmax over TIME (objectives.py:499-500).  One 64-thread block per
// sample; lanes over the vocabulary (coalesced rows of V floats), serial loop over T (<= a few hundred).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void ce_time_fwd_kernel(const float* __restrict__ lg, const float* __restrict__ tg,
                                                         float* __restrict__ loss, float* __restrict__ row, int T,
                                                         int V, int trows) {
  const int b = blockIdx.x;
  const float* L = lg + (size_t)b * T * V;
  const float* Tg = tg + (size_t)(b % trows) * T * V;
  float rsum = 0.f;
  for (int v = threadIdx.x; v < V; v += 64) {
    float mx = -INFINITY;
    for (int t = 0; t < T; ++t) mx = fmaxf(mx, L[t * V + v]);
    float se = 0.f, dot = 0.f, ts = 0.f;
    for (int t = 0; t < T; ++t) {
      float l = L[t * V + v], g = Tg[t * V + v];
      se += expf(l - mx);
      dot += g * l;
      ts += g;
    }
    float lse = mx + logf(se);
    float ls = lse * ts - dot;
    if (loss) loss[(size_t)b * V + v] = ls;
    rsum += ls;
  }
  rsum = wave_sum(rsum);
  if (row && threadIdx.x == 0) row[b] = rsum;
}

__global__ __launch_bounds__(64) void ce_time_bwd_kernel(const float* __restrict__ lg, const float* __restrict__ tg,
                                                         const float* __restrict__ g, const float* __restrict__ grow,
                                                         float* __restrict__ dl, int T, int V, int trows) {
  const int b = blockIdx.x;
  const float* L = lg + (size_t)b * T * V;
  const float* Tg = tg + (size_t)(b % trows) * T * V;
  float* D = dl + (size_t)b * T * V;
  for (int v = threadIdx.x; v < V; v += 64) {
    const float gv = g ? g[(size_t)b * V + v] : grow[b];
    float mx = -INFINITY;
    for (int t = 0; t < T; ++t) mx = fmaxf(mx, L[t * V + v]);
    float se = 0.f, ts = 0.f;
    for (int t = 0; t < T; ++t) {
      se += expf(L[t * V + v] - mx);
      ts += Tg[t * V + v];
    }
    const float inv = 1.0f / se;
    for (int t = 0; t < T; ++t) D[t * V + v] = gv * (expf(L[t * V + v] - mx) * inv * ts - Tg[t * V + v]);
  }
}

// T*V <= CE_TILE (the text towers: 32 x 27): the sample's logits and targets are staged once with coalesced,
// independent loads (the per-column loops above chase T dependent global loads per pass: 25 us at B = 128), the
// column statistics come out of LDS, and in backward all 256 threads write the gradient tile.
#define CE_TILE 4096
__global__ __launch_bounds__(256) void ce_time_fwd_tile_kernel(const float* __restrict__ lg, const float* __restrict__ tg,
                                                              float* __restrict__ loss, float* __restrict__ row, int T,
                                                              int V, float seed, float* __restrict__ dl, int trows) {
  // dl != NULL: every row's upstream gradient is the constant `seed`; the logit gradient is written from the same
  // staged tile (see bce_rowsum_kernel<true>)
  __shared__ float sl[CE_TILE], st[CE_TILE];
  __shared__ float s_mx[256], s_k[256];
  __shared__ float red[4];
  MMVAE_TRACE_STAMP(29);
  const int b = blockIdx.x, n = T * V;
  const float* L = lg + (size_t)b * n;
  const float* Tg = tg + (size_t)(b % trows) * n;
  for (int e = threadIdx.x; e < n; e += 256) {
    sl[e] = L[e];
    st[e] = Tg[e];
  }
  __syncthreads();
  float rsum = 0.f;
  for (int v = threadIdx.x; v < V; v += 256) {
    float mx = -INFINITY;
    for (int t = 0; t < T; ++t) mx = fmaxf(mx, sl[t * V + v]);
    float se = 0.f, dot = 0.f, ts = 0.f;
    for (int t = 0; t < T; ++t) {
      const float l = sl[t * V + v], g = st[t * V + v];
      se += expf(l - mx);
      dot += g * l;
      ts += g;
    }
    const float ls = (mx + logf(se)) * ts - dot;
    if (loss) loss[(size_t)b * V + v] = ls;
    rsum += ls;
    if (dl) {
      s_mx[v] = mx;
      s_k[v] = ts / se;
    }
  }
  rsum = block_sum_256(rsum, red);
  if (row && threadIdx.x == 0) row[b] = rsum;
  if (dl) {     // (block_sum_256 synchronised the workgroup: s_mx / s_k are visible)
    float* D = dl + (size_t)b * n;
    for (int e = threadIdx.x; e < n; e += 256) {
      const int v = e % V;
      D[e] = seed * (expf(sl[e] - s_mx[v]) * s_k[v] - st[e]);
    }
  }
}
__global__ __launch_bounds__(256) void ce_time_bwd_tile_kernel(const float* __restrict__ lg, const float* __restrict__ tg,
                                                              const float* __restrict__ g, const float* __restrict__ grow,
                                                              float* __restrict__ dl, int T, int V, int trows) {
  __shared__ float sl[CE_TILE], st[CE_TILE];
  __shared__ float s_mx[256], s_k[256], s_gv[256];   // per column: max, ts / sum exp, upstream gradient (V <= 256)
  const int b = blockIdx.x, n = T * V;
  const float* L = lg + (size_t)b * n;
  const float* Tg = tg + (size_t)(b % trows) * n;
  for (int e = threadIdx.x; e < n; e += 256) {
    sl[e] = L[e];
    st[e] = Tg[e];
  }
  __syncthreads();
  for (int v = threadIdx.x; v < V; v += 256) {
    float mx = -INFINITY;
    for (int t = 0; t < T; ++t) mx = fmaxf(mx, sl[t * V + v]);
    float se = 0.f, ts = 0.f;
    for (int t = 0; t < T; ++t) {
      se += expf(sl[t * V + v] - mx);
      ts += st[t * V + v];
    }
    s_mx[v] = mx;
    s_k[v] = ts / se;
    s_gv[v] = g ? g[(size_t)b * V + v] : grow[b];
  }
  __syncthreads();
  float* D = dl + (size_t)b * n;
  for (int e = threadIdx.x; e < n; e += 256) {
    const int v = e % V;
    D[e] = s_gv[v] * (expf(sl[e] - s_mx[v]) * s_k[v] - st[e]);
  }
}

extern "C" int mmvae_ce_over_time_fwd(const float* logits, const float* target, float* loss, float* row_loss, int B,
                                      int T, int V, int target_rows, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(logits && target && (loss || row_loss) && B > 0 && T > 0 && V > 0 && target_rows > 0);
  if (T * V <= CE_TILE && V <= 256)
    hipLaunchKernelGGL(ce_time_fwd_tile_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, target, loss,
                       row_loss, T, V, 0.f, nullptr, target_rows);
  else
    hipLaunchKernelGGL(ce_time_fwd_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, logits, target, loss, row_loss, T,
                       V, target_rows);
  return mmvae_launch_status();
}
/* row sums + logit gradient for a constant upstream gradient `seed` in one launch; MMVAE_ERR_UNSUPPORTED when the
 * (T, V) tile does not fit the single-workgroup kernel (use fwd + bwd) */
extern "C" int mmvae_ce_over_time_seeded(const float* logits, const float* target, float* row_loss, float seed,
                                         float* dlogits, int B, int T, int V, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(logits && target && row_loss && dlogits && B > 0 && T > 0 && V > 0);
  if (!(T * V <= CE_TILE && V <= 256)) return MMVAE_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(ce_time_fwd_tile_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, target,
                     (float*)nullptr, row_loss, T, V, seed, dlogits, B);
  return mmvae_launch_status();
}
extern "C" int mmvae_ce_over_time_bwd(const float* logits, const float* target, const float* g, const float* g_row,
                                      float* dlogits, int B, int T, int V, int target_rows, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(logits && target && (g || g_row) && dlogits && B > 0 && T > 0 && V > 0 && target_rows > 0);
  if (T * V <= CE_TILE && V <= 256)
    hipLaunchKernelGGL(ce_time_bwd_tile_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, target, g, g_row,
                       dlogits, T, V, target_rows);
  else
    hipLaunchKernelGGL(ce_time_bwd_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, logits, target, g, g_row, dlogits,
                       T, V, target_rows);
  return mmvae_launch_status();
}

// ---------------------------------------------------------------------------------------------
// ELBO assembly: out[k] = sum_n W[k,n] * sum_b V[n,b]
// ---------------------------------------------------------------------------------------------
#define LC_MAX_ROWS 32
#define LC_MAX_OUT 4
struct lincomb_w {
  float w[LC_MAX_OUT * LC_MAX_ROWS];
};

__global__ __launch_bounds__(256) void lincomb_fwd_kernel(const float* __restrict__ V, lincomb_w W,
                                                          float* __restrict__ out, int n_rows, int B, int n_out) {
  __shared__ float red[4];
  __shared__ float rs[LC_MAX_ROWS];
  for (int n = 0; n < n_rows; ++n) {
    float a = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) a += V[(size_t)n * B + b];
    a = block_sum_256(a, red);
    if (threadIdx.x == 0) rs[n] = a;
  }
  __syncthreads();
  if (threadIdx.x < n_out) {
    float o = 0.f;
    for (int n = 0; n < n_rows; ++n) o += W.w[threadIdx.x * LC_MAX_ROWS + n] * rs[n];
    out[threadIdx.x] = o;
  }
}
__global__ __launch_bounds__(256) void lincomb_bwd_kernel(const float* __restrict__ gout, lincomb_w W,
                                                          float* __restrict__ dV, int n_rows, int B, int n_out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_rows * B) return;
  const int n = i / B;
  float g = 0.f;
  for (int k = 0; k < n_out; ++k) g += gout[k] * W.w[k * LC_MAX_ROWS + n];
  dV[i] = g;
}

static int pack_w(const float* W_host, int n_rows, int n_out, lincomb_w* w) {
  if (n_rows > LC_MAX_ROWS || n_out > LC_MAX_OUT) return MMVAE_ERR_UNSUPPORTED;
  for (int k = 0; k < n_out; ++k)
    for (int n = 0; n < n_rows; ++n) w->w[k * LC_MAX_ROWS + n] = W_host[k * n_rows + n];
  return MMVAE_OK;
}
extern "C" int mmvae_lincomb_rows_fwd(const float* V, const float* W_host, float* out, int n_rows, int B, int n_out,
                                      mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(V && W_host && out && n_rows > 0 && B > 0 && n_out > 0);
  lincomb_w w;
  int rc = pack_w(W_host, n_rows, n_out, &w);
  if (rc) return rc;
  hipLaunchKernelGGL(lincomb_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, V, w, out, n_rows, B, n_out);
  return mmvae_launch_status();
}
extern "C" int mmvae_lincomb_rows_bwd(const float* gout, const float* W_host, float* dV, int n_rows, int B, int n_out,
                                      mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(gout && W_host && dV && n_rows > 0 && B > 0 && n_out > 0);
  lincomb_w w;
  int rc = pack_w(W_host, n_rows, n_out, &w);
  if (rc) return rc;
  hipLaunchKernelGGL(lincomb_bwd_kernel, dim3((n_rows * B + 255) / 256), dim3(256), 0, (hipStream_t)stream, gout, w,
                     dV, n_rows, B, n_out);
  return mmvae_launch_status();
}

// Row-pointer forms: the rows live in separate tensors (one per loss term) and the upstream gradients of the n_out
// outputs arrive as separate scalars -- no cat / select / fill kernels around the two launches.
__global__ __launch_bounds__(256) void lincomb_rowptrs_fwd_kernel(mmvae_rowptrs_t rows, lincomb_w W,
                                                                  float* __restrict__ out, mmvae_rowptrs_t dunit,
                                                                  int has_dunit, int n_rows, int B, int n_out) {
  __shared__ float red[4];
  __shared__ float rs[LC_MAX_ROWS];
  for (int n = 0; n < n_rows; ++n) {
    const float* V = rows.p[n];
    float a = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) a += V[b];
    a = block_sum_256(a, red);
    if (threadIdx.x == 0) rs[n] = a;
    // gradient of output 0 with respect to this row for a unit upstream gradient: the constant W[0][n]
    if (has_dunit) {
      float* d = const_cast<float*>(dunit.p[n]);
      for (int b = threadIdx.x; b < B; b += 256) d[b] = W.w[n];
    }
  }
  __syncthreads();
  if (threadIdx.x < n_out) {
    float o = 0.f;
    for (int n = 0; n < n_rows; ++n) o += W.w[threadIdx.x * LC_MAX_ROWS + n] * rs[n];
    out[threadIdx.x] = o;
  }
}
__global__ __launch_bounds__(256) void lincomb_rowptrs_bwd_kernel(mmvae_gptrs_t g, lincomb_w W, mmvae_rowptrs_t drows,
                                                                  int n_rows, int B, int n_out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_rows * B) return;
  const int n = i / B, b = i - n * B;
  float v = 0.f;
  for (int k = 0; k < n_out; ++k)
    if (g.g[k]) v += g.g[k][0] * W.w[k * LC_MAX_ROWS + n];
  const_cast<float*>(drows.p[n])[b] = v;
}
extern "C" int mmvae_lincomb_rowptrs_fwd(const mmvae_rowptrs_t* rows, const float* W_host, float* out,
                                         const mmvae_rowptrs_t* d_unit, int n_rows, int B, int n_out,
                                         mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(rows && W_host && out && n_rows > 0 && B > 0 && n_out > 0);
  lincomb_w w;
  int rc = pack_w(W_host, n_rows, n_out, &w);
  if (rc) return rc;
  mmvae_rowptrs_t du;
  if (d_unit) du = *d_unit;
  hipLaunchKernelGGL(lincomb_rowptrs_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, *rows, w, out, du,
                     d_unit ? 1 : 0, n_rows, B, n_out);
  return mmvae_launch_status();
}
extern "C" int mmvae_lincomb_rowptrs_bwd(const mmvae_gptrs_t* gout, const float* W_host, const mmvae_rowptrs_t* drows,
                                         int n_rows, int B, int n_out, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(gout && W_host && drows && n_rows > 0 && B > 0 && n_out > 0);
  lincomb_w w;
  int rc = pack_w(W_host, n_rows, n_out, &w);
  if (rc) return rc;
  hipLaunchKernelGGL(lincomb_rowptrs_bwd_kernel, dim3((n_rows * B + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     *gout, w, *drows, n_rows, B, n_out);
  return mmvae_launch_status();
}

// ---------------------------------------------------------------------------------------------
// MoE ELBO assembly (models/mmvae_models.py:61-77 + BaseObjective.elbo objectives.py:54-67)
//   wc[b] = exp(lw[b]) * r[b]                                  (importance-weighted cross term)
//   loss  = ( sum_n w_n rowsum_n + n_nz * beta * sum kld ) / M,   n_nz = #rows with w_n rowsum_n != 0:
//   the reference drops rows whose sum is exactly 0 (`lp.sum() != 0`, a host sync there) and its broadcast
//   subtracts beta*kld.sum() once per surviving row -- reproduced on the device, no sync.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void expmul_kernel(const float* __restrict__ lw, const float* __restrict__ r,
                                                     const float* __restrict__ g, float* __restrict__ o0,
                                                     float* __restrict__ o1, int n, int bwd) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float e = expf(lw[i]);
  if (!bwd) o0[i] = e * r[i];
  else {
    o0[i] = g[i] * e * r[i];  // d/dlw
    o1[i] = g[i] * e;         // d/dr
  }
}
extern "C" int mmvae_expmul_fwd(const float* lw, const float* r, float* out, int n, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(lw && r && out && n > 0);
  hipLaunchKernelGGL(expmul_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, lw, r, nullptr, out,
                     nullptr, n, 0);
  return mmvae_launch_status();
}
extern "C" int mmvae_expmul_bwd(const float* lw, const float* r, const float* g, float* dlw, float* dr, int n,
                                mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(lw && r && g && dlw && dr && n > 0);
  hipLaunchKernelGGL(expmul_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, lw, r, g, dlw, dr, n, 1);
  return mmvae_launch_status();
}
__global__ __launch_bounds__(256) void moe_elbo_fwd_kernel(const float* __restrict__ rows, lincomb_w W,
                                                           const float* __restrict__ kld, float* __restrict__ out,
                                                           int n_rows, int M, int B, float beta) {
  __shared__ float red[4];
  __shared__ float acc[2];
  if (threadIdx.x == 0) acc[0] = acc[1] = 0.f;
  float total = 0.f, nnz = 0.f;
  for (int n = 0; n < n_rows; ++n) {
    float a = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) a += rows[(size_t)n * B + b];
    a = block_sum_256(a, red) * W.w[n];
    total += a;
    nnz += (a != 0.f) ? 1.f : 0.f;
  }
  float k = 0.f;
  for (int i = threadIdx.x; i < M * B; i += 256) k += kld[i];
  k = block_sum_256(k, red);
  if (threadIdx.x == 0) {
    out[0] = (total + nnz * beta * k) / (float)M;
    out[1] = nnz;
  }
}
__global__ __launch_bounds__(256) void moe_elbo_bwd_kernel(const float* __restrict__ g, const float* __restrict__ out,
                                                           lincomb_w W, float* __restrict__ drows,
                                                           float* __restrict__ dkld, int n_rows, int M, int B,
                                                           float beta) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const float gg = g[0] / (float)M;
  if (i < n_rows * B) drows[i] = gg * W.w[i / B];
  if (i < M * B) dkld[i] = gg * out[1] * beta;
}
extern "C" int mmvae_moe_elbo_fwd(const float* rows, const float* W_host, const float* kld, float* out, int n_rows,
                                  int M, int B, float beta, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(rows && W_host && kld && out && n_rows > 0 && M > 0 && B > 0);
  lincomb_w w;
  int rc = pack_w(W_host, n_rows, 1, &w);
  if (rc) return rc;
  hipLaunchKernelGGL(moe_elbo_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, rows, w, kld, out, n_rows, M, B,
                     beta);
  return mmvae_launch_status();
}
extern "C" int mmvae_moe_elbo_bwd(const float* g, const float* out, const float* W_host, float* drows, float* dkld,
                                  int n_rows, int M, int B, float beta, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(g && out && W_host && drows && dkld && n_rows > 0 && M > 0 && B > 0);
  lincomb_w w;
  int rc = pack_w(W_host, n_rows, 1, &w);
  if (rc) return rc;
  const int n = (n_rows > M ? n_rows : M) * B;
  hipLaunchKernelGGL(moe_elbo_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, g, out, w, drows,
                     dkld, n_rows, M, B, beta);
  return mmvae_launch_status();
}

MMVAE_TRACE_SETTER(loss)
