// Encoder heads + fused latent op (product of experts -> reparameterise -> analytic KL) for gfx950.
// Elementwise / small-reduction work over (experts x B x D) with D <= 256: one wavefront per sample,
// lanes over the latent dimension, wave-shuffle reductions; no LDS, no atomics (deterministic).
#include "common.hpp"
#include <type_traits>

// one wave per sample up to this many waves, then samples round robin (round 4: 128 -> 512; a wave's samples are a serial
// chain of L2 round trips: same box, ms/step of the cfg2 step at batch 512 / 1000 with 128 / 256 / 512 waves: 0.928 / 0.921 /
// 0.919 and 1.581 / 1.564 / 1.557)
#define POE_MAX_WAVES 512
#define POE_SLOTS 4  // D <= 64 * POE_SLOTS

// ---------------------------------------------------------------------------------------------
// VaeComponent.process_output (models/encoders.py:49-54): lv = softmax(u, -1) + 1e-6 on h[:, D:2D]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_softmax_fwd_kernel(float* __restrict__ h, int B, int D) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float* u = h + (size_t)b * 2 * D + D;
  float mx = -INFINITY;
  for (int d = lane; d < D; d += 64) mx = fmaxf(mx, u[d]);
  mx = wave_max(mx);
  float s = 0.f;
  for (int d = lane; d < D; d += 64) s += expf(u[d] - mx);
  s = wave_sum(s);
  const float inv = 1.0f / s;
  for (int d = lane; d < D; d += 64) u[d] = expf(u[d] - mx) * inv + 1e-6f;
}

// dh[:, D:] <- s * (dlv - sum(s * dlv)),  s = lv - 1e-6
__global__ __launch_bounds__(256) void head_softmax_bwd_kernel(const float* __restrict__ h, float* __restrict__ dh,
                                                               int B, int D) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const float* lv = h + (size_t)b * 2 * D + D;
  float* g = dh + (size_t)b * 2 * D + D;
  float dot = 0.f;
  for (int d = lane; d < D; d += 64) dot += (lv[d] - 1e-6f) * g[d];
  dot = wave_sum(dot);
  for (int d = lane; d < D; d += 64) g[d] = (lv[d] - 1e-6f) * (g[d] - dot);
}

extern "C" int mmvae_head_softmax_fwd(float* h, int B, int D, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(h && B > 0 && D > 0);
  hipLaunchKernelGGL(head_softmax_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, h, B, D);
  return mmvae_launch_status();
}
extern "C" int mmvae_head_softmax_bwd(const float* h, float* dh, int B, int D, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(h && dh && B > 0 && D > 0);
  hipLaunchKernelGGL(head_softmax_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, h, dh, B, D);
  return mmvae_launch_status();
}

// ---------------------------------------------------------------------------------------------
// prior sigma: softmax(theta) * D   (models/mmvae_models.py:274-276), computed per wave
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void prior_sigma(const float* __restrict__ theta, int D, int lane, float sp[POE_SLOTS],
                                            float sm[POE_SLOTS]) {
  float mx = -INFINITY;
#pragma unroll
  for (int s = 0; s < POE_SLOTS; ++s) {
    int d = lane + 64 * s;
    if (d < D) mx = fmaxf(mx, theta[d]);
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int s = 0; s < POE_SLOTS; ++s) {
    int d = lane + 64 * s;
    sm[s] = d < D ? expf(theta[d] - mx) : 0.f;
    sum += sm[s];
  }
  sum = wave_sum(sum);
#pragma unroll
  for (int s = 0; s < POE_SLOTS; ++s) {
    sm[s] = sm[s] / sum;
    sp[s] = sm[s] * (float)D;
  }
}

// statistics of softmax(u[0..D)) for one row: max and 1 / sum exp  (raw heads: lv = softmax(u) + 1e-6 computed here
// instead of by a head_softmax launch per tower and direction)
__device__ __forceinline__ void row_softmax_stats(const float* __restrict__ u, int D, int lane, float* mx_out,
                                                  float* inv_out) {
  float mx = -INFINITY;
#pragma unroll
  for (int s = 0; s < POE_SLOTS; ++s) {
    const int d = lane + 64 * s;
    if (d < D) mx = fmaxf(mx, u[d]);
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int s = 0; s < POE_SLOTS; ++s) {
    const int d = lane + 64 * s;
    if (d < D) sum += expf(u[d] - mx);
  }
  sum = wave_sum(sum);
  *mx_out = mx;
  *inv_out = 1.0f / sum;
}

// element e of the counter-based standard-normal stream `key` (see randn_kernel: Box-Muller over the hash pair of
// element pair e >> 1; the even element takes the cosine branch)
__device__ __forceinline__ uint32_t randn_key(const uint32_t* __restrict__ state) {
  return drop_fmix(state[0] ^ (state[1] * 0x9E3779B1u) ^ 0x632BE5ABu);
}
__device__ __forceinline__ float randn_elem(uint32_t key, long e) {
  const long i = e >> 1;
  const uint32_t h1 = drop_fmix(key + (uint32_t)(2 * i) * 0x9E3779B1u);
  const uint32_t h2 = drop_fmix(key + (uint32_t)(2 * i + 1) * 0x9E3779B1u);
  const float u1 = ((float)(h1 >> 8) + 1.0f) * (1.0f / 16777216.0f);   // (0, 1]
  const float u2 = (float)(h2 >> 8) * (1.0f / 16777216.0f);            // [0, 1)
  const float r = sqrtf(-2.0f * logf(u1));
  float sn, cs;
  sincosf(6.283185307179586f * u2, &sn, &cs);
  return (e & 1) ? r * sn : r * cs;
}
// the last workgroup of a launch that consumed the stream advances its counter
__device__ __forceinline__ void randn_advance(uint32_t* __restrict__ state) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t ticket = atomicAdd(state + 2, 1u);
    if (ticket == gridDim.x - 1) {
      state[2] = 0u;
      state[1] += 1u;
    }
  }
}

__device__ __forceinline__ float kl_elem(float mu, float s, float sp) {
  float r = s / sp, m = mu / sp;
  float vr = r * r;
  return 0.5f * (vr + m * m - 1.0f - logf(vr));
}

__global__ __launch_bounds__(256) void poe_fwd_kernel(mmvae_poe_fwd_args a, const float* __restrict__ theta,
                                                      float* __restrict__ joint, float* __restrict__ kl, int E,
                                                      int with_prior, int n_z, unsigned kl_mask, int B, int D, int ld,
                                                      int raw, uint32_t* __restrict__ rng) {
  MMVAE_TRACE_STAMP(22);
  // rng != NULL: the noise is drawn HERE (element i*B*D + b*D + d of the generator's current draw, exactly what
  // mmvae_randn would have put into a (n_z, B, D) tensor) and written to a.eps for the backward pass -- one launch
  // less in front of the fusion
  const uint32_t rkey = rng ? randn_key(rng) : 0u;
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * 4;
  float sp[POE_SLOTS], sm[POE_SLOTS];
  prior_sigma(theta, D, lane, sp, sm);
  for (int b = wave; b < B; b += nwaves) {
    float klacc[MMVAE_MAX_EXPERTS + 1];
#pragma unroll
    for (int j = 0; j <= MMVAE_MAX_EXPERTS; ++j) klacc[j] = 0.f;
    float umx[MMVAE_MAX_EXPERTS], uinv[MMVAE_MAX_EXPERTS];
    if (raw) {
#pragma unroll
      for (int e = 0; e < MMVAE_MAX_EXPERTS; ++e)
        if (e < E) row_softmax_stats(a.lv[e] + (size_t)b * ld, D, lane, &umx[e], &uinv[e]);
    }
#pragma unroll
    for (int s = 0; s < POE_SLOTS; ++s) {
      const int d = lane + 64 * s;
      if (d < D) {
      const size_t o = (size_t)b * D + d;    // contiguous (B,D) tensors
      const size_t oi = (size_t)b * ld + d;  // strided expert tensors
      float P = 0.f, S = 0.f;
#pragma unroll
      for (int e = 0; e < MMVAE_MAX_EXPERTS; ++e) {
        if (e >= E) continue;
        const float mu = a.mu[e][oi];
        float lv = a.lv[e][oi];
        if (raw) lv = expf(lv - umx[e]) * uinv[e] + 1e-6f;
        const float T = 1.0f / (expf(lv) + 1e-8f);
        P += T;
        S += mu * T;
        if (kl_mask & (1u << e)) klacc[e] += kl_elem(mu, lv, sp[s]);
      }
      if (with_prior == 1) P += 1.0f / (1.0f + 1e-8f);
      float muJ = S / P, varJ = 1.0f / P;
      if (with_prior == 2) {   // no product: the "joint" is expert 0 itself, sigma = its lv (MoE posteriors)
        muJ = a.mu[0][oi];
        varJ = a.lv[0][oi];
      }
      joint[o] = muJ;
      joint[(size_t)B * D + o] = varJ;
      if (kl_mask & (1u << E)) klacc[E] += kl_elem(muJ, varJ, sp[s]);
#pragma unroll
      for (int i = 0; i < MMVAE_MAX_EXPERTS; ++i) {
        if (i >= n_z) continue;
        float ev;
        if (rng) {
          ev = randn_elem(rkey, (long)i * B * D + (long)o);
          const_cast<float*>(a.eps[i])[o] = ev;
        } else {
          ev = a.eps[i][o];
        }
        a.z[i][o] = muJ + varJ * ev;
      }
      }
    }
    if (kl) {
#pragma unroll
      for (int j = 0; j <= MMVAE_MAX_EXPERTS; ++j) {
        if (j > E) continue;
        float v = wave_sum(klacc[j]);   // rows outside kl_mask accumulate nothing: written as 0
        if (lane == 0) kl[(size_t)j * B + b] = v;
      }
    }
  }
  if (rng) randn_advance(rng);
}

// Backward.  Per element (all for fixed b,d):
//   G_mu  = sum_i dz_i            + gk_J * muJ / sp^2
//   G_var = sum_i dz_i * eps_i    + gk_J * (varJ / sp^2 - 1 / varJ)
//   dmu_e = G_mu * T_e * varJ                                  + gk_e * mu_e / sp^2
//   dlv_e = [G_mu (mu_e - muJ) varJ - G_var varJ^2] * (-exp(lv_e) T_e^2)  + gk_e * (lv_e / sp^2 - 1 / lv_e)
//   dsp  += sum_j gk_j * (1 - (s_j^2 + mu_j^2) / sp^2) / sp
// COHERENT: the partial rows were written by other workgroups of the SAME launch with agent-scope write-through stores
// (poe_ws_store); read them with agent-scope loads that bypass the (per-XCD, non-coherent) L2
template <bool COHERENT = false>
__device__ __forceinline__ void poe_theta_body(const float* __restrict__ theta, const float* __restrict__ ws,
                                               float* __restrict__ dtheta, int nrows, int D, int accumulate,
                                               float (*part)[64 * POE_SLOTS]);
__device__ __forceinline__ void poe_ws_store(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Last-workgroup election without __threadfence(): a release fence writes back -- and the matching acquire
// invalidates -- the XCD's whole L2, under the feet of whatever else is running (the other tower's kernels share the
// GPU with this launch).  The partial rows go out as agent-scope write-through stores; vmcnt(0) says they have
// arrived; the ticket is a relaxed agent-scope atomic; the elected workgroup reads with agent-scope loads.
__device__ __forceinline__ bool poe_last_workgroup(int* __restrict__ ticket, int* last_lds) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const int t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *last_lds = t == (int)gridDim.x - 1;
    if (*last_lds) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  return *last_lds != 0;
}
__global__ __launch_bounds__(256) void poe_bwd_kernel(mmvae_poe_bwd_args a, const float* __restrict__ theta,
                                                      const float* __restrict__ dkl, float* __restrict__ ws, int E,
                                                      int with_prior, int n_z, unsigned kl_mask, int B, int D, int ld,
                                                      int raw, float* __restrict__ dtheta, int* __restrict__ ticket,
                                                      int accumulate, unsigned acc_packed) {
  MMVAE_TRACE_STAMP(23);
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * 4;
  float sp[POE_SLOTS], sm[POE_SLOTS], dsp[POE_SLOTS];
  prior_sigma(theta, D, lane, sp, sm);
#pragma unroll
  for (int s = 0; s < POE_SLOTS; ++s) dsp[s] = 0.f;
  for (int b = wave; b < B; b += nwaves) {
    float gk[MMVAE_MAX_EXPERTS + 1];
#pragma unroll
    for (int j = 0; j <= MMVAE_MAX_EXPERTS; ++j)
      gk[j] = (j <= E && dkl && (kl_mask & (1u << j))) ? dkl[(size_t)j * B + b] : 0.f;
    float umx[MMVAE_MAX_EXPERTS], uinv[MMVAE_MAX_EXPERTS], udot[MMVAE_MAX_EXPERTS];
#pragma unroll
    for (int e = 0; e < MMVAE_MAX_EXPERTS; ++e) udot[e] = 0.f;
    if (raw) {
#pragma unroll
      for (int e = 0; e < MMVAE_MAX_EXPERTS; ++e)
        if (e < E) row_softmax_stats(a.lv[e] + (size_t)b * ld, D, lane, &umx[e], &uinv[e]);
    }
#pragma unroll
    for (int s = 0; s < POE_SLOTS; ++s) {
      const int d = lane + 64 * s;
      if (d < D) {
      const size_t o = (size_t)b * D + d;    // contiguous (B,D) tensors
      const size_t oi = (size_t)b * ld + d;  // strided expert tensors
      const float isp2 = 1.0f / (sp[s] * sp[s]);
      float mu[MMVAE_MAX_EXPERTS], lv[MMVAE_MAX_EXPERTS], T[MMVAE_MAX_EXPERTS];
      float P = 0.f, S = 0.f;
#pragma unroll
      for (int e = 0; e < MMVAE_MAX_EXPERTS; ++e) {
        if (e >= E) continue;
        mu[e] = a.mu[e][oi];
        lv[e] = a.lv[e][oi];
        if (raw) lv[e] = expf(lv[e] - umx[e]) * uinv[e] + 1e-6f;
        T[e] = 1.0f / (expf(lv[e]) + 1e-8f);
        P += T[e];
        S += mu[e] * T[e];
      }
      if (with_prior == 1) P += 1.0f / (1.0f + 1e-8f);
      float muJ = S / P, varJ = 1.0f / P;
      if (with_prior == 2) {
        muJ = mu[0];
        varJ = lv[0];
      }
      float Gmu = 0.f, Gvar = 0.f;
#pragma unroll
      for (int i = 0; i < MMVAE_MAX_EXPERTS; ++i) {
        if (i >= n_z) continue;
        const float g = a.dz[i][o];
        Gmu += g;
        Gvar += g * a.eps[i][o];
      }
      float dspe = 0.f;
      if (gk[E] != 0.f) {
        Gmu += gk[E] * muJ * isp2;
        Gvar += gk[E] * (varJ * isp2 - 1.0f / varJ);
        dspe += gk[E] * (1.0f - (varJ * varJ + muJ * muJ) * isp2) / sp[s];
      }
#pragma unroll
      for (int e = 0; e < MMVAE_MAX_EXPERTS; ++e) {
        if (e >= E) continue;
        float dmu = Gmu * T[e] * varJ;
        float dT = Gmu * (mu[e] - muJ) * varJ - Gvar * varJ * varJ;
        float dlv = dT * (-expf(lv[e]) * T[e] * T[e]);
        if (with_prior == 2) {   // direct: z = mu + lv * eps
          dmu = Gmu;
          dlv = Gvar;
        }
        if (gk[e] != 0.f) {
          dmu += gk[e] * mu[e] * isp2;
          dlv += gk[e] * (lv[e] * isp2 - 1.0f / lv[e]);
          dspe += gk[e] * (1.0f - (lv[e] * lv[e] + mu[e] * mu[e]) * isp2) / sp[s];
        }
        // (acc_packed bit e: expert e's gradient tensor already holds another call's contribution to these columns -- the
        // same head output read by several fusion calls, DMVAE's joint / shared / private: add instead of an addition launch)
        const bool acc_e = (acc_packed >> e) & 1u;
        a.dmu[e][oi] = acc_e ? a.dmu[e][oi] + dmu : dmu;
        a.dlv[e][oi] = acc_e ? a.dlv[e][oi] + dlv : dlv;
        udot[e] += (lv[e] - 1e-6f) * dlv;
      }
      dsp[s] += dspe;
      }
    }
    if (raw) {      // through lv = softmax(u) + 1e-6:  du = s (dlv - sum(s dlv)),  s = lv - 1e-6
#pragma unroll
      for (int e = 0; e < MMVAE_MAX_EXPERTS; ++e) {
        if (e >= E) continue;
        const float dot = wave_sum(udot[e]);
#pragma unroll
        for (int s = 0; s < POE_SLOTS; ++s) {
          const int d = lane + 64 * s;
          if (d < D) {
            const size_t oi = (size_t)b * ld + d;
            const float sv = expf(a.lv[e][oi] - umx[e]) * uinv[e];
            a.dlv[e][oi] = sv * (a.dlv[e][oi] - dot);
          }
        }
      }
    }
  }
#pragma unroll
  for (int s = 0; s < POE_SLOTS; ++s) {
    const int d = lane + 64 * s;
    if (d < D) {
      if (ticket) poe_ws_store(ws + (size_t)wave * D + d, dsp[s]);
      else ws[(size_t)wave * D + d] = dsp[s];
    }
  }
  if (ticket) {
    // the prior-parameter gradient in the same launch: the last workgroup to finish folds every wave's partial row
    // (a second one-workgroup launch sat on the backward critical path between the fusion and the encoders)
    __shared__ float part[4][64 * POE_SLOTS];
    __shared__ int last;
    if (!poe_last_workgroup(ticket, &last)) return;
    poe_theta_body<true>(theta, ws, dtheta, nwaves, D, accumulate, part);
  }
}

// dtheta_d (+)= D * s_d * (dsp_d - sum_k s_k dsp_k), dsp = sum over the per-wave partial rows.
// 4 waves split the rows, 8 independent loads in flight per lane, LDS combine (D <= 256).
template <bool COHERENT>
__device__ __forceinline__ void poe_theta_body(const float* __restrict__ theta, const float* __restrict__ ws,
                                               float* __restrict__ dtheta, int nrows, int D, int accumulate,
                                               float (*part)[64 * POE_SLOTS]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  auto ld = [&](size_t i) -> float {
    return COHERENT ? __hip_atomic_load(ws + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ws[i];
  };
#pragma unroll
  for (int s = 0; s < POE_SLOTS; ++s) {
    const int d = lane + 64 * s;
    float acc = 0.f;
    if (d < D) {
      int r = wave;
      for (; r + 28 < nrows; r += 32) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ld((size_t)(r + 4 * u) * D + d);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
      }
      for (; r < nrows; r += 4) acc += ld((size_t)r * D + d);
    }
    part[wave][d] = acc;
  }
  __syncthreads();
  if (wave != 0) return;
  float sp[POE_SLOTS], sm[POE_SLOTS], dsp[POE_SLOTS];
  prior_sigma(theta, D, lane, sp, sm);
  float dot = 0.f;
#pragma unroll
  for (int s = 0; s < POE_SLOTS; ++s) {
    const int d = lane + 64 * s;
    dsp[s] = part[0][d] + part[1][d] + part[2][d] + part[3][d];
    dot += dsp[s] * sm[s];
  }
  dot = wave_sum(dot);
#pragma unroll
  for (int s = 0; s < POE_SLOTS; ++s) {
    const int d = lane + 64 * s;
    if (d < D) {
      float g = (float)D * sm[s] * (dsp[s] - dot);
      dtheta[d] = accumulate ? dtheta[d] + g : g;
    }
  }
}
__global__ __launch_bounds__(256) void poe_theta_kernel(const float* __restrict__ theta, const float* __restrict__ ws,
                                                        float* __restrict__ dtheta, int nrows, int D, int accumulate) {
  __shared__ float part[4][64 * POE_SLOTS];
  poe_theta_body(theta, ws, dtheta, nrows, D, accumulate, part);
}

// ---- D <= 64 fast paths -------------------------------------------------------------------------------------------
// The generic kernels above walk POE_SLOTS column slots and MMVAE_MAX_EXPERTS experts with run-time bounds: every
// bound is a branch, every branch a basic block the compiler cannot move loads across, so a launch was a chain of
// four to six dependent L2 round trips (prior row, one softmax row per expert, the expert rows again, noise /
// upstream gradients) for a few hundred flops: 8.6 / 12.5 us, on the critical path of BOTH towers.  With the expert
// and sample counts as template arguments and one column per lane, every load of a row is issued before the first
// use, inactive lanes load a clamped column and only their stores are predicated.  Same arithmetic, same order.
template <int E, int NZ>
__global__ __launch_bounds__(256) void poe_fwd_fast_kernel(mmvae_poe_fwd_args a, const float* __restrict__ theta,
                                                           float* __restrict__ joint, float* __restrict__ kl,
                                                           int with_prior, unsigned kl_mask, int B, int D, int ld,
                                                           int raw, uint32_t* __restrict__ rng) {
  MMVAE_TRACE_STAMP(22);
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * 4;
  const bool live = lane < D;
  const int dc = live ? lane : D - 1;
  const float th = theta[dc];
  const uint32_t rkey = rng ? randn_key(rng) : 0u;
  float sp = 1.f;
  bool have_prior = false;
  for (int b = wave; b < B; b += nwaves) {
    const size_t o = (size_t)b * D + dc, oi = (size_t)b * ld + dc;
    float mu[E], lv[E], ep[NZ > 0 ? NZ : 1];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      mu[e] = a.mu[e][oi];
      lv[e] = a.lv[e][oi];
    }
    if (!rng) {
#pragma unroll
      for (int i = 0; i < NZ; ++i) ep[i] = a.eps[i][o];
    }
    if (!have_prior) {      // prior sigma = softmax(theta) * D
      const float mx = wave_max(live ? th : -INFINITY);
      const float ex = live ? expf(th - mx) : 0.f;
      const float sum = wave_sum(ex);
      sp = (ex / sum) * (float)D;
      have_prior = true;
    }
    const float lv0_raw = lv[0];
    if (raw) {
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const float mx = wave_max(live ? lv[e] : -INFINITY);
        const float ex = live ? expf(lv[e] - mx) : 0.f;
        const float inv = 1.0f / wave_sum(ex);
        lv[e] = expf(lv[e] - mx) * inv + 1e-6f;
      }
    }
    float klacc[E + 1];
    float P = 0.f, S = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const float T = 1.0f / (expf(lv[e]) + 1e-8f);
      P += T;
      S += mu[e] * T;
      klacc[e] = (live && (kl_mask & (1u << e))) ? kl_elem(mu[e], lv[e], sp) : 0.f;
    }
    if (with_prior == 1) P += 1.0f / (1.0f + 1e-8f);
    float muJ = S / P, varJ = 1.0f / P;
    if (with_prior == 2) {
      muJ = mu[0];
      varJ = lv0_raw;
    }
    klacc[E] = (live && (kl_mask & (1u << E))) ? kl_elem(muJ, varJ, sp) : 0.f;
    if (live) {
      joint[o] = muJ;
      joint[(size_t)B * D + o] = varJ;
    }
#pragma unroll
    for (int i = 0; i < NZ; ++i) {
      float ev;
      if (rng) {
        ev = randn_elem(rkey, (long)i * B * D + (long)o);
        if (live) const_cast<float*>(a.eps[i])[o] = ev;
      } else {
        ev = ep[i];
      }
      if (live) a.z[i][o] = muJ + varJ * ev;
    }
    if (kl) {
#pragma unroll
      for (int j = 0; j <= E; ++j) {
        const float v = wave_sum(klacc[j]);
        if (lane == 0) kl[(size_t)j * B + b] = v;
      }
    }
  }
  if (rng) randn_advance(rng);
}

template <int E, int NZ>
__global__ __launch_bounds__(256) void poe_bwd_fast_kernel(mmvae_poe_bwd_args a, const float* __restrict__ theta,
                                                           const float* __restrict__ dkl, float* __restrict__ ws,
                                                           int with_prior, unsigned kl_mask, int B, int D, int ld,
                                                           int raw, float* __restrict__ dtheta,
                                                           int* __restrict__ ticket, int accumulate, unsigned acc_packed) {
  MMVAE_TRACE_STAMP(23);
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * 4;
  const bool live = lane < D;
  const int dc = live ? lane : D - 1;
  const float th = theta[dc];
  float sp = 1.f, dsp = 0.f;
  bool have_prior = false;
  for (int b = wave; b < B; b += nwaves) {
    const size_t o = (size_t)b * D + dc, oi = (size_t)b * ld + dc;
    float gk[E + 1], mu[E], lv[E], T[E], dz[NZ > 0 ? NZ : 1], ep[NZ > 0 ? NZ : 1];
#pragma unroll
    for (int j = 0; j <= E; ++j) gk[j] = (dkl && (kl_mask & (1u << j))) ? dkl[(size_t)j * B + b] : 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      mu[e] = a.mu[e][oi];
      lv[e] = a.lv[e][oi];
    }
#pragma unroll
    for (int i = 0; i < NZ; ++i) {
      dz[i] = a.dz[i][o];
      ep[i] = a.eps[i][o];
    }
    if (!have_prior) {
      const float mx = wave_max(live ? th : -INFINITY);
      const float ex = live ? expf(th - mx) : 0.f;
      const float sum = wave_sum(ex);
      sp = (ex / sum) * (float)D;
      have_prior = true;
    }
    float sv[E];      // softmax(u) of the raw heads
    if (raw) {
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const float mx = wave_max(live ? lv[e] : -INFINITY);
        const float ex = live ? expf(lv[e] - mx) : 0.f;
        const float inv = 1.0f / wave_sum(ex);
        sv[e] = expf(lv[e] - mx) * inv;
        lv[e] = sv[e] + 1e-6f;
      }
    }
    const float isp2 = 1.0f / (sp * sp);
    float P = 0.f, S = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      T[e] = 1.0f / (expf(lv[e]) + 1e-8f);
      P += T[e];
      S += mu[e] * T[e];
    }
    if (with_prior == 1) P += 1.0f / (1.0f + 1e-8f);
    float muJ = S / P, varJ = 1.0f / P;
    if (with_prior == 2) {
      muJ = mu[0];
      varJ = lv[0];
    }
    float Gmu = 0.f, Gvar = 0.f;
#pragma unroll
    for (int i = 0; i < NZ; ++i) {
      Gmu += dz[i];
      Gvar += dz[i] * ep[i];
    }
    float dspe = 0.f;
    if (gk[E] != 0.f) {
      Gmu += gk[E] * muJ * isp2;
      Gvar += gk[E] * (varJ * isp2 - 1.0f / varJ);
      dspe += gk[E] * (1.0f - (varJ * varJ + muJ * muJ) * isp2) / sp;
    }
    float dlv[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      float dmu = Gmu * T[e] * varJ;
      const float dT = Gmu * (mu[e] - muJ) * varJ - Gvar * varJ * varJ;
      dlv[e] = dT * (-expf(lv[e]) * T[e] * T[e]);
      if (with_prior == 2) {
        dmu = Gmu;
        dlv[e] = Gvar;
      }
      if (gk[e] != 0.f) {
        dmu += gk[e] * mu[e] * isp2;
        dlv[e] += gk[e] * (lv[e] * isp2 - 1.0f / lv[e]);
        dspe += gk[e] * (1.0f - (lv[e] * lv[e] + mu[e] * mu[e]) * isp2) / sp;
      }
      if (live) a.dmu[e][oi] = ((acc_packed >> e) & 1u) ? a.dmu[e][oi] + dmu : dmu;
    }
    if (live) dsp += dspe;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      if (raw) {      // through lv = softmax(u) + 1e-6:  du = s (dlv - sum(s dlv)),  s = lv - 1e-6
        const float dot = wave_sum(live ? (lv[e] - 1e-6f) * dlv[e] : 0.f);
        dlv[e] = sv[e] * (dlv[e] - dot);
      }
      if (live) a.dlv[e][oi] = ((acc_packed >> e) & 1u) ? a.dlv[e][oi] + dlv[e] : dlv[e];
    }
  }
  if (live) {
    if (ticket) poe_ws_store(ws + (size_t)wave * D + lane, dsp);
    else ws[(size_t)wave * D + lane] = dsp;
  }
  if (ticket) {
    __shared__ float part[4][64 * POE_SLOTS];
    __shared__ int last;
    if (!poe_last_workgroup(ticket, &last)) return;
    poe_theta_body<true>(theta, ws, dtheta, nwaves, D, accumulate, part);
  }
}

// (E, n_z) -> instantiation; false when the generic kernels have to run
template <typename F>
static inline bool poe_fast_visit(int E, int n_z, int D, F&& f) {
  if (D > 64 || E < 1 || E > 3 || n_z < 0 || n_z > 3) return false;
#define POE_CASE(EE, ZZ) if (E == EE && n_z == ZZ) { f(std::integral_constant<int, EE>{}, std::integral_constant<int, ZZ>{}); return true; }
  POE_CASE(1, 0) POE_CASE(1, 1) POE_CASE(1, 2) POE_CASE(1, 3)
  POE_CASE(2, 0) POE_CASE(2, 1) POE_CASE(2, 2) POE_CASE(2, 3)
  POE_CASE(3, 0) POE_CASE(3, 1) POE_CASE(3, 2) POE_CASE(3, 3)
#undef POE_CASE
  return false;
}

static inline int poe_blocks(int B) {
  int waves = B < POE_MAX_WAVES ? B : POE_MAX_WAVES;
  return (waves + 3) / 4;
}

extern "C" size_t mmvae_poe_ws_floats(int B, int D) { return (size_t)poe_blocks(B) * 4 * D; }

extern "C" int mmvae_poe_reparam_kl_fwd(const mmvae_poe_fwd_args* a, const float* theta, float* joint, float* kl,
                                        int E, int with_prior, int n_z, unsigned kl_mask, int B, int D, int ld_in,
                                        int raw_heads, uint32_t* rng_state, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(a && theta && joint && B > 0 && D > 0 && E > 0 && ld_in >= D);
  if (E > MMVAE_MAX_EXPERTS || n_z > MMVAE_MAX_EXPERTS || D > 64 * POE_SLOTS) return MMVAE_ERR_UNSUPPORTED;
  if (with_prior == 2 && E != 1) return MMVAE_ERR_ARG;
  if (kl_mask && !kl) return MMVAE_ERR_ARG;
  if (!poe_fast_visit(E, n_z, D, [&](auto e, auto z) {
        hipLaunchKernelGGL((poe_fwd_fast_kernel<decltype(e)::value, decltype(z)::value>), dim3(poe_blocks(B)), dim3(256),
                           0, (hipStream_t)stream, *a, theta, joint, kl, with_prior, kl_mask, B, D, ld_in,
                           raw_heads ? 1 : 0, rng_state);
      }))
    hipLaunchKernelGGL(poe_fwd_kernel, dim3(poe_blocks(B)), dim3(256), 0, (hipStream_t)stream, *a, theta, joint, kl, E,
                       with_prior, n_z, kl_mask, B, D, ld_in, raw_heads ? 1 : 0, rng_state);
  return mmvae_launch_status();
}

extern "C" int mmvae_poe_reparam_kl_bwd(const mmvae_poe_bwd_args* a, const float* theta, const float* dkl,
                                        float* dtheta, float* ws, int* ticket, int E, int with_prior, int n_z,
                                        unsigned kl_mask, int B, int D, int ld_in, int raw_heads, int accumulate,
                                        mmvae_stream_t stream) {
  return mmvae_poe_reparam_kl_bwd_acc(a, theta, dkl, dtheta, ws, ticket, E, with_prior, n_z, kl_mask, B, D, ld_in, raw_heads,
                                      accumulate, 0u, stream);
}
extern "C" int mmvae_poe_reparam_kl_bwd_acc(const mmvae_poe_bwd_args* a, const float* theta, const float* dkl,
                                            float* dtheta, float* ws, int* ticket, int E, int with_prior, int n_z,
                                            unsigned kl_mask, int B, int D, int ld_in, int raw_heads, int accumulate,
                                            unsigned acc_packed, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(a && theta && ws && B > 0 && D > 0 && E > 0 && ld_in >= D);
  if (acc_packed && raw_heads) return MMVAE_ERR_ARG;      // (the raw-head form uses the gradient tensor as scratch)
  if (E > MMVAE_MAX_EXPERTS || n_z > MMVAE_MAX_EXPERTS || D > 64 * POE_SLOTS) return MMVAE_ERR_UNSUPPORTED;
  const int nb = poe_blocks(B);
  const bool one_launch = dtheta && ticket;
  if (!poe_fast_visit(E, n_z, D, [&](auto e, auto z) {
        hipLaunchKernelGGL((poe_bwd_fast_kernel<decltype(e)::value, decltype(z)::value>), dim3(nb), dim3(256), 0,
                           (hipStream_t)stream, *a, theta, dkl, ws, with_prior, kl_mask, B, D, ld_in, raw_heads ? 1 : 0,
                           one_launch ? dtheta : nullptr, one_launch ? ticket : nullptr, accumulate, acc_packed);
      }))
    hipLaunchKernelGGL(poe_bwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, *a, theta, dkl, ws, E, with_prior,
                       n_z, kl_mask, B, D, ld_in, raw_heads ? 1 : 0, one_launch ? dtheta : nullptr,
                       one_launch ? ticket : nullptr, accumulate, acc_packed);
  if (dtheta && !one_launch)
    hipLaunchKernelGGL(poe_theta_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, theta, ws, dtheta, nb * 4, D,
                       accumulate);
  return mmvae_launch_status();
}

// ---------------------------------------------------------------------------------------------
// MoE importance weights (models/mmvae_models.py:56-62):
//   lw[b] = sum_d [ log N(z; mu_r, s_r) - log N(z; mu_o, s_o) ],  z and (mu_o, s_o) detached.
// packed_* = (B, 2D) [mu | sigma] head outputs.  One wave per sample.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void normal_logratio_fwd_kernel(const float* __restrict__ pr,
                                                                  const float* __restrict__ po,
                                                                  const float* __restrict__ z, float* __restrict__ lw,
                                                                  int B, int D) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float a = 0.f;
  for (int d = lane; d < D; d += 64) {
    const float zz = z[(size_t)b * D + d];
    const float mr = pr[(size_t)b * 2 * D + d], sr = pr[(size_t)b * 2 * D + D + d];
    const float mo = po[(size_t)b * 2 * D + d], so = po[(size_t)b * 2 * D + D + d];
    const float tr = (zz - mr) / sr, to = (zz - mo) / so;
    a += (-0.5f * tr * tr - logf(sr)) - (-0.5f * to * to - logf(so));
  }
  a = wave_sum(a);
  if (lane == 0) lw[b] = a;
}
// d lw / d mu_r = (z - mu_r) / s_r^2 ; d lw / d s_r = (z - mu_r)^2 / s_r^3 - 1 / s_r
__global__ __launch_bounds__(256) void normal_logratio_bwd_kernel(const float* __restrict__ pr,
                                                                  const float* __restrict__ z,
                                                                  const float* __restrict__ g, float* __restrict__ dpr,
                                                                  int B, int D) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * D) return;
  const int b = i / D, d = i - b * D;
  const float zz = z[i], mr = pr[(size_t)b * 2 * D + d], sr = pr[(size_t)b * 2 * D + D + d];
  const float t = (zz - mr) / sr, gb = g[b];
  dpr[(size_t)b * 2 * D + d] = gb * t / sr;
  dpr[(size_t)b * 2 * D + D + d] = gb * (t * t - 1.0f) / sr;
}
extern "C" int mmvae_normal_logratio_fwd(const float* packed_r, const float* packed_o, const float* z, float* lw,
                                         int B, int D, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(packed_r && packed_o && z && lw && B > 0 && D > 0);
  hipLaunchKernelGGL(normal_logratio_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, packed_r,
                     packed_o, z, lw, B, D);
  return mmvae_launch_status();
}
extern "C" int mmvae_normal_logratio_bwd(const float* packed_r, const float* z, const float* g, float* dpacked_r,
                                         int B, int D, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(packed_r && z && g && dpacked_r && B > 0 && D > 0);
  hipLaunchKernelGGL(normal_logratio_bwd_kernel, dim3((B * D + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     packed_r, z, g, dpacked_r, B, D);
  return mmvae_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Standard-normal noise for the reparameterised samples (torch.distributions rsample -> _standard_normal in the
// reference).  Counter-based (murmur3 finaliser over (seed, call counter, element)) + Box-Muller, two outputs per
// thread.  torch.randn inside a captured hipGraph makes every replay run two extra fill kernels (the generator's
// seed/offset tensors); this kernel keeps its state {seed, counter, ticket} on the device and bumps the counter
// itself when its last workgroup finishes.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void randn_kernel(float* __restrict__ out, long n, uint32_t* __restrict__ state) {
  const uint32_t key = randn_key(state);
  const long pairs = (n + 1) >> 1;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < pairs; i += (long)gridDim.x * 256) {
    const uint32_t h1 = drop_fmix(key + (uint32_t)(2 * i) * 0x9E3779B1u);
    const uint32_t h2 = drop_fmix(key + (uint32_t)(2 * i + 1) * 0x9E3779B1u);
    const float u1 = ((float)(h1 >> 8) + 1.0f) * (1.0f / 16777216.0f);   // (0, 1]
    const float u2 = (float)(h2 >> 8) * (1.0f / 16777216.0f);            // [0, 1)
    const float r = sqrtf(-2.0f * logf(u1));
    float sn, cs;
    sincosf(6.283185307179586f * u2, &sn, &cs);
    out[2 * i] = r * cs;
    if (2 * i + 1 < n) out[2 * i + 1] = r * sn;
  }
  randn_advance(state);
}
extern "C" int mmvae_randn(float* out, long n, uint32_t* state, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(out && state && n > 0);
  long blocks = ((n + 1) / 2 + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(randn_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, n, state);
  return mmvae_launch_status();
}

// ---- debug: device wall-clock marker ---------------------------------------------------------------------------
// One-thread kernel that stores the 100 MHz device wall clock into `slot`: dropped into a captured step at phase
// boundaries (tools/phase_timeline.py) it gives the true per-stream timeline of a graph replay, which a profiler's
// per-dispatch instrumentation distorts at this kernel size.
__global__ void timestamp_kernel(long long* slot) { *slot = (long long)wall_clock64(); }
extern "C" int mmvae_debug_timestamp(long long* slot, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(slot);
  hipLaunchKernelGGL(timestamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, slot);
  return mmvae_launch_status();
}

// debug: one thread that spins for `ticks` wall-clock ticks (10 ns each), then stamps slot[0] = start, slot[1] = end.
// Synthetic DAGs of these (tools/probe/graph_sched.py) show how a captured multi-stream graph is really scheduled.
__global__ void spin_kernel(long long* slot, long long ticks) {
  const long long t0 = (long long)wall_clock64();
  long long t = t0;
  while (t - t0 < ticks) t = (long long)wall_clock64();
  slot[0] = t0;
  slot[1] = t;
}
extern "C" int mmvae_debug_spin(long long* slot, long long ticks, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(slot);
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, slot, ticks);
  return mmvae_launch_status();
}

// ---------------------------------------------------------------------------------------------
// latent samples -> decoder inputs (PoE: one decoder call decodes the samples of several subsets as one batch; DMVAE: a
// decoder's own / joint / cross passes as one batch of [shared | private] rows) and the transpose.  Forward: up to 16 strided
// 2-D copies in one launch.  Backward: the gradient of every SOURCE = the sum of the blocks that read it, in block order.
// (torch.cat / repeat for the inputs and autograd's per-consumer additions for the gradients were 5 .. 13 launches per step.)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rows_fan_fwd_kernel(mmvae_fan_blocks_t t) {
  const int k = blockIdx.y;
  const int W = t.width[k];
  const long n = (long)t.B * W;
  const float* __restrict__ src = t.src[k];
  float* __restrict__ dst = t.dst[k];
  const int ls = t.ld_src[k], ld = t.ld_dst[k];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int b = (int)(i / W), d = (int)(i - (long)b * W);
    dst[(size_t)b * ld + d] = src[(size_t)b * ls + d];
  }
}
__global__ __launch_bounds__(256) void rows_fan_bwd_kernel(mmvae_fan_sum_t t) {
  const int s = blockIdx.y;
  const int W = t.width[s], ng = t.n_g[s];
  const long n = (long)t.B * W;
  float* __restrict__ out = t.out[s];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int b = (int)(i / W), d = (int)(i - (long)b * W);
    float a = t.g[s][0][(size_t)b * t.ld[s][0] + d];
    for (int j = 1; j < ng; ++j) a += t.g[s][j][(size_t)b * t.ld[s][j] + d];
    out[i] = a;
  }
}
extern "C" int mmvae_rows_fan_fwd(const mmvae_fan_blocks_t* blocks, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(blocks && blocks->n >= 1 && blocks->n <= MMVAE_FAN_MAX_BLOCKS && blocks->B > 0);
  int wmax = 0;
  for (int k = 0; k < blocks->n; ++k) {
    MMVAE_CHECK_ARG(blocks->src[k] && blocks->dst[k] && blocks->width[k] > 0 && blocks->ld_src[k] >= blocks->width[k] &&
                    blocks->ld_dst[k] >= blocks->width[k]);
    wmax = blocks->width[k] > wmax ? blocks->width[k] : wmax;
  }
  long bx = ((long)blocks->B * wmax + 255) / 256;
  if (bx > 256) bx = 256;
  hipLaunchKernelGGL(rows_fan_fwd_kernel, dim3((unsigned)bx, blocks->n), dim3(256), 0, (hipStream_t)stream, *blocks);
  return mmvae_launch_status();
}
extern "C" int mmvae_rows_fan_bwd(const mmvae_fan_sum_t* sums, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(sums && sums->n >= 1 && sums->n <= MMVAE_FAN_MAX_SRC && sums->B > 0);
  int wmax = 0;
  for (int s = 0; s < sums->n; ++s) {
    MMVAE_CHECK_ARG(sums->out[s] && sums->width[s] > 0 && sums->n_g[s] >= 1 && sums->n_g[s] <= MMVAE_FAN_MAX_USES);
    for (int j = 0; j < sums->n_g[s]; ++j) MMVAE_CHECK_ARG(sums->g[s][j] && sums->ld[s][j] >= sums->width[s]);
    wmax = sums->width[s] > wmax ? sums->width[s] : wmax;
  }
  long bx = ((long)sums->B * wmax + 255) / 256;
  if (bx > 256) bx = 256;
  hipLaunchKernelGGL(rows_fan_bwd_kernel, dim3((unsigned)bx, sums->n), dim3(256), 0, (hipStream_t)stream, *sums);
  return mmvae_launch_status();
}

MMVAE_TRACE_SETTER(latent)
