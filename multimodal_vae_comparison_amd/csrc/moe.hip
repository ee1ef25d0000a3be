// MoE with K samples per posterior and the DReG objective -- the reference's shipped configs/config_mnistsvhn.yml
// (mixing moe, obj dreg, K 30, prior laplace):
//   MOE.forward                     models/mmvae_models.py:80-117   (q_m.rsample([K]), own + cross decoding)
//   MOE.objective, non-elbo branch  models/mmvae_models.py:63-78
//   MultimodalObjective.dreg / _m_dreg_looser   models/objectives.py:361-387
// plus the Laplace pieces `prior: laplace` switches on for the elbo objective (KL(Laplace || Normal), the importance
// ratio under Laplace posteriors).  All HBM-bound elementwise / row-reduction work on (M, K, B, D) <= a few MB.
#include "common.hpp"

#define MOE_SLOTS 4   // D <= 256: lane owns d = lane + 64 s

// softmax(theta) [sm] and the prior sigma softmax(theta) * D [sp] (MOE.pz_params, models/mmvae_models.py:28-30)
__device__ __forceinline__ void moe_prior_sigma(const float* __restrict__ theta, int D, int lane, float sp[MOE_SLOTS],
                                                float sm[MOE_SLOTS]) {
  float mx = -INFINITY;
#pragma unroll
  for (int s = 0; s < MOE_SLOTS; ++s) {
    const int d = lane + 64 * s;
    if (d < D) mx = fmaxf(mx, theta[d]);
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int s = 0; s < MOE_SLOTS; ++s) {
    const int d = lane + 64 * s;
    sm[s] = d < D ? expf(theta[d] - mx) : 0.f;
    sum += sm[s];
  }
  sum = wave_sum(sum);
#pragma unroll
  for (int s = 0; s < MOE_SLOTS; ++s) {
    sm[s] = sm[s] / sum;
    sp[s] = sm[s] * (float)D;
  }
}

#define HALF_LOG_2PI_F 0.9189385332046727f
#define LOG_2_F 0.6931471805599453f
// log q(z) of one coordinate: torch.distributions.Normal / Laplace .log_prob with scale s
__device__ __forceinline__ float logq_elem(float z, float mu, float s, int laplace) {
  if (laplace) return -logf(2.0f * s) - fabsf(z - mu) / s;
  const float d = z - mu;
  return -(d * d) / (2.0f * (s * s)) - logf(s) - HALF_LOG_2PI_F;
}

// ---------------------------------------------------------------------------------------------
// forward: one wave per (r, k, b).  z_r[k,b,:] = mu_r[b,:] + s_r[b,:] * e_r[k,b,:];
//   lat[r,k,b] = sum_d log N(z; 0, sp_d) - beta * log-mean-exp_m sum_d log q_m(z)
//   (dreg: beta = 1, objectives.py:368-372; iwae: the objective's beta, objectives.py:356)
//   pi[r,k,b,m] = softmax_m (sum_d log q_m(z))  -- d lat / d (log q_m row sum) = -beta pi, kept for the backward pass
// ---------------------------------------------------------------------------------------------
template <int M>
__global__ __launch_bounds__(256) void moe_ksample_fwd_kernel(mmvae_moe_k_args a, const float* __restrict__ theta,
                                                              float* __restrict__ lat, float* __restrict__ pi, int K,
                                                              int B, int D, float beta) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)M * K * B) return;
  const int b = (int)(row % B);
  const int k = (int)((row / B) % K);
  const int r = (int)(row / ((long)B * K));
  float sp[MOE_SLOTS], sm[MOE_SLOTS];
  moe_prior_sigma(theta, D, lane, sp, sm);
  float lpz = 0.f, lq[M];
#pragma unroll
  for (int m = 0; m < M; ++m) lq[m] = 0.f;
  const float* __restrict__ pr = a.packed[r] + (size_t)b * 2 * D;
  const size_t zoff = ((size_t)k * B + b) * D;
#pragma unroll
  for (int s = 0; s < MOE_SLOTS; ++s) {
    const int d = lane + 64 * s;
    if (d < D) {
      const float z = pr[d] + pr[D + d] * a.eps[r][zoff + d];
      a.z[r][zoff + d] = z;
      lpz += -(z * z) / (2.0f * (sp[s] * sp[s])) - logf(sp[s]) - HALF_LOG_2PI_F;
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const float* __restrict__ pm = a.packed[m] + (size_t)b * 2 * D;
        lq[m] += logq_elem(z, pm[d], pm[D + d], a.laplace[m]);
      }
    }
  }
  lpz = wave_sum(lpz);
  float mx = -INFINITY;
#pragma unroll
  for (int m = 0; m < M; ++m) {
    lq[m] = wave_sum(lq[m]);
    mx = fmaxf(mx, lq[m]);
  }
  float se = 0.f;
#pragma unroll
  for (int m = 0; m < M; ++m) se += expf(lq[m] - mx);
  if (lane == 0) {
    lat[row] = lpz - beta * (mx + logf(se) - logf((float)M));
#pragma unroll
    for (int m = 0; m < M; ++m) pi[row * M + m] = expf(lq[m] - mx) / se;
  }
}

// ---------------------------------------------------------------------------------------------
// backward: one wave per sample b, lanes over d, serial over (r, k): no atomics, every (b, d) gradient is one
// register sum.  Inputs: dlat (M,K,B), dz_r (K,B,D) from the decoders (NULL = none).
//   d lat / d z       = -z / sp^2 - beta sum_m pi_m d log q_m / d z
//   d lat / d (mu_m)  = -beta pi_m d log q_m / d mu_m,  same for the scale
//   z = mu_r + s_r e  =>  dmu_r += dz_total, ds_r += dz_total * e
//   theta: dsp_d = sum g (z^2 / sp^3 - 1 / sp);  dtheta_j = D sm_j (dsp_j - sum_d dsp_d sm_d)  -> dtheta_rows[b, :]
// ---------------------------------------------------------------------------------------------
template <int M>
__global__ __launch_bounds__(256) void moe_ksample_bwd_kernel(mmvae_moe_k_bwd_args a, const float* __restrict__ theta,
                                                              const float* __restrict__ dlat,
                                                              const float* __restrict__ pi,
                                                              float* __restrict__ dtheta_rows, int K, int B, int D,
                                                              float beta) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float sp[MOE_SLOTS], sm[MOE_SLOTS], dsp[MOE_SLOTS];
  moe_prior_sigma(theta, D, lane, sp, sm);
  float dot = 0.f;
#pragma unroll
  for (int s = 0; s < MOE_SLOTS; ++s) {
    const int d = lane + 64 * s;
    dsp[s] = 0.f;
    if (d >= D) continue;
    float mu[M], sc[M], dmu[M], dsc[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      mu[m] = a.packed[m][(size_t)b * 2 * D + d];
      sc[m] = a.packed[m][(size_t)b * 2 * D + D + d];
      dmu[m] = 0.f;
      dsc[m] = 0.f;
    }
    const float isp = 1.0f / sp[s];
#pragma unroll
    for (int r = 0; r < M; ++r) {
      for (int k = 0; k < K; ++k) {
        const size_t row = ((size_t)r * K + k) * B + b;
        const size_t zi = ((size_t)k * B + b) * D + d;
        const float g = dlat[row];
        const float z = a.z[r][zi], e = a.eps[r][zi];
        float dz = a.dz[r] ? a.dz[r][zi] : 0.f;
        dz += g * (-z * isp * isp);
        dsp[s] += g * (z * z * isp * isp * isp - isp);
#pragma unroll
        for (int m = 0; m < M; ++m) {
          const float c = -g * beta * pi[row * M + m];
          const float is = 1.0f / sc[m];
          const float df = z - mu[m];
          if (a.laplace[m]) {
            const float sg = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
            dz += c * (-sg * is);
            dmu[m] += c * (sg * is);
            dsc[m] += c * (-is + fabsf(df) * is * is);
          } else {
            const float t = df * is;
            dz += c * (-t * is);
            dmu[m] += c * (t * is);
            dsc[m] += c * ((t * t - 1.0f) * is);
          }
        }
        dmu[r] += dz;
        dsc[r] += dz * e;
      }
    }
#pragma unroll
    for (int m = 0; m < M; ++m) {
      a.dpacked[m][(size_t)b * 2 * D + d] = dmu[m];
      a.dpacked[m][(size_t)b * 2 * D + D + d] = dsc[m];
    }
    dot += dsp[s] * sm[s];
  }
  dot = wave_sum(dot);
  if (dtheta_rows) {
#pragma unroll
    for (int s = 0; s < MOE_SLOTS; ++s) {
      const int d = lane + 64 * s;
      if (d < D) dtheta_rows[(size_t)b * D + d] = (float)D * sm[s] * (dsp[s] - dot);
    }
  }
}

extern "C" int mmvae_moe_ksample_fwd(const mmvae_moe_k_args* a, const float* theta, float* lat, float* pi, int M, int K,
                                     int B, int D, float beta, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(a && theta && lat && pi && K > 0 && B > 0 && D > 0);
  if (M < 2 || M > MMVAE_MOE_MAX_MODS || D > 64 * MOE_SLOTS) return MMVAE_ERR_UNSUPPORTED;
  for (int m = 0; m < M; ++m) MMVAE_CHECK_ARG(a->packed[m] && a->eps[m] && a->z[m]);
  const long rows = (long)M * K * B;
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  switch (M) {
    case 2: hipLaunchKernelGGL(moe_ksample_fwd_kernel<2>, grid, block, 0, st, *a, theta, lat, pi, K, B, D, beta); break;
    case 3: hipLaunchKernelGGL(moe_ksample_fwd_kernel<3>, grid, block, 0, st, *a, theta, lat, pi, K, B, D, beta); break;
    default: hipLaunchKernelGGL(moe_ksample_fwd_kernel<4>, grid, block, 0, st, *a, theta, lat, pi, K, B, D, beta); break;
  }
  return mmvae_launch_status();
}

extern "C" int mmvae_moe_ksample_bwd(const mmvae_moe_k_bwd_args* a, const float* theta, const float* dlat,
                                     const float* pi, float* dtheta_rows, int M, int K, int B, int D, float beta,
                                     mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(a && theta && dlat && pi && K > 0 && B > 0 && D > 0);
  if (M < 2 || M > MMVAE_MOE_MAX_MODS || D > 64 * MOE_SLOTS) return MMVAE_ERR_UNSUPPORTED;
  for (int m = 0; m < M; ++m) MMVAE_CHECK_ARG(a->packed[m] && a->eps[m] && a->z[m] && a->dpacked[m]);
  const dim3 grid((unsigned)((B + 3) / 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  switch (M) {
    case 2: hipLaunchKernelGGL(moe_ksample_bwd_kernel<2>, grid, block, 0, st, *a, theta, dlat, pi, dtheta_rows, K, B, D, beta); break;
    case 3: hipLaunchKernelGGL(moe_ksample_bwd_kernel<3>, grid, block, 0, st, *a, theta, dlat, pi, dtheta_rows, K, B, D, beta); break;
    default: hipLaunchKernelGGL(moe_ksample_bwd_kernel<4>, grid, block, 0, st, *a, theta, dlat, pi, dtheta_rows, K, B, D, beta); break;
  }
  return mmvae_launch_status();
}

// ---------------------------------------------------------------------------------------------
// DReG loss (objectives.py:375-387): lw[r,k] = sum_b lat[r,k,b] - lam_r (sum_b own_r[k,b] + sum_b cross_r[k,b])
// (own / cross = POSITIVE per-sample reconstruction sums, i.e. -lpx / llik_scaling), all sums in fp64 (the reference's
// lprob terms are double);  w = softmax_k(lw) (detached);  loss = -(1/M) sum_r sum_k w lw.
// out (doubles): [0] loss | lw (M*K) | w (M*K) | rec (M*2*K) = lpx_own, lpx_cross per r.  One workgroup.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dreg_loss_fwd_kernel(const float* __restrict__ lat, mmvae_dreg_rows rows,
                                                            double* __restrict__ out, int M, int K, int B) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* lw = out + 1;
  double* w = lw + (size_t)M * K;
  double* rec = w + (size_t)M * K;
  for (int p = wave; p < M * K; p += 4) {
    const int r = p / K, k = p - r * K;
    double sa = 0.0, so = 0.0, sc = 0.0;
    for (int b = lane; b < B; b += 64) {
      sa += (double)lat[(size_t)p * B + b];
      so += (double)rows.own[r][(size_t)k * B + b];
      sc += (double)rows.cross[r][(size_t)k * B + b];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      sa += __shfl_xor(sa, o, 64);
      so += __shfl_xor(so, o, 64);
      sc += __shfl_xor(sc, o, 64);
    }
    if (lane == 0) {
      const double lam = (double)rows.lam[r];
      rec[((size_t)r * 2 + 0) * K + k] = -lam * so;
      rec[((size_t)r * 2 + 1) * K + k] = -lam * sc;
      lw[p] = sa - lam * so - lam * sc;
    }
  }
  __syncthreads();
  __shared__ double part[MMVAE_MOE_MAX_MODS];
  if (threadIdx.x < M) {
    const int r = threadIdx.x;
    double mx = -INFINITY;
    for (int k = 0; k < K; ++k) mx = fmax(mx, lw[r * K + k]);
    double se = 0.0;
    for (int k = 0; k < K; ++k) se += exp(lw[r * K + k] - mx);
    const double lse = mx + log(se);
    double acc = 0.0;
    for (int k = 0; k < K; ++k) {
      const double wk = exp(lw[r * K + k] - lse);
      w[r * K + k] = wk;
      acc += wk * lw[r * K + k];
    }
    part[r] = acc;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int r = 0; r < M; ++r) t += part[r];
    out[0] = -t / (double)M;
  }
}

// backward: d loss / d lw[r,k] = -w[r,k] / M (w is detached)  =>  dlat[r,k,b] = -g w / M,
// d own_r[k,b] = d cross_r[k,b] = +g w lam_r / M
__global__ __launch_bounds__(256) void dreg_loss_bwd_kernel(const double* __restrict__ out, const double* __restrict__ g,
                                                            mmvae_dreg_rows_grad rows, float* __restrict__ dlat, int M,
                                                            int K, int B) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)M * K * B) return;
  const int p = (int)(i / B), b = (int)(i - (long)p * B);
  const int r = p / K, k = p - r * K;
  const double* w = out + 1 + (size_t)M * K;
  const double c = g[0] * w[p] / (double)M;
  dlat[i] = (float)(-c);
  const float v = (float)(c * (double)rows.lam[r]);
  rows.own[r][(size_t)k * B + b] = v;
  rows.cross[r][(size_t)k * B + b] = v;
}

extern "C" int mmvae_dreg_loss_fwd(const float* lat, const mmvae_dreg_rows* rows, double* out, int M, int K, int B,
                                   mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(lat && rows && out && K > 0 && B > 0);
  if (M < 2 || M > MMVAE_MOE_MAX_MODS) return MMVAE_ERR_UNSUPPORTED;
  for (int m = 0; m < M; ++m) MMVAE_CHECK_ARG(rows->own[m] && rows->cross[m]);
  hipLaunchKernelGGL(dreg_loss_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, lat, *rows, out, M, K, B);
  return mmvae_launch_status();
}

extern "C" int mmvae_dreg_loss_bwd(const double* out, const double* g, const mmvae_dreg_rows_grad* rows, float* dlat,
                                   int M, int K, int B, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(out && g && rows && dlat && K > 0 && B > 0);
  if (M < 2 || M > MMVAE_MOE_MAX_MODS) return MMVAE_ERR_UNSUPPORTED;
  for (int m = 0; m < M; ++m) MMVAE_CHECK_ARG(rows->own[m] && rows->cross[m]);
  const long n = (long)M * K * B;
  hipLaunchKernelGGL(dreg_loss_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, g,
                     *rows, dlat, M, K, B);
  return mmvae_launch_status();
}

// ---------------------------------------------------------------------------------------------
// IWAE loss (MultimodalObjective.iwae, objectives.py:342-359): per sample b over the M*K importance weights
//   lw[r,k,b] = lat[r,k,b] - lam_r (own_r[k,b] + cross_r[k,b])     (lat already holds log p(z) - beta lqz, :356)
//   loss = - sum_b ( logsumexp_{r,k} lw[.,.,b] - log(M K) )        (fp64: the reference's lprob terms are double)
// out (doubles): [0] loss | lse (B) | rec (M,2,K*B) = lpx_own, lpx_cross per r (the logged reconstruction_loss).
// One thread per sample (M K <= a few hundred serial fp64 terms, B threads), then one workgroup folds the B terms
// in a fixed order (no atomics).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void iwae_loss_rows_kernel(const float* __restrict__ lat, mmvae_dreg_rows rows,
                                                             double* __restrict__ out, int M, int K, int B) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  double* lse = out + 1;
  double* rec = lse + B;
  const size_t KB = (size_t)K * B;
  double mx = -INFINITY;
  for (int r = 0; r < M; ++r) {
    const double lam = (double)rows.lam[r];
    for (int k = 0; k < K; ++k) {
      const size_t i = (size_t)k * B + b;
      const double o = -lam * (double)rows.own[r][i], c = -lam * (double)rows.cross[r][i];
      rec[((size_t)r * 2 + 0) * KB + i] = o;
      rec[((size_t)r * 2 + 1) * KB + i] = c;
      mx = fmax(mx, (double)lat[(size_t)r * KB + i] + o + c);
    }
  }
  double se = 0.0;
  for (int r = 0; r < M; ++r)
    for (int k = 0; k < K; ++k) {
      const size_t i = (size_t)k * B + b;
      se += exp((double)lat[(size_t)r * KB + i] + rec[((size_t)r * 2 + 0) * KB + i] + rec[((size_t)r * 2 + 1) * KB + i] - mx);
    }
  lse[b] = mx + log(se);
}
__global__ __launch_bounds__(256) void iwae_loss_sum_kernel(double* __restrict__ out, int MK, int B) {
  __shared__ double part[256];
  double acc = 0.0;
  for (int b = threadIdx.x; b < B; b += 256) acc += out[1 + b];
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = -(part[0] - (double)B * log((double)MK));
}
// backward: d loss / d lw[r,k,b] = -g softmax_{rk}(lw)[b]  =>  dlat = that, d own = d cross = +g softmax lam_r
__global__ __launch_bounds__(256) void iwae_loss_bwd_kernel(const float* __restrict__ lat, const double* __restrict__ out,
                                                            const double* __restrict__ g, mmvae_dreg_rows_grad rows,
                                                            float* __restrict__ dlat, int M, int K, int B) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const size_t KB = (size_t)K * B;
  if (i >= (long)M * KB) return;
  const int r = (int)(i / KB);
  const size_t j = (size_t)i - (size_t)r * KB;
  const int b = (int)(j % B);
  const double* rec = out + 1 + B;
  const double lw = (double)lat[i] + rec[((size_t)r * 2 + 0) * KB + j] + rec[((size_t)r * 2 + 1) * KB + j];
  const double w = g[0] * exp(lw - out[1 + b]);
  dlat[i] = (float)(-w);
  const float v = (float)(w * (double)rows.lam[r]);
  rows.own[r][j] = v;
  rows.cross[r][j] = v;
}
extern "C" size_t mmvae_iwae_loss_out_doubles(int M, int K, int B) { return (size_t)1 + B + (size_t)M * 2 * K * B; }
extern "C" int mmvae_iwae_loss_fwd(const float* lat, const mmvae_dreg_rows* rows, double* out, int M, int K, int B,
                                   mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(lat && rows && out && K > 0 && B > 0);
  if (M < 2 || M > MMVAE_MOE_MAX_MODS) return MMVAE_ERR_UNSUPPORTED;
  for (int m = 0; m < M; ++m) MMVAE_CHECK_ARG(rows->own[m] && rows->cross[m]);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(iwae_loss_rows_kernel, dim3((B + 255) / 256), dim3(256), 0, st, lat, *rows, out, M, K, B);
  hipLaunchKernelGGL(iwae_loss_sum_kernel, dim3(1), dim3(256), 0, st, out, M * K, B);
  return mmvae_launch_status();
}
extern "C" int mmvae_iwae_loss_bwd(const float* lat, const double* out, const double* g, const mmvae_dreg_rows_grad* rows,
                                   float* dlat, int M, int K, int B, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(lat && out && g && rows && dlat && K > 0 && B > 0);
  if (M < 2 || M > MMVAE_MOE_MAX_MODS) return MMVAE_ERR_UNSUPPORTED;
  for (int m = 0; m < M; ++m) MMVAE_CHECK_ARG(rows->own[m] && rows->cross[m]);
  const long n = (long)M * K * B;
  hipLaunchKernelGGL(iwae_loss_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, lat,
                     out, g, *rows, dlat, M, K, B);
  return mmvae_launch_status();
}

// ---------------------------------------------------------------------------------------------
// KL(Laplace(mu, s) || Normal(0, 1)) summed over d (torch.distributions.kl._kl_laplace_normal through
// utils.kl_divergence, utils.py:399-402; MOE.objective :45 with `prior: laplace`): packed (B, 2D) -> kl (B)
//   per coordinate: -0.5 log(2 s^2 / pi) + s^2 + 0.5 mu^2 - 1
// ---------------------------------------------------------------------------------------------
#define PI_F 3.14159265358979323846f
__global__ __launch_bounds__(256) void kl_laplace_normal_fwd_kernel(const float* __restrict__ p, float* __restrict__ kl,
                                                                    int B, int D) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float acc = 0.f;
  for (int d = lane; d < D; d += 64) {
    const float mu = p[(size_t)b * 2 * D + d], s = p[(size_t)b * 2 * D + D + d];
    const float ratio = s * s;
    acc += -0.5f * logf(2.0f * ratio / PI_F) + ratio + 0.5f * (mu * mu) - 1.0f;
  }
  acc = wave_sum(acc);
  if (lane == 0) kl[b] = acc;
}
// d/dmu = mu;  d/ds = -1/s + 2 s
__global__ __launch_bounds__(256) void kl_laplace_normal_bwd_kernel(const float* __restrict__ p,
                                                                    const float* __restrict__ g, float* __restrict__ dp,
                                                                    int B, int D) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * D) return;
  const int b = i / D, d = i - b * D;
  const float mu = p[(size_t)b * 2 * D + d], s = p[(size_t)b * 2 * D + D + d], gb = g[b];
  dp[(size_t)b * 2 * D + d] = gb * mu;
  dp[(size_t)b * 2 * D + D + d] = gb * (2.0f * s - 1.0f / s);
}
extern "C" int mmvae_kl_laplace_normal_fwd(const float* packed, float* kl, int B, int D, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(packed && kl && B > 0 && D > 0);
  hipLaunchKernelGGL(kl_laplace_normal_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, packed, kl, B,
                     D);
  return mmvae_launch_status();
}
extern "C" int mmvae_kl_laplace_normal_bwd(const float* packed, const float* g, float* dpacked, int B, int D,
                                           mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(packed && g && dpacked && B > 0 && D > 0);
  hipLaunchKernelGGL(kl_laplace_normal_bwd_kernel, dim3((B * D + 255) / 256), dim3(256), 0, (hipStream_t)stream, packed,
                     g, dpacked, B, D);
  return mmvae_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Laplace importance ratio of the elbo objective under `prior: laplace` (models/mmvae_models.py:56-62 with
// qz_x = Laplace):  lw[b] = sum_d [ log Lap(z; mu_r, s_r) - log Lap(z; mu_o, s_o) ],  gradient into packed_r only.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void laplace_logratio_fwd_kernel(const float* __restrict__ pr,
                                                                   const float* __restrict__ po,
                                                                   const float* __restrict__ z, float* __restrict__ lw,
                                                                   int B, int D) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float a = 0.f;
  for (int d = lane; d < D; d += 64) {
    const float zz = z[(size_t)b * D + d];
    a += logq_elem(zz, pr[(size_t)b * 2 * D + d], pr[(size_t)b * 2 * D + D + d], 1) -
         logq_elem(zz, po[(size_t)b * 2 * D + d], po[(size_t)b * 2 * D + D + d], 1);
  }
  a = wave_sum(a);
  if (lane == 0) lw[b] = a;
}
__global__ __launch_bounds__(256) void laplace_logratio_bwd_kernel(const float* __restrict__ pr,
                                                                   const float* __restrict__ z,
                                                                   const float* __restrict__ g, float* __restrict__ dpr,
                                                                   int B, int D) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * D) return;
  const int b = i / D, d = i - b * D;
  const float df = z[i] - pr[(size_t)b * 2 * D + d], is = 1.0f / pr[(size_t)b * 2 * D + D + d], gb = g[b];
  const float sg = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
  dpr[(size_t)b * 2 * D + d] = gb * sg * is;
  dpr[(size_t)b * 2 * D + D + d] = gb * (-is + fabsf(df) * is * is);
}
extern "C" int mmvae_laplace_logratio_fwd(const float* packed_r, const float* packed_o, const float* z, float* lw,
                                          int B, int D, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(packed_r && packed_o && z && lw && B > 0 && D > 0);
  hipLaunchKernelGGL(laplace_logratio_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, packed_r,
                     packed_o, z, lw, B, D);
  return mmvae_launch_status();
}
extern "C" int mmvae_laplace_logratio_bwd(const float* packed_r, const float* z, const float* g, float* dpacked_r,
                                          int B, int D, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(packed_r && z && g && dpacked_r && B > 0 && D > 0);
  hipLaunchKernelGGL(laplace_logratio_bwd_kernel, dim3((B * D + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     packed_r, z, g, dpacked_r, B, D);
  return mmvae_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Standard-Laplace noise e = -sign(u) log1p(-|u|), u ~ U(-1, 1)  (torch.distributions.Laplace.rsample:
// z = loc + scale * e).  Same counter-based stream construction as mmvae_randn (state {seed, counter, ticket}; the
// last workgroup advances the counter); a distinct key constant keeps the two streams independent.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rand_laplace_kernel(float* __restrict__ out, long n,
                                                           uint32_t* __restrict__ state) {
  const uint32_t key = drop_fmix(state[0] ^ (state[1] * 0x9E3779B1u) ^ 0x1B873593u);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const uint32_t h = drop_fmix(key + (uint32_t)i * 0x9E3779B1u);
    const float u = ((float)(h >> 9) + 0.5f) * (1.0f / 4194304.0f) - 1.0f;      // (-1, 1) exactly: never 0 or +-1
    const float m = -log1pf(-fabsf(u));
    out[i] = u < 0.f ? -m : m;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t ticket = atomicAdd(state + 2, 1u);
    if (ticket == gridDim.x - 1) {
      state[2] = 0u;
      state[1] += 1u;
    }
  }
}
extern "C" int mmvae_rand_laplace(float* out, long n, uint32_t* state, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(out && state && n > 0);
  long blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(rand_laplace_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, n, state);
  return mmvae_launch_status();
}
