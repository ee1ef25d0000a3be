// Helper kernels of the ResNet-50 image tower (`encoder: CNN`, reference models/encoders.py:86-127 = torchvision's
// resnet50 -> SiLU -> heads).  Inside the tower activations are NHWC = (rows = B*H*W, C) matrices:
// every 1x1 convolution is then a plain GEMM on the MFMA kernels of gemm.hip (mmvae_linear_*), every k x k convolution
// an im2col pass + the same GEMMs (columns ordered (c, kh, kw) = the native (Cout, Cin, k, k) weight layout, so the
// weight tensor is used as it is stored), BatchNorm / pooling are HBM-bound column-statistics and elementwise kernels.
// Layers emit PRE-activations (package convention): the ReLU after a BatchNorm is applied by the consumer while it
// stages its input (GEMM x_act, im2col in_act, pool in_act) and by the data-gradient epilogues.
#include "common.hpp"

// ---------------------------------------------------------------------------------------------
// im2col / col2im, square kernel K, stride S, padding P.
//   x: (B,H,W,C) [nchw = 0] or (B,C,H,W) [nchw = 1, the stem's image input]
//   cols[(b,oh,ow), c*K*K + kh*K + kw] = act(x[b, oh*S-P+kh, ow*S-P+kw, c])   (0 outside)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ x, float* __restrict__ cols, int B, int H,
                                                     int W, int C, int K, int S, int P, int Ho, int Wo, int act,
                                                     int nchw) {
  const int KK = K * K, N = C * KK;
  const long total = (long)B * Ho * Wo * N;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int j = (int)(e % N);
    const long r = e / N;
    const int ow = (int)(r % Wo), oh = (int)((r / Wo) % Ho), b = (int)(r / ((long)Wo * Ho));
    const int c = j / KK, t = j - c * KK, kh = t / K, kw = t - kh * K;
    const int ih = oh * S - P + kh, iw = ow * S - P + kw;
    float v = 0.f;
    if (ih >= 0 && ih < H && iw >= 0 && iw < W) {
      v = nchw ? x[(((size_t)b * C + c) * H + ih) * W + iw] : x[(((size_t)b * H + ih) * W + iw) * C + c];
      v = apply_in_act(v, act);
    }
    cols[e] = v;
  }
}
// dx[b,h,w,c] = act'(x[b,h,w,c]) * sum over the taps that read this pixel of dcols[...]   (NHWC only)
__global__ __launch_bounds__(256) void col2im_kernel(const float* __restrict__ dcols, const float* __restrict__ x,
                                                     float* __restrict__ dx, int B, int H, int W, int C, int K, int S,
                                                     int P, int Ho, int Wo, int act) {
  const int KK = K * K, N = C * KK;
  const long total = (long)B * H * W * C;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = (int)(e % C);
    const long r = e / C;
    const int w = (int)(r % W), h = (int)((r / W) % H), b = (int)(r / ((long)W * H));
    float acc = 0.f;
    for (int kh = 0; kh < K; ++kh) {
      const int th = h + P - kh;
      if (th < 0 || th % S) continue;
      const int oh = th / S;
      if (oh >= Ho) continue;
      for (int kw = 0; kw < K; ++kw) {
        const int tw = w + P - kw;
        if (tw < 0 || tw % S) continue;
        const int ow = tw / S;
        if (ow >= Wo) continue;
        acc += dcols[(((size_t)b * Ho + oh) * Wo + ow) * N + c * KK + kh * K + kw];
      }
    }
    if (act == MMVAE_ACT_RELU) acc = x[e] > 0.f ? acc : 0.f;
    dx[e] = acc;
  }
}
static inline unsigned ew_blocks(long n) {
  long b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}
extern "C" int mmvae_im2col(const float* x, float* cols, int B, int H, int W, int C, int K, int S, int P, int in_act,
                            int nchw, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(x && cols && B > 0 && H > 0 && W > 0 && C > 0 && K > 0 && S > 0 && P >= 0);
  const int Ho = (H + 2 * P - K) / S + 1, Wo = (W + 2 * P - K) / S + 1;
  MMVAE_CHECK_ARG(Ho > 0 && Wo > 0);
  hipLaunchKernelGGL(im2col_kernel, dim3(ew_blocks((long)B * Ho * Wo * C * K * K)), dim3(256), 0, (hipStream_t)stream, x,
                     cols, B, H, W, C, K, S, P, Ho, Wo, in_act, nchw);
  return mmvae_launch_status();
}
extern "C" int mmvae_col2im(const float* dcols, const float* x, float* dx, int B, int H, int W, int C, int K, int S, int P,
                            int in_act, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(dcols && dx && B > 0 && H > 0 && W > 0 && C > 0 && K > 0 && S > 0 && P >= 0);
  if (in_act != MMVAE_ACT_NONE && in_act != MMVAE_ACT_RELU) return MMVAE_ERR_UNSUPPORTED;
  MMVAE_CHECK_ARG(in_act == MMVAE_ACT_NONE || x);
  const int Ho = (H + 2 * P - K) / S + 1, Wo = (W + 2 * P - K) / S + 1;
  hipLaunchKernelGGL(col2im_kernel, dim3(ew_blocks((long)B * H * W * C)), dim3(256), 0, (hipStream_t)stream, dcols, x,
                     dx, B, H, W, C, K, S, P, Ho, Wo, in_act);
  return mmvae_launch_status();
}

// ---------------------------------------------------------------------------------------------
// BatchNorm2d in training mode over an (M, C) NHWC matrix (torch.nn.BatchNorm2d: biased variance for the
// normalisation, unbiased for running_var, momentum 0.1, eps 1e-5; per-rank statistics under data parallelism).
//   stats:  workgroup (column tile of 64, row block) -> partial (mean, M2) of its rows, two sweeps (no cancellation)
//   apply:  every thread merges its column's partials (Chan), y = (x - mean) rstd gamma + beta [+ relu?(res)],
//           workgroup row 0 stores mean / rstd and updates the running statistics.
// ---------------------------------------------------------------------------------------------
// Row blocks: 64 rows each (4 rows per thread of the 16-slice kernels below, register resident), at most 256 blocks.
// (First version: 256 rows per block walked one scalar load at a time by 4 row slices -- 64 dependent memory round
// trips per sweep; at the shipped CdSprites+ batch of 24 every BatchNorm kernel took 20 us for < 6 MB and the four of
// them were 52 % of the step.)
#define BN_ROWS_PER_BLOCK 64
#define BN_MAX_BLOCKS 256
#define BN_RT 8                      // rows a thread may keep in registers (float4 each)
static inline int bn_row_blocks(int M) {
  int nb = (M + BN_ROWS_PER_BLOCK - 1) / BN_ROWS_PER_BLOCK;
  return nb > BN_MAX_BLOCKS ? BN_MAX_BLOCKS : (nb < 1 ? 1 : nb);
}
static inline int bn_rows_per(int M, int nblk) {
  const int rp = (M + nblk - 1) / nblk;
  return (rp + 15) / 16 * 16;        // a multiple of the 16 row slices
}
extern "C" int mmvae_bn_row_blocks(int M) { return bn_row_blocks(M); }
extern "C" size_t mmvae_bn_ws_floats(int M, int C) { return (size_t)bn_row_blocks(M) * C * 2; }

__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, float* __restrict__ part, int M,
                                                       int C, int rows_per) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane, blk = blockIdx.y;
  const int r0 = blk * rows_per, r1 = min(M, r0 + rows_per);
  const int n = max(r1 - r0, 0);
  float s = 0.f;
  if (c < C)
    for (int r = r0 + rl; r < r1; r += 4) s += x[(size_t)r * C + c];
  red[rl][lane] = s;
  __syncthreads();
  const float mean = n > 0 ? (red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]) / (float)n : 0.f;
  __syncthreads();
  float q = 0.f;
  if (c < C)
    for (int r = r0 + rl; r < r1; r += 4) {
      const float d = x[(size_t)r * C + c] - mean;
      q += d * d;
    }
  red[rl][lane] = q;
  __syncthreads();
  if (rl == 0 && c < C) {
    part[((size_t)blk * C + c) * 2] = mean;
    part[((size_t)blk * C + c) * 2 + 1] = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
  }
}
// merge of the row blocks' (count, mean, M2) for one column, in double
__device__ __forceinline__ void bn_merge(const float* __restrict__ part, int c, int C, int nblk, int M, int rows_per,
                                         double* mean_out, double* m2_out) {
  double n = 0.0, mean = 0.0, m2 = 0.0;
  for (int b = 0; b < nblk; ++b) {
    const int r0 = b * rows_per;
    const double nb = (double)max(min(M, r0 + rows_per) - r0, 0);
    if (nb <= 0.0) continue;
    const double mb = (double)part[((size_t)b * C + c) * 2], qb = (double)part[((size_t)b * C + c) * 2 + 1];
    const double d = mb - mean, tot = n + nb;
    mean += d * nb / tot;
    m2 += qb + d * d * n * nb / tot;
    n = tot;
  }
  *mean_out = mean;
  *m2_out = m2;
}
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ part,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ res, float* __restrict__ y,
                                                       float* __restrict__ save_mean, float* __restrict__ save_rstd,
                                                       float* __restrict__ run_mean, float* __restrict__ run_var,
                                                       int M, int C, int nblk, int rows_per, float eps, float momentum,
                                                       int res_relu, int eval_mode) {
  const int lane = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  if (c >= C) return;
  double mean = 0.0, m2 = 0.0;
  float mu, rstd;
  if (eval_mode) {      // model.eval(): the running statistics (torch.nn.BatchNorm2d with self.training == False)
    mu = run_mean[c];
    rstd = 1.0f / sqrtf(run_var[c] + eps);
  } else {
    bn_merge(part, c, C, nblk, M, rows_per, &mean, &m2);
    mu = (float)mean;
    rstd = (float)(1.0 / sqrt(m2 / (double)M + (double)eps));
  }
  if (blockIdx.y == 0 && rl == 0) {
    save_mean[c] = mu;
    save_rstd[c] = rstd;
    if (run_mean && !eval_mode) {
      run_mean[c] = (1.0f - momentum) * run_mean[c] + momentum * mu;
      const float unbiased = (float)(m2 / (double)(M > 1 ? M - 1 : 1));
      run_var[c] = (1.0f - momentum) * run_var[c] + momentum * unbiased;
    }
  }
  const float g = gamma[c] * rstd, bt = beta[c] - mu * gamma[c] * rstd;
  const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
  for (int r = r0 + rl; r < r1; r += 4) {
    float v = x[(size_t)r * C + c] * g + bt;
    if (res) {
      const float rv = res[(size_t)r * C + c];
      v += res_relu ? fmaxf(rv, 0.f) : rv;
    }
    y[(size_t)r * C + c] = v;
  }
}
// ---- 16-byte forms (C % 4 == 0): workgroup = 16 channel quads (64 channels) x 16 row slices ------------------------
// every load is a float4 over 4 channels; a thread's rows are independent loads issued back to back; when a row block
// has <= 16 * BN_RT rows the first sweep's values stay in registers for the second one.
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
// sum over the 16 row slices of a float4 per channel quad; red: [16][16] float4
__device__ __forceinline__ float4 bn_slice_sum(float4 v, float4* red, int cq, int rs) {
  __syncthreads();
  red[rs * 16 + cq] = v;
  __syncthreads();
  float4 t = red[cq];
#pragma unroll
  for (int i = 1; i < 16; ++i) t = f4add(t, red[i * 16 + cq]);
  return t;
}
__global__ __launch_bounds__(256) void bn_stats4_kernel(const float* __restrict__ x, float* __restrict__ part, int M,
                                                        int C, int rows_per) {
  __shared__ float4 red[256];
  const int cq = threadIdx.x & 15, rs = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + 4 * cq, blk = blockIdx.y;
  const bool cok = c < C;
  const int r0 = blk * rows_per, r1 = min(M, r0 + rows_per);
  const int n = max(r1 - r0, 0);
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  const float inv_n = n > 0 ? 1.0f / (float)n : 0.f;
  if (rows_per <= 16 * BN_RT) {
    float4 v[BN_RT];
#pragma unroll
    for (int i = 0; i < BN_RT; ++i) {
      const int r = r0 + rs + 16 * i;
      v[i] = (cok && r < r1) ? *reinterpret_cast<const float4*>(x + (size_t)r * C + c) : z;
    }
    float4 s = z;
#pragma unroll
    for (int i = 0; i < BN_RT; ++i) s = f4add(s, v[i]);
    s = bn_slice_sum(s, red, cq, rs);
    const float4 mean = make_float4(s.x * inv_n, s.y * inv_n, s.z * inv_n, s.w * inv_n);
    float4 q = z;
#pragma unroll
    for (int i = 0; i < BN_RT; ++i) {
      if (r0 + rs + 16 * i < r1) {
        const float dx = v[i].x - mean.x, dy = v[i].y - mean.y, dz = v[i].z - mean.z, dw = v[i].w - mean.w;
        q.x += dx * dx; q.y += dy * dy; q.z += dz * dz; q.w += dw * dw;
      }
    }
    q = bn_slice_sum(q, red, cq, rs);
    if (rs == 0 && cok) {
      float* o = part + ((size_t)blk * C + c) * 2;
      *reinterpret_cast<float4*>(o) = make_float4(mean.x, q.x, mean.y, q.y);
      *reinterpret_cast<float4*>(o + 4) = make_float4(mean.z, q.z, mean.w, q.w);
    }
    return;
  }
  float4 s = z;
  if (cok) {
    int r = r0 + rs;
    for (; r + 48 < r1; r += 64) {
      const float4 a = *reinterpret_cast<const float4*>(x + (size_t)r * C + c);
      const float4 b = *reinterpret_cast<const float4*>(x + (size_t)(r + 16) * C + c);
      const float4 d = *reinterpret_cast<const float4*>(x + (size_t)(r + 32) * C + c);
      const float4 e = *reinterpret_cast<const float4*>(x + (size_t)(r + 48) * C + c);
      s = f4add(s, f4add(f4add(a, b), f4add(d, e)));
    }
    for (; r < r1; r += 16) s = f4add(s, *reinterpret_cast<const float4*>(x + (size_t)r * C + c));
  }
  s = bn_slice_sum(s, red, cq, rs);
  const float4 mean = make_float4(s.x * inv_n, s.y * inv_n, s.z * inv_n, s.w * inv_n);
  float4 q = z;
  if (cok) {
    auto sq = [&](const float4& v) {
      const float dx = v.x - mean.x, dy = v.y - mean.y, dz = v.z - mean.z, dw = v.w - mean.w;
      q.x += dx * dx; q.y += dy * dy; q.z += dz * dz; q.w += dw * dw;
    };
    int r = r0 + rs;
    for (; r + 48 < r1; r += 64) {
      const float4 a = *reinterpret_cast<const float4*>(x + (size_t)r * C + c);
      const float4 b = *reinterpret_cast<const float4*>(x + (size_t)(r + 16) * C + c);
      const float4 d = *reinterpret_cast<const float4*>(x + (size_t)(r + 32) * C + c);
      const float4 e = *reinterpret_cast<const float4*>(x + (size_t)(r + 48) * C + c);
      sq(a); sq(b); sq(d); sq(e);
    }
    for (; r < r1; r += 16) sq(*reinterpret_cast<const float4*>(x + (size_t)r * C + c));
  }
  q = bn_slice_sum(q, red, cq, rs);
  if (rs == 0 && cok) {
    float* o = part + ((size_t)blk * C + c) * 2;
    *reinterpret_cast<float4*>(o) = make_float4(mean.x, q.x, mean.y, q.y);
    *reinterpret_cast<float4*>(o + 4) = make_float4(mean.z, q.z, mean.w, q.w);
  }
}
__global__ __launch_bounds__(256) void bn_apply4_kernel(const float* __restrict__ x, const float* __restrict__ part,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ res, float* __restrict__ y,
                                                        float* __restrict__ save_mean, float* __restrict__ save_rstd,
                                                        float* __restrict__ run_mean, float* __restrict__ run_var,
                                                        int M, int C, int nblk, int rows_per, float eps, float momentum,
                                                        int res_relu, int eval_mode) {
  __shared__ double sm[16][64];         // per row slice and channel: its share of a sum over the row blocks
  const int cq = threadIdx.x & 15, rs = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + 4 * cq;
  const bool cok = c < C;
  // this thread's rows: issued first, they fly under the merge of the statistics
  const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
  const bool regs = rows_per <= 16 * BN_RT;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 v[BN_RT], rv[BN_RT];
  if (regs) {
#pragma unroll
    for (int i = 0; i < BN_RT; ++i) {
      const int r = r0 + rs + 16 * i;
      const bool ok = cok && r < r1;
      v[i] = ok ? *reinterpret_cast<const float4*>(x + (size_t)r * C + c) : z;
      rv[i] = (ok && res) ? *reinterpret_cast<const float4*>(res + (size_t)r * C + c) : z;
    }
  }
  float mu[4], rstd[4];
  if (eval_mode) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      mu[k] = cok ? run_mean[c + k] : 0.f;
      rstd[k] = cok ? 1.0f / sqrtf(run_var[c + k] + eps) : 0.f;
    }
  } else {
    // exact two-pass combination of the row blocks' (count, mean, M2), in double and without a division per block
    // (a Chan merge per block and channel -- 128 double divisions per thread -- made this kernel 14 us when its three
    // siblings took 6): mean = sum n_b mean_b / M, M2 = sum [M2_b + n_b (mean_b - mean)^2].  Slice rs adds row blocks
    // rs, rs + 16, ...; the 16 slice sums are added in slice order.
    auto slice_total = [&](const double (&v)[4], double (&tot)[4]) {
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 4; ++k) sm[rs][4 * cq + k] = v[k];
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        double t = 0.0;
        for (int i = 0; i < 16; ++i) t += sm[i][4 * cq + k];
        tot[k] = t;
      }
    };
    double sw[4] = {0, 0, 0, 0}, mean[4], sq[4] = {0, 0, 0, 0}, m2[4];
    if (cok)
      for (int b = rs; b < nblk; b += 16) {
        const int br0 = b * rows_per;
        const double nb = (double)max(min(M, br0 + rows_per) - br0, 0);
        const float* pp = part + ((size_t)b * C + c) * 2;
        const float4 p0 = *reinterpret_cast<const float4*>(pp), p1 = *reinterpret_cast<const float4*>(pp + 4);
        sw[0] += nb * (double)p0.x; sw[1] += nb * (double)p0.z; sw[2] += nb * (double)p1.x; sw[3] += nb * (double)p1.z;
      }
    slice_total(sw, mean);
#pragma unroll
    for (int k = 0; k < 4; ++k) mean[k] /= (double)M;
    if (cok)
      for (int b = rs; b < nblk; b += 16) {
        const int br0 = b * rows_per;
        const double nb = (double)max(min(M, br0 + rows_per) - br0, 0);
        const float* pp = part + ((size_t)b * C + c) * 2;
        const float4 p0 = *reinterpret_cast<const float4*>(pp), p1 = *reinterpret_cast<const float4*>(pp + 4);
        const double d0 = (double)p0.x - mean[0], d1 = (double)p0.z - mean[1], d2 = (double)p1.x - mean[2],
                     d3 = (double)p1.z - mean[3];
        sq[0] += (double)p0.y + nb * d0 * d0; sq[1] += (double)p0.w + nb * d1 * d1;
        sq[2] += (double)p1.y + nb * d2 * d2; sq[3] += (double)p1.w + nb * d3 * d3;
      }
    slice_total(sq, m2);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      mu[k] = (float)mean[k];
      rstd[k] = (float)(1.0 / sqrt(m2[k] / (double)M + (double)eps));
      if (blockIdx.y == 0 && rs == 0 && cok && run_mean) {
        run_mean[c + k] = (1.0f - momentum) * run_mean[c + k] + momentum * mu[k];
        const float unbiased = (float)(m2[k] / (double)(M > 1 ? M - 1 : 1));
        run_var[c + k] = (1.0f - momentum) * run_var[c + k] + momentum * unbiased;
      }
    }
  }
  if (!cok) return;
  if (blockIdx.y == 0 && rs == 0) {
    *reinterpret_cast<float4*>(save_mean + c) = make_float4(mu[0], mu[1], mu[2], mu[3]);
    *reinterpret_cast<float4*>(save_rstd + c) = make_float4(rstd[0], rstd[1], rstd[2], rstd[3]);
  }
  const float4 gm = *reinterpret_cast<const float4*>(gamma + c), be = *reinterpret_cast<const float4*>(beta + c);
  const float g[4] = {gm.x * rstd[0], gm.y * rstd[1], gm.z * rstd[2], gm.w * rstd[3]};
  const float bt[4] = {be.x - mu[0] * g[0], be.y - mu[1] * g[1], be.z - mu[2] * g[2], be.w - mu[3] * g[3]};
  auto out = [&](int r, const float4& xv, const float4& resv) {
    float4 o = make_float4(xv.x * g[0] + bt[0], xv.y * g[1] + bt[1], xv.z * g[2] + bt[2], xv.w * g[3] + bt[3]);
    if (res) {
      o.x += res_relu ? fmaxf(resv.x, 0.f) : resv.x;
      o.y += res_relu ? fmaxf(resv.y, 0.f) : resv.y;
      o.z += res_relu ? fmaxf(resv.z, 0.f) : resv.z;
      o.w += res_relu ? fmaxf(resv.w, 0.f) : resv.w;
    }
    *reinterpret_cast<float4*>(y + (size_t)r * C + c) = o;
  };
  if (regs) {
#pragma unroll
    for (int i = 0; i < BN_RT; ++i) {
      const int r = r0 + rs + 16 * i;
      if (r < r1) out(r, v[i], rv[i]);
    }
  } else {
    for (int r = r0 + rs; r < r1; r += 64) {
      float4 a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int rr = r + 16 * u;
        a[u] = rr < r1 ? *reinterpret_cast<const float4*>(x + (size_t)rr * C + c) : z;
        b[u] = (rr < r1 && res) ? *reinterpret_cast<const float4*>(res + (size_t)rr * C + c) : z;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (r + 16 * u < r1) out(r + 16 * u, a[u], b[u]);
    }
  }
}
__global__ __launch_bounds__(256) void bn_bwd_stats4_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ save_mean,
                                                            const float* __restrict__ save_rstd, float* __restrict__ part,
                                                            int M, int C, int rows_per) {
  __shared__ float4 red[256];
  const int cq = threadIdx.x & 15, rs = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + 4 * cq, blk = blockIdx.y;
  const bool cok = c < C;
  const int r0 = blk * rows_per, r1 = min(M, r0 + rows_per);
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 s0 = z, s1 = z;
  if (cok) {
    const float4 mu = *reinterpret_cast<const float4*>(save_mean + c), rsd = *reinterpret_cast<const float4*>(save_rstd + c);
    auto acc = [&](const float4& g, const float4& xv) {
      s0 = f4add(s0, g);
      s1.x += g.x * ((xv.x - mu.x) * rsd.x); s1.y += g.y * ((xv.y - mu.y) * rsd.y);
      s1.z += g.z * ((xv.z - mu.z) * rsd.z); s1.w += g.w * ((xv.w - mu.w) * rsd.w);
    };
    for (int r = r0 + rs; r < r1; r += 64) {
      float4 g[4], xv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int rr = r + 16 * u;
        g[u] = rr < r1 ? *reinterpret_cast<const float4*>(dy + (size_t)rr * C + c) : z;
        xv[u] = rr < r1 ? *reinterpret_cast<const float4*>(x + (size_t)rr * C + c) : mu;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc(g[u], xv[u]);
    }
  }
  s0 = bn_slice_sum(s0, red, cq, rs);
  s1 = bn_slice_sum(s1, red, cq, rs);
  if (rs == 0 && cok) {
    *reinterpret_cast<float4*>(part + (size_t)blk * 2 * C + c) = s0;
    *reinterpret_cast<float4*>(part + (size_t)blk * 2 * C + C + c) = s1;
  }
}
__global__ __launch_bounds__(256) void bn_bwd_apply4_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ save_mean,
                                                            const float* __restrict__ save_rstd,
                                                            const float* __restrict__ part, const float* __restrict__ res,
                                                            float* __restrict__ dx, float* __restrict__ dres,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int M,
                                                            int C, int nblk, int rows_per, int res_relu, int accumulate,
                                                            int eval_mode) {
  __shared__ float4 red[256];
  const int cq = threadIdx.x & 15, rs = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + 4 * cq;
  const bool cok = c < C;
  const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
  const bool regs = rows_per <= 16 * BN_RT;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 gv[BN_RT], xv[BN_RT];
  if (regs) {
#pragma unroll
    for (int i = 0; i < BN_RT; ++i) {
      const int r = r0 + rs + 16 * i;
      const bool ok = cok && r < r1;
      gv[i] = ok ? *reinterpret_cast<const float4*>(dy + (size_t)r * C + c) : z;
      xv[i] = ok ? *reinterpret_cast<const float4*>(x + (size_t)r * C + c) : z;
    }
  }
  // column sums of the row blocks' partials: slice rs adds blocks rs, rs + 16, ...; then the 16 slices in order
  float4 a = z, b = z;
  if (cok)
    for (int k = rs; k < nblk; k += 16) {
      a = f4add(a, *reinterpret_cast<const float4*>(part + (size_t)k * 2 * C + c));
      b = f4add(b, *reinterpret_cast<const float4*>(part + (size_t)k * 2 * C + C + c));
    }
  const float4 sdy = bn_slice_sum(a, red, cq, rs);
  const float4 sdyx = bn_slice_sum(b, red, cq, rs);
  if (!cok) return;
  if (blockIdx.y == 0 && rs == 0 && accumulate != MMVAE_ACC_DEFER) {
    float4 og = sdyx, ob = sdy;
    if (accumulate) {
      og = f4add(og, *reinterpret_cast<const float4*>(dgamma + c));
      ob = f4add(ob, *reinterpret_cast<const float4*>(dbeta + c));
    }
    *reinterpret_cast<float4*>(dgamma + c) = og;
    *reinterpret_cast<float4*>(dbeta + c) = ob;
  }
  const float4 mu = *reinterpret_cast<const float4*>(save_mean + c), rsd = *reinterpret_cast<const float4*>(save_rstd + c);
  const float4 gm = *reinterpret_cast<const float4*>(gamma + c);
  const float invM = 1.0f / (float)M;
  const float gr[4] = {gm.x * rsd.x, gm.y * rsd.y, gm.z * rsd.z, gm.w * rsd.w};
  auto out = [&](int r, const float4& g, const float4& xx) {
    const size_t i = (size_t)r * C + c;
    float4 o;
    if (eval_mode) {
      o = make_float4(gr[0] * g.x, gr[1] * g.y, gr[2] * g.z, gr[3] * g.w);
    } else {
      o.x = gr[0] * (g.x - sdy.x * invM - ((xx.x - mu.x) * rsd.x) * sdyx.x * invM);
      o.y = gr[1] * (g.y - sdy.y * invM - ((xx.y - mu.y) * rsd.y) * sdyx.y * invM);
      o.z = gr[2] * (g.z - sdy.z * invM - ((xx.z - mu.z) * rsd.z) * sdyx.z * invM);
      o.w = gr[3] * (g.w - sdy.w * invM - ((xx.w - mu.w) * rsd.w) * sdyx.w * invM);
    }
    *reinterpret_cast<float4*>(dx + i) = o;
    if (dres) {
      float4 d = g;
      if (res_relu) {
        const float4 rv = *reinterpret_cast<const float4*>(res + i);
        d.x = rv.x > 0.f ? g.x : 0.f; d.y = rv.y > 0.f ? g.y : 0.f;
        d.z = rv.z > 0.f ? g.z : 0.f; d.w = rv.w > 0.f ? g.w : 0.f;
      }
      *reinterpret_cast<float4*>(dres + i) = d;
    }
  };
  if (regs) {
#pragma unroll
    for (int i = 0; i < BN_RT; ++i) {
      const int r = r0 + rs + 16 * i;
      if (r < r1) out(r, gv[i], xv[i]);
    }
  } else {
    for (int r = r0 + rs; r < r1; r += 64) {
      float4 g4[4], x4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int rr = r + 16 * u;
        g4[u] = rr < r1 ? *reinterpret_cast<const float4*>(dy + (size_t)rr * C + c) : z;
        x4[u] = rr < r1 ? *reinterpret_cast<const float4*>(x + (size_t)rr * C + c) : z;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (r + 16 * u < r1) out(r + 16 * u, g4[u], x4[u]);
    }
  }
}
static inline bool bn_vec_ok(int C, const void* a, const void* b, const void* c, const void* d) {
  return (C & 3) == 0 && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d) & 15) == 0;
}

extern "C" int mmvae_bn_train_fwd(const float* x, const float* gamma, const float* beta, const float* res, float* y,
                                  float* save_mean, float* save_rstd, float* run_mean, float* run_var, float* ws, int M,
                                  int C, float eps, float momentum, int res_relu, int eval_mode,
                                  mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(x && gamma && beta && y && save_mean && save_rstd && ws && M > 0 && C > 0);
  MMVAE_CHECK_ARG(!eval_mode || (run_mean && run_var));
  const int nblk = bn_row_blocks(M), rows_per = bn_rows_per(M, nblk);
  const dim3 grid((C + 63) / 64, nblk);
  if (bn_vec_ok(C, x, y, res, ws) && bn_vec_ok(C, gamma, beta, save_mean, save_rstd)) {
    if (!eval_mode) hipLaunchKernelGGL(bn_stats4_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, ws, M, C, rows_per);
    hipLaunchKernelGGL(bn_apply4_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, ws, gamma, beta, res, y, save_mean,
                       save_rstd, run_mean, run_var, M, C, nblk, rows_per, eps, momentum, res_relu, eval_mode);
    return mmvae_launch_status();
  }
  if (!eval_mode) hipLaunchKernelGGL(bn_stats_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, ws, M, C, rows_per);
  hipLaunchKernelGGL(bn_apply_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, ws, gamma, beta, res, y, save_mean,
                     save_rstd, run_mean, run_var, M, C, nblk, rows_per, eps, momentum, res_relu, eval_mode);
  return mmvae_launch_status();
}
// backward.  xhat = (x - mean) rstd;  dgamma = sum dy xhat, dbeta = sum dy;
//   dx = gamma rstd (dy - dbeta / M - xhat dgamma / M);   dres = dy [* (res > 0) when the residual went through a ReLU]
// stage 1: row-block partials of (sum dy, sum dy xhat) -> part (nblk, C, 2); stage 2: every thread sums its column's
// partials, writes dx (and dres); row block 0 writes dgamma / dbeta partial-free (or leaves the partials for the
// caller's fold: accumulate = MMVAE_ACC_DEFER).
__global__ __launch_bounds__(256) void bn_bwd_stats_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ save_mean,
                                                           const float* __restrict__ save_rstd, float* __restrict__ part,
                                                           int M, int C, int rows_per) {
  __shared__ float red[2][4][64];
  const int lane = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane, blk = blockIdx.y;
  const int r0 = blk * rows_per, r1 = min(M, r0 + rows_per);
  float s0 = 0.f, s1 = 0.f;
  if (c < C) {
    const float mu = save_mean[c], rs = save_rstd[c];
    for (int r = r0 + rl; r < r1; r += 4) {
      const float g = dy[(size_t)r * C + c];
      s0 += g;
      s1 += g * ((x[(size_t)r * C + c] - mu) * rs);
    }
  }
  red[0][rl][lane] = s0;
  red[1][rl][lane] = s1;
  __syncthreads();
  if (rl == 0 && c < C) {
    part[(size_t)blk * 2 * C + c] = red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane];
    part[(size_t)blk * 2 * C + C + c] = red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane];
  }
}
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ save_mean,
                                                           const float* __restrict__ save_rstd,
                                                           const float* __restrict__ part, const float* __restrict__ res,
                                                           float* __restrict__ dx, float* __restrict__ dres,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, int M,
                                                           int C, int nblk, int rows_per, int res_relu, int accumulate,
                                                           int eval_mode) {
  const int lane = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  if (c >= C) return;
  float sdy = 0.f, sdyx = 0.f;
  for (int b = 0; b < nblk; ++b) {
    sdy += part[(size_t)b * 2 * C + c];
    sdyx += part[(size_t)b * 2 * C + C + c];
  }
  if (blockIdx.y == 0 && rl == 0 && accumulate != MMVAE_ACC_DEFER) {
    dgamma[c] = accumulate ? dgamma[c] + sdyx : sdyx;
    dbeta[c] = accumulate ? dbeta[c] + sdy : sdy;
  }
  const float mu = save_mean[c], rs = save_rstd[c], gr = gamma[c] * rs, invM = 1.0f / (float)M;
  const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
  for (int r = r0 + rl; r < r1; r += 4) {
    const size_t i = (size_t)r * C + c;
    const float g = dy[i];
    const float xh = (x[i] - mu) * rs;
    dx[i] = eval_mode ? gr * g : gr * (g - sdy * invM - xh * sdyx * invM);   // eval: the statistics are constants
    if (dres) dres[i] = (res_relu && !(res[i] > 0.f)) ? 0.f : g;
  }
}
extern "C" int mmvae_bn_train_bwd(const float* dy, const float* x, const float* gamma, const float* save_mean,
                                  const float* save_rstd, const float* res, float* dx, float* dres, float* dgamma,
                                  float* dbeta, float* ws, int M, int C, int res_relu, int accumulate, int eval_mode,
                                  mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(dy && x && gamma && save_mean && save_rstd && dx && ws && M > 0 && C > 0);
  MMVAE_CHECK_ARG(accumulate == MMVAE_ACC_DEFER || (dgamma && dbeta));
  MMVAE_CHECK_ARG(!dres || !res_relu || res);
  const int nblk = bn_row_blocks(M), rows_per = bn_rows_per(M, nblk);
  const dim3 grid((C + 63) / 64, nblk);
  if (bn_vec_ok(C, dy, x, dx, ws) && bn_vec_ok(C, gamma, save_mean, save_rstd, res) && bn_vec_ok(C, dres, dgamma, dbeta, nullptr)) {
    hipLaunchKernelGGL(bn_bwd_stats4_kernel, grid, dim3(256), 0, (hipStream_t)stream, dy, x, save_mean, save_rstd, ws, M,
                       C, rows_per);
    hipLaunchKernelGGL(bn_bwd_apply4_kernel, grid, dim3(256), 0, (hipStream_t)stream, dy, x, gamma, save_mean, save_rstd,
                       ws, res, dx, dres, dgamma, dbeta, M, C, nblk, rows_per, res_relu, accumulate, eval_mode);
    return mmvae_launch_status();
  }
  hipLaunchKernelGGL(bn_bwd_stats_kernel, grid, dim3(256), 0, (hipStream_t)stream, dy, x, save_mean, save_rstd, ws, M, C,
                     rows_per);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, grid, dim3(256), 0, (hipStream_t)stream, dy, x, gamma, save_mean, save_rstd, ws,
                     res, dx, dres, dgamma, dbeta, M, C, nblk, rows_per, res_relu, accumulate, eval_mode);
  return mmvae_launch_status();
}

// ---------------------------------------------------------------------------------------------
// MaxPool2d(3, stride 2, padding 1) on relu(x), NHWC; idx = flat input position (h*W + w) of the FIRST maximum in
// window scan order (torch.nn.functional.max_pool2d: strict '>' comparison), kept for the backward pass.
// backward (gather form, deterministic): dx[b,h,w,c] = (x > 0) * sum over the <= 4 windows containing (h,w) whose
// recorded maximum is this pixel.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          int* __restrict__ idx, int B, int H, int W, int C, int Ho,
                                                          int Wo, int act) {
  const long total = (long)B * Ho * Wo * C;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = (int)(e % C);
    const long r = e / C;
    const int ow = (int)(r % Wo), oh = (int)((r / Wo) % Ho), b = (int)(r / ((long)Wo * Ho));
    float best = -INFINITY;
    int bi = -1;
    for (int kh = 0; kh < 3; ++kh) {
      const int ih = 2 * oh - 1 + kh;
      if (ih < 0 || ih >= H) continue;
      for (int kw = 0; kw < 3; ++kw) {
        const int iw = 2 * ow - 1 + kw;
        if (iw < 0 || iw >= W) continue;
        const float v = apply_in_act(x[(((size_t)b * H + ih) * W + iw) * C + c], act);
        if (v > best || bi < 0) {
          best = v;
          bi = ih * W + iw;
        }
      }
    }
    y[e] = best;
    idx[e] = bi;
  }
}
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const int* __restrict__ idx,
                                                          const float* __restrict__ x, float* __restrict__ dx, int B,
                                                          int H, int W, int C, int Ho, int Wo, int act) {
  const long total = (long)B * H * W * C;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = (int)(e % C);
    const long r = e / C;
    const int w = (int)(r % W), h = (int)((r / W) % H), b = (int)(r / ((long)W * H));
    float acc = 0.f;
    const int me = h * W + w;
    for (int oh = (h + 1 - 2 + 1) / 2; oh <= (h + 1) / 2; ++oh) {      // windows with 2 oh - 1 <= h <= 2 oh + 1
      if (oh < 0 || oh >= Ho) continue;
      for (int ow = (w) / 2; ow <= (w + 1) / 2; ++ow) {
        if (ow < 0 || ow >= Wo) continue;
        const size_t o = (((size_t)b * Ho + oh) * Wo + ow) * C + c;
        if (idx[o] == me) acc += dy[o];
      }
    }
    if (act == MMVAE_ACT_RELU && !(x[e] > 0.f)) acc = 0.f;
    dx[e] = acc;
  }
}
extern "C" int mmvae_maxpool3x3s2_fwd(const float* x, float* y, int* idx, int B, int H, int W, int C, int in_act,
                                      mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(x && y && idx && B > 0 && H > 1 && W > 1 && C > 0);
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(ew_blocks((long)B * Ho * Wo * C)), dim3(256), 0, (hipStream_t)stream, x, y,
                     idx, B, H, W, C, Ho, Wo, in_act);
  return mmvae_launch_status();
}
extern "C" int mmvae_maxpool3x3s2_bwd(const float* dy, const int* idx, const float* x, float* dx, int B, int H, int W,
                                      int C, int in_act, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(dy && idx && x && dx && B > 0 && H > 1 && W > 1 && C > 0);
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(ew_blocks((long)B * H * W * C)), dim3(256), 0, (hipStream_t)stream, dy, idx,
                     x, dx, B, H, W, C, Ho, Wo, in_act);
  return mmvae_launch_status();
}

// ---------------------------------------------------------------------------------------------
// AdaptiveAvgPool2d(1) on relu(x): (B, HW, C) -> (B, C), and its backward
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B,
                                                          int HW, int C, int act) {
  const long total = (long)B * C;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = (int)(e % C), b = (int)(e / C);
    float s = 0.f;
    for (int p = 0; p < HW; ++p) s += apply_in_act(x[((size_t)b * HW + p) * C + c], act);
    y[e] = s / (float)HW;
  }
}
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          float* __restrict__ dx, int B, int HW, int C, int act) {
  const long total = (long)B * HW * C;
  const float inv = 1.0f / (float)HW;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = (int)(e % C), b = (int)(e / ((long)HW * C));
    float g = dy[(size_t)b * C + c] * inv;
    if (act == MMVAE_ACT_RELU && !(x[e] > 0.f)) g = 0.f;
    dx[e] = g;
  }
}
extern "C" int mmvae_avgpool_fwd(const float* x, float* y, int B, int HW, int C, int in_act, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(x && y && B > 0 && HW > 0 && C > 0);
  hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(ew_blocks((long)B * C)), dim3(256), 0, (hipStream_t)stream, x, y, B, HW, C,
                     in_act);
  return mmvae_launch_status();
}
extern "C" int mmvae_avgpool_bwd(const float* dy, const float* x, float* dx, int B, int HW, int C, int in_act,
                                 mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(dy && x && dx && B > 0 && HW > 0 && C > 0);
  hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(ew_blocks((long)B * HW * C)), dim3(256), 0, (hipStream_t)stream, dy, x, dx,
                     B, HW, C, in_act);
  return mmvae_launch_status();
}
