// Flat-buffer Adam(amsgrad) + partial-sum reduction + fill, gfx950.  Pure HBM streaming: 16-byte
// accesses, grid capped at 2048 blocks with a grid-stride loop.
#include "common.hpp"

// torch.optim.Adam(amsgrad=True) (reference: models/trainer.py:79-81), torch's single-tensor formula:
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; vmax = max(vmax, v)
//   p -= (lr / bc1) * m / (sqrt(vmax) / sqrt(bc2) + eps)
// ONE spelling of the update for every kernel that applies it (adam_amsgrad_kernel, adam_fold_kernel): explicit fused
// multiply-adds and no compiler contraction, so that the same inputs give the same bits wherever the update is inlined
// (left to -ffp-contract=fast the quotient's denominator was fused in one kernel and not in the other: 1-ulp differences
// in a handful of parameters per step).
__device__ __forceinline__ void adam_update1(float& P, const float G, float& M, float& V, float& X, const float b1,
                                             const float b2, const float eps, const float gscale, const float lr_bc1,
                                             const float inv_sqrt_bc2) {
#pragma clang fp contract(off)
  const float gr = G * gscale;
  M = fmaf(b1, M, (1.0f - b1) * gr);
  V = fmaf(b2, V, ((1.0f - b2) * gr) * gr);
  X = fmaxf(X, V);
  const float den = fmaf(sqrtf(X), inv_sqrt_bc2, eps);
  P = P - (lr_bc1 * M) / den;
}

__global__ __launch_bounds__(256) void adam_amsgrad_kernel(float* __restrict__ p, float* __restrict__ g,
                                                           float* __restrict__ m, float* __restrict__ v,
                                                           float* __restrict__ vmax, long n, float lr,
                                                           int step, int* __restrict__ step_dev, float b1,
                                                           float b2, float eps, float gscale, int zero_grad) {
  MMVAE_TRACE_STAMP(20);
  // bias corrections from the step count; the count lives in device memory when the launch is replayed
  // from a captured graph (kernel arguments are frozen at capture time).
  __shared__ float bc[2];
  __shared__ double pw[2];
  if (threadIdx.x == 0) {
    // step < 0 with a device counter: this launch IS step (*step_dev + 1); the last workgroup to finish stores it.
    // In that mode the buffer also carries beta1^count, beta2^count as doubles (bytes 8..23; zero = not there yet):
    // one multiply per step instead of two double-precision pow() in front of every launch (~2 us of a ~15 us kernel).
    const int st = step_dev ? step_dev[0] + (step < 0 ? 1 : 0) : step;
    double p1, p2;
    if (step_dev && step < 0) {
      const double* run = reinterpret_cast<const double*>(step_dev + 2);
      const double r1 = run[0], r2 = run[1];
      p1 = st == 1 ? (double)b1 : (r1 > 0.0 ? r1 * (double)b1 : pow((double)b1, (double)st));
      p2 = st == 1 ? (double)b2 : (r2 > 0.0 ? r2 * (double)b2 : pow((double)b2, (double)st));
      pw[0] = p1;
      pw[1] = p2;
    } else {
      p1 = pow((double)b1, (double)st);
      p2 = pow((double)b2, (double)st);
    }
    bc[0] = (float)((double)lr / (1.0 - p1));
    bc[1] = (float)(1.0 / sqrt(1.0 - p2));
  }
  __syncthreads();
  const float lr_bc1 = bc[0], inv_sqrt_bc2 = bc[1];
  const long n4 = n >> 2;
  const long stride = (long)gridDim.x * 256;
  auto update4 = [&](float4& P, const float4& G, float4& M, float4& Vv, float4& X) {
    float* pp = &P.x;
    const float* gg = &G.x;
    float* mm = &M.x;
    float* vv = &Vv.x;
    float* xx = &X.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) adam_update1(pp[k], gg[k], mm[k], vv[k], xx[k], b1, b2, eps, gscale, lr_bc1, inv_sqrt_bc2);
  };
  float4* p4 = reinterpret_cast<float4*>(p);
  float4* g4 = reinterpret_cast<float4*>(g);
  float4* m4 = reinterpret_cast<float4*>(m);
  float4* v4 = reinterpret_cast<float4*>(v);
  float4* x4 = reinterpret_cast<float4*>(vmax);
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  // two float4 per array per thread and iteration: ten 16-byte loads in flight before the first use
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  for (; i + stride < n4; i += 2 * stride) {
    const long j = i + stride;
    float4 P0 = p4[i], G0 = g4[i], M0 = m4[i], V0 = v4[i], X0 = x4[i];
    float4 P1 = p4[j], G1 = g4[j], M1 = m4[j], V1 = v4[j], X1 = x4[j];
    update4(P0, G0, M0, V0, X0);
    update4(P1, G1, M1, V1, X1);
    p4[i] = P0; m4[i] = M0; v4[i] = V0; x4[i] = X0;
    p4[j] = P1; m4[j] = M1; v4[j] = V1; x4[j] = X1;
    if (zero_grad) { g4[i] = zero4; g4[j] = zero4; }
  }
  if (i < n4) {
    float4 P0 = p4[i], G0 = g4[i], M0 = m4[i], V0 = v4[i], X0 = x4[i];
    update4(P0, G0, M0, V0, X0);
    p4[i] = P0; m4[i] = M0; v4[i] = V0; x4[i] = X0;
    if (zero_grad) g4[i] = zero4;
  }
  // tail
  for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    float pi = p[i], mi = m[i], vi = v[i], xi = vmax[i];
    adam_update1(pi, g[i], mi, vi, xi, b1, b2, eps, gscale, lr_bc1, inv_sqrt_bc2);
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
    vmax[i] = xi;
    if (zero_grad) g[i] = 0.f;
  }
  if (step_dev && step < 0) {     // {count, finished-workgroup ticket}: no separate counter-bump launch
    __syncthreads();
    if (threadIdx.x == 0) {
      const int ticket = atomicAdd(step_dev + 1, 1);
      if (ticket == (int)gridDim.x - 1) {
        step_dev[1] = 0;
        step_dev[0] += 1;
        double* run = reinterpret_cast<double*>(step_dev + 2);
        run[0] = pw[0];
        run[1] = pw[1];
      }
    }
  }
}

extern "C" int mmvae_adam_amsgrad_flat(float* p, float* g, float* m, float* v, float* vmax, long n, float lr,
                                       float beta1, float beta2, float eps, int step, int* step_dev,
                                       float grad_scale, int zero_grad, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(p && g && m && v && vmax && n > 0 && (step > 0 || step_dev));
  if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)vmax) & 15) != 0) return MMVAE_ERR_ARG;
  if (step < 0 && (((uintptr_t)step_dev) & 7) != 0) return MMVAE_ERR_ARG;
  long blocks = ((n >> 3) + 255) / 256;    // two float4 per thread
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adam_amsgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, vmax,
                     n, lr, step, step_dev, beta1, beta2, eps, grad_scale, zero_grad);
  return mmvae_launch_status();
}

// AdaBelief (Zhuang et al., NeurIPS 2020, Algorithm 2) as the reference configures it: `optimizer: adabelief` ->
// adabelief_pytorch.AdaBelief(lr, eps=1e-16, betas=(0.9, 0.999), weight_decouple=True, rectify=False) (reference
// models/trainer.py:82-86; the package is a dependency the reference does not vendor and this image does not have:
// PARITY UNPINNED, restated from the paper and the package's update order, weight_decay = 0 = its default):
//   m = b1 m + (1-b1) g ; s = b2 s + (1-b2) (g - m)^2 + eps   (the package adds eps to the stored s, as the paper does)
//   p -= (lr / bc1) * m / (sqrt(s) / sqrt(bc2) + eps)
// (1 - beta) comes from the host in double -> float, as torch hands `value=1 - beta2` to its kernels: 1.0f - 0.999f is
// 1.3e-5 off 0.001)
__device__ __forceinline__ void adabelief_update1(float& P, const float G, float& M, float& S, const float b1, const float b2,
                                                  const float omb1, const float omb2, const float eps, const float gscale,
                                                  const float lr_bc1, const float inv_sqrt_bc2) {
#pragma clang fp contract(off)
  const float gr = G * gscale;
  M = fmaf(b1, M, omb1 * gr);
  const float d = gr - M;
  S = fmaf(b2, S, (omb2 * d) * d) + eps;
  const float den = fmaf(sqrtf(S), inv_sqrt_bc2, eps);
  P = P - (lr_bc1 * M) / den;
}

// same step-count protocol as adam_amsgrad_kernel (step < 0: device block {count, ticket, beta1^count, beta2^count})
__global__ __launch_bounds__(256) void adabelief_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ sv, long n, float lr, int step,
                                                        int* __restrict__ step_dev, float b1, float b2, float omb1,
                                                        float omb2, float eps, float gscale, int zero_grad) {
  __shared__ float bc[2];
  __shared__ double pw[2];
  if (threadIdx.x == 0) {
    const int st = step_dev ? step_dev[0] + (step < 0 ? 1 : 0) : step;
    double p1, p2;
    if (step_dev && step < 0) {
      const double* run = reinterpret_cast<const double*>(step_dev + 2);
      const double r1 = run[0], r2 = run[1];
      p1 = st == 1 ? (double)b1 : (r1 > 0.0 ? r1 * (double)b1 : pow((double)b1, (double)st));
      p2 = st == 1 ? (double)b2 : (r2 > 0.0 ? r2 * (double)b2 : pow((double)b2, (double)st));
      pw[0] = p1;
      pw[1] = p2;
    } else {
      p1 = pow((double)b1, (double)st);
      p2 = pow((double)b2, (double)st);
    }
    bc[0] = (float)((double)lr / (1.0 - p1));
    bc[1] = (float)(1.0 / sqrt(1.0 - p2));
  }
  __syncthreads();
  const float lr_bc1 = bc[0], inv_sqrt_bc2 = bc[1];
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    float pi = p[i], mi = m[i], si = sv[i];
    adabelief_update1(pi, g[i], mi, si, b1, b2, omb1, omb2, eps, gscale, lr_bc1, inv_sqrt_bc2);
    p[i] = pi;
    m[i] = mi;
    sv[i] = si;
    if (zero_grad) g[i] = 0.f;
  }
  if (step_dev && step < 0) {
    __syncthreads();
    if (threadIdx.x == 0) {
      const int ticket = atomicAdd(step_dev + 1, 1);
      if (ticket == (int)gridDim.x - 1) {
        step_dev[1] = 0;
        step_dev[0] += 1;
        double* run = reinterpret_cast<double*>(step_dev + 2);
        run[0] = pw[0];
        run[1] = pw[1];
      }
    }
  }
}

extern "C" int mmvae_adabelief_flat(float* p, float* g, float* m, float* s, long n, float lr, double beta1, double beta2,
                                    float eps, int step, int* step_dev, float grad_scale, int zero_grad,
                                    mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(p && g && m && s && n > 0 && (step > 0 || step_dev));
  if (step < 0 && (((uintptr_t)step_dev) & 7) != 0) return MMVAE_ERR_ARG;
  long blocks = (n + 1023) / 1024;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adabelief_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, s, n, lr, step,
                     step_dev, (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), eps, grad_scale,
                     zero_grad);
  return mmvae_launch_status();
}

// dst[i] (+)= sum_r src[r*stride + i].  Block = 64 columns x 4 row slices; every thread keeps 8 independent
// loads in flight (a serial "a += src[r]" chain costs one memory latency per row: 26 us for 64 rows).
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                          int n_rows, long len, long stride, int accumulate) {
  __shared__ float part[4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  for (long c0 = (long)blockIdx.x * 64; c0 < len; c0 += (long)gridDim.x * 64) {
    const long i = c0 + cx;
    float a = 0.f;
    if (i < len) {
      int r = ry;
      for (; r + 28 < n_rows; r += 32) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(r + 4 * u) * stride + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) a += v[u];
      }
      for (; r < n_rows; r += 4) a += src[(size_t)r * stride + i];
    }
    part[ry][cx] = a;
    __syncthreads();
    if (ry == 0 && i < len) {
      const float t = part[0][cx] + part[1][cx] + part[2][cx] + part[3][cx];
      dst[i] = accumulate ? dst[i] + t : t;
    }
    __syncthreads();
  }
}
extern "C" int mmvae_reduce_rows(const float* src, float* dst, int n_rows, long len, long stride, int accumulate,
                                 mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(src && dst && n_rows > 0 && len > 0);
  long blocks = (len + 63) / 64;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, n_rows,
                     len, stride, accumulate);
  return mmvae_launch_status();
}

// Many partial-sum reductions in ONE launch: dst_s[i] += sum_r src_s[r*stride_s + i] for up to 64 segments.
// Used at the end of backward: every weight-gradient kernel of the step leaves its split partials in a private
// region of the step arena and registers a segment; a single launch folds all of them into the flat gradient
// buffer (40 tiny reduce launches per step otherwise).  Same 64-column x 4-row-slice blocks as reduce_rows.
// A block covers RS_COLS = 256 columns of one head segment: 64 lanes x float4 (16-byte loads when the whole chain is
// 16-byte aligned and a multiple of 4 long, else four dword columns per lane) x 4 row slices, 8 loads in flight.
#define RS_COLS 256
// Optional rider: the ELBO assembly out_k = sum_n W[k][n] * sum_b rows_n[b] (mmvae_lincomb_rowptrs_fwd) as ONE extra
// workgroup of the same launch.  The logged loss values only need the forward's row sums, but as a launch of their own
// they sat in the serial tail of the step between this fold and the optimiser.
#define RS_LC_ROWS 32
#define RS_LC_OUT 4
struct rs_lincomb_t {
  const float* rows[RS_LC_ROWS];
  float w[RS_LC_OUT * RS_LC_ROWS];
  float* out;
  int n_rows, B, n_out;
};
__device__ __forceinline__ void rs_lincomb_body(const rs_lincomb_t& lc) {
  __shared__ float rs[RS_LC_ROWS];
  // one wave per row (rows are short: B values), 4 rows in flight
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int n = wave; n < lc.n_rows; n += 4) {
    const float* V = lc.rows[n];
    float a = 0.f;
    for (int b = lane; b < lc.B; b += 64) a += V[b];
    a = wave_sum(a);
    if (lane == 0) rs[n] = a;
  }
  __syncthreads();
  if ((int)threadIdx.x < lc.n_out) {
    float o = 0.f;
    for (int n = 0; n < lc.n_rows; ++n) o += lc.w[threadIdx.x * RS_LC_ROWS + n] * rs[n];
    lc.out[threadIdx.x] = o;
  }
}

// Block geometry of the partial-sum kernels: NRG row groups x (256 / NRG) float4 lanes = 1024 / NRG columns.  Wide (NRG 4,
// 256 columns) for the usual 16 - 64 partial rows; narrow (NRG 16, 64 columns) for segments with many rows (the
// 3-channel conv layers leave 512).  The row walk is a block's serial chain -- 8 loads in flight per step -- and a
// 512-row segment took 16 steps in the wide form: the longest chain of the whole end-of-step launch (the step's time
// followed the row count at ~0.6 us per 32 rows).
#define RS_NARROW_ROWS 128
__host__ __device__ static inline int rs_cols(bool narrow) { return narrow ? 64 : RS_COLS; }

// a += sum over rows ry, ry + NRG, ... of the 4 columns [i, i + 4) of one partial region
template <int NRG>
__device__ __forceinline__ void rs_row_sum(const float* __restrict__ src, const int n_rows, const long stride, const long i,
                                           const long len, const bool vec, const int ry, float4& a) {
  if (i >= len) return;
  if (vec) {
    int r = ry;
    for (; r + 7 * NRG < n_rows; r += 8 * NRG) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(src + (size_t)(r + NRG * u) * stride + i);
#pragma unroll
      for (int u = 0; u < 8; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
    }
    for (; r < n_rows; r += NRG) {
      const float4 v = *reinterpret_cast<const float4*>(src + (size_t)r * stride + i);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  } else {     // unaligned chain: the same 4 columns as dwords, still 32 independent loads in flight
    float* ap = &a.x;
    int r = ry;
    for (; r + 7 * NRG < n_rows; r += 8 * NRG) {
      float v[8][4];
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c) v[u][c] = src[(size_t)(r + NRG * u) * stride + (i + c < len ? i + c : i)];
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c) ap[c] += v[u][c];
    }
    for (; r < n_rows; r += NRG) {
      float v[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = src[(size_t)r * stride + (i + c < len ? i + c : i)];
#pragma unroll
      for (int c = 0; c < 4; ++c) ap[c] += v[c];
    }
  }
}

template <bool TAIL>
__global__ __launch_bounds__(256) void reduce_segments_kernel(mmvae_reduce_segments_t t, rs_lincomb_t lc,
                                                              unsigned long long narrow) {
  MMVAE_TRACE_STAMP(21);
  if (TAIL && blockIdx.x == gridDim.x - 1) {
    rs_lincomb_body(lc);
    return;
  }
  __shared__ float4 part[256];      // [row group][float4 lane]
  int sg = 0;
  while (sg + 1 < t.n && (int)blockIdx.x >= t.blk0[sg + 1]) ++sg;   // uniform scan over head segments, n <= 64
  float* __restrict__ dst = t.dst[sg];
  const long len = t.len[sg];
  const bool nar = narrow >> sg & 1ull;
  const int LN = nar ? 16 : 64, NRG = 256 / LN;                       // float4 lanes per row group, row groups
  const int cx = threadIdx.x & (LN - 1), ry = threadIdx.x / LN;
  const long i = ((long)blockIdx.x - t.blk0[sg]) * (4 * LN) + 4 * cx;
  bool vec = (len & 3) == 0 && (((uintptr_t)dst) & 15) == 0;
  for (int cur = sg; cur >= 0; cur = t.next[cur])
    vec = vec && (((uintptr_t)t.src[cur]) & 15) == 0 && (t.stride[cur] & 3) == 0;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  // a parameter used several times in one backward (PoE re-runs its towers per subset) has several partial
  // regions with the SAME destination: they are chained (`next`) and summed by the same block -- two blocks
  // doing dst += concurrently would race.
  for (int cur = sg; cur >= 0; cur = t.next[cur]) {
    if (nar) rs_row_sum<16>(t.src[cur], t.rows[cur], t.stride[cur], i, len, vec, ry, a);
    else rs_row_sum<4>(t.src[cur], t.rows[cur], t.stride[cur], i, len, vec, ry, a);
  }
  part[ry * LN + cx] = a;
  __syncthreads();
  if (ry == 0) {
    float4 sm = part[cx];
    for (int k = 1; k < NRG; ++k) {
      const float4 pk = part[k * LN + cx];
      sm.x += pk.x; sm.y += pk.y; sm.z += pk.z; sm.w += pk.w;
    }
    const float sv[4] = {sm.x, sm.y, sm.z, sm.w};
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (i + c < len) dst[i + c] += sv[c];
  }
}
static int reduce_segments_impl(const mmvae_reduce_segments_t* table, const rs_lincomb_t* tail, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(table && table->n > 0 && table->n <= MMVAE_MAX_SEGMENTS);
  mmvae_reduce_segments_t t = *table;
  // chain segments that share a destination; chain heads are compacted to the front and only they get workgroups
  mmvae_reduce_segments_t o;
  int n_heads = 0;
  int head_of[MMVAE_MAX_SEGMENTS], tail_of[MMVAE_MAX_SEGMENTS];
  for (int s = 0; s < t.n; ++s) {
    if (!t.src[s] || !t.dst[s] || t.rows[s] <= 0 || t.len[s] <= 0) return MMVAE_ERR_ARG;
    int h = -1;
    for (int k = 0; k < n_heads; ++k)
      if (t.dst[head_of[k]] == t.dst[s] && t.len[head_of[k]] == t.len[s]) { h = k; break; }
    if (h < 0) { head_of[n_heads] = s; tail_of[n_heads] = s; ++n_heads; }
    else tail_of[h] = s;
  }
  // build output table: heads first (slots 0..n_heads-1), chained members after
  int pos = n_heads;
  for (int k = 0; k < n_heads; ++k) {
    const int hs = head_of[k];
    o.src[k] = t.src[hs]; o.dst[k] = t.dst[hs]; o.rows[k] = t.rows[hs]; o.len[k] = t.len[hs];
    o.stride[k] = t.stride[hs]; o.next[k] = -1;
    int last = k;
    for (int s = hs + 1; s < t.n; ++s) {
      if (t.dst[s] == t.dst[hs] && t.len[s] == t.len[hs]) {
        o.src[pos] = t.src[s]; o.dst[pos] = t.dst[s]; o.rows[pos] = t.rows[s]; o.len[pos] = t.len[s];
        o.stride[pos] = t.stride[s]; o.next[pos] = -1; o.blk0[pos] = 0x7fffffff;
        o.next[last] = pos;
        last = pos++;
      }
    }
  }
  o.n = n_heads;   // the device scan only walks head slots
  int blocks = 0;
  unsigned long long narrow = 0;
  for (int k = 0; k < n_heads; ++k) {
    int max_rows = 0;
    for (int cur = k; cur >= 0; cur = o.next[cur]) max_rows = o.rows[cur] > max_rows ? o.rows[cur] : max_rows;
    const bool nar = max_rows >= RS_NARROW_ROWS;
    if (nar) narrow |= 1ull << k;
    o.blk0[k] = blocks;
    blocks += (o.len[k] + rs_cols(nar) - 1) / rs_cols(nar);
  }
  t = o;
  if (tail) {
    hipLaunchKernelGGL(reduce_segments_kernel<true>, dim3(blocks + 1), dim3(256), 0, (hipStream_t)stream, t, *tail, narrow);
  } else {
    rs_lincomb_t none{};
    hipLaunchKernelGGL(reduce_segments_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t, none, narrow);
  }
  return mmvae_launch_status();
}
extern "C" int mmvae_reduce_segments(const mmvae_reduce_segments_t* table, mmvae_stream_t stream) {
  return reduce_segments_impl(table, nullptr, stream);
}
extern "C" int mmvae_reduce_segments_lincomb(const mmvae_reduce_segments_t* table, const mmvae_rowptrs_t* rows,
                                             const float* W_host, float* out, int n_rows, int B, int n_out,
                                             mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(rows && W_host && out && n_rows > 0 && B > 0 && n_out > 0);
  if (n_rows > RS_LC_ROWS || n_out > RS_LC_OUT) return MMVAE_ERR_UNSUPPORTED;
  rs_lincomb_t lc{};
  for (int n = 0; n < n_rows; ++n) lc.rows[n] = rows->p[n];
  for (int k = 0; k < n_out; ++k)
    for (int n = 0; n < n_rows; ++n) lc.w[k * RS_LC_ROWS + n] = W_host[k * n_rows + n];
  lc.out = out; lc.n_rows = n_rows; lc.B = B; lc.n_out = n_out;
  return reduce_segments_impl(table, &lc, stream);
}


// ---- fold + Adam in ONE launch (the serial tail of a one-GPU step: fold 16-18 us + gap + Adam 13 us before) ----------
// The flat gradient of a step = what the backward kernels wrote straight into `g` PLUS the split partials registered as
// segments.  Workgroups [0, fold_blocks) each own 256 columns of one destination range: they sum the partials exactly as
// reduce_segments_kernel does (same order, same float4 lanes), add the direct contribution, and apply the Adam update to
// those 256 elements themselves -- the folded gradient never travels through memory.  Workgroups behind them run the
// plain Adam body over the gaps between the destination ranges (2048 elements each); the optional last workgroup is the
// ELBO assembly.  Bit-identical to mmvae_reduce_segments followed by mmvae_adam_amsgrad_flat.
#define AF_PLAIN 2048
struct af_seg_t {
  const float* src;
  int dst_off, len, stride;   // destination = g + dst_off
  short rows, next;
  int blk0;
};
struct af_table_t {
  af_seg_t seg[MMVAE_MAX_SEGMENTS];      // heads first, sorted by dst_off; chained members behind
  int pblk0[MMVAE_MAX_SEGMENTS + 2];     // first workgroup of plain range k = [end of head k-1, start of head k)
  int n_heads, fold_blocks, plain_blocks;
  unsigned long long narrow;             // bit k: head k is folded by narrow blocks (64 columns x 16 row groups)
};
struct af_adam_t {
  float *p, *g, *m, *v, *vmax;
  long lo, hi;      // this launch covers elements [lo, hi) of the flat buffers (mmvae_adam_fold_range; the whole buffer otherwise)
  int* step_dev;
  float lr, b1, b2, eps, gscale;
  int zero_grad;
  int advance;      // 1: this launch closes step (*step_dev + 1) -- takes tickets, the last workgroup advances the counter
};

template <bool TAIL>
__global__ __launch_bounds__(256) void adam_fold_kernel(af_adam_t A, af_table_t t, rs_lincomb_t lc) {
  MMVAE_TRACE_STAMP(20);
  const int bid = blockIdx.x;
  const int work_blocks = t.fold_blocks + t.plain_blocks;
  if (TAIL && bid == work_blocks) {       // does not take a ticket: the step counter waits for work_blocks tickets
    rs_lincomb_body(lc);
    return;
  }
  __shared__ float bc[2];
  __shared__ double pw[2];
  __shared__ float4 part[256];      // [row group][float4 lane]
  if (threadIdx.x == 0) {      // this launch IS step (*step_dev + 1), see adam_amsgrad_kernel
    const int st = A.step_dev[0] + 1;
    const double* run = reinterpret_cast<const double*>(A.step_dev + 2);
    const double r1 = run[0], r2 = run[1];
    const double p1 = st == 1 ? (double)A.b1 : (r1 > 0.0 ? r1 * (double)A.b1 : pow((double)A.b1, (double)st));
    const double p2 = st == 1 ? (double)A.b2 : (r2 > 0.0 ? r2 * (double)A.b2 : pow((double)A.b2, (double)st));
    pw[0] = p1;
    pw[1] = p2;
    bc[0] = (float)((double)A.lr / (1.0 - p1));
    bc[1] = (float)(1.0 / sqrt(1.0 - p2));
  }
  const float b1 = A.b1, b2 = A.b2, eps = A.eps, gscale = A.gscale;
  auto update1 = [&](float& P, float G, float& M, float& V, float& X, float lr_bc1, float inv_sqrt_bc2) {
    adam_update1(P, G, M, V, X, b1, b2, eps, gscale, lr_bc1, inv_sqrt_bc2);
  };
  if (bid < t.fold_blocks) {
    int sg = 0;
    while (sg + 1 < t.n_heads && bid >= t.seg[sg + 1].blk0) ++sg;      // uniform scan over the heads
    const long len = t.seg[sg].len;
    const long off = t.seg[sg].dst_off;
    const bool nar = t.narrow >> sg & 1ull;                            // block geometry: see rs_row_sum
    const int LN = nar ? 16 : 64, NRG = 256 / LN, cols = 4 * LN;
    const int cx = threadIdx.x & (LN - 1), ry = threadIdx.x / LN;
    const long i = ((long)bid - t.seg[sg].blk0) * cols + 4 * cx;
    bool vec = (len & 3) == 0;
    for (int cur = sg; cur >= 0; cur = t.seg[cur].next)
      vec = vec && (((uintptr_t)t.seg[cur].src) & 15) == 0 && (t.seg[cur].stride & 3) == 0;
    // this thread's own column (tid) of the block's `cols`: issued before the partial sums, consumed after them
    const long e = ((long)bid - t.seg[sg].blk0) * cols + threadIdx.x;
    const bool mine = (int)threadIdx.x < cols && e < len;
    const long fe = off + e;
    float P = 0.f, G = 0.f, M = 0.f, V = 0.f, X = 0.f;
    if (mine) { P = A.p[fe]; G = A.g[fe]; M = A.m[fe]; V = A.v[fe]; X = A.vmax[fe]; }
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int cur = sg; cur >= 0; cur = t.seg[cur].next) {
      if (nar) rs_row_sum<16>(t.seg[cur].src, t.seg[cur].rows, t.seg[cur].stride, i, len, vec, ry, a);
      else rs_row_sum<4>(t.seg[cur].src, t.seg[cur].rows, t.seg[cur].stride, i, len, vec, ry, a);
    }
    part[ry * LN + cx] = a;
    __syncthreads();
    if (mine) {
      const float* pf = reinterpret_cast<const float*>(&part[0]);
      const int c = threadIdx.x;                 // column c = lane (c >> 2), component (c & 3)
      float sm = pf[c];
      for (int k = 1; k < NRG; ++k) sm += pf[k * cols + c];
      G += sm;
      update1(P, G, M, V, X, bc[0], bc[1]);
      A.p[fe] = P; A.m[fe] = M; A.v[fe] = V; A.vmax[fe] = X;
      A.g[fe] = A.zero_grad ? 0.f : G;
    }
  } else {
    const int pb = bid - t.fold_blocks;
    int k = 0;
    while (k < t.n_heads && pb >= t.pblk0[k + 1]) ++k;                  // uniform scan over the plain ranges
    const long rs = k == 0 ? A.lo : (long)t.seg[k - 1].dst_off + t.seg[k - 1].len;
    const long re = k == t.n_heads ? A.hi : (long)t.seg[k].dst_off;
    const long cs = rs + (long)(pb - t.pblk0[k]) * AF_PLAIN;
    const long ce = cs + AF_PLAIN < re ? cs + AF_PLAIN : re;
    const long a0 = (cs + 3) & ~3L, a1 = ce & ~3L;                      // float4 body [a0, a1), dword edges
    float4* p4 = reinterpret_cast<float4*>(A.p);
    float4* g4 = reinterpret_cast<float4*>(A.g);
    float4* m4 = reinterpret_cast<float4*>(A.m);
    float4* v4 = reinterpret_cast<float4*>(A.v);
    float4* x4 = reinterpret_cast<float4*>(A.vmax);
    const long q0 = (a0 >> 2) + threadIdx.x, q1 = q0 + 256, qe = a1 >> 2;
    const bool h0 = a0 < a1 && q0 < qe, h1 = a0 < a1 && q1 < qe;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 P0 = z, G0 = z, M0 = z, V0 = z, X0 = z, P1 = z, G1 = z, M1 = z, V1 = z, X1 = z;
    // the ten state loads go out BEFORE the barrier behind which thread 0's bias corrections (a dependent global load of
    // the step counter + double-precision divisions) become visible: round 6, the launch is the serial tail of the step
    if (h0) { P0 = p4[q0]; G0 = g4[q0]; M0 = m4[q0]; V0 = v4[q0]; X0 = x4[q0]; }
    if (h1) { P1 = p4[q1]; G1 = g4[q1]; M1 = m4[q1]; V1 = v4[q1]; X1 = x4[q1]; }
    __syncthreads();
    const float lr_bc1 = bc[0], inv_sqrt_bc2 = bc[1];
    if (a0 < a1) {
      if (h0) {
        update1(P0.x, G0.x, M0.x, V0.x, X0.x, lr_bc1, inv_sqrt_bc2);
        update1(P0.y, G0.y, M0.y, V0.y, X0.y, lr_bc1, inv_sqrt_bc2);
        update1(P0.z, G0.z, M0.z, V0.z, X0.z, lr_bc1, inv_sqrt_bc2);
        update1(P0.w, G0.w, M0.w, V0.w, X0.w, lr_bc1, inv_sqrt_bc2);
        p4[q0] = P0; m4[q0] = M0; v4[q0] = V0; x4[q0] = X0;
        if (A.zero_grad) g4[q0] = z;
      }
      if (h1) {
        update1(P1.x, G1.x, M1.x, V1.x, X1.x, lr_bc1, inv_sqrt_bc2);
        update1(P1.y, G1.y, M1.y, V1.y, X1.y, lr_bc1, inv_sqrt_bc2);
        update1(P1.z, G1.z, M1.z, V1.z, X1.z, lr_bc1, inv_sqrt_bc2);
        update1(P1.w, G1.w, M1.w, V1.w, X1.w, lr_bc1, inv_sqrt_bc2);
        p4[q1] = P1; m4[q1] = M1; v4[q1] = V1; x4[q1] = X1;
        if (A.zero_grad) g4[q1] = z;
      }
    }
    // dword edges: [cs, min(a0, ce)) and [max(a1, a0), ce): at most 3 + 3 elements
    {
      const long lo_end = a0 < ce ? a0 : ce;
      const long hi_beg = a1 > lo_end ? a1 : lo_end;
      long e = -1;
      if ((long)threadIdx.x < lo_end - cs) e = cs + threadIdx.x;
      else if (threadIdx.x >= 64 && (long)(threadIdx.x - 64) < ce - hi_beg) e = hi_beg + (threadIdx.x - 64);
      if (e >= 0) {
        float P = A.p[e], G = A.g[e], M = A.m[e], V = A.v[e], X = A.vmax[e];
        update1(P, G, M, V, X, lr_bc1, inv_sqrt_bc2);
        A.p[e] = P; A.m[e] = M; A.v[e] = V; A.vmax[e] = X;
        if (A.zero_grad) A.g[e] = 0.f;
      }
    }
  }
  __syncthreads();
  if (A.advance && threadIdx.x == 0) {
    const int ticket = atomicAdd(A.step_dev + 1, 1);
    if (ticket == work_blocks - 1) {
      A.step_dev[1] = 0;
      A.step_dev[0] += 1;
      double* run = reinterpret_cast<double*>(A.step_dev + 2);
      run[0] = pw[0];
      run[1] = pw[1];
    }
  }
}

static int adam_fold_impl(float* p, float* g, float* m, float* v, float* vmax, long n, long lo, long hi, int advance,
                          float lr, float beta1, float beta2, float eps, int* step_dev, float grad_scale, int zero_grad,
                          const mmvae_reduce_segments_t* table, const mmvae_rowptrs_t* rows, const float* W_host, float* out,
                          int n_rows, int B, int n_out, mmvae_stream_t stream);

extern "C" int mmvae_adam_fold_flat(float* p, float* g, float* m, float* v, float* vmax, long n, float lr, float beta1,
                                    float beta2, float eps, int* step_dev, float grad_scale, int zero_grad,
                                    const mmvae_reduce_segments_t* table, const mmvae_rowptrs_t* rows,
                                    const float* W_host, float* out, int n_rows, int B, int n_out,
                                    mmvae_stream_t stream) {
  return adam_fold_impl(p, g, m, v, vmax, n, 0, n, 1, lr, beta1, beta2, eps, step_dev, grad_scale, zero_grad, table, rows,
                        W_host, out, n_rows, B, n_out, stream);
}

// One step's update in SEVERAL launches (round 6): elements [lo, hi) of the flat buffers only, with the segments of `table`
// whose destination lies inside that range (the others are skipped: they belong to another launch of the same step; a
// segment that straddles lo or hi is an error).  Every launch of a step reads the same step number; exactly ONE of them --
// the last one in stream order -- passes advance_step = 1 and closes the step.  The captured training step runs the decoders'
// + prior's range (the second range of the flat buffer, FlatParams.split) behind the text encoder's backward, beside the image
// encoder's, so that only the encoders' parameters are left for the serial launch at the end of the step.  Per element the
// same arithmetic in the same order as mmvae_adam_fold_flat over the whole buffer: bit-identical.  table may be NULL or have
// no segment inside the range (plain Adam over [lo, hi)).
extern "C" int mmvae_adam_fold_range(float* p, float* g, float* m, float* v, float* vmax, long n, long lo, long hi,
                                     int advance_step, float lr, float beta1, float beta2, float eps, int* step_dev,
                                     float grad_scale, int zero_grad, const mmvae_reduce_segments_t* table,
                                     const mmvae_rowptrs_t* rows, const float* W_host, float* out, int n_rows, int B,
                                     int n_out, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(lo >= 0 && lo < hi && hi <= n && (lo & 3) == 0);
  return adam_fold_impl(p, g, m, v, vmax, n, lo, hi, advance_step ? 1 : 0, lr, beta1, beta2, eps, step_dev, grad_scale,
                        zero_grad, table, rows, W_host, out, n_rows, B, n_out, stream);
}

static int adam_fold_impl(float* p, float* g, float* m, float* v, float* vmax, long n, long lo, long hi, int advance,
                          float lr, float beta1, float beta2, float eps, int* step_dev, float grad_scale, int zero_grad,
                          const mmvae_reduce_segments_t* table, const mmvae_rowptrs_t* rows, const float* W_host, float* out,
                          int n_rows, int B, int n_out, mmvae_stream_t stream) {
  const bool whole = lo == 0 && hi == n;
  MMVAE_CHECK_ARG(p && g && m && v && vmax && n > 0 && n < (1L << 31) && step_dev && (table || !whole));
  MMVAE_CHECK_ARG(!table || (table->n >= (whole ? 1 : 0) && table->n <= MMVAE_MAX_SEGMENTS));
  if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)vmax) & 15) != 0) return MMVAE_ERR_ARG;
  if ((((uintptr_t)step_dev) & 7) != 0) return MMVAE_ERR_ARG;
  mmvae_reduce_segments_t inside;      // the segments of this launch's range
  inside.n = 0;
  for (int s = 0; table && s < table->n; ++s) {
    if (!table->dst[s] || table->len[s] <= 0) return MMVAE_ERR_ARG;
    const float *d0 = table->dst[s], *d1 = table->dst[s] + table->len[s];
    if (d1 <= g + lo || d0 >= g + hi) {
      if (whole) return MMVAE_ERR_ARG;       // (outside the flat buffer)
      continue;
    }
    if (d0 < g + lo || d1 > g + hi) return MMVAE_ERR_ARG;      // straddles a range boundary
    const int k = inside.n++;
    inside.src[k] = table->src[s]; inside.dst[k] = table->dst[s]; inside.rows[k] = table->rows[s];
    inside.len[k] = table->len[s]; inside.stride[k] = table->stride[s];
  }
  const mmvae_reduce_segments_t& t = inside;
  // heads = distinct destination ranges, sorted by offset; members with the same (dst, len) chained behind their head
  int head[MMVAE_MAX_SEGMENTS], n_heads = 0;
  for (int s = 0; s < t.n; ++s) {
    if (!t.src[s] || !t.dst[s] || t.rows[s] <= 0 || t.rows[s] > 32767 || t.len[s] <= 0) return MMVAE_ERR_ARG;
    if (t.dst[s] < g || t.dst[s] + t.len[s] > g + n || t.stride[s] < 0 || t.stride[s] >= (1L << 31)) return MMVAE_ERR_ARG;
    bool seen = false;
    for (int k = 0; k < n_heads; ++k)
      if (t.dst[head[k]] == t.dst[s] && t.len[head[k]] == t.len[s]) { seen = true; break; }
    if (!seen) head[n_heads++] = s;
  }
  for (int a = 1; a < n_heads; ++a)         // insertion sort by destination
    for (int b = a; b > 0 && t.dst[head[b]] < t.dst[head[b - 1]]; --b) { const int x = head[b]; head[b] = head[b - 1]; head[b - 1] = x; }
  for (int k = 1; k < n_heads; ++k)         // overlapping destination ranges would race
    if (t.dst[head[k - 1]] + t.len[head[k - 1]] > t.dst[head[k]]) return MMVAE_ERR_ARG;
  af_table_t o{};
  int pos = n_heads, blocks = 0;
  for (int k = 0; k < n_heads; ++k) {
    const int hs = head[k];
    o.seg[k] = af_seg_t{t.src[hs], (int)(t.dst[hs] - g), (int)t.len[hs], (int)t.stride[hs], (short)t.rows[hs], (short)-1, blocks};
    int max_rows = t.rows[hs];
    for (int s = 0; s < t.n; ++s)
      if (s != hs && t.dst[s] == t.dst[hs] && t.len[s] == t.len[hs] && t.rows[s] > max_rows) max_rows = t.rows[s];
    const bool nar = max_rows >= RS_NARROW_ROWS;       // the same choice as reduce_segments_impl: same summation order
    if (nar) o.narrow |= 1ull << k;
    blocks += (int)((t.len[hs] + rs_cols(nar) - 1) / rs_cols(nar));
    int last = k;
    for (int s = 0; s < t.n; ++s) {
      if (s == hs || t.dst[s] != t.dst[hs] || t.len[s] != t.len[hs]) continue;
      o.seg[pos] = af_seg_t{t.src[s], (int)(t.dst[s] - g), (int)t.len[s], (int)t.stride[s], (short)t.rows[s], (short)-1, 0x7fffffff};
      o.seg[last].next = (short)pos;
      last = pos++;
    }
  }
  o.n_heads = n_heads;
  o.fold_blocks = blocks;
  int pblocks = 0;
  for (int k = 0; k <= n_heads; ++k) {
    const long rs = k == 0 ? lo : (long)o.seg[k - 1].dst_off + o.seg[k - 1].len;
    const long re = k == n_heads ? hi : (long)o.seg[k].dst_off;
    o.pblk0[k] = pblocks;
    pblocks += (int)((re - rs + AF_PLAIN - 1) / AF_PLAIN);
  }
  o.pblk0[n_heads + 1] = pblocks;
  o.plain_blocks = pblocks;
  af_adam_t A{p, g, m, v, vmax, lo, hi, step_dev, lr, beta1, beta2, eps, grad_scale, zero_grad, advance};
  if (blocks + pblocks == 0) return MMVAE_ERR_ARG;
  if (rows) {
    MMVAE_CHECK_ARG(W_host && out && n_rows > 0 && B > 0 && n_out > 0);
    if (n_rows > RS_LC_ROWS || n_out > RS_LC_OUT) return MMVAE_ERR_UNSUPPORTED;
    rs_lincomb_t lc{};
    for (int i = 0; i < n_rows; ++i) lc.rows[i] = rows->p[i];
    for (int k = 0; k < n_out; ++k)
      for (int i = 0; i < n_rows; ++i) lc.w[k * RS_LC_ROWS + i] = W_host[k * n_rows + i];
    lc.out = out; lc.n_rows = n_rows; lc.B = B; lc.n_out = n_out;
    hipLaunchKernelGGL(adam_fold_kernel<true>, dim3(blocks + pblocks + 1), dim3(256), 0, (hipStream_t)stream, A, o, lc);
  } else {
    rs_lincomb_t none{};
    hipLaunchKernelGGL(adam_fold_kernel<false>, dim3(blocks + pblocks), dim3(256), 0, (hipStream_t)stream, A, o, none);
  }
  return mmvae_launch_status();
}

__global__ __launch_bounds__(256) void fill_kernel(float* __restrict__ p, long n, float value) {
  const long gs = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += gs) p[i] = value;
}
extern "C" int mmvae_fill(float* p, long n, float value, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(p && n > 0);
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, n, value);
  return mmvae_launch_status();
}

__global__ void step_inc_kernel(int* s) { *s += 1; }
extern "C" int mmvae_step_inc(int* step_dev, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(step_dev);
  hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev);
  return mmvae_launch_status();
}

extern "C" int mmvae_version(void) { return 1; }
extern "C" const char* mmvae_arch(void) { return "gfx950"; }

MMVAE_TRACE_SETTER(optim)
