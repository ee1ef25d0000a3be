// Fused convolution + BatchNorm engine of the ResNet-50 image tower (`encoder: CNN`, reference models/encoders.py:86-127 =
// torchvision resnet50 -> SiLU -> heads).  gfx950 only.
//
// Activations are NHWC (rows = B*H*W, C) fp32 matrices; convolution weights are stored channels-last, (Cout, kh, kw, Cin)
// in memory behind the (Cout, Cin, kh, kw) parameter view, so that for every filter tap the reduction runs over
// contiguous input channels.  A k x k convolution is then a sum over its taps of 1x1 GEMMs whose A rows are shifted
// pixels: the im2col matrix only ever exists as 32/64-row tiles in LDS (a per-geometry table gives, for a row and a tap,
// the source pixel or -1 for the zero padding).
//
// What is fused around the fp32 MFMA (v_mfma_f32_32x32x2_f32) tiles -- no BatchNorm or elementwise kernel is left
// between two convolutions:
//   forward   A prologue: relu(bn(Y_prev)) = max(fma(y - mean, gamma rstd, beta), 0) of the producer's RAW output;
//             epilogue: raw output + per-column (mean, M2) of the tile; the last workgroup of a column tile merges the
//             row tiles' partials (Chan, double), emits mean / rstd / gamma rstd and moves the running statistics.
//   dgrad     A prologue: the BatchNorm backward of the consumer side, dY = G p + Y q + r per channel (p, q, r from the
//             statistics sum G, sum G xhat); epilogue: (+ shortcut gradient), ReLU mask recomputed from the producer's
//             raw output, the statistics of THAT BatchNorm's backward, last workgroup: dgamma, dbeta, p, q, r.
//   wgrad     A prologue as dgrad (transposed), B prologue as forward; split over the pixel rows with the partial tiles
//             summed in a fixed order by the last workgroup of an output tile (deterministic, no atomics on data).
// Cross-workgroup hand-over inside a launch: agent-scope write-through stores, a relaxed agent-scope ticket,
// agent-scope loads in the elected workgroup (latent.hip: poe_last_workgroup explains why not __threadfence()).
#include "common.hpp"

#define RC_PRE_NONE 0
#define RC_PRE_RELU 1
#define RC_PRE_BN_RELU 2
#define RC_MASK_NONE 0
#define RC_MASK_RAW 1
#define RC_MASK_BN 2

__device__ __forceinline__ float rc_bn(float y, float mean, float sc, float beta) { return fmaf(y - mean, sc, beta); }
__device__ __forceinline__ void rc_st(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float rc_ld(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// sum of n values `stride` floats apart written by other workgroups of this launch, in index order, with the loads of
// 8 values in flight together (a loop of dependent agent-scope loads costs a memory round trip per value)
__device__ __forceinline__ float rc_sum_strided(const float* __restrict__ p, size_t stride, int n, float t) {
  int z = 0;
  for (; z + 8 <= n; z += 8) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = rc_ld(p + (size_t)(z + i) * stride);
#pragma unroll
    for (int i = 0; i < 8; ++i) t += v[i];
  }
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = rc_ld(p + (size_t)min(z + i, n - 1) * stride);
#pragma unroll
  for (int i = 0; i < 8; ++i) t += z + i < n ? v[i] : 0.f;
  return t;
}
// `expected` workgroups take a ticket; the last one gets true (and re-arms the ticket for the next launch)
__device__ __forceinline__ bool rc_last_workgroup(unsigned* __restrict__ ticket, unsigned expected, int* last_lds) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *last_lds = t == expected - 1;
    if (*last_lds) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  return *last_lds != 0;
}

// Staging of an R x BK operand tile into LDS as [k][r] (pitch R + 1, odd: conflict-free for both the transposing store
// and the per-lane MFMA fragment reads).  KMAJOR: consecutive threads walk k (the source is k-contiguous), a thread's
// slots are rows rl + i RSTEP; otherwise consecutive threads walk the rows and the slots are k = kl + i KSTEP.
template <int R, int BK, bool KMAJOR>
struct RcStg {
  static constexpr int PER = R * BK / 256;
  static constexpr int RSTEP = 256 / BK, KSTEP = 256 / R, RP = R + 1;
  int rl, kl;
  __device__ __forceinline__ void init(int tid) {
    if (KMAJOR) { kl = tid % BK; rl = tid / BK; } else { rl = tid % R; kl = tid / R; }
  }
  __device__ __forceinline__ int row(int i) const { return KMAJOR ? rl + i * RSTEP : rl; }
  __device__ __forceinline__ int kk(int i) const { return KMAJOR ? kl : kl + i * KSTEP; }
  __device__ __forceinline__ void store(float* __restrict__ S, const float (&v)[PER]) const {
    float* d = S + kl * RP + rl;
#pragma unroll
    for (int i = 0; i < PER; ++i) d[KMAJOR ? i * RSTEP : i * KSTEP * RP] = v[i];
  }
};

// 4 wavefronts over a BM x BN tile: WM x WN waves own 32 x 32 sub-tiles, the remaining factor WK splits every BK-deep
// stage (64 x 64: 2 x 2 x 1; 32 x 32: 1 x 1 x 4 -- the layers with few output tiles and a deep reduction).
template <int BM, int BN, int BK>
struct RcTile {
  static constexpr int WM = BM / 32, WN = BN / 32, WK = 4 / (WM * WN), KW = BK / WK;
  static constexpr int AP = BM + 1, BP = BN + 1;
  static constexpr int STAGE = BK * AP + BK * BP, OUT = WK * BM * BP;
  static constexpr int SMEM = STAGE > OUT ? STAGE : OUT;
  static constexpr int RG = 256 / BN, NR = BM / RG;   // epilogue: thread = (column, row group), NR rows each
  int wm, wn, wk, li, lh;
  __device__ __forceinline__ void init(int tid) {
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    wk = wave / (WM * WN);
    wm = (wave % (WM * WN)) / WN;
    wn = wave % WN;
    li = lane & 31;
    lh = lane >> 5;
  }
  __device__ __forceinline__ void mma(const float* __restrict__ As, const float* __restrict__ Bs, f32x16& acc) const {
    const float* a = As + (wk * KW + lh) * AP + wm * 32 + li;
    const float* b = Bs + (wk * KW + lh) * BP + wn * 32 + li;
#pragma unroll
    for (int kk = 0; kk < KW; kk += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk * AP], b[kk * BP], acc, 0, 0, 0);
  }
  // accumulators -> LDS tile(s) [wk][BM][BP]; the caller syncs before and after
  __device__ __forceinline__ void spill(float* __restrict__ T, const f32x16& acc) const {
#pragma unroll
    for (int r = 0; r < 16; ++r)
      T[(wk * BM + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * BP + wn * 32 + li] = acc[r];
  }
  __device__ __forceinline__ static float tile_at(const float* __restrict__ T, int row, int col) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < WK; ++w) t += T[(w * BM + row) * BP + col];
    return t;
  }
};

// sum over the RG row groups of a column (cs: 256 floats), every thread gets the total; fixed order
template <int BN>
__device__ __forceinline__ float rc_colsum(float* __restrict__ cs, float v, int col, int rg) {
  constexpr int RG = 256 / BN;
  __syncthreads();
  cs[rg * BN + col] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int g = 0; g < RG; ++g) t += cs[g * BN + col];
  return t;
}

// ---------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------
struct RcBnFwd {          // the BatchNorm that follows the convolution (statistics of the raw output)
  const float* gamma;
  const float* beta;
  float* run_mean;
  float* run_var;
  float* mean;            // out (C): batch mean (eval: running mean)
  float* rstd;            // out (C)
  float* sc;              // out (C): gamma * rstd
  float* part;            // (row tiles, C, 2)
  unsigned* counter;      // (column tiles)
  float eps, momentum;
  int eval;
};
struct RcFwdArgs {
  const float* x;         // (Min, Cin)
  const float* w;         // (Cout, T, Cin)
  const float* xmean;     // prologue of RC_PRE_BN_RELU: relu(fma(x - xmean, xsc, xbeta))
  const float* xsc;
  const float* xbeta;
  const int* tbl;         // (T, M) source row of (tap, output row), -1 = padding; NULL: identity (T = 1, Min = M)
  float* y;               // (M, Cout)
  int M, Cin, Cout, T, pre;
  RcBnFwd bn;             // bn.part == NULL: no statistics
};

template <int BN>
__device__ __forceinline__ void rc_bn_fwd_finalize(const RcBnFwd& bn, double* __restrict__ dl, int nparts, int BMrows,
                                                   int M, int C, int n, int col, int rg) {
  constexpr int RG = 256 / BN;
  // pass 1: mean = sum cnt_p mean_p / M; pass 2: M2 = sum [M2_p + cnt_p (mean_p - mean)^2]  (exact two-pass merge of
  // the row tiles' (mean, M2); this thread's partials stay in registers between the passes when they fit)
  constexpr int KEEP = 16;
  float pm[KEEP], pq[KEEP];
  double s = 0.0;
  int cnt_i = 0;
  for (int p0 = rg; p0 < nparts; p0 += RG * KEEP) {
#pragma unroll
    for (int i = 0; i < KEEP; ++i) {
      const int p = p0 + i * RG;
      pm[i] = rc_ld(bn.part + ((size_t)min(p, nparts - 1) * C + n) * 2);
    }
#pragma unroll
    for (int i = 0; i < KEEP; ++i) {
      const int p = p0 + i * RG;
      if (p < nparts) s += (double)min(BMrows, M - p * BMrows) * (double)pm[i];
    }
    ++cnt_i;
  }
  __syncthreads();
  dl[rg * BN + col] = s;
  __syncthreads();
  double mean = 0.0;
  for (int g = 0; g < RG; ++g) mean += dl[g * BN + col];
  mean /= (double)M;
  double q = 0.0;
  for (int p0 = rg; p0 < nparts; p0 += RG * KEEP) {
    if (cnt_i > 1) {
#pragma unroll
      for (int i = 0; i < KEEP; ++i) {
        const int p = p0 + i * RG;
        pm[i] = rc_ld(bn.part + ((size_t)min(p, nparts - 1) * C + n) * 2);
      }
    }
#pragma unroll
    for (int i = 0; i < KEEP; ++i) {
      const int p = p0 + i * RG;
      pq[i] = rc_ld(bn.part + ((size_t)min(p, nparts - 1) * C + n) * 2 + 1);
    }
#pragma unroll
    for (int i = 0; i < KEEP; ++i) {
      const int p = p0 + i * RG;
      if (p < nparts) {
        const double d = (double)pm[i] - mean;
        q += (double)pq[i] + (double)min(BMrows, M - p * BMrows) * d * d;
      }
    }
  }
  __syncthreads();
  dl[rg * BN + col] = q;
  __syncthreads();
  if (rg == 0) {
    double c2 = 0.0;
    for (int g = 0; g < RG; ++g) c2 += dl[g * BN + col];
    const double cn = (double)M, var = c2 / cn;
    double rs = 1.0 / sqrt(var + (double)bn.eps);
    if (bn.eval) {
      mean = (double)bn.run_mean[n];
      rs = 1.0 / sqrt((double)bn.run_var[n] + (double)bn.eps);
    } else if (bn.run_mean) {
      const double mo = (double)bn.momentum, unb = cn > 1.0 ? c2 / (cn - 1.0) : var;
      bn.run_mean[n] = (float)((1.0 - mo) * (double)bn.run_mean[n] + mo * mean);
      bn.run_var[n] = (float)((1.0 - mo) * (double)bn.run_var[n] + mo * unb);
    }
    bn.mean[n] = (float)mean;
    bn.rstd[n] = (float)rs;
    bn.sc[n] = (float)((double)bn.gamma[n] * rs);
  }
}

template <int BM, int BN, int BK>
__global__ __launch_bounds__(256) void rc_fwd_kernel(RcFwdArgs a) {
  using TL = RcTile<BM, BN, BK>;
  using SA = RcStg<BM, BK, true>;
  using SB = RcStg<BN, BK, true>;
  __shared__ __attribute__((aligned(16))) float smem[TL::SMEM];
  __shared__ float cs[256];
  __shared__ int last;
  float* As = smem;
  float* Bs = smem + BK * TL::AP;
  const int tid = threadIdx.x;
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
  TL tl;
  tl.init(tid);
  SA sa;
  SB sb;
  sa.init(tid);
  sb.init(tid);
  int src[SA::PER], wrow[SB::PER];
  float ra[SA::PER], rb[SB::PER];
  float pm = 0.f, ps = 1.f, pb = 0.f;
  unsigned oka = 0;
#pragma unroll
  for (int i = 0; i < SB::PER; ++i) wrow[i] = (n0 + sb.row(i)) * a.T;
  int ltap = 0, lc0 = 0;
  bool newtap = true;
  auto load = [&]() {
    if (newtap) {
#pragma unroll
      for (int i = 0; i < SA::PER; ++i) {
        const int r = m0 + sa.row(i);
        src[i] = r < a.M ? (a.tbl ? a.tbl[(size_t)ltap * a.M + r] : r) : -1;
      }
      newtap = false;
    }
    const int c = lc0 + sa.kl;
    if (a.pre == RC_PRE_BN_RELU) { pm = a.xmean[c]; ps = a.xsc[c]; pb = a.xbeta[c]; }
    oka = 0;
#pragma unroll
    for (int i = 0; i < SA::PER; ++i) {
      const bool ok = src[i] >= 0;
      oka |= (ok ? 1u : 0u) << i;
      ra[i] = a.x[ok ? (size_t)src[i] * a.Cin + c : 0];
    }
#pragma unroll
    for (int i = 0; i < SB::PER; ++i) rb[i] = a.w[(size_t)(wrow[i] + ltap) * a.Cin + lc0 + sb.kl];
    lc0 += BK;
    if (lc0 >= a.Cin) { lc0 = 0; ++ltap; newtap = true; }
  };
  auto store = [&]() {
    float va[SA::PER];
#pragma unroll
    for (int i = 0; i < SA::PER; ++i) {
      float v = ra[i];
      if (a.pre == RC_PRE_BN_RELU) v = fmaxf(rc_bn(v, pm, ps, pb), 0.f);
      else if (a.pre == RC_PRE_RELU) v = fmaxf(v, 0.f);
      va[i] = (oka >> i & 1u) ? v : 0.f;
    }
    sa.store(As, va);
    sb.store(Bs, rb);
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int nstage = a.T * (a.Cin / BK);
  load();
#pragma unroll 1
  for (int s = 0; s < nstage; ++s) {
    store();
    __syncthreads();
    if (s + 1 < nstage) load();
    tl.mma(As, Bs, acc);
    __syncthreads();
  }
  tl.spill(smem, acc);
  __syncthreads();
  const int col = tid % BN, rg = tid / BN, n = n0 + col;
  float v[TL::NR], s = 0.f;
  const int cnt = min(BM, a.M - m0);
#pragma unroll
  for (int j = 0; j < TL::NR; ++j) {
    const int row = rg + j * TL::RG;
    v[j] = TL::tile_at(smem, row, col);
    if (row < cnt) {
      a.y[(size_t)(m0 + row) * a.Cout + n] = v[j];
      s += v[j];
    }
  }
  if (!a.bn.part) return;
  const float mean_t = rc_colsum<BN>(cs, s, col, rg) / (float)cnt;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < TL::NR; ++j)
    if (rg + j * TL::RG < cnt) q = fmaf(v[j] - mean_t, v[j] - mean_t, q);
  const float m2_t = rc_colsum<BN>(cs, q, col, rg);
  if (rg == 0) {
    rc_st(a.bn.part + ((size_t)blockIdx.y * a.Cout + n) * 2, mean_t);
    rc_st(a.bn.part + ((size_t)blockIdx.y * a.Cout + n) * 2 + 1, m2_t);
  }
  if (!rc_last_workgroup(a.bn.counter + blockIdx.x, gridDim.y, &last)) return;
  rc_bn_fwd_finalize<BN>(a.bn, reinterpret_cast<double*>(smem), gridDim.y, BM, a.M, a.Cout, n, col, rg);
}

// ---------------------------------------------------------------------------------------------------------------------
// BatchNorm backward statistics (shared by the dgrad epilogue and the stand-alone kernel)
// ---------------------------------------------------------------------------------------------------------------------
struct RcStat {           // a BatchNorm whose output gradient G the kernel holds: sum G, sum G xhat over the rows
  const float* Y;         // (rows, C) raw input of that BatchNorm
  const float* mean;
  const float* rstd;
  const float* gamma;
  float* pqr;             // out (3, C): the input gradient is G p + Y q + r
  float* dgamma;
  float* dbeta;
  float* part;            // (row tiles, C, 2)
  unsigned* counter;      // (column tiles)
  int acc;                // add to dgamma / dbeta instead of overwriting
  int eval;
};

template <int BN>
__device__ __forceinline__ void rc_stat_finalize(const RcStat& st, double* __restrict__ dl, int nparts, int M, int C, int n,
                                                 int col, int rg) {
  constexpr int RG = 256 / BN;
  double s1 = 0.0, s2 = 0.0;
  for (int p0 = rg; p0 < nparts; p0 += RG * 8) {
    float v1[8], v2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int p = p0 + i * RG;
      v1[i] = rc_ld(st.part + ((size_t)min(p, nparts - 1) * C + n) * 2);
      v2[i] = rc_ld(st.part + ((size_t)min(p, nparts - 1) * C + n) * 2 + 1);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (p0 + i * RG < nparts) { s1 += (double)v1[i]; s2 += (double)v2[i]; }
  }
  __syncthreads();
  dl[(rg * BN + col) * 2] = s1;
  dl[(rg * BN + col) * 2 + 1] = s2;
  __syncthreads();
  if (rg == 0) {
    s1 = 0.0; s2 = 0.0;
    for (int g = 0; g < RG; ++g) { s1 += dl[(g * BN + col) * 2]; s2 += dl[(g * BN + col) * 2 + 1]; }
    if (st.dbeta) st.dbeta[n] = (float)((st.acc ? (double)st.dbeta[n] : 0.0) + s1);
    if (st.dgamma) st.dgamma[n] = (float)((st.acc ? (double)st.dgamma[n] : 0.0) + s2);
    const double rs = (double)st.rstd[n], p = (double)st.gamma[n] * rs;
    double q = 0.0, r = 0.0;
    if (!st.eval) {
      q = -p * rs * (s2 / (double)M);
      r = -p * (s1 / (double)M) - q * (double)st.mean[n];
    }
    st.pqr[n] = (float)p;
    st.pqr[C + n] = (float)q;
    st.pqr[2 * C + n] = (float)r;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// data gradient
// ---------------------------------------------------------------------------------------------------------------------
struct RcDgradArgs {
  const float* G;         // (M, Cout) gradient of the BatchNorm output behind this convolution ...
  const float* Y;         // ... its raw input (this convolution's output) ...
  const float* pqr;       // ... and (3, Cout): dY = G p + Y q + r.  NULL: dY = G
  const float* w;         // (Cout, T, Cin)
  const int* tbl;         // (T, Min): output row feeding (tap, input row), -1 = none; NULL: identity
  const float* add;       // (Min, Cin) added before the mask (the shortcut's gradient) or NULL
  const float* mY;        // mask source (Min, Cin): RC_MASK_RAW mY > 0, RC_MASK_BN bn(mY) > 0
  const float* mmean;
  const float* msc;
  const float* mbeta;
  float* out;             // (Min, Cin)
  int M, Min, Cin, Cout, T, mask, nstat;
  RcStat st[2];
};

template <int BM, int BN, int BK>
__global__ __launch_bounds__(256) void rc_dgrad_kernel(RcDgradArgs a) {
  using TL = RcTile<BM, BN, BK>;
  using SA = RcStg<BM, BK, true>;
  using SB = RcStg<BN, BK, false>;
  __shared__ __attribute__((aligned(16))) float smem[TL::SMEM];
  __shared__ float cs[256];
  __shared__ int last;
  float* As = smem;
  float* Bs = smem + BK * TL::AP;
  const int tid = threadIdx.x;
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;     // n0: input-channel tile
  TL tl;
  tl.init(tid);
  SA sa;
  SB sb;
  sa.init(tid);
  sb.init(tid);
  int src[SA::PER];
  float rg_[SA::PER], ry[SA::PER], rb[SB::PER];
  float pp = 1.f, pq = 0.f, pr = 0.f;
  unsigned oka = 0;
  int ltap = 0, lk0 = 0;
  bool newtap = true;
  auto load = [&]() {
    if (newtap) {
#pragma unroll
      for (int i = 0; i < SA::PER; ++i) {
        const int r = m0 + sa.row(i);
        src[i] = r < a.Min ? (a.tbl ? a.tbl[(size_t)ltap * a.Min + r] : r) : -1;
      }
      newtap = false;
    }
    const int k = lk0 + sa.kl;
    if (a.pqr) { pp = a.pqr[k]; pq = a.pqr[a.Cout + k]; pr = a.pqr[2 * a.Cout + k]; }
    oka = 0;
#pragma unroll
    for (int i = 0; i < SA::PER; ++i) {
      const bool ok = src[i] >= 0;
      oka |= (ok ? 1u : 0u) << i;
      const size_t o = ok ? (size_t)src[i] * a.Cout + k : 0;
      rg_[i] = a.G[o];
      ry[i] = a.pqr ? a.Y[o] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < SB::PER; ++i)
      rb[i] = a.w[((size_t)(lk0 + sb.kk(i)) * a.T + ltap) * a.Cin + n0 + sb.rl];
    lk0 += BK;
    if (lk0 >= a.Cout) { lk0 = 0; ++ltap; newtap = true; }
  };
  auto store = [&]() {
    float va[SA::PER];
#pragma unroll
    for (int i = 0; i < SA::PER; ++i) {
      const float v = a.pqr ? fmaf(rg_[i], pp, fmaf(ry[i], pq, pr)) : rg_[i];
      va[i] = (oka >> i & 1u) ? v : 0.f;
    }
    sa.store(As, va);
    sb.store(Bs, rb);
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int nstage = a.T * (a.Cout / BK);
  load();
#pragma unroll 1
  for (int s = 0; s < nstage; ++s) {
    store();
    __syncthreads();
    if (s + 1 < nstage) load();
    tl.mma(As, Bs, acc);
    __syncthreads();
  }
  tl.spill(smem, acc);
  __syncthreads();
  const int col = tid % BN, rg = tid / BN, c = n0 + col;
  const int cnt = min(BM, a.Min - m0);
  float mm = 0.f, ms = 1.f, mb = 0.f;
  if (a.mask == RC_MASK_BN) { mm = a.mmean[c]; ms = a.msc[c]; mb = a.mbeta[c]; }
  float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f}, tm[2] = {0.f, 0.f}, tr[2] = {0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 2; ++t)
    if (t < a.nstat) { tm[t] = a.st[t].mean[c]; tr[t] = a.st[t].rstd[c]; }
#pragma unroll
  for (int j = 0; j < TL::NR; ++j) {
    const int row = rg + j * TL::RG;
    if (row < cnt) {
      const size_t o = (size_t)(m0 + row) * a.Cin + c;
      float g = TL::tile_at(smem, row, col);
      if (a.add) g += a.add[o];
      float my = 0.f;
      if (a.mask != RC_MASK_NONE) {
        my = a.mY[o];
        const float z = a.mask == RC_MASK_BN ? rc_bn(my, mm, ms, mb) : my;
        g = z > 0.f ? g : 0.f;
      }
      a.out[o] = g;
#pragma unroll
      for (int t = 0; t < 2; ++t)
        if (t < a.nstat) {
          const float yv = (a.st[t].Y == a.mY && a.mask != RC_MASK_NONE) ? my : a.st[t].Y[o];
          s1[t] += g;
          s2[t] = fmaf(g, (yv - tm[t]) * tr[t], s2[t]);
        }
    }
  }
  if (a.nstat == 0) return;
#pragma unroll
  for (int t = 0; t < 2; ++t)
    if (t < a.nstat) {
      const float a1 = rc_colsum<BN>(cs, s1[t], col, rg), a2 = rc_colsum<BN>(cs, s2[t], col, rg);
      if (rg == 0) {
        rc_st(a.st[t].part + ((size_t)blockIdx.y * a.Cin + c) * 2, a1);
        rc_st(a.st[t].part + ((size_t)blockIdx.y * a.Cin + c) * 2 + 1, a2);
      }
    }
  if (!rc_last_workgroup(a.st[0].counter + blockIdx.x, gridDim.y, &last)) return;
#pragma unroll
  for (int t = 0; t < 2; ++t)
    if (t < a.nstat) rc_stat_finalize<BN>(a.st[t], reinterpret_cast<double*>(smem), gridDim.y, a.Min, a.Cin, c, col, rg);
}

// stand-alone statistics of a BatchNorm backward whose G was produced elsewhere (pooling backward, tests):
// grid (C / 64, row tiles of 64)
__global__ __launch_bounds__(256) void rc_stat_kernel(const float* __restrict__ G, RcStat st, int M, int C) {
  __shared__ __attribute__((aligned(16))) float smem[64 * 4 * 2 * 2];
  __shared__ float cs[256];
  __shared__ int last;
  const int tid = threadIdx.x, col = tid % 64, rg = tid / 64, c = blockIdx.x * 64 + col;
  const int m0 = blockIdx.y * 64, cnt = min(64, M - m0);
  const float tm = st.mean[c], tr = st.rstd[c];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll 4
  for (int row = rg; row < cnt; row += 4) {
    const size_t o = (size_t)(m0 + row) * C + c;
    const float g = G[o];
    s1 += g;
    s2 = fmaf(g, (st.Y[o] - tm) * tr, s2);
  }
  const float a1 = rc_colsum<64>(cs, s1, col, rg), a2 = rc_colsum<64>(cs, s2, col, rg);
  if (rg == 0) {
    rc_st(st.part + ((size_t)blockIdx.y * C + c) * 2, a1);
    rc_st(st.part + ((size_t)blockIdx.y * C + c) * 2 + 1, a2);
  }
  if (!rc_last_workgroup(st.counter + blockIdx.x, gridDim.y, &last)) return;
  rc_stat_finalize<64>(st, reinterpret_cast<double*>(smem), gridDim.y, M, C, c, col, rg);
}

// ---------------------------------------------------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------------------------------------------------
struct RcWgradArgs {
  const float* G;         // (M, Cout), Y, pqr: dY = G p + Y q + r (pqr NULL: dY = G)
  const float* Y;
  const float* pqr;
  const float* x;         // (Min, Cin) raw input, consumed through `pre` as in the forward pass
  const float* xmean;
  const float* xsc;
  const float* xbeta;
  const int* tbl;         // (T, M) as forward
  float* dw;              // (Cout, T, Cin)
  float* ws;              // nz > 1: (nz, Cout T Cin) partial tiles
  unsigned* counter;      // nz > 1: one ticket per output tile
  int M, Cin, Cout, T, pre, acc, nz, kper;
};

template <int BM, int BN, int BK>
__global__ __launch_bounds__(256) void rc_wgrad_kernel(RcWgradArgs a) {
  using TL = RcTile<BM, BN, BK>;
  using SA = RcStg<BM, BK, false>;
  using SB = RcStg<BN, BK, false>;
  __shared__ __attribute__((aligned(16))) float smem[TL::SMEM];
  __shared__ int last;
  float* As = smem;
  float* Bs = smem + BK * TL::AP;
  const int tid = threadIdx.x;
  const int c0 = blockIdx.x * BN, n0 = blockIdx.y * BM;
  const int tap = blockIdx.z / a.nz, zi = blockIdx.z % a.nz;
  const int kbeg = zi * a.kper, kend = min(a.M, kbeg + a.kper);
  TL tl;
  tl.init(tid);
  SA sa;
  SB sb;
  sa.init(tid);
  sb.init(tid);
  float rg_[SA::PER], ry[SA::PER], rb[SB::PER];
  unsigned oka = 0, okb = 0;
  const int n = n0 + sa.rl, c = c0 + sb.rl;
  float pp = 1.f, pq = 0.f, pr = 0.f, pm = 0.f, ps = 1.f, pb = 0.f;
  if (a.pqr) { pp = a.pqr[n]; pq = a.pqr[a.Cout + n]; pr = a.pqr[2 * a.Cout + n]; }
  if (a.pre == RC_PRE_BN_RELU) { pm = a.xmean[c]; ps = a.xsc[c]; pb = a.xbeta[c]; }
  const int* tb = a.tbl ? a.tbl + (size_t)tap * a.M : nullptr;
  auto load = [&](int k0) {
    oka = 0;
    okb = 0;
#pragma unroll
    for (int i = 0; i < SA::PER; ++i) {
      const int m = k0 + sa.kk(i);
      const bool ok = m < kend;
      oka |= (ok ? 1u : 0u) << i;
      const size_t o = ok ? (size_t)m * a.Cout + n : 0;
      rg_[i] = a.G[o];
      ry[i] = a.pqr ? a.Y[o] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < SB::PER; ++i) {
      const int m = k0 + sb.kk(i);
      int sr = -1;
      if (m < kend) sr = tb ? tb[m] : m;
      const bool ok = sr >= 0;
      okb |= (ok ? 1u : 0u) << i;
      rb[i] = a.x[ok ? (size_t)sr * a.Cin + c : 0];
    }
  };
  auto store = [&]() {
    float va[SA::PER], vb[SB::PER];
#pragma unroll
    for (int i = 0; i < SA::PER; ++i) {
      const float v = a.pqr ? fmaf(rg_[i], pp, fmaf(ry[i], pq, pr)) : rg_[i];
      va[i] = (oka >> i & 1u) ? v : 0.f;
    }
#pragma unroll
    for (int i = 0; i < SB::PER; ++i) {
      float v = rb[i];
      if (a.pre == RC_PRE_BN_RELU) v = fmaxf(rc_bn(v, pm, ps, pb), 0.f);
      else if (a.pre == RC_PRE_RELU) v = fmaxf(v, 0.f);
      vb[i] = (okb >> i & 1u) ? v : 0.f;
    }
    sa.store(As, va);
    sb.store(Bs, vb);
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (kbeg < kend) load(kbeg);
#pragma unroll 1
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    store();
    __syncthreads();
    if (k0 + BK < kend) load(k0 + BK);
    tl.mma(As, Bs, acc);
    __syncthreads();
  }
  tl.spill(smem, acc);
  __syncthreads();
  const int col = tid % BN, rg = tid / BN;
  const size_t numel = (size_t)a.Cout * a.T * a.Cin;
  float v[TL::NR];
#pragma unroll
  for (int j = 0; j < TL::NR; ++j) v[j] = TL::tile_at(smem, rg + j * TL::RG, col);
  auto idx = [&](int j) { return ((size_t)(n0 + rg + j * TL::RG) * a.T + tap) * a.Cin + c0 + col; };
  if (a.nz == 1) {
#pragma unroll
    for (int j = 0; j < TL::NR; ++j) a.dw[idx(j)] = a.acc ? a.dw[idx(j)] + v[j] : v[j];
    return;
  }
#pragma unroll
  for (int j = 0; j < TL::NR; ++j) rc_st(a.ws + (size_t)zi * numel + idx(j), v[j]);
  unsigned* ticket = a.counter + ((size_t)tap * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  if (!rc_last_workgroup(ticket, (unsigned)a.nz, &last)) return;
#pragma unroll
  for (int j = 0; j < TL::NR; ++j) {
    a.dw[idx(j)] = rc_sum_strided(a.ws + idx(j), numel, a.nz, a.acc ? a.dw[idx(j)] : 0.f);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// elementwise pieces
// ---------------------------------------------------------------------------------------------------------------------
// end of a bottleneck: s_out = bn3(Y3) + (bn_d(Yd) | relu?(s_in)); 4 channels per thread
__global__ __launch_bounds__(256) void rc_blockout_kernel(const float* __restrict__ Y3, const float* __restrict__ m3,
                                                          const float* __restrict__ sc3, const float* __restrict__ b3,
                                                          const float* __restrict__ R, const float* __restrict__ mr,
                                                          const float* __restrict__ scr, const float* __restrict__ br,
                                                          int res_relu, float* __restrict__ out, long n4, int C) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const int c = (int)((i * 4) % C);
  const float4 y = reinterpret_cast<const float4*>(Y3)[i], r = reinterpret_cast<const float4*>(R)[i];
  const float4 m = *reinterpret_cast<const float4*>(m3 + c), s = *reinterpret_cast<const float4*>(sc3 + c),
               b = *reinterpret_cast<const float4*>(b3 + c);
  float4 o;
  if (mr) {
    const float4 m2 = *reinterpret_cast<const float4*>(mr + c), s2 = *reinterpret_cast<const float4*>(scr + c),
                 b2 = *reinterpret_cast<const float4*>(br + c);
    o.x = rc_bn(y.x, m.x, s.x, b.x) + rc_bn(r.x, m2.x, s2.x, b2.x);
    o.y = rc_bn(y.y, m.y, s.y, b.y) + rc_bn(r.y, m2.y, s2.y, b2.y);
    o.z = rc_bn(y.z, m.z, s.z, b.z) + rc_bn(r.z, m2.z, s2.z, b2.z);
    o.w = rc_bn(y.w, m.w, s.w, b.w) + rc_bn(r.w, m2.w, s2.w, b2.w);
  } else {
    o.x = rc_bn(y.x, m.x, s.x, b.x) + (res_relu ? fmaxf(r.x, 0.f) : r.x);
    o.y = rc_bn(y.y, m.y, s.y, b.y) + (res_relu ? fmaxf(r.y, 0.f) : r.y);
    o.z = rc_bn(y.z, m.z, s.z, b.z) + (res_relu ? fmaxf(r.z, 0.f) : r.z);
    o.w = rc_bn(y.w, m.w, s.w, b.w) + (res_relu ? fmaxf(r.w, 0.f) : r.w);
  }
  reinterpret_cast<float4*>(out)[i] = o;
}

// out = bn(Y) with the engine's own expression (what the prologues and masks evaluate): diagnostics / ReLU-mask export
__global__ __launch_bounds__(256) void rc_bn_apply_kernel(const float* __restrict__ Y, const float* __restrict__ m,
                                                          const float* __restrict__ sc, const float* __restrict__ b,
                                                          float* __restrict__ out, long n, int C) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int c = (int)(i % C);
  out[i] = rc_bn(Y[i], m[c], sc[c], b[c]);
}

// source-row tables of a k x k / stride S / padding P convolution over (B, H, W) pixels:
//   fwd[tap][(b,oh,ow)] = row of (b, oh S - P + kh, ow S - P + kw) or -1
//   bwd[tap][(b,ih,iw)] = row of the output pixel that reads (b,ih,iw) through tap (kh,kw) or -1
__global__ __launch_bounds__(256) void rc_tables_kernel(int* __restrict__ fwd, int* __restrict__ bwd, int B, int H, int W,
                                                        int Ho, int Wo, int K, int S, int P) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long M = (long)B * Ho * Wo, Min = (long)B * H * W;
  const int T = K * K;
  if (i < M * T) {
    const int tap = (int)(i / M);
    const long m = i % M;
    const int b = (int)(m / (Ho * Wo)), oh = (int)(m / Wo % Ho), ow = (int)(m % Wo);
    const int ih = oh * S - P + tap / K, iw = ow * S - P + tap % K;
    fwd[i] = (ih >= 0 && ih < H && iw >= 0 && iw < W) ? (b * H + ih) * W + iw : -1;
  }
  if (i < Min * T) {
    const int tap = (int)(i / Min);
    const long m = i % Min;
    const int b = (int)(m / (H * W)), ih = (int)(m / W % H), iw = (int)(m % W);
    const int th = ih + P - tap / K, tw = iw + P - tap % K;
    int r = -1;
    if (th >= 0 && tw >= 0 && th % S == 0 && tw % S == 0 && th / S < Ho && tw / S < Wo) r = (b * Ho + th / S) * Wo + tw / S;
    bwd[i] = r;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------------
static inline bool rc_small(long tiles64) { return tiles64 < 256; }
static inline int rc_bk_small(int K) { return K % 128 == 0 ? 128 : 64; }

extern "C" int mmvae_rc_row_tile(int M, int N) {   // rows per statistics partial for an (M, N) output
  const long t64 = (long)((M + 63) / 64) * (N / 64);
  return rc_small(t64) ? 32 : 64;
}

extern "C" int mmvae_rc_tables(int* fwd, int* bwd, int B, int H, int W, int K, int S, int P, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(fwd && bwd && B > 0 && H > 0 && W > 0 && K > 0 && S > 0);
  const int Ho = (H + 2 * P - K) / S + 1, Wo = (W + 2 * P - K) / S + 1;
  const long n = (long)B * (long)max(H * W, Ho * Wo) * K * K;
  hipLaunchKernelGGL(rc_tables_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, fwd, bwd, B, H,
                     W, Ho, Wo, K, S, P);
  return mmvae_launch_status();
}

extern "C" int mmvae_rc_conv_fwd(const float* x, const float* w, const float* xmean, const float* xsc, const float* xbeta,
                                 const int* tbl, float* y, int M, int Cin, int Cout, int T, int pre,
                                 const float* gamma, const float* beta, float* run_mean, float* run_var, float* mean,
                                 float* rstd, float* sc, float* part, unsigned* counter, float eps, float momentum,
                                 int eval, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(x && w && y && M > 0 && Cin % 64 == 0 && Cout % 64 == 0 && T >= 1);
  MMVAE_CHECK_ARG(pre != RC_PRE_BN_RELU || (xmean && xsc && xbeta));
  MMVAE_CHECK_ARG(!part || (gamma && beta && mean && rstd && sc && counter));
  RcFwdArgs a{x, w, xmean, xsc, xbeta, tbl, y, M, Cin, Cout, T, pre,
              {gamma, beta, run_mean, run_var, mean, rstd, sc, part, counter, eps, momentum, eval}};
  hipStream_t st = (hipStream_t)stream;
  if (mmvae_rc_row_tile(M, Cout) == 64) {
    hipLaunchKernelGGL((rc_fwd_kernel<64, 64, 32>), dim3(Cout / 64, (M + 63) / 64), dim3(256), 0, st, a);
  } else if (rc_bk_small(Cin) == 128) {
    hipLaunchKernelGGL((rc_fwd_kernel<32, 32, 128>), dim3(Cout / 32, (M + 31) / 32), dim3(256), 0, st, a);
  } else {
    hipLaunchKernelGGL((rc_fwd_kernel<32, 32, 64>), dim3(Cout / 32, (M + 31) / 32), dim3(256), 0, st, a);
  }
  return mmvae_launch_status();
}

extern "C" int mmvae_rc_conv_dgrad(const float* G, const float* Y, const float* pqr, const float* w, const int* tbl,
                                   const float* add, int mask, const float* mY, const float* mmean, const float* msc,
                                   const float* mbeta, float* out, int M, int Min, int Cin, int Cout, int T, int nstat,
                                   const mmvae_rc_stat_t* st0, const mmvae_rc_stat_t* st1, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(G && w && out && M > 0 && Min > 0 && Cin % 64 == 0 && Cout % 64 == 0 && T >= 1);
  MMVAE_CHECK_ARG((!pqr || Y) && nstat >= 0 && nstat <= 2 && (nstat < 1 || st0) && (nstat < 2 || st1));
  MMVAE_CHECK_ARG(mask == RC_MASK_NONE || mY);
  MMVAE_CHECK_ARG(mask != RC_MASK_BN || (mmean && msc && mbeta));
  RcDgradArgs a{G, Y, pqr, w, tbl, add, mY, mmean, msc, mbeta, out, M, Min, Cin, Cout, T, mask, nstat, {}};
  const mmvae_rc_stat_t* sts[2] = {st0, st1};
  for (int t = 0; t < nstat; ++t)
    a.st[t] = RcStat{sts[t]->Y, sts[t]->mean, sts[t]->rstd, sts[t]->gamma, sts[t]->pqr, sts[t]->dgamma, sts[t]->dbeta,
                     sts[t]->part, sts[t]->counter, sts[t]->acc, sts[t]->eval};
  hipStream_t s = (hipStream_t)stream;
  if (mmvae_rc_row_tile(Min, Cin) == 64) {
    hipLaunchKernelGGL((rc_dgrad_kernel<64, 64, 32>), dim3(Cin / 64, (Min + 63) / 64), dim3(256), 0, s, a);
  } else if (rc_bk_small(Cout) == 128) {
    hipLaunchKernelGGL((rc_dgrad_kernel<32, 32, 128>), dim3(Cin / 32, (Min + 31) / 32), dim3(256), 0, s, a);
  } else {
    hipLaunchKernelGGL((rc_dgrad_kernel<32, 32, 64>), dim3(Cin / 32, (Min + 31) / 32), dim3(256), 0, s, a);
  }
  return mmvae_launch_status();
}

extern "C" int mmvae_rc_bn_bwd_stats(const float* G, const mmvae_rc_stat_t* st, int M, int C, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(G && st && st->Y && st->pqr && st->part && st->counter && M > 0 && C % 64 == 0);
  RcStat s{st->Y, st->mean, st->rstd, st->gamma, st->pqr, st->dgamma, st->dbeta, st->part, st->counter, st->acc, st->eval};
  hipLaunchKernelGGL(rc_stat_kernel, dim3(C / 64, (M + 63) / 64), dim3(256), 0, (hipStream_t)stream, G, s, M, C);
  return mmvae_launch_status();
}

// split of the pixel rows of a weight gradient: enough workgroups to fill the chip, >= 256 rows per split
extern "C" int mmvae_rc_wgrad_splits(int M, int Cin, int Cout, int T) {
  const long t64 = (long)(Cout / 64) * (Cin / 64) * T;
  const long tiles = rc_small(t64) ? t64 * 4 : t64;
  long nz = (768 + tiles - 1) / tiles;
  const long maxz = (M + 255) / 256;
  if (nz > maxz) nz = maxz;
  if (nz > 64) nz = 64;
  return (int)(nz < 1 ? 1 : nz);
}
extern "C" size_t mmvae_rc_wgrad_ws_floats(int M, int Cin, int Cout, int T) {
  const int nz = mmvae_rc_wgrad_splits(M, Cin, Cout, T);
  return nz > 1 ? (size_t)nz * Cout * T * Cin : 0;
}
extern "C" size_t mmvae_rc_wgrad_tickets(int Cin, int Cout, int T) { return (size_t)(Cout / 32) * (Cin / 32) * T; }

extern "C" int mmvae_rc_conv_wgrad(const float* G, const float* Y, const float* pqr, const float* x, const float* xmean,
                                   const float* xsc, const float* xbeta, const int* tbl, float* dw, float* ws,
                                   unsigned* counter, int M, int Cin, int Cout, int T, int pre, int accumulate,
                                   mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(G && x && dw && M > 0 && Cin % 64 == 0 && Cout % 64 == 0 && T >= 1 && (!pqr || Y));
  MMVAE_CHECK_ARG(pre != RC_PRE_BN_RELU || (xmean && xsc && xbeta));
  const int nz = mmvae_rc_wgrad_splits(M, Cin, Cout, T);
  MMVAE_CHECK_ARG(nz == 1 || (ws && counter));
  const long t64 = (long)(Cout / 64) * (Cin / 64) * T;
  const bool small = rc_small(t64);
  const int bk = small ? 128 : 32;
  int kper = (M + nz - 1) / nz;
  kper = (kper + bk - 1) / bk * bk;
  RcWgradArgs a{G, Y, pqr, x, xmean, xsc, xbeta, tbl, dw, ws, counter, M, Cin, Cout, T, pre, accumulate ? 1 : 0, nz, kper};
  hipStream_t s = (hipStream_t)stream;
  if (!small) hipLaunchKernelGGL((rc_wgrad_kernel<64, 64, 32>), dim3(Cin / 64, Cout / 64, T * nz), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((rc_wgrad_kernel<32, 32, 128>), dim3(Cin / 32, Cout / 32, T * nz), dim3(256), 0, s, a);
  return mmvae_launch_status();
}

extern "C" int mmvae_rc_blockout(const float* Y3, const float* m3, const float* sc3, const float* b3, const float* R,
                                 const float* mr, const float* scr, const float* br, int res_relu, float* out, long rows,
                                 int C, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(Y3 && m3 && sc3 && b3 && R && out && rows > 0 && C % 4 == 0 && (!mr || (scr && br)));
  const long n4 = rows * C / 4;
  hipLaunchKernelGGL(rc_blockout_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, Y3, m3, sc3,
                     b3, R, mr, scr, br, res_relu, out, n4, C);
  return mmvae_launch_status();
}

extern "C" int mmvae_rc_bn_apply(const float* Y, const float* mean, const float* sc, const float* beta, float* out,
                                 long rows, int C, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(Y && mean && sc && beta && out && rows > 0 && C > 0);
  const long n = rows * C;
  hipLaunchKernelGGL(rc_bn_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, Y, mean, sc,
                     beta, out, n, C);
  return mmvae_launch_status();
}
