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

// workgroup coordinates of a job (its own blockIdx / gridDim when the job is a launch of its own; a slice of the
// linear grid when several independent jobs share one launch: rc_group_kernel)
struct RcBlk { int bx, by, bz, gx, gy; };

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
  float* ws;              // nz > 1: (nz, M, Cout) partial outputs
  unsigned* tile_ticket;  // nz > 1: one per output tile
  int M, Cin, Cout, T, pre, nz;
  RcBnFwd bn;             // bn.part == NULL: no statistics
};

#define RC_GROUP 16   // row tiles whose statistics partials one representative workgroup condenses
// Merge of the partial pairs p in [pb, pe) of column n, all 256 threads of the workgroup (thread = (col, rg)), loads of 8
// partials per thread in flight.  MEANVAR: pairs are (mean, M2) of rows_per rows each (the part that covers the end of
// the M rows: fewer), merged as double sums shifted by the first part's mean -- no division per part, no cancellation
// that double does not absorb; out (count, mean, M2).  Otherwise plain sums of both members; out (-, sum0, sum1).
template <int BN, bool MEANVAR>
__device__ __forceinline__ void rc_merge(const float* __restrict__ part, int C, int n, int pb, int pe, int rows_per, int M,
                                         double* __restrict__ dl, int col, int rg, double& o0, double& o1, double& o2) {
  constexpr int RG = 256 / BN;
  const double r = MEANVAR ? (double)rc_ld(part + ((size_t)pb * C + n) * 2) : 0.0;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (int p0 = pb + rg; p0 < pe; p0 += RG * 8) {
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int p = min(p0 + i * RG, pe - 1);
      a[i] = rc_ld(part + ((size_t)p * C + n) * 2);
      b[i] = rc_ld(part + ((size_t)p * C + n) * 2 + 1);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int p = p0 + i * RG;
      if (p < pe) {
        if (MEANVAR) {
          const double cnt = (double)min(rows_per, M - p * rows_per), d = (double)a[i] - r;
          s0 += cnt;
          s1 += cnt * d;
          s2 += (double)b[i] + cnt * d * d;
        } else {
          s1 += (double)a[i];
          s2 += (double)b[i];
        }
      }
    }
  }
  __syncthreads();
  dl[(rg * BN + col) * 3] = s0;
  dl[(rg * BN + col) * 3 + 1] = s1;
  dl[(rg * BN + col) * 3 + 2] = s2;
  __syncthreads();
  s0 = 0.0; s1 = 0.0; s2 = 0.0;
#pragma unroll
  for (int g = 0; g < RG; ++g) { s0 += dl[(g * BN + col) * 3]; s1 += dl[(g * BN + col) * 3 + 1]; s2 += dl[(g * BN + col) * 3 + 2]; }
  if (MEANVAR) { o0 = s0; o1 = r + s1 / s0; o2 = s2 - s1 * s1 / s0; } else { o0 = 0.0; o1 = s1; o2 = s2; }
}

// Where the statistics of one column tile live.  Level 0: one partial pair per row tile; with more than RC_GROUP row
// tiles the last workgroup of every group of RC_GROUP condenses them into one level-1 pair (so that a ticket never sees
// more than RC_GROUP ... nrow / RC_GROUP arrivals: ~50 ns each, serialised on one address) and the representatives
// elect the finalizer.  counter: [column tiles] final tickets, then [column tiles][groups] group tickets.
struct RcLevels {
  int nrow, ngrp, grp, g0, g1;
  __device__ __forceinline__ RcLevels(int nrow_, int by) : nrow(nrow_) {
    ngrp = nrow > RC_GROUP ? (nrow + RC_GROUP - 1) / RC_GROUP : 0;
    grp = by / RC_GROUP;
    g0 = grp * RC_GROUP;
    g1 = min(nrow, g0 + RC_GROUP);
  }
  __device__ __forceinline__ float* level1(float* part, int C) const { return part + (size_t)nrow * C * 2; }
};

template <int BN>
__device__ __forceinline__ void rc_bn_fwd_finalize(const RcBnFwd& bn, double mean, double m2, int M, int n) {
  const double cn = (double)M, var = m2 / cn;
  double rs = 1.0 / sqrt(var + (double)bn.eps);
  if (bn.eval) {
    mean = (double)bn.run_mean[n];
    rs = 1.0 / sqrt((double)bn.run_var[n] + (double)bn.eps);
  } else if (bn.run_mean) {
    const double mo = (double)bn.momentum, unb = cn > 1.0 ? m2 / (cn - 1.0) : var;
    bn.run_mean[n] = (float)((1.0 - mo) * (double)bn.run_mean[n] + mo * mean);
    bn.run_var[n] = (float)((1.0 - mo) * (double)bn.run_var[n] + mo * unb);
  }
  bn.mean[n] = (float)mean;
  bn.rstd[n] = (float)rs;
  bn.sc[n] = (float)((double)bn.gamma[n] * rs);
}

// The complete values of this workgroup's output tile.  nz == 1: the LDS tile.  nz > 1 (split of the reduction over
// blockIdx.z, 32 x 32 tiles only): every split leaves its partial tile in ws [z][rows][N]; the last one to arrive sums
// them in z order (NR x 8 loads in flight) and carries on alone -- false for the others.
template <class TL, int BM, int BN>
__device__ __forceinline__ bool rc_tile_values(const float* __restrict__ smem, float* __restrict__ ws,
                                               unsigned* __restrict__ ticket, int nz, int zi, size_t zstride, int m0, int n0,
                                               int N, int cnt, int col, int rg, float (&v)[TL::NR], int* last) {
#pragma unroll
  for (int j = 0; j < TL::NR; ++j) v[j] = TL::tile_at(smem, rg + j * TL::RG, col);
  if (nz == 1) return true;
  if constexpr (TL::NR <= 4) {
#pragma unroll
    for (int j = 0; j < TL::NR; ++j) {
      const int row = rg + j * TL::RG;
      if (row < cnt) rc_st(ws + (size_t)zi * zstride + (size_t)(m0 + row) * N + n0 + col, v[j]);
    }
    if (!rc_last_workgroup(ticket, (unsigned)nz, last)) return false;
#pragma unroll
    for (int j = 0; j < TL::NR; ++j) v[j] = 0.f;
    for (int z0 = 0; z0 < nz; z0 += 8) {
      float t[TL::NR][8];
#pragma unroll
      for (int j = 0; j < TL::NR; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i)
          t[j][i] = rc_ld(ws + (size_t)min(z0 + i, nz - 1) * zstride + (size_t)(m0 + min(rg + j * TL::RG, cnt - 1)) * N +
                          n0 + col);
#pragma unroll
      for (int j = 0; j < TL::NR; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (z0 + i < nz) v[j] += t[j][i];
    }
    return true;
  } else {
    return false;   // the dispatcher never splits the 64 x 64 tiling
  }
}

template <int BM, int BN, int BK>
__device__ __forceinline__ void rc_fwd_body(const RcFwdArgs& a, const RcBlk k, float* __restrict__ smem, float* __restrict__ cs,
                                          int* __restrict__ lastp) {
  using TL = RcTile<BM, BN, BK>;
  using SA = RcStg<BM, BK, true>;
  using SB = RcStg<BN, BK, true>;
  float* As = smem;
  float* Bs = smem + BK * TL::AP;
  const int tid = threadIdx.x;
  const int n0 = k.bx * BN, m0 = k.by * BM;
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
  // this workgroup's share of the T * Cin / BK stages (k.bz of nz)
  const int nstage_all = a.T * (a.Cin / BK), sper = (nstage_all + a.nz - 1) / a.nz;
  const int s_beg = k.bz * sper, s_end = min(nstage_all, s_beg + sper);
  int ltap = s_beg / (a.Cin / BK), lc0 = (s_beg % (a.Cin / BK)) * BK;
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
  if (s_beg < s_end) load();
#pragma unroll 1
  for (int s = s_beg; s < s_end; ++s) {
    store();
    __syncthreads();
    if (s + 1 < s_end) load();
    tl.mma(As, Bs, acc);
    __syncthreads();
  }
  tl.spill(smem, acc);
  __syncthreads();
  const int col = tid % BN, rg = tid / BN, n = n0 + col;
  const int cnt = min(BM, a.M - m0);
  float v[TL::NR], s = 0.f;
  if (!rc_tile_values<TL, BM, BN>(smem, a.ws, a.tile_ticket + k.by * k.gx + k.bx, a.nz, k.bz,
                                  (size_t)a.M * a.Cout, m0, n0, a.Cout, cnt, col, rg, v, lastp))
    return;
#pragma unroll
  for (int j = 0; j < TL::NR; ++j) {
    const int row = rg + j * TL::RG;
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
    rc_st(a.bn.part + ((size_t)k.by * a.Cout + n) * 2, mean_t);
    rc_st(a.bn.part + ((size_t)k.by * a.Cout + n) * 2 + 1, m2_t);
  }
  const RcLevels lv(k.gy, k.by);
  double* dl = reinterpret_cast<double*>(smem);
  double o0, o1, o2;
  const float* src_part = a.bn.part;
  int np = lv.nrow, rows_per = BM;
  if (lv.ngrp) {
    if (!rc_last_workgroup(a.bn.counter + k.gx + k.bx * lv.ngrp + lv.grp, lv.g1 - lv.g0, lastp)) return;
    rc_merge<BN, true>(a.bn.part, a.Cout, n, lv.g0, lv.g1, BM, a.M, dl, col, rg, o0, o1, o2);
    float* l1 = lv.level1(a.bn.part, a.Cout);
    if (rg == 0) {
      rc_st(l1 + ((size_t)lv.grp * a.Cout + n) * 2, (float)o1);
      rc_st(l1 + ((size_t)lv.grp * a.Cout + n) * 2 + 1, (float)o2);
    }
    src_part = l1;
    np = lv.ngrp;
    rows_per = BM * RC_GROUP;
  }
  if (!rc_last_workgroup(a.bn.counter + k.bx, np, lastp)) return;
  rc_merge<BN, true>(src_part, a.Cout, n, 0, np, rows_per, a.M, dl, col, rg, o0, o1, o2);
  if (rg == 0) rc_bn_fwd_finalize<BN>(a.bn, o1, o2, a.M, n);
}

template <int BM, int BN, int BK>
__global__ __launch_bounds__(256) void rc_fwd_kernel(RcFwdArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[RcTile<BM, BN, BK>::SMEM];
  __shared__ float cs[256];
  __shared__ int last;
  rc_fwd_body<BM, BN, BK>(a, RcBlk{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.x, (int)gridDim.y}, smem, cs,
                          &last);
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

__device__ __forceinline__ void rc_stat_finalize(const RcStat& st, double s1, double s2, int M, int C, int n) {
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

// the hierarchical election + merge of the backward statistics of nstat BatchNorms (partials already stored)
template <int BN>
__device__ __forceinline__ void rc_stat_tail(const RcStat* st, int nstat, int nrow, int by, int bx, int ncoltiles, int M,
                                             int C, int n, double* dl, int col, int rg, int* last) {
  const RcLevels lv(nrow, by);
  double o0, o1, o2;
  int np = lv.nrow;
  bool l1 = false;
  if (lv.ngrp) {
    if (!rc_last_workgroup(st[0].counter + ncoltiles + bx * lv.ngrp + lv.grp, lv.g1 - lv.g0, last)) return;
#pragma unroll
    for (int t = 0; t < 2; ++t)
      if (t < nstat) {
        rc_merge<BN, false>(st[t].part, C, n, lv.g0, lv.g1, 0, 0, dl, col, rg, o0, o1, o2);
        float* d = lv.level1(st[t].part, C);
        if (rg == 0) {
          rc_st(d + ((size_t)lv.grp * C + n) * 2, (float)o1);
          rc_st(d + ((size_t)lv.grp * C + n) * 2 + 1, (float)o2);
        }
      }
    np = lv.ngrp;
    l1 = true;
  }
  if (!rc_last_workgroup(st[0].counter + bx, np, last)) return;
#pragma unroll
  for (int t = 0; t < 2; ++t)
    if (t < nstat) {
      rc_merge<BN, false>(l1 ? lv.level1(st[t].part, C) : st[t].part, C, n, 0, np, 0, 0, dl, col, rg, o0, o1, o2);
      if (rg == 0) rc_stat_finalize(st[t], o1, o2, M, C, n);
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
  const float* add;       // added before the mask (the shortcut's gradient) or NULL: (Min, Cin), or ...
  const int* add_tbl;     // ... with add_tbl (Min): row add_tbl[m] of `add` (-1: nothing), a strided projection's gradient
  const float* mY;        // mask source (Min, Cin): RC_MASK_RAW mY > 0, RC_MASK_BN bn(mY) > 0
  const float* mmean;
  const float* msc;
  const float* mbeta;
  float* out;             // (Min, Cin)
  float* ws;              // nz > 1: (nz, Min, Cin) partial outputs
  unsigned* tile_ticket;
  int M, Min, Cin, Cout, T, mask, nstat, nz;
  RcStat st[2];
};

template <int BM, int BN, int BK>
__device__ __forceinline__ void rc_dgrad_body(const RcDgradArgs& a, const RcBlk k, float* __restrict__ smem, float* __restrict__ cs,
                                          int* __restrict__ lastp) {
  using TL = RcTile<BM, BN, BK>;
  using SA = RcStg<BM, BK, true>;
  using SB = RcStg<BN, BK, false>;
  float* As = smem;
  float* Bs = smem + BK * TL::AP;
  const int tid = threadIdx.x;
  const int n0 = k.bx * BN, m0 = k.by * BM;     // n0: input-channel tile
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
  const int nstage_all = a.T * (a.Cout / BK), sper = (nstage_all + a.nz - 1) / a.nz;
  const int s_beg = k.bz * sper, s_end = min(nstage_all, s_beg + sper);
  int ltap = s_beg / (a.Cout / BK), lk0 = (s_beg % (a.Cout / BK)) * BK;
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
  if (s_beg < s_end) load();
#pragma unroll 1
  for (int s = s_beg; s < s_end; ++s) {
    store();
    __syncthreads();
    if (s + 1 < s_end) load();
    tl.mma(As, Bs, acc);
    __syncthreads();
  }
  tl.spill(smem, acc);
  __syncthreads();
  const int col = tid % BN, rg = tid / BN, c = n0 + col;
  const int cnt = min(BM, a.Min - m0);
  float v[TL::NR];
  if (!rc_tile_values<TL, BM, BN>(smem, a.ws, a.tile_ticket + k.by * k.gx + k.bx, a.nz, k.bz,
                                  (size_t)a.Min * a.Cin, m0, n0, a.Cin, cnt, col, rg, v, lastp))
    return;
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
      float g = v[j];
      if (a.add) {
        if (a.add_tbl) {
          const int r = a.add_tbl[m0 + row];
          if (r >= 0) g += a.add[(size_t)r * a.Cin + c];
        } else {
          g += a.add[o];
        }
      }
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
        rc_st(a.st[t].part + ((size_t)k.by * a.Cin + c) * 2, a1);
        rc_st(a.st[t].part + ((size_t)k.by * a.Cin + c) * 2 + 1, a2);
      }
    }
  rc_stat_tail<BN>(a.st, a.nstat, k.gy, k.by, k.bx, k.gx, a.Min, a.Cin, c,
                   reinterpret_cast<double*>(smem), col, rg, lastp);
}

template <int BM, int BN, int BK>
__global__ __launch_bounds__(256) void rc_dgrad_kernel(RcDgradArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[RcTile<BM, BN, BK>::SMEM];
  __shared__ float cs[256];
  __shared__ int last;
  rc_dgrad_body<BM, BN, BK>(a, RcBlk{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.x, (int)gridDim.y}, smem, cs,
                          &last);
}

// stand-alone statistics of a BatchNorm backward whose G was produced elsewhere (pooling backward, tests):
// grid (C / 64, row tiles of 64)
__global__ __launch_bounds__(256) void rc_stat_kernel(const float* __restrict__ G, RcStat st, int M, int C) {
  __shared__ __attribute__((aligned(16))) float smem[256 * 3 * 2];
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
  rc_stat_tail<64>(&st, 1, gridDim.y, blockIdx.y, blockIdx.x, gridDim.x, M, C, c, reinterpret_cast<double*>(smem), col, rg,
                   &last);
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
__device__ __forceinline__ void rc_wgrad_body(const RcWgradArgs& a, const RcBlk k, float* __restrict__ smem, float* __restrict__ cs,
                                          int* __restrict__ lastp) {
  using TL = RcTile<BM, BN, BK>;
  using SA = RcStg<BM, BK, false>;
  using SB = RcStg<BN, BK, false>;
  float* As = smem;
  float* Bs = smem + BK * TL::AP;
  const int tid = threadIdx.x;
  const int c0 = k.bx * BN, n0 = k.by * BM;
  const int tap = k.bz / a.nz, zi = k.bz % a.nz;
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
  unsigned* ticket = a.counter + ((size_t)tap * k.gy + k.by) * k.gx + k.bx;
  if (!rc_last_workgroup(ticket, (unsigned)a.nz, lastp)) return;
#pragma unroll
  for (int j = 0; j < TL::NR; ++j) {
    a.dw[idx(j)] = rc_sum_strided(a.ws + idx(j), numel, a.nz, a.acc ? a.dw[idx(j)] : 0.f);
  }
}

template <int BM, int BN, int BK>
__global__ __launch_bounds__(256) void rc_wgrad_kernel(RcWgradArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[RcTile<BM, BN, BK>::SMEM];
  __shared__ float cs[256];
  __shared__ int last;
  rc_wgrad_body<BM, BN, BK>(a, RcBlk{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.x, (int)gridDim.y}, smem, cs,
                          &last);
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
// several independent jobs in ONE launch (a layer's data and weight gradient; a block's first convolution and its
// projection shortcut): at batch 24 every job alone leaves most of the chip idle and costs a dependent launch
// ---------------------------------------------------------------------------------------------------------------------
#define RC_KIND_FWD 0
#define RC_KIND_DGRAD 1
#define RC_KIND_WGRAD 2
#define RC_CFG_L32 0      // 64 x 64 tiles, 32-deep stages
#define RC_CFG_S128 1     // 32 x 32 tiles, 4 waves split 128-deep stages
#define RC_CFG_S64 2      // 32 x 32 tiles, 64-deep stages (64-channel reductions)
struct RcPlan { int kind, cfg, gx, gy, gz; };
struct RcGroup {
  RcFwdArgs f[MMVAE_RC_MAX_JOBS];
  RcDgradArgs d[MMVAE_RC_MAX_JOBS];
  RcWgradArgs w[MMVAE_RC_MAX_JOBS];
  RcPlan plan[MMVAE_RC_MAX_JOBS];
  int blk0[MMVAE_RC_MAX_JOBS + 1];
  int n;
};

template <int BM, int BN, int BK>
__device__ __forceinline__ void rc_job(const RcGroup* __restrict__ g, int kind, int p, const RcBlk k, float* smem, float* cs,
                                       int* lastp) {
  if (kind == RC_KIND_FWD) {
    const RcFwdArgs a = g->f[p];
    rc_fwd_body<BM, BN, BK>(a, k, smem, cs, lastp);
  } else if (kind == RC_KIND_DGRAD) {
    const RcDgradArgs a = g->d[p];
    rc_dgrad_body<BM, BN, BK>(a, k, smem, cs, lastp);
  } else {
    const RcWgradArgs a = g->w[p];
    rc_wgrad_body<BM, BN, BK>(a, k, smem, cs, lastp);
  }
}

// (the job table is read through the kernel-argument segment pointer: indexing the by-value parameter with the
// workgroup's job number makes the compiler copy all 2.6 KB of it into scratch -- 2.5 KB per lane, 10 x the run time)
__global__ __launch_bounds__(256) void rc_group_kernel(RcGroup g_) {
  __shared__ __attribute__((aligned(16))) float smem[RcTile<32, 32, 128>::SMEM];
  __shared__ float cs[256];
  __shared__ int last;
  const RcGroup* __restrict__ g = (const RcGroup*)__builtin_amdgcn_kernarg_segment_ptr();
  int p = 0;
#pragma unroll
  for (int q = 1; q < MMVAE_RC_MAX_JOBS; ++q)
    if (q < g->n && (int)blockIdx.x >= g->blk0[q]) p = q;
  const int local = blockIdx.x - g->blk0[p];
  const RcPlan pl = g->plan[p];
  const RcBlk k{local % pl.gx, (local / pl.gx) % pl.gy, local / (pl.gx * pl.gy), pl.gx, pl.gy};
  if (pl.cfg == RC_CFG_L32) rc_job<64, 64, 32>(g, pl.kind, p, k, smem, cs, &last);
  else if (pl.cfg == RC_CFG_S128) rc_job<32, 32, 128>(g, pl.kind, p, k, smem, cs, &last);
  else rc_job<32, 32, 64>(g, pl.kind, p, k, smem, cs, &last);
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

// split of the reduction (taps x channels) of a forward / data-gradient GEMM with an (M, N) output over blockIdx.z: only
// the 32 x 32 tiling (few tiles), >= 2 stages per split, enough workgroups for two per CU
extern "C" int mmvae_rc_conv_splits(int M, int N, int K, int T) {
  if (mmvae_rc_row_tile(M, N) == 64) return 1;
  const long tiles = (long)((M + 31) / 32) * (N / 32);
  const int nstage = T * (K / rc_bk_small(K));
  long nz = (768 + tiles - 1) / tiles;
  if (nz > (nstage + 1) / 2) nz = (nstage + 1) / 2;
  if (nz > 16) nz = 16;
  return (int)(nz < 1 ? 1 : nz);
}
extern "C" size_t mmvae_rc_conv_ws_floats(int M, int N, int K, int T) {
  const int nz = mmvae_rc_conv_splits(M, N, K, T);
  return nz > 1 ? (size_t)nz * M * N : 0;
}

extern "C" int mmvae_rc_tables(int* fwd, int* bwd, int B, int H, int W, int K, int S, int P, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(fwd && bwd && B > 0 && H > 0 && W > 0 && K > 0 && S > 0);
  const int Ho = (H + 2 * P - K) / S + 1, Wo = (W + 2 * P - K) / S + 1;
  const long n = (long)B * (long)max(H * W, Ho * Wo) * K * K;
  hipLaunchKernelGGL(rc_tables_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, fwd, bwd, B, H,
                     W, Ho, Wo, K, S, P);
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

static RcStat rc_stat_of(const mmvae_rc_stat_t& s) {
  return RcStat{s.Y, s.mean, s.rstd, s.gamma, s.pqr, s.dgamma, s.dbeta, s.part, s.counter, s.acc, s.eval};
}

static int rc_plan_fwd(const mmvae_rc_fwd_t& j, RcFwdArgs& a, RcPlan& pl) {
  MMVAE_CHECK_ARG(j.x && j.w && j.y && j.M > 0 && j.Cin % 64 == 0 && j.Cout % 64 == 0 && j.T >= 1);
  MMVAE_CHECK_ARG(j.pre != RC_PRE_BN_RELU || (j.xmean && j.xsc && j.xbeta));
  MMVAE_CHECK_ARG(!j.part || (j.gamma && j.beta && j.mean && j.rstd && j.sc && j.counter));
  const int nz = mmvae_rc_conv_splits(j.M, j.Cout, j.Cin, j.T);
  MMVAE_CHECK_ARG(nz == 1 || (j.ws && j.tile_ticket));
  a = RcFwdArgs{j.x, j.w, j.xmean, j.xsc, j.xbeta, j.tbl, j.y, j.ws, j.tile_ticket, j.M, j.Cin, j.Cout, j.T, j.pre, nz,
                {j.gamma, j.beta, j.run_mean, j.run_var, j.mean, j.rstd, j.sc, j.part, j.counter, j.eps, j.momentum, j.eval}};
  if (mmvae_rc_row_tile(j.M, j.Cout) == 64) pl = RcPlan{RC_KIND_FWD, RC_CFG_L32, j.Cout / 64, (j.M + 63) / 64, 1};
  else pl = RcPlan{RC_KIND_FWD, rc_bk_small(j.Cin) == 128 ? RC_CFG_S128 : RC_CFG_S64, j.Cout / 32, (j.M + 31) / 32, nz};
  return MMVAE_OK;
}

static int rc_plan_dgrad(const mmvae_rc_dgrad_t& j, RcDgradArgs& a, RcPlan& pl) {
  MMVAE_CHECK_ARG(j.G && j.w && j.out && j.M > 0 && j.Min > 0 && j.Cin % 64 == 0 && j.Cout % 64 == 0 && j.T >= 1);
  MMVAE_CHECK_ARG((!j.pqr || j.Y) && j.nstat >= 0 && j.nstat <= 2);
  MMVAE_CHECK_ARG(j.mask == RC_MASK_NONE || j.mY);
  MMVAE_CHECK_ARG(j.mask != RC_MASK_BN || (j.mmean && j.msc && j.mbeta));
  const int nz = mmvae_rc_conv_splits(j.Min, j.Cin, j.Cout, j.T);
  MMVAE_CHECK_ARG(nz == 1 || (j.ws && j.tile_ticket));
  a = RcDgradArgs{j.G, j.Y, j.pqr, j.w, j.tbl, j.add, j.add_tbl, j.mY, j.mmean, j.msc, j.mbeta, j.out, j.ws, j.tile_ticket,
                  j.M, j.Min, j.Cin, j.Cout, j.T, j.mask, j.nstat, nz, {}};
  for (int t = 0; t < j.nstat; ++t) {
    MMVAE_CHECK_ARG(j.st[t].Y && j.st[t].pqr && j.st[t].part && j.st[t].counter);
    a.st[t] = rc_stat_of(j.st[t]);
  }
  if (mmvae_rc_row_tile(j.Min, j.Cin) == 64) pl = RcPlan{RC_KIND_DGRAD, RC_CFG_L32, j.Cin / 64, (j.Min + 63) / 64, 1};
  else pl = RcPlan{RC_KIND_DGRAD, rc_bk_small(j.Cout) == 128 ? RC_CFG_S128 : RC_CFG_S64, j.Cin / 32, (j.Min + 31) / 32, nz};
  return MMVAE_OK;
}

static int rc_plan_wgrad(const mmvae_rc_wgrad_t& j, RcWgradArgs& a, RcPlan& pl) {
  MMVAE_CHECK_ARG(j.G && j.x && j.dw && j.M > 0 && j.Cin % 64 == 0 && j.Cout % 64 == 0 && j.T >= 1 && (!j.pqr || j.Y));
  MMVAE_CHECK_ARG(j.pre != RC_PRE_BN_RELU || (j.xmean && j.xsc && j.xbeta));
  const int nz = mmvae_rc_wgrad_splits(j.M, j.Cin, j.Cout, j.T);
  MMVAE_CHECK_ARG(nz == 1 || (j.ws && j.counter));
  const bool small = rc_small((long)(j.Cout / 64) * (j.Cin / 64) * j.T);
  const int bk = small ? 128 : 32;
  int kper = (j.M + nz - 1) / nz;
  kper = (kper + bk - 1) / bk * bk;
  a = RcWgradArgs{j.G, j.Y, j.pqr, j.x, j.xmean, j.xsc, j.xbeta, j.tbl, j.dw, j.ws, j.counter, j.M, j.Cin, j.Cout, j.T, j.pre,
                  j.accumulate ? 1 : 0, nz, kper};
  if (!small) pl = RcPlan{RC_KIND_WGRAD, RC_CFG_L32, j.Cin / 64, j.Cout / 64, j.T * nz};
  else pl = RcPlan{RC_KIND_WGRAD, RC_CFG_S128, j.Cin / 32, j.Cout / 32, j.T * nz};
  return MMVAE_OK;
}

template <class A, class KL, class KS128, class KS64>
static void rc_launch_one(const A& a, const RcPlan& pl, hipStream_t st, KL kl, KS128 k128, KS64 k64) {
  const dim3 grid(pl.gx, pl.gy, pl.gz);
  if (pl.cfg == RC_CFG_L32) hipLaunchKernelGGL(kl, grid, dim3(256), 0, st, a);
  else if (pl.cfg == RC_CFG_S128) hipLaunchKernelGGL(k128, grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(k64, grid, dim3(256), 0, st, a);
}

extern "C" int mmvae_rc_launch(const mmvae_rc_job_t* jobs, int n, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(jobs && n >= 1 && n <= MMVAE_RC_MAX_JOBS);
  hipStream_t st = (hipStream_t)stream;
  static thread_local RcGroup g;     // ~2.5 KB of kernel arguments, assembled in place
  g.n = n;
  g.blk0[0] = 0;
  for (int p = 0; p < n; ++p) {
    int rc;
    if (jobs[p].kind == RC_KIND_FWD) rc = rc_plan_fwd(jobs[p].f, g.f[p], g.plan[p]);
    else if (jobs[p].kind == RC_KIND_DGRAD) rc = rc_plan_dgrad(jobs[p].d, g.d[p], g.plan[p]);
    else if (jobs[p].kind == RC_KIND_WGRAD) rc = rc_plan_wgrad(jobs[p].w, g.w[p], g.plan[p]);
    else return MMVAE_ERR_ARG;
    if (rc != MMVAE_OK) return rc;
    g.blk0[p + 1] = g.blk0[p] + g.plan[p].gx * g.plan[p].gy * g.plan[p].gz;
  }
  if (n == 1) {       // a job alone keeps its own kernel (and its name in a profile)
    const RcPlan& pl = g.plan[0];
    if (pl.kind == RC_KIND_FWD)
      rc_launch_one(g.f[0], pl, st, rc_fwd_kernel<64, 64, 32>, rc_fwd_kernel<32, 32, 128>, rc_fwd_kernel<32, 32, 64>);
    else if (pl.kind == RC_KIND_DGRAD)
      rc_launch_one(g.d[0], pl, st, rc_dgrad_kernel<64, 64, 32>, rc_dgrad_kernel<32, 32, 128>, rc_dgrad_kernel<32, 32, 64>);
    else
      rc_launch_one(g.w[0], pl, st, rc_wgrad_kernel<64, 64, 32>, rc_wgrad_kernel<32, 32, 128>, rc_wgrad_kernel<32, 32, 64>);
    return mmvae_launch_status();
  }
  hipLaunchKernelGGL(rc_group_kernel, dim3(g.blk0[n]), dim3(256), 0, st, g);
  return mmvae_launch_status();
}

extern "C" int mmvae_rc_bn_bwd_stats(const float* G, const mmvae_rc_stat_t* st, int M, int C, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(G && st && st->Y && st->pqr && st->part && st->counter && M > 0 && C % 64 == 0);
  hipLaunchKernelGGL(rc_stat_kernel, dim3(C / 64, (M + 63) / 64), dim3(256), 0, (hipStream_t)stream, G, rc_stat_of(*st), M, C);
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
