// Fused convolution + BatchNorm engine of the ResNet-50 image tower (`encoder: CNN`, reference models/encoders.py:86-127 =
// torchvision resnet50 -> SiLU -> heads).  gfx950 only.
//
// Activations are NHWC (rows = B*H*W, C) fp32 matrices; convolution weights are stored channels-last, (Cout, kh, kw, Cin)
// in memory behind the (Cout, Cin, kh, kw) parameter view, so that for every filter tap the reduction runs over
// contiguous input channels.  A k x k convolution is then a sum over its taps of 1x1 GEMMs whose A rows are shifted
// pixels: the im2col matrix only ever exists as 64-row tiles in LDS (the source pixel of a row and a tap is integer
// arithmetic on the row's (b, h, w), zero for the padding).
//
// One tiling: workgroup = 4 wavefronts = a 64 x 64 output tile (2 x 2 waves of 32 x 32 on v_mfma_f32_32x32x2_f32), 32-deep
// stages (RC_BK) staged global -> registers -> LDS [k][row] (pitch 65), the next stage's loads in flight under the MFMAs.  Jobs
// with few tiles split their reduction (taps x channels, or the pixel rows of a weight gradient) over blockIdx.z; the
// partial accumulators go to a workspace in the MFMA register layout and the last workgroup of a tile sums them in a
// fixed order (deterministic; no atomics on data) and runs the epilogue.
//
// What is fused around the tiles -- no BatchNorm, ReLU or im2col kernel is left between two convolutions:
//   forward   A prologue: relu(bn(Y_prev)) = max(fma(y - mean, gamma rstd, beta), 0) of the producer's RAW output;
//             epilogue: raw output + per-column (mean, M2) of the tile; the last workgroup of a column tile merges the
//             row tiles' partials (double), emits mean / rstd / gamma rstd and moves the running statistics.
//   dgrad     A prologue: the BatchNorm backward of the consumer side, dY = G p + Y q + r per channel (p, q, r from the
//             statistics sum G, sum G xhat); epilogue: (+ shortcut gradient), ReLU mask recomputed from the producer's
//             raw output, the statistics of THAT BatchNorm's backward, last workgroup: dgamma, dbeta, p, q, r.
//   wgrad     A prologue as dgrad (transposed), B prologue as forward.
// Cross-workgroup hand-over inside a launch: agent-scope write-through stores, a relaxed agent-scope ticket,
// agent-scope loads in the elected workgroup (latent.hip: poe_last_workgroup explains why not __threadfence()).  Every
// such step is a ~1.5 us round trip at device scope: loads of partials are issued 8+ at a time, and tickets with more
// than 16 arrivals are split in two levels.
#include <cstdlib>

#include "common.hpp"

// RC_PROBE build (tools/probe): wall-clock stamps (100 MHz) of thread 0 of every workgroup at fixed points of a job
#ifdef RC_PROBE
__device__ long long* rc_probe_ptr = nullptr;
extern "C" int mmvae_rc_probe(long long* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(rc_probe_ptr), &buf, sizeof(buf)) == hipSuccess ? 0 : 3;
}
#define RC_STAMP(k_, i)                                                                                     \
  do {                                                                                                      \
    if (threadIdx.x == 0 && rc_probe_ptr)                                                                   \
      rc_probe_ptr[((size_t)((k_).bz * (k_).gy + (k_).by) * (k_).gx + (k_).bx) * 8 + (i)] = wall_clock64(); \
  } while (0)
#define RC_STAMP_WAIT(k_, i)                                        \
  do {                                                              \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     \
    RC_STAMP(k_, i);                                                \
  } while (0)
#else
#define RC_STAMP(k_, i)
#define RC_STAMP_WAIT(k_, i)
#endif

#define RC_PRE_NONE 0
#define RC_PRE_RELU 1
#define RC_PRE_BN_RELU 2
#define RC_MASK_NONE 0
#define RC_MASK_RAW 1
#define RC_MASK_BN 2
// -DRC_HALF_B (make probe_rc_halfb; tools/probe/rc_time.py): an UPPER BOUND for what 128-row tiles could buy -- two row
// tiles sharing one staged weight tile stage the B operand half as often -- measured without building them: the odd stages
// neither load nor store their B tile (results are wrong, the instruction stream is the shared-B kernel's per row tile)
#ifdef RC_HALF_B
#define RC_B_STAGE(st) ((st) == 0)
#else
#define RC_B_STAGE(st) true
#endif
#define RC_BM 64
#ifndef RC_BK
#define RC_BK 32
#endif
#define RC_NS (RC_BK / 4)      // staging slots of a thread per 64 x RC_BK operand tile
#define RC_RSTEP (256 / RC_BK) // rows between a thread's slots when it walks a k-contiguous source
#define RC_AP 65      // LDS row pitch of a staged operand ([k][row]; odd: conflict-free transposing stores and fragment reads)

__device__ __forceinline__ float rc_bn(float y, float mean, float sc, float beta) { return fmaf(y - mean, sc, beta); }
__device__ __forceinline__ void rc_st(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float rc_ld(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

// Linear workgroup number of a job -> its coordinates.  Workgroup i of a launch runs on XCD i % 8 (tools/probe/
// xcd_scope.hip: without exception), and every XCD has its own L2: with `xcd` the row tile `by` always lands on XCD
// by % 8, in every launch, so that the rows one launch wrote through an XCD's L2 are read on that XCD by the next (the
// plain order puts row tile `by` wherever gx happens to send it).  `local` must be congruent to the hardware workgroup
// number mod 8; the job's grid is padded to 8 * ceil(gy / 8) * gx * gz workgroups, the surplus leaves at once (false).
__device__ __forceinline__ bool rc_blk_of(int local, int gx, int gy, int gz, int xcd, RcBlk& k) {
  if (!xcd) {
    if (local >= gx * gy * gz) return false;
    k = RcBlk{local % gx, (local / gx) % gy, local / (gx * gy), gx, gy};
    return true;
  }
  const int x = local & 7, q = local >> 3, per = gx * gz;
  const int ny = (gy - x + 7) >> 3;            // row tiles x, x + 8, ... of this XCD
  if (q >= ny * per) return false;
  const int i = q / per, rem = q - i * per;
  k = RcBlk{rem % gx, x + 8 * i, rem / gx, gx, gy};
  return true;
}

// pixel geometry of a convolution: (B, H, W) input pixels -> (B, Ho, Wo) output pixels, KW x (T / KW) taps, stride S,
// padding P.  T == 1 && S == 1: rows map to themselves.
struct RcGeom { int H, W, Ho, Wo, KW, S, P; };

// wave / lane roles over the 64 x 64 tile: wave (wm, wn) owns rows [32 wm, +32) x columns [32 wn, +32); accumulator
// register r of lane (li, lh) is element (row_of(r), col())
struct RcWave {
  int wm, wn, li, lh;
  __device__ __forceinline__ void init(int tid) {
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    wm = wave >> 1;
    wn = wave & 1;
    li = lane & 31;
    lh = lane >> 5;
  }
  __device__ __forceinline__ int row_of(int r) const { return wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh; }
  __device__ __forceinline__ int col() const { return wn * 32 + li; }
  // Half of a stage's fragments are read before its first MFMA, the other half under the first eight (hipcc otherwise sinks
  // every ds_read next to its consumer: {2 ds_read2_b32, s_waitcnt lgkmcnt(0), 2 MFMAs} x 8)
  __device__ __forceinline__ void mma(const float* __restrict__ As, const float* __restrict__ Bs, f32x16& acc) const {
    const float* a = As + lh * RC_AP + wm * 32 + li;
    const float* b = Bs + lh * RC_AP + wn * 32 + li;
    float av[2][RC_BK / 4], bv[2][RC_BK / 4];
#pragma unroll
    for (int i = 0; i < RC_BK / 4; ++i) {
      av[0][i] = a[2 * i * RC_AP];
      bv[0][i] = b[2 * i * RC_AP];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < RC_BK / 4; ++i) {
      av[1][i] = a[(RC_BK / 2 + 2 * i) * RC_AP];
      bv[1][i] = b[(RC_BK / 2 + 2 * i) * RC_AP];
    }
#pragma unroll
    for (int i = 0; i < RC_BK / 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0][i], bv[0][i], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < RC_BK / 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1][i], bv[1][i], acc, 0, 0, 0);
  }
};

// Staging slots of a thread for a 64 x 32 operand tile: k-contiguous sources (KMAJOR) are walked with consecutive
// threads along k (slot i = row rl + 8 i), row-contiguous ones along the rows (slot i = k kl + 4 i); 8 slots each.
template <bool KMAJOR>
struct RcStg {
  int rl, kl;
  __device__ __forceinline__ void init(int tid) {
    if (KMAJOR) { kl = tid % RC_BK; rl = tid / RC_BK; } else { rl = tid & 63; kl = tid >> 6; }
  }
  __device__ __forceinline__ int row(int i) const { return KMAJOR ? rl + i * RC_RSTEP : rl; }
  __device__ __forceinline__ int kk(int i) const { return KMAJOR ? kl : kl + i * 4; }
  __device__ __forceinline__ void store(float* __restrict__ S, const float (&v)[RC_NS]) const {
    float* d = S + kl * RC_AP + rl;
#pragma unroll
    for (int i = 0; i < RC_NS; ++i) d[KMAJOR ? i * RC_RSTEP : i * 4 * RC_AP] = v[i];
  }
};

// column sums over the tile's 64 rows of per-lane partial sums (a lane: its 16 rows of column col()); cs: 128 floats
__device__ __forceinline__ float rc_colsum(float* __restrict__ cs, float v, const RcWave& w) {
  v += __shfl_xor(v, 32, 64);
  __syncthreads();
  if (w.lh == 0) cs[w.wm * 64 + w.col()] = v;
  __syncthreads();
  return cs[w.col()] + cs[64 + w.col()];
}

// The complete accumulators of this workgroup's tile.  nz == 1: its own.  nz > 1 (reduction split over blockIdx.z):
// every split leaves its accumulators in ws [z][...] at (row, col) -> base + row * rstride + col; the last one to arrive
// sums them in z order (16 x 4 loads in flight) and carries on alone -- false for the others.
__device__ __forceinline__ bool rc_acc_reduce(f32x16& acc, float* __restrict__ ws, unsigned* __restrict__ ticket, int nz,
                                              int zi, size_t zstride, size_t base, size_t rstride, int cnt,
                                              const RcWave& w, int* last) {
  if (nz == 1) return true;
  float* mine = ws + (size_t)zi * zstride + base + w.col();
#pragma unroll
  for (int r = 0; r < 16; ++r)
    if (w.row_of(r) < cnt) rc_st(mine + (size_t)w.row_of(r) * rstride, acc[r]);
  if (!rc_last_workgroup(ticket, (unsigned)nz, last)) return false;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const float* all = ws + base + w.col();
  for (int z0 = 0; z0 < nz; z0 += 4) {
    float t[16][4];
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        t[r][i] = rc_ld(all + (size_t)min(z0 + i, nz - 1) * zstride + (size_t)min(w.row_of(r), cnt - 1) * rstride);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (z0 + i < nz) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += t[r][i];
      }
  }
  return true;
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
  float* part;            // (row tiles [+ groups], C, 2)
  unsigned* counter;      // (column tiles) final tickets, then (column tiles, groups)
  float eps, momentum;
  int eval;
};
struct RcFwdArgs {
  const float* x;         // (B H W, Cin)
  const float* w;         // (Cout, T, Cin)
  const float* xmean;     // prologue of RC_PRE_BN_RELU: relu(fma(x - xmean, xsc, xbeta))
  const float* xsc;
  const float* xbeta;
  float* y;               // (M = B Ho Wo, Cout)
  float* ws;              // nz > 1: (nz, M, Cout) partial outputs
  unsigned* tile_ticket;  // nz > 1: one per output tile
  int M, Cin, Cout, T, pre, nz;
  RcGeom g;
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


// store of a finished tile + the statistics of the BatchNorm behind it (partials, election, merge, finalize)
__device__ __forceinline__ void rc_fwd_tail(const f32x16& acc, float* __restrict__ y, int M, int Cout, const RcBnFwd& bn,
                                            const RcBlk k, int m0, int n0, int cnt, const RcWave& wv,
                                            float* __restrict__ smem, float* __restrict__ cs, int* __restrict__ lastp) {
  const int tid = threadIdx.x;
  const int n = n0 + wv.col();
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r)
    if (wv.row_of(r) < cnt) {
      y[(size_t)(m0 + wv.row_of(r)) * Cout + n] = acc[r];
      s += acc[r];
    }
  if (!bn.part) return;
  const float mean_t = rc_colsum(cs, s, wv) / (float)cnt;
  float q = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r)
    if (wv.row_of(r) < cnt) q = fmaf(acc[r] - mean_t, acc[r] - mean_t, q);
  const float m2_t = rc_colsum(cs, q, wv);
  if (wv.wm == 0 && wv.lh == 0) {
    rc_st(bn.part + ((size_t)k.by * Cout + n) * 2, mean_t);
    rc_st(bn.part + ((size_t)k.by * Cout + n) * 2 + 1, m2_t);
  }
  // election (two levels beyond RC_GROUP row tiles) and merge: thread = (column, one of 4 part groups)
  const int col = tid & 63, rg = tid >> 6, nc = n0 + col;
  const RcLevels lv(k.gy, k.by);
  double* dl = reinterpret_cast<double*>(smem);
  double o0, o1, o2;
  const float* src_part = bn.part;
  int np = lv.nrow, rows_per = RC_BM;
  if (lv.ngrp) {
    if (!rc_last_workgroup(bn.counter + k.gx + k.bx * lv.ngrp + lv.grp, lv.g1 - lv.g0, lastp)) return;
    rc_merge<64, true>(bn.part, Cout, nc, lv.g0, lv.g1, RC_BM, M, dl, col, rg, o0, o1, o2);
    float* l1 = lv.level1(bn.part, Cout);
    if (rg == 0) {
      rc_st(l1 + ((size_t)lv.grp * Cout + nc) * 2, (float)o1);
      rc_st(l1 + ((size_t)lv.grp * Cout + nc) * 2 + 1, (float)o2);
    }
    src_part = l1;
    np = lv.ngrp;
    rows_per = RC_BM * RC_GROUP;
  }
  if (!rc_last_workgroup(bn.counter + k.bx, np, lastp)) return;
  rc_merge<64, true>(src_part, Cout, nc, 0, np, rows_per, M, dl, col, rg, o0, o1, o2);
  if (rg == 0) rc_bn_fwd_finalize<64>(bn, o1, o2, M, nc);
}

__device__ __forceinline__ void rc_fwd_body(const RcFwdArgs& a, const RcBlk k, float* __restrict__ smem, float* __restrict__ cs,
                                            int* __restrict__ lastp) {
  const int tid = threadIdx.x;
  const int n0 = k.bx * 64, m0 = k.by * RC_BM;
  RcWave wv;
  wv.init(tid);
  // Round 4: both operands of a forward stage are k-contiguous (x rows along Cin, channels-last weight rows along Cin), so
  // a thread stages FOUR consecutive k of TWO rows with one 16-byte load each (it was one float of eight rows: 16 dword
  // loads with their 64-bit address arithmetic per stage, and the per-tap geometry of eight rows -- VALU work that ADDS to
  // the fp32 MFMA time on a SIMD).  LDS image unchanged ([k][row], pitch 65): the four k go to four rows of it, the 32
  // lanes of a store hit banks (4 k4 + row) mod 32, all distinct.
  static_assert(RC_BK == 32, "float4 staging: 8 k-quads x 32 rows per pass");
  constexpr int NR = 2;                               // rows per thread: r0 and r0 + 32
  const int k4 = (tid & 7) * 4, r0 = tid >> 3;
  const bool ident = a.T == 1 && a.g.S == 1;
  int pix[NR], hw[NR], src[NR], wrow[NR];
  // Round 5: TWO register sets -- the loads of stage s + 2 are issued while stage s is multiplied and stage s + 1 waits in
  // the other set (stamps at batch 24: 1.35 us per 32-deep stage against 0.43 us of MFMA with one stage in flight; a
  // launch there is 6-9 stages per workgroup).  The loop is unrolled by two so that the set of every access is static.
  float4 ra[2][NR], rb[2][NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int r = m0 + r0 + 32 * i;
    wrow[i] = (n0 + r0 + 32 * i) * a.T;
    if (ident) {
      pix[i] = r < a.M ? r : -1;
      hw[i] = 0;
    } else {
      const int hwo = a.g.Ho * a.g.Wo;
      const int b = r / hwo, rem = r - b * hwo, oh = rem / a.g.Wo, ow = rem - oh * a.g.Wo;
      const int h0 = oh * a.g.S - a.g.P, w0 = ow * a.g.S - a.g.P;
      pix[i] = (b * a.g.H + h0) * a.g.W + w0;
      hw[i] = r < a.M ? ((h0 + 0x4000) << 16 | (w0 + 0x4000)) : -1;
    }
  }
  float4 pm[2], ps[2], pb[2];
  pm[0] = pm[1] = pb[0] = pb[1] = make_float4(0.f, 0.f, 0.f, 0.f);
  ps[0] = ps[1] = make_float4(1.f, 1.f, 1.f, 1.f);
  unsigned oka[2] = {0u, 0u};
  // this workgroup's share of the T * Cin / 32 stages (blockIdx.z of nz)
  const int spc = a.Cin / RC_BK, nstage_all = a.T * spc, sper = (nstage_all + a.nz - 1) / a.nz;
  const int s_beg = k.bz * sper, s_end = min(nstage_all, s_beg + sper);
  int ltap = s_beg / spc, lc0 = (s_beg - ltap * spc) * RC_BK;
  bool newtap = true;
  auto load = [&](const int st) {
    if (newtap) {
      const int kh = ltap / a.g.KW, kw = ltap - kh * a.g.KW;
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        if (ident) {
          src[i] = pix[i];
        } else {
          const int ih = (hw[i] >> 16) - 0x4000 + kh, iw = (hw[i] & 0xffff) - 0x4000 + kw;
          const bool ok = hw[i] != -1 && (unsigned)ih < (unsigned)a.g.H && (unsigned)iw < (unsigned)a.g.W;
          src[i] = ok ? pix[i] + kh * a.g.W + kw : -1;
        }
      }
      newtap = false;
    }
    const int c = lc0 + k4;
    if (a.pre == RC_PRE_BN_RELU) {
      pm[st] = *reinterpret_cast<const float4*>(a.xmean + c);
      ps[st] = *reinterpret_cast<const float4*>(a.xsc + c);
      pb[st] = *reinterpret_cast<const float4*>(a.xbeta + c);
    }
    oka[st] = 0;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const bool ok = src[i] >= 0;
      oka[st] |= (ok ? 1u : 0u) << i;
      ra[st][i] = *reinterpret_cast<const float4*>(a.x + (ok ? (size_t)src[i] * a.Cin + c : 0));
    }
#pragma unroll
    for (int i = 0; i < NR; ++i)
      if (RC_B_STAGE(st)) rb[st][i] = *reinterpret_cast<const float4*>(a.w + (size_t)(wrow[i] + ltap) * a.Cin + c);
    lc0 += RC_BK;
    if (lc0 >= a.Cin) { lc0 = 0; ++ltap; newtap = true; }
  };
  const float lo = a.pre == RC_PRE_NONE ? -INFINITY : 0.f;
  auto store = [&](const int st, float* __restrict__ As_, float* __restrict__ Bs_) {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      float v[4] = {ra[st][i].x, ra[st][i].y, ra[st][i].z, ra[st][i].w};
      const float m4[4] = {pm[st].x, pm[st].y, pm[st].z, pm[st].w}, s4[4] = {ps[st].x, ps[st].y, ps[st].z, ps[st].w};
      const float b4[4] = {pb[st].x, pb[st].y, pb[st].z, pb[st].w};
      const bool ok = oka[st] >> i & 1u;
      float* da = As_ + k4 * RC_AP + r0 + 32 * i;
      float* db = Bs_ + k4 * RC_AP + r0 + 32 * i;
      const float w4[4] = {rb[st][i].x, rb[st][i].y, rb[st][i].z, rb[st][i].w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // one branch-free form for the three prologues: (mean, sc, beta) = (0, 1, 0) outside RC_PRE_BN_RELU -- fma(t - 0, 1,
        // 0) is t itself -- and the clamp floor -inf for RC_PRE_NONE (the per-element mode tests were 30 branches a stage)
        // (select, not v_max: a NaN input stays a NaN in every mode, as torch.relu / the plain pass-through keep it)
        const float u = rc_bn(v[q], m4[q], s4[q], b4[q]);
        const float t = u < lo ? lo : u;
        da[q * RC_AP] = ok ? t : 0.f;
        if (RC_B_STAGE(st)) db[q * RC_AP] = w4[q];
      }
    }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // two LDS stage buffers (the next stage is stored while the other waves still read the current one: one barrier per
  // stage) fed from the two register sets
  float* const b0 = smem;
  float* const b1 = smem + 2 * RC_BK * RC_AP;
  if (s_beg < s_end) load(0);
  if (s_beg + 1 < s_end) load(1);
  if (s_beg < s_end) store(0, b0, b0 + RC_BK * RC_AP);
  __syncthreads();
#pragma unroll 1
  for (int s = s_beg; s < s_end; s += 2) {
    if (s + 2 < s_end) load(0);
    wv.mma(b0, b0 + RC_BK * RC_AP, acc);
    if (s + 1 < s_end) store(1, b1, b1 + RC_BK * RC_AP);
    __syncthreads();
    if (s + 1 >= s_end) break;
    if (s + 3 < s_end) load(1);
    wv.mma(b1, b1 + RC_BK * RC_AP, acc);
    if (s + 2 < s_end) store(0, b0, b0 + RC_BK * RC_AP);
    __syncthreads();
  }
  const int cnt = min(RC_BM, a.M - m0);
  if (!rc_acc_reduce(acc, a.ws, a.tile_ticket + k.by * k.gx + k.bx, a.nz, k.bz, (size_t)a.M * a.Cout,
                     (size_t)m0 * a.Cout + n0, a.Cout, cnt, wv, lastp))
    return;
  rc_fwd_tail(acc, a.y, a.M, a.Cout, a.bn, k, m0, n0, cnt, wv, smem, cs, lastp);
}

__global__ __launch_bounds__(256, 3) void rc_fwd_kernel(RcFwdArgs a, int gx, int gy, int gz, int xcd) {
  __shared__ __attribute__((aligned(16))) float smem[4 * RC_BK * RC_AP];
  __shared__ float cs[128];
  __shared__ int last;
  RcBlk k;
  if (!rc_blk_of((int)blockIdx.x, gx, gy, gz, xcd, k)) return;
  rc_fwd_body(a, k, smem, cs, &last);
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
  float* part;            // (row tiles [+ groups], C, 2)
  unsigned* counter;      // as RcBnFwd
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
  const float* add;       // added before the mask (the shortcut's gradient) or NULL: (Min, Cin), or ...
  const int* add_tbl;     // ... with add_tbl (Min): row add_tbl[m] of `add` (-1: nothing), a strided projection's gradient
  const float* mY;        // mask source (Min, Cin): RC_MASK_RAW mY > 0, RC_MASK_BN bn(mY) > 0
  const float* mmean;
  const float* msc;
  const float* mbeta;
  float* out;             // (Min, Cin)
  float* ws;              // nz > 1: (nz, Min, Cin) partial outputs
  unsigned* tile_ticket;
  const int* row_map;     // stride 2, even H and W: the input pixels ordered by parity class (4 x Min / 4 rows), see below
  int M, Min, Cin, Cout, T, mask, nstat, nz, cls_tiles;
  RcGeom g;
  RcStat st[2];
};

__device__ __forceinline__ void rc_dgrad_body(const RcDgradArgs& a, const RcBlk k, float* __restrict__ smem,
                                              float* __restrict__ cs, int* __restrict__ lastp) {
  const int tid = threadIdx.x;
  // Stride 2 (row_map): an input pixel (ih, iw) is read through tap (kh, kw) only when ih + P - kh and iw + P - kw are
  // even, i.e. through 1, 2, 2 or 4 of a 3 x 3 filter's taps depending on the parities of (ih, iw).  Walking the pixels in
  // raster order makes every 64-row tile run all 9 taps with three quarters of its (row, tap) pairs zero; so the rows are
  // walked class by class (row tile -> class k.by / cls_tiles, rows row_map[class * Min / 4 + ...]) and a tile only runs
  // its class's taps: 9 / 4 instead of 9 tap passes per pixel.
  const bool cls = a.row_map != nullptr;
  const int pcl = cls ? k.by / a.cls_tiles : 0, rows_c = cls ? a.Min / 4 : a.Min;
  const int n0 = k.bx * 64, m0 = (cls ? k.by - pcl * a.cls_tiles : k.by) * RC_BM;     // n0: input-channel tile
  const int* rmap = cls ? a.row_map + (size_t)pcl * rows_c : nullptr;
  const int kh0 = cls ? ((pcl >> 1) + a.g.P) & 1 : 0, kw0 = cls ? ((pcl & 1) + a.g.P) & 1 : 0;
  const int KH = a.T / a.g.KW, nkw = cls ? (a.g.KW - kw0 + 1) / 2 : a.g.KW, ntap = cls ? ((KH - kh0 + 1) / 2) * nkw : a.T;
  RC_STAMP(k, 0);
  RcWave wv;
  wv.init(tid);
  // float4 staging as in rc_fwd_body: A = (G, Y) rows, four consecutive Cout channels of two pixels per thread; B = the
  // weights' [k][n] view (n = Cin contiguous), four consecutive n of two k per thread
  static_assert(RC_BK == 32, "float4 staging: 8 k-quads x 32 rows per pass");
  constexpr int NR = 2;
  const int k4 = (tid & 7) * 4, r0 = tid >> 3;            // A: k quad, row (and row + 32)
  const int nb4 = (tid & 15) * 4, kb = tid >> 4;          // B: n quad, k (and k + 16)
  // this thread's 2 input pixels: bh = (b, ih + P, iw + P) packed; the output pixel that reads pixel (ih, iw) through
  // tap (kh, kw) is ((ih + P - kh) / S, (iw + P - kw) / S) when both divide and lie inside
  const bool ident = a.T == 1 && a.g.S == 1;
  int pb_[NR], hw[NR], src[NR];
  float4 rg_[2][NR], ry[2][NR], rb[2][NR];        // two register sets: loads run two stages ahead (see rc_fwd_body)
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int rr = m0 + r0 + 32 * i;
    const int r = cls ? rmap[min(rr, rows_c - 1)] : rr;
    if (ident) {
      pb_[i] = r < a.Min ? r : -1;
      hw[i] = 0;
    } else {
      const int hwi = a.g.H * a.g.W;
      const int b = r / hwi, rem = r - b * hwi, ih = rem / a.g.W, iw = rem - ih * a.g.W;
      pb_[i] = b * a.g.Ho;
      hw[i] = rr < rows_c ? ((ih + a.g.P) << 16 | (iw + a.g.P)) : -1;
    }
  }
  float4 pp[2], pq[2], pr[2];
  pp[0] = pp[1] = make_float4(1.f, 1.f, 1.f, 1.f);
  pq[0] = pq[1] = pr[0] = pr[1] = make_float4(0.f, 0.f, 0.f, 0.f);
  unsigned oka[2] = {0u, 0u};
  int wtap = 0;           // the filter tap (kh KW + kw) of the stage being loaded
  const int spc = a.Cout / RC_BK, nstage_all = ntap * spc, sper = (nstage_all + a.nz - 1) / a.nz;
  const int s_beg = k.bz * sper, s_end = min(nstage_all, s_beg + sper);
  int ltap = s_beg / spc, lk0 = (s_beg - ltap * spc) * RC_BK;
  bool newtap = true;
  auto load = [&](const int st) {
    if (newtap) {
      const int ta = ltap / nkw, tb = ltap - ta * nkw;
      const int kh = cls ? kh0 + 2 * ta : ta, kw = cls ? kw0 + 2 * tb : tb;
      wtap = kh * a.g.KW + kw;
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        if (ident) {
          src[i] = pb_[i];
        } else {
          const int th = (hw[i] >> 16) - kh, tw = (hw[i] & 0xffff) - kw;
          int oh = th, ow = tw;
          bool ok = hw[i] != -1 && th >= 0 && tw >= 0;
          if (a.g.S == 2) {
            ok = ok && ((th | tw) & 1) == 0;
            oh = th >> 1;
            ow = tw >> 1;
          } else if (a.g.S != 1) {
            ok = ok && th % a.g.S == 0 && tw % a.g.S == 0;
            oh = th / a.g.S;
            ow = tw / a.g.S;
          }
          ok = ok && oh < a.g.Ho && ow < a.g.Wo;
          src[i] = ok ? (pb_[i] + oh) * a.g.Wo + ow : -1;
        }
      }
      newtap = false;
    }
    const int kc = lk0 + k4;
    if (a.pqr) {
      pp[st] = *reinterpret_cast<const float4*>(a.pqr + kc);
      pq[st] = *reinterpret_cast<const float4*>(a.pqr + a.Cout + kc);
      pr[st] = *reinterpret_cast<const float4*>(a.pqr + 2 * a.Cout + kc);
    }
    oka[st] = 0;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const bool ok = src[i] >= 0;
      oka[st] |= (ok ? 1u : 0u) << i;
      const size_t o = ok ? (size_t)src[i] * a.Cout + kc : 0;
      rg_[st][i] = *reinterpret_cast<const float4*>(a.G + o);
      ry[st][i] = a.pqr ? *reinterpret_cast<const float4*>(a.Y + o) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < NR; ++i)
      if (RC_B_STAGE(st))
        rb[st][i] = *reinterpret_cast<const float4*>(a.w + ((size_t)(lk0 + kb + 16 * i) * a.T + wtap) * a.Cin + n0 + nb4);
    lk0 += RC_BK;
    if (lk0 >= a.Cout) { lk0 = 0; ++ltap; newtap = true; }
  };
  auto store = [&](const int st, float* __restrict__ As_, float* __restrict__ Bs_) {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const float g4[4] = {rg_[st][i].x, rg_[st][i].y, rg_[st][i].z, rg_[st][i].w};
      const float y4[4] = {ry[st][i].x, ry[st][i].y, ry[st][i].z, ry[st][i].w};
      const float p4[4] = {pp[st].x, pp[st].y, pp[st].z, pp[st].w}, q4[4] = {pq[st].x, pq[st].y, pq[st].z, pq[st].w};
      const float r4[4] = {pr[st].x, pr[st].y, pr[st].z, pr[st].w};
      const bool ok = oka[st] >> i & 1u;
      float* da = As_ + k4 * RC_AP + r0 + 32 * i;
#pragma unroll
      for (int q = 0; q < 4; ++q) {      // (p, q, r) = (1, 0, 0) and Y = 0 without a BatchNorm behind: the same fma chain
        const float v = fmaf(g4[q], p4[q], fmaf(y4[q], q4[q], r4[q]));
        da[q * RC_AP] = ok ? v : 0.f;
      }
      float* db = Bs_ + (kb + 16 * i) * RC_AP + nb4;
      if (RC_B_STAGE(st)) { db[0] = rb[st][i].x; db[1] = rb[st][i].y; db[2] = rb[st][i].z; db[3] = rb[st][i].w; }
    }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float* const b0 = smem;
  float* const b1 = smem + 2 * RC_BK * RC_AP;
  if (s_beg < s_end) load(0);
  if (s_beg + 1 < s_end) load(1);
  if (s_beg < s_end) {
    RC_STAMP_WAIT(k, 1);
    store(0, b0, b0 + RC_BK * RC_AP);
  }
  __syncthreads();
#pragma unroll 1
  for (int s = s_beg; s < s_end; s += 2) {
    if (s + 2 < s_end) load(0);
    wv.mma(b0, b0 + RC_BK * RC_AP, acc);
    if (s + 1 < s_end) store(1, b1, b1 + RC_BK * RC_AP);
    __syncthreads();
    if (s + 1 >= s_end) break;
    if (s + 3 < s_end) load(1);
    wv.mma(b1, b1 + RC_BK * RC_AP, acc);
    if (s + 2 < s_end) store(0, b0, b0 + RC_BK * RC_AP);
    __syncthreads();
  }
  RC_STAMP(k, 2);
  const int cnt = min(RC_BM, rows_c - m0);
  // (workspace rows by row tile, not by pixel: in class order the tiles of different classes share local row numbers)
  if (!rc_acc_reduce(acc, a.ws, a.tile_ticket + k.by * k.gx + k.bx, a.nz, k.bz, (size_t)k.gy * RC_BM * a.Cin,
                     (size_t)k.by * RC_BM * a.Cin + n0, a.Cin, cnt, wv, lastp))
    return;
  RC_STAMP(k, 3);
  const int c = n0 + wv.col();
  float mm = 0.f, ms = 1.f, mb = 0.f;
  if (a.mask == RC_MASK_BN) { mm = a.mmean[c]; ms = a.msc[c]; mb = a.mbeta[c]; }
  float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f}, tm[2] = {0.f, 0.f}, tr[2] = {0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 2; ++t)
    if (t < a.nstat) { tm[t] = a.st[t].mean[c]; tr[t] = a.st[t].rstd[c]; }
  // the epilogue in two halves of 8 accumulator rows: the operands of a half first (all in flight), then its arithmetic.
  // st[0].Y is the mask source itself at the BatchNorm-mask call sites (one load serves both).  Every mode test is hoisted
  // out of the per-row loops (round 5: the workgroup-uniform tests inside them were ~100 branches per tile) and the
  // arithmetic is one form for all modes: no mask = a mask source of 1, a raw mask = the BatchNorm form with (0, 1, 0).
  const bool st0_is_mask = a.nstat > 0 && a.mask != RC_MASK_NONE && a.st[0].Y == a.mY;
  const float* const y0p = a.nstat > 0 ? a.st[0].Y : nullptr;
  const float* const y1p = a.nstat > 1 ? a.st[1].Y : nullptr;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    size_t off[8];          // element offsets of this half's accumulator registers in the (rows, Cin) tensors
    int arow[8];
    float vadd[8], vmy[8], vy0[8], vy1[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int rr = m0 + min(wv.row_of(8 * h + q), cnt - 1);
      arow[q] = cls ? rmap[rr] : rr;
      off[q] = (size_t)arow[q] * a.Cin + c;
      vadd[q] = 0.f;
      vmy[q] = 1.f;
      vy0[q] = vy1[q] = 0.f;
    }
    if (a.add && a.add_tbl) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int rr = a.add_tbl[arow[q]];
        const float t = a.add[rr >= 0 ? (size_t)rr * a.Cin + c : 0];
        vadd[q] = rr >= 0 ? t : 0.f;
      }
    } else if (a.add) {
#pragma unroll
      for (int q = 0; q < 8; ++q) vadd[q] = a.add[off[q]];
    }
    if (a.mask != RC_MASK_NONE) {
#pragma unroll
      for (int q = 0; q < 8; ++q) vmy[q] = a.mY[off[q]];
    }
    if (y0p && !st0_is_mask) {
#pragma unroll
      for (int q = 0; q < 8; ++q) vy0[q] = y0p[off[q]];
    }
    if (y1p) {
#pragma unroll
      for (int q = 0; q < 8; ++q) vy1[q] = y1p[off[q]];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int r = 8 * h + q;
      const bool live = wv.row_of(r) < cnt;
      const float z = rc_bn(vmy[q], mm, ms, mb);              // raw mask / no mask: (mm, ms, mb) = (0, 1, 0)
      const float g = (live && z > 0.f) ? acc[r] + vadd[q] : 0.f;
      if (live) a.out[off[q]] = g;
      const float y0 = st0_is_mask ? vmy[q] : vy0[q];
      s1[0] += g;                                             // (unused without statistics; rows past the tile add 0)
      s2[0] = fmaf(g, (y0 - tm[0]) * tr[0], s2[0]);
      s2[1] = fmaf(g, (vy1[q] - tm[1]) * tr[1], s2[1]);
    }
  }
  s1[1] = s1[0];
  RC_STAMP_WAIT(k, 4);
  if (a.nstat == 0) return;
#pragma unroll
  for (int t = 0; t < 2; ++t)
    if (t < a.nstat) {
      const float a1 = rc_colsum(cs, s1[t], wv), a2 = rc_colsum(cs, s2[t], wv);
      if (wv.wm == 0 && wv.lh == 0) {
        rc_st(a.st[t].part + ((size_t)k.by * a.Cin + c) * 2, a1);
        rc_st(a.st[t].part + ((size_t)k.by * a.Cin + c) * 2 + 1, a2);
      }
    }
  RC_STAMP_WAIT(k, 5);
  rc_stat_tail<64>(a.st, a.nstat, k.gy, k.by, k.bx, k.gx, a.Min, a.Cin, n0 + (tid & 63), reinterpret_cast<double*>(smem),
                   tid & 63, tid >> 6, lastp);
  RC_STAMP_WAIT(k, 6);
}

__global__ __launch_bounds__(256, 3) void rc_dgrad_kernel(RcDgradArgs a, int gx, int gy, int gz, int xcd) {
  __shared__ __attribute__((aligned(16))) float smem[4 * RC_BK * RC_AP];
  __shared__ float cs[128];
  __shared__ int last;
  RcBlk k;
  if (!rc_blk_of((int)blockIdx.x, gx, gy, gz, xcd, k)) return;
  rc_dgrad_body(a, k, smem, cs, &last);
}

// stand-alone statistics of a BatchNorm backward whose G was produced elsewhere (pooling backward, tests):
// grid (C / 64, row tiles of 64)
// With dyp != NULL the gradient is made here as well: the backward of AdaptiveAvgPool2d(1) on relu(xs),
// G[(b, hw), c] = dyp[b, c] / HW * (xs > 0), stored to Gout (the stack's last BatchNorm sits right under the pooling).
// With idx != NULL: the backward of the stem's MaxPool2d(3, 2, 1) on relu(bn(Y)): G[(b,h,w), c] = (bn(Y) > 0) * sum of the
// dyp entries of the <= 4 windows around (h, w) whose recorded maximum is this pixel; (msc, mbeta) with st.mean normalise.
__global__ __launch_bounds__(256) void rc_stat_kernel(const float* __restrict__ G, RcStat st, int M, int C,
                                                      const float* __restrict__ dyp, const float* __restrict__ xs,
                                                      float* __restrict__ Gout, int HW, const int* __restrict__ idx,
                                                      const float* __restrict__ msc, const float* __restrict__ mbeta,
                                                      int H, int W, int Ho, int Wo) {
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
    float g;
    if (idx) {
      const int r = m0 + row, b = r / (H * W), me = r - b * (H * W), h = me / W, w = me - h * W;
      g = 0.f;
      for (int oh = h / 2; oh <= (h + 1) / 2; ++oh) {
        if (oh >= Ho) continue;
        for (int ow = w / 2; ow <= (w + 1) / 2; ++ow) {
          if (ow >= Wo) continue;
          const size_t oo = (((size_t)b * Ho + oh) * Wo + ow) * C + c;
          if (idx[oo] == me) g += dyp[oo];
        }
      }
      if (!(rc_bn(st.Y[o], tm, msc[c], mbeta[c]) > 0.f)) g = 0.f;
      Gout[o] = g;
    } else if (dyp) {
      g = xs[o] > 0.f ? dyp[(size_t)((m0 + row) / HW) * C + c] * (1.0f / (float)HW) : 0.f;
      Gout[o] = g;
    } else {
      g = G[o];
    }
    s1 += g;
    s2 = fmaf(g, (st.Y[o] - tm) * tr, s2);
  }
  __syncthreads();
  cs[rg * 64 + col] = s1;
  __syncthreads();
  const float a1 = cs[col] + cs[64 + col] + cs[128 + col] + cs[192 + col];
  __syncthreads();
  cs[rg * 64 + col] = s2;
  __syncthreads();
  const float a2 = cs[col] + cs[64 + col] + cs[128 + col] + cs[192 + col];
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
  const float* x;         // (B H W, Cin) raw input, consumed through `pre` as in the forward pass
  const float* xmean;
  const float* xsc;
  const float* xbeta;
  const int* tbl;         // (T, M) source row of (tap, output row), -1 = padding; NULL: identity
  float* dw;              // (Cout, T, Cin)
  float* ws;              // nz > 1: (nz, Cout T Cin) partial tiles
  unsigned* counter;      // nz > 1: one ticket per output tile
  int M, Cin, Cout, T, pre, acc, nz, kper;
};

__device__ __forceinline__ void rc_wgrad_body(const RcWgradArgs& a, const RcBlk k, float* __restrict__ smem,
                                              float* __restrict__ cs, int* __restrict__ lastp) {
  const int tid = threadIdx.x;
  const int c0 = k.bx * 64, n0 = k.by * 64;
  const int tap = k.bz / a.nz, zi = k.bz - tap * a.nz;
  const int kbeg = zi * a.kper, kend = min(a.M, kbeg + a.kper);
  RcWave wv;
  wv.init(tid);
  // float4 staging (as rc_fwd_body): both operands are row-contiguous here -- G / Y along Cout, x along Cin -- so a thread
  // stages four consecutive columns of two reduction rows (pixels kb and kb + 16 of the stage) per operand
  static_assert(RC_BK == 32, "float4 staging: 16 column quads x 16 rows per pass");
  constexpr int NR = 2;
  const int q4 = (tid & 15) * 4, kb = tid >> 4;
  float4 rg_[2][NR], ry[2][NR], rb[2][NR];    // two register sets: loads run two stages ahead (see rc_fwd_body)
  int sr[NR];                 // source rows of the stage being loaded next (one stage ahead of the data)
  unsigned oka[2] = {0u, 0u}, okb[2] = {0u, 0u};
  const int n = n0 + q4, c = c0 + q4;
  float4 pp = make_float4(1.f, 1.f, 1.f, 1.f), pq = make_float4(0.f, 0.f, 0.f, 0.f), pr = pq, pm = pq, ps = pp, pb = pq;
  if (a.pqr) {
    pp = *reinterpret_cast<const float4*>(a.pqr + n);
    pq = *reinterpret_cast<const float4*>(a.pqr + a.Cout + n);
    pr = *reinterpret_cast<const float4*>(a.pqr + 2 * a.Cout + n);
  }
  if (a.pre == RC_PRE_BN_RELU) {
    pm = *reinterpret_cast<const float4*>(a.xmean + c);
    ps = *reinterpret_cast<const float4*>(a.xsc + c);
    pb = *reinterpret_cast<const float4*>(a.xbeta + c);
  }
  const int* tb = a.tbl ? a.tbl + (size_t)tap * a.M : nullptr;
  const float lo = a.pre == RC_PRE_NONE ? -INFINITY : 0.f;
  auto rows = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int m = k0 + kb + 16 * i;
      sr[i] = m < kend ? (tb ? tb[m] : m) : -1;
    }
  };
  auto load = [&](const int st, int k0) {
    oka[st] = 0;
    okb[st] = 0;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int m = k0 + kb + 16 * i;
      const bool ok = m < kend;
      oka[st] |= (ok ? 1u : 0u) << i;
      const size_t o = ok ? (size_t)m * a.Cout + n : 0;
      rg_[st][i] = *reinterpret_cast<const float4*>(a.G + o);
      ry[st][i] = a.pqr ? *reinterpret_cast<const float4*>(a.Y + o) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const bool ok = sr[i] >= 0;
      okb[st] |= (ok ? 1u : 0u) << i;
      rb[st][i] = *reinterpret_cast<const float4*>(a.x + (ok ? (size_t)sr[i] * a.Cin + c : 0));
    }
    if (k0 + RC_BK < kend) rows(k0 + RC_BK);      // the table entries of the stage after: not a dependent round trip then
  };
  auto store = [&](const int st, float* __restrict__ As_, float* __restrict__ Bs_) {
    const float p4[4] = {pp.x, pp.y, pp.z, pp.w}, qq4[4] = {pq.x, pq.y, pq.z, pq.w}, r4[4] = {pr.x, pr.y, pr.z, pr.w};
    const float m4[4] = {pm.x, pm.y, pm.z, pm.w}, s4[4] = {ps.x, ps.y, ps.z, ps.w}, b4[4] = {pb.x, pb.y, pb.z, pb.w};
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const float g4[4] = {rg_[st][i].x, rg_[st][i].y, rg_[st][i].z, rg_[st][i].w};
      const float y4[4] = {ry[st][i].x, ry[st][i].y, ry[st][i].z, ry[st][i].w};
      const float x4[4] = {rb[st][i].x, rb[st][i].y, rb[st][i].z, rb[st][i].w};
      const bool oa = oka[st] >> i & 1u, ob = okb[st] >> i & 1u;
      float* da = As_ + (kb + 16 * i) * RC_AP + q4;
      float* db = Bs_ + (kb + 16 * i) * RC_AP + q4;
#pragma unroll
      for (int q = 0; q < 4; ++q) {      // branch-free forms of both prologues (see rc_fwd_body)
        const float v = fmaf(g4[q], p4[q], fmaf(y4[q], qq4[q], r4[q]));
        da[q] = oa ? v : 0.f;
        const float u = rc_bn(x4[q], m4[q], s4[q], b4[q]);
        const float t = u < lo ? lo : u;      // (NaN-preserving, see the forward's prologue)
        db[q] = ob ? t : 0.f;
      }
    }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float* const b0 = smem;
  float* const b1 = smem + 2 * RC_BK * RC_AP;
  if (kbeg < kend) {
    rows(kbeg);
    load(0, kbeg);
  }
  if (kbeg + RC_BK < kend) load(1, kbeg + RC_BK);
  if (kbeg < kend) store(0, b0, b0 + RC_BK * RC_AP);
  __syncthreads();
#pragma unroll 1
  for (int k0 = kbeg; k0 < kend; k0 += 2 * RC_BK) {
    if (k0 + 2 * RC_BK < kend) load(0, k0 + 2 * RC_BK);
    wv.mma(b0, b0 + RC_BK * RC_AP, acc);
    if (k0 + RC_BK < kend) store(1, b1, b1 + RC_BK * RC_AP);
    __syncthreads();
    if (k0 + RC_BK >= kend) break;
    if (k0 + 3 * RC_BK < kend) load(1, k0 + 3 * RC_BK);
    wv.mma(b1, b1 + RC_BK * RC_AP, acc);
    if (k0 + 2 * RC_BK < kend) store(0, b0, b0 + RC_BK * RC_AP);
    __syncthreads();
  }
  const size_t numel = (size_t)a.Cout * a.T * a.Cin, rstride = (size_t)a.T * a.Cin;
  const size_t base = ((size_t)n0 * a.T + tap) * a.Cin + c0;
  unsigned* ticket = a.counter + ((size_t)tap * k.gy + k.by) * k.gx + k.bx;
  if (!rc_acc_reduce(acc, a.ws, ticket, a.nz, zi, numel, base, rstride, 64, wv, lastp)) return;
  float* d = a.dw + base + wv.col();
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float* p = d + (size_t)wv.row_of(r) * rstride;
    *p = a.acc ? *p + acc[r] : acc[r];
  }
}

__global__ __launch_bounds__(256, 3) void rc_wgrad_kernel(RcWgradArgs a, int gx, int gy, int gz, int xcd) {
  __shared__ __attribute__((aligned(16))) float smem[4 * RC_BK * RC_AP];
  __shared__ float cs[128];
  __shared__ int last;
  RcBlk k;
  if (!rc_blk_of((int)blockIdx.x, gx, gy, gz, xcd, k)) return;
  rc_wgrad_body(a, k, smem, cs, &last);
}

// ---------------------------------------------------------------------------------------------------------------------
// the stem: conv1 (7x7 / 2, 3 -> 64 channels) straight from the NCHW image batch, its BatchNorm statistics in the epilogue;
// max pooling that normalises + rectifies while it reads; and the weight gradient.  The reduction index is k = tap * Cimg
// + c (the channels-last weight row), walked 32 at a time: K = 147 is 5 stages, the last one partial.
// ---------------------------------------------------------------------------------------------------------------------
struct RcStemFwdArgs {
  const float* img;       // (B, Cimg, H, W)
  const float* w;         // (Cout, T, Cimg)
  float* y;               // (M = B Ho Wo, Cout)
  int M, Cimg, Cout, T;
  RcGeom g;
  RcBnFwd bn;
};

__global__ __launch_bounds__(256) void rc_stem_fwd_kernel(RcStemFwdArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[4 * RC_BK * RC_AP];
  __shared__ float cs[128];
  __shared__ int last;
  float* As = smem;
  float* Bs = smem + RC_BK * RC_AP;
  const RcBlk k{(int)blockIdx.x, (int)blockIdx.y, 0, (int)gridDim.x, (int)gridDim.y};
  const int tid = threadIdx.x;
  const int n0 = k.bx * 64, m0 = k.by * RC_BM;
  RcWave wv;
  wv.init(tid);
  RcStg<true> sa, sb;
  sa.init(tid);
  sb.init(tid);
  const int K = a.T * a.Cimg, hwi = a.g.H * a.g.W;
  int pix[RC_NS], hw[RC_NS];
  float ra[RC_NS], rb[RC_NS];
#pragma unroll
  for (int i = 0; i < RC_NS; ++i) {
    const int r = m0 + sa.row(i);
    const int hwo = a.g.Ho * a.g.Wo;
    const int b = r / hwo, rem = r - b * hwo, oh = rem / a.g.Wo, ow = rem - oh * a.g.Wo;
    const int h0 = oh * a.g.S - a.g.P, w0 = ow * a.g.S - a.g.P;
    pix[i] = b * a.Cimg * hwi + h0 * a.g.W + w0;
    hw[i] = r < a.M ? ((h0 + 0x4000) << 16 | (w0 + 0x4000)) : -1;
  }
  unsigned oka = 0;
  auto load = [&](int k0) {
    const int kk = k0 + sa.kl;
    const int tap = kk / a.Cimg, c = kk - tap * a.Cimg, kh = tap / a.g.KW, kw = tap - kh * a.g.KW;
    const int koff = c * hwi + kh * a.g.W + kw;
    oka = 0;
#pragma unroll
    for (int i = 0; i < RC_NS; ++i) {
      const int ih = (hw[i] >> 16) - 0x4000 + kh, iw = (hw[i] & 0xffff) - 0x4000 + kw;
      const bool ok = kk < K && hw[i] != -1 && (unsigned)ih < (unsigned)a.g.H && (unsigned)iw < (unsigned)a.g.W;
      oka |= (ok ? 1u : 0u) << i;
      ra[i] = a.img[ok ? pix[i] + koff : 0];
    }
#pragma unroll
    for (int i = 0; i < RC_NS; ++i) {
      const float t = a.w[(size_t)(n0 + sb.row(i)) * K + min(k0 + sb.kl, K - 1)];
      rb[i] = k0 + sb.kl < K ? t : 0.f;
    }
  };
  auto store = [&](float* __restrict__ As_, float* __restrict__ Bs_) {
    float va[RC_NS];
#pragma unroll
    for (int i = 0; i < RC_NS; ++i) va[i] = (oka >> i & 1u) ? ra[i] : 0.f;
    sa.store(As_, va);
    sb.store(Bs_, rb);
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  load(0);
  store(As, Bs);
  __syncthreads();
  int cur = 0;
#pragma unroll 1
  for (int k0 = 0; k0 < K; k0 += RC_BK) {
    const int o = cur * 2 * RC_BK * RC_AP, on = (cur ^ 1) * 2 * RC_BK * RC_AP;
    if (k0 + RC_BK < K) load(k0 + RC_BK);
    wv.mma(As + o, Bs + o, acc);
    if (k0 + RC_BK < K) store(As + on, Bs + on);
    __syncthreads();
    cur ^= 1;
  }
  rc_fwd_tail(acc, a.y, a.M, a.Cout, a.bn, k, m0, n0, min(RC_BM, a.M - m0), wv, smem, cs, &last);
}

// dw (Cout, K = T Cimg) (+)= dY^T patches(img), dY = G p + Y q + r; tbl (2, M): per output pixel the image offset of its
// window origin (b Cimg H W + h0 W + w0) and (h0, w0) packed as in the forward kernel.  grid (ceil(K / 64), Cout / 64, nz)
struct RcStemWgradArgs {
  const float* G;
  const float* Y;
  const float* pqr;
  const float* img;
  const int* tbl;
  float* dw;
  float* ws;              // nz > 1: (nz, Cout, Kpad = 64 gridDim.x)
  unsigned* counter;
  int M, Cimg, Cout, T, acc, nz, kper;
  RcGeom g;
};

__global__ __launch_bounds__(256) void rc_stem_wgrad_kernel(RcStemWgradArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[4 * RC_BK * RC_AP];
  __shared__ int last;
  float* As = smem;
  float* Bs = smem + RC_BK * RC_AP;
  const int tid = threadIdx.x;
  const int c0 = blockIdx.x * 64, n0 = blockIdx.y * 64, zi = blockIdx.z;
  const int kbeg = zi * a.kper, kend = min(a.M, kbeg + a.kper);
  const int K = a.T * a.Cimg, Kpad = 64 * gridDim.x, hwi = a.g.H * a.g.W;
  RcWave wv;
  wv.init(tid);
  RcStg<false> sa, sb;
  sa.init(tid);
  sb.init(tid);
  float rg_[RC_NS], ry[RC_NS], rb[RC_NS];
  int tp[RC_NS], th[RC_NS];
  unsigned oka = 0, okb = 0;
  const int n = n0 + sa.rl, j = c0 + sb.rl;
  const int tap = j / a.Cimg, c = j - tap * a.Cimg, kh = tap / a.g.KW, kw = tap - kh * a.g.KW;
  const int koff = c * hwi + kh * a.g.W + kw;
  float pp = 1.f, pq = 0.f, pr = 0.f;
  if (a.pqr) { pp = a.pqr[n]; pq = a.pqr[a.Cout + n]; pr = a.pqr[2 * a.Cout + n]; }
  auto rows = [&](int k0) {
#pragma unroll
    for (int i = 0; i < RC_NS; ++i) {
      const int m = k0 + sb.kk(i);
      tp[i] = a.tbl[min(m, a.M - 1)];
      th[i] = m < kend ? a.tbl[a.M + min(m, a.M - 1)] : -1;
    }
  };
  auto load = [&](int k0) {
    oka = 0;
    okb = 0;
#pragma unroll
    for (int i = 0; i < RC_NS; ++i) {
      const int m = k0 + sa.kk(i);
      const bool ok = m < kend;
      oka |= (ok ? 1u : 0u) << i;
      const size_t o = ok ? (size_t)m * a.Cout + n : 0;
      rg_[i] = a.G[o];
      ry[i] = a.pqr ? a.Y[o] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < RC_NS; ++i) {
      const int ih = (th[i] >> 16) - 0x4000 + kh, iw = (th[i] & 0xffff) - 0x4000 + kw;
      const bool ok = j < K && th[i] != -1 && (unsigned)ih < (unsigned)a.g.H && (unsigned)iw < (unsigned)a.g.W;
      okb |= (ok ? 1u : 0u) << i;
      rb[i] = a.img[ok ? tp[i] + koff : 0];
    }
    if (k0 + RC_BK < kend) rows(k0 + RC_BK);
  };
  auto store = [&](float* __restrict__ As_, float* __restrict__ Bs_) {
    float va[RC_NS], vb[RC_NS];
#pragma unroll
    for (int i = 0; i < RC_NS; ++i) {
      const float v = a.pqr ? fmaf(rg_[i], pp, fmaf(ry[i], pq, pr)) : rg_[i];
      va[i] = (oka >> i & 1u) ? v : 0.f;
      vb[i] = (okb >> i & 1u) ? rb[i] : 0.f;
    }
    sa.store(As_, va);
    sb.store(Bs_, vb);
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (kbeg < kend) {
    rows(kbeg);
    load(kbeg);
  }
  if (kbeg < kend) store(As, Bs);
  __syncthreads();
  int cur = 0;
#pragma unroll 1
  for (int k0 = kbeg; k0 < kend; k0 += RC_BK) {
    const int o = cur * 2 * RC_BK * RC_AP, on = (cur ^ 1) * 2 * RC_BK * RC_AP;
    if (k0 + RC_BK < kend) load(k0 + RC_BK);
    wv.mma(As + o, Bs + o, acc);
    if (k0 + RC_BK < kend) store(As + on, Bs + on);
    __syncthreads();
    cur ^= 1;
  }
  unsigned* ticket = a.counter + blockIdx.y * gridDim.x + blockIdx.x;
  if (!rc_acc_reduce(acc, a.ws, ticket, a.nz, zi, (size_t)a.Cout * Kpad, (size_t)n0 * Kpad + c0, Kpad, 64, wv, &last)) return;
  const int jj = c0 + wv.col();
  if (jj < K) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float* p = a.dw + (size_t)(n0 + wv.row_of(r)) * K + jj;
      *p = a.acc ? *p + acc[r] : acc[r];
    }
  }
}

// MaxPool2d(3, 2, 1) on relu(bn(y)) of the raw stem output y (B H W, C): out (B Ho Wo, C), idx = h W + w of the first
// maximum (torch: strict '>')
__global__ __launch_bounds__(256) void rc_maxpool_fwd_kernel(const float* __restrict__ y, const float* __restrict__ mean,
                                                             const float* __restrict__ sc, const float* __restrict__ beta,
                                                             float* __restrict__ out, int* __restrict__ idx, int B, int H,
                                                             int W, int C, int Ho, int Wo) {
  const long total = (long)B * Ho * Wo * C;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = (int)(e % C);
    const long r = e / C;
    const int ow = (int)(r % Wo), oh = (int)((r / Wo) % Ho), b = (int)(r / ((long)Wo * Ho));
    const float m = mean[c], s = sc[c], bt = beta[c];
    float best = -INFINITY;
    int bi = -1;
    for (int kh = 0; kh < 3; ++kh) {
      const int ih = 2 * oh - 1 + kh;
      if (ih < 0 || ih >= H) continue;
      for (int kw = 0; kw < 3; ++kw) {
        const int iw = 2 * ow - 1 + kw;
        if (iw < 0 || iw >= W) continue;
        const float v = fmaxf(rc_bn(y[(((size_t)b * H + ih) * W + iw) * C + c], m, s, bt), 0.f);
        if (v > best || bi < 0) {
          best = v;
          bi = ih * W + iw;
        }
      }
    }
    out[e] = best;
    idx[e] = bi;
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
// several independent jobs in ONE launch (a layer's data and weight gradient + the projection shortcut's; a block's first
// convolution and its projection): at batch 24 every job alone leaves most of the chip idle and costs a dependent launch
// ---------------------------------------------------------------------------------------------------------------------
#define RC_KIND_FWD 0
#define RC_KIND_DGRAD 1
#define RC_KIND_WGRAD 2
struct RcPlan { int kind, gx, gy, gz, xcd; };
union RcAny {
  RcFwdArgs f;
  RcDgradArgs d;
  RcWgradArgs w;
};
struct RcGroup {            // kernel arguments: < 4 KB
  RcAny job[MMVAE_RC_MAX_JOBS];
  RcPlan plan[MMVAE_RC_MAX_JOBS];
  int blk0[MMVAE_RC_MAX_JOBS + 1];
  int n;
};
static_assert(sizeof(RcGroup) <= 4000, "rc_group_kernel's arguments must fit the 4 KB kernel-argument segment");

// (the job table is read through the kernel-argument segment pointer: indexing the by-value parameter with the
// workgroup's job number makes the compiler copy all of it into scratch -- 2.5 KB per lane, 10 x the run time)
__global__ __launch_bounds__(256, 3) void rc_group_kernel(RcGroup g_) {
  __shared__ __attribute__((aligned(16))) float smem[4 * RC_BK * RC_AP];
  __shared__ float cs[128];
  __shared__ int last;
  const RcGroup* __restrict__ g = (const RcGroup*)__builtin_amdgcn_kernarg_segment_ptr();
  int p = 0;
#pragma unroll
  for (int q = 1; q < MMVAE_RC_MAX_JOBS; ++q)
    if (q < g->n && (int)blockIdx.x >= g->blk0[q]) p = q;
  const int local = blockIdx.x - g->blk0[p];          // (blk0 are multiples of 8)
  const RcPlan pl = g->plan[p];
  RcBlk k;
  if (!rc_blk_of(local, pl.gx, pl.gy, pl.gz, pl.xcd, k)) return;
  if (pl.kind == RC_KIND_FWD) {
    const RcFwdArgs a = g->job[p].f;
    rc_fwd_body(a, k, smem, cs, &last);
  } else if (pl.kind == RC_KIND_DGRAD) {
    const RcDgradArgs a = g->job[p].d;
    rc_dgrad_body(a, k, smem, cs, &last);
  } else {
    const RcWgradArgs a = g->job[p].w;
    rc_wgrad_body(a, k, smem, cs, &last);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------------
// workgroups a split job aims for (measured on the shipped CdSprites+ config at
// batch 24: 128 / 256 / 512 for the forward and data-gradient jobs 4.48 / 4.18 / 4.22 ms per step, 64 / 128 / 256 / 512+
// for the weight gradients 4.69 / 4.30 / 4.20 / 4.18)
static inline long rc_target() { return 256L; }

// row tile -> XCD placement of the forward / data-gradient jobs (kept in the launch plumbing, off).  MEASURED in round 3, no gain: the
// shipped CdSprites+ step 8.95 against 8.95 ms at batch 128 (every layer has >= 8 row tiles there) -- what one launch
// leaves in an XCD's L2 is not what the next launch's misses are about (L2 hit rate 0.55 either way: weights and the
// other tiles' rows) -- and 4.57 against 4.16 ms at batch 24, where layers 3 and 4 have 6 and 2 row tiles and the
// placement leaves most XCDs idle.
static inline int rc_xcd() { return 0; }
static inline int rc_plan_blocks(const RcPlan& pl) {
  const int n = pl.xcd ? 8 * ((pl.gy + 7) / 8) * pl.gx * pl.gz : pl.gx * pl.gy * pl.gz;
  return (n + 7) / 8 * 8;
}
static inline long rc_target_w() { return 512L; }

extern "C" int mmvae_rc_row_tile(int M, int N) { return RC_BM; }   // rows per statistics partial

// split of the reduction (T taps x K channels) of a forward / data-gradient GEMM with an (M, N) output over
// blockIdx.z: >= 2 stages per split, enough workgroups for two per CU
extern "C" int mmvae_rc_conv_splits(int M, int N, int K, int T) {
  const long tiles = (long)((M + 63) / 64) * (N / 64);
  const int nstage = T * (K / RC_BK);
  long nz = (rc_target() + tiles - 1) / tiles;
  if (nz > (nstage + 1) / 2) nz = (nstage + 1) / 2;
  if (nz > 32) nz = 32;
  return (int)(nz < 1 ? 1 : nz);
}
extern "C" size_t mmvae_rc_conv_ws_floats(int M, int N, int K, int T) {      // rows by 64-row tile, + 4 tiles for the
  const int nz = mmvae_rc_conv_splits(M, N, K, T);                             // parity-class order of stride-2 gradients
  return nz > 1 ? (size_t)nz * ((size_t)(M + 63) / 64 * 64 + 256) * N : 0;
}

extern "C" int mmvae_rc_tables(int* fwd, int* bwd, int B, int H, int W, int K, int S, int P, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(fwd && bwd && B > 0 && H > 0 && W > 0 && K > 0 && S > 0);
  const int Ho = (H + 2 * P - K) / S + 1, Wo = (W + 2 * P - K) / S + 1;
  const long n = (long)B * (long)max(H * W, Ho * Wo) * K * K;
  hipLaunchKernelGGL(rc_tables_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, fwd, bwd, B, H,
                     W, Ho, Wo, K, S, P);
  return mmvae_launch_status();
}

// split of the pixel rows of a weight gradient: enough workgroups to fill the chip, >= 128 rows per split
extern "C" int mmvae_rc_wgrad_splits(int M, int Cin, int Cout, int T) {
  const long tiles = (long)(Cout / 64) * (Cin / 64) * T;
  long nz = (rc_target_w() + tiles - 1) / tiles;
  const long maxz = (M + 127) / 128;
  if (nz > maxz) nz = maxz;
  if (nz > 64) nz = 64;
  return (int)(nz < 1 ? 1 : nz);
}
extern "C" size_t mmvae_rc_wgrad_ws_floats(int M, int Cin, int Cout, int T) {
  const int nz = mmvae_rc_wgrad_splits(M, Cin, Cout, T);
  return nz > 1 ? (size_t)nz * Cout * T * Cin : 0;
}
extern "C" size_t mmvae_rc_wgrad_tickets(int Cin, int Cout, int T) { return (size_t)(Cout / 64) * (Cin / 64) * T; }

static RcStat rc_stat_of(const mmvae_rc_stat_t& s) {
  return RcStat{s.Y, s.mean, s.rstd, s.gamma, s.pqr, s.dgamma, s.dbeta, s.part, s.counter, s.acc, s.eval};
}
static bool rc_geom_ok(const mmvae_rc_geom_t& g, int T) {
  return g.KW >= 1 && T % g.KW == 0 && g.S >= 1 && g.H > 0 && g.W > 0 && g.Ho > 0 && g.Wo > 0 && g.H < 0x4000 && g.W < 0x4000;
}
static RcGeom rc_geom_of(const mmvae_rc_geom_t& g) { return RcGeom{g.H, g.W, g.Ho, g.Wo, g.KW, g.S, g.P}; }

static int rc_plan_fwd(const mmvae_rc_fwd_t& j, RcFwdArgs& a, RcPlan& pl) {
  MMVAE_CHECK_ARG(j.x && j.w && j.y && j.M > 0 && j.Cin % 64 == 0 && j.Cout % 64 == 0 && j.T >= 1 && rc_geom_ok(j.g, j.T));
  MMVAE_CHECK_ARG(j.pre != RC_PRE_BN_RELU || (j.xmean && j.xsc && j.xbeta));
  MMVAE_CHECK_ARG(!j.part || (j.gamma && j.beta && j.mean && j.rstd && j.sc && j.counter));
  MMVAE_CHECK_ARG((long)j.M * j.Cout < (1L << 31) && (long)j.M * j.Cin * j.g.S * j.g.S < (1L << 31));
  const int nz = mmvae_rc_conv_splits(j.M, j.Cout, j.Cin, j.T);
  MMVAE_CHECK_ARG(nz == 1 || (j.ws && j.tile_ticket));
  a = RcFwdArgs{j.x, j.w, j.xmean, j.xsc, j.xbeta, j.y, j.ws, j.tile_ticket, j.M, j.Cin, j.Cout, j.T, j.pre, nz, rc_geom_of(j.g),
                {j.gamma, j.beta, j.run_mean, j.run_var, j.mean, j.rstd, j.sc, j.part, j.counter, j.eps, j.momentum, j.eval}};
  pl = RcPlan{RC_KIND_FWD, j.Cout / 64, (j.M + 63) / 64, nz, rc_xcd()};
  return MMVAE_OK;
}

static int rc_plan_dgrad(const mmvae_rc_dgrad_t& j, RcDgradArgs& a, RcPlan& pl) {
  MMVAE_CHECK_ARG(j.G && j.w && j.out && j.M > 0 && j.Min > 0 && j.Cin % 64 == 0 && j.Cout % 64 == 0 && j.T >= 1 &&
                  rc_geom_ok(j.g, j.T));
  MMVAE_CHECK_ARG((!j.pqr || j.Y) && j.nstat >= 0 && j.nstat <= 2);
  MMVAE_CHECK_ARG(j.mask == RC_MASK_NONE || j.mY);
  MMVAE_CHECK_ARG(j.mask != RC_MASK_BN || (j.mmean && j.msc && j.mbeta));
  int nz = mmvae_rc_conv_splits(j.Min, j.Cin, j.Cout, j.T), cls_tiles = 0, row_tiles = (j.Min + 63) / 64;
  if (j.row_map) {      // parity-class order of a stride-2 data gradient: 4 x Min / 4 rows; the one-tap class has
    MMVAE_CHECK_ARG(j.g.S == 2 && j.T > 1 && j.g.H % 2 == 0 && j.g.W % 2 == 0 && j.Min % 4 == 0);   // Cout / 32 stages
    if (nz > j.Cout / RC_BK) nz = j.Cout / RC_BK;
    cls_tiles = (j.Min / 4 + 63) / 64;
    row_tiles = 4 * cls_tiles;
  }
  MMVAE_CHECK_ARG(nz == 1 || (j.ws && j.tile_ticket));
  a = RcDgradArgs{j.G, j.Y, j.pqr, j.w, j.add, j.add_tbl, j.mY, j.mmean, j.msc, j.mbeta, j.out, j.ws, j.tile_ticket, j.row_map,
                  j.M, j.Min, j.Cin, j.Cout, j.T, j.mask, j.nstat, nz, cls_tiles, rc_geom_of(j.g), {}};
  for (int t = 0; t < j.nstat; ++t) {
    MMVAE_CHECK_ARG(j.st[t].Y && j.st[t].pqr && j.st[t].part && j.st[t].counter);
    a.st[t] = rc_stat_of(j.st[t]);
  }
  pl = RcPlan{RC_KIND_DGRAD, j.Cin / 64, row_tiles, nz, rc_xcd()};
  return MMVAE_OK;
}

static int rc_plan_wgrad(const mmvae_rc_wgrad_t& j, RcWgradArgs& a, RcPlan& pl) {
  MMVAE_CHECK_ARG(j.G && j.x && j.dw && j.M > 0 && j.Cin % 64 == 0 && j.Cout % 64 == 0 && j.T >= 1 && (!j.pqr || j.Y));
  MMVAE_CHECK_ARG(j.pre != RC_PRE_BN_RELU || (j.xmean && j.xsc && j.xbeta));
  const int nz = mmvae_rc_wgrad_splits(j.M, j.Cin, j.Cout, j.T);
  MMVAE_CHECK_ARG(nz == 1 || (j.ws && j.counter));
  int kper = (j.M + nz - 1) / nz;
  kper = (kper + RC_BK - 1) / RC_BK * RC_BK;
  a = RcWgradArgs{j.G, j.Y, j.pqr, j.x, j.xmean, j.xsc, j.xbeta, j.tbl, j.dw, j.ws, j.counter, j.M, j.Cin, j.Cout, j.T, j.pre,
                  j.accumulate ? 1 : 0, nz, kper};
  pl = RcPlan{RC_KIND_WGRAD, j.Cin / 64, j.Cout / 64, j.T * nz, 0};
  return MMVAE_OK;
}

extern "C" int mmvae_rc_launch(const mmvae_rc_job_t* jobs, int n, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(jobs && n >= 1 && n <= MMVAE_RC_MAX_JOBS);
  hipStream_t st = (hipStream_t)stream;
  static thread_local RcGroup g;     // < 4 KB of kernel arguments, assembled in place
  g.n = n;
  g.blk0[0] = 0;
  for (int p = 0; p < n; ++p) {
    int rc;
    if (jobs[p].kind == RC_KIND_FWD) rc = rc_plan_fwd(jobs[p].f, g.job[p].f, g.plan[p]);
    else if (jobs[p].kind == RC_KIND_DGRAD) rc = rc_plan_dgrad(jobs[p].d, g.job[p].d, g.plan[p]);
    else if (jobs[p].kind == RC_KIND_WGRAD) rc = rc_plan_wgrad(jobs[p].w, g.job[p].w, g.plan[p]);
    else return MMVAE_ERR_ARG;
    if (rc != MMVAE_OK) return rc;
    g.blk0[p + 1] = g.blk0[p] + rc_plan_blocks(g.plan[p]);
  }
  if (n == 1) {       // a job alone keeps its own kernel (and its name in a profile)
    const RcPlan& pl = g.plan[0];
    const dim3 grid(rc_plan_blocks(pl));
    if (pl.kind == RC_KIND_FWD)
      hipLaunchKernelGGL(rc_fwd_kernel, grid, dim3(256), 0, st, g.job[0].f, pl.gx, pl.gy, pl.gz, pl.xcd);
    else if (pl.kind == RC_KIND_DGRAD)
      hipLaunchKernelGGL(rc_dgrad_kernel, grid, dim3(256), 0, st, g.job[0].d, pl.gx, pl.gy, pl.gz, pl.xcd);
    else hipLaunchKernelGGL(rc_wgrad_kernel, grid, dim3(256), 0, st, g.job[0].w, pl.gx, pl.gy, pl.gz, pl.xcd);
    return mmvae_launch_status();
  }
  hipLaunchKernelGGL(rc_group_kernel, dim3(g.blk0[n]), dim3(256), 0, st, g);
  return mmvae_launch_status();
}

extern "C" int mmvae_rc_bn_bwd_stats(const float* G, const mmvae_rc_stat_t* st, int M, int C, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(G && st && st->Y && st->pqr && st->part && st->counter && M > 0 && C % 64 == 0);
  hipLaunchKernelGGL(rc_stat_kernel, dim3(C / 64, (M + 63) / 64), dim3(256), 0, (hipStream_t)stream, G, rc_stat_of(*st), M, C,
                     (const float*)nullptr, (const float*)nullptr, (float*)nullptr, 1, (const int*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, 1, 1, 1, 1);
  return mmvae_launch_status();
}

extern "C" int mmvae_rc_pool_bwd_stats(const float* dy, const float* x, float* G, const mmvae_rc_stat_t* st, int B, int HW,
                                       int C, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(dy && x && G && st && st->Y && st->pqr && st->part && st->counter && B > 0 && HW > 0 && C % 64 == 0);
  const int M = B * HW;
  hipLaunchKernelGGL(rc_stat_kernel, dim3(C / 64, (M + 63) / 64), dim3(256), 0, (hipStream_t)stream, (const float*)nullptr,
                     rc_stat_of(*st), M, C, dy, x, G, HW, (const int*)nullptr, (const float*)nullptr, (const float*)nullptr, 1, 1, 1,
                     1);
  return mmvae_launch_status();
}

extern "C" int mmvae_rc_maxpool_fwd(const float* y, const float* mean, const float* sc, const float* beta, float* out,
                                    int* idx, int B, int H, int W, int C, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(y && mean && sc && beta && out && idx && B > 0 && H > 1 && W > 1 && C > 0);
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long n = (long)B * Ho * Wo * C;
  hipLaunchKernelGGL(rc_maxpool_fwd_kernel, dim3((unsigned)min((n + 255) / 256, 65535L * 16)), dim3(256), 0,
                     (hipStream_t)stream, y, mean, sc, beta, out, idx, B, H, W, C, Ho, Wo);
  return mmvae_launch_status();
}

extern "C" int mmvae_rc_maxpool_bwd_stats(const float* dy, const int* idx, float* G, const float* sc, const float* beta,
                                          const mmvae_rc_stat_t* st, int B, int H, int W, int C, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(dy && idx && G && sc && beta && st && st->Y && st->mean && st->pqr && st->part && st->counter && B > 0 &&
                  H > 1 && W > 1 && C % 64 == 0);
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1, M = B * H * W;
  hipLaunchKernelGGL(rc_stat_kernel, dim3(C / 64, (M + 63) / 64), dim3(256), 0, (hipStream_t)stream, (const float*)nullptr,
                     rc_stat_of(*st), M, C, dy, (const float*)nullptr, G, 1, idx, sc, beta, H, W, Ho, Wo);
  return mmvae_launch_status();
}

extern "C" int mmvae_rc_stem_fwd(const float* img, const float* w, float* y, int M, int Cimg, int Cout, int T,
                                 const mmvae_rc_geom_t* g, const float* gamma, const float* beta, float* run_mean,
                                 float* run_var, float* mean, float* rstd, float* sc, float* part, unsigned* counter,
                                 float eps, float momentum, int eval, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(img && w && y && g && M > 0 && Cimg > 0 && Cout % 64 == 0 && T >= 1 && rc_geom_ok(*g, T));
  MMVAE_CHECK_ARG(part && gamma && beta && mean && rstd && sc && counter);
  RcStemFwdArgs a{img, w, y, M, Cimg, Cout, T, rc_geom_of(*g),
                  {gamma, beta, run_mean, run_var, mean, rstd, sc, part, counter, eps, momentum, eval}};
  hipLaunchKernelGGL(rc_stem_fwd_kernel, dim3(Cout / 64, (M + 63) / 64), dim3(256), 0, (hipStream_t)stream, a);
  return mmvae_launch_status();
}

extern "C" int mmvae_rc_stem_wgrad_splits(int M, int Cimg, int Cout, int T) {
  const long tiles = (long)(Cout / 64) * ((T * Cimg + 63) / 64);
  long nz = (rc_target_w() + tiles - 1) / tiles;
  const long maxz = (M + 127) / 128;
  if (nz > maxz) nz = maxz;
  if (nz > 128) nz = 128;
  return (int)(nz < 1 ? 1 : nz);
}
extern "C" size_t mmvae_rc_stem_wgrad_ws_floats(int M, int Cimg, int Cout, int T) {
  return (size_t)mmvae_rc_stem_wgrad_splits(M, Cimg, Cout, T) * Cout * (((size_t)T * Cimg + 63) / 64 * 64);
}

extern "C" int mmvae_rc_stem_wgrad(const float* G, const float* Y, const float* pqr, const float* img, const int* tbl,
                                   float* dw, float* ws, unsigned* counter, int M, int Cimg, int Cout, int T,
                                   const mmvae_rc_geom_t* g, int accumulate, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(G && img && tbl && dw && ws && counter && g && M > 0 && Cimg > 0 && Cout % 64 == 0 && T >= 1 && (!pqr || Y) &&
                  rc_geom_ok(*g, T));
  const int nz = mmvae_rc_stem_wgrad_splits(M, Cimg, Cout, T);
  int kper = (M + nz - 1) / nz;
  kper = (kper + RC_BK - 1) / RC_BK * RC_BK;
  RcStemWgradArgs a{G, Y, pqr, img, tbl, dw, ws, counter, M, Cimg, Cout, T, accumulate ? 1 : 0, nz, kper, rc_geom_of(*g)};
  hipLaunchKernelGGL(rc_stem_wgrad_kernel, dim3((T * Cimg + 63) / 64, Cout / 64, nz), dim3(256), 0, (hipStream_t)stream, a);
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

// ---------------------------------------------------------------------------------------------------------------------
// Global average pooling of the tower (torchvision resnet50.avgpool on relu(x)), the two kernels of the tower that are not
// GEMM-shaped (folded in from csrc/resnet.hip in round 4)
// ---------------------------------------------------------------------------------------------------------------------
static inline unsigned rc_ew_blocks(long n) {
  long b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
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
  hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(rc_ew_blocks((long)B * C)), dim3(256), 0, (hipStream_t)stream, x, y, B, HW, C,
                     in_act);
  return mmvae_launch_status();
}
extern "C" int mmvae_avgpool_bwd(const float* dy, const float* x, float* dx, int B, int HW, int C, int in_act,
                                 mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(dy && x && dx && B > 0 && HW > 0 && C > 0);
  hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(rc_ew_blocks((long)B * HW * C)), dim3(256), 0, (hipStream_t)stream, dy, x, dx,
                     B, HW, C, in_act);
  return mmvae_launch_status();
}
