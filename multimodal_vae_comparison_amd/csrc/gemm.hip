// Strided fp32 GEMM on v_mfma_f32_32x32x2_f32 for the small dense layers of the towers
// (nn.Linear fwd / dgrad / wgrad, attention projections, FFN).  gfx950 only.
//
//   C[M,N] = ep( bias + act(A)[M,K] * B[K,N] )
//
// Workgroup = 4 wavefronts.  Two tilings share one body:
//   KSPLIT = 1 : 128 x 32 output tile, wave w owns rows [32w, 32w+32), K staged 32 deep;
//   KSPLIT = 4 / 8 : 32 x 32 output tile, the 4 (8) waves split every 128 (256)-deep K stage and the
//                partial accumulators are summed through LDS (used when M*N is too small to fill the
//                chip: batch-128 linears, weight gradients).
// A and B tiles are staged global -> registers -> LDS as [k][m] / [k][n] (row pitch odd => both the
// transposing store and the per-lane fragment read are bank-conflict free), with the next stage's global
// loads issued before the current stage's MFMAs.  MFMA operand layout (32x32x2 f32): lane l supplies
// A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31]; accumulator register r of lane l is
// C[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31].
// gridDim.z > 1 = split-K across workgroups: raw partial tiles go to `ws` ([z][M][N] then [z][M] row sums)
// and are summed by mmvae_reduce_rows (deterministic, no atomics).
#include "common.hpp"

#define GEMM_BK1 64   // K depth of a stage of the 128 x 32 tiling

struct GemmArgs {
  const float* A;
  const float* B;
  const float* bias;
  float* aux;  // read for MUL_* epilogues, written (pre-activation) for EP_GELU
  float* C;
  float* a_rowsum;
  float* ws;
  int M, N, K;
  long sam, sak, sbk, sbn, ldc;
  int a_act, b_act, ep, accumulate, kper;
};

// Staging of one operand tile [R rows (m or n)][BK] into LDS as [k][r] (pitch RP).  The tile is walked with a
// per-thread base offset and a constant stride, so a slot costs one address add, one select and one LDS store
// (the first version decoded (row, k) and did 64-bit index math per element: ~25 VALU per slot and 30 KB of code,
// which at batch 128 is what the kernel's time went into).
//   KMAJOR (k contiguous in memory):  slot i = (row r0 + i*RSTEP, k kl);   RSTEP = NT / BK
//   else   (row contiguous):          slot i = (row rl, k kl0 + i*KSTEP);  KSTEP = NT / R
template <int R, int BK, int NT, bool KMAJOR>
struct OperandStage {
  static constexpr int PER_T = R * BK / NT;
  static constexpr int RSTEP = NT / BK, KSTEP = NT / R;
  static constexpr int RP = R + 1;
  int rl, kl;            // this thread's first (row, k) inside the tile
  long rstride, kstride; // element strides of row / k in memory
  unsigned rvalid;       // KMAJOR: per-slot row validity (stage independent)
  int rows_left;         // rows valid in this tile
  __device__ __forceinline__ void init(int tid, int row0, int nrows, long rs, long ks) {
    if (KMAJOR) { kl = tid % BK; rl = tid / BK; } else { rl = tid % R; kl = tid / R; }
    rstride = rs; kstride = ks;
    rows_left = nrows - row0;
    rvalid = 0;
#pragma unroll
    for (int i = 0; i < PER_T; ++i) {
      const int r = KMAJOR ? rl + i * RSTEP : rl;
      rvalid |= (r < rows_left ? 1u : 0u) << i;
    }
  }
  // Unconditional loads of stage [k0, kend) from clamped addresses (a predicated load makes hipcc wait vmcnt(0)
  // per element); returns the validity mask that store() applies.
  __device__ __forceinline__ unsigned load(const float* __restrict__ base, int k0, int kend, float (&v)[PER_T]) const {
    unsigned ok = 0;
    const long off = (long)rl * rstride + (long)(k0 + kl) * kstride;
    if (KMAJOR) {
      ok = (k0 + kl < kend) ? rvalid : 0u;
      const long step = (long)RSTEP * rstride;
#pragma unroll
      for (int i = 0; i < PER_T; ++i) v[i] = base[(ok >> i & 1u) ? off + i * step : 0];
    } else {
      const bool rok = rl < rows_left;
      const long step = (long)KSTEP * kstride;
#pragma unroll
      for (int i = 0; i < PER_T; ++i) {
        const bool e = rok && (k0 + kl + i * KSTEP < kend);
        ok |= (e ? 1u : 0u) << i;
        v[i] = base[e ? off + i * step : 0];
      }
    }
    return ok;
  }
  __device__ __forceinline__ void store(float* __restrict__ S, const float (&v)[PER_T], unsigned ok) const {
    float* d = S + kl * RP + rl;
#pragma unroll
    for (int i = 0; i < PER_T; ++i) d[KMAJOR ? i * RSTEP : i * KSTEP * RP] = (ok >> i & 1u) ? v[i] : 0.f;
  }
};

template <int N>
__device__ __forceinline__ void act_inplace(float (&v)[N], int act) {
  if (act == MMVAE_ACT_SILU) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = dev_silu(v[i]);
  } else if (act == MMVAE_ACT_RELU) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = fmaxf(v[i], 0.f);
  } else if (act == MMVAE_ACT_GELU) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = dev_gelu(v[i]);
  }
}

#include "gemm_b16.inc"

// Output phase shared by the staged and the register-operand bodies: (KSPLIT > 1) sum the waves' partial accumulators
// through LDS (`red`: KSPLIT x 16 x 64 floats, `rsr`: KSPLIT x 32), then bias / epilogue / accumulate / store.
template <int KSPLIT>
__device__ __forceinline__ void gemm_finish(const GemmArgs& g, const int bx, const int m0, const int n0, const int bz,
                                            const int nz, const f32x16& acc, float asum, float* __restrict__ red,
                                            float* __restrict__ rsr) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const bool partial = nz > 1;
  float* Cout = partial ? g.ws + (size_t)bz * g.M * g.N : g.C;
  const long ldc = partial ? g.N : g.ldc;
  float* rs_out = partial ? g.ws + (size_t)nz * g.M * g.N + (size_t)bz * g.M : g.a_rowsum;
  const int ep = partial ? MMVAE_EP_NONE : g.ep;
  const bool acc_out = !partial && g.accumulate;

  asum += __shfl_xor(asum, 32, 64);

  const bool plain = ep == MMVAE_EP_NONE && !acc_out;   // the common case: no per-element branching
  const float bias_v = (!partial && g.bias && n0 + li < g.N) ? g.bias[n0 + li] : 0.f;   // every emit of a lane has col n0+li
  auto emit = [&](int row, int col, float v) {
    if (row < g.M && col < g.N) {
      v += bias_v;
      const long o = (long)row * ldc + col;
      if (!plain) {
        float av = 0.f;
        if (ep_reads_aux(ep)) av = g.aux[o];
        if (ep == MMVAE_EP_GELU && g.aux) g.aux[o] = v;
        v = apply_epilogue(v, av, ep);
        if (acc_out) v += Cout[o];
      }
      Cout[o] = v;
    }
  };

  if (KSPLIT == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) emit(m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, n0 + li, acc[r]);
    if (g.a_rowsum && bx == 0 && lh == 0) {
      const int row = m0 + wave * 32 + li;
      if (row < g.M) rs_out[row] = acc_out ? rs_out[row] + asum : asum;
    }
  } else {
    #pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
    if (lh == 0) rsr[wave * 32 + li] = asum;
    __syncthreads();
    constexpr int RPW = 16 / KSPLIT > 0 ? 16 / KSPLIT : 1;  // accumulator registers summed and emitted per wave
    if (wave < 16 / RPW) {
#pragma unroll
      for (int q = 0; q < RPW; ++q) {
        const int r = wave * RPW + q;
        float v = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < KSPLIT; ++w2) v += red[(w2 * 16 + r) * 64 + lane];
        emit(m0 + (r & 3) + 8 * (r >> 2) + 4 * lh, n0 + li, v);
      }
    }
    if (g.a_rowsum && bx == 0 && wave == 0 && lh == 0) {
      const int row = m0 + li;
      float v = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < KSPLIT; ++w2) v += rsr[w2 * 32 + li];
      if (row < g.M) rs_out[row] = acc_out ? rs_out[row] + v : v;
    }
  }
}

template <int KSPLIT, bool A_KMAJOR, bool B_KMAJOR>
__device__ __forceinline__ void gemm_body(const GemmArgs& g, const int bx, const int by, const int bz, const int nz,
                                          float* __restrict__ As, float* __restrict__ Bs) {
  constexpr int NT = KSPLIT == 8 ? 512 : 256;  // threads
  constexpr int BM = KSPLIT == 1 ? 128 : 32;
  constexpr int BN = 32;
  constexpr int BK = KSPLIT == 1 ? GEMM_BK1 : 32 * KSPLIT;  // KSPLIT > 1: every wave owns a 32-deep K slice of a stage
  constexpr int KW = KSPLIT == 1 ? BK : 32;                 // K steps (x2) a wave walks per stage
  constexpr int AP = BM + 1;  // LDS row pitch of As (odd)
  constexpr int BP = BN + 1;
  using StA = OperandStage<BM, BK, NT, A_KMAJOR>;
  using StB = OperandStage<BN, BK, NT, B_KMAJOR>;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int m0 = by * BM, n0 = bx * BN;
  const int kbeg = bz * g.kper;
  const int kend = min(g.K, kbeg + g.kper);

  // Staging in two phases so that all of a stage's global loads are in flight together: (1) predicated loads
  // into registers, (2) activation + LDS stores.
  StA sa;
  StB sb;
  sa.init(tid, 0, g.M - m0, g.sam, g.sak);
  sb.init(tid, 0, g.N - n0, g.sbn, g.sbk);
  const float* Abase = g.A + (long)m0 * g.sam;
  const float* Bbase = g.B + (long)n0 * g.sbn;
  float ra[StA::PER_T], rb[StB::PER_T];
  unsigned oka = 0, okb = 0;
  auto load_stage = [&](int k0) {
    oka = sa.load(Abase, k0, kend, ra);
    okb = sb.load(Bbase, k0, kend, rb);
  };
  auto store_stage = [&]() {
    if (g.a_act != MMVAE_ACT_NONE) act_inplace(ra, g.a_act);
    if (g.b_act != MMVAE_ACT_NONE) act_inplace(rb, g.b_act);
    sa.store(As, ra, oka);
    sb.store(Bs, rb, okb);
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float asum = 0.f;

  const int a_off = (KSPLIT == 1 ? wave * 32 : 0) + li;   // m within the staged tile
  const int k_off = (KSPLIT == 1 ? 0 : wave * 32) + lh;   // k within the staged tile

  if (kbeg < kend) load_stage(kbeg);
#pragma unroll 1
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    store_stage();
    __syncthreads();
    if (k0 + BK < kend) load_stage(k0 + BK);
#pragma unroll
    for (int kk = 0; kk < KW; kk += 2) {
      const float a = As[(k_off + kk) * AP + a_off];
      const float b = Bs[(k_off + kk) * BP + li];
      asum += a;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }

  gemm_finish<KSPLIT>(g, bx, m0, n0, bz, nz, acc, asum, As, Bs);
}

template <int KSPLIT, bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(KSPLIT == 8 ? 512 : 256) void gemm_kernel(GemmArgs g) {
  constexpr int BM = KSPLIT == 1 ? 128 : 32;
  constexpr int BK = KSPLIT == 1 ? GEMM_BK1 : 32 * KSPLIT;
  __shared__ float As[BK * (BM + 1)];
  __shared__ float Bs[BK * 33];
  gemm_body<KSPLIT, A_KMAJOR, B_KMAJOR>(g, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.z, As, Bs);
}

// Several independent GEMMs in ONE launch (same tiling): workgroup id -> (problem, tile).  Used for the data- and
// weight-gradient GEMMs of a Linear layer, which only share their inputs: each alone fills a fraction of the chip.
// Problem 0 is the data gradient (A k-major, B n-major), problem 1 the weight gradient (A m-major, B n-major).
#define GEMM_GROUP_MAX 2
struct GemmGroup {
  GemmArgs g[GEMM_GROUP_MAX];
  int blk0[GEMM_GROUP_MAX + 1];
  int nx[GEMM_GROUP_MAX], ny[GEMM_GROUP_MAX], nz[GEMM_GROUP_MAX];
  int n;
};
__global__ __launch_bounds__(256) void gemm_grouped_kernel(GemmGroup grp) {
  MMVAE_TRACE_STAMP(24);
  __shared__ float As[128 * 33];
  __shared__ float Bs[128 * 33];
  const int p = ((int)blockIdx.x >= grp.blk0[1]) ? 1 : 0;
  const int local = blockIdx.x - grp.blk0[p];
  const int bx = local % grp.nx[p], t = local / grp.nx[p];
  const int by = t % grp.ny[p], bz = t / grp.ny[p];
  if (p == 0) gemm_body<4, true, false>(grp.g[0], bx, by, bz, grp.nz[0], As, Bs);
  else gemm_body<4, false, false>(grp.g[1], bx, by, bz, grp.nz[1], As, Bs);
}

// ---- register-operand body (batch-128 linears) -------------------------------------------------------------
// With <= 256 output rows a Linear layer is a handful of 32x32 tiles with a 512-deep reduction: the staged body
// above spends its time in the global -> register -> LDS -> register round trip and two barriers per stage, not in
// its 16 MFMAs.  fp32 MFMA wants ONE row per lane (A[i = li][k], B[k][j = li]) so here every lane loads its own
// operand slots straight from L2 into registers -- float4 along k for k-contiguous operands, coalesced dwords for
// row-contiguous ones -- all of a slice's loads in flight at once, no LDS and no barrier before the MFMAs.  The 8
// waves of a workgroup take interleaved DEPTH-deep slices of the reduction; gemm_finish sums them.
// Reduction slot of lane (li, lh), register (q, j): k = k0 + 8 q + 4 lh + j.
// Requires (checked by the dispatcher): k-contiguous operands 16-byte aligned with K and the row stride % 4 == 0.
template <int DEPTH, bool KMAJOR>
__device__ __forceinline__ void rgemm_load(const float* __restrict__ rowp, const long kstride, const int k0,
                                           const int kend, const int K, const int lh, float (&v)[DEPTH / 8][4]) {
#pragma unroll
  for (int q = 0; q < DEPTH / 8; ++q) {
    const int kq = k0 + 8 * q + 4 * lh;
    if (KMAJOR) {
      const float4 t = *reinterpret_cast<const float4*>(rowp + min(kq, K - 4));
      const bool ok = kq < kend;
      v[q][0] = ok ? t.x : 0.f; v[q][1] = ok ? t.y : 0.f; v[q][2] = ok ? t.z : 0.f; v[q][3] = ok ? t.w : 0.f;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float t = rowp[(long)min(kq + j, K - 1) * kstride];
        v[q][j] = kq + j < kend ? t : 0.f;
      }
    }
  }
}

template <int DEPTH, bool A_KMAJOR, bool B_KMAJOR>
__device__ __forceinline__ void rgemm_body(const GemmArgs& g, const int bx, const int by, const int bz, const int nz,
                                           float* __restrict__ red, float* __restrict__ rsr) {
  constexpr int NW = 8, NQ = DEPTH / 8;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int m0 = by * 32, n0 = bx * 32;
  const int kbeg = bz * g.kper;
  const int kend = min(g.K, kbeg + g.kper);
  // rows / columns past the edge are clamped, not zeroed: they only feed accumulator rows / columns that are never
  // written.  The reduction tail is zeroed.
  const float* Ap = g.A + (long)min(m0 + li, g.M - 1) * g.sam;
  const float* Bp = g.B + (long)min(n0 + li, g.N - 1) * g.sbn;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float asum = 0.f;
#ifdef GEMM_PROBE
  long stamps[8];
  int nst = 0;
#define GSTAMP() stamps[nst++] = clock64()
#else
#define GSTAMP()
#endif
  GSTAMP();
#pragma unroll 1
  for (int k0 = kbeg + wave * DEPTH; k0 < kend; k0 += NW * DEPTH) {
    float a[NQ][4], b[NQ][4];
    rgemm_load<DEPTH, A_KMAJOR>(Ap, g.sak, k0, kend, g.K, lh, a);
    rgemm_load<DEPTH, B_KMAJOR>(Bp, g.sbk, k0, kend, g.K, lh, b);
    GSTAMP();
#ifdef GEMM_PROBE
    __builtin_amdgcn_s_waitcnt(0);
    GSTAMP();
#endif
    if (g.a_act != MMVAE_ACT_NONE) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) act_inplace(a[q], g.a_act);
    }
    if (g.b_act != MMVAE_ACT_NONE) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) act_inplace(b[q], g.b_act);
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        asum += a[q][j];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][j], b[q][j], acc, 0, 0, 0);
      }
  }
  GSTAMP();
  gemm_finish<NW>(g, bx, m0, n0, bz, nz, acc, asum, red, rsr);
#ifdef GEMM_PROBE
  GSTAMP();
  if (g.ws && lane == 0) {
    long* o = reinterpret_cast<long*>(g.ws) + ((long)(by * gridDim.x + bx) * NW + wave) * 8;
    for (int i = 0; i < 8; ++i) o[i] = i < nst ? stamps[i] : 0;
  }
#endif
}

// 16 x 16 tiles on v_mfma_f32_16x16x4_f32 for outputs of <= 128 32x32 tiles (a 128 x 512 layer: 256 workgroups
// instead of 64, so the MFMA phase and the operand loads spread over the whole chip).  The texture unit walks about
// one 128-byte line per clock per CU: a float4 load of 16 rows x 64 contiguous bytes touches 16 lines per kilobyte
// where the 32-row form touches 32, and a workgroup only pulls 64 KB.
// Operand layout: lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15]; accumulator register r of
// lane l is C[4 (l >> 4) + r][l & 15].  Reduction slot of lane (l & 15, kq = l >> 4), register (q, j):
// k = k0 + 16 q + 4 kq + j.
template <int DEPTH, bool KMAJOR>
__device__ __forceinline__ void rgemm16_load(const float* __restrict__ rowp, const long kstride, const int k0,
                                             const int kend, const int K, const int kq4, float (&v)[DEPTH / 16][4]) {
#pragma unroll
  for (int q = 0; q < DEPTH / 16; ++q) {
    const int kq = k0 + 16 * q + 4 * kq4;
    if (KMAJOR) {
      const float4 t = *reinterpret_cast<const float4*>(rowp + min(kq, K - 4));
      const bool ok = kq < kend;
      v[q][0] = ok ? t.x : 0.f; v[q][1] = ok ? t.y : 0.f; v[q][2] = ok ? t.z : 0.f; v[q][3] = ok ? t.w : 0.f;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float t = rowp[(long)min(kq + j, K - 1) * kstride];
        v[q][j] = kq + j < kend ? t : 0.f;
      }
    }
  }
}

template <int DEPTH, bool A_KMAJOR, bool B_KMAJOR>
__device__ __forceinline__ void rgemm16_body(const GemmArgs& g, const int bx, const int by, float* __restrict__ red) {
  constexpr int NW = 8, NQ = DEPTH / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, kq4 = lane >> 4;
  const int m0 = by * 16, n0 = bx * 16;
  const int kend = g.K;
  const float* Ap = g.A + (long)min(m0 + l16, g.M - 1) * g.sam;
  const float* Bp = g.B + (long)min(n0 + l16, g.N - 1) * g.sbn;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int k0 = wave * DEPTH; k0 < kend; k0 += NW * DEPTH) {
    float a[NQ][4], b[NQ][4];
    rgemm16_load<DEPTH, A_KMAJOR>(Ap, g.sak, k0, kend, g.K, kq4, a);
    rgemm16_load<DEPTH, B_KMAJOR>(Bp, g.sbk, k0, kend, g.K, kq4, b);
    if (g.a_act != MMVAE_ACT_NONE) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) act_inplace(a[q], g.a_act);
    }
    if (g.b_act != MMVAE_ACT_NONE) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) act_inplace(b[q], g.b_act);
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q][j], b[q][j], acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[(wave * 4 + r) * 64 + lane] = acc[r];
  __syncthreads();
  if (tid < 256) {
    const int r = tid >> 6;
    float v = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < NW; ++w2) v += red[(w2 * 4 + r) * 64 + lane];
    const int row = m0 + 4 * kq4 + r, col = n0 + l16;
    if (row < g.M && col < g.N) {
      if (g.bias) v += g.bias[col];
      const long o = (long)row * g.ldc + col;
      const int ep = g.ep;
      if (ep != MMVAE_EP_NONE || g.accumulate) {
        float av = 0.f;
        if (ep_reads_aux(ep)) av = g.aux[o];
        if (ep == MMVAE_EP_GELU && g.aux) g.aux[o] = v;
        v = apply_epilogue(v, av, ep);
        if (g.accumulate) v += g.C[o];
      }
      g.C[o] = v;
    }
  }
}

template <int DEPTH, bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(512) void rgemm16_kernel(GemmArgs g) {
  MMVAE_TRACE_STAMP(25);
  __shared__ float red[8 * 4 * 64];
  rgemm16_body<DEPTH, A_KMAJOR, B_KMAJOR>(g, blockIdx.x, blockIdx.y, red);
}

template <int DEPTH, bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(512) void rgemm_kernel(GemmArgs g) {
  MMVAE_TRACE_STAMP(26);
  __shared__ float red[8 * 16 * 64];
  __shared__ float rsr[8 * 32];
  rgemm_body<DEPTH, A_KMAJOR, B_KMAJOR>(g, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.z, red, rsr);
}

// data + weight gradient of one Linear layer in one launch (see gemm_grouped_kernel), register-operand bodies
template <int D0, int D1, bool T16>
__global__ __launch_bounds__(512) void rgemm_grouped_kernel(GemmGroup grp) {
  MMVAE_TRACE_STAMP(27);
  __shared__ float red[8 * 16 * 64];
  __shared__ float rsr[8 * 32];
  const int p = ((int)blockIdx.x >= grp.blk0[1]) ? 1 : 0;
  const int local = blockIdx.x - grp.blk0[p];
  const int bx = local % grp.nx[p], t = local / grp.nx[p];
  const int by = t % grp.ny[p], bz = t / grp.ny[p];
  if (p == 0) {
    if (T16) rgemm16_body<D0, true, false>(grp.g[0], bx, by, red);
    else rgemm_body<D0, true, false>(grp.g[0], bx, by, bz, grp.nz[0], red, rsr);
  } else {
    rgemm_body<D1, false, false>(grp.g[1], bx, by, bz, grp.nz[1], red, rsr);
  }
}

static inline bool rgemm_aligned(const float* p, long row_stride, int K) {
  return ((uintptr_t)p & 15) == 0 && (row_stride & 3) == 0 && (K & 3) == 0 && K >= 4;
}
static inline int rgemm_depth(int K) { return K >= 384 ? 64 : 16; }
static bool rgemm_enabled() { return true; }


// ------------------------------------------------------------------------------------------------
// Large GEMMs (round 2: the MNIST MLP towers at M = K*B = 7680 rows, the ResNet-50 tower's NHWC convolutions, the
// text towers' weight gradients at large batches): 128 x BN output tile per workgroup (BN = 128 | 64), 4 waves as 2 x 2,
// each wave a 64 x BN/2 block = 2 x (BN/64) accumulators of v_mfma_f32_32x32x2_f32; K staged 16 deep, double
// buffered in LDS as [k][m] / [k][n] (pitch +4: the transposing dword stores of a k-contiguous operand and the
// b128 stores of a row-contiguous one are both conflict free, fragment reads are consecutive lanes = consecutive
// banks); operands come in as float4 (all of a stage's loads in flight, next stage's loads issued before this
// stage's MFMAs).  Same GemmArgs / split-K partial layout / epilogue semantics as gemm_kernel.
// Requires 16-byte aligned operands along their contiguous axis (the dispatcher checks; else the staged kernel runs).
// ------------------------------------------------------------------------------------------------
// Round 3: BM is a template parameter too.  128 x (128 | 64) for the big grids; 64 x 64 (one accumulator per wave) for
// the batch-row linears at mid-size batches (M = 512 .. 2000 rows x 512 columns: 128 workgroups at M = 1000, where the
// register-operand kernel -- built for M <= 256 -- ran at 10 TFLOP/s and 128-row tiles would leave 3/4 of the chip idle).
#define GB_BK 16
template <int BM, int BN, bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(256) void gemm_big_kernel(GemmArgs g) {
  constexpr int BK = GB_BK;
  constexpr int AP = BM + 4, BP = BN + 4;
  constexpr int TM = BM / 64;                 // 32-row MFMA tiles per wave along M (2 | 1)
  constexpr int TN = BN / 64;                 // 32-column MFMA tiles per wave along N
  constexpr int NA4 = BM * BK / 4 / 256;      // float4 of the A tile per thread (2 | 1)
  constexpr int NB4 = BN * BK / 4 / 256;      // float4 of the B tile per thread (2 | 1)
  __shared__ __attribute__((aligned(16))) float As[2][BK * AP];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * BP];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int nz = gridDim.z, bz = blockIdx.z;
  const int kbeg = bz * g.kper, kend = min(g.K, kbeg + g.kper);

  // ---- staging slots.  k-contiguous operand: float4 = 4 k of one row (row = slot % R, k4 = slot / R);
  //      row-contiguous operand: float4 = 4 rows of one k (r4 = slot % (R/4), k = slot / (R/4))
  long a_off[NA4];
  int a_lds[NA4], a_k[NA4];
  bool a_ok[NA4];
#pragma unroll
  for (int i = 0; i < NA4; ++i) {
    const int q = tid + 256 * i;
    if (A_KMAJOR) {
      const int m = q % BM, k4 = q / BM;
      a_k[i] = 4 * k4;
      a_ok[i] = m0 + m < g.M;
      a_off[i] = (long)(m0 + (a_ok[i] ? m : 0)) * g.sam + 4 * k4;
      a_lds[i] = 4 * k4 * AP + m;
    } else {
      const int m4 = q % (BM / 4), k = q / (BM / 4);
      a_k[i] = k;
      a_ok[i] = m0 + 4 * m4 < g.M;            // M % 4 == 0 (dispatcher): a float4 is all in or all out
      a_off[i] = (long)k * g.sak + m0 + (a_ok[i] ? 4 * m4 : 0);
      a_lds[i] = k * AP + 4 * m4;
    }
  }
  long b_off[NB4];
  int b_lds[NB4], b_k[NB4];
  bool b_ok[NB4];
#pragma unroll
  for (int i = 0; i < NB4; ++i) {
    const int q = tid + 256 * i;
    if (B_KMAJOR) {
      const int n = q % BN, k4 = q / BN;
      b_k[i] = 4 * k4;
      b_ok[i] = n0 + n < g.N;
      b_off[i] = (long)(n0 + (b_ok[i] ? n : 0)) * g.sbn + 4 * k4;
      b_lds[i] = 4 * k4 * BP + n;
    } else {
      const int n4 = q % (BN / 4), k = q / (BN / 4);
      b_k[i] = k;
      b_ok[i] = n0 + 4 * n4 < g.N;
      b_off[i] = (long)k * g.sbk + n0 + (b_ok[i] ? 4 * n4 : 0);
      b_lds[i] = k * BP + 4 * n4;
    }
  }
  float4 ra[NA4], rb[NB4];
  unsigned oka = 0, okb = 0;
  auto load_stage = [&](int k0) {
    oka = okb = 0;
#pragma unroll
    for (int i = 0; i < NA4; ++i) {
      // k-contiguous: the float4 covers k0+a_k .. +3 (K % 4 == 0: all in or all out); row-contiguous: one k
      const bool ok = a_ok[i] && (k0 + a_k[i] < kend);
      oka |= (ok ? 1u : 0u) << i;
      const long o = ok ? (A_KMAJOR ? a_off[i] + k0 : a_off[i] + (long)k0 * g.sak) : 0;
      ra[i] = *reinterpret_cast<const float4*>(g.A + o);
    }
#pragma unroll
    for (int i = 0; i < NB4; ++i) {
      const bool ok = b_ok[i] && (k0 + b_k[i] < kend);
      okb |= (ok ? 1u : 0u) << i;
      const long o = ok ? (B_KMAJOR ? b_off[i] + k0 : b_off[i] + (long)k0 * g.sbk) : 0;
      rb[i] = *reinterpret_cast<const float4*>(g.B + o);
    }
  };
  auto act4 = [&](float4& v, int act) {
    float t[4] = {v.x, v.y, v.z, v.w};
    act_inplace(t, act);
    v = make_float4(t[0], t[1], t[2], t[3]);
  };
  auto store_stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NA4; ++i) {
      float4 v = (oka >> i & 1u) ? ra[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      if (g.a_act != MMVAE_ACT_NONE) act4(v, g.a_act);
      float* d = As[buf] + a_lds[i];
      if (A_KMAJOR) { d[0] = v.x; d[AP] = v.y; d[2 * AP] = v.z; d[3 * AP] = v.w; }
      else *reinterpret_cast<float4*>(d) = v;
    }
#pragma unroll
    for (int i = 0; i < NB4; ++i) {
      float4 v = (okb >> i & 1u) ? rb[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      if (g.b_act != MMVAE_ACT_NONE) act4(v, g.b_act);
      float* d = Bs[buf] + b_lds[i];
      if (B_KMAJOR) { d[0] = v.x; d[BP] = v.y; d[2 * BP] = v.z; d[3 * BP] = v.w; }
      else *reinterpret_cast<float4*>(d) = v;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float asum = 0.f;                                   // row sums of act(A) (the bias gradient when A = dy^T)
  const bool want_rs = g.a_rowsum != nullptr && blockIdx.x == 0;

  int buf = 0;
  if (kbeg < kend) {
    load_stage(kbeg);
    store_stage(0);
  }
  __syncthreads();
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    const bool more = k0 + BK < kend;
    if (more) load_stage(k0 + BK);
    const float* __restrict__ a_s = As[buf] + lh * AP + wm * (BM / 2) + li;
    const float* __restrict__ b_s = Bs[buf] + lh * BP + wn * (BN / 2) + li;
    // operands of k pair kp+1 are read from LDS while the MFMAs of pair kp issue (hipcc otherwise sinks every ds_read
    // next to its consumer, see conv_gather.inc)
    float av[2][TM], bv[2][TN];
    auto load_ops = [&](int kp, int ob) {
#pragma unroll
      for (int i = 0; i < TM; ++i) av[ob][i] = a_s[2 * kp * AP + 32 * i];
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[ob][j] = b_s[2 * kp * BP + 32 * j];
    };
    load_ops(0, 0);
#pragma unroll
    for (int kp = 0; kp < BK / 2; ++kp) {
      if (kp + 1 < BK / 2) load_ops(kp + 1, (kp + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kp & 1][i], bv[kp & 1][j], acc[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (want_rs && tid < BM) {
#pragma unroll
      for (int k = 0; k < BK; ++k) asum += As[buf][k * AP + tid];
    }
    if (more) store_stage(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  // ---- output: bias / epilogue / accumulate, or the raw partial of this K split ----
  const bool partial = nz > 1;
  float* Cout = partial ? g.ws + (size_t)bz * g.M * g.N : g.C;
  const long ldc = partial ? g.N : g.ldc;
  const int ep = partial ? MMVAE_EP_NONE : g.ep;
  const bool acc_out = !partial && g.accumulate;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn * (BN / 2) + 32 * j + li;
    const float bias_v = (!partial && g.bias && col < g.N) ? g.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * (BM / 2) + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < g.M && col < g.N) {
          float v = acc[i][j][r] + bias_v;
          const long o = (long)row * ldc + col;
          if (ep != MMVAE_EP_NONE || acc_out) {
            float av = 0.f;
            if (ep_reads_aux(ep)) av = g.aux[o];
            if (ep == MMVAE_EP_GELU && g.aux) g.aux[o] = v;
            v = apply_epilogue(v, av, ep);
            if (acc_out) v += Cout[o];
          }
          Cout[o] = v;
        }
      }
    }
  }
  if (want_rs && tid < BM && m0 + tid < g.M) {
    float* rs_out = partial ? g.ws + (size_t)nz * g.M * g.N + (size_t)bz * g.M : g.a_rowsum;
    rs_out[m0 + tid] = acc_out ? rs_out[m0 + tid] + asum : asum;
  }
}

// shapes the large-tile kernel takes: enough 128-row tiles to put >= 2 workgroups on every CU, float4-loadable
// operands.  Returns the N tile (128 | 64) or 0.
static inline int gemm_big_bn(const GemmArgs& g, int nz, bool ak, bool bk_major) {
  if (g.K < 16 || g.N < 32 || g.M < 128) return 0;
  auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
  if (!al16(g.A) || !al16(g.B)) return 0;
  if (ak) { if ((g.sam & 3) || (g.K & 3) || (g.kper & 3)) return 0; }
  else { if ((g.sak & 3) || (g.M & 3)) return 0; }
  if (bk_major) { if ((g.sbn & 3) || (g.K & 3) || (g.kper & 3)) return 0; }
  else { if ((g.sbk & 3) || (g.N & 3)) return 0; }
  const long rows = (g.M + 127) / 128;
  if (rows * ((g.N + 127) / 128) * nz >= 512 && g.N > 64) return 128;
  if (rows * ((g.N + 63) / 64) * nz >= 384) return 64;
  // (A 64 x 64-tile variant for mid-size problems past the register-operand regime was measured in round 3 and removed in
  // round 4: 1000 x 512 x 512 took 34 us on it against 17.7 us on the register-operand kernel.)
  return 0;
}
static void gemm_big_launch(const GemmArgs& g, int bn, int nz, bool ak, bool bk_major, hipStream_t st) {
  constexpr int bm = 128;
  const dim3 grid((g.N + bn - 1) / bn, (g.M + bm - 1) / bm, nz);
#define GB_LAUNCH(BM, BN)                                                                                   \
  do {                                                                                                      \
    if (ak && bk_major) hipLaunchKernelGGL((gemm_big_kernel<BM, BN, true, true>), grid, dim3(256), 0, st, g);   \
    else if (ak) hipLaunchKernelGGL((gemm_big_kernel<BM, BN, true, false>), grid, dim3(256), 0, st, g);         \
    else if (!bk_major) hipLaunchKernelGGL((gemm_big_kernel<BM, BN, false, false>), grid, dim3(256), 0, st, g); \
    else hipLaunchKernelGGL((gemm_big_kernel<BM, BN, false, true>), grid, dim3(256), 0, st, g);                 \
  } while (0)
  if (bn == 128) GB_LAUNCH(128, 128);
  else GB_LAUNCH(128, 64);
#undef GB_LAUNCH
}

// tiling variant, reduction length per split and number of splits actually used for (M, N, K, requested splitk)
static int gemm_split_plan(int M, int N, int K, int splitk, int* variant_out, int* kper_out) {
  if (splitk < 1) splitk = 1;
  const int ntn = (N + 31) / 32;
  const long tiles128 = (long)((M + 127) / 128) * ntn, tiles32 = (long)((M + 31) / 32) * ntn;
  int variant = 4;
  if (tiles128 >= 256) variant = 1;   // measured: below that the single-stage 32x32 tiles are faster
  else if (K >= 384 && tiles32 <= 512 && splitk == 1) variant = 8;
  const int bk = variant == 1 ? GEMM_BK1 : 32 * variant;
  int kper = (K + splitk - 1) / splitk;
  kper = (kper + bk - 1) / bk * bk;
  if (variant_out) *variant_out = variant;
  if (kper_out) *kper_out = kper;
  return (K + kper - 1) / kper;
}
extern "C" int mmvae_gemm_splits(int M, int N, int K, int splitk) { return gemm_split_plan(M, N, K, splitk, nullptr, nullptr); }

extern "C" size_t mmvae_gemm_ws_floats(int M, int N, int splitk) {
  return splitk > 1 ? (size_t)splitk * ((size_t)M * N + M) : 0;
}

extern "C" int mmvae_gemm_f32(const float* A, const float* Bm, const float* bias, const float* aux, float* C,
                              float* a_rowsum, float* ws, int M, int N, int K, long sam, long sak, long sbk, long sbn,
                              long ldc, int a_act, int b_act, int ep_mode, int accumulate, int splitk,
                              mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(A && Bm && C && M > 0 && N > 0 && K > 0);
  if (splitk < 1) splitk = 1;
  if (splitk > 1 && (!ws || bias || ep_mode != MMVAE_EP_NONE)) return MMVAE_ERR_ARG;
  if (ep_reads_aux(ep_mode) && !aux) return MMVAE_ERR_ARG;
  GemmArgs g;
  g.A = A; g.B = Bm; g.bias = bias; g.aux = const_cast<float*>(aux); g.C = C; g.a_rowsum = a_rowsum; g.ws = ws;
  g.M = M; g.N = N; g.K = K; g.sam = sam; g.sak = sak; g.sbk = sbk; g.sbn = sbn; g.ldc = ldc;
  g.a_act = a_act; g.b_act = b_act; g.ep = ep_mode;
  g.accumulate = accumulate ? 1 : 0;   // DEFER without a split == plain accumulation
  // wide layers at large row counts: the LDS-tiled split-bf16 kernel (gemm_b16.inc); forward and data gradient
  if (splitk == 1 && !a_rowsum && ep_mode != MMVAE_EP_GELU && gb_act_ok(a_act) && gb_act_ok(b_act) && gb_shape_ok(M, N, K) &&
      M <= G16_MAX_ROWS) {
    const bool akc = sak == 1, bkc = (sbk == 1 && sbn != 1);
    const long lda = akc ? sam : sak, ldb = bkc ? sbn : sbk;
    const bool a_ok = (akc || sam == 1) && gb_al16(A) && (lda & 3) == 0 && (akc || (M & 3) == 0);
    const bool b_ok = (bkc || sbn == 1) && gb_al16(Bm) && (ldb & 3) == 0;
    if (a_ok && b_ok && (!akc || !bkc || true)) {
      GbArgs b{A, Bm, bias, aux, C, nullptr, nullptr, M, N, K, lda, ldb, ldc, a_act, b_act, ep_mode, g.accumulate, K};
      const dim3 grid((N + G16_TM - 1) / G16_TM, (M + G16_TM - 1) / G16_TM, 1);
      hipStream_t st = (hipStream_t)stream;
      // fewer tiles than CUs: two k-groups per tile (twice the waves, half the chain)
      const bool ks2 = (long)grid.x * grid.y < 256 && K >= 4 * G16_BK;
#define G16_LAUNCH(AK_, BK_)                                                                                   \
  do {                                                                                                         \
    if (ks2) hipLaunchKernelGGL((gemm_b16_kernel<AK_, BK_, 2>), grid, dim3(512), 0, st, b);                     \
    else hipLaunchKernelGGL((gemm_b16_kernel<AK_, BK_, 1>), grid, dim3(256), 0, st, b);                         \
  } while (0)
      if (akc && bkc) G16_LAUNCH(true, true);
      else if (akc) G16_LAUNCH(true, false);
      else if (bkc) G16_LAUNCH(false, true);
      else G16_LAUNCH(false, false);
#undef G16_LAUNCH
      return mmvae_launch_status();
    }
  }
  const int ntn = (N + 31) / 32;
  // Tiling choice (all that matters at batch 128 is the length of the serial load -> MFMA chain per workgroup):
  //   K <= 128 and many rows : 128x32 tiles, one stage, no cross-wave reduction            (KSPLIT 1)
  //   K >= 384, few tiles    : 32x32 tile, 8 waves x 256-deep stages (half the serial stages)  (KSPLIT 8)
  //   otherwise              : 32x32 tile, 4 waves x 128-deep stages                       (KSPLIT 4)
  const long tiles128 = (long)((M + 127) / 128) * ntn, tiles32 = (long)((M + 31) / 32) * ntn;
  int variant, kper;
  const int nz = gemm_split_plan(M, N, K, splitk, &variant, &kper);
  g.kper = kper;
  if (nz > 1 && !ws) return MMVAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  // operand orientation is a template argument (per-slot address stride folds to a constant): the three forms a
  // Linear layer needs are  fwd: A k-major, B k-major;  dgrad: A k-major, B n-major;  wgrad: A m-major, B n-major
  const bool ak = sak == 1, bk_major = (sbk == 1 && sbn != 1);
  if (!ak && sam != 1) return MMVAE_ERR_UNSUPPORTED;
  if (!bk_major && sbn != 1) return MMVAE_ERR_UNSUPPORTED;
  // large problems: 128 x 128 | 64 tiles, float4 staging (same kper / partial layout)
  if (const int big_bn = gemm_big_bn(g, nz, ak, bk_major)) {
    gemm_big_launch(g, big_bn, nz, ak, bk_major, st);
    int rc = mmvae_launch_status();
    if (rc) return rc;
    if (nz > 1 && accumulate != MMVAE_ACC_DEFER) {
      rc = mmvae_reduce_rows(ws, C, nz, (long)M * N, (long)M * N, accumulate, stream);
      if (rc) return rc;
      if (a_rowsum) rc = mmvae_reduce_rows(ws + (size_t)nz * M * N, a_rowsum, nz, M, M, accumulate, stream);
    }
    return rc;
  }
  // split reduction over workgroups with few output tiles (the text towers' (L*N)-row weight gradients: 162 x 54
  // outputs, 4096 rows): same kper / partial layout as the staged kernel, register-operand body
  const bool rsplit = true;
  if (rgemm_enabled() && rsplit && nz > 1 && tiles32 <= 64 && kper <= 1024 && !ak && !bk_major) {
    const dim3 rgrid(ntn, (M + 31) / 32, nz);
    if (rgemm_depth(kper) == 64) hipLaunchKernelGGL((rgemm_kernel<64, false, false>), rgrid, dim3(512), 0, st, g);
    else hipLaunchKernelGGL((rgemm_kernel<16, false, false>), rgrid, dim3(512), 0, st, g);
    int rc = mmvae_launch_status();
    if (rc) return rc;
    if (accumulate != MMVAE_ACC_DEFER) {
      rc = mmvae_reduce_rows(ws, C, nz, (long)M * N, (long)M * N, accumulate, stream);
      if (rc) return rc;
      if (a_rowsum) rc = mmvae_reduce_rows(ws + (size_t)nz * M * N, a_rowsum, nz, M, M, accumulate, stream);
    }
    return rc;
  }
  // few tiles, short reduction: register-operand kernel (no LDS staging)
  if (rgemm_enabled() && splitk == 1 && tiles32 <= 1024 && K <= 1024) {
    // k-contiguous operands that are not float4-aligned take the strided (dword) loads with a k stride of 1
    const bool ak4 = ak && rgemm_aligned(A, sam, K), bk4 = bk_major && rgemm_aligned(Bm, sbn, K);
    g.kper = K;
    const dim3 rgrid(ntn, (M + 31) / 32, 1);
#define RGEMM_LAUNCH(D)                                                                                  \
  do {                                                                                                   \
    if (ak4 && bk4) hipLaunchKernelGGL((rgemm_kernel<D, true, true>), rgrid, dim3(512), 0, st, g);        \
    else if (ak4) hipLaunchKernelGGL((rgemm_kernel<D, true, false>), rgrid, dim3(512), 0, st, g);         \
    else if (!bk4) hipLaunchKernelGGL((rgemm_kernel<D, false, false>), rgrid, dim3(512), 0, st, g);       \
    else hipLaunchKernelGGL((rgemm_kernel<D, false, true>), rgrid, dim3(512), 0, st, g);                  \
  } while (0)
    if (tiles32 <= 128 && !a_rowsum) {   // 16 x 16 tiles
      const dim3 grid16((N + 15) / 16, (M + 15) / 16, 1);
#define RGEMM16_LAUNCH(D)                                                                                   \
  do {                                                                                                      \
    if (ak4 && bk4) hipLaunchKernelGGL((rgemm16_kernel<D, true, true>), grid16, dim3(512), 0, st, g);        \
    else if (ak4) hipLaunchKernelGGL((rgemm16_kernel<D, true, false>), grid16, dim3(512), 0, st, g);         \
    else if (!bk4) hipLaunchKernelGGL((rgemm16_kernel<D, false, false>), grid16, dim3(512), 0, st, g);       \
    else hipLaunchKernelGGL((rgemm16_kernel<D, false, true>), grid16, dim3(512), 0, st, g);                  \
  } while (0)
      if (rgemm_depth(K) == 64) RGEMM16_LAUNCH(64);
      else RGEMM16_LAUNCH(16);
#undef RGEMM16_LAUNCH
      return mmvae_launch_status();
    }
    if (rgemm_depth(K) == 64) RGEMM_LAUNCH(64);
    else RGEMM_LAUNCH(16);
#undef RGEMM_LAUNCH
    return mmvae_launch_status();
  }
  const dim3 grid(ntn, variant == 1 ? (M + 127) / 128 : (M + 31) / 32, nz), block(variant == 8 ? 512 : 256);
#define GEMM_LAUNCH(V)                                                                            \
  do {                                                                                            \
    if (ak && bk_major) hipLaunchKernelGGL((gemm_kernel<V, true, true>), grid, block, 0, st, g);   \
    else if (ak) hipLaunchKernelGGL((gemm_kernel<V, true, false>), grid, block, 0, st, g);         \
    else if (!bk_major) hipLaunchKernelGGL((gemm_kernel<V, false, false>), grid, block, 0, st, g); \
    else hipLaunchKernelGGL((gemm_kernel<V, false, true>), grid, block, 0, st, g);                 \
  } while (0)
  if (variant == 1) GEMM_LAUNCH(1);
  else if (variant == 8) GEMM_LAUNCH(8);
  else GEMM_LAUNCH(4);
#undef GEMM_LAUNCH
  int rc = mmvae_launch_status();
  if (rc) return rc;
  if (nz > 1 && accumulate != MMVAE_ACC_DEFER) {
    rc = mmvae_reduce_rows(ws, C, nz, (long)M * N, (long)M * N, accumulate, stream);
    if (rc) return rc;
    if (a_rowsum) rc = mmvae_reduce_rows(ws + (size_t)nz * M * N, a_rowsum, nz, M, M, accumulate, stream);
  }
  return rc;
}

// ---- nn.Linear wrappers ---------------------------------------------------------------------------
int mmvae_proj32_fwd_launch(const float* x, const float* w, const float* b, float* y, int M, int N, mmvae_stream_t stream);   // ffn.hip
extern "C" int mmvae_linear_fwd(const float* x, const float* w, const float* b, float* aux, float* y, int M, int N,
                                int K, long ldx, int x_act, int ep_mode, mmvae_stream_t stream) {
  // many rows of a 32-wide input into 32 .. 128 outputs (the action towers' QKV projection): one wave per 32 rows
  if (K == 32 && ldx == 32 && M >= 1024 && N >= 32 && N <= 128 && (N & 31) == 0 && x_act == MMVAE_ACT_NONE &&
      ep_mode == MMVAE_EP_NONE && x && w && y && ((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)b) & 15) == 0))
    return mmvae_proj32_fwd_launch(x, w, b, y, M, N, stream);
  // y[m,n] = sum_k x[m,k] w[n,k]: A = x (sam = ldx, sak = 1), B(k,n) = w[n*K + k]
  return mmvae_gemm_f32(x, w, b, aux, y, nullptr, nullptr, M, N, K, ldx, 1, 1, K, N, x_act, MMVAE_ACT_NONE, ep_mode, 0,
                        1, stream);
}
extern "C" int mmvae_linear_bwd_data(const float* dy, const float* w, const float* aux, float* dx, int M, int N,
                                     int K, int ep_mode, int accumulate, mmvae_stream_t stream) {
  // dx[m,k] = sum_n dy[m,n] w[n,k]: A = dy (M x N), B(n,k) = w[n*K + k]
  return mmvae_gemm_f32(dy, w, nullptr, aux, dx, nullptr, nullptr, M, K, N, N, 1, K, 1, K, MMVAE_ACT_NONE, MMVAE_ACT_NONE,
                        ep_mode, accumulate, 1, stream);
}
static int wgrad_splitk(int M, int N, int K) {
  // reduction length is M (rows); output N x K.  Split so that tiles * splits ~ 512 workgroups.
  if (M >= 2048 && N >= 128 && K >= 32 && !(N & 3) && !(K & 3)) {
    // long reductions into a sizeable output run on the large-tile kernel (128 x 64 tiles): splits for ITS grid
    const long big = (long)((N + 127) / 128) * ((K + 63) / 64);
    int s = (int)((512 + big - 1) / big);
    const int maxs = M / 256;
    if (s > maxs) s = maxs;
    if (s > 64) s = 64;
    if (s >= 1) return s;
  }
  const long tiles = (long)((N + 31) / 32) * ((K + 31) / 32);
  int s = (int)(512 / (tiles > 0 ? tiles : 1));
  const int maxs = (M + 127) / 128;
  if (s > maxs) s = maxs;
  if (s < 1) s = 1;
  if (s > 64) s = 64;
  return s;
}
extern "C" int mmvae_linear_bwd_weight_splits(int M, int N, int K) {
  // mirrors the kper rounding of mmvae_gemm_f32 for the (N x K) = dy^T x problem with reduction length M
  int sk = wgrad_splitk(M, N, K);
  // gemm problem: rows N, cols K, reduction M
  const long tiles128 = (long)((N + 127) / 128) * ((K + 31) / 32), tiles32 = (long)((N + 31) / 32) * ((K + 31) / 32);
  int variant = 4;
  if (tiles128 >= 256) variant = 1;
  else if (M >= 384 && tiles32 <= 512 && sk == 1) variant = 8;
  const int bk = variant == 1 ? GEMM_BK1 : 32 * variant;
  int kper = (M + sk - 1) / sk;
  kper = (kper + bk - 1) / bk * bk;
  return (M + kper - 1) / kper;
}
extern "C" size_t mmvae_linear_bwd_weight_ws_floats(int M, int N, int K) {
  return mmvae_gemm_ws_floats(N, K, wgrad_splitk(M, N, K));
}
extern "C" int mmvae_linear_bwd_weight(const float* dy, const float* x, float* dw, float* db, float* ws, int M, int N,
                                       int K, long ldx, int x_act, int accumulate, mmvae_stream_t stream) {
  // dw[n,k] = sum_m dy[m,n] act(x[m,k]): "A"(n,m) = dy[m*N + n] (sam = 1, sak = N), "B"(m,k) = x[m*ldx + k];
  // db[n] = row sum of "A".
  const int sk = wgrad_splitk(M, N, K);
  return mmvae_gemm_f32(dy, x, nullptr, nullptr, dw, db, ws, N, K, M, 1, N, ldx, 1, K, MMVAE_ACT_NONE, x_act,
                        MMVAE_EP_NONE, accumulate, sk, stream);
}

// ---- several weight gradients in ONE launch ------------------------------------------------------------------
// The text towers' fused layer leaves 4 (encoder) or 6 (decoder) independent (L*N)-row weight gradients behind it,
// each a handful of 32x32 tiles times its split count: 32 .. 128 workgroups and 6 - 16 us apiece when launched one
// after the other.  Here every job keeps the tiling, split plan and partial layout mmvae_linear_bwd_weight would
// give it (bit-identical results) and the workgroups of all jobs share one grid.
struct GemmBatch {
  GemmArgs g[MMVAE_WGRAD_BATCH_MAX];
  int blk0[MMVAE_WGRAD_BATCH_MAX + 1];
  int nx[MMVAE_WGRAD_BATCH_MAX], ny[MMVAE_WGRAD_BATCH_MAX], nz[MMVAE_WGRAD_BATCH_MAX];
  unsigned deep;   // bit p: job p reduces in 64-deep slices
  int n;
};
// ANY_DEEP = false: no job of the launch takes the 64-deep slices -- 60 VGPRs instead of 172, so the workgroups fit
// beside the conv kernels of the other stream instead of waiting for a whole free CU
template <bool ANY_DEEP>
__global__ __launch_bounds__(512) void rgemm_batch_kernel(GemmBatch bt) {
  MMVAE_TRACE_STAMP(39);
  __shared__ float red[8 * 16 * 64];
  __shared__ float rsr[8 * 32];
  int p = 0;
  for (int q = 1; q < bt.n; ++q) p = ((int)blockIdx.x >= bt.blk0[q]) ? q : p;
  p = __builtin_amdgcn_readfirstlane(p);
  const int local = blockIdx.x - bt.blk0[p];
  const int bx = local % bt.nx[p], t = local / bt.nx[p];
  const int by = t % bt.ny[p], bz = t / bt.ny[p];
  const GemmArgs g = bt.g[p];
  if (ANY_DEEP && (bt.deep >> p & 1u)) rgemm_body<64, false, false>(g, bx, by, bz, bt.nz[p], red, rsr);
  else rgemm_body<16, false, false>(g, bx, by, bz, bt.nz[p], red, rsr);
}

// fills `g` / grid of one job when it runs on the register-operand body with m-major dy and n-major x; false otherwise
static bool wgrad_batch_plan(const mmvae_wgrad_job_t& j, GemmArgs& g, int& nx, int& ny, int& nz, bool& deep) {
  if (!j.dy || !j.x || !j.dw || j.M <= 0 || j.N <= 0 || j.K <= 0) return false;
  const int sk = wgrad_splitk(j.M, j.N, j.K);
  // gemm problem: rows N, cols K, reduction M
  g.A = j.dy; g.B = j.x; g.bias = nullptr; g.aux = nullptr; g.C = j.dw; g.a_rowsum = j.db; g.ws = j.ws;
  g.M = j.N; g.N = j.K; g.K = j.M; g.sam = 1; g.sak = j.N; g.sbk = j.ldx; g.sbn = 1; g.ldc = j.K;
  g.a_act = MMVAE_ACT_NONE; g.b_act = j.x_act; g.ep = MMVAE_EP_NONE; g.accumulate = j.accumulate ? 1 : 0;
  int variant, kper;
  nz = gemm_split_plan(g.M, g.N, g.K, sk, &variant, &kper);
  g.kper = kper;
  if (nz > 1 && !j.ws) return false;
  if (gemm_big_bn(g, nz, false, false)) return false;
  nx = (g.N + 31) / 32; ny = (g.M + 31) / 32;
  const long tiles32 = (long)nx * ny;
  const bool rsplit = true;
  if (!rgemm_enabled()) return false;
  if (nz > 1) {
    if (!(rsplit && tiles32 <= 64 && kper <= 1024 && j.accumulate == MMVAE_ACC_DEFER)) return false;
    deep = rgemm_depth(kper) == 64;
    return true;
  }
  // one split: mmvae_gemm_f32 takes the same body unless it prefers 16 x 16 tiles (no row sums asked for)
  if (!(sk == 1 && tiles32 <= 1024 && g.K <= 1024) || (tiles32 <= 128 && !j.db)) return false;
  g.kper = g.K;
  deep = rgemm_depth(g.K) == 64;
  return true;
}

extern "C" int mmvae_linear_bwd_weight_batch(const mmvae_wgrad_job_t* jobs, int n_jobs, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(jobs && n_jobs > 0);
  bool ok = n_jobs >= 2 && n_jobs <= MMVAE_WGRAD_BATCH_MAX;
  GemmBatch bt;
  bt.deep = 0; bt.n = n_jobs; bt.blk0[0] = 0;
  for (int p = 0; ok && p < n_jobs; ++p) {
    bool deep = false;
    ok = wgrad_batch_plan(jobs[p], bt.g[p], bt.nx[p], bt.ny[p], bt.nz[p], deep);
    if (!ok) break;
    if (deep) bt.deep |= 1u << p;
    bt.blk0[p + 1] = bt.blk0[p] + bt.nx[p] * bt.ny[p] * bt.nz[p];
  }
  if (!ok) {   // any job outside that regime: one launch per job, as before
    for (int p = 0; p < n_jobs; ++p) {
      const mmvae_wgrad_job_t& j = jobs[p];
      const int rc = mmvae_linear_bwd_weight(j.dy, j.x, j.dw, j.db, j.ws, j.M, j.N, j.K, j.ldx, j.x_act, j.accumulate, stream);
      if (rc) return rc;
    }
    return MMVAE_OK;
  }
  if (bt.deep) hipLaunchKernelGGL(rgemm_batch_kernel<true>, dim3(bt.blk0[n_jobs]), dim3(512), 0, (hipStream_t)stream, bt);
  else hipLaunchKernelGGL(rgemm_batch_kernel<false>, dim3(bt.blk0[n_jobs]), dim3(512), 0, (hipStream_t)stream, bt);
  return mmvae_launch_status();
}

// Fused nn.Linear backward: dx = ep(dy W) and dW (+)= dy^T act(x), db (+)= colsum(dy) in ONE grouped launch.
// Falls back to two launches when either problem wants a different tiling.
// rows up to which the grouped backward runs on the register-operand bodies (round 4: 256 -> 1024; same box, ms/step of
// the cfg2 step at batch 512 / 1000: 0.996 -> 0.958, 1.660 -> 1.635; 2048 and 4096 measured level on the K-sample workloads)
constexpr int RGEMM_BWD_MAX_M = 1024;
static inline bool linear_bwd_rgemm(int M, int N) { return rgemm_enabled() && M <= RGEMM_BWD_MAX_M && (N & 3) == 0 && N >= 4; }
// the tiled split-bf16 grouped launch (gemm_b16.inc): data gradient (M x K, reduction N) + weight gradient (N x K, reduction M)
static inline bool linear_bwd_b16(int M, int N, int K) { return gb_shape_ok(M, K, N) && M <= G16_MAX_ROWS; }
extern "C" size_t mmvae_linear_bwd_ws_floats(int M, int N, int K) {
  if (linear_bwd_b16(M, N, K)) {
    const int nz = gb_wgrad_nz(M, N, K);
    return nz > 1 ? (size_t)nz * ((size_t)N * K + N) : 0;
  }
  return linear_bwd_rgemm(M, N) ? 0 : mmvae_linear_bwd_weight_ws_floats(M, N, K);
}
extern "C" int mmvae_linear_bwd_splits(int M, int N, int K) {
  if (linear_bwd_b16(M, N, K)) return gb_wgrad_nz(M, N, K);
  return linear_bwd_rgemm(M, N) ? 1 : mmvae_linear_bwd_weight_splits(M, N, K);
}
extern "C" int mmvae_linear_bwd(const float* dy, const float* x, const float* w, const float* aux, float* dx,
                                float* dw, float* db, float* ws, int M, int N, int K, long ldx, int x_act, int ep_mode,
                                int accumulate, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(dy && x && w && dx && dw && M > 0 && N > 0 && K > 0);
  if (ep_reads_aux(ep_mode) && !aux) return MMVAE_ERR_ARG;
  // problem 0 (data):   C[M,K] = dy[M,N] W[N,K]            reduction N
  // problem 1 (weight): C[N,K] = dy^T[N,M] act(x)[M,K]     reduction M, split over workgroups
  if (linear_bwd_b16(M, N, K)) {
    if (!gb_al16(dy) || !gb_al16(x) || !gb_al16(w) || (ldx & 3) != 0) return MMVAE_ERR_ARG;   // float4 staging
    const int nz = gb_wgrad_nz(M, N, K);
    if (nz > 1 && !ws) return MMVAE_ERR_ARG;
    GbGroup grp;
    grp.g[0] = GbArgs{dy, w, nullptr, aux, dx, nullptr, nullptr, M, K, N, (long)N, (long)K, (long)K, MMVAE_ACT_NONE,
                      MMVAE_ACT_NONE, ep_mode, 0, N};
    grp.g[1] = GbArgs{dy, x, nullptr, nullptr, dw, db, ws, N, K, M, (long)N, ldx, (long)K, MMVAE_ACT_NONE, x_act,
                      MMVAE_EP_NONE, accumulate ? 1 : 0, gb_wgrad_kper(M, gb_wgrad_splits(M, N, K))};
    grp.nx[0] = (K + G16_TM - 1) / G16_TM; grp.ny[0] = (M + G16_TM - 1) / G16_TM; grp.nz[0] = 1;
    grp.nx[1] = (K + G16_TM - 1) / G16_TM; grp.ny[1] = (N + G16_TM - 1) / G16_TM; grp.nz[1] = nz;
    grp.blk0[0] = 0;
    grp.blk0[1] = grp.nx[0] * grp.ny[0];
    grp.blk0[2] = grp.blk0[1] + grp.nx[1] * grp.ny[1] * nz;
    if (grp.blk0[2] < 256 && M >= 4 * G16_BK && N >= 4 * G16_BK)
      hipLaunchKernelGGL(gemm_b16_linear_bwd_kernel<2>, dim3(grp.blk0[2]), dim3(512), 0, (hipStream_t)stream, grp);
    else
      hipLaunchKernelGGL(gemm_b16_linear_bwd_kernel<1>, dim3(grp.blk0[2]), dim3(256), 0, (hipStream_t)stream, grp);
    int rc = mmvae_launch_status();
    if (rc) return rc;
    if (nz > 1 && accumulate != MMVAE_ACC_DEFER) {
      rc = mmvae_reduce_rows(ws, dw, nz, (long)N * K, (long)N * K, accumulate, stream);
      if (rc) return rc;
      if (db) rc = mmvae_reduce_rows(ws + (size_t)nz * N * K, db, nz, N, N, accumulate, stream);
    }
    return rc;
  }
  if (linear_bwd_rgemm(M, N)) {
    if (!rgemm_aligned(dy, N, N)) return MMVAE_ERR_ARG;   // this regime needs a 16-byte aligned dy
    GemmGroup grp;
    GemmArgs& gd = grp.g[0];
    gd.A = dy; gd.B = w; gd.bias = nullptr; gd.aux = const_cast<float*>(aux); gd.C = dx; gd.a_rowsum = nullptr; gd.ws = nullptr;
    gd.M = M; gd.N = K; gd.K = N; gd.sam = N; gd.sak = 1; gd.sbk = K; gd.sbn = 1; gd.ldc = K;
    gd.a_act = MMVAE_ACT_NONE; gd.b_act = MMVAE_ACT_NONE; gd.ep = ep_mode; gd.accumulate = 0; gd.kper = N;
    GemmArgs& gw = grp.g[1];
    gw.A = dy; gw.B = x; gw.bias = nullptr; gw.aux = nullptr; gw.C = dw; gw.a_rowsum = db; gw.ws = nullptr;
    gw.M = N; gw.N = K; gw.K = M; gw.sam = 1; gw.sak = N; gw.sbk = ldx; gw.sbn = 1; gw.ldc = K;
    gw.a_act = MMVAE_ACT_NONE; gw.b_act = x_act; gw.ep = MMVAE_EP_NONE; gw.accumulate = accumulate ? 1 : 0; gw.kper = M;
    grp.n = 2;
    const bool t16 = (long)((M + 31) / 32) * ((K + 31) / 32) <= 128;   // data gradient on 16 x 16 tiles
    const int td = t16 ? 16 : 32;
    grp.nx[0] = (K + td - 1) / td; grp.ny[0] = (M + td - 1) / td; grp.nz[0] = 1;
    grp.nx[1] = (K + 31) / 32; grp.ny[1] = (N + 31) / 32; grp.nz[1] = 1;
    grp.blk0[0] = 0;
    grp.blk0[1] = grp.nx[0] * grp.ny[0];
    grp.blk0[2] = grp.blk0[1] + grp.nx[1] * grp.ny[1];
    const dim3 rgrid(grp.blk0[2]);
    hipStream_t st = (hipStream_t)stream;
    const bool deep = rgemm_depth(N) == 64;
    if (deep && t16) hipLaunchKernelGGL((rgemm_grouped_kernel<64, 16, true>), rgrid, dim3(512), 0, st, grp);
    else if (deep) hipLaunchKernelGGL((rgemm_grouped_kernel<64, 16, false>), rgrid, dim3(512), 0, st, grp);
    else if (t16) hipLaunchKernelGGL((rgemm_grouped_kernel<16, 16, true>), rgrid, dim3(512), 0, st, grp);
    else hipLaunchKernelGGL((rgemm_grouped_kernel<16, 16, false>), rgrid, dim3(512), 0, st, grp);
    return mmvae_launch_status();
  }
  const long t128_d = (long)((M + 127) / 128) * ((K + 31) / 32), t32_d = (long)((M + 31) / 32) * ((K + 31) / 32);
  // (a 512-deep data gradient alone prefers the 8-wave tiling, but one grouped launch beats two: +4 % on the step)
  const bool d_ok = t128_d < 256;
  const int nz = mmvae_linear_bwd_weight_splits(M, N, K);
  const long t128_w = (long)((N + 127) / 128) * ((K + 31) / 32), t32_w = (long)((N + 31) / 32) * ((K + 31) / 32);
  const int sk = wgrad_splitk(M, N, K);
  const bool w_ok = t128_w < 256 && !(M >= 384 && t32_w <= 512 && sk == 1);
  if (!d_ok || !w_ok) {
    int rc = mmvae_linear_bwd_weight(dy, x, dw, db, ws, M, N, K, ldx, x_act, accumulate, stream);
    if (rc) return rc;
    return mmvae_linear_bwd_data(dy, w, aux, dx, M, N, K, ep_mode, 0, stream);
  }
  if (nz > 1 && !ws) return MMVAE_ERR_ARG;
  GemmGroup grp;
  GemmArgs& gd = grp.g[0];
  gd.A = dy; gd.B = w; gd.bias = nullptr; gd.aux = const_cast<float*>(aux); gd.C = dx; gd.a_rowsum = nullptr; gd.ws = nullptr;
  gd.M = M; gd.N = K; gd.K = N; gd.sam = N; gd.sak = 1; gd.sbk = K; gd.sbn = 1; gd.ldc = K;
  gd.a_act = MMVAE_ACT_NONE; gd.b_act = MMVAE_ACT_NONE; gd.ep = ep_mode; gd.accumulate = 0;
  gd.kper = (N + 127) / 128 * 128;
  GemmArgs& gw = grp.g[1];
  gw.A = dy; gw.B = x; gw.bias = nullptr; gw.aux = nullptr; gw.C = dw; gw.a_rowsum = db; gw.ws = ws;
  gw.M = N; gw.N = K; gw.K = M; gw.sam = 1; gw.sak = N; gw.sbk = ldx; gw.sbn = 1; gw.ldc = K;
  gw.a_act = MMVAE_ACT_NONE; gw.b_act = x_act; gw.ep = MMVAE_EP_NONE; gw.accumulate = accumulate ? 1 : 0;
  int kper = (M + sk - 1) / sk;
  kper = (kper + 127) / 128 * 128;
  gw.kper = kper;
  grp.n = 2;
  grp.nx[0] = (K + 31) / 32; grp.ny[0] = (M + 31) / 32; grp.nz[0] = 1;
  grp.nx[1] = (K + 31) / 32; grp.ny[1] = (N + 31) / 32; grp.nz[1] = nz;
  grp.blk0[0] = 0;
  grp.blk0[1] = grp.nx[0] * grp.ny[0];
  grp.blk0[2] = grp.blk0[1] + grp.nx[1] * grp.ny[1] * nz;
  hipLaunchKernelGGL(gemm_grouped_kernel, dim3(grp.blk0[2]), dim3(256), 0, (hipStream_t)stream, grp);
  int rc = mmvae_launch_status();
  if (rc) return rc;
  if (nz > 1 && accumulate != MMVAE_ACC_DEFER) {
    rc = mmvae_reduce_rows(ws, dw, nz, (long)N * K, (long)N * K, accumulate, stream);
    if (rc) return rc;
    if (db) rc = mmvae_reduce_rows(ws + (size_t)nz * N * K, db, nz, N, N, accumulate, stream);
  }
  return rc;
}

MMVAE_TRACE_SETTER(gemm)
