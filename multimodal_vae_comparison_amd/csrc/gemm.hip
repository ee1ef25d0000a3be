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

template <int KSPLIT>
__device__ __forceinline__ void gemm_body(const GemmArgs& g, const int bx, const int by, const int bz, const int nz,
                                          float* __restrict__ As, float* __restrict__ Bs) {
  constexpr int NT = KSPLIT == 8 ? 512 : 256;  // threads
  constexpr int BM = KSPLIT == 1 ? 128 : 32;
  constexpr int BN = 32;
  constexpr int BK = KSPLIT == 1 ? 32 : 32 * KSPLIT;  // every wave owns a 32-deep K slice of a stage
  constexpr int AP = BM + 1;  // LDS row pitch of As (odd)
  constexpr int BP = BN + 1;
  constexpr int A_PER_T = BM * BK / NT;  // 16
  constexpr int B_PER_T = BK * BN / NT;  // 4 or 16
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int m0 = by * BM, n0 = bx * BN;
  const int kbeg = bz * g.kper;
  const int kend = min(g.K, kbeg + g.kper);
  const bool a_kmajor = (g.sak == 1);
  const bool b_kmajor = (g.sbk == 1 && g.sbn != 1);

  // Staging in two phases so that all of a stage's global loads are in flight together: (1) unconditional
  // loads from clamped addresses into registers, (2) select/activation + LDS stores.  (A load followed by a
  // data-dependent branch makes hipcc wait vmcnt(0) per element: measured 8 us per stage.)
  float ra[A_PER_T], rb[B_PER_T];
  unsigned va = 0, vb = 0;

  auto load_stage = [&](int k0) {
    va = vb = 0;
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) {
      const int e = i * NT + tid;
      int ml, kl;
      if (a_kmajor) { kl = e % BK; ml = e / BK; } else { ml = e % BM; kl = e / BM; }
      const int m = m0 + ml, k = k0 + kl;
      const bool ok = m < g.M && k < kend;
      va |= (ok ? 1u : 0u) << i;
      ra[i] = g.A[ok ? (long)m * g.sam + (long)k * g.sak : 0];
    }
#pragma unroll
    for (int i = 0; i < B_PER_T; ++i) {
      const int e = i * NT + tid;
      int nl, kl;
      if (b_kmajor) { kl = e % BK; nl = e / BK; } else { nl = e % BN; kl = e / BN; }
      const int n = n0 + nl, k = k0 + kl;
      const bool ok = n < g.N && k < kend;
      vb |= (ok ? 1u : 0u) << i;
      rb[i] = g.B[ok ? (long)k * g.sbk + (long)n * g.sbn : 0];
    }
  };
  auto store_stage = [&]() {
    if (g.a_act != MMVAE_ACT_NONE) {
#pragma unroll
      for (int i = 0; i < A_PER_T; ++i) ra[i] = apply_in_act(ra[i], g.a_act);
    }
    if (g.b_act != MMVAE_ACT_NONE) {
#pragma unroll
      for (int i = 0; i < B_PER_T; ++i) rb[i] = apply_in_act(rb[i], g.b_act);
    }
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) {
      const int e = i * NT + tid;
      int ml, kl;
      if (a_kmajor) { kl = e % BK; ml = e / BK; } else { ml = e % BM; kl = e / BM; }
      As[kl * AP + ml] = (va >> i & 1u) ? ra[i] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < B_PER_T; ++i) {
      const int e = i * NT + tid;
      int nl, kl;
      if (b_kmajor) { kl = e % BK; nl = e / BK; } else { nl = e % BN; kl = e / BN; }
      Bs[kl * BP + nl] = (vb >> i & 1u) ? rb[i] : 0.f;
    }
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float asum = 0.f;

  const int a_off = (KSPLIT == 1 ? wave * 32 : 0) + li;   // m within the staged tile
  const int k_off = (KSPLIT == 1 ? 0 : wave * 32) + lh;   // k within the staged tile

  if (kbeg < kend) load_stage(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    store_stage();
    __syncthreads();
    if (k0 + BK < kend) load_stage(k0 + BK);
#pragma unroll
    for (int kk = 0; kk < 32; kk += 2) {
      const float a = As[(k_off + kk) * AP + a_off];
      const float b = Bs[(k_off + kk) * BP + li];
      asum += a;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }

  const bool partial = nz > 1;
  float* Cout = partial ? g.ws + (size_t)bz * g.M * g.N : g.C;
  const long ldc = partial ? g.N : g.ldc;
  float* rs_out = partial ? g.ws + (size_t)nz * g.M * g.N + (size_t)bz * g.M : g.a_rowsum;
  const int ep = partial ? MMVAE_EP_NONE : g.ep;
  const bool acc_out = !partial && g.accumulate;

  asum += __shfl_xor(asum, 32, 64);

  auto emit = [&](int row, int col, float v) {
    if (row < g.M && col < g.N) {
      if (!partial && g.bias) v += g.bias[col];
      const long o = (long)row * ldc + col;
      float av = 0.f;
      if (ep_reads_aux(ep)) av = g.aux[o];
      if (ep == MMVAE_EP_GELU && g.aux) g.aux[o] = v;
      v = apply_epilogue(v, av, ep);
      Cout[o] = acc_out ? Cout[o] + v : v;
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
    float* red = As;  // KSPLIT waves x 16 regs x 64 lanes floats <= BK*AP
    float* rsr = Bs;  // KSPLIT x 32 row sums
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

template <int KSPLIT>
__global__ __launch_bounds__(KSPLIT == 8 ? 512 : 256) void gemm_kernel(GemmArgs g) {
  constexpr int BM = KSPLIT == 1 ? 128 : 32;
  constexpr int BK = KSPLIT == 1 ? 32 : 32 * KSPLIT;
  __shared__ float As[BK * (BM + 1)];
  __shared__ float Bs[BK * 33];
  gemm_body<KSPLIT>(g, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.z, As, Bs);
}

// Several independent GEMMs in ONE launch (same tiling): workgroup id -> (problem, tile).  Used for the data- and
// weight-gradient GEMMs of a Linear layer, which only share their inputs: each alone fills a fraction of the chip.
#define GEMM_GROUP_MAX 4
struct GemmGroup {
  GemmArgs g[GEMM_GROUP_MAX];
  int blk0[GEMM_GROUP_MAX + 1];
  int nx[GEMM_GROUP_MAX], ny[GEMM_GROUP_MAX], nz[GEMM_GROUP_MAX];
  int n;
};
__global__ __launch_bounds__(256) void gemm_grouped_kernel(GemmGroup grp) {
  __shared__ float As[128 * 33];
  __shared__ float Bs[128 * 33];
  int p = 0;
  while (p + 1 < grp.n && (int)blockIdx.x >= grp.blk0[p + 1]) ++p;
  const int local = blockIdx.x - grp.blk0[p];
  const int bx = local % grp.nx[p], t = local / grp.nx[p];
  const int by = t % grp.ny[p], bz = t / grp.ny[p];
  gemm_body<4>(grp.g[p], bx, by, bz, grp.nz[p], As, Bs);
}

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
  const int ntn = (N + 31) / 32;
  // Tiling choice (all that matters at batch 128 is the length of the serial load -> MFMA chain per workgroup):
  //   K <= 128 and many rows : 128x32 tiles, one stage, no cross-wave reduction            (KSPLIT 1)
  //   K >= 384, few tiles    : 32x32 tile, 8 waves x 256-deep stages (half the serial stages)  (KSPLIT 8)
  //   otherwise              : 32x32 tile, 4 waves x 128-deep stages                       (KSPLIT 4)
  const long tiles128 = (long)((M + 127) / 128) * ntn, tiles32 = (long)((M + 31) / 32) * ntn;
  int variant = 4;
  if (tiles128 >= 256) variant = 1;   // measured: below that the single-stage 32x32 tiles are faster
  else if (K >= 384 && tiles32 <= 512 && splitk == 1) variant = 8;
  const int bk = variant == 1 ? 32 : 32 * variant;
  int kper = (K + splitk - 1) / splitk;
  kper = (kper + bk - 1) / bk * bk;
  const int nz = (K + kper - 1) / kper;
  g.kper = kper;
  if (nz > 1 && !ws) return MMVAE_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (variant == 1)
    hipLaunchKernelGGL(gemm_kernel<1>, dim3(ntn, (M + 127) / 128, nz), dim3(256), 0, st, g);
  else if (variant == 8)
    hipLaunchKernelGGL(gemm_kernel<8>, dim3(ntn, (M + 31) / 32, nz), dim3(512), 0, st, g);
  else
    hipLaunchKernelGGL(gemm_kernel<4>, dim3(ntn, (M + 31) / 32, nz), dim3(256), 0, st, g);
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
extern "C" int mmvae_linear_fwd(const float* x, const float* w, const float* b, float* aux, float* y, int M, int N,
                                int K, long ldx, int x_act, int ep_mode, mmvae_stream_t stream) {
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
  const int bk = variant == 1 ? 32 : 32 * variant;
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

// Fused nn.Linear backward: dx = ep(dy W) and dW (+)= dy^T act(x), db (+)= colsum(dy) in ONE grouped launch.
// Falls back to two launches when either problem wants a different tiling.
extern "C" size_t mmvae_linear_bwd_ws_floats(int M, int N, int K) { return mmvae_linear_bwd_weight_ws_floats(M, N, K); }
extern "C" int mmvae_linear_bwd(const float* dy, const float* x, const float* w, const float* aux, float* dx,
                                float* dw, float* db, float* ws, int M, int N, int K, long ldx, int x_act, int ep_mode,
                                int accumulate, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(dy && x && w && dx && dw && M > 0 && N > 0 && K > 0);
  if (ep_reads_aux(ep_mode) && !aux) return MMVAE_ERR_ARG;
  // problem 0 (data):   C[M,K] = dy[M,N] W[N,K]            reduction N
  // problem 1 (weight): C[N,K] = dy^T[N,M] act(x)[M,K]     reduction M, split over workgroups
  const long t128_d = (long)((M + 127) / 128) * ((K + 31) / 32), t32_d = (long)((M + 31) / 32) * ((K + 31) / 32);
  const bool d_ok = t128_d < 256 && !(N >= 384 && t32_d <= 512);
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
