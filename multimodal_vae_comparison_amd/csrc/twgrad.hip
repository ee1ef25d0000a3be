// All weight gradients behind one fused text layer in ONE launch (gfx950): dW_j = dY_j^T X_j, db_j = colsum(dY_j) for
// up to 8 independent jobs (the in_proj / out_proj / linear1 / linear2 [/ cross out_proj / cross value rows] of
// torch.nn.TransformerEncoderLayer / TransformerDecoderLayer, models/encoders.py:828-837, decoders.py:708-723), over the
// (L*N)-row tensors the layer kernels leave in HBM.
//
// These are tall-skinny reductions: (L*N = 4096 .. 32000 rows) x (32 .. 162 columns) operands, <= 162 x 128 outputs.
// The split-K register-operand GEMMs they replace tiled the OUTPUT (16 x 16 tiles x splits), so every operand row was
// fetched once per output tile column (L2 hit 0.27, 64 % of wave cycles parked on s_waitcnt: profiles/r03_pmc_gemm_txt.txt).
// Here the output of a job is cut into a few UNITS of 64 x 64 (two columns per lane on each side), a wave owns one unit
// over its own slice of the rows and keeps the unit's 2 x 2 accumulator tiles (v_mfma_f32_32x32x2_f32) in registers:
//   * no LDS staging, no barrier in the row loop: the MFMA's k index is the ROW, so lane (i, half) feeds the k-step
//     with row 2s + half, columns {2i, 2i+1}: one 8-byte load per operand per k-step, 256 contiguous bytes per row and
//     half-wave, every operand byte fetched once per unit column (x is shared by a job's <= 3 units through L2);
//   * 8 k-steps (16 rows) per chunk, the next chunk's 16 loads in flight under the 32 MFMAs of the current one;
//   * the four waves of a workgroup own adjacent row slices of the same unit and sum their accumulators through LDS,
//     one partial row per workgroup goes to the step arena and is folded with every other split gradient
//     (mmvae_reduce_segments / mmvae_adam_fold_flat: fixed order, no atomics).
#include "common.hpp"

namespace tw {

constexpr int MAXJ = 8;
constexpr int CH = 8;    // k-steps per chunk (16 rows): 206 VGPRs, two waves per SIMD (16 k-steps spill)

struct Job {
  const float* dy;
  const float* x;
  float* ws;
  int M, N, K;
  int nz, rs;          // splits, rows per split (multiple of 8)
  int aw, bw;          // floats per lane on the dY / X side (1 | 2)
  int n_sk;            // units along K
  int wg0;             // first workgroup of this job
  int pitch;           // floats per partial row: [N * K weight sums | N bias sums], rounded up to even
};
struct Args {
  Job job[MAXJ];
  int n_jobs;
};

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int acc_row(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

template <int W>
struct Frag {
  float v[CH][W];
};

// rows m0 + 2s + lh (s < CH) of a (M, LD) matrix, columns col .. col + W - 1; FULL: every row and column is in range
template <int W, bool FULL>
__device__ __forceinline__ void load_chunk(Frag<W>& f, const float* __restrict__ base, const int LD, const int m0,
                                           const int m_end, const int lh, const int col, const bool col_ok) {
  const float* p = base + (size_t)(m0 + lh) * LD + col;
  const size_t step = (size_t)2 * LD;
#pragma unroll
  for (int s = 0; s < CH; ++s, p += step) {
    if (FULL) {
      if (W == 2) {
        const float2 t = *reinterpret_cast<const float2*>(p);
        f.v[s][0] = t.x;
        f.v[s][W - 1] = t.y;
      } else {
        f.v[s][0] = *p;
      }
    } else {
      const bool ok = col_ok && (m0 + 2 * s + lh) < m_end;
      const float* q = ok ? p : base;
      if (W == 2) {
        const float2 t = *reinterpret_cast<const float2*>(q);
        f.v[s][0] = ok ? t.x : 0.f;
        f.v[s][W - 1] = ok ? t.y : 0.f;
      } else {
        const float t = *q;
        f.v[s][0] = ok ? t : 0.f;
      }
    }
  }
}

template <int AW, int BW>
__device__ __forceinline__ void unit_body(const Job& jb, const int unit, const int split, float* __restrict__ red) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int s_n = unit / jb.n_sk, s_k = unit - s_n * jb.n_sk;
  const int N = jb.N, K = jb.K;
  const int colA = s_n * 32 * AW + AW * li, colB = s_k * 32 * BW + BW * li;
  const bool okA = colA < N, okB = colB < K;     // AW == 2: N even, so colA + 1 < N as well
  const int q = jb.rs >> 2;
  int r0 = split * jb.rs + wave * q;
  int r1 = r0 + q;
  if (r1 > jb.M) r1 = jb.M;
  if (r0 > r1) r0 = r1;
  // a whole unit inside the matrix lets the row loop run unchecked loads
  const bool cols_full = (s_n * 32 * AW + 32 * AW <= N) && (s_k * 32 * BW + 32 * BW <= K);

  f32x16 acc[AW][BW];
#pragma unroll
  for (int a = 0; a < AW; ++a)
#pragma unroll
    for (int b = 0; b < BW; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float bsum[AW];
#pragma unroll
  for (int a = 0; a < AW; ++a) bsum[a] = 0.f;

  Frag<AW> fa[2];
  Frag<BW> fb[2];
  auto load = [&](int buf, int m0) {
    if (cols_full && m0 + 2 * CH <= r1) {
      load_chunk<AW, true>(fa[buf], jb.dy, N, m0, r1, lh, colA, true);
      load_chunk<BW, true>(fb[buf], jb.x, K, m0, r1, lh, colB, true);
    } else {
      load_chunk<AW, false>(fa[buf], jb.dy, N, m0, r1, lh, colA, okA);
      load_chunk<BW, false>(fb[buf], jb.x, K, m0, r1, lh, colB, okB);
    }
  };
  auto mma = [&](int buf) {
#pragma unroll
    for (int s = 0; s < CH; ++s) {
#pragma unroll
      for (int a = 0; a < AW; ++a) {
        bsum[a] += fa[buf].v[s][a];
#pragma unroll
        for (int b = 0; b < BW; ++b) acc[a][b] = mfma(fa[buf].v[s][a], fb[buf].v[s][b], acc[a][b]);
      }
    }
  };
  if (r0 < r1) {
    load(0, r0);
    int m = r0;
    while (true) {
      if (m + 2 * CH < r1) load(1, m + 2 * CH);
      mma(0);
      m += 2 * CH;
      if (m >= r1) break;
      if (m + 2 * CH < r1) load(0, m + 2 * CH);
      mma(1);
      m += 2 * CH;
      if (m >= r1) break;
    }
  }

  // ---- sum the four waves' accumulators through LDS (waves 1-3 publish, wave 0 adds and stores) ----
  constexpr int NR = AW * BW * 16 + AW;      // registers per lane
  float* mine = red + (size_t)(wave - 1) * NR * 64 + lane;
  if (wave > 0) {
#pragma unroll
    for (int a = 0; a < AW; ++a)
#pragma unroll
      for (int b = 0; b < BW; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) mine[((a * BW + b) * 16 + r) * 64] = acc[a][b][r];
#pragma unroll
    for (int a = 0; a < AW; ++a) mine[(AW * BW * 16 + a) * 64] = bsum[a];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll 1
    for (int w = 0; w < 3; ++w) {
      const float* o = red + (size_t)w * NR * 64 + lane;
#pragma unroll
      for (int a = 0; a < AW; ++a)
#pragma unroll
        for (int b = 0; b < BW; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][b][r] += o[((a * BW + b) * 16 + r) * 64];
#pragma unroll
      for (int a = 0; a < AW; ++a) bsum[a] += o[(AW * BW * 16 + a) * 64];
    }
    float* pw = jb.ws + (size_t)split * jb.pitch;
    if (okB) {
#pragma unroll
      for (int a = 0; a < AW; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = s_n * 32 * AW + AW * acc_row(r, lh) + a;
          if (n < N) {
            float* o = pw + (size_t)n * K + colB;
            if (BW == 2) *reinterpret_cast<float2*>(o) = make_float2(acc[a][0][r], acc[a][BW - 1][r]);
            else *o = acc[a][0][r];
          }
        }
    }
    if (s_k == 0) {
#pragma unroll
      for (int a = 0; a < AW; ++a) bsum[a] += __shfl_xor(bsum[a], 32, 64);
      if (lh == 0 && okA) {
        float* pb = pw + (size_t)N * K + colA;
#pragma unroll
        for (int a = 0; a < AW; ++a) pb[a] = bsum[a];
      }
    }
  }
}

__global__ __launch_bounds__(256, 2) void txt_wgrad_kernel(const Args args) {
  MMVAE_TRACE_STAMP(35);
  __shared__ __attribute__((aligned(16))) float red[3 * (4 * 16 + 2) * 64];
  const int b = blockIdx.x;
  Job jb = args.job[0];
#pragma unroll
  for (int j = 1; j < MAXJ; ++j)
    if (j < args.n_jobs && b >= args.job[j].wg0) jb = args.job[j];
  const int local = b - jb.wg0;
  const int unit = local / jb.nz, split = local - unit * jb.nz;
  if (jb.aw == 2 && jb.bw == 2) unit_body<2, 2>(jb, unit, split, red);
  else if (jb.aw == 2) unit_body<2, 1>(jb, unit, split, red);
  else if (jb.bw == 2) unit_body<1, 2>(jb, unit, split, red);
  else unit_body<1, 1>(jb, unit, split, red);
}

// one partial row of a job = its N * K weight sums followed by its N bias sums: ONE fold segment per job (weight and bias
// are adjacent in the flat gradient buffer) instead of two -- the step's closing fold + Adam launch takes 64 entries
static inline int row_pitch(int N, int K) { return (N * K + N + 1) & ~1; }
static inline int plan_splits(int M) {
  int nz = M / 128;
  if (nz < 1) nz = 1;
  if (nz > 64) nz = 64;
  return nz;
}

}  // namespace tw

extern "C" int mmvae_txt_wgrad_splits(int M, int N, int K) {
  (void)N;
  (void)K;
  return tw::plan_splits(M);
}
extern "C" size_t mmvae_txt_wgrad_ws_floats(int M, int N, int K) {
  return (size_t)tw::plan_splits(M) * (size_t)tw::row_pitch(N, K);
}
extern "C" int mmvae_txt_wgrad_supported(int M, int N, int K) {
  if (M < 1 || N < 1 || K < 1) return 0;
  if ((N > 32 && (N & 1)) || (K > 32 && (K & 1))) return 0;
  return 1;
}

extern "C" int mmvae_txt_wgrad(const mmvae_txt_wgrad_job_t* jobs, int n_jobs, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(jobs && n_jobs >= 1 && n_jobs <= tw::MAXJ);
  tw::Args a;
  int wg = 0;
  for (int j = 0; j < n_jobs; ++j) {
    const mmvae_txt_wgrad_job_t& s = jobs[j];
    MMVAE_CHECK_ARG(s.dy && s.x && s.ws);
    if (!mmvae_txt_wgrad_supported(s.M, s.N, s.K)) return MMVAE_ERR_UNSUPPORTED;
    tw::Job& d = a.job[j];
    d.dy = s.dy; d.x = s.x; d.ws = s.ws;
    d.M = s.M; d.N = s.N; d.K = s.K;
    d.pitch = tw::row_pitch(s.N, s.K);
    d.aw = s.N > 32 ? 2 : 1;
    d.bw = s.K > 32 ? 2 : 1;
    if (d.aw == 2 && ((uintptr_t)s.dy & 7)) return MMVAE_ERR_ARG;
    if (d.bw == 2 && (((uintptr_t)s.x & 7) || ((uintptr_t)s.ws & 7))) return MMVAE_ERR_ARG;
    d.nz = tw::plan_splits(s.M);
    d.rs = ((s.M + d.nz - 1) / d.nz + 7) / 8 * 8;
    const int n_sn = (s.N + 32 * d.aw - 1) / (32 * d.aw);
    d.n_sk = (s.K + 32 * d.bw - 1) / (32 * d.bw);
    d.wg0 = wg;
    wg += n_sn * d.n_sk * d.nz;
  }
  for (int j = n_jobs; j < tw::MAXJ; ++j) a.job[j] = a.job[0];
  a.n_jobs = n_jobs;
  hipLaunchKernelGGL(tw::txt_wgrad_kernel, dim3(wg), dim3(256), 0, (hipStream_t)stream, a);
  return mmvae_launch_status();
}

MMVAE_TRACE_SETTER(twgrad)
