// One workgroup = one sequence: a whole post-norm Transformer layer of the text towers out of LDS (gfx950).
//
// The text towers run at T <= 32 tokens, d_model 54 / 32, ff 128 (models/encoders.py:790-837, decoders.py:668-723):
// op by op that is ~10 launches per layer and direction, each a 4096-row GEMM or row kernel that is pure launch +
// load latency on a 256-CU chip.  T <= 32 is exactly one 32-row MFMA tile, and a layer's weights (25 K floats)
// stream from L2, so the whole layer is computed per sequence by one 4-wave workgroup:
//
//   forward  (mmvae_txt_layer_fwd):  x -> QKV -> masked softmax attention (one wave per head) -> out_proj
//            -> +x, dropout, LayerNorm1 [-> decoder: value-path cross attention over the length-1 memory -> LayerNorm2]
//            -> linear1 -> GELU, dropout -> linear2 -> +, dropout, LayerNorm -> y
//   backward (mmvae_txt_layer_bwd):  the whole data-gradient chain in reverse; it also leaves the per-GEMM output
//            gradients in HBM, from which the weight gradients are ordinary (L*N)-row GEMMs (mmvae_linear_bwd_weight).
//
// MFMA use (v_mfma_f32_32x32x2_f32): rows = tokens.  A operands are read from LDS as ds_read_b64 along k (row pitch
// 2*odd: the 32 rows of a lane group hit 32 different bank pairs); weights are NOT staged: lane j reads its own weight
// row (k-contiguous, float2) or column (coalesced dwords) straight from L2 into registers.  Attention computes
// S^T = K Q^T so that a lane owns one query column: the softmax is a 16-register + one-shuffle reduction and P feeds
// the P V MFMA as the A operand directly from the accumulator registers.
// Dropout masks are the counter-based masks of the op-by-op kernels (same keys, same element indices), so both paths
// produce identical masks for a given DropoutState.
#include "common.hpp"

namespace tl {

constexpr int T = 32;

__host__ __device__ constexpr int r4(int v) { return (v + 3) / 4 * 4; }
// LDS row pitch for a [32][v] buffer read as ds_read_b64 along k: even, >= r4(v), pitch/2 odd
__host__ __device__ constexpr int pitch(int v) { return (r4(v) / 2) % 2 == 0 ? r4(v) + 2 : r4(v); }

template <int D_, int FF_, int NH_, bool DEC_>
struct Geom {
  static constexpr int D = D_, FF = FF_, NH = NH_, HD = D / NH;
  static constexpr bool DEC = DEC_;
  static constexpr int PX = pitch(D), PH = pitch(FF), PQ = (3 * D) | 1, PG = pitch(3 * D);
  static_assert(D % 2 == 0 && FF % 2 == 0 && D % NH == 0 && HD <= 32 && NH <= 4, "shape");
};

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// accumulator register r of lane (li, lh) holds row acc_row(r, lh), column li
__device__ __forceinline__ int acc_row(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

struct Lane {
  int tid, lane, wave, li, lh;
};
__device__ __forceinline__ Lane lane_ids() {
  Lane l;
  l.tid = threadIdx.x;
  l.lane = l.tid & 63;
  l.wave = __builtin_amdgcn_readfirstlane(l.tid >> 6);
  l.li = l.lane & 31;
  l.lh = l.lane >> 5;
  return l;
}

// Weight operand of one 32-column tile, held in registers.  The weights are not staged through LDS: every lane reads
// its own column's K values straight from L2.  `load` is separate from `mma` so that a phase can issue the NEXT
// GEMM's weight loads before its own barrier (the only thing a one-workgroup-per-sequence kernel can overlap).
template <int K>
struct WTile {
  static constexpr int KQ = (K + 3) / 4;
  float2 b[KQ];
  // W [N][K] row-major (k contiguous): lane's row n, k = 4q + 2 lh + {0, 1}
  __device__ __forceinline__ void load_kc(const float* __restrict__ W, const int N, const int nt, const Lane& l) {
    const int n = nt * 32 + l.li;
    const bool nv = n < N;
    const float* wr = W + (size_t)(nv ? n : 0) * K;
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      const int k = 4 * q + 2 * l.lh;
      const bool kv = (4 * q + 4 <= K) || (k < K);
      b[q] = *reinterpret_cast<const float2*>(wr + (kv ? k : 0));
      if (!kv || !nv) b[q] = make_float2(0.f, 0.f);
    }
  }
  // W [K][NJ] row-major (j contiguous): lane's column j
  __device__ __forceinline__ void load_jc(const float* __restrict__ W, const int NJ, const int nt, const Lane& l) {
    const int j = nt * 32 + l.li;
    const bool jv = j < NJ;
    const float* wc = W + (jv ? j : 0);
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      const int k = 4 * q + 2 * l.lh;
      const bool kv0 = (4 * q + 4 <= K) || (k < K), kv1 = (4 * q + 4 <= K) || (k + 1 < K);
      b[q].x = wc[(size_t)(kv0 ? k : 0) * NJ];
      b[q].y = wc[(size_t)(kv1 ? k + 1 : 0) * NJ];
      if (!kv0 || !jv) b[q].x = 0.f;
      if (!kv1 || !jv) b[q].y = 0.f;
    }
  }
  __device__ __forceinline__ f32x16 mma(const float* __restrict__ A, const int PA, const Lane& l) const {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* ar = A + l.li * PA + 2 * l.lh;
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      const float2 a = *reinterpret_cast<const float2*>(ar + 4 * q);
      acc = mfma(a.x, b[q].x, acc);
      acc = mfma(a.y, b[q].y, acc);
    }
    return acc;
  }
};

// C[32][N] = A_lds[32][K] * (weights); waves take the 32-column tiles round robin; the first tile of every wave may
// have been prefetched into `pre` (prefetch_* below).  epi(col, col_valid, acc) consumes one tile.
template <int K, bool KC, typename Epi>
__device__ __forceinline__ void gemm_tiles(const float* __restrict__ A, const int PA, const float* __restrict__ W,
                                           const int N, const Lane& l, WTile<K>& pre, const bool have_pre, Epi&& epi) {
  for (int nt = l.wave; nt * 32 < N; nt += 4) {
    if (!(have_pre && nt == l.wave)) {
      if (KC) pre.load_kc(W, N, nt, l);
      else pre.load_jc(W, N, nt, l);
    }
    const f32x16 acc = pre.mma(A, PA, l);
    const int col = nt * 32 + l.li;
    epi(col, col < N, acc);
  }
}
template <int K, bool KC>
__device__ __forceinline__ void prefetch_tile(WTile<K>& pre, const float* __restrict__ W, const int N, const Lane& l) {
  if (l.wave * 32 < N) {
    if (KC) pre.load_kc(W, N, l.wave, l);
    else pre.load_jc(W, N, l.wave, l);
  }
}
template <int K, typename Epi>
__device__ __forceinline__ void gemm_kc(const float* __restrict__ A, const int PA, const float* __restrict__ W,
                                        const int N, const Lane& l, Epi&& epi) {
  WTile<K> w;
  gemm_tiles<K, true>(A, PA, W, N, l, w, false, epi);
}
template <int K, typename Epi>
__device__ __forceinline__ void gemm_jc(const float* __restrict__ A, const int PA, const float* __restrict__ W,
                                        const int NJ, const Lane& l, Epi&& epi) {
  WTile<K> w;
  gemm_tiles<K, false>(A, PA, W, NJ, l, w, false, epi);
}

// rows t < L of a (L, N, COLS) tensor for sequence n -> LDS [32][P]: all loads issued before the first store
// (clamped addresses, no predicated loads), rows >= L zero.
template <int COLS>
__device__ __forceinline__ void stage_rows(float* __restrict__ dst, const int P, const float* __restrict__ src,
                                           const size_t N, const int n, const int L, const Lane& l) {
  constexpr int NIT = (T * COLS + 255) / 256;
  float v[NIT];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int e = l.tid + 256 * i;
    const int t = e / COLS, c = e - t * COLS;
    const int tc = t < L ? t : L - 1;
    v[i] = src[((size_t)tc * N + n) * COLS + c];
  }
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int e = l.tid + 256 * i;
    const int t = e / COLS, c = e - t * COLS;
    if (e < T * COLS) dst[t * P + c] = t < L ? v[i] : 0.f;
  }
}

// LayerNorm over the D columns of the 32 rows of R (LDS): 8 lanes per row.  out(t, c, xhat, y) per element,
// row_out(t, rstd) once per row.
template <int D, typename Out, typename RowOut>
__device__ __forceinline__ void layernorm_rows(const float* __restrict__ R, const int PR, const float* __restrict__ gamma,
                                               const float* __restrict__ beta, const Lane& l, Out&& out, RowOut&& row_out) {
  constexpr int NJ = (D + 7) / 8;
  const int t = l.tid >> 3, sub = l.tid & 7;
  float v[NJ];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = sub + 8 * j;
    v[j] = c < D ? R[t * PR + c] : 0.f;
    s += v[j];
  }
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  s += __shfl_xor(s, 4, 64);
  const float mean = s / (float)D;
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const float dv = (sub + 8 * j) < D ? v[j] - mean : 0.f;
    ss += dv * dv;
  }
  ss += __shfl_xor(ss, 1, 64);
  ss += __shfl_xor(ss, 2, 64);
  ss += __shfl_xor(ss, 4, 64);
  const float rs = rsqrtf(ss / (float)D + 1e-5f);
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = sub + 8 * j;
    if (c < D) {
      const float xh = (v[j] - mean) * rs;
      out(t, c, xh, xh * gamma[c] + beta[c]);
    }
  }
  if (sub == 0) row_out(t, rs);
}

// dr = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma, over the rows of DY / XH (LDS)
template <int D, typename Out>
__device__ __forceinline__ void layernorm_bwd_rows(const float* __restrict__ DY, const float* __restrict__ XH, const int P,
                                                   const float* __restrict__ gamma, const float* __restrict__ rstd_row,
                                                   const Lane& l, Out&& out) {
  constexpr int NJ = (D + 7) / 8;
  const int t = l.tid >> 3, sub = l.tid & 7;
  float g[NJ], xh[NJ];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = sub + 8 * j;
    const bool ok = c < D;
    xh[j] = ok ? XH[t * P + c] : 0.f;
    g[j] = ok ? DY[t * P + c] * gamma[c] : 0.f;
    s1 += g[j];
    s2 += g[j] * xh[j];
  }
  s1 += __shfl_xor(s1, 1, 64); s2 += __shfl_xor(s2, 1, 64);
  s1 += __shfl_xor(s1, 2, 64); s2 += __shfl_xor(s2, 2, 64);
  s1 += __shfl_xor(s1, 4, 64); s2 += __shfl_xor(s2, 4, 64);
  const float m1 = s1 / (float)D, m2 = s2 / (float)D, rs = rstd_row[t];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = sub + 8 * j;
    if (c < D) out(t, c, rs * (g[j] - m1 - xh[j] * m2));
  }
}

// Scores of one head, transposed: lane (li = query, lh) register r = key acc_row(r, lh).  Returns the normalised
// probabilities pn[] (pre-dropout).  Q at column q0, K at column k0 of the rows of QB (pitch PQ, odd).
template <int HD>
__device__ __forceinline__ void attn_probs(const float* __restrict__ QB, const int PQ, const int q0, const int k0,
                                           const float* __restrict__ s_valid, const int L, const Lane& l, float (&pn)[16]) {
  constexpr int S = (HD + 1) / 2;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const float* kr = QB + l.li * PQ + k0 + l.lh;   // A[i = key][k = dd]
  const float* qr = QB + l.li * PQ + q0 + l.lh;   // B[k = dd][j = query]
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const bool dv = (2 * s + 2 <= HD) || (2 * s + l.lh < HD);
    const float a = dv ? kr[2 * s] : 0.f;
    const float b = dv ? qr[2 * s] : 0.f;
    acc = mfma(a, b, acc);
  }
  const float scale = 1.0f / sqrtf((float)HD);
  float mx = -INFINITY;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int key = acc_row(r, l.lh);
    float sc = acc[r] * scale;
    if (key >= L || s_valid[key] == 0.f) sc = -INFINITY;
    pn[r] = sc;
    mx = fmaxf(mx, sc);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    pn[r] = expf(pn[r] - mx);
    sum += pn[r];
  }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
#pragma unroll
  for (int r = 0; r < 16; ++r) pn[r] *= inv;
}

}  // namespace tl

using namespace tl;

// ------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------
template <typename G>
__global__ __launch_bounds__(256) void txt_layer_fwd_kernel(const float* __restrict__ x, const uint8_t* __restrict__ valid,
                                                            const float* __restrict__ mem, float* __restrict__ y,
                                                            const mmvae_txt_layer_w_t w, const mmvae_txt_layer_saved_t sv,
                                                            const mmvae_txt_layer_drop_t dr, const int L, const int N,
                                                            const int time_mean, const float* __restrict__ head_w,
                                                            const float* __restrict__ head_b,
                                                            float* __restrict__ heads, const int HN) {
  MMVAE_TRACE_STAMP(16 + (G::DEC ? 1 : 0));
  constexpr int D = G::D, FF = G::FF, NH = G::NH, HD = G::HD, PX = G::PX, PH = G::PH, PQ = G::PQ;
  constexpr int NBUF = G::DEC ? 5 : 4;
  constexpr int SMEM = NBUF * T * PX + T * PQ + T * PH + 64 + 64;
  __shared__ __attribute__((aligned(16))) float smem[SMEM];
  float* XB = smem;                 // layer input (residual of block 1)
  float* AB = XB + T * PX;          // attention output / cross-attention value rows
  float* RB = AB + T * PX;          // pre-LayerNorm sums
  float* X1B = RB + T * PX;         // LayerNorm1 output
  float* X2B = G::DEC ? X1B + T * PX : X1B;   // decoder: LayerNorm2 output (input of the FFN block)
  float* QB = smem + NBUF * T * PX; // qkv, pitch PQ
  float* HB = QB + T * PQ;          // dropout(gelu(linear1)), pitch PH
  float* s_valid = HB + T * PH;     // [32] 1 = real token
  float* s_vec = s_valid + 64;      // decoder: value projection of the memory row [D]
  const Lane l = lane_ids();
  const int n = blockIdx.x;
  const size_t LN = (size_t)N;

  for (int i = l.tid; i < SMEM / 4; i += 256) reinterpret_cast<float4*>(smem)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  WTile<D> wd;       // prefetched first weight tile of the next K = D GEMM
  WTile<FF> wf;      // ... of linear2
  prefetch_tile<D, true>(wd, w.in_w, 3 * D, l);
  stage_rows<D>(XB, PX, x, LN, n, L, l);
  if (l.tid < T) s_valid[l.tid] = (l.tid < L && valid[(size_t)n * L + l.tid] != 0) ? 1.f : 0.f;
  __syncthreads();

  // ---- QKV projection ----
  gemm_tiles<D, true>(XB, PX, w.in_w, 3 * D, l, wd, true, [&](int col, bool cv, const f32x16& acc) {
    if (!cv) return;
    const float b = w.in_b[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = acc_row(r, l.lh);
      const float v = acc[r] + b;
      QB[t * PQ + col] = v;
      if (t < L) sv.qkv[((size_t)t * LN + n) * (3 * D) + col] = v;
    }
  });
  prefetch_tile<D, true>(wd, w.out_w, D, l);
  __syncthreads();

  // ---- attention: one wave per head ----
  if (l.wave < NH) {
    const int h = l.wave;
    float p[16];
    attn_probs<HD>(QB, PQ, h * HD, D + h * HD, s_valid, L, l, p);
    const DropKey dk = drop_key(dr.attn);
    const uint32_t drow = (uint32_t)((((size_t)n * NH + h) * L + l.li) * L);   // query = li
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    const float* vr = QB + 2 * D + h * HD + l.li;    // B[k = key][j = dv = li]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = acc_row(r, l.lh);
      if (sv.probs && l.li < L && key < L) sv.probs[(((size_t)n * NH + h) * L + l.li) * L + key] = p[r];
      const float pd = p[r] * drop_mul(dk, drow + key);
      const float b = l.li < HD ? vr[key * PQ] : 0.f;
      o = mfma(pd, b, o);
    }
    if (l.li < HD) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int t = acc_row(r, l.lh);
        AB[t * PX + h * HD + l.li] = o[r];
        if (t < L) sv.ao[((size_t)t * LN + n) * D + h * HD + l.li] = o[r];
      }
    }
  }
  __syncthreads();

  // ---- out_proj, + x, dropout1 -> RB ----
  {
    const DropKey dk = drop_key(dr.drop1);
    gemm_tiles<D, true>(AB, PX, w.out_w, D, l, wd, true, [&](int col, bool cv, const f32x16& acc) {
      if (!cv) return;
      const float b = w.out_b[col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int t = acc_row(r, l.lh);
        const uint32_t idx = (uint32_t)(((size_t)t * LN + n) * D + col);
        RB[t * PX + col] = XB[t * PX + col] + (acc[r] + b) * drop_mul(dk, idx);
      }
    });
  }
  if (!G::DEC) prefetch_tile<D, true>(wd, w.l1_w, FF, l);
  __syncthreads();
  layernorm_rows<D>(RB, PX, w.n1_g, w.n1_b, l,
                    [&](int t, int c, float xh, float yv) {
                      X1B[t * PX + c] = yv;
                      if (t < L) {
                        const size_t o = ((size_t)t * LN + n) * D + c;
                        sv.xhat1[o] = xh;
                        sv.x1[o] = yv;
                      }
                    },
                    [&](int t, float rs) { if (t < L) sv.rstd1[(size_t)t * LN + n] = rs; });
  __syncthreads();

  if (G::DEC) {
    // ---- cross attention over the length-1 memory: softmax == 1, so ca[t] = out_proj_x(dropout_t,h(v)), v = W_v mem + b_v
    if (l.tid < D) {
      const float* mr = mem + (size_t)n * D;
      const float* wr = w.x_in_w + (size_t)l.tid * D;
      float a = w.x_in_b[l.tid];
      for (int k = 0; k < D; ++k) a += mr[k] * wr[k];
      s_vec[l.tid] = a;
      sv.vproj[(size_t)n * D + l.tid] = a;
    }
    __syncthreads();
    {
      const DropKey dk = drop_key(dr.xattn);
      for (int e = l.tid; e < T * D; e += 256) {
        const int t = e / D, c = e - t * D, h = c / HD;
        const float v = s_vec[c] * drop_mul(dk, (uint32_t)(((size_t)n * NH + h) * L + t));
        AB[t * PX + c] = v;
        if (t < L) sv.vb[((size_t)t * LN + n) * D + c] = v;
      }
    }
    __syncthreads();
    {
      const DropKey dk = drop_key(dr.drop2);
      gemm_kc<D>(AB, PX, w.x_out_w, D, l, [&](int col, bool cv, const f32x16& acc) {
        if (!cv) return;
        const float b = w.x_out_b[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int t = acc_row(r, l.lh);
          const uint32_t idx = (uint32_t)(((size_t)t * LN + n) * D + col);
          RB[t * PX + col] = X1B[t * PX + col] + (acc[r] + b) * drop_mul(dk, idx);
        }
      });
    }
    prefetch_tile<D, true>(wd, w.l1_w, FF, l);
    __syncthreads();
    layernorm_rows<D>(RB, PX, w.n2_g, w.n2_b, l,
                      [&](int t, int c, float xh, float yv) {
                        X2B[t * PX + c] = yv;
                        if (t < L) {
                          const size_t o = ((size_t)t * LN + n) * D + c;
                          sv.xhat2[o] = xh;
                          sv.x2[o] = yv;
                        }
                      },
                      [&](int t, float rs) { if (t < L) sv.rstd2[(size_t)t * LN + n] = rs; });
    __syncthreads();
  }

  // ---- FFN: linear1 -> GELU, dropout -> linear2, + residual, dropout -> LayerNorm ----
  {
    const DropKey dk = drop_key(dr.ffn);
    gemm_tiles<D, true>(X2B, PX, w.l1_w, FF, l, wd, true, [&](int col, bool cv, const f32x16& acc) {
      if (!cv) return;
      const float b = w.l1_b[col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int t = acc_row(r, l.lh);
        const size_t o = ((size_t)t * LN + n) * FF + col;
        const float h1 = acc[r] + b;
        const float g = dev_gelu(h1) * drop_mul(dk, (uint32_t)o);
        HB[t * PH + col] = g;
        if (t < L) {
          sv.h1[o] = h1;
          sv.g[o] = g;
        }
      }
    });
  }
  prefetch_tile<FF, true>(wf, w.l2_w, D, l);
  __syncthreads();
  {
    const DropKey dk = drop_key(G::DEC ? dr.drop3 : dr.drop2);
    gemm_tiles<FF, true>(HB, PH, w.l2_w, D, l, wf, true, [&](int col, bool cv, const f32x16& acc) {
      if (!cv) return;
      const float b = w.l2_b[col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int t = acc_row(r, l.lh);
        const uint32_t idx = (uint32_t)(((size_t)t * LN + n) * D + col);
        RB[t * PX + col] = X2B[t * PX + col] + (acc[r] + b) * drop_mul(dk, idx);
      }
    });
  }
  __syncthreads();
  layernorm_rows<D>(RB, PX, G::DEC ? w.n3_g : w.n2_g, G::DEC ? w.n3_b : w.n2_b, l,
                    [&](int t, int c, float xh, float yv) {
                      if (t < L) {
                        const size_t o = ((size_t)t * LN + n) * D + c;
                        sv.xhatf[o] = xh;
                        if (time_mean) RB[t * PX + c] = yv;   // same lane read this element: in-place is safe
                        else y[o] = yv;
                      }
                    },
                    [&](int t, float rs) { if (t < L) sv.rstdf[(size_t)t * LN + n] = rs; });
  if (time_mean) {   // y (N, D) = mean over the L frames (the encoder's pooling): no (L, N, D) round trip, no launch
    __syncthreads();
    float zc = 0.f;
    if (l.tid < D) {
      float a = 0.f;
      for (int t = 0; t < L; ++t) a += RB[t * PX + l.tid];
      zc = a * (1.0f / (float)L);
      y[(size_t)n * D + l.tid] = zc;
    }
    if (head_w) {   // the posterior heads on the pooled feature: heads[n, j] = head_w[j, :] . z + head_b[j] (a 54 x 64
                    // GEMV per sequence: cheaper here than one more launch between the encoder and the fusion)
      __syncthreads();
      if (l.tid < D) RB[l.tid] = zc;
      __syncthreads();
      const int q = l.tid & 3;
      for (int j = l.tid >> 2; j < HN; j += 64) {
        float a = 0.f;
        for (int c = q; c < D; c += 4) a += head_w[(size_t)j * D + c] * RB[c];
        a += __shfl_xor(a, 1, 64);
        a += __shfl_xor(a, 2, 64);
        if (q == 0) heads[(size_t)n * HN + j] = a + head_b[j];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// backward (data-gradient chain); LayerNorm gamma/beta partials: lnws[n][k][2][D], k = 0 first norm ... last norm
// ------------------------------------------------------------------------------------------------------------
template <typename G>
__global__ __launch_bounds__(256) void txt_layer_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ valid,
                                                            float* __restrict__ dx, float* __restrict__ dmem,
                                                            const mmvae_txt_layer_w_t w, const mmvae_txt_layer_saved_t sv,
                                                            const mmvae_txt_layer_grads_t gr,
                                                            const mmvae_txt_layer_drop_t dr, const int L, const int N,
                                                            const int time_mean) {
  MMVAE_TRACE_STAMP(18 + (G::DEC ? 1 : 0));
  constexpr int D = G::D, FF = G::FF, NH = G::NH, HD = G::HD, PX = G::PX, PH = G::PH, PQ = G::PQ, PG = G::PG;
  constexpr int NLN = G::DEC ? 3 : 2;
  constexpr int PP = 33;
  constexpr int SMEM = 4 * T * PX + T * PQ + T * PG + T * PH + NH * T * PP + 64 + 3 * 32 + 64;
  __shared__ __attribute__((aligned(16))) float smem[SMEM];
  float* DYB = smem;                // incoming gradient of the current block's LayerNorm output
  float* XHB = DYB + T * PX;        // xhat of that LayerNorm
  float* DRB = XHB + T * PX;        // gradient of the pre-norm sum (= residual path gradient)
  float* GAB = DRB + T * PX;        // A operand of the next data-gradient GEMM (masked branch gradient)
  float* QB = GAB + T * PX;         // qkv (recomputed softmax), pitch PQ
  float* DQB = QB + T * PQ;         // d qkv, pitch PG
  float* DHB = DQB + T * PG;        // d h1, pitch PH
  float* PB = DHB + T * PH;         // per head [query][key] scratch for the transposed operands, pitch 33
  float* s_valid = PB + NH * T * PP;
  float* s_rstd = s_valid + 64;     // [3][32]
  float* s_vec = s_rstd + 96;       // decoder: d v (gradient of the value projection) [D]
  const Lane l = lane_ids();
  const int n = blockIdx.x;
  const size_t LN = (size_t)N;

  for (int i = l.tid; i < SMEM / 4; i += 256) reinterpret_cast<float4*>(smem)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  WTile<D> wd;        // prefetched first weight tile of the next K = D data-gradient GEMM
  prefetch_tile<D, false>(wd, w.l2_w, FF, l);
  if (time_mean) {   // dy (N, D) is the gradient of the mean over frames: every row gets dy / L
    for (int e = l.tid; e < T * D; e += 256) {
      const int t = e / D, c = e - t * D;
      DYB[t * PX + c] = t < L ? dy[(size_t)n * D + c] * (1.0f / (float)L) : 0.f;
    }
  } else {
    stage_rows<D>(DYB, PX, dy, LN, n, L, l);
  }
  stage_rows<D>(XHB, PX, sv.xhatf, LN, n, L, l);
  stage_rows<3 * D>(QB, PQ, sv.qkv, LN, n, L, l);
  if (l.tid < T) {
    s_valid[l.tid] = (l.tid < L && valid[(size_t)n * L + l.tid] != 0) ? 1.f : 0.f;
    const size_t row = (size_t)l.tid * LN + n;
    s_rstd[l.tid] = l.tid < L ? sv.rstd1[row] : 0.f;
    s_rstd[32 + l.tid] = (G::DEC && l.tid < L) ? sv.rstd2[row] : 0.f;
    s_rstd[64 + l.tid] = l.tid < L ? sv.rstdf[row] : 0.f;
  }
  __syncthreads();

  // column sums of dy * xhat and dy over this sequence's rows -> LayerNorm parameter gradient partials
  auto ln_param_partials = [&](int k) {
    if (l.tid < D) {
      float sg = 0.f, sb = 0.f;
      for (int t = 0; t < T; ++t) {
        const float d = DYB[t * PX + l.tid];
        sg += d * XHB[t * PX + l.tid];
        sb += d;
      }
      float* o = gr.lnws + (((size_t)n * NLN + k) * 2) * D;
      o[l.tid] = sg;
      o[D + l.tid] = sb;
    }
  };

  // ================= FFN block =================
  ln_param_partials(NLN - 1);
  {
    const DropKey dk = drop_key(G::DEC ? dr.drop3 : dr.drop2);
    layernorm_bwd_rows<D>(DYB, XHB, PX, G::DEC ? w.n3_g : w.n2_g, s_rstd + 64, l, [&](int t, int c, float v) {
      DRB[t * PX + c] = v;
      const size_t o = ((size_t)t * LN + n) * D + c;
      const float m = v * drop_mul(dk, (uint32_t)o);
      GAB[t * PX + c] = m;
      if (t < L) gr.d_f[o] = m;           // gradient of linear2's output
    });
  }
  WTile<FF> wf;
  prefetch_tile<FF, false>(wf, w.l1_w, D, l);
  __syncthreads();
  {
    const DropKey dk = drop_key(dr.ffn);
    gemm_tiles<D, false>(GAB, PX, w.l2_w, FF, l, wd, true, [&](int col, bool cv, const f32x16& acc) {
      if (!cv) return;
      float hv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {      // all 16 loads in flight before the first use
        const int t = acc_row(r, l.lh);
        hv[r] = sv.h1[((size_t)(t < L ? t : L - 1) * LN + n) * FF + col];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int t = acc_row(r, l.lh);
        const size_t o = ((size_t)t * LN + n) * FF + col;
        float v = 0.f;
        if (t < L) {
          v = acc[r] * drop_mul(dk, (uint32_t)o) * dev_gelu_grad(hv[r]);
          gr.d_h1[o] = v;                 // gradient of linear1's output
        }
        DHB[t * PH + col] = v;
      }
    });
  }
  __syncthreads();
  // d(FFN input) = DRB + DHB * W1  -> DYB (gradient of the previous LayerNorm's output)
  gemm_tiles<FF, false>(DHB, PH, w.l1_w, D, l, wf, true, [&](int col, bool cv, const f32x16& acc) {
    if (!cv) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = acc_row(r, l.lh);
      DYB[t * PX + col] = DRB[t * PX + col] + acc[r];
    }
  });
  __syncthreads();

  if (G::DEC) {
    // ================= cross-attention block =================
    stage_rows<D>(XHB, PX, sv.xhat2, LN, n, L, l);
    __syncthreads();
    ln_param_partials(1);
    {
      const DropKey dk = drop_key(dr.drop2);
      layernorm_bwd_rows<D>(DYB, XHB, PX, w.n2_g, s_rstd + 32, l, [&](int t, int c, float v) {
        DRB[t * PX + c] = v;
        const size_t o = ((size_t)t * LN + n) * D + c;
        const float m = v * drop_mul(dk, (uint32_t)o);
        GAB[t * PX + c] = m;
        if (t < L) gr.d_ca[o] = m;        // gradient of the cross out_proj's output
      });
    }
    __syncthreads();
    // d vb = GAB * W_ox ; d v[c] = sum_t mask(t, h) * d vb[t][c]
    gemm_jc<D>(GAB, PX, w.x_out_w, D, l, [&](int col, bool cv, const f32x16& acc) {
      if (!cv) return;
#pragma unroll
      for (int r = 0; r < 16; ++r) DHB[acc_row(r, l.lh) * PH + col] = acc[r];
    });
    __syncthreads();
    if (l.tid < D) {
      const DropKey dk = drop_key(dr.xattn);
      const int h = l.tid / HD;
      float a = 0.f;
      for (int t = 0; t < L; ++t) a += DHB[t * PH + l.tid] * drop_mul(dk, (uint32_t)(((size_t)n * NH + h) * L + t));
      s_vec[l.tid] = a;
      gr.d_v[(size_t)n * D + l.tid] = a;  // gradient of the value projection's output (N, D)
    }
    __syncthreads();
    if (l.tid < D) {                      // d mem = d v * W_v
      float a = 0.f;
      for (int c = 0; c < D; ++c) a += s_vec[c] * w.x_in_w[(size_t)c * D + l.tid];
      dmem[(size_t)n * D + l.tid] = a;
    }
    // the residual path: DYB = DRB (gradient of LayerNorm1's output)
    for (int e = l.tid; e < T * D; e += 256) {
      const int t = e / D, c = e - t * D;
      DYB[t * PX + c] = DRB[t * PX + c];
    }
    __syncthreads();
  }

  // ================= self-attention block =================
  stage_rows<D>(XHB, PX, sv.xhat1, LN, n, L, l);
  prefetch_tile<D, false>(wd, w.out_w, D, l);
  __syncthreads();
  ln_param_partials(0);
  {
    const DropKey dk = drop_key(dr.drop1);
    layernorm_bwd_rows<D>(DYB, XHB, PX, w.n1_g, s_rstd, l, [&](int t, int c, float v) {
      DRB[t * PX + c] = v;
      const size_t o = ((size_t)t * LN + n) * D + c;
      const float m = v * drop_mul(dk, (uint32_t)o);
      GAB[t * PX + c] = m;
      if (t < L) gr.d_a[o] = m;           // gradient of out_proj's output
    });
  }
  __syncthreads();
  // d(attention output) -> XHB (reused)
  gemm_tiles<D, false>(GAB, PX, w.out_w, D, l, wd, true, [&](int col, bool cv, const f32x16& acc) {
    if (!cv) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) XHB[acc_row(r, l.lh) * PX + col] = acc[r];
  });
  WTile<3 * D> wq;    // in_proj columns for the last GEMM: in flight under the attention backward
  prefetch_tile<3 * D, false>(wq, w.in_w, D, l);
  __syncthreads();

  // attention backward, one wave per head; DO = XHB
  float ds[16];
  if (l.wave < NH) {
    const int h = l.wave;
    float* P = PB + h * T * PP;
    float p[16];
    attn_probs<HD>(QB, PQ, h * HD, D + h * HD, s_valid, L, l, p);
    const DropKey dk = drop_key(dr.attn);
    const uint32_t drow = (uint32_t)((((size_t)n * NH + h) * L + l.li) * L);
    float mk[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = acc_row(r, l.lh);
      mk[r] = drop_mul(dk, drow + key);
      P[l.li * PP + key] = p[r] * mk[r];          // [query][key]: dropped probabilities for d V
    }
    // d P^T [key][query] = sum_dv V[key][dv] DO[query][dv]   (same register layout as the scores)
    f32x16 dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) dp[r] = 0.f;
    {
      constexpr int S = (HD + 1) / 2;
      const float* vr = QB + l.li * PQ + 2 * D + h * HD + l.lh;
      const float* dor = XHB + l.li * PX + h * HD + l.lh;
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const bool dv = (2 * s + 2 <= HD) || (2 * s + l.lh < HD);
        dp = mfma(dv ? vr[2 * s] : 0.f, dv ? dor[2 * s] : 0.f, dp);
      }
    }
    float dot = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      dp[r] *= mk[r];                               // through the dropout
      dot += dp[r] * p[r];
    }
    dot += __shfl_xor(dot, 32, 64);
    const float scale = 1.0f / sqrtf((float)HD);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = acc_row(r, l.lh);
      float v = p[r] * (dp[r] - dot) * scale;
      if (!(v == v) || key >= L || l.li >= L) v = 0.f;   // masked / padded entries: p = 0 (and no NaN from 0 * inf)
      ds[r] = v;
    }
  }
  __syncthreads();   // P visible (cross-lane), all waves
  if (l.wave < NH) {
    const int h = l.wave;
    float* P = PB + h * T * PP;
    // d V[key][dv] = sum_query Pd[query][key] DO[query][dv]
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int q = 2 * s + l.lh;
      const float a = P[q * PP + l.li];                                   // A[i = key][k = query]
      const float b = l.li < HD ? XHB[q * PX + h * HD + l.li] : 0.f;       // B[k = query][j = dv]
      acc = mfma(a, b, acc);
    }
    if (l.li < HD) {
#pragma unroll
      for (int r = 0; r < 16; ++r) DQB[acc_row(r, l.lh) * PG + 2 * D + h * HD + l.li] = acc[r];
    }
    // d Q[query][dd] = sum_key dS[query][key] K[key][dd]: A straight from the registers
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = acc_row(r, l.lh);
      const float b = l.li < HD ? QB[key * PQ + D + h * HD + l.li] : 0.f;
      acc = mfma(ds[r], b, acc);
    }
    if (l.li < HD) {
#pragma unroll
      for (int r = 0; r < 16; ++r) DQB[acc_row(r, l.lh) * PG + h * HD + l.li] = acc[r];
    }
  }
  __syncthreads();   // every lane has read P for d V
  if (l.wave < NH) {
    float* P = PB + l.wave * T * PP;
#pragma unroll
    for (int r = 0; r < 16; ++r) P[l.li * PP + acc_row(r, l.lh)] = ds[r];   // [query][key]
  }
  __syncthreads();
  if (l.wave < NH) {
    const int h = l.wave;
    float* P = PB + h * T * PP;
    // d K[key][dd] = sum_query dS[query][key] Q[query][dd]
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int q = 2 * s + l.lh;
      const float a = P[q * PP + l.li];
      const float b = l.li < HD ? QB[q * PQ + h * HD + l.li] : 0.f;
      acc = mfma(a, b, acc);
    }
    if (l.li < HD) {
#pragma unroll
      for (int r = 0; r < 16; ++r) DQB[acc_row(r, l.lh) * PG + D + h * HD + l.li] = acc[r];
    }
  }
  __syncthreads();
  for (int e = l.tid; e < T * 3 * D; e += 256) {
    const int t = e / (3 * D), c = e - t * (3 * D);
    if (t < L) gr.d_qkv[((size_t)t * LN + n) * (3 * D) + c] = DQB[t * PG + c];
  }
  // dx = DRB + DQB * W_in
  gemm_tiles<3 * D, false>(DQB, PG, w.in_w, D, l, wq, true, [&](int col, bool cv, const f32x16& acc) {
    if (!cv) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = acc_row(r, l.lh);
      if (t < L) dx[((size_t)t * LN + n) * D + col] = DRB[t * PX + col] + acc[r];
    }
  });
}

// ------------------------------------------------------------------------------------------------------------
template <typename F>
static inline bool txt_layer_visit(int D, int FF, int NH, int dec, F&& f) {
  if (NH != 2 || FF != 128) return false;
  if (D == 54 && !dec) { f(Geom<54, 128, 2, false>{}); return true; }
  if (D == 32 && dec) { f(Geom<32, 128, 2, true>{}); return true; }
  if (D == 32 && !dec) { f(Geom<32, 128, 2, false>{}); return true; }
  if (D == 54 && dec) { f(Geom<54, 128, 2, true>{}); return true; }
  if (D == 16 && dec) { f(Geom<16, 128, 2, true>{}); return true; }     // BASELINE configs[0]: latent_dim 16
  if (D == 24 && dec) { f(Geom<24, 128, 2, true>{}); return true; }     // the shipped config_cdspritesplus.yml: n_latents 24
  return false;
}

// csrc/txtwave.hip: the same layer as one wave per sequence, activations chained through registers (round 4, default).
// MMVAE_TXT_WAVE=0 keeps the workgroup-per-sequence kernels of this file (same saved tensors, same masks).
int txt_wave_fwd_dispatch(const float* x, const uint8_t* valid, const float* mem, float* y, const mmvae_txt_layer_w_t& wv,
                          const mmvae_txt_layer_saved_t& sv, const mmvae_txt_layer_drop_t& d, int L, int N, int D, int FF,
                          int NH, int dec, int time_mean, const float* head_w, const float* head_b, float* heads, int HN,
                          hipStream_t stream);
int txt_wave_bwd_dispatch(const float* dy, const uint8_t* valid, float* dx, float* dmem, const mmvae_txt_layer_w_t& wv,
                          const mmvae_txt_layer_saved_t& sv, const mmvae_txt_layer_grads_t& gv,
                          const mmvae_txt_layer_drop_t& d, int L, int N, int D, int FF, int NH, int dec, int time_mean,
                          hipStream_t stream);
// which kernels serve a layer: { forward on the wave kernels, backward on them from this many sequences (encoder-width
// layers, d <= 32 decoder layers) }.  Environment defaults, or mmvae_txt_layer_plan() (tests run both forms in one process).
static int g_txt_plan[3] = {-1, -1, -1};
static inline void txt_plan_init() {
  if (g_txt_plan[0] >= 0) return;
  const int m = getenv("MMVAE_TXT_WAVE") ? atoi(getenv("MMVAE_TXT_WAVE")) : 3;    // bit 0: forward, bit 1: backward
  g_txt_plan[0] = m & 1;
  const int never = 1 << 30;
  g_txt_plan[1] = !(m & 2) ? never : getenv("MMVAE_TXT_WAVE_BWD_MIN_N") ? atoi(getenv("MMVAE_TXT_WAVE_BWD_MIN_N")) : 384;
  g_txt_plan[2] = !(m & 2) ? never : getenv("MMVAE_TXT_WAVE_BWD_MIN_N_DEC") ? atoi(getenv("MMVAE_TXT_WAVE_BWD_MIN_N_DEC")) : g_txt_plan[1];
}
extern "C" int mmvae_txt_layer_plan(int fwd_wave, int bwd_min_n, int bwd_min_n_dec) {
  txt_plan_init();
  if (fwd_wave >= 0) g_txt_plan[0] = fwd_wave ? 1 : 0;
  if (bwd_min_n >= 0) g_txt_plan[1] = bwd_min_n;
  if (bwd_min_n_dec >= 0) g_txt_plan[2] = bwd_min_n_dec;
  return MMVAE_OK;
}

extern "C" int mmvae_txt_layer_supported(int L, int D, int FF, int NH, int dec) {
  if (L < 1 || L > tl::T) return 0;
  return txt_layer_visit(D, FF, NH, dec, [](auto) {}) ? 1 : 0;
}
extern "C" size_t mmvae_txt_layer_lnws_floats(int N, int D, int dec) { return (size_t)N * (dec ? 3 : 2) * 2 * D; }

extern "C" int mmvae_txt_layer_fwd(const float* x, const uint8_t* valid, const float* mem, float* y,
                                   const mmvae_txt_layer_w_t* w, const mmvae_txt_layer_saved_t* saved,
                                   const mmvae_txt_layer_drop_t* drop, int L, int N, int D, int FF, int NH, int dec,
                                   int time_mean, const float* head_w, const float* head_b, float* heads, int HN,
                                   mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(x && valid && y && w && saved && N > 0);
  if (head_w && !(time_mean && head_b && heads && HN > 0)) return MMVAE_ERR_ARG;
  if (dec && !mem) return MMVAE_ERR_ARG;
  if (L < 1 || L > tl::T) return MMVAE_ERR_UNSUPPORTED;
  mmvae_txt_layer_drop_t d;
  if (drop) d = *drop;
  else {
    const mmvae_dropout_t z = {nullptr, 0u, 0u, 0.f};
    d.attn = d.drop1 = d.xattn = d.drop2 = d.ffn = d.drop3 = z;
  }
  const mmvae_txt_layer_w_t wv = *w;
  const mmvae_txt_layer_saved_t sv = *saved;
  txt_plan_init();
  if (g_txt_plan[0] && !sv.probs)      // (the wave kernel does not export the attention weights)
    return txt_wave_fwd_dispatch(x, valid, mem, y, wv, sv, d, L, N, D, FF, NH, dec, time_mean, head_w, head_b, heads, HN,
                                 (hipStream_t)stream);
  if (!txt_layer_visit(D, FF, NH, dec, [&](auto g) {
        using G = decltype(g);
        hipLaunchKernelGGL((txt_layer_fwd_kernel<G>), dim3(N), dim3(256), 0, (hipStream_t)stream, x, valid, mem, y, wv, sv,
                           d, L, N, time_mean, head_w, head_b, heads, HN);
      }))
    return MMVAE_ERR_UNSUPPORTED;
  return mmvae_launch_status();
}

extern "C" int mmvae_txt_layer_bwd(const float* dy, const uint8_t* valid, float* dx, float* dmem,
                                   const mmvae_txt_layer_w_t* w, const mmvae_txt_layer_saved_t* saved,
                                   const mmvae_txt_layer_grads_t* grads, const mmvae_txt_layer_drop_t* drop, int L, int N,
                                   int D, int FF, int NH, int dec, int time_mean, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(dy && valid && dx && w && saved && grads && N > 0);
  if (dec && !dmem) return MMVAE_ERR_ARG;
  if (L < 1 || L > tl::T) return MMVAE_ERR_UNSUPPORTED;
  mmvae_txt_layer_drop_t d;
  if (drop) d = *drop;
  else {
    const mmvae_dropout_t z = {nullptr, 0u, 0u, 0.f};
    d.attn = d.drop1 = d.xattn = d.drop2 = d.ffn = d.drop3 = z;
  }
  const mmvae_txt_layer_w_t wv = *w;
  const mmvae_txt_layer_saved_t sv = *saved;
  const mmvae_txt_layer_grads_t gv = *grads;
  // the wave-per-sequence backward (csrc/txtwave.hip) wins once the sequences outnumber the CUs' 4-wave slots; below
  // that, the 54-wide encoder layer's chain is longer on one SIMD than spread over four (measured in the step: batch
  // 128 0.405 vs 0.415 ms with it, batch 512 1.108 vs 1.089, batch 1000 1.950 vs 1.865): from 384 sequences (MMVAE_TXT_WAVE_BWD_MIN_N)
  txt_plan_init();
  if (N >= ((dec && D <= 32) ? g_txt_plan[2] : g_txt_plan[1]))
    return txt_wave_bwd_dispatch(dy, valid, dx, dmem, wv, sv, gv, d, L, N, D, FF, NH, dec, time_mean, (hipStream_t)stream);
  if (!txt_layer_visit(D, FF, NH, dec, [&](auto g) {
        using G = decltype(g);
        hipLaunchKernelGGL((txt_layer_bwd_kernel<G>), dim3(N), dim3(256), 0, (hipStream_t)stream, dy, valid, dx, dmem, wv,
                           sv, gv, d, L, N, time_mean);
      }))
    return MMVAE_ERR_UNSUPPORTED;
  return mmvae_launch_status();
}

MMVAE_TRACE_SETTER(txtlayer)
