// Feed-forward block of the action Transformer towers as three fused launches (gfx950, fp32 MFMA):
//
//     y = W2 . dropout(gelu(W1 . x + b1)) + b2          d_model = 32, FF = a multiple of 32 (1024 in the reference)
//
// (torch.nn.TransformerEncoderLayer / DecoderLayer `linear2(dropout(activation(linear1(x))))`, reference
// models/encoders.py:706-716, models/decoders.py:589-600.)  Op by op the (rows x FF) hidden activation -- 52 MB for the
// reference's 12 800 rows -- is written by linear1, read and rewritten by the activation-dropout pass, read by linear2,
// and the same again twice in backward: ~0.5 GB of HBM traffic per layer and step.  Here it never leaves the registers:
// every wave produces 32 x 32 tiles of it with 16 MFMAs, applies bias / GELU / the dropout mask (same element index as
// mmvae_dropout_act: row * FF + column) on the accumulators and feeds them straight into the next MFMAs; backward
// recomputes the tiles (K = 32: cheaper than reading them).
//
// Operand trick used throughout: v_mfma_f32_32x32x2_f32 leaves C[i][j] in lane (j, half) register r with
// i = 8 (r >> 2) + 4 half + (r & 3).  The same register, read as the A operand of k-step s = r, supplies
// A[row = lane & 31][k = the accumulator's i]: as long as the B operand walks k in that same order (k_s(half) =
// 8 (s >> 2) + 4 half + (s & 3)) an accumulator tile is the next GEMM's A operand without a transposition through LDS.
// Computing a tile as mfma(A = W1 rows, B = x rows) or as mfma(A = x rows, B = W1 rows) -- the same operand registers
// with their roles swapped -- puts either the x row or the FF column on the lanes, whichever the consumer needs.
#include "common.hpp"

#define FFN_D 32

struct FfnArgs {
  const float *x, *dy, *w1, *b1, *w2, *b2;
  float *y, *dx, *ws;
  int M, FF, rows_per_slice, rowlen;
  mmvae_dropout_t drop;
};

// GELU and its derivative on the accumulators.  ocml's erff costs ~45 VALU instructions, which the op-by-op pass hides
// behind HBM; here they sit between MFMAs (same-box A/B at 12 800 x 1024: forward 48.6 -> 39.9 us, data gradient
// 59.7 -> 52.2, weight gradients 82.1 -> 65.4).  Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7 on erf, i.e. <= 1e-7 |x|
// on gelu: two orders below the 1e-4 parity bar) with the hardware exp / rcp; exp(-x^2 / 2) serves the erf tail AND
// the normal density of gelu'.  (Measured and dropped on top of it: register prefetch of the next chunk's operands --
// 180 / 256 VGPRs, forward 44.0, data gradient 64.3 us -- and the dropout hash with a hoisted index multiply and an
// integer threshold: no change.  The kernels are bound by the ~180 VALU cycles per hidden element -- two quarter-rate
// integer multiplies of the mask hash, exp and rcp of the GELU -- between the MFMAs, not by operand latency.)
__device__ __forceinline__ void ffn_gelu_parts(float x, float& tail, float& e) {
  // tail = 0.5 erfc(|x| / sqrt 2) = 0.5 poly(t) exp(-x^2 / 2), t = 1 / (1 + p |x| / sqrt 2)
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678f, ax, 1.0f));
  e = __expf(-0.5f * x * x);
  float poly = fmaf(t, 1.061405429f, -1.453152027f);
  poly = fmaf(t, poly, 1.421413741f);
  poly = fmaf(t, poly, -0.284496736f);
  poly = fmaf(t, poly, 0.254829592f);
  tail = 0.5f * (poly * t) * e;
}
__device__ __forceinline__ float ffn_gelu(float x) {
  float tail, e;
  ffn_gelu_parts(x, tail, e);
  return x * (x >= 0.f ? 1.0f - tail : tail);
}
__device__ __forceinline__ void ffn_gelu_both(float x, float& g, float& dg) {
  float tail, e;
  ffn_gelu_parts(x, tail, e);
  const float cdf = x >= 0.f ? 1.0f - tail : tail;
  g = x * cdf;
  dg = fmaf(x * 0.3989422804f, e, cdf);
}

// lane (li, lh) of a wave: values row[2 kk + lh], kk = 0..15, of the 32-float row at p (zeros if !ok)
__device__ __forceinline__ void ffn_row_pairs(const float* __restrict__ p, bool ok, int lh, float (&v)[16]) {
  float4 q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) q[j] = ok ? *reinterpret_cast<const float4*>(p + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    v[2 * j] = lh ? q[j].y : q[j].x;
    v[2 * j + 1] = lh ? q[j].w : q[j].z;
  }
}
__device__ __forceinline__ f32x16 ffn_zero() {
  f32x16 a;
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = 0.f;
  return a;
}
// the accumulator index i of register r in lane half lh
__device__ __forceinline__ int ffn_i(int r, int lh) { return 8 * (r >> 2) + 4 * lh + (r & 3); }

// sum of the 4 waves' 32 x 32 accumulators (register r <-> row ffn_i(r), lane li <-> column), + bias, -> out rows
__device__ __forceinline__ void ffn_reduce_store(const f32x16& acc, float* __restrict__ red, int wave, int lane,
                                                 float* __restrict__ out, int R0, int M, const float* __restrict__ bias,
                                                 const float* __restrict__ add = nullptr) {
  const int li = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
  __syncthreads();
  const float b = bias ? bias[li] : 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 4 * wave + i;
    const float s = (red[r * 64 + lane] + red[(16 + r) * 64 + lane]) + (red[(32 + r) * 64 + lane] + red[(48 + r) * 64 + lane]);
    const int row = R0 + ffn_i(r, lh);
    if (row < M) out[(size_t)row * FFN_D + li] = s + b + (add ? add[(size_t)row * FFN_D + li] : 0.f);
  }
}

// The same reduction with the LayerNorm that consumes the block as its epilogue (round 6): out = LayerNorm(dropout(y) + r),
// y = the summed tile + bias.  Thread (wave, li, lh) holds four rows' column li: a row's 32 columns are the 32 lanes of a
// half-wave, so the row statistics are five xor-shuffles each; saves xhat / rstd as mmvae_layernorm_residual_fwd does (same
// dropout mask: element row * 32 + column).
struct FfnLn {
  const float *r, *gamma, *beta;      // residual rows (M, 32), LayerNorm weight / bias
  float *xhat, *rstd;
  mmvae_dropout_t drop;
};
__device__ __forceinline__ float ffn_half_row_sum(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  v += __shfl_xor(v, 16, 64);
  return v;
}
__device__ __forceinline__ void ffn_reduce_ln_store(const f32x16& acc, float* __restrict__ red, int wave, int lane,
                                                    float* __restrict__ out, int R0, int M, const float* __restrict__ bias,
                                                    const FfnLn& ln) {
  const int li = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
  __syncthreads();
  const float b = bias[li], g = ln.gamma[li], bt = ln.beta[li];
  const DropKey dk = drop_key(ln.drop);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 4 * wave + i;
    const float s = (red[r * 64 + lane] + red[(16 + r) * 64 + lane]) + (red[(32 + r) * 64 + lane] + red[(48 + r) * 64 + lane]);
    const int row = R0 + ffn_i(r, lh);
    const bool ok = row < M;
    const size_t o = (size_t)(ok ? row : 0) * FFN_D + li;
    const float v = ok ? (s + b) * drop_mul(dk, (uint32_t)o) + ln.r[o] : 0.f;
    const float mean = ffn_half_row_sum(v) * (1.0f / 32.0f);
    const float d = v - mean;
    const float rs = rsqrtf(ffn_half_row_sum(d * d) * (1.0f / 32.0f) + 1e-5f);
    const float xh = d * rs;
    if (ok) {
      ln.xhat[o] = xh;
      out[o] = xh * g + bt;
      if (li == 0) ln.rstd[row] = rs;
    }
  }
}

// ---- forward: workgroup = one 32-row block, wave w = FF chunks w, w + 4, ... ------------------------------------------
__global__ __launch_bounds__(256) void ffn32_fwd_kernel(FfnArgs a) {
  __shared__ float red[4 * 16 * 64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int R0 = blockIdx.x * 32, FF = a.FF;
  const DropKey dk = drop_key(a.drop);
  float xv[16];
  ffn_row_pairs(a.x + (size_t)(R0 + li) * FFN_D, R0 + li < a.M, lh, xv);
  f32x16 acc2 = ffn_zero();
  const uint32_t drow = (uint32_t)(R0 + li) * (uint32_t)FF;
  for (int c0 = wave * 32; c0 < FF; c0 += 128) {
    float w1v[16];
    ffn_row_pairs(a.w1 + (size_t)(c0 + li) * FFN_D, true, lh, w1v);
    float4 bq[4], w2q[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      bq[q] = *reinterpret_cast<const float4*>(a.b1 + c0 + 8 * q + 4 * lh);
      w2q[q] = *reinterpret_cast<const float4*>(a.w2 + (size_t)li * FF + c0 + 8 * q + 4 * lh);
    }
    // h^T tile: register r <-> FF column c0 + ffn_i(r), lane li <-> x row
    f32x16 h = ffn_zero();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) h = __builtin_amdgcn_mfma_f32_32x32x2f32(w1v[kk], xv[kk], h, 0, 0, 0);
    uint32_t hh = 0u;      // registers r, r + 1 (r even) are consecutive columns of an even-aligned pair: one mask hash
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float bias = (&bq[r >> 2].x)[r & 3];
      if ((r & 1) == 0) hh = drop_pair_hash(dk, (drow + (uint32_t)(c0 + ffn_i(r, lh))) >> 1);
      const float act = ffn_gelu(h[r] + bias) * ((r & 1) ? drop_pair_hi(dk, hh) : drop_pair_lo(dk, hh));
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(act, (&w2q[r >> 2].x)[r & 3], acc2, 0, 0, 0);
    }
  }
  ffn_reduce_store(acc2, red, wave, lane, a.y, R0, a.M, a.b2);
}

// ---- backward, data gradient: dx = (dy W2 . mask . gelu'(h)) W1, same decomposition as forward ---------------------------
__global__ __launch_bounds__(256) void ffn32_bwd_data_kernel(FfnArgs a) {
  __shared__ float red[4 * 16 * 64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int R0 = blockIdx.x * 32, FF = a.FF;
  const DropKey dk = drop_key(a.drop);
  float xv[16], dyv[16];
  ffn_row_pairs(a.x + (size_t)(R0 + li) * FFN_D, R0 + li < a.M, lh, xv);
  ffn_row_pairs(a.dy + (size_t)(R0 + li) * FFN_D, R0 + li < a.M, lh, dyv);
  f32x16 accdx = ffn_zero();
  const uint32_t drow = (uint32_t)(R0 + li) * (uint32_t)FF;
  for (int c0 = wave * 32; c0 < FF; c0 += 128) {
    float w1v[16], w2t[16], w1n[16];
    ffn_row_pairs(a.w1 + (size_t)(c0 + li) * FFN_D, true, lh, w1v);
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) w2t[kk] = a.w2[(size_t)(2 * kk + lh) * FF + c0 + li];     // W2[out][ff = c0 + li]
#pragma unroll
    for (int s = 0; s < 16; ++s) w1n[s] = a.w1[(size_t)(c0 + ffn_i(s, lh)) * FFN_D + li];    // W1[ff_s][k = li]
    float4 bq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const float4*>(a.b1 + c0 + 8 * q + 4 * lh);
    f32x16 h = ffn_zero(), da = ffn_zero();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      h = __builtin_amdgcn_mfma_f32_32x32x2f32(w1v[kk], xv[kk], h, 0, 0, 0);
      da = __builtin_amdgcn_mfma_f32_32x32x2f32(w2t[kk], dyv[kk], da, 0, 0, 0);
    }
    uint32_t hh = 0u;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pre = h[r] + (&bq[r >> 2].x)[r & 3];
      float g_, dg;
      ffn_gelu_both(pre, g_, dg);
      if ((r & 1) == 0) hh = drop_pair_hash(dk, (drow + (uint32_t)(c0 + ffn_i(r, lh))) >> 1);
      const float dh = da[r] * ((r & 1) ? drop_pair_hi(dk, hh) : drop_pair_lo(dk, hh)) * dg;
      accdx = __builtin_amdgcn_mfma_f32_32x32x2f32(dh, w1n[r], accdx, 0, 0, 0);
    }
  }
  ffn_reduce_store(accdx, red, wave, lane, a.dx, R0, a.M, nullptr);
}

// ---- backward, weight gradients: wave = one 32-column FF chunk, walks the row blocks of its row slice ----------------
// partial row p (= row slice) of ws: [dW1 (FF x 32) | db1 (FF) | dW2 (32 x FF) | db2 (32)], rowlen floats apart
__global__ __launch_bounds__(256) void ffn32_bwd_weight_kernel(FfnArgs a) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int FF = a.FF, c0 = (blockIdx.x * 4 + wave) * 32;
  if (c0 >= FF) return;
  const DropKey dk = drop_key(a.drop);
  float w1v[16], w2t[16];
  ffn_row_pairs(a.w1 + (size_t)(c0 + li) * FFN_D, true, lh, w1v);
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) w2t[kk] = a.w2[(size_t)(2 * kk + lh) * FF + c0 + li];
  const float b1v = a.b1[c0 + li];
  f32x16 accw1 = ffn_zero(), accw2 = ffn_zero();
  float db1 = 0.f, db2 = 0.f;
  const bool want_db2 = blockIdx.x == 0 && wave == 0;
  const int r_beg = blockIdx.y * a.rows_per_slice, r_end = min(a.M, r_beg + a.rows_per_slice);
  for (int R0 = r_beg; R0 < r_end; R0 += 32) {
    float xv[16], dyv[16], xn[16], dyn[16];
    const bool rok = R0 + li < r_end;
    ffn_row_pairs(a.x + (size_t)(R0 + li) * FFN_D, rok, lh, xv);
    ffn_row_pairs(a.dy + (size_t)(R0 + li) * FFN_D, rok, lh, dyv);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int row = R0 + ffn_i(s, lh);
      const bool ok = row < r_end;
      xn[s] = ok ? a.x[(size_t)row * FFN_D + li] : 0.f;
      dyn[s] = ok ? a.dy[(size_t)row * FFN_D + li] : 0.f;
    }
    // h, da tiles: register r <-> row R0 + ffn_i(r), lane li <-> FF column c0 + li
    f32x16 h = ffn_zero(), da = ffn_zero();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      h = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[kk], w1v[kk], h, 0, 0, 0);
      da = __builtin_amdgcn_mfma_f32_32x32x2f32(dyv[kk], w2t[kk], da, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = R0 + ffn_i(r, lh);
      const float pre = h[r] + b1v;
      const float m = drop_mul(dk, (uint32_t)row * (uint32_t)FF + (uint32_t)(c0 + li));
      float g_, dg;
      ffn_gelu_both(pre, g_, dg);
      const float act = g_ * m;                               // rows >= r_end: multiplied by dy = 0 below
      const float dh = da[r] * m * dg;                        // ... and da = 0 there
      db1 += dh;
      db2 += dyn[r];
      accw1 = __builtin_amdgcn_mfma_f32_32x32x2f32(dh, xn[r], accw1, 0, 0, 0);     // dW1[ff][k = li]
      accw2 = __builtin_amdgcn_mfma_f32_32x32x2f32(dyn[r], act, accw2, 0, 0, 0);   // dW2[out][ff = c0 + li]
    }
  }
  float* __restrict__ part = a.ws + (size_t)blockIdx.y * a.rowlen;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    part[(size_t)(c0 + ffn_i(r, lh)) * FFN_D + li] = accw1[r];
    part[(size_t)FF * FFN_D + FF + (size_t)ffn_i(r, lh) * FF + c0 + li] = accw2[r];
  }
  db1 += __shfl_xor(db1, 32, 64);
  db2 += __shfl_xor(db2, 32, 64);
  if (lh == 0) {
    part[(size_t)FF * FFN_D + c0 + li] = db1;
    if (want_db2) part[(size_t)FF * FFN_D + FF + (size_t)FFN_D * FF + li] = db2;
  }
}

#include "ffn_b16.inc"

static inline bool ffn_al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
static inline int ffn_slices(int M, int FF) {
  // row slices of the weight-gradient launch: ~2048 waves in flight, 8 to 13 row blocks per wave
  const int nrb = (M + 31) / 32, chunks = FF / 32;
  int s = (2048 + chunks - 1) / chunks;
  const int lo = (nrb + 12) / 13;
  if (s < lo) s = lo;
  if (s > nrb) s = nrb;
  if (s > 512) s = 512;
  return s < 1 ? 1 : s;
}
extern "C" int mmvae_ffn32_supported(int d, int FF) { return d == FFN_D && FF >= 32 && FF % 32 == 0; }
extern "C" int mmvae_ffn32_bwd_parts(int M, int FF) { return ffn_slices(M, FF); }
extern "C" size_t mmvae_ffn32_bwd_rowlen(int FF) { return (size_t)FFN_D * FF + FF + (size_t)FFN_D * FF + FFN_D; }

extern "C" int mmvae_ffn32_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                               float* y, int M, int FF, const mmvae_dropout_t* drop, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(x && w1 && b1 && w2 && b2 && y && M > 0);
  if (!mmvae_ffn32_supported(FFN_D, FF) || (long)M * FF >= (1L << 32)) return MMVAE_ERR_UNSUPPORTED;
  if (!ffn_al16(x) || !ffn_al16(w1) || !ffn_al16(b1) || !ffn_al16(w2)) return MMVAE_ERR_ARG;
  FfnArgs a{x, nullptr, w1, b1, w2, b2, y, nullptr, nullptr, M, FF, 0, 0, drop_arg(drop)};
  hipLaunchKernelGGL(ffn32_fwd_kernel, dim3((M + 31) / 32), dim3(256), 0, (hipStream_t)stream, a);
  return mmvae_launch_status();
}
// dx (M x 32) and the partial rows of the weight / bias gradients: mmvae_ffn32_bwd_parts(M, FF) rows of
// mmvae_ffn32_bwd_rowlen(FF) floats in ws, each [dW1 (FF,32) | db1 (FF) | dW2 (32,FF) | db2 (32)]; every element of
// every row is written.  dx may be NULL (no data gradient wanted) and so may ws (data gradient only): the two launches are
// independent, a caller can put them on different streams.
extern "C" int mmvae_ffn32_bwd(const float* x, const float* dy, const float* w1, const float* b1, const float* w2,
                               float* dx, float* ws, int M, int FF, const mmvae_dropout_t* drop,
                               mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(x && dy && w1 && b1 && w2 && (ws || dx) && M > 0);
  if (!mmvae_ffn32_supported(FFN_D, FF) || (long)M * FF >= (1L << 32)) return MMVAE_ERR_UNSUPPORTED;
  if (!ffn_al16(x) || !ffn_al16(dy) || !ffn_al16(w1) || !ffn_al16(b1)) return MMVAE_ERR_ARG;
  const int S = ffn_slices(M, FF);
  int rps = ((M + S - 1) / S + 31) / 32 * 32;
  FfnArgs a{x, dy, w1, b1, w2, nullptr, nullptr, dx, ws, M, FF, rps, (int)mmvae_ffn32_bwd_rowlen(FF), drop_arg(drop)};
  if (dx) hipLaunchKernelGGL(ffn32_bwd_data_kernel, dim3((M + 31) / 32), dim3(256), 0, (hipStream_t)stream, a);
  if (ws) hipLaunchKernelGGL(ffn32_bwd_weight_kernel, dim3((FF / 32 + 3) / 4, S), dim3(256), 0, (hipStream_t)stream, a);
  return mmvae_launch_status();
}

// ---- the same three launches on split-bf16 MFMA (ffn_b16.inc) ---------------------------------------------------------
extern "C" size_t mmvae_ffn32_wsplit_bytes(int FF) { return (size_t)4 * 3 * FFN_D * 2 * (size_t)FF; }
extern "C" size_t mmvae_ffn32_rsplit_bytes(int M) { return (size_t)4 * 3 * FFN_D * 2 * (size_t)((M + 31) / 32 * 32); }
extern "C" int mmvae_ffn32_prep_weights_many(const float* const* w1, const float* const* w2, void* const* wsplit, int n,
                                             int FF, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(w1 && w2 && wsplit && n > 0 && n <= FFN_PREP_MAX);
  if (!mmvae_ffn32_supported(FFN_D, FF)) return MMVAE_ERR_UNSUPPORTED;
  FfnPrepJobs jobs;
  for (int i = 0; i < FFN_PREP_MAX; ++i) {
    const int j = i < n ? i : 0;
    MMVAE_CHECK_ARG(w1[j] && w2[j] && wsplit[j] && ffn_al16(wsplit[j]));
    jobs.w1[i] = w1[j];
    jobs.w2[i] = w2[j];
    jobs.out[i] = (unsigned short*)wsplit[j];
  }
  hipLaunchKernelGGL(ffn32_prep_weights_kernel, dim3((16 * FF + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, jobs, FF);
  return mmvae_launch_status();
}
extern "C" int mmvae_ffn32_prep_weights(const float* w1, const float* w2, void* wsplit, int FF, mmvae_stream_t stream) {
  return mmvae_ffn32_prep_weights_many(&w1, &w2, &wsplit, 1, FF, stream);
}
extern "C" int mmvae_ffn32_fwd_b16(const float* x, const void* wsplit, const float* b1, const float* b2, float* y, int M,
                                   int FF, const mmvae_dropout_t* drop, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(x && wsplit && b1 && b2 && y && M > 0);
  if (!mmvae_ffn32_supported(FFN_D, FF) || (long)M * FF >= (1L << 32)) return MMVAE_ERR_UNSUPPORTED;
  if (!ffn_al16(x) || !ffn_al16(wsplit) || !ffn_al16(b1)) return MMVAE_ERR_ARG;
  FfnB16Args a{x, nullptr, b1, b2, (const unsigned short*)wsplit, nullptr, y, nullptr, nullptr, M, 0, FF, 0, 0, drop_arg(drop)};
  hipLaunchKernelGGL(ffn32_fwd_b16_kernel<false>, dim3((M + 31) / 32), dim3(256), 0, (hipStream_t)stream, a, FfnLn{});
  return mmvae_launch_status();
}
// ... with the LayerNorm behind the block as the launch's epilogue: y = LayerNorm(dropout(ffn(x)) + r) (models/encoders.py:
// 706-716: `src = norm2(src + dropout2(ff))`), xhat / rstd saved as mmvae_layernorm_residual_fwd saves them
extern "C" int mmvae_ffn32_fwd_b16_ln(const float* x, const void* wsplit, const float* b1, const float* b2, const float* r,
                                      const float* gamma, const float* beta, float* y, float* xhat, float* rstd, int M, int FF,
                                      const mmvae_dropout_t* drop, const mmvae_dropout_t* ln_drop, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(x && wsplit && b1 && b2 && r && gamma && beta && y && xhat && rstd && M > 0);
  if (!mmvae_ffn32_supported(FFN_D, FF) || (long)M * FF >= (1L << 32)) return MMVAE_ERR_UNSUPPORTED;
  if (!ffn_al16(x) || !ffn_al16(wsplit) || !ffn_al16(b1)) return MMVAE_ERR_ARG;
  FfnB16Args a{x, nullptr, b1, b2, (const unsigned short*)wsplit, nullptr, y, nullptr, nullptr, M, 0, FF, 0, 0, drop_arg(drop)};
  FfnLn ln{r, gamma, beta, xhat, rstd, drop_arg(ln_drop)};
  hipLaunchKernelGGL(ffn32_fwd_b16_kernel<true>, dim3((M + 31) / 32), dim3(256), 0, (hipStream_t)stream, a, ln);
  return mmvae_launch_status();
}
// as mmvae_ffn32_bwd; rsplit (mmvae_ffn32_rsplit_bytes(M) bytes, needed with ws) receives the split images of x and dy
// that the weight-gradient launch reads -- scratch of THIS call, in stream order
extern "C" int mmvae_ffn32_bwd_b16(const float* x, const float* dy, const void* wsplit, const float* b1, float* dx, float* ws,
                                   void* rsplit, const float* dx_add, int M, int FF, const mmvae_dropout_t* drop,
                                   mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(x && dy && wsplit && b1 && (ws || dx) && (!ws || rsplit) && M > 0);
  if (!mmvae_ffn32_supported(FFN_D, FF) || (long)M * FF >= (1L << 32)) return MMVAE_ERR_UNSUPPORTED;
  if (!ffn_al16(x) || !ffn_al16(dy) || !ffn_al16(wsplit) || !ffn_al16(b1) || !ffn_al16(rsplit)) return MMVAE_ERR_ARG;
  const int S = ffn_slices(M, FF), Mpad = (M + 31) / 32 * 32;
  int rps = ((M + S - 1) / S + 31) / 32 * 32;
  FfnB16Args a{x, dy, b1, dx_add, (const unsigned short*)wsplit, (const unsigned short*)rsplit, nullptr, dx, ws, M, Mpad, FF,
               rps, (int)mmvae_ffn32_bwd_rowlen(FF), drop_arg(drop)};
  hipStream_t st = (hipStream_t)stream;
  if (dx) hipLaunchKernelGGL(ffn32_bwd_data_b16_kernel, dim3((M + 31) / 32), dim3(256), 0, st, a);
  if (ws) {
    hipLaunchKernelGGL(ffn32_prep_rows_kernel, dim3((16 * Mpad + 255) / 256), dim3(256), 0, st, x, dy, (unsigned short*)rsplit, M,
                       Mpad);
    hipLaunchKernelGGL(ffn32_bwd_weight_b16_kernel, dim3((FF / 32 + 3) / 4, S), dim3(256), 0, st, a);
  }
  return mmvae_launch_status();
}

// ---- out_proj + dropout + residual + LayerNorm of a d_model-32 Transformer layer in ONE launch (round 6) ----------------
//   y = LayerNorm( dropout(x W^T + b) + r ),  x, r, y (M, 32), W (32, 32) [out][in]     (models/encoders.py:706-716:
//   `src = norm1(src + dropout1(self_attn(...)))` of nn.TransformerEncoderLayer; the attention's out_proj is the Linear)
// It replaces a 12 800 x 32 x 32 GEMM launch (4.9 - 10.9 us on the generic bodies) and the LayerNorm launch behind it (5.1 us)
// with one pass over the rows: a wave owns 32 rows, lane (li, lh) = row R0 + li; the product is 16 fp32 MFMAs with the
// TOKENS as the B operand, so that the accumulator leaves row li's 32 outputs in lanes li and li + 32 (16 each) -- the
// LayerNorm's row sums are in-lane adds + one exchange between the wave's halves, the stores four float4 per lane.
// Saves xhat / rstd exactly as mmvae_layernorm_residual_fwd does (same dropout mask: element row * 32 + column), so the
// backward is the two existing launches (LayerNorm backward, Linear backward on x and W).
__global__ __launch_bounds__(256) void proj32_ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                            const float* __restrict__ b, const float* __restrict__ r,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ y, float* __restrict__ xhat,
                                                            float* __restrict__ rstd, int M, mmvae_dropout_t drop) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int row = (blockIdx.x * 4 + wave) * 32 + li;
  const bool ok = row < M;
  const DropKey dk = drop_key(drop);
  float xv[16], wv[16];
  ffn_row_pairs(x + (size_t)(ok ? row : 0) * FFN_D, ok, lh, xv);
  ffn_row_pairs(W + (size_t)li * FFN_D, true, lh, wv);
  f32x16 acc = ffn_zero();
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[kk], xv[kk], acc, 0, 0, 0);
  // register r of lane (li, lh) = output column ffn_i(r, lh) of row li: four runs of four consecutive columns 8 q + 4 lh
  const size_t o = (size_t)(ok ? row : 0) * FFN_D + 4 * lh;
  float v[16];
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 bq = *reinterpret_cast<const float4*>(b + 8 * q + 4 * lh);
    const float4 rq = ok ? *reinterpret_cast<const float4*>(r + o + 8 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 m = make_float4(1.f, 1.f, 1.f, 1.f);
    if (dk.on) {
      const uint32_t idx = (uint32_t)(o + 8 * q);
      const uint32_t h0 = drop_pair_hash(dk, idx >> 1), h1 = drop_pair_hash(dk, (idx >> 1) + 1);
      m = make_float4(drop_pair_lo(dk, h0), drop_pair_hi(dk, h0), drop_pair_lo(dk, h1), drop_pair_hi(dk, h1));
    }
    v[4 * q] = (acc[4 * q] + bq.x) * m.x + rq.x;
    v[4 * q + 1] = (acc[4 * q + 1] + bq.y) * m.y + rq.y;
    v[4 * q + 2] = (acc[4 * q + 2] + bq.z) * m.z + rq.z;
    v[4 * q + 3] = (acc[4 * q + 3] + bq.w) * m.w + rq.w;
    s += (v[4 * q] + v[4 * q + 1]) + (v[4 * q + 2] + v[4 * q + 3]);
  }
  s += __shfl_xor(s, 32, 64);
  const float mean = s * (1.0f / 32.0f);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    v[i] -= mean;
    ss += v[i] * v[i];
  }
  ss += __shfl_xor(ss, 32, 64);
  const float rs = rsqrtf(ss * (1.0f / 32.0f) + 1e-5f);
  if (ok) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 g = *reinterpret_cast<const float4*>(gamma + 8 * q + 4 * lh);
      const float4 bt = *reinterpret_cast<const float4*>(beta + 8 * q + 4 * lh);
      const float4 xh = make_float4(v[4 * q] * rs, v[4 * q + 1] * rs, v[4 * q + 2] * rs, v[4 * q + 3] * rs);
      *reinterpret_cast<float4*>(xhat + o + 8 * q) = xh;
      *reinterpret_cast<float4*>(y + o + 8 * q) = make_float4(xh.x * g.x + bt.x, xh.y * g.y + bt.y, xh.z * g.z + bt.z, xh.w * g.w + bt.w);
    }
    if (lh == 0) rstd[row] = rs;
  }
}
extern "C" int mmvae_proj32_ln_fwd(const float* x, const float* w, const float* b, const float* r, const float* gamma,
                                   const float* beta, float* y, float* xhat, float* rstd, int M, const mmvae_dropout_t* drop,
                                   mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(x && w && b && r && gamma && beta && y && xhat && rstd && M > 0 && (long)M * FFN_D < (1L << 32));
  if (!ffn_al16(x) || !ffn_al16(w) || !ffn_al16(b) || !ffn_al16(r) || !ffn_al16(gamma) || !ffn_al16(beta) || !ffn_al16(y) ||
      !ffn_al16(xhat))
    return MMVAE_ERR_ARG;
  hipLaunchKernelGGL(proj32_ln_fwd_kernel, dim3((M + 127) / 128), dim3(256), 0, (hipStream_t)stream, x, w, b, r, gamma, beta, y,
                     xhat, rstd, M, drop_arg(drop));
  return mmvae_launch_status();
}

// ---- y = x W^T + b for a 32-wide input and N = 32 / 64 / 96 / 128 outputs over many rows (round 6) -----------------------
// The action towers' packed QKV projection (12 800 x 32 -> 96: models/encoders.py:706-716 via nn.MultiheadAttention's in_proj)
// ran 10.9 us on the generic 128 x 32-tile GEMM.  Same mapping as proj32_ln_fwd_kernel: a wave owns 32 rows (lane = row, the
// tokens are the MFMA's B operand), one 32 x 32 x 32 product per 32 output columns, four float4 stores per lane and tile.
template <int NT>      // NT = N / 32 output tiles; every tile's weight rows are in flight before the first MFMA (with the
                        // loads inside the tile loop a wave paid the L2 round trip once per tile: 12.5 us whatever M)
__global__ __launch_bounds__(256) void proj32_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                         const float* __restrict__ b, float* __restrict__ y, int M, int N) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int rbase = (blockIdx.x * 4 + wave) * 32, row = rbase + li;
  const bool ok = row < M;
  float xv[16], wv[NT][16], bv[NT];
  ffn_row_pairs(x + (size_t)(ok ? row : 0) * FFN_D, ok, lh, xv);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    ffn_row_pairs(W + (size_t)(32 * t + li) * FFN_D, true, lh, wv[t]);
    bv[t] = b ? b[32 * t + li] : 0.f;
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    // rows as the A operand: register r of lane (li, lh) = y[row ffn_i(r, lh)][column 32 t + li] -- a store instruction writes
    // 32 consecutive floats of two rows
    f32x16 acc = ffn_zero();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[kk], wv[t][kk], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rr = rbase + ffn_i(r, lh);
      if (rr < M) y[(size_t)rr * N + 32 * t + li] = acc[r] + bv[t];
    }
  }
}
// (called from mmvae_linear_fwd: csrc/gemm.hip)
int mmvae_proj32_fwd_launch(const float* x, const float* w, const float* b, float* y, int M, int N, mmvae_stream_t stream) {
  const dim3 grid((M + 127) / 128);
  hipStream_t st = (hipStream_t)stream;
  switch (N / 32) {
    case 1: hipLaunchKernelGGL(proj32_fwd_kernel<1>, grid, dim3(256), 0, st, x, w, b, y, M, N); break;
    case 2: hipLaunchKernelGGL(proj32_fwd_kernel<2>, grid, dim3(256), 0, st, x, w, b, y, M, N); break;
    case 3: hipLaunchKernelGGL(proj32_fwd_kernel<3>, grid, dim3(256), 0, st, x, w, b, y, M, N); break;
    default: hipLaunchKernelGGL(proj32_fwd_kernel<4>, grid, dim3(256), 0, st, x, w, b, y, M, N); break;
  }
  return mmvae_launch_status();
}
