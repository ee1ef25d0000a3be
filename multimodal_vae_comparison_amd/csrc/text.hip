// Text-tower kernels for gfx950: embedding + positional-encoding quirk, masked softmax attention for
// sequences <= 64, residual LayerNorm, time reductions, decoder output permute+mask.
// All of these are tiny (T <= 64, d <= 64): one wavefront per row / per (sample, head), wave-shuffle
// reductions, LDS only for the attention tiles.  No atomics: parameter gradients go through per-block
// partial rows + mmvae_reduce_rows.
#include "common.hpp"

// ---------------------------------------------------------------------------------------------
// Embedding(one-hot.long()) + PositionalEncoding (models/encoders.py:833-835, nn_modules.py:430-438)
//   mode 0: out[((t*B+b)*V+v)*2+e] = emb[oh[b,t,v]][e] + pe[b][e]
//   mode 1: out[((b*T+t)*V+v)*2+e] = emb[oh[b,t,v]][e] + pe[B==1 ? 0 : t][e]      (memory relabelled as (T,B,2V))
// B0 (round 5): the one-hot tensor has B0 <= B rows and output row b reads row b % B0 of it AND of the table: R = B / B0
// passes of the SAME batch as one call keep every sample's positional term (indexed by its position in the ORIGINAL batch,
// nn_modules.py:432-438) -- POE's per-subset encoder passes (mmvae_models.py:159-187) without materialising the repeat.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_pe_fwd_kernel(const float* __restrict__ oh, const float* __restrict__ emb,
                                                           const float* __restrict__ pe, float* __restrict__ out, int B,
                                                           int T, int V, int mode, int B0, mmvae_dropout_t drop) {
  const int n = B * T * V, n0 = B0 * T * V;
  MMVAE_TRACE_STAMP(30);
  const DropKey dk = drop_key(drop);
  const float e00 = emb[0], e01 = emb[1], e10 = emb[2], e11 = emb[3];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int v = i % V, bt = i / V, t = bt % T, b = bt / T;
    const bool one = oh[i % n0] != 0.f;  // .long() of a 0/1 float
    const int pos = mode == 0 ? b % B0 : (B0 == 1 ? 0 : t);
    const size_t o = mode == 0 ? (((size_t)t * B + b) * V + v) * 2 : (size_t)i * 2;
    float2 r;
    r.x = ((one ? e10 : e00) + pe[pos * 2]) * drop_mul(dk, (uint32_t)o);
    r.y = ((one ? e11 : e01) + pe[pos * 2 + 1]) * drop_mul(dk, (uint32_t)o + 1u);
    *reinterpret_cast<float2*>(out + o) = r;
  }
}

// per-block partial of demb rows 0/1: ws[block][4] = {sum dx0 | oh=0, sum dx1 | oh=0, sum dx0 | oh=1, sum dx1 | oh=1}
__global__ __launch_bounds__(256) void embed_pe_bwd_kernel(const float* __restrict__ oh, const float* __restrict__ dx,
                                                           float* __restrict__ ws, int B, int T, int V, int mode,
                                                           int B0, mmvae_dropout_t drop) {
  __shared__ float red[4];
  MMVAE_TRACE_STAMP(31);
  const int n = B * T * V, n0 = B0 * T * V;
  const DropKey dk = drop_key(drop);
  float a00 = 0.f, a01 = 0.f, a10 = 0.f, a11 = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int v = i % V, bt = i / V, t = bt % T, b = bt / T;
    const size_t o = mode == 0 ? (((size_t)t * B + b) * V + v) * 2 : (size_t)i * 2;
    float2 g = *reinterpret_cast<const float2*>(dx + o);
    g.x *= drop_mul(dk, (uint32_t)o);
    g.y *= drop_mul(dk, (uint32_t)o + 1u);
    if (oh[i % n0] != 0.f) { a10 += g.x; a11 += g.y; } else { a00 += g.x; a01 += g.y; }
  }
  a00 = block_sum_256(a00, red);
  a01 = block_sum_256(a01, red);
  a10 = block_sum_256(a10, red);
  a11 = block_sum_256(a11, red);
  if (threadIdx.x == 0) {
    float* w = ws + (size_t)blockIdx.x * 4;
    w[0] = a00; w[1] = a01; w[2] = a10; w[3] = a11;
  }
}

static inline int embed_blocks(int B, int T, int V) {
  long n = (long)B * T * V;
  long b = (n + 255) / 256;
  return (int)(b > 256 ? 256 : (b < 1 ? 1 : b));
}
extern "C" int mmvae_embed_bwd_rows(int B, int T, int V) { return embed_blocks(B, T, V); }
extern "C" size_t mmvae_embed_ws_floats(int B, int T, int V) { return (size_t)embed_blocks(B, T, V) * 4; }

extern "C" int mmvae_embed_pe_fwd(const float* onehot, const float* emb, const float* pe, float* x, int B, int T,
                                  int V, int mode, int B0, const mmvae_dropout_t* drop, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(onehot && emb && pe && x && B > 0 && T > 0 && V > 1 && B0 > 0 && B % B0 == 0);
  MMVAE_CHECK_ARG(B0 == B || mode == 0);      // repeated passes only in the batch-indexed branch (the other relabels memory)
  hipLaunchKernelGGL(embed_pe_fwd_kernel, dim3(embed_blocks(B, T, V)), dim3(256), 0, (hipStream_t)stream, onehot, emb,
                     pe, x, B, T, V, mode, B0, drop_arg(drop));
  return mmvae_launch_status();
}
extern "C" int mmvae_embed_pe_bwd(const float* onehot, const float* dx, float* demb, float* ws, int B, int T, int V,
                                  int mode, int B0, int accumulate, const mmvae_dropout_t* drop, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(onehot && dx && ws && B > 0 && T > 0 && V > 1 && B0 > 0 && B % B0 == 0 && (B0 == B || mode == 0));
  MMVAE_CHECK_ARG(accumulate == MMVAE_ACC_DEFER || demb);
  const int nb = embed_blocks(B, T, V);
  hipLaunchKernelGGL(embed_pe_bwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, onehot, dx, ws, B, T, V, mode,
                     B0, drop_arg(drop));
  int rc = mmvae_launch_status();
  if (rc || accumulate == MMVAE_ACC_DEFER) return rc;
  if (!accumulate && V > 2) {
    rc = mmvae_fill(demb + 4, (long)(V - 2) * 2, 0.f, stream);  // rows >= 2 are never looked up
    if (rc) return rc;
  }
  return mmvae_reduce_rows(ws, demb, nb, 4, 4, accumulate, stream);
}

// ---------------------------------------------------------------------------------------------
// Attention core of nn.MultiheadAttention for L, S <= 128, head_dim <= 32.  One workgroup of AM threads (AM = 64 or
// 128: one or two wavefronts) per (sample n, head h); thread = query row or key column.  Rows of K / V / Q / dO live
// in LDS with pitch 36 floats and are consumed as broadcast ds_read_b128 (all lanes read the same row => conflict
// free); the L x S score / probability tile lives in LDS with pitch AM + 1 (thread = row or thread = column are both
// conflict free).  The 128 form serves the action towers' Ta = 100 sequences (models/datasets.py:887).
// ---------------------------------------------------------------------------------------------
#define ATT_MAX 128
// head_dim <= HD (template: 32 for the text towers' 27, 16 for the action towers' D/2 = 16 -- half the LDS reads and
// FMAs of the padded form); padded columns are zero; row pitch HD + 4 floats (16-byte aligned rows)
#define ATT_HD_MAX 32

// rows x 32 floats -> LDS, 8 independent loads in flight per thread (a load -> store loop waits one memory
// latency per iteration)
template <int NT, int HD>
__device__ __forceinline__ void att_stage_rows(float* __restrict__ dst, const float* __restrict__ src, int rows, int N,
                                               int n, long ld, int col0, int hd, int tid) {
  const int total = rows * HD;
  if ((hd & 3) == 0 && (col0 & 3) == 0 && (ld & 3) == 0 && (((uintptr_t)src) & 15) == 0) {
    // 16-byte form (the action towers: head_dim 16, T = 100): one round of loads instead of four -- with 128 threads
    // per workgroup the staging's memory round trips, not the arithmetic, were most of the kernel
    const int q4 = hd >> 2, tot4 = rows * q4;
    for (int e0 = 0; e0 < tot4; e0 += NT * 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u * NT + tid;
        const int r = e / q4, c = e - r * q4;
        v[u] = *reinterpret_cast<const float4*>(src + (e < tot4 ? ((size_t)r * N + n) * ld + col0 + 4 * c
                                                                 : (size_t)n * ld + col0));
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u * NT + tid;
        if (e < tot4) {
          const int r = e / q4, c = e - r * q4;
          *reinterpret_cast<float4*>(dst + r * (HD + 4) + 4 * c) = v[u];
        }
      }
    }
    // zero padding columns hd .. 31
    const int padq = (HD - hd) >> 2;
    for (int e = tid; e < rows * padq; e += NT) {
      const int r = e / padq, c = e - r * padq;
      *reinterpret_cast<float4*>(dst + r * (HD + 4) + hd + 4 * c) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return;
  }
  for (int e0 = 0; e0 < total; e0 += NT * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u * NT + tid;
      const int r = e / HD, d = e - r * HD;
      const bool ok = e < total && d < hd;
      v[u] = src[ok ? ((size_t)r * N + n) * ld + col0 + d : (size_t)n * ld + col0];
      v[u] = ok ? v[u] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u * NT + tid;
      if (e < total) dst[(e / HD) * (HD + 4) + (e % HD)] = v[u];
    }
  }
}
// L x S probability tile -> LDS (pitch SP), same batching
template <int NT, int SP>
__device__ __forceinline__ void att_stage_tile(float* __restrict__ dst, const float* __restrict__ src, int L, int S,
                                               int tid) {
  const int total = L * S;
  const float invS = 1.0f / (float)S;
  if ((S & 3) == 0 && (((uintptr_t)src) & 15) == 0) {
    const int tot4 = total >> 2, s4 = S >> 2;
    for (int e0 = 0; e0 < tot4; e0 += NT * 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u * NT + tid;
        v[u] = reinterpret_cast<const float4*>(src)[e < tot4 ? e : 0];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u * NT + tid;
        if (e < tot4) {
          const int l = e / s4, c = e - l * s4;
          float* d = dst + l * SP + 4 * c;
          d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
        }
      }
    }
    return;
  }
  for (int e0 = 0; e0 < total; e0 += NT * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u * NT + tid;
      v[u] = src[e < total ? e : 0];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u * NT + tid;
      if (e < total) {
        const int l = (int)(((float)e + 0.5f) * invS);
        dst[l * SP + (e - l * S)] = v[u];
      }
    }
  }
}
template <int HD>
__device__ __forceinline__ float att_dot(const float* __restrict__ row, const float (&x)[HD]) {
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
  for (int d = 0; d < HD; d += 4) {
    const float4 k4 = *reinterpret_cast<const float4*>(row + d);
    a0 += x[d] * k4.x;
    a1 += x[d + 1] * k4.y;
    a2 += x[d + 2] * k4.z;
    a3 += x[d + 3] * k4.w;
  }
  return (a0 + a1) + (a2 + a3);
}
template <int HD>
__device__ __forceinline__ void att_axpy(float (&acc)[HD], float p, const float* __restrict__ row) {
#pragma unroll
  for (int d = 0; d < HD; d += 4) {
    const float4 v4 = *reinterpret_cast<const float4*>(row + d);
    acc[d] += p * v4.x;
    acc[d + 1] += p * v4.y;
    acc[d + 2] += p * v4.z;
    acc[d + 3] += p * v4.w;
  }
}

// Round 5: the saved probabilities carry the attention-weight dropout decision in their SIGN bit (probabilities are >= 0;
// a dropped weight is stored negated, -0.0 included), so that no backward kernel has to regenerate the mask: the MFMA
// backward spent ~190 mask hashes per lane on it, ~6.5 of its 28 us at the action towers' shape.
__device__ __forceinline__ float at_signed(float p, float m) {
  return m == 0.f ? __uint_as_float(__float_as_uint(p) | 0x80000000u) : p;
}
__device__ __forceinline__ float at_mask_of(float v, float inv_keep) { return (__float_as_uint(v) >> 31) ? 0.f : inv_keep; }

template <int AM, int HD>
__global__ __launch_bounds__(AM) void attn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                      const float* __restrict__ v, const uint8_t* __restrict__ kpm,
                                                      float* __restrict__ out, float* __restrict__ probs, int L, int S,
                                                      int N, int H, int hd, long ldq, long ldk, long ldv,
                                                      int mask_is_valid, mmvae_dropout_t drop) {
  constexpr int SP = AM + 1;
  __shared__ __attribute__((aligned(16))) float sk[AM * (HD + 4)];
  __shared__ __attribute__((aligned(16))) float sv[AM * (HD + 4)];
  __shared__ float sp[AM * SP];
  __shared__ float smask[AM];
  __shared__ float sinv[AM];
  const int n = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
  att_stage_rows<AM, HD>(sk, k, S, N, n, ldk, h * hd, hd, lane);
  att_stage_rows<AM, HD>(sv, v, S, N, n, ldv, h * hd, hd, lane);
  // kpm bytes: 1 = ignore this key (key_padding_mask) or, with mask_is_valid, the batch's own validity mask
  // (1 = real token) read in place -- no conversion kernel
  if (lane < S) smask[lane] = (kpm && ((kpm[(size_t)n * S + lane] != 0) != (mask_is_valid != 0))) ? 1.f : 0.f;
  __syncthreads();
  const float scale = 1.0f / sqrtf((float)hd);
  if (lane < L) {
    float qr[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) qr[d] = d < hd ? q[((size_t)lane * N + n) * ldq + h * hd + d] * scale : 0.f;
    float mx = -INFINITY;
    for (int s = 0; s < S; ++s) {
      float sc = att_dot<HD>(sk + s * (HD + 4), qr);
      if (smask[s] != 0.f) sc = -INFINITY;
      sp[lane * SP + s] = sc;
      mx = fmaxf(mx, sc);
    }
    float o[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = 0.f;
    float sum = 0.f;
    const DropKey dk = drop_key(drop);   // dropout on the attention weights (after the softmax, before P V)
    const uint32_t drow = (uint32_t)((((size_t)n * H + h) * L + lane) * S);
    for (int s = 0; s < S; ++s) {
      const float p = expf(sp[lane * SP + s] - mx);
      sp[lane * SP + s] = p;
      sum += p;
      att_axpy<HD>(o, p * drop_mul(dk, drow + s), sv + s * (HD + 4));
    }
    const float inv = 1.0f / sum;
    sinv[lane] = inv;
    float* orow = out + ((size_t)lane * N + n) * ((size_t)H * hd) + h * hd;
#pragma unroll
    for (int d = 0; d < HD; ++d)
      if (d < hd) orow[d] = o[d] * inv;
  }
  __syncthreads();
  // normalised probabilities, written row by row with lane = key index (coalesced)
  float* P = probs + ((size_t)n * H + h) * L * S;
  if (lane < S) {
    const DropKey dk = drop_key(drop);
    for (int l = 0; l < L; ++l)
      P[(size_t)l * S + lane] = at_signed(sp[l * SP + lane] * sinv[l], drop_mul(dk, (uint32_t)((((size_t)n * H + h) * L + l) * S + lane)));
  }
}

// dV[s] = sum_l P[l,s] dO[l];  dP[l,s] = dO[l].V[s];  dS = P (dP - sum_s P dP);
// dQ[l] = scale * sum_s dS[l,s] K[s];  dK[s] = scale * sum_l dS[l,s] Q[l]
// ONE L x S tile in LDS: column pass for dV on P, row pass that overwrites P with dS (the dP dot products are computed
// twice instead of being parked in a second tile: 2 x 66 KB would not fit beside the four row arrays at AM = 128),
// column pass for dK on dS.
template <int AM, int HD>
__global__ __launch_bounds__(AM) void attn_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                      const float* __restrict__ v, const float* __restrict__ probs,
                                                      const float* __restrict__ dout, float* __restrict__ dq,
                                                      float* __restrict__ dk, float* __restrict__ dv, int L, int S,
                                                      int N, int H, int hd, long ldq, long ldk, long ldv,
                                                      mmvae_dropout_t drop) {
  constexpr int SP = AM + 1;
  __shared__ __attribute__((aligned(16))) float sq[AM * (HD + 4)];
  __shared__ __attribute__((aligned(16))) float sk[AM * (HD + 4)];
  __shared__ __attribute__((aligned(16))) float sv[AM * (HD + 4)];
  __shared__ __attribute__((aligned(16))) float sdo[AM * (HD + 4)];
  __shared__ float sp[AM * SP];
  const int n = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
  const long E = (long)H * hd;
  att_stage_rows<AM, HD>(sk, k, S, N, n, ldk, h * hd, hd, lane);
  att_stage_rows<AM, HD>(sv, v, S, N, n, ldv, h * hd, hd, lane);
  att_stage_rows<AM, HD>(sq, q, L, N, n, ldq, h * hd, hd, lane);
  att_stage_rows<AM, HD>(sdo, dout, L, N, n, E, h * hd, hd, lane);
  att_stage_tile<AM, SP>(sp, probs + ((size_t)n * H + h) * L * S, L, S, lane);
  __syncthreads();
  const float scale = 1.0f / sqrtf((float)hd);
  const DropKey dkey = drop_key(drop);
  if (lane < S) {     // dV sees the dropped weights
    const int s = lane;
    float dvr[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) dvr[d] = 0.f;
    for (int l = 0; l < L; ++l) {
      const float pv = sp[l * SP + s];
      att_axpy<HD>(dvr, fabsf(pv) * at_mask_of(pv, dkey.inv_keep), sdo + l * (HD + 4));
    }
    float* dvrow = dv + ((size_t)s * N + n) * ldv + h * hd;
#pragma unroll
    for (int d = 0; d < HD; ++d)
      if (d < hd) dvrow[d] = dvr[d];
  }
  __syncthreads();
  if (lane < L) {
    const int l = lane;
    float dor[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) dor[d] = sdo[l * (HD + 4) + d];
    float delta = 0.f;
    for (int s = 0; s < S; ++s) {
      const float pv = sp[l * SP + s];
      delta += fabsf(pv) * att_dot<HD>(sv + s * (HD + 4), dor) * at_mask_of(pv, dkey.inv_keep);   // through the weight dropout
    }
    float dqr[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) dqr[d] = 0.f;
    for (int s = 0; s < S; ++s) {
      const float pv = sp[l * SP + s];
      const float dp = att_dot<HD>(sv + s * (HD + 4), dor) * at_mask_of(pv, dkey.inv_keep);
      const float ds = fabsf(pv) * (dp - delta);
      sp[l * SP + s] = ds;
      att_axpy<HD>(dqr, ds, sk + s * (HD + 4));
    }
    float* dqrow = dq + ((size_t)l * N + n) * ldq + h * hd;
#pragma unroll
    for (int d = 0; d < HD; ++d)
      if (d < hd) dqrow[d] = dqr[d] * scale;
  }
  __syncthreads();
  if (lane < S) {
    const int s = lane;
    float dkr[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) dkr[d] = 0.f;
    for (int l = 0; l < L; ++l) att_axpy<HD>(dkr, sp[l * SP + s], sq + l * (HD + 4));
    float* dkrow = dk + ((size_t)s * N + n) * ldk + h * hd;
#pragma unroll
    for (int d = 0; d < HD; ++d)
      if (d < hd) dkrow[d] = dkr[d] * scale;
  }
}

// ---------------------------------------------------------------------------------------------
// MFMA form for 64 < L, S <= 128 with head_dim <= 16 (the action towers: Ta = 100, D / 2 = 16).  In the thread-per-row
// kernels above every lane walks all S keys one after the other (two passes of ~250 VALU cycles per key: 38 us forward,
// 66 us backward at Ta = 100 however many threads there are).  Here the four waves of a workgroup own 32 query rows (or
// 32 keys) each and the three products of forward (Q K^T, P V) and five of backward (P^T dO, dO V^T, dS K, dS^T Q) are
// v_mfma_f32_32x32x2_f32 tiles whose operands are read from LDS with conflict-free pitches (rows: 17 floats, the
// L x S tile: 129); the softmax runs with two lanes per row (64 keys each).  Same arithmetic definition, dropout mask
// indices and -inf semantics as the kernels above.
// ---------------------------------------------------------------------------------------------
#define AT_HD 16
#define AT_HP 17
#define AT_SP 129
__device__ __forceinline__ int at_i(int r, int lh) { return 8 * (r >> 2) + 4 * lh + (r & 3); }
// rows x hd floats (row r at src[(r N + n) ld + col0 ..]) -> dst[r][AT_HP], zero padded to 128 rows x 16 columns
__device__ __forceinline__ void at_stage(float* __restrict__ dst, const float* __restrict__ src, int rows, int N, int n,
                                         long ld, int col0, int hd, float mul, int tid) {
  if (hd == AT_HD && (col0 & 3) == 0 && (ld & 3) == 0 && (((uintptr_t)src) & 15) == 0) {
    // 4 x 16-byte loads per row, two rows' worth per thread in flight (512 quads for 128 rows)
    float4 v[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = tid + 256 * u, r = e >> 2, c = e & 3;
      v[u] = r < rows ? *reinterpret_cast<const float4*>(src + ((size_t)r * N + n) * ld + col0 + 4 * c)
                      : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = tid + 256 * u, r = e >> 2, c = e & 3;
      float* d = dst + r * AT_HP + 4 * c;
      d[0] = v[u].x * mul; d[1] = v[u].y * mul; d[2] = v[u].z * mul; d[3] = v[u].w * mul;
    }
    return;
  }
  for (int e = tid; e < 128 * AT_HD; e += 256) {
    const int r = e >> 4, d = e & 15;
    const bool ok = r < rows && d < hd;
    const float v = src[ok ? ((size_t)r * N + n) * ld + col0 + d : (size_t)n * ld + col0];
    dst[r * AT_HP + d] = ok ? v * mul : 0.f;
  }
}
// max / sum over the 32 lanes of each wave half
__device__ __forceinline__ float at_half_max(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float at_half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__global__ __launch_bounds__(256) void attn_mfma_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ v, const uint8_t* __restrict__ kpm,
                                                            float* __restrict__ out, float* __restrict__ probs, int L,
                                                            int S, int N, int H, int hd, long ldq, long ldk, long ldv,
                                                            int mask_is_valid, mmvae_dropout_t drop) {
  __shared__ float sq[128 * AT_HP], sk[128 * AT_HP], sv[128 * AT_HP];
  __shared__ float sp[128 * AT_SP];
  __shared__ float smask[128];
  const int n = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 31, lh = lane >> 5;
  const float scale = 1.0f / sqrtf((float)hd);
  at_stage(sq, q, L, N, n, ldq, h * hd, hd, scale, tid);
  at_stage(sk, k, S, N, n, ldk, h * hd, hd, 1.0f, tid);
  at_stage(sv, v, S, N, n, ldv, h * hd, hd, 1.0f, tid);
  if (tid < 128)
    smask[tid] = (tid >= S || (kpm && ((kpm[(size_t)n * S + tid] != 0) != (mask_is_valid != 0)))) ? 1.f : 0.f;
  __syncthreads();
  const int l0 = wave * 32;
  if (l0 >= L) return;                      // (no barrier below: every wave works on its own 32 rows of the tile)
  // ---- scores of rows l0 .. l0+31 against all keys: 4 accumulator tiles, register r <-> row l0 + at_i(r), lane li <->
  //      key 32 kb + li.  The softmax runs on the accumulators: per row an in-lane maximum / sum over the 4 tiles and a
  //      reduction over the 32 lanes of the half. ----
  float qa[AT_HD / 2];
#pragma unroll
  for (int kk = 0; kk < AT_HD / 2; ++kk) qa[kk] = sq[(l0 + li) * AT_HP + 2 * kk + lh];
  f32x16 acc[4];
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[kb][r] = 0.f;
    if (kb * 32 < S) {
#pragma unroll
      for (int kk = 0; kk < AT_HD / 2; ++kk)
        acc[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[kk], sk[(kb * 32 + li) * AT_HP + 2 * kk + lh], acc[kb], 0, 0, 0);
    }
    const bool masked = smask[kb * 32 + li] != 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[kb][r] = masked ? -INFINITY : acc[kb][r];
  }
  const DropKey dkey = drop_key(drop);
  const uint32_t dbase = (uint32_t)(((size_t)n * H + h) * L * S);
  float* P = probs + ((size_t)n * H + h) * L * S;
  float inv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float mx = at_half_max(fmaxf(fmaxf(acc[0][r], acc[1][r]), fmaxf(acc[2][r], acc[3][r])));
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const float pe = expf(acc[kb][r] - mx);     // exp(-inf - mx) = 0 for masked keys; NaN for a fully masked row
      acc[kb][r] = pe;
      sum += pe;
    }
    inv[r] = 1.0f / at_half_sum(sum);
  }
  // normalised probabilities straight from the registers (lanes over the keys); dropped, unnormalised weights -> tile
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    const int s_ = kb * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int l = l0 + at_i(r, lh);
      const float pe = acc[kb][r];
      const float m = drop_mul(dkey, dbase + (uint32_t)(l * S + s_));
      if (s_ < S && l < L) P[(size_t)l * S + s_] = at_signed(pe * inv[r], m);
      sp[l * AT_SP + s_] = s_ < S ? pe * m : 0.f;
    }
  }
  // ---- O = (P . mask) V ----
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
  const int ksteps = (S + 1) >> 1;
  const float* prow = sp + (l0 + li) * AT_SP;
#pragma unroll 4
  for (int kk = 0; kk < ksteps; ++kk) {
    const int s_ = 2 * kk + lh;
    const float b = li < AT_HD ? sv[s_ * AT_HP + li] : 0.f;
    o = __builtin_amdgcn_mfma_f32_32x32x2f32(prow[s_], b, o, 0, 0, 0);
  }
  if (li < hd) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int l = l0 + at_i(r, lh);
      if (l < L) out[((size_t)l * N + n) * ((size_t)H * hd) + h * hd + li] = o[r] * inv[r];
    }
  }
}

__global__ __launch_bounds__(256) void attn_mfma_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ v, const float* __restrict__ probs,
                                                            const float* __restrict__ dout, float* __restrict__ dq,
                                                            float* __restrict__ dk, float* __restrict__ dv, int L, int S,
                                                            int N, int H, int hd, long ldq, long ldk, long ldv,
                                                            mmvae_dropout_t drop) {
  __shared__ float sq[128 * AT_HP], sk[128 * AT_HP], sv[128 * AT_HP], sdo[128 * AT_HP];
  __shared__ float sp[128 * AT_SP];
  const int n = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 31, lh = lane >> 5;
  const long E = (long)H * hd;
  const float scale = 1.0f / sqrtf((float)hd);
  at_stage(sq, q, L, N, n, ldq, h * hd, hd, 1.0f, tid);
  at_stage(sk, k, S, N, n, ldk, h * hd, hd, 1.0f, tid);
  at_stage(sv, v, S, N, n, ldv, h * hd, hd, 1.0f, tid);
  at_stage(sdo, dout, L, N, n, E, h * hd, hd, 1.0f, tid);
  {   // P tile, zero padded to 128 x 128
    const float* P = probs + ((size_t)n * H + h) * L * S;
    if ((S & 3) == 0 && (((uintptr_t)P) & 15) == 0) {
      // 32 quads per row slot: 8 independent 16-byte loads per thread and round (two rounds)
#pragma unroll
      for (int e0 = 0; e0 < 128 * 32; e0 += 256 * 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = e0 + u * 256 + tid, l = e >> 5, c = e & 31;
          v[u] = (l < L && 4 * c < S) ? *reinterpret_cast<const float4*>(P + (size_t)l * S + 4 * c)
                                      : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = e0 + u * 256 + tid, l = e >> 5, c = e & 31;
          float* d = sp + l * AT_SP + 4 * c;
          d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
        }
      }
    } else {
      for (int e = tid; e < 128 * 128; e += 256) {
        const int l = e >> 7, s_ = e & 127;
        sp[l * AT_SP + s_] = (l < L && s_ < S) ? P[(size_t)l * S + s_] : 0.f;
      }
    }
  }
  __syncthreads();
  const DropKey dkey = drop_key(drop);
  const uint32_t dbase = (uint32_t)(((size_t)n * H + h) * L * S);
  const int b0 = wave * 32;                      // this wave's block of keys (dV, dK) and of query rows (dP, dQ)
  const int lsteps = (L + 1) >> 1, ssteps = (S + 1) >> 1;
  // ---- dV[s][d] = sum_l (P . mask)[l][s] dO[l][d] ----
  if (b0 < S) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 4
    for (int kk = 0; kk < lsteps; ++kk) {
      const int l = 2 * kk + lh;
      const float pv = sp[l * AT_SP + b0 + li];
      const float a = fabsf(pv) * at_mask_of(pv, dkey.inv_keep);
      const float b = li < AT_HD ? sdo[l * AT_HP + li] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    if (li < hd) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int s_ = b0 + at_i(r, lh);
        if (s_ < S) dv[((size_t)s_ * N + n) * ldv + h * hd + li] = acc[r];
      }
    }
  }
  __syncthreads();
  // ---- dS = P (dP . mask - delta), dP = dO V^T, delta[l] = sum_s P (dP . mask): rows b0 .. b0+31, in place over P ----
  if (b0 < L) {
    float da[AT_HD / 2];
#pragma unroll
    for (int kk = 0; kk < AT_HD / 2; ++kk) da[kk] = sdo[(b0 + li) * AT_HP + 2 * kk + lh];
    f32x16 dp[4];
    float part[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) part[r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) dp[kb][r] = 0.f;
      if (kb * 32 < S) {
#pragma unroll
        for (int kk = 0; kk < AT_HD / 2; ++kk)
          dp[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(da[kk], sv[(kb * 32 + li) * AT_HP + 2 * kk + lh], dp[kb], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int l = b0 + at_i(r, lh), s_ = kb * 32 + li;
          const float pv = sp[l * AT_SP + s_];
          dp[kb][r] *= at_mask_of(pv, dkey.inv_keep);
          part[r] += fabsf(pv) * dp[kb][r];
        }
      }
    }
    // row sums over the 32 key lanes of each half
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int o_ = 16; o_ > 0; o_ >>= 1) part[r] += __shfl_xor(part[r], o_, 64);
    }
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      if (kb * 32 < S) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int idx = (b0 + at_i(r, lh)) * AT_SP + kb * 32 + li;
          sp[idx] = fabsf(sp[idx]) * (dp[kb][r] - part[r]);
        }
      }
    }
  }
  __syncthreads();
  // ---- dQ[l][d] = scale sum_s dS[l][s] K[s][d] ----
  if (b0 < L) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 4
    for (int kk = 0; kk < ssteps; ++kk) {
      const int s_ = 2 * kk + lh;
      const float b = li < AT_HD ? sk[s_ * AT_HP + li] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sp[(b0 + li) * AT_SP + s_], b, acc, 0, 0, 0);
    }
    if (li < hd) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int l = b0 + at_i(r, lh);
        if (l < L) dq[((size_t)l * N + n) * ldq + h * hd + li] = acc[r] * scale;
      }
    }
  }
  // ---- dK[s][d] = scale sum_l dS[l][s] Q[l][d] ----
  if (b0 < S) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 4
    for (int kk = 0; kk < lsteps; ++kk) {
      const int l = 2 * kk + lh;
      const float b = li < AT_HD ? sq[l * AT_HP + li] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sp[l * AT_SP + b0 + li], b, acc, 0, 0, 0);
    }
    if (li < hd) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int s_ = b0 + at_i(r, lh);
        if (s_ < S) dk[((size_t)s_ * N + n) * ldk + h * hd + li] = acc[r] * scale;
      }
    }
  }
}
// ---- forward, transposed form (round 5): one WAVE per (sample, head, 32 query rows), scores as S^T = K Q^T --------------
// The tile's register r of lane (li, lh) is then score(key 32 kb + at_i(r, lh), query l0 + li): a query's whole row of
// keys lives in ONE lane pair, so the softmax is in-lane arithmetic plus one exchange with the other half (the kernel above
// reduces 16 rows over 32 lanes each: 160 shuffles), consecutive registers are consecutive keys (one mask hash per pair,
// 16-byte stores of the probabilities), and the normalised, dropped tile IS the A operand of P V (the accumulator-as-
// operand layout of ffn.hip: reduction index = key at_i(r, lh), so V is read in that order) -- no L x S tile in LDS at all.
// 16 KB of LDS per wave instead of 92 KB per workgroup: the 4 x N x H waves of a launch are all resident at once.
__global__ __launch_bounds__(64) void attn_t_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                        const float* __restrict__ v, const uint8_t* __restrict__ kpm,
                                                        float* __restrict__ out, float* __restrict__ probs, int L, int S,
                                                        int N, int H, int hd, long ldq, long ldk, long ldv,
                                                        int mask_is_valid, mmvae_dropout_t drop) {
  __shared__ __attribute__((aligned(16))) float sk[128 * AT_HP + 16], sv[128 * AT_HP + 16], sq[32 * AT_HP];
  __shared__ float smask[128];
  const int n = blockIdx.x, h = blockIdx.y, l0 = blockIdx.z * 32, lane = threadIdx.x, li = lane & 31, lh = lane >> 5;
  const float scale = 1.0f / sqrtf((float)hd);
  {   // K, V: 128 rows x 4 quads each (rows >= S and columns >= hd: zeros; hd a multiple of 4), Q: this wave's 32 rows;
      // all loads in flight before the LDS writes
    float4 kq[8], vq[8], qq[2];
    const int nq = hd >> 2;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = lane + 64 * u, r = e >> 2, c = e & 3;
      const bool ok = r < S && c < nq;
      kq[u] = ok ? *reinterpret_cast<const float4*>(k + ((size_t)r * N + n) * ldk + h * hd + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
      vq[u] = ok ? *reinterpret_cast<const float4*>(v + ((size_t)r * N + n) * ldv + h * hd + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = lane + 64 * u, r = e >> 2, c = e & 3;
      qq[u] = (l0 + r < L && c < nq) ? *reinterpret_cast<const float4*>(q + ((size_t)(l0 + r) * N + n) * ldq + h * hd + 4 * c)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = lane + 64 * u, r = e >> 2, c = e & 3;
      float* dk_ = sk + r * AT_HP + 4 * c;
      dk_[0] = kq[u].x; dk_[1] = kq[u].y; dk_[2] = kq[u].z; dk_[3] = kq[u].w;
      float* dv_ = sv + r * AT_HP + 4 * c;
      dv_[0] = vq[u].x; dv_[1] = vq[u].y; dv_[2] = vq[u].z; dv_[3] = vq[u].w;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = lane + 64 * u, r = e >> 2, c = e & 3;
      float* dq_ = sq + r * AT_HP + 4 * c;
      dq_[0] = qq[u].x * scale; dq_[1] = qq[u].y * scale; dq_[2] = qq[u].z * scale; dq_[3] = qq[u].w * scale;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int s_ = lane + 64 * u;
      smask[s_] = (s_ >= S || (kpm && ((kpm[(size_t)n * S + s_] != 0) != (mask_is_valid != 0)))) ? 1.f : 0.f;
    }
  }
  __syncthreads();
  float qb[AT_HD / 2];
#pragma unroll
  for (int kk = 0; kk < AT_HD / 2; ++kk) qb[kk] = sq[li * AT_HP + 2 * kk + lh];
  f32x16 acc[4];
  float mx = -INFINITY;
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[kb][r] = 0.f;
    if (kb * 32 < S) {
#pragma unroll
      for (int kk = 0; kk < AT_HD / 2; ++kk)
        acc[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(sk[(kb * 32 + li) * AT_HP + 2 * kk + lh], qb[kk], acc[kb], 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc[kb][r] = smask[kb * 32 + at_i(r, lh)] != 0.f ? -INFINITY : acc[kb][r];
      mx = fmaxf(mx, acc[kb][r]);
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int kb = 0; kb < 4; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc[kb][r] = __expf(acc[kb][r] - mx);          // exp(-inf - mx) = 0 for masked keys; NaN for a fully masked row
      sum += acc[kb][r];
    }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
  const DropKey dkey = drop_key(drop);
  const int l = l0 + li;
  const uint32_t dbase = (uint32_t)(((size_t)n * H + h) * L * S) + (uint32_t)(l * S);
  float* P = probs + ((size_t)n * H + h) * L * S + (size_t)l * S;
  const bool vec = (S & 3) == 0 && (((uintptr_t)probs) & 15) == 0;
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    if (kb * 32 < S) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int key0 = kb * 32 + 8 * g + 4 * lh;              // registers 4 g .. 4 g + 3: keys key0 .. key0 + 3
        float p4[4], m4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) p4[j] = acc[kb][4 * g + j] * inv;
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
          const uint32_t i0 = dbase + (uint32_t)(key0 + j);
          if ((i0 & 1u) == 0u) {
            const uint32_t hh = drop_pair_hash(dkey, i0 >> 1);
            m4[j] = drop_pair_lo(dkey, hh);
            m4[j + 1] = drop_pair_hi(dkey, hh);
          } else {
            m4[j] = drop_mul(dkey, i0);
            m4[j + 1] = drop_mul(dkey, i0 + 1);
          }
        }
        if (l < L) {      // saved for backward with the dropout decision in the sign bit (at_signed)
          if (vec && key0 + 3 < S) {
            *reinterpret_cast<float4*>(P + key0) = make_float4(at_signed(p4[0], m4[0]), at_signed(p4[1], m4[1]),
                                                                at_signed(p4[2], m4[2]), at_signed(p4[3], m4[3]));
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (key0 + j < S) P[key0 + j] = at_signed(p4[j], m4[j]);
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) p4[j] *= m4[j];
        // O += (P . mask) V over these four keys: the tile's registers are the A operand, reduction index = key
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float b = li < AT_HD ? sv[(key0 + j) * AT_HP + li] : 0.f;
          o = __builtin_amdgcn_mfma_f32_32x32x2f32(p4[j], b, o, 0, 0, 0);
        }
      }
    }
  }
  if (li < hd) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int lo = l0 + at_i(r, lh);
      if (lo < L) out[((size_t)lo * N + n) * ((size_t)H * hd) + h * hd + li] = o[r];
    }
  }
}
// ---- backward, register form (round 6): the L x S tile never enters LDS --------------------------------------------------
// attn_mfma_bwd_kernel stages Q, K, V, dO AND the 128 x 128 probability tile (92 KB of LDS: one workgroup per CU), rewrites
// the tile in place between four barrier-separated phases and reads one LDS word per lane and MFMA out of it: 29.4 us at
// 128 x 2 x 100 x 100 for 0.33 GFLOP.  Here the probabilities are read from global memory straight into the accumulator
// layouts the products want and dP is computed twice, once per orientation:
//   phase A, wave = 32 QUERIES (lane li = query): dP^T = V dO^T per 32-key tile (register r = key at_i(r, lh)), P^T read with
//     four 16-byte loads per tile and lane, delta[query] = sum_key P dP.mask is in-lane adds + ONE exchange between the wave's
//     halves (the tile form reduces 16 rows over 32 lanes: 80 shuffles), dS^T stays in registers and IS the B operand of
//     dQ^T = K^T dS^T (accumulator-as-operand: reduction index = key at_i(s, lh), K read from LDS in that order);
//   phase C, wave = 32 KEYS (lane li = key): per 32-query tile dP = dO V^T (register r = query), P read as 16 coalesced words,
//     delta from LDS (128 floats), and the dropped probabilities / dS are the B operands of dV^T = dO^T P.mask and
//     dK^T = Q^T dS with the reduction index = query at_i(s, lh); accumulated over the query tiles in registers, no cross-wave sum.
// 35 KB of LDS (Q, K, V, dO rows + delta): four workgroups per CU; 256 MFMAs per wave, two barriers.
__global__ __launch_bounds__(256) void attn_t_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                         const float* __restrict__ v, const float* __restrict__ probs,
                                                         const float* __restrict__ dout, float* __restrict__ dq,
                                                         float* __restrict__ dk, float* __restrict__ dv, int L, int S, int N,
                                                         int H, int hd, long ldq, long ldk, long ldv, mmvae_dropout_t drop) {
  __shared__ float sq[128 * AT_HP], sk[128 * AT_HP], sv[128 * AT_HP], sdo[128 * AT_HP];
  __shared__ float sdelta[128];
  const int n = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 31, lh = lane >> 5;
  const long E = (long)H * hd;
  const float scale = 1.0f / sqrtf((float)hd);
  at_stage(sq, q, L, N, n, ldq, h * hd, hd, 1.0f, tid);
  at_stage(sk, k, S, N, n, ldk, h * hd, hd, 1.0f, tid);
  at_stage(sv, v, S, N, n, ldv, h * hd, hd, 1.0f, tid);
  at_stage(sdo, dout, L, N, n, E, h * hd, hd, 1.0f, tid);
  const float* __restrict__ P = probs + ((size_t)n * H + h) * L * S;
  const bool vec = (S & 3) == 0 && (((uintptr_t)P) & 15) == 0;
  const float inv_keep = drop_key(drop).inv_keep;
  const int b0 = wave * 32;
  if (tid < 128) sdelta[tid] = 0.f;      // (rows >= L are read with a zero probability in front of them: keep them finite)
  __syncthreads();
  // ================= phase A: this wave's 32 queries =================
  if (b0 < L) {
    const int l = b0 + li;
    const bool lok = l < L;
    float da[AT_HD / 2];
#pragma unroll
    for (int kk = 0; kk < AT_HD / 2; ++kk) da[kk] = sdo[l * AT_HP + 2 * kk + lh];
    f32x16 ds[4];                 // dP^T, then dS^T: register r <-> key 32 kb + at_i(r, lh), lane <-> query
    float pa[4][16];              // |P^T|
    float part = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { ds[kb][r] = 0.f; pa[kb][r] = 0.f; }
      if (kb * 32 < S) {
#pragma unroll
        for (int kk = 0; kk < AT_HD / 2; ++kk)
          ds[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[(kb * 32 + li) * AT_HP + 2 * kk + lh], da[kk], ds[kb], 0, 0, 0);
        const float* prow = P + (size_t)(lok ? l : 0) * S + kb * 32 + 4 * lh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int key0 = kb * 32 + 8 * g + 4 * lh;
          float p4[4] = {0.f, 0.f, 0.f, 0.f};
          if (vec) {
            if (lok && key0 < S) {
              const float4 t = *reinterpret_cast<const float4*>(prow + 8 * g);
              p4[0] = t.x; p4[1] = t.y; p4[2] = t.z; p4[3] = t.w;
            }
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (lok && key0 + j < S) p4[j] = prow[8 * g + j];
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int r = 4 * g + j;
            const float dpm = ds[kb][r] * at_mask_of(p4[j], inv_keep);      // dP . mask
            pa[kb][r] = fabsf(p4[j]);
            ds[kb][r] = dpm;
            part += pa[kb][r] * dpm;
          }
        }
      }
    }
    const float delta = part + __shfl_xor(part, 32, 64);
    if (lh == 0) sdelta[l] = delta;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) ds[kb][r] = pa[kb][r] * (ds[kb][r] - delta);
    // ---- dQ^T[d][query] = sum_key K[key][d] dS^T[key][query]: lane (d = li, lh) feeds K[32 kb + at_i(s, lh)][li] ----
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      if (kb * 32 < S) {
#pragma unroll
        for (int s_ = 0; s_ < 16; ++s_) {
          const float a = li < AT_HD ? sk[(kb * 32 + at_i(s_, lh)) * AT_HP + li] : 0.f;
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, ds[kb][s_], acc, 0, 0, 0);
        }
      }
    }
    if (lok) {      // register r < 8 of lane (li, lh) = dQ[l][d = at_i(r, lh)]: two float4 per lane
      float* o = dq + ((size_t)l * N + n) * ldq + h * hd;
      if (4 * lh < hd) *reinterpret_cast<float4*>(o + 4 * lh) = make_float4(acc[0] * scale, acc[1] * scale, acc[2] * scale, acc[3] * scale);
      if (8 + 4 * lh < hd)
        *reinterpret_cast<float4*>(o + 8 + 4 * lh) = make_float4(acc[4] * scale, acc[5] * scale, acc[6] * scale, acc[7] * scale);
    }
  }
  __syncthreads();
  // ================= phase C: this wave's 32 keys =================
  if (b0 < S) {
    const int s_key = b0 + li;
    const bool sok = s_key < S;
    float vb[AT_HD / 2];
#pragma unroll
    for (int kk = 0; kk < AT_HD / 2; ++kk) vb[kk] = sv[s_key * AT_HP + 2 * kk + lh];
    f32x16 accv, acck;
#pragma unroll
    for (int r = 0; r < 16; ++r) accv[r] = acck[r] = 0.f;
#pragma unroll 1
    for (int qt = 0; qt < 4; ++qt) {
      if (qt * 32 >= L) break;
      f32x16 dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) dp[r] = 0.f;
#pragma unroll
      for (int kk = 0; kk < AT_HD / 2; ++kk)
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(sdo[(qt * 32 + li) * AT_HP + 2 * kk + lh], vb[kk], dp, 0, 0, 0);
      float pm[16], dsv[16];      // register r <-> query 32 qt + at_i(r, lh), lane <-> key
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int l = qt * 32 + at_i(r, lh);
        const float pv = (sok && l < L) ? P[(size_t)l * S + s_key] : 0.f;
        const float m = at_mask_of(pv, inv_keep), pabs = fabsf(pv);
        pm[r] = pabs * m;
        dsv[r] = pabs * (dp[r] * m - sdelta[l]);
      }
#pragma unroll
      for (int s_ = 0; s_ < 16; ++s_) {
        const int l = qt * 32 + at_i(s_, lh);
        const float ao = li < AT_HD ? sdo[l * AT_HP + li] : 0.f;
        const float aq = li < AT_HD ? sq[l * AT_HP + li] : 0.f;
        accv = __builtin_amdgcn_mfma_f32_32x32x2f32(ao, pm[s_], accv, 0, 0, 0);
        acck = __builtin_amdgcn_mfma_f32_32x32x2f32(aq, dsv[s_], acck, 0, 0, 0);
      }
    }
    if (sok) {
      float* ov = dv + ((size_t)s_key * N + n) * ldv + h * hd;
      float* ok_ = dk + ((size_t)s_key * N + n) * ldk + h * hd;
      if (4 * lh < hd) {
        *reinterpret_cast<float4*>(ov + 4 * lh) = make_float4(accv[0], accv[1], accv[2], accv[3]);
        *reinterpret_cast<float4*>(ok_ + 4 * lh) = make_float4(acck[0] * scale, acck[1] * scale, acck[2] * scale, acck[3] * scale);
      }
      if (8 + 4 * lh < hd) {
        *reinterpret_cast<float4*>(ov + 8 + 4 * lh) = make_float4(accv[4], accv[5], accv[6], accv[7]);
        *reinterpret_cast<float4*>(ok_ + 8 + 4 * lh) = make_float4(acck[4] * scale, acck[5] * scale, acck[6] * scale, acck[7] * scale);
      }
    }
  }
}
static inline bool attn_t_ok(const float* q, const float* k, const float* v, int hd, long ldq, long ldk, long ldv) {
  return hd <= AT_HD && (hd & 3) == 0 && ((ldq | ldk | ldv) & 3) == 0 &&
         ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v)) & 15) == 0;
}
// the MFMA kernels take over from the thread-per-row ones past 32 rows (round 5: it was 64 -- the text decoder of a PoE
// subset WITHOUT the text modality decodes all 45 positions of its modality: 20 / 33 us per call at 32 x 2 x 45 x 45)
// runtime switch (the tests run both; MMVAE_ATTN_T_BWD=1 in the environment starts with the register form).  Measured, same
// box (profiles/r06_attn_t_bwd.txt): alone 24.4 -> 22.8 us at 128 x 2 x 100 x 100, in the cfg5 step 2.932 -> 2.942 ms -- level,
// so the tile form stays the default: both spend most of their time staging four strided operand matrices and waiting on
// LDS operands, not in their 256 MFMAs (6.8 us)
static int attn_t_bwd_flag = -1;
static bool attn_t_bwd_enabled() {
  if (attn_t_bwd_flag < 0) {
    const char* e = getenv("MMVAE_ATTN_T_BWD");
    attn_t_bwd_flag = (e && e[0] == '1') ? 1 : 0;
  }
  return attn_t_bwd_flag != 0;
}
extern "C" int mmvae_attn_t_bwd_set(int on) {
  const int was = attn_t_bwd_enabled() ? 1 : 0;
  attn_t_bwd_flag = on ? 1 : 0;
  return was;
}
static inline bool attn_use_mfma(int L, int S, int hd) {
  return hd <= AT_HD && (L > 32 || S > 32) && L <= 128 && S <= 128;
}

extern "C" int mmvae_attn_fwd(const float* q, const float* k, const float* v, const uint8_t* kpm, float* out,
                              float* probs, int L, int S, int N, int H, int hd, long ldq, long ldk, long ldv,
                              int mask_is_valid, const mmvae_dropout_t* drop, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(q && k && v && out && probs && L > 0 && S > 0 && N > 0 && H > 0 && hd > 0);
  if (L > ATT_MAX || S > ATT_MAX || hd > ATT_HD_MAX) return MMVAE_ERR_UNSUPPORTED;
  if (attn_use_mfma(L, S, hd) && attn_t_ok(q, k, v, hd, ldq, ldk, ldv)) {
    hipLaunchKernelGGL(attn_t_fwd_kernel, dim3(N, H, (L + 31) / 32), dim3(64), 0, (hipStream_t)stream, q, k, v, kpm, out, probs,
                       L, S, N, H, hd, ldq, ldk, ldv, mask_is_valid, drop_arg(drop));
    return mmvae_launch_status();
  }
  if (attn_use_mfma(L, S, hd)) {
    hipLaunchKernelGGL(attn_mfma_fwd_kernel, dim3(N, H), dim3(256), 0, (hipStream_t)stream, q, k, v, kpm, out, probs, L, S,
                       N, H, hd, ldq, ldk, ldv, mask_is_valid, drop_arg(drop));
    return mmvae_launch_status();
  }
#define ATT_FWD(AM, HD)                                                                                              \
  hipLaunchKernelGGL((attn_fwd_kernel<AM, HD>), dim3(N, H), dim3(AM), 0, (hipStream_t)stream, q, k, v, kpm, out, probs, \
                     L, S, N, H, hd, ldq, ldk, ldv, mask_is_valid, drop_arg(drop))
  if (L <= 64 && S <= 64) { if (hd <= 16) ATT_FWD(64, 16); else ATT_FWD(64, 32); }
  else { if (hd <= 16) ATT_FWD(128, 16); else ATT_FWD(128, 32); }
#undef ATT_FWD
  return mmvae_launch_status();
}
extern "C" int mmvae_attn_bwd(const float* q, const float* k, const float* v, const float* probs, const float* dout,
                              float* dq, float* dk, float* dv, int L, int S, int N, int H, int hd, long ldq, long ldk,
                              long ldv, const mmvae_dropout_t* drop, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(q && k && v && probs && dout && dq && dk && dv && L > 0 && S > 0 && N > 0 && H > 0 && hd > 0);
  if (L > ATT_MAX || S > ATT_MAX || hd > ATT_HD_MAX) return MMVAE_ERR_UNSUPPORTED;
  if (attn_use_mfma(L, S, hd)) {
    // the register form needs 16-byte aligned head slices (its dq / dk / dv stores are float4), as the transposed forward does
    const bool t_ok = attn_t_bwd_enabled() && attn_t_ok(q, k, v, hd, ldq, ldk, ldv) && attn_t_ok(dq, dk, dv, hd, ldq, ldk, ldv) &&
                      (((uintptr_t)dout) & 15) == 0;
    if (t_ok)
      hipLaunchKernelGGL(attn_t_bwd_kernel, dim3(N, H), dim3(256), 0, (hipStream_t)stream, q, k, v, probs, dout, dq, dk, dv, L,
                         S, N, H, hd, ldq, ldk, ldv, drop_arg(drop));
    else
      hipLaunchKernelGGL(attn_mfma_bwd_kernel, dim3(N, H), dim3(256), 0, (hipStream_t)stream, q, k, v, probs, dout, dq, dk,
                         dv, L, S, N, H, hd, ldq, ldk, ldv, drop_arg(drop));
    return mmvae_launch_status();
  }
#define ATT_BWD(AM, HD)                                                                                              \
  hipLaunchKernelGGL((attn_bwd_kernel<AM, HD>), dim3(N, H), dim3(AM), 0, (hipStream_t)stream, q, k, v, probs, dout, dq, \
                     dk, dv, L, S, N, H, hd, ldq, ldk, ldv, drop_arg(drop))
  if (L <= 64 && S <= 64) { if (hd <= 16) ATT_BWD(64, 16); else ATT_BWD(64, 32); }
  else { if (hd <= 16) ATT_BWD(128, 16); else ATT_BWD(128, 32); }
#undef ATT_BWD
  return mmvae_launch_status();
}

// ---------------------------------------------------------------------------------------------
// y = LayerNorm(x + r) (torch.nn.LayerNorm, eps 1e-5, biased variance); one wavefront per row.
// r: NULL, same shape, or broadcast over time (r_rows = N rows, row index = row % N).
// ---------------------------------------------------------------------------------------------
#define LN_SLOTS 4  // d <= 256
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ r,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ y, float* __restrict__ xhat,
                                                     float* __restrict__ rstd, int rows, int d, int r_rows,
                                                     mmvae_dropout_t drop) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (size_t)row * d;
  const float* rr = r ? r + (size_t)(r_rows > 0 ? row % r_rows : row) * d : nullptr;
  const DropKey dkey = drop_key(drop);   // dropout on x (the sub-layer output), not on the residual
  float vals[LN_SLOTS];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_SLOTS; ++i) {
    const int c = lane + 64 * i;
    float v = 0.f;
    if (c < d) v = xr[c] * drop_mul(dkey, (uint32_t)((size_t)row * d + c)) + (rr ? rr[c] : 0.f);
    vals[i] = v;
    s += v;
  }
  const float mean = wave_sum(s) / (float)d;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < LN_SLOTS; ++i) {
    const int c = lane + 64 * i;
    const float dv = c < d ? vals[i] - mean : 0.f;
    ss += dv * dv;
  }
  const float rs = rsqrtf(wave_sum(ss) / (float)d + 1e-5f);
#pragma unroll
  for (int i = 0; i < LN_SLOTS; ++i) {
    const int c = lane + 64 * i;
    if (c < d) {
      const float xh = (vals[i] - mean) * rs;
      xhat[(size_t)row * d + c] = xh;
      y[(size_t)row * d + c] = xh * gamma[c] + beta[c];
    }
  }
  if (lane == 0) rstd[row] = rs;
}

// dsum = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma;  per-block partials of dgamma, dbeta
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ xhat,
                                                     const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                     float* __restrict__ dsum, float* __restrict__ dxd,
                                                     float* __restrict__ ws, int rows, int d, int rows_per_block,
                                                     mmvae_dropout_t drop) {
  __shared__ float sg[4][2 * 64 * LN_SLOTS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const DropKey dkey = drop_key(drop);
  float dg[LN_SLOTS], db[LN_SLOTS];
#pragma unroll
  for (int i = 0; i < LN_SLOTS; ++i) dg[i] = db[i] = 0.f;
  const int row_beg = blockIdx.x * rows_per_block;
  const int row_end = min(rows, row_beg + rows_per_block);
  for (int row = row_beg + wave; row < row_end; row += 4) {
    float g[LN_SLOTS], xh[LN_SLOTS];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_SLOTS; ++i) {
      const int c = lane + 64 * i;
      g[i] = xh[i] = 0.f;
      if (c < d) {
        const float dyv = dy[(size_t)row * d + c];
        xh[i] = xhat[(size_t)row * d + c];
        g[i] = dyv * gamma[c];
        dg[i] += dyv * xh[i];
        db[i] += dyv;
      }
      s1 += g[i];
      s2 += g[i] * xh[i];
    }
    s1 = wave_sum(s1) / (float)d;
    s2 = wave_sum(s2) / (float)d;
    const float rs = rstd[row];
#pragma unroll
    for (int i = 0; i < LN_SLOTS; ++i) {
      const int c = lane + 64 * i;
      if (c < d) {
        const float ds = rs * (g[i] - s1 - xh[i] * s2);
        dsum[(size_t)row * d + c] = ds;
        if (dxd) dxd[(size_t)row * d + c] = ds * drop_mul(dkey, (uint32_t)((size_t)row * d + c));
      }
    }
  }
#pragma unroll
  for (int i = 0; i < LN_SLOTS; ++i) {
    sg[wave][lane + 64 * i] = dg[i];
    sg[wave][64 * LN_SLOTS + lane + 64 * i] = db[i];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < d; c += 256) {
    float a = 0.f, b = 0.f;
    for (int w = 0; w < 4; ++w) {
      a += sg[w][c];
      b += sg[w][64 * LN_SLOTS + c];
    }
    ws[(size_t)blockIdx.x * 2 * d + c] = a;
    ws[(size_t)blockIdx.x * 2 * d + d + c] = b;
  }
}

// ---- d == 32 (the action towers' d_model; round 5): a row is EIGHT threads with one float4 each, 32 rows per workgroup
// and pass -- coalesced 16-byte accesses and every lane busy, where the generic kernels give a 32-float row a whole wave
// (half its lanes idle, one dependent load round per row).  Same arithmetic, same dropout mask (element row * 32 + c; one
// hash per even-aligned pair), same partial-row layout of the parameter gradients.
__device__ __forceinline__ float ln32_sum8(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  return v;
}
__device__ __forceinline__ float4 ln32_mask4(const DropKey& dk, uint32_t idx) {      // idx % 4 == 0
  const uint32_t h0 = drop_pair_hash(dk, idx >> 1), h1 = drop_pair_hash(dk, (idx >> 1) + 1);
  return make_float4(drop_pair_lo(dk, h0), drop_pair_hi(dk, h0), drop_pair_lo(dk, h1), drop_pair_hi(dk, h1));
}
__global__ __launch_bounds__(256) void ln32_fwd_kernel(const float* __restrict__ x, const float* __restrict__ r,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ y, float* __restrict__ xhat,
                                                       float* __restrict__ rstd, int rows, int r_rows, mmvae_dropout_t drop) {
  const int q = threadIdx.x & 7, row = blockIdx.x * 32 + (threadIdx.x >> 3);
  if (row >= rows) return;
  const DropKey dkey = drop_key(drop);
  const size_t o = (size_t)row * 32 + 4 * q;
  float4 v = *reinterpret_cast<const float4*>(x + o);
  if (dkey.on) {
    const float4 m = ln32_mask4(dkey, (uint32_t)o);
    v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
  }
  if (r) {
    const float4 rv = *reinterpret_cast<const float4*>(r + (size_t)(r_rows > 0 ? row % r_rows : row) * 32 + 4 * q);
    v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
  }
  const float mean = ln32_sum8((v.x + v.y) + (v.z + v.w)) * (1.0f / 32.0f);
  const float4 dv = make_float4(v.x - mean, v.y - mean, v.z - mean, v.w - mean);
  const float rs = rsqrtf(ln32_sum8((dv.x * dv.x + dv.y * dv.y) + (dv.z * dv.z + dv.w * dv.w)) * (1.0f / 32.0f) + 1e-5f);
  const float4 g = *reinterpret_cast<const float4*>(gamma + 4 * q), b = *reinterpret_cast<const float4*>(beta + 4 * q);
  const float4 xh = make_float4(dv.x * rs, dv.y * rs, dv.z * rs, dv.w * rs);
  *reinterpret_cast<float4*>(xhat + o) = xh;
  *reinterpret_cast<float4*>(y + o) = make_float4(xh.x * g.x + b.x, xh.y * g.y + b.y, xh.z * g.z + b.z, xh.w * g.w + b.w);
  if (q == 0) rstd[row] = rs;
}
__global__ __launch_bounds__(256) void ln32_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ xhat,
                                                       const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                       float* __restrict__ dsum, float* __restrict__ dxd,
                                                       float* __restrict__ ws, int rows, int rows_per_block,
                                                       mmvae_dropout_t drop) {
  __shared__ float sg[32][65];
  const int q = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const DropKey dkey = drop_key(drop);
  const float4 gm = *reinterpret_cast<const float4*>(gamma + 4 * q);
  float4 dg = make_float4(0.f, 0.f, 0.f, 0.f), db = dg;
  const int row_beg = blockIdx.x * rows_per_block, row_end = min(rows, row_beg + rows_per_block);
  for (int row = row_beg + rl; row < row_end; row += 32) {
    const size_t o = (size_t)row * 32 + 4 * q;
    const float4 d4 = *reinterpret_cast<const float4*>(dy + o), xh = *reinterpret_cast<const float4*>(xhat + o);
    const float rs = rstd[row];
    const float4 g = make_float4(d4.x * gm.x, d4.y * gm.y, d4.z * gm.z, d4.w * gm.w);
    dg.x += d4.x * xh.x; dg.y += d4.y * xh.y; dg.z += d4.z * xh.z; dg.w += d4.w * xh.w;
    db.x += d4.x; db.y += d4.y; db.z += d4.z; db.w += d4.w;
    const float s1 = ln32_sum8((g.x + g.y) + (g.z + g.w)) * (1.0f / 32.0f);
    const float s2 = ln32_sum8((g.x * xh.x + g.y * xh.y) + (g.z * xh.z + g.w * xh.w)) * (1.0f / 32.0f);
    const float4 ds = make_float4(rs * (g.x - s1 - xh.x * s2), rs * (g.y - s1 - xh.y * s2), rs * (g.z - s1 - xh.z * s2),
                                  rs * (g.w - s1 - xh.w * s2));
    *reinterpret_cast<float4*>(dsum + o) = ds;
    if (dxd) {
      float4 m = make_float4(1.f, 1.f, 1.f, 1.f);
      if (dkey.on) m = ln32_mask4(dkey, (uint32_t)o);
      *reinterpret_cast<float4*>(dxd + o) = make_float4(ds.x * m.x, ds.y * m.y, ds.z * m.z, ds.w * m.w);
    }
  }
  float* srow = sg[rl];
  srow[4 * q] = dg.x; srow[4 * q + 1] = dg.y; srow[4 * q + 2] = dg.z; srow[4 * q + 3] = dg.w;
  srow[32 + 4 * q] = db.x; srow[32 + 4 * q + 1] = db.y; srow[32 + 4 * q + 2] = db.z; srow[32 + 4 * q + 3] = db.w;
  __syncthreads();
  if (threadIdx.x < 64) {
    float a = 0.f;
#pragma unroll 8
    for (int i = 0; i < 32; ++i) a += sg[i][threadIdx.x];
    ws[(size_t)blockIdx.x * 64 + threadIdx.x] = a;          // [dgamma (32) | dbeta (32)]
  }
}

static inline bool ln32_ok(const void* a, const void* b, const void* c, const void* d_) {
  return ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c) | ((uintptr_t)d_)) & 15) == 0;
}
static inline int ln_blocks(int rows, int d, int* rpb) {
  if (d == 32) {      // ln32_bwd_kernel: 32 rows per pass, whole passes per workgroup
    int nb = (rows + 31) / 32;
    if (nb > 1024) nb = 1024;
    *rpb = ((rows + nb - 1) / nb + 31) / 32 * 32;
    return (rows + *rpb - 1) / *rpb;
  }
  // a wave walks its rows one after the other (one dependent load round per row): at most ~4 rows per wave, i.e. 16 per
  // workgroup, up to 1024 workgroups (the action towers' 12 800-row LayerNorms: 26.5 -> 6 us; 128 workgroups of 100 rows
  // before).  The per-workgroup [dgamma | dbeta] partials grow with it: 1024 x 2d floats, folded with the others.
  int nb = (rows + 15) / 16;
  if (nb > 1024) nb = 1024;
  *rpb = (rows + nb - 1) / nb;
  return (rows + *rpb - 1) / *rpb;
}
extern "C" int mmvae_layernorm_bwd_rows(int rows, int d) {
  int rpb;
  return ln_blocks(rows, d, &rpb);
}
extern "C" size_t mmvae_layernorm_ws_floats(int rows, int d) {
  int rpb;
  return (size_t)ln_blocks(rows, d, &rpb) * 2 * d;
}
extern "C" int mmvae_layernorm_residual_fwd(const float* x, const float* r, const float* gamma, const float* beta,
                                            float* y, float* xhat, float* rstd, int rows, int d, int r_rows,
                                            const mmvae_dropout_t* drop, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(x && gamma && beta && y && xhat && rstd && rows > 0 && d > 0);
  if (d > 64 * LN_SLOTS) return MMVAE_ERR_UNSUPPORTED;
  if (d == 32 && ln32_ok(x, y, xhat, r) && ln32_ok(gamma, beta, nullptr, nullptr)) {
    hipLaunchKernelGGL(ln32_fwd_kernel, dim3((rows + 31) / 32), dim3(256), 0, (hipStream_t)stream, x, r, gamma, beta, y, xhat,
                       rstd, rows, r_rows, drop_arg(drop));
    return mmvae_launch_status();
  }
  hipLaunchKernelGGL(ln_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, r, gamma, beta, y,
                     xhat, rstd, rows, d, r_rows, drop_arg(drop));
  return mmvae_launch_status();
}
// dgamma and dbeta must be adjacent when both given separately is not required: two reductions are issued.
extern "C" int mmvae_layernorm_residual_bwd(const float* dy, const float* xhat, const float* rstd, const float* gamma,
                                            float* dsum, float* dx_drop, float* dgamma, float* dbeta, float* ws,
                                            int rows, int d, int accumulate, const mmvae_dropout_t* drop,
                                            mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(dy && xhat && rstd && gamma && dsum && ws && rows > 0 && d > 0);
  MMVAE_CHECK_ARG(accumulate == MMVAE_ACC_DEFER || (dgamma && dbeta));
  if (d > 64 * LN_SLOTS) return MMVAE_ERR_UNSUPPORTED;
  int rpb;
  const int nb = ln_blocks(rows, d, &rpb);
  // (a misaligned view falls back to the generic kernel, as the forward does: same 2 d floats per block in ws)
  if (d == 32 && ln32_ok(dy, xhat, dsum, dx_drop) && ln32_ok(gamma, nullptr, nullptr, nullptr)) {
    hipLaunchKernelGGL(ln32_bwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, dy, xhat, rstd, gamma, dsum, dx_drop, ws,
                       rows, rpb, drop_arg(drop));
  } else {
    hipLaunchKernelGGL(ln_bwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, dy, xhat, rstd, gamma, dsum, dx_drop,
                       ws, rows, d, rpb, drop_arg(drop));
  }
  int rc = mmvae_launch_status();
  if (rc || accumulate == MMVAE_ACC_DEFER) return rc;
  if (dbeta == dgamma + d) return mmvae_reduce_rows(ws, dgamma, nb, 2L * d, 2L * d, accumulate, stream);
  rc = mmvae_reduce_rows(ws, dgamma, nb, d, 2L * d, accumulate, stream);
  if (rc) return rc;
  return mmvae_reduce_rows(ws + d, dbeta, nb, d, 2L * d, accumulate, stream);
}

// ---------------------------------------------------------------------------------------------
// time reductions over x (L,N,d): y[n,c] = scale * sum_l x[l,n,c];  and the broadcast backward
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void time_sum_serial_kernel(const float* __restrict__ x, float* __restrict__ y, int L,
                                                              long nd, float scale) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= nd) return;
  float a = 0.f;
  for (int l = 0; l < L; ++l) a += x[(size_t)l * nd + i];
  y[i] = a * scale;
}
// (round 6: from TIME_SUM_WIDE_L steps on -- a thread per column walking all L rows left 16 workgroups with 100
// dependent-latency loads each at the action encoder's pooling: 28 us for 1.6 MB -- 64 columns x 4 row groups per workgroup,
// two accumulators per thread, the four partial sums added in a fixed order; short sequences keep the serial sum)
#define TIME_SUM_WIDE_L 32
__global__ __launch_bounds__(256) void time_sum_kernel(const float* __restrict__ x, float* __restrict__ y, int L,
                                                       long nd, float scale) {
  __shared__ float part[4][64];
  const int c = threadIdx.x & 63, r = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + c;
  float a0 = 0.f, a1 = 0.f;
  if (i < nd) {
    int l = r;
    for (; l + 4 < L; l += 8) {
      a0 += x[(size_t)l * nd + i];
      a1 += x[(size_t)(l + 4) * nd + i];
    }
    if (l < L) a0 += x[(size_t)l * nd + i];
  }
  part[r][c] = a0 + a1;
  __syncthreads();
  if (r == 0 && i < nd) y[i] = (((part[0][c] + part[1][c]) + part[2][c]) + part[3][c]) * scale;
}
__global__ __launch_bounds__(256) void time_bcast_kernel(const float* __restrict__ dy, float* __restrict__ dx, int L,
                                                         long nd, float scale) {
  const int c = threadIdx.x & 63, r = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + c;
  if (i >= nd) return;
  const float g = dy[i] * scale;
  for (int l = r; l < L; l += 4) dx[(size_t)l * nd + i] = g;
}
extern "C" int mmvae_mean_over_time_fwd(const float* x, float* y, int L, int N, int d, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(x && y && L > 0 && N > 0 && d > 0);
  const long nd = (long)N * d;
  if (L < TIME_SUM_WIDE_L)
    hipLaunchKernelGGL(time_sum_serial_kernel, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y,
                       L, nd, 1.0f / (float)L);
  else
    hipLaunchKernelGGL(time_sum_kernel, dim3((unsigned)((nd + 63) / 64)), dim3(256), 0, (hipStream_t)stream, x, y, L,
                       nd, 1.0f / (float)L);
  return mmvae_launch_status();
}
extern "C" int mmvae_mean_over_time_bwd(const float* dy, float* dx, int L, int N, int d, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(dy && dx && L > 0 && N > 0 && d > 0);
  const long nd = (long)N * d;
  hipLaunchKernelGGL(time_bcast_kernel, dim3((unsigned)((nd + 63) / 64)), dim3(256), 0, (hipStream_t)stream, dy, dx,
                     L, nd, 1.0f / (float)L);
  return mmvae_launch_status();
}
extern "C" int mmvae_sum_over_time(const float* x, float* y, int L, int N, int d, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(x && y && L > 0 && N > 0 && d > 0);
  const long nd = (long)N * d;
  if (L < TIME_SUM_WIDE_L)
    hipLaunchKernelGGL(time_sum_serial_kernel, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y,
                       L, nd, 1.0f);
  else
    hipLaunchKernelGGL(time_sum_kernel, dim3((unsigned)((nd + 63) / 64)), dim3(256), 0, (hipStream_t)stream, x, y, L,
                       nd, 1.0f);
  return mmvae_launch_status();
}

// ---------------------------------------------------------------------------------------------
// decoder output: (T,B,V) -> (B,T,V) * mask[b,t]   (models/decoders.py:722) and its transpose
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void permute_mask_kernel(const float* __restrict__ x, const uint8_t* __restrict__ m,
                                                           float* __restrict__ y, int T, int B, int V, int fwd) {
  MMVAE_TRACE_STAMP(32);
  const long n = (long)T * B * V;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    // i indexes the (B,T,V) side
    const int v = (int)(i % V);
    const long bt = i / V;
    const int t = (int)(bt % T), b = (int)(bt / T);
    const long j = ((long)t * B + b) * V + v;  // (T,B,V) side
    // `output[~mask.T] = 0` is a masked WRITE (decoders.py:612,720): exact zeros forward, exact zero gradient backward
    // even where the incoming value is NaN / inf -- a select, not a multiplication
    const bool keep = m[bt] != 0;
    if (fwd) y[i] = keep ? x[j] : 0.f; else y[j] = keep ? x[i] : 0.f;
  }
}
// ... the first Tk of the T steps only: y (B,Tk,V) = x[:Tk] permuted and masked (mask (B,T)); the transpose writes the
// gradient of ALL T steps, zero from step Tk on (a PoE subset without the text modality decodes at full length and compares
// the first `mask length` steps: objectives.py:30-52 slice after decoders.py:722 -- slicing the permuted output instead cost
// a copy forward and a zero-fill + copy backward)
__global__ __launch_bounds__(256) void permute_mask_head_kernel(const float* __restrict__ x, const uint8_t* __restrict__ m,
                                                                float* __restrict__ y, int T, int B, int V, int Tk,
                                                                int fwd) {
  if (fwd) {
    const long n = (long)Tk * B * V;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {   // i: (B,Tk,V) side
      const int v = (int)(i % V);
      const long bt = i / V;
      const int t = (int)(bt % Tk), b = (int)(bt / Tk);
      y[i] = m[(long)b * T + t] != 0 ? x[((long)t * B + b) * V + v] : 0.f;
    }
  } else {
    const long n = (long)T * B * V;
    for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < n; j += (long)gridDim.x * 256) {   // j: (T,B,V) side
      const int v = (int)(j % V);
      const long tb = j / V;
      const int b = (int)(tb % B), t = (int)(tb / B);
      y[j] = (t < Tk && m[(long)b * T + t] != 0) ? x[((long)b * Tk + t) * V + v] : 0.f;
    }
  }
}
extern "C" int mmvae_permute_mask_head_fwd(const float* x, const uint8_t* mask, float* y, int T, int B, int V, int Tk,
                                           mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(x && mask && y && T > 0 && B > 0 && V > 0 && Tk > 0 && Tk <= T);
  long blocks = ((long)Tk * B * V + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(permute_mask_head_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, mask, y, T,
                     B, V, Tk, 1);
  return mmvae_launch_status();
}
extern "C" int mmvae_permute_mask_head_bwd(const float* dy, const uint8_t* mask, float* dx, int T, int B, int V, int Tk,
                                           mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(dy && mask && dx && T > 0 && B > 0 && V > 0 && Tk > 0 && Tk <= T);
  long blocks = ((long)T * B * V + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(permute_mask_head_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dy, mask, dx, T,
                     B, V, Tk, 0);
  return mmvae_launch_status();
}
extern "C" int mmvae_permute_mask_fwd(const float* x, const uint8_t* mask, float* y, int T, int B, int V,
                                      mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(x && mask && y && T > 0 && B > 0 && V > 0);
  long blocks = ((long)T * B * V + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(permute_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, mask, y, T, B,
                     V, 1);
  return mmvae_launch_status();
}
extern "C" int mmvae_permute_mask_bwd(const float* dy, const uint8_t* mask, float* dx, int T, int B, int V,
                                      mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(dy && mask && dx && T > 0 && B > 0 && V > 0);
  long blocks = ((long)T * B * V + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(permute_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dy, mask, dx, T, B,
                     V, 0);
  return mmvae_launch_status();
}

// ---------------------------------------------------------------------------------------------
// dropout utilities (train mode)
// ---------------------------------------------------------------------------------------------
__global__ void dropout_advance_kernel(uint32_t* st, uint32_t slot) {
  const uint32_t c = st[1] + 1u;
  st[1] = c;
  st[2 + slot] = c;
}
extern "C" int mmvae_dropout_advance(uint32_t* state, uint32_t slot, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(state && slot < MMVAE_DROPOUT_SLOTS);
  hipLaunchKernelGGL(dropout_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, slot);
  return mmvae_launch_status();
}
// call 0 of up to MMVAE_DROPOUT_ADVANCE_MAX towers in one launch (start of a training step, before the towers fork onto
// their streams): one graph node instead of a one-thread launch at the head of every tower's chain
struct DropAdvanceMany {
  uint32_t* st[MMVAE_DROPOUT_ADVANCE_MAX];
  int n;
};
__global__ void dropout_advance_many_kernel(DropAdvanceMany a) {
  MMVAE_TRACE_STAMP(33);
  const int i = threadIdx.x;
  if (i < a.n) {
    uint32_t* st = a.st[i];
    const uint32_t c = st[1] + 1u;
    st[1] = c;
    st[2] = c;      // slot 0
  }
}
extern "C" int mmvae_dropout_advance_many(uint32_t* const* states, int n, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(states && n > 0 && n <= MMVAE_DROPOUT_ADVANCE_MAX);
  DropAdvanceMany a;
  for (int i = 0; i < n; ++i) {
    if (!states[i]) return MMVAE_ERR_ARG;
    for (int j = 0; j < i; ++j)
      if (states[j] == states[i]) return MMVAE_ERR_ARG;      // one advance per state
    a.st[i] = states[i];
  }
  a.n = n;
  hipLaunchKernelGGL(dropout_advance_many_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
  return mmvae_launch_status();
}
__global__ __launch_bounds__(256) void dropout_mask_kernel(float* __restrict__ out, long n, mmvae_dropout_t drop) {
  const DropKey dk = drop_key(drop);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = drop_mul(dk, (uint32_t)i);
}
extern "C" int mmvae_dropout_mask(const mmvae_dropout_t* drop, float* out, long n, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(drop && out && n > 0);
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, n, *drop);
  return mmvae_launch_status();
}
__global__ __launch_bounds__(256) void dropout_act_kernel(const float* __restrict__ a, const float* __restrict__ x,
                                                          float* __restrict__ out, long n, int act, int bwd,
                                                          mmvae_dropout_t drop) {
  // fwd: out = act(x) * m ; bwd: out = a(=dy) * m * act'(x)
  MMVAE_TRACE_STAMP(34);
  const DropKey dk = drop_key(drop);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float m = drop_mul(dk, (uint32_t)i);
    const float xv = x[i];
    float v;
    if (!bwd) v = apply_in_act(xv, act) * m;
    else v = a[i] * m * (act == MMVAE_ACT_GELU ? dev_gelu_grad(xv) : 1.0f);
    out[i] = v;
  }
}
extern "C" int mmvae_dropout_act_fwd(const float* x, float* y, long n, int act, const mmvae_dropout_t* drop,
                                     mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(x && y && n > 0 && (act == MMVAE_ACT_NONE || act == MMVAE_ACT_GELU));
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(dropout_act_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, nullptr, x, y, n,
                     act, 0, drop_arg(drop));
  return mmvae_launch_status();
}
extern "C" int mmvae_dropout_act_bwd(const float* dy, const float* x, float* dx, long n, int act,
                                     const mmvae_dropout_t* drop, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(dy && x && dx && n > 0 && (act == MMVAE_ACT_NONE || act == MMVAE_ACT_GELU));
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(dropout_act_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dy, x, dx, n, act,
                     1, drop_arg(drop));
  return mmvae_launch_status();
}
// out[l,n,c] = v[n,c] * mask[(n*H + c/hd)*L + l]   (attention-weight dropout over a length-1 memory)
// forward: one thread per OUTPUT element (round 5: it was one thread per (n, c) walking the L steps -- 16 workgroups at the
// action decoder's 128 x 32 x 100, 20 us per call); backward: one workgroup per n, its 256 threads = (l mod 8, c), summed
// over the 8 time groups through LDS in a fixed order.
__global__ __launch_bounds__(256) void head_bcast_dropout_fwd_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                                     int L, int N, int H, int hd, mmvae_dropout_t drop) {
  const DropKey dk = drop_key(drop);
  const int E = H * hd;
  const long ne = (long)N * E, tot = ne * L;
  for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < tot; j += (long)gridDim.x * 256) {
    const int l = (int)(j / ne);
    const long i = j - (long)l * ne;
    const int n = (int)(i / E), c = (int)(i - (long)n * E), h = c / hd;
    dst[j] = src[i] * drop_mul(dk, (uint32_t)(((size_t)n * H + h) * L) + (uint32_t)l);
  }
}
#define HB_TG 8
__global__ __launch_bounds__(256) void head_bcast_dropout_bwd_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                                     int L, int N, int H, int hd, mmvae_dropout_t drop) {
  __shared__ float red[256];
  const DropKey dk = drop_key(drop);
  const int E = H * hd, n = blockIdx.x;
  const long ne = (long)N * E;
  // E <= 32: thread = (time group tid / 32, channel tid % 32); wider rows: channel loop
  for (int c0 = 0; c0 < E; c0 += 32) {
    const int c = c0 + (threadIdx.x & 31), tg = threadIdx.x >> 5;
    float a = 0.f;
    if (c < E) {
      const uint32_t base = (uint32_t)(((size_t)n * H + c / hd) * L);
      for (int l = tg; l < L; l += HB_TG) a += src[(size_t)l * ne + (size_t)n * E + c] * drop_mul(dk, base + l);
    }
    red[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x < 32 && c < E) {
      float s_ = 0.f;
#pragma unroll
      for (int g = 0; g < HB_TG; ++g) s_ += red[g * 32 + threadIdx.x];
      dst[(size_t)n * E + c] = s_;
    }
    __syncthreads();
  }
}
// y[t,b,:] = dropout(x[t,b,:] + pe[t,:]): the time positional encoding of the action Transformer towers
// (PositionalEncoding.forward's try-branch, models/nn_modules.py:430-438); x == NULL: the decoders' time queries
// PE(zeros).  Backward is mmvae_dropout_act_bwd with MMVAE_ACT_NONE (same element index).
__global__ __launch_bounds__(256) void add_pe_dropout_kernel(const float* __restrict__ x, const float* __restrict__ pe,
                                                             float* __restrict__ y, int T, long BD, int D,
                                                             mmvae_dropout_t drop) {
  const DropKey dk = drop_key(drop);
  const long n = (long)T * BD;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int t = (int)(i / BD), d = (int)(i % D);
    y[i] = ((x ? x[i] : 0.f) + pe[(size_t)t * D + d]) * drop_mul(dk, (uint32_t)i);
  }
}
extern "C" int mmvae_add_pe_dropout_fwd(const float* x, const float* pe, float* y, int T, int B, int D,
                                        const mmvae_dropout_t* drop, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(pe && y && T > 0 && B > 0 && D > 0);
  const long n = (long)T * B * D;
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(add_pe_dropout_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, pe, y, T,
                     (long)B * D, D, drop_arg(drop));
  return mmvae_launch_status();
}

extern "C" int mmvae_head_bcast_dropout_fwd(const float* v, float* out, int L, int N, int H, int hd,
                                            const mmvae_dropout_t* drop, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(v && out && L > 0 && N > 0 && H > 0 && hd > 0);
  long blocks = ((long)L * N * H * hd + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(head_bcast_dropout_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, v, out, L, N,
                     H, hd, drop_arg(drop));
  return mmvae_launch_status();
}
extern "C" int mmvae_head_bcast_dropout_bwd(const float* dout, float* dv, int L, int N, int H, int hd,
                                            const mmvae_dropout_t* drop, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(dout && dv && L > 0 && N > 0 && H > 0 && hd > 0);
  hipLaunchKernelGGL(head_bcast_dropout_bwd_kernel, dim3((unsigned)N), dim3(256), 0, (hipStream_t)stream, dout, dv, L, N, H,
                     hd, drop_arg(drop));
  return mmvae_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Input expansion on the device (the step in front of the path).  The reference's datasets hold
// fp32 images `torch.tensor(uint8) / 255` (models/datasets.py:251-254) and fp32 one-hot text with a mask column
// (`one_hot_encode`, utils.py:414-421; `lengths_to_mask`, utils.py:239; models/datasets.py:272-281): 49 KB + 3.5 KB per
// sample over PCIe.  Shipping the uint8 pixels and one token id per character (12 KB + 0.13 KB) and expanding here is
// bit-identical and a quarter of the host traffic.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void expand_u8_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, long n) {
  const long stride = (long)gridDim.x * 256 * 4;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n && (((uintptr_t)(src + i)) & 3) == 0 && (((uintptr_t)(dst + i)) & 15) == 0) {
      const uint32_t w = *reinterpret_cast<const uint32_t*>(src + i);
      float4 o;      // IEEE division, as torch's true_divide of the converted value
      o.x = __fdiv_rn((float)(w & 255u), 255.0f);
      o.y = __fdiv_rn((float)((w >> 8) & 255u), 255.0f);
      o.z = __fdiv_rn((float)((w >> 16) & 255u), 255.0f);
      o.w = __fdiv_rn((float)(w >> 24), 255.0f);
      *reinterpret_cast<float4*>(dst + i) = o;
    } else {
      for (long j = i; j < n && j < i + 4; ++j) dst[j] = __fdiv_rn((float)src[j], 255.0f);
    }
  }
}
extern "C" int mmvae_expand_image_u8(const uint8_t* src, float* dst, long n, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(src && dst && n > 0);
  long blocks = (n + 1023) / 1024;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(expand_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, n);
  return mmvae_launch_status();
}
// tokens (B,T) int32: character index in the alphabet, -1 = not in the alphabet (an all-zero row, as one_hot_encode);
// lengths (B): characters per sample.  onehot (B,T,V) fp32, mask (B,T) bytes (1 = data, 0 = padding).
__global__ __launch_bounds__(256) void expand_tokens_kernel(const int32_t* __restrict__ tok, const int32_t* __restrict__ len,
                                                            float* __restrict__ onehot, uint8_t* __restrict__ mask,
                                                            int B, int T, int V) {
  const long n = (long)B * T * V;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int v = (int)(i % V);
    const long bt = i / V;
    const int t = (int)(bt % T), b = (int)(bt / T);
    const bool in = t < len[b];
    onehot[i] = (in && tok[bt] == v) ? 1.0f : 0.0f;
    if (v == 0 && mask) mask[bt] = in ? 1 : 0;
  }
}
extern "C" int mmvae_expand_text_tokens(const int32_t* tokens, const int32_t* lengths, float* onehot, uint8_t* mask,
                                        int B, int T, int V, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(tokens && lengths && onehot && B > 0 && T > 0 && V > 0);
  long blocks = ((long)B * T * V + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(expand_tokens_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, tokens, lengths,
                     onehot, mask, B, T, V);
  return mmvae_launch_status();
}

MMVAE_TRACE_SETTER(text)
