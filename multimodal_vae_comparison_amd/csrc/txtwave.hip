// One WAVE = one sequence: a whole post-norm Transformer layer of the text towers chained through registers (gfx950).
//
// Same arithmetic, same saved tensors and same dropout masks as csrc/txtlayer.hip (torch.nn.TransformerEncoderLayer /
// TransformerDecoderLayer over a length-1 memory: models/encoders.py:806-837, models/decoders.py:686-723), other mapping.
// txtlayer.hip gives a sequence a 4-wave workgroup and 68-98 KB of LDS: every GEMM phase is load -> barrier -> <= 2 tiles
// of MFMA per wave -> barrier, 57-67 % of the wave cycles parked, two workgroups per CU at most and none beside a conv
// workgroup of the other tower (profiles/r03_pmc_gemm_txt.txt; at batch 1000 the four layer kernels are 0.7 ms of a
// 2.0 ms step).  Here T <= 32 tokens sit on the LANES of one wave for the whole layer:
//
//   * "token layout" (TL): a (32 x C) activation is C/32 accumulator tiles; lane (token, half) register r holds
//     column 32 t + 8 (r >> 2) + 4 half + (r & 3) -- exactly what v_mfma_f32_32x32x2_f32 leaves when the tokens were
//     its B operand.  Y = X W^T is then mfma(A = W rows, B = X registers): the k-step that consumes register r of X
//     takes W[n][that column] from lane n -- four consecutive k per 16-byte load straight from L2 -- and Y comes out in
//     TL again.  A whole MLP / projection chain therefore never leaves the registers: no LDS, no barrier, no transposition.
//   * swapped operand roles, mfma(A = X registers, B = W rows), leave the result with the OUTPUT COLUMN on the lanes and
//     the tokens in the registers ("DL").  V is produced in DL, so that O^T = V^T P^T is mfma(A = V (DL), B = P) with P the
//     softmax of S^T = K Q^T (A = K (TL), B = Q (TL): lane = query, registers = keys): the attention needs no transposition
//     either, the softmax is 16 registers + one cross-half shuffle, and O^T lands in TL for out_proj.
//   * LayerNorm is a per-lane sum over the lane's registers + one shuffle with lane ^ 32.
//   * weights stream through a 3-slot register ring of 16-k-step units, loaded two units (>= 2 K cycles of MFMA) ahead
//     across GEMM boundaries, so a single wave per SIMD still hides its L2 latency.
// One 64-thread workgroup per sequence: 128 sequences occupy 128 of the chip's 1024 SIMDs and no LDS beyond a pooling
// scratch; 1000 sequences are one round of one wave per SIMD.
#include <type_traits>
#include <utility>

#include "common.hpp"

namespace tv {

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
// a select between two pointers loses the address space and turns the access into flat_load / flat_store (which also
// counts on lgkmcnt): cast it back to global
typedef __attribute__((address_space(1))) float gfloat;
typedef __attribute__((address_space(1))) f4u gf4u;
__device__ __forceinline__ void gstore(float* p, const float v) { *(gfloat*)p = v; }
__device__ __forceinline__ void gstore4(float* p, const f4u v) { *(gf4u*)p = v; }
__device__ __forceinline__ float gload(const float* p) { return *(const gfloat*)p; }
__device__ __forceinline__ f4u gload4(const float* p) { return *(const gf4u*)p; }
template <int V>
using IC = std::integral_constant<int, V>;
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(IC<I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

__host__ __device__ constexpr int cmin(int a, int b) { return a < b ? a : b; }
__host__ __device__ constexpr int tcol(int r) { return 8 * (r >> 2) + (r & 3); }   // + 4 * half

template <int D_, int FF_, int NH_, bool DEC_>
struct Geom {
  static constexpr int D = D_, FF = FF_, NH = NH_, HD = D / NH;
  static constexpr bool DEC = DEC_;
  static constexpr int DT = (D + 31) / 32, FT = FF / 32;
  static constexpr bool PACK = (HD % 8 == 0) && D <= 32;     // all heads in one 32-column tile
  static constexpr int AT = PACK ? 1 : NH;                   // attention tiles
  static constexpr int AV = PACK ? D : HD;                   // valid columns of an attention tile
  static_assert(D % 2 == 0 && D <= 64 && FF % 32 == 0 && FF <= 128 && D % NH == 0 && HD <= 32, "shape");
  __host__ __device__ static constexpr int dvalid(int i) { return cmin(32, D - 32 * i); }
  __host__ __device__ static constexpr int abase(int a) { return PACK ? 0 : a * HD; }
};

// -DTV_PROBE (tools/probe/txtwave_stamps.py): workgroup 0 stamps s_memtime at the phase boundaries of the layer
#ifdef TV_PROBE
__device__ long long tv_stamps[2][32];
#define TV_STAMP(dir, i)                                                        \
  do {                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                          \
    const long long t__ = __builtin_amdgcn_s_memtime();                         \
    if (blockIdx.x == 0 && threadIdx.x == 0) tv_stamps[dir][i] = t__;           \
    __builtin_amdgcn_sched_barrier(0);                                          \
  } while (0)
#else
#define TV_STAMP(dir, i)
#endif

struct Ctx {
  int lane, li, lh;
  int n, N, L;
  float* trash;        // this lane's 16-byte slot for masked stores (see tl_store)
  const float* zero;   // >= 64 readable zeros
};
__device__ const uint32_t tv_zero[64] = {0};          // zeros: weight rows of padding lanes, generator words of a disabled dropout site

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}

// GELU / GELU' as csrc/ffn.hip: Abramowitz-Stegun 7.1.26 (|erf error| <= 1.5e-7) on the hardware exp / rcp.  ocml's erff
// is ~45 VALU instructions per element, which a single wave per SIMD cannot hide behind anything.
__device__ __forceinline__ void gelu_parts(float x, float& tail, float& e) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678f, ax, 1.0f));
  e = __expf(-0.5f * x * x);
  float poly = fmaf(t, 1.061405429f, -1.453152027f);
  poly = fmaf(t, poly, 1.421413741f);
  poly = fmaf(t, poly, -0.284496736f);
  poly = fmaf(t, poly, 0.254829592f);
  tail = 0.5f * (poly * t) * e;
}
__device__ __forceinline__ float gelu(float x) {
  float tail, e;
  gelu_parts(x, tail, e);
  return x * (x >= 0.f ? 1.0f - tail : tail);
}
__device__ __forceinline__ float gelu_grad(float x) {
  float tail, e;
  gelu_parts(x, tail, e);
  const float cdf = x >= 0.f ? 1.0f - tail : tail;
  return fmaf(x * 0.3989422804f, e, cdf);
}

// dropout, branch-free (a uniform `if (!on)` is a basic-block boundary too): the key of a disabled site has thr = 0 and
// inv_keep = 1, so every element is "kept" with multiplier 1.  Same masks as drop_mul per element.
__device__ __forceinline__ DropKey drop_key_nb(const mmvae_dropout_t& d) {
  DropKey k;
  k.on = d.state != nullptr && d.p > 0.f;
  const uint32_t* st = k.on ? d.state : tv_zero;
  k.p = k.on ? d.p : 0.f;
  k.inv_keep = 1.0f / (1.0f - k.p);
  k.thr = (uint32_t)(k.p * 65536.0f + 0.5f);
  k.key = drop_fmix(st[0] ^ (st[2 + (k.on ? d.slot : 0u)] * 0x9E3779B1u) ^ (d.site * 0x85EBCA77u + 0x165667B1u));
  return k;
}
__device__ __forceinline__ float nb_lo(const DropKey& k, uint32_t h) { return (h & 0xFFFFu) >= k.thr ? k.inv_keep : 0.0f; }
__device__ __forceinline__ float nb_hi(const DropKey& k, uint32_t h) { return (h >> 16) >= k.thr ? k.inv_keep : 0.0f; }
__device__ __forceinline__ float drop1_nb(const DropKey& k, const uint32_t idx) {
  const uint32_t h = drop_pair_hash(k, idx >> 1);
  return ((idx & 1u) ? (h >> 16) : (h & 0xFFFFu)) >= k.thr ? k.inv_keep : 0.0f;
}
// multipliers of 4 consecutive elements idx0 .. idx0 + 3
__device__ __forceinline__ void drop4_even(const DropKey& k, const uint32_t idx0, float (&m)[4]) {   // idx0 even
  const uint32_t h0 = drop_pair_hash(k, idx0 >> 1), h1 = drop_pair_hash(k, (idx0 >> 1) + 1);
  m[0] = nb_lo(k, h0);
  m[1] = nb_hi(k, h0);
  m[2] = nb_lo(k, h1);
  m[3] = nb_hi(k, h1);
}
__device__ __forceinline__ void drop4_any(const DropKey& k, const uint32_t idx0, float (&m)[4]) {
  // any parity, still without a branch: elements idx0 + b are halves (odd + b) of the 16-bit sequence h0 | h1 | h2
  const uint32_t p = idx0 >> 1;
  const uint32_t sh = (idx0 & 1u) << 4;
  const uint32_t h0 = drop_pair_hash(k, p), h1 = drop_pair_hash(k, p + 1), h2 = drop_pair_hash(k, p + 2);
  const uint32_t w01 = __funnelshift_r(h0, h1, sh), w23 = __funnelshift_r(h1, h2, sh);
  m[0] = nb_lo(k, w01);
  m[1] = nb_hi(k, w01);
  m[2] = nb_lo(k, w23);
  m[3] = nb_hi(k, w23);
}

// ---- one weight unit: the <= 16 k-steps that consume one TL tile of the input ---------------------------------------
// KC: W[n][k] (k contiguous) -- lane (n, half) loads W[nbase + n][kbase + 8 g + 4 half + 0..3];  !KC: W[k][n].
// Rows n >= NVALID and columns >= KVALID read as zero.
template <bool KC, int NVALID, int KVALID>
__device__ __forceinline__ void wload(float (&w)[16], const float* __restrict__ W, const int ldw, const int nbase,
                                      const int kbase, const Ctx& c) {
  // lanes without an output row read a row of zeros (one pointer select per unit instead of one value select per k-step)
  const bool nok = NVALID >= 32 || c.li < NVALID;
  if (KC) {
    const float* p = nok ? W + (size_t)(nbase + c.li) * ldw + kbase + 4 * c.lh : c.zero;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (8 * g < KVALID) {
        if (8 * g + 8 <= KVALID) {
          const f4u v = gload4(p + 8 * g);
#pragma unroll
          for (int b = 0; b < 4; ++b) w[4 * g + b] = v[b];
        } else {
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const bool ok = (8 * g + 4 * c.lh + b) < KVALID;
            w[4 * g + b] = gload(ok ? p + 8 * g + b : c.zero);
          }
        }
      }
    }
  } else {
    const float* p = W + (size_t)(kbase + 4 * c.lh) * ldw + nbase + (nok ? c.li : 0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (8 * g + b < KVALID) {
          const bool ok = nok && ((8 * g + 4 + b < KVALID) || c.lh == 0);     // the row exists for this half
          w[4 * g + b] = gload(ok ? p + (size_t)(8 * g + b) * ldw : c.zero);
        }
      }
    }
  }
}
// SWAP false: acc (TL: lane = token, registers = output columns) += W_unit x;  true: (DL: lane = output column,
// registers = tokens)
template <int KVALID, bool SWAP>
__device__ __forceinline__ void mma_unit(f32x16& acc, const float (&w)[16], const f32x16& x) {
#pragma unroll
  for (int r = 0; r < 16; ++r)
    if (tcol(r) < KVALID) acc = SWAP ? mfma(x[r], w[r], acc) : mfma(w[r], x[r], acc);
}

// sum / max over the two halves of the wave (lane ^ 32) without the LDS crossbar: v_permlane32_swap
__device__ __forceinline__ float half_sum(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float half_max(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// ---- TL tiles <-> (L, N, C) tensors: p = the lane's row + the tile's first column + 4 * half ---------------------------
// A divergent `if (ok) store` splits the kernel into basic blocks, and hipcc schedules -- and counts s_waitcnt -- per block:
// with ~120 of them the first version had no MFMA / VALU overlap at all and 45 full vmcnt(0) drains.  Masked lanes
// therefore store too, into a per-lane trash slot: the layer is ONE straight-line region.
template <int NVALID>
__device__ __forceinline__ void tl_store(float* __restrict__ p, const f32x16& v, const bool tok_ok, const Ctx& c) {
  float* const tr = c.trash;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if (8 * g < NVALID) {
      if (8 * g + 8 <= NVALID) {
        f4u o;
        o[0] = v[4 * g]; o[1] = v[4 * g + 1]; o[2] = v[4 * g + 2]; o[3] = v[4 * g + 3];
        gstore4(tok_ok ? p + 8 * g : tr, o);
      } else {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const bool ok = tok_ok && (8 * g + 4 * c.lh + b) < NVALID;
          gstore(ok ? p + 8 * g + b : tr + b, v[4 * g + b]);
        }
      }
    }
  }
}
// `safe` = an address that may always be read (masked lanes load it and discard the value)
template <int NVALID>
__device__ __forceinline__ f32x16 tl_load(const float* __restrict__ p, const float* __restrict__ safe, const bool tok_ok,
                                          const Ctx& c) {
  f32x16 v = zero16();
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if (8 * g < NVALID) {
      if (8 * g + 8 <= NVALID) {
        const f4u o = gload4(tok_ok ? p + 8 * g : safe);
#pragma unroll
        for (int b = 0; b < 4; ++b) v[4 * g + b] = tok_ok ? o[b] : 0.f;
      } else {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const bool ok = tok_ok && (8 * g + 4 * c.lh + b) < NVALID;
          const float o = gload(ok ? p + 8 * g + b : safe);
          v[4 * g + b] = ok ? o : 0.f;
        }
      }
    }
  }
  return v;
}
// a vector (bias, LayerNorm weight) in TL (p includes + 4 * half): the same 16 columns for every token of the half
template <int NVALID>
__device__ __forceinline__ f32x16 vec_load(const float* __restrict__ p, const Ctx& c) {
  return tl_load<NVALID>(p, p, true, c);
}

template <int NVALID>
__device__ __forceinline__ bool col_ok(const int r, const Ctx& c) {
  return NVALID >= 32 || (tcol(r) + 4 * c.lh) < NVALID;
}

// LayerNorm over the D columns of a TL activation (padding columns hold zeros).  out(i, r, xhat, y), returns rstd.
template <typename G, typename Out>
__device__ __forceinline__ float layernorm_tl(const f32x16 (&v)[G::DT], const float* __restrict__ gamma,
                                              const float* __restrict__ beta, const Ctx& c, Out&& out) {
  constexpr int D = G::D;
  f32x16 gm[G::DT], bt[G::DT];
  static_for<G::DT>([&](auto i) {
    gm[i] = vec_load<G::dvalid(i)>(gamma + 32 * i + 4 * c.lh, c);
    bt[i] = vec_load<G::dvalid(i)>(beta + 32 * i + 4 * c.lh, c);
  });
  float s = 0.f;
  static_for<G::DT>([&](auto i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s += v[i][r];
  });
  s = half_sum(s);
  const float mean = s / (float)D;
  float ss = 0.f;
  static_for<G::DT>([&](auto i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float d = col_ok<G::dvalid(i)>(r, c) ? v[i][r] - mean : 0.f;
      ss += d * d;
    }
  });
  ss = half_sum(ss);
  const float rs = rsqrtf(ss / (float)D + 1e-5f);
  static_for<G::DT>([&](auto i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bool ok = col_ok<G::dvalid(i)>(r, c);
      const float xh = ok ? (v[i][r] - mean) * rs : 0.f;
      out(i, r, xh, ok ? xh * gm[i][r] + bt[i][r] : 0.f);
    }
  });
  return rs;
}

}  // namespace tv

using namespace tv;

// ------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------
namespace tv {

template <typename T>
__device__ __forceinline__ T* bptr(T* base, const uint32_t byte_off) {      // uniform base + 32-bit lane offset
  return reinterpret_cast<T*>(reinterpret_cast<uintptr_t>(base) + byte_off);
}

// The layer's weight units in PROGRAM ORDER, grouped into BATCHES of <= 4 units: while batch b computes (its MFMAs and
// the vector work that program order places behind them), the weights of batch b + 1 are already in flight into the
// other half of a two-batch register buffer.  A batch begins with those loads and a full scheduling barrier:
// hipcc otherwise sinks every weight load next to its first use (one 16-byte load ahead of 4 MFMAs: the L2 latency
// then stalls half of every group), and nothing else keeps it from hoisting the MFMAs above the loads.
enum { WS_IN = 0, WS_OUT, WS_XOUT, WS_L1, WS_L2 };
struct UD {
  int ws, ldw, nbase, kbase, nvalid, kvalid, swap;
  int kc = 1;      // W[n][k] (k contiguous); 0: W[k][n]
};
struct BD {
  int start, len;
};
template <typename G>
struct FwdPlan {
  static constexpr int D = G::D, DT = G::DT, FT = G::FT, AT = G::AT;
  static_assert(FT >= 2 && 2 * DT <= 4, "feed-forward interleave / batch size");
  static constexpr int NA = AT * 3 * DT, NB = DT * AT, NC = G::DEC ? DT * DT : 0, NF = 2 * FT * DT;
  static constexpr int A0 = 0, B0 = NA, C0 = B0 + NB, F0 = C0 + NC, NU = F0 + NF;
  // in_proj: per attention tile Q, K (token layout), V (column layout)
  __host__ __device__ static constexpr int posA(int a, int part, int j) { return A0 + (a * 3 + part) * DT + j; }
  // out_proj (output tile i, attention tile a): a-major, so that a finished head's units can run under the next softmax
  __host__ __device__ static constexpr int posB(int i, int a) { return B0 + a * DT + i; }
  __host__ __device__ static constexpr int posC(int i, int j) { return C0 + i * DT + j; }
  // feed-forward block interleaved: L1(0) L1(1) [L1(2) L2(.,0)] [L1(3) L2(.,1)] ... [L2(.,FT-1)]
  __host__ __device__ static constexpr int posL1(int i, int j) {
    return F0 + (i < 2 ? i * DT : 2 * DT + (i - 2) * 2 * DT) + j;
  }
  __host__ __device__ static constexpr int posL2(int i, int jj) {
    int p = F0 + 2 * DT;
    for (int g = 0; g < jj; ++g) p += ((g + 2 < FT) ? DT : 0) + DT;
    return p + ((jj + 2 < FT) ? DT : 0) + i;
  }
  // batches
  static constexpr int NBAT = 3 * AT + AT + (G::DEC ? 1 : 0) + 2 + FT;
  __host__ __device__ static constexpr int batA(int a, int part) { return 3 * a + part; }
  __host__ __device__ static constexpr int batB(int a) { return 3 * AT + a; }
  __host__ __device__ static constexpr int batC() { return 4 * AT; }
  __host__ __device__ static constexpr int batL1(int i) { return 4 * AT + (G::DEC ? 1 : 0) + i; }        // i = 0, 1
  __host__ __device__ static constexpr int batG(int jj) { return 4 * AT + (G::DEC ? 1 : 0) + 2 + jj; }   // [L1(jj+2)] L2(.,jj)
  struct Table {
    UD u[NU];
    BD b[NBAT + 1];
    int bat_of[NU];
  };
};
template <typename G>
__host__ __device__ constexpr typename FwdPlan<G>::Table make_fwd_table() {
  using P = FwdPlan<G>;
  typename P::Table t{};
  for (int a = 0; a < G::AT; ++a)
    for (int part = 0; part < 3; ++part) {
      t.b[P::batA(a, part)] = BD{P::posA(a, part, 0), G::DT};
      for (int j = 0; j < G::DT; ++j)
        t.u[P::posA(a, part, j)] = UD{WS_IN, G::D, part * G::D + G::abase(a), 32 * j, G::AV, G::dvalid(j), part == 2};
    }
  for (int a = 0; a < G::AT; ++a) {
    t.b[P::batB(a)] = BD{P::posB(0, a), G::DT};
    for (int i = 0; i < G::DT; ++i) t.u[P::posB(i, a)] = UD{WS_OUT, G::D, 32 * i, G::abase(a), G::dvalid(i), G::AV, 0};
  }
  if (G::DEC) {
    t.b[P::batC()] = BD{P::posC(0, 0), G::DT * G::DT};
    for (int i = 0; i < G::DT; ++i)
      for (int j = 0; j < G::DT; ++j)
        t.u[P::posC(i, j)] = UD{WS_XOUT, G::D, 32 * i, 32 * j, G::dvalid(i), G::dvalid(j), 0};
  }
  for (int i = 0; i < G::FT; ++i)
    for (int j = 0; j < G::DT; ++j) t.u[P::posL1(i, j)] = UD{WS_L1, G::D, 32 * i, 32 * j, 32, G::dvalid(j), 0};
  for (int i = 0; i < G::DT; ++i)
    for (int jj = 0; jj < G::FT; ++jj) t.u[P::posL2(i, jj)] = UD{WS_L2, G::FF, 32 * i, 32 * jj, G::dvalid(i), 32, 0};
  t.b[P::batL1(0)] = BD{P::posL1(0, 0), G::DT};
  t.b[P::batL1(1)] = BD{P::posL1(1, 0), G::DT};
  for (int jj = 0; jj < G::FT; ++jj) {
    const bool l1 = jj + 2 < G::FT;
    t.b[P::batG(jj)] = BD{l1 ? P::posL1(jj + 2, 0) : P::posL2(0, jj), (l1 ? G::DT : 0) + G::DT};
  }
  t.b[P::NBAT] = BD{P::NU, 0};
  for (int b = 0; b < P::NBAT; ++b)
    for (int k = 0; k < t.b[b].len; ++k) t.bat_of[t.b[b].start + k] = b;
  return t;
}
template <typename G>
inline constexpr typename FwdPlan<G>::Table fwd_table = make_fwd_table<G>();

}  // namespace tv

using namespace tv;

// Program order puts every batch of INDEPENDENT MFMAs in front of the VALU block it can hide (a wave issues in order: a
// dependent MFMA chain leaves ~60 of every 64 cycles to whatever follows it in the stream): K's MFMAs before Q's bias /
// store, the next head's Q (or the finished head's out_proj units) before a softmax, linear1 tile i + 1 and the linear2
// units of tile i - 1 before tile i's GELU / dropout.  Bias / LayerNorm vectors are loaded one batch early as well.
template <typename G, bool FULL>
__global__ __launch_bounds__(64) void txt_wave_fwd_kernel(const float* __restrict__ x, const uint8_t* __restrict__ valid,
                                                          const float* __restrict__ mem, float* __restrict__ y,
                                                          const mmvae_txt_layer_w_t w, const mmvae_txt_layer_saved_t sv,
                                                          const mmvae_txt_layer_drop_t dr, const int L, const int N,
                                                          const int time_mean, const float* __restrict__ head_w,
                                                          const float* __restrict__ head_b, float* __restrict__ heads,
                                                          const int HN, float* __restrict__ trash) {
  MMVAE_TRACE_STAMP(16 + (G::DEC ? 1 : 0));
  using P = FwdPlan<G>;
  constexpr int D = G::D, FF = G::FF, NH = G::NH, HD = G::HD, DT = G::DT, FT = G::FT, AT = G::AT, AV = G::AV;
  constexpr bool DEC = G::DEC, PACK = G::PACK;
  constexpr int HPT = PACK ? NH : 1;                  // heads per attention tile
  constexpr int PP = D | 1;                           // pooling scratch pitch
  __shared__ float pool[32 * PP + 64];
  Ctx c;
  c.lane = threadIdx.x;
  c.li = c.lane & 31;
  c.lh = c.lane >> 5;
  c.n = blockIdx.x;
  c.N = N;
  c.L = L;
  c.trash = trash + 4 * c.lane;
  c.zero = reinterpret_cast<const float*>(tv_zero);
  const int tok = c.li;
  const bool tok_ok = FULL || tok < L;                               // FULL: L == 32, every lane owns a token
  const uint32_t row = (uint32_t)(tok_ok ? tok : 0) * N + c.n;      // this lane's row of every (L, N, .) tensor
  const uint32_t rD = row * (D * 4) + 16 * c.lh, rF = row * (FF * 4) + 16 * c.lh, rQ = row * (3 * D * 4) + 16 * c.lh;

  float wb[2][4][16];
  auto batch_load = [&](auto b_) {
    constexpr int b = b_;
    if constexpr (b < P::NBAT) {
      constexpr BD bd = fwd_table<G>.b[b];
      static_for<bd.len>([&](auto k) {
        constexpr UD d = fwd_table<G>.u[bd.start + k];
        const float* W = d.ws == WS_IN ? w.in_w : d.ws == WS_OUT ? w.out_w : d.ws == WS_XOUT ? w.x_out_w
                       : d.ws == WS_L1 ? w.l1_w : w.l2_w;
        wload<true, d.nvalid, d.kvalid>(wb[b & 1][k], W, d.ldw, d.nbase, d.kbase, c);
      });
    }
  };
  // start of batch b: the next batch's weights go out, and nothing moves across this point
  auto begin = [&](auto b_) {
    constexpr int b = b_;
    batch_load(IC<b + 1>{});
    __builtin_amdgcn_sched_barrier(0);
  };
  auto mma = [&](auto u_, f32x16& acc, const f32x16& xop) {
    constexpr int u = u_;
    constexpr UD d = fwd_table<G>.u[u];
    constexpr int b = fwd_table<G>.bat_of[u];
    mma_unit<d.kvalid, d.swap != 0>(acc, wb[b & 1][u - fwd_table<G>.b[b].start], xop);
  };
  TV_STAMP(0, 0);
  batch_load(IC<0>{});

  // ---- layer input (TL) and the key mask ----
  f32x16 xin[DT];
  static_for<DT>([&](auto i) { xin[i] = tl_load<G::dvalid(i)>(bptr(x, rD + 128 * i), x, tok_ok, c); });
  const unsigned long long vb64 = __ballot(tok_ok && valid[(size_t)c.n * L + (tok_ok ? tok : 0)] != 0);
  const uint32_t key_bits = (uint32_t)vb64;           // bit k: key k is a real token

  TV_STAMP(0, 1);
  // ================= self-attention =================
  f32x16 ao[AT], oacc[DT];
  static_for<AT>([&](auto a) { ao[a] = zero16(); });
  static_for<DT>([&](auto i) { oacc[i] = zero16(); });
  f32x16 bo[DT];                                      // out_proj bias (loaded under the last attention tile)
  {
    const DropKey dk = drop_key_nb(dr.attn);
    const uint32_t vstride = (uint32_t)N * (3 * D * 4);                           // one token of (L, N, 3D), bytes
    const uint32_t voff = ((uint32_t)(4 * c.lh) * N + c.n) * (3 * D * 4) + 4 * c.li;   // V store: token 4 half, column li
    const bool dv_ok = c.li < AV;
    f32x16 q, k, vt, qn = zero16();
    f32x16 bq = vec_load<AV>(w.in_b + G::abase(0) + 4 * c.lh, c), bqn, bk;
    float bv;
    begin(IC<P::batA(0, 0)>{});
    static_for<DT>([&](auto j) { mma(IC<P::posA(0, 0, j)>{}, qn, xin[j]); });
    static_for<AT>([&](auto a_) {
      constexpr int a = a_;
      q = qn;
      k = zero16();
      bk = vec_load<AV>(w.in_b + D + G::abase(a) + 4 * c.lh, c);
      begin(IC<P::batA(a, 1)>{});
      static_for<DT>([&](auto j) { mma(IC<P::posA(a, 1, j)>{}, k, xin[j]); });
      // Q: bias, store (under K's MFMAs)
#pragma unroll
      for (int r = 0; r < 16; ++r) q[r] += bq[r];        // (padding columns: weight rows and bias read as zero)
      tl_store<AV>(bptr(sv.qkv, rQ + 4 * G::abase(a)), q, tok_ok, c);
      vt = zero16();
      bv = w.in_b[2 * D + G::abase(a) + (dv_ok ? c.li : 0)];
      begin(IC<P::batA(a, 2)>{});
      static_for<DT>([&](auto j) { mma(IC<P::posA(a, 2, j)>{}, vt, xin[j]); });
#pragma unroll
      for (int r = 0; r < 16; ++r) k[r] += bk[r];
      tl_store<AV>(bptr(sv.qkv, rQ + 4 * (D + G::abase(a))), k, tok_ok, c);
      // independent matrix work in front of the softmax: the next tile's Q, or out_proj on the finished tiles
      if constexpr (a + 1 < AT) {
        qn = zero16();
        bqn = vec_load<AV>(w.in_b + G::abase(a + 1) + 4 * c.lh, c);
        begin(IC<P::batA(a + 1, 0)>{});
      } else {
        static_for<DT>([&](auto i) { bo[i] = vec_load<G::dvalid(i)>(w.out_b + 32 * i + 4 * c.lh, c); });
        if constexpr (!PACK && a > 0) begin(IC<P::batB(0)>{});
        else __builtin_amdgcn_sched_barrier(0);
      }
      f32x16 s[HPT];
      static_for<HPT>([&](auto hh) {
        constexpr int c0 = PACK ? (int)hh * HD : 0;        // the head's columns inside the tile: [c0, c0 + HD)
        s[hh] = zero16();
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (tcol(r) >= c0 && tcol(r) < c0 + HD) s[hh] = mfma(k[r], q[r], s[hh]);   // S^T[key][query]
      });
      {   // V (column layout): bias, store
        const int cb = 2 * D + G::abase(a);
        const float b = dv_ok ? bv : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          vt[r] += b;
          const bool ok = dv_ok && (FULL || tcol(r) + 4 * c.lh < L);
          gstore(ok ? bptr(sv.qkv, voff + 4 * cb + tcol(r) * vstride) : c.trash, vt[r]);
        }
      }
      if constexpr (a + 1 < AT) {
        static_for<DT>([&](auto j) { mma(IC<P::posA(a + 1, 0, j)>{}, qn, xin[j]); });
      } else if constexpr (!PACK) {
        static_for<a>([&](auto a2) {
          if constexpr (a2 > 0) begin(IC<P::batB(a2)>{});
          static_for<DT>([&](auto i) { mma(IC<P::posB(i, a2)>{}, oacc[i], ao[a2]); });
        });
      }
      static_for<HPT>([&](auto hh) {
        constexpr int h = PACK ? (int)hh : a;
        constexpr int c0 = PACK ? h * HD : 0;
        const float scale = 1.0f / sqrtf((float)HD);
        float p[16];
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = tcol(r) + 4 * c.lh;
          float sc = s[hh][r] * scale;
          if ((!FULL && key >= L) || !((key_bits >> key) & 1u)) sc = -INFINITY;
          p[r] = sc;
          mx = fmaxf(mx, sc);
        }
        mx = half_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          p[r] = expf(p[r] - mx);
          sum += p[r];
        }
        sum = half_sum(sum);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int r = 0; r < 16; ++r) p[r] *= inv;
        const uint32_t prow = (((uint32_t)c.n * NH + h) * L + tok) * L;     // query = this lane's token
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float m[4];
          if constexpr (FULL) drop4_even(dk, prow + 8 * g + 4 * c.lh, m);     // L == 32: the row starts on an even index
          else drop4_any(dk, prow + 8 * g + 4 * c.lh, m);
#pragma unroll
          for (int b = 0; b < 4; ++b) p[4 * g + b] *= m[b];
        }
        const bool mine = !PACK || (c.li >= c0 && c.li < c0 + HD);       // this lane's dv belongs to head h
#pragma unroll
        for (int r = 0; r < 16; ++r) ao[a] = mfma(mine ? vt[r] : 0.f, p[r], ao[a]);   // O^T[dv][query]
      });
      tl_store<AV>(bptr(sv.ao, rD + 4 * G::abase(a)), ao[a], tok_ok, c);
      if constexpr (a + 1 < AT) bq = bqn;
      TV_STAMP(0, 10 + a);
    });
    begin(IC<P::batB(PACK ? 0 : AT - 1)>{});
    static_for<DT>([&](auto i) { mma(IC<P::posB(i, PACK ? 0 : AT - 1)>{}, oacc[i], ao[PACK ? 0 : AT - 1]); });
  }

  TV_STAMP(0, 2);
  // ================= out_proj epilogue: + x, dropout1, LayerNorm1 =================
  f32x16 x1[DT];
  {
    const DropKey dk = drop_key_nb(dr.drop1);
    static_for<DT>([&](auto i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float m[4];
        drop4_even(dk, (rD >> 2) + 32 * i + 8 * g, m);
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
          const int r = 4 * g + bb;
          x1[i][r] = col_ok<G::dvalid(i)>(r, c) ? xin[i][r] + (oacc[i][r] + bo[i][r]) * m[bb] : 0.f;
        }
      }
    });
    f32x16 xh[DT];
    const float rs = layernorm_tl<G>(x1, w.n1_g, w.n1_b, c, [&](auto i, int r, float xhv, float yv) {
      xh[i][r] = xhv;
      x1[i][r] = yv;
    });
    static_for<DT>([&](auto i) {
      tl_store<G::dvalid(i)>(bptr(sv.xhat1, rD + 128 * i), xh[i], tok_ok, c);
      tl_store<G::dvalid(i)>(bptr(sv.x1, rD + 128 * i), x1[i], tok_ok, c);
    });
    gstore((tok_ok && c.lh == 0) ? sv.rstd1 + row : c.trash, rs);
  }

  TV_STAMP(0, 3);
  // ================= decoder: value-path cross attention over the length-1 memory, LayerNorm2 =================
  f32x16 x2[DT];      // input of the feed-forward block
  if constexpr (DEC) {
    // v = W_v mem + b_v: lane d < D owns one output (softmax over one key == 1)
    float vp = 0.f;
    {
      const int d = c.lane < D ? c.lane : 0;
      const float* mr = mem + (size_t)c.n * D;
      const float* wr = w.x_in_w + (size_t)d * D;
      float a = w.x_in_b[d];
#pragma unroll
      for (int k4 = 0; k4 < D / 4; ++k4) {
        const f4u wv = *reinterpret_cast<const f4u*>(wr + 4 * k4);
        const f4u mv = *reinterpret_cast<const f4u*>(mr + 4 * k4);
        a += mv[0] * wv[0];
        a += mv[1] * wv[1];
        a += mv[2] * wv[2];
        a += mv[3] * wv[3];
      }
#pragma unroll
      for (int k = D / 4 * 4; k < D; ++k) a += mr[k] * wr[k];
      vp = a;
      gstore(c.lane < D ? sv.vproj + (size_t)c.n * D + c.lane : c.trash, a);
    }
    f32x16 vb[DT];
    {
      const DropKey dk = drop_key_nb(dr.xattn);
      float hm[NH];                                       // the (sequence, head, token) mask of this lane's token
#pragma unroll
      for (int h = 0; h < NH; ++h) hm[h] = drop1_nb(dk, ((uint32_t)c.n * NH + h) * L + tok);
      static_for<DT>([&](auto i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c0 = 32 * i + tcol(r);                 // column of half 0; half 1: + 4
          const float v0 = c0 < D ? __shfl(vp, c0 < D ? c0 : 0, 64) * hm[(c0 < D ? c0 : 0) / HD] : 0.f;
          const float v1 = c0 + 4 < D ? __shfl(vp, c0 + 4 < D ? c0 + 4 : 0, 64) * hm[(c0 + 4 < D ? c0 + 4 : 0) / HD] : 0.f;
          vb[i][r] = c.lh ? v1 : v0;
        }
        tl_store<G::dvalid(i)>(bptr(sv.vb, rD + 128 * i), vb[i], tok_ok, c);
      });
    }
    {
      const DropKey dk = drop_key_nb(dr.drop2);
      f32x16 cacc[DT], bx[DT];
      static_for<DT>([&](auto i) { bx[i] = vec_load<G::dvalid(i)>(w.x_out_b + 32 * i + 4 * c.lh, c); });
      begin(IC<P::batC()>{});
      static_for<DT>([&](auto i) {
        cacc[i] = zero16();
        static_for<DT>([&](auto j) { mma(IC<P::posC(i, j)>{}, cacc[i], vb[j]); });
      });
      static_for<DT>([&](auto i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float m[4];
          drop4_even(dk, (rD >> 2) + 32 * i + 8 * g, m);
#pragma unroll
          for (int bb = 0; bb < 4; ++bb) {
            const int r = 4 * g + bb;
            x2[i][r] = col_ok<G::dvalid(i)>(r, c) ? x1[i][r] + (cacc[i][r] + bx[i][r]) * m[bb] : 0.f;
          }
        }
      });
    }
    f32x16 xh[DT];
    const float rs = layernorm_tl<G>(x2, w.n2_g, w.n2_b, c, [&](auto i, int r, float xhv, float yv) {
      xh[i][r] = xhv;
      x2[i][r] = yv;
    });
    static_for<DT>([&](auto i) {
      tl_store<G::dvalid(i)>(bptr(sv.xhat2, rD + 128 * i), xh[i], tok_ok, c);
      tl_store<G::dvalid(i)>(bptr(sv.x2, rD + 128 * i), x2[i], tok_ok, c);
    });
    gstore((tok_ok && c.lh == 0) ? sv.rstd2 + row : c.trash, rs);
  } else {
    static_for<DT>([&](auto i) { x2[i] = x1[i]; });
  }

  TV_STAMP(0, 4);
  // ================= feed-forward block: linear1 -> GELU, dropout -> linear2, interleaved =================
  f32x16 gh[FT], yacc[DT], hacc[2], b1[2], b2[DT];
  static_for<DT>([&](auto i) { yacc[i] = zero16(); });
  {
    const DropKey dk = drop_key_nb(dr.ffn);
    auto lin1 = [&](auto i_) {          // linear1 tile i -> hacc[i & 1]
      constexpr int i = i_;
      hacc[i & 1] = zero16();
      static_for<DT>([&](auto j) { mma(IC<P::posL1(i, j)>{}, hacc[i & 1], x2[j]); });
    };
    auto epi1 = [&](auto i_) {          // bias, save h1, GELU, dropout, save g -> gh[i]
      constexpr int i = i_;
      f32x16& acc = hacc[i & 1];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += b1[i & 1][r];
      tl_store<32>(bptr(sv.h1, rF + 128 * i), acc, tok_ok, c);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float m[4];
        drop4_even(dk, (rF >> 2) + 32 * i + 8 * g, m);
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) gh[i][4 * g + bb] = gelu(acc[4 * g + bb]) * m[bb];
      }
      tl_store<32>(bptr(sv.g, rF + 128 * i), gh[i], tok_ok, c);
    };
    b1[0] = vec_load<32>(w.l1_b + 4 * c.lh, c);
    begin(IC<P::batL1(0)>{});
    lin1(IC<0>{});
    b1[1] = vec_load<32>(w.l1_b + 32 + 4 * c.lh, c);
    begin(IC<P::batL1(1)>{});
    lin1(IC<1>{});
    epi1(IC<0>{});
    static_for<FT>([&](auto jj_) {
      constexpr int jj = jj_;
      if constexpr (jj + 2 < FT) b1[jj & 1] = vec_load<32>(w.l1_b + 32 * (jj + 2) + 4 * c.lh, c);
      if constexpr (jj == FT - 1)
        static_for<DT>([&](auto i) { b2[i] = vec_load<G::dvalid(i)>(w.l2_b + 32 * i + 4 * c.lh, c); });
      begin(IC<P::batG(jj)>{});
      if constexpr (jj + 2 < FT) lin1(IC<jj + 2>{});
      static_for<DT>([&](auto i) { mma(IC<P::posL2(i, jj)>{}, yacc[i], gh[jj]); });
      if constexpr (jj + 1 < FT) epi1(IC<jj + 1>{});
    });
  }
  TV_STAMP(0, 5);
  f32x16 yo[DT];
  {
    const DropKey dk = drop_key_nb(DEC ? dr.drop3 : dr.drop2);
    static_for<DT>([&](auto i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float m[4];
        drop4_even(dk, (rD >> 2) + 32 * i + 8 * g, m);
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
          const int r = 4 * g + bb;
          yo[i][r] = col_ok<G::dvalid(i)>(r, c) ? x2[i][r] + (yacc[i][r] + b2[i][r]) * m[bb] : 0.f;
        }
      }
    });
    f32x16 xh[DT];
    const float rs = layernorm_tl<G>(yo, DEC ? w.n3_g : w.n2_g, DEC ? w.n3_b : w.n2_b, c,
                                     [&](auto i, int r, float xhv, float yv) {
                                       xh[i][r] = xhv;
                                       yo[i][r] = yv;
                                     });
    static_for<DT>([&](auto i) { tl_store<G::dvalid(i)>(bptr(sv.xhatf, rD + 128 * i), xh[i], tok_ok, c); });
    gstore((tok_ok && c.lh == 0) ? sv.rstdf + row : c.trash, rs);
  }
  TV_STAMP(0, 6);
  if (!time_mean) {
    static_for<DT>([&](auto i) { tl_store<G::dvalid(i)>(bptr(y, rD + 128 * i), yo[i], tok_ok, c); });
    TV_STAMP(0, 7);
    return;
  }
  // ---- y (N, D) = mean over the L frames (the encoder's pooling) [+ the posterior heads on the pooled feature] ----
  static_for<DT>([&](auto i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int col = 32 * i + tcol(r) + 4 * c.lh;
      if (col < D) pool[tok * PP + col] = yo[i][r];
    }
  });
  __syncthreads();
  float zc = 0.f;
  if (c.lane < D) {
    float a = 0.f;
    for (int t = 0; t < L; ++t) a += pool[t * PP + c.lane];
    zc = a * (1.0f / (float)L);
    y[(size_t)c.n * D + c.lane] = zc;
  }
  if (head_w) {
    float* z = pool + 32 * PP;
    if (c.lane < D) z[c.lane] = zc;
    __syncthreads();
    for (int j = c.lane; j < HN; j += 64) {
      const float* hw = head_w + (size_t)j * D;
      float a = 0.f;
#pragma unroll 9
      for (int k = 0; k < D; ++k) a += hw[k] * z[k];
      heads[(size_t)c.n * HN + j] = a + head_b[j];
    }
  }
}


// ------------------------------------------------------------------------------------------------------------
// backward (data-gradient chain); same outputs as txt_layer_bwd_kernel: dx, dmem, the output gradient of every GEMM
// (d_f, d_h1, d_ca, d_v, d_a, d_qkv) for the weight-gradient launch, LayerNorm parameter partials lnws[n][k][2][D]
// ------------------------------------------------------------------------------------------------------------
namespace tv {

// All data-gradient GEMMs read the weights "the other way": dX = dY W, i.e. W[k][n] with the OUTPUT column contiguous
// (kc = 0).  Program order: the feed-forward block interleaved like the forward (d gelu tile i + 1 and the linear1
// data-gradient units of tile i - 1 in front of tile i's GELU' / dropout), [cross out_proj,] out_proj per attention
// tile, in_proj per (attention tile, Q | K | V).
template <typename G>
struct BwdPlan {
  static constexpr int D = G::D, DT = G::DT, FT = G::FT, AT = G::AT;
  static constexpr int NF = 2 * FT * DT, NC = G::DEC ? DT * DT : 0, NB = AT * DT, NX = AT * 3 * DT;
  static constexpr int F0 = 0, C0 = NF, B0 = C0 + NC, X0 = B0 + NB, NU = X0 + NX;
  __host__ __device__ static constexpr int posG(int i, int j) { return F0 + (i < 2 ? i * DT : 2 * DT + (i - 2) * 2 * DT) + j; }
  __host__ __device__ static constexpr int posH(int i, int jj) {      // linear1 data gradient (output tile i, hidden tile jj)
    int p = F0 + 2 * DT;
    for (int g = 0; g < jj; ++g) p += ((g + 2 < FT) ? DT : 0) + DT;
    return p + ((jj + 2 < FT) ? DT : 0) + i;
  }
  __host__ __device__ static constexpr int posC(int i, int j) { return C0 + i * DT + j; }
  __host__ __device__ static constexpr int posB(int a, int j) { return B0 + a * DT + j; }
  __host__ __device__ static constexpr int posX(int a, int part, int i) { return X0 + (a * 3 + part) * DT + i; }
  static constexpr int NBAT = 2 + FT + (G::DEC ? 1 : 0) + AT + 3 * AT;
  __host__ __device__ static constexpr int batG(int i) { return i; }                                    // i = 0, 1
  __host__ __device__ static constexpr int batF(int jj) { return 2 + jj; }
  __host__ __device__ static constexpr int batC() { return 2 + FT; }
  __host__ __device__ static constexpr int batB(int a) { return 2 + FT + (G::DEC ? 1 : 0) + a; }
  __host__ __device__ static constexpr int batX(int a, int part) { return 2 + FT + (G::DEC ? 1 : 0) + AT + 3 * a + part; }
  struct Table {
    UD u[NU];
    BD b[NBAT + 1];
    int bat_of[NU];
  };
};
template <typename G>
__host__ __device__ constexpr typename BwdPlan<G>::Table make_bwd_table() {
  using P = BwdPlan<G>;
  typename P::Table t{};
  for (int i = 0; i < G::FT; ++i)
    for (int j = 0; j < G::DT; ++j) t.u[P::posG(i, j)] = UD{WS_L2, G::FF, 32 * i, 32 * j, 32, G::dvalid(j), 0, 0};
  for (int i = 0; i < G::DT; ++i)
    for (int jj = 0; jj < G::FT; ++jj) t.u[P::posH(i, jj)] = UD{WS_L1, G::D, 32 * i, 32 * jj, G::dvalid(i), 32, 0, 0};
  t.b[P::batG(0)] = BD{P::posG(0, 0), G::DT};
  t.b[P::batG(1)] = BD{P::posG(1, 0), G::DT};
  for (int jj = 0; jj < G::FT; ++jj) {
    const bool l1 = jj + 2 < G::FT;
    t.b[P::batF(jj)] = BD{l1 ? P::posG(jj + 2, 0) : P::posH(0, jj), (l1 ? G::DT : 0) + G::DT};
  }
  if (G::DEC) {
    t.b[P::batC()] = BD{P::posC(0, 0), G::DT * G::DT};
    for (int i = 0; i < G::DT; ++i)
      for (int j = 0; j < G::DT; ++j)
        t.u[P::posC(i, j)] = UD{WS_XOUT, G::D, 32 * i, 32 * j, G::dvalid(i), G::dvalid(j), 0, 0};
  }
  for (int a = 0; a < G::AT; ++a) {
    t.b[P::batB(a)] = BD{P::posB(a, 0), G::DT};
    for (int j = 0; j < G::DT; ++j) t.u[P::posB(a, j)] = UD{WS_OUT, G::D, G::abase(a), 32 * j, G::AV, G::dvalid(j), 0, 0};
    for (int part = 0; part < 3; ++part) {
      t.b[P::batX(a, part)] = BD{P::posX(a, part, 0), G::DT};
      for (int i = 0; i < G::DT; ++i)
        t.u[P::posX(a, part, i)] = UD{WS_IN, G::D, 32 * i, part * G::D + G::abase(a), G::dvalid(i), G::AV, 0, 0};
    }
  }
  t.b[P::NBAT] = BD{P::NU, 0};
  for (int b = 0; b < P::NBAT; ++b)
    for (int k = 0; k < t.b[b].len; ++k) t.bat_of[t.b[b].start + k] = b;
  return t;
}
template <typename G>
inline constexpr typename BwdPlan<G>::Table bwd_table = make_bwd_table<G>();

// LayerNorm backward over the D columns of TL tiles: dr = rstd (g - mean(g) - xhat mean(g xhat)), g = dy gamma
template <typename G>
__device__ __forceinline__ void ln_bwd_tl(const f32x16 (&dyv)[G::DT], const f32x16 (&xh)[G::DT], const float* __restrict__ gamma,
                                          const float rstd, const Ctx& c, f32x16 (&dr)[G::DT]) {
  constexpr int D = G::D;
  f32x16 gg[G::DT];
  float s1 = 0.f, s2 = 0.f;
  static_for<G::DT>([&](auto i) {
    const f32x16 gm = vec_load<G::dvalid(i)>(gamma + 32 * i + 4 * c.lh, c);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      gg[i][r] = dyv[i][r] * gm[r];
      s1 += gg[i][r];
      s2 += gg[i][r] * xh[i][r];
    }
  });
  s1 = half_sum(s1);
  s2 = half_sum(s2);
  const float m1 = s1 / (float)D, m2 = s2 / (float)D;
  static_for<G::DT>([&](auto i) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
      dr[i][r] = col_ok<G::dvalid(i)>(r, c) ? rstd * (gg[i][r] - m1 - xh[i][r] * m2) : 0.f;
  });
}

// 32 x 32 register tile (lane (i, half), register r <-> column tcol(r) + 4 half) -> its transpose, through a
// [32][33] LDS scratch of this wave
__device__ __forceinline__ f32x16 tile_transpose(float* __restrict__ tp, const f32x16& v, const Ctx& c) {
#pragma unroll
  for (int r = 0; r < 16; ++r) tp[c.li * 33 + tcol(r) + 4 * c.lh] = v[r];
  __syncthreads();
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = tp[(tcol(r) + 4 * c.lh) * 33 + c.li];
  __syncthreads();
  return o;
}

}  // namespace tv

template <typename G, bool FULL>
__global__ __launch_bounds__(64) void txt_wave_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ valid,
                                                          float* __restrict__ dx, float* __restrict__ dmem,
                                                          const mmvae_txt_layer_w_t w, const mmvae_txt_layer_saved_t sv,
                                                          const mmvae_txt_layer_grads_t gr, const mmvae_txt_layer_drop_t dr_,
                                                          const int L, const int N, const int time_mean,
                                                          float* __restrict__ trash) {
  MMVAE_TRACE_STAMP(18 + (G::DEC ? 1 : 0));
  using P = BwdPlan<G>;
  constexpr int D = G::D, FF = G::FF, NH = G::NH, HD = G::HD, DT = G::DT, FT = G::FT, AT = G::AT, AV = G::AV;
  constexpr bool DEC = G::DEC, PACK = G::PACK;
  constexpr int HPT = PACK ? NH : 1, NLN = DEC ? 3 : 2;
  constexpr int PP = D | 1;
  __shared__ float cs[32 * PP];          // column sums over the tokens
  __shared__ float tp[32 * 33];          // tile transposes
  __shared__ float vec[64];              // decoder: d v broadcast
  Ctx c;
  c.lane = threadIdx.x;
  c.li = c.lane & 31;
  c.lh = c.lane >> 5;
  c.n = blockIdx.x;
  c.N = N;
  c.L = L;
  c.trash = trash + 4 * c.lane;
  c.zero = reinterpret_cast<const float*>(tv_zero);
  const int tok = c.li;
  const bool tok_ok = FULL || tok < L;
  const uint32_t row = (uint32_t)(tok_ok ? tok : 0) * N + c.n;
  const uint32_t rD = row * (D * 4) + 16 * c.lh, rF = row * (FF * 4) + 16 * c.lh, rQ = row * (3 * D * 4) + 16 * c.lh;

  float wb[2][4][16];
  auto batch_load = [&](auto b_) {
    constexpr int b = b_;
    if constexpr (b < P::NBAT) {
      constexpr BD bd = bwd_table<G>.b[b];
      static_for<bd.len>([&](auto k) {
        constexpr UD d = bwd_table<G>.u[bd.start + k];
        const float* W = d.ws == WS_IN ? w.in_w : d.ws == WS_OUT ? w.out_w : d.ws == WS_XOUT ? w.x_out_w
                       : d.ws == WS_L1 ? w.l1_w : w.l2_w;
        wload<false, d.nvalid, d.kvalid>(wb[b & 1][k], W, d.ldw, d.nbase, d.kbase, c);
      });
    }
  };
  auto begin = [&](auto b_) {
    constexpr int b = b_;
    batch_load(IC<b + 1>{});
    __builtin_amdgcn_sched_barrier(0);
  };
  auto mma = [&](auto u_, f32x16& acc, const f32x16& xop) {
    constexpr int u = u_;
    constexpr UD d = bwd_table<G>.u[u];
    constexpr int b = bwd_table<G>.bat_of[u];
    mma_unit<d.kvalid, false>(acc, wb[b & 1][u - bwd_table<G>.b[b].start], xop);
  };
  // column sums over the tokens of a TL activation: lane col < D returns sum_t v[t][col]
  auto col_sum = [&](const f32x16 (&v)[DT]) {
    static_for<DT>([&](auto i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int col = 32 * i + tcol(r) + 4 * c.lh;
        if (col < D) cs[tok * PP + col] = v[i][r];
      }
    });
    __syncthreads();
    float a = 0.f;
    const int col = c.lane < D ? c.lane : 0;
#pragma unroll 8
    for (int t = 0; t < 32; ++t) a += cs[t * PP + col];
    __syncthreads();
    return a;
  };
  // LayerNorm weight / bias gradient partials of this sequence: lnws[n][k][0] = sum_t dy xhat, [1] = sum_t dy
  auto ln_partials = [&](const f32x16 (&dyv)[DT], const f32x16 (&xh)[DT], const int k) {
    f32x16 pr[DT];
    static_for<DT>([&](auto i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) pr[i][r] = dyv[i][r] * xh[i][r];
    });
    const float sg = col_sum(pr), sb = col_sum(dyv);
    float* o = gr.lnws + (((size_t)c.n * NLN + k) * 2) * D;
    gstore(c.lane < D ? o + c.lane : c.trash, sg);
    gstore(c.lane < D ? o + D + c.lane : c.trash, sb);
  };
  TV_STAMP(1, 0);
  batch_load(IC<0>{});

  // ---- incoming gradient (rows t >= L are zero) and the last LayerNorm ----
  f32x16 g[DT], xh[DT], dr[DT];
  static_for<DT>([&](auto i) {
    if (time_mean) {   // dy (N, D) is the gradient of the mean over frames: every row gets dy / L
      const f32x16 v = vec_load<G::dvalid(i)>(dy + (size_t)c.n * D + 32 * i + 4 * c.lh, c);
      const float sc = tok_ok ? 1.0f / (float)L : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) g[i][r] = v[r] * sc;
    } else {
      g[i] = tl_load<G::dvalid(i)>(bptr(dy, rD + 128 * i), dy, tok_ok, c);
    }
    xh[i] = tl_load<G::dvalid(i)>(bptr(sv.xhatf, rD + 128 * i), sv.xhatf, tok_ok, c);
  });
  ln_partials(g, xh, NLN - 1);
  ln_bwd_tl<G>(g, xh, DEC ? w.n3_g : w.n2_g, gload(tok_ok ? sv.rstdf + row : c.zero), c, dr);

  TV_STAMP(1, 1);
  // ================= feed-forward block =================
  f32x16 dyn[DT];        // gradient of the FFN block's input (LayerNorm2 output for the decoder, LayerNorm1 for the encoder)
  {
    f32x16 df[DT];
    {
      const DropKey dk = drop_key_nb(DEC ? dr_.drop3 : dr_.drop2);
      static_for<DT>([&](auto i) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          float m[4];
          drop4_even(dk, (rD >> 2) + 32 * i + 8 * gq, m);
#pragma unroll
          for (int bb = 0; bb < 4; ++bb) df[i][4 * gq + bb] = dr[i][4 * gq + bb] * m[bb];
        }
        tl_store<G::dvalid(i)>(bptr(gr.d_f, rD + 128 * i), df[i], tok_ok, c);      // gradient of linear2's output
      });
    }
    const DropKey dk = drop_key_nb(dr_.ffn);
    f32x16 dh[FT], xacc[DT], gacc[2], hv[2];
    static_for<DT>([&](auto i) { xacc[i] = zero16(); });
    auto ling = [&](auto i_) {          // d(gelu output) tile i = df W2[:, tile i] -> gacc[i & 1]
      constexpr int i = i_;
      gacc[i & 1] = zero16();
      static_for<DT>([&](auto j) { mma(IC<P::posG(i, j)>{}, gacc[i & 1], df[j]); });
    };
    auto epig = [&](auto i_) {          // through the dropout and GELU': d h1 tile i
      constexpr int i = i_;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        float m[4];
        drop4_even(dk, (rF >> 2) + 32 * i + 8 * gq, m);
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
          const int r = 4 * gq + bb;
          dh[i][r] = tok_ok ? gacc[i & 1][r] * m[bb] * gelu_grad(hv[i & 1][r]) : 0.f;
        }
      }
      tl_store<32>(bptr(gr.d_h1, rF + 128 * i), dh[i], tok_ok, c);                 // gradient of linear1's output
    };
    hv[0] = tl_load<32>(bptr(sv.h1, rF), sv.h1, tok_ok, c);
    begin(IC<P::batG(0)>{});
    ling(IC<0>{});
    hv[1] = tl_load<32>(bptr(sv.h1, rF + 128), sv.h1, tok_ok, c);
    begin(IC<P::batG(1)>{});
    ling(IC<1>{});
    epig(IC<0>{});
    static_for<FT>([&](auto jj_) {
      constexpr int jj = jj_;
      if constexpr (jj + 2 < FT) hv[jj & 1] = tl_load<32>(bptr(sv.h1, rF + 128 * (jj + 2)), sv.h1, tok_ok, c);
      begin(IC<P::batF(jj)>{});
      if constexpr (jj + 2 < FT) ling(IC<jj + 2>{});
      static_for<DT>([&](auto i) { mma(IC<P::posH(i, jj)>{}, xacc[i], dh[jj]); });
      if constexpr (jj + 1 < FT) epig(IC<jj + 1>{});
    });
    static_for<DT>([&](auto i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) dyn[i][r] = dr[i][r] + xacc[i][r];
    });
  }

  TV_STAMP(1, 2);
  if constexpr (DEC) {
    // ================= cross-attention block (value path over the length-1 memory) =================
    static_for<DT>([&](auto i) { xh[i] = tl_load<G::dvalid(i)>(bptr(sv.xhat2, rD + 128 * i), sv.xhat2, tok_ok, c); });
    ln_partials(dyn, xh, 1);
    ln_bwd_tl<G>(dyn, xh, w.n2_g, gload(tok_ok ? sv.rstd2 + row : c.zero), c, dr);
    f32x16 dca[DT], dvb[DT];
    {
      const DropKey dk = drop_key_nb(dr_.drop2);
      static_for<DT>([&](auto i) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          float m[4];
          drop4_even(dk, (rD >> 2) + 32 * i + 8 * gq, m);
#pragma unroll
          for (int bb = 0; bb < 4; ++bb) dca[i][4 * gq + bb] = dr[i][4 * gq + bb] * m[bb];
        }
        tl_store<G::dvalid(i)>(bptr(gr.d_ca, rD + 128 * i), dca[i], tok_ok, c);    // gradient of the cross out_proj's output
      });
    }
    begin(IC<P::batC()>{});
    static_for<DT>([&](auto i) {
      dvb[i] = zero16();
      static_for<DT>([&](auto j) { mma(IC<P::posC(i, j)>{}, dvb[i], dca[j]); });
    });
    {   // d v[c] = sum_t mask(t, h(c)) d vb[t][c]
      const DropKey dk = drop_key_nb(dr_.xattn);
      float hm[NH];
#pragma unroll
      for (int h = 0; h < NH; ++h) hm[h] = tok_ok ? drop1_nb(dk, ((uint32_t)c.n * NH + h) * L + tok) : 0.f;
      static_for<DT>([&](auto i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c0 = 32 * i + tcol(r);
          const float m0 = hm[(c0 < D ? c0 : 0) / HD], m1 = hm[(c0 + 4 < D ? c0 + 4 : 0) / HD];
          dvb[i][r] *= c.lh ? m1 : m0;
        }
      });
      const float dv = col_sum(dvb);
      gstore(c.lane < D ? gr.d_v + (size_t)c.n * D + c.lane : c.trash, dv);      // gradient of the value projection (N, D)
      if (c.lane < D) vec[c.lane] = dv;
      __syncthreads();
      float a = 0.f;                                                           // d mem = d v W_v
      const int k = c.lane < D ? c.lane : 0;
#pragma unroll 8
      for (int cc = 0; cc < D; ++cc) a += vec[cc] * w.x_in_w[(size_t)cc * D + k];
      gstore(c.lane < D ? dmem + (size_t)c.n * D + c.lane : c.trash, a);
      __syncthreads();
    }
    static_for<DT>([&](auto i) { dyn[i] = dr[i]; });      // the residual path: gradient of LayerNorm1's output
  }

  TV_STAMP(1, 3);
  // ================= self-attention block =================
  static_for<DT>([&](auto i) { xh[i] = tl_load<G::dvalid(i)>(bptr(sv.xhat1, rD + 128 * i), sv.xhat1, tok_ok, c); });
  ln_partials(dyn, xh, 0);
  ln_bwd_tl<G>(dyn, xh, w.n1_g, gload(tok_ok ? sv.rstd1 + row : c.zero), c, dr);
  f32x16 da[DT];
  {
    const DropKey dk = drop_key_nb(dr_.drop1);
    static_for<DT>([&](auto i) {
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        float m[4];
        drop4_even(dk, (rD >> 2) + 32 * i + 8 * gq, m);
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) da[i][4 * gq + bb] = dr[i][4 * gq + bb] * m[bb];
      }
      tl_store<G::dvalid(i)>(bptr(gr.d_a, rD + 128 * i), da[i], tok_ok, c);        // gradient of out_proj's output
    });
  }
  TV_STAMP(1, 4);
  const unsigned long long vb64 = __ballot(tok_ok && valid[(size_t)c.n * L + (tok_ok ? tok : 0)] != 0);
  const uint32_t key_bits = (uint32_t)vb64;
  f32x16 dao[AT], xacc[DT];
  static_for<DT>([&](auto i) { xacc[i] = zero16(); });
  static_for<AT>([&](auto a) {
    dao[a] = zero16();
    begin(IC<P::batB(a)>{});
    static_for<DT>([&](auto j) { mma(IC<P::posB(a, j)>{}, dao[a], da[j]); });      // d(attention output), tile a
  });
  TV_STAMP(1, 5);
  {
    const DropKey dk = drop_key_nb(dr_.attn);
    const uint32_t vstride = (uint32_t)N * (3 * D * 4);
    const uint32_t voff = ((uint32_t)(4 * c.lh) * N + c.n) * (3 * D * 4) + 4 * c.li;   // column layout: token 4 half, column li
    const bool dl_ok = c.li < AV;
    static_for<AT>([&](auto a_) {
      constexpr int a = a_;
      // operands of this attention tile: Q, K, V in token layout now; Q, K in column layout (lane = dd, registers = tokens)
      // and d O in column layout only when the products that consume them start (register pressure)
      const f32x16 qt = tl_load<AV>(bptr(sv.qkv, rQ + 4 * G::abase(a)), sv.qkv, tok_ok, c);
      const f32x16 kt = tl_load<AV>(bptr(sv.qkv, rQ + 4 * (D + G::abase(a))), sv.qkv, tok_ok, c);
      const f32x16 vt = tl_load<AV>(bptr(sv.qkv, rQ + 4 * (2 * D + G::abase(a))), sv.qkv, tok_ok, c);
      f32x16 ds[HPT], pd[HPT];
      static_for<HPT>([&](auto hh) {
        constexpr int h = PACK ? (int)hh : a;
        constexpr int c0 = PACK ? h * HD : 0;
        f32x16 s = zero16(), dp = zero16();
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (tcol(r) >= c0 && tcol(r) < c0 + HD) {
            s = mfma(kt[r], qt[r], s);                 // S^T[key][query]
            dp = mfma(vt[r], dao[a][r], dp);           // d P^T[key][query] = sum_dv V[key][dv] dO[query][dv]
          }
        const float scale = 1.0f / sqrtf((float)HD);
        float p[16], mk[16];
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = tcol(r) + 4 * c.lh;
          float sc = s[r] * scale;
          if ((!FULL && key >= L) || !((key_bits >> key) & 1u)) sc = -INFINITY;
          p[r] = sc;
          mx = fmaxf(mx, sc);
        }
        mx = half_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          p[r] = expf(p[r] - mx);
          sum += p[r];
        }
        sum = half_sum(sum);
        const float inv = 1.0f / sum;
        const uint32_t prow = (((uint32_t)c.n * NH + h) * L + tok) * L;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          float m[4];
          if constexpr (FULL) drop4_even(dk, prow + 8 * gq + 4 * c.lh, m);
          else drop4_any(dk, prow + 8 * gq + 4 * c.lh, m);
#pragma unroll
          for (int bb = 0; bb < 4; ++bb) mk[4 * gq + bb] = m[bb];
        }
        float dot = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          p[r] *= inv;
          dp[r] *= mk[r];                              // through the dropout
          dot += dp[r] * p[r];
        }
        dot = half_sum(dot);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = tcol(r) + 4 * c.lh;
          float v = p[r] * (dp[r] - dot) * scale;
          if (!(v == v) || (!FULL && (key >= L || tok >= L))) v = 0.f;   // masked / padded entries (and no NaN from 0 * inf)
          ds[hh][r] = v;
          pd[hh][r] = (!FULL && tok >= L) ? 0.f : p[r] * mk[r];
        }
      });
      __builtin_amdgcn_sched_barrier(0);        // (Q, K, V in token layout are dead from here on)
      f32x16 dq = zero16(), dkk = zero16(), dvv = zero16();
      {
        f32x16 kd;                               // K in column layout
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool ok = dl_ok && (FULL || tcol(r) + 4 * c.lh < L);
          kd[r] = gload(ok ? bptr(sv.qkv, voff + 4 * (D + G::abase(a)) + tcol(r) * vstride) : c.zero);
        }
        static_for<HPT>([&](auto hh) {
          constexpr int c0 = PACK ? (int)hh * HD : 0;
          const bool mine = !PACK || (c.li >= c0 && c.li < c0 + HD);
#pragma unroll
          for (int r = 0; r < 16; ++r) dq = mfma(mine ? kd[r] : 0.f, ds[hh][r], dq);    // d Q^T[dd][query]
        });
      }
      {
        f32x16 qd;                               // Q in column layout
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool ok = dl_ok && (FULL || tcol(r) + 4 * c.lh < L);
          qd[r] = gload(ok ? bptr(sv.qkv, voff + 4 * G::abase(a) + tcol(r) * vstride) : c.zero);
        }
        static_for<HPT>([&](auto hh) {
          constexpr int c0 = PACK ? (int)hh * HD : 0;
          const bool mine = !PACK || (c.li >= c0 && c.li < c0 + HD);
          const f32x16 dsT = tile_transpose(tp, ds[hh], c);                            // lane = key, registers = queries
#pragma unroll
          for (int r = 0; r < 16; ++r) dkk = mfma(mine ? qd[r] : 0.f, dsT[r], dkk);    // d K^T[dd][key]
        });
      }
      {
        const f32x16 dod = tile_transpose(tp, dao[a], c);     // d O in column layout (lane = dv, registers = queries)
        static_for<HPT>([&](auto hh) {
          constexpr int c0 = PACK ? (int)hh * HD : 0;
          const bool mine = !PACK || (c.li >= c0 && c.li < c0 + HD);
          const f32x16 pdT = tile_transpose(tp, pd[hh], c);
#pragma unroll
          for (int r = 0; r < 16; ++r) dvv = mfma(mine ? dod[r] : 0.f, pdT[r], dvv);   // d V^T[dv][key]
        });
      }
      tl_store<AV>(bptr(gr.d_qkv, rQ + 4 * G::abase(a)), dq, tok_ok, c);
      tl_store<AV>(bptr(gr.d_qkv, rQ + 4 * (D + G::abase(a))), dkk, tok_ok, c);
      tl_store<AV>(bptr(gr.d_qkv, rQ + 4 * (2 * D + G::abase(a))), dvv, tok_ok, c);
      // d x += d qkv W_in, this tile's rows of the in_proj
      TV_STAMP(1, 10 + 2 * a);
      begin(IC<P::batX(a, 0)>{});
      static_for<DT>([&](auto i) { mma(IC<P::posX(a, 0, i)>{}, xacc[i], dq); });
      begin(IC<P::batX(a, 1)>{});
      static_for<DT>([&](auto i) { mma(IC<P::posX(a, 1, i)>{}, xacc[i], dkk); });
      begin(IC<P::batX(a, 2)>{});
      static_for<DT>([&](auto i) { mma(IC<P::posX(a, 2, i)>{}, xacc[i], dvv); });
      TV_STAMP(1, 11 + 2 * a);
    });
  }
  static_for<DT>([&](auto i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) xacc[i][r] += dr[i][r];
    tl_store<G::dvalid(i)>(bptr(dx, rD + 128 * i), xacc[i], tok_ok, c);
  });
  TV_STAMP(1, 7);
}


#ifdef TV_PROBE
extern "C" int mmvae_txt_wave_stamps(long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(tv::tv_stamps), sizeof(long long) * 64) == hipSuccess ? 0 : 1;
}
#endif

// masked stores land here (device memory nobody reads; one 16-byte slot per lane, shared by all waves)
__device__ float tv_trash_buf[64 * 4];
// (a __device__ symbol has one address PER DEVICE: cached by device ordinal, never across devices -- ADVICE r4)
static float* tv_trash_ptr() {
  static float* p[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (!p[dev] && hipGetSymbolAddress(reinterpret_cast<void**>(&p[dev]), HIP_SYMBOL(tv_trash_buf)) != hipSuccess)
    p[dev] = nullptr;
  return p[dev];
}

// ------------------------------------------------------------------------------------------------------------
template <typename F>
static inline bool txt_wave_visit(int D, int FF, int NH, int dec, F&& f) {
  if (NH != 2 || FF != 128) return false;
  if (D == 54 && !dec) { f(tv::Geom<54, 128, 2, false>{}); return true; }
  if (D == 32 && dec) { f(tv::Geom<32, 128, 2, true>{}); return true; }
  if (D == 32 && !dec) { f(tv::Geom<32, 128, 2, false>{}); return true; }
  if (D == 54 && dec) { f(tv::Geom<54, 128, 2, true>{}); return true; }
  if (D == 16 && dec) { f(tv::Geom<16, 128, 2, true>{}); return true; }
  if (D == 24 && dec) { f(tv::Geom<24, 128, 2, true>{}); return true; }     // the shipped config_cdspritesplus.yml: n_latents 24
  return false;
}

int txt_wave_fwd_dispatch(const float* x, const uint8_t* valid, const float* mem, float* y, const mmvae_txt_layer_w_t& wv,
                          const mmvae_txt_layer_saved_t& sv, const mmvae_txt_layer_drop_t& d, int L, int N, int D, int FF,
                          int NH, int dec, int time_mean, const float* head_w, const float* head_b, float* heads, int HN,
                          hipStream_t stream) {
  if (!tv_trash_ptr()) return MMVAE_ERR_LAUNCH;
  if (!txt_wave_visit(D, FF, NH, dec, [&](auto g) {
        using G = decltype(g);
        if (L == 32)
          hipLaunchKernelGGL((txt_wave_fwd_kernel<G, true>), dim3(N), dim3(64), 0, stream, x, valid, mem, y, wv, sv, d, L,
                             N, time_mean, head_w, head_b, heads, HN, tv_trash_ptr());
        else
          hipLaunchKernelGGL((txt_wave_fwd_kernel<G, false>), dim3(N), dim3(64), 0, stream, x, valid, mem, y, wv, sv, d, L,
                             N, time_mean, head_w, head_b, heads, HN, tv_trash_ptr());
      }))
    return MMVAE_ERR_UNSUPPORTED;
  return mmvae_launch_status();
}

int txt_wave_bwd_dispatch(const float* dy, const uint8_t* valid, float* dx, float* dmem, const mmvae_txt_layer_w_t& wv,
                          const mmvae_txt_layer_saved_t& sv, const mmvae_txt_layer_grads_t& gv,
                          const mmvae_txt_layer_drop_t& d, int L, int N, int D, int FF, int NH, int dec, int time_mean,
                          hipStream_t stream) {
  if (!tv_trash_ptr()) return MMVAE_ERR_LAUNCH;
  if (!txt_wave_visit(D, FF, NH, dec, [&](auto g) {
        using G = decltype(g);
        if (L == 32)
          hipLaunchKernelGGL((txt_wave_bwd_kernel<G, true>), dim3(N), dim3(64), 0, stream, dy, valid, dx, dmem, wv, sv, gv, d,
                             L, N, time_mean, tv_trash_ptr());
        else
          hipLaunchKernelGGL((txt_wave_bwd_kernel<G, false>), dim3(N), dim3(64), 0, stream, dy, valid, dx, dmem, wv, sv, gv, d,
                             L, N, time_mean, tv_trash_ptr());
      }))
    return MMVAE_ERR_UNSUPPORTED;
  return mmvae_launch_status();
}

MMVAE_TRACE_SETTER(txtwave)
