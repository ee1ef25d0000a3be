// Shared device helpers for the gfx950 kernels.  Wavefront = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mmvae_hip.h"

#define MMVAE_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MMVAE_CHECK_ARG(cond) \
  do {                        \
    if (!(cond)) return MMVAE_ERR_ARG; \
  } while (0)

static inline int mmvae_launch_status() { return hipGetLastError() == hipSuccess ? MMVAE_OK : MMVAE_ERR_LAUNCH; }

// v_exp_f32 + v_rcp_f32 (1 ulp each); an IEEE division here costs ~10 more VALU instructions per element
__device__ __forceinline__ float dev_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float dev_silu(float x) { return x * dev_sigmoid(x); }
__device__ __forceinline__ float dev_silu_grad(float x) {
  float s = dev_sigmoid(x);
  return s * (1.0f + x * (1.0f - s));
}
// exact (erf) GELU, torch.nn.functional.gelu default
__device__ __forceinline__ float dev_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float dev_gelu_grad(float x) {
  float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
  float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

__device__ __forceinline__ float apply_in_act(float v, int act) {
  switch (act) {
    case MMVAE_ACT_SILU: return dev_silu(v);
    case MMVAE_ACT_RELU: return fmaxf(v, 0.0f);
    case MMVAE_ACT_GELU: return dev_gelu(v);
    default: return v;
  }
}

// epilogue: v = acc (+bias already added by the caller); aux_v = value of aux at the same location (or 0)
__device__ __forceinline__ float apply_epilogue(float v, float aux_v, int ep) {
  switch (ep) {
    case MMVAE_EP_RELU: return fmaxf(v, 0.0f);
    case MMVAE_EP_MUL_RELU_MASK: return aux_v > 0.0f ? v : 0.0f;
    case MMVAE_EP_MUL_SILU_GRAD: return v * dev_silu_grad(aux_v);
    case MMVAE_EP_GELU: return dev_gelu(v);
    case MMVAE_EP_MUL_GELU_GRAD: return v * dev_gelu_grad(aux_v);
    case MMVAE_EP_SIGMOID_CLAMP: return fminf(fmaxf(dev_sigmoid(v), 1e-6f), 1.0f - 1e-6f);
    case MMVAE_EP_SIGMOID: return dev_sigmoid(v);
    case MMVAE_EP_ADD_AUX: return v + aux_v;
    default: return v;
  }
}
__host__ __device__ __forceinline__ bool ep_reads_aux(int ep) {
  return ep == MMVAE_EP_MUL_RELU_MASK || ep == MMVAE_EP_MUL_SILU_GRAD || ep == MMVAE_EP_MUL_GELU_GRAD || ep == MMVAE_EP_ADD_AUX;
}

// ---- counter-based dropout -----------------------------------------------------------------------------
__device__ __forceinline__ uint32_t drop_fmix(uint32_t h) {  // murmur3 finaliser
  h ^= h >> 16;
  h *= 0x85ebca6bu;
  h ^= h >> 13;
  h *= 0xc2b2ae35u;
  h ^= h >> 16;
  return h;
}
struct DropKey {
  uint32_t key, thr;
  float p, inv_keep;
  bool on;
};
__device__ __forceinline__ DropKey drop_key(const mmvae_dropout_t& d) {
  DropKey k;
  k.on = d.state != nullptr && d.p > 0.f;
  k.p = d.p;
  k.inv_keep = k.on ? 1.0f / (1.0f - d.p) : 1.0f;
  k.thr = (uint32_t)(d.p * 65536.0f + 0.5f);      // p in units of 2^-16 (|p - thr / 65536| <= 7.7e-6)
  k.key = k.on ? drop_fmix(d.state[0] ^ (d.state[2 + d.slot] * 0x9E3779B1u) ^ (d.site * 0x85EBCA77u + 0x165667B1u)) : 0u;
  return k;
}
// Round 3: ONE hash serves TWO consecutive elements -- element idx takes the 16-bit half (idx & 1) of
// fmix(key + (idx >> 1) C) and is kept when that half >= p 2^16.  The murmur finaliser costs two quarter-rate integer
// multiplies (+ one for the index): kernels whose lanes own runs of consecutive elements (the fused feed-forward block:
// 16 hidden elements per lane and tile between its MFMAs) now pay it once per pair (drop_pair_hash / _lo / _hi);
// everything else keeps calling drop_mul per element and gets the same mask.
__device__ __forceinline__ uint32_t drop_pair_hash(const DropKey& k, uint32_t pair) {
  return drop_fmix(k.key + pair * 0x9E3779B1u);
}
__device__ __forceinline__ float drop_pair_lo(const DropKey& k, uint32_t h) {
  return (!k.on || (h & 0xFFFFu) >= k.thr) ? k.inv_keep : 0.0f;
}
__device__ __forceinline__ float drop_pair_hi(const DropKey& k, uint32_t h) {
  return (!k.on || (h >> 16) >= k.thr) ? k.inv_keep : 0.0f;
}
// multiplicative mask of element idx: 0 or 1/(1-p)
__device__ __forceinline__ float drop_mul(const DropKey& k, uint32_t idx) {
  if (!k.on) return 1.0f;
  const uint32_t h = drop_pair_hash(k, idx >> 1);
  return ((idx & 1u) ? (h >> 16) : (h & 0xFFFFu)) >= k.thr ? k.inv_keep : 0.0f;
}
static inline mmvae_dropout_t drop_arg(const mmvae_dropout_t* d) {
  mmvae_dropout_t z = {nullptr, 0u, 0u, 0.f};
  return d ? *d : z;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// sum over a 256-thread block; result valid in every thread.  `red` = >= 4 floats of LDS.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// ---- debug build only (-DMMVAE_TRACE, tools/probe/trace_step.py): thread 0 of workgroup 0 of the instrumented kernels
// stores the device wall clock into a per-module table, so the start time of every phase of a REAL captured step can
// be read back without adding graph nodes (markers change the graph's topology and with it hipGraph's scheduling) or
// a profiler (which changes the timing).  Compiles to nothing in the product build.
#ifdef MMVAE_TRACE
// table[0] = event counter; event k = (id, wall clock) at table[8 + 2 k], k modulo 2048: EVERY launch of an instrumented
// kernel leaves one, so repeated kernels (the grouped GEMMs, the two text layers) show up once per launch, in time order
static __device__ long long* mmvae_trace_table __attribute__((unused)) = nullptr;
#define MMVAE_TRACE_STAMP(id)                                                      \
  do {                                                                             \
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) { \
      long long* t__ = mmvae_trace_table;                                          \
      if (t__) {                                                                   \
        const unsigned long long k__ = atomicAdd(reinterpret_cast<unsigned long long*>(t__), 1ull) & 2047ull; \
        t__[8 + 2 * k__] = (long long)(id);                                        \
        t__[9 + 2 * k__] = (long long)wall_clock64();                              \
      }                                                                            \
    }                                                                              \
  } while (0)
#define MMVAE_TRACE_SETTER(module)                                                          \
  extern "C" int mmvae_trace_set_##module(long long* table) {                               \
    return hipMemcpyToSymbol(HIP_SYMBOL(mmvae_trace_table), &table, sizeof(table)) == hipSuccess ? 0 : 1; \
  }
#else
#define MMVAE_TRACE_STAMP(id)
#define MMVAE_TRACE_SETTER(module)
#endif
