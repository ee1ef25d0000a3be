// Micro-probe for VERDICT r5 item 3 (text towers on split-bf16 MFMA): the GEMM skeleton of the wave-per-sequence text layer
// (csrc/txtwave.hip) -- one wave per sequence, activations chained through the accumulator registers, a weight unit of 32
// outputs x 32 reduction steps per GEMM step, weights streamed from L2 one unit ahead -- once on v_mfma_f32_32x32x2_f32 (16
// MFMAs per unit, weights as fp32 rows) and once on split-bf16 (activation tile split in registers once per produced tile,
// pre-split permuted three-plane weight images, 2 k-steps x 6 v_mfma_f32_32x32x16_bf16 per unit).  U units per launch, a new
// activation tile every second unit (the layer's D = 54 has two tiles per activation).  No softmax / GELU / LayerNorm / dropout:
// this is the part of the layer the matrix core choice changes.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libtxtunit.so txt_unit_b16.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ void split3(float x, float& x0, float& x1, float& x2) {
  x0 = __uint_as_float(__float_as_uint(x) & 0xFFFF0000u);
  const float r1 = x - x0;
  x1 = __uint_as_float(__float_as_uint(r1) & 0xFFFF0000u);
  x2 = r1 - x1;
}
__device__ __forceinline__ unsigned pack(float lo, float hi) {
  return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}
__device__ __forceinline__ void split8(const float* f, u32x4 (&p)[3]) {
  float a[3][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) split3(f[j], a[0][j], a[1][j], a[2][j]);
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int d = 0; d < 4; ++d) p[t][d] = pack(a[t][2 * d], a[t][2 * d + 1]);
}

template <int U>
__global__ __launch_bounds__(64) void unit_chain_f32(const float* __restrict__ W, float* __restrict__ out) {
  const int lane = threadIdx.x, li = lane & 31, lh = lane >> 5;
  f32x16 x, acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) { x[r] = 0.001f * (float)(lane + r + blockIdx.x % 7); acc[r] = 0.f; }
  float w[2][16];
  auto wload = [&](float (&d)[16], int u) {
    const float* p = W + ((size_t)u * 32 + li) * 32 + 4 * lh;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f4u v = *reinterpret_cast<const f4u*>(p + 8 * g);
#pragma unroll
      for (int b = 0; b < 4; ++b) d[4 * g + b] = v[b];
    }
  };
  wload(w[0], 0);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (u + 1 < U) wload(w[(u + 1) & 1], u + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[u & 1][r], x[r], acc, 0, 0, 0);
    if (u & 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { x[r] = acc[r] * 0.05f; acc[r] = 0.f; }
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) out[((size_t)blockIdx.x * 16 + r) * 64 + lane] = x[r] + acc[r];
}

template <int U>
__global__ __launch_bounds__(64) void unit_chain_b16(const unsigned short* __restrict__ Wimg, float* __restrict__ out) {
  const int lane = threadIdx.x, li = lane & 31, lh = lane >> 5;
  f32x16 x, acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) { x[r] = 0.001f * (float)(lane + r + blockIdx.x % 7); acc[r] = 0.f; }
  u32x4 xb[2][3], w[2][2][3];
  auto xsplit = [&]() {
    float f[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) f[r] = x[r];
    split8(&f[0], xb[0]);
    split8(&f[8], xb[1]);
  };
  auto wload = [&](u32x4 (&d)[2][3], int u) {      // unit image: [3 planes][32 rows][32 k permuted]
    const unsigned short* p = Wimg + (size_t)u * 3 * 1024 + li * 32 + 8 * lh;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int t = 0; t < 3; ++t) d[m][t] = *reinterpret_cast<const u32x4*>(p + t * 1024 + 16 * m);
  };
  xsplit();
  wload(w[0], 0);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (u + 1 < U) wload(w[(u + 1) & 1], u + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#define MM(ta, tb) \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w[u & 1][m][ta]), __builtin_bit_cast(bf16x8, xb[m][tb]), acc, 0, 0, 0)
      MM(0, 2); MM(2, 0); MM(1, 1); MM(0, 1); MM(1, 0); MM(0, 0);
#undef MM
    }
    if (u & 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { x[r] = acc[r] * 0.05f; acc[r] = 0.f; }
      xsplit();
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) out[((size_t)blockIdx.x * 16 + r) * 64 + lane] = x[r] + acc[r];
}

extern "C" int probe_f32(const float* W, float* out, int N, void* stream) {
  hipLaunchKernelGGL(unit_chain_f32<36>, dim3(N), dim3(64), 0, (hipStream_t)stream, W, out);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
extern "C" int probe_b16(const unsigned short* Wimg, float* out, int N, void* stream) {
  hipLaunchKernelGGL(unit_chain_b16<36>, dim3(N), dim3(64), 0, (hipStream_t)stream, Wimg, out);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
