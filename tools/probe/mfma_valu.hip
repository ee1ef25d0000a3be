// Probe: does independent fp32 VALU work hide behind a dependent chain of v_mfma_f32_32x32x2_f32 (and, for contrast,
// v_mfma_f32_32x32x16_bf16) in ONE wave?  K independent v_fma_f32 are placed between consecutive MFMAs.
// hipcc --offload-arch=gfx950 -O3 -shared -fPIC mfma_valu.hip -o mfma_valu.so
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
template <int K, bool BF16>
__global__ void mfma_valu_kernel(float* out, long long* ticks, int iters) {
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float a = threadIdx.x * 0.001f, b = 1.0f;
  bf16x8 ab, bb;
  for (int i = 0; i < 8; ++i) { ab[i] = (short)(threadIdx.x + i); bb[i] = (short)(i * 3); }
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = a + i;
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (BF16) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc, 0, 0, 0);
      else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
      for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[k % 16]) : "v"(b));
    }
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc[r] + v[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <bool BF16>
static void launch(int k, float* out, long long* ticks, int iters, int blocks, int threads, hipStream_t st) {
#define CASE(K) if (k == K) hipLaunchKernelGGL((mfma_valu_kernel<K, BF16>), dim3(blocks), dim3(threads), 0, st, out, ticks, iters);
  CASE(0) CASE(2) CASE(4) CASE(8) CASE(12) CASE(16) CASE(24)
#undef CASE
}
extern "C" int mfma_valu(float* out, long long* ticks, int iters, int k, int bf16, int blocks, int threads, void* stream) {
  if (bf16) launch<true>(k, out, ticks, iters, blocks, threads, (hipStream_t)stream);
  else launch<false>(k, out, ticks, iters, blocks, threads, (hipStream_t)stream);
  return (int)hipGetLastError();
}
