// Probe: issue rate of v_mfma_f32_32x32x2_f32 from one wave (dependent chain / two independent chains) and from two
// waves per SIMD.  hipcc --offload-arch=gfx950 -O3 -shared -fPIC mfma_rate.hip -o mfma_rate.so
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void mfma_rate_kernel(float* out, long long* ticks, int iters) {
  f32x16 acc[NACC];
  for (int n = 0; n < NACC; ++n)
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  float a = threadIdx.x * 0.001f, b = 1.0f;
  __syncthreads();
  const long long t0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u % NACC], 0, 0, 0);
  }
  const long long t1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int n = 0; n < NACC; ++n)
    for (int r = 0; r < 16; ++r) s += acc[n][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    ticks[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2] = t1 - t0;
    ticks[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2 + 1] = w1 - w0;
  }
}
extern "C" int mfma_rate(float* out, long long* ticks, int iters, int nacc, int blocks, int threads, void* stream) {
  if (nacc == 1) hipLaunchKernelGGL(mfma_rate_kernel<1>, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, out, ticks, iters);
  else if (nacc == 2) hipLaunchKernelGGL(mfma_rate_kernel<2>, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, out, ticks, iters);
  else hipLaunchKernelGGL(mfma_rate_kernel<4>, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, out, ticks, iters);
  return (int)hipGetLastError();
}
