// Probe: v_mfma_f32_32x32x2_f32 with both operands read from LDS, as the conv kernels do it (per 8 MFMAs: 8 ds_read2_b32
// issued one channel ahead), against operands from registers; and with the B operand shared by two accumulators.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC mfma_lds.hip -o mfma_lds.so
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define PITCH 33
// MODE 0: registers only; 1: A and B from LDS (1 read2 per MFMA); 2: two accumulators share each B read (0.75 per MFMA)
template <int MODE>
__global__ __launch_bounds__(256) void mfma_lds_kernel(float* out, long long* ticks, int iters) {
  __shared__ float sa[64 * PITCH * 2], sb[64 * PITCH * 2];
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  for (int i = threadIdx.x; i < 64 * PITCH * 2; i += 256) { sa[i] = i * 1e-4f; sb[i] = 1.0f - i * 1e-5f; }
  __syncthreads();
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  const float* pa = sa + li * PITCH + lh;       // A[i = li][k = 2 kk + lh], k contiguous: read2 offsets 0, 2
  const float* pb = sb + lh * PITCH + li;       // B[k = 2 kk + lh][j = li]
  float av[2][8], bv[2][8], cv[2][8];
  auto load_ops = [&](int c, int slot) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      av[slot][t] = pa[(c & 1) * 16 + 2 * t];
      bv[slot][t] = pb[((c & 1) * 16 + 2 * t) * PITCH];
      if (MODE == 2) cv[slot][t] = pa[32 * PITCH + (c & 1) * 16 + 2 * t];
    }
  };
  const long long t0 = clock64();
  if (MODE == 0) {
    float a = lane * 0.001f, b = 1.0f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int t = 0; t < 8; ++t) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
    }
  } else {
    load_ops(0, 0);
    for (int i = 0; i < iters; i += 2) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        load_ops(i + u + 1, (u + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][t], bv[u][t], acc0, 0, 0, 0);
          if (MODE == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(cv[u][t], bv[u][t], acc1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) ticks[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}
extern "C" int mfma_lds(float* out, long long* ticks, int iters, int mode, int blocks, void* stream) {
  if (mode == 0) hipLaunchKernelGGL(mfma_lds_kernel<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, ticks, iters);
  else if (mode == 1) hipLaunchKernelGGL(mfma_lds_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, ticks, iters);
  else hipLaunchKernelGGL(mfma_lds_kernel<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, ticks, iters);
  return (int)hipGetLastError();
}
