// Global average pooling of the ResNet-50 image tower (`encoder: CNN`, reference models/encoders.py:86-127 = torchvision's
// resnet50 -> SiLU -> heads).  Everything else of the tower -- the stem, the 16 bottlenecks, their
// BatchNorms and the max pooling -- is the fused engine of rconv.hip; activations are NHWC = (rows = B*H*W, C) matrices.
#include "common.hpp"

static inline unsigned ew_blocks(long n) {
  long b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

// ---------------------------------------------------------------------------------------------
// AdaptiveAvgPool2d(1) on relu(x): (B, HW, C) -> (B, C), and its backward
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B,
                                                          int HW, int C, int act) {
  const long total = (long)B * C;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = (int)(e % C), b = (int)(e / C);
    float s = 0.f;
    for (int p = 0; p < HW; ++p) s += apply_in_act(x[((size_t)b * HW + p) * C + c], act);
    y[e] = s / (float)HW;
  }
}
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          float* __restrict__ dx, int B, int HW, int C, int act) {
  const long total = (long)B * HW * C;
  const float inv = 1.0f / (float)HW;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = (int)(e % C), b = (int)(e / ((long)HW * C));
    float g = dy[(size_t)b * C + c] * inv;
    if (act == MMVAE_ACT_RELU && !(x[e] > 0.f)) g = 0.f;
    dx[e] = g;
  }
}
extern "C" int mmvae_avgpool_fwd(const float* x, float* y, int B, int HW, int C, int in_act, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(x && y && B > 0 && HW > 0 && C > 0);
  hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(ew_blocks((long)B * C)), dim3(256), 0, (hipStream_t)stream, x, y, B, HW, C,
                     in_act);
  return mmvae_launch_status();
}
extern "C" int mmvae_avgpool_bwd(const float* dy, const float* x, float* dx, int B, int HW, int C, int in_act,
                                 mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(dy && x && dx && B > 0 && HW > 0 && C > 0);
  hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(ew_blocks((long)B * HW * C)), dim3(256), 0, (hipStream_t)stream, dy, x, dx,
                     B, HW, C, in_act);
  return mmvae_launch_status();
}
