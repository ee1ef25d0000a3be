// Generic 2-D convolution family (any channel counts, kernel K x K, stride, padding) -- the first correct path for
// the towers outside the CdSprites+ shapes: Enc_SVHN / Dec_SVHN (models/encoders.py:434-478, models/decoders.py:101-147:
// 32->64, 64->64, 64->128 k4 s2 p0 and ConvT 128->64 k4 s1 p0, 64->64, 64->32).  Plain fp32 FMA, one thread per
// output element; the MFMA kernels of conv.hip keep serving the 3/32-channel k4-s2-p1 layers.  Not yet tuned: the
// wave's 64 consecutive outputs share (image, output channel), so weights are wave-uniform (scalar loads) and the
// input reads of a wave are contiguous rows.
//
//   gather form : y[b,o,oh,ow] = ep(bias[o] + sum_{c,kh,kw} act(x[b,c,oh*S-P+kh, ow*S-P+kw]) * W(o,c,kh,kw))
//                 = Conv2d forward (W = w[o][c]) and ConvTranspose2d input-gradient (W = w[c_in][c_out] read transposed)
//   scatter form: y[b,o,oh,ow] = ep(bias[o] + sum_{c,kh,kw : oh = ih*S-P+kh} act(x[b,c,ih,iw]) * W(o,c,kh,kw))
//                 = ConvTranspose2d forward and Conv2d input-gradient
//   wgrad       : dW(ps,ql,kh,kw) (+)= sum_{b,i,j} small[b,ps,i,j] * act(large[b,ql,i*S-P+kh, j*S-P+kw])
// W(o,c,kh,kw) = w[o*wo + c*wc + kh*K + kw] with the two strides given by the caller.
#include "common.hpp"

struct GConvArgs {
  const float* x;
  const float* w;
  const float* bias;
  const float* aux;
  float* y;
  int B, Cin, Cout, Hin, Win, Hout, Wout, K, S, P, in_act, ep;
  long wo, wc;   // weight strides of the output / reduced channel
};

__device__ __forceinline__ float gconv_act(float v, int act) {
  return act == MMVAE_ACT_RELU ? fmaxf(v, 0.f) : act == MMVAE_ACT_SILU ? dev_silu(v) : v;
}
__device__ __forceinline__ float gconv_ep(float v, float a, int ep) {
  switch (ep) {
    case MMVAE_EP_RELU: return fmaxf(v, 0.f);
    case MMVAE_EP_MUL_RELU_MASK: return a > 0.f ? v : 0.f;
    case MMVAE_EP_MUL_SILU_GRAD: return v * dev_silu_grad(a);
    case MMVAE_EP_SIGMOID_CLAMP: return fminf(fmaxf(dev_sigmoid(v), 1e-6f), 1.0f - 1e-6f);
    case MMVAE_EP_SIGMOID: return dev_sigmoid(v);
    default: return v;
  }
}

__global__ __launch_bounds__(256) void gconv_gather_kernel(GConvArgs a) {
  const long n = (long)a.B * a.Cout * a.Hout * a.Wout;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int ow = (int)(i % a.Wout);
    long t = i / a.Wout;
    const int oh = (int)(t % a.Hout);
    t /= a.Hout;
    const int o = (int)(t % a.Cout), b = (int)(t / a.Cout);
    float acc = a.bias ? a.bias[o] : 0.f;
    const float* xb = a.x + (size_t)b * a.Cin * a.Hin * a.Win;
    for (int c = 0; c < a.Cin; ++c) {
      const float* wp = a.w + (size_t)o * a.wo + (size_t)c * a.wc;
      const float* xc = xb + (size_t)c * a.Hin * a.Win;
      for (int kh = 0; kh < a.K; ++kh) {
        const int ih = oh * a.S - a.P + kh;
        if (ih < 0 || ih >= a.Hin) continue;
        for (int kw = 0; kw < a.K; ++kw) {
          const int iw = ow * a.S - a.P + kw;
          if (iw < 0 || iw >= a.Win) continue;
          acc += gconv_act(xc[ih * a.Win + iw], a.in_act) * wp[kh * a.K + kw];
        }
      }
    }
    a.y[i] = gconv_ep(acc, a.aux ? a.aux[i] : 0.f, a.ep);
  }
}

__global__ __launch_bounds__(256) void gconv_scatter_kernel(GConvArgs a) {
  const long n = (long)a.B * a.Cout * a.Hout * a.Wout;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int ow = (int)(i % a.Wout);
    long t = i / a.Wout;
    const int oh = (int)(t % a.Hout);
    t /= a.Hout;
    const int o = (int)(t % a.Cout), b = (int)(t / a.Cout);
    float acc = a.bias ? a.bias[o] : 0.f;
    const float* xb = a.x + (size_t)b * a.Cin * a.Hin * a.Win;
    for (int kh = 0; kh < a.K; ++kh) {
      const int th = oh + a.P - kh;
      if (th < 0 || th % a.S) continue;
      const int ih = th / a.S;
      if (ih >= a.Hin) continue;
      for (int kw = 0; kw < a.K; ++kw) {
        const int tw = ow + a.P - kw;
        if (tw < 0 || tw % a.S) continue;
        const int iw = tw / a.S;
        if (iw >= a.Win) continue;
        const float* xp = xb + (size_t)ih * a.Win + iw;
        const float* wp = a.w + (size_t)o * a.wo + kh * a.K + kw;
        for (int c = 0; c < a.Cin; ++c)
          acc += gconv_act(xp[(size_t)c * a.Hin * a.Win], a.in_act) * wp[(size_t)c * a.wc];
      }
    }
    a.y[i] = gconv_ep(acc, a.aux ? a.aux[i] : 0.f, a.ep);
  }
}

struct GWgradArgs {
  const float* small;   // (B, Ps, Hs, Ws)
  const float* large;   // (B, Ql, Hl, Wl)
  float* dw;
  int B, Ps, Ql, Hs, Ws, Hl, Wl, K, S, P, small_act, large_act, accumulate;
  long sp, sq;          // dw strides of the small / large channel
};
// one workgroup per (ps, ql) pair: K*K <= 16 accumulators per thread over the (b, i, j) positions
__global__ __launch_bounds__(256) void gconv_wgrad_kernel(GWgradArgs a) {
  __shared__ float red[4];
  const int ps = blockIdx.x % a.Ps, ql = blockIdx.x / a.Ps;
  float acc[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.f;
  const long npos = (long)a.B * a.Hs * a.Ws;
  for (long p = threadIdx.x; p < npos; p += 256) {
    const int j = (int)(p % a.Ws);
    long t = p / a.Ws;
    const int i = (int)(t % a.Hs), b = (int)(t / a.Hs);
    const float sv = gconv_act(a.small[(((size_t)b * a.Ps + ps) * a.Hs + i) * a.Ws + j], a.small_act);
    const float* lp = a.large + ((size_t)b * a.Ql + ql) * a.Hl * a.Wl;
    for (int kh = 0; kh < a.K; ++kh) {
      const int ih = i * a.S - a.P + kh;
      if (ih < 0 || ih >= a.Hl) continue;
      for (int kw = 0; kw < a.K; ++kw) {
        const int iw = j * a.S - a.P + kw;
        if (iw < 0 || iw >= a.Wl) continue;
        acc[kh * a.K + kw] += sv * gconv_act(lp[ih * a.Wl + iw], a.large_act);
      }
    }
  }
  for (int k = 0; k < a.K * a.K; ++k) {
    const float v = block_sum_256(acc[k], red);
    if (threadIdx.x == 0) {
      float* d = a.dw + (size_t)ps * a.sp + (size_t)ql * a.sq + k;
      *d = a.accumulate ? *d + v : v;
    }
  }
}
// db[ch] (+)= sum_{b,h,w} act?(t[b,ch,h,w])
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ t, float* __restrict__ db, int B, int C,
                                                          int HW, int accumulate) {
  __shared__ float red[4];
  const int ch = blockIdx.x;
  float acc = 0.f;
  for (long p = threadIdx.x; p < (long)B * HW; p += 256) {
    const int b = (int)(p / HW), r = (int)(p % HW);
    acc += t[((size_t)b * C + ch) * HW + r];
  }
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) db[ch] = accumulate ? db[ch] + acc : acc;
}

static inline unsigned gconv_blocks(long n) {
  long b = (n + 255) / 256;
  return (unsigned)(b > 65535 * 16 ? 65535 * 16 : b);
}
static inline bool gconv_ep_ok(int ep) {
  return ep == MMVAE_EP_NONE || ep == MMVAE_EP_RELU || ep == MMVAE_EP_MUL_RELU_MASK || ep == MMVAE_EP_MUL_SILU_GRAD ||
         ep == MMVAE_EP_SIGMOID_CLAMP || ep == MMVAE_EP_SIGMOID;
}

// Conv2d forward: x (B,Cin,Hin,Win), w [Cout][Cin][K][K] -> y (B,Cout,Hout,Wout), Hout = (Hin + 2P - K) / S + 1
extern "C" int mmvae_conv2d_generic_fwd(const float* x, const float* w, const float* bias, const float* aux, float* y,
                                        int B, int Cin, int Cout, int Hin, int Win, int K, int S, int P, int in_act,
                                        int ep_mode, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(x && w && y && B > 0 && Cin > 0 && Cout > 0 && K > 0 && K <= 4 && S > 0 && P >= 0);
  if (!gconv_ep_ok(ep_mode) || in_act == MMVAE_ACT_GELU) return MMVAE_ERR_UNSUPPORTED;
  const int Hout = (Hin + 2 * P - K) / S + 1, Wout = (Win + 2 * P - K) / S + 1;
  if (Hout < 1 || Wout < 1) return MMVAE_ERR_ARG;
  GConvArgs a{x, w, bias, aux, y, B, Cin, Cout, Hin, Win, Hout, Wout, K, S, P, in_act, ep_mode, (long)Cin * K * K,
              (long)K * K};
  hipLaunchKernelGGL(gconv_gather_kernel, dim3(gconv_blocks((long)B * Cout * Hout * Wout)), dim3(256), 0,
                     (hipStream_t)stream, a);
  return mmvae_launch_status();
}
// Conv2d input gradient: dy (B,Cout,Hout,Wout), w [Cout][Cin][K][K] -> dx (B,Cin,Hin,Win) (scatter form over dy)
extern "C" int mmvae_conv2d_generic_dgrad(const float* dy, const float* w, const float* aux, float* dx, int B, int Cin,
                                          int Cout, int Hin, int Win, int K, int S, int P, int ep_mode,
                                          mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(dy && w && dx && B > 0 && K > 0 && K <= 4 && S > 0);
  if (!gconv_ep_ok(ep_mode)) return MMVAE_ERR_UNSUPPORTED;
  const int Hout = (Hin + 2 * P - K) / S + 1, Wout = (Win + 2 * P - K) / S + 1;
  // "input" of the scatter = dy (Cout channels, Hout x Wout), "output" = dx (Cin channels); W(o = c_in, c = c_out)
  GConvArgs a{dy, w, nullptr, aux, dx, B, Cout, Cin, Hout, Wout, Hin, Win, K, S, P, MMVAE_ACT_NONE, ep_mode,
              (long)K * K, (long)Cin * K * K};
  hipLaunchKernelGGL(gconv_scatter_kernel, dim3(gconv_blocks((long)B * Cin * Hin * Win)), dim3(256), 0,
                     (hipStream_t)stream, a);
  return mmvae_launch_status();
}
// ConvTranspose2d forward: x (B,Cin,Hin,Win), w [Cin][Cout][K][K] -> y (B,Cout,Hout,Wout), Hout = (Hin-1) S - 2P + K
extern "C" int mmvae_convT2d_generic_fwd(const float* x, const float* w, const float* bias, const float* aux, float* y,
                                         int B, int Cin, int Cout, int Hin, int Win, int K, int S, int P, int in_act,
                                         int ep_mode, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(x && w && y && B > 0 && Cin > 0 && Cout > 0 && K > 0 && K <= 4 && S > 0 && P >= 0);
  if (!gconv_ep_ok(ep_mode) || in_act == MMVAE_ACT_GELU) return MMVAE_ERR_UNSUPPORTED;
  const int Hout = (Hin - 1) * S - 2 * P + K, Wout = (Win - 1) * S - 2 * P + K;
  GConvArgs a{x, w, bias, aux, y, B, Cin, Cout, Hin, Win, Hout, Wout, K, S, P, in_act, ep_mode, (long)K * K,
              (long)Cout * K * K};
  hipLaunchKernelGGL(gconv_scatter_kernel, dim3(gconv_blocks((long)B * Cout * Hout * Wout)), dim3(256), 0,
                     (hipStream_t)stream, a);
  return mmvae_launch_status();
}
// ConvTranspose2d input gradient: dy (B,Cout,Hout,Wout), w [Cin][Cout][K][K] -> dx (B,Cin,Hin,Win) (gather form over dy)
extern "C" int mmvae_convT2d_generic_dgrad(const float* dy, const float* w, const float* aux, float* dx, int B, int Cin,
                                           int Cout, int Hin, int Win, int K, int S, int P, int ep_mode,
                                           mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(dy && w && dx && B > 0 && K > 0 && K <= 4 && S > 0);
  if (!gconv_ep_ok(ep_mode)) return MMVAE_ERR_UNSUPPORTED;
  const int Hout = (Hin - 1) * S - 2 * P + K, Wout = (Win - 1) * S - 2 * P + K;
  GConvArgs a{dy, w, nullptr, aux, dx, B, Cout, Cin, Hout, Wout, Hin, Win, K, S, P, MMVAE_ACT_NONE, ep_mode,
              (long)Cout * K * K, (long)K * K};
  hipLaunchKernelGGL(gconv_gather_kernel, dim3(gconv_blocks((long)B * Cin * Hin * Win)), dim3(256), 0,
                     (hipStream_t)stream, a);
  return mmvae_launch_status();
}
// Weight + bias gradients.  Conv2d: dw[o][c] (+)= dy (x) act(x), db[o] (+)= sum dy.
extern "C" int mmvae_conv2d_generic_wgrad(const float* dy, const float* x, float* dw, float* db, int B, int Cin, int Cout,
                                          int Hin, int Win, int K, int S, int P, int x_act, int accumulate,
                                          mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(dy && x && dw && B > 0 && K > 0 && K <= 4);
  const int Hout = (Hin + 2 * P - K) / S + 1, Wout = (Win + 2 * P - K) / S + 1;
  GWgradArgs a{dy, x, dw, B, Cout, Cin, Hout, Wout, Hin, Win, K, S, P, MMVAE_ACT_NONE, x_act, accumulate ? 1 : 0,
               (long)Cin * K * K, (long)K * K};
  hipLaunchKernelGGL(gconv_wgrad_kernel, dim3(Cout * Cin), dim3(256), 0, (hipStream_t)stream, a);
  if (db)
    hipLaunchKernelGGL(channel_sum_kernel, dim3(Cout), dim3(256), 0, (hipStream_t)stream, dy, db, B, Cout, Hout * Wout,
                       accumulate ? 1 : 0);
  return mmvae_launch_status();
}
// ConvTranspose2d: dw[c][o] (+)= act(x) (x) dy, db[o] (+)= sum dy.
extern "C" int mmvae_convT2d_generic_wgrad(const float* dy, const float* x, float* dw, float* db, int B, int Cin,
                                           int Cout, int Hin, int Win, int K, int S, int P, int x_act, int accumulate,
                                           mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(dy && x && dw && B > 0 && K > 0 && K <= 4);
  const int Hout = (Hin - 1) * S - 2 * P + K, Wout = (Win - 1) * S - 2 * P + K;
  GWgradArgs a{x, dy, dw, B, Cin, Cout, Hin, Win, Hout, Wout, K, S, P, x_act, MMVAE_ACT_NONE, accumulate ? 1 : 0,
               (long)Cout * K * K, (long)K * K};
  hipLaunchKernelGGL(gconv_wgrad_kernel, dim3(Cin * Cout), dim3(256), 0, (hipStream_t)stream, a);
  if (db)
    hipLaunchKernelGGL(channel_sum_kernel, dim3(Cout), dim3(256), 0, (hipStream_t)stream, dy, db, B, Cout, Hout * Wout,
                       accumulate ? 1 : 0);
  return mmvae_launch_status();
}

// y = sigmoid(x) ; dx = dy * y * (1 - y)   (Dec_MNIST's nn.Sigmoid, models/decoders.py:266)
__global__ __launch_bounds__(256) void sigmoid_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      float* __restrict__ out, long n, int bwd) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    if (!bwd) out[i] = dev_sigmoid(a[i]);
    else out[i] = a[i] * b[i] * (1.0f - b[i]);
  }
}
extern "C" int mmvae_sigmoid_fwd(const float* x, float* y, long n, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(x && y && n > 0);
  hipLaunchKernelGGL(sigmoid_kernel, dim3(gconv_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, nullptr, y, n, 0);
  return mmvae_launch_status();
}
extern "C" int mmvae_sigmoid_bwd(const float* dy, const float* y, float* dx, long n, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(dy && y && dx && n > 0);
  hipLaunchKernelGGL(sigmoid_kernel, dim3(gconv_blocks(n)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, n, 1);
  return mmvae_launch_status();
}

// ---------------------------------------------------------------------------------------------
// grouped bias (a ConvTranspose2d on a 1x1 input is the GEMM (B, Cin) x (Cin, Cout*K*K); its bias repeats over the
// K*K positions of a channel)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bias_group_add_kernel(float* __restrict__ y, const float* __restrict__ bias,
                                                             long n4, int N4, int G4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float bv = bias[(int)(i % N4) / G4];
    float4 v = reinterpret_cast<float4*>(y)[i];
    v.x += bv; v.y += bv; v.z += bv; v.w += bv;
    reinterpret_cast<float4*>(y)[i] = v;
  }
}
// stage 1: workgroup (c, part) sums its block of rows of channel c's G columns -> ws[part * C + c]
__global__ __launch_bounds__(256) void bias_group_grad_kernel(const float* __restrict__ dy, float* __restrict__ ws,
                                                              int rows, int C, int G, int rows_per) {
  __shared__ float red[4];
  const int c = blockIdx.x, part = blockIdx.y;
  const long N = (long)C * G;
  const int r0 = part * rows_per, r1 = min(rows, r0 + rows_per);
  float acc = 0.f;
  for (long e = threadIdx.x; e < (long)(r1 - r0) * G; e += 256) {
    const long r = e / G, g = e - r * G;
    acc += dy[(r0 + r) * N + (long)c * G + g];
  }
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) ws[(size_t)part * C + c] = acc;
}
// 16-byte form: workgroup = (chunk of 256 column quads, block of rows); a thread walks its quad down the rows with
// coalesced float4 loads (8 in flight), adds the 4 components, and G / 4 neighbouring threads add up to one channel.
// (The form above reads 64 contiguous bytes per row and channel: 181 us for the 63 MB of the MNIST-SVHN decoder's
// 7680 x 2048 gradient; this one streams whole rows.)
__global__ __launch_bounds__(256) void bias_group_grad4_kernel(const float* __restrict__ dy, float* __restrict__ ws,
                                                               int rows, int C, int G, int rows_per) {
  __shared__ float sums[256];
  const int part = blockIdx.y, q4 = G >> 2;
  const long n4 = (long)C * q4;                          // float4 per row
  const long col = (long)blockIdx.x * 256 + threadIdx.x;
  const int r0 = part * rows_per, r1 = min(rows, r0 + rows_per);
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col < n4) {
    const float4* p = reinterpret_cast<const float4*>(dy) + col;
    int r = r0;
    for (; r + 7 < r1; r += 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(r + u) * n4];
#pragma unroll
      for (int u = 0; u < 8; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
    }
    for (; r < r1; ++r) {
      const float4 v = p[(size_t)r * n4];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  }
  sums[threadIdx.x] = (a.x + a.y) + (a.z + a.w);
  __syncthreads();
  const int per = 256 / q4;                              // channels of this chunk
  if ((int)threadIdx.x < per) {
    const long c = (long)blockIdx.x * per + threadIdx.x;
    if (c < C) {
      float t = 0.f;
      for (int i = 0; i < q4; ++i) t += sums[threadIdx.x * q4 + i];
      ws[(size_t)part * C + c] = t;
    }
  }
}
extern "C" int mmvae_bias_group_parts(int rows) {
  int parts = (rows + 63) / 64;
  return parts < 1 ? 1 : (parts > 128 ? 128 : parts);
}
extern "C" size_t mmvae_bias_group_ws_floats(int rows, int C) { return (size_t)mmvae_bias_group_parts(rows) * C; }
extern "C" int mmvae_bias_group_add(float* y, const float* bias, int rows, int C, int G, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(y && bias && rows > 0 && C > 0 && G > 0);
  if (G % 4) return MMVAE_ERR_UNSUPPORTED;
  const long n4 = (long)rows * C * G / 4;
  long blocks = (n4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(bias_group_add_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, y, bias, n4,
                     C * G / 4, G / 4);
  return mmvae_launch_status();
}
extern "C" int mmvae_bias_group_grad(const float* dy, float* db, float* ws, int rows, int C, int G, int accumulate,
                                     mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(dy && db && ws && rows > 0 && C > 0 && G > 0);
  const int parts = mmvae_bias_group_parts(rows);
  const int rows_per = (rows + parts - 1) / parts;
  const int q4 = G >> 2;
  if ((G & 3) == 0 && q4 <= 256 && 256 % q4 == 0 && (((uintptr_t)dy) & 15) == 0) {
    const long n4 = (long)C * q4;
    hipLaunchKernelGGL(bias_group_grad4_kernel, dim3((unsigned)((n4 + 255) / 256), parts), dim3(256), 0,
                       (hipStream_t)stream, dy, ws, rows, C, G, rows_per);
  } else {
    hipLaunchKernelGGL(bias_group_grad_kernel, dim3(C, parts), dim3(256), 0, (hipStream_t)stream, dy, ws, rows, C, G,
                       rows_per);
  }
  int rc = mmvae_launch_status();
  if (rc || accumulate == MMVAE_ACC_DEFER) return rc;
  return mmvae_reduce_rows(ws, db, parts, C, C, accumulate, stream);
}
