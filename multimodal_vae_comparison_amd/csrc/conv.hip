// Convolution family (4x4 / stride 2 / pad 1) for gfx950: one translation unit so that the three kernel forms can
// also be combined into fused backward launches.
//   conv_gather.inc  : "gather" implicit GEMM   (Conv2d forward, ConvTranspose2d input-gradient)
//   conv_scatter.inc : "scatter" implicit GEMM  (ConvTranspose2d forward, Conv2d input-gradient)
//   conv_wgrad.inc   : weight / bias gradients
// The C-ABI entry points of include/mmvae_hip.h are at the bottom.
#include "conv_common.hpp"

#include "conv_gather.inc"
#include "conv_gather_b16.inc"
#include "conv_scatter.inc"
#include "conv_scatter_b16.inc"
#include "conv_wgrad.inc"

// ------------------------------------------------------------------------------------------------
// Fused backward of one layer: the input-gradient kernel and the weight-gradient kernel only share their inputs
// and each alone fills a fraction of the chip at batch 128 (256-512 resp. 256 workgroups of one wave per SIMD), so
// both run in ONE launch: workgroups [0, n_w) execute the weight-gradient body, the rest the input-gradient body.
// ------------------------------------------------------------------------------------------------
constexpr int cmax(int a, int b) { return a > b ? a : b; }

template <typename G, typename W, bool SPEC = false>
__global__ __launch_bounds__(256) void conv2d_bwd_fused_kernel(ConvScatterArgs ad, ConvWgradArgs aw, int n_w, int w_gx) {
  MMVAE_TRACE_STAMP(10 + G::LGH);
  __shared__ __attribute__((aligned(16))) float smem[cmax(G::SMEM, W::SMEM)];
  if ((int)blockIdx.x < n_w) conv_wgrad_body<W, SPEC>(aw, blockIdx.x % w_gx, blockIdx.x / w_gx, w_gx, smem);
  else   // (the data-gradient tiles XCD-contiguous, see xcd_contiguous: needs the body's first workgroup on XCD 0)
    conv_scatter_body<G>(ad, (n_w & 7) ? (int)blockIdx.x - n_w : xcd_contiguous(blockIdx.x - n_w, gridDim.x - n_w), smem);
}
template <typename G, typename W, bool SPEC = false>
__global__ __launch_bounds__(256) void convT_bwd_fused_kernel(ConvGatherArgs ad, ConvWgradArgs aw, int n_w, int w_gx) {
  MMVAE_TRACE_STAMP(14 - G::LGH);
  __shared__ __attribute__((aligned(16))) float smem[cmax(G::SMEM, W::SMEM)];
  if ((int)blockIdx.x < n_w) conv_wgrad_body<W, SPEC>(aw, blockIdx.x % w_gx, blockIdx.x / w_gx, w_gx, smem);
  else
    conv_gather_body<G, G::RWONLY>(ad, (n_w & 7) ? (int)blockIdx.x - n_w : xcd_contiguous(blockIdx.x - n_w, gridDim.x - n_w),
                                   smem);
}

// Conv2d backward with the data gradient on the split-bf16 scatter body (conv_scatter_b16.inc; input activation NONE)
template <typename GS, typename W, bool SPEC = false>
__global__ __launch_bounds__(256) void conv2d_bwd_fused_b16_kernel(ConvScatterArgs ad, ConvWgradArgs aw, int n_w, int w_gx) {
  MMVAE_TRACE_STAMP(10 + GS::LGH);
  __shared__ __attribute__((aligned(16))) unsigned char smem[cmax(GS::SMEM_TOTAL, 4 * W::SMEM)];
  if ((int)blockIdx.x < n_w)
    conv_wgrad_body<W, SPEC>(aw, blockIdx.x % w_gx, blockIdx.x / w_gx, w_gx, reinterpret_cast<float*>(smem));
  else
    conv_scatter_b16_body<GS, MMVAE_ACT_NONE>(
        ad, (n_w & 7) ? (int)blockIdx.x - n_w : xcd_contiguous(blockIdx.x - n_w, gridDim.x - n_w), 0, smem);
}

// the same launch with the data gradient on the split-bf16 gather body (conv_gather_b16.inc; input activation NONE)
template <typename GB, typename W, bool SPEC = false>
__global__ __launch_bounds__(256) void convT_bwd_fused_b16_kernel(ConvGatherArgs ad, ConvWgradArgs aw, int n_w, int w_gx) {
  MMVAE_TRACE_STAMP(14 - GB::LGH);
  __shared__ __attribute__((aligned(16))) unsigned char smem[cmax(GB::SMEM_TOTAL, 4 * W::SMEM)];
  if ((int)blockIdx.x < n_w)
    conv_wgrad_body<W, SPEC>(aw, blockIdx.x % w_gx, blockIdx.x / w_gx, w_gx, reinterpret_cast<float*>(smem));
  else
    conv_gather_b16p_body<GB, MMVAE_ACT_NONE>(
        ad, (n_w & 7) ? (int)blockIdx.x - n_w : xcd_contiguous(blockIdx.x - n_w, gridDim.x - n_w), 0, smem);
}

// 32-channel weight gradients: 8-channel chunks (4 x splits workgroups) when 16-channel chunks would leave the chip
// half empty.
// (Round 2, in the step: with at most 8 macro tiles per split -- batch <= 128 on the 32-wide maps -- the 16-channel
// chunks win, 0.4162 vs 0.4231 ms/step over three pairs: half as many weight-gradient workgroups beside the data-gradient
// ones and the text tower's; with longer splits -- batch 256: 1.27 vs 1.33 ms -- the 8-channel chunks keep the
// weight-gradient body from becoming the launch's long pole.)
// (Round 4: with the text tower's forward on the wave-per-sequence kernels -- no LDS, 128 SIMDs -- and its weight
// gradients in one launch per layer, the 8-channel chunks win at batch 128 too: 0.4110 -> 0.4050 and 0.4110 -> 0.4047
// ms/step, two same-box rounds; the short-split exception is gone.)
static inline bool wgrad_qc8(int nsplit, int n_macro) {
  (void)n_macro;
  return nsplit * 2 < 256;
}

static inline int dact_ep(int act) {
  return act == MMVAE_ACT_SILU ? MMVAE_EP_MUL_SILU_GRAD : act == MMVAE_ACT_RELU ? MMVAE_EP_MUL_RELU_MASK
                                                        : act == MMVAE_ACT_GELU ? MMVAE_EP_MUL_GELU_GRAD : MMVAE_EP_NONE;
}
static int conv_bwd_reduce(float* ws, float* dw, float* db, int B, int Q, int Hs, int nbias, int accumulate,
                           mmvae_stream_t stream) {
  if (accumulate == MMVAE_ACC_DEFER) return MMVAE_OK;
  int rows, rowlen, bias_col;
  conv_wgrad_layout(B, 32, Q, Hs, &rows, &rowlen, &bias_col);
  int rc = mmvae_reduce_rows(ws, dw, rows, bias_col, rowlen, accumulate, stream);
  if (rc) return rc;
  if (db) rc = mmvae_reduce_rows(ws + bias_col, db, rows, nbias, rowlen, accumulate, stream);
  return rc;
}

// Conv2d backward: dx = (dy (*) w) * act'(x), dw (+)= dy (x) act(x), db (+)= sum dy.  x (B,Cin,2Hout,2Hout).
extern "C" int mmvae_conv2d_k4s2_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, float* db,
                                     float* ws, int B, int Cin, int Cout, int Hout, int x_act, int accumulate,
                                     mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(dy && x && w && dx && dw && ws && B > 0);
  hipStream_t st = (hipStream_t)stream;
  if (Cin != 32 || Cout != 32 || Hout < 4 || Hout > 16 || (Hout & (Hout - 1))) {   // not a fused shape: two launches
    int rc = conv_wgrad_dispatch(dy, x, dw, db, ws, B, Cout, Cin, Hout, MMVAE_ACT_NONE, x_act, db ? 1 : 0, accumulate, st);
    if (rc) return rc;
    return conv_scatter_dispatch(dy, w, nullptr, x, dx, B, Cout, Cin, Hout, MMVAE_ACT_NONE, dact_ep(x_act), st);
  }
  const int ep = dact_ep(x_act);
  ConvScatterArgs ad{dy, w, nullptr, x, dx, B, MMVAE_ACT_NONE, ep, CONV_CO};
  const long tiles = ((long)B * Hout * Hout + 31) / 32;
  const int n_macro = wgrad_n_macro(B, Hout), nsplit = wgrad_splits(n_macro, 32);
  ConvWgradArgs aw{dy, x, ws, B, MMVAE_ACT_NONE, x_act, db ? 1 : 0, n_macro, 32, (long)32 * 32 * 16 + 32};
  // one instantiation per layer shape: the scatter plan is a function of the map size at a given batch, but both
  // template arguments must be compile-time, so the (few) combinations are enumerated by the two visitors
  bool launched = false;
  // (at batch 128 the fused launch is faster on the fp32 body: 0.391 vs 0.400 ms/step; from batch 256 on 80 KB per workgroup pays)
  if (conv_split_bf16_enabled() && Hout == 16 && tiles >= 2048) {       // the layer whose data gradient the split-bf16 scatter body serves
    using GS = ScatterB16Geom<32, 4>;
    const int n_d = (int)(((long)B * GS::HIN + GS::NR - 1) / GS::NR);
    auto go = [&](auto wg) {
      using W = decltype(wg);
      const int n_w = nsplit * W::NCH;
      if (wgrad_spec(n_macro, nsplit))
        hipLaunchKernelGGL((conv2d_bwd_fused_b16_kernel<GS, W, true>), dim3(n_w + n_d), dim3(256), 0, st, ad, aw, n_w, nsplit);
      else
        hipLaunchKernelGGL((conv2d_bwd_fused_b16_kernel<GS, W>), dim3(n_w + n_d), dim3(256), 0, st, ad, aw, n_w, nsplit);
    };
    if (wgrad_qc8(nsplit, n_macro)) go(WgradGeom<32, 4, 8>{});
    else go(WgradGeom<32, 4>{});
    int rc0 = mmvae_launch_status();
    if (rc0) return rc0;
    return conv_bwd_reduce(ws, dw, db, B, Cin, Hout, 32, accumulate, stream);
  }
  scatter_visit(Hout, scatter_plan(tiles, 32, Hout), [&](auto g) {
    using G = decltype(g);
    if constexpr (G::LGH <= 4) {
      const int n_d = (int)scatter_grid(B, Hout, G::TM);
      if (wgrad_qc8(nsplit, n_macro)) {
        using W = WgradGeom<32, G::LGH, 8>;
        const int n_w = nsplit * W::NCH;
        if (wgrad_spec(n_macro, nsplit))
          hipLaunchKernelGGL((conv2d_bwd_fused_kernel<G, W, true>), dim3(n_w + n_d), dim3(256), 0, st, ad, aw, n_w, nsplit);
        else
          hipLaunchKernelGGL((conv2d_bwd_fused_kernel<G, W>), dim3(n_w + n_d), dim3(256), 0, st, ad, aw, n_w, nsplit);
      } else {
        using W = WgradGeom<32, G::LGH>;
        const int n_w = nsplit * W::NCH;
        hipLaunchKernelGGL((conv2d_bwd_fused_kernel<G, W>), dim3(n_w + n_d), dim3(256), 0, st, ad, aw, n_w, nsplit);
      }
      launched = true;
    }
  });
  if (!launched) return MMVAE_ERR_UNSUPPORTED;
  int rc = mmvae_launch_status();
  if (rc) return rc;
  return conv_bwd_reduce(ws, dw, db, B, Cin, Hout, 32, accumulate, stream);
}

// ConvTranspose2d backward: dx = (dy (*) w) * act'(x), dw (+)= act(x) (x) dy, db (+)= sum dy.  dy (B,Cout,2Hin,2Hin).
extern "C" int mmvae_convT2d_k4s2_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, float* db,
                                      float* ws, int B, int Cin, int Cout, int Hin, int x_act, int accumulate,
                                      mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(dy && x && w && dx && dw && ws && B > 0);
  hipStream_t st = (hipStream_t)stream;
  const int ep = dact_ep(x_act);
  const bool fused = Cin == 32 && (Cout == 32 || Cout == 3) && Hin >= 4 && Hin <= 32 && !(Hin & (Hin - 1)) &&
                     !(Cout == 3 && Hin != 32) && !(Cout == 32 && Hin > 16);
  if (!fused) {
    int rc = conv_wgrad_dispatch(x, dy, dw, db, ws, B, Cin, Cout, Hin, x_act, MMVAE_ACT_NONE, db ? 2 : 0, accumulate, st);
    if (rc) return rc;
    return conv_gather_dispatch(dy, w, nullptr, x, dx, B, Cout, Cin, 2 * Hin, MMVAE_ACT_NONE, ep, st);
  }
  // input gradient = gather conv over dy (2Hin x 2Hin, Cout channels) -> (Hin x Hin, 32 channels)
  ConvGatherArgs ad{dy, w, nullptr, x, dx, B, MMVAE_ACT_NONE, ep, CONV_CO};
  const long tiles = ((long)B * Hin * Hin + 31) / 32;
  const int n_macro = wgrad_n_macro(B, Hin), nsplit = wgrad_splits(n_macro, Cout);
  ConvWgradArgs aw{x, dy, ws, B, x_act, MMVAE_ACT_NONE, db ? 2 : 0, n_macro, 32, (long)32 * Cout * 16 + 32};
  bool launched = false;
  // the 32-channel layers whose data gradient the split-bf16 gather body serves (as conv_gather_b16_launch)
  auto b16 = [&](auto gb) {
    using GB = decltype(gb);
    const int n_d = (int)((long)B * GB::HOUT / GB::NR);
    auto go = [&](auto wg) {
      using W = decltype(wg);
      const int n_w = nsplit * W::NCH;
      if (wgrad_spec(n_macro, nsplit))
        hipLaunchKernelGGL((convT_bwd_fused_b16_kernel<GB, W, true>), dim3(n_w + n_d), dim3(256), 0, st, ad, aw, n_w, nsplit);
      else
        hipLaunchKernelGGL((convT_bwd_fused_b16_kernel<GB, W>), dim3(n_w + n_d), dim3(256), 0, st, ad, aw, n_w, nsplit);
    };
    if constexpr (GB::CIN == 32) {
      if (wgrad_qc8(nsplit, n_macro)) go(WgradGeom<32, GB::LGH - 1, 8>{});
      else go(WgradGeom<32, GB::LGH - 1>{});
    } else {
      go(WgradGeom<GB::CIN, GB::LGH - 1>{});
    }
    launched = true;
  };
  const bool sb = conv_split_bf16_enabled();
  if (!sb) {}
  else if (Cout == 3 && Hin == 32 && tiles >= 1024) b16(GatherB16Geom<3, 6, 4, 3>{});
  else if (Cout == 32 && Hin == 16 && tiles >= 512) b16(GatherB16Geom<32, 5, 4, 4>{});
  else if (Cout == 32 && Hin == 8 && tiles >= 1024) b16(GatherB16Geom<32, 4, 2, 4>{});
  else if (Cout == 32 && Hin == 8 && tiles >= 256) b16(GatherB16Geom<32, 4, 1, 8>{});
  if (launched) {
    int rc0 = mmvae_launch_status();
    if (rc0) return rc0;
    return conv_bwd_reduce(ws, dw, db, B, Cout, Hin, Cout, accumulate, stream);
  }
  gather_visit(Cout, 2 * Hin, gather_plan(Cout, tiles, 2 * Hin), [&](auto g) {
    using G = decltype(g);
    if constexpr (G::LGH >= 3 && (G::CIN == 32 || G::LGH == 6)) {
      const int n_d = (int)gather_grid(B, Hin, G::TM);    // (weight gradient: small map = the gather's output map)
      if (G::CIN == 32 && wgrad_qc8(nsplit, n_macro)) {
        using W = WgradGeom<G::CIN, G::LGH - 1, G::CIN == 32 ? 8 : WGRAD_QC_DEFAULT(G::CIN)>;
        const int n_w = nsplit * W::NCH;
        if (wgrad_spec(n_macro, nsplit))
          hipLaunchKernelGGL((convT_bwd_fused_kernel<G, W, true>), dim3(n_w + n_d), dim3(256), 0, st, ad, aw, n_w, nsplit);
        else
          hipLaunchKernelGGL((convT_bwd_fused_kernel<G, W>), dim3(n_w + n_d), dim3(256), 0, st, ad, aw, n_w, nsplit);
      } else {
        using W = WgradGeom<G::CIN, G::LGH - 1>;
        const int n_w = nsplit * W::NCH;
        if (wgrad_spec(n_macro, nsplit))
          hipLaunchKernelGGL((convT_bwd_fused_kernel<G, W, true>), dim3(n_w + n_d), dim3(256), 0, st, ad, aw, n_w, nsplit);
        else
          hipLaunchKernelGGL((convT_bwd_fused_kernel<G, W>), dim3(n_w + n_d), dim3(256), 0, st, ad, aw, n_w, nsplit);
      }
      launched = true;
    }
  });
  if (!launched) return MMVAE_ERR_UNSUPPORTED;
  int rc = mmvae_launch_status();
  if (rc) return rc;
  return conv_bwd_reduce(ws, dw, db, B, Cout, Hin, Cout, accumulate, stream);
}

extern "C" int mmvae_conv2d_k4s2_fwd(const float* x, const float* w, const float* bias, const float* aux, float* y,
                                     int B, int Cin, int Cout, int Hin, int in_act, int ep_mode,
                                     mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(x && w && y && B > 0);
  return conv_gather_dispatch(x, w, bias, aux, y, B, Cin, Cout, Hin, in_act, ep_mode, (hipStream_t)stream);
}
extern "C" int mmvae_conv2d_k4s2_dgrad(const float* dy, const float* w, const float* aux, float* dx, int B, int Cin,
                                       int Cout, int Hout, int ep_mode, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(dy && w && dx && B > 0);
  // reduce over the conv's output channels, produce its input channels; w is [Cout][Cin] = [red][out]
  return conv_scatter_dispatch(dy, w, nullptr, aux, dx, B, Cout, Cin, Hout, MMVAE_ACT_NONE, ep_mode,
                               (hipStream_t)stream);
}
extern "C" int mmvae_conv2d_k4s2_wgrad(const float* dy, const float* x, float* dw, float* db, float* ws, int B,
                                       int Cin, int Cout, int Hout, int x_act, int accumulate,
                                       mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(dy && x && dw && B > 0);
  return conv_wgrad_dispatch(dy, x, dw, db, ws, B, Cout, Cin, Hout, MMVAE_ACT_NONE, x_act, 1, accumulate,
                             (hipStream_t)stream);
}
extern "C" int mmvae_convT2d_k4s2_fwd(const float* x, const float* w, const float* bias, const float* aux, float* y,
                                      int B, int Cin, int Cout, int Hin, int in_act, int ep_mode,
                                      mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(x && w && y && B > 0);
  return conv_scatter_dispatch(x, w, bias, aux, y, B, Cin, Cout, Hin, in_act, ep_mode, (hipStream_t)stream);
}
extern "C" int mmvae_convT3_bce_seeded(const float* x, const float* w, const float* bias, const float* target, float* row,
                                       float* dlogit, float* part, unsigned* ticket, int B, int in_act, float seed,
                                       mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(x && w && target && row && dlogit && B > 0);
  MMVAE_CHECK_ARG(in_act == MMVAE_ACT_NONE || in_act == MMVAE_ACT_RELU || in_act == MMVAE_ACT_SILU);
  MMVAE_CHECK_ARG(convT3_strips(B) == 1 || (part && ticket));
  MMVAE_CHECK_ARG((((uintptr_t)target | (uintptr_t)dlogit) & 15) == 0);
  ConvT3Args t{x, w, bias, dlogit, target, row, part, ticket, B, in_act, MMVAE_EP_SIGMOID_CLAMP, seed};
  convT3_launch<true>(t, (hipStream_t)stream);
  return mmvae_launch_status();
}
extern "C" int mmvae_convT2d_k4s2_dgrad(const float* dy, const float* w, const float* aux, float* dx, int B, int Cin,
                                        int Cout, int Hin, int ep_mode, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(dy && w && dx && B > 0);
  // dx[c] = sum_o dy[o] (gathered) * w[c][o]: w is [Cin][Cout] = [out][red]
  return conv_gather_dispatch(dy, w, nullptr, aux, dx, B, Cout, Cin, 2 * Hin, MMVAE_ACT_NONE, ep_mode,
                              (hipStream_t)stream);
}
extern "C" int mmvae_convT2d_k4s2_wgrad(const float* x, const float* dy, float* dw, float* db, float* ws, int B,
                                        int Cin, int Cout, int Hin, int x_act, int accumulate,
                                        mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(x && dy && dw && B > 0);
  return conv_wgrad_dispatch(x, dy, dw, db, ws, B, Cin, Cout, Hin, x_act, MMVAE_ACT_NONE, 2, accumulate,
                             (hipStream_t)stream);
}
extern "C" size_t mmvae_conv_wgrad_ws_floats(int B, int Csmall, int Clarge, int Hsmall) {
  return conv_wgrad_ws_floats(B, Csmall, Clarge, Hsmall);
}
extern "C" int mmvae_conv_wgrad_layout(int B, int Csmall, int Clarge, int Hsmall, int* rows, int* rowlen,
                                       int* bias_col) {
  MMVAE_CHECK_ARG(rows && rowlen && bias_col);
  conv_wgrad_layout(B, Csmall, Clarge, Hsmall, rows, rowlen, bias_col);
  return MMVAE_OK;
}

MMVAE_TRACE_SETTER(conv)

// split_bf16: 1 / 0 sets the GEMM core of the layers both cores cover (conv_common.hpp), < 0 keeps it; returns the old value
extern "C" int mmvae_conv_plan(int split_bf16) {
  const int old = g_conv_split_bf16;
  if (split_bf16 >= 0) g_conv_split_bf16 = split_bf16 ? 1 : 0;
  return old;
}
