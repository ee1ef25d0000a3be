// C-ABI entry points of the convolution family (include/mmvae_hip.h) mapped onto the three kernel forms.
#include "conv_common.hpp"

int conv_gather_dispatch(const float* x, const float* w, const float* bias, const float* aux, float* y, int B,
                         int Cred, int Cout, int Hin, int in_act, int ep, hipStream_t st);
int conv_scatter_dispatch(const float* x, const float* w, const float* bias, const float* aux, float* y, int B,
                          int Cred, int Cout, int Hin, int in_act, int ep, hipStream_t st);
int conv_wgrad_dispatch(const float* small, const float* large, float* dw, float* db, float* ws, int B, int P, int Q,
                        int Hs, int small_act, int large_act, int bias_from, int accumulate, hipStream_t st);
size_t conv_wgrad_ws_floats(int B, int Q, int Hs);
void conv_wgrad_layout(int B, int Q, int Hs, int* rows, int* rowlen, int* bias_col);

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
  (void)Csmall;
  return conv_wgrad_ws_floats(B, Clarge, Hsmall);
}
extern "C" int mmvae_conv_wgrad_layout(int B, int Csmall, int Clarge, int Hsmall, int* rows, int* rowlen,
                                       int* bias_col) {
  (void)Csmall;
  MMVAE_CHECK_ARG(rows && rowlen && bias_col);
  conv_wgrad_layout(B, Clarge, Hsmall, rows, rowlen, bias_col);
  return MMVAE_OK;
}
