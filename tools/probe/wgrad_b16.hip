// probe build of the split-bf16 weight-gradient body (csrc/conv_wgrad_b16.inc) beside the fp32-MFMA one
#include "../../multimodal_vae_comparison_amd/csrc/conv_gather.inc"
#include "../../multimodal_vae_comparison_amd/csrc/conv_gather_b16.inc"
#include "../../multimodal_vae_comparison_amd/csrc/conv_wgrad.inc"
#include "conv_wgrad_b16.inc"

// small (B,32,Hs,Hs), large (B,32,2Hs,2Hs) -> partial rows in ws (layout of conv_wgrad_layout); b16 = 0: fp32-MFMA kernel
extern "C" int probe_wgrad(const float* small, const float* large, float* ws, int B, int Hs, int small_act, int large_act,
                           int bias_from, int b16, hipStream_t st) {
  const int P = 32, Q = 32;
  const int n_macro = wgrad_n_macro(B, Hs);
  int nsplit = wgrad_splits(n_macro, Q);
  if (b16 > 1) nsplit = n_macro < b16 ? n_macro : b16;       // probe: another number of position splits
  const long dwlen = (long)P * Q * 16, rowlen = dwlen + wgrad_bias_slots(P, Q);
  ConvWgradArgs a{small, large, ws, B, small_act, large_act, bias_from, n_macro, P, rowlen};
  if (b16) {
    if (Hs == 16) hipLaunchKernelGGL((conv_wgrad_b16_kernel<WgradB16Geom<4>, MMVAE_ACT_NONE, MMVAE_ACT_RELU>), dim3(nsplit, 4, 1), dim3(256), 0, st, a);
    else if (Hs == 8) hipLaunchKernelGGL((conv_wgrad_b16_kernel<WgradB16Geom<3>, MMVAE_ACT_NONE, MMVAE_ACT_RELU>), dim3(nsplit, 4, 1), dim3(256), 0, st, a);
    else return 1;
  } else {
    if (Hs == 16) hipLaunchKernelGGL((conv_wgrad_kernel<WgradGeom<32, 4, 8>, true>), dim3(nsplit, 4, 1), dim3(256), 0, st, a);
    else if (Hs == 8) hipLaunchKernelGGL((conv_wgrad_kernel<WgradGeom<32, 3, 8>, true>), dim3(nsplit, 4, 1), dim3(256), 0, st, a);
    else return 1;
  }
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
extern "C" int probe_wgrad_rows(int B, int Hs, int* rows, int* rowlen, int* bias_col) {
  conv_wgrad_layout(B, 32, 32, Hs, rows, rowlen, bias_col);
  return 0;
}
