// probe build of the split-bf16 gather convolution (csrc/conv_gather_b16.inc) beside the fp32-MFMA one of the library
#include "../../multimodal_vae_comparison_amd/csrc/conv_gather.inc"
#include "../../multimodal_vae_comparison_amd/csrc/conv_gather_b16.inc"

template <int LGH, int TM, int CC, int CIN = 32>
static int launch(const ConvGatherArgs& a, int Cout, hipStream_t st) {
  using G = GatherB16Geom<CIN, LGH, TM, CC>;
  hipLaunchKernelGGL((conv_gather_b16p_kernel<G, MMVAE_ACT_RELU>), dim3(a.B * G::HOUT / G::NR, Cout / 32), dim3(256), 0, st, a);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
// geom = 10 * CC-code (1: 8 channels per chunk, 2: 4) + TM
extern "C" int probe_conv_b16(const float* x, const float* w, const float* bias, const float* aux, float* y, int B,
                              int Cin, int Cout, int Hin, int in_act, int ep, int geom, hipStream_t st) {
  ConvGatherArgs a{x, w, bias, aux, y, B, in_act, ep, Cout};
  if (Cout % 32 || in_act != MMVAE_ACT_RELU) return 1;
  if (Cin == 3 && Hin == 64) return launch<6, 4, 3, 3>(a, Cout, st);
  if (Cin != 32) return 1;
  if (Hin == 32) {
    if (geom == 12) return launch<5, 2, 8>(a, Cout, st);
    if (geom == 14) return launch<5, 4, 8>(a, Cout, st);
    if (geom == 22) return launch<5, 2, 4>(a, Cout, st);
    if (geom == 24) return launch<5, 4, 4>(a, Cout, st);
  } else if (Hin == 16) {
    if (geom == 11) return launch<4, 1, 8>(a, Cout, st);
    if (geom == 12) return launch<4, 2, 8>(a, Cout, st);
    if (geom == 21) return launch<4, 1, 4>(a, Cout, st);
    if (geom == 22) return launch<4, 2, 4>(a, Cout, st);
  }
  return 1;
}
extern "C" int probe_conv_f32(const float* x, const float* w, const float* bias, const float* aux, float* y, int B,
                              int Cin, int Cout, int Hin, int in_act, int ep, int plan, hipStream_t st) {
  return conv_gather_dispatch(x, w, bias, aux, y, B, Cin, Cout, Hin, in_act, ep, st, plan);
}
