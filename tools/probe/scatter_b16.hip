// probe build of the split-bf16 scatter convolution (csrc/conv_scatter_b16.inc) beside the fp32-MFMA one
#include "../../multimodal_vae_comparison_amd/csrc/conv_gather.inc"
#include "../../multimodal_vae_comparison_amd/csrc/conv_gather_b16.inc"
#include "../../multimodal_vae_comparison_amd/csrc/conv_scatter.inc"
#include "../../multimodal_vae_comparison_amd/csrc/conv_scatter_b16.inc"
// b16 = 1: split-bf16 kernel, 0: fp32-MFMA kernel (plan chosen by the library's rule)
extern "C" int probe_scatter(const float* x, const float* w, const float* bias, const float* aux, float* y, int B, int Cred,
                             int Cout, int Hin, int in_act, int ep, int b16, hipStream_t st) {
  ConvScatterArgs a{x, w, bias, aux, y, B, in_act, ep, Cout};
  const unsigned ngrp = Cout / 32;
  const long tiles = ((long)B * Hin * Hin + 31) / 32 * ngrp;
  if (b16) {
    bool ok = false;
    if (Hin == 16 && Cred == 32) ok = conv_scatter_b16_launch_geom<ScatterB16Geom<32, 4>>(a, ngrp, st);
    else if (Hin == 8 && Cred == 32) ok = conv_scatter_b16_launch_geom<ScatterB16Geom<32, 3>>(a, ngrp, st);
    else if (Hin == 8 && Cred == 64) ok = conv_scatter_b16_launch_geom<ScatterB16Geom<64, 3>>(a, ngrp, st);
    if (!ok) return 1;
    return hipGetLastError() == hipSuccess ? 0 : 2;
  }
  const int plan = scatter_plan(tiles, Cred, Hin);
  const bool ok = scatter_visit(Hin, plan, [&](auto g) {
    using G = decltype(g);
    hipLaunchKernelGGL((conv_scatter_kernel<G>), dim3(scatter_grid(B, Hin, G::TM), ngrp), dim3(256), 0, st, a);
  }, Cred);
  return ok ? 0 : 1;
}
