// Timing probe (not part of the product): the gather convolution with pieces compiled out (-DCONV_PROBE=bits).
#include "../../multimodal_vae_comparison_amd/csrc/conv_common.hpp"
#include "../../multimodal_vae_comparison_amd/csrc/conv_gather.inc"
extern "C" int probe_gather_stamps(const float* x, const float* w, const float* b, float* y, long long* stamps, int B,
                                   int Hin, int plan, int act, void* stream) {
  return conv_gather_dispatch(x, w, b, reinterpret_cast<const float*>(stamps), y, B, 32, 32, Hin, act, 0,
                              (hipStream_t)stream, plan);
}
extern "C" int probe_gather(const float* x, const float* w, const float* b, float* y, int B, int Hin, int plan, int act,
                            void* stream) {
  return conv_gather_dispatch(x, w, b, nullptr, y, B, 32, 32, Hin, act, 0, (hipStream_t)stream, plan);
}
