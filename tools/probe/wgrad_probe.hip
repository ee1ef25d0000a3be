// Timing probe (not part of the product): the weight-gradient body with clock64 stamps per phase (-DWGRAD_PROBE).
#define WGRAD_PROBE
#include "../../multimodal_vae_comparison_amd/csrc/conv_common.hpp"
#include "../../multimodal_vae_comparison_amd/csrc/conv_wgrad.inc"
// (the deferred mode never reduces; the product's own copy of the dispatcher must not be picked up instead of this one)
extern "C" int mmvae_reduce_rows(const float*, float*, int, long, long, int, mmvae_stream_t) { return MMVAE_ERR_UNSUPPORTED; }
extern "C" int probe_wgrad(const float* dy, const float* x, float* dw, float* db, float* ws, long long* stamps, int B,
                           int Q, int Hs, int x_act, void* stream) {
  if (hipMemcpyToSymbol(HIP_SYMBOL(wgrad_stamp_buf), &stamps, sizeof(stamps)) != hipSuccess) return -1;
  return conv_wgrad_dispatch(dy, x, dw, db, ws, B, 32, Q, Hs, MMVAE_ACT_NONE, x_act, 1, MMVAE_ACC_DEFER, (hipStream_t)stream);
}
