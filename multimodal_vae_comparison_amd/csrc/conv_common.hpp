// Shared pieces of the 4x4 / stride-2 / pad-1 implicit-GEMM convolution kernels (gfx950).
#pragma once
#include "common.hpp"

#define CONV_CO 32   // MFMA N tile = all 32 output channels of the towers
#define CONV_CC 8    // input channels staged per LDS chunk (K = 128 per chunk)

// Which GEMM core serves the layers both cores cover: the split-bf16 bodies (conv_*_b16.inc; default) or the fp32-MFMA
// bodies everywhere (mmvae_conv_plan(0): the tests compare the two on the same inputs in one process).  A runtime switch
// of the dispatchers; there are no compile-time variants of the product kernels.
inline int g_conv_split_bf16 = 1;
static inline bool conv_split_bf16_enabled() { return g_conv_split_bf16 != 0; }

__host__ __device__ __forceinline__ int ilog2i(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

// Exact floor(e / d) for 0 <= e < 2^20 and small d through one float multiply (a runtime integer division costs
// ~40 VALU instructions; the staging-slot decode below does a few of them per slot).  (e + 0.5) / d is at least
// 0.5 / d away from an integer, far above the float rounding error at these magnitudes.
__device__ __forceinline__ int fdiv_small(int e, float inv_d) { return (int)(((float)e + 0.5f) * inv_d); }

// Geometry of one workgroup's macro tile: `nr` consecutive global rows (row = b*H + h) of an H x H map.
// Either a slice of rows inside one image (nr <= H) or `nimg` whole images (nr = nimg * H).
struct MacroTile {
  int b0;     // first image
  int h0;     // first row inside the image (0 when whole images)
  int nrow;   // rows per image in this tile
  int nimg;   // images in this tile
};
__device__ __forceinline__ MacroTile macro_tile(int first_row, int nr, int H) {
  MacroTile t;
  t.b0 = first_row / H;
  if (nr <= H) {
    t.h0 = first_row - t.b0 * H;
    t.nrow = nr;
    t.nimg = 1;
  } else {
    t.h0 = 0;
    t.nrow = H;
    t.nimg = nr / H;
  }
  return t;
}

// epilogues the convolution towers use (no GELU: keeps erff out of the conv kernels)
__device__ __forceinline__ float conv_epilogue(float v, float aux_v, int ep) {
  switch (ep) {
    case MMVAE_EP_RELU: return fmaxf(v, 0.0f);
    case MMVAE_EP_MUL_RELU_MASK: return aux_v > 0.0f ? v : 0.0f;
    case MMVAE_EP_MUL_SILU_GRAD: return v * dev_silu_grad(aux_v);
    case MMVAE_EP_SIGMOID_CLAMP: return fminf(fmaxf(dev_sigmoid(v), 1e-6f), 1.0f - 1e-6f);
    case MMVAE_EP_SIGMOID: return dev_sigmoid(v);
    default: return v;
  }
}
__host__ __device__ __forceinline__ bool conv_ep_supported(int ep) {
  return ep == MMVAE_EP_NONE || ep == MMVAE_EP_RELU || ep == MMVAE_EP_MUL_RELU_MASK || ep == MMVAE_EP_MUL_SILU_GRAD ||
         ep == MMVAE_EP_SIGMOID_CLAMP || ep == MMVAE_EP_SIGMOID;
}

// Workgroups are handed to the 8 XCDs round robin (physical id p runs on XCD p % 8, each with its own L2).  Consecutive
// tiles share two of their ten staged input rows (and the four tiles of an image all of them, one row apart): the remap
// gives every XCD a CONTIGUOUS eighth of the tiles, so a halo row is an L2 hit instead of a second HBM fetch.
__device__ __forceinline__ int xcd_contiguous(unsigned p, unsigned n) {
  return (n & 7u) ? (int)p : (int)((p & 7u) * (n >> 3) + (p >> 3));
}
