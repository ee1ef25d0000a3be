// Weight gradient of the 4x4 / stride-2 / pad-1 convolutions on fp32 MFMA (gfx950):
//
//     dw[p,q,kh,kw] = sum_{b,oh,ow} small[b,p,oh,ow] * act(large[b,q,2oh-1+kh,2ow-1+kw])
//
// nn.Conv2d:          small = dy (Cout = 32 ch), large = x  (Cin ch)  -> dw[o,c,kh,kw], db[o] = sum small
// nn.ConvTranspose2d: small = x  (Cin  = 32 ch), large = dy (Cout ch) -> dw[c,o,kh,kw], db[o] = sum large
//
// GEMM view: M = 32 small channels, N = Q*16 (large channel, tap), K = B*Hs*Ws positions (long reduction).
// A[i = p][k = pos]  = small_lds[p][pos]                      pitch MP+1  (== 1 mod 32: conflict free)
// B[k = pos][j]      = large_lds[q][2oh+kh][2ow+kw], j = (q_local, kh, kw); row pitch == 4 and channel pitch
//                      == 16 (mod 32) => the 32 lanes of a half-wave hit 32 distinct banks.
// grid.y = chunk of 8 large channels (4 N tiles = one per wave).  grid.x = position splits; every workgroup
// walks its macro tiles (128 positions) accumulating in registers and writes ONE partial dw to the
// workspace; mmvae_reduce_rows sums the partials (deterministic; float atomics would run at 1.3 TB/s and
// reorder sums).  Bias gradients ride along as per-lane VALU sums of the operands already in registers.
#include "conv_common.hpp"

struct ConvWgradArgs {
  const float* small;
  const float* large;
  float* ws;
  int B, Hs, lgWs, small_act, large_act, bias_from, n_macro;
};

template <int Q>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(ConvWgradArgs a) {
  constexpr int P = 32;
  constexpr int QC = Q >= 8 ? 8 : 4;              // large channels staged per workgroup
  constexpr int NT = QC / 2;                      // N tiles (2 channels x 16 taps each)
  constexpr int PSPLIT = 4 / NT;                  // position halves when fewer than 4 N tiles
  constexpr int L_MAX = 5888, S_MAX = 32 * 129;
  constexpr long ROWLEN = (long)P * Q * 16 + 32;  // one partial: dw then 32 bias slots
  __shared__ float s_large[L_MAX];
  __shared__ float s_small[S_MAX];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int nt = wave % NT, half = wave / NT;
  const int chunk = blockIdx.y;
  const int Hs = a.Hs, Ws = a.Hs, lgWs = a.lgWs, Hl = 2 * Hs, Wl = 2 * Ws;
  const int NR = Hs >= 32 ? 128 / Ws : 8;  // small rows per macro tile
  const int RSraw = Wl + 2;
  const int RS = Wl <= 32 ? 36 : 68;  // == 4 (mod 32)

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float asum = 0.f, bsum = 0.f;

  const int ql = nt * 2 + (li >> 4), kh = (li >> 2) & 3, kw = li & 3;

  for (int mtile = blockIdx.x; mtile < a.n_macro; mtile += gridDim.x) {
    const int first_row = mtile * NR;
    const MacroTile mt = macro_tile(first_row, NR, Hs);
    const int NRin = 2 * mt.nrow + 2;
    int CS = NRin * RS;
    CS += (16 - (CS & 31) + 32) & 31;  // == 16 (mod 32)
    const int IS = QC * CS;
    const int MP = mt.nimg * mt.nrow * Ws;  // positions in this macro tile
    const int PSm = MP + 1;

    __syncthreads();  // previous tile's MFMAs are done with LDS
    // ---- stage small: s_small[p*PSm + pos] ----
    {
      const int per_img = mt.nrow * Ws;
      for (int e = tid; e < P * MP; e += 256) {
        const int p = e / MP, pos = e - p * MP;
        const int img = pos / per_img, within = pos - img * per_img;
        const int b = mt.b0 + img;
        float v = 0.f;
        if (b < a.B) v = apply_in_act(a.small[((size_t)(b * P + p) * Hs + mt.h0) * Ws + within], a.small_act);
        s_small[p * PSm + pos] = v;
      }
    }
    // ---- stage large planes with zero halo ----
    {
      const int plane_n = NRin * RSraw;
      for (int img = 0; img < mt.nimg; ++img) {
        const int b = mt.b0 + img;
        for (int qq = 0; qq < QC; ++qq) {
          const int q = chunk * QC + qq;
          const float* plane = a.large + ((size_t)(b * Q + q)) * Hl * Wl;
          float* dst = s_large + img * IS + qq * CS;
          for (int e = tid; e < plane_n; e += 256) {
            const int lr = e / RSraw, col = e - lr * RSraw;
            const int ih = 2 * mt.h0 - 1 + lr, iw = col - 1;
            float v = 0.f;
            if (q < Q && b < a.B && ih >= 0 && ih < Hl && iw >= 0 && iw < Wl)
              v = apply_in_act(plane[ih * Wl + iw], a.large_act);
            dst[lr * RS + col] = v;
          }
        }
      }
    }
    __syncthreads();
    // ---- MFMAs over this wave's share of the positions ----
    const int rows_total = mt.nimg * mt.nrow;
    const int r_beg = half * (rows_total / PSPLIT), r_end = r_beg + rows_total / PSPLIT;
    const int bbase = ql * CS + kh * RS + kw + 2 * lh;
    const int abase = li * PSm + lh;
    for (int rr = r_beg; rr < r_end; ++rr) {
      const int img = rr / mt.nrow, oh_l = rr - img * mt.nrow;
      const float* brow = s_large + img * IS + (2 * oh_l) * RS + bbase;
      const float* arow = s_small + abase + rr * Ws;
#pragma unroll 4
      for (int ow0 = 0; ow0 < Ws; ow0 += 2) {
        const float av = arow[ow0];
        const float bv = brow[2 * ow0];
        asum += av;
        bsum += bv;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
      }
    }
  }

  // ---- write this wave's partial ----
  float* row = a.ws + ((size_t)blockIdx.x * PSPLIT + half) * ROWLEN;
  const int ncol = chunk * QC * 16 + nt * 32 + li;
  if (ncol < Q * 16) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int p = (r & 3) + 8 * (r >> 2) + 4 * lh;
      row[(size_t)p * (Q * 16) + ncol] = acc[r];
    }
  }
  if (a.bias_from == 1) {
    const float v = asum + __shfl_xor(asum, 32, 64);
    if (chunk == 0 && nt == 0 && lh == 0) row[(size_t)P * Q * 16 + li] = v;
  } else if (a.bias_from == 2) {
    // every interior pixel of `large` is read exactly once by the taps (kh,kw) in {1,2}^2
    float v = ((kh == 1 || kh == 2) && (kw == 1 || kw == 2)) ? bsum : 0.f;
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 32, 64);
    const int q = chunk * QC + ql;
    if ((lane & 47) == 0 && q < Q) row[(size_t)P * Q * 16 + q] = v;  // lanes 0 and 16
  }
}

static int wgrad_splits(int n_macro, int Q) {
  const int target = Q >= 8 ? 64 : 256;
  return n_macro < target ? n_macro : target;
}
static int wgrad_rows(int n_macro, int Q) { return wgrad_splits(n_macro, Q) * (Q >= 8 ? 1 : 2); }
static int wgrad_n_macro(int B, int Hs) {
  const int NR = Hs >= 32 ? 128 / Hs : 8;
  return (int)(((long)B * Hs + NR - 1) / NR);
}

size_t conv_wgrad_ws_floats(int B, int Q, int Hs) {
  return (size_t)wgrad_rows(wgrad_n_macro(B, Hs), Q) * ((size_t)32 * Q * 16 + 32);
}

// small (B,32,Hs,Hs), large (B,Q,2Hs,2Hs) -> dw [32][Q][4][4], db (bias_from 1: 32 entries, 2: Q entries)
int conv_wgrad_dispatch(const float* small, const float* large, float* dw, float* db, float* ws, int B, int P, int Q,
                        int Hs, int small_act, int large_act, int bias_from, int accumulate, hipStream_t st) {
  if (P != 32 || (Q != 32 && Q != 3)) return MMVAE_ERR_UNSUPPORTED;
  if (Hs < 4 || Hs > 32 || (Hs & (Hs - 1))) return MMVAE_ERR_UNSUPPORTED;
  if (!ws) return MMVAE_ERR_ARG;
  if (!db) bias_from = 0;
  const int n_macro = wgrad_n_macro(B, Hs);
  const int nsplit = wgrad_splits(n_macro, Q);
  ConvWgradArgs a{small, large, ws, B, Hs, ilog2i(Hs), small_act, large_act, bias_from, n_macro};
  if (Q == 32)
    hipLaunchKernelGGL(conv_wgrad_kernel<32>, dim3(nsplit, 4), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(conv_wgrad_kernel<3>, dim3(nsplit, 1), dim3(256), 0, st, a);
  int rc = mmvae_launch_status();
  if (rc) return rc;
  const int rows = wgrad_rows(n_macro, Q);
  const long dwlen = (long)32 * Q * 16, rowlen = dwlen + 32;
  rc = mmvae_reduce_rows(ws, dw, rows, dwlen, rowlen, accumulate, st);
  if (rc) return rc;
  if (bias_from) rc = mmvae_reduce_rows(ws + dwlen, db, rows, bias_from == 1 ? 32 : Q, rowlen, accumulate, st);
  return rc;
}
