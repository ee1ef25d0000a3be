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

  // Every macro tile has the same shape, so the staging index math is done once per thread: slot -> (global
  // offset relative to the tile origin, LDS index, halo flags).  Per tile only the origin and three uniform
  // flags change, and the next tile's loads are issued before the current tile's MFMAs.
  const MacroTile mt0 = macro_tile(0, NR, Hs);
  const int nimg = mt0.nimg, nrow = mt0.nrow;
  const int NRin = 2 * nrow + 2;
  int CS = NRin * RS;
  CS += (16 - (CS & 31) + 32) & 31;  // == 16 (mod 32)
  const int IS = QC * CS;
  const int per_img = nrow * Ws;
  const int MP = nimg * per_img;  // positions per macro tile
  const int PSm = MP + 1;
  const int lgMP = ilog2i(MP);  // MP is a power of two (32, 64 or 128)
  const int plane_n = NRin * RSraw;
  const int lregion = nimg * QC * plane_n;
  const float inv_plane = 1.0f / (float)plane_n, inv_RSraw = 1.0f / (float)RSraw;
  constexpr int SS = S_MAX / 32 * 32 / 256 + 1;  // 17 >= 32*128/256
  constexpr int LS = 20;                         // >= 8 ch * 18 rows * 34 cols / 256
  int s_off[SS], l_off[LS], l_idx[LS];
  unsigned s_ok = 0, s_img1 = 0, l_ok = 0, l_top = 0, l_bot = 0, l_img1 = 0;
#pragma unroll
  for (int i = 0; i < SS; ++i) {
    const int e = tid + 256 * i;
    const int p = e >> lgMP, pos = e - (p << lgMP);
    const int img = pos >= per_img ? 1 : 0, within = pos - img * per_img;   // nimg <= 2
    const bool ok = e < P * MP;
    s_ok |= (ok ? 1u : 0u) << i;
    s_img1 |= ((ok && img >= 1) ? 1u : 0u) << i;
    s_off[i] = ok ? ((img * P + p) * Hs) * Ws + within : 0;
  }
#pragma unroll
  for (int i = 0; i < LS; ++i) {
    const int e = tid + 256 * i;
    const int pl = fdiv_small(e, inv_plane), rem = e - pl * plane_n;
    const int img = pl / QC, qq = pl - img * QC;
    const int lr = fdiv_small(rem, inv_RSraw), col = rem - lr * RSraw;
    const int q = chunk * QC + qq, iw = col - 1;
    const bool in = e < lregion;
    const bool ok = in && q < Q && iw >= 0 && iw < Wl;
    l_ok |= (ok ? 1u : 0u) << i;
    l_top |= ((in && lr == 0) ? 1u : 0u) << i;
    l_bot |= ((in && lr == NRin - 1) ? 1u : 0u) << i;
    l_img1 |= ((in && img >= 1) ? 1u : 0u) << i;
    l_off[i] = ok ? ((img * Q + q) * Hl + lr) * Wl + iw : 0;
    l_idx[i] = in ? img * IS + qq * CS + lr * RS + col : -1;
  }
  float sv[SS], lv[LS];
  unsigned s_val = 0, l_val = 0;
  auto load_tile = [&](int mtile) {
    const MacroTile mt = macro_tile(mtile * NR, NR, Hs);
    const bool img1bad = nimg > 1 && (mt.b0 + 1) >= a.B;   // nimg <= 2 here
    s_val = s_ok & ~(img1bad ? s_img1 : 0u);
    l_val = l_ok & ~(mt.h0 == 0 ? l_top : 0u) & ~(mt.h0 + nrow == Hs ? l_bot : 0u) & ~(img1bad ? l_img1 : 0u);
    const float* sb = a.small + ((size_t)mt.b0 * P * Hs + mt.h0) * Ws;
    const float* lb = a.large + ((long)mt.b0 * Q * Hl + 2 * mt.h0 - 1) * Wl;
#pragma unroll
    for (int i = 0; i < SS; ++i) sv[i] = sb[(s_val >> i & 1u) ? s_off[i] : 0];
#pragma unroll
    for (int i = 0; i < LS; ++i) {
      const long o = (l_val >> i & 1u) ? (long)l_off[i] : (long)(Wl + 1);   // (Wl+1) keeps the dummy address in range
      lv[i] = lb[o];
    }
  };
  auto store_tile = [&]() {
    if (a.small_act != MMVAE_ACT_NONE) {
#pragma unroll
      for (int i = 0; i < SS; ++i) sv[i] = apply_in_act(sv[i], a.small_act);
    }
    if (a.large_act != MMVAE_ACT_NONE) {
#pragma unroll
      for (int i = 0; i < LS; ++i) lv[i] = apply_in_act(lv[i], a.large_act);
    }
#pragma unroll
    for (int i = 0; i < SS; ++i) {
      const int e = tid + 256 * i;
      if (s_ok >> i & 1u) {
        const int p = e >> lgMP;
        s_small[e + p] = (s_val >> i & 1u) ? sv[i] : 0.f;   // p*PSm + pos = e + p
      }
    }
#pragma unroll
    for (int i = 0; i < LS; ++i)
      if (l_idx[i] >= 0) s_large[l_idx[i]] = (l_val >> i & 1u) ? lv[i] : 0.f;
  };

  const int rows_total = nimg * nrow;
  const int r_beg = half * (rows_total / PSPLIT), r_end = r_beg + rows_total / PSPLIT;
  const int bbase = ql * CS + kh * RS + kw + 2 * lh;
  const int abase = li * PSm + lh;

  if ((int)blockIdx.x < a.n_macro) load_tile(blockIdx.x);
  for (int mtile = blockIdx.x; mtile < a.n_macro; mtile += gridDim.x) {
    __syncthreads();  // previous tile's MFMAs are done with LDS
    store_tile();
    __syncthreads();
    if (mtile + (int)gridDim.x < a.n_macro) load_tile(mtile + gridDim.x);
    // Positions are walked in groups of 16 (= 8 MFMAs, k = 2 positions each).  s_small is position-linear, so the
    // A operands of a group are 8 consecutive pairs; the B operand address is decoded per MFMA from the
    // (uniform) position.  Group g+1's operands are read from LDS while group g's MFMAs issue.
    {
      const int pbeg = r_beg * Ws, ngrp = (r_end - r_beg) * Ws / 16;
      float av[2][8], bv[2][8];
      auto load_grp = [&](int gi, int buf) {
        const int p0 = pbeg + 16 * gi;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const int p = p0 + 2 * t;
          const int rr = p >> lgWs, ow0 = p & (Ws - 1);
          const int img = rr >= nrow ? 1 : 0, oh_l = rr - img * nrow;
          av[buf][t] = s_small[abase + p];
          bv[buf][t] = s_large[img * IS + (2 * oh_l) * RS + 2 * ow0 + bbase];
        }
      };
      auto mfma_grp = [&](int buf) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          asum += av[buf][t];
          bsum += bv[buf][t];
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[buf][t], bv[buf][t], acc, 0, 0, 0);
        }
      };
      if (ngrp > 0) load_grp(0, 0);
      for (int gi = 0; gi < ngrp; gi += 2) {
        if (gi + 1 < ngrp) load_grp(gi + 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_grp(0);
        __builtin_amdgcn_sched_barrier(0);
        if (gi + 1 < ngrp) {
          if (gi + 2 < ngrp) load_grp(gi + 2, 0);
          __builtin_amdgcn_sched_barrier(0);
          mfma_grp(1);
          __builtin_amdgcn_sched_barrier(0);
        }
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

void conv_wgrad_layout(int B, int Q, int Hs, int* rows, int* rowlen, int* bias_col) {
  *rows = wgrad_rows(wgrad_n_macro(B, Hs), Q);
  *bias_col = 32 * Q * 16;
  *rowlen = *bias_col + 32;
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
  if (accumulate == MMVAE_ACC_DEFER) return rc;   // partials stay in ws; the caller registers the segments
  const int rows = wgrad_rows(n_macro, Q);
  const long dwlen = (long)32 * Q * 16, rowlen = dwlen + 32;
  rc = mmvae_reduce_rows(ws, dw, rows, dwlen, rowlen, accumulate, st);
  if (rc) return rc;
  if (bias_from) rc = mmvae_reduce_rows(ws + dwlen, db, rows, bias_from == 1 ? 32 : Q, rowlen, accumulate, st);
  return rc;
}
