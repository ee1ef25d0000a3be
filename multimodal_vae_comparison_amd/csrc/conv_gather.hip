// "Gather" form of the 4x4 / stride-2 / pad-1 convolution as an implicit GEMM on fp32 MFMA (gfx950):
//
//     y[b,o,oh,ow] = ep( bias[o] + sum_{c,kh,kw} act(x[b,c,2oh-1+kh,2ow-1+kw]) * w[o,c,kh,kw] )
//
// = nn.Conv2d forward (models/encoders.py:214-217) and nn.ConvTranspose2d input-gradient.
// GEMM view: M = B*Hout*Wout output pixels, N = 32 output channels (one v_mfma_f32_32x32x2_f32 column
// block), K = Cin*16.  One wavefront owns a 32-pixel x 32-channel accumulator (16 VGPRs).
//
// Workgroup = 4 waves = TM pixel tiles x KS K-slices (TM*KS = 4): layers with few pixels split K across the
// waves and reduce through LDS so that a batch-128 step still fills 256 CUs.
// Per K chunk of 8 input channels the workgroup stages (a) the input rows its pixels touch, including the
// zero halo, activation applied once per staged element, as [img][c][row][col] and (b) the weight slice as
// [k][o] with pitch 33, then every wave issues 64/KS MFMAs whose operands are single ds_read_b32 with
// immediate offsets:  A[i = pixel][k] = in[c][2oh+kh][2ow+kw]  (lane half selects kw parity),
//                     B[k][j = o]     = w_lds[k][o].
#include "conv_common.hpp"

struct ConvGatherArgs {
  const float* x;
  const float* w;
  const float* bias;
  const float* aux;
  float* y;
  int B, Hin, Hout, lgW, in_act, ep;
};

template <int CIN, int TM>
__global__ __launch_bounds__(256) void conv_gather_kernel(ConvGatherArgs a) {
  constexpr int KS = 4 / TM;
  constexpr int CC = CIN < CONV_CC ? CIN : CONV_CC;
  constexpr int NCHUNK = CIN / CC;
  constexpr int CPW = CC / KS;         // channels of a chunk handled by one wave
  constexpr int WP = CONV_CO + 1;      // weight LDS pitch
  constexpr int IN_MAX = 6400;         // >= max staged input floats (8 img x 8 ch x 10 x 10) and >= 4096 (reduce)
  static_assert(CC % KS == 0, "K split must divide the channel chunk");
  __shared__ float s_in[IN_MAX];
  __shared__ float s_w[CC * 16 * WP];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int tm = wave % TM, ks = wave / TM;
  const int Hin = a.Hin, Win = a.Hin, Hout = a.Hout, Wout = a.Hout, lgW = a.lgW;
  const int Mtot = a.B * Hout * Wout;
  const int rows_per_tile = Wout >= 32 ? 1 : 32 / Wout;
  const int NR = TM * rows_per_tile;
  const int first_row = blockIdx.x * NR;
  const MacroTile mt = macro_tile(first_row, NR, Hout);
  const int NRin = 2 * mt.nrow + 2, RS = Win + 2, CS = NRin * RS, IS = CC * CS;

  // Per-thread staging slots.  The chunk's LDS region [0, nimg*IS) is contiguous ([img][c][row][col]); thread t
  // owns elements t, t+256, ...  Their global offsets (relative to the chunk's first channel of image b0) and
  // validity never change across chunks, so the index math (two runtime divisions per slot) is done once and
  // every chunk's loads are issued back to back with nothing but an address add in between.
  constexpr int MAXSLOT = IN_MAX / 256;
  const int region = mt.nimg * IS;
  int soff[MAXSLOT];
  unsigned svalid = 0;
  const float inv_CS = 1.0f / (float)CS, inv_RS = 1.0f / (float)RS;
#pragma unroll
  for (int sidx = 0; sidx < MAXSLOT; ++sidx) {
    soff[sidx] = 0;
    if (sidx * 256 < region) {   // uniform: slots beyond the region cost nothing
      const int e = tid + 256 * sidx;
      const int pl = fdiv_small(e, inv_CS), rem = e - pl * CS;
      const int img = pl / CC, cl = pl - img * CC;
      const int lr = fdiv_small(rem, inv_RS), col = rem - lr * RS;
      const int ih = 2 * mt.h0 - 1 + lr, iw = col - 1;
      const bool ok = e < region && (mt.b0 + img) < a.B && ih >= 0 && ih < Hin && iw >= 0 && iw < Win;
      svalid |= (ok ? 1u : 0u) << sidx;
      soff[sidx] = ok ? ((img * CIN + cl) * Hin + ih) * Win + iw : 0;
    }
  }
  const float* xbase = a.x + (size_t)mt.b0 * CIN * Hin * Win;
  constexpr int WSLOT = CC * 16 * CONV_CO / 256;
  float rv[MAXSLOT], rw[WSLOT];
  auto load_chunk = [&](int ch) {
    const float* xch = xbase + (size_t)ch * CC * Hin * Win;
#pragma unroll
    for (int sidx = 0; sidx < MAXSLOT; ++sidx)
      if (tid + 256 * sidx < region) rv[sidx] = xch[soff[sidx]];
#pragma unroll
    for (int i = 0; i < WSLOT; ++i) {
      const int e = i * 256 + tid;
      const int kl = e % (CC * 16), o = e / (CC * 16);
      rw[i] = a.w[((size_t)o * CIN + ch * CC) * 16 + kl];
    }
  };
  auto store_chunk = [&]() {
    if (a.in_act != MMVAE_ACT_NONE) {
#pragma unroll
      for (int sidx = 0; sidx < MAXSLOT; ++sidx) rv[sidx] = apply_in_act(rv[sidx], a.in_act);
    }
#pragma unroll
    for (int sidx = 0; sidx < MAXSLOT; ++sidx) {
      const int e = tid + 256 * sidx;
      if (e < region) s_in[e] = (svalid >> sidx & 1u) ? rv[sidx] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < WSLOT; ++i) {
      const int e = i * 256 + tid;
      const int kl = e % (CC * 16), o = e / (CC * 16);
      s_w[kl * WP + o] = rw[i];
    }
  };

  // per-lane A-operand base
  const int tile_p0 = (blockIdx.x * TM + tm) * 32;
  int abase;
  {
    int p = tile_p0 + li;
    if (p > Mtot - 1) p = Mtot - 1;
    const int R = p >> lgW, ow = p & (Wout - 1);
    int rl = R - first_row;
    if (rl < 0) rl = 0;
    int img_l = 0, oh_l = rl;
    if (mt.nimg > 1) {
      img_l = rl >> lgW;  // Hout == Wout
      oh_l = rl & (Hout - 1);
    }
    abase = img_l * IS + (2 * oh_l) * RS + 2 * ow + lh + ks * CPW * CS;
  }
  const int wbase = (ks * CPW * 16 + lh) * WP + li;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  load_chunk(0);
  for (int ch = 0; ch < NCHUNK; ++ch) {
    if (ch > 0) __syncthreads();
    store_chunk();
    __syncthreads();
    if (ch + 1 < NCHUNK) load_chunk(ch + 1);  // next chunk's loads fly under this chunk's MFMAs
    // ---- MFMAs: operands of channel cc+1 are read from LDS while the 8 MFMAs of channel cc issue ----
    {
      float av[2][8], bv[2][8];
      auto load_ops = [&](int cc, int buf) {
#pragma unroll
        for (int kh = 0; kh < 4; ++kh) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            av[buf][kh * 2 + j] = s_in[abase + cc * CS + kh * RS + 2 * j];
            bv[buf][kh * 2 + j] = s_w[wbase + (cc * 16 + kh * 4 + 2 * j) * WP];
          }
        }
      };
      load_ops(0, 0);
#pragma unroll
      for (int cc = 0; cc < CPW; ++cc) {
        if (cc + 1 < CPW) load_ops(cc + 1, (cc + 1) & 1);
        // keep the reads above the MFMAs: hipcc's scheduler otherwise sinks every ds_read next to its consumer
        // (one LDS round trip exposed per MFMA pair)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 8; ++t)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cc & 1][t], bv[cc & 1][t], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  // ---- epilogue ----
  const float bias = a.bias ? a.bias[li] : 0.f;
  auto store4 = [&](int q, float v0, float v1, float v2, float v3) {
    const int p0 = tile_p0 + 8 * q + 4 * lh;
    if (p0 < Mtot) {
      const int R = p0 >> lgW, ow = p0 & (Wout - 1);
      const int b = R >> lgW, oh = R & (Hout - 1);
      const size_t off = (((size_t)b * CONV_CO + li) * Hout + oh) * Wout + ow;
      float4 av = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ep_reads_aux(a.ep)) av = *reinterpret_cast<const float4*>(a.aux + off);
      float4 o;
      o.x = apply_epilogue(v0 + bias, av.x, a.ep);
      o.y = apply_epilogue(v1 + bias, av.y, a.ep);
      o.z = apply_epilogue(v2 + bias, av.z, a.ep);
      o.w = apply_epilogue(v3 + bias, av.w, a.ep);
      *reinterpret_cast<float4*>(a.y + off) = o;
    }
  };
  if (KS == 1) {
#pragma unroll
    for (int q = 0; q < 4; ++q) store4(q, acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
  } else {
    __syncthreads();
    float* red = s_in;
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
#pragma unroll
    for (int qq = 0; qq < 4 / KS; ++qq) {
      const int q = ks * (4 / KS) + qq;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float s = 0.f;
#pragma unroll
        for (int k2 = 0; k2 < KS; ++k2) s += red[((k2 * TM + tm) * 16 + 4 * q + e) * 64 + lane];
        v[e] = s;
      }
      store4(q, v[0], v[1], v[2], v[3]);
    }
  }
}

template <int CIN>
static int launch_gather(const ConvGatherArgs& a, hipStream_t st) {
  const int Hout = a.Hout;
  const int rows_per_tile = Hout >= 32 ? 1 : 32 / Hout;
  const long total_rows = (long)a.B * Hout;
  const long tiles = ((long)a.B * Hout * Hout + 31) / 32;
  // TM tiles per workgroup: keep >= ~256 workgroups when possible
  int TM = 4;
  if (CIN >= CONV_CC) {
    if (tiles < 512 * 4) TM = 2;
    if (tiles < 512 * 2) TM = 1;
  }
  const int NR = TM * rows_per_tile;
  const unsigned grid = (unsigned)((total_rows + NR - 1) / NR);
  if (TM == 4)
    hipLaunchKernelGGL((conv_gather_kernel<CIN, 4>), dim3(grid), dim3(256), 0, st, a);
  else if (TM == 2)
    hipLaunchKernelGGL((conv_gather_kernel<CIN, (CIN >= CONV_CC ? 2 : 4)>), dim3(grid), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((conv_gather_kernel<CIN, (CIN >= CONV_CC ? 1 : 4)>), dim3(grid), dim3(256), 0, st, a);
  return mmvae_launch_status();
}

// x (B,Cred,Hin,Hin), w [32][Cred][4][4] -> y (B,32,Hin/2,Hin/2)
int conv_gather_dispatch(const float* x, const float* w, const float* bias, const float* aux, float* y, int B,
                         int Cred, int Cout, int Hin, int in_act, int ep, hipStream_t st) {
  if (Cout != CONV_CO) return MMVAE_ERR_UNSUPPORTED;
  const int Hout = Hin / 2;
  if (Hin < 8 || Hin > 64 || (Hin & (Hin - 1))) return MMVAE_ERR_UNSUPPORTED;
  if (ep_reads_aux(ep) && !aux) return MMVAE_ERR_ARG;
  ConvGatherArgs a{x, w, bias, aux, y, B, Hin, Hout, ilog2i(Hout), in_act, ep};
  if (Cred == 32) return launch_gather<32>(a, st);
  if (Cred == 3) return launch_gather<3>(a, st);
  return MMVAE_ERR_UNSUPPORTED;
}
