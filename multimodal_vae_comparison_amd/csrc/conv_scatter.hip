// "Scatter" form of the 4x4 / stride-2 / pad-1 convolution on fp32 MFMA (gfx950):
//
//     y[b,o,oh,ow] = ep( bias[o] + sum_{c,kh,kw : oh = 2ih-1+kh, ow = 2iw-1+kw} x[b,c,ih,iw] * w[c,o,kh,kw] )
//
// = nn.ConvTranspose2d forward (models/decoders.py:91-95) and nn.Conv2d input-gradient.
// Each output pixel (2r+ph, 2s+pw) sees exactly 2x2 of the 4x4 taps, chosen by its parity class (ph,pw):
//   ph = 0 : (kh=1, ih=r), (kh=3, ih=r-1)        ph = 1 : (kh=0, ih=r+1), (kh=2, ih=r)      (same for columns)
// so the op is four stride-1 2x2-tap convolutions over the INPUT grid, K = Cin*4 each.
//
// Cout = 32 (conv_scatter_kernel): M tile = 32 input-grid positions, four 32x32 accumulators (one per class)
//   per wave on v_mfma_f32_32x32x2_f32; lane half = column tap.  Same TM x KS workgroup shapes as the gather
//   kernel.  The two column classes of a lane are interleaved in registers => 32-byte contiguous stores.
// Cout = 3 (conv_scatter3_kernel): N = 3 channels x 4 classes = 12 <= 16, v_mfma_f32_16x16x4_f32 with a dense
//   K = Cin x (3x3 input offsets); the weight matrix holds zeros where a class does not use an offset (44 %
//   dense) -- cheaper than wasting 29 of 32 MFMA columns, and the layer is bound by its 6.3 MB/128-sample
//   output anyway.  Epilogue fuses bias + sigmoid + clamp (decoders.py:96-97).
#include "conv_common.hpp"

struct ConvScatterArgs {
  const float* x;
  const float* w;
  const float* bias;
  const float* aux;
  float* y;
  int B, Hin, lgW, in_act, ep;
};

template <int TM>
__global__ __launch_bounds__(256) void conv_scatter_kernel(ConvScatterArgs a) {
  constexpr int CIN = 32;
  constexpr int KS = 4 / TM;
  constexpr int CC = CONV_CC;
  constexpr int NCHUNK = CIN / CC;
  constexpr int CPW = CC / KS;
  constexpr int IN_MAX = 4608;  // >= 8 img x 8 ch x 6 x 6 = 2304, 8ch x 10 x 18 = 1440; >= 4096 for the reduce
  __shared__ float s_in[IN_MAX];
  __shared__ float s_w[4 * CC * 4 * CONV_CO];  // [cls][cl][th][tw][o]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int tm = wave % TM, ks = wave / TM;
  const int Hin = a.Hin, Win = a.Hin, lgW = a.lgW;
  const int Hout = 2 * Hin, Wout = 2 * Win;
  const int Mtot = a.B * Hin * Win;  // input-grid positions
  const int rows_per_tile = Win >= 32 ? 1 : 32 / Win;
  const int NR = TM * rows_per_tile;
  const int first_row = blockIdx.x * NR;
  const MacroTile mt = macro_tile(first_row, NR, Hin);
  const int NRin = mt.nrow + 2, RS = Win + 2, CS = NRin * RS, IS = CC * CS;

  // staging slots precomputed once (see conv_gather.hip): region [0, nimg*IS) is contiguous in LDS
  constexpr int MAXSLOT = IN_MAX / 256;
  const int region = mt.nimg * IS;
  int soff[MAXSLOT];
  unsigned svalid = 0;
  const float inv_CS = 1.0f / (float)CS, inv_RS = 1.0f / (float)RS;
#pragma unroll
  for (int sidx = 0; sidx < MAXSLOT; ++sidx) {
    soff[sidx] = 0;
    if (sidx * 256 < region) {
      const int e = tid + 256 * sidx;
      const int pl = fdiv_small(e, inv_CS), rem = e - pl * CS;
      const int img = pl / CC, cl = pl - img * CC;
      const int lr = fdiv_small(rem, inv_RS), col = rem - lr * RS;
      const int ih = mt.h0 - 1 + lr, iw = col - 1;
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
      const int o = e & 31, t = (e >> 5) & 15, cl = e >> 9;
      rw[i] = a.w[((size_t)(ch * CC + cl) * CONV_CO + o) * 16 + t];
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
    // weights: w[(c*32 + o)*16 + kh*4 + kw] -> s_w[(((cls*CC + cl)*2 + th)*2 + tw)*32 + o]
#pragma unroll
    for (int i = 0; i < WSLOT; ++i) {
      const int e = i * 256 + tid;
      const int o = e & 31, t = (e >> 5) & 15, cl = e >> 9;
      const int kh = t >> 2, kw = t & 3;
      const int ph = 1 - (kh & 1), th = kh >> 1, pw = 1 - (kw & 1), tw = kw >> 1;
      const int cls = ph * 2 + pw;
      s_w[(((cls * CC + cl) * 2 + th) * 2 + tw) * CONV_CO + o] = rw[i];
    }
  };

  const int tile_p0 = (blockIdx.x * TM + tm) * 32;
  int abase;
  {
    int p = tile_p0 + li;
    if (p > Mtot - 1) p = Mtot - 1;
    const int R = p >> lgW, s = p & (Win - 1);
    int rl = R - first_row;
    if (rl < 0) rl = 0;
    int img_l = 0, r_l = rl;
    if (mt.nimg > 1) {
      img_l = rl >> lgW;
      r_l = rl & (Hin - 1);
    }
    // A for class (ph,pw), taps (th, tw = lane half): in[c][r_l + 1 + ph - th][s + 1 + pw - lh]
    abase = img_l * IS + (r_l + 1) * RS + (s + 1 - lh) + ks * CPW * CS;
  }
  const int wbase = (ks * CPW * 4 + lh) * CONV_CO + li;  // + ((cls*CC + cc)*2 + th)*2*32

  f32x16 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

  load_chunk(0);
  for (int ch = 0; ch < NCHUNK; ++ch) {
    if (ch > 0) __syncthreads();
    store_chunk();
    __syncthreads();
    if (ch + 1 < NCHUNK) load_chunk(ch + 1);
    // operands of channel cc+1 are read from LDS while the 8 MFMAs (4 classes x 2 row taps) of channel cc issue
    {
      float av[2][8], bv[2][8];
      auto load_ops = [&](int cc, int buf) {
#pragma unroll
        for (int cls = 0; cls < 4; ++cls) {
#pragma unroll
          for (int th = 0; th < 2; ++th) {
            const int ph = cls >> 1, pw = cls & 1;
            av[buf][cls * 2 + th] = s_in[abase + cc * CS + (ph - th) * RS + pw];
            bv[buf][cls * 2 + th] = s_w[wbase + (((cls * CC + cc) * 2 + th) * 2) * CONV_CO];
          }
        }
      };
      load_ops(0, 0);
#pragma unroll
      for (int cc = 0; cc < CPW; ++cc) {
        if (cc + 1 < CPW) load_ops(cc + 1, (cc + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);   // keep the reads above the MFMAs (see conv_gather.hip)
#pragma unroll
        for (int cls = 0; cls < 4; ++cls)
#pragma unroll
          for (int th = 0; th < 2; ++th)
            acc[cls] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cc & 1][cls * 2 + th], bv[cc & 1][cls * 2 + th],
                                                            acc[cls], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  const float bias = a.bias ? a.bias[li] : 0.f;
  // one row parity `ph` of a 4-position group: 8 contiguous outputs = two float4
  auto store8 = [&](int q, int ph, const float* e0, const float* e1) {
    const int p0 = tile_p0 + 8 * q + 4 * lh;
    if (p0 < Mtot) {
      const int R = p0 >> lgW, s = p0 & (Win - 1);
      const int b = R >> lgW, r = R & (Hin - 1);
      const size_t off = (((size_t)b * CONV_CO + li) * Hout + 2 * r + ph) * Wout + 2 * s;
      float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
      if (ep_reads_aux(a.ep)) {
        a0 = *reinterpret_cast<const float4*>(a.aux + off);
        a1 = *reinterpret_cast<const float4*>(a.aux + off + 4);
      }
      float4 o0, o1;
      o0.x = apply_epilogue(e0[0] + bias, a0.x, a.ep);
      o0.y = apply_epilogue(e1[0] + bias, a0.y, a.ep);
      o0.z = apply_epilogue(e0[1] + bias, a0.z, a.ep);
      o0.w = apply_epilogue(e1[1] + bias, a0.w, a.ep);
      o1.x = apply_epilogue(e0[2] + bias, a1.x, a.ep);
      o1.y = apply_epilogue(e1[2] + bias, a1.y, a.ep);
      o1.z = apply_epilogue(e0[3] + bias, a1.z, a.ep);
      o1.w = apply_epilogue(e1[3] + bias, a1.w, a.ep);
      *reinterpret_cast<float4*>(a.y + off) = o0;
      *reinterpret_cast<float4*>(a.y + off + 4) = o1;
    }
  };

  if (KS == 1) {
#pragma unroll
    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float e0[4], e1[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          e0[e] = acc[ph * 2 + 0][4 * q + e];
          e1[e] = acc[ph * 2 + 1][4 * q + e];
        }
        store8(q, ph, e0, e1);
      }
  } else {
    float* red = s_in;  // [wave][16][64], one row parity (2 classes) does not fit => one class per pass
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      float sum[2][4 / KS][4];
#pragma unroll
      for (int pw = 0; pw < 2; ++pw) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[ph * 2 + pw][r];
        __syncthreads();
#pragma unroll
        for (int qq = 0; qq < 4 / KS; ++qq) {
          const int q = ks * (4 / KS) + qq;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float s = 0.f;
#pragma unroll
            for (int k2 = 0; k2 < KS; ++k2) s += red[((k2 * TM + tm) * 16 + 4 * q + e) * 64 + lane];
            sum[pw][qq][e] = s;
          }
        }
      }
#pragma unroll
      for (int qq = 0; qq < 4 / KS; ++qq) store8(ks * (4 / KS) + qq, ph, sum[0][qq], sum[1][qq]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Cout = 3: y (B,3,2H,2H) from x (B,32,H,H), H % 16 == 0.  Workgroup = 4 waves x 2 tiles of 16 positions
// = 128 positions (128/W rows of one image); all 32 input channels staged at once.
// ------------------------------------------------------------------------------------------------
struct ConvScatter3Args {
  const float* x;
  const float* w;  // [32][3][4][4]
  const float* bias;
  float* y;
  int B, Hin, lgW, in_act, ep;
};

__global__ __launch_bounds__(256) void conv_scatter3_kernel(ConvScatter3Args a) {
  constexpr int CIN = 32, CO = 3;
  constexpr int IN_MAX = CIN * 208 + 64;  // CS = 6 rows x 34 cols = 204 -> 208 (== 16 mod 32)
  __shared__ float s_in[IN_MAX];
  __shared__ float s_b[9 * CIN * 16];  // B[k = o9*32 + c][n = co*4 + ph*2 + pw]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, kq = lane >> 4;
  const int Hin = a.Hin, Win = a.Hin, lgW = a.lgW, Hout = 2 * Hin, Wout = 2 * Win;
  const int Mtot = a.B * Hin * Win;
  const int NR = 128 / Win;  // rows of the input grid per workgroup (Win in {16,32,64,128})
  const int first_row = blockIdx.x * NR;
  const int b0 = first_row / Hin, h0 = first_row - b0 * Hin;
  const int NRin = NR + 2, RS = Win + 2;
  int CS = NRin * RS;
  CS += (16 - (CS & 31) + 32) & 31;  // CS == 16 (mod 32): the 4 k-lanes of a 32-lane LDS group hit disjoint banks

  // stage all input channels (zero halo): fixed-trip unrolled loops so that the loads are issued back to back
  {
    const int total = CIN * NRin * RS;
    const float inv_PL = 1.0f / (float)(NRin * RS), inv_RS3 = 1.0f / (float)RS;
    constexpr int NS = (IN_MAX + 255) / 256;
    float rv[NS];
    int lo[NS];
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx) {
      const int e = tid + 256 * sidx;
      const int c = fdiv_small(e, inv_PL), rem = e - c * (NRin * RS);
      const int lr = fdiv_small(rem, inv_RS3), col = rem - lr * RS;
      const int ih = h0 - 1 + lr, iw = col - 1;
      const bool ok = e < total && b0 < a.B && ih >= 0 && ih < Hin && iw >= 0 && iw < Win;
      lo[sidx] = e < total ? c * CS + rem : -1;
      const float v = a.x[ok ? (((size_t)b0 * CIN + c) * Hin + ih) * Win + iw : 0];
      rv[sidx] = ok ? v : 0.f;
    }
    if (a.in_act != MMVAE_ACT_NONE) {
#pragma unroll
      for (int sidx = 0; sidx < NS; ++sidx) rv[sidx] = apply_in_act(rv[sidx], a.in_act);
    }
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx)
      if (lo[sidx] >= 0) s_in[lo[sidx]] = rv[sidx];
  }
  // build the dense B matrix
#pragma unroll
  for (int it = 0; it < 9 * CIN * 16 / 256; ++it) {
    const int e = tid + 256 * it;
    const int n = e & 15, k = e >> 4;
    const int c = k & 31, o9 = k >> 5;
    const int dr = o9 / 3 - 1, ds = o9 % 3 - 1;
    int kh = -1, kw = -1, co = 0;
    if (n < CO * 4) {
      co = n >> 2;
      const int ph = (n >> 1) & 1, pw = n & 1;
      // ph = 0: dr = 0 -> kh 1, dr = -1 -> kh 3 ; ph = 1: dr = +1 -> kh 0, dr = 0 -> kh 2
      if (ph == 0) kh = dr == 0 ? 1 : (dr == -1 ? 3 : -1); else kh = dr == 1 ? 0 : (dr == 0 ? 2 : -1);
      if (pw == 0) kw = ds == 0 ? 1 : (ds == -1 ? 3 : -1); else kw = ds == 1 ? 0 : (ds == 0 ? 2 : -1);
    }
    const bool ok = kh >= 0 && kw >= 0;
    const float v = a.w[ok ? ((size_t)c * CO + co) * 16 + kh * 4 + kw : 0];
    s_b[e] = ok ? v : 0.f;
  }
  __syncthreads();

  f32x4 acc[2];
  int abase[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int pl = (wave * 2 + t) * 16 + i16;  // position within the workgroup's 128
    const int r_l = pl >> lgW, s = pl & (Win - 1);
    abase[t] = kq * CS + (r_l + 1) * RS + s + 1;
  }
  const int bbase = kq * 16 + i16;
#pragma unroll
  for (int o9 = 0; o9 < 9; ++o9) {
    const int dr = o9 / 3 - 1, ds = o9 % 3 - 1;
#pragma unroll
    for (int c0 = 0; c0 < CIN; c0 += 4) {
      const float bv = s_b[bbase + (o9 * CIN + c0) * 16];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float av = s_in[abase[t] + c0 * CS + dr * RS + ds];
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[t], 0, 0, 0);
      }
    }
  }

  // epilogue: lane (n = i16, rows 4*kq + reg).  Column-parity partner lane (n ^ 1) holds the neighbour pixel.
  const int n = i16, co = n >> 2, ph = (n >> 1) & 1, pw = n & 1;
  const float bias = (a.bias && n < CO * 4) ? a.bias[co] : 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    float v[4], pv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] = apply_epilogue(acc[t][r] + bias, 0.f, a.ep);
      pv[r] = __shfl_xor(v[r], 1, 64);
    }
    const int pl = (wave * 2 + t) * 16 + 4 * kq;  // first of this lane's 4 consecutive positions
    const int p = blockIdx.x * 128 + pl;
    if (n < CO * 4 && p < Mtot) {
      const int r_l = pl >> lgW, s = pl & (Win - 1);
      const size_t off = (((size_t)b0 * CO + co) * Hout + 2 * (h0 + r_l) + ph) * Wout + 2 * s;
      // even lane writes positions 0,1 (4 floats), odd lane positions 2,3
      float4 o;
      if (pw == 0) { o.x = v[0]; o.y = pv[0]; o.z = v[1]; o.w = pv[1]; *reinterpret_cast<float4*>(a.y + off) = o; }
      else { o.x = pv[2]; o.y = v[2]; o.z = pv[3]; o.w = v[3]; *reinterpret_cast<float4*>(a.y + off + 4) = o; }
    }
  }
}

// x (B,32,Hin,Hin), w [32][Cout][4][4] -> y (B,Cout,2Hin,2Hin)
int conv_scatter_dispatch(const float* x, const float* w, const float* bias, const float* aux, float* y, int B,
                          int Cred, int Cout, int Hin, int in_act, int ep, hipStream_t st) {
  if (Cred != 32) return MMVAE_ERR_UNSUPPORTED;
  if (Hin < 4 || Hin > 64 || (Hin & (Hin - 1))) return MMVAE_ERR_UNSUPPORTED;
  if (ep_reads_aux(ep) && !aux) return MMVAE_ERR_ARG;
  if (Cout == 3) {
    if ((Hin != 16 && Hin != 32) || ep_reads_aux(ep)) return MMVAE_ERR_UNSUPPORTED;
    ConvScatter3Args a{x, w, bias, y, B, Hin, ilog2i(Hin), in_act, ep};
    const long total = (long)B * Hin * Hin;
    hipLaunchKernelGGL(conv_scatter3_kernel, dim3((unsigned)((total + 127) / 128)), dim3(256), 0, st, a);
    return mmvae_launch_status();
  }
  if (Cout != CONV_CO) return MMVAE_ERR_UNSUPPORTED;
  ConvScatterArgs a{x, w, bias, aux, y, B, Hin, ilog2i(Hin), in_act, ep};
  const int rows_per_tile = Hin >= 32 ? 1 : 32 / Hin;
  const long total_rows = (long)B * Hin;
  const long tiles = ((long)B * Hin * Hin + 31) / 32;
  int TM = 4;
  if (tiles < 512 * 4) TM = 2;
  if (tiles < 512 * 2) TM = 1;
  const int NR = TM * rows_per_tile;
  const unsigned grid = (unsigned)((total_rows + NR - 1) / NR);
  if (TM == 4)
    hipLaunchKernelGGL(conv_scatter_kernel<4>, dim3(grid), dim3(256), 0, st, a);
  else if (TM == 2)
    hipLaunchKernelGGL(conv_scatter_kernel<2>, dim3(grid), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(conv_scatter_kernel<1>, dim3(grid), dim3(256), 0, st, a);
  return mmvae_launch_status();
}
