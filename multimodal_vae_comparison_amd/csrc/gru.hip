// Enc_TxtRNN (reference models/encoders.py:840-869): Embedding -> bidirectional one-layer GRU(512) -> output[-1] ->
// sum of the two directions -> Linear -> chunk -> softmax + eta.  A DEFINED path (the reference's forward crashes on its
// own batch format): see oracle/mmvae_oracle.py: enc_txt_rnn for what is computed and why.
//
// MI355X mapping.  The input side of every gate is a table look-up: x_t is an embedding row, so
// W_i x_t = (W_i E^T)[:, id_t] -- ONE (3H x V) GEMM per step of training (V = 27) instead of a (T B x H x 3H) one; the
// gate kernels read Pt[j][id].  The recurrence h_t = cell(x_t, h_{t-1}) is T dependent steps of a (B x H) x (H x 3H)
// product: one launch per time step (gru_step_fwd_kernel: the product on v_mfma_f32_16x16x4_f32 tiles straight from
// L2-resident operands -- W_hh is 3 MB and re-read every step --, the three gate pre-activations of one (row, unit) in
// one lane, gates and the state update in the epilogue).  A persistent kernel would need a grid-wide barrier per step
// (the whole batch's h feeds every output tile): 4 - 5 us on this chip (MI355X_MICROARCH.md, barrier-xcd), more than the
// launch boundary it would replace.  `output[-1]` makes the reverse direction a SINGLE cell step from h = 0 on the last
// token (gru_cell0_*): elementwise.  Backward = T steps of [gate derivatives (elementwise) + dh_{t-1} += dGh W_hh
// (mmvae_linear_bwd_data, accumulate)], then the weight gradients of all steps as two large reductions
// (mmvae_linear_bwd_weight over T B rows).
#include "common.hpp"

// ---------------------------------------------------------------------------------------------
// token ids of a one-hot batch: data (B,T,V) -> ids (T,B) [sequence-first, as nn.GRU's default layout] and the
// canonical one-hot rows oh (T*B, V) of those ids (an all-zero padding row is token 0: torch.argmax returns the first
// maximum).  One thread per (b, t).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gru_ids_kernel(const float* __restrict__ data, int* __restrict__ ids,
                                                      float* __restrict__ oh, int B, int T, int V) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * T) return;
  const int b = i / T, t = i - b * T;
  const float* row = data + (size_t)i * V;
  int best = 0;
  float bv = row[0];
  for (int v = 1; v < V; ++v) {
    const float x = row[v];
    if (x > bv) {
      bv = x;
      best = v;
    }
  }
  const size_t o = (size_t)t * B + b;
  ids[o] = best;
  for (int v = 0; v < V; ++v) oh[o * V + v] = v == best ? 1.0f : 0.0f;
}
extern "C" int mmvae_gru_token_ids(const float* onehot, int* ids, float* onehot_tb, int B, int T, int V,
                                   mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(onehot && ids && onehot_tb && B > 0 && T > 0 && V > 0);
  hipLaunchKernelGGL(gru_ids_kernel, dim3((B * T + 255) / 256), dim3(256), 0, (hipStream_t)stream, onehot, ids,
                     onehot_tb, B, T, V);
  return mmvae_launch_status();
}

// ---------------------------------------------------------------------------------------------
// One forward time step.  Workgroup = 16 batch rows x 16 hidden units, 8 waves splitting the H-deep reduction (64 per
// wave and round), 3 accumulators (r, z, n rows of W_hh) on v_mfma_f32_16x16x4_f32: lane l supplies
// A[i = l & 15][k = l >> 4] = h_prev row and B[k][j = l & 15] = W_hh row (gate * H + j0 + j), both k-contiguous: one
// float4 per 4 k (rgemm16 in gemm.hip is the same operand scheme).  Epilogue (256 threads, one (row, unit) each):
//   gx_g = Pt[(g H + j) V + id] + b_ih[g H + j]        (the embedding-folded input projection)
//   r = sigma(gx_r + gh_r), z = sigma(gx_z + gh_z), q = gh_n, n = tanh(gx_n + r q), h = (1 - z) n + z h_prev
// saved for the backward pass: r, z, n, q (B,H each).
// ---------------------------------------------------------------------------------------------
struct GruStepArgs {
  const float* hprev;   // (B,H)
  const float* whh;     // (3H,H)
  const float* bhh;     // (3H)
  const float* pt;      // (3H,V) = W_ih E^T
  const float* bih;     // (3H)
  const int* ids;       // (B) token ids of this step
  float* h;             // (B,H) out
  float* r;
  float* z;
  float* n;
  float* q;             // (B,H) each, saved
  int B, H, V;
};
__global__ __launch_bounds__(512) void gru_step_fwd_kernel(GruStepArgs a) {
  __shared__ float red[3 * 8 * 4 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, kq4 = lane >> 4;
  const int j0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
  const int H = a.H;
  const float* Ap = a.hprev + (size_t)min(m0 + l16, a.B - 1) * H;
  const float* Br = a.whh + (size_t)(j0 + l16) * H;
  const float* Bz = Br + (size_t)H * H;
  const float* Bn = Bz + (size_t)H * H;
  f32x4 ar = {0.f, 0.f, 0.f, 0.f}, az = ar, an = ar;
#pragma unroll 1
  for (int k0 = wave * 64; k0 < H; k0 += 8 * 64) {
    float4 va[4], vr[4], vz[4], vn[4];
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {        // all 16 loads of the slice in flight before the first MFMA
      const int k = k0 + 16 * qd + 4 * kq4;
      va[qd] = *reinterpret_cast<const float4*>(Ap + k);
      vr[qd] = *reinterpret_cast<const float4*>(Br + k);
      vz[qd] = *reinterpret_cast<const float4*>(Bz + k);
      vn[qd] = *reinterpret_cast<const float4*>(Bn + k);
    }
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const float x[4] = {va[qd].x, va[qd].y, va[qd].z, va[qd].w};
      const float wr[4] = {vr[qd].x, vr[qd].y, vr[qd].z, vr[qd].w};
      const float wz[4] = {vz[qd].x, vz[qd].y, vz[qd].z, vz[qd].w};
      const float wn[4] = {vn[qd].x, vn[qd].y, vn[qd].z, vn[qd].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ar = __builtin_amdgcn_mfma_f32_16x16x4f32(x[j], wr[j], ar, 0, 0, 0);
        az = __builtin_amdgcn_mfma_f32_16x16x4f32(x[j], wz[j], az, 0, 0, 0);
        an = __builtin_amdgcn_mfma_f32_16x16x4f32(x[j], wn[j], an, 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    red[((0 * 8 + wave) * 4 + r) * 64 + lane] = ar[r];
    red[((1 * 8 + wave) * 4 + r) * 64 + lane] = az[r];
    red[((2 * 8 + wave) * 4 + r) * 64 + lane] = an[r];
  }
  __syncthreads();
  if (tid < 256) {
    const int r = tid >> 6;                  // accumulator register: C[4 (l >> 4) + r][l & 15]
    float gh[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      float v = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < 8; ++w2) v += red[((g * 8 + w2) * 4 + r) * 64 + lane];
      gh[g] = v;
    }
    const int row = m0 + 4 * kq4 + r, j = j0 + l16;
    if (row < a.B) {
      const int id = a.ids[row];
      const float gxr = a.pt[(size_t)j * a.V + id] + a.bih[j];
      const float gxz = a.pt[(size_t)(H + j) * a.V + id] + a.bih[H + j];
      const float gxn = a.pt[(size_t)(2 * H + j) * a.V + id] + a.bih[2 * H + j];
      const float rg = 1.0f / (1.0f + expf(-(gxr + gh[0] + a.bhh[j])));
      const float zg = 1.0f / (1.0f + expf(-(gxz + gh[1] + a.bhh[H + j])));
      const float qv = gh[2] + a.bhh[2 * H + j];
      const float ng = tanhf(gxn + rg * qv);
      const size_t o = (size_t)row * H + j;
      const float hp = a.hprev[o];
      a.h[o] = (1.0f - zg) * ng + zg * hp;
      a.r[o] = rg;
      a.z[o] = zg;
      a.n[o] = ng;
      a.q[o] = qv;
    }
  }
}

// Forward over all T steps.  hs (T+1, B, H): hs[0] must be zero (h_0 = 0), hs[t+1] = h after step t; ids (T,B);
// saved (4, T, B, H) = r | z | n | q.  H % 64 == 0 (8 waves x 16 k per MFMA group x 4 floats) and 16-byte aligned rows.
extern "C" int mmvae_gru_forward(const float* pt, const float* bih, const float* whh, const float* bhh, const int* ids,
                                 float* hs, float* saved, int T, int B, int H, int V, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(pt && bih && whh && bhh && ids && hs && saved && T > 0 && B > 0 && H > 0 && V > 0);
  if (H % 64 != 0 || ((uintptr_t)whh & 15) || ((uintptr_t)hs & 15)) return MMVAE_ERR_UNSUPPORTED;
  const size_t BH = (size_t)B * H, TBH = (size_t)T * BH;
  const dim3 grid(H / 16, (B + 15) / 16);
  for (int t = 0; t < T; ++t) {
    GruStepArgs a;
    a.hprev = hs + (size_t)t * BH;
    a.whh = whh; a.bhh = bhh; a.pt = pt; a.bih = bih;
    a.ids = ids + (size_t)t * B;
    a.h = hs + (size_t)(t + 1) * BH;
    a.r = saved + (size_t)t * BH;
    a.z = saved + TBH + (size_t)t * BH;
    a.n = saved + 2 * TBH + (size_t)t * BH;
    a.q = saved + 3 * TBH + (size_t)t * BH;
    a.B = B; a.H = H; a.V = V;
    hipLaunchKernelGGL(gru_step_fwd_kernel, grid, dim3(512), 0, (hipStream_t)stream, a);
  }
  return mmvae_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Gate derivatives of one backward time step (elementwise over (B,H)).  dh (B,H) holds d loss / d h_t on entry; on exit
// it holds the DIRECT part of d loss / d h_{t-1} (dh * z) -- the caller then accumulates dGh W_hh into it.
//   dn = dh (1 - z), dz = dh (h_prev - n);  da_n = dn (1 - n^2);  dr = da_n q;  dq = da_n r
//   da_r = dr r (1 - r), da_z = dz z (1 - z)
//   dGx (B,3H) = [da_r | da_z | da_n]   (input side: gathers into d Pt and d b_ih)
//   dGh (B,3H) = [da_r | da_z | dq]     (hidden side: d W_hh, d b_hh, d h_prev)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gru_gates_bwd_kernel(float* __restrict__ dh, const float* __restrict__ r,
                                                            const float* __restrict__ z, const float* __restrict__ n,
                                                            const float* __restrict__ q, const float* __restrict__ hprev,
                                                            float* __restrict__ dgx, float* __restrict__ dgh, int B,
                                                            int H) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * H) return;
  const int b = (int)(i / H), j = (int)(i - (long)b * H);
  const float g = dh[i], rv = r[i], zv = z[i], nv = n[i], qv = q[i], hp = hprev[i];
  const float dan = g * (1.0f - zv) * (1.0f - nv * nv);
  const float daz = g * (hp - nv) * zv * (1.0f - zv);
  const float dar = dan * qv * rv * (1.0f - rv);
  const size_t o = (size_t)b * 3 * H + j;
  dgx[o] = dar; dgx[o + H] = daz; dgx[o + 2 * H] = dan;
  dgh[o] = dar; dgh[o + H] = daz; dgh[o + 2 * H] = dan * rv;
  dh[i] = g * zv;
}

// Backward over all T steps: dh (B,H) in = d loss / d h_T (destroyed), out = d loss / d h_0 (unused by the tower);
// dgx, dgh (T, B, 3H) out.  The weight / table / bias gradients are reductions over these two arrays (caller).
extern "C" int mmvae_gru_backward(float* dh, const float* whh, const float* hs, const float* saved, float* dgx,
                                  float* dgh, int T, int B, int H, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(dh && whh && hs && saved && dgx && dgh && T > 0 && B > 0 && H > 0);
  const size_t BH = (size_t)B * H, TBH = (size_t)T * BH;
  const unsigned blocks = (unsigned)((BH + 255) / 256);
  for (int t = T - 1; t >= 0; --t) {
    float* gx = dgx + (size_t)t * B * 3 * H;
    float* gh = dgh + (size_t)t * B * 3 * H;
    hipLaunchKernelGGL(gru_gates_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dh, saved + (size_t)t * BH,
                       saved + TBH + (size_t)t * BH, saved + 2 * TBH + (size_t)t * BH, saved + 3 * TBH + (size_t)t * BH,
                       hs + (size_t)t * BH, gx, gh, B, H);
    // dh_{t-1} += dGh_t W_hh      ((B x 3H) x (3H x H), reduction 3H)
    const int rc = mmvae_linear_bwd_data(gh, whh, nullptr, dh, B, 3 * H, H, MMVAE_EP_NONE, 1, stream);
    if (rc) return rc;
  }
  return mmvae_launch_status();
}

// ---------------------------------------------------------------------------------------------
// The reverse direction at the LAST time position = one cell step from h = 0 (`output[-1]`, encoders.py:862):
//   r = sigma(gx_r + b_hr), z = sigma(gx_z + b_hz), n = tanh(gx_n + r b_hn), h = (1 - z) n;   out = h_fwd + h
// (the sum of the two directions, encoders.py:864, is folded in).  saved (3,B,H) = r | z | n.
// Backward: dGx (B,3H) = [da_r | da_z | da_n], dGh (B,3H) = [da_r | da_z | da_n r] (d b_hh; W_hh_reverse multiplies
// h = 0 and gets no gradient).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gru_cell0_fwd_kernel(const float* __restrict__ pt, const float* __restrict__ bih,
                                                            const float* __restrict__ bhh, const int* __restrict__ ids,
                                                            const float* __restrict__ hfwd, float* __restrict__ out,
                                                            float* __restrict__ saved, int B, int H, int V) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * H) return;
  const int b = (int)(i / H), j = (int)(i - (long)b * H);
  const int id = ids[b];
  const float gxr = pt[(size_t)j * V + id] + bih[j];
  const float gxz = pt[(size_t)(H + j) * V + id] + bih[H + j];
  const float gxn = pt[(size_t)(2 * H + j) * V + id] + bih[2 * H + j];
  const float rg = 1.0f / (1.0f + expf(-(gxr + bhh[j])));
  const float zg = 1.0f / (1.0f + expf(-(gxz + bhh[H + j])));
  const float ng = tanhf(gxn + rg * bhh[2 * H + j]);
  out[i] = hfwd[i] + (1.0f - zg) * ng;
  const size_t BH = (size_t)B * H;
  saved[i] = rg;
  saved[BH + i] = zg;
  saved[2 * BH + i] = ng;
}
__global__ __launch_bounds__(256) void gru_cell0_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ saved,
                                                            const float* __restrict__ bhh, float* __restrict__ dgx,
                                                            float* __restrict__ dgh, int B, int H) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * H) return;
  const int b = (int)(i / H), j = (int)(i - (long)b * H);
  const size_t BH = (size_t)B * H;
  const float g = dout[i], rv = saved[i], zv = saved[BH + i], nv = saved[2 * BH + i];
  const float dan = g * (1.0f - zv) * (1.0f - nv * nv);
  const float daz = -g * nv * zv * (1.0f - zv);
  const float dar = dan * bhh[2 * H + j] * rv * (1.0f - rv);
  const size_t o = (size_t)b * 3 * H + j;
  dgx[o] = dar; dgx[o + H] = daz; dgx[o + 2 * H] = dan;
  dgh[o] = dar; dgh[o + H] = daz; dgh[o + 2 * H] = dan * rv;
}
extern "C" int mmvae_gru_cell0_fwd(const float* pt, const float* bih, const float* bhh, const int* ids, const float* hfwd,
                                   float* out, float* saved, int B, int H, int V, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(pt && bih && bhh && ids && hfwd && out && saved && B > 0 && H > 0 && V > 0);
  hipLaunchKernelGGL(gru_cell0_fwd_kernel, dim3((unsigned)(((size_t)B * H + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, pt, bih, bhh, ids, hfwd, out, saved, B, H, V);
  return mmvae_launch_status();
}
extern "C" int mmvae_gru_cell0_bwd(const float* dout, const float* saved, const float* bhh, float* dgx, float* dgh, int B,
                                   int H, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(dout && saved && bhh && dgx && dgh && B > 0 && H > 0);
  hipLaunchKernelGGL(gru_cell0_bwd_kernel, dim3((unsigned)(((size_t)B * H + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, dout, saved, bhh, dgx, dgh, B, H);
  return mmvae_launch_status();
}
