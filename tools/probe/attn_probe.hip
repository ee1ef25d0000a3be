#include "../../multimodal_vae_comparison_amd/csrc/common.hpp"
#define AT_HD 16
#define AT_HP 17
#define AT_SP 129
__device__ __forceinline__ int at_i(int r, int lh) { return 8 * (r >> 2) + 4 * lh + (r & 3); }
// rows x hd floats (row r at src[(r N + n) ld + col0 ..]) -> dst[r][AT_HP], zero padded to 128 rows x 16 columns
__device__ __forceinline__ void at_stage(float* __restrict__ dst, const float* __restrict__ src, int rows, int N, int n,
                                         long ld, int col0, int hd, float mul, int tid) {
  for (int e = tid; e < 128 * AT_HD; e += 256) {
    const int r = e >> 4, d = e & 15;
    const bool ok = r < rows && d < hd;
    const float v = src[ok ? ((size_t)r * N + n) * ld + col0 + d : (size_t)n * ld + col0];
    dst[r * AT_HP + d] = ok ? v * mul : 0.f;
  }
}
__global__ __launch_bounds__(256) void attn_mfma_fwd_probe(long long* stamps, const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ v, const uint8_t* __restrict__ kpm,
                                                            float* __restrict__ out, float* __restrict__ probs, int L,
                                                            int S, int N, int H, int hd, long ldq, long ldk, long ldv,
                                                            int mask_is_valid, mmvae_dropout_t drop) {
  __shared__ float sq[128 * AT_HP], sk[128 * AT_HP], sv[128 * AT_HP];
  __shared__ float sp[128 * AT_SP];
  __shared__ float smask[128], sinv[128];
  const int n = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 31, lh = lane >> 5;
  long long st[8]; int ns = 0;
#define ST() st[ns++] = clock64()
  ST();
  const float scale = 1.0f / sqrtf((float)hd);
  at_stage(sq, q, L, N, n, ldq, h * hd, hd, scale, tid);
  at_stage(sk, k, S, N, n, ldk, h * hd, hd, 1.0f, tid);
  at_stage(sv, v, S, N, n, ldv, h * hd, hd, 1.0f, tid);
  if (tid < 128)
    smask[tid] = (tid >= S || (kpm && ((kpm[(size_t)n * S + tid] != 0) != (mask_is_valid != 0)))) ? 1.f : 0.f;
  __syncthreads();
  ST();
  const int l0 = wave * 32;
  if (l0 >= L) return;                      // (no barrier below: every wave works on its own 32 rows of the tile)
  const int nkb = (S + 31) >> 5;            // key blocks
  // ---- scores: rows l0 .. l0+31, all keys ----
  float qa[AT_HD / 2];
#pragma unroll
  for (int kk = 0; kk < AT_HD / 2; ++kk) qa[kk] = sq[(l0 + li) * AT_HP + 2 * kk + lh];
  for (int kb = 0; kb < nkb; ++kb) {
    const int s0 = kb * 32;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < AT_HD / 2; ++kk)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[kk], sk[(s0 + li) * AT_HP + 2 * kk + lh], acc, 0, 0, 0);
    const bool masked = smask[s0 + li] != 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) sp[(l0 + at_i(r, lh)) * AT_SP + s0 + li] = masked ? -INFINITY : acc[r];
  }
  ST();
  // ---- softmax: lane (li, lh) = row l0 + li, keys [64 lh, 64 lh + 64) ----
  const int row = l0 + li, kbeg = 64 * lh, kend = min(S, kbeg + 64);
  float* prow = sp + row * AT_SP;
  float mx = -INFINITY;
  for (int s_ = kbeg; s_ < kend; ++s_) mx = fmaxf(mx, prow[s_]);
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
  for (int s_ = kbeg; s_ < kend; ++s_) {
    const float p = expf(prow[s_] - mx);
    prow[s_] = p;
    sum += p;
  }
  for (int s_ = max(kend, kbeg); s_ < kbeg + 64; ++s_) prow[s_] = 0.f;      // padded keys of the P V product
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
  if (lh == 0) sinv[row] = inv;
  ST();
  // ---- O = (P . mask) V ----
  const DropKey dkey = drop_key(drop);
  const uint32_t drow = (uint32_t)((((size_t)n * H + h) * L + row) * S);
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
  const int ksteps = (S + 1) >> 1;
  for (int kk = 0; kk < ksteps; ++kk) {
    const int s_ = 2 * kk + lh;
    const float a = prow[s_] * drop_mul(dkey, drow + (uint32_t)s_);
    const float b = li < AT_HD ? sv[s_ * AT_HP + li] : 0.f;
    o = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, o, 0, 0, 0);
  }
  ST();
  if (li < hd) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int l = l0 + at_i(r, lh);
      if (l < L) out[((size_t)l * N + n) * ((size_t)H * hd) + h * hd + li] = o[r] * sinv[l];
    }
  }
  ST();
  // ---- normalised probabilities of this wave's rows, lanes over the keys ----
  float* P = probs + ((size_t)n * H + h) * L * S;
  for (int l = l0; l < min(L, l0 + 32); ++l) {
    const float iv = sinv[l];
    for (int s_ = lane; s_ < S; s_ += 64) P[(size_t)l * S + s_] = sp[l * AT_SP + s_] * iv;
  }
  ST();
  if (lane == 0) for (int i = 0; i < 8; ++i) stamps[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 8 + i] = i < ns ? st[i] : 0;
}

extern "C" int attn_probe(long long* stamps, const float* q, const float* k, const float* v, const unsigned char* kpm, float* out, float* probs,
                          int L, int S, int N, int H, int hd, long ld, void* stream) {
  mmvae_dropout_t z = {nullptr, 0u, 0u, 0.f};
  hipLaunchKernelGGL(attn_mfma_fwd_probe, dim3(N, H), dim3(256), 0, (hipStream_t)stream, stamps, q, k, v, kpm, out, probs, L, S, N, H, hd, ld, ld, ld, 1, z);
  return (int)hipGetLastError();
}
