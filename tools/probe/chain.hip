// A chain of batch-row linear layers in ONE launch (round 5): the latency-bound middle of the step.
//
// At the batch sizes the step is quoted on (128 rows) the image towers' MLPs -- Enc_CNN2.lin1 -> heads
// (models/encoders.py:194,218-223,49-54), Dec_CNN.lin1 -> lin2 -> lin3 (models/decoders.py:58-60,86-88) and their data
// gradients -- are 67 MFLOP GEMMs that take 5-8 us each as launches of their own: fill the chip, fetch operands from
// L2 / HBM, drain, next launch.  Here a chain of up to CH_MAX_STAGES such layers is one launch of persistent workgroups:
//
//   * a stage is column-sliced: workgroup (rb, cb) owns the 16 x 16 output tile (row block rb, column block cb) of every
//     stage (v_mfma_f32_16x16x4_f32, 8 waves split the reduction, LDS only for the cross-wave sum) -- 256 workgroups for
//     128 rows x 512 columns, one per CU;
//   * row block rb lives on XCD rb % 8 (workgroup id w runs on XCD w % 8): the 32 workgroups that exchange a row block's
//     activations share an L2 (a speed matter only -- nothing below depends on the placement);
//   * hand-over between stages per ROW BLOCK, not per grid: outputs are stored write-through (sc1), every storing wave
//     drains (s_waitcnt vmcnt(0)), the workgroup's barrier, one lane adds to the row block's arrival counter (agent scope);
//     the consumers of that row block poll the counter with sc1 loads (one wave, s_sleep), barrier, and read the rows with
//     sc1 buffer loads straight into registers (cdna_hip_programming.md, Guideline 16: the all-sc1 form, no fences);
//   * a stage's WEIGHT operands do not depend on the hand-over: they are fetched into registers BEFORE the poll, so the
//     wait hides their latency;
//   * the counters clean themselves: every workgroup takes an exit ticket once its last poll has matched, and the one
//     that draws the last ticket zeroes the block (nobody polls any more) -- no memset node, graph replays start clean;
//   * every spin is bounded (20 ms of wall clock): a timeout sets a sticky word the host can read and the launch ends.
//
// Forward stages compute y = act(x) W^T + b, backward stages ("transposed") dx = (dy W) * act'(saved pre-activation).
#include "common.hpp"

#define CH_MAX_STAGES MMVAE_CHAIN_MAX_STAGES
#define CH_MAX_RB 16
#define CH_EXIT 64
#define CH_TMO 65

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct ChainArgs {
  const float* x;
  int ldx, M, nrb, n;
  unsigned total;       // workgroups that take an exit ticket
  unsigned* sync;
  mmvae_chain_stage_t st[CH_MAX_STAGES];
};

__device__ __forceinline__ unsigned ch_ld(unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ch_st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// one wave polls one word (relaxed sc1 loads); bounded by the wall clock (100 MHz)
__device__ __forceinline__ void ch_wait(unsigned* c, unsigned need, unsigned* tmo) {
  if (ch_ld(c) >= need) return;
  const unsigned long long t0 = wall_clock64();
  while (ch_ld(c) < need) {
    __builtin_amdgcn_s_sleep(2);
    if (wall_clock64() - t0 > 2000000ull) {
      ch_st(tmo, 1u);
      break;
    }
  }
}

// one stage of one workgroup: the 16 x 16 tile (rb, cb) of y = A B, A = act(x rows of the row block), B from st.w
template <bool TR>
__device__ __forceinline__ void ch_stage(const ChainArgs& a, const mmvae_chain_stage_t& st, const int s, const float* xin,
                                         const int ldx, const int rb, const int cb, unsigned* wait_on, const unsigned need,
                                         unsigned* tmo, unsigned* exitc, const bool take_ticket, unsigned& ticket,
                                         float* red) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, kq4 = lane >> 4;
  const int K = st.n_in, N = st.n_out, nq = (K + 15) >> 4;
  const int n0 = cb * 16;
  const int col = min(n0 + l16, N - 1);
  const int row = min(rb * 16 + l16, a.M - 1);
  // lane (l16, kq4) holds k = 16 q + 4 kq4 + 0..3 of the quads q = wave, wave + 8, ... (K % 4 == 0: a float4 is inside or
  // outside the reduction; outside ones are fetched from k = 0 and zeroed)
  int kk[4];
  bool ok[4];
#pragma unroll
  for (int qi = 0; qi < 4; ++qi) {
    const int k = 16 * (wave + 8 * qi) + 4 * kq4;
    ok[qi] = k < K;
    kk[qi] = ok[qi] ? k : 0;
  }
  // ---- weight operand B[k][j = l16]: independent of the hand-over, so in flight across the poll
  float b[4][4];
  if (!TR) {                       // W (N, K): k-contiguous rows
    const float* wp = st.w + (long)col * K;
#pragma unroll
    for (int qi = 0; qi < 4; ++qi) {
      const float4 t = *reinterpret_cast<const float4*>(wp + kk[qi]);
      b[qi][0] = t.x; b[qi][1] = t.y; b[qi][2] = t.z; b[qi][3] = t.w;
    }
  } else {                         // W (K, N): the data gradient dy W
    const float* wp = st.w + col;
#pragma unroll
    for (int qi = 0; qi < 4; ++qi)
#pragma unroll
      for (int j = 0; j < 4; ++j) b[qi][j] = wp[(long)(kk[qi] + j) * N];
  }
  // ---- hand-over: this row block's rows from the previous stage
  if (wait_on) {
    if (wave == 0) ch_wait(wait_on, need, tmo);
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      // (no instruction: keeps the loads below the poll)
    if (take_ticket && tid == 0) ticket = __hip_atomic_fetch_add(exitc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // ---- activation operand A[i = l16][k]: sc1 loads (they bypass this CU's L1), all in flight at once
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xin), 0, (int)((long)a.M * ldx * 4), 0x00020000);
  // (the whole vector is bit-cast: element-indexing the builtin's result made hipcc 7.2 narrow the load to ONE dword and
  // feed the same element to all four MFMAs)
  f32x4 at[4];
#pragma unroll
  for (int qi = 0; qi < 4; ++qi)
    at[qi] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (row * ldx + kk[qi]) * 4, 0, 16));
  float av[4][4];
#pragma unroll
  for (int qi = 0; qi < 4; ++qi) {
    av[qi][0] = at[qi].x; av[qi][1] = at[qi].y; av[qi][2] = at[qi].z; av[qi][3] = at[qi].w;
  }
  if (st.in_act == MMVAE_ACT_RELU) {
#pragma unroll
    for (int qi = 0; qi < 4; ++qi)
#pragma unroll
      for (int j = 0; j < 4; ++j) av[qi][j] = fmaxf(av[qi][j], 0.f);
  } else if (st.in_act == MMVAE_ACT_SILU) {
#pragma unroll
    for (int qi = 0; qi < 4; ++qi)
#pragma unroll
      for (int j = 0; j < 4; ++j) av[qi][j] = dev_silu(av[qi][j]);
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int qi = 0; qi < 4; ++qi) {
    if (wave + 8 * qi < nq) {      // (wave-uniform)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ok[qi] ? av[qi][j] : 0.f, b[qi][j], acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[(wave * 4 + r) * 64 + lane] = acc[r];
  __syncthreads();
  if (tid < 256) {
    const int r = tid >> 6;
    float v = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < 8; ++w2) v += red[(w2 * 4 + r) * 64 + lane];
    const int orow = rb * 16 + 4 * kq4 + r, ocol = n0 + l16;
    if (orow < a.M && ocol < N) {
      if (st.bias) v += st.bias[ocol];
      const long o = (long)orow * N + ocol;
      if (st.ep == MMVAE_EP_MUL_RELU_MASK) v = st.aux[o] > 0.f ? v : 0.f;
      else if (st.ep == MMVAE_EP_MUL_SILU_GRAD) v *= dev_silu_grad(st.aux[o]);
      __hip_atomic_store(st.y + o, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // global_store_dword sc1
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains before the workgroup signals
  __syncthreads();
}

__global__ __launch_bounds__(512) void linear_chain_kernel(ChainArgs a) {
  __shared__ float red[8 * 4 * 64];
  const int tid = threadIdx.x;
  const unsigned wg = blockIdx.x;
  const int rb = (int)((wg & 7u) + 8u * (wg >> 8)), cb = (int)((wg >> 3) & 31u);
  if (rb >= a.nrb) return;
  unsigned* const ctr = a.sync;
  unsigned* const exitc = a.sync + CH_EXIT;
  unsigned* const tmo = a.sync + CH_TMO;
  int lastpoll = 0;      // last stage in which this workgroup waits for a hand-over
  for (int s = 1; s < a.n; ++s)
    if (cb < ((a.st[s].n_out + 15) >> 4)) lastpoll = s;
  unsigned ticket = 0;
  if (lastpoll == 0 && tid == 0) ticket = __hip_atomic_fetch_add(exitc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const float* xin = a.x;
  int ldx = a.ldx;
#pragma unroll 1
  for (int s = 0; s < a.n; ++s) {
    const mmvae_chain_stage_t st = a.st[s];
    if (cb < ((st.n_out + 15) >> 4)) {
      unsigned* const wait_on = s > 0 ? ctr + (s - 1) * CH_MAX_RB + rb : nullptr;
      const unsigned need = s > 0 ? (unsigned)((a.st[s - 1].n_out + 15) >> 4) : 0u;
      if (st.transposed) ch_stage<true>(a, st, s, xin, ldx, rb, cb, wait_on, need, tmo, exitc, s == lastpoll, ticket, red);
      else ch_stage<false>(a, st, s, xin, ldx, rb, cb, wait_on, need, tmo, exitc, s == lastpoll, ticket, red);
      if (tid == 0 && s + 1 < a.n)
        __hip_atomic_fetch_add(ctr + s * CH_MAX_RB + rb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    xin = st.y;
    ldx = st.n_out;
  }
  // the last exit ticket: every workgroup is past its last poll -- leave the block zeroed for the next launch
  if (tid == 0 && ticket == a.total - 1u) {
    for (int i = 0; i < CH_MAX_STAGES * CH_MAX_RB; ++i) ch_st(ctr + i, 0u);
    ch_st(exitc, 0u);
  }
}

extern "C" int mmvae_linear_chain_supported(int M, const int* widths, int n_stages) {
  if (n_stages < 1 || n_stages > CH_MAX_STAGES || M < 1 || M > 16 * CH_MAX_RB) return 0;
  for (int i = 0; i <= n_stages; ++i)
    if (widths[i] < 4 || widths[i] > 512 || (widths[i] & 3)) return 0;
  return 1;
}
extern "C" size_t mmvae_linear_chain_sync_words(void) { return 96; }

extern "C" int mmvae_linear_chain(const float* x, long ldx, const mmvae_chain_stage_t* stages, int n_stages, int M,
                                  unsigned* sync, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(x && stages && sync && n_stages >= 1 && n_stages <= CH_MAX_STAGES && M >= 1 && M <= 16 * CH_MAX_RB);
  MMVAE_CHECK_ARG((ldx & 3) == 0 && ((uintptr_t)x & 15) == 0 && (long)M * ldx * 4 < (1l << 31));
  ChainArgs a;
  a.x = x;
  a.ldx = (int)ldx;
  a.M = M;
  a.nrb = (M + 15) / 16;
  a.n = n_stages;
  a.total = (unsigned)a.nrb * 32u;
  a.sync = sync;
  int prev = -1;
  for (int s = 0; s < n_stages; ++s) {
    const mmvae_chain_stage_t& st = stages[s];
    MMVAE_CHECK_ARG(st.w && st.y && st.n_in >= 4 && st.n_in <= 512 && st.n_out >= 4 && st.n_out <= 512);
    MMVAE_CHECK_ARG((st.n_in & 3) == 0 && (st.n_out & 3) == 0 && ((uintptr_t)st.w & 15) == 0 && ((uintptr_t)st.y & 15) == 0);
    MMVAE_CHECK_ARG(prev < 0 || st.n_in == prev);
    MMVAE_CHECK_ARG(s > 0 || st.n_in <= ldx);
    MMVAE_CHECK_ARG(!ep_reads_aux(st.ep) || st.aux);
    MMVAE_CHECK_ARG(st.ep == MMVAE_EP_NONE || st.ep == MMVAE_EP_MUL_RELU_MASK || st.ep == MMVAE_EP_MUL_SILU_GRAD);
    prev = st.n_out;
    a.st[s] = st;
  }
  const unsigned grid = (unsigned)((a.nrb + 7) / 8) * 256u;
  hipLaunchKernelGGL(linear_chain_kernel, dim3(grid), dim3(512), 0, (hipStream_t)stream, a);
  return mmvae_launch_status();
}
