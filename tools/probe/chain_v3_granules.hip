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
//   * hand-over between stages per ROW BLOCK, not per grid.  Besides the plain copy later launches read, a stage's outputs
//     are stored as 8-byte granules {value, tag = this launch's epoch + 1}, ONE sc1 (write-through) store each, and every
//     storing wave adds 1 to the row block's arrival counter straight behind its stores -- no drain, no barrier, no fence on
//     the producer side.  A consumer waits with ONE lane polling that counter, then every wave reads its 16 granules per lane
//     with sc1 loads and checks the tags: the data is its own proof (cdna_hip_programming.md, Guideline 16, R2), so a counter
//     that overtook the stores only costs another sweep.  (Measured on the way: sc1 stores -> vmcnt(0) -> barrier -> counter
//     -> poll -> barrier -> sc1 loads cost ~6 us per stage alone on the chip; waiting by sweeping the granules themselves
//     from all 512 lanes starved the rest of the chip.)
//   * a stage's WEIGHT operands do not depend on the hand-over: they are fetched into registers BEFORE the sweep, so the
//     wait hides their latency (in the step they come from HBM: Adam has just rewritten them);
//   * nothing to re-zero between launches: the epoch lives on the device, every workgroup reads it and then takes an exit
//     ticket, the last ticket moves it on; the arrival counters are monotonic (target = a multiple of the epoch) -- graph
//     replays start clean, no memset node;
//   * every sweep is bounded (20 ms of wall clock): a timeout sets a sticky word the host can read and the launch ends.
//
// Forward stages compute y = act(x) W^T + b, backward stages ("transposed") dx = (dy W) * act'(saved pre-activation).
#include "common.hpp"

#define CH_MAX_STAGES MMVAE_CHAIN_MAX_STAGES
#define CH_MAX_RB 16
#define CH_EXIT 0
#define CH_TMO 1
#define CH_EPOCH 2
#define CH_CTR 4      // [stage][row block] arrival counters: monotonic, 4 * (column blocks) per launch

typedef unsigned long long u64;

struct ChainArgs {
  const float* x;
  int ldx, M, nrb, n;
  unsigned total;       // workgroups that take an exit ticket
  unsigned* sync;
  u64* gran[CH_MAX_STAGES];      // hand-over copy of stage s's output: (rows, n_out) granules {value, tag}
  mmvae_chain_stage_t st[CH_MAX_STAGES];
};

__device__ __forceinline__ unsigned ch_ld(unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ch_st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// one stage of one workgroup: the 16 x 16 tile (rb, cb) of y = A B, A = act(x rows of the row block), B from st.w
template <bool TR>
__device__ __forceinline__ void ch_stage(const ChainArgs& a, const mmvae_chain_stage_t& st, const int s, const float* xin,
                                         const int ldx, const int rb, const int cb, const u64* gin, u64* gout,
                                         unsigned* wait_on, const unsigned need, unsigned* arrive, unsigned* tmo,
                                         const unsigned* s_epoch, float* red) {
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
  // ---- weight operand B[k][j = l16]: independent of the hand-over, so in flight across the wait
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
  // ---- activation operand A[i = l16][k]
  // (whole vectors are bit-cast: element-indexing the builtin's result made hipcc 7.2 narrow the load to ONE dword and
  // feed the same element to all four MFMAs)
  float av[4][4];
  if (!gin) {      // stage 0: x was written by an earlier launch
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xin), 0, (int)((long)a.M * ldx * 4), 0x00020000);
    f32x4 at[4];
#pragma unroll
    for (int qi = 0; qi < 4; ++qi)
      at[qi] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (row * ldx + kk[qi]) * 4, 0, 0));
#pragma unroll
    for (int qi = 0; qi < 4; ++qi) {
      av[qi][0] = at[qi].x; av[qi][1] = at[qi].y; av[qi][2] = at[qi].z; av[qi][3] = at[qi].w;
    }
  } else {
    // The hand-over: the rows come as 8-byte granules {value, tag = this launch's epoch + 1}, each written by ONE sc1 store of
    // the producing workgroup: the data IS the flag (Guideline 16, R2), so nothing orders the producer's stores -- no drain,
    // no barrier, no fence.  WAITING on the payload itself is what must not happen: 256 workgroups x 512 lanes re-reading
    // 64 KB each per pass starved everything else on the chip (the second version: the step got slower and, beside a
    // bandwidth-bound kernel on another stream, sweeps ran into their 20 ms bound).  So the wait is ONE lane polling the row
    // block's arrival counter (each storing wave adds 1 right behind its stores, unordered with them), and the tags are the
    // proof: normally the first sweep after the counter finds every tag in place, else it sweeps again.
    if (wave == 0) {
      unsigned long long t0w = 0;
      while ((int)(ch_ld(wait_on) - need) < 0) {
        __builtin_amdgcn_s_sleep(2);
        const unsigned long long now = wall_clock64();
        if (t0w == 0) t0w = now;
        else if (now - t0w > 2000000ull) {
          if (lane == 0) ch_st(tmo, 1u);
          break;
        }
      }
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<u64*>(gin), 0, (int)((long)a.nrb * 16 * ldx * 8), 0x00020000);
    const unsigned tag = __hip_atomic_load(s_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + 1u;
    f32x4 g[8];
    unsigned long long t0 = 0;
    for (;;) {
#pragma unroll
      for (int qi = 0; qi < 4; ++qi) {
        const int off = (row * ldx + kk[qi]) * 8;
        g[2 * qi] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16));
        g[2 * qi + 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16, 0, 16));
      }
      bool good = true;
#pragma unroll
      for (int i = 0; i < 8; ++i) {      // (copies first: __builtin_bit_cast straight on a vector ELEMENT reads element 0, hipcc 7.2)
        const float t0g = g[i].y, t1g = g[i].w;
        good = good && __float_as_uint(t0g) == tag && __float_as_uint(t1g) == tag;
      }
      if (__all(good)) break;
      __builtin_amdgcn_s_sleep(8);
      const unsigned long long now = wall_clock64();
      if (t0 == 0) t0 = now;
      else if (now - t0 > 2000000ull) {      // 20 ms at 100 MHz: give up (sticky flag for the host), never hang
        if (lane == 0) ch_st(tmo, 1u);
        break;
      }
    }
#pragma unroll
    for (int qi = 0; qi < 4; ++qi) {
      av[qi][0] = g[2 * qi].x; av[qi][1] = g[2 * qi].z; av[qi][2] = g[2 * qi + 1].x; av[qi][3] = g[2 * qi + 1].z;
    }
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
  // cross-wave sum through LDS; `red` alternates between two buffers from stage to stage, so ONE barrier per stage is enough
  // (a wave that runs ahead into the next stage writes the other buffer)
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
      if (gout) {      // ONE 8-byte sc1 store per granule: the next stage's consumers see value and tag together
        const unsigned tag = *s_epoch + 1u;
        __hip_atomic_store(gout + o, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      }
      st.y[o] = v;     // the copy later launches read (saved pre-activation / per-layer gradient)
    }
    // (every storing wave for itself, straight behind its stores: the counter may overtake them, the tags cannot lie)
    if (arrive && lane == 0) __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ __launch_bounds__(512) void linear_chain_kernel(ChainArgs a) {
  __shared__ float red[2][8 * 4 * 64];
  __shared__ unsigned s_epoch;
  const int tid = threadIdx.x;
  const unsigned wg = blockIdx.x;
  const int rb = (int)((wg & 7u) + 8u * (wg >> 8)), cb = (int)((wg >> 3) & 31u);
  if (rb >= a.nrb) return;
  unsigned* const exitc = a.sync + CH_EXIT;
  unsigned* const tmo = a.sync + CH_TMO;
  unsigned* const epoch = a.sync + CH_EPOCH;
  // this launch's epoch (the granule tag is epoch + 1); every workgroup reads it, THEN takes an exit ticket: whoever draws
  // the last ticket knows that every workgroup holds its copy and moves the epoch on for the next launch (stream order
  // puts that launch behind this one's end).  Nothing else is shared, nothing to re-zero: graph replays start clean.
  unsigned ticket = 0, ep0 = 0;
  if (tid == 0) {
    ep0 = ch_ld(epoch);
    s_epoch = ep0;
  }
  __syncthreads();
  if (tid == 0) ticket = __hip_atomic_fetch_add(exitc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const float* xin = a.x;
  int ldx = a.ldx;
#pragma unroll 1
  for (int s = 0; s < a.n; ++s) {
    const mmvae_chain_stage_t st = a.st[s];
    if (cb < ((st.n_out + 15) >> 4)) {
      const u64* gin = s > 0 ? a.gran[s - 1] : nullptr;
      u64* gout = s + 1 < a.n ? a.gran[s] : nullptr;
      // counters are never reset: a launch adds 4 per participating column block, so this launch's target is a multiple of
      // its epoch (wrap-safe signed comparison; the block belongs to ONE call site with ONE shape)
      unsigned* const wait_on = s > 0 ? a.sync + CH_CTR + (s - 1) * CH_MAX_RB + rb : nullptr;
      const unsigned need = s > 0 ? (s_epoch + 1u) * 4u * (unsigned)((a.st[s - 1].n_out + 15) >> 4) : 0u;
      unsigned* const arrive = gout ? a.sync + CH_CTR + s * CH_MAX_RB + rb : nullptr;
      if (st.transposed) ch_stage<true>(a, st, s, xin, ldx, rb, cb, gin, gout, wait_on, need, arrive, tmo, &s_epoch, red[s & 1]);
      else ch_stage<false>(a, st, s, xin, ldx, rb, cb, gin, gout, wait_on, need, arrive, tmo, &s_epoch, red[s & 1]);
    }
    xin = st.y;
    ldx = st.n_out;
  }
  if (tid == 0 && ticket == a.total - 1u) {
    ch_st(exitc, 0u);
    ch_st(epoch, ep0 + 1u);      // (wraps naturally, as the counters do: their targets are multiples of it mod 2^32)
  }
}

extern "C" int mmvae_linear_chain_supported(int M, const int* widths, int n_stages) {
  if (n_stages < 1 || n_stages > CH_MAX_STAGES || M < 1 || M > 16 * CH_MAX_RB) return 0;
  for (int i = 0; i <= n_stages; ++i)
    if (widths[i] < 4 || widths[i] > 512 || (widths[i] & 3)) return 0;
  return 1;
}
extern "C" size_t mmvae_linear_chain_sync_words(void) { return CH_CTR + CH_MAX_STAGES * CH_MAX_RB; }
// hand-over scratch of one call site: granule copies of every stage output but the last
extern "C" size_t mmvae_linear_chain_scratch_bytes(int M, const int* widths, int n_stages) {
  size_t b = 0;
  const size_t rows = (size_t)((M + 15) / 16) * 16;
  for (int s = 1; s < n_stages; ++s) b += rows * (size_t)widths[s] * 8;
  return b;
}

extern "C" int mmvae_linear_chain(const float* x, long ldx, const mmvae_chain_stage_t* stages, int n_stages, int M,
                                  unsigned* sync, void* scratch, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(x && stages && sync && n_stages >= 1 && n_stages <= CH_MAX_STAGES && M >= 1 && M <= 16 * CH_MAX_RB);
  MMVAE_CHECK_ARG((ldx & 3) == 0 && ((uintptr_t)x & 15) == 0 && (long)M * ldx * 4 < (1l << 31));
  MMVAE_CHECK_ARG(n_stages == 1 || (scratch && ((uintptr_t)scratch & 15) == 0));
  ChainArgs a;
  a.x = x;
  a.ldx = (int)ldx;
  a.M = M;
  a.nrb = (M + 15) / 16;
  a.n = n_stages;
  a.total = (unsigned)a.nrb * 32u;
  a.sync = sync;
  int prev = -1;
  char* sp = (char*)scratch;
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
    a.gran[s] = nullptr;
    if (s + 1 < n_stages) {
      a.gran[s] = (u64*)sp;
      sp += (size_t)a.nrb * 16 * st.n_out * 8;
    }
  }
  const unsigned grid = (unsigned)((a.nrb + 7) / 8) * 256u;
  hipLaunchKernelGGL(linear_chain_kernel, dim3(grid), dim3(512), 0, (hipStream_t)stream, a);
  return mmvae_launch_status();
}
