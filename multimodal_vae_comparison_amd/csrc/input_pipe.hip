// Native input step of the captured training step (the caller side of the path): one packed
// pinned host batch -> ONE host-to-device copy on the pipe's own copy stream, under the step that is running ->
// device expansion into the step's static inputs right before the next replay.  Replaces the reference's
// DataLoader -> `.to(device)` of fp32 images / one-hot text in front of every step (models/dataloader.py:120-126,
// pin_memory=True, moved by lightning's batch transfer; datasets.py:251-254, :272-281 build the fp32 tensors on the host).
//
// Everything one step needs -- wait for the staged batch, the expansion launches, "staging consumed", and the next
// copy behind that -- is ONE library call: at 0.41 ms per step the graph launch itself keeps the host thread ~75 %
// busy, and the same sequence issued as eight separate runtime calls from Python made the loop host-bound
// (0.53 ms per step; tools/probe/input_pipeline_time.py).
#include "common.hpp"
#include <stdlib.h>

// The H2D transfer as a kernel on the copy stream: the pinned batch is device-visible, 16-byte loads over the host
// link, a small grid (it shares the chip with the running step).  hipMemcpyAsync of the same 1.6 MB costs the host
// thread > 100 us per call while a graph is in flight (tools/probe/input_pipeline_time.py: "+ copy only").
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void pull_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) dst[i] = __builtin_nontemporal_load(src + i);
}

// The same transfer as a node INSIDE the captured step (round 3): the source is slot (count % n_ring) of a ring of pinned
// batches whose addresses sit in a device table, and the launch advances the count itself (last-workgroup ticket, as the
// noise generators do) -- the graph replays with no per-step runtime call besides its own launch.
__global__ __launch_bounds__(256) void pull_ring_kernel(const u32x4* const* __restrict__ ring, int n_ring,
                                                        unsigned* __restrict__ ctr, u32x4* __restrict__ dst, size_t off16,
                                                        size_t n16, int advance) {
  const u32x4* __restrict__ src = ring[ctr[0] % (unsigned)n_ring] + off16;
  dst += off16;
  const size_t stride = (size_t)gridDim.x * 256;
  // 8 independent 16-byte loads over the host link in flight per thread before the first store (a round trip is ~2 us)
  for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < n16; i0 += 8 * stride) {
    u32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const size_t i = i0 + u * stride;
      if (i < n16) v[u] = __builtin_nontemporal_load(src + i);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const size_t i = i0 + u * stride;
      if (i < n16) dst[i] = v[u];
    }
  }
  __syncthreads();
  if (advance && threadIdx.x == 0) {
    const unsigned ticket = atomicAdd(ctr + 1, 1u);
    if (ticket == gridDim.x - 1) {      // every workgroup has read ctr[0] (it did before its copy loop)
      ctr[1] = 0u;
      ctr[0] += 1u;
    }
  }
}
extern "C" int mmvae_input_ring_pull(const void* const* ring_dev, int n_ring, unsigned* ctr_dev, void* staging,
                                     size_t offset, size_t bytes, int advance, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(ring_dev && n_ring > 0 && ctr_dev && staging && bytes > 0);
  if ((((uintptr_t)staging) | bytes | offset) & 15) return MMVAE_ERR_ARG;
  const size_t n16 = bytes / 16;
  size_t blocks = (n16 + 8 * 256 - 1) / (8 * 256);
  if (blocks > 64) blocks = 64;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(pull_ring_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const u32x4* const*>(ring_dev), n_ring, ctr_dev, static_cast<u32x4*>(staging),
                     offset / 16, n16, advance);
  return mmvae_launch_status();
}

struct mmvae_input_pipe {
  hipStream_t copy;
  hipEvent_t staged;     // recorded on the copy stream after the H2D copy
  hipEvent_t consumed;   // recorded on the step's stream after the expansion launches
  void* staging;         // device buffer (owned by the caller)
  size_t bytes;
  int staged_once;
  int dma;               // MMVAE_INPUT_PIPE_DMA=1: hipMemcpyAsync instead of the pull kernel
};

extern "C" int mmvae_input_pipe_create(mmvae_input_pipe_t** out, void* staging, size_t bytes) {
  MMVAE_CHECK_ARG(out && staging && bytes > 0);
  mmvae_input_pipe* p = new mmvae_input_pipe();
  p->staging = staging;
  p->bytes = bytes;
  p->staged_once = 0;
  p->dma = getenv("MMVAE_INPUT_PIPE_DMA") && atoi(getenv("MMVAE_INPUT_PIPE_DMA")) == 1;
  if (hipStreamCreateWithFlags(&p->copy, hipStreamNonBlocking) != hipSuccess) { delete p; return MMVAE_ERR_LAUNCH; }
  if (hipEventCreateWithFlags(&p->staged, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&p->consumed, hipEventDisableTiming) != hipSuccess) {
    (void)hipStreamDestroy(p->copy);
    delete p;
    return MMVAE_ERR_LAUNCH;
  }
  *out = p;
  return MMVAE_OK;
}

extern "C" int mmvae_input_pipe_destroy(mmvae_input_pipe_t* p) {
  if (!p) return MMVAE_OK;
  (void)hipStreamSynchronize(p->copy);
  (void)hipEventDestroy(p->staged);
  (void)hipEventDestroy(p->consumed);
  (void)hipStreamDestroy(p->copy);
  delete p;
  return MMVAE_OK;
}

extern "C" int mmvae_input_pipe_prefetch(mmvae_input_pipe_t* p, const void* host_packed) {
  MMVAE_CHECK_ARG(p && host_packed);
  // (an event that was never recorded is complete: the first copy does not wait)
  if (hipStreamWaitEvent(p->copy, p->consumed, 0) != hipSuccess) return MMVAE_ERR_LAUNCH;
  const bool al16 = (((uintptr_t)host_packed | (uintptr_t)p->staging | p->bytes) & 15) == 0;
  if (p->dma || !al16) {
    if (hipMemcpyAsync(p->staging, host_packed, p->bytes, hipMemcpyHostToDevice, p->copy) != hipSuccess) return MMVAE_ERR_LAUNCH;
  } else {
    const size_t n16 = p->bytes / 16;
    size_t blocks = (n16 + 255) / 256;
    if (blocks > 64) blocks = 64;
    hipLaunchKernelGGL(pull_kernel, dim3((unsigned)blocks), dim3(256), 0, p->copy, static_cast<const u32x4*>(host_packed),
                       static_cast<u32x4*>(p->staging), n16);
    if (mmvae_launch_status()) return MMVAE_ERR_LAUNCH;
  }
  if (hipEventRecord(p->staged, p->copy) != hipSuccess) return MMVAE_ERR_LAUNCH;
  p->staged_once = 1;
  return MMVAE_OK;
}

extern "C" int mmvae_input_pipe_commit(mmvae_input_pipe_t* p, const mmvae_input_mod_t* mods, int n_mods,
                                       const void* next_host_packed, mmvae_stream_t stream) {
  MMVAE_CHECK_ARG(p && mods && n_mods > 0 && n_mods <= MMVAE_INPUT_MAX_MODS);
  if (!p->staged_once) return MMVAE_ERR_ARG;      // nothing was prefetched
  hipStream_t st = (hipStream_t)stream;
  if (hipStreamWaitEvent(st, p->staged, 0) != hipSuccess) return MMVAE_ERR_LAUNCH;
  const uint8_t* base = static_cast<const uint8_t*>(p->staging);
  for (int i = 0; i < n_mods; ++i) {
    const mmvae_input_mod_t& m = mods[i];
    int rc;
    if (m.kind == MMVAE_INPUT_IMAGE_U8) {
      if (m.src_off + (size_t)m.n > p->bytes) return MMVAE_ERR_ARG;
      rc = mmvae_expand_image_u8(base + m.src_off, m.dst, m.n, stream);
    } else if (m.kind == MMVAE_INPUT_TEXT_TOKENS) {
      if ((m.src_off & 3) || (m.len_off & 3) || m.src_off + (size_t)m.B * m.T * 4 > p->bytes ||
          m.len_off + (size_t)m.B * 4 > p->bytes)
        return MMVAE_ERR_ARG;
      rc = mmvae_expand_text_tokens(reinterpret_cast<const int32_t*>(base + m.src_off),
                                    reinterpret_cast<const int32_t*>(base + m.len_off), m.dst, m.mask, m.B, m.T, m.V,
                                    stream);
    } else {
      return MMVAE_ERR_ARG;
    }
    if (rc) return rc;
  }
  if (hipEventRecord(p->consumed, st) != hipSuccess) return MMVAE_ERR_LAUNCH;
  return next_host_packed ? mmvae_input_pipe_prefetch(p, next_host_packed) : MMVAE_OK;
}
