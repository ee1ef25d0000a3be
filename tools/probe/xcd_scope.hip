// How long does a hand-over between two workgroups take at device (agent) scope, and how long when both sit on the same
// XCD and only bypass their L1 (sc0 loads / L2-executed atomics)?  And: which XCD does workgroup i of a 1-D grid run on?
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/xcd_scope.hip -o tools/probe/xcd_scope ; run on the GPU box
// Measured (MI355X, idle GPU): workgroup i of a grid (linear order x, y, z) runs on XCC i % 8, no exception in 2000;
// publish -> seen at agent scope 0.50 us (worst 0.77); a dependent agent-scope fetch-add 0.24 us, a dependent agent-scope
// load 0.09 us.  A plain store is NOT seen by an sc0 (L1-bypassing) load of a workgroup on the same XCD within the
// polling bound: the XCD-local hand-over would need more than the load's cache bits, and at 0.25-0.5 us the device-scope
// one is not what makes the ResNet engine's steps take 1.5-2 us -- the memory system loaded with operand traffic is.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}
__device__ __forceinline__ unsigned ld_sc0(const unsigned* base, unsigned byte_off) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
  return __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 1);
}
__global__ void who(unsigned* ids) {
  if (threadIdx.x == 0) ids[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = xcc_id();
}
// workgroups [0, 8) publish a time stamp after `delay` polls of the clock, workgroups [8, 16) wait for the one of
// workgroup (i - 8) (same XCD under round-robin placement); mode 0: agent-scope store / load, 1: plain store + sc0 load
__global__ void pingpong(unsigned* flag, long long* out, unsigned* xcc, int mode, int round) {
  const int w = blockIdx.x;
  if (threadIdx.x != 0) return;
  xcc[w] = xcc_id();
  if (w < 8) {
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 2000) {}          // 20 us: the pollers are surely running
    const unsigned stamp = (unsigned)(wall_clock64() & 0x7fffffff) | 1u;
    if (mode == 0) __hip_atomic_store(flag + w * 64 + round, stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else flag[w * 64 + round] = stamp;
  } else {
    unsigned v = 0;
    const unsigned* p = flag + (w - 8) * 64 + round;
    for (int it = 0; v == 0 && it < 4000000; ++it) {          // bounded: a stale read must not hang the box
      if (mode == 0) v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else v = ld_sc0(flag, (unsigned)(((w - 8) * 64 + round) * 4));
      asm volatile("" ::: "memory");                          // (the buffer-load builtin is a plain read: keep it in the loop)
    }
    if (v == 0) { out[w - 8] = -1; return; }
    const unsigned now = (unsigned)(wall_clock64() & 0x7fffffff);
    out[w - 8] = (long long)((now - (v & ~1u)) & 0x7fffffff);
  }
}
// a chain of dependent fetch-adds on one address by one thread: round trip of an atomic at agent / workgroup scope
__global__ void atomic_chain(unsigned* ctr, long long* out, int n) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  long long t0 = wall_clock64();
  unsigned v = 0;
  for (int i = 0; i < n; ++i) v = __hip_atomic_fetch_add(ctr + (v & 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  long long t1 = wall_clock64();
  for (int i = 0; i < n; ++i) v = __hip_atomic_fetch_add(ctr + 2 + (v & 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  long long t2 = wall_clock64();
  out[0] = t1 - t0;
  out[1] = t2 - t1;
  out[2] = v;
}
// dependent loads: agent scope vs sc0 vs plain (L1 hit after the first)
__global__ void load_chain(unsigned* buf, long long* out, int n) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  unsigned v = 0;
  long long t0 = wall_clock64();
  for (int i = 0; i < n; ++i) v = __hip_atomic_load(buf + (v & 1023), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  long long t1 = wall_clock64();
  for (int i = 0; i < n; ++i) { v = ld_sc0(buf, (v & 1023) * 4); asm volatile("" ::: "memory"); }
  long long t2 = wall_clock64();
  for (int i = 0; i < n; ++i) v = buf[v & 1023];
  long long t3 = wall_clock64();
  out[0] = t1 - t0; out[1] = t2 - t1; out[2] = t3 - t2; out[3] = v;
}
int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  unsigned *ids, *flag, *xcc, *ctr, *buf;
  long long* out;
  hipMalloc(&ids, 4096 * 4); hipMalloc(&flag, 8 * 64 * 4); hipMalloc(&xcc, 64 * 4); hipMalloc(&ctr, 64); hipMalloc(&buf, 4096);
  hipMalloc(&out, 64 * 8);
  hipMemset(flag, 0, 8 * 64 * 4); hipMemset(ctr, 0, 64); hipMemset(buf, 0, 4096);
  std::vector<unsigned> h(4096);
  who<<<dim3(64), 64>>>(ids);
  hipMemcpy(h.data(), ids, 64 * 4, hipMemcpyDeviceToHost);
  printf("1-D grid of 64, XCC of workgroup i:"); for (int i = 0; i < 64; ++i) printf(" %u", h[i]); printf("\n"); fflush(stdout);
  who<<<dim3(4, 3, 5), 64>>>(ids);
  hipMemcpy(h.data(), ids, 60 * 4, hipMemcpyDeviceToHost);
  printf("grid (4,3,5), XCC in linear order x + 4 y + 12 z:"); for (int i = 0; i < 60; ++i) printf(" %u", h[i]); printf("\n");
  who<<<dim3(2000), 256>>>(ids);
  hipMemcpy(h.data(), ids, 2000 * 4, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 2000; ++i) bad += (h[i] != (unsigned)((i + h[0]) % 8));
  printf("1-D grid of 2000 x 256 threads: %d workgroups off the round-robin pattern (first on XCC %u)\n", bad, h[0]);
  for (int mode = 0; mode < 2; ++mode) {
    double sum = 0; int n = 0; long long worst = 0;
    std::vector<long long> o(8); std::vector<unsigned> x(16);
    for (int r = 0; r < 20; ++r) {
      pingpong<<<16, 64>>>(flag, out, xcc, mode, r + 20 * mode);
      hipDeviceSynchronize();
      hipMemcpy(o.data(), out, 8 * 8, hipMemcpyDeviceToHost);
      hipMemcpy(x.data(), xcc, 16 * 4, hipMemcpyDeviceToHost);
      for (int i = 0; i < 8; ++i) {
        if (x[i] != x[i + 8]) printf("  (pair %d on XCCs %u / %u)\n", i, x[i], x[i + 8]);
        if (o[i] < 0) { printf("  (pair %d: never seen)\n", i); continue; }
        sum += o[i]; ++n; if (o[i] > worst) worst = o[i];
      }
    }
    printf("publish -> seen, %s: mean %.2f us, worst %.2f us\n", mode == 0 ? "agent-scope store + agent-scope load" : "plain store + sc0 load (same XCD)", sum / n * 0.01, worst * 0.01);
  }
  std::vector<long long> o(4);
  atomic_chain<<<1, 64>>>(ctr, out, 1000); hipDeviceSynchronize();
  hipMemcpy(o.data(), out, 32, hipMemcpyDeviceToHost);
  printf("dependent fetch-add round trip: agent scope %.2f us, workgroup scope (executed in the XCD's L2) %.2f us\n", o[0] * 0.01 / 1000, o[1] * 0.01 / 1000);
  load_chain<<<1, 64>>>(buf, out, 1000); hipDeviceSynchronize();
  hipMemcpy(o.data(), out, 32, hipMemcpyDeviceToHost);
  printf("dependent load round trip: agent scope %.2f us, sc0 (L1 bypass) %.2f us, plain %.2f us\n", o[0] * 0.01 / 1000, o[1] * 0.01 / 1000, o[2] * 0.01 / 1000);
  return 0;
}
