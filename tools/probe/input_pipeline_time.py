"""Where the input step's time goes: captured cfg2 step with (a) nothing, (b) expansion of an already staged batch only,
(c) H2D copy on the copy stream only, (d) both (= MultimodalVAE.prefetch_compact / commit_prefetched), per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device("cuda", 0)
tr, desc, meta = bench._build("cfg2", 128, dev, 0, 1, 1)
B, T, V = 128, 32, 27
g = torch.Generator().manual_seed(3)
host = []
for _ in range(4):
    u8 = torch.randint(0, 256, (B, 3, 64, 64), generator=g, dtype=torch.uint8).pin_memory()
    tok = torch.randint(0, V, (B, T), generator=g, dtype=torch.int32).pin_memory()
    lens = torch.randint(3, T + 1, (B,), generator=g, dtype=torch.int32); lens[0] = T
    host.append({"mod_1": {"u8": u8}, "mod_2": {"tokens": tok, "lengths": lens.pin_memory()}})
tr.prefetch_compact(host[0]); tr.commit_prefetched(); torch.cuda.synchronize()
i = [0]
def run(pre, n=300):
    for _ in range(20):
        pre(); tr.fused_step(1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        pre(); tr.fused_step(1)
    t1 = time.perf_counter()      # everything enqueued: (t1 - t0) / n close to the total = the host is the limiter
    torch.cuda.synchronize()
    enq[0] = (t1 - t0) / n * 1e6
    return (time.perf_counter() - t0) / n * 1e6
enq = [0.0]
def both():
    tr.commit_prefetched(); i[0] += 1; tr.prefetch_compact(host[i[0] & 3])
def copy_only():
    i[0] += 1; tr.prefetch_compact(host[i[0] & 3])
def host_only():       # the same python work without any GPU operation of the input step
    i[0] += 1
print(f"replay only          {run(lambda: None):7.1f} us/step   (host enqueue {enq[0]:.1f})")
print(f"+ expansion only     {run(tr.commit_prefetched):7.1f}   (host enqueue {enq[0]:.1f})")
print(f"+ copy only          {run(copy_only):7.1f}   (host enqueue {enq[0]:.1f})")
print(f"+ copy + expansion   {run(both):7.1f}   (host enqueue {enq[0]:.1f})")
host = [tr.pack_compact_pinned(h) for h in host]
tr._staging = {}
tr.prefetch_compact(host[0]); tr.commit_prefetched(); torch.cuda.synchronize()
print(f"+ ONE copy + expansion {run(both):7.1f}   (pack_compact_pinned; host enqueue {enq[0]:.1f})")
print(f"  same, 100 steps      {run(both, 100):7.1f}   (host enqueue {enq[0]:.1f})")
pipe = tr.input_pipe(host[0])
pipe.prefetch(host[0])
def native():
    i[0] += 1; pipe.step(host[i[0] & 3])
print(f"+ native pipe (one call) {run(native):7.1f}   (host enqueue {enq[0]:.1f})")
print(f"replay only (again)  {run(lambda: None):7.1f}   (host enqueue {enq[0]:.1f})")
