"""How does hipGraph schedule a stream-captured DAG?  Synthetic graphs of one-thread spin kernels (no resource
contention): each prints when every node started/ended relative to the first node."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodal_vae_comparison_amd import hipops as H

L = H.lib()
buf = torch.zeros(512, dtype=torch.int64, device="cuda")
names = []


def spin(name, us):
    names.append(name)
    i = len(names) - 1
    rc = L.mmvae_debug_spin(buf.data_ptr() + 16 * i, int(us * 100), torch.cuda.current_stream().cuda_stream)
    assert rc == 0


def run(title, build):
    names.clear()
    buf.zero_()
    main = torch.cuda.Stream()
    with torch.cuda.stream(main):
        build(main, lambda: None)     # eager warm-up (code object load)
        torch.cuda.synchronize()
        names.clear()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=main):
            build(main, lambda: None)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    v = buf.cpu().tolist()
    t0 = min(v[2 * i] for i in range(len(names)))
    print(f"--- {title}")
    for i, n in sorted(enumerate(names), key=lambda x: v[2 * x[0]]):
        print(f"   {n:10s} start {(v[2 * i] - t0) / 100:7.1f}  end {(v[2 * i + 1] - t0) / 100:7.1f}")


side = torch.cuda.Stream()
side2 = torch.cuda.Stream()


def two_chains(first):
    def build(main, _):
        spin("P", 10)
        side.wait_stream(main)
        def a():
            for i in range(5):
                spin(f"A{i}", 10)
        def b():
            with torch.cuda.stream(side):
                for i in range(3):
                    spin(f"B{i}", 10)
        (a, b)[first](); (b, a)[first]()
        main.wait_stream(side)
        spin("J", 10)
    return build


run("P -> main A0..A4 | side B0..B2 -> J   (A captured first)", two_chains(0))
run("P -> main A0..A4 | side B0..B2 -> J   (B captured first)", two_chains(1))


def mid_join(first):
    """two phases with a join + fork in the middle, like enc -> poe -> dec"""
    def build(main, _):
        spin("R", 5)
        side.wait_stream(main)
        for i in range(2):
            spin(f"A{i}", 10)
        with torch.cuda.stream(side):
            for i in range(3):
                spin(f"B{i}", 10)
        main.wait_stream(side)
        spin("P", 10)
        side.wait_stream(main)
        def c():
            for i in range(4):
                spin(f"C{i}", 10)
        def d():
            with torch.cuda.stream(side):
                for i in range(2):
                    spin(f"D{i}", 10)
        (c, d)[first](); (d, c)[first]()
        main.wait_stream(side)
        spin("J", 5)
    return build


run("R -> A|B -> P(main) -> C(main)|D(side) -> J   (C captured first)", mid_join(0))
run("R -> A|B -> P(main) -> C(main)|D(side) -> J   (D captured first)", mid_join(1))


def fuse_on_side(first):
    def build(main, _):
        spin("R", 5)
        side.wait_stream(main)
        for i in range(2):
            spin(f"A{i}", 10)
        with torch.cuda.stream(side):
            for i in range(3):
                spin(f"B{i}", 10)
            side.wait_stream(main)
            spin("P", 10)
        main.wait_stream(side)
        def c():
            with torch.cuda.stream(side):
                for i in range(4):
                    spin(f"C{i}", 10)
        def d():
            for i in range(2):
                spin(f"D{i}", 10)
        (c, d)[first](); (d, c)[first]()
        main.wait_stream(side)
        spin("J", 5)
    return build


run("R -> A(main)|B(side) -> P(side) -> C(side)|D(main) -> J   (C captured first)", fuse_on_side(0))
run("R -> A(main)|B(side) -> P(side) -> C(side)|D(main) -> J   (D captured first)", fuse_on_side(1))


def satisfied_wait(with_wait):
    """main: P, A0..A7; side: B0 (short).  Optionally A4 additionally waits for B0, which finished long before."""
    def build(main, _):
        spin("P", 5)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            spin("B0", 5)
        for i in range(8):
            if i == 4 and with_wait:
                main.wait_stream(side)
            spin(f"A{i}", 10)
        main.wait_stream(side)
        spin("J", 5)
    return build


run("main chain A0..A7, side B0; no extra wait", satisfied_wait(False))
run("main chain A0..A7, side B0; A4 waits for (long finished) B0", satisfied_wait(True))


def late_side_work(n_extra):
    """like the end of a training step: main tail T0,T1 after a join; side gets `n_extra` small kernels after the join
    event, which need one more join at the very end"""
    def build(main, _):
        spin("P", 5)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            for i in range(3):
                spin(f"B{i}", 10)
        for i in range(3):
            spin(f"A{i}", 10)
        main.wait_stream(side)
        if n_extra:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                spin("L", 5)
        spin("T0", 15)
        spin("T1", 15)
        if n_extra:
            main.wait_stream(side)
        spin("E", 1)
    return build


run("tail after join, nothing on side", late_side_work(0))
run("tail after join, one side kernel beside the tail + final join", late_side_work(1))
