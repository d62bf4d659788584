#!/usr/bin/env python
"""How long does hipGraphLaunch keep the host / leave the GPU idle?  A chain of N small kernels captured into one graph:
host time of replay(), GPU time of the replay, and back-to-back replays (does launch k+1 overlap execution k?)."""
import sys, time
import torch
dev = torch.device("cuda")
x = torch.randn(1 << 22, device=dev)
for N, work in ((400, 1), (400, 16), (1600, 1)):
    ys = [torch.empty_like(x) for _ in range(2)]
    def chain():
        a = x
        for i in range(N):
            b = ys[i & 1]
            for _ in range(work):
                torch.mul(a, 1.0001, out=b)
            a = b
    chain(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        chain()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter(); chain(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"N={N} x{work}: eager   host {1e3 * (t1 - t0):7.2f} ms, done {1e3 * (t2 - t0):7.2f} ms")
    for reps in (1, 4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record()
        for _ in range(reps):
            g.replay()
        e1.record(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"N={N} x{work}: {reps} replay host {1e3 * (t1 - t0):7.2f} ms, done {1e3 * (t2 - t0):7.2f} ms, GPU events {e0.elapsed_time(e1):7.2f} ms")

# two different graphs alternating with an eager copy in front of each (the training loop's pattern: static-input copy, replay)
print("--- alternating graphs A (400 x 8 kernels) and B (800 x 4), eager copy before each replay")
def mk(N, work):
    ys = [torch.empty_like(x) for _ in range(2)]
    inp = torch.empty_like(x)
    def chain():
        a = inp
        for i in range(N):
            b = ys[i & 1]
            for _ in range(work):
                torch.mul(a, 1.0001, out=b)
            a = b
    chain(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        chain()
    return g, inp
gA, inA = mk(400, 8)
gB, inB = mk(800, 4)
def one(g):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
tA, tB = one(gA), one(gB)
for reps in (1, 4, 8):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    host = []
    t0 = time.perf_counter(); e0.record()
    for _ in range(reps):
        for g, inp in ((gA, inA), (gB, inB)):
            h0 = time.perf_counter()
            inp.copy_(x)
            g.replay()
            host.append(1e3 * (time.perf_counter() - h0))
    e1.record(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{reps} x (A, B): single A {tA:.2f} ms, single B {tB:.2f} ms -> sum {reps * (tA + tB):.2f}; GPU events {e0.elapsed_time(e1):.2f} ms; "
          f"host total {1e3 * (t1 - t0):.2f} ms; per launch host ms: {' '.join(f'{h:.2f}' for h in host)}")
