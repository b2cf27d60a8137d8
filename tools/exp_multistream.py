#!/usr/bin/env python3
"""Experiment: R independent 4096x4096 launches (dequant -> bf16, or GEMV) captured into one HIP graph from 1 / 2 / 4 streams
(fork / join inside the capture), HBM-cold rotation; us per launch = replay time / R.  Tells how much of the per-launch time is
the dependent-kernel boundary that independent launches could overlap."""
import os
import statistics
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "torch-bnb-fp4_amd"), os.path.join(REPO, "tests")]
import torch  # noqa: E402

import hipabi  # noqa: E402

M = K = 4096
R, dev, n = 64, torch.device("cuda", 0), M * K
gen = torch.Generator(device=dev).manual_seed(0)
packed = [torch.randint(0, 256, (n // 2,), dtype=torch.uint8, device=dev, generator=gen) for _ in range(R)]
absmax = [torch.rand(n // 64, device=dev, generator=gen) * 0.1 + 0.01 for _ in range(R)]
outs = [torch.empty(n, dtype=torch.bfloat16, device=dev) for _ in range(R)]
x = torch.randn(K, device=dev).to(torch.bfloat16)
ys = [torch.empty(M, dtype=torch.bfloat16, device=dev) for _ in range(R)]


def capture_multi(launch, nstreams):
    main = torch.cuda.Stream()
    side = [torch.cuda.Stream() for _ in range(nstreams - 1)]
    streams = [main] + side
    with torch.cuda.stream(main):
        for i in range(R):
            launch(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=main):
            ev = torch.cuda.Event()
            ev.record(main)
            for s in side:
                s.wait_event(ev)
            for i in range(R):
                with torch.cuda.stream(streams[i % nstreams]):
                    launch(i)
            for s in side:
                e = torch.cuda.Event()
                e.record(s)
                main.wait_event(e)
    torch.cuda.synchronize()
    return g


def timeit(g, reps=9):
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / R)
    return statistics.median(ts), min(ts)


dq = lambda i: hipabi.dequantize(packed[i], absmax[i], 64, n, torch.bfloat16, out=outs[i])
gv = lambda i: hipabi.gemv(x, packed[i], absmax[i], M, K, 64)
for name, fn, nbytes in (("dequant bf16", dq, n // 2 + n // 16 + 2 * n), ("gemv bf16", gv, n // 2 + n // 16 + 4 * K)):
    for ns in (1, 2, 3, 4):
        med, mn = timeit(capture_multi(fn, ns))
        print(f"{name} 4096x4096, {R} independent launches over {ns} stream(s): {med:6.3f} us per launch (min {mn:6.3f}) = {nbytes / med / 1e3:6.0f} GB/s", flush=True)
