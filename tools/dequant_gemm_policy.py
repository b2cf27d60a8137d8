#!/usr/bin/env python3
"""Does the store policy of the dequant matter to the GEMM that consumes its output right away (the reference's
batch>1 path)?  dequant (plain vs non-temporal stores) + F.linear, 4096x4096 bf16, HBM-cold rotation, graph replay."""
import os
import statistics
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "torch-bnb-fp4_amd"), os.path.join(REPO, "tests")]
import torch  # noqa: E402

import hipabi  # noqa: E402

M = K = 4096
R, dev, n = 48, torch.device("cuda", 0), M * K
gen = torch.Generator(device=dev).manual_seed(0)
packed = [torch.randint(0, 256, (n // 2,), dtype=torch.uint8, device=dev, generator=gen) for _ in range(R)]
absmax = [torch.rand(n // 64, device=dev, generator=gen) * 0.1 + 0.01 for _ in range(R)]
wbuf = [torch.empty(M, K, dtype=torch.bfloat16, device=dev) for _ in range(R)]


def capture(fn):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    torch.cuda.synchronize()
    return g.replay


def timeit(replay, launches, reps=5):
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); replay(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / launches)
    return statistics.median(ts)


for B in (2, 16, 64, 256, 2048):
    x = torch.randn(B, K, device=dev).to(torch.bfloat16)
    res = {}
    for name, variant in (("plain L4", 4), ("plain L8", 8), ("nt L4", 4 | 256)):
        hipabi.set_variant("dequant", variant)

        def both():
            for i in range(R):
                hipabi.dequantize(packed[i], absmax[i], 64, n, torch.bfloat16, out=wbuf[i].view(-1))
                torch.nn.functional.linear(x, wbuf[i])

        def dq_only():
            for i in range(R):
                hipabi.dequantize(packed[i], absmax[i], 64, n, torch.bfloat16, out=wbuf[i].view(-1))

        res[name] = (timeit(capture(both), R), timeit(capture(dq_only), R))
    print(f"batch {B:5d}: " + "   ".join(f"{k}: dequant {v[1]:6.2f} + gemm -> {v[0]:7.2f} us" for k, v in res.items()), flush=True)
