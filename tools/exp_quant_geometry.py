#!/usr/bin/env python3
"""Round 5: the quantiser's two kernels side by side - the persistent one (512-thread workgroups, one tile in flight behind the one
being ranked; `fp4_hip_set_variant("quantize", k)` = k workgroups per CU) and the one-shot tiles kernel (256-thread workgroups, all of
a lane's loads up front; variant 1000 + loads per lane) - per launch (HBM-cold rotation) and over a stack in one launch, at the
shapes a decoder quantises at load time."""
import os
import statistics
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "torch-bnb-fp4_amd"), os.path.join(REPO, "tests")]
import torch  # noqa: E402

import hipabi  # noqa: E402

dev = torch.device("cuda", 0)


def capture(fn):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    torch.cuda.synchronize()
    return g.replay


def timeit(replay, launches, reps=9, warm=4):
    ts = []
    for i in range(reps + warm):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); replay(); b.record(); b.synchronize()
        if i >= warm:
            ts.append(a.elapsed_time(b) * 1e3 / launches)
    return statistics.median(ts)


for dt, name, isz in ((torch.bfloat16, "bf16", 2), (torch.float32, "f32", 4)):
    for M, K in ((1024, 4096), (4096, 4096), (14336, 4096)):
        n = M * K
        R = max(4, min(32, int(1.2e9 / (n * isz))))
        big = (torch.randn(R * n, device=dev) * 0.02).to(dt)
        nbytes = n * isz + n // 2 + 4 * (n // 64)
        for label, variant in (("persistent x4", 4), ("persistent x8", 8), ("tiles 1 load", 1001), ("tiles 2 loads", 1002), ("tiles 4 loads", 1004)):
            hipabi.set_variant("quantize", variant)
            cold = capture(lambda: [hipabi.quantize(big[i * n:(i + 1) * n], 64) for i in range(R)])
            c = timeit(cold, R)
            stack = capture(lambda: [hipabi.quantize(big, 64) for _ in range(3)])
            s_us = timeit(stack, 3)
            print(f"quantize {name} {M:5d}x{K} {label:14s} per launch {c:7.2f} us = {nbytes / c / 1e3:6.0f} GB/s   stack of {R:2d}: {s_us / R:7.2f} us per matrix = "
                  f"{R * nbytes / s_us / 1e3:6.0f} GB/s", flush=True)
        hipabi.set_variant("quantize", 0)
        del big
