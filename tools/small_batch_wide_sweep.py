#!/usr/bin/env python3
"""1..16 activation rows: the default small-batch dispatch next to the one-pass (wide) kernels forced at every workgroup shape
(fp4_hip_set_variant("gemm_small", 1 << 12) + ("gemm_wide", cfg)), HBM-cold, HIP-graph replay, bf16.  usage: M K [M K ...]"""
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


def timeit(replay, launches, reps=7):
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); replay(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / launches)
    return statistics.median(ts)


args = [int(v) for v in sys.argv[1:]]
for M, K in zip(args[0::2], args[1::2]):
    n = M * K
    R = max(8, min(64, int(1.2e9 / (n * 0.5625))))
    gen = torch.Generator(device=dev).manual_seed(0)
    packed = [torch.randint(0, 256, (n // 2,), dtype=torch.uint8, device=dev, generator=gen) for _ in range(R)]
    absmax = [torch.rand(n // 64, device=dev, generator=gen) * 0.1 + 0.01 for _ in range(R)]
    for B in (2, 4, 8, 12, 16):
        x = torch.randn(B, K, device=dev).to(torch.bfloat16)
        res = {}
        hipabi.set_variant("gemm_small", -1)
        hipabi.set_variant("gemm_wide", -1)
        res["default"] = timeit(capture(lambda: [hipabi.gemm_small(x, packed[i], absmax[i], M, K, 64) for i in range(R)]), R)
        for cfg in (-1, 1, 2, 3, 4):
            hipabi.set_variant("gemm_small", 1 << 12)
            hipabi.set_variant("gemm_wide", cfg)
            try:
                res[f"wide cfg {cfg}"] = timeit(capture(lambda: [hipabi.gemm_small(x, packed[i], absmax[i], M, K, 64) for i in range(R)]), R)
            except AssertionError:
                res[f"wide cfg {cfg}"] = float("nan")
        hipabi.set_variant("gemm_small", -1)
        hipabi.set_variant("gemm_wide", -1)
        print(f"{M}x{K} bf16 batch {B:2d}: " + "  ".join(f"{k} {v:6.2f}" for k, v in res.items()), flush=True)
    del packed, absmax
    torch.cuda.empty_cache()
