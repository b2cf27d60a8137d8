#!/usr/bin/env python3
"""17..64 activation rows: fp4_hip_gemm_small (ceil(B/16) launches, each streaming the weight once) next to the reference's
batch > 1 path (dequantise to bf16 + hipBLASLt GEMM), HBM-cold rotation, HIP-graph replay.  Where is the crossover?"""
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


shapes = [(4096, 4096), (14336, 4096), (4096, 14336), (28672, 4096)]
if len(sys.argv) > 2 and sys.argv[1].isdigit():
    shapes = [(int(sys.argv[1]), int(sys.argv[2]))]
for M, K in shapes:
    n = M * K
    R = max(6, min(48, int(1.0e9 / (n * 0.5625))))
    gen = torch.Generator(device=dev).manual_seed(0)
    packed = [torch.randint(0, 256, (n // 2,), dtype=torch.uint8, device=dev, generator=gen) for _ in range(R)]
    absmax = [torch.rand(n // 64, device=dev, generator=gen) * 0.1 + 0.01 for _ in range(R)]
    wbuf = [torch.empty(M, K, dtype=torch.bfloat16, device=dev) for _ in range(min(R, 8))]
    x1 = torch.randn(K, device=dev).to(torch.bfloat16)
    t_gemv = timeit(capture(lambda: [hipabi.gemv(x1, packed[i], absmax[i], M, K, 64) for i in range(R)]), R)
    print(f"{M}x{K} bf16 (R={R}): batch-1 gemv {t_gemv:.2f} us", flush=True)
    for B in (2, 8, 16, 24, 32, 48, 64, 96, 128):
        x = torch.randn(B, K, device=dev).to(torch.bfloat16)
        t_small, t_cfg = None, {}
        if B <= 128:
            t_small = timeit(capture(lambda: [hipabi.gemm_small(x, packed[i], absmax[i], M, K, 64) for i in range(R)]), R)
        if 16 < B <= 64 and "--cfgs" in sys.argv:
            for cfg in (0, 1, 2, 3, 4):
                hipabi.set_variant("gemm_wide", cfg)
                t_cfg[cfg] = timeit(capture(lambda: [hipabi.gemm_small(x, packed[i], absmax[i], M, K, 64) for i in range(R)]), R)
            hipabi.set_variant("gemm_wide", -1)

        def ref():
            for i in range(R):
                w = wbuf[i % len(wbuf)]
                hipabi.dequantize(packed[i], absmax[i], 64, n, torch.bfloat16, out=w.view(-1))
                torch.nn.functional.linear(x, w)
        t_ref = timeit(capture(ref), R)
        t_ws = None
        if B <= 64:
            ws = torch.empty(max(16, -(-(K // 64) // 8) * B * M * 4), dtype=torch.uint8, device=dev)
            t_ws = timeit(capture(lambda: [hipabi.gemm_small_ws(x, packed[i], absmax[i], M, K, 64, workspace=ws) for i in range(R)]), R)
        s = f"{t_small:7.2f} us" if t_small is not None else "      - "
        print(f"   rows {B:3d}: gemm_small {s}   dequant + hipBLASLt {t_ref:7.2f} us" + (f"   ratio {t_ref / t_small:4.2f}x" if t_small else "") + (f"   with workspace {t_ws:7.2f} us" if t_ws else "")
              + ("   [16-row launches / one pass with 16 / 32 / 64 / 128 rows per workgroup: " + " ".join(f"{t_cfg[c]:.2f}" for c in sorted(t_cfg)) + "]" if t_cfg else ""), flush=True)
    del packed, absmax, wbuf
    torch.cuda.empty_cache()
