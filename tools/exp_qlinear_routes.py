#!/usr/bin/env python3
"""Round 5: the two GEMM routes of the qlinear* ops side by side (direct hipBLASLt with cached plans vs at::linear) - DEVICE time per call from
HIP-graph replays and HOST time per eager call, from 2 rows to prefill sizes, so that the cached-plan route is shown not to cost GEMM speed."""
import os
import statistics
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "torch-bnb-fp4_amd")]
import torch  # noqa: E402

import torch_bnb_fp4 as P  # noqa: E402

dev = torch.device("cuda", 0)


def graph_us(fn, n=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / n)
    return statistics.median(ts[2:])


def host_us(fn, n=600):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return t


for dt in (torch.bfloat16, torch.float16, torch.float32):
    for M, K in ((4096, 4096), (14336, 4096), (4096, 14336)):
        g = torch.Generator(device=dev).manual_seed(1)
        packed = torch.randint(0, 256, (M * K // 2, 1), dtype=torch.uint8, device=dev, generator=g)
        absmax = torch.rand(M * K // 64, device=dev, generator=g) * 0.02 + 0.002
        bias = torch.randn(M, device=dev, generator=g).to(dt)
        for rows in (2, 16, 128, 1024, 4096):
            if dt == torch.float32 and rows > 1024:
                continue
            x = torch.randn(rows, K, device=dev, generator=g).to(dt)
            res = {}
            with torch.inference_mode():
                for route in ("hipblaslt", "aten"):
                    P.ext.set_qlinear_gemm(route)
                    fn = lambda: P.ext.qlinear_bias(x, packed, absmax, M, K, 64, bias)
                    res[route] = (graph_us(fn), host_us(fn))
            P.ext.set_qlinear_gemm("hipblaslt")
            d, a = res["hipblaslt"], res["aten"]
            print(f"{str(dt).replace('torch.', ''):9s} {M:5d}x{K:<5d} rows {rows:5d}: device us/call direct {d[0]:9.1f}  at::linear {a[0]:9.1f}  ({d[0] / a[0]:.3f}x)   "
                  f"host us/call direct {d[1]:6.1f}  at::linear {a[1]:6.1f}", flush=True)
        del packed, absmax
