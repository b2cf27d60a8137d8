#!/usr/bin/env python3
"""Fused small-batch FP4 GEMM (fp4_hip_gemm_small) vs the reference's batch>1 path (dequant + hipBLASLt GEMM),
4096x4096 bf16, HBM-cold rotation over 64 weights, HIP-graph replay."""
import os
import statistics
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "torch-bnb-fp4_amd"), os.path.join(REPO, "tests")]
import torch  # noqa: E402

import hipabi  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
QUICK = "--quick" in sys.argv  # default dispatch only, next to the batch-1 GEMV on the same weights (the "same bytes" yardstick)
dev, n = torch.device("cuda", 0), M * K
R = max(8, min(64, int(1.2e9 / (n * 0.5625))))
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


def timeit(replay, launches, reps=7):
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); replay(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / launches)
    return statistics.median(ts)


if QUICK:
    x1 = torch.randn(K, device=dev).to(torch.bfloat16)
    t_gemv = timeit(capture(lambda: [hipabi.gemv(x1, packed[i], absmax[i], M, K, 64) for i in range(R)]), R)
    line = [f"{M}x{K} bf16: gemv {t_gemv:6.2f} us |"]
    for B in (2, 4, 8, 16):
        x = torch.randn(B, K, device=dev).to(torch.bfloat16)
        t = timeit(capture(lambda: [hipabi.gemm_small(x, packed[i], absmax[i], M, K, 64) for i in range(R)]), R)
        line.append(f"batch {B:2d}: {t:6.2f} us ({t / t_gemv:4.2f}x)")
    print("  ".join(line), flush=True)
    sys.exit(0)

for B in (1, 2, 3, 4, 8, 12, 16):
    x = torch.randn(B, K, device=dev).to(torch.bfloat16)
    per_kernel = {}
    for kname, kv in (("valu", 0), ("mfma rt1", 1 | (1 << 4)), ("mfma rt2", 1 | (2 << 4)), ("mfma rt1 no-xstage", 1 | (1 << 4) | (1 << 9)),
                      ("mfma persistent", 1 | (2 << 10)), ("mfma one-shot", 1 | (1 << 10))):
        if kv == 0 and B > 8:
            continue
        hipabi.set_variant("gemm_small", kv)
        per_kernel[kname] = timeit(capture(lambda: [hipabi.gemm_small(x, packed[i], absmax[i], M, K, 64) for i in range(R)]), R)
    hipabi.set_variant("gemm_small", -1)
    fused = capture(lambda: [hipabi.gemm_small(x, packed[i], absmax[i], M, K, 64) for i in range(R)])

    def ref():
        for i in range(R):
            hipabi.dequantize(packed[i], absmax[i], 64, n, torch.bfloat16, out=wbuf[i].view(-1))
            torch.nn.functional.linear(x, wbuf[i])

    t_f, t_r = timeit(fused, R), timeit(capture(ref), R)
    print(f"{M}x{K} bf16 batch {B:2d}: fused {t_f:6.2f} us ({', '.join(f'{k} {v:.2f}' for k, v in per_kernel.items())})   "
          f"dequant+hipBLASLt {t_r:6.2f} us   speedup {t_r / t_f:4.1f}x", flush=True)
