#!/usr/bin/env python3
"""Every FP4 kernel once per decode shape, HBM-cold (rotating buffers), for rocprofv3:

    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python tools/profile_ops.py
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d OUT -- python tools/profile_ops.py      (then WRITE_SIZE)

Shapes: Mistral-7B / Llama-3-8B projections (q/o, fused qkv, k/v, fused gate|up, down) + the 4096x4096 bench shape.
Prints the algorithmic bytes of each (op, shape) so the counters can be set against them."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "tests"), REPO]
import torch  # noqa: E402

import hipabi  # noqa: E402

dev = torch.device("cuda", 0)
SHAPES = [(4096, 4096), (1024, 4096), (6144, 4096), (28672, 4096), (4096, 14336)]
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 12
gen = torch.Generator(device=dev).manual_seed(0)
for M, K in SHAPES:
    n = M * K
    R = max(4, min(24, int(600e6 / (n * 0.5625))))
    packed = [torch.randint(0, 256, (n // 2,), dtype=torch.uint8, device=dev, generator=gen) for _ in range(R)]
    absmax = [torch.rand(n // 64, device=dev, generator=gen) * 0.1 + 0.01 for _ in range(R)]
    x = torch.randn(K, device=dev).to(torch.bfloat16)
    x4 = torch.randn(4, K, device=dev).to(torch.bfloat16)
    out = torch.empty(n, dtype=torch.bfloat16, device=dev)
    for i in range(REPS):
        hipabi.gemv(x, packed[i % R], absmax[i % R], M, K, 64)
    for i in range(REPS):
        hipabi.gemm_small(x4, packed[i % R], absmax[i % R], M, K, 64)
    if n <= 4096 * 14336:
        for i in range(REPS):
            hipabi.dequantize(packed[i % R], absmax[i % R], 64, n, torch.bfloat16, out=out)
    torch.cuda.synchronize()
    print(f"{M}x{K}: gemv {n // 2 + n // 16 + 2 * (M + K)} B, gemm_small(4) {n // 2 + n // 16 + 8 * (M + K)} B, "
          f"dequant bf16 {n // 2 + n // 16 + 2 * n} B", flush=True)
    del packed, absmax, out
n = 4096 * 4096
ws = [torch.randn(n, device=dev).to(torch.bfloat16) for _ in range(16)]
for i in range(REPS * 2):
    hipabi.quantize(ws[i % 16], 64)
wf = [torch.randn(n, device=dev) for _ in range(8)]
for i in range(REPS * 2):
    hipabi.quantize(wf[i % 8], 64)
torch.cuda.synchronize()
print(f"quantize 4096x4096: bf16 {2 * n + n // 2 + n // 16} B, f32 {4 * n + n // 2 + n // 16} B")
