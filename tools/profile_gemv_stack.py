#!/usr/bin/env python3
"""Steady-state GEMV for counter collection: ONE tall weight (rows x K, default 262144 x 4096 = 64 stacked 4096 x 4096
matrices, 604 MB: far beyond the Infinity Cache) streamed by a handful of launches, so that launch boundaries are
negligible and rocprofv3 --pmc counters describe the kernel's steady state.

    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY ... -- python3 tools/profile_gemv_stack.py [rows K reps]
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "tests"), REPO]
import torch  # noqa: E402

import hipabi  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
dev = torch.device("cuda", 0)
n = rows * K
gen = torch.Generator(device=dev).manual_seed(0)
packed = torch.randint(0, 256, (n // 2,), dtype=torch.uint8, device=dev, generator=gen)
absmax = torch.rand(n // 64, device=dev, generator=gen) * 0.1 + 0.01
x = torch.randn(K, device=dev).to(torch.bfloat16)
for _ in range(2):
    hipabi.gemv(x, packed, absmax, rows, K, 64)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
ev[0].record()
for i in range(reps):
    hipabi.gemv(x, packed, absmax, rows, K, 64)
    ev[i + 1].record()
torch.cuda.synchronize()
us = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(reps))
nbytes = n // 2 + n // 16 + 2 * (rows + K)
print(f"gemv {rows}x{K} bf16: median {us[len(us) // 2]:.1f} us = {nbytes / us[len(us) // 2] / 1e3:.0f} GB/s (algorithmic {nbytes} B)")
