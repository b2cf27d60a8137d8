#!/usr/bin/env python3
"""A few HBM-cold launches of the FP4 quantiser (4096x4096 bf16 -> packed + absmax) for rocprofv3 counter passes."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "torch-bnb-fp4_amd"), os.path.join(REPO, "tests")]
import torch  # noqa: E402

import hipabi  # noqa: E402

dev = torch.device("cuda", 0)
n = 4096 * 4096
ws = [torch.randn(n, device=dev).to(torch.bfloat16) for _ in range(24)]
for rep in range(4):
    for w in ws:
        hipabi.quantize(w, 64)
torch.cuda.synchronize()
print("done")
