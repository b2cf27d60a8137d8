#!/usr/bin/env python3
"""Host-side cost of one decode-layer call (eager, no graph): where the microseconds go between Python and the kernel."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "torch-bnb-fp4_amd")]
import torch  # noqa: E402
from torch import nn  # noqa: E402

import torch_bnb_fp4 as pkg  # noqa: E402

dev = torch.device("cuda", 0)
M = K = 4096
lin = nn.Linear(K, M).to(dev).to(torch.bfloat16)
fp4 = pkg.TorchFP4Linear(pkg.swap_linear_with_bnb_linear(lin).to(dev))
qd = fp4.quant_data
x = torch.randn(1, K, device=dev, dtype=torch.bfloat16)
fp4(x)


def bench(name, fn, n=3000):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"{name:46s} host issue {t_issue / n * 1e6:6.2f} us/call   with GPU drain {t_all / n * 1e6:6.2f} us/call", flush=True)


with torch.inference_mode():
    bench("ext.gemv_fp4_bias (raw extension op)", lambda: pkg.ext.gemv_fp4_bias(x, qd._B_t, qd.absmax, qd.code, 64, qd.qtype, qd._shape_list, qd.bias))
    bench("QuantData.forward", lambda: qd.forward(x))
    bench("TorchFP4Linear.__call__", lambda: fp4(x))
    bench("functional gemm_4bit_inference (dtype lookup)", lambda: pkg.gemm_4bit_inference(x, qd._B_t, qd.absmax, qd.code, 64, torch.bfloat16, qd.quant_state.shape))
    bench("dense nn.Linear bf16 (hipBLASLt)", lambda: lin(x))
    bench("torch.add (one trivial torch op)", lambda: torch.add(x, x))
    xb = torch.randn(4, K, device=dev, dtype=torch.bfloat16)
    bench("TorchFP4Linear batch 4 (dequant + GEMM)", lambda: fp4(xb), n=1000)
