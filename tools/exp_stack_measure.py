#!/usr/bin/env python3
"""Why two measurements of the steady-state GEMV (one launch over a 262144 x 4096 stack) disagree: the data (uniform
random bytes vs quantised Gaussian weights) x the timing method (events around eager launches vs a 12-launch graph)."""
import os
import statistics
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "tests"), REPO]
import torch  # noqa: E402

import hipabi  # noqa: E402

dev = torch.device("cuda", 0)
rows, K = 262144, 4096
n = rows * K
nbytes = n // 2 + n // 16 + 2 * (rows + K)
x = torch.randn(K, device=dev).to(torch.bfloat16)


def eager(p, a, reps=8):
    for _ in range(2):
        hipabi.gemv(x, p, a, rows, K, 64)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        hipabi.gemv(x, p, a, rows, K, 64)
        ev[i + 1].record()
    torch.cuda.synchronize()
    return statistics.median(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(reps))


DQ_OUT = torch.empty(n, dtype=torch.bfloat16, device=dev)


def dq_eager(p, a, reps=20):
    for _ in range(2):
        hipabi.dequantize(p, a, 64, n, torch.bfloat16, out=DQ_OUT)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        hipabi.dequantize(p, a, 64, n, torch.bfloat16, out=DQ_OUT)
        ev[i + 1].record()
    torch.cuda.synchronize()
    return statistics.median(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(reps))


def graph(p, a, launches=12, reps=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        hipabi.gemv(x, p, a, rows, K, 64)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            for _ in range(launches):
                hipabi.gemv(x, p, a, rows, K, 64)
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / launches)
    return statistics.median(out)


gen = torch.Generator(device=dev).manual_seed(0)
data = {}
data["uniform bytes"] = (torch.randint(0, 256, (n // 2,), dtype=torch.uint8, device=dev, generator=gen),
                         torch.rand(n // 64, device=dev, generator=gen) * 0.1 + 0.01)
w = torch.randn(4096 * 4096, device=dev, generator=gen).to(torch.bfloat16)
qp, qa = hipabi.quantize(w, 64)
data["quantised gaussian (one matrix x 64)"] = (qp.reshape(-1).repeat(64), qa.reshape(-1).repeat(64))
data["zeros"] = (torch.zeros(n // 2, dtype=torch.uint8, device=dev), torch.ones(n // 64, device=dev))
graphs = {}
for name, (p, a) in data.items():  # 200 ms of the kernel first: clocks settled before anything is timed
    for _ in range(2000):
        hipabi.gemv(x, p, a, rows, K, 64)
    torch.cuda.synchronize()
for rnd in range(4):  # interleaved rounds: drift shows as disagreement between the rounds
    for name, (p, a) in data.items():
        e, g = eager(p, a, 40), graph(p, a, 12, 9)
        d = dq_eager(p, a)
        dqb = n // 2 + n // 16 + 2 * n
        print(f"round {rnd} {name:40s} dequant eager {d:7.1f} us = {dqb / d / 1e3:6.0f} GB/s", flush=True)
        print(f"round {rnd} {name:40s} eager {e:7.1f} us = {nbytes / e / 1e3:6.0f} GB/s   graph(12) {g:7.1f} us = {nbytes / g / 1e3:6.0f} GB/s", flush=True)
