#!/usr/bin/env python3
"""Kernel-geometry sweep on one GPU: times every dequant / GEMV variant HBM-cold (rotating over R
distinct weights, HIP-graph replay) and prints a table.  Usage: python tools/sweep.py [dequant|gemv|all] [M K]"""
import os
import statistics
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "torch-bnb-fp4_amd"), os.path.join(REPO, "tests")]
import torch  # noqa: E402

import hipabi  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "all"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
K = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
R = max(8, min(64, int(700e6 / (M * K * 0.5625))))
dev = torch.device("cuda", 0)
n = M * K
gen = torch.Generator(device=dev).manual_seed(0)
packed = [torch.randint(0, 256, (n // 2,), dtype=torch.uint8, device=dev, generator=gen) for _ in range(R)]
absmax = [torch.rand(n // 64, device=dev, generator=gen) * 0.1 + 0.01 for _ in range(R)]


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
    return statistics.median(ts), min(ts)


if what in ("dequant", "all"):
    for dt, name, isz in ((torch.bfloat16, "bf16", 2), (torch.float16, "f16", 2), (torch.float32, "f32", 4)):
        outs = [torch.empty(n, dtype=dt, device=dev) for _ in range(min(R, 32))]
        nbytes = n // 2 + 4 * (n // 64) + n * isz
        for variant in ((-1, 1, 2, 4, 8, 16, 1 | 256, 2 | 256, 4 | 256, 8 | 256, 16 | 256) if isz == 4 else (-1, 1, 2, 4, 8, 1 | 256, 2 | 256, 4 | 256)):  # what the library builds
            hipabi.set_variant("dequant", variant)
            cold = capture(lambda: [hipabi.dequantize(packed[i], absmax[i], 64, n, dt, out=outs[i % len(outs)]) for i in range(R)])
            hot = capture(lambda: [hipabi.dequantize(packed[0], absmax[0], 64, n, dt, out=outs[0]) for i in range(R)])
            c, cm = timeit(cold, R)
            h, hm = timeit(hot, R)
            print(f"dequant {name} {M}x{K} loads={variant & 255 if variant >= 0 else -1:2d} nt={variant >> 8 if variant >= 0 else -1} cold {c:7.2f} us (min {cm:7.2f}) = {nbytes / c / 1e3:7.0f} GB/s   hot {h:7.2f} us = {nbytes / h / 1e3:7.0f} GB/s", flush=True)
        del outs
    hipabi.set_variant("dequant", -1)

if what in ("gemv", "all"):
    for dt, name, isz in ((torch.bfloat16, "bf16", 2), (torch.float16, "f16", 2)):
        x = torch.randn(K, device=dev).to(dt)
        ys = [torch.empty(M, dtype=dt, device=dev) for _ in range(R)]
        nbytes = n // 2 + 4 * (n // 64) + (K + M) * isz
        for (r, w, u) in [(1, 4, 2), (1, 8, 2)]:  # the LDS geometries the library builds
            hipabi.set_variant("gemv", r | (w << 8) | (u << 16))
            reps = 4
            cold = capture(lambda: [hipabi.gemv(x, packed[i % R], absmax[i % R], M, K, 64) for i in range(R * reps)])
            hot = capture(lambda: [hipabi.gemv(x, packed[0], absmax[0], M, K, 64) for i in range(R * reps)])
            c, cm = timeit(cold, R * reps)
            h, hm = timeit(hot, R * reps)
            print(f"gemv {name} {M}x{K} rows={r} waves={w:2d} unroll={u} cold {c:6.2f} us (min {cm:6.2f}) = {nbytes / c / 1e3:6.0f} GB/s   hot {h:6.2f} us = {nbytes / h / 1e3:6.0f} GB/s", flush=True)
    x = torch.randn(K, device=dev)
    hipabi.set_variant("gemv", 0)
    cold = capture(lambda: [hipabi.gemv(x, packed[i % R], absmax[i % R], M, K, 64) for i in range(R * 2)])
    c, cm = timeit(cold, R * 2)
    print(f"gemv f32 (LDS-x kernel) {M}x{K} cold {c:6.2f} us = {(n // 2 + 4 * (n // 64) + (K + M) * 4) / c / 1e3:6.0f} GB/s", flush=True)
    for it in (1, 2, 4):
        hipabi.set_variant("gemv", (3 << 24) | it)
        cold = capture(lambda: [hipabi.gemv(x, packed[i % R], absmax[i % R], M, K, 64) for i in range(R * 2)])
        c, cm = timeit(cold, R * 2)
        print(f"gemv f32 reg-x iters={it} {M}x{K} cold {c:6.2f} us = {(n // 2 + 4 * (n // 64) + (K + M) * 4) / c / 1e3:6.0f} GB/s", flush=True)
    hipabi.set_variant("gemv", -1)
    cold = capture(lambda: [hipabi.gemv(x, packed[i % R], absmax[i % R], M, K, 64) for i in range(R * 2)])
    c, cm = timeit(cold, R * 2)
    print(f"gemv f32 {M}x{K} cold {c:6.2f} us = {(n // 2 + 4 * (n // 64) + (K + M) * 4) / c / 1e3:6.0f} GB/s", flush=True)

if what == "regx":
    # register-x GEMV geometries at one shape: iters x {auto K split, 8-way split over 8 waves}; variant word = 1<<24 | ks<<8 | iters
    for dt, name, isz in ((torch.bfloat16, "bf16", 2),):
        x = torch.randn(K, device=dev).to(dt)
        nbytes = n // 2 + 4 * (n // 64) + (K + M) * isz
        res = []
        for ks in (0, 5, 6, 7, 8):
            for it in (1, 2, 4):
                hipabi.set_variant("gemv", (1 << 24) | (ks << 8) | it)
                try:
                    cold = capture(lambda: [hipabi.gemv(x, packed[i % R], absmax[i % R], M, K, 64) for i in range(R * 2)])
                except AssertionError as e:
                    continue
                c, cm = timeit(cold, R * 2)
                print(f"gemv regx {name} {M}x{K} ks={ks} iters={it} cold {c:6.2f} us (min {cm:6.2f}) = {nbytes / c / 1e3:6.0f} GB/s", flush=True)
        hipabi.set_variant("gemv", -1)
        cold = capture(lambda: [hipabi.gemv(x, packed[i % R], absmax[i % R], M, K, 64) for i in range(R * 2)])
        c, cm = timeit(cold, R * 2)
        print(f"gemv default {name} {M}x{K} cold {c:6.2f} us (min {cm:6.2f}) = {nbytes / c / 1e3:6.0f} GB/s", flush=True)

if what in ("quant", "all"):
    for dt, name, isz in ((torch.bfloat16, "bf16", 2), (torch.float16, "f16", 2), (torch.float32, "f32", 4)):
        RW = 16
        ws = [torch.randn(n, device=dev).to(dt) for _ in range(RW)]
        nbytes = n * isz + n // 2 + 4 * (n // 64)
        for wg in ([0] if "--grids" not in sys.argv else [0, 1, 2, 3, 4, 6, 8, 16, 1 << 20]):
            hipabi.set_variant("quantize", wg)
            cold = capture(lambda: [hipabi.quantize(ws[i % RW], 64) for i in range(RW * 2)])
            c, cm = timeit(cold, RW * 2)
            tag = "" if wg == 0 else f" wg/cu={wg}"
            print(f"quantize {name} {M}x{K} bs64{tag} cold {c:7.2f} us (min {cm:7.2f}) = {nbytes / c / 1e3:7.0f} GB/s", flush=True)
        hipabi.set_variant("quantize", 0)
