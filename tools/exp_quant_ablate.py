#!/usr/bin/env python3
"""Times fp4_hip_quantize_blockwise of each library given on the command line (tools/exp_quant_ablate.sh builds the product library with
parts of the quantiser removed): bf16 / f16 / f32 input, 4096 x 4096, blocksize 64, with bench.py's launch structure - a HIP graph of R
launches rotating over R distinct weights (HBM-cold), HIP events - and as one launch over the stack of R (steady state)."""
import ctypes
import os
import statistics
import sys

import torch

dev = torch.device("cuda", 0)
M = K = 4096
n, R, BS = M * K, 32, 64
DT = {torch.float16: 0, torch.float32: 1, torch.bfloat16: 2}


def capture(fn):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    torch.cuda.synchronize()
    return g.replay


def timeit(replay, launches, reps=9, warm=3):
    ts = []
    for i in range(reps + warm):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); replay(); b.record(); b.synchronize()
        if i >= warm:
            ts.append(a.elapsed_time(b) * 1e3 / launches)
    return statistics.median(ts), min(ts)


def main():
    libs = sorted(sys.argv[1:], key=lambda p: (0 if "base" in p else 1, p))
    gen = torch.Generator(device=dev).manual_seed(0)
    for dt, name, isz in ((torch.bfloat16, "bf16", 2), (torch.float16, "f16", 2), (torch.float32, "f32", 4)):
        big = (torch.randn(R * n, device=dev, generator=gen) * 0.02).to(dt)
        ws = [big[i * n:(i + 1) * n] for i in range(R)]
        qp = torch.empty(R * n // 2, dtype=torch.uint8, device=dev)
        qa = torch.empty(R * n // BS, dtype=torch.float32, device=dev)
        nbytes = n * isz + n // 2 + 4 * (n // BS)
        for path in libs:
            lib = ctypes.CDLL(os.path.abspath(path))
            vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
            lib.fp4_hip_quantize_blockwise.argtypes = [vp, i32, vp, vp, i64, i32, vp]
            lib.fp4_hip_set_variant.argtypes = [ctypes.c_char_p, i32]

            def q(w, count, off=0):
                s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
                rc = lib.fp4_hip_quantize_blockwise(w.data_ptr(), DT[dt], qp.data_ptr() + off * (n // 2), qa.data_ptr() + off * 4 * (n // BS), count, BS, s)
                assert rc == 0, rc

            tag = os.path.basename(path).replace("libfp4_quant_abl_", "").replace(".so", "")
            for kern, variant in (("tiles", 0), ("persistent", 4)):  # the one-shot tiles kernel (default) / the persistent one, 4 workgroups per CU
                lib.fp4_hip_set_variant(b"quantize", variant)
                cold = capture(lambda: [q(ws[i], n, i) for i in range(R)])
                c, cm = timeit(cold, R)
                stack = capture(lambda: [q(big, R * n) for _ in range(4)])
                timeit(stack, 4, reps=6, warm=0)
                s_us, s_min = timeit(stack, 4)
                print(f"quantize {name} 4096x4096 bs64 {tag:16s} {kern:10s} per launch {c:6.2f} us (min {cm:6.2f}) = {nbytes / c / 1e3:6.0f} GB/s   "
                      f"stack of {R}: {s_us / R:6.2f} us per matrix = {R * nbytes / s_us / 1e3:6.0f} GB/s", flush=True)
            lib.fp4_hip_set_variant(b"quantize", 0)
        del big, ws, qp, qa


if __name__ == "__main__":
    main()
