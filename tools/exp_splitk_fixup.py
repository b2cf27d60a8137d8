#!/usr/bin/env python3
"""(Needs profiles/r04_splitk_fixup_attempt.patch applied to csrc/ - the attempt was measured slower and is not in the library; results and
conclusion: profiles/r04_splitk_single_launch_fixup_attempt.txt.)

Round-4 experiment: 9..64 activation rows on short weights - the x-stationary split-K kernel as ONE launch with a last-arriver
fix-up (fp4_hip_set_variant("gemm_splitk", 1)) against the default dispatch of fp4_hip_gemm_small_ws, HBM-cold, HIP-graph replay.
Every fix-up output is first compared with the default dispatch's output for the same operands (parity-tested kernels; both round a
float32 sum of the same products once, so they may differ by one bf16 ulp where the summation order matters), three launches in a row on
one workspace (the counters must come back to zero by themselves).  The runs recorded in profiles/ checked against the float64 product
through the test-side oracle; tools/ must not import it, so the committed script does not."""
import ctypes
import os
import statistics
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "torch-bnb-fp4_amd"), os.path.join(REPO, "tests")]
import torch  # noqa: E402

import hipabi  # noqa: E402

dev = torch.device("cuda", 0)
BS = 64


def capture(fn):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    torch.cuda.synchronize()
    return g.replay


def timeit(replay, launches, reps=9):
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); replay(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / launches)
    return statistics.median(ts)


def ws_call(x, P, A, M, K, out, ws):
    l = hipabi.lib()
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
    l.fp4_hip_gemm_small_ws.argtypes = [vp, vp, vp, vp, vp, vp, i64, i64, i64, i32, i32, i32, vp, i64, vp]
    B = x.shape[0]
    rc = l.fp4_hip_gemm_small_ws(hipabi._ptr(x), hipabi._ptr(P), hipabi._ptr(A), None, None, hipabi._ptr(out), B, M, K, BS, hipabi.DT[x.dtype], 0,
                                 hipabi._ptr(ws), 0 if ws is None else ws.numel(), hipabi._stream())
    assert rc == 0, hipabi.last_error()


def ws_bytes(B, M, K):
    l = hipabi.lib()
    l.fp4_hip_gemm_small_ws_bytes.restype = ctypes.c_int64
    l.fp4_hip_gemm_small_ws_bytes.argtypes = [ctypes.c_int64] * 3 + [ctypes.c_int] * 2
    return l.fp4_hip_gemm_small_ws_bytes(B, M, K, BS, hipabi.BF16)


shapes = [(4096, 4096), (6144, 4096), (1024, 4096), (4096, 14336), (4096, 11008), (5120, 13824), (4096, 8192)]
if len(sys.argv) > 2:
    shapes = [(int(sys.argv[1]), int(sys.argv[2]))]
for M, K in shapes:
    n = M * K
    R = max(6, min(48, int(1.0e9 / (n * 0.5625))))
    gen = torch.Generator(device=dev).manual_seed(0)
    packed = [torch.randint(0, 256, (n // 2,), dtype=torch.uint8, device=dev, generator=gen) for _ in range(R)]
    absmax = [torch.rand(n // 64, device=dev, generator=gen) * 0.1 + 0.01 for _ in range(R)]
    for B in (9, 12, 16, 24, 32, 48, 64):
        x = torch.randn(B, K, device=dev).to(torch.bfloat16)
        out = torch.empty(B, M, dtype=torch.bfloat16, device=dev)
        # default dispatch (with the workspace it asks for, if any)
        hipabi.set_variant("gemm_splitk", 0)
        nb0 = ws_bytes(B, M, K)
        ws0 = torch.empty(max(nb0, 16), dtype=torch.uint8, device=dev) if nb0 else None
        ws_call(x, packed[0], absmax[0], M, K, out, ws0)
        ref = out.clone()
        t_def = timeit(capture(lambda: [ws_call(x, packed[i], absmax[i], M, K, out, ws0) for i in range(R)]), R)
        # single launch with the fix-up (FIXUP_MODE: 1 = no fences, 3 = release fence, 5 = acquire fence, 7 = both)
        hipabi.set_variant("gemm_splitk", int(os.environ.get("FIXUP_MODE", "1")))
        nb1 = ws_bytes(B, M, K)
        line = f"{M}x{K} bf16 rows {B:3d}: default {t_def:6.2f} us (ws {nb0 >> 10} KiB)"
        if nb1:
            ws1 = torch.zeros(nb1, dtype=torch.uint8, device=dev)
            outs = []
            for _ in range(3):
                o = torch.empty(B, M, dtype=torch.bfloat16, device=dev)
                ws_call(x, packed[0], absmax[0], M, K, o, ws1)
                outs.append(o)
            torch.cuda.synchronize()
            bad = int(((outs[0].float() - ref.float()).abs() > 2.0 ** -7 * ref.float().abs() + 1e-3).sum().item())
            same = torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
            counters_zero = bool((ws1[: ((M + 15) // 16 * 4 + 255) // 256 * 256] == 0).all())
            t_fix = timeit(capture(lambda: [ws_call(x, packed[i], absmax[i], M, K, out, ws1) for i in range(R)]), R)
            line += f"   fix-up {t_fix:6.2f} us (ws {nb1 >> 10} KiB)  bad {bad}  repeatable {same}  counters zero {counters_zero}"
        else:
            line += "   fix-up: n/a"
        hipabi.set_variant("gemm_splitk", 0)
        print(line, flush=True)
    del packed, absmax
    torch.cuda.empty_cache()
