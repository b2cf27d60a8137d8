import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "torch-bnb-fp4_amd"), os.path.join(REPO, "tests")]
import numpy as np, torch
import hipabi
from gpu_util import NPDT, bits, dev, np_bits, to_dev
from oracle import fp4_oracle as o
g = dict(np.load(os.path.join(REPO, "tests/golden/fp4_golden.npz")))
for tag in "abcde":
    packed, am, n = g[f"kat4{tag}_packed"], g[f"kat4{tag}_absmax"], int(g[f"kat4{tag}_n"])
    for dt in (torch.float32, torch.float16, torch.bfloat16):
        out = hipabi.dequantize(to_dev(packed), to_dev(am), 64, n, dt)
        want = np_bits(o.dequantize(packed, am, 64, n, NPDT[dt]))
        got = bits(out)
        bad = np.nonzero(got != want)[0]
        print(tag, n, dt, "mismatches", bad.size, bad[:10], [hex(v) for v in got[bad[:6]]], [hex(v) for v in want[bad[:6]]])
