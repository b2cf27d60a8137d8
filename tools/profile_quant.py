#!/usr/bin/env python3
"""A few HBM-cold launches of the FP4 quantiser (bf16 -> packed + absmax) for rocprofv3 counter passes.
    python tools/profile_quant.py [variant [M K]]     variant: 0 = the library's choice, 4 = persistent kernel, 1001 / 1002 / 1004 = tiles kernel
With `--summarise <counter_collection.csv>` it prints the per-kernel mean of every counter in a rocprofv3 PMC csv instead."""
import os
import sys

if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
    import collections
    import csv
    import statistics

    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(sys.argv[2])):
        if "quantize" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0].replace("void fp4::(anonymous namespace)::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in sorted(acc.items()):
        print(k, {c: round(statistics.mean(v), 1) for c, v in sorted(cs.items())}, "launches", len(next(iter(cs.values()))))
    sys.exit(0)

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "torch-bnb-fp4_amd"), os.path.join(REPO, "tests")]
import torch  # noqa: E402

import hipabi  # noqa: E402

variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
M, K = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (4096, 4096)
dev = torch.device("cuda", 0)
n = M * K
ws = [torch.randn(n, device=dev).to(torch.bfloat16) for _ in range(max(4, min(24, int(8e8 / (2 * n)))))]
hipabi.set_variant("quantize", variant)
for rep in range(4):
    for w in ws:
        hipabi.quantize(w, 64)
torch.cuda.synchronize()
hipabi.set_variant("quantize", 0)
print("done")
