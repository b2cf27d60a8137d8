#!/usr/bin/env python3
"""Turns two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, as the TCC slots require) into
per-launch HBM traffic for the FP4 kernels, with the gfx950 corrections of MI355X_MICROARCH.md (HBM section):
FETCH_SIZE (KB) counts 64 B per 128-B request on streaming reads -> x2; WRITE_SIZE (KB) is exact.

`_source_sha256` records the digest of every kernel source / header / build.py at collection time (tools/source_digest.py).

usage: tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
"""
import collections
import csv
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from source_digest import file_digests  # noqa: E402


def per_kernel(path, counter):
    """Mean counter value per (kernel, grid): one kernel name serves several problem sizes in bench.py (the 4096x4096 launches
    of the timed region and the one-launch stack of R weights), so launches are grouped by grid size as well; the most
    frequent grid of a kernel keeps the bare name (that is the timed region's), the others get a `grid=N` suffix."""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "fp4::" in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("(unsigned")[0].split("(void")[0].replace("void fp4::(anonymous namespace)::", "").strip()
            acc[name][int(r["Grid_Size"])].append(float(r["Counter_Value"]))
    out = {}
    for name, grids in acc.items():
        main = max(grids, key=lambda g: len(grids[g]))
        for g, v in grids.items():
            out[name if g == main else f"{name} grid={g}"] = (statistics.mean(v), len(v))
    return out


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"_method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `python bench.py --no-cpu "
                  "--steps 3 --warmup 1`; per-launch means; FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B), WRITE_SIZE as is; KB = 1024 B"}
# the sources these counters were taken from (the profile runs from the tree it sits in): bench.py marks the figure stale when they change
out["_source_sha256"] = file_digests()
for k in sorted(set(fetch) & set(write)):
    f_kb, nf = fetch[k]
    w_kb, nw = write[k]
    out[k] = {"launches_sampled": [nf, nw], "FETCH_SIZE_kb_raw": round(f_kb, 2), "WRITE_SIZE_kb_raw": round(w_kb, 2),
              "hbm_read_bytes": int(round(2 * f_kb * 1024)), "hbm_write_bytes": int(round(w_kb * 1024)),
              "traffic_bytes": int(round((2 * f_kb + w_kb) * 1024))}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
