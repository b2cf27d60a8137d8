#!/usr/bin/env python3
"""Per-kernel timing of a rocprofv3 kernel trace of `bench.py`, from the dispatches' start / end TIMESTAMPS.

rocprofv3's --stats table reports a kernel's average DURATION (end - start of each dispatch), per kernel NAME.  For the
short kernels of bench.py that is not directly the time a launch costs: what bench.py measures with HIP events is the
launch-to-launch interval, one kernel name serves several problem sizes (per-name averages mix them), and the trace
itself changes the timing (dispatches are serialised, ~1 us is added to each).  This tool reports duration, interval and
gap per kernel AND grid from the dispatch timestamps and rebuilds the timed step from them, against the bench line of
the SAME profiled run:   sum of durations per step  <=  that run's ms_per_step.

usage: tools/trace_summary.py <kernel_trace.csv> <out.json> [--bench-json bench_line.json]
"""
import collections
import csv
import json
import re
import statistics
import sys


def short(name: str) -> str:
    name = name.split("(unsigned")[0].split("(void")[0]
    name = name.replace("void fp4::(anonymous namespace)::", "").strip()
    return re.sub(r"\s+", " ", name)[:90]


def main():
    trace, out_path = sys.argv[1], sys.argv[2]
    bench = None
    if "--bench-json" in sys.argv:
        bench = json.loads(open(sys.argv[sys.argv.index("--bench-json") + 1]).read().strip().splitlines()[-1])
    rows = []
    for r in csv.DictReader(open(trace)):
        if "fp4::" not in r["Kernel_Name"]:
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), int(r["Grid_Size_X"]),
                     int(r["Workgroup_Size_X"]), int(r["VGPR_Count"]), int(r["LDS_Block_Size"])))
    rows.sort()
    groups = collections.defaultdict(list)
    for i, row in enumerate(rows):
        groups[(row[2], row[3], row[4])].append(i)
    out = {"_method": "rocprofv3 --kernel-trace --output-format csv over `python3 bench.py --no-cpu --steps 5 --warmup 2`; duration = "
                      "End_Timestamp - Start_Timestamp of a dispatch; interval = start-to-start distance to the NEXT dispatch when that "
                      "is the same kernel within 50 us (i.e. inside one back-to-back graph replay); overlap = duration - interval where "
                      "positive; all in us", "kernels": {}}
    for key, idx in sorted(groups.items(), key=lambda kv: -len(kv[1])):
        name, grid, wg = key
        dur = [(rows[i][1] - rows[i][0]) / 1e3 for i in idx]
        interval, gap = [], []
        for i in idx:
            if i + 1 < len(rows) and (rows[i + 1][2], rows[i + 1][3], rows[i + 1][4]) == key:
                d = (rows[i + 1][0] - rows[i][0]) / 1e3
                if d < 50.0:
                    interval.append(d)
                    gap.append((rows[i + 1][0] - rows[i][1]) / 1e3)
        if len(dur) < 8:
            continue
        rec = {"launches": len(dur), "grid_threads": grid, "workgroup": wg, "vgprs": rows[idx[0]][5], "lds_bytes": rows[idx[0]][6],
               "duration_us": {"mean": round(statistics.mean(dur), 3), "median": round(statistics.median(dur), 3),
                               "min": round(min(dur), 3), "max": round(max(dur), 3)}}
        if len(interval) >= 8:
            rec["interval_us"] = {"mean": round(statistics.mean(interval), 3), "median": round(statistics.median(interval), 3),
                                  "samples": len(interval)}
            rec["gap_next_start_minus_this_end_us"] = {"mean": round(statistics.mean(gap), 3), "median": round(statistics.median(gap), 3),
                                                       "share_overlapping": round(sum(g < 0 for g in gap) / len(gap), 3)}
        out["kernels"][f"{name} grid={grid} wg={wg}"] = rec
    if bench is not None:
        R, GR = bench["config"]["matrices_per_step"], bench["config"]["gemv_passes_per_step"]
        dq = next((v for k, v in out["kernels"].items() if k.startswith("dequant_tiles_kernel<2, 4, true>") and v["grid_threads"] == 524288), None)
        gv = next((v for k, v in out["kernels"].items() if k.startswith("gemv16_regx_kernel<2, 4, 1, 2") and "interval_us" in v), None)
        if dq and gv and "interval_us" in dq:
            step_from_intervals = (R * dq["interval_us"]["mean"] + R * GR * gv["interval_us"]["mean"]) / 1e3
            step_from_durations = (R * dq["duration_us"]["mean"] + R * GR * gv["duration_us"]["mean"]) / 1e3
            out["step_reconstruction"] = {
                "launches_per_step": {"dequant": R, "gemv": R * GR},
                "ms_per_step_from_intervals": round(step_from_intervals, 4),
                "ms_per_step_from_durations": round(step_from_durations, 4),
                "ms_per_step_bench_hip_events": bench["ms_per_step"],
                "bench_dequant_us": bench["dequant_us_per_matrix"], "bench_gemv_us": bench["gemv_us_per_layer"],
                "note": "compare only within this profiled run: the kernel trace serialises dispatches (median gap between one dispatch's end "
                        "and the next one's start: 0) and adds ~1 us to kernels this short, so the sum of durations stays below the run's own "
                        "ms_per_step, and both are above the un-profiled figures of bench.py",
            }
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out.get("step_reconstruction", {}), indent=1))
    for k, v in list(out["kernels"].items())[:6]:
        print(k, v["duration_us"], v.get("interval_us"), v.get("gap_next_start_minus_this_end_us"))


if __name__ == "__main__":
    main()
