#!/bin/bash
# One command for the evidence the bench line cites (run on the GPU box):  tools/collect_profiles.sh OUTDIR
#   1. un-profiled `python3 bench.py`                                              -> OUTDIR/bench.json
#   2. rocprofv3 --kernel-trace --stats over the same command (--no-cpu, 5 steps)  -> OUTDIR/trace/*  + OUTDIR/profiled_run_line.json
#   3. tools/trace_summary.py: per-(kernel, grid) durations, the roofline rows (per-launch and stack-of-R) and the tracer's inflation
#   4. two PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs with --kernel-trace only, as the microarchitecture guide prescribes)
#      -> tools/pmc_traffic.py -> OUTDIR/traffic.json
# Copy bench.json, the stats csv, trace_summary.json, profiled_run_line.json and traffic.json into profiles/ as rNN_*.
set -e
out=${1:-gpurun_out/profiles_run}
mkdir -p "$out"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp FP4_BENCH_C4=${FP4_BENCH_C4:-1}
python3 bench.py > "$out/bench.json" 2> "$out/bench.err"
echo "bench: $(head -c 300 "$out/bench.json")"
FP4_BENCH_C4=0 FP4_BENCH_C3=0 FP4_BENCH_QUANT_STACK=0 rocprofv3 --kernel-trace --stats -d "$out/trace" -o trace --output-format csv -- python3 bench.py --no-cpu --steps 5 --warmup 2 \
    > "$out/profiled_run_line.json" 2> "$out/trace.err"
python3 tools/trace_summary.py "$(find "$out/trace" -name '*kernel_trace.csv' | head -1)" "$out/trace_summary.json" \
    --bench-json "$out/profiled_run_line.json" --unprofiled-json "$out/bench.json" > "$out/trace_summary.txt"
cp "$(find "$out/trace" -name '*kernel_stats.csv' | head -1)" "$out/kernel_stats.csv"
for c in FETCH_SIZE WRITE_SIZE; do
    FP4_BENCH_C4=0 FP4_BENCH_C3=0 FP4_BENCH_QUANT_STACK=0 rocprofv3 --kernel-trace --pmc $c -d "$out/pmc_$c" -o pmc --output-format csv -- python3 bench.py --no-cpu --steps 3 --warmup 1 \
        > "$out/pmc_$c.line" 2> "$out/pmc_$c.err"
done
python3 tools/pmc_traffic.py "$(find "$out/pmc_FETCH_SIZE" -name '*counter_collection.csv' | head -1)" \
    "$(find "$out/pmc_WRITE_SIZE" -name '*counter_collection.csv' | head -1)" "$out/traffic.json" > /dev/null
rm -rf "$out"/pmc_*/ "$out"/trace/*kernel_trace.csv 2>/dev/null || true   # the raw traces are tens of MB; the summaries are what is kept
tail -25 "$out/trace_summary.txt"
