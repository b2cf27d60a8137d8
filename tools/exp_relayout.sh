#!/bin/bash
# Round-3 bounded experiment: does a load-time re-layout of the packed nibbles (and the full-rate v_bitop3_b32 sign merge) speed the
# batch-1 GEMV up at the long / tall decode shapes?  Three builds of csrc/gemv_fp4.hip (baseline, -DFP4_EXP_BITOP3, -DFP4_EXP_RELAID)
# linked with tools/exp_gemv.hip (cross-compiled beforehand into build_tmp/exp/, see profiles/r03_gemv_relayout_attempt.txt), timed
# HBM-cold and cache-hot through the C ABI, then SQ counters of each build at 4096 x 14336.   Usage: tools/exp_relayout.sh OUTDIR
#
# Build first (works without a GPU; the binaries travel to the GPU box with the snapshot):  tools/exp_relayout.sh --build
set -e
cd "$(dirname "$0")/.."
if [ "$1" = "--build" ]; then
    python3 torch-bnb-fp4_amd/build.py --no-ext > /dev/null   # the other translation units' objects (build_tmp/obj) are linked as they are
    mkdir -p build_tmp/exp
    objs=$(ls build_tmp/obj/*.o | grep -v gemv_fp4)
    for v in base: bitop3:-DFP4_EXP_BITOP3 relaid:-DFP4_EXP_RELAID; do
        name=${v%%:*}; flag=${v#*:}
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=16 $flag \
            -Iinclude -Itorch-bnb-fp4_amd/csrc -c torch-bnb-fp4_amd/csrc/gemv_fp4.hip -o build_tmp/exp/gemv_$name.o
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $flag -Iinclude -Itorch-bnb-fp4_amd/csrc \
            -c tools/exp_gemv.hip -o build_tmp/exp/tool_$name.o
        /opt/rocm/bin/hipcc --offload-arch=gfx950 build_tmp/exp/tool_$name.o build_tmp/exp/gemv_$name.o $objs -o build_tmp/exp/exp_gemv_$name
    done
    exit 0
fi
out=${1:-gpurun_out/r3_relayout}
mkdir -p "$out"
for shape in "4096 4096" "28672 4096" "4096 14336"; do
    for b in base bitop3 relaid; do
        echo "### $b  $shape" | tee -a "$out/timing.txt"
        ./build_tmp/exp/exp_gemv_$b $shape quick 2>&1 | grep -E "^M=|floor L2|gemv default" | tee -a "$out/timing.txt"
    done
done
# a second pass in the opposite build order (clock / thermal drift shows up as a difference between the two passes)
for shape in "4096 14336" "28672 4096" "4096 4096"; do
    for b in relaid bitop3 base; do
        echo "### $b  $shape (second pass)" | tee -a "$out/timing.txt"
        ./build_tmp/exp/exp_gemv_$b $shape quick 2>&1 | grep -E "gemv default" | tee -a "$out/timing.txt"
    done
done
export TMPDIR=/tmp
for b in base bitop3 relaid; do
    rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU \
        -d "$out/pmc_$b" -o pmc --output-format csv -- ./build_tmp/exp/exp_gemv_$b 4096 14336 quick > "$out/pmc_$b.log" 2>&1 || echo "pmc $b failed" | tee -a "$out/timing.txt"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for b in ("base", "bitop3", "relaid"):
    files = glob.glob(f"{out}/pmc_{b}/**/*counter_collection.csv", recursive=True)
    if not files:
        print(b, "no counter file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for row in csv.DictReader(open(files[0])):
        k = row["Kernel_Name"]
        if "gemv16_regx" not in k: continue
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == "SQ_WAVES": n[k] += 1
    for k, c in agg.items():
        d = max(1, n[k])
        line = f"{b:7s} {k[:60]:60s} launches {d:4d} " + " ".join(f"{name}={v / d:.0f}" for name, v in sorted(c.items()))
        print(line)
        open(f"{out}/sq_counters.txt", "a").write(line + "\n")
PY
