#!/bin/bash
# Round 5: SQ counters of the quantiser's two kernels (rocprofv3 --pmc, counters in their own passes with --kernel-trace only), at 4096x4096 and
# 14336x4096 bf16: waves, vector / LDS instructions per wave, wave cycles and what they wait for.    tools/exp_quant_counters.sh OUTDIR  (GPU box)
out=${1:-gpurun_out/quant_counters}
mkdir -p "$out"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for shape in "4096 4096" "14336 4096"; do
    for variant in 4 1002; do
        tag="v${variant}_$(echo $shape | tr ' ' x)"
        for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
            d="$out/pmc_${tag}_$(echo $set | cut -d' ' -f1)"
            rocprofv3 --kernel-trace --pmc $set -d "$d" -o pmc --output-format csv -- python3 tools/profile_quant.py $variant $shape > "$d.log" 2>&1
            echo "### variant $variant (4 = persistent, 1002 = tiles with 2 loads per lane)  $shape  [$set]" | tee -a "$out/counters.txt"
            python3 tools/profile_quant.py --summarise "$(find "$d" -name '*counter_collection.csv' | head -1)" | tee -a "$out/counters.txt"
            rm -rf "$d"
        done
    done
done
