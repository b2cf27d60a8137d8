#!/bin/bash
# Round 4: where does the batch-1 GEMV's per-launch time go at the HEADLINE shape?  The same-run probe of bench.py reads the GEMV's
# 9.45 MB bare in 2.9 us per launch (tools/stream_probe.hip geometry) against 3.8 us for the GEMV - round 1's "floor" kernel (3.6 us)
# loaded the scales lane by lane and strided its addresses over the grid, and under-estimated the headroom.  Five builds of
# csrc/gemv_fp4.hip - as shipped, and with the x loads / the scale loads / the decode + dot arithmetic / all three removed (FP4_ABL_*,
# results meaningless by construction) - linked with tools/exp_gemv.hip, timed HBM-cold and cache-hot next to both floors.
#   tools/exp_gemv_ablate.sh --build      (here, no GPU needed; binaries travel with the snapshot in build_tmp/exp/)
#   tools/exp_gemv_ablate.sh OUTDIR       (on the GPU box)
set -e
cd "$(dirname "$0")/.."
if [ "$1" = "--build" ]; then
    python3 torch-bnb-fp4_amd/build.py --no-ext > /dev/null
    mkdir -p build_tmp/exp
    objs=$(ls build_tmp/obj/*.o | grep -v gemv_fp4)
    for v in base: nox:-DFP4_ABL_NOX noabsmax:-DFP4_ABL_NOABSMAX nocompute:-DFP4_ABL_NOCOMPUTE "all:-DFP4_ABL_NOX -DFP4_ABL_NOABSMAX -DFP4_ABL_NOCOMPUTE"; do
        name=${v%%:*}; flag=${v#*:}
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=16 $flag \
            -Iinclude -Itorch-bnb-fp4_amd/csrc -c torch-bnb-fp4_amd/csrc/gemv_fp4.hip -o build_tmp/exp/gemv_abl_$name.o
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Itorch-bnb-fp4_amd/csrc \
            -c tools/exp_gemv.hip -o build_tmp/exp/tool_abl.o
        /opt/rocm/bin/hipcc --offload-arch=gfx950 build_tmp/exp/tool_abl.o build_tmp/exp/gemv_abl_$name.o $objs -o build_tmp/exp/exp_gemv_abl_$name
    done
    exit 0
fi
out=${1:-gpurun_out/gemv_ablate}
mkdir -p "$out"
for shape in "4096 4096" "6144 4096" "4096 14336" "28672 4096"; do
    for b in base nox noabsmax nocompute all; do
        echo "### $b  $shape" | tee -a "$out/ablation.txt"
        ./build_tmp/exp/exp_gemv_abl_$b $shape quick 2>&1 | grep -E "^M=|empty kernel|floor|gemv default" | tee -a "$out/ablation.txt"
    done
done
