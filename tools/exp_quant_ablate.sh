#!/bin/bash
# Round 5: where does the quantiser's per-launch time go?  bench.py's `roofline_quantize` puts it at 0.77 of what this box needs to read
# 33.6 MB and write 9.4 MB back to back (steady state), the largest gap to the box's own streams among the kernels.  Builds of
# csrc/quantize_fp4.hip as shipped and with the ranking / the stores / the block-maximum pass / the loads removed (FP4_ABL_*, results
# meaningless by construction) - and with the round-1..4 ranking arithmetic (FP4_EXP_QUANT_OLD_ENCODE, results identical) - each linked into a complete library of its own and timed by tools/exp_quant_ablate.py.
#   tools/exp_quant_ablate.sh --build      (here, no GPU needed; the libraries travel with the snapshot in build_tmp/exp/)
#   tools/exp_quant_ablate.sh OUTDIR       (on the GPU box)
set -e
cd "$(dirname "$0")/.."
if [ "$1" = "--build" ]; then
    python3 torch-bnb-fp4_amd/build.py --no-ext > /dev/null
    mkdir -p build_tmp/exp
    objs=$(ls build_tmp/obj/*.o | grep -v quantize_fp4)
    for v in base: old_encode:-DFP4_EXP_QUANT_OLD_ENCODE norank:-DFP4_ABL_NORANK nostore:-DFP4_ABL_NOSTORE noabsmax:-DFP4_ABL_NOABSMAX noload:-DFP4_ABL_NOLOAD \
             "noload_nostore:-DFP4_ABL_NOLOAD -DFP4_ABL_NOSTORE" "norank_noabsmax:-DFP4_ABL_NORANK -DFP4_ABL_NOABSMAX"; do
        name=${v%%:*}; flag=${v#*:}
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -mllvm -amdgpu-kernarg-preload-count=16 \
            -Wno-dangling-else $flag -Iinclude -Itorch-bnb-fp4_amd/csrc -c torch-bnb-fp4_amd/csrc/quantize_fp4.hip -o build_tmp/exp/quant_abl_$name.o
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fvisibility=hidden build_tmp/exp/quant_abl_$name.o $objs -o build_tmp/exp/libfp4_quant_abl_$name.so
    done
    exit 0
fi
out=${1:-gpurun_out/quant_ablate}
mkdir -p "$out"
python3 tools/exp_quant_ablate.py build_tmp/exp/libfp4_quant_abl_*.so | tee "$out/ablation.txt"
