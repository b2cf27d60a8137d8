#!/bin/bash
# N > 1 rehearsals on a ONE-GPU box (run on the GPU box):  tools/rehearse_ranks.sh OUTDIR
# The ranks share the device and gloo stands in for RCCL (FP4_BENCH_BACKEND=gloo); everything else is the production N > 1 path
# of bench.py: group evidence from real collectives, per-rank rates, K-split leg, strong-scaling leg, C5 leg with the one-shot all-reduce.
#   1. python bench.py --gpus 2              (self-launched workers)                     -> OUTDIR/bench_2rank_gloo.json
#   2. python bench.py --gpus 4 --matrices 32                                            -> OUTDIR/bench_4rank_gloo.json
#   3. the DRIVER's form: torchrun from outside, no HSA_* variable preset                -> OUTDIR/bench_2rank_torchrun_gloo.json
out=${1:-gpurun_out/ranks}
mkdir -p "$out"
cd "$(dirname "$0")/.."
export FP4_BENCH_BACKEND=gloo
python3 bench.py --gpus 2 > "$out/bench_2rank_gloo.json" 2> "$out/bench_2rank.err"; echo "2 ranks rc=$?"
python3 bench.py --gpus 4 --matrices 32 > "$out/bench_4rank_gloo.json" 2> "$out/bench_4rank.err"; echo "4 ranks rc=$?"
env -u HSA_ENABLE_IPC_MODE_LEGACY python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 2 > "$out/bench_2rank_torchrun_gloo.json" 2> "$out/bench_2rank_torchrun.err"; echo "torchrun form rc=$?"
