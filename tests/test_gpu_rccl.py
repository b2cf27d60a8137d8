"""The RCCL (`nccl` backend) branch of the tensor-parallel layers on real hardware.

A one-GPU box cannot host two RCCL ranks (RCCL refuses two ranks on one device), so this is a WORLD-SIZE-1 group: small, but
it is the real communicator - `init_process_group("nccl", device_id=...)`, `all_reduce` / `all_gather` on device tensors
through `torch_bnb_fp4.parallel`'s helpers (the same calls the N > 1 layers make), and a HIP-graph capture with the
collective INSIDE (`capture_error_mode="thread_local"`, what `bench.py`'s FP4_BENCH_C5_GRAPH=1 leg does).  Runs in a child
process so the process group never leaks into the other tests.  The 2-rank data path itself is covered with gloo staging in
test_gpu_parallel.py; the 8-GPU run is the driver's.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, sys
sys.path[:0] = [os.path.join(%(repo)r, "tests"), os.path.join(%(repo)r, "torch-bnb-fp4_amd"), %(repo)r]
import numpy as np
import torch
import torch.distributed as dist

os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(%(port)d), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
res = {"backend": dist.get_backend()}
try:
    from torch_bnb_fp4 import parallel as par
    from oracle import c_oracle

    M, K, BS = 4096, 4096, 64
    rng = np.random.default_rng(5)
    packed, am = c_oracle.quantize((rng.standard_normal(M * K) * 0.03).astype(np.float32), BS)
    P, A = torch.from_numpy(packed).to(dev).view(-1, 1), torch.from_numpy(am).to(dev)
    x = torch.from_numpy(rng.standard_normal(K).astype(np.float32)).to(dev).to(torch.bfloat16).view(1, K)
    # helpers on device tensors: with one rank the sum / the gather is the identity, through RCCL
    t = torch.arange(4096, device=dev, dtype=torch.float32) * 0.5
    res["all_reduce_identity"] = bool(torch.equal(par._all_reduce_sum(t.clone(), None), t))
    res["all_gather_identity"] = bool(torch.equal(par._all_gather_last(t.view(1, -1), 1, None), t.view(1, -1)))
    res["host_staging_off"] = not par._on_host_backend(None)
    # the K-split layer: world == 1 shards nothing; its result must be the plain GEMV's
    row = par.RowParallelFP4Linear(P, A, (M, K), BS)
    col = par.ColumnParallelFP4Linear(P, A, (M, K), BS)
    y_row, y_col = row(x), col(x)
    res["row_equals_col"] = bool(torch.equal(y_row, y_col))
    # HIP graph with the RCCL collective inside the capture, replayed on new inputs
    from torch_bnb_fp4._ext import ext

    static_x = x.clone()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        def step():
            part = ext.gemv_fp4_partial(static_x, row.quant_data.A.t(), row.quant_data.absmax, BS, [M, K])
            dist.all_reduce(part)
            return part.to(torch.bfloat16)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            y_static = step()
    torch.cuda.synchronize()
    ok = True
    for i in range(4):
        static_x.copy_(x * (1.0 + 0.25 * i))
        g.replay()
        torch.cuda.synchronize()
        ok = ok and torch.equal(y_static, row(static_x))
    res["graph_with_rccl_inside_equal"] = bool(ok)
    # the one-shot communicator degenerates cleanly to one rank: out = T(partial) + bias + residual
    comm = par.oneshot_comm(None)
    part = torch.randn(4096, device=dev)
    b, r = torch.randn(4096, device=dev).bfloat16(), torch.randn(4096, device=dev).bfloat16()
    got = comm.reduce(part, torch.bfloat16, b, r)
    res["oneshot_world1_equal"] = bool(torch.equal(got, (part.bfloat16() + b) + r))
    comm.check()
    res["oneshot_memory_kind"] = comm.memory_kind
finally:
    dist.destroy_process_group()
print("RESULT " + json.dumps(res))
"""


def test_rccl_world_size_one_collectives_and_graph_capture():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", CHILD % {"repo": REPO, "port": port}], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
    assert line, p.stdout[-2000:]
    res = json.loads(line[-1][7:])
    assert res["backend"] == "nccl"
    for key in ("all_reduce_identity", "all_gather_identity", "host_staging_off", "row_equals_col", "graph_with_rccl_inside_equal",
                "oneshot_world1_equal"):
        assert res[key], res
