"""Tensor-parallel layers on real kernels: two ranks share cuda:0 and talk over gloo (the one-GPU box cannot host an
RCCL group), so everything but the transport is the production path: row shards + all-gather for the column-parallel
layer, re-packed column shards + fp4_hip_gemv_partial + f32 all-reduce for the row-parallel layer."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import c_oracle, fp4_oracle as o

pytestmark = pytest.mark.gpu
M, K, BS = 512, 2048, 64


def _case():
    rng = np.random.default_rng(99)
    w = (rng.standard_normal(M * K) * 0.03).astype(np.float32)
    packed, am = c_oracle.quantize(w, BS)
    return packed, am, rng.standard_normal(M).astype(np.float32) * 0.1, rng.standard_normal(K).astype(np.float32)


def _worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torch_bnb_fp4 import parallel as par

        dev = torch.device("cuda", 0)
        packed, am, bias, x = _case()
        P, A = torch.from_numpy(packed).to(dev).view(-1, 1), torch.from_numpy(am).to(dev)
        B = torch.from_numpy(bias).to(dev).to(torch.bfloat16)
        xt = torch.from_numpy(x).to(dev).to(torch.bfloat16)
        col = par.ColumnParallelFP4Linear(P, A, (M, K), BS, bias=B)
        row = par.RowParallelFP4Linear(P, A, (M, K), BS, bias=B)
        res = {"col1": col(xt.view(1, K)), "row1": row(xt.view(1, 1, K)), "col4": col(xt.repeat(4, 1)), "row4": row(xt.repeat(4, 1))}
        if rank == 0:
            q.put({k: v.float().cpu().numpy() for k, v in res.items()})
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_tensor_parallel_two_ranks_on_one_gpu():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    packed, am, bias, x = _case()
    xb = torch.from_numpy(x).to(torch.bfloat16).float().numpy().astype(np.float64)
    bb = torch.from_numpy(bias).to(torch.bfloat16).float().numpy().astype(np.float64)
    want = c_oracle.gemv_f64(xb, packed, am, M, K, BS) + bb
    tol = 2.0**-7 * np.abs(want) + 2e-3
    assert res["col1"].shape == (1, M) and res["row1"].shape == (1, 1, M) and res["col4"].shape == (4, M)
    for key in ("col1", "row1"):
        assert (np.abs(res[key].reshape(-1) - want) <= tol).all(), key
    for key in ("col4", "row4"):
        err = np.abs(res[key][2] - want)  # batch path: the weight itself is rounded to bf16 before the GEMM
        assert (err <= tol + 2e-2).all(), (key, err.max(), int(err.argmax()), res[key][2][err.argmax()], want[err.argmax()])
