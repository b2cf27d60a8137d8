"""world_size-2 checks of the tensor-parallel layer on CPU (gloo): sharding arithmetic against the oracle,
and the Column/Row-parallel modules end to end with the oracle-backed fake extension standing in for the GPU ops."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import fp4_oracle as o

M, K, BS = 128, 512, 64


def _case():
    rng = np.random.default_rng(42)
    w = (rng.standard_normal(M * K) * 0.05).astype(np.float32)
    packed, am = o.quantize_fp4(w, BS)
    bias = rng.standard_normal(M).astype(np.float32) * 0.1
    x = rng.standard_normal(K).astype(np.float32)
    return packed, am, bias, x


def test_shard_arithmetic_matches_full_dequant():
    import torch_bnb_fp4.parallel as par

    packed, am, _, x = _case()
    full = o.dequantize_f32(packed, am, BS, M * K).reshape(M, K)
    P, A = torch.from_numpy(packed).view(-1, 1), torch.from_numpy(am)
    for world in (2, 4, 8):
        rows, partial = [], np.zeros(M)
        for r in range(world):
            p, a, shp = par.shard_rows(P, A, (M, K), BS, r, world)
            rows.append(o.dequantize_f32(p.numpy().reshape(-1), a.numpy(), BS, shp[0] * shp[1]).reshape(shp))
            p, a, shp = par.shard_cols(P, A, (M, K), BS, r, world)
            wc = o.dequantize_f32(p.numpy().reshape(-1), a.numpy(), BS, shp[0] * shp[1]).reshape(shp)
            assert np.array_equal(wc, full[:, r * shp[1]:(r + 1) * shp[1]])
            partial += o.gemv_exact(x[r * shp[1]:(r + 1) * shp[1]], p.numpy().reshape(-1), a.numpy(), shp[0], shp[1], BS)
        assert np.array_equal(np.concatenate(rows), full)
        assert np.allclose(partial, o.gemv_exact(x, packed, am, M, K, BS), rtol=1e-12)
    with pytest.raises(ValueError):
        par.shard_cols(P, A, (M, K), BS, 0, 16)  # 32 columns per rank < blocksize
    with pytest.raises(ValueError):
        par.shard_rows(P, A, (M, K), BS, 0, 3)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys

        here = os.path.dirname(os.path.abspath(__file__))
        sys.path[:0] = [here]
        import torch_bnb_fp4 as pkg
        from fake_ext import FakeExt
        from torch_bnb_fp4 import functional as F_mod, fused as fused_mod, parallel as par, quant_data as qd_mod

        fake = FakeExt(pkg.ext)
        F_mod.ext = fake
        qd_mod.ext = fake
        par.ext = fake
        fused_mod.ext = fake
        packed, am, bias, x = _case()
        P, A, B = torch.from_numpy(packed).view(-1, 1), torch.from_numpy(am), torch.from_numpy(bias)
        xt = torch.from_numpy(x)
        col = par.ColumnParallelFP4Linear(P, A, (M, K), BS, bias=B)
        row = par.RowParallelFP4Linear(P, A, (M, K), BS, bias=B)
        res = {
            "col1": col(xt.view(1, K)), "row1": row(xt.view(1, K)), "row3d": row(xt.view(1, 1, K)),
            "colN": col(torch.stack([xt, 2 * xt])), "rowN": row(torch.stack([xt, 2 * xt])),
            "calls": list(fake.calls),
        }
        # the fused tensor-parallel forms: several M-split projections in one launch per rank, gate|up with silu(g)*u fused
        xb = xt.to(torch.bfloat16).view(1, K)
        col_ng = par.ColumnParallelFP4Linear(P, A, (M, K), BS, gather_output=False)
        P2 = torch.flip(P, dims=[0]).contiguous()
        col2_ng = par.ColumnParallelFP4Linear(P2, A, (M, K), BS, gather_output=False)
        qkv = par.FusedColumnParallelFP4([(P, A, (M, K)), (P2, A, (M, K)), (P, A, (M, K))], BS)
        ya, yb = col_ng(xb), col2_ng(xb)
        res["fused_cat_equal"] = bool(torch.equal(qkv(xb), torch.cat([ya, yb, ya], dim=-1))) and qkv.split_sizes == [M // world] * 3
        gu = par.FusedColumnParallelFP4([(P, A, (M, K)), (P2, A, (M, K))], BS, epilogue="silu_mul")
        want = torch.nn.functional.silu(ya) * yb
        got = gu(xb)
        res["fused_silu_shape"] = tuple(got.shape) == (1, M // world)
        res["fused_silu_close"] = float((got.float() - want.float()).abs().max()) <= 0.02 * float(want.float().abs().max()) + 1e-3
        res["row_residual"] = row(xt.view(1, K), residual=torch.ones(1, M))
        if rank == 0:
            out_q.put({k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in res.items()})
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_column_and_row_parallel_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    packed, am, bias, x = _case()
    want = o.gemv_exact(x, packed, am, M, K, BS) + bias
    for key, shape in (("col1", (1, M)), ("row1", (1, M)), ("row3d", (1, 1, M))):
        assert res[key].shape == shape and np.allclose(res[key].reshape(-1), want, rtol=1e-4, atol=1e-5), key
    for key in ("colN", "rowN"):
        assert res[key].shape == (2, M)
        assert np.allclose(res[key][0], want, rtol=1e-3, atol=1e-4) and np.allclose(res[key][1], 2 * want - bias, rtol=1e-3, atol=1e-4)
    assert "gemv_fp4_partial" in res["calls"] and "gemv_fp4_bias" in res["calls"]
    assert res["fused_cat_equal"] and res["fused_silu_shape"] and res["fused_silu_close"]
    assert np.allclose(res["row_residual"].reshape(-1), want + 1.0, rtol=1e-4, atol=1e-5)
