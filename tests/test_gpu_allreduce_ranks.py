"""The one-shot all-reduce (fp4_hip_allreduce_oneshot) with MORE than two writers, on one GPU.

With two ranks a + b is commutative, so a wrong summation order, a wrong slot stride (``rank * capacity``), a parity
buffer re-used too early or a ``world`` loop that stops short cannot show.  Here every check is ORDER-SENSITIVE: the
partials span six decades, so the f32 sum taken in rank order 0..G-1 differs in its bits from the sum taken in any other
order (asserted), and the kernel's output must equal that rank-order sum bit for bit on every rank.

Two set-ups, because a GPU box admits at most six processes on its card (the test runner is one of them):

* world 4, one process per rank, slot buffers mapped through IPC handles (the production transport), gloo for set-up;
* world 3 / 5 as streams of ONE process (every "rank" a stream with its own slot buffer, peers addressed directly), and
  world 8 as FOUR processes with two such ranks each (IPC between processes, direct pointers inside one): the kernel, its
  slot arithmetic and its epoch / parity protocol are the same in every set-up; only the mapping differs.  Polling is
  bounded by wall time, so a rank that is not co-scheduled ends in the NaN / status path, never in a hang.

The reference has no multi-GPU path (its only device handling is /root/reference/csrc/torch_fp4.cpp:47); spec: SURVEY 8e."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

SIZES = [4096, 1000, 16384, 1, 4096, 4096, 257]  # element counts of consecutive calls (epochs 1..7; both parities, ragged tails)
CAPACITY = 16384


def partial_of(rank: int, call: int, n: int) -> np.ndarray:
    """Rank ``rank``'s f32 partial in call ``call``: magnitudes over six decades, both signs -> the f32 sum is order-sensitive."""
    rng = np.random.default_rng(7919 * call + rank)
    return (rng.standard_normal(n) * 10.0 ** rng.uniform(-3, 3, n)).astype(np.float32)


def ordered_sum(parts) -> np.ndarray:
    """f32 sum in list order, one rounded add per rank - what the kernel does (``sum = 0; sum += slot[j]``)."""
    acc = np.zeros_like(parts[0], dtype=np.float32)
    for p in parts:
        acc = (acc + p).astype(np.float32)
    return acc


def expected(world: int, call: int, n: int, dtype: torch.dtype, bias, residual) -> torch.Tensor:
    s = torch.from_numpy(ordered_sum([partial_of(r, call, n) for r in range(world)]))
    if dtype == torch.float32:
        t = s
        if bias is not None:
            t = t + bias
        if residual is not None:
            t = t + residual
        return t
    t = s.to(dtype)
    if bias is not None:
        t = (t.float() + bias.float()).to(dtype)
    if residual is not None:
        t = (t.float() + residual.float()).to(dtype)
    return t


def extras(call: int, n: int, dtype: torch.dtype):
    """Bias / residual of a call (the same on every rank): none, bias, residual, both - cycling with the call number."""
    g = torch.Generator().manual_seed(31 * call + 5)
    bias = torch.randn(n, generator=g).to(dtype) if call % 4 in (1, 3) else None
    residual = torch.randn(n, generator=g).to(dtype) if call % 4 in (2, 3) else None
    return bias, residual


def same_bits(a: torch.Tensor, b: torch.Tensor) -> bool:
    a, b = a.detach().cpu().contiguous(), b.detach().cpu().contiguous()
    iv = torch.int32 if a.dtype == torch.float32 else torch.int16
    return a.dtype == b.dtype and bool(torch.equal(a.view(iv), b.view(iv)))


def finite_part_ok(y: torch.Tensor, want: torch.Tensor) -> bool:
    """For a mismatching output: is every element that is NOT NaN bit-equal to the expected one?  (A bounded time-out writes NaN, never
    a wrong finite value; anything else in a mismatch is a kernel bug, whatever the status word says.  `want` never holds a NaN.)"""
    y, want = y.detach().cpu().contiguous(), want.detach().cpu().contiguous()
    if y.dtype != want.dtype or y.shape != want.shape:
        return False
    iv = torch.int32 if y.dtype == torch.float32 else torch.int16
    keep = ~torch.isnan(y.float())
    return bool(torch.equal(y.view(iv)[keep], want.view(iv)[keep]))


def test_the_check_is_order_sensitive():
    """Precondition of everything below (pure numpy): rank order and reversed order give different f32 bits."""
    for world in (3, 4, 5, 8):
        parts = [partial_of(r, 1, 4096) for r in range(world)]
        fwd, rev = ordered_sum(parts), ordered_sum(parts[::-1])
        rot = ordered_sum(parts[1:] + parts[:1])
        assert (fwd.view(np.uint32) != rev.view(np.uint32)).mean() > 0.2, world
        assert (fwd.view(np.uint32) != rot.view(np.uint32)).mean() > 0.2, world


# ---- world 4, one process per rank, IPC-mapped slot buffers ------------------------------------------------------------
def _ipc_worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = {"rank": rank}
    try:
        from oracle import c_oracle
        from torch_bnb_fp4 import comm as comm_mod, parallel as par
        from torch_bnb_fp4._ext import ext

        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        comm = comm_mod.OneShotAllReduce(None, capacity=CAPACITY, timeout_us=5_000_000)
        bad = []
        digests = []
        call = 0
        for dtype in (torch.float32, torch.bfloat16, torch.float16):
            for n in SIZES:
                call += 1
                bias, residual = extras(call, n, dtype)
                y = comm.reduce(torch.from_numpy(partial_of(rank, call, n)).to(dev), dtype,
                                None if bias is None else bias.to(dev), None if residual is None else residual.to(dev))
                if not same_bits(y, expected(world, call, n, dtype, bias, residual)):
                    bad.append((str(dtype), call, n))
                digests.append(y.cpu().view(torch.uint8).numpy().tobytes())
        comm.check()
        res["bad"] = bad
        res["calls"] = call
        res["status"] = comm.status()
        # every rank's outputs are bit-identical to every other rank's
        everyone = [None] * world
        dist.all_gather_object(everyone, digests)
        res["ranks_agree"] = all(e == everyone[0] for e in everyone)

        # layer level: K-split FP4 weight, f32 activations so that nothing hides the order (the epilogue does not round);
        # the kernel's sum must equal the host-side f32 sum of the per-rank gemv_fp4_partial outputs taken in rank order
        M, K, BS = 4096, 4096, 64
        rng = np.random.default_rng(4242)
        w = (rng.standard_normal(M * K) * 0.03).astype(np.float32)
        packed, am = c_oracle.quantize(w, BS)
        P, A = torch.from_numpy(packed).to(dev).view(-1, 1), torch.from_numpy(am).to(dev)
        layer_bad = []
        for dtype in (torch.float32, torch.bfloat16):
            x = torch.from_numpy(rng.standard_normal(K).astype(np.float32) * 10.0 ** rng.uniform(-2, 2, K).astype(np.float32)).to(dtype).to(dev)
            one = par.RowParallelFP4Linear(P, A, (M, K), BS, allreduce="oneshot")
            y = one(x.view(1, K))
            ks = K // world
            p, a, local = par.shard_cols(P, A, (M, K), BS, rank, world)
            part = ext.gemv_fp4_partial(x[rank * ks:(rank + 1) * ks].view(1, ks).contiguous(), p.t(), a, BS, list(local)).cpu()
            parts = [torch.empty_like(part) for _ in range(world)]
            dist.all_gather(parts, part)
            want = torch.from_numpy(ordered_sum([t.numpy().reshape(-1) for t in parts])).to(dtype)
            if not same_bits(y.view(-1), want):
                layer_bad.append(str(dtype))
            rev = torch.from_numpy(ordered_sum([t.numpy().reshape(-1) for t in parts[::-1]]))
            res[f"layer_order_matters_{dtype}"] = bool((rev.view(torch.int32) != want.float().view(torch.int32)).any()) if dtype == torch.float32 else True
        res["layer_bad"] = layer_bad
        par.oneshot_comm(None).check()
        dist.barrier()

        # bounded polling with several waiters: the LAST rank stays away from one call of a fresh communicator; every other
        # rank gives up after 0.2 s, names the missing rank and returns NaN - nobody hangs
        lonely = comm_mod.OneShotAllReduce(None, capacity=1024, timeout_us=200_000)
        if rank != world - 1:
            y = lonely.reduce(torch.ones(1024, device=dev), torch.bfloat16)
            try:
                lonely.check()
                res["timeout"] = "no error raised"
            except RuntimeError as exc:
                res["timeout"] = str(exc)
            res["timeout_nan"] = bool(torch.isnan(y.float()).all().item())
        dist.barrier()
        # ... and the group's view of the same situation: the rank that stayed away has no time-out of its own (a local check()
        # would pass there while its peers raise), check_collective() raises on EVERY rank and names the ranks that gave up
        lonely2 = comm_mod.OneShotAllReduce(None, capacity=1024, timeout_us=200_000)
        if rank != world - 1:
            lonely2.reduce(torch.ones(1024, device=dev), torch.bfloat16)
        else:
            lonely2.check()  # the local view of the absent rank: nothing to report
        try:
            lonely2.check_collective()
            res["collective"] = "no error raised"
        except RuntimeError as exc:
            res["collective"] = str(exc)
        lonely2.check_collective()  # cleared on the ranks that had timed out: the next collective check passes everywhere
        dist.barrier()
        lonely2.close()
        lonely.close()
        comm.close()
        q.put(res)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_world_4_one_process_per_rank_ipc_slots():
    world = 4
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ipc_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(r["rank"] for r in results) == list(range(world))
    for r in results:
        assert r["bad"] == [] and r["layer_bad"] == [], r
        assert r["calls"] == 3 * len(SIZES) and r["status"][0] == r["calls"] and r["status"][1:] == (0, 0, 0), r["status"]
        assert r["ranks_agree"]
        assert r["layer_order_matters_torch.float32"]
        if r["rank"] != world - 1:
            assert f"timed out waiting for rank {world - 1}" in r["timeout"] and r["timeout_nan"], r
        # the collective check raised on every rank - the absent one included - naming each rank that gave up
        assert all(f"rank {k} (waiting for rank {world - 1}" in r["collective"] for k in range(world - 1)), r["collective"]
        assert f"this is rank {r['rank']}" in r["collective"], r["collective"]


# ---- world 3 / 5 as streams of ONE process, world 8 as 4 processes x 2 streams ------------------------------------------
def _grid_worker(proc, nproc, per, port, q):
    """``per`` ranks of this process (global ranks proc*per .. proc*per+per-1), each with a stream and a slot buffer of its own;
    buffers of other processes are mapped through IPC handles, those of this process addressed directly."""
    from torch_bnb_fp4._ext import ext
    from torch_bnb_fp4.dtypes import ScalarType

    world = nproc * per
    dist = None
    if nproc > 1:
        import torch.distributed as dist

        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        dist.init_process_group("gloo", rank=proc, world_size=nproc)
    try:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        own = [ext.comm_alloc(world, CAPACITY, 0) for _ in range(per)]  # (address, 64-byte handle, memory kind)
        opened = []
        if nproc > 1:
            everyone = [None] * nproc
            dist.all_gather_object(everyone, (proc, [bytes(h) for _, h, _ in own]))
            peers = []
            for pr, handles in sorted(everyone):
                for i, h in enumerate(handles):
                    if pr == proc:
                        peers.append(own[i][0])
                    else:
                        opened.append(ext.comm_open(h, 0))
                        peers.append(opened[-1])
            dist.barrier()
        else:
            peers = [o_[0] for o_ in own]
        mine = [proc * per + i for i in range(per)]
        streams = {r: torch.cuda.Stream() for r in mine}
        bad, call = [], 0
        finite_wrong = 0  # mismatches (over ALL of them, not the first few kept in `bad`) holding a wrong value that is not NaN
        for dtype in (torch.float32, torch.bfloat16):
            for n in SIZES:
                call += 1
                bias, residual = extras(call, n, dtype)
                bias_d = None if bias is None else bias.to(dev)
                res_d = None if residual is None else residual.to(dev)
                parts = {r: torch.from_numpy(partial_of(r, call, n)).to(dev) for r in mine}
                torch.cuda.synchronize()
                ys = []
                # launch in an order that is NOT the rank order (the sum must still be taken in rank order)
                for r in sorted(mine, key=lambda r: (r * 5 + call) % world):
                    with torch.cuda.stream(streams[r]):
                        ys.append((r, ext.allreduce_oneshot(parts[r], peers, r, CAPACITY, ScalarType.from_torch_dtype(dtype).value,
                                                            bias_d, res_d, 3_000_000)))
                torch.cuda.synchronize()
                want = expected(world, call, n, dtype, bias, residual)
                for r, y in ys:
                    if not same_bits(y, want):
                        bad.append((str(dtype), call, n, r, int(torch.isnan(y.float()).sum().item())))
                        finite_wrong += not finite_part_ok(y, want)
        # back-to-back calls WITHOUT a host synchronisation in between: a rank may run one call ahead of a peer that still
        # reads (the double buffering by epoch parity is what makes that safe)
        n = 4096
        first = call + 1
        ys = {}
        staged = {(i, r): torch.from_numpy(partial_of(r, first + i, n)).to(dev) for i in range(12) for r in mine}
        torch.cuda.synchronize()
        for i in range(12):
            for r in mine:
                with torch.cuda.stream(streams[r]):
                    ys[(i, r)] = ext.allreduce_oneshot(staged[(i, r)], peers, r, CAPACITY, ScalarType.float32.value, None, None, 3_000_000)
        torch.cuda.synchronize()
        for i in range(12):
            want = expected(world, first + i, n, torch.float32, None, None)
            for r in mine:
                if not same_bits(ys[(i, r)], want):
                    bad.append(("pipelined", first + i, n, r, int(torch.isnan(ys[(i, r)]).sum().item())))
                    finite_wrong += not finite_part_ok(ys[(i, r)], want)
        calls = call + 12
        layer_bad = []
        if world == 8:
            # C5's o-projection at its real sharding: the eight 4096x512 K-split partials of one 4096x4096 weight, reduced by
            # the kernel, against the host-side rank-order f32 sum of the same eight fp4_hip_gemv_partial outputs
            from oracle import c_oracle
            from torch_bnb_fp4 import parallel as par

            M, K, BS = 4096, 4096, 64
            rng = np.random.default_rng(808)
            packed, am = c_oracle.quantize((rng.standard_normal(M * K) * 0.03).astype(np.float32), BS)
            P, A = torch.from_numpy(packed).to(dev).view(-1, 1), torch.from_numpy(am).to(dev)
            x = torch.from_numpy(rng.standard_normal(K).astype(np.float32)).to(torch.bfloat16).to(dev)
            res = torch.from_numpy(rng.standard_normal(M).astype(np.float32)).to(torch.bfloat16).to(dev)
            ks = K // world
            parts = []
            for r in range(world):
                p, a, local = par.shard_cols(P, A, (M, K), BS, r, world)
                parts.append(ext.gemv_fp4_partial(x[r * ks:(r + 1) * ks].view(1, ks).contiguous(), p.t(), a, BS, list(local)).view(-1))
            torch.cuda.synchronize()
            for dtype, residual in ((torch.float32, None), (torch.bfloat16, res)):
                ys = []
                for r in mine:
                    with torch.cuda.stream(streams[r]):
                        ys.append(ext.allreduce_oneshot(parts[r], peers, r, CAPACITY, ScalarType.from_torch_dtype(dtype).value, None, residual, 3_000_000))
                torch.cuda.synchronize()
                calls += 1
                acc = torch.from_numpy(ordered_sum([t.cpu().numpy() for t in parts]))
                want = acc if dtype == torch.float32 else (acc.to(dtype).float() + res.float().cpu()).to(dtype)
                for y in ys:
                    if not same_bits(y, want):
                        layer_bad.append(str(dtype))
        status = [tuple(ext.comm_status(o_[0])) for o_ in own]
        if dist is not None:
            dist.barrier()  # peers may still be reading our slots
        for ptr in opened:
            ext.comm_close(ptr)
        for o_ in own:
            ext.comm_free(o_[0])
        q.put({"proc": proc, "bad": bad[:8], "n_bad": len(bad), "finite_wrong": finite_wrong, "layer_bad": layer_bad, "calls": calls,
               "status": status})
        if dist is not None:
            dist.barrier()
    finally:
        if dist is not None:
            dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("nproc,per", [(1, 3), (1, 5), (4, 2)])
def test_more_than_two_writers_streams_and_processes(nproc, per):
    """World 3 and 5 as streams of one process; world 8 as four processes with two ranks each (five processes on the card
    with the test runner: a GPU box admits six).  Eight streams in ONE process were measured not to be co-scheduled even
    with GPU_MAX_HW_QUEUES=16 (the eighth kernel waits behind one of the seven spinning ones; every rank then takes the
    bounded time-out, NaN + status word, as designed) - hence the split."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    saved = os.environ.get("GPU_MAX_HW_QUEUES")
    os.environ["GPU_MAX_HW_QUEUES"] = "8"  # read by the children's HIP runtime at start-up: a hardware queue per stream
    try:
        procs = [ctx.Process(target=_grid_worker, args=(i, nproc, per, port, q)) for i in range(nproc)]
        for p in procs:
            p.start()
    finally:
        if saved is None:
            os.environ.pop("GPU_MAX_HW_QUEUES", None)
        else:
            os.environ["GPU_MAX_HW_QUEUES"] = saved
    results = [q.get(timeout=600) for _ in range(nproc)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(r["proc"] for r in results) == list(range(nproc))
    if nproc == 1:
        # Several ranks as streams of ONE process need as many co-resident kernels as ranks; where the runtime does not grant that
        # (fewer hardware queues than streams), a rank waits behind a spinning one and the bounded time-out fires (NaN + status
        # word, by design).  That is an environment limit, not a kernel result: skip - but only if a rank's status word names a
        # time-out AND, over ALL mismatching outputs, every element that is not NaN carries the expected bits; a single wrong finite
        # value - next to NaNs or not - still fails.  (The multi-process set-ups above and below do not depend on it.)
        r = results[0]
        timed_out = any(st[2] != 0 and st[3] > 0 for st in r["status"])
        if timed_out and r["n_bad"] > 0 and r["finite_wrong"] == 0 and r["layer_bad"] == []:
            pytest.skip(f"{per} kernels of one process were not co-scheduled on this box (bounded time-out taken): {r['status'][0]}")
    for r in results:
        assert r["n_bad"] == 0 and r["layer_bad"] == [], r
        for st in r["status"]:  # every rank's header: all calls completed, no busy workgroup, no time-out
            assert st == (r["calls"], 0, 0, 0), r["status"]
