"""Tensor-parallel use of FP4 Linear weights across the GPUs of one node (one process per GPU,
``torch.distributed`` with the ``nccl`` backend = RCCL over xGMI).

The reference has no multi-GPU code at all; this module is the SURVEY section 8e design:

* quant blocks are flat over the row-major weight and ``K % blocksize == 0``, so a **row range** of W is a
  contiguous slice of both ``packed`` and ``absmax`` -> M-split ("column-parallel") shards need no re-packing and
  no collective for the GEMV itself (outputs are disjoint; an all-gather only if the consumer wants all of y);
* a **column range** is strided -> K-split ("row-parallel") shards are re-packed once at load time into their own
  contiguous ``[M, K/G/2]`` bytes + ``[M*K/G/bs]`` scales; each rank produces a full-length f32 partial
  (``gemv_fp4_partial``) and the partials are summed with one all-reduce (16 KiB at M = 4096: latency-bound),
  then rounded to the activation dtype once.  ``allreduce="dist"`` uses ``torch.distributed`` (RCCL);
  ``allreduce="oneshot"`` the peer-slot kernel of :mod:`torch_bnb_fp4.comm` (one launch, HIP-graph capturable).
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch
import torch.distributed as dist
from torch import nn

from ._ext import ext
from .dtypes import ScalarType
from .nn import QuantState, fp4_code
from .quant_data import QuantData


def shard_rows(packed: torch.Tensor, absmax: torch.Tensor, shape: Sequence[int], blocksize: int, rank: int, world: int
               ) -> Tuple[torch.Tensor, torch.Tensor, Tuple[int, int]]:
    """Rows ``[rank*M/world, (rank+1)*M/world)`` of an FP4 weight: plain contiguous slices (views)."""
    M, K = int(shape[0]), int(shape[1])
    if M % world or K % blocksize or K % 2:
        raise ValueError(f"cannot row-shard a {M}x{K} weight (blocksize {blocksize}) {world} ways")
    rows = M // world
    p = packed.reshape(-1)[rank * rows * K // 2:(rank + 1) * rows * K // 2].reshape(-1, 1)
    a = absmax.reshape(-1)[rank * rows * K // blocksize:(rank + 1) * rows * K // blocksize]
    return p, a, (rows, K)


def shard_cols(packed: torch.Tensor, absmax: torch.Tensor, shape: Sequence[int], blocksize: int, rank: int, world: int
               ) -> Tuple[torch.Tensor, torch.Tensor, Tuple[int, int]]:
    """Columns ``[rank*K/world, (rank+1)*K/world)`` re-packed into a contiguous ``[M, K/world]`` FP4 weight."""
    M, K = int(shape[0]), int(shape[1])
    if K % world or (K // world) % blocksize or (K // world) % 2:
        raise ValueError(f"cannot column-shard a {M}x{K} weight (blocksize {blocksize}) {world} ways")
    ks = K // world
    p = packed.reshape(M, K // 2)[:, rank * ks // 2:(rank + 1) * ks // 2].contiguous().reshape(-1, 1)
    a = absmax.reshape(M, K // blocksize)[:, rank * ks // blocksize:(rank + 1) * ks // blocksize].contiguous().reshape(-1)
    return p, a, (M, ks)


def concat_rows(weights: Sequence[Tuple[torch.Tensor, torch.Tensor, Sequence[int]]], blocksize: int
                ) -> Tuple[torch.Tensor, torch.Tensor, Tuple[int, int]]:
    """Stack FP4 weights that share ``in_features`` along the output dimension (fused QKV / gate-up projections):
    rows are independent and ``K % blocksize == 0``, so this is a plain concatenation of bytes and of scales, and one
    GEMV launch then serves all of them."""
    K = int(weights[0][2][1])
    if any(int(w[2][1]) != K for w in weights) or K % blocksize:
        raise ValueError("concat_rows needs equal in_features, divisible by the blocksize")
    packed = torch.cat([w[0].reshape(-1) for w in weights]).reshape(-1, 1)
    absmax = torch.cat([w[1].reshape(-1) for w in weights])
    return packed, absmax, (sum(int(w[2][0]) for w in weights), K)


def _on_host_backend(group) -> bool:
    return dist.get_backend(group) == "gloo"


def _all_reduce_sum(t: torch.Tensor, group) -> torch.Tensor:
    """Sum across ranks; with the gloo backend (CPU rehearsal of the RCCL path) GPU tensors are staged through the host."""
    if t.is_cuda and _on_host_backend(group):
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        return h.to(t.device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def _all_gather_last(y: torch.Tensor, world: int, group) -> torch.Tensor:
    src = y.contiguous().cpu() if (y.is_cuda and _on_host_backend(group)) else y.contiguous()
    parts = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(parts, src, group=group)
    return torch.cat(parts, dim=-1).to(y.device)


_COMMS = {}
ONESHOT_CAPACITY = 65536  # elements: the largest partial RowParallelFP4Linear sends through the one-shot path (8 MiB of slots at G = 8)


def oneshot_comm(group=None, capacity: int = ONESHOT_CAPACITY):
    """The process group's shared :class:`~torch_bnb_fp4.comm.OneShotAllReduce`, created on first use (a COLLECTIVE call: every
    rank must reach it together, outside any HIP-graph capture) and sized once for the path's own limit - it is never re-created
    behind the caller's back (a regrow would run collectives, an allocation and a device synchronisation inside ``forward``)."""
    from .comm import OneShotAllReduce

    key = (group, torch.cuda.current_device())  # the group object itself (id() values are recycled after GC)
    comm = _COMMS.get(key)
    if comm is None:
        comm = _COMMS[key] = OneShotAllReduce(group, capacity=max(int(capacity), ONESHOT_CAPACITY))
    elif comm.capacity < capacity:
        raise ValueError(f"oneshot_comm: {capacity} elements exceed the communicator's capacity of {comm.capacity}; larger partials go "
                         "through torch.distributed (allreduce='dist')")
    return comm


def check_oneshot(group=None) -> None:
    """Synchronous health check of the group's one-shot communicator, for a sync point of the decode loop OUTSIDE graph capture:
    raises if a reduction timed out waiting for a peer since the last check (its outputs were NaN).  No-op if none exists yet."""
    comm = _COMMS.get((group, torch.cuda.current_device()))
    if comm is not None:
        comm.check()


def check_oneshot_collective(group=None) -> None:
    """:func:`check_oneshot` for the whole group (collective - every rank calls it at the same sync point): if a reduction timed out
    on ANY rank since the last check, EVERY rank raises, so no rank carries on with activations its peers do not share
    (:meth:`OneShotAllReduce.check_collective`).  No-op on every rank if the group has no communicator yet."""
    comm = _COMMS.get((group, torch.cuda.current_device()))
    if comm is not None:
        comm.check_collective()


def close_oneshot(group=None) -> None:
    """Release the group's one-shot communicator (collective: peers unmap each other's buffers after a barrier)."""
    comm = _COMMS.pop((group, torch.cuda.current_device()), None)
    if comm is not None:
        comm.close()


def _quant_data(packed, absmax, shape, blocksize, bias, use_codebook_dequant=True) -> QuantData:
    state = QuantState(absmax, shape, fp4_code().to(packed.device), blocksize)
    return QuantData(packed, state, state.shape, original_lin=None, bias=bias, use_codebook_dequant=use_codebook_dequant)


class ColumnParallelFP4Linear(nn.Module):
    """M-split: this rank owns ``out_features / world`` rows (and that slice of the bias)."""

    def __init__(self, packed, absmax, shape, blocksize: int = 64, bias: Optional[torch.Tensor] = None, group=None,
                 gather_output: bool = True):
        super().__init__()
        self.group = group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        p, a, local = shard_rows(packed, absmax, shape, blocksize, self.rank, self.world)
        b = None if bias is None else bias.reshape(-1)[self.rank * local[0]:(self.rank + 1) * local[0]]
        self.quant_data = _quant_data(p, a, local, blocksize, b)
        self.gather_output = gather_output
        self.out_features, self.in_features = int(shape[0]), int(shape[1])

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y = self.quant_data.forward(x)
        if not self.gather_output or self.world == 1:
            return y
        return _all_gather_last(y, self.world, self.group)


class FusedColumnParallelFP4(nn.Module):
    """Several M-split projections that share their input as ONE launch per rank (q|k|v, or gate|up with the activation in
    the epilogue): each weight is row-sharded first, then this rank's shards are row-concatenated (or, for ``silu_mul``,
    row-interleaved), so the local output is ``cat([W_i[shard] x])`` - the same values the separate column-parallel layers
    would give, without a gather.  ``split_sizes`` are the local output widths."""

    def __init__(self, weights, blocksize: int = 64, group=None, epilogue: Optional[str] = None):
        super().__init__()
        from .fused import EPILOGUE_NONE, EPILOGUE_SILU_MUL_PAIRS, FusedFP4Linear, interleave_rows

        self.group = group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        shards = [shard_rows(p, a, shp, blocksize, self.rank, self.world) for p, a, shp in weights]
        if epilogue == "silu_mul":
            if len(shards) != 2 or shards[0][2] != shards[1][2]:
                raise ValueError("silu_mul needs exactly a gate and an up projection of the same shape")
            packed, absmax, local = interleave_rows((shards[0][0], shards[0][1]), (shards[1][0], shards[1][1]), shards[0][2], blocksize)
            self.layer = FusedFP4Linear.from_packed(packed, absmax, local, blocksize, None, EPILOGUE_SILU_MUL_PAIRS)
            self.split_sizes = [shards[0][2][0]]
        elif epilogue is None:
            packed, absmax, local = concat_rows(shards, blocksize)
            self.layer = FusedFP4Linear.from_packed(packed, absmax, local, blocksize, None, EPILOGUE_NONE)
            self.split_sizes = [s[2][0] for s in shards]
        else:
            raise ValueError(f"unknown epilogue {epilogue!r}")
        self.in_features = int(weights[0][2][1])

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.layer(x)


class RowParallelFP4Linear(nn.Module):
    """K-split: this rank owns ``in_features / world`` columns; outputs are summed across ranks in f32."""

    _mm_f32_out = {}  # (dtype, device type) -> does torch.mm(a, b, out_dtype=float32) exist for these operands?  Probed once, tiny.

    @classmethod
    def _has_mm_f32_out(cls, like: torch.Tensor) -> bool:
        """Capability probe on 16x16 operands - never inferred from a failure of the real GEMM (an out-of-memory error there is a
        RuntimeError too, and must surface as such instead of quietly moving every later call to the slower f32 path)."""
        key = (like.dtype, like.device.type)
        ok = cls._mm_f32_out.get(key)
        if ok is None:
            try:
                a = torch.zeros(16, 16, dtype=like.dtype, device=like.device)
                ok = torch.mm(a, a, out_dtype=torch.float32).dtype == torch.float32
            except torch.cuda.OutOfMemoryError:
                raise
            except (NotImplementedError, RuntimeError, TypeError):
                ok = False
            cls._mm_f32_out[key] = ok
        return ok

    def __init__(self, packed, absmax, shape, blocksize: int = 64, bias: Optional[torch.Tensor] = None, group=None,
                 input_is_parallel: bool = False, allreduce: str = "dist", reduce_single_rank: bool = False):
        super().__init__()
        if allreduce not in ("dist", "oneshot"):
            raise ValueError(f"allreduce must be 'dist' or 'oneshot', got {allreduce!r}")
        self.group = group
        self.allreduce = allreduce
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        # A one-rank group needs no exchange and normally gets none.  reduce_single_rank=True issues the collective all the same (a sum
        # over one rank is the identity): the only way to send this layer's real call sequence through RCCL / the one-shot kernel on a
        # one-GPU box (bench.py FP4_BENCH_FORCE_GROUP=1).
        self.reduces = self.world > 1 or bool(reduce_single_rank)
        p, a, local = shard_cols(packed, absmax, shape, blocksize, self.rank, self.world)
        self.quant_data = _quant_data(p, a, local, blocksize, None)
        self.bias = bias
        self.blocksize = blocksize
        self.local_shape = local
        self.input_is_parallel = input_is_parallel
        self.out_features, self.in_features = int(shape[0]), int(shape[1])

    def check(self) -> None:
        """Raise if a one-shot reduction of this layer's group timed out since the last check (synchronises the device; call it at
        a sync point of the decode loop, never inside graph capture).  ``allreduce="dist"`` has nothing to check."""
        if self.allreduce == "oneshot":
            check_oneshot(self.group)

    def forward(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        ks = self.local_shape[1]
        xs = x if self.input_is_parallel else x[..., self.rank * ks:(self.rank + 1) * ks]
        qd = self.quant_data
        single = xs.numel() == ks
        if single:  # single token: fused GEMV, raw f32 accumulator out
            part = ext.gemv_fp4_partial(xs.reshape(1, ks).contiguous(), qd.A.t(), qd.absmax, self.blocksize, list(self.local_shape))
            part = part.view(*x.shape[:-1], self.out_features)
        else:
            # batch / sequence: the reference's batch semantics (dequantise to the activation dtype, then a dense GEMM,
            # torch_bnb_fp4/__init__.py:423-436) with the shard's product kept in f32 - 16-bit operands on the matrix cores, f32
            # accumulation AND f32 output (aten::mm.dtype), so the partial is not rounded to T before the ranks are summed.  (Round 2
            # dequantised the shard to f32 and multiplied in f32: twice the transient weight and the far slower f32 GEMM at prefill
            # sizes.)  f32 activations, or a build without the f32-output GEMM, take that f32 path.
            if not qd.compute_dtype_set:
                qd.set_compute_type(xs)
            x2 = xs.reshape(-1, ks)
            part = None
            if x2.dtype in (torch.float16, torch.bfloat16) and self._has_mm_f32_out(x2):
                w16 = ext.dequantize_fp4_codebook(qd.A, qd.absmax, qd.code, qd.M, qd.N, qd.blocksize, qd.numel,
                                                  ScalarType.from_torch_dtype(x2.dtype).value)
                part = torch.mm(x2, w16.t(), out_dtype=torch.float32)  # errors here (out of memory included) are the caller's to see
                del w16
            if part is None:
                w32 = ext.dequantize_fp4_codebook(qd.A, qd.absmax, qd.code, qd.M, qd.N, qd.blocksize, qd.numel, ScalarType.float32.value)
                part = torch.nn.functional.linear(x2.float(), w32)
            part = part.view(*x.shape[:-1], self.out_features)
        if self.reduces and self.allreduce == "oneshot" and part.is_cuda and part.numel() <= ONESHOT_CAPACITY:
            # (latency-bound sizes only: decode and small batches; a prefill-sized partial goes through torch.distributed below)
            # one launch: publish into every peer's slots, gather, sum in rank order, round once, bias / residual on top
            comm = oneshot_comm(self.group)
            bias = None if self.bias is None else self.bias.to(x.dtype)
            if bias is not None and part.numel() != self.out_features:
                bias = bias.expand(part.shape).contiguous()
            return comm.reduce(part, x.dtype, bias, residual)
        if self.reduces:
            part = _all_reduce_sum(part, self.group)
        y = part.to(x.dtype)
        if self.bias is not None:
            y = y + self.bias.to(x.dtype)
        return y if residual is None else y + residual
