"""Peer-slot communicator for the one-shot all-reduce of the K-split layers (``fp4_hip_allreduce_oneshot``).

One :class:`OneShotAllReduce` per process group: every rank allocates a slot buffer on its GPU, the 64-byte IPC handles
travel once through ``torch.distributed`` (any backend: this is setup, not the data path), every rank maps its peers'
buffers, and from then on a reduction is ONE kernel launch on the current stream - no RCCL call, no host
synchronisation, HIP-graph capturable.  The reference has no multi-GPU path; design: SURVEY section 8e.

All ranks must call :meth:`reduce` the same number of times in the same order on one stream each.  A peer that never
shows up does not hang the GPU: the kernel gives up after ``timeout_us``, writes NaN and records the event in the
buffer's status word; :meth:`check` (synchronous) raises on it.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist

from ._ext import ext
from .dtypes import ScalarType

_KINDS = {0: "uncached", 1: "fine-grained", 2: "default"}


def _device_identity(device: torch.device) -> str:
    """Something that tells two GPUs of one node apart (for "do my peers sit on other devices?")."""
    props = torch.cuda.get_device_properties(device)
    uuid = getattr(props, "uuid", None)
    return str(uuid) if uuid is not None else f"{os.uname().nodename}:{device.index}:{getattr(props, 'pci_bus_id', '')}"


class OneShotAllReduce:
    """Launch-time requirement: ``HSA_ENABLE_IPC_MODE_LEGACY=0`` must be in the environment BEFORE the process starts (the host
    driver only supports dmabuf IPC; the HIP runtime reads the variable when it initialises, so setting it here would be too late).

    Memory: the slot buffer is written by peers, so it is allocated uncached (or fine-grained) where the runtime offers it.  If
    only plain ``hipMalloc`` memory is available (``memory_kind == "default"``) AND a peer sits on another device, the owner's L2
    may serve stale lines for peer-written slots: the tag-in-granule design turns that into time-outs and NaN rather than wrong
    sums, but the path is refused unless ``FP4_COMM_ALLOC=default`` asks for it explicitly."""

    def __init__(self, group=None, capacity: int = 16384, device: Optional[torch.device] = None, timeout_us: int = 2_000_000):
        if not dist.is_initialized():
            raise RuntimeError("OneShotAllReduce needs an initialised torch.distributed process group")
        if os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") != "0":
            import warnings

            warnings.warn("OneShotAllReduce: HSA_ENABLE_IPC_MODE_LEGACY=0 is not in this process's environment; it has to be exported "
                          "before the process starts, or hipIpcGetMemHandle / hipIpcOpenMemHandle fail with 'invalid argument'")
        self.group = group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.capacity = int(capacity)
        self.timeout_us = int(timeout_us)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        # (set-up failures are made COLLECTIVE: what goes wrong on one rank is raised on every rank at once, never one exception here and
        #  the other ranks parked in the next collective until the process group's time-out)
        try:
            self._own, handle, kind = ext.comm_alloc(self.world, self.capacity, self.device.index)
            alloc_problem = None
        except Exception as exc:
            self._own, handle, kind, alloc_problem = None, b"", -1, f"allocating / exporting the slot buffer failed: {exc}"
        self.memory_kind = _KINDS.get(kind, str(kind))
        handles = [None] * self.world
        dist.all_gather_object(handles, (self.rank, os.getpid(), bytes(handle), _device_identity(self.device), self.memory_kind, alloc_problem),
                               group=group)
        failed = [(r, why) for r, _, _, _, _, why in handles if why]
        if failed:
            if self._own is not None:
                ext.comm_free(self._own)
                self._own = None
            raise RuntimeError("OneShotAllReduce: " + "; ".join(f"rank {r}: {why}" for r, why in failed)
                               + f" (HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')!r} in this process). Use allreduce='dist' (RCCL)")
        handles = [h[:5] for h in handles]
        other_devices = sorted({r for r, _, _, ident, _ in handles if ident != _device_identity(self.device)})
        plain = sorted({r for r, _, _, _, k in handles if k == "default"})
        if other_devices and plain and os.environ.get("FP4_COMM_ALLOC") != "default":
            ext.comm_free(self._own)
            self._own = None
            raise RuntimeError(f"OneShotAllReduce: rank(s) {plain} could only allocate plain device memory (memory_kind 'default'; this rank: "
                               f"'{self.memory_kind}') and the group spans several devices (peers of rank {self.rank} on other devices: "
                               f"{other_devices}): peer-written slots may be served stale from the owner's L2.  Use allreduce='dist' "
                               "(RCCL), or set FP4_COMM_ALLOC=default to accept time-outs / NaN as the failure mode")
        # Mapping the peers' buffers is the step that has never run across devices (one-GPU boxes only): a failure on ONE rank must
        # become the same, immediate error on EVERY rank - not one exception here and seven ranks parked in the barrier below until
        # the process group's time-out.  So every rank reports how its mappings went before anybody proceeds.
        self._peers, self._opened = [], []
        problem = None
        for r, pid, h, _, _ in handles:
            if r == self.rank:
                self._peers.append(self._own)
            elif pid == os.getpid():
                problem = problem or "two ranks in one process cannot share an IPC handle"
                self._peers.append(0)
            else:
                try:
                    p = ext.comm_open(h, self.device.index)
                except Exception as exc:  # hipIpcOpenMemHandle refused (peer access, IPC mode, driver): reported collectively below
                    problem = problem or f"mapping rank {r}'s buffer failed: {exc}"
                    self._peers.append(0)
                    continue
                self._peers.append(p)
                self._opened.append(p)
        reports = [None] * self.world
        dist.all_gather_object(reports, (self.rank, problem), group=group)
        failed = [(r, why) for r, why in reports if why]
        if failed:
            for p in self._opened:
                try:
                    ext.comm_close(p)
                except Exception:
                    pass
            dist.barrier(group=group)  # every peer has unmapped our buffer before it is freed
            ext.comm_free(self._own)
            self._own, self._peers, self._opened = None, [], []
            raise RuntimeError("OneShotAllReduce: the peer buffers could not be mapped on every rank ("
                               + "; ".join(f"rank {r}: {why}" for r, why in failed)
                               + f"); HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')!r} in this process. "
                               "Use allreduce='dist' (RCCL)")
        dist.barrier(group=group)  # nobody starts reducing before every buffer is mapped everywhere

    def reduce(self, partial: torch.Tensor, out_dtype: torch.dtype, bias: Optional[torch.Tensor] = None,
               residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``T(sum over ranks of partial) (+ bias) (+ residual)``; ``partial`` is this rank's float32 tensor."""
        if partial.numel() > self.capacity:
            raise ValueError(f"OneShotAllReduce: {partial.numel()} elements exceed the capacity of {self.capacity}")
        return ext.allreduce_oneshot(partial.contiguous(), self._peers, self.rank, self.capacity,
                                     ScalarType.from_torch_dtype(out_dtype).value, bias, residual, self.timeout_us)

    def status(self):
        """``(completed calls, busy workgroups, status word, lanes that timed out)`` - synchronises the device."""
        torch.cuda.synchronize(self.device)
        return tuple(ext.comm_status(self._own))

    def reset_status(self) -> None:
        """Zero the sticky status word and the timed-out lane count (synchronises the device; the epoch is kept, so the ranks stay
        in step).  Call it at a sync point with no reduction of this rank in flight."""
        torch.cuda.synchronize(self.device)
        ext.comm_clear_status(self._own)

    def check(self) -> None:
        """Raise if a reduction of THIS rank timed out since the last check; the status is cleared before raising, so one transient
        time-out is reported once.  Local view only: the peer the rank gave up on may hold a valid result for the same call, so after
        this raises on one rank the ranks' activations differ.  The communicator stays usable only while no rank lags by a whole call
        (slots are double-buffered by call parity: a rank that carries on rewrites call n's slot at call n + 2, and a peer further
        behind than that times out in turn and writes NaN - never a wrong finite value).  Where the decision to continue matters, use
        :meth:`check_collective`, which makes it for the whole group."""
        epoch, _, status, lanes = self.status()
        if status:
            self.reset_status()
            raise RuntimeError(self._timeout_message(status, lanes, epoch))

    def _timeout_message(self, status: int, lanes: int, epoch: int) -> str:
        return (f"OneShotAllReduce: rank {self.rank} timed out waiting for rank {(status & 0xFF) - 1} in call "
                f"{status >> 8} ({lanes} lanes gave up; {epoch} calls completed): outputs of that call are NaN")

    def check_collective(self) -> None:
        """The group's view, for the sync point of a decode loop (collective: every rank must call it; never inside graph capture).
        Each rank's status word is gathered over the process group the communicator was built on; if ANY rank timed out since the
        last check, every rank clears its status and raises the same error, naming the ranks that gave up - so no rank carries on
        with a hidden state its peers do not share."""
        epoch, _, status, lanes = self.status()
        mine = torch.tensor([self.rank, status, lanes, epoch], dtype=torch.int64)
        backend = dist.get_backend(self.group)
        if backend == "nccl":
            mine = mine.to(self.device)
        got = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(got, mine, group=self.group)
        bad = [tuple(int(v) for v in g.tolist()) for g in got if int(g[1]) != 0]
        if bad:
            if status:
                self.reset_status()
            raise RuntimeError("OneShotAllReduce: a reduction timed out on " + "; ".join(
                f"rank {r} (waiting for rank {(st & 0xFF) - 1} in call {st >> 8}, {ln} lanes, {ep} calls completed)" for r, st, ln, ep in bad)
                + f" - reported on every rank of the group (this is rank {self.rank}); outputs of those calls are NaN on the ranks named")

    def close(self) -> None:
        if getattr(self, "_own", None) is None:
            return
        torch.cuda.synchronize(self.device)
        try:
            dist.barrier(group=self.group)  # peers may still be reading our slots
        except Exception:
            pass
        for p in self._opened:
            ext.comm_close(p)
        ext.comm_free(self._own)
        self._own, self._peers, self._opened = None, [], []
