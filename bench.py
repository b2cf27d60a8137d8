#!/usr/bin/env python3
"""Headline benchmark: FP4 dequant GB/s (% of HBM peak) + fused-GEMV us/layer, 4096x4096 bf16.

One "step" = one pass of the hot path over a batch of R distinct 4096x4096 FP4 weight matrices
(blocksize 64): R blockwise dequants to bf16, then GEMV_REPS x R fused batch-1 GEMVs, every launch
through the C ABI of libtorch_bnb_fp4_hip.so.  The R matrices (R x 9.4 MB packed+scales, R x 32 MiB
of outputs) rotate through far more than the 256 MiB Infinity Cache, so every launch streams from
HBM ("HBM-cold"); same-buffer "hot" figures are reported separately.  Launches are replayed from
HIP graphs so the host never paces the GPU; the timed region still contains one kernel boundary
per launch, exactly as a decode loop would.

Contract: `python bench.py --gpus N --steps K --warmup W`.  N > 1: one rank per GPU; either started under torchrun
(WORLD_SIZE set) or plainly, in which case this script starts its own N workers as child processes
(`python -m torch.distributed.run ...`, launch_workers below) BEFORE it touches the GPU and relays rank 0's line;
W untimed warm-up steps, then exactly K timed steps between barrier + synchronize, MAX over ranks,
rank 0 prints ONE JSON line.  `value` = whole-job dequant GB/s (algorithmic bytes, inputs resident
in HBM); the GEMV figures, the HIP-event roofline of the dequant kernel and a pure-torch CPU
dequant baseline timed on the host cores ride along in the same line.  No data-path collective:
rows of W are independent, so ranks process independent shards (weak scaling).

stdout carries the ONE JSON line and nothing else (claim_stdout: RCCL's version banner, gloo's notes and every other print go to stderr).

Environment switches (all optional; none changes the timed region):
  FP4_BENCH_BACKEND=gloo      N > 1 with the ranks SHARING devices and gloo standing in for RCCL (one-GPU rehearsal of the N > 1 path)
  FP4_BENCH_FORCE_GROUP=1     take the N > 1 code path with one rank (the only way through real RCCL on a one-GPU box)
  FP4_BENCH_C3=0              skip the sanity-MLP leg (BASELINE config 3: the reference's published table, re-measured; N = 1)
  FP4_BENCH_C4=0 / FP4_BENCH_C4_LAYERS=n      skip / shorten the Mistral-7B decode leg (N = 1)
  FP4_BENCH_C5=0 / FP4_BENCH_C5_LAYERS=n      skip / shorten the Llama-3-8B tensor-parallel leg (N > 1)
  FP4_BENCH_C5_ONESHOT=0      leave the one-shot all-reduce out of that leg;  FP4_BENCH_C5_GRAPH=1  also capture the RCCL variant in a HIP graph
  FP4_BENCH_QUANT_STACK=0     leave the quantiser's stack-of-R launch out (PMC passes: its persistent grid cannot be told from the per-matrix one)
  FP4_BENCH_CPU_THREADS=n     cap the host threads of the cpu_baseline leg (default: the cgroup's share, at most 64)
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import socket
import statistics
import subprocess
import sys
import time

# The host driver of this pool only supports dmabuf IPC, and the HIP runtime reads this variable once, when it initialises: it has to be
# in the environment before ANYTHING touches the GPU (RCCL's own peer mappings and the one-shot all-reduce's slot buffers both go through
# hipIpcGetMemHandle).  Module top, before `import torch`, so that the driver's own torchrun (which does not pass through
# launch_workers below) gets it as well.  HSA_IPC_ENV_AT_START records what the launcher had given us, for the result line.
HSA_IPC_ENV_AT_START = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (REPO, os.path.join(REPO, "torch-bnb-fp4_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

M = K = 4096
BLOCKSIZE = 64
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy rate is reported next to it
F16, F32, BF16 = 0, 1, 2


def dequant_bytes(m, k, bs, itemsize):
    return m * k // 2 + 4 * (m * k // bs) + m * k * itemsize


def gemv_bytes(m, k, bs, itemsize):
    return m * k // 2 + 4 * (m * k // bs) + k * itemsize + m * itemsize


class Lib:
    """The product library, bound through its C ABI (include/torch_bnb_fp4_hip.h)."""

    def __init__(self):
        import torch_bnb_fp4  # product package: fails loudly if the HIP extension is missing

        self.pkg = torch_bnb_fp4
        self.l = ctypes.CDLL(torch_bnb_fp4.HIP_LIBRARY_PATH)
        vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
        self.l.fp4_hip_dequantize_blockwise.argtypes = [vp, vp, vp, i32, i64, i32, i32, i32, vp]
        self.l.fp4_hip_gemv.argtypes = [vp, vp, vp, vp, vp, i64, i64, i32, i32, vp]
        self.l.fp4_hip_gemv_partial.argtypes = [vp, vp, vp, vp, i64, i64, i32, i32, vp]
        self.l.fp4_hip_gemm_small.argtypes = [vp, vp, vp, vp, vp, i64, i64, i64, i32, i32, vp]
        self.l.fp4_hip_quantize_blockwise.argtypes = [vp, i32, vp, vp, i64, i32, vp]
        self.l.fp4_hip_last_error.restype = ctypes.c_char_p

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(self.l.fp4_hip_last_error().decode())

    def dequant(self, packed, absmax, out, n, dtype=BF16, flags=0):
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        self._check(self.l.fp4_hip_dequantize_blockwise(packed.data_ptr(), absmax.data_ptr(), out.data_ptr(), BLOCKSIZE, n, dtype, 0, flags, s))

    def gemv(self, x, packed, absmax, out, m, k, dtype=BF16):
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        self._check(self.l.fp4_hip_gemv(x.data_ptr(), packed.data_ptr(), absmax.data_ptr(), None, out.data_ptr(), m, k, BLOCKSIZE, dtype, s))

    def gemm_small(self, x, packed, absmax, out, b, m, k, dtype=BF16):
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        self._check(self.l.fp4_hip_gemm_small(x.data_ptr(), packed.data_ptr(), absmax.data_ptr(), None, out.data_ptr(), b, m, k, BLOCKSIZE, dtype, s))

    def quantize(self, w, packed, absmax, n, dtype=BF16):
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        self._check(self.l.fp4_hip_quantize_blockwise(w.data_ptr(), dtype, packed.data_ptr(), absmax.data_ptr(), n, BLOCKSIZE, s))


def tp_ksplit_leg(lib, dist, backend, rank, world, dev, x, packed0, absmax0, barrier):
    """N > 1 only, outside the timed region: the one place the path has a real exchange step.  One 4096x4096 weight is
    column-sharded over the ranks (re-packed, torch_bnb_fp4.parallel.shard_cols); each rank computes its f32 partial
    (fp4_hip_gemv_partial) and the partials meet in a 16 KiB all-reduce (RCCL over xGMI): latency-bound at this size."""
    from torch_bnb_fp4 import parallel as par

    ks = K // world
    p_s, a_s, _ = par.shard_cols(packed0.view(-1, 1), absmax0, (M, K), BLOCKSIZE, rank, world)
    x_s = x[rank * ks:(rank + 1) * ks].contiguous()
    part = torch.empty(M, dtype=torch.float32, device=dev)
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def reduce_():
        if backend == "nccl":
            dist.all_reduce(part)
        else:  # gloo rehearsal: stage through the host
            h = part.cpu()
            dist.all_reduce(h)
            part.copy_(h)

    def step():
        lib._check(lib.l.fp4_hip_gemv_partial(x_s.data_ptr(), p_s.data_ptr(), a_s.data_ptr(), part.data_ptr(), M, ks, BLOCKSIZE, BF16, s))
        reduce_()

    out = {}
    for name, fn in (("gemv_partial_plus_allreduce_us", step), ("allreduce_16KiB_f32_us", reduce_)):
        for _ in range(20):
            fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(200):
            fn()
        barrier()
        out[name] = round((time.perf_counter() - t0) / 200 * 1e6, 2)
    out["note"] = f"K split {world} ways ({ks} columns per GPU), eager launches, backend {backend}; the collective is latency-bound"
    return out


def _rank_identity(rank, local, dev):
    """What this rank runs on, gathered from every rank into the N > 1 line: enough to tell N processes on N devices from N
    processes sharing one (uuid where the runtime exposes it, PCI bus id and device index otherwise)."""
    from torch_bnb_fp4 import comm

    p = torch.cuda.get_device_properties(dev)
    bus = getattr(p, "pci_bus_id", None)
    return {"rank": rank, "local_rank": local, "pid": os.getpid(), "host": os.uname().nodename, "device_index": dev.index,
            "device": comm._device_identity(dev), "pci_bus_id": None if bus is None else int(bus), "arch": getattr(p, "gcnArchName", ""),
            "hsa_ipc_env_at_start": HSA_IPC_ENV_AT_START, "hsa_ipc_env": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}


def group_proof(dist, backend, rank, world, local, dev):
    """N > 1: evidence, from REAL collectives on the data-path backend, that `world` ranks on (how many) distinct devices took part -
    so that a reader of the line does not have to trust `n_gpus` (an environment variable).  Tensors live on the GPU for the nccl (= RCCL)
    backend, i.e. the all-reduces below cross xGMI when the ranks sit on different devices; gloo (the one-GPU rehearsal) stages on the host."""
    where = dev if backend == "nccl" else "cpu"
    one = torch.ones(1, dtype=torch.int64, device=where)
    dist.all_reduce(one)  # every rank contributes 1
    tag = torch.tensor([rank + 1], dtype=torch.int64, device=where)
    dist.all_reduce(tag)  # every rank contributes rank + 1: N(N+1)/2 only if ranks 0..N-1 are each there once
    ids = [None] * world
    dist.all_gather_object(ids, _rank_identity(rank, local, dev))
    # a float sum whose value depends on every contribution, checked on every rank against the closed form (a transport that
    # delivered the right COUNT but the wrong DATA would show here)
    probe = (torch.arange(1024, dtype=torch.float32, device=where) + 1.0) * float(rank + 1)
    dist.all_reduce(probe)
    want = (torch.arange(1024, dtype=torch.float32, device=where) + 1.0) * float(world * (world + 1) // 2)
    data_ok = torch.tensor([int(torch.equal(probe, want))], dtype=torch.int64, device=where)
    dist.all_reduce(data_ok, op=dist.ReduceOp.MIN)
    try:
        rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception as exc:
        rccl = f"unavailable ({type(exc).__name__})"
    devices = sorted({i["device"] for i in ids})
    return {
        "backend": backend,
        "collective_library": ("RCCL " + rccl) if backend == "nccl" else f"{backend} (host-staged rehearsal; RCCL {rccl} not used)",
        "ranks_seen": int(one.item()),
        "rank_checksum": int(tag.item()),
        "rank_checksum_expected": world * (world + 1) // 2,
        "allreduce_data_ok_on_every_rank": bool(data_ok.item()),
        "distinct_devices": len(devices),
        "distinct_pids": len({(i["host"], i["pid"]) for i in ids}),
        "hosts": sorted({i["host"] for i in ids}),
        "ranks": sorted(ids, key=lambda i: i["rank"]),
        "ok": bool(one.item() == world and tag.item() == world * (world + 1) // 2 and data_ok.item() == 1
                   and (backend != "nccl" or len(devices) == world)),
    }


STAGES = ("started", "process-group-ready", "buffers-ready", "timed-region-done", "done")


def stage(name):
    """Workers tell the rank-side watchdog - and, under our own launching parent, the parent - how far they got.  The parent's channel is
    a small FILE per rank (FP4_BENCH_STAGE_DIR), never stdout: several ranks share one stdout pipe, writes above PIPE_BUF are not atomic,
    and rank 0's ONE JSON line must not be interleaved with anything."""
    if _WATCHDOG is not None:
        _WATCHDOG.stage = name
    d = os.environ.get("FP4_BENCH_STAGE_DIR")
    if d:
        try:
            with open(os.path.join(d, f"rank{os.environ.get('RANK', '0')}"), "a") as f:
                f.write(name + "\n")
        except OSError:
            pass


class Watchdog:
    """Every rank's own deadline (the driver starts the N > 1 ranks under torchrun itself, so the parent's deadline in
    launch_workers is not there to help): when it passes, the rank says on stderr which stage it was in, rank 0 prints
    the headline line if the timed region had already finished (marked "incomplete": the optional legs after it did
    not), and the process leaves with code 124 - a reason and an exit, never a silent hang."""

    def __init__(self, seconds, rank):
        import threading

        self.seconds, self.rank, self.stage, self.provisional = float(seconds), rank, "started", None
        self._timer = threading.Timer(self.seconds, self._fire)
        self._timer.daemon = True
        self._timer.start()

    def _fire(self):
        print(f"bench.py: rank {self.rank}: deadline of {self.seconds:.0f} s passed in stage '{self.stage}'; giving up", file=sys.stderr, flush=True)
        if self.provisional is not None:
            line = dict(self.provisional)
            line["incomplete"] = f"deadline of {self.seconds:.0f} s passed in stage '{self.stage}': figures after the timed region are missing"
            print_result(line)
        os._exit(124)

    def cancel(self):
        self._timer.cancel()


_WATCHDOG = None
_RESULT_OUT = None  # the process's ORIGINAL stdout, kept for the one result line (claim_stdout)


def claim_stdout():
    """stdout carries ONE JSON line - but native libraries write there too: RCCL prints a five-line version banner ("RCCL version : ...",
    "HIP version : ...", host name ...) on stdout when its first communicator comes up, gloo prints "[Gloo] Rank N is connected ..."
    (found with the one-rank RCCL rehearsal of round 4: a reader taking "the line on stdout" would have met the banner first).  So the
    worker keeps a private duplicate of the original stdout for its result line and points file descriptor 1 at stderr for everybody
    else - C-level printf included - before anything initialises the GPU or a process group."""
    global _RESULT_OUT
    if _RESULT_OUT is None:
        try:
            sys.stdout.flush()
            keep = os.dup(1)
            os.dup2(2, 1)
            _RESULT_OUT = os.fdopen(keep, "w")
        except OSError:  # no usable stdout / stderr descriptor (a caller that closed them): print where print goes, as before
            _RESULT_OUT = sys.stdout
    return _RESULT_OUT


def print_result(line):
    out = _RESULT_OUT or sys.stdout
    print(json.dumps(line), file=out, flush=True)


def _kill_group(proc, grace_s=5.0):
    """End the worker group WE started (its own session / process group id == proc.pid): TERM, then KILL."""
    import signal

    for sig, wait_s in ((signal.SIGTERM, grace_s), (signal.SIGKILL, 10.0)):
        try:
            os.killpg(proc.pid, sig)
        except (ProcessLookupError, PermissionError):
            pass
        try:
            proc.wait(timeout=wait_s)
            return
        except subprocess.TimeoutExpired:
            continue


def _die_with_parent():
    """Runs in the torchrun child between fork and exec: the workers live in a session of their own (so that the deadline can kill
    exactly them), which also takes them out of reach of a killpg aimed at this parent - so ask the kernel to TERM the child when
    the parent goes, however it goes (SIGKILL included; torchrun's own TERM handler then ends its ranks)."""
    try:
        import signal

        ctypes.CDLL(None, use_errno=True).prctl(1, int(signal.SIGTERM), 0, 0, 0)  # PR_SET_PDEATHSIG
    except Exception:
        pass


def _forward_termination(proc):
    """A TERM / HUP / INT sent to the launching parent (a driver's time limit, say) ends the worker group too, then the parent leaves
    with 128 + signal (through SystemExit, so that every `finally` on the way runs).  Returns a callable that puts the previous handlers back.  (Handlers can only be set from the main thread;
    elsewhere _die_with_parent is the only line of defence.)"""
    import signal

    previous = {}

    def handler(signum, _frame):
        # Nothing that waits in here: the main thread may be inside proc.wait() holding Popen's wait lock, and a handler that
        # waited on the same process would spin through both grace periods.  Leave through the normal flow instead - SystemExit
        # unwinds _wait_for_workers (whose `except BaseException` ends the worker group) and launch_workers' `finally`
        # (handlers restored, the stage directory removed) - and the process exits with 128 + signal.
        print(f"bench.py: signal {signum} received: ending the worker group", file=sys.stderr, flush=True)
        try:
            os.killpg(proc.pid, signal.SIGTERM)  # start the workers' shutdown at once; _kill_group escalates if they linger
        except (ProcessLookupError, PermissionError):
            pass
        raise SystemExit(128 + signum)

    for sig in (signal.SIGTERM, signal.SIGHUP, signal.SIGINT):
        try:
            previous[sig] = signal.signal(sig, handler)
        except ValueError:
            break

    def restore():
        for sig, old in previous.items():
            try:
                signal.signal(sig, old)
            except ValueError:
                pass

    return restore


def launch_workers(n, argv, script=None, timeout_s=None, out=None, deadline_s=480.0):
    """`python bench.py --gpus N` without torchrun: start the N ranks as CHILD processes (never exec: a process that has
    touched the GPU must not be replaced, and this parent stays GPU-free so that it can relay), wait for them, print
    rank 0's JSON line on our stdout and return the children's exit code (non-zero if any rank failed: no retry).

    The whole run has ONE deadline (`deadline_s`, `--deadline`; `timeout_s` is its older name): when it passes, the worker
    group (a session of its own) is killed, the stage every rank last reported is printed - so a rank stuck in RCCL
    initialisation reads "rank 5: never got past 'started'" instead of a silent driver kill - and the exit code is 124."""
    import threading

    if timeout_s is not None:
        deadline_s = timeout_s
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["FP4_BENCH_DEADLINE_S"] = str(deadline_s)
    import tempfile

    stage_dir = tempfile.mkdtemp(prefix="fp4_bench_stages_")
    env["FP4_BENCH_STAGE_DIR"] = stage_dir
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), script or os.path.abspath(__file__), *argv]
    out = out or sys.stdout
    t_start = time.monotonic()
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, start_new_session=True, preexec_fn=_die_with_parent)
    state = {"line": None}
    restore = _forward_termination(proc)

    def last_stages():
        got = {}
        for r in range(n):
            try:
                lines = open(os.path.join(stage_dir, f"rank{r}")).read().split()
                if lines:
                    got[r] = lines[-1]
            except OSError:
                pass
        return got

    def pump():  # rank 0 prints exactly one JSON line; anything else a child writes to stdout goes to our stderr
        for raw in proc.stdout:
            txt = raw.strip()
            at = txt.find('{"metric"')
            if at >= 0:  # tolerate foreign text glued to the line (several processes share the pipe)
                try:
                    obj, end = json.JSONDecoder().raw_decode(txt[at:])
                    state["line"] = json.dumps(obj)
                    txt = (txt[:at] + " " + txt[at + end:]).strip()
                except ValueError:
                    pass
            if txt:
                print(txt, file=sys.stderr, flush=True)

    reader = threading.Thread(target=pump, daemon=True)
    reader.start()
    try:
        return _wait_for_workers(proc, reader, state, last_stages, n, deadline_s, t_start, out)
    finally:
        import shutil

        restore()
        shutil.rmtree(stage_dir, ignore_errors=True)


def _wait_for_workers(proc, reader, state, last_stages, n, deadline_s, t_start, out):
    try:
        rc = proc.wait(timeout=deadline_s)
    except subprocess.TimeoutExpired:
        _kill_group(proc)
        reader.join(timeout=5.0)
        print(f"bench.py: the {n}-rank worker group did not finish within the deadline of {deadline_s:.0f} s "
              f"(ran {time.monotonic() - t_start:.0f} s); killed.  Last stage reported by each rank:", file=sys.stderr, flush=True)
        stages = last_stages()
        for r in range(n):
            last = stages.get(r)
            what = f"last stage '{last}'" if last else "no stage reported (never reached main(): import or launcher problem)"
            if last == "started":
                what += " - never finished process-group initialisation (RCCL / rendezvous)"
            print(f"  rank {r}: {what}", file=sys.stderr, flush=True)
        if state["line"] is None:
            print("  no result line was produced", file=sys.stderr, flush=True)
        return 124
    except BaseException:
        _kill_group(proc)
        raise
    reader.join(timeout=10.0)
    line = state["line"]
    if line is not None:
        print(line, file=out, flush=True)
    if rc != 0:
        stages = last_stages()
        stuck = [f"rank {r}: '{stages.get(r, 'nothing')}'" for r in range(n) if stages.get(r) != "done"]
        print(f"bench.py: the {n}-rank worker group exited with code {rc}" + (f" (did not finish: {', '.join(stuck)})" if stuck else ""),
              file=sys.stderr, flush=True)
        return rc
    if line is None:
        print("bench.py: the worker group printed no result line", file=sys.stderr, flush=True)
        return 1
    return 0


def strong_split_leg(lib, dist, backend, rank, world, dev, x, packed, absmax, barrier):
    """N > 1 only, outside the timed region: STRONG scaling of the headline shape.  Every 4096x4096 weight is split by rows
    over the ranks - a row range is a contiguous slice of the packed bytes and of the scales, no re-packing and no collective - and each
    rank dequantises / GEMVs only its M/N rows.  At 8 ranks that is 512 rows = 5.4 MB of dequant traffic per GPU and launch: the launch
    boundary, not HBM, sets the time, which is what the figure is there to show."""
    rows = M // world
    n_s = rows * K
    R = len(packed)
    p_s = [p[rank * n_s // 2:(rank + 1) * n_s // 2] for p in packed]
    a_s = [a[rank * n_s // BLOCKSIZE:(rank + 1) * n_s // BLOCKSIZE] for a in absmax]
    o_s = [torch.empty(n_s, dtype=torch.bfloat16, device=dev) for _ in range(min(R, 16))]
    y_s = torch.empty(rows, dtype=torch.bfloat16, device=dev)
    dq = capture(lambda: [lib.dequant(p_s[i], a_s[i], o_s[i % len(o_s)], n_s) for i in range(R)])
    gv = capture(lambda: [lib.gemv(x, p_s[i], a_s[i], y_s, rows, K) for i in range(R)])
    out = {"rows_per_gpu": rows}
    for name, rp in (("dequant_us", dq), ("gemv_us", gv)):
        time_replays(rp, 3, R)
        barrier()
        us = time_replays(rp, 7, R)[0]
        t = torch.tensor([us], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        out[name] = round(t.item(), 3)
    out["dequant_aggregate_gbps"] = round(world * dequant_bytes(rows, K, BLOCKSIZE, 2) / out["dequant_us"] / 1e3, 1)
    out["gemv_aggregate_gbps"] = round(world * gemv_bytes(rows, K, BLOCKSIZE, 2) / out["gemv_us"] / 1e3, 1)
    out["note"] = (f"one 4096x4096 weight split by rows over {world} GPUs (contiguous slices, no collective): max over ranks of the us per "
                   "launch, HBM-cold, HIP-graph replay; launch-bound at this size - compare with the weak-scaling headline")
    return out


def _ipc_env():
    """The state of the variable the peer mappings depend on: what the launcher gave this process, and what it ran with."""
    return {"hsa_ipc_env_at_start": HSA_IPC_ENV_AT_START, "hsa_ipc_env": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
            "fp4_comm_alloc": os.environ.get("FP4_COMM_ALLOC")}


def c5_leg(dist, backend, rank, world, dev, barrier, tokens=16):
    """N > 1 only, outside the timed region: BASELINE config 5 - every FP4 Linear of a Llama-3-8B shaped decoder through
    Column/RowParallelFP4Linear (q/k/v/gate/up M-split, o/down K-split + one f32 all-reduce each), batch-1 decode."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import decode_bench as db

    cfg = dict(db.MODELS["llama3-8b"])
    cfg["layers"] = int(os.environ.get("FP4_BENCH_C5_LAYERS", cfg["layers"]))
    out = {"model": "llama3-8b shapes", "layers": cfg["layers"], "backend": backend}
    # eager first (always valid); a HIP graph around a live RCCL communicator only on request (FP4_BENCH_C5_GRAPH=1),
    # the one-shot peer-slot all-reduce (no RCCL call inside the step) is always captured
    modes = [("dist", os.environ.get("FP4_BENCH_C5_GRAPH", "0") == "1" and backend == "nccl")]
    if os.environ.get("FP4_BENCH_C5_ONESHOT", "1") == "1":
        # The peer-slot path has only ever run with ranks sharing one GPU: before it is trusted across xGMI, ONE reduction
        # with a 50 ms give-up is checked against torch.distributed on every rank; any doubt and the leg is skipped.
        try:
            from torch_bnb_fp4 import parallel as par

            comm = par.oneshot_comm(None)
            probe = torch.arange(4096, device=dev, dtype=torch.float32) * (rank + 1) * 0.25
            # first call with a generous bound: it carries each rank's one-time costs (code-object load, first touch of the peers'
            # mappings), which can skew the ranks by more than the 50 ms the check proper allows
            comm.timeout_us = 2_000_000
            comm.reduce(probe, torch.float32)
            torch.cuda.synchronize()
            barrier()
            comm.timeout_us = 50_000
            got = comm.reduce(probe, torch.float32)
            want = probe.clone()
            if backend == "nccl":
                dist.all_reduce(want)
            else:
                h = want.cpu()
                dist.all_reduce(h)
                want = h.to(dev)
            ok = torch.tensor([int(comm.status()[2] == 0 and torch.equal(got, want))], device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            out["oneshot_selfcheck"] = {"ok": bool(ok.item()), "memory_kind": comm.memory_kind, "status_word": int(comm.status()[2]),
                                        "timed_out_lanes": int(comm.status()[3]), **_ipc_env()}
            comm.timeout_us = 1_000_000  # the leg itself: host jitter between the ranks' launches must not read as a missing peer
            if ok.item():
                modes.append(("oneshot", True))
        except Exception as exc:
            out["oneshot_selfcheck"] = {"ok": False, "error": repr(exc)[:300], **_ipc_env()}
    for ar, graph in modes:
        try:
            token, h0, meta = db.build_token_fn(cfg, dev, torch.bfloat16, world, rank, allreduce=ar, lm_head=False, epilogues=True,
                                                tensor_parallel=True)
            t = db.time_tokens(token, h0, tokens, graph=graph, barrier=barrier)
            best = t["graph_s"] or t["eager_s"]
            out[ar] = {"eager_ms_per_token": round(t["eager_s"] * 1e3, 3),
                       "graph_ms_per_token": None if t["graph_s"] is None else round(t["graph_s"] * 1e3, 3),
                       "graph_error": t["graph_error"],
                       "fp4_gbps_per_gpu": round(meta["fp4_bytes_per_token_per_gpu"] / best / 1e9, 1)}
            if ar == "oneshot":
                from torch_bnb_fp4 import parallel as par

                st = par.oneshot_comm(None).status()
                out[ar]["timeouts"] = int(st[3])  # lanes that gave up waiting for a peer (must be 0 for the figure to count)
                # collective (every rank is here: time_tokens ends in a barrier): raises on EVERY rank (-> "error" below) if a reduction
                # of the leg timed out on ANY rank - rank 0's own status word says nothing about a peer whose outputs were NaN
                par.check_oneshot_collective(None)
            out["allreduces_per_token"] = meta["allreduces_per_token"]  # issued inside the K-split layers (a one-rank group included)
            out["collective_ranks"] = meta["collective_ranks"]  # 1 = FP4_BENCH_FORCE_GROUP rehearsal: the calls run, no data crosses a link
            out["fp4_bytes_per_token_per_gpu"] = meta["fp4_bytes_per_token_per_gpu"]
            del token, h0
        except Exception as exc:
            out[ar] = {"error": repr(exc)[:300]}
        torch.cuda.empty_cache()
    out["note"] = ("q|k|v shards in one launch, gate|up shards interleaved with silu(g)*u in the epilogue, residual adds inside the "
                   "K-split layers; 2 all-reduces of 16 KiB f32 per layer (latency-bound); eager launches are host-bound; attention "
                   "replaced by identity, lm_head left out")
    return out


def fused_epilogue_leg(lib, dev):
    """us per launch of fp4_hip_gemv_fused at Mistral-7B / Llama-3-8B layer shapes, HBM-cold rotation, HIP-graph replay."""
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
    lib.l.fp4_hip_gemv_fused.argtypes = [vp, vp, vp, vp, vp, vp, i64, i64, i32, i32, i32, vp]
    out = {}
    for name, (m, k), epi in (("o_plus_residual_4096x4096", (4096, 4096), 0), ("gate_up_silu_mul_28672x4096", (28672, 4096), 1),
                              ("down_plus_residual_4096x14336", (4096, 14336), 0)):
        n = m * k
        r = max(8, min(32, int(1.0e9 / (n * 0.5625))))
        gen = torch.Generator(device=dev).manual_seed(5)
        packed = [torch.randint(0, 256, (n // 2,), dtype=torch.uint8, device=dev, generator=gen) for _ in range(r)]
        absmax = [torch.rand(n // BLOCKSIZE, device=dev, generator=gen) * 0.02 + 0.002 for _ in range(r)]
        x = torch.randn(k, device=dev).to(torch.bfloat16)
        m_out = m // 2 if epi else m
        res = torch.randn(m_out, device=dev).to(torch.bfloat16)
        y = torch.empty(m_out, dtype=torch.bfloat16, device=dev)

        def run():
            s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            for i in range(r):
                lib._check(lib.l.fp4_hip_gemv_fused(x.data_ptr(), packed[i].data_ptr(), absmax[i].data_ptr(), None, res.data_ptr(),
                                                    y.data_ptr(), m, k, BLOCKSIZE, BF16, epi, s))

        def plain():
            for i in range(r):
                lib.gemv(x, packed[i], absmax[i], yp, m, k)

        yp = torch.empty(m, dtype=torch.bfloat16, device=dev)
        out[name] = round(time_replays(capture(run), 5, r)[0], 3)
        out[name.split("_")[0] + "_plain_gemv_same_weight"] = round(time_replays(capture(plain), 5, r)[0], 3)
        del packed, absmax
    # q|k|v row-concatenated (4096 + 1024 + 1024 rows): a plain GEMV, the fourth FP4 launch of a decoder layer
    m, k = 6144, 4096
    n = m * k
    r = 32
    gen = torch.Generator(device=dev).manual_seed(6)
    packed = [torch.randint(0, 256, (n // 2,), dtype=torch.uint8, device=dev, generator=gen) for _ in range(r)]
    absmax = [torch.rand(n // BLOCKSIZE, device=dev, generator=gen) * 0.02 + 0.002 for _ in range(r)]
    x = torch.randn(k, device=dev).to(torch.bfloat16)
    yq = torch.empty(m, dtype=torch.bfloat16, device=dev)
    out["qkv_row_concat_plain_gemv_6144x4096"] = round(time_replays(capture(lambda: [lib.gemv(x, packed[i], absmax[i], yq, m, k) for i in range(r)]), 5, r)[0], 3)
    return out


def c3_leg(pkg):
    """Outside the timed region: BASELINE config 3, the only speed table the reference itself publishes (README.md:100-159), re-measured
    on this build and this box: TestModel(768, 2048, 4, 64), [1,768] / [2,768], fp32 / fp16 / bf16, dense vs FP4, eager and HIP-graph
    replay, the README's own figures beside them and the FP4-calls / dense-nn.Linear / GELU split (tools/sanity_bench.py; bounded)."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import sanity_bench as sb

    t0 = time.perf_counter()
    table = sb.c3_table(pkg, verbose=False)  # (the prose that explains the table: tools/sanity_bench.py, profiles/r05_c3_sanity_mlp.json)
    table["leg_seconds"] = round(time.perf_counter() - t0, 1)
    return table


def c4_leg(dev, tokens=24, fused_us=None):
    """Outside the timed region: ms per token of batch-1 decode through 32 Mistral-7B shaped layers (+ dense lm_head), replayed
    from a HIP graph: as separate launches, with q|k|v and gate|up row-concatenated, with the fused epilogues, and with the fused
    epilogues over `lean` glue (ONE elementwise launch per layer in place of the six-launch attention stand-in) - plus the sum of the
    FP4 kernels' own per-launch figures of this run (`fused_us`, HBM-cold) and the dense lm_head, so that the line says by itself how
    much of a token is FP4 kernels at their per-launch floor and how much is stand-in glue and launch boundaries."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import decode_bench as db

    cfg = dict(db.MODELS["mistral7b"])
    cfg["layers"] = int(os.environ.get("FP4_BENCH_C4_LAYERS", cfg["layers"]))
    out = {"model": "mistral7b shapes", "layers": cfg["layers"], "dtype": "bf16",
           "note": "synthetic FP4 bytes; attention replaced by identity; lm_head dense bf16; HIP-graph replay"}
    for name, kw in (("separate_launches", {}), ("row_concat", {"fuse": True}), ("row_concat_plus_epilogues", {"epilogues": True}),
                     ("row_concat_plus_epilogues_lean_glue", {"epilogues": True, "lean_glue": True})):
        token, h0, meta = db.build_token_fn(cfg, dev, torch.bfloat16, **kw)
        t = db.time_tokens(token, h0, tokens, graph=True)
        out[name] = {"graph_ms_per_token": None if t["graph_s"] is None else round(t["graph_s"] * 1e3, 3),
                     "eager_ms_per_token": round(t["eager_s"] * 1e3, 3), "fp4_launches_per_token": meta["fp4_linear_calls_per_token"]}
        out["hbm_floor_ms_per_token_at_8TBps"] = round((meta["fp4_bytes_per_token_per_gpu"] + meta["lm_head_bytes"]) / 8e12 * 1e3, 3)
        del token, h0
        torch.cuda.empty_cache()
    out["row_concat_plus_epilogues_lean_glue"]["glue"] = "one elementwise launch per layer, no rescale (tools/decode_bench.py --lean-glue)"
    keys = ("qkv_row_concat_plain_gemv_6144x4096", "o_plus_residual_4096x4096", "gate_up_silu_mul_28672x4096", "down_plus_residual_4096x14336")
    if fused_us and all(k in fused_us for k in keys):
        # the dense bf16 lm_head GEMV (262 MB: larger than the 256 MiB Infinity Cache, so every replay streams it from HBM)
        head = torch.nn.Linear(cfg["hidden"], cfg["vocab"], bias=False, device=dev, dtype=torch.bfloat16)
        h = torch.randn(1, cfg["hidden"], device=dev).to(torch.bfloat16)
        with torch.inference_mode():
            head_us = time_replays(capture(lambda: [head(h) for _ in range(4)]), 5, 4)[0]
        del head
        per_layer_us = sum(fused_us[k] for k in keys)
        kernels_ms = (cfg["layers"] * per_layer_us + head_us) / 1e3
        out["fp4_kernels_sum_ms"] = round(cfg["layers"] * per_layer_us / 1e3, 3)
        out["lm_head_dense_gemv_ms"] = round(head_us / 1e3, 3)
        out["decomposition"] = {
            "per_layer_fp4_launch_us": {k: fused_us[k] for k in keys},
            "fp4_kernels_plus_lm_head_ms": round(kernels_ms, 3),
            "glue_and_boundaries_ms": {n_: round(out[n_]["graph_ms_per_token"] - kernels_ms, 3)
                                       for n_ in ("row_concat_plus_epilogues", "row_concat_plus_epilogues_lean_glue") if out[n_]["graph_ms_per_token"]},
            "note": "fp4_kernels_sum_ms = layers x the four per-launch figures of `fused_epilogue_us` (same run, HBM-cold, one kernel boundary each); "
                    "what is left of a graph-replayed token is the attention stand-in's elementwise launches (six per layer by default, one with lean "
                    "glue) and their boundaries - not FP4 work",
        }
    return out


def _probe_lib():
    probe = ctypes.CDLL(os.path.join(REPO, "tools", "libfp4_stream_probe.so"))
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
    probe.fp4_probe_stream.argtypes = [i32, vp, vp, i64, vp]
    probe.fp4_probe_stream.restype = i32
    probe.fp4_probe_bytes.argtypes = [i32, i64]
    probe.fp4_probe_bytes.restype = i64
    return probe


PROBE_MODES = (("write_only", 0), ("read_only", 1), ("copy", 2), ("dequant_mix_1r_4w", 3), ("dequant_mix_1r_4w_loads_barrier_stores", 4))


def box_stream_leg(packed, outs):
    """What THIS box streams, in this very run, with the dequant kernel's own access geometry and no arithmetic (tools/stream_probe.hip:
    256-thread workgroups, 4 KiB contiguous per wave, 16-byte non-temporal accesses): write only, read only, copy, and the kernel's
    1-read : 4-write mix, each as R launches of 32 MiB of stores (or loads) rotating over the bench's own buffers - HBM-cold, HIP-graph
    replay, HIP events, exactly like the headline.  The spec figure (8 TB/s) stays the roofline peak; this says how far the kernel is from
    what the silicon in front of it delivers for its traffic shape."""
    probe = _probe_lib()
    R = len(outs)
    n = outs[0].numel() * outs[0].element_size()  # 32 MiB: the output of one 4096x4096 -> bf16 dequant
    assert packed[0].numel() * 4 >= n and n % 16384 == 0
    out = {}
    for name, mode in PROBE_MODES:
        def run(mode=mode):
            s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            for i in range(R):
                src = packed[i] if mode >= 3 else outs[(i + R // 2) % R]
                rc = probe.fp4_probe_stream(mode, src.data_ptr(), outs[i].data_ptr(), n, s)
                if rc != 0:
                    raise RuntimeError(f"fp4_probe_stream(mode {mode}) failed with code {rc}")

        rp = capture(run)
        time_replays(rp, 3, R)
        us = time_replays(rp, 7, R)[0]
        out[name] = round(probe.fp4_probe_bytes(mode, n) / us / 1e3, 1)
    # ... and a bare read of exactly the GEMV's bytes per launch (9 453 568 B = 577 probe tiles), same rotation: the batch-1 GEMV's
    # per-launch figure is launch + first byte + stream, and this is what that costs on this box with no arithmetic and no x at all
    n_gv = gemv_bytes(M, K, BLOCKSIZE, 2)
    if n_gv % 16384 == 0 and n_gv <= n:
        def run_gv():
            s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            for i in range(R):
                rc = probe.fp4_probe_stream(1, outs[(i + R // 2) % R].data_ptr(), outs[i].data_ptr(), n_gv, s)
                if rc != 0:
                    raise RuntimeError(f"fp4_probe_stream(read, {n_gv} B) failed with code {rc}")

        rp = capture(run_gv)
        time_replays(rp, 3, R)
        out["read_only_gemv_bytes_us"] = round(time_replays(rp, 7, R)[0], 3)
    return out


def box_stream_stack_leg(big_p, big_o):
    """The same four streams as ONE launch over the stack of R outputs (R x 32 MiB; the copy moves one half of it onto the other): the
    steady-state counterpart, away from the launch boundary - measured like `dequant_stack_of_R_one_launch_gbps` (four launches per
    replay, after ~25 ms of the same kernel)."""
    probe = _probe_lib()
    total = big_o.numel() * big_o.element_size()
    assert total % 32768 == 0 and big_p.numel() * 4 >= total
    out = {}
    for name, mode in PROBE_MODES:
        n = total // 2 if mode == 2 else total
        src = big_p if mode >= 3 else big_o
        dst_ptr = big_o.data_ptr() + (n if mode == 2 else 0)

        def run():
            s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            for _ in range(4):
                rc = probe.fp4_probe_stream(mode, src.data_ptr(), dst_ptr, n, s)
                if rc != 0:
                    raise RuntimeError(f"fp4_probe_stream(mode {mode}) failed with code {rc}")

        rp = capture(run)
        time_replays(rp, 15, 4)
        us = time_replays(rp, 9, 4)[0]
        out[name] = round(probe.fp4_probe_bytes(mode, n) / us / 1e3, 1)
    return out


def box_yardsticks(rf, read_bytes, write_bytes, per_launch, stack):
    """What this box would need to read `read_bytes` and write `write_bytes` back to back at its own bare stream rates (same run, same
    buffers, tools/stream_probe.hip), next to what the kernel achieved: an ESTIMATE from the per-launch probes (their read rate is
    launch-limited, so it is no ceiling - a kernel that overlaps its two directions beats it), one from the steady-state probes over
    the stack of R, and `best_of_both` (the better rate per direction): the one figure of the three a kernel should not exceed."""
    def serial(read_rate, write_rate):
        return round((read_bytes + write_bytes) / (write_bytes / write_rate + read_bytes / read_rate), 1)

    est = {"per_launch_probes": serial(per_launch["read_only"], per_launch["write_only"])}
    if "write_only" in stack:
        est["stack_probes"] = serial(stack["read_only"], stack["write_only"])
        est["best_of_both"] = serial(max(per_launch["read_only"], stack["read_only"]), max(per_launch["write_only"], stack["write_only"]))
    rf["box_serial_rw_estimate_gbps"] = est
    rf["frac_of_box_serial_rw"] = {k: round(rf["achieved"] / v, 4) for k, v in est.items() if k != "stack_probes"}
    if "steady_state_gbps" in rf and "stack_probes" in est:
        rf["steady_state_frac_of_box_serial_rw"] = {k: round(rf["steady_state_gbps"] / est[k], 4) for k in ("stack_probes", "best_of_both")}


def capture(fn):
    """Capture fn() into a HIP graph (after one eager run) and return a replay callable."""
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    # thread_local: helper threads of the process (e.g. the RCCL watchdog at N > 1) must not invalidate the capture
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        fn()
    torch.cuda.synchronize()
    return g.replay


def time_replays(replay, reps, launches):
    """Median us per launch over `reps` event-timed replays of a graph holding `launches` kernels."""
    out = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        replay()
        b.record()
        b.synchronize()
        out.append(a.elapsed_time(b) * 1e3 / launches)
    return statistics.median(out), out


def committed_traffic(prefix: str, profiles_dir: str | None = None, repo: str = REPO):
    """Per-launch HBM bytes of a kernel from the committed PMC profile (profiles/rNN_traffic*.json, produced by
    tools/pmc_traffic.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this very script): the
    latest round's file, and within a round the one marked "final" if there is one.  Returns (bytes, source, stale, reason):
    the profile records the SHA-256 of the sources it was taken from, and a figure whose kernel has been edited since
    (or whose profile predates the digests) is kept but marked stale (tools/source_digest.py)."""
    import glob
    import re

    if os.path.join(REPO, "tools") not in sys.path:
        sys.path.insert(0, os.path.join(REPO, "tools"))
    import source_digest

    def order(path):
        name = os.path.basename(path)
        m = re.match(r"r(\d+)_", name)
        return (int(m.group(1)) if m else -1, "final" in name, name)

    files = sorted(glob.glob(os.path.join(profiles_dir or os.path.join(repo, "profiles"), "r*_traffic*.json")), key=order)
    if not files:
        return None, None, None, None
    data = json.load(open(files[-1]))
    for name, rec in data.items():
        if name.startswith(prefix):
            is_stale, why = source_digest.stale(data.get("_source_sha256"), name, repo)
            return rec["traffic_bytes"], f"{os.path.relpath(files[-1], repo)}:{name}", is_stale, why
    return None, None, None, None


def latest_profile(suffix: str) -> str:
    """The newest round's committed profiles/rNN_<suffix> (by round number), as a path relative to the repository."""
    import glob
    import re

    files = glob.glob(os.path.join(REPO, "profiles", f"r*_{suffix}"))
    if not files:
        return f"profiles/rNN_{suffix} (none committed)"
    best = max(files, key=lambda f: int(re.match(r"r(\d+)_", os.path.basename(f)).group(1)))
    return os.path.relpath(best, REPO)


def device_info(dev):
    """What the box says it is (record the device next to the roofline it is priced against)."""
    try:
        p = torch.cuda.get_device_properties(dev)
        return {"name": p.name, "arch": getattr(p, "gcnArchName", ""), "compute_units": p.multi_processor_count,
                "hbm_gib": round(p.total_memory / 2**30, 1), "torch": torch.__version__, "hip": torch.version.hip}
    except Exception as exc:  # never let a property lookup take the line down
        return {"error": repr(exc)[:120]}


def usable_cores() -> int:
    """Host cores this process may really use: affinity mask, capped by the cgroup CPU quota if there is one."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("FP4_BENCH_CPU_THREADS", "64"))))


def cpu_baseline(budget_s=12.0):
    """Pure-torch dequant of the same 4096x4096 -> bf16 problem on the host cores."""
    from oracle import torch_cpu

    ncores = usable_cores()
    torch.set_num_threads(ncores)
    g = torch.Generator().manual_seed(0)
    packed = torch.randint(0, 256, (M * K // 2,), dtype=torch.uint8, generator=g)
    absmax = torch.rand(M * K // BLOCKSIZE, generator=g) * 0.1 + 0.01
    table = torch_cpu.code_table()
    times = []
    t_end = time.perf_counter() + budget_s
    for i in range(13):
        t0 = time.perf_counter()
        torch_cpu.dequantize(packed, absmax, M, K, BLOCKSIZE, torch.bfloat16, table)
        dt = time.perf_counter() - t0
        if i >= 3 or time.perf_counter() > t_end:
            times.append(dt)
        if time.perf_counter() > t_end and times:
            break
    med = statistics.median(times)
    x = torch.randn(1, K)
    t0 = time.perf_counter()
    torch_cpu.gemv(x, packed, absmax, M, K, BLOCKSIZE)
    gemv_s = time.perf_counter() - t0
    return {
        "value": round(dequant_bytes(M, K, BLOCKSIZE, 2) / med / 1e9, 3),
        "unit": "GB/s",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": f"{len(times)} timed pure-torch CPU dequants of one 4096x4096 bs64 weight to bf16 (median {med * 1e3:.1f} ms; "
                  f"os.cpu_count()={os.cpu_count()}, usable={ncores}, torch threads={torch.get_num_threads()}); "
                  f"CPU dequant+GEMV once: {gemv_s * 1e3:.1f} ms",
        "ms_per_matrix": round(med * 1e3, 3),
        "gemv_us_per_layer": round(gemv_s * 1e6, 1),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--matrices", type=int, default=64, help="distinct 4096x4096 FP4 weights per step (R)")
    ap.add_argument("--gemv-reps", type=int, default=4, help="GEMV passes over the R weights per step")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--deadline", type=float, default=480.0,
                    help="N > 1 started without torchrun: seconds after which the worker group is killed and the run fails with a reason")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: nothing below this line has touched the GPU yet (importing torch does not)
        sys.exit(launch_workers(args.gpus, sys.argv[1:], deadline_s=args.deadline))
    if args.gpus != world:
        sys.exit(f"--gpus {args.gpus} does not match WORLD_SIZE={world}")
    claim_stdout()  # from here on file descriptor 1 is stderr for everyone but print_result()
    # one rank per GPU; FP4_BENCH_BACKEND=gloo lets the N > 1 code path be rehearsed on a box with fewer GPUs
    backend = os.environ.get("FP4_BENCH_BACKEND", "nccl")
    # FP4_BENCH_FORCE_GROUP=1: take the whole N > 1 code path (process group, group evidence, per-rank gather, K-split / strong-split /
    # C5 legs) with ONE rank - the only way to run that path through real RCCL on a one-GPU box (RCCL refuses two ranks on one device)
    grouped = world > 1 or os.environ.get("FP4_BENCH_FORCE_GROUP") == "1"
    if grouped and "MASTER_ADDR" not in os.environ:  # plain `python bench.py` with the switch: a rendezvous of our own
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sk.getsockname()[1]), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    ndev = torch.cuda.device_count()
    if world > 1 and backend == "nccl" and ndev < world:
        sys.exit(f"--gpus {world} needs {world} visible GPUs for the RCCL backend, found {ndev} "
                 "(FP4_BENCH_BACKEND=gloo rehearses the N > 1 path with ranks sharing devices)")
    local_dev = local % max(1, ndev)
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    global _WATCHDOG
    # under our own parent the rank's deadline sits a little inside the parent's, so that the rank's reason is printed first
    own = float(os.environ["FP4_BENCH_DEADLINE_S"]) - 15.0 if "FP4_BENCH_DEADLINE_S" in os.environ else args.deadline
    _WATCHDOG = Watchdog(max(5.0, own), rank)
    stage("started")
    if grouped:
        import datetime

        import torch.distributed as dist

        # a rendezvous / RCCL initialisation that does not complete raises here with its reason (well inside the parent's deadline)
        # (the same timeout then governs every collective of the group: it must cover the rank-0-only extras the other ranks sit
        #  out in the final barrier - about half a minute - so it is a large share of the deadline, not a few seconds)
        init_timeout = datetime.timedelta(seconds=max(60.0, min(300.0, float(os.environ.get("FP4_BENCH_DEADLINE_S", str(args.deadline))) * 0.6)))
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev, timeout=init_timeout)
            else:
                dist.init_process_group(backend, timeout=init_timeout)
        except Exception as exc:
            print(f"bench.py: rank {rank}: process-group initialisation ({backend}) failed: {type(exc).__name__}: {exc}", file=sys.stderr, flush=True)
            raise
    proof = None
    if grouped:
        try:
            proof = group_proof(dist, backend, rank, world, local, dev)
        except Exception as exc:  # (an error every rank meets alike must not cost the headline; a one-rank failure ends in the time-outs)
            proof = {"ok": False, "error": repr(exc)[:300]}
    stage("process-group-ready")

    lib = Lib()
    R, GR = args.matrices, args.gemv_reps
    n = M * K
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    packed = [torch.randint(0, 256, (n // 2,), dtype=torch.uint8, device=dev, generator=gen) for _ in range(R)]
    absmax = [torch.rand(n // BLOCKSIZE, device=dev, generator=gen) * 0.1 + 0.01 for _ in range(R)]
    outs = [torch.empty(n, dtype=torch.bfloat16, device=dev) for _ in range(R)]
    x = torch.randn(K, device=dev, generator=gen).to(torch.bfloat16)
    ys = [torch.empty(M, dtype=torch.bfloat16, device=dev) for _ in range(R)]

    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        def dq_pass():
            for i in range(R):
                lib.dequant(packed[i], absmax[i], outs[i], n)

        def gv_pass():
            for _ in range(GR):
                for i in range(R):
                    lib.gemv(x, packed[i], absmax[i], ys[i], M, K)

        dq_replay = capture(dq_pass)
        gv_replay = capture(gv_pass)
        stage("buffers-ready")

        def barrier():
            if grouped:
                dist.barrier()
            torch.cuda.synchronize()

        for _ in range(args.warmup):
            dq_replay()
            gv_replay()
        barrier()
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
        t0 = time.perf_counter()
        for s in range(args.steps):
            ev[s][0].record()
            dq_replay()
            ev[s][1].record()
            gv_replay()
            ev[s][2].record()
        barrier()
        wall = time.perf_counter() - t0
        dq_ms = [e[0].elapsed_time(e[1]) for e in ev]
        gv_ms = [e[1].elapsed_time(e[2]) for e in ev]
        dq_total_s, gv_total_s = sum(dq_ms) / 1e3, sum(gv_ms) / 1e3

        times = torch.tensor([wall, dq_total_s, gv_total_s], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        per_rank = None
        if grouped:
            mine = torch.tensor([dq_total_s, gv_total_s], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            got = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(got, mine)  # every rank's own event-timed totals: stragglers show as a spread, not only as the MAX
            per_rank = [g.tolist() for g in got]
            dist.all_reduce(times, op=dist.ReduceOp.MAX)
        wall, dq_total_s, gv_total_s = times.tolist()
        line = None
        if rank == 0:
            dq_b, gv_b = dequant_bytes(M, K, BLOCKSIZE, 2), gemv_bytes(M, K, BLOCKSIZE, 2)
            dq_launches, gv_launches = args.steps * R, args.steps * R * GR
            dq_us, gv_us = dq_total_s * 1e6 / dq_launches, gv_total_s * 1e6 / gv_launches
            dq_gbps_gpu = dq_b / dq_us / 1e3
            gv_gbps_gpu = gv_b / gv_us / 1e3
            line = {
                "metric": "FP4 dequant GB/s (% HBM peak) + fused-GEMV us/layer, 4096x4096 bf16",
                "value": round(dq_gbps_gpu * world, 1),
                "unit": "GB/s",
                "n_gpus": world,
                "steps": args.steps,
                "warmup": args.warmup,
                "ms_per_step": round(wall * 1e3 / args.steps, 4),
                "higher_is_better": True,
                "scaling": "weak",
                "vs_baseline": None,
                "dtype": "bf16",
                "data": "synthetic",
                "config": {
                    "workload": f"FP4 4096x4096 blocksize 64 -> bf16: per step {R} distinct weights, each dequantised once and "
                                f"GEMV'd {GR}x (batch 1), HBM-cold rotation over {R * (dq_b) / 1e9:.2f} GB, HIP-graph replay",
                    "M": M, "K": K, "blocksize": BLOCKSIZE, "matrices_per_step": R, "gemv_passes_per_step": GR,
                    "parallelism": f"independent row shards x{world}, no collective",
                },
                "device": device_info(dev),
            "pct_hbm_peak": round(100 * dq_gbps_gpu / HBM_PEAK_GBPS, 2),
                "dequant_us_per_matrix": round(dq_us, 3),
                "gemv_us_per_layer": round(gv_us, 3),
                "gemv_gbps": round(gv_gbps_gpu * world, 1),
                "gemv_pct_hbm_peak": round(100 * gv_gbps_gpu / HBM_PEAK_GBPS, 2),
                "roofline": {
                    "bound": "hbm",
                    "kernel": "dequant_tiles_kernel<bf16> (fp4_hip_dequantize_blockwise)",
                    "achieved": round(dq_gbps_gpu, 1),
                    "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s",
                    "frac": round(dq_gbps_gpu / HBM_PEAK_GBPS, 4),
                    "traffic": None,
                    "bytes_per_launch": dq_b,
                    "avg_launch_us": round(dq_us, 3),
                    "method": "HIP events around each graph replay of R back-to-back launches on the launch stream; includes one kernel "
                              f"boundary per launch; rocprofv3 trace of this command: {latest_profile('bench_kernel_trace_summary.json')} (roofline_rows)",
                },
                "roofline_gemv": {
                    "bound": "hbm", "kernel": "gemv16_regx_kernel<bf16> (fp4_hip_gemv)", "achieved": round(gv_gbps_gpu, 1),
                    "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(gv_gbps_gpu / HBM_PEAK_GBPS, 4), "traffic": None,
                    "bytes_per_launch": gv_b, "avg_launch_us": round(gv_us, 3),
                },
            }
            for key, prefix in (("roofline", "dequant_tiles_kernel<2,"), ("roofline_gemv", "gemv16_regx_kernel<2,")):
                traffic, src, is_stale, why = committed_traffic(prefix)
                line[key]["traffic"] = traffic
                line[key]["traffic_source"] = src
                if traffic is not None:
                    line[key]["traffic_stale"] = bool(is_stale)
                    if is_stale:
                        line[key]["traffic_stale_reason"] = why
            def spread(ms, launches):  # median and inter-quartile range over the timed steps, us per launch (rank 0)
                us = sorted(v * 1e3 / launches for v in ms)
                q = statistics.quantiles(us, n=4) if len(us) >= 4 else [us[0], us[len(us) // 2], us[-1]]
                return {"median_us": round(q[1], 3), "iqr_us": round(q[2] - q[0], 3), "min_us": round(us[0], 3), "max_us": round(us[-1], 3)}

            if grouped:
                # row e evidence: who took part (real collectives on the data-path backend) and how evenly the ranks ran
                line["group"] = proof
                dq_rates = [dq_b * dq_launches / t[0] / 1e9 for t in per_rank]
                gv_us_r = [t[1] * 1e6 / gv_launches for t in per_rank]
                line["per_rank"] = {
                    "dequant_gbps": [round(v, 1) for v in dq_rates], "dequant_gbps_min": round(min(dq_rates), 1),
                    "dequant_gbps_max": round(max(dq_rates), 1), "dequant_gbps_sum": round(sum(dq_rates), 1),
                    "gemv_us": [round(v, 3) for v in gv_us_r], "gemv_us_min": round(min(gv_us_r), 3), "gemv_us_max": round(max(gv_us_r), 3),
                    "note": "each rank's own HIP-event totals over the timed steps; `value` is world x the SLOWEST rank's rate",
                }
            line["dequant_step_spread"] = spread(dq_ms, R)
            line["gemv_step_spread"] = spread(gv_ms, R * GR)
            _WATCHDOG.provisional = line
        stage("timed-region-done")

        # ---- secondary figures (outside the timed region) ------------------------------------
        extra = {}
        if grouped:
            try:
                extra["tp_ksplit"] = tp_ksplit_leg(lib, dist, backend, rank, world, dev, x, packed[0], absmax[0], barrier)
            except Exception as exc:  # never let the optional leg take the headline line down
                extra["tp_ksplit"] = {"error": repr(exc)[:300]}
            try:
                extra["strong_scaling_row_split"] = strong_split_leg(lib, dist, backend, rank, world, dev, x, packed, absmax, barrier)
            except Exception as exc:
                extra["strong_scaling_row_split"] = {"error": repr(exc)[:300]}
            if os.environ.get("FP4_BENCH_C5", "1") == "1":
                try:
                    extra["c5_llama3_8b_tp"] = c5_leg(dist, backend, rank, world, dev, barrier)
                except Exception as exc:
                    extra["c5_llama3_8b_tp"] = {"error": repr(exc)[:300]}
        if rank == 0:
            hot_dq = capture(lambda: [lib.dequant(packed[0], absmax[0], outs[0], n) for _ in range(32)])
            hot_gv = capture(lambda: [lib.gemv(x, packed[0], absmax[0], ys[0], M, K) for _ in range(128)])
            extra["dequant_hot_us"] = round(time_replays(hot_dq, 10, 32)[0], 3)
            extra["gemv_hot_us"] = round(time_replays(hot_gv, 10, 128)[0], 3)
            # comparator for the "fused GEMV faster than dequant + hipBLASLt GEMV" target, same inputs
            x2 = x.view(1, K)
            def dq_gemm():
                for i in range(R):
                    lib.dequant(packed[i], absmax[i], outs[i], n, flags=1)  # KEEP_CACHED: what the batch>1 path uses
                    torch.nn.functional.linear(x2, outs[i].view(M, K))
            extra["dequant_plus_hipblaslt_gemv_us"] = round(time_replays(capture(dq_gemm), 5, R)[0], 3)
            def gemm_only():
                for i in range(R):
                    torch.nn.functional.linear(x2, outs[i].view(M, K))
            extra["hipblaslt_gemv_bf16_dense_us"] = round(time_replays(capture(gemm_only), 5, R)[0], 3)
            # other dtypes / table, HBM-cold, for the record
            for name, dt, tdt in (("f16", F16, torch.float16), ("f32", F32, torch.float32)):
                o32 = [torch.empty(n, dtype=tdt, device=dev) for _ in range(16)]
                rp = capture(lambda: [lib.dequant(packed[i], absmax[i], o32[i % 16], n, dt) for i in range(R)])
                us = time_replays(rp, 5, R)[0]
                extra[f"dequant_{name}_us"] = round(us, 3)
                extra[f"dequant_{name}_gbps"] = round(dequant_bytes(M, K, BLOCKSIZE, o32[0].element_size()) / us / 1e3, 1)
                del o32, rp
            # the rows SURVEY 8(f) marks "next", HBM-cold like the headline: fused small batch (2..16 activation rows on one
            # matrix-core tile, 32 / 64 rows in one pass over the weight, 128 rows as two such passes: same weight traffic as the
            # GEMV per pass) and the quantiser (reads the bf16 weights the dequant above just wrote)
            sb = {}
            for b in (2, 4, 8, 16, 32, 64, 128):
                xb = torch.randn(b, K, device=dev).to(torch.bfloat16)
                yb = torch.empty(b, M, dtype=torch.bfloat16, device=dev)
                rp = capture(lambda: [lib.gemm_small(xb, packed[i], absmax[i], yb, b, M, K) for i in range(R)])
                sb[str(b)] = round(time_replays(rp, 5, R)[0], 3)
                del rp
            extra["small_batch_us_by_rows"] = sb
            # the reference's path for the same rows: dequantise to bf16, then the dense GEMM (hipBLASLt)
            x64 = torch.randn(64, K, device=dev).to(torch.bfloat16)
            def dq_gemm64():
                for i in range(R):
                    lib.dequant(packed[i], absmax[i], outs[i], n)
                    torch.nn.functional.linear(x64, outs[i].view(M, K))
            extra["dequant_plus_hipblaslt_gemm_64_rows_us"] = round(time_replays(capture(dq_gemm64), 5, R)[0], 3)
            qp, qa = torch.empty(n // 2, dtype=torch.uint8, device=dev), torch.empty(n // BLOCKSIZE, dtype=torch.float32, device=dev)
            rp = capture(lambda: [lib.quantize(outs[i % len(outs)], qp, qa, n) for i in range(R)])
            us = time_replays(rp, 5, R)[0]
            extra["quantize_bf16_us"] = round(us, 3)
            extra["quantize_bf16_gbps"] = round(dequant_bytes(M, K, BLOCKSIZE, 2) / us / 1e3, 1)
            del rp, qp, qa
            # launch-overhead-free context: the same kernels on one tall stack of R weights ([R*4096, 4096]) per launch
            # (input buffers are re-used as views; nothing new is read from the host).  Four launches per graph replay, so
            # that the ~10-20 us a replay itself costs is spread thin (a single ~100 us launch per replay reads 20 % low).
            big_p, big_a = torch.cat(packed), torch.cat(absmax)
            big_o = torch.empty(R * n, dtype=torch.bfloat16, device=dev)
            big_y = torch.empty(R * M, dtype=torch.bfloat16, device=dev)
            # Each figure is taken after ~25 ms of the same kernel: the first replays after a change of kernel run at clocks
            # that have not settled and read 5-15 % low (tools/exp_stack_measure.py, profiles/r02_stack_measurement_method.txt).
            rp = capture(lambda: [lib.dequant(big_p, big_a, big_o, R * n) for _ in range(4)])
            time_replays(rp, 15, 4)
            us = time_replays(rp, 9, 4)[0]
            extra["dequant_stack_of_R_one_launch_gbps"] = round(R * dequant_bytes(M, K, BLOCKSIZE, 2) / us / 1e3, 1)
            # the quantiser over the same stack (reads the bf16 values the stacked dequant just wrote; outputs go over the inputs' buffers)
            # (FP4_BENCH_QUANT_STACK=0, set by the PMC passes of tools/collect_profiles.sh: the quantiser's grid is persistent - the
            #  same for every size - so its counters could not be told apart from the per-matrix launches' by grid like the others')
            if os.environ.get("FP4_BENCH_QUANT_STACK", "1") == "1":
                big_qp, big_qa = torch.empty(R * n // 2, dtype=torch.uint8, device=dev), torch.empty(R * n // BLOCKSIZE, dtype=torch.float32, device=dev)
                rp = capture(lambda: [lib.quantize(big_o, big_qp, big_qa, R * n) for _ in range(4)])
                time_replays(rp, 15, 4)
                us = time_replays(rp, 9, 4)[0]
                extra["quantize_stack_of_R_one_launch_gbps"] = round(R * dequant_bytes(M, K, BLOCKSIZE, 2) / us / 1e3, 1)
                del big_qp, big_qa
            rp = capture(lambda: [lib.gemv(x, big_p, big_a, big_y, R * M, K) for _ in range(12)])
            time_replays(rp, 20, 12)
            us = time_replays(rp, 9, 12)[0]
            extra["gemv_stack_of_R_one_launch_gbps"] = round(gemv_bytes(R * M, K, BLOCKSIZE, 2) / us / 1e3, 1)
            try:  # the bare streams of the same geometry over the same stack (steady state of the box, same run)
                extra["box_stream_stack_of_R_gbps"] = box_stream_stack_leg(big_p, big_o)
            except Exception as exc:
                extra["box_stream_stack_of_R_gbps"] = {"error": repr(exc)[:200]}
            del big_p, big_a, big_o, big_y
            # end-to-end through the Python op surface (host overhead visible, like the reference's README table)
            code = lib.pkg.ext.code_table("tree").to(dev)
            B = packed[0].view(-1, 1).t()
            for _ in range(20):
                lib.pkg.gemm_4bit_inference(x2, B, absmax[0], code, BLOCKSIZE, torch.bfloat16, [M, K])
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(200):
                lib.pkg.gemm_4bit_inference(x2, B, absmax[0], code, BLOCKSIZE, torch.bfloat16, [M, K])
            torch.cuda.synchronize()
            extra["gemv_python_op_us"] = round((time.perf_counter() - t1) / 200 * 1e6, 2)
            # device copy rate of this box, the practical HBM ceiling
            src = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
            dst = torch.empty_like(src)
            cp = capture(lambda: dst.copy_(src))
            us = time_replays(cp, 5, 1)[0]
            extra["device_copy_gbps"] = round(2 * src.numel() / us / 1e3, 1)
            del src, dst
            # ... and what it streams with the dequant kernel's own access geometry (the ceiling that means something for this kernel)
            try:
                extra["box_stream_gbps"] = box_stream_leg(packed, outs)
            except Exception as exc:
                extra["box_stream_gbps"] = {"error": repr(exc)[:200]}
            # the fused decode epilogues at the layer shapes they serve (HBM-cold like the headline): h + o(a), silu(g)*u over the
            # interleaved gate|up weight, h + down(act); same weight traffic as the plain GEMV of that shape
            try:
                extra["fused_epilogue_us"] = fused_epilogue_leg(lib, dev)
            except Exception as exc:
                extra["fused_epilogue_us"] = {"error": repr(exc)[:200]}
            # BASELINE config 3 (driver-visible): the reference's own sanity-MLP speed table on this build, with its README figures beside it
            if world == 1 and os.environ.get("FP4_BENCH_C3", "1") == "1":
                try:
                    extra["c3_sanity_mlp"] = c3_leg(lib.pkg)
                except Exception as exc:
                    extra["c3_sanity_mlp"] = {"error": repr(exc)[:300]}
                torch.cuda.empty_cache()
            # BASELINE config 4 (driver-visible): batch-1 decode through every FP4 Linear of a Mistral-7B shaped model
            if world == 1 and os.environ.get("FP4_BENCH_C4", "1") == "1":
                try:
                    extra["c4_mistral7b_decode"] = c4_leg(dev, fused_us=extra.get("fused_epilogue_us"))
                except Exception as exc:
                    extra["c4_mistral7b_decode"] = {"error": repr(exc)[:300]}

    if rank == 0:
        line.update(extra)
        # the same kernels away from the launch boundary (one launch over a stack of R weights): what the kernel itself streams at
        if "dequant_stack_of_R_one_launch_gbps" in extra:
            line["roofline"]["steady_state_gbps"] = extra["dequant_stack_of_R_one_launch_gbps"]
            line["roofline"]["steady_state_frac"] = round(extra["dequant_stack_of_R_one_launch_gbps"] / HBM_PEAK_GBPS, 4)
            line["roofline_gemv"]["steady_state_gbps"] = extra["gemv_stack_of_R_one_launch_gbps"]
            line["roofline_gemv"]["steady_state_frac"] = round(extra["gemv_stack_of_R_one_launch_gbps"] / HBM_PEAK_GBPS, 4)
        if "quantize_bf16_us" in extra:
            # row f1's kernel (load-time, not on the decode path) priced like the other two: 33.6 MB read, 9.4 MB written per launch
            q_b = dequant_bytes(M, K, BLOCKSIZE, 2)
            q_gbps = q_b / extra["quantize_bf16_us"] / 1e3
            line["roofline_quantize"] = {
                "bound": "hbm", "kernel": "quantize_kernel<bf16> (fp4_hip_quantize_blockwise)", "achieved": round(q_gbps, 1), "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": round(q_gbps / HBM_PEAK_GBPS, 4), "traffic": None, "bytes_per_launch": q_b,
                "avg_launch_us": extra["quantize_bf16_us"],
                "method": "HIP events around graph replays of R back-to-back launches, HBM-cold rotation over the R bf16 weights (outside the timed region)",
            }
            traffic, src, is_stale, why = committed_traffic("quantize_")
            if traffic is not None:
                line["roofline_quantize"].update({"traffic": traffic, "traffic_source": src, "traffic_stale": bool(is_stale)})
                if is_stale:
                    line["roofline_quantize"]["traffic_stale_reason"] = why
            if "quantize_stack_of_R_one_launch_gbps" in extra:
                line["roofline_quantize"]["steady_state_gbps"] = extra["quantize_stack_of_R_one_launch_gbps"]
                line["roofline_quantize"]["steady_state_frac"] = round(extra["quantize_stack_of_R_one_launch_gbps"] / HBM_PEAK_GBPS, 4)
        bs_, st_ = extra.get("box_stream_gbps", {}), extra.get("box_stream_stack_of_R_gbps", {})
        if "write_only" in bs_:
            rf = line["roofline"]
            if "read_only_gemv_bytes_us" in bs_:  # same run: a bare read of the GEMV's bytes per launch vs the GEMV
                bs_ = dict(bs_)
                bare = bs_.pop("read_only_gemv_bytes_us")
                line["roofline_gemv"]["box_bare_read_same_bytes_us"] = bare
                line["roofline_gemv"]["frac_of_box_bare_read_per_launch"] = round(bare / line["roofline_gemv"]["avg_launch_us"], 4)
            rf["box_stream_gbps"] = bs_
            if "write_only" in st_:
                rf["box_stream_stack_of_R_gbps"] = st_
                if "steady_state_gbps" in line["roofline_gemv"]:
                    line["roofline_gemv"]["steady_state_frac_of_box_read_stream"] = round(line["roofline_gemv"]["steady_state_gbps"] / st_["read_only"], 4)
            box_yardsticks(rf, M * K // 2 + 4 * (M * K // BLOCKSIZE), M * K * 2, bs_, st_)
            if "roofline_quantize" in line:  # reads the bf16 weight, writes the packed bytes + scales
                box_yardsticks(line["roofline_quantize"], M * K * 2, M * K // 2 + 4 * (M * K // BLOCKSIZE), bs_, st_)
            rf["box_stream_note"] = ("tools/stream_probe.hip: the dequant kernel's access geometry without arithmetic, per 32 MiB launch and over the stack of R; "
                                     "box_serial_rw_estimate_gbps = bytes / (written / write_only + read / read_only); `best_of_both` (better bare rate per "
                                     "direction) is the one meant as a ceiling; `frac` stays against the 8 TB/s spec (DESIGN.md section 3)")
        if "dequant_plus_hipblaslt_gemv_us" in extra:
            line["fused_gemv_speedup_vs_dequant_hipblaslt"] = round(extra["dequant_plus_hipblaslt_gemv_us"] / gv_us, 2)
        if world == 1 and not args.no_cpu:
            line["cpu_baseline"] = cpu_baseline()
        _WATCHDOG.provisional = None
        print_result(line)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()
    stage("done")
    _WATCHDOG.cancel()


if __name__ == "__main__":
    main()
