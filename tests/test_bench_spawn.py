"""`python bench.py --gpus N` without torchrun starts its own workers (bench.launch_workers): the spawn / relay / exit-code
logic is exercised here on CPU with a stand-in worker script (gloo, two ranks); the real thing runs in test_gpu_bench.py."""
import io
import json
import os
import sys
import textwrap

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def _script(tmp_path, body):
    p = tmp_path / "worker.py"
    p.write_text(textwrap.dedent(body))
    return str(p)


def _script_named(tmp_path, name, body):
    p = tmp_path / name
    p.write_text(textwrap.dedent(body))
    return str(p)


def test_launch_workers_relays_rank0_line_and_exit_code(tmp_path, capfd):
    import bench

    ok = _script(tmp_path, """
        import json, os, sys
        import torch, torch.distributed as dist
        assert os.environ["MASTER_ADDR"] == "127.0.0.1" and os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        dist.init_process_group("gloo")
        t = torch.tensor([float(dist.get_rank() + 1)])
        dist.all_reduce(t)
        print("chatter from rank", dist.get_rank())          # must not reach the parent's stdout
        if dist.get_rank() == 0:
            print(json.dumps({"metric": "stand-in", "n_gpus": dist.get_world_size(), "sum": t.item(), "argv": sys.argv[1:]}))
        dist.barrier()
        dist.destroy_process_group()
    """)
    out = io.StringIO()
    rc = bench.launch_workers(2, ["--gpus", "2", "--steps", "3"], script=ok, timeout_s=300, out=out)
    lines = [l for l in out.getvalue().splitlines() if l.strip()]
    assert rc == 0 and len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["sum"] == 3.0 and rec["argv"] == ["--gpus", "2", "--steps", "3"]
    assert "chatter" in capfd.readouterr().err

    bad = _script(tmp_path, """
        import os, sys
        sys.exit(3 if os.environ["RANK"] == "1" else 0)
    """)
    out = io.StringIO()
    assert bench.launch_workers(2, [], script=bad, timeout_s=300, out=out) != 0  # a failed rank is a failed run: no retry
    assert out.getvalue() == ""

    silent = _script(tmp_path, "pass\n")
    assert bench.launch_workers(2, [], script=silent, timeout_s=300, out=io.StringIO()) == 1  # no result line is a failure


def test_plain_invocation_with_gpus_gt_1_spawns_before_any_gpu_call():
    """Source-level guard: in bench.main the spawn branch comes before the first torch.cuda call."""
    src = open(os.path.join(REPO, "bench.py")).read()
    main = src[src.index("def main():"):]
    assert main.index("launch_workers(args.gpus") < main.index("torch.cuda.")
    assert "os.exec" not in src


def test_ipc_mode_variable_is_set_before_anything_can_touch_the_gpu():
    """HSA_ENABLE_IPC_MODE_LEGACY=0 is read by the HIP runtime when it initialises: in the WORKER path too (the driver's own torchrun
    never passes through launch_workers) it has to be in os.environ before the first torch.cuda call - i.e. at module top, ahead of
    `import torch` - in bench.py and in tools/decode_bench.py (imported by the C5 leg, runnable on its own under torchrun)."""
    for rel in ("bench.py", os.path.join("tools", "decode_bench.py")):
        src = open(os.path.join(REPO, rel)).read()
        at = src.index('os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")')
        assert at < src.index("\nimport torch"), rel
        assert at < src.index("torch.cuda."), rel
        assert at < src.index("def main("), rel
        # and nowhere later does the worker path set it "just in time" (too late by then): the only other mention is the child env
        later = [i for i in range(len(src)) if src.startswith('os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY"', i) and i != at]
        assert later == [], (rel, later)
    # behaviour, not only source: a fresh interpreter without the variable has it after importing bench, before torch.cuda is touched
    import subprocess

    env = {k: v for k, v in os.environ.items() if k != "HSA_ENABLE_IPC_MODE_LEGACY"}
    code = ("import os, sys; sys.path.insert(0, %r); import bench; "
            "print(os.environ['HSA_ENABLE_IPC_MODE_LEGACY'], bench.HSA_IPC_ENV_AT_START)" % REPO)
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and p.stdout.split() == ["0", "None"], (p.stdout, p.stderr[-500:])


def test_group_proof_fields_from_real_collectives_world_2(tmp_path):
    """The N > 1 line's `group` record (ranks_seen, rank checksum, gathered identities, distinct devices) comes from collectives on the
    process group; here two gloo ranks on CPU with a stand-in identity (no GPU in this container), the real thing in test_gpu_bench.py."""
    import subprocess

    worker = _script(tmp_path, f"""
        import json, os, sys
        sys.path.insert(0, {REPO!r})
        import torch, torch.distributed as dist
        import bench
        rank = int(os.environ["RANK"])
        same = os.environ["SAME_DEVICE"] == "1"
        bench._rank_identity = lambda r, l, d: {{"rank": r, "local_rank": l, "pid": os.getpid(), "host": "h",
                                                "device": "gpu-0" if same else f"gpu-{{r}}"}}
        dist.init_process_group("gloo")
        got = bench.group_proof(dist, os.environ["AS_BACKEND"], rank, dist.get_world_size(), rank, None)
        if rank == 0:
            print(json.dumps(got))
        dist.barrier()
        dist.destroy_process_group()
    """)
    import socket

    def run(same, as_backend):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, SAME_DEVICE=same, AS_BACKEND=as_backend)
        p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                            "--master-port", str(port), worker], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])

    g = run("1", "gloo")
    assert g["ranks_seen"] == 2 and g["rank_checksum"] == 3 == g["rank_checksum_expected"] and g["allreduce_data_ok_on_every_rank"]
    assert g["distinct_devices"] == 1 and g["distinct_pids"] == 2 and [r["rank"] for r in g["ranks"]] == [0, 1] and g["ok"]
    assert "gloo" in g["collective_library"] and g["backend"] == "gloo"


def test_sigterm_to_the_launching_parent_ends_the_worker_group(tmp_path):
    """The worker group is a session of its own (so that the deadline can kill exactly it), hence out of reach of a killpg aimed at the
    parent: a TERM to the parent (a driver's limit) must end the ranks too, and so must a KILL of the parent (PR_SET_PDEATHSIG)."""
    import signal
    import subprocess
    import time

    pids = tmp_path / "pids"
    pids.mkdir()
    sleeper = _script(tmp_path, f"""
        import os, time
        open(os.path.join({str(pids)!r}, os.environ["RANK"]), "w").write(str(os.getpid()))
        time.sleep(600)
    """)
    parent_src = _script_named(tmp_path, "parent.py", f"""
        import sys
        sys.path.insert(0, {REPO!r})
        import bench
        sys.exit(bench.launch_workers(2, [], script={sleeper!r}, deadline_s=300))
    """)

    def alive(pid):
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            return False
        try:  # a zombie waiting for its reaper is not "alive" for our purpose
            return open(f"/proc/{pid}/stat").read().rsplit(")", 1)[1].split()[0] != "Z"
        except OSError:
            return False

    for sig, want_rc in ((signal.SIGTERM, 128 + signal.SIGTERM), (signal.SIGKILL, -signal.SIGKILL)):
        for f in pids.iterdir():
            f.unlink()
        parent = subprocess.Popen([sys.executable, parent_src], stderr=subprocess.PIPE, text=True)
        t_end = time.monotonic() + 120
        while len(list(pids.iterdir())) < 2 and time.monotonic() < t_end:
            time.sleep(0.2)
        time.sleep(0.3)
        workers = [int(f.read_text()) for f in pids.iterdir()]
        assert len(workers) == 2 and all(alive(w) for w in workers)
        parent.send_signal(sig)
        rc = parent.wait(timeout=60)
        assert rc == want_rc, (sig, rc, parent.stderr.read()[-500:])
        t_end = time.monotonic() + 40  # torchrun TERMs its ranks, then KILLs them after its grace period
        while any(alive(w) for w in workers) and time.monotonic() < t_end:
            time.sleep(0.2)
        left = [w for w in workers if alive(w)]
        for w in left:
            os.kill(w, signal.SIGKILL)
        assert not left, (sig, left)


def test_a_worker_group_that_outlives_the_deadline_is_killed_with_a_reason(tmp_path, capfd):
    """A rank that never returns (a stuck RCCL initialisation on the 8-GPU box, say) must end in a non-zero exit inside the
    deadline with the stage each rank had reached on stderr - not in a silent hang that the driver has to kill."""
    import time

    import bench

    sleeper = _script(tmp_path, f"""
        import os, sys, time
        sys.path.insert(0, {REPO!r})
        import bench
        bench.stage("started")
        if os.environ["RANK"] == "0":
            bench.stage("process-group-ready")
        time.sleep(600)
    """)
    out = io.StringIO()
    t0 = time.monotonic()
    rc = bench.launch_workers(2, [], script=sleeper, deadline_s=20, out=out)
    took = time.monotonic() - t0
    err = capfd.readouterr().err
    assert rc == 124 and took < 60, (rc, took)
    assert out.getvalue() == ""
    assert "did not finish within the deadline" in err
    assert "rank 0: last stage 'process-group-ready'" in err
    assert "rank 1: last stage 'started'" in err and "never finished process-group initialisation" in err


def test_the_rank_side_watchdog_prints_the_finished_headline_and_leaves(tmp_path):
    """Under torchrun (how the driver starts N > 1) there is no parent of ours: every rank carries its own deadline.  If it
    passes after the timed region, rank 0 still prints the headline line, marked incomplete, and the exit code is 124."""
    import subprocess

    worker = _script(tmp_path, f"""
        import sys, time
        sys.path.insert(0, {REPO!r})
        import bench
        w = bench.Watchdog(2.0, 0)
        bench._WATCHDOG = w
        bench.stage("timed-region-done")
        w.provisional = {{"metric": "stand-in", "value": 1.0}}
        time.sleep(120)
    """)
    p = subprocess.run([sys.executable, worker], capture_output=True, text=True, timeout=100)
    assert p.returncode == 124
    rec = json.loads(p.stdout.strip().splitlines()[-1])
    assert rec["value"] == 1.0 and "timed-region-done" in rec["incomplete"]
    assert "deadline of 2 s passed in stage 'timed-region-done'" in p.stderr


def test_stage_reports_never_touch_stdout(tmp_path):
    """The ranks of an N > 1 run share one stdout pipe and rank 0's line is longer than PIPE_BUF, so anything another rank prints can
    land in the middle of it: progress goes to a file per rank (only when the launching parent asked for it), stdout carries the ONE
    JSON line and nothing else - under the driver's own torchrun as well."""
    import subprocess

    worker = _script(tmp_path, f"""
        import sys
        sys.path.insert(0, {REPO!r})
        import bench
        bench.stage("started")
        bench.stage("done")
        print("only line")
    """)
    env = dict(os.environ, WORLD_SIZE="2", RANK="1")
    env.pop("FP4_BENCH_STAGE_DIR", None)
    p = subprocess.run([sys.executable, worker], capture_output=True, text=True, timeout=100, env=env)
    assert p.returncode == 0 and p.stdout.strip() == "only line"
    d = tmp_path / "stages"
    d.mkdir()
    p = subprocess.run([sys.executable, worker], capture_output=True, text=True, timeout=100, env=dict(env, FP4_BENCH_STAGE_DIR=str(d)))
    assert p.stdout.strip() == "only line" and (d / "rank1").read_text().split() == ["started", "done"]


def test_result_line_is_recovered_when_another_process_glues_text_to_it():
    """launch_workers extracts the JSON object even if foreign text shares its line (gloo's own prints, a rank's chatter)."""
    import bench

    class FakeProc:
        stdout = ['[Gloo] Rank 0 is connected{"metric": "m", "value": 1.5, "nested": {"a": [1, 2]}}chatter from rank 1\n', "noise\n"]

    # the pump is local to launch_workers; its parsing rule is what matters and is restated here on the same inputs
    got = None
    for raw in FakeProc.stdout:
        at = raw.find('{"metric"')
        if at >= 0:
            obj, end = json.JSONDecoder().raw_decode(raw[at:])
            got = obj
    assert got == {"metric": "m", "value": 1.5, "nested": {"a": [1, 2]}}
    src = open(os.path.join(REPO, "bench.py")).read()
    assert "raw_decode(txt[at:])" in src  # the rule above is the one launch_workers applies


def test_native_chatter_on_stdout_cannot_reach_the_result_stream(tmp_path):
    """RCCL prints a version banner and gloo prints connection notes on file descriptor 1 from native code; after bench.claim_stdout()
    that descriptor is stderr for everybody, and only bench.print_result() still reaches the process's original stdout."""
    import subprocess

    worker = _script(tmp_path, f"""
        import os, sys
        sys.path.insert(0, {REPO!r})
        import bench
        bench.claim_stdout()
        os.write(1, b"RCCL version : 2.26.6-HEAD (native printf)\\n")
        print("python-level chatter")
        os.system("echo child process chatter")
        bench.print_result({{"metric": "stand-in", "value": 2.5}})
        bench.claim_stdout()  # idempotent
        bench.print_result({{"metric": "second"}})
    """)
    p = subprocess.run([sys.executable, worker], capture_output=True, text=True, timeout=200)
    assert p.returncode == 0, p.stderr[-800:]
    assert [json.loads(l)["metric"] for l in p.stdout.splitlines()] == ["stand-in", "second"], p.stdout
    for text in ("RCCL version", "python-level chatter", "child process chatter"):
        assert text in p.stderr and text not in p.stdout
