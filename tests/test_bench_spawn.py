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


def test_a_worker_group_that_outlives_the_deadline_is_killed_with_a_reason(tmp_path, capfd):
    """A rank that never returns (a stuck RCCL initialisation on the 8-GPU box, say) must end in a non-zero exit inside the
    deadline with the stage each rank had reached on stderr - not in a silent hang that the driver has to kill."""
    import time

    import bench

    sleeper = _script(tmp_path, """
        import os, sys, time
        rank = os.environ["RANK"]
        print(f"@@fp4-bench rank={rank} stage=started", flush=True)
        if rank == "0":
            print(f"@@fp4-bench rank={rank} stage=process-group-ready", flush=True)
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


def test_stage_markers_never_reach_stdout_without_our_parent(tmp_path):
    """The driver starts N > 1 under its own torchrun and expects ONE JSON line on stdout: the progress markers are only written
    when launch_workers (which swallows them) asked for them."""
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
    env.pop("FP4_BENCH_STAGES", None)
    p = subprocess.run([sys.executable, worker], capture_output=True, text=True, timeout=100, env=env)
    assert p.returncode == 0 and p.stdout.strip() == "only line"
    p = subprocess.run([sys.executable, worker], capture_output=True, text=True, timeout=100, env=dict(env, FP4_BENCH_STAGES="1"))
    assert "@@fp4-bench rank=1 stage=started" in p.stdout
