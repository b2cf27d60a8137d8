"""bench.py end to end on the GPU box: the N = 1 line's schema, and `python bench.py --gpus 2` WITHOUT torchrun (the script
starts its own two workers; FP4_BENCH_BACKEND=gloo lets both ranks share the one GPU of the box, everything but the
transport being the N > 1 production path, including the tensor-parallel C5 leg)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCHEMA = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
          "dtype", "data", "config", "roofline"}


def _run(args, env_extra):
    env = dict(os.environ, **env_extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line_schema():
    rec = _run(["--steps", "2", "--warmup", "1", "--matrices", "8", "--no-cpu"], {"FP4_BENCH_C4_LAYERS": "2"})
    assert SCHEMA <= set(rec) and rec["n_gpus"] == 1 and rec["steps"] == 2 and rec["unit"] == "GB/s" and rec["dtype"] == "bf16"
    r = rec["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert 0.2 < r["frac"] < 1.0 and "workload" in rec["config"] and "model" not in rec["config"]
    # driver-visible secondary legs (outside the timed region): fused epilogues at layer shapes, config C4 in three forms
    fe, c4 = rec["fused_epilogue_us"], rec["c4_mistral7b_decode"]
    assert "error" not in fe and fe["gate_up_silu_mul_28672x4096"] > 0 and "error" not in c4, (fe, c4)
    assert c4["row_concat_plus_epilogues"]["graph_ms_per_token"] < c4["separate_launches"]["graph_ms_per_token"]


def test_two_ranks_self_launched_over_gloo():
    rec = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--matrices", "8"],
               {"FP4_BENCH_BACKEND": "gloo", "FP4_BENCH_C5_LAYERS": "2"})
    assert SCHEMA <= set(rec) and rec["n_gpus"] == 2 and rec["scaling"] == "weak"
    assert "error" not in rec["tp_ksplit"], rec["tp_ksplit"]
    ss = rec["strong_scaling_row_split"]  # one 4096x4096 weight split by rows over the ranks (strong scaling, launch-bound)
    assert "error" not in ss and ss["rows_per_gpu"] == 2048 and ss["dequant_us"] > 0 and ss["gemv_us"] > 0, ss
    c5 = rec["c5_llama3_8b_tp"]
    assert "error" not in c5 and c5["allreduces_per_token"] == 4 and "error" not in c5["dist"], c5
    _check_group_evidence(rec, 2)


def _check_group_evidence(rec, world):
    """the line itself says who took part - counted and summed by real collectives, identities gathered from every rank."""
    g, pr = rec["group"], rec["per_rank"]
    assert g["ranks_seen"] == world and g["rank_checksum"] == g["rank_checksum_expected"] == world * (world + 1) // 2, g
    assert g["allreduce_data_ok_on_every_rank"] is True and g["ok"] is True and g["backend"] == "gloo", g
    assert [r["rank"] for r in g["ranks"]] == list(range(world)) and g["distinct_pids"] == world, g
    assert g["distinct_devices"] == 1, g  # a one-GPU box: the ranks share the device, and the line says so
    assert all(r["hsa_ipc_env"] == "0" and r["arch"].startswith("gfx950") for r in g["ranks"]), g["ranks"]
    assert "RCCL" in g["collective_library"], g
    assert len(pr["dequant_gbps"]) == world and 0 < pr["dequant_gbps_min"] <= pr["dequant_gbps_max"], pr
    assert abs(rec["value"] - world * pr["dequant_gbps_min"]) <= 0.02 * rec["value"], (rec["value"], pr)
    sc = rec["c5_llama3_8b_tp"]["oneshot_selfcheck"]
    assert sc["ok"] is True and sc["hsa_ipc_env"] == "0" and sc["status_word"] == 0, sc


def test_two_ranks_started_the_way_the_driver_starts_them():
    """The command the driver types for N > 1: torchrun from OUTSIDE (bench.launch_workers and the environment it prepares are not
    involved), `--nproc-per-node 2`, 127.0.0.1 rendezvous - and no HSA_* variable preset by the caller: bench.py's own module-top
    default must be in time for the peer mappings of the one-shot all-reduce (the C5 leg's self-check says whether it was)."""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("HSA_ENABLE_IPC_MODE_LEGACY", "WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(FP4_BENCH_BACKEND="gloo", FP4_BENCH_C5_LAYERS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--matrices", "8"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=REPO)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip() and not l.startswith("[Gloo]")]
    assert len(lines) == 1 and lines[0].startswith('{"metric"'), p.stdout[-3000:]
    rec = json.loads(lines[0])
    assert SCHEMA <= set(rec) and rec["n_gpus"] == 2 and "incomplete" not in rec
    assert all(r["hsa_ipc_env_at_start"] is None for r in rec["group"]["ranks"]), rec["group"]["ranks"]  # nothing was preset: the default did it
    _check_group_evidence(rec, 2)
    for leg in ("tp_ksplit", "strong_scaling_row_split"):
        assert "error" not in rec[leg], rec[leg]
    c5 = rec["c5_llama3_8b_tp"]
    assert "error" not in c5["dist"] and "error" not in c5.get("oneshot", {}), c5
