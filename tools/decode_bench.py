#!/usr/bin/env python3
"""Batch-1 decode through every FP4 Linear of a Mistral-7B / Llama-3-8B shaped model (BASELINE configs 4 and 5).

Synthetic weights of the real shapes (no checkpoints offline): per decoder layer q,o 4096x4096, k,v 1024x4096,
gate,up 14336x4096, down 4096x14336, all FP4 blocksize 64; lm_head stays a dense bf16 Linear, as the reference's
default `ignore_layer_names=["lm_head"]` leaves it (torch_bnb_fp4/__init__.py:788).  One "token" = the dependent chain
of the 7 x L fused GEMVs (+ the elementwise glue between them and the lm_head GEMV), run through the package's
QuantData / tensor-parallel modules, eagerly and replayed from a HIP graph.  Attention itself is not part of this path
and is replaced by an identity on q (the GEMV traffic is what is being measured).

    python tools/decode_bench.py [--model mistral7b|llama3-8b] [--layers 32] [--tokens 64] [--fuse] [--epilogues]
    torchrun --nproc-per-node N tools/decode_bench.py --model llama3-8b      # tensor parallel: q/k/v/gate/up M-split,
                                                                              # o/down K-split + all-reduce

`build_token_fn` / `time_tokens` are imported by bench.py for the N > 1 "C5" leg.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# dmabuf IPC only on this pool, and the HIP runtime reads the variable once when it initialises: before anything touches the GPU
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (REPO, os.path.join(REPO, "torch-bnb-fp4_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import torch  # noqa: E402

MODELS = {"mistral7b": dict(hidden=4096, kv=1024, inter=14336, vocab=32000, layers=32),
          "llama3-8b": dict(hidden=4096, kv=1024, inter=14336, vocab=128256, layers=32)}
BS = 64


def fp4_bytes(m, k):
    return m * k // 2 + 4 * (m * k // BS)


def build_token_fn(cfg, dev, dtype, world=1, rank=0, group=None, fuse=False, epilogues=False, batch=1,
                   reference_dispatch=False, allreduce="dist", lm_head=True, seed=7, lean_glue=False, tensor_parallel=None):
    """Builds the FP4 layers of a `cfg`-shaped decoder and returns (token_fn, h0, meta).

    world == 1: QuantData dispatchers (the product's single-GPU path).  world > 1: Column/RowParallelFP4Linear
    (q/k/v/gate/up M-split without a gather, o/down K-split with one f32 all-reduce each: 2 all-reduces per layer).
    fuse: q|k|v and gate|up as one launch each (row concatenation).  epilogues: on top of that, silu(gate)*up and the
    residual adds run in the GEMV epilogue (torch_bnb_fp4.fused).  tensor_parallel: None = (world > 1); True builds the tensor-parallel
    modules even for a one-rank group (bench.py's FP4_BENCH_FORCE_GROUP rehearsal of the N > 1 path through real RCCL on one GPU)."""
    import torch_bnb_fp4 as pkg
    from torch_bnb_fp4 import parallel as par

    H, KV, I, V, L = cfg["hidden"], cfg["kv"], cfg["inter"], cfg["vocab"], cfg["layers"]
    tp = world > 1 if tensor_parallel is None else bool(tensor_parallel)
    gen = torch.Generator(device=dev).manual_seed(seed)  # the same full weights on every rank; the modules shard them

    def fp4_weight(m, k):
        packed = torch.randint(0, 256, (m * k // 2, 1), dtype=torch.uint8, device=dev, generator=gen)
        absmax = (torch.rand(m * k // BS, device=dev, generator=gen) * 0.02 + 0.002) * (0.25 if lean_glue else 1.0)
        return packed, absmax

    code = pkg.ext.code_table("tree").to(dev)

    def qd_of(packed, absmax, m, k):
        state = pkg.QuantState(absmax, (m, k), code, BS)
        return pkg.QuantData(packed, state, state.shape, original_lin=None, bias=None,
                             small_batch_fused=not reference_dispatch)

    def linear(m, k, kind):
        if isinstance(m, (list, tuple)):  # fused rows
            packed, absmax, (m, k) = par.concat_rows([(*fp4_weight(mi, k), (mi, k)) for mi in m], BS)
        else:
            packed, absmax = fp4_weight(m, k)
        if not tp:
            return qd_of(packed, absmax, m, k).forward
        if kind == "col":
            return par.ColumnParallelFP4Linear(packed, absmax, (m, k), BS, group=group, gather_output=False)
        # a forced one-rank group (tensor_parallel=True at world 1) still issues its collectives, so that the rehearsal line's
        # `allreduces_per_token` is what ran and not what would have run
        return par.RowParallelFP4Linear(packed, absmax, (m, k), BS, group=group, input_is_parallel=True, allreduce=allreduce,
                                        reduce_single_rank=(world == 1))

    layers = []
    for _ in range(L):
        if epilogues and not tp:
            from torch_bnb_fp4 import fused

            ly = dict(qkv=linear([H, KV, KV], H, "col"),
                      o=fused.FusedFP4Linear.from_packed(*fp4_weight(H, H), (H, H), BS),
                      gate_up=fused.FusedFP4Linear.gate_up_from_packed(fp4_weight(I, H), fp4_weight(I, H), (I, H), BS),
                      down=fused.FusedFP4Linear.from_packed(*fp4_weight(H, I), (H, I), BS))
        elif fuse and not tp:
            ly = dict(qkv=linear([H, KV, KV], H, "col"), o=linear(H, H, "row"), gate_up=linear([I, I], H, "col"),
                      down=linear(H, I, "row"))
        elif (fuse or epilogues) and tp:
            # tensor parallel with the same fusions: q|k|v shards in one launch, gate|up shards interleaved with silu(g)*u in the
            # epilogue, the residual adds inside the K-split layers (in the one-shot all-reduce's epilogue when that is used)
            w = lambda m, k: (*fp4_weight(m, k), (m, k))
            ly = dict(tp_qkv=par.FusedColumnParallelFP4([w(H, H), w(KV, H), w(KV, H)], BS, group),
                      o=linear(H, H, "row"),
                      tp_gate_up=par.FusedColumnParallelFP4([w(I, H), w(I, H)], BS, group, epilogue="silu_mul"),
                      down=linear(H, I, "row"))
        else:
            ly = dict(q=linear(H, H, "col"), k=linear(KV, H, "col"), v=linear(KV, H, "col"), o=linear(H, H, "row"),
                      gate=linear(I, H, "col"), up=linear(I, H, "col"), down=linear(H, I, "row"))
        layers.append(ly)
    head = torch.nn.Linear(H, V, bias=False, device=dev, dtype=dtype) if lm_head else None
    h0 = torch.randn(batch, H, device=dev, generator=gen).to(dtype)
    silu = torch.nn.functional.silu
    # stand-in for attention: keeps k and v live and dependent.  Default: five small launches (two sums, add, scale, add), about what
    # RoPE + cache append + attention cost a real decoder layer in launches; lean_glue: ONE elementwise launch, and no rescale of h
    # (the weights' scales keep magnitudes bounded instead) - the floor of everything that is not an FP4 Linear.
    if lean_glue:
        def attn(q, k, v):
            return torch.addcmul(q, k[..., :1], v[..., :1], value=0.0)
        rescale = 1.0
    else:
        def attn(q, k, v):
            return q + 0.0 * (k.sum() + v.sum())
        rescale = 0.5

    def token(h):
        for ly in layers:
            if "tp_qkv" in ly:
                q, k, v = ly["tp_qkv"](h).split(ly["tp_qkv"].split_sizes, dim=-1)
                a = attn(q, k, v)
                h = ly["o"](a.contiguous(), residual=h)
                h = ly["down"](ly["tp_gate_up"](h), residual=h)
                h = h * rescale if rescale != 1.0 else h
                continue
            if "gate_up" in ly and epilogues and not tp:
                q, k, v = ly["qkv"](h).split([H, KV, KV], dim=-1)
                a = attn(q, k, v)
                h = ly["o"](a.contiguous(), residual=h)       # h + o(a), one launch
                h = ly["down"](ly["gate_up"](h), residual=h)  # silu(g)*u in the gate|up epilogue, + h in down's
                h = h * rescale if rescale != 1.0 else h
                continue
            if "qkv" in ly:
                q, k, v = ly["qkv"](h).split([H, KV, KV], dim=-1)
                a = attn(q, k, v)
                h = h + ly["o"](a.contiguous())
                g, u = ly["gate_up"](h).split([I, I], dim=-1)
                h = h + ly["down"](silu(g) * u)
                h = h * rescale if rescale != 1.0 else h
                continue
            q, k, v = ly["q"](h), ly["k"](h), ly["v"](h)
            a = attn(q, k, v)
            h = h + ly["o"](a)
            h = h + ly["down"](silu(ly["gate"](h)) * ly["up"](h))
            h = h * rescale if rescale != 1.0 else h  # keep magnitudes bounded over many layers of random weights
        return head(h) if head is not None else h

    per_token_fp4 = L * (2 * fp4_bytes(H, H) + 2 * fp4_bytes(KV, H) + 2 * fp4_bytes(I, H) + fp4_bytes(H, I)) // world
    meta = dict(layers=L, fp4_bytes_per_token_per_gpu=per_token_fp4, lm_head_bytes=(V * H * 2 if lm_head else 0),
                fp4_linear_calls_per_token=(4 if (fuse or epilogues) else 7) * L,
                allreduces_per_token=(2 * L if tp else 0),  # issued by the K-split layers (also by a forced one-rank group: reduce_single_rank)
                collective_ranks=world if tp else 0)
    return token, h0, meta


def time_tokens(token, h0, tokens, graph=True, barrier=None, warmup=3):
    """Eager and (optionally) HIP-graph-replayed seconds per token; `barrier` brackets the timed loops at world > 1."""
    sync = barrier or torch.cuda.synchronize
    out = {"eager_s": None, "graph_s": None, "graph_error": None}
    with torch.inference_mode():
        for _ in range(warmup):
            y = token(h0)
        sync()
        t0 = time.perf_counter()
        for _ in range(tokens):
            y = token(h0)
        sync()
        out["eager_s"] = (time.perf_counter() - t0) / tokens
        if not graph:
            return out
        try:
            g = torch.cuda.CUDAGraph()
            static_h = h0.clone()
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                token(static_h)
                torch.cuda.synchronize()
                # thread_local: helper threads of the process (the RCCL watchdog at world > 1) must not break the capture
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    static_out = token(static_h)
            torch.cuda.synchronize()
            for _ in range(warmup):
                g.replay()
            sync()
            t0 = time.perf_counter()
            for _ in range(tokens):
                g.replay()
            sync()
            out["graph_s"] = (time.perf_counter() - t0) / tokens
            assert torch.isfinite(static_out.float()).all()
        except Exception as exc:  # graph capture of collectives may be unavailable
            out["graph_error"] = repr(exc)[:200]
    return out


def main():
    import torch.distributed as dist

    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="mistral7b", choices=MODELS)
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--tokens", type=int, default=64)
    ap.add_argument("--dtype", default="bfloat16")
    ap.add_argument("--fuse", action="store_true", help="one GEMV for q|k|v and one for gate|up (row concatenation)")
    ap.add_argument("--epilogues", action="store_true",
                    help="--fuse plus silu(gate)*up and the residual adds inside the GEMV epilogues (torch_bnb_fp4.fused)")
    ap.add_argument("--batch", type=int, default=1, help="sequences decoded together (activation rows per Linear call)")
    ap.add_argument("--reference-dispatch", action="store_true",
                    help="batch > 1 through dequant + GEMM like the reference, instead of the fused small-batch kernels")
    ap.add_argument("--allreduce", default="dist", choices=("dist", "oneshot"),
                    help="world > 1: torch.distributed all-reduce (RCCL) or the one-shot peer-slot kernel")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--lean-glue", action="store_true",
                    help="attention stand-in as ONE elementwise launch and no rescale: what is left besides the FP4 Linears is minimal")
    args = ap.parse_args()
    cfg = dict(MODELS[args.model])
    if args.layers:
        cfg["layers"] = args.layers
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    # RCCL (version banner) and gloo (connection notes) print on file descriptor 1 from native code: keep the original stdout for the
    # result line, point descriptor 1 at stderr for everything else (same as bench.claim_stdout)
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    backend = os.environ.get("FP4_BENCH_BACKEND", "nccl")
    local_dev = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    dtype = getattr(torch, args.dtype)
    token, h0, meta = build_token_fn(cfg, dev, dtype, world, rank, fuse=args.fuse, epilogues=args.epilogues, batch=args.batch,
                                     reference_dispatch=args.reference_dispatch, allreduce=args.allreduce, lean_glue=args.lean_glue)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    t = time_tokens(token, h0, args.tokens, graph=not args.no_graph and not (world > 1 and backend != "nccl" and args.allreduce == "dist"),
                    barrier=barrier)
    if t["graph_error"] and rank == 0:
        print("graph capture failed:", t["graph_error"], file=sys.stderr)
    if world > 1 and args.allreduce == "oneshot":
        # sync point outside capture: a reduction that timed out waiting for a peer wrote NaN - the figure must not be reported
        from torch_bnb_fp4 import parallel as par

        par.check_oneshot_collective(None)  # every rank is here; a time-out on ANY rank fails the run on every rank
    if rank == 0:
        best = t["graph_s"] or t["eager_s"]
        per_token_fp4 = meta["fp4_bytes_per_token_per_gpu"]
        print(json.dumps({
            "model": args.model, "layers": meta["layers"], "n_gpus": world, "dtype": args.dtype,
            "fp4_linear_calls_per_token": meta["fp4_linear_calls_per_token"], "epilogues_fused": bool(args.epilogues and world == 1),
            "glue": "lean (1 elementwise launch per layer)" if args.lean_glue else "default (6 small launches per layer)",
            "fp4_bytes_per_token_per_gpu": per_token_fp4, "eager_ms_per_step": round(t["eager_s"] * 1e3, 3),
            "graph_ms_per_step": None if t["graph_s"] is None else round(t["graph_s"] * 1e3, 3),
            "batch": args.batch, "batch_path": "reference dispatch (dequant + GEMM)" if args.reference_dispatch and args.batch > 1 else "fused",
            "allreduces_per_token": meta["allreduces_per_token"], "allreduce": args.allreduce if world > 1 else None,
            "tokens_per_s": round(args.batch / best, 1), "fp4_stream_gbps_per_gpu": round(per_token_fp4 / best / 1e9, 1),
            "hbm_floor_ms_per_token_at_8TBps": round((per_token_fp4 + meta["lm_head_bytes"]) / 8e12 * 1e3, 3),
            "data": "synthetic random FP4 bytes + scales; attention replaced by identity; lm_head dense " + args.dtype,
        }), file=result_out, flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
