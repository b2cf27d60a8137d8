#!/usr/bin/env python3
"""Batch-1 decode through every FP4 Linear of a Mistral-7B / Llama-3-8B shaped model (BASELINE configs 4 and 5).

Synthetic weights of the real shapes (no checkpoints offline): per decoder layer q,o 4096x4096, k,v 1024x4096,
gate,up 14336x4096, down 4096x14336, all FP4 blocksize 64; lm_head stays a dense bf16 Linear, as the reference's
default `ignore_layer_names=["lm_head"]` leaves it (torch_bnb_fp4/__init__.py:788).  One "token" = the dependent chain
of the 7 x L fused GEMVs (+ the elementwise glue between them and the lm_head GEMV), run through the package's
TorchFP4Linear modules, eagerly and replayed from a HIP graph.  Attention itself is not part of this path and is
replaced by an identity on q (the GEMV traffic is what is being measured).

    python tools/decode_bench.py [--model mistral7b|llama3-8b] [--layers 32] [--tokens 64]
    torchrun --nproc-per-node N tools/decode_bench.py --model llama3-8b      # tensor parallel: q/k/v/gate/up M-split,
                                                                              # o/down K-split + RCCL all-reduce
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "torch-bnb-fp4_amd")]
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MODELS = {"mistral7b": dict(hidden=4096, kv=1024, inter=14336, vocab=32000, layers=32),
          "llama3-8b": dict(hidden=4096, kv=1024, inter=14336, vocab=128256, layers=32)}
BS = 64


def fp4_bytes(m, k):
    return m * k // 2 + 4 * (m * k // BS)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="mistral7b", choices=MODELS)
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--tokens", type=int, default=64)
    ap.add_argument("--dtype", default="bfloat16")
    ap.add_argument("--fuse", action="store_true", help="one GEMV for q|k|v and one for gate|up (row concatenation)")
    ap.add_argument("--batch", type=int, default=1, help="sequences decoded together (activation rows per Linear call)")
    ap.add_argument("--reference-dispatch", action="store_true",
                    help="batch > 1 through dequant + GEMM like the reference, instead of the fused small-batch kernels")
    args = ap.parse_args()
    cfg = dict(MODELS[args.model])
    if args.layers:
        cfg["layers"] = args.layers
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    import torch_bnb_fp4 as pkg
    from torch_bnb_fp4 import parallel as par

    dtype = getattr(torch, args.dtype)
    H, KV, I, V, L = cfg["hidden"], cfg["kv"], cfg["inter"], cfg["vocab"], cfg["layers"]
    gen = torch.Generator(device=dev).manual_seed(7)

    def fp4_weight(m, k):
        packed = torch.randint(0, 256, (m * k // 2, 1), dtype=torch.uint8, device=dev, generator=gen)
        absmax = torch.rand(m * k // BS, device=dev, generator=gen) * 0.02 + 0.002
        return packed, absmax

    def linear(m, k, kind):
        if isinstance(m, (list, tuple)):  # fused rows
            packed, absmax, (m, k) = par.concat_rows([(*fp4_weight(mi, k), (mi, k)) for mi in m], BS)
        else:
            packed, absmax = fp4_weight(m, k)
        if world == 1:
            state = pkg.QuantState(absmax, (m, k), pkg.ext.code_table("tree").to(dev), BS)
            qd = pkg.QuantData(packed, state, state.shape, original_lin=None, bias=None,
                               small_batch_fused=not args.reference_dispatch)
            return qd.forward
        if kind == "col":
            return par.ColumnParallelFP4Linear(packed, absmax, (m, k), BS, gather_output=False)
        return par.RowParallelFP4Linear(packed, absmax, (m, k), BS, input_is_parallel=True)

    if args.fuse and world == 1:
        layers = [dict(qkv=linear([H, KV, KV], H, "col"), o=linear(H, H, "row"), gate_up=linear([I, I], H, "col"),
                       down=linear(H, I, "row")) for _ in range(L)]
    else:
        layers = [dict(q=linear(H, H, "col"), k=linear(KV, H, "col"), v=linear(KV, H, "col"), o=linear(H, H, "row"),
                       gate=linear(I, H, "col"), up=linear(I, H, "col"), down=linear(H, I, "row")) for _ in range(L)]
    lm_head = torch.nn.Linear(H, V, bias=False, device=dev, dtype=dtype)
    h0 = torch.randn(args.batch, H, device=dev, generator=gen).to(dtype)

    def token(h):
        for ly in layers:
            if "qkv" in ly:
                q, k, v = ly["qkv"](h).split([H, KV, KV], dim=-1)
                a = q + 0.0 * (k.sum() + v.sum())
                h = h + ly["o"](a.contiguous())
                g, u = ly["gate_up"](h).split([I, I], dim=-1)
                h = (h + ly["down"](torch.nn.functional.silu(g) * u)) * 0.5
                continue
            q, k, v = ly["q"](h), ly["k"](h), ly["v"](h)
            a = q + 0.0 * (k.sum() + v.sum())  # stand-in for attention: keeps k, v live and dependent
            h = h + ly["o"](a)
            h = h + ly["down"](torch.nn.functional.silu(ly["gate"](h)) * ly["up"](h))
            h = h * 0.5  # keep magnitudes bounded over many layers of random weights
        return lm_head(h)

    per_token_fp4 = L * (2 * fp4_bytes(H, H) + 2 * fp4_bytes(KV, H) + 2 * fp4_bytes(I, H) + fp4_bytes(H, I)) // world
    with torch.inference_mode():
        for _ in range(3):
            out = token(h0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.tokens):
            out = token(h0)
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / args.tokens
        graph_s = None
        try:
            g = torch.cuda.CUDAGraph()
            static_h = h0.clone()
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                token(static_h)
                torch.cuda.synchronize()
                with torch.cuda.graph(g):
                    static_out = token(static_h)
            torch.cuda.synchronize()
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.tokens):
                g.replay()
            torch.cuda.synchronize()
            graph_s = (time.perf_counter() - t0) / args.tokens
            assert torch.isfinite(static_out.float()).all()
        except Exception as exc:  # graph capture of collectives may be unavailable
            if rank == 0:
                print("graph capture failed:", repr(exc)[:200], file=sys.stderr)
    if rank == 0:
        best = graph_s or eager
        print(json.dumps({
            "model": args.model, "layers": L, "n_gpus": world, "dtype": args.dtype, "fp4_linear_calls_per_token": (4 if (args.fuse and world == 1) else 7) * L,
            "fp4_bytes_per_token_per_gpu": per_token_fp4, "eager_ms_per_step": round(eager * 1e3, 3),
            "graph_ms_per_step": None if graph_s is None else round(graph_s * 1e3, 3),
            "batch": args.batch, "batch_path": "reference dispatch (dequant + GEMM)" if args.reference_dispatch and args.batch > 1 else "fused",
            "tokens_per_s": round(args.batch / best, 1), "fp4_stream_gbps_per_gpu": round(per_token_fp4 / best / 1e9, 1),
            "hbm_floor_ms_per_token_at_8TBps": round((per_token_fp4 + V * H * 2 // 1) / 8e12 * 1e3, 3),
            "data": "synthetic random FP4 bytes + scales; attention replaced by identity; lm_head dense " + args.dtype,
        }))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
