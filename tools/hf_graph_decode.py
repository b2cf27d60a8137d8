#!/usr/bin/env python3
"""BASELINE config 4 through a REAL Hugging Face model class, without the host pacing the GPU: a Mistral-7B-shaped
`MistralForCausalLM` (random weights, no hub download) with every Linear but lm_head as FP4, greedy decode over a static KV
cache, ONE decode step captured into a HIP graph (torch_bnb_fp4.GraphedStep) and replayed per token.

The reference's own harness (examples/speed_test_mistral_7b.py) times `model.generate`, which is host-bound on this box at
~9 ms per token (tools/hf_mistral_speed.py: 81 tok/s dense bf16, ~107 tok/s after the swap).  Here the same model code runs
at what its kernels cost.  Prints one JSON line: tokens/s eager and graph-replayed, for dense bf16 and for FP4 (plain swap,
and with the gated-MLP fusion), and checks that the graph-replayed greedy tokens equal the eager ones.

    python tools/hf_graph_decode.py [layers=32] [new_tokens=64]
"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "torch-bnb-fp4_amd")]
import torch  # noqa: E402
from transformers import MistralConfig, MistralForCausalLM, StaticCache  # noqa: E402

import torch_bnb_fp4 as pkg  # noqa: E402

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 32
new_tokens = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda", 0)
cfg = MistralConfig(vocab_size=32000, hidden_size=4096, intermediate_size=14336, num_hidden_layers=layers, num_attention_heads=32,
                    num_key_value_heads=8, max_position_embeddings=4096, sliding_window=None)
torch.manual_seed(0)
with torch.device(dev):
    model = MistralForCausalLM(cfg).to(torch.bfloat16).eval()
prompt = torch.randint(0, 32000, (1, 32), device=dev)
MAX_LEN = 32 + new_tokens + 8


def decode(m, use_graph: bool):
    """Greedy decode of `new_tokens` tokens after a prefill; returns (tokens, seconds per decoded token)."""
    cache = StaticCache(config=m.config, max_cache_len=MAX_LEN)
    with torch.inference_mode():
        pos = torch.arange(prompt.shape[1], device=dev)
        out = m(input_ids=prompt, past_key_values=cache, cache_position=pos, use_cache=True)
        tok = out.logits[:, -1:].argmax(-1)
        cur = torch.tensor([prompt.shape[1]], device=dev)

        ar = torch.arange(MAX_LEN, device=dev).view(1, 1, 1, MAX_LEN)

        def step(t, p):
            # mask and positions as TENSOR functions of the position: transformers derives them from host-side cache state
            # otherwise, which a graph replay would freeze at its capture-time value
            o = m(input_ids=t, attention_mask=ar <= p.view(1, 1, 1, 1), position_ids=p.view(1, 1), past_key_values=cache,
                  cache_position=p, use_cache=True)
            return o.logits[:, -1:].argmax(-1)

        if use_graph:
            saved = [layer.cumulative_length.clone() for layer in cache.layers]
            runner = pkg.GraphedStep(step, tok, cur, warmup=1)
            for layer, c in zip(cache.layers, saved):
                layer.cumulative_length.copy_(c)  # the warm-up advanced the cache's device-side write position; the capture ran nothing
        else:
            runner = step
        toks = [tok.clone()]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(new_tokens - 1):
            tok = runner(tok, cur)
            toks.append(tok.clone())
            cur = cur + 1
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (new_tokens - 1)
    return torch.cat(toks, dim=1), dt


res = {"model": f"Mistral-7B shapes, {layers} layers, random weights", "prompt_tokens": 32, "new_tokens": new_tokens}
t_dense_e, s = decode(model, False)
res["dense_bf16_eager_tok_s"] = round(1 / s, 1)
t_dense_g, s = decode(model, True)
res["dense_bf16_graph_tok_s"] = round(1 / s, 1)
res["dense_graph_tokens_equal_eager"] = bool(torch.equal(t_dense_e, t_dense_g))
model = pkg.recursively_replace_with_fp4_linear(model, as_dtype=torch.bfloat16, device=dev)
t_e, s = decode(model, False)
res["fp4_eager_tok_s"] = round(1 / s, 1)
t_g, s = decode(model, True)
res["fp4_graph_tok_s"] = round(1 / s, 1)
res["fp4_graph_tokens_equal_eager"] = bool(torch.equal(t_e, t_g))
res["gated_mlps_fused"] = pkg.fuse_gated_mlps(model)
t_f, s = decode(model, True)
res["fp4_fused_mlp_graph_tok_s"] = round(1 / s, 1)
# random weights make the logits noise: a 1-ulp difference in a fused silu flips an argmax sooner or later and the greedy sequences
# part ways from there (a property of the random model, not of the kernels; the layers are checked bit-level in tests/test_gpu_fused.py)
same = (t_f == t_g).view(-1).int()
res["fused_tokens_equal_unfused_until"] = int(same.cumprod(0).sum())
res["gpu_mem_gb"] = round(torch.cuda.memory_allocated() / 1e9, 2)
print(json.dumps(res))
