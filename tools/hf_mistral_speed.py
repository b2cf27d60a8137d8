#!/usr/bin/env python3
"""The reference's examples/speed_test_mistral_7b.py, restated offline: a Mistral-7B-shaped HF model with random weights
(no hub download), `generate` timed before and after every nn.Linear except lm_head is swapped for TorchFP4Linear.
Prints tokens/s like the reference (examples/speed_test_mistral_7b.py:71-130); there is no bitsandbytes column on ROCm."""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "torch-bnb-fp4_amd")]
import torch  # noqa: E402
from transformers import MistralConfig, MistralForCausalLM  # noqa: E402

import torch_bnb_fp4 as pkg  # noqa: E402

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 32
new_tokens = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda", 0)
cfg = MistralConfig(vocab_size=32000, hidden_size=4096, intermediate_size=14336, num_hidden_layers=layers, num_attention_heads=32,
                    num_key_value_heads=8, max_position_embeddings=4096, sliding_window=None)
torch.manual_seed(0)
t0 = time.perf_counter()
with torch.device(dev):
    model = MistralForCausalLM(cfg).to(torch.bfloat16).eval()
print(f"built random Mistral-7B-shaped model ({sum(p.numel() for p in model.parameters()) / 1e9:.2f} B params) in {time.perf_counter() - t0:.0f} s", flush=True)
ids = torch.randint(0, 32000, (1, 32), device=dev)


def tok_per_s(m, runs=2):
    best = 0.0
    for i in range(runs + 1):  # first run discarded
        torch.cuda.synchronize()
        t = time.perf_counter()
        out = m.generate(ids, max_new_tokens=new_tokens, min_new_tokens=new_tokens, do_sample=False, use_cache=True, pad_token_id=0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        if i:
            best = max(best, (out.shape[1] - ids.shape[1]) / dt)
    return best


with torch.inference_mode():
    dense = tok_per_s(model)
    t0 = time.perf_counter()
    model = pkg.recursively_replace_with_fp4_linear(model, as_dtype=torch.bfloat16, device=dev)
    torch.cuda.synchronize()
    swap_s = time.perf_counter() - t0
    n_fp4 = sum(isinstance(m, pkg.TorchFP4Linear) for m in model.modules())
    fp4 = tok_per_s(model)
print(json.dumps({"model": f"Mistral-7B shapes, {layers} layers, random weights", "prompt_tokens": 32, "new_tokens": new_tokens,
                  "dense_bf16_tokens_per_s": round(dense, 1), "torch_bnb_fp4_amd_tokens_per_s": round(fp4, 1), "fp4_layers": n_fp4,
                  "quantise_and_swap_s": round(swap_s, 2), "mode": "HF generate, greedy, eager (no graph capture)",
                  "gpu_mem_gb_after": round(torch.cuda.memory_allocated() / 1e9, 2)}))
