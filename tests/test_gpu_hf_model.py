"""Drop-in check on a real model class (BASELINE config 4 in miniature): a randomly initialised HF Llama (no download)
has every nn.Linear except lm_head replaced by TorchFP4Linear; prefill (seq > 1 -> dequant + GEMM path) and cached
decode steps ([1, 1, hidden] -> fused GEMV path) must match the same model carrying the dequantised weights densely."""
import copy

import numpy as np
import pytest
import torch
from torch import nn

from gpu_util import dev

pytestmark = pytest.mark.gpu
transformers = pytest.importorskip("transformers")


def _dense_twin(model, P):
    """Copy of `model` whose Linear weights are replaced by dequant(quant(w.half())) - what the FP4 layers represent."""
    twin = copy.deepcopy(model)
    for name, mod in twin.named_modules():
        if isinstance(mod, nn.Linear) and "lm_head" not in name:
            w = mod.weight.data
            packed, absmax = P.quantize_fp4(w.to(torch.float16), 64)
            mod.weight.data = P.dequantize_fp4(packed, absmax, 64, w.shape[0], w.shape[1], w.dtype)
    return twin


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_tiny_llama_prefill_and_cached_decode(dtype):
    import torch_bnb_fp4 as P
    from transformers import LlamaConfig, LlamaForCausalLM

    torch.manual_seed(0)
    cfg = LlamaConfig(vocab_size=1000, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, max_position_embeddings=128, tie_word_embeddings=False)
    model = LlamaForCausalLM(cfg).to(dev()).to(dtype).eval()
    twin = _dense_twin(model, P).eval()
    fp4 = P.recursively_replace_with_fp4_linear(copy.deepcopy(model), as_dtype=dtype, device=dev())
    n_fp4 = sum(isinstance(m, P.TorchFP4Linear) for m in fp4.modules())
    assert n_fp4 == 2 * 7 and isinstance(fp4.lm_head, nn.Linear) and not isinstance(fp4.lm_head, P.TorchFP4Linear)

    ids = torch.randint(0, 1000, (1, 6), device=dev())
    with torch.inference_mode():
        out_ref = twin(ids, use_cache=True)
        out_fp4 = fp4(ids, use_cache=True)
        # prefill: same dequantised weights, same GEMM -> identical up to bf16/fp16 GEMM noise
        ref, got = out_ref.logits.float(), out_fp4.logits.float()
        assert got.shape == ref.shape == (1, 6, 1000)
        assert (got - ref).abs().max().item() <= 0.03 * (1 + ref.abs().max().item())
        # three cached decode steps: the FP4 layers now see [1, 1, hidden] activations -> fused GEMV
        past_ref, past_fp4 = out_ref.past_key_values, out_fp4.past_key_values
        nxt = ref[:, -1].argmax(-1, keepdim=True)
        for _ in range(3):
            o_ref = twin(nxt, past_key_values=past_ref, use_cache=True)
            o_fp4 = fp4(nxt, past_key_values=past_fp4, use_cache=True)
            past_ref, past_fp4 = o_ref.past_key_values, o_fp4.past_key_values
            r, g = o_ref.logits.float(), o_fp4.logits.float()
            assert (g - r).abs().max().item() <= 0.05 * (1 + r.abs().max().item())
            assert np.corrcoef(g.cpu().numpy().ravel(), r.cpu().numpy().ravel())[0, 1] > 0.999
            nxt = r[:, -1].argmax(-1, keepdim=True)
        # greedy generation runs end to end through generate()
        gen = fp4.generate(ids, max_new_tokens=5, do_sample=False)
        assert gen.shape == (1, 11)
        # gated-MLP fusion (gate|up in one launch with silu(g) * u in its epilogue): same model, same numbers - the fused path
        # rounds where the separate ops round, so cached decode agrees with the unfused FP4 model to rounding noise
        fused = P.recursively_replace_with_fp4_linear(copy.deepcopy(model), as_dtype=dtype, device=dev())  # same bytes as fp4
        assert P.fuse_gated_mlps(fused) == cfg.num_hidden_layers
        assert all(isinstance(l.mlp, P.FusedGatedMLP) for l in fused.model.layers)
        o_a = fp4(ids, use_cache=True)
        o_b = fused(ids, use_cache=True)  # prefill: unfused fallback inside the fused module
        assert (o_a.logits.float() - o_b.logits.float()).abs().max().item() <= 0.02 * (1 + o_a.logits.float().abs().max().item())
        nxt = o_a.logits[:, -1].argmax(-1, keepdim=True)
        d_a = fp4(nxt, past_key_values=o_a.past_key_values, use_cache=True).logits.float()
        d_b = fused(nxt, past_key_values=o_b.past_key_values, use_cache=True).logits.float()  # decode: fused epilogue
        assert (d_a - d_b).abs().max().item() <= 0.02 * (1 + d_a.abs().max().item())
        assert fused.generate(ids, max_new_tokens=5, do_sample=False).shape == (1, 11)
