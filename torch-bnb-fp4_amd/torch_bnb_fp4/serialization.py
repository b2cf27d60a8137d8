"""bitsandbytes-format (de)serialisation of FP4 Linear layers.

A pre-quantised bitsandbytes checkpoint stores, per ``Linear4bit`` weight (key names as consumed by
``transformers/quantizers/quantizer_bnb_4bit.py:172-184``):

    <prefix>weight                                   uint8  [numel/2, 1]   packed nibbles
    <prefix>weight.absmax                            float32 [numel/blocksize]
    <prefix>weight.quant_map                         float32 [16]           the code (k/12)
    <prefix>weight.quant_state.bitsandbytes__fp4     uint8  [len]          utf-8 JSON: quant_type, blocksize, dtype, shape
    <prefix>bias                                     (optional)

The reference cannot load or save its layers at all (its wrapped module is hidden in a python list,
torch_bnb_fp4/__init__.py:644); this module round-trips :class:`TorchFP4Linear` through exactly that format, so
FP4 safetensors written by bitsandbytes/transformers load straight into the MI355X path without re-quantising.
Nested (double-quantised) absmax is rejected, as in the reference (README.md:223-224).
"""
from __future__ import annotations

import json
from typing import Dict, Mapping, Optional

import torch

from .linear import TorchFP4Linear
from .nn import LinearFP4, Params4bit, QuantState

_STATE_KEY = "weight.quant_state.bitsandbytes__fp4"


def _pack_json(d: dict) -> torch.Tensor:
    return torch.tensor(list(json.dumps(d).encode("utf-8")), dtype=torch.uint8)


def _unpack_json(t: torch.Tensor) -> dict:
    return json.loads(bytes(t.detach().cpu().to(torch.uint8).tolist()).decode("utf-8"))


def fp4_linear_to_bnb_state(layer: TorchFP4Linear, prefix: str = "") -> Dict[str, torch.Tensor]:
    """State-dict entries of one layer in bitsandbytes' 4-bit layout (tensors moved to the CPU)."""
    qd = layer.quant_data
    meta = {"quant_type": "fp4", "blocksize": int(qd.blocksize), "dtype": str(qd.quant_state.dtype).replace("torch.", ""),
            "shape": [int(qd.M), int(qd.N)]}
    out = {
        prefix + "weight": qd.A.detach().cpu().reshape(-1, 1),
        prefix + "weight.absmax": qd.absmax.detach().cpu(),
        prefix + "weight.quant_map": qd.code.detach().cpu(),
        prefix + _STATE_KEY: _pack_json(meta),
    }
    if layer.bias is not None:
        out[prefix + "bias"] = layer.bias.detach().cpu()
    return out


def fp4_linear_from_bnb_state(state: Mapping[str, torch.Tensor], prefix: str = "", device="cuda",
                              use_codebook_dequant: bool = True, name: str = "") -> TorchFP4Linear:
    """Build a :class:`TorchFP4Linear` from bitsandbytes-format entries (no re-quantisation)."""
    if prefix + "weight.nested_absmax" in state:
        raise ValueError("nested (double-quantised) absmax is not supported")
    if prefix + _STATE_KEY not in state:
        raise KeyError(f"{prefix + _STATE_KEY} not found: not a bitsandbytes FP4 weight (NF4 is not supported)")
    meta = _unpack_json(state[prefix + _STATE_KEY])
    if meta.get("quant_type") != "fp4":
        raise ValueError(f"quant_type {meta.get('quant_type')!r} is not fp4")
    M, K = (int(v) for v in meta["shape"])
    bs = int(meta["blocksize"])
    dev = torch.device(device)
    packed = state[prefix + "weight"].to(dev).reshape(-1, 1).contiguous()
    absmax = state[prefix + "weight.absmax"].to(dev).float().contiguous()
    code = state[prefix + "weight.quant_map"].to(dev).float().contiguous()
    if packed.dtype != torch.uint8 or packed.numel() != (M * K + 1) // 2 or absmax.numel() != -(-M * K // bs) or code.numel() != 16:
        raise ValueError("inconsistent FP4 state: packed/absmax/quant_map sizes do not match the recorded shape")
    bias = state.get(prefix + "bias")
    shell = LinearFP4(K, M, bias=bias is not None, device="meta")
    qs = QuantState(absmax, (M, K), code, bs, getattr(torch, meta.get("dtype", "float16")))
    shell._parameters["weight"] = Params4bit(packed, False, qs, bs, "fp4")
    if bias is not None:
        shell._parameters["bias"] = torch.nn.Parameter(bias.to(dev), requires_grad=False)
    return TorchFP4Linear(shell, use_codebook_dequant=use_codebook_dequant, name=name)


def save_fp4_model(model: torch.nn.Module, path: str) -> None:
    """Write every :class:`TorchFP4Linear` of ``model`` (bitsandbytes layout) and every other tensor of its
    ``state_dict`` to one safetensors file."""
    from safetensors.torch import save_file

    tensors: Dict[str, torch.Tensor] = {}
    fp4_prefixes = []
    for name, mod in model.named_modules():
        if isinstance(mod, TorchFP4Linear):
            prefix = name + "." if name else ""
            fp4_prefixes.append(prefix)
            tensors.update(fp4_linear_to_bnb_state(mod, prefix))
    for key, val in model.state_dict().items():
        if not any(key.startswith(p) for p in fp4_prefixes):
            tensors[key] = val.detach().cpu().contiguous()
    save_file({k: v.contiguous() for k, v in tensors.items()}, path)


def load_fp4_layers(model: torch.nn.Module, path: str, device="cuda", use_codebook_dequant: bool = True) -> torch.nn.Module:
    """Replace, in ``model``, every ``nn.Linear`` for which ``path`` holds a bitsandbytes FP4 weight by a
    :class:`TorchFP4Linear` built from the stored bytes; other tensors are loaded with ``load_state_dict(strict=False)``."""
    from safetensors.torch import load_file

    state = load_file(path)
    prefixes = sorted(k[: -len(_STATE_KEY)] for k in state if k.endswith(_STATE_KEY))
    consumed = set()
    for prefix in prefixes:
        parent_name, _, child = prefix.rstrip(".").rpartition(".")
        parent = model.get_submodule(parent_name) if parent_name else model
        layer = fp4_linear_from_bnb_state(state, prefix, device, use_codebook_dequant, name=prefix.rstrip("."))
        if child:
            parent._modules[child] = layer
        else:
            model = layer
        consumed.update(k for k in state if k.startswith(prefix))
    rest = {k: v for k, v in state.items() if k not in consumed}
    if rest and isinstance(model, torch.nn.Module):
        model.load_state_dict(rest, strict=False)
    return model
