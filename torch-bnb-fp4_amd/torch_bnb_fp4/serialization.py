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
from typing import Dict, Mapping

import torch

from .linear import TorchFP4Linear
from .nn import LinearFP4, Params4bit, QuantState

_STATE_KEY = "weight.quant_state.bitsandbytes__fp4"


def _pack_json(d: dict) -> torch.Tensor:
    return torch.tensor(list(json.dumps(d).encode("utf-8")), dtype=torch.uint8)


def _unpack_json(t: torch.Tensor) -> dict:
    return json.loads(bytes(t.detach().cpu().to(torch.uint8).tolist()).decode("utf-8"))


def _bnb_entries(prefix: str, packed, absmax, code, blocksize: int, shape, dtype, bias) -> Dict[str, torch.Tensor]:
    meta = {"quant_type": "fp4", "blocksize": int(blocksize), "dtype": str(dtype).replace("torch.", ""), "shape": [int(shape[0]), int(shape[1])]}
    out = {
        prefix + "weight": packed.detach().cpu().reshape(-1, 1),
        prefix + "weight.absmax": absmax.detach().float().cpu(),
        prefix + "weight.quant_map": code.detach().float().cpu(),
        prefix + _STATE_KEY: _pack_json(meta),
    }
    if bias is not None:
        out[prefix + "bias"] = bias.detach().cpu()
    return out


def fp4_linear_to_bnb_state(layer: TorchFP4Linear, prefix: str = "") -> Dict[str, torch.Tensor]:
    """State-dict entries of one layer in bitsandbytes' 4-bit layout (tensors moved to the CPU)."""
    qd = layer.quant_data
    return _bnb_entries(prefix, qd.A, qd.absmax, qd.code, qd.blocksize, (qd.M, qd.N), qd.quant_state.dtype, layer.bias)


def fused_linear_to_bnb_state(layer, prefix: str = "", pair_prefixes=None, dtype=None) -> Dict[str, torch.Tensor]:
    """A :class:`~torch_bnb_fp4.fused.FusedFP4Linear` in bitsandbytes' layout.  A plain one (residual epilogue) is one entry under
    ``prefix``; a gate|up one is DE-INTERLEAVED into the two projections it was built from and written under ``pair_prefixes``
    (the rows are a load-time permutation of the bnb bytes, not a new format) - without names for the pair it cannot be saved."""
    from .fused import EPILOGUE_SILU_MUL_PAIRS, deinterleave_rows

    qd = layer.quant_data
    if dtype is None:  # the dtype the weight was quantised from, as its quant_state recorded it
        dtype = getattr(qd.quant_state, "dtype", torch.float16)
    if layer.epilogue != EPILOGUE_SILU_MUL_PAIRS:
        return _bnb_entries(prefix, qd.A, qd.absmax, qd.code, qd.blocksize, (qd.M, qd.N), dtype, layer.bias)
    if not pair_prefixes:
        raise ValueError(f"{prefix or 'layer'}: a gate|up FusedFP4Linear can only be saved as its two projections; it is not inside a "
                         "FusedGatedMLP that remembers their names - save the unfused model (before fuse_gated_mlps) instead")
    (pg, ag), (pu, au), shape = deinterleave_rows(qd.A, qd.absmax, (qd.M, qd.N), qd.blocksize)
    bias = layer.bias
    bg = None if bias is None else bias.reshape(-1, 2)[:, 0].contiguous()
    bu = None if bias is None else bias.reshape(-1, 2)[:, 1].contiguous()
    out = _bnb_entries(pair_prefixes[0], pg, ag, qd.code, qd.blocksize, shape, dtype, bg)
    out.update(_bnb_entries(pair_prefixes[1], pu, au, qd.code, qd.blocksize, shape, dtype, bu))
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
    """Write every FP4 layer of ``model`` (bitsandbytes layout) and every other tensor of its ``state_dict`` to one safetensors
    file.  :class:`TorchFP4Linear` layers are written as they are; the fused layers of :mod:`torch_bnb_fp4.fused` are written as
    the plain projections they were built from (a gated MLP's interleaved gate|up weight is de-interleaved under the two names
    the unfused model uses), so the file always loads into a fresh, UNFUSED model with :func:`load_fp4_layers` - after which
    ``fuse_gated_mlps`` can be applied again.  A fused layer that cannot be expressed that way raises instead of being written as
    tensors no loader would recognise."""
    from safetensors.torch import save_file

    from .fused import FusedFP4Linear
    from .surgery import FusedGatedMLP

    # Tensor-parallel wrappers hold ONE RANK'S shard (row slices, re-packed column ranges, shard-wise concatenations): written as they
    # are they would read back as complete but wrong-sized plain layers, without the collective and without forward(x, residual).
    try:
        from . import parallel as _par

        tp_types = (_par.ColumnParallelFP4Linear, _par.RowParallelFP4Linear, _par.FusedColumnParallelFP4)
    except Exception:  # torch.distributed not built in: no such modules can exist in the model either
        tp_types = ()
    for name, mod in model.named_modules():
        if tp_types and isinstance(mod, tp_types):
            raise ValueError(f"save_fp4_model: '{name}' is a {type(mod).__name__} holding one rank's shard of its weight; save the "
                             "unsharded model (before the tensor-parallel layers are built) and shard again after loading")

    tensors: Dict[str, torch.Tensor] = {}
    fp4_prefixes = []
    gated = {}  # prefix of a FusedGatedMLP's gate_up child -> (prefix for gate, prefix for up, dtype)
    for name, mod in model.named_modules():
        if isinstance(mod, FusedGatedMLP):
            base = name + "." if name else ""
            gated[base + "gate_up."] = (base + mod.projection_names[0] + ".", base + mod.projection_names[1] + ".", mod.quant_dtype)
    for name, mod in model.named_modules():
        prefix = name + "." if name else ""
        if isinstance(mod, TorchFP4Linear):
            fp4_prefixes.append(prefix)
            tensors.update(fp4_linear_to_bnb_state(mod, prefix))
        elif isinstance(mod, FusedFP4Linear):
            fp4_prefixes.append(prefix)
            g = gated.get(prefix)
            tensors.update(fused_linear_to_bnb_state(mod, prefix, None if g is None else g[:2], None if g is None else g[2]))
    for key, val in model.state_dict().items():
        if not any(key.startswith(p) for p in fp4_prefixes):
            tensors[key] = val.detach().cpu().contiguous()
    save_file({k: v.contiguous() for k, v in tensors.items()}, path)


def load_fp4_layers(model: torch.nn.Module, path: str, device="cuda", use_codebook_dequant: bool = True,
                    strict: bool = True) -> torch.nn.Module:
    """Replace, in ``model``, every ``nn.Linear`` for which ``path`` holds a bitsandbytes FP4 weight by a
    :class:`TorchFP4Linear` built from the stored bytes; the file's other tensors are loaded into the model.  Tensors of the
    file that the model has no place for raise a ``KeyError`` (``strict=False``: they are listed in ``model.fp4_unexpected_keys``
    instead) - a checkpoint is never half-applied silently."""
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
    unexpected = []
    if rest and isinstance(model, torch.nn.Module):
        unexpected = list(model.load_state_dict(rest, strict=False).unexpected_keys)
    if unexpected and strict:
        raise KeyError(f"load_fp4_layers: {len(unexpected)} tensor(s) of {path} have no place in the model (saved from a model with a "
                       f"different structure?): {unexpected[:8]}{' ...' if len(unexpected) > 8 else ''}")
    if isinstance(model, torch.nn.Module):
        model.fp4_unexpected_keys = unexpected
    return model
