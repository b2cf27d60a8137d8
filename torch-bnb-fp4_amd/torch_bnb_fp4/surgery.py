"""Model surgery: nn.Linear -> FP4 layer -> :class:`TorchFP4Linear`.

Counterpart of the reference's module-walking helpers (torch_bnb_fp4/__init__.py:717-922) with
the same names, keyword arguments and defaults.  The nn.Linear -> FP4 step, which the reference
delegates to bitsandbytes (``bnb.nn.LinearFP4`` + ``Params4bit.cuda()``), uses this package's own
:class:`~torch_bnb_fp4.nn.LinearFP4` and HIP quantiser, so no bitsandbytes install is needed.
"""
from __future__ import annotations

import logging
from typing import List, Optional, TypeVar

import torch
from torch import nn

from .functional import quantize_fp4
from .linear import TorchFP4Linear
from .nn import FP4_LINEAR_TYPES, LinearFP4, Params4bit, QuantState, fp4_code

T_Model = TypeVar("T_Model", bound=nn.Module)
log = logging.getLogger(__name__)


@torch.no_grad()
def swap_linear_with_bnb_linear(linear: nn.Linear, dtype=torch.float16) -> LinearFP4:
    """New (still dense) ``LinearFP4`` holding clones of ``linear``'s weight and bias; it quantises
    when moved to a GPU (reference :717-747)."""
    # built on the meta device: nn.Linear.__init__ would otherwise allocate and randomly initialise a full dense
    # weight on the CPU only to have it replaced (18 s of a 7B model's 224 layers)
    fp4 = LinearFP4(input_features=linear.in_features, output_features=linear.out_features,
                    bias=linear.bias is not None, compute_dtype=dtype, device="meta")
    fp4.weight = Params4bit(linear.weight.data.clone().detach(), False, None, fp4.blocksize, "fp4")
    if linear.bias is not None:
        fp4.bias = nn.Parameter(linear.bias.data.clone().detach(), requires_grad=False)
    fp4.requires_grad_(False)
    return fp4


def check_if_name_contained_in_list(name: str, names_list) -> bool:
    """True when any entry of ``names_list`` is a substring of ``name`` (reference :750-756)."""
    return any(entry in name for entry in names_list)


def todevice_if_necessary(module, device):
    """Make sure an FP4 layer's weight really is packed uint8 on ``device``; quantise it directly
    if moving the module did not (reference :759-778)."""
    if module.weight.data.dtype != torch.uint8 and isinstance(module, FP4_LINEAR_TYPES):
        module = module.to(device)
        w = module.weight
        if not (w.data.device == torch.device(device) and w.data.dtype == torch.uint8):
            log.debug("layer was not quantised by the device move; quantising its weight directly")
            dense = w.data.to(device=device, dtype=torch.float16)
            packed, absmax = quantize_fp4(dense, module.blocksize)
            state = QuantState(absmax, dense.shape, fp4_code().to(device), module.blocksize, w.data.dtype)
            module.weight = Params4bit(packed, False, state, module.blocksize, "fp4")
    return module


def _device_is_gpu(device) -> bool:
    kind = device.type if hasattr(device, "type") else str(device).split(":")[0]
    return kind == "cuda"


def _to_fp4_linear(layer: nn.Module, device, as_dtype, use_codebook_dequant: bool, name: str) -> TorchFP4Linear:
    """nn.Linear or FP4 layer -> TorchFP4Linear on ``device``."""
    if not isinstance(layer, FP4_LINEAR_TYPES):
        layer = swap_linear_with_bnb_linear(layer, dtype=as_dtype)
    layer = layer.to(device)
    if getattr(layer.weight, "quant_state", None) is None:
        layer = todevice_if_necessary(layer, device)
    return TorchFP4Linear(lin=layer, use_codebook_dequant=use_codebook_dequant, name=name)


def recursively_replace_with_fp4_linear(
    module: T_Model,
    as_dtype=torch.float16,
    use_codebook_dequant=True,
    device: torch.device = torch.device("cuda" if torch.cuda.is_available() else "cpu"),
    return_final_module: bool = True,
    only_replace_bnb_layers: bool = False,
    ignore_layer_names: List[str] = ["lm_head"],
    parent="",
    debug: bool = False,
) -> Optional[T_Model]:
    """Replace every nn.Linear / FP4 linear below ``module`` by a :class:`TorchFP4Linear`.

    Same contract as the reference (:781-922): children whose *own* name contains an entry of
    ``ignore_layer_names`` are skipped together with their subtree; ``only_replace_bnb_layers``
    leaves plain nn.Linear alone; a root that is itself a Linear is converted and returned;
    ``named_children()`` dedupes shared modules, so a Linear object reused in several slots is
    swapped only in the first one (the reference's sanity model relies on exactly that).
    """
    assert _device_is_gpu(device), "Device type must be cuda!"
    prefix = parent + "." if parent != "" else ""
    swapped_dense = False
    for name, child in module.named_children():
        child_name = prefix + name
        if check_if_name_contained_in_list(name, ignore_layer_names):
            if debug:
                print(f"Ignoring name: {child_name}, as it is in the ignore list")
            continue
        if isinstance(child, (nn.Linear,) + FP4_LINEAR_TYPES):
            is_fp4 = isinstance(child, FP4_LINEAR_TYPES)
            if not is_fp4 and only_replace_bnb_layers:
                if debug:
                    print(f"Ignoring {child_name}, as only_replace_bnb_layers=True")
                continue
            if debug:
                print(f"Replacing {'FP4 layer ' if is_fp4 else ''}{child_name} with TorchFP4Linear.")
            module._modules[name] = _to_fp4_linear(child, device, as_dtype, use_codebook_dequant, child_name)
            swapped_dense |= not is_fp4
        elif isinstance(child, nn.Module):
            recursively_replace_with_fp4_linear(child, as_dtype=as_dtype, use_codebook_dequant=use_codebook_dequant,
                                                device=device, return_final_module=False,
                                                only_replace_bnb_layers=only_replace_bnb_layers,
                                                ignore_layer_names=ignore_layer_names, parent=child_name, debug=debug)
    if isinstance(module, (nn.Linear,) + FP4_LINEAR_TYPES):
        is_fp4 = isinstance(module, FP4_LINEAR_TYPES)
        if is_fp4 or not only_replace_bnb_layers:
            if debug:
                print(f"Replacing {parent} with TorchFP4Linear.")
            module = _to_fp4_linear(module, device, as_dtype, use_codebook_dequant, parent)
            swapped_dense |= not is_fp4
        elif debug:
            print(f"Ignoring {parent}, as only_replace_bnb_layers=True")
    if swapped_dense:
        torch.cuda.empty_cache()  # the dense copies are gone; give their blocks back (:919-920)
    if return_final_module:
        return module


def set_small_batch_fused(module: nn.Module, enabled: bool = True) -> int:
    """Route 2..128 activation rows of every :class:`TorchFP4Linear` below ``module`` to the fused small-batch kernels
    (``enabled=True``) or back to the reference's dispatch, dequant + GEMM for every batch > 1
    (torch_bnb_fp4/__init__.py:592,616-617; the default, so that a converted model behaves like the reference's).
    Not part of the reference surface.  Returns the number of layers touched.  Batched decode through Mistral-7B shapes:
    3 237 tok/s fused vs 963 tok/s through the reference dispatch at 8 sequences (profiles/)."""
    n = 0
    for m in module.modules():
        if isinstance(m, TorchFP4Linear):
            m.quant_data.small_batch_fused = bool(enabled)
            n += 1
    return n



class FusedGatedMLP(nn.Module):
    """``down(silu(gate(x)) * up(x))`` with the gate and up projections as ONE launch whose epilogue applies the activation
    and the product (:mod:`torch_bnb_fp4.fused`); what :func:`fuse_gated_mlps` puts in place of a Llama / Mistral style MLP."""

    def __init__(self, gate: TorchFP4Linear, up: TorchFP4Linear, down: nn.Module, names=("gate_proj", "up_proj")):
        super().__init__()
        from .fused import FusedFP4Linear

        self.gate_up = FusedFP4Linear.gate_up(gate, up)
        self.down_proj = down
        # the two projections' names in the unfused model: save_fp4_model writes the interleaved weight back under them
        # (bitsandbytes layout, one entry per projection), so that the file loads into a fresh, unfused model
        self.projection_names = (str(names[0]), str(names[1]))
        self.quant_dtype = gate.quant_data.quant_state.dtype

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.down_proj(self.gate_up(x))


def fuse_gated_mlps(module: nn.Module, gate: str = "gate_proj", up: str = "up_proj", down: str = "down_proj", act: str = "act_fn") -> int:
    """After :func:`recursively_replace_with_fp4_linear`: replace every sub-module that looks like a SiLU-gated MLP (children
    ``gate`` / ``up`` / ``down`` with the first two :class:`TorchFP4Linear` of equal shape, activation ``act`` a SiLU) by a
    :class:`FusedGatedMLP`.  Single-token calls then pay one launch for gate + up + activation + product instead of four; other
    shapes run the unfused sequence.  Returns how many were replaced.  Not in the reference (its surface stops at the Linear)."""
    count = 0
    for name, child in list(module.named_children()):
        g, u, d, a = (getattr(child, n, None) for n in (gate, up, down, act))
        silu = isinstance(a, nn.SiLU) or "silu" in type(a).__name__.lower() or a is nn.functional.silu
        if (isinstance(g, TorchFP4Linear) and isinstance(u, TorchFP4Linear) and isinstance(d, nn.Module) and silu
                and (g.quant_data.M, g.quant_data.N, g.quant_data.blocksize) == (u.quant_data.M, u.quant_data.N, u.quant_data.blocksize)
                ):
            module._modules[name] = FusedGatedMLP(g, u, d, names=(gate, up))
            count += 1
        else:
            count += fuse_gated_mlps(child, gate, up, down, act)
    return count
