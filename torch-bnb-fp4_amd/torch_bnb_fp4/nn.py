"""bitsandbytes-free stand-ins for the FP4 layer types the reference consumes.

The reference wraps ``bitsandbytes.nn.LinearFP4`` / ``Linear4bit`` modules whose ``weight`` is a
``Params4bit`` carrying a ``QuantState`` (torch_bnb_fp4/__init__.py:649-662).  bitsandbytes
(``<0.43``, requirements.txt:1) is a CUDA package and is not available on the MI355X boxes, so this
module provides the same attribute contract - ``weight.data`` = packed ``uint8[numel/2, 1]``,
``weight.quant_state.{absmax, code, blocksize, shape, dtype}``, ``bias``, ``in_features`` /
``out_features`` - with quantisation done by this package's HIP quantiser the first time the
module is moved to a GPU (what ``Params4bit.cuda()`` does in bitsandbytes).  Real bitsandbytes
modules, when importable, are accepted everywhere these are (duck typing on ``quant_state``).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import nn

from ._ext import ext
from .functional import quantize_fp4

try:  # optional: accept genuine bitsandbytes layers too
    import bitsandbytes as _bnb  # type: ignore
    from bitsandbytes.nn.modules import Linear4bit as _BnbLinear4bit, LinearFP4 as _BnbLinearFP4, Params4bit as _BnbParams4bit  # type: ignore

    HAVE_BITSANDBYTES = True
except Exception:  # pragma: no cover - bitsandbytes is absent in this image
    _bnb = None
    _BnbLinear4bit = _BnbLinearFP4 = _BnbParams4bit = ()
    HAVE_BITSANDBYTES = False


def fp4_code() -> torch.Tensor:
    """The 16-entry FP4 code as bitsandbytes stores it in ``quant_state.code`` (k/12 in f32)."""
    return ext.code_table("tree")


class QuantState:
    """Attribute-compatible subset of ``bitsandbytes.functional.QuantState`` (FP4, no nested absmax)."""

    def __init__(self, absmax: torch.Tensor, shape, code: torch.Tensor, blocksize: int = 64,
                 dtype: torch.dtype = torch.float16, quant_type: str = "fp4"):
        self.absmax = absmax
        self.shape = torch.Size(shape)
        self.code = code
        self.blocksize = int(blocksize)
        self.dtype = dtype
        self.quant_type = quant_type
        self.nested = False

    def to(self, device) -> "QuantState":
        self.absmax = self.absmax.to(device)
        self.code = self.code.to(device)
        return self


class Params4bit(nn.Parameter):
    """``nn.Parameter`` that is either a dense float weight (``quant_state is None``) or packed FP4."""

    def __new__(cls, data: Optional[torch.Tensor] = None, requires_grad: bool = False,
                quant_state: Optional[QuantState] = None, blocksize: int = 64, quant_type: str = "fp4"):
        if data is None:
            data = torch.empty(0)
        self = torch.Tensor._make_subclass(cls, data, requires_grad)
        self.quant_state = quant_state
        self.blocksize = blocksize
        self.quant_type = quant_type
        return self

    def __deepcopy__(self, memo):
        return Params4bit(self.data.clone(), self.requires_grad, self.quant_state, self.blocksize, self.quant_type)

    @classmethod
    def quantized_from(cls, dense: torch.Tensor, device, blocksize: int = 64) -> "Params4bit":
        """Quantise like ``Params4bit.cuda()``: the weight is cast to fp16 first, then blockwise FP4."""
        w = dense.detach().contiguous().to(device=device, dtype=torch.float16)
        packed, absmax = quantize_fp4(w, blocksize)
        state = QuantState(absmax, dense.shape, fp4_code().to(device), blocksize, dense.dtype)
        return cls(packed, False, state, blocksize, "fp4")


class LinearFP4(nn.Linear):
    """Drop-in for ``bitsandbytes.nn.LinearFP4``: dense until it reaches a GPU, FP4 afterwards."""

    def __init__(self, input_features: int, output_features: int, bias: bool = True,
                 compute_dtype: Optional[torch.dtype] = None, blocksize: int = 64, device=None):
        super().__init__(input_features, output_features, bias, device=device)
        self.compute_dtype = compute_dtype
        self.blocksize = blocksize
        self.weight = Params4bit(self.weight.data, False, None, blocksize, "fp4")

    def _apply(self, fn, recurse=True):
        w = self._parameters.pop("weight")
        try:
            super()._apply(fn, recurse)  # bias and buffers take the generic path
        finally:
            self._parameters["weight"] = w
        probe = fn(torch.empty(0, dtype=torch.float16, device=w.device))
        if getattr(w, "quant_state", None) is None:
            if probe.device.type == "cuda":
                self._parameters["weight"] = Params4bit.quantized_from(w.data, probe.device, self.blocksize)
            else:
                self._parameters["weight"] = Params4bit(fn(w.data), False, None, self.blocksize, "fp4")
        elif probe.device != w.device:
            w.quant_state.to(probe.device)
            self._parameters["weight"] = Params4bit(w.data.to(probe.device), False, w.quant_state, w.blocksize, w.quant_type)
        return self

    def forward(self, x: torch.Tensor) -> torch.Tensor:  # dense fallback is only for un-quantised (CPU) use
        if getattr(self.weight, "quant_state", None) is None:
            return nn.functional.linear(x, self.weight.to(x.dtype), None if self.bias is None else self.bias.to(x.dtype))
        raise RuntimeError("LinearFP4 holds packed FP4 weights; wrap it in TorchFP4Linear to run it")


Linear4bit = LinearFP4

# what counts as "a 4-bit linear" / "4-bit params" for isinstance checks
FP4_LINEAR_TYPES: Tuple[type, ...] = (LinearFP4,) + tuple(t for t in (_BnbLinear4bit, _BnbLinearFP4) if isinstance(t, type))
FP4_PARAM_TYPES: Tuple[type, ...] = (Params4bit,) + tuple(t for t in (_BnbParams4bit,) if isinstance(t, type))


def is_fp4_params(p) -> bool:
    """``Params4bit`` of either provenance, or anything that quacks like one."""
    return isinstance(p, FP4_PARAM_TYPES) or (isinstance(p, torch.Tensor) and hasattr(p, "quant_state"))
