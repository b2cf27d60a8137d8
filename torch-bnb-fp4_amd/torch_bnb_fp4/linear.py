"""``TorchFP4Linear``: nn.Module shell around :class:`QuantData`.

Counterpart of the reference's class of the same name (torch_bnb_fp4/__init__.py:621-714).
Unlike the reference, which hides the wrapped layer in a python list so that nothing is
registered (:644), the packed weight, scales and bias are registered buffers here, so
``state_dict()`` round-trips and ``.to(device)`` moves the layer.
"""
from __future__ import annotations

import torch
from torch import nn

from .nn import is_fp4_params
from .quant_data import QuantData


class TorchFP4Linear(nn.Module):
    def __init__(self, lin, use_codebook_dequant: bool = True, name: str = ""):
        super().__init__()
        self.lin = [lin]  # kept out of module registration, like the reference
        self.in_features = lin.in_features
        self.out_features = lin.out_features
        self.use_codebook_dequant = use_codebook_dequant
        self.name = name
        w = lin.weight
        if not is_fp4_params(w):
            raise ValueError("Linear is not a bnb linear and is not quantized, and I have no idea what to do with that rn.")
        if getattr(w, "quant_state", None) is None or w.device.type != "cuda" or w.data.dtype != torch.uint8:
            raise ValueError(
                f"Linear weights are not quantized, and I have no idea what to do with that rn. Weights are {w.data.dtype}")
        self.quant_data = QuantData(w.data, w.quant_state, w.quant_state.shape, bias=lin.bias, original_lin=lin,
                                    use_codebook_dequant=self.use_codebook_dequant)
        self.register_buffer("qweight", self.quant_data.A, persistent=True)
        self.register_buffer("absmax", self.quant_data.absmax, persistent=True)
        self.register_buffer("code", self.quant_data.code, persistent=True)
        self.register_buffer("bias", None if self.quant_data.bias is None else self.quant_data.bias.detach(), persistent=True)

    def _sync_from_buffers(self) -> None:
        """The registered buffers are the single source of truth: the dispatcher (and the wrapped layer kept in ``self.lin``)
        are re-pointed at them, so nothing keeps a stale or old-device copy alive."""
        self.quant_data.rebind(self.qweight, self.absmax, self.code, self.bias)
        self._buffers["absmax"], self._buffers["code"] = self.quant_data.absmax, self.quant_data.code
        if self.quant_data.bias is not None:
            self._buffers["bias"] = self.quant_data.bias
        lin = self.lin[0]
        w = getattr(lin, "weight", None)
        if w is not None and getattr(w, "quant_state", None) is not None:
            w.data = self.qweight
            w.quant_state.absmax, w.quant_state.code = self.absmax, self.code
        if getattr(lin, "bias", None) is not None and self.bias is not None and lin.bias.device != self.bias.device:
            lin.bias.data = lin.bias.data.to(self.bias.device)

    def _apply(self, fn, recurse=True):
        # only device moves are honoured: the packed bytes and f32 scales never change dtype
        probe = fn(torch.empty(0, dtype=torch.float16, device=self.qweight.device))
        if probe.device != self.qweight.device:
            mv = lambda t: None if t is None else t.to(probe.device)
            for name in ("qweight", "absmax", "code", "bias"):
                self._buffers[name] = mv(self._buffers[name])
            self._sync_from_buffers()
        return self

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self._sync_from_buffers()  # load_state_dict copies into the buffers in place; the bias may need its cast redone

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        qd = self.quant_data
        if not qd.compute_dtype_set and x.numel():
            qd.set_compute_type(x)  # casts the bias to the activation dtype (reference :417-421) ...
            if qd.bias is not None:
                self._buffers["bias"] = qd.bias  # ... and the registered buffer follows, so state_dict() shows what the kernels use
        return qd.forward(x)

    def __repr__(self) -> str:
        dt = getattr(getattr(self, "quant_data", None), "o_type", None)
        return (f"TorchFP4Linear(in_features={self.in_features}, out_features={self.out_features}, "
                f"bias={self.lin[0].bias is not None}" + (f", dtype={dt})" if hasattr(self, "quant_data") else ")"))

    @classmethod
    def fuse(cls, layers, name: str = "") -> "TorchFP4Linear":
        """One layer computing ``cat([l(x) for l in layers], -1)`` in a single launch (fused QKV / gate-up projections):
        rows of an FP4 weight are independent and ``in_features % blocksize == 0``, so the packed bytes, scales and
        biases are simply concatenated.  Bit-identical to the separate layers on the GEMV path."""
        from .nn import LinearFP4, Params4bit, QuantState
        from .parallel import concat_rows

        qds = [l.quant_data for l in layers]
        bs = qds[0].blocksize
        if any(q.blocksize != bs or q.N != qds[0].N for q in qds):
            raise ValueError("fuse() needs layers with the same in_features and blocksize")
        if len({q.bias is None for q in qds}) != 1:
            raise ValueError("fuse() needs either every layer or no layer to carry a bias")
        packed, absmax, (M, K) = concat_rows([(q.A, q.absmax, (q.M, q.N)) for q in qds], bs)
        shell = LinearFP4(K, M, bias=qds[0].bias is not None, device="meta")
        state = QuantState(absmax, (M, K), qds[0].code, bs, qds[0].quant_state.dtype)
        shell._parameters["weight"] = Params4bit(packed, False, state, bs, "fp4")
        if qds[0].bias is not None:
            shell._parameters["bias"] = nn.Parameter(torch.cat([q.bias.reshape(-1) for q in qds]), requires_grad=False)
        return cls(shell, use_codebook_dequant=layers[0].use_codebook_dequant, name=name)

    @classmethod
    def from_linear(cls, linear, use_codebook_dequant: bool = False, name: str = "") -> "TorchFP4Linear":
        """Wrap an already-quantised FP4 layer (note the reference's default of ``False`` here, :699)."""
        return cls(linear, use_codebook_dequant=use_codebook_dequant, name=name)
