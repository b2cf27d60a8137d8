"""Decode-step epilogue fusion on top of the reference's Linear surface.

The reference's op surface stops at the Linear: a decoder layer then pays separate launches for ``silu(gate(h)) * up(h)``
and for each residual add - where the reference itself already pays one for the bias (torch_bnb_fp4/__init__.py:603-613).
At batch 1 every one of those launches is a ~1.5 us kernel boundary next to GEMVs of 3-9 us, so they are folded into the
GEMV's epilogue (``fp4_hip_gemv_fused``):

* :class:`FusedFP4Linear` ``(x, residual=None)`` -> ``residual + Linear(x)`` in one launch;
* :meth:`FusedFP4Linear.gate_up` interleaves the rows of a gate and an up projection (rows of an FP4 weight are
  independent, so this is a row permutation of bytes and scales done once) and returns a layer computing
  ``silu(gate(x)) * up(x)`` in one launch.

Every intermediate is rounded to the activation dtype exactly where the separate torch ops would round it, so the result
equals the unfused sequence bit for bit (``exp`` is the device library's, as in torch's silu).  2..128 activation rows
(batched decode) take the same epilogues on the small-batch kernels (``fp4_hip_gemm_small_fused``); larger inputs, or shapes
the fused kernels do not cover, run the unfused sequence through :class:`QuantData`.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch
from torch import nn

from ._ext import ext
from .nn import QuantState, fp4_code
from .quant_data import QuantData

EPILOGUE_NONE = 0
EPILOGUE_SILU_MUL_PAIRS = 1


def interleave_rows(a: Tuple[torch.Tensor, torch.Tensor], b: Tuple[torch.Tensor, torch.Tensor], shape: Sequence[int], blocksize: int
                    ) -> Tuple[torch.Tensor, torch.Tensor, Tuple[int, int]]:
    """Rows ``a0, b0, a1, b1, ...`` of two FP4 weights of the same ``[M, K]`` shape as one ``[2M, K]`` weight
    (``(packed, absmax)`` each).  ``K % blocksize == 0``, so a row is a contiguous run of bytes and of scales."""
    M, K = int(shape[0]), int(shape[1])
    if K % blocksize or K % 2:
        raise ValueError(f"interleave_rows needs in_features ({K}) divisible by the blocksize ({blocksize})")
    (pa, sa), (pb, sb) = a, b
    if pa.numel() != M * K // 2 or pb.numel() != M * K // 2 or sa.numel() != M * K // blocksize or sb.numel() != M * K // blocksize:
        raise ValueError("interleave_rows: both weights must have the given [M, K] shape")
    packed = torch.stack([pa.reshape(M, K // 2), pb.reshape(M, K // 2)], dim=1).reshape(-1, 1)
    absmax = torch.stack([sa.reshape(M, K // blocksize), sb.reshape(M, K // blocksize)], dim=1).reshape(-1)
    return packed, absmax, (2 * M, K)


def deinterleave_rows(packed: torch.Tensor, absmax: torch.Tensor, shape: Sequence[int], blocksize: int):
    """Inverse of :func:`interleave_rows`: the even rows and the odd rows of a ``[2M, K]`` FP4 weight as two ``[M, K]`` weights,
    ``((packed_a, absmax_a), (packed_b, absmax_b), (M, K))`` - what a checkpoint in bitsandbytes' layout stores for a gate and an
    up projection."""
    M2, K = int(shape[0]), int(shape[1])
    if M2 % 2 or K % blocksize or K % 2:
        raise ValueError(f"deinterleave_rows needs an even row count and in_features ({K}) divisible by the blocksize ({blocksize})")
    p = packed.reshape(M2 // 2, 2, K // 2)
    a = absmax.reshape(M2 // 2, 2, K // blocksize)
    return ((p[:, 0].reshape(-1, 1).contiguous(), a[:, 0].reshape(-1).contiguous()),
            (p[:, 1].reshape(-1, 1).contiguous(), a[:, 1].reshape(-1).contiguous()), (M2 // 2, K))


class FusedFP4Linear(nn.Module):
    """An FP4 Linear whose single-token path runs ``fp4_hip_gemv_fused``.

    ``epilogue == EPILOGUE_NONE``: ``forward(x, residual=None)`` = ``Linear(x) (+ residual)``.
    ``epilogue == EPILOGUE_SILU_MUL_PAIRS``: the weight holds interleaved gate / up rows, ``forward(x)`` =
    ``silu(gate(x)) * up(x)`` with ``out_features`` = half the weight's rows."""

    def __init__(self, quant_data: QuantData, epilogue: int = EPILOGUE_NONE):
        super().__init__()
        if epilogue not in (EPILOGUE_NONE, EPILOGUE_SILU_MUL_PAIRS):
            raise ValueError(f"unknown epilogue {epilogue}")
        if epilogue == EPILOGUE_SILU_MUL_PAIRS and quant_data.M % 2:
            raise ValueError("the gate|up epilogue needs an even number of weight rows")
        self.quant_data = quant_data
        self.epilogue = epilogue
        self.in_features = int(quant_data.N)
        self.out_features = int(quant_data.M) // (2 if epilogue == EPILOGUE_SILU_MUL_PAIRS else 1)
        self._fused_ok = True  # cleared the first time the kernel reports the shape as not covered
        self._small_ok = True
        self.register_buffer("qweight", quant_data.A, persistent=True)
        self.register_buffer("absmax", quant_data.absmax, persistent=True)
        self.register_buffer("bias", None if quant_data.bias is None else quant_data.bias.detach(), persistent=True)

    # -- constructors ------------------------------------------------------------------------------------------------
    @classmethod
    def from_packed(cls, packed, absmax, shape, blocksize: int = 64, bias: Optional[torch.Tensor] = None,
                    epilogue: int = EPILOGUE_NONE, dtype: torch.dtype = torch.float16) -> "FusedFP4Linear":
        """``dtype``: the dtype the weight was quantised from (checkpoint metadata only; the kernels follow the activation)."""
        state = QuantState(absmax, shape, fp4_code().to(packed.device), blocksize, dtype)
        return cls(QuantData(packed, state, state.shape, original_lin=None, bias=bias), epilogue)

    @classmethod
    def gate_up_from_packed(cls, gate, up, shape, blocksize: int = 64, gate_bias: Optional[torch.Tensor] = None,
                            up_bias: Optional[torch.Tensor] = None) -> "FusedFP4Linear":
        """``gate`` / ``up``: ``(packed, absmax)`` of two ``[M, K]`` projections -> one layer computing silu(gate(x)) * up(x)."""
        packed, absmax, full = interleave_rows(gate, up, shape, blocksize)
        if (gate_bias is None) != (up_bias is None):
            raise ValueError("gate and up must both carry a bias or neither")
        bias = None if gate_bias is None else torch.stack([gate_bias.reshape(-1), up_bias.reshape(-1)], dim=1).reshape(-1)
        return cls.from_packed(packed, absmax, full, blocksize, bias, EPILOGUE_SILU_MUL_PAIRS)

    @classmethod
    def from_linear(cls, layer) -> "FusedFP4Linear":
        """From a :class:`TorchFP4Linear` (shares its packed weight)."""
        qd = layer.quant_data
        return cls.from_packed(qd.A, qd.absmax, (qd.M, qd.N), qd.blocksize, qd.bias, dtype=getattr(qd.quant_state, "dtype", torch.float16))

    @classmethod
    def gate_up(cls, gate_layer, up_layer) -> "FusedFP4Linear":
        """From the gate and up :class:`TorchFP4Linear` of a gated MLP."""
        g, u = gate_layer.quant_data, up_layer.quant_data
        if (g.M, g.N, g.blocksize) != (u.M, u.N, u.blocksize):
            raise ValueError("gate_up() needs two projections of the same shape and blocksize")
        return cls.gate_up_from_packed((g.A, g.absmax), (u.A, u.absmax), (g.M, g.N), g.blocksize, g.bias, u.bias)

    def _apply(self, fn, recurse=True):
        # Only device moves are honoured, exactly as in TorchFP4Linear._apply: the packed bytes and the f32 scales never
        # change dtype.  (nn.Module._apply would run model.half() / .to(torch.bfloat16) over the registered buffers and
        # round absmax to 16 bits - silently degraded scales in the running layer and in state_dict().)
        probe = fn(torch.empty(0, dtype=torch.float16, device=self.qweight.device))
        if probe.device != self.qweight.device:
            mv = lambda t: None if t is None else t.to(probe.device)
            for name in ("qweight", "absmax", "bias"):
                self._buffers[name] = mv(self._buffers[name])
            qd = self.quant_data
            qd.rebind(self.qweight, self.absmax, qd.code.to(probe.device), self.bias)
            self._buffers["absmax"] = qd.absmax
            if qd.bias is not None:
                self._buffers["bias"] = qd.bias
        return self

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self.quant_data.rebind(self.qweight, self.absmax, self.quant_data.code, self.bias)

    # -- forward -----------------------------------------------------------------------------------------------------
    def _unfused(self, x: torch.Tensor, residual: Optional[torch.Tensor]) -> torch.Tensor:
        y = self.quant_data.forward(x)
        if self.epilogue == EPILOGUE_SILU_MUL_PAIRS:
            y = nn.functional.silu(y[..., 0::2]) * y[..., 1::2]
        return y if residual is None else y + residual

    def forward(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        qd = self.quant_data
        K = x.shape[-1]
        if not qd.compute_dtype_set and x.numel():
            qd.set_compute_type(x)
            if qd.bias is not None:
                self._buffers["bias"] = qd.bias  # the buffer follows the cast to the compute dtype
        if (self._fused_ok and x.numel() == K and K == self.in_features and x.ndim in (2, 3) and K % qd.blocksize == 0
                and x.dtype == qd.o_type):
            if not x.is_contiguous():
                x = x.contiguous()
            try:
                return ext.gemv_fp4_fused(x, qd._B_t, qd.absmax, qd.blocksize, qd._shape_list, qd.bias, residual, self.epilogue)
            except RuntimeError as exc:
                if "not available" not in str(exc):
                    raise
                self._fused_ok = False  # shape outside the fused kernel's coverage: unfused sequence from now on
        rows = x.numel() // K if K else 0
        if (self._small_ok and 2 <= rows <= 128 and K == self.in_features and x.dtype == qd.o_type and x.dtype in (torch.float16, torch.bfloat16)
                and ((qd.blocksize == 64 and K % 64 == 0) or (rows <= 8 and K % qd.blocksize == 0 and K % 32 == 0 and K <= 4096))):
            try:  # batched decode: the same epilogues on the small-batch kernels
                return ext.gemm_small_fp4_fused(x.contiguous(), qd._B_t, qd.absmax, qd.blocksize, qd._shape_list, qd.bias, residual,
                                                self.epilogue)
            except RuntimeError as exc:
                if "not covered" not in str(exc):
                    raise
                self._small_ok = False
        return self._unfused(x, residual)

    def extra_repr(self) -> str:
        kind = "silu(gate)*up" if self.epilogue == EPILOGUE_SILU_MUL_PAIRS else "linear(+residual)"
        return f"in_features={self.in_features}, out_features={self.out_features}, epilogue={kind}"
