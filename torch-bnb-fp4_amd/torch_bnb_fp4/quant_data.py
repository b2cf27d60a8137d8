"""``QuantData``: the packed weight of one Linear plus the GEMV / dequant+GEMM dispatcher.

Counterpart of the reference's ``QuantData`` (torch_bnb_fp4/__init__.py:340-618); the dispatch
rules of ``forward`` are part of the parity contract and are reproduced branch for branch:

====================================  =========================================
input                                 path
====================================  =========================================
any dim of size 0                     empty tensor of the output shape (:580-589)
numel == last dim, K % blocksize != 0 ``qlinear``                        (:593-594)
numel == last dim, 2-D or 3-D         fused GEMV, bias added after       (:603-613)
numel == last dim, other ranks        ``qlinear``                        (:614-615)
everything else (batch or seq > 1)    ``qlinear`` = dequant + F.linear   (:616-617)
====================================  =========================================

Two extensions.  ``fuse_bias`` (ON by default) folds the post-GEMV ``out += bias`` into the kernel epilogue: the table above
still holds and the result is bit-identical (``T(T(sum) + bias)``), there is just one launch fewer; ``fuse_bias=False`` runs the
reference's two-step sequence literally.  ``small_batch_fused`` (OFF by default, because it changes the table's last row) sends
2..128 activation rows (f32 activations: 2..8, one f32 GEMV launch per row) to the fused small-batch kernels instead of dequant + GEMM.
"""
from __future__ import annotations

from math import prod
from typing import Optional, Tuple

import torch

from ._ext import ext
from .dtypes import ScalarType
from .functional import dequantize_fp4_codebook_invoke_qtype, dequantize_fp4_qtype


class QuantData:
    def __init__(self, A: torch.Tensor, state, shape: Tuple[int, int], original_lin=None,
                 bias: Optional[torch.Tensor] = None, use_codebook_dequant: Optional[bool] = True,
                 allow_reduced_precision_linear: Optional[bool] = False, fuse_bias: bool = True,
                 small_batch_fused: bool = False):
        self.use_codebook_dequant = use_codebook_dequant
        self.A = A
        self.absmax = state.absmax.float()
        self.blocksize = state.blocksize
        self.M, self.N = shape[0], shape[1]
        self.code = state.code.float()
        self.o_type = None
        self.qtype = None
        self.quant_state = state
        # same precedence as the reference (:387): the wrapped layer's bias wins over the argument
        self.bias = original_lin.bias if hasattr(original_lin, "bias") else bias
        self.original_lin = original_lin
        self.compute_dtype_set = False
        self.numel = prod(shape)
        # fuse the post-GEMV `out += bias` into the kernel epilogue (bit-identical, one launch fewer)
        self.fuse_bias = fuse_bias
        # opt-in: 2..128 activation rows go to the fused small-batch kernels instead of dequant + GEMM (the reference
        # always dequantises for batch > 1, :616-617; same result up to rounding, ~5x less HBM traffic)
        self.small_batch_fused = small_batch_fused
        # per-call constants of the decode path, built once (224 calls per token in a 7B model)
        self._B_t = A.t()
        self._shape_list = [int(shape[0]), int(shape[1])]
        if allow_reduced_precision_linear:
            self.qlinear = self._qlinear_low_precision_codebook if use_codebook_dequant else self._qlinear_low_precision_normal
        else:
            self.qlinear = self._dequant_linear
        self.dequantize = self._dequantize_codebook if use_codebook_dequant else self._dequantize_normal

    def rebind(self, A: torch.Tensor, absmax: torch.Tensor, code: torch.Tensor, bias: Optional[torch.Tensor]) -> None:
        """Point the dispatcher at new storage for the packed weight / scales / code / bias (after a device move or a
        ``load_state_dict`` of the owning module); the bias is cast to the compute dtype if that is already fixed."""
        self.A, self.absmax, self.code = A, absmax.float(), code.float()
        self._B_t = A.t()
        if bias is not None and self.compute_dtype_set:
            bias = bias.to(dtype=self.o_type)
        self.bias = bias
        qs = self.quant_state
        if hasattr(qs, "absmax"):
            qs.absmax, qs.code = self.absmax, self.code

    # -- compute dtype ---------------------------------------------------------------------------
    def set_compute_type(self, x: torch.Tensor) -> None:
        """First call fixes the compute dtype to the activation's and casts the bias to it (:403-421)."""
        self.o_type = x.dtype
        self.qtype = ScalarType.from_torch_dtype(x.dtype).value
        if self.bias is not None:
            self.bias = self.bias.to(dtype=self.o_type)
        self.compute_dtype_set = True

    # -- dequant ---------------------------------------------------------------------------------
    def _dequantize_codebook(self) -> torch.Tensor:
        return dequantize_fp4_codebook_invoke_qtype(self.A, self.absmax, self.code, self.blocksize, self.M, self.N,
                                                    self.numel, self.qtype)

    def _dequantize_normal(self) -> torch.Tensor:
        return dequantize_fp4_qtype(self.A, self.absmax, self.blocksize, self.M, self.N, self.qtype)

    def _dequant_linear(self, A: torch.Tensor) -> torch.Tensor:
        # Same maths as the reference's F.linear(A, self.dequantize(), self.bias) (:423-436), done by the
        # extension's qlinear* ops in one call: the dequant there keeps its output cache-resident for the GEMM
        # (plain stores; standalone dequantize() streams it out non-temporally), and one Python round trip is saved.
        if A.dtype != self.o_type:
            return torch.nn.functional.linear(A, self.dequantize(), self.bias)
        return self._qlinear_low_precision_codebook(A) if self.use_codebook_dequant else self._qlinear_low_precision_normal(A)

    # -- fused paths -----------------------------------------------------------------------------
    def _qgemv(self, A: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        # the reference hands the packed [numel/2, 1] tensor over transposed (:486); it stays contiguous
        if self._B_t.data_ptr() != self.A.data_ptr():  # the packed tensor was replaced (e.g. a device move)
            self._B_t = self.A.t()
        if bias is not None:
            return ext.gemv_fp4_bias(A, self._B_t, self.absmax, self.code, self.blocksize, self.qtype, self._shape_list, bias)
        return ext.gemv_fp4(A, self._B_t, self.absmax, self.code, self.blocksize, self.qtype, self._shape_list)

    def _qlinear_low_precision_normal(self, A: torch.Tensor) -> torch.Tensor:
        if self.bias is None:
            return ext.qlinear(A, self.A, self.absmax, self.M, self.N, self.blocksize)
        return ext.qlinear_bias(A, self.A, self.absmax, self.M, self.N, self.blocksize, self.bias)

    def _qlinear_low_precision_codebook(self, A: torch.Tensor) -> torch.Tensor:
        if self.bias is None:
            return ext.qlinear_codebook(A, self.A, self.absmax, self.code, self.M, self.N, self.blocksize)
        return ext.qlinear_codebook_bias(A, self.A, self.absmax, self.code, self.M, self.N, self.blocksize, self.bias)

    # -- dispatcher ------------------------------------------------------------------------------
    def forward(self, A: torch.Tensor) -> torch.Tensor:
        total = A.numel()
        K = A.shape[-1]
        # hot path first (224 calls per token in a 7B model; eager decode is host-bound): single token, 2-D or 3-D
        if total == K and total != 0 and self.compute_dtype_set and K % self.blocksize == 0:
            nd = A.ndim
            if nd == 2 or nd == 3:
                if not A.is_contiguous():
                    A = A.contiguous()
                bias = self.bias
                if nd == 2:
                    out = self._qgemv(A, bias if self.fuse_bias else None)
                else:
                    out = self._qgemv(A.view(-1, K), bias if self.fuse_bias else None).view(A.shape[0], 1, -1)
                if bias is not None and not self.fuse_bias:
                    out += bias
                return out
        if total == 0:
            w_shape = self.quant_state.shape
            tail = w_shape[1:] if K == w_shape[0] else w_shape[:1]
            return torch.empty(A.shape[:-1] + tail, dtype=A.dtype, device=A.device)
        if not self.compute_dtype_set:
            self.set_compute_type(A)
            return self.forward(A)
        # everything that is not a 2-D / 3-D single token with K % blocksize == 0 (:593-594, :614-617)
        rows = total // K
        if self.small_batch_fused and 2 <= rows <= 128 and (
                (A.dtype in (torch.float16, torch.bfloat16)
                 and ((self.blocksize == 64 and K % 64 == 0) or (rows <= 8 and K % self.blocksize == 0 and K % 32 == 0 and K <= 4096)))
                or (A.dtype == torch.float32 and rows <= 8 and K % self.blocksize == 0)):  # f32: one f32 GEMV per row, up to 8 rows
            return ext.gemm_small_fp4(A.contiguous(), self.A.t(), self.absmax, self.blocksize, self._shape_list, self.bias)
        return self.qlinear(A)
