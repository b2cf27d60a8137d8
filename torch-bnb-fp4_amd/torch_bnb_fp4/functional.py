"""Functional wrappers, 1:1 over the extension ops.

Same names, argument order and defaults as the reference's wrappers
(torch_bnb_fp4/__init__.py:87-337); every call lands in a hand-written gfx950 kernel through
``torch_bnb_fp4_ext`` -> ``libtorch_bnb_fp4_hip.so``.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

from ._ext import ext
from .dtypes import ScalarType


@torch.no_grad()
def dequantize_fp4(qweight: torch.Tensor, absmax: torch.Tensor, blocksize: int, M: int, N: int,
                   dtype: torch.dtype = torch.float16) -> torch.Tensor:
    """FP4 -> ``dtype`` [M, N] with the tree constants (reference :87-121)."""
    return ext.dequantize_fp4(qweight, absmax, blocksize, M, N, ScalarType.from_torch_dtype(dtype).value)


@torch.no_grad()
def dequantize_fp4_qtype(qweight: torch.Tensor, absmax: torch.Tensor, blocksize: int, M: int, N: int,
                         dtype=ScalarType.bfloat16.value) -> torch.Tensor:
    """Same, taking the extension enum directly (reference :298-337)."""
    return ext.dequantize_fp4(qweight, absmax, blocksize, M, N, dtype)


@torch.no_grad()
def dequantize_fp4_codebook_invoke_qtype(qweight: torch.Tensor, absmax: torch.Tensor, code: torch.Tensor, blocksize: int,
                                         M: int, N: int, numel: int, qtype) -> torch.Tensor:
    """FP4 -> T [M, N] with the codebook table; ``qtype`` is the extension enum (reference :124-168)."""
    return ext.dequantize_fp4_codebook(qweight, absmax, code, M, N, blocksize, numel, qtype)


@torch.no_grad()
def dequantize_fp4_codebook_invoke(qweight: torch.Tensor, absmax: torch.Tensor, code: torch.Tensor, blocksize: int,
                                   M: int, N: int, numel: int, qtype: torch.dtype) -> torch.Tensor:
    """Same, taking a torch dtype (reference :171-217)."""
    return ext.dequantize_fp4_codebook(qweight, absmax, code, M, N, blocksize, numel,
                                       ScalarType.from_torch_dtype(qtype).value)


@torch.no_grad()
def gemm_4bit_inference(A: torch.Tensor, B: torch.Tensor, absmax: torch.Tensor, code: torch.Tensor, blocksize: int,
                        dtype: torch.dtype = torch.float16, Bshape: Optional[Sequence[int]] = None) -> torch.Tensor:
    """Fused batch-1 GEMV ``A[1,K] @ W[M,K]^T`` over packed FP4 ``B`` (reference :220-257)."""
    return ext.gemv_fp4(A, B, absmax, code, blocksize, ScalarType.from_torch_dtype(dtype).value, list(Bshape))


@torch.no_grad()
def gemm_4bit_inference_qtype(A: torch.Tensor, B: torch.Tensor, absmax: torch.Tensor, code: torch.Tensor, blocksize: int,
                              dtype=ScalarType.bfloat16.value, Bshape: Optional[List[int]] = None,
                              bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Same, taking the extension enum (reference :260-295).  ``bias`` (an extension of the
    reference signature) fuses the reference's separate ``out += bias`` into the kernel epilogue."""
    if bias is not None:
        return ext.gemv_fp4_bias(A, B, absmax, code, blocksize, dtype, list(Bshape), bias)
    return ext.gemv_fp4(A, B, absmax, code, blocksize, dtype, list(Bshape))


@torch.no_grad()
def quantize_fp4(W: torch.Tensor, blocksize: int = 64) -> Tuple[torch.Tensor, torch.Tensor]:
    """bitsandbytes-format blockwise FP4 quantisation on the GPU: ``(packed uint8[n/2, 1], absmax f32[n/bs])``.

    Stands in for ``bitsandbytes.functional.quantize_fp4`` (called by the reference at :775)."""
    return ext.quantize_fp4(W.contiguous(), blocksize)
