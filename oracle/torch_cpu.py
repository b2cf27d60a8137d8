"""Pure-torch CPU dequant / GEMV: the host-core baseline BASELINE.md section 3 asks to be
timed next to the GPU path (no custom kernels, no bitsandbytes).  Test infrastructure only.

Arithmetic follows csrc/dequant_fp4_optimized.cu:165-166 of the reference:
out = T(code[nibble] * absmax[block]) with the high nibble first.
"""
from __future__ import annotations

import torch

from . import fp4_oracle as _o


def code_table(name: str = "codebook") -> torch.Tensor:
    return torch.from_numpy(_o.table(name).copy())


@torch.no_grad()
def dequantize(packed: torch.Tensor, absmax: torch.Tensor, M: int, K: int, blocksize: int, dtype: torch.dtype,
               table: torch.Tensor | None = None) -> torch.Tensor:
    t = code_table() if table is None else table
    p = packed.reshape(-1)
    nib = torch.stack([p >> 4, p & 15], 1).reshape(-1)[: M * K]
    out = t[nib.long()] * absmax.reshape(-1).repeat_interleave(blocksize)[: M * K]
    return out.to(dtype).view(M, K)


@torch.no_grad()
def gemv(x: torch.Tensor, packed: torch.Tensor, absmax: torch.Tensor, M: int, K: int, blocksize: int) -> torch.Tensor:
    w = dequantize(packed, absmax, M, K, blocksize, torch.float32)
    return x.float().reshape(1, K) @ w.t()
