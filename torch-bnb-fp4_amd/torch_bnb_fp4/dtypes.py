"""Python-side mirror of the extension's ``ScalarType`` enum.

Counterpart of ``ScalarType`` in the reference (torch_bnb_fp4/__init__.py:22-84): members carry
the extension's enum value so ``.value`` can be handed straight to the ops.
"""
from __future__ import annotations

from enum import Enum

import torch

from ._ext import ext

_TORCH = {"bfloat16": torch.bfloat16, "float16": torch.float16, "float32": torch.float32}


class ScalarType(Enum):
    bfloat16 = ext.ScalarType.bfloat16
    float16 = ext.ScalarType.float16
    float32 = ext.ScalarType.float32

    @classmethod
    def from_torch_dtype(cls, dtype: torch.dtype) -> "ScalarType":
        for name, td in _TORCH.items():
            if dtype == td:
                return cls[name]
        raise ValueError(f"Unsupported dtype {dtype}")

    @classmethod
    def from_str(cls, dtype: str) -> "ScalarType":
        if dtype in _TORCH:
            return cls[dtype]
        raise ValueError(f"Unsupported dtype {dtype}")

    @property
    def torch_dtype(self) -> torch.dtype:
        # the reference's property compares against non-existent members and always raises
        # (torch_bnb_fp4/__init__.py:75-84); this one works
        return _TORCH[self.name]
