"""Loads the native extension ``torch_bnb_fp4_ext`` (built in-tree by ``build.py``).

There is deliberately no fallback: if the extension is missing the import fails, loudly, so a
GPU box can never run this package on a silent eager/CPU path.
"""
from __future__ import annotations

import importlib
import os
import sys

import torch  # noqa: F401  (must be imported first: the extension resolves libamdhip64 / libc10 through it)

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG_DIR)


def _load():
    if _ROOT not in sys.path:
        sys.path.insert(0, _ROOT)
    try:
        return importlib.import_module("torch_bnb_fp4_ext")
    except ImportError as exc:  # pragma: no cover - exercised only on a broken install
        raise ImportError(
            "torch_bnb_fp4: the HIP extension 'torch_bnb_fp4_ext' is not built or cannot be loaded "
            f"({exc}). Build it with `python {os.path.join(_ROOT, 'build.py')}` (needs hipcc, targets gfx950). "
            "There is no CPU or eager fallback."
        ) from exc


ext = _load()
HIP_LIBRARY_PATH = os.path.join(_PKG_DIR, "lib", "libtorch_bnb_fp4_hip.so")
