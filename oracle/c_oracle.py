"""ctypes loader for oracle/libfp4_oracle.so (the C restatement).  Test infrastructure only."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfp4_oracle.so")

F16, F32, BF16 = 0, 1, 2
TABLE_CODEBOOK, TABLE_TREE = 0, 1
_DT = {"float16": F16, "float32": F32, "bfloat16": BF16}
_TB = {"codebook": TABLE_CODEBOOK, "tree": TABLE_TREE}


def build(force: bool = False) -> str:
    override = os.environ.get("FP4_ORACLE_LIB")  # e.g. the sanitizer build (make -C oracle asan); never set in normal runs
    if override:
        return override
    src = os.path.join(_HERE, "fp4_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libfp4_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        l = ctypes.CDLL(build())
        vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
        l.fp4_oracle_table.argtypes = [i32, vp]
        l.fp4_oracle_dequant.argtypes = [vp, vp, vp, i32, i64, i32, i32]
        l.fp4_oracle_gemv_f64.argtypes = [vp, vp, vp, vp, i64, i64, i32, i32]
        l.fp4_oracle_quantize.argtypes = [vp, vp, vp, i64, i32]
        for f in (l.fp4_oracle_table, l.fp4_oracle_dequant, l.fp4_oracle_gemv_f64, l.fp4_oracle_quantize):
            f.restype = None
        _lib = l
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def table(name: str) -> np.ndarray:
    out = np.empty(16, np.float32)
    lib().fp4_oracle_table(_TB[name], _p(out))
    return out


def dequantize(packed, absmax, blocksize: int, n: int, dtype: str, table_name: str = "codebook") -> np.ndarray:
    packed = np.ascontiguousarray(packed, np.uint8).reshape(-1)
    absmax = np.ascontiguousarray(absmax, np.float32).reshape(-1)
    out = np.empty(n, np.float32 if dtype == "float32" else np.uint16)
    lib().fp4_oracle_dequant(_p(packed), _p(absmax), _p(out), blocksize, n, _DT[dtype], _TB[table_name])
    return out.view(np.float16) if dtype == "float16" else out


def gemv_f64(x, packed, absmax, M: int, K: int, blocksize: int, table_name: str = "codebook") -> np.ndarray:
    x = np.ascontiguousarray(x, np.float64).reshape(-1)
    packed = np.ascontiguousarray(packed, np.uint8).reshape(-1)
    absmax = np.ascontiguousarray(absmax, np.float32).reshape(-1)
    out = np.empty(M, np.float64)
    lib().fp4_oracle_gemv_f64(_p(x), _p(packed), _p(absmax), _p(out), M, K, blocksize, _TB[table_name])
    return out


def quantize(w, blocksize: int = 64):
    w = np.ascontiguousarray(w, np.float32).reshape(-1)
    n = w.size
    packed = np.empty((n + 1) // 2, np.uint8)
    absmax = np.empty((n + blocksize - 1) // blocksize, np.float32)
    lib().fp4_oracle_quantize(_p(w), _p(packed), _p(absmax), n, blocksize)
    return packed, absmax
