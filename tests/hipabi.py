"""ctypes binding of include/torch_bnb_fp4_hip.h for the tests: the GPU parity tests call the HIP
kernels through the C ABI, with torch used only to own device memory and the stream."""
from __future__ import annotations

import ctypes
import os
import re

import torch  # noqa: F401  (first: the library resolves libamdhip64 through torch's copy)

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "include", "torch_bnb_fp4_hip.h")
LIB_PATH = os.path.join(REPO, "torch-bnb-fp4_amd", "torch_bnb_fp4", "lib", "libtorch_bnb_fp4_hip.so")

F16, F32, BF16 = 0, 1, 2
TABLE_CODEBOOK, TABLE_TREE = 0, 1
OK, ERR_INVALID, ERR_UNSUPPORTED, ERR_LAUNCH = 0, 1, 2, 3
DT = {torch.float16: F16, torch.float32: F32, torch.bfloat16: BF16}


def declared_symbols():
    """Every function the header declares (FP4_HIP_API <ret> name(...))."""
    text = open(HEADER).read()
    return re.findall(r"FP4_HIP_API\s+[\w\s\*]+?\b(fp4_hip_\w+)\s*\(", text)


_lib = None


def lib():
    global _lib
    if _lib is None:
        l = ctypes.CDLL(LIB_PATH)
        vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
        l.fp4_hip_abi_version.restype = i32
        l.fp4_hip_last_error.restype = ctypes.c_char_p
        l.fp4_hip_code_table.argtypes = [i32, vp]
        l.fp4_hip_dequantize_blockwise.argtypes = [vp, vp, vp, i32, i64, i32, i32, i32, vp]
        l.fp4_hip_gemv.argtypes = [vp, vp, vp, vp, vp, i64, i64, i32, i32, vp]
        l.fp4_hip_gemv_partial.argtypes = [vp, vp, vp, vp, i64, i64, i32, i32, vp]
        l.fp4_hip_gemv_fused.argtypes = [vp, vp, vp, vp, vp, vp, i64, i64, i32, i32, i32, vp]
        l.fp4_hip_gemv_fused.restype = i32
        l.fp4_hip_gemm_small_fused.argtypes = [vp, vp, vp, vp, vp, vp, i64, i64, i64, i32, i32, i32, vp]
        l.fp4_hip_gemm_small_fused.restype = i32
        l.fp4_hip_gemm_small.argtypes = [vp, vp, vp, vp, vp, i64, i64, i64, i32, i32, vp]
        l.fp4_hip_gemm_small.restype = i32
        l.fp4_hip_quantize_blockwise.argtypes = [vp, i32, vp, vp, i64, i32, vp]
        l.fp4_hip_set_variant.argtypes = [ctypes.c_char_p, i32]
        for f in (l.fp4_hip_code_table, l.fp4_hip_dequantize_blockwise, l.fp4_hip_gemv, l.fp4_hip_gemv_partial, l.fp4_hip_quantize_blockwise,
                  l.fp4_hip_set_variant):
            f.restype = i32
        _lib = l
    return _lib


def last_error() -> str:
    return lib().fp4_hip_last_error().decode()


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


AUTO, KEEP_CACHED, STREAM = 0, 1, 2


def dequantize(packed: torch.Tensor, absmax: torch.Tensor, blocksize: int, n: int, dtype: torch.dtype, table: int = TABLE_CODEBOOK,
               out: torch.Tensor | None = None, flags: int = AUTO) -> torch.Tensor:
    if out is None:
        out = torch.empty(n, dtype=dtype, device=packed.device)
    rc = lib().fp4_hip_dequantize_blockwise(_ptr(packed), _ptr(absmax), _ptr(out), blocksize, n, DT[dtype], table, flags, _stream())
    assert rc == OK, (rc, last_error())
    return out


def gemv(x: torch.Tensor, packed: torch.Tensor, absmax: torch.Tensor, M: int, K: int, blocksize: int,
         bias: torch.Tensor | None = None) -> torch.Tensor:
    out = torch.empty(M, dtype=x.dtype, device=x.device)
    rc = lib().fp4_hip_gemv(_ptr(x), _ptr(packed), _ptr(absmax), _ptr(bias), _ptr(out), M, K, blocksize, DT[x.dtype], _stream())
    assert rc == OK, (rc, last_error())
    return out


EPILOGUE_NONE, EPILOGUE_SILU_MUL_PAIRS = 0, 1


def gemv_fused(x: torch.Tensor, packed: torch.Tensor, absmax: torch.Tensor, M: int, K: int, blocksize: int,
               bias: torch.Tensor | None = None, residual: torch.Tensor | None = None, epilogue: int = EPILOGUE_NONE,
               out: torch.Tensor | None = None, expect_ok: bool = True):
    if out is None:
        out = torch.empty(M // 2 if epilogue == EPILOGUE_SILU_MUL_PAIRS else M, dtype=x.dtype, device=x.device)
    rc = lib().fp4_hip_gemv_fused(_ptr(x), _ptr(packed), _ptr(absmax), _ptr(bias), _ptr(residual), _ptr(out), M, K, blocksize,
                                  DT[x.dtype], epilogue, _stream())
    if expect_ok:
        assert rc == OK, (rc, last_error())
        return out
    return rc


def gemm_small(x: torch.Tensor, packed: torch.Tensor, absmax: torch.Tensor, M: int, K: int, blocksize: int,
               bias: torch.Tensor | None = None, expect_ok: bool = True):
    B = x.numel() // K
    out = torch.empty(B, M, dtype=x.dtype, device=x.device)
    rc = lib().fp4_hip_gemm_small(_ptr(x), _ptr(packed), _ptr(absmax), _ptr(bias), _ptr(out), B, M, K, blocksize, DT[x.dtype], _stream())
    if expect_ok is None:  # the caller looks at both
        return rc, out
    if expect_ok:
        assert rc == OK, (rc, last_error())
        return out
    return rc


def gemm_small_ws(x: torch.Tensor, packed: torch.Tensor, absmax: torch.Tensor, M: int, K: int, blocksize: int,
                  bias: torch.Tensor | None = None, residual: torch.Tensor | None = None, epilogue: int = EPILOGUE_NONE,
                  workspace: torch.Tensor | None = None):
    """fp4_hip_gemm_small_ws with a workspace of the size the library asks for (or the one given); returns (out, bytes asked)."""
    B = x.numel() // K
    l = lib()
    l.fp4_hip_gemm_small_ws_bytes.restype = ctypes.c_int64
    l.fp4_hip_gemm_small_ws_bytes.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int]
    want = l.fp4_hip_gemm_small_ws_bytes(B, M, K, blocksize, DT[x.dtype])
    if workspace is None and want > 0:
        workspace = torch.empty(want, dtype=torch.uint8, device=x.device)
    out = torch.empty(B, M // 2 if epilogue == EPILOGUE_SILU_MUL_PAIRS else M, dtype=x.dtype, device=x.device)
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
    l.fp4_hip_gemm_small_ws.argtypes = [vp, vp, vp, vp, vp, vp, i64, i64, i64, i32, i32, i32, vp, i64, vp]
    rc = l.fp4_hip_gemm_small_ws(_ptr(x), _ptr(packed), _ptr(absmax), _ptr(bias), _ptr(residual), _ptr(out), B, M, K, blocksize, DT[x.dtype],
                                 epilogue, _ptr(workspace), 0 if workspace is None else workspace.numel(), _stream())
    assert rc == OK, (rc, last_error())
    return out, want


def gemm_small_fused(x: torch.Tensor, packed: torch.Tensor, absmax: torch.Tensor, M: int, K: int, blocksize: int,
                     bias: torch.Tensor | None = None, residual: torch.Tensor | None = None, epilogue: int = EPILOGUE_NONE,
                     expect_ok: bool = True):
    B = x.numel() // K
    out = torch.empty(B, M // 2 if epilogue == EPILOGUE_SILU_MUL_PAIRS else M, dtype=x.dtype, device=x.device)
    rc = lib().fp4_hip_gemm_small_fused(_ptr(x), _ptr(packed), _ptr(absmax), _ptr(bias), _ptr(residual), _ptr(out), B, M, K, blocksize,
                                        DT[x.dtype], epilogue, _stream())
    if expect_ok:
        assert rc == OK, (rc, last_error())
        return out
    return rc


def gemv_partial(x: torch.Tensor, packed: torch.Tensor, absmax: torch.Tensor, M: int, K: int, blocksize: int) -> torch.Tensor:
    out = torch.empty(M, dtype=torch.float32, device=x.device)
    rc = lib().fp4_hip_gemv_partial(_ptr(x), _ptr(packed), _ptr(absmax), _ptr(out), M, K, blocksize, DT[x.dtype], _stream())
    assert rc == OK, (rc, last_error())
    return out


def quantize(w: torch.Tensor, blocksize: int = 64):
    n = w.numel()
    packed = torch.empty((n + 1) // 2, dtype=torch.uint8, device=w.device)
    absmax = torch.empty((n + blocksize - 1) // blocksize, dtype=torch.float32, device=w.device)
    rc = lib().fp4_hip_quantize_blockwise(_ptr(w), DT[w.dtype], _ptr(packed), _ptr(absmax), n, blocksize, _stream())
    assert rc == OK, (rc, last_error())
    return packed, absmax


def set_variant(kernel: str, variant: int) -> None:
    rc = lib().fp4_hip_set_variant(kernel.encode(), variant)
    assert rc == OK, (rc, last_error())
