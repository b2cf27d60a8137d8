"""A CPU test double for ``torch_bnb_fp4_ext``: same op names and signatures, computed with the
oracle, recording which op was called.  Lets the host-side dispatch logic be tested without a GPU."""
from __future__ import annotations

import numpy as np
import torch

from oracle import fp4_oracle as o

_NP = {torch.float16: "float16", torch.float32: "float32", torch.bfloat16: "bfloat16"}


def _to_torch(arr: np.ndarray, dtype: torch.dtype) -> torch.Tensor:
    if dtype == torch.bfloat16:
        return torch.from_numpy(arr.astype(np.int16)).view(torch.bfloat16)
    return torch.from_numpy(arr.copy())


class FakeExt:
    def __init__(self, real_ext):
        self.ScalarType = real_ext.ScalarType
        self._dt = {real_ext.ScalarType.float16: torch.float16, real_ext.ScalarType.float32: torch.float32,
                    real_ext.ScalarType.bfloat16: torch.bfloat16}
        self.calls = []

    def _deq(self, A, absmax, M, N, blocksize, n, dtype, table):
        out = o.dequantize(A.numpy().reshape(-1), absmax.numpy(), blocksize, n, _NP[dtype], table)
        full = torch.zeros(M * N, dtype=dtype)
        full[:n] = _to_torch(out, dtype)
        return full.view(M, N)

    def dequantize_fp4(self, A, absmax, blocksize, M, N, o_type):
        self.calls.append("dequantize_fp4")
        return self._deq(A, absmax, M, N, blocksize, M * N, self._dt[o_type], "tree")

    def dequantize_fp4_codebook(self, A, absmax, codebook, M, N, blocksize, n, dtype):
        self.calls.append("dequantize_fp4_codebook")
        return self._deq(A, absmax, M, N, blocksize, n, self._dt[dtype], "codebook")

    def _gemv(self, A, B, absmax, blocksize, dtype, Bshape, bias):
        M, K = Bshape
        assert A.is_contiguous() and A.numel() == K and A.dtype == self._dt[dtype]
        y = o.gemv_exact(A.float().numpy().reshape(-1), B.numpy().reshape(-1), absmax.numpy(), M, K, blocksize)
        out = torch.from_numpy(y).to(A.dtype)
        if bias is not None:
            out = out + bias
        return out.view(*A.shape[:-1], M).clone()

    def gemv_fp4(self, A, B, absmax, datatype, blocksize, dtype, Bshape):
        self.calls.append("gemv_fp4")
        return self._gemv(A, B, absmax, blocksize, dtype, Bshape, None)

    def gemv_fp4_bias(self, A, B, absmax, datatype, blocksize, dtype, Bshape, bias):
        self.calls.append("gemv_fp4_bias")
        return self._gemv(A, B, absmax, blocksize, dtype, Bshape, bias)

    def gemv_fp4_fused(self, A, B, absmax, blocksize, Bshape, bias, residual, epilogue):
        self.calls.append("gemv_fp4_fused")
        M, K = Bshape
        if epilogue == 1 and (A.dtype == torch.float32 or K > 16384):
            raise RuntimeError("fp4_hip_gemv_fused: the gate|up epilogue is not available for this shape")
        assert A.is_contiguous() and A.numel() == K
        y = o.gemv_exact(A.float().numpy().reshape(-1), B.numpy().reshape(-1), absmax.numpy(), M, K, blocksize)
        nb = None if bias is None else bias.float().numpy()
        nr = None if residual is None else residual.float().numpy().reshape(-1)
        if A.dtype == torch.float32:
            t = y.astype(np.float32) + (0 if nb is None else nb) + (0 if nr is None else nr)
        elif epilogue == 1:
            t = o.linear_epilogue(y, _NP[A.dtype], nb)
            t = o.silu_mul_epilogue(t[0::2], t[1::2], _NP[A.dtype], nr)
        else:
            t = o.linear_epilogue(y, _NP[A.dtype], nb, nr)
        return torch.from_numpy(np.asarray(t, np.float32)).to(A.dtype).view(*A.shape[:-1], -1)

    def gemm_small_fp4_fused(self, A, B, absmax, blocksize, Bshape, bias, residual, epilogue):
        self.calls.append("gemm_small_fp4_fused")
        M, K = Bshape
        w = o.dequantize_f32(B.numpy().reshape(-1), absmax.numpy(), blocksize, M * K).reshape(M, K).astype(np.float64)
        y = A.float().numpy().reshape(-1, K).astype(np.float64) @ w.T
        if bias is not None:
            y = y + bias.float().numpy().astype(np.float64)
        nr = None if residual is None else residual.float().numpy().reshape(y.shape[0], -1)
        if epilogue == 1:
            t = o.silu_mul_epilogue(y[:, 0::2], y[:, 1::2], _NP[A.dtype], nr)
        else:
            t = o.linear_epilogue(y, _NP[A.dtype], None, nr)
        return torch.from_numpy(np.asarray(t, np.float32)).to(A.dtype).view(*A.shape[:-1], -1)

    def gemm_small_fp4(self, A, B, absmax, blocksize, Bshape, bias):
        self.calls.append("gemm_small_fp4")
        M, K = Bshape
        w = torch.from_numpy(o.dequantize_f32(B.numpy().reshape(-1), absmax.numpy(), blocksize, M * K)).view(M, K)
        y = torch.nn.functional.linear(A.float(), w, None if bias is None else bias.float())
        return y.to(A.dtype)

    def gemv_fp4_partial(self, A, B, absmax, blocksize, Bshape):
        self.calls.append("gemv_fp4_partial")
        M, K = Bshape
        assert A.is_contiguous() and A.numel() == K
        y = o.gemv_exact(A.float().numpy().reshape(-1), B.numpy().reshape(-1), absmax.numpy(), M, K, blocksize)
        return torch.from_numpy(y).float().view(1, M)

    def _qlinear(self, name, A_in, A, absmax, M, N, blocksize, table, bias=None):
        self.calls.append(name)
        w = self._deq(A, absmax, M, N, blocksize, M * N, A_in.dtype, table)
        return torch.nn.functional.linear(A_in, w, bias)

    def qlinear(self, A_in, A, absmax, M, N, blocksize):
        return self._qlinear("qlinear", A_in, A, absmax, M, N, blocksize, "tree")

    def qlinear_bias(self, A_in, A, absmax, M, N, blocksize, bias):
        return self._qlinear("qlinear_bias", A_in, A, absmax, M, N, blocksize, "tree", bias)

    def qlinear_codebook(self, A_in, A, absmax, codebook, M, N, blocksize):
        return self._qlinear("qlinear_codebook", A_in, A, absmax, M, N, blocksize, "codebook")

    def qlinear_codebook_bias(self, A_in, A, absmax, codebook, M, N, blocksize, bias):
        return self._qlinear("qlinear_codebook_bias", A_in, A, absmax, M, N, blocksize, "codebook", bias)

    def quantize_fp4(self, W, blocksize):
        self.calls.append("quantize_fp4")
        packed, am = o.quantize_fp4(W.float().numpy().reshape(-1), blocksize)
        return torch.from_numpy(packed).view(-1, 1), torch.from_numpy(am)

    def code_table(self, name):
        return torch.from_numpy(o.table(name).copy())
