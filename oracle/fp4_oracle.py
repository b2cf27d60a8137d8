"""CPU oracle for the FP4 dequant / fused-GEMV hot path (numpy restatement).

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  The product path (``torch-bnb-fp4_amd/``) never does, and fails
loudly when its HIP extension is missing.

What it restates (reference = aredden/torch-bnb-fp4, paths relative to the
reference checkout):

* the two 16-entry code tables            csrc/dequant_fp4_optimized.cu:28-46 (codebook op, GEMV)
                                          csrc/dequant_fp4_optimized.cu:55-76 (tree op)
* nibble order / absmax index / tail      csrc/dequant_fp4_optimized.cu:107-121, 156-169
* f32 multiply then RNE convert           csrc/dequant_fp4_optimized.cu:78-87, 165-166
* GEMV row/byte/absmax indexing           csrc/gemv_fp4_optimized.cu:99-156
* GEMV numerics of the reference kernel   csrc/gemv_fp4_optimized.cu:87-95, 128-129, 146-152
* dispatch rules of the Linear shell      torch_bnb_fp4/__init__.py:560-618
* the FP4 blockwise *quantiser* lives in bitsandbytes (pinned ``<0.43`` by the
  reference's requirements.txt:1, not vendored, not installed here).  Its
  published algorithm (absmax per block, midpoint thresholds, nibble map) is
  restated in :func:`quantize_fp4`; call sites in the reference:
  torch_bnb_fp4/__init__.py:736-747, 775-777.

Pinning status
--------------
The reference cannot be compiled here (its .cu files need nvcc + CUB, neither
exists in the image) nor imported (``bitsandbytes`` and the built extension are
absent: ordinary ModuleNotFoundError).  The reference holds no test vectors.
The oracle is therefore pinned by what the reference *does* hold:

* the code-table literals of the .cu files (tests/test_oracle.py re-derives
  the hex constants below from the decimal literals with a real C compiler,
  via oracle/fp4_oracle.c), and
* the published acceptance statistic, mean|dense - fp4| in [0.045, 0.065] with
  the nine README values 0.049-0.057 (README.md:90-91,113-115,137-139,161-163;
  sanity_check.py:130-179), reproduced in tests/test_oracle.py.

The dequant/GEMV consumer side is fully specified by the .cu sources, so its
parity is pinned by construction from those sources.  The *quantiser*
(bitsandbytes) is "parity unpinned": no reference-held vector covers its exact
tie/threshold bits.
"""
from __future__ import annotations

import numpy as np

# ---------------------------------------------------------------------------
# Code tables, as IEEE-754 binary32 bit patterns.
# nibble layout: bit 3 = sign, bits 2..0 = magnitude index
# (csrc/dequant_fp4_optimized.cu:56-75).
# ---------------------------------------------------------------------------

# ``CODE_PARAM`` literals of csrc/dequant_fp4_optimized.cu:30-45 and
# csrc/gemv_fp4_optimized.cu:34-49, rounded to binary32 the way a C compiler
# rounds a decimal float literal.
CODEBOOK_MAG_BITS = (
    0x00000000,  # 0.00000f
    0x3BAAAAAA,  # 5.208333e-03f
    0x3F2AAAAB,  # 0.6666667f
    0x3F800000,  # 1.000000f
    0x3EAAAA9F,  # 0.333333f
    0x3F000000,  # 0.500000f
    0x3E2AAAAD,  # 0.1666667f
    0x3E800000,  # 0.250000f
)
# literals of ``dequantize_fp4_tree`` (csrc/dequant_fp4_optimized.cu:60-75);
# these coincide with k/12 rounded to binary32, i.e. bitsandbytes' own table.
TREE_MAG_BITS = (
    0x00000000,  # 0.00000000f
    0x3BAAAAAB,  # 5.208333333e-03f
    0x3F2AAAAB,  # 0.66666667f
    0x3F800000,  # 1.00000000f
    0x3EAAAAAB,  # 0.33333333f
    0x3F000000,  # 0.50000000f
    0x3E2AAAAB,  # 0.16666667f
    0x3E800000,  # 0.25000000f
)

TABLE_CODEBOOK = "codebook"
TABLE_TREE = "tree"


def _table_from_mag_bits(mag_bits) -> np.ndarray:
    bits = np.array(list(mag_bits) + [b | 0x80000000 for b in mag_bits], dtype=np.uint32)
    return bits.view(np.float32)


CODEBOOK_TABLE = _table_from_mag_bits(CODEBOOK_MAG_BITS)  # float32[16]
TREE_TABLE = _table_from_mag_bits(TREE_MAG_BITS)  # float32[16]


def table(name: str) -> np.ndarray:
    if name == TABLE_CODEBOOK:
        return CODEBOOK_TABLE
    if name == TABLE_TREE:
        return TREE_TABLE
    raise ValueError(f"unknown table {name!r}")


# 12 x |code| : the integer-ish FP4 magnitudes bitsandbytes documents
# (0, 0.0625, 8, 12, 4, 6, 2, 3).  All are exact in fp16 and bf16.
C12_MAG = np.array([0.0, 0.0625, 8.0, 12.0, 4.0, 6.0, 2.0, 3.0], dtype=np.float32)

# ---------------------------------------------------------------------------
# rounding helpers
# ---------------------------------------------------------------------------


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """binary32 -> bfloat16 bit patterns, round-to-nearest-even.

    Restates ``__float2bfloat16_rn`` (csrc/dequant_fp4_optimized.cu:79-81).
    NaN is kept NaN (quietened); it cannot arise from finite code x absmax.
    """
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    rounded = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    out = (rounded & 0xFFFF).astype(np.uint16)
    nan = np.isnan(x)
    if nan.any():
        out = np.where(nan, ((u >> 16) | 0x0040).astype(np.uint16), out)
    return out


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    return (np.ascontiguousarray(b, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def round_to_bf16(x: np.ndarray) -> np.ndarray:
    """float64/float32 -> nearest bf16 value (RNE, single rounding), as float32."""
    v = np.asarray(x, dtype=np.float64)
    m, e = np.frexp(v)  # v = m * 2**e, 0.5 <= |m| < 1
    e = np.maximum(e, -125)  # bf16 subnormals share the exponent of 2**-126
    ulp = np.ldexp(1.0, e - 8)
    r = np.round(v / ulp) * ulp
    r = np.where(np.isfinite(v), r, v)
    return r.astype(np.float32)


def round_to_f16(x: np.ndarray) -> np.ndarray:
    """float64/float32 -> nearest fp16 value (RNE, single rounding), as float32."""
    return np.asarray(x).astype(np.float16).astype(np.float32)


def round_to(dtype: str):
    return {"float32": lambda v: np.asarray(v).astype(np.float32), "float16": round_to_f16, "bfloat16": round_to_bf16}[dtype]


# ---------------------------------------------------------------------------
# dequant
# ---------------------------------------------------------------------------


def unpack_nibbles(packed: np.ndarray) -> np.ndarray:
    """uint8[nbytes] -> uint8[2*nbytes]; the HIGH nibble is the even element.

    csrc/dequant_fp4_optimized.cu:117-118,165-166; csrc/gemv_fp4_optimized.cu:128-129.
    """
    p = np.ascontiguousarray(packed, dtype=np.uint8).reshape(-1)
    out = np.empty(p.size * 2, dtype=np.uint8)
    out[0::2] = p >> 4
    out[1::2] = p & 0x0F
    return out


def absmax_index(n: int, blocksize: int) -> np.ndarray:
    """Index into ``absmax`` for each of the n output elements.

    The reference looks absmax up once per thread, for the first of the 8
    packed bytes (16 elements) the thread owns:
    ``absmax[(i + threadIdx.x*8) / (blocksize/2)]``
    (csrc/dequant_fp4_optimized.cu:110,159,177).  For blocksize % 16 == 0 this
    is ``element // blocksize``.
    """
    e = np.arange(n, dtype=np.int64)
    first_byte_of_thread = (e // 16) * 8
    return first_byte_of_thread // (blocksize // 2)


def dequantize_f32(packed, absmax, blocksize: int, n: int, table_name: str = TABLE_CODEBOOK) -> np.ndarray:
    """float32[n] = f32(code[nibble]) * f32(absmax[block]) - one f32 multiply.

    csrc/dequant_fp4_optimized.cu:165-166 (codebook) / :60-75,117-118 (tree:
    ``c*absmax*sign``; the sign multiply is exact, so it equals ``(+-c)*absmax``).
    """
    nib = unpack_nibbles(packed)[:n]
    t = table(table_name)
    am = np.ascontiguousarray(absmax, dtype=np.float32).reshape(-1)
    idx = absmax_index(n, blocksize)
    return (t[nib] * am[idx]).astype(np.float32)


def dequantize(packed, absmax, blocksize: int, n: int, dtype: str, table_name: str = TABLE_CODEBOOK) -> np.ndarray:
    """Dequantise to ``dtype``; returns float32 / float16 arrays, or uint16 bit
    patterns for ``"bfloat16"`` (numpy has no bf16).

    csrc/dequant_fp4_optimized.cu:78-87: identity / __float2half_rn /
    __float2bfloat16_rn applied to the f32 product.
    """
    f = dequantize_f32(packed, absmax, blocksize, n, table_name)
    if dtype == "float32":
        return f
    if dtype == "float16":
        return f.astype(np.float16)  # IEEE RNE, subnormals kept
    if dtype == "bfloat16":
        return f32_to_bf16_bits(f)
    raise ValueError(dtype)


# ---------------------------------------------------------------------------
# GEMV
# ---------------------------------------------------------------------------


def gemv_exact(x, packed, absmax, M: int, K: int, blocksize: int, table_name: str = TABLE_CODEBOOK) -> np.ndarray:
    """float64[M] = x @ dequant_f32(W)^T accumulated in float64.

    Row r of W starts at packed byte r*K/2 and its absmax at (r*K)/blocksize
    (csrc/gemv_fp4_optimized.cu:100-103,108).  ``x`` holds the activation's
    exact values (already rounded to its storage dtype).
    """
    w = dequantize_f32(packed, absmax, blocksize, M * K, table_name).reshape(M, K).astype(np.float64)
    return w @ np.asarray(x, dtype=np.float64).reshape(K)


def gemv_reference_emulated(x, packed, absmax, M: int, K: int, blocksize: int, dtype: str, fused: bool = False) -> np.ndarray:
    """Emulates the *numerics* of the reference GEMV kernels; returns float32[M]
    holding values of ``dtype``.

    half / bf16 kernel (csrc/gemv_fp4_optimized.cu:60-157): the code table and
    the absmax are rounded to T (:92-95,103), ``quant_map[n]*absmax`` is a T
    multiply (:128-129), the per-lane accumulator is T (:87,146-148); only the
    32-lane reduction is float (:80,152).  fp32 kernel (:159-259): all float.
    Lane l of the 32-lane warp owns elements [32l, 32l+32) of every 1024-element
    K step (:99).  ``fused`` selects a single-rounding multiply-add for the
    per-lane accumulate (what a contracting compiler emits) instead of a
    rounded multiply followed by a rounded add.
    """
    assert K % 32 == 0, "the reference's GEMV gate implies K % 32 == 0 (torch_bnb_fp4/__init__.py:593)"
    rt = round_to(dtype)
    nib = unpack_nibbles(packed)[: M * K].reshape(M, K)
    qm = rt(CODEBOOK_TABLE)  # quant_map[i] = T(code[i])
    am = np.ascontiguousarray(absmax, dtype=np.float32).reshape(-1)
    # absidx = (2*ldb*row + inner_idx) / blocksize with ldb = K/2, looked up
    # once per lane per 32-element chunk (:100-103)
    chunk_first = (np.arange(M)[:, None] * K + (np.arange(K // 32) * 32)[None, :]) // blocksize
    am_chunk = rt(am[chunk_first])  # [M, K/32]  (T(absmax))
    xv = rt(np.asarray(x, dtype=np.float64).reshape(K))
    nchunks = K // 32
    acc = np.zeros((M, 32), dtype=np.float64)  # per-lane local_C, values of T
    for c0 in range(0, nchunks, 32):  # one K step of 1024 elements
        lanes = min(32, nchunks - c0)
        sl = slice(c0, c0 + lanes)
        a_c = am_chunk[:, sl].astype(np.float64)  # [M, lanes]
        for j in range(32):
            col = (np.arange(c0, c0 + lanes) * 32 + j)
            wq = qm[nib[:, col]].astype(np.float64)  # [M, lanes]
            b = rt(wq * a_c).astype(np.float64)  # local_B = quant_map*absmax in T
            xa = xv[col].astype(np.float64)[None, :]
            if fused:
                acc[:, :lanes] = rt(xa * b + acc[:, :lanes])
            else:
                acc[:, :lanes] = rt(rt(xa * b).astype(np.float64) + acc[:, :lanes])
    # cub::WarpReduce<float>::Sum: shuffle-down tree in float (:152)
    red = acc.astype(np.float32)
    off = 1
    while off < 32:
        shifted = np.zeros_like(red)
        shifted[:, : 32 - off] = red[:, off:]
        red = (red + shifted).astype(np.float32)
        off *= 2
    return rt(red[:, 0])  # out[row] = T(local_C)  (:154-156)


# ---------------------------------------------------------------------------
# quantiser (bitsandbytes' algorithm; "parity unpinned", see module docstring)
# ---------------------------------------------------------------------------

# midpoints between neighbouring code magnitudes / 12, compared with strict '>'
QUANT_THRESHOLDS = np.array(
    [0.00260417, 0.0859375, 0.20833333, 0.29166667, 0.4166667, 0.583333, 0.8333333], dtype=np.float32
)
# magnitude rank (0 = smallest) -> 3-bit code:  0, .0052, .1667, .25, .3333, .5, .6667, 1
RANK_TO_CODE = np.array([0b000, 0b001, 0b110, 0b111, 0b100, 0b101, 0b010, 0b011], dtype=np.uint8)


def linear_epilogue(y, dtype: str, bias=None, residual=None) -> np.ndarray:
    """What follows a GEMV in the reference and in the model code, one rounded op at a time: ``out = T(y)``, then the
    in-place ``out += bias`` (torch_bnb_fp4/__init__.py:608-613), then the caller's ``h + out``; each is an f32 add
    rounded to T (what torch does for 16-bit tensors).  y may be float64 (the exact sum) or already T-valued."""
    rt = round_to(dtype)
    t = rt(y)
    if bias is not None:
        t = rt(t.astype(np.float32) + np.asarray(bias, np.float32))
    if residual is not None:
        t = rt(t.astype(np.float32) + np.asarray(residual, np.float32))
    return t


def silu_mul_epilogue(gate, up, dtype: str, residual=None) -> np.ndarray:
    """``silu(gate) * up`` as the model code runs it on T tensors (e.g. transformers' LlamaMLP ``act_fn(gate_proj(x)) *
    up_proj(x)``): gate and up are T-valued; silu = x / (1 + exp(-x)) evaluated in f32 and rounded to T; the product is
    rounded to T; an optional residual add is one more rounded op.  exp here is numpy's f32 exp, which may differ from the
    device library's by an ulp of f32 - visible in the T result only when that lands on a rounding boundary."""
    rt = round_to(dtype)
    g = rt(gate).astype(np.float32)
    u = rt(up).astype(np.float32)
    s = rt(g / (np.float32(1.0) + np.exp(-g, dtype=np.float32)))
    t = rt(s.astype(np.float32) * u)
    if residual is not None:
        t = rt(t.astype(np.float32) + np.asarray(residual, np.float32))
    return t


def quantize_fp4(w, blocksize: int = 64):
    """Blockwise FP4 quantisation of a flat float array.

    Returns ``(packed uint8[ceil(n/2)], absmax float32[ceil(n/blocksize)])``.
    Per block (flat, row-major): absmax = max|w|; x = w * (1/absmax) in f32;
    magnitude -> nearest code by the thresholds above; sign -> bit 3; the even
    element goes to the HIGH nibble.  An all-zero block has absmax 0 and
    encodes every element as 0 (1/0 = inf, 0*inf = NaN, every compare false).
    """
    w = np.ascontiguousarray(w, dtype=np.float32).reshape(-1)
    n = w.size
    nblocks = -(-n // blocksize)
    padded = np.zeros(nblocks * blocksize, dtype=np.float32)
    padded[:n] = w
    blk = padded.reshape(nblocks, blocksize)
    absmax = np.abs(blk).max(axis=1).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        inv = (np.float32(1.0) / absmax).astype(np.float32)
        xn = (blk * inv[:, None]).astype(np.float32)
    mag = np.abs(xn)
    rank = np.zeros(mag.shape, dtype=np.int64)
    for t in QUANT_THRESHOLDS:
        rank += mag > t  # NaN compares false -> rank 0
    code = RANK_TO_CODE[rank]
    code = code | np.where(xn < 0, np.uint8(8), np.uint8(0)).astype(np.uint8)
    code = code.reshape(-1)[:n]
    if n & 1:
        code = np.concatenate([code, np.zeros(1, np.uint8)])
    packed = ((code[0::2] << 4) | code[1::2]).astype(np.uint8)
    return packed, absmax


# ---------------------------------------------------------------------------
# dispatch rules of QuantData.forward (torch_bnb_fp4/__init__.py:560-618)
# ---------------------------------------------------------------------------


def expected_dispatch(shape, K: int, blocksize: int) -> str:
    """Which branch the reference's ``QuantData.forward`` takes for an input of
    ``shape`` (last dim K): "empty" | "gemv" | "qlinear"."""
    numel = int(np.prod(shape)) if len(shape) else 1
    if numel == 0:
        return "empty"  # :580-589
    if numel == shape[-1]:  # :592 single token, batch 1
        if shape[-1] % blocksize != 0:
            return "qlinear"  # :593-594
        if len(shape) in (2, 3):
            return "gemv"  # :603-613
        return "qlinear"  # :614-615
    return "qlinear"  # :616-617


# ---------------------------------------------------------------------------
# algorithmic byte counts
# ---------------------------------------------------------------------------


def dequant_bytes(M: int, K: int, blocksize: int, out_itemsize: int) -> int:
    n = M * K
    return n // 2 + 4 * (n // blocksize) + n * out_itemsize


def gemv_bytes(M: int, K: int, blocksize: int, itemsize: int) -> int:
    n = M * K
    return n // 2 + 4 * (n // blocksize) + K * itemsize + M * itemsize
