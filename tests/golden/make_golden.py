"""Generates tests/golden/fp4_golden.npz from the numpy oracle (oracle/fp4_oracle.py).

The reference holds no test vectors of its own and can be neither compiled nor imported in
this image (see the oracle header), so these known-answer vectors are produced by the oracle,
whose tables are pinned to the reference's literals.  Run from the repo root:

    python tests/golden/make_golden.py

The file is data only (inputs + expected outputs); everything is seeded and deterministic.
"""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import fp4_oracle as o  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fp4_golden.npz")
g = {}

# (1) exhaustive byte table: every byte value, absmax = 1 -> the two tables themselves
allbytes = np.arange(256, dtype=np.uint8)
ones = np.ones(512 // 64, np.float32)
g["kat1_packed"] = allbytes
for tb in ("codebook", "tree"):
    g[f"kat1_{tb}_f32"] = o.dequantize(allbytes, ones, 64, 512, "float32", tb)

# (2) absmax-index KAT: nibble 3 (= 1.0) everywhere, absmax[b] = 2**((b % 8) - 4)
for bs in (64, 128, 32, 256):
    n = bs * 24
    packed = np.full(n // 2, 0x33, np.uint8)
    am = (2.0 ** ((np.arange(n // bs) % 8) - 4)).astype(np.float32)
    g[f"kat2_bs{bs}_packed"], g[f"kat2_bs{bs}_absmax"] = packed, am
    g[f"kat2_bs{bs}_f32"] = o.dequantize(packed, am, bs, n, "float32")

# (3) rounding KAT: every nibble against absmax values that land on bf16 / fp16 ties, fp16
# subnormals and a log-uniform random sweep; one quant block (64 elements = 4 x 16 nibbles) per absmax
rng = np.random.default_rng(1234)
special = np.array([1.0, 1 + 2.0**-8, 1 + 3 * 2.0**-8, 1 + 2.0**-11, 1 + 3 * 2.0**-11, 0.01, 0.0117, 2.0**-14, 3 * 2.0**-16,
                    2.0**-24, 6.1e-5, 65504.0, 7e4, 1e-30, 3.0e38, 0.0], np.float32)
sweep = np.exp(rng.uniform(np.log(1e-6), np.log(1e3), 2032)).astype(np.float32)
am3 = np.concatenate([special, sweep])
nibs = np.tile(np.arange(16, dtype=np.uint8), 4)  # 64 nibbles per block
blockbytes = ((nibs[0::2] << 4) | nibs[1::2]).astype(np.uint8)
packed3 = np.tile(blockbytes, am3.size)
g["kat3_packed"], g["kat3_absmax"] = packed3, am3
n3 = am3.size * 64
with np.errstate(over="ignore"):
    for tb in ("codebook", "tree"):
        g[f"kat3_{tb}_f32"] = o.dequantize(packed3, am3, 64, n3, "float32", tb)
        g[f"kat3_{tb}_f16"] = o.dequantize(packed3, am3, 64, n3, "float16", tb).view(np.uint16)
        g[f"kat3_{tb}_bf16"] = o.dequantize(packed3, am3, 64, n3, "bfloat16", tb)

# (4) tail KATs: n not a multiple of the tile, odd n, tiny n
for tag, n in (("a", 24 * 40), ("b", 65536 + 777), ("c", 1), ("d", 31), ("e", 64 * 2048 + 1)):
    w = rng.standard_normal(n).astype(np.float32)
    packed, am = o.quantize_fp4(w, 64)
    g[f"kat4{tag}_packed"], g[f"kat4{tag}_absmax"], g[f"kat4{tag}_n"] = packed, am, np.int64(n)
    h = hashlib.sha256()
    for dt in ("float32", "float16", "bfloat16"):
        h.update(np.ascontiguousarray(o.dequantize(packed, am, 64, n, dt)).tobytes())
    g[f"kat4{tag}_sha256"] = np.frombuffer(h.digest(), np.uint8)

# (5) config C1: seed-0 1024x1024 N(0,1) weight -> quantise -> dequant f32: digests + a 4 KiB slice
w = np.random.default_rng(0).standard_normal(1024 * 1024).astype(np.float32)
packed, am = o.quantize_fp4(w, 64)
out = o.dequantize(packed, am, 64, w.size, "float32")
g["c1_packed_sha256"] = np.frombuffer(hashlib.sha256(packed.tobytes()).digest(), np.uint8)
g["c1_absmax_sha256"] = np.frombuffer(hashlib.sha256(am.tobytes()).digest(), np.uint8)
g["c1_out_sha256"] = np.frombuffer(hashlib.sha256(out.tobytes()).digest(), np.uint8)
g["c1_out_slice"] = out[123456 : 123456 + 1024].copy()
g["c1_w_slice"] = w[123456 : 123456 + 1024].copy()

# (6) GEMV: small exact cases (float64 answers) incl. K with idle lanes and M not a multiple of 4
for tag, (M, K) in (("a", (66, 768)), ("b", (8, 4096)), ("c", (5, 64))):
    w = (rng.standard_normal(M * K) * 0.02).astype(np.float32)
    packed, am = o.quantize_fp4(w, 64)
    x = o.round_to_bf16(rng.standard_normal(K))  # representable in bf16 (hence f32); fp16 tests re-round
    g[f"gemv{tag}_packed"], g[f"gemv{tag}_absmax"], g[f"gemv{tag}_x"] = packed, am, x
    g[f"gemv{tag}_shape"] = np.array([M, K], np.int64)
    g[f"gemv{tag}_exact"] = o.gemv_exact(x, packed, am, M, K, 64)

np.savez_compressed(OUT, **g)
print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(g), "arrays")
