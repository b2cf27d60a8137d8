"""GPU parity of the HIP FP4 quantiser (C ABI) against the oracle's restatement of bitsandbytes'
algorithm: bit-exact codes and scales.  (The quantiser itself is "parity unpinned" with respect to
bitsandbytes - see oracle/fp4_oracle.py - so the oracle is the specification here.)"""
import numpy as np
import pytest
import torch

import hipabi
from gpu_util import dev, to_dev
from oracle import c_oracle, fp4_oracle as o

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("bs", [32, 64, 128, 256, 512, 1024, 2048, 4096])
@pytest.mark.parametrize("tiles", [3, 16])
def test_quantize_whole_tiles_matches_oracle(dtype, bs, tiles):
    """n a multiple of 8192: the one-shot tiles kernel where blocks fit a wave (bs <= 512), the persistent kernel otherwise, and the
    persistent kernel forced - all bit-exact (zero blocks, a zero element, values at every magnitude)."""
    n = 8192 * tiles
    rng = np.random.default_rng(bs + tiles)
    w = (rng.standard_normal(n) * 10.0 ** rng.uniform(-3, 3, n)).astype(np.float32)
    w[bs : 2 * bs] = 0.0
    w[5] = 0.0
    _check(torch.from_numpy(w).to(dtype), bs)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("bs", [32, 64, 128, 256, 512, 1024, 2048, 4096])
def test_quantize_matches_oracle(dtype, bs):
    n = 4096 * 37 + 8 * 5 + 3  # ragged: partial last block, partial last dword, odd n
    rng = np.random.default_rng(bs)
    w = rng.standard_normal(n).astype(np.float32)
    w[bs : 2 * bs] = 0.0  # all-zero block
    w[5] = 0.0
    w_t = torch.from_numpy(w).to(dtype).to(dev())
    packed, absmax = hipabi.quantize(w_t, bs)
    want_p, want_a = o.quantize_fp4(w_t.float().cpu().numpy(), bs)
    assert np.array_equal(absmax.cpu().numpy(), want_a)
    got = packed.cpu().numpy()
    if n % 2:  # the unused low nibble of the last byte is zero in both
        assert got[-1] & 0x0F == 0
    assert np.array_equal(got, want_p)


def _check(w_t, bs):
    """Every kernel against the oracle: the library's own choice (variant 0: by dtype and size, csrc/quantize_fp4.hip), the persistent
    kernel forced (4 workgroups per CU) and the one-shot tiles kernel with 1 / 2 / 4 loads per lane (1001 / 1002 / 1004; it applies to
    whole tiles with blocks of at most 512 elements and hands everything else to the persistent kernel)."""
    want_p, want_a = o.quantize_fp4(w_t.float().cpu().numpy().reshape(-1), bs)
    for variant in (0, 4, 1001, 1002, 1004):
        hipabi.set_variant("quantize", variant)
        packed, absmax = hipabi.quantize(w_t.to(dev()), bs)
        np.testing.assert_array_equal(absmax.cpu().numpy(), want_a)  # NaN == NaN here
        bad = np.flatnonzero(packed.cpu().numpy() != want_p)
        assert bad.size == 0, (variant, bad[:8], packed.cpu().numpy()[bad[:8]], want_p[bad[:8]])
    hipabi.set_variant("quantize", 0)


def test_quantize_threshold_neighbourhoods():
    """absmax 1 makes x == w, so every threshold, its two f32 neighbours, both signs, -0.0 and the smallest
    denormals hit the ranking rule directly (the kernel ranks on bit patterns, the oracle with float compares)."""
    t = np.asarray(o.QUANT_THRESHOLDS, dtype=np.float32)
    vals = np.concatenate([t, np.nextafter(t, np.float32(2)), np.nextafter(t, np.float32(0)),
                           np.asarray([0.0, 1e-45, 1e-40, 1.1754944e-38, 0.99999994], np.float32)])
    vals = np.concatenate([vals, -vals]).astype(np.float32)
    blocks = []
    for i in range(0, vals.size, 63):
        chunk = np.zeros(64, np.float32)
        c = vals[i : i + 63]
        chunk[: c.size] = c
        chunk[63] = 1.0 if (i // 63) % 2 == 0 else -1.0
        blocks.append(chunk)
    _check(torch.from_numpy(np.concatenate(blocks)), 64)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("order", ["sorted", "shuffled"])
def test_quantize_every_16bit_pattern(dtype, order):
    """All 65536 bit patterns of the input type as weights - subnormals, infinities and NaNs included - once sorted by
    magnitude (blocks of near-equal values, subnormal / infinite / NaN absmax) and once shuffled."""
    bits = torch.arange(-32768, 32768, dtype=torch.int32).to(torch.int16)
    w = bits.view(dtype)
    if order == "sorted":
        w = w[torch.argsort((bits.to(torch.int32) & 0x7FFF), stable=True)]
    else:
        w = w[torch.randperm(w.numel(), generator=torch.Generator().manual_seed(3))]
    for bs in (64, 32, 256):
        _check(w, bs)


def test_quantize_nonfinite_scale_blocks():
    """f32 blocks whose absmax is subnormal (1/absmax = inf, so 0*inf = NaN), infinite or NaN."""
    rng = np.random.default_rng(5)
    w = rng.standard_normal(64 * 8).astype(np.float32)
    w[0:64] = rng.choice(np.asarray([0.0, 1e-39, -1e-39, 5e-40, -3e-45, -0.0], np.float32), 64)
    w[64:128] *= np.float32(1e-41)
    w[130] = np.inf
    w[200] = -np.inf
    w[201] = 0.0
    w[260] = np.nan
    w[330] = 3e38
    w[331] = -1e-45
    _check(torch.from_numpy(w), 64)
    _check(torch.from_numpy(w), 256)


def test_quantize_full_matrix_and_roundtrip():
    M = K = 4096
    g = torch.Generator(device="cpu").manual_seed(0)
    w = (torch.randn(M, K, generator=g) * 0.02).to(torch.float16)
    packed, absmax = hipabi.quantize(w.to(dev()), 64)
    want_p, want_a = c_oracle.quantize(w.float().numpy().reshape(-1), 64)
    assert np.array_equal(packed.cpu().numpy(), want_p) and np.array_equal(absmax.cpu().numpy(), want_a)
    # quantise -> dequantise -> quantise is a fixed point (size-independent property)
    deq = hipabi.dequantize(packed, absmax, 64, M * K, torch.float32, hipabi.TABLE_TREE)
    p2, a2 = hipabi.quantize(deq, 64)
    assert torch.equal(a2, absmax)
    assert torch.equal(hipabi.dequantize(p2, a2, 64, M * K, torch.float32, hipabi.TABLE_TREE), deq)


def test_torch_ext_quantize():
    import torch_bnb_fp4 as pkg

    w = torch.randn(300, 128, device=dev(), dtype=torch.bfloat16)
    packed, absmax = pkg.quantize_fp4(w, 64)
    assert packed.shape == (300 * 128 // 2, 1) and packed.dtype == torch.uint8 and absmax.shape == (300 * 128 // 64,)
    want_p, want_a = o.quantize_fp4(w.float().cpu().numpy().reshape(-1), 64)
    assert np.array_equal(packed.cpu().numpy().reshape(-1), want_p) and np.array_equal(absmax.cpu().numpy(), want_a)
    with pytest.raises(RuntimeError):
        pkg.quantize_fp4(w, 48)
