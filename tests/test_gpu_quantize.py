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
