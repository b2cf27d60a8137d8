"""Maximum sizes: weights of 2^32 elements and more (2 GiB of packed nibbles), where 32-bit byte offsets stop being enough.

The fast GEMV / small-batch geometries address the weight through buffer descriptors with one 32-bit offset per lane and are only
dispatched while M * K < 2^32 (`csrc/gemv_fp4.hip` dispatch_regx, `csrc/gemm_small_fp4.hip`); at and beyond that size the entry points
must fall back to 64-bit addressing - silently and correctly, the reference takes any (M, K) that fits the card
(/root/reference/csrc/gemv_fp4_optimized.cu:289-299 derives everything from Bshape).  Nothing below the boundary exercises those branches,
so this file does: M x K = 262144 x 16384 = exactly 2^32, and 2^32 + one row.  EVERY output row is checked against the float64 product
formed on the device by the pure-torch oracle (oracle/torch_cpu.py), which is tied to the C oracle on sampled rows (first, last and the
rows on either side of each 2^31 / 2^32-byte offset boundary).  Same bar as tests/test_gpu_gemv.py."""
import numpy as np
import pytest
import torch

import hipabi
from gpu_util import HALF_ULP, dev
from oracle import c_oracle, torch_cpu
from test_gpu_dispatch_table import BS, _device_f64_product, _random_fp4

pytestmark = pytest.mark.gpu
K = 16384


@pytest.fixture(autouse=True)
def _default_variant():
    hipabi.set_variant("gemv", -1)
    yield
    hipabi.set_variant("gemv", -1)


@pytest.fixture(scope="module")
def huge():
    """One weight of 2^32 + K elements shared by the cases (2 GiB + 8 KiB of packed bytes, 256 MiB of scales)."""
    M = (1 << 32) // K + 1
    free, _ = torch.cuda.mem_get_info()
    if free < 24 << 30:
        pytest.skip("needs ~24 GiB of free device memory")
    packed, absmax = _random_fp4(M * K, 20260104)
    yield M, packed, absmax
    del packed, absmax
    torch.cuda.empty_cache()


def _sample_rows(M):
    per_row_bytes = K // 2
    marks = [0, M - 1] + [b // per_row_bytes + d for b in (1 << 31, (1 << 32) // 2, 3 << 30) for d in (-1, 0, 1)]
    rng = np.random.default_rng(5)
    return np.unique(np.clip(np.concatenate([np.array(marks), rng.integers(0, M, 40)]), 0, M - 1))


def _check(y, exact_d, scale_d, dtype, what):
    tol = HALF_ULP[dtype] * 1.01 * exact_d.abs() + 1e-5 * scale_d + 1e-30
    err = (y.double() - exact_d).abs()
    bad = int((err > tol).sum().item())
    assert bad == 0, (what, dtype, bad, float((err / tol).max().item()), int((err / tol).argmax().item()))


@pytest.mark.parametrize("rows", ["exactly_2_pow_32", "one_row_more"])
def test_gemv_at_and_beyond_2_pow_32_elements(huge, rows):
    M_all, packed_d, absmax_d = huge
    M = M_all - 1 if rows == "exactly_2_pow_32" else M_all
    assert M * K >= 2**32
    P, A = packed_d[: M * K // 2], absmax_d[: M * K // BS]
    table_d = torch_cpu.code_table("codebook").to(dev())
    sample = _sample_rows(M)
    p_rows = P.view(M, K // 2)[sample].cpu().numpy().reshape(-1)
    a_rows = A.view(M, K // BS)[sample].cpu().numpy().reshape(-1)
    for dtype in (torch.bfloat16, torch.float16, torch.float32):
        g = torch.Generator().manual_seed(99)
        x_t = torch.randn(K, generator=g).to(dtype).to(dev())
        x64 = x_t.double()
        exact_d = _device_f64_product(P, A, x64, M, K, table_d)
        scale_d = _device_f64_product(P, A, x64.abs(), M, K, table_d, magnitudes=True)
        want_rows = c_oracle.gemv_f64(x64.cpu().numpy(), p_rows, a_rows, len(sample), K, BS)
        assert np.allclose(exact_d[torch.from_numpy(sample).to(dev())].cpu().numpy(), want_rows, rtol=1e-11, atol=1e-13)
        y = hipabi.gemv(x_t, P, A, M, K, BS)
        _check(y, exact_d, scale_d, dtype, "gemv")
        if dtype == torch.float32:
            continue
        # the K-split building block (raw f32 accumulator) and the bias / residual epilogue go through the same dispatcher
        part = hipabi.gemv_partial(x_t, P, A, M, K, BS)
        _check(part, exact_d, scale_d, torch.float32, "gemv_partial")
        res = torch.randn(M, generator=g).to(dtype).to(dev())
        fused = hipabi.gemv_fused(x_t, P, A, M, K, BS, None, res)
        assert torch.equal(fused, (y.float() + res.float()).to(dtype))
        # gate|up pairs: either computed (and then equal to torch's ops on the plain rows) or reported as unsupported - never wrong
        if M % 2 == 0:
            out = torch.empty(M // 2, dtype=dtype, device=dev())
            rc = hipabi.gemv_fused(x_t, P, A, M, K, BS, None, None, hipabi.EPILOGUE_SILU_MUL_PAIRS, out=out, expect_ok=False)
            assert rc in (hipabi.OK, hipabi.ERR_UNSUPPORTED), (rc, hipabi.last_error())
            if rc == hipabi.OK:
                ref = torch.nn.functional.silu(y[0::2]) * y[1::2]
                close = (out.float() - ref.float()).abs() <= 2.0 ** -6 * ref.float().abs() + 1e-6
                assert bool(close.all())
        del exact_d, scale_d, y, part, fused


def test_small_batch_at_2_pow_32_elements(huge):
    """A few activation rows on the same weight (fp4_hip_gemm_small): whichever kernel the dispatcher falls back to, every output
    element meets the GEMV's bar."""
    M_all, packed_d, absmax_d = huge
    M = M_all - 1
    P, A = packed_d[: M * K // 2], absmax_d[: M * K // BS]
    table_d = torch_cpu.code_table("codebook").to(dev())
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(7)
    for B in (3, 16):
        xb = torch.randn(B, K, generator=g).to(dtype).to(dev())
        rc, out = hipabi.gemm_small(xb, P, A, M, K, BS, expect_ok=None)
        assert rc in (hipabi.OK, hipabi.ERR_UNSUPPORTED), (B, rc, hipabi.last_error())
        if rc != hipabi.OK:
            assert "2^32" in hipabi.last_error() or "large" in hipabi.last_error(), hipabi.last_error()  # refused with a reason: the caller takes dequant + GEMM
            continue
        for b in (0, B - 1):
            x64 = xb[b].double()
            exact_d = _device_f64_product(P, A, x64, M, K, table_d)
            scale_d = _device_f64_product(P, A, x64.abs(), M, K, table_d, magnitudes=True)
            _check(out[b], exact_d, scale_d, dtype, f"gemm_small rows={B} row {b}")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_quantiser_and_dequant_past_2_to_the_31_elements(dtype):
    """The quantiser (both kernels) and the dequant on 2^31 + 2^22 elements: element offsets past 2^31, byte offsets of the f32 input past
    2^33.  Blocks are independent, so a weight made of one 2^20-element pattern repeated must quantise to that pattern's packed bytes and
    scales repeated - and the pattern's own result is held against the oracle, bit for bit.  The dequant of the huge result must likewise
    be the pattern's dequant repeated (checked by comparing every repetition on the device)."""
    from oracle import fp4_oracle as o

    free, _ = torch.cuda.mem_get_info()
    unit, reps = 1 << 20, (1 << 11) + 4
    n = unit * reps
    if free < (n * (4 + torch.empty((), dtype=dtype).element_size()) + (4 << 30)):
        pytest.skip("needs ~24 GiB of free device memory")
    rng = np.random.default_rng(11)
    pat = (rng.standard_normal(unit) * 10.0 ** rng.uniform(-2, 1, unit)).astype(np.float32)
    pat[64:128] = 0.0
    pat_t = torch.from_numpy(pat).to(dtype).to(dev())
    want_p, want_a = o.quantize_fp4(pat_t.float().cpu().numpy(), BS)
    w = pat_t.repeat(reps)
    assert w.numel() == n and n > (1 << 31)
    for variant in (4, 1002, 0):  # persistent kernel, tiles kernel, the library's choice
        hipabi.set_variant("quantize", variant)
        packed, absmax = hipabi.quantize(w, BS)
        p2, a2 = packed.view(reps, unit // 2), absmax.view(reps, unit // BS)
        assert np.array_equal(p2[0].cpu().numpy(), want_p) and np.array_equal(a2[0].cpu().numpy(), want_a), variant
        assert bool((p2 == p2[0]).all()) and bool((a2 == a2[0]).all()), variant  # every repetition, the ones past 2^31 included
    hipabi.set_variant("quantize", 0)
    del w
    out = hipabi.dequantize(packed, absmax, BS, n, torch.bfloat16)
    o2 = out.view(reps, unit).view(torch.int16)
    want = o.dequantize(want_p, want_a, BS, unit, "bfloat16", "codebook")
    assert np.array_equal(o2[0].cpu().numpy().view(np.uint16), want) and bool((o2 == o2[0]).all())
