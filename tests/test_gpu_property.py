"""Randomised GPU parity (hypothesis): arbitrary lengths, block sizes, dtypes, tables and pointer offsets for the
dequant, arbitrary shapes for the GEMV - every draw is checked against the oracle with the same bars as the
hand-picked cases (bit-exact / half-ulp)."""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings, strategies as st

import hipabi
from gpu_util import NPDT, bits, dev, np_bits, to_dev, torch_values
from oracle import c_oracle, fp4_oracle as o

pytestmark = pytest.mark.gpu
DT = [torch.float32, torch.float16, torch.bfloat16]
COMMON = dict(deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)


@settings(max_examples=60, **COMMON)
@given(n=st.integers(1, 300000), bs_log=st.integers(1, 13), odd_bs=st.booleans(), dt=st.sampled_from(DT),
       table=st.sampled_from([("codebook", hipabi.TABLE_CODEBOOK), ("tree", hipabi.TABLE_TREE)]), off=st.sampled_from([0, 0, 0, 1, 4, 16]),
       flags=st.sampled_from([0, 1, 2]), seed=st.integers(0, 2**31))
def test_dequant_any_shape(n, bs_log, odd_bs, dt, table, off, flags, seed):
    bs = (1 << bs_log) * (3 if odd_bs and bs_log < 10 else 1)  # even, sometimes not a power of two
    rng = np.random.default_rng(seed)
    packed = rng.integers(0, 256, (n + 1) // 2 + off, dtype=np.uint8)
    am = (rng.random(n // 2 + 8, dtype=np.float32) * 2 + 1e-3).astype(np.float32)  # long enough for any index rule
    want = np_bits(o.dequantize(packed[off:], am, bs, n, NPDT[dt], table[0]))
    out = hipabi.dequantize(to_dev(packed)[off:], to_dev(am), bs, n, dt, table[1], flags=flags)
    assert np.array_equal(bits(out), want)


@settings(max_examples=40, **COMMON)
@given(M=st.integers(1, 300), kc=st.integers(1, 96), bs=st.sampled_from([32, 64, 128, 256]), dt=st.sampled_from(DT),
       bias=st.booleans(), seed=st.integers(0, 2**31))
def test_gemv_any_shape(M, kc, bs, dt, bias, seed):
    K = kc * 32
    rng = np.random.default_rng(seed)
    w = (rng.standard_normal(M * K) * 0.05).astype(np.float32)
    if K % bs:
        bs = 32
    packed, am = c_oracle.quantize(w, bs)
    x_t = torch_values(rng.standard_normal(K), dt)
    b_t = torch_values(rng.standard_normal(M) * 0.1, dt) if bias else None
    P, A = to_dev(packed), to_dev(am)
    y = hipabi.gemv(x_t, P, A, M, K, bs, bias=b_t)
    if bias:  # exact bias semantics: T(T(sum) + bias) for 16-bit, f32 add for f32
        plain = hipabi.gemv(x_t, P, A, M, K, bs)
        assert torch.equal(y, plain + b_t)
        y = plain
    xv = x_t.float().cpu().numpy().astype(np.float64)
    exact = c_oracle.gemv_f64(xv, packed, am, M, K, bs)
    scale = np.abs(o.dequantize_f32(packed, am, bs, M * K).reshape(M, K).astype(np.float64)) @ np.abs(xv)
    half_ulp = {torch.bfloat16: 2.0**-8, torch.float16: 2.0**-11, torch.float32: 0.0}[dt]
    assert (np.abs(y.float().cpu().numpy() - exact) <= half_ulp * 1.01 * np.abs(exact) + 1e-5 * scale + 1e-30).all()


@settings(max_examples=150, **COMMON)
@given(M=st.integers(1, 40000), kc=st.integers(1, 512), dt=st.sampled_from(DT), bias=st.booleans(), seed=st.integers(0, 2**31))
def test_gemv_any_large_shape(M, kc, dt, bias, seed):
    """The same draw at the sizes where the dispatcher's M- and K-dependent rules live (rows per workgroup, band counts, the LDS
    fallback beyond K = 16384, up to 0.65 G weights): EVERY row against the float64 product formed on the device by the pure-torch
    oracle, that evaluation tied to the C oracle on 16 sampled rows (first and last included)."""
    from oracle import torch_cpu

    K = kc * 32
    bs = 64 if K % 64 == 0 else 32
    g = torch.Generator(device=dev()).manual_seed(seed)
    packed_d = torch.randint(0, 256, (M * K // 2,), dtype=torch.uint8, device=dev(), generator=g)
    absmax_d = torch.rand(M * K // bs, device=dev(), generator=g) * 0.05 + 0.005
    rng = np.random.default_rng(seed)
    x_t = torch_values(rng.standard_normal(K), dt)
    b_t = torch_values(rng.standard_normal(M) * 0.1, dt) if bias else None
    y = hipabi.gemv(x_t, packed_d, absmax_d, M, K, bs, bias=b_t)
    if bias:
        plain = hipabi.gemv(x_t, packed_d, absmax_d, M, K, bs)
        assert torch.equal(y, plain + b_t)
        y = plain
    table_d = torch_cpu.code_table("codebook").to(dev())
    x64 = x_t.double()

    def product(xv, magnitudes):  # float64 x @ W^T (or |x| @ |W|^T) over row chunks of <= 32 Mi weights
        out = torch.empty(M, dtype=torch.float64, device=dev())
        step = max(1, (1 << 25) // K)
        for r0 in range(0, M, step):
            r1 = min(M, r0 + step)
            p = packed_d[r0 * K // 2:r1 * K // 2]
            w = torch_cpu.dequantize(p & 0x77 if magnitudes else p, absmax_d[r0 * K // bs:r1 * K // bs], r1 - r0, K, bs, torch.float32, table_d)
            out[r0:r1] = w.double() @ xv
        return out

    exact_d, scale_d = product(x64, False), product(x64.abs(), True)
    rows = np.unique(np.concatenate([[0, M - 1], rng.integers(0, M, 14)]))
    p_rows = packed_d.view(M, K // 2)[rows].cpu().numpy().reshape(-1)
    a_rows = absmax_d.view(M, K // bs)[rows].cpu().numpy().reshape(-1)
    want = c_oracle.gemv_f64(x64.cpu().numpy(), p_rows, a_rows, len(rows), K, bs)
    assert np.allclose(exact_d[torch.from_numpy(rows).to(dev())].cpu().numpy(), want, rtol=1e-11, atol=1e-13)
    half_ulp = {torch.bfloat16: 2.0**-8, torch.float16: 2.0**-11, torch.float32: 0.0}[dt]
    tol = half_ulp * 1.01 * exact_d.abs() + 1e-5 * scale_d + 1e-30
    err = (y.double() - exact_d).abs()
    assert int((err > tol).sum().item()) == 0, (M, K, dt, float((err / tol).max().item()))
    # the residual epilogue on whichever geometry was picked: one more rounded add on top of the plain result, bit for bit
    res = torch_values(rng.standard_normal(M), dt)
    fused = hipabi.gemv_fused(x_t, packed_d, absmax_d, M, K, bs, None, res)
    assert torch.equal(fused, y + res if dt == torch.float32 else (y.float() + res.float()).to(dt))
    part = hipabi.gemv_partial(x_t, packed_d, absmax_d, M, K, bs) if dt != torch.float32 else None
    if part is not None:  # the K-split building block: the raw f32 accumulator, same sum
        assert int(((part.double() - exact_d).abs() > 1e-5 * scale_d + 1e-30).sum().item()) == 0


@settings(max_examples=120, **COMMON)
@given(B=st.integers(1, 128), M=st.integers(1, 40000), kc=st.integers(1, 256), dt=st.sampled_from([torch.bfloat16, torch.float16]),
       bias=st.booleans(), seed=st.integers(0, 2**31))
def test_small_batch_any_large_shape(B, M, kc, dt, bias, seed):
    """fp4_hip_gemm_small over the whole range its dispatcher decides on - 1..128 activation rows, up to 40 000 weight rows, K any multiple
    of 64 up to 16 384 (matrix-core one-shot / persistent, VALU, one-pass wide kernels with 1..4 column tiles, the 16-row splits): either
    the shape is refused with a reason (UNSUPPORTED: the caller takes dequant + GEMM) or EVERY output element meets the GEMV's bar against
    the float64 product formed on the device by the pure-torch oracle (tied to the C oracle on sampled weight rows)."""
    from oracle import torch_cpu

    K, bs = kc * 64, 64
    g = torch.Generator(device=dev()).manual_seed(seed)
    packed_d = torch.randint(0, 256, (M * K // 2,), dtype=torch.uint8, device=dev(), generator=g)
    absmax_d = torch.rand(M * K // bs, device=dev(), generator=g) * 0.05 + 0.005
    rng = np.random.default_rng(seed)
    x_t = torch_values(rng.standard_normal((B, K)), dt)
    b_t = torch_values(rng.standard_normal(M) * 0.1, dt) if bias else None
    rc, y = hipabi.gemm_small(x_t, packed_d, absmax_d, M, K, bs, bias=b_t, expect_ok=None)
    assert rc in (hipabi.OK, hipabi.ERR_UNSUPPORTED), (rc, hipabi.last_error())
    if rc != hipabi.OK:
        assert "not covered" in hipabi.last_error() or "dtype" in hipabi.last_error(), hipabi.last_error()
        return
    table_d = torch_cpu.code_table("codebook").to(dev())
    x64 = x_t.double()
    bias64 = b_t.double() if bias else torch.zeros(M, dtype=torch.float64, device=dev())
    exact = torch.empty(B, M, dtype=torch.float64, device=dev())
    scale = torch.empty(B, M, dtype=torch.float64, device=dev())
    step = max(1, (1 << 24) // K)
    for r0 in range(0, M, step):
        r1 = min(M, r0 + step)
        p, a = packed_d[r0 * K // 2:r1 * K // 2], absmax_d[r0 * K // bs:r1 * K // bs]
        w = torch_cpu.dequantize(p, a, r1 - r0, K, bs, torch.float32, table_d).double()
        exact[:, r0:r1] = x64 @ w.t() + bias64[r0:r1]
        scale[:, r0:r1] = x64.abs() @ w.abs().t() + bias64[r0:r1].abs()
    rows = np.unique(np.concatenate([[0, M - 1], rng.integers(0, M, 6)]))
    p_rows = packed_d.view(M, K // 2)[rows].cpu().numpy().reshape(-1)
    a_rows = absmax_d.view(M, K // bs)[rows].cpu().numpy().reshape(-1)
    want = c_oracle.gemv_f64(x64[B - 1].cpu().numpy(), p_rows, a_rows, len(rows), K, bs) + bias64[torch.from_numpy(rows).to(dev())].cpu().numpy()
    assert np.allclose(exact[B - 1, torch.from_numpy(rows).to(dev())].cpu().numpy(), want, rtol=1e-11, atol=1e-13)
    half_ulp = {torch.bfloat16: 2.0**-8, torch.float16: 2.0**-11}[dt]
    tol = half_ulp * 1.01 * exact.abs() + 1e-5 * scale + 1e-30
    err = (y.double() - exact).abs()
    assert int((err > tol).sum().item()) == 0, (B, M, K, dt, bias, float((err / tol).max().item()))
    # the same call with the scratch buffer the library asks for (what the torch ops do): 33..64 rows on short weights with long rows then
    # run as split-K over workgroups + an ordered reducing launch - a different summation order, the same bar
    y_ws, asked = hipabi.gemm_small_ws(x_t, packed_d, absmax_d, M, K, bs, bias=b_t)
    err = (y_ws.double() - exact).abs()
    assert int((err > tol).sum().item()) == 0, ("ws", asked, B, M, K, dt, bias, float((err / tol).max().item()))
    if asked == 0:
        assert torch.equal(y_ws, y)
    # the fused epilogues on whatever kernel the dispatcher picked: residual = one more rounded add on top of the plain result (bit-exact);
    # gate|up pairs = torch's own silu(g) * u on the plain result's rows (same device exp: <= 1 ulp, >= 99.8 % identical)
    res = torch_values(rng.standard_normal((B, M)), dt)
    fused = hipabi.gemm_small_fused(x_t, packed_d, absmax_d, M, K, bs, bias=b_t, residual=res)
    assert torch.equal(fused, (y.float() + res.float()).to(dt))
    if M % 2 == 0:
        rc2 = hipabi.gemm_small_fused(x_t, packed_d, absmax_d, M, K, bs, bias=b_t, epilogue=hipabi.EPILOGUE_SILU_MUL_PAIRS, expect_ok=False)
        assert rc2 in (hipabi.OK, hipabi.ERR_UNSUPPORTED), (rc2, hipabi.last_error())
        if rc2 == hipabi.OK:
            gu = hipabi.gemm_small_fused(x_t, packed_d, absmax_d, M, K, bs, bias=b_t, epilogue=hipabi.EPILOGUE_SILU_MUL_PAIRS)
            ref = torch.nn.functional.silu(y[:, 0::2]) * y[:, 1::2]
            a, b = gu.view(torch.int16).int() & 0xFFFF, ref.contiguous().view(torch.int16).int() & 0xFFFF
            a = torch.where((a & 0x8000) != 0, 0x8000 - a, a)
            b = torch.where((b & 0x8000) != 0, 0x8000 - b, b)
            d = (a - b).abs()
            assert int(d.max().item()) <= 1 and float((d == 0).float().mean().item()) >= 0.998, (B, M, K, dt, int(d.max().item()))
