"""GPU parity of the fused GEMV epilogues (fp4_hip_gemv_fused) through the C ABI.

The sums themselves are covered by test_gpu_gemv.py (float64 oracle).  What is new here is elementwise and therefore
held to a bit-exact bar wherever arithmetic allows:

* bias / residual epilogue: out == T(T(T(sum) + bias) + residual) BIT FOR BIT, where T(sum) is what the plain GEMV
  (fp4_hip_gemv, parity-tested on its own) returns for the same operands and the adds are redone by the numpy oracle
  (oracle.linear_epilogue) - the kernel computes the same sum either way, so any difference is the epilogue's;
* gate|up epilogue: against oracle.silu_mul_epilogue on the plain GEMV's gate / up rows: bit-exact except where the f32
  exp of numpy and of the device library differ by an ulp that lands on a rounding boundary of T - at most 1 ulp of T
  on at most 0.2 % of the elements; and against torch's own silu() * up on the GPU (same device library): >= 99.9 %
  identical, never more than 1 ulp of T apart.
"""
import numpy as np
import pytest
import torch

import hipabi
from gpu_util import NPDT, bits, case, dev, to_dev, torch_values
from oracle import fp4_oracle as o

pytestmark = pytest.mark.gpu
DT16 = [torch.bfloat16, torch.float16]
# every K class of the register-x geometry: 1 / 2 / 4 bands, deep slices, 5 / 6 / 7 / 8 bands, plus the decode shapes
SHAPES = [(4096, 4096), (28672, 4096), (4096, 14336), (6144, 4096), (64, 768), (66, 1024), (258, 2048), (130, 11008), (70, 7168),
          (36, 28672), (132, 5120), (70, 6144), (4100, 8192), (34, 16384), (2, 64)]


def make_case(M, K, seed=None, bs=64):
    """(packed, absmax, x, bias, res) of the session-cached case of this shape (gpu_util.case: quantised once by the CPU oracle and
    shared by every test function and parameter that works on the shape)."""
    c = case(M, K, bs, 1000003 * M + K if seed is None else seed)
    return c.packed, c.am, c.x, c.bias, c.res


def dev_case(M, K, bs=64):
    c = case(M, K, bs, 1000003 * M + K)
    return c.P, c.A


def as_np(t):
    return t.float().cpu().numpy()


@pytest.mark.parametrize("dtype", DT16 + [torch.float32])
@pytest.mark.parametrize("M,K", SHAPES + [(5, 64), (33, 24), (257, 2112)])
def test_bias_residual_epilogue_is_bit_exact(dtype, M, K):
    bs = 64 if K % 64 == 0 else 8
    packed, am, x, bias, res = make_case(M, K, bs=bs)
    P, A = dev_case(M, K, bs)
    x_t, b_t, r_t = torch_values(x, dtype), torch_values(bias, dtype), torch_values(res, dtype)
    plain = hipabi.gemv(x_t, P, A, M, K, bs)
    for use_bias, use_res in ((False, True), (True, True), (True, False), (False, False)):
        got = hipabi.gemv_fused(x_t, P, A, M, K, bs, b_t if use_bias else None, r_t if use_res else None)
        if dtype == torch.float32:  # f32: plain f32 adds in the same order
            want = as_np(plain)
            want = want + as_np(b_t) if use_bias else want
            want = want + as_np(r_t) if use_res else want
        else:
            want = o.linear_epilogue(as_np(plain), NPDT[dtype], as_np(b_t) if use_bias else None, as_np(r_t) if use_res else None)
        assert np.array_equal(as_np(got).view(np.uint32), np.asarray(want, np.float32).view(np.uint32)), (use_bias, use_res)
    # in place: residual aliases out (h = h + Linear(a))
    h = r_t.clone()
    hipabi.gemv_fused(x_t, P, A, M, K, bs, None, h, out=h)
    assert torch.equal(h, hipabi.gemv_fused(x_t, P, A, M, K, bs, None, r_t))


def ulp_distance(a_bits, b_bits):
    a, b = a_bits.astype(np.int64), b_bits.astype(np.int64)
    a = np.where(a & 0x8000, 0x8000 - a, a)  # sign-magnitude -> a monotone integer line
    b = np.where(b & 0x8000, 0x8000 - b, b)
    return np.abs(a - b)


@pytest.mark.parametrize("dtype", DT16)
@pytest.mark.parametrize("with_bias", [False, True])
@pytest.mark.parametrize("M,K", [s for s in SHAPES if s[0] % 2 == 0])
def test_gate_up_epilogue(dtype, M, K, with_bias):
    packed, am, x, bias, res = make_case(M, K)
    P, A = dev_case(M, K)
    x_t = torch_values(x * 2.0, dtype)  # gate values spread over a few units: silu is exercised off its linear part
    b_t = torch_values(bias, dtype) if with_bias else None
    r_t = torch_values(res[: M // 2], dtype)
    plain = hipabi.gemv(x_t, P, A, M, K, 64, b_t)  # rows 2i = gate_i, 2i+1 = up_i, each already T(+bias)
    g, u = plain[0::2].contiguous(), plain[1::2].contiguous()
    got = hipabi.gemv_fused(x_t, P, A, M, K, 64, b_t, None, hipabi.EPILOGUE_SILU_MUL_PAIRS)
    got_r = hipabi.gemv_fused(x_t, P, A, M, K, 64, b_t, r_t, hipabi.EPILOGUE_SILU_MUL_PAIRS)
    # 1. torch's own ops on the GPU (same device exp): the sequence a model runs after two separate Linears
    ref = torch.nn.functional.silu(g) * u
    d = ulp_distance(bits(got), bits(ref))
    assert d.max() <= 1 and (d == 0).mean() >= 0.999, (int(d.max()), float((d == 0).mean()))
    d = ulp_distance(bits(got_r), bits(ref + r_t))
    assert d.max() <= 1 and (d == 0).mean() >= 0.999
    # 2. the numpy oracle (its exp may differ from the device's by an f32 ulp)
    want = o.silu_mul_epilogue(as_np(g), as_np(u), NPDT[dtype])
    want_bits = want.astype(np.float16).view(np.uint16) if dtype == torch.float16 else o.f32_to_bf16_bits(want)
    d = ulp_distance(bits(got), want_bits)
    assert d.max() <= 1 and (d == 0).mean() >= 0.998, (int(d.max()), float((d == 0).mean()))


def test_gate_up_against_float64_oracle_end_to_end():
    """The whole fused launch against float64: silu(g*) * u* with g*, u* the exact sums; the tolerance is the rounding chain
    of the unfused model code (three roundings to bf16 + the GEMV's own bar)."""
    M, K = 28672, 4096
    c = case(M, K, 64, 1000003 * M + K)
    packed, am, x = c.packed, c.am, c.x
    x_t = torch_values(x, torch.bfloat16)
    got = as_np(hipabi.gemv_fused(x_t, c.P, c.A, M, K, 64, None, None, hipabi.EPILOGUE_SILU_MUL_PAIRS)).astype(np.float64)
    exact, _ = c.exact(x_t, tag=("x", torch.bfloat16))
    g, u = exact[0::2], exact[1::2]
    sil = g / (1.0 + np.exp(-g))
    want = sil * u
    # d(silu)/dg is within [-0.1, 1.1]; each of g, u, silu, product carries <= half a bf16 ulp of relative error
    tol = 2.0**-8 * 1.02 * (np.abs(want) * 3 + 1.1 * np.abs(g) * np.abs(u)) + 1e-6
    assert (np.abs(got - want) <= tol).all(), float((np.abs(got - want) / tol).max())


def test_unsupported_combinations_are_reported_not_computed():
    M, K = 64, 768
    packed, am, x, _, _ = make_case(M, K, 3)
    P, A = to_dev(packed), to_dev(am)
    rc = hipabi.gemv_fused(torch_values(x, torch.float32), P, A, M, K, 64, None, None, hipabi.EPILOGUE_SILU_MUL_PAIRS, expect_ok=False)
    assert rc == hipabi.ERR_UNSUPPORTED and "not available" in hipabi.last_error()
    rc = hipabi.gemv_fused(torch_values(x, torch.bfloat16), P, A, M - 1, K, 64, None, None, hipabi.EPILOGUE_SILU_MUL_PAIRS, expect_ok=False)
    assert rc == hipabi.ERR_INVALID  # odd row count
    rc = hipabi.gemv_fused(torch_values(x, torch.bfloat16), P, A, M, K, 64, None, None, 7, expect_ok=False)
    assert rc == hipabi.ERR_INVALID
    # K > 16384 (LDS geometry: a pair's rows sit in different waves)
    M2, K2 = 8, 32768
    packed, am, x, _, _ = make_case(M2, K2, 4)
    rc = hipabi.gemv_fused(torch_values(x, torch.bfloat16), to_dev(packed), to_dev(am), M2, K2, 64, None, None,
                           hipabi.EPILOGUE_SILU_MUL_PAIRS, expect_ok=False)
    assert rc == hipabi.ERR_UNSUPPORTED


def test_fused_layers_match_the_unfused_modules_and_capture_into_a_graph():
    """Module level: FusedFP4Linear (residual) and the gate|up layer against the unfused TorchFP4Linear sequence, eagerly and
    replayed from a HIP graph; a [2, K] input falls back to the unfused sequence with the same semantics."""
    import torch_bnb_fp4 as pkg
    from torch_bnb_fp4 import fused

    H, I = 512, 1408
    g = torch.Generator().manual_seed(5)
    mk = lambda o_, i_: torch.nn.Linear(i_, o_, bias=True).to(torch.bfloat16)
    torch.manual_seed(11)
    gate, up, down = mk(I, H), mk(I, H), mk(H, I)
    fp = [pkg.TorchFP4Linear(pkg.swap_linear_with_bnb_linear(l, dtype=torch.bfloat16).to(dev())) for l in (gate, up, down)]
    gu = fused.FusedFP4Linear.gate_up(fp[0], fp[1]).to(dev())
    dn = fused.FusedFP4Linear.from_linear(fp[2]).to(dev())
    h = torch.randn(1, H, generator=g).to(torch.bfloat16).to(dev())
    want = h + fp[2](torch.nn.functional.silu(fp[0](h)) * fp[1](h))
    got = dn(gu(h), residual=h)
    assert gu.out_features == I and got.shape == want.shape
    d = ulp_distance(bits(got), bits(want))
    assert d.max() <= 2 and (d == 0).mean() >= 0.99  # a 1-ulp silu difference can propagate through down's sum
    # graph capture: every launch is capturable (no allocation outside torch's pool, no sync)
    s = torch.cuda.Stream()
    static_h = h.clone()
    with torch.cuda.stream(s):
        dn(gu(static_h), residual=static_h)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = dn(gu(static_h), residual=static_h)
    static_h.copy_(h * 0.5)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, dn(gu(h * 0.5), residual=h * 0.5))
    # state_dict carries the packed weight, the scales and the (compute-dtype) bias; loading new scales reaches the kernel
    sd = {k: v.clone() for k, v in dn.state_dict().items()}  # (state_dict() hands out the live buffers; load copies in place)
    assert set(sd) == {"qweight", "absmax", "bias"} and sd["bias"].dtype == torch.bfloat16
    sd2 = {k: v.clone() for k, v in sd.items()}
    sd2["absmax"] = sd2["absmax"] * 2.0
    mid = gu(h)
    before = dn(mid)
    dn.load_state_dict(sd2)
    after = dn(mid)
    assert (after.float() - (2.0 * (before.float() - sd["bias"].float()) + sd["bias"].float())).abs().max().item() <= 0.03 * after.float().abs().max().item() + 0.02
    dn.load_state_dict(sd)
    assert torch.equal(dn(mid), before)
    # the packaged form of the same thing
    step = pkg.GraphedStep(lambda t: dn(gu(t), residual=t), h)
    for scale in (1.0, 0.75, -0.5):
        assert torch.equal(step(h * scale), dn(gu(h * scale), residual=h * scale))
    with pytest.raises(ValueError):
        step(torch.cat([h, h]))
    # batch of 2: the small-batch kernels with the same epilogues
    h2 = torch.cat([h, h * 0.25])
    got2 = dn(gu(h2), residual=h2)
    want2 = h2 + fp[2](torch.nn.functional.silu(fp[0](h2)) * fp[1](h2))
    assert got2.shape == want2.shape and (got2.float() - want2.float()).abs().max() <= 2e-2 * want2.float().abs().max()
    # 40 and 100 rows (I = 1408 = 22 quant blocks: the one-pass kernels with a ragged last step; 100 rows: two chunks), 3-D input
    for rows in (40, 100):
        hb = (torch.randn(rows, H, generator=g) * 0.5).to(torch.bfloat16).to(dev()).view(2, rows // 2, H)
        gotb = dn(gu(hb), residual=hb)
        wantb = hb + fp[2](torch.nn.functional.silu(fp[0](hb)) * fp[1](hb))
        assert gotb.shape == wantb.shape == (2, rows // 2, H)
        assert (gotb.float() - wantb.float()).abs().max() <= 2e-2 * wantb.float().abs().max()


@pytest.mark.parametrize("dtype", DT16)
@pytest.mark.parametrize("wide", [-1, 0, 1, 2, 3, 4, 5])  # 17..64 rows: default dispatch, 16-row launches, the one-pass kernels' 16 / 32 / 64 / 128-row workgroups
@pytest.mark.parametrize("B", [24, 40, 64, 100])
@pytest.mark.parametrize("M,K", [(4096, 4096), (258, 2048), (130, 14336), (66, 1024)])
def test_wide_batch_epilogues(dtype, wide, B, M, K):
    """The same epilogues on 17..128 rows, on every path those take (the one-pass kernel's epilogue is its own code)."""
    if wide > 0 and B > 64 and wide != 3:
        pytest.skip("above 64 rows the chunks take the same kernels as the 33..64-row cases")
    hipabi.set_variant("gemm_wide", wide)
    try:
        test_small_batch_epilogues(dtype, B, M, K)
    finally:
        hipabi.set_variant("gemm_wide", -1)


@pytest.mark.parametrize("dtype", DT16)
@pytest.mark.parametrize("B", [1, 2, 4, 8, 11, 16, 24])
@pytest.mark.parametrize("M,K", [(4096, 4096), (6144, 4096), (258, 2048), (130, 14336), (66, 1024), (36, 8192), (66, 768), (34, 1280)])
def test_small_batch_epilogues(dtype, B, M, K):
    """fp4_hip_gemm_small_fused (batched decode): the residual epilogue is BIT-EXACT against the numpy oracle applied to the plain
    small-batch product (same kernel, same sum); the gated epilogue within 1 ulp of torch's silu(g) * u on that product's
    gate / up columns, >= 99.8 % identical; every kernel family (persistent / one-shot matrix-core, VALU fallback) and the
    17..32-row split over two launches."""
    packed, am, x, bias, res = make_case(M, K)
    rng = np.random.default_rng(B)
    P, A = dev_case(M, K)
    xb = torch_values(rng.standard_normal((B, K)).astype(np.float32), dtype)
    b_t = torch_values(bias, dtype)
    r_t = torch_values(rng.standard_normal((B, M)).astype(np.float32), dtype)
    rc = hipabi.gemm_small(xb, P, A, M, K, 64, b_t, expect_ok=False)
    if rc != hipabi.OK:
        assert rc == hipabi.ERR_UNSUPPORTED  # shape outside the small-batch kernels: the fused entry must say the same
        assert hipabi.gemm_small_fused(xb, P, A, M, K, 64, b_t, r_t, expect_ok=False) == hipabi.ERR_UNSUPPORTED
        return
    plain = hipabi.gemm_small(xb, P, A, M, K, 64, b_t)
    got = hipabi.gemm_small_fused(xb, P, A, M, K, 64, b_t, r_t)
    want = o.linear_epilogue(as_np(plain), NPDT[dtype], None, as_np(r_t))
    assert np.array_equal(as_np(got).view(np.uint32), np.asarray(want, np.float32).view(np.uint32))
    assert torch.equal(hipabi.gemm_small_fused(xb, P, A, M, K, 64, b_t, None), plain)
    # gate | up
    rh = r_t[:, : M // 2].contiguous()
    gu = hipabi.gemm_small_fused(xb, P, A, M, K, 64, b_t, None, hipabi.EPILOGUE_SILU_MUL_PAIRS)
    gur = hipabi.gemm_small_fused(xb, P, A, M, K, 64, b_t, rh, hipabi.EPILOGUE_SILU_MUL_PAIRS)
    ref = torch.nn.functional.silu(plain[:, 0::2]) * plain[:, 1::2]
    assert gu.shape == (B, M // 2)
    for a, b in ((gu, ref), (gur, ref + rh)):
        d = ulp_distance(bits(a), bits(b))
        assert d.max() <= 1 and (d == 0).mean() >= 0.998, (int(d.max()), float((d == 0).mean()))
