"""GPU parity of the fused FP4 GEMV through the C ABI.

Floating-point work: tolerance-based (the reference's own GEMV is not bit-reproducible either: it
accumulates per lane in half/bf16, csrc/gemv_fp4_optimized.cu:87,146-148).  Two bars, both written
down here:

1. closeness to the exact result  y* = x @ dequant_f32(W)^T  (float64, oracle.gemv_exact):
   |y - y*| <= ulp_T(y*)/2 * 1.01 + 1e-5 * sum_k |x_k w_rk|      (one final rounding to T plus f32
   accumulation noise; ulp_T/2 = 2^-8 |y| for bf16, 2^-11 |y| for fp16, 0 for f32);
2. at least as close to y* as an emulation of the reference kernel's arithmetic
   (oracle.gemv_reference_emulated), in mean absolute error.
"""
import numpy as np
import pytest
import torch

import hipabi
from gpu_util import HALF_ULP, NPDT, assert_within_bar, case, dev, to_dev, torch_values
from oracle import c_oracle, fp4_oracle as o

pytestmark = pytest.mark.gpu
DTYPES = [torch.bfloat16, torch.float16, torch.float32]
# every 16-bit geometry the library builds for this shape: the LDS geometry (north-star mapping at 4 and 8 waves) + the register-x
# family at 1 / 2 / 4 row pairs per group (round 3 removed the sweep-only instantiations, profiles/r03_kernel_inventory.txt)
VARIANTS16 = [r | (w << 8) | (u << 16) for (r, w, u) in [(1, 4, 2), (1, 8, 2)]] + [(1 << 24) | it for it in (1, 2, 4)]


@pytest.fixture(autouse=True)
def _default_variant():
    hipabi.set_variant("gemv", -1)
    yield
    hipabi.set_variant("gemv", -1)


def make_case(M, K, bs=64, seed=0, wscale=0.02):
    """(packed, absmax, x) of the session-cached case (gpu_util.case): the oracle quantises each (shape, seed) once."""
    c = case(M, K, bs, seed, wscale)
    return c.packed, c.am, c.x


def shape_seed(M, K):
    """One seed per shape, so that every test function working on a shape shares its quantised weight and float64 answers."""
    return 1000003 * M + K


def check_case(y: torch.Tensor, c, dtype, x_t=None):
    """Bar 1 for the case's own activation vector (its float64 answer is computed once per dtype and shared)."""
    if x_t is None:
        x_t = torch_values(c.x, dtype)
    exact, scale = c.exact(x_t, tag=("x", dtype))
    return exact, assert_within_bar(y, exact, scale, dtype)


def check(y: torch.Tensor, x_t: torch.Tensor, packed, am, M, K, bs, dtype, bias=None):
    """Bar 1 for explicit operands (no caching: golden vectors and one-off inputs)."""
    xv = x_t.float().cpu().numpy().astype(np.float64)
    exact = c_oracle.gemv_f64(xv, packed, am, M, K, bs) if K % 2 == 0 else o.gemv_exact(xv, packed, am, M, K, bs)
    wabs = np.abs(o.dequantize_f32(packed, am, bs, M * K).reshape(M, K).astype(np.float64))
    scale = wabs @ np.abs(xv)
    if bias is not None:
        # bias semantics are exact (T(T(sum) + bias)), checked separately; compare pre-bias here
        raise AssertionError("use check() without bias")
    return exact, assert_within_bar(y, exact, scale, dtype)


@pytest.mark.parametrize("tag", list("abc"))
@pytest.mark.parametrize("dtype", DTYPES)
def test_golden_cases_and_reference_emulation(golden, tag, dtype):
    M, K = (int(v) for v in golden[f"gemv{tag}_shape"])
    packed, am, x = golden[f"gemv{tag}_packed"], golden[f"gemv{tag}_absmax"], golden[f"gemv{tag}_x"]
    x_t = torch_values(x, dtype)
    y = hipabi.gemv(x_t, to_dev(packed), to_dev(am), M, K, 64)
    exact, err = check(y, x_t, packed, am, M, K, 64, dtype)
    xv = x_t.float().cpu().numpy()
    emu = o.gemv_reference_emulated(xv, packed, am, M, K, 64, NPDT[dtype]).astype(np.float64)
    emu_err = np.abs(emu - o.gemv_exact(xv, packed, am, M, K, 64))
    assert err.mean() <= emu_err.mean() * 1.05 + 1e-12, (err.mean(), emu_err.mean())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,K", [(4096, 4096), (4096, 14336)])
def test_bar2_reference_emulation_at_decode_shapes(dtype, M, K):
    """Bar 2 at the headline shape and at the down-projection, default geometry: the kernel's mean |error| against the
    float64 answer is no larger than that of an emulation of the REFERENCE kernel's arithmetic (T-rounded table and
    absmax, T multiply, T accumulate per lane, f32 tree reduce: csrc/gemv_fp4_optimized.cu:87-156).  The kernel runs on
    the full matrix; the (row-independent) emulation is evaluated on the first 512 rows to bound the test's run time."""
    rows = 512
    c = case(M, K, seed=shape_seed(M, K))
    packed, am = c.packed, c.am
    x_t = torch_values(c.x, dtype)
    y = hipabi.gemv(x_t, c.P, c.A, M, K, 64)
    exact, err = check_case(y, c, dtype, x_t)  # bar 1 on every row
    xv = x_t.float().cpu().numpy()
    p_s, a_s = packed[: rows * K // 2], am[: rows * K // 64]
    emu = o.gemv_reference_emulated(xv, p_s, a_s, rows, K, 64, NPDT[dtype]).astype(np.float64)
    emu_err = np.abs(emu - exact[:rows])
    assert err[:rows].mean() <= emu_err.mean() * 1.05 + 1e-12, (err[:rows].mean(), emu_err.mean())
    # and the emulation itself stays within the reference's own accuracy class (a sanity check of the emulation)
    assert emu_err.mean() <= 40 * HALF_ULP[dtype] * np.abs(exact[:rows]).mean() + 1e-3


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("variant", VARIANTS16)
def test_4096x4096_every_variant(dtype, variant):
    M = K = 4096
    c = case(M, K, seed=shape_seed(M, K))
    x_t = torch_values(c.x, dtype)
    hipabi.set_variant("gemv", variant)
    y = hipabi.gemv(x_t, c.P, c.A, M, K, 64)
    check_case(y, c, dtype, x_t)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,K", [(14336, 4096), (4096, 14336), (1024, 4096), (2048, 768), (64, 2048), (66, 768), (5, 64), (1, 32),
                                 (3, 8192), (257, 2112), (130, 11008), (70, 13824), (130, 7168), (37, 28672), (5120, 5120), (70, 10240), (133, 6144), (4099, 8192)])
def test_model_shapes(dtype, M, K):
    c = case(M, K, seed=shape_seed(M, K))
    x_t = torch_values(c.x, dtype)
    y = hipabi.gemv(x_t, c.P, c.A, M, K, 64)
    check_case(y, c, dtype, x_t)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("iters", [1, 2, 4])
@pytest.mark.parametrize("M,K", [(14336, 4096), (4096, 14336), (1024, 4096), (2048, 768), (64, 2048), (66, 768), (5, 64), (1, 32),
                                 (3, 8192), (257, 2112), (100, 16384), (7, 32768), (33, 1024), (19, 2048), (130, 11008), (9, 12288)])
def test_register_x_geometry_all_band_splits(dtype, iters, M, K):  # noqa: D401
    """The second GEMV geometry (x in registers, K split across waves): every KSPLIT x G instantiation,
    ragged M, idle lanes, and the K > 16384 fall-back to the LDS geometry."""
    c = case(M, K, seed=shape_seed(M, K))
    x_t = torch_values(c.x, dtype)
    hipabi.set_variant("gemv", (1 << 24) | iters)
    y = hipabi.gemv(x_t, c.P, c.A, M, K, 64)
    check_case(y, c, dtype, x_t)


# (M, K, bands, row pairs per group): five / six bands at 1 / 2 / 4 row pairs; seven and eight bands at the one depth-specific setting
# that is built (7168 -> 2, 14336 -> 4, 28672 -> 2; 8192 -> 2 or 4); the last three land on the standard split (their band
# geometries are not built: 12288 six-band, 16384 / 32768 eight-band) and must still be right.
BAND_CASES = ([(M, K, b, it) for (M, K, b) in [(66, 5120, 5), (131, 10240, 5), (2048, 5120, 5), (66, 6144, 6)] for it in (1, 2, 4)] +
              [(66, 7168, 7, 2), (5, 7168, 7, 2), (4096, 14336, 7, 4), (130, 14336, 7, 4), (37, 28672, 7, 2),
               (4100, 8192, 8, 2), (4100, 8192, 8, 4), (66, 8192, 8, 2), (35, 12288, 6, 2), (40, 16384, 8, 2), (9, 32768, 8, 2)])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,K,bands,iters", BAND_CASES)
def test_register_x_five_six_seven_band_geometries(dtype, M, K, bands, iters):
    """K = 5 / 6 / 7 / 8 band widths: as many waves as bands, 1 / 2 / 4 groups per lane, ragged M."""
    c = case(M, K, seed=shape_seed(M, K))
    x_t = torch_values(c.x, dtype)
    try:
        hipabi.set_variant("gemv", (1 << 24) | (bands << 8) | iters)
        y = hipabi.gemv(x_t, c.P, c.A, M, K, 64)
    finally:
        hipabi.set_variant("gemv", -1)
    check_case(y, c, dtype, x_t)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,K,bs", [(37, 96, 32), (16, 80, 16), (9, 4096, 128), (12, 1024, 1024), (20, 256, 4096), (7, 66, 2),
                                    (33, 2048, 32)])
def test_blocksizes_and_generic_shapes(dtype, M, K, bs):
    packed, am, x = make_case(M, K, bs=bs, seed=bs)
    x_t = torch_values(x, dtype)
    y = hipabi.gemv(x_t, to_dev(packed), to_dev(am), M, K, bs)
    check(y, x_t, packed, am, M, K, bs, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_fused_bias_equals_separate_add_bitwise(dtype):
    M, K = 1000, 1024
    packed, am, x = make_case(M, K, seed=3)
    x_t = torch_values(x, dtype)
    bias = torch_values(np.random.default_rng(4).standard_normal(M) * 0.1, dtype)
    P, A = to_dev(packed), to_dev(am)
    plain = hipabi.gemv(x_t, P, A, M, K, 64)
    fused = hipabi.gemv(x_t, P, A, M, K, 64, bias=bias)
    assert torch.equal(fused, plain + bias)  # the reference's `out += bias` (torch_bnb_fp4/__init__.py:608-613)


def test_extreme_activations_and_signs():
    # large-magnitude bf16 activations (beyond fp16 range) and a weight row of all -1.0 / +1.0 codes
    M, K = 8, 2048
    nib = np.where(np.arange(M * K) % 2 == 0, 3, 11).astype(np.uint8)  # +1, -1 alternating
    packed = ((nib[0::2] << 4) | nib[1::2]).astype(np.uint8)
    am = np.full(M * K // 64, 0.5, np.float32)
    x = np.zeros(K, np.float32)
    x[0::2] = 3.0e5
    x[1::2] = 1.0e5
    x_t = torch_values(x, torch.bfloat16)
    y = hipabi.gemv(x_t, to_dev(packed), to_dev(am), M, K, 64)
    xv = x_t.float().cpu().numpy().astype(np.float64)
    want = 0.5 * (xv[0::2].sum() - xv[1::2].sum())
    assert np.allclose(y.float().cpu().numpy(), want, rtol=2.0**-8)


def test_torch_ext_gemv_shapes_and_errors():
    import torch_bnb_fp4 as pkg

    M, K = 192, 256
    packed, am, x = make_case(M, K, seed=9)
    B = to_dev(packed).view(-1, 1).t()  # what QuantData passes: [1, numel/2], still contiguous
    absmax, code = to_dev(am), pkg.ext.code_table("tree").to(dev())
    for dt in DTYPES:
        x_t = torch_values(x, dt).view(1, K)
        y2 = pkg.gemm_4bit_inference(x_t, B, absmax, code, 64, dt, torch.Size([M, K]))
        y3 = pkg.gemm_4bit_inference(x_t.view(1, 1, K), B, absmax, code, 64, dt, [M, K])
        assert y2.shape == (1, M) and y3.shape == (1, 1, M) and y2.dtype == dt
        assert torch.equal(y2.view(-1), y3.view(-1))
        assert torch.equal(y2.view(-1), hipabi.gemv(x_t.view(-1), to_dev(packed), absmax, M, K, 64))
        yq = pkg.gemm_4bit_inference_qtype(x_t, B, absmax, code, 64, pkg.ScalarType.from_torch_dtype(dt).value, [M, K])
        assert torch.equal(yq, y2)
    with pytest.raises(RuntimeError, match="batch-1"):
        pkg.gemm_4bit_inference(torch.zeros(2, K, device=dev(), dtype=torch.float16), B, absmax, code, 64, torch.float16, [M, K])
    with pytest.raises(RuntimeError, match="dtype"):
        pkg.gemm_4bit_inference(torch.zeros(1, K, device=dev(), dtype=torch.float16), B, absmax, code, 64, torch.bfloat16, [M, K])
    with pytest.raises(RuntimeError, match="fp32 absmax"):
        pkg.gemm_4bit_inference(torch.zeros(1, K, device=dev(), dtype=torch.float16), B, absmax.half(), code, 64, torch.float16, [M, K])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_partial_f32_output_and_k_split_sum(dtype):
    """fp4_hip_gemv_partial: the raw f32 accumulator; column shards (re-packed as torch_bnb_fp4.parallel does)
    sum to the full product - the data path of the row-parallel layer, minus the all-reduce."""
    import torch_bnb_fp4.parallel as par

    M, K, G = 512, 4096, 8
    packed, am, x = make_case(M, K, seed=77)
    x_t = torch_values(x, dtype)
    P, A = to_dev(packed).view(-1, 1), to_dev(am)
    full = hipabi.gemv_partial(x_t, P, A, M, K, 64)
    xv = x_t.float().cpu().numpy().astype(np.float64)
    exact = c_oracle.gemv_f64(xv, packed, am, M, K, 64)
    scale = np.abs(o.dequantize_f32(packed, am, 64, M * K).reshape(M, K).astype(np.float64)) @ np.abs(xv)
    assert full.dtype == torch.float32
    assert (np.abs(full.cpu().numpy() - exact) <= 1e-5 * scale).all()
    acc = torch.zeros(M, dtype=torch.float32, device=dev())
    for r in range(G):
        p, a, (m, ks) = par.shard_cols(P, A, (M, K), 64, r, G)
        acc += hipabi.gemv_partial(x_t[r * ks:(r + 1) * ks].contiguous(), p, a, m, ks, 64)
    assert (np.abs(acc.cpu().numpy() - exact) <= 1e-5 * scale).all()
    # row shards are plain slices: concatenated shard outputs == the unsharded GEMV, bit for bit
    y = hipabi.gemv(x_t, P, A, M, K, 64)
    ys = [hipabi.gemv(x_t, *par.shard_rows(P, A, (M, K), 64, r, G)[:2], M // G, K, 64) for r in range(G)]
    assert torch.equal(torch.cat(ys), y)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("kernel", [-1, 0, 1 | (1 << 4), 1 | (2 << 4), 1 | (2 << 10)])  # default; VALU kernel; matrix-core kernel: 1 / 2 row tiles; the persistent form wherever it applies (K = 4096)
@pytest.mark.parametrize("B", [1, 2, 4, 5, 8, 9, 13, 16])
@pytest.mark.parametrize("M,K", [(4096, 4096), (1024, 4096), (2048, 768), (66, 2048), (257, 1024), (300, 8192), (40, 14336), (33, 512),
                                 (130, 11008), (48, 1472), (70, 5120), (36, 13824)])
def test_small_batch_fused_gemm(dtype, kernel, B, M, K):
    """fp4_hip_gemm_small: 1..16 activation rows against the float64 product, same bar as the GEMV (one rounding);
    both kernels, every shape class (VALU band splits, MFMA blocks-per-wave 1/2/4/8, ragged M, B not a power of two).  K % 512 != 0
    beyond the VALU kernel's reach (11008 = Llama-2-7B's down projection, 1472 = 23 blocks) lands on the one-pass kernels of
    gemm_wide_fp4.hip with one column tile and a ragged last step; so do, in the default dispatch, K = 5120 from 5 rows and K = 13824
    (one or two quant blocks per wave and pass on the 16-row matrix-core kernel)."""
    c = case(M, K, seed=shape_seed(M, K))
    x_t, b_t, exact, scale = c.rows(B, B, dtype)
    kernel_is_mfma = kernel & 1
    valu_ok = B <= 8 and ((K // 32 <= 128) or (K // 32 <= 256 and B <= 4) or (K // 32 <= 512 and B <= 2))
    mfma_ok = K % 512 == 0
    hipabi.set_variant("gemm_small", kernel)
    try:
        assert valu_ok or mfma_ok or K % 64 == 0
        y = hipabi.gemm_small(x_t, c.P, c.A, M, K, 64, bias=b_t)
    finally:
        hipabi.set_variant("gemm_small", -1)
    assert_within_bar(y, exact, scale, dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B", [1, 2, 3, 4, 5, 7, 8, 9, 13, 16])
@pytest.mark.parametrize("M", [33, 1024, 4096, 8200, 20000])
def test_small_batch_persistent_kernel(dtype, B, M):
    """The persistent matrix-core kernel (K = 4096; x image for <= 4 / <= 8 rows, register-resident B fragments above; by
    default only on tall weights) forced at every size:
    fewer tiles than workgroups, exactly one tile each, ragged last tile, and several tiles per workgroup (M = 20000:
    1250 tiles over 512 / 256 workgroups, i.e. the prefetch-next-tile loop).  Same bar as every other small-batch kernel, next to
    the one-shot kernel on the same operands."""
    K = 4096
    c = case(M, K, seed=shape_seed(M, K))
    x_t, b_t, exact, scale = c.rows(B, B + M, dtype)
    P, A = c.P, c.A
    try:
        hipabi.set_variant("gemm_small", 1 | (2 << 10))
        y = hipabi.gemm_small(x_t, P, A, M, K, 64, bias=b_t)
        hipabi.set_variant("gemm_small", 1 | (1 << 10) | (1 << 4))
        y_one_shot = hipabi.gemm_small(x_t, P, A, M, K, 64, bias=b_t)
    finally:
        hipabi.set_variant("gemm_small", -1)
    # two different K partitions over the eight waves (one 8-block slice per wave vs two 4-block passes): both meet the bar, and they
    # agree to within the rounding of the last f32 additions
    assert_within_bar(y, exact, scale, dtype)
    assert_within_bar(y_one_shot, exact, scale, dtype)
    assert float((y == y_one_shot).float().mean()) > 0.9


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B", [17, 24, 32, 33, 64])
@pytest.mark.parametrize("M,K", [(4096, 4096), (300, 8192), (66, 2048)])
def test_small_batch_beyond_16_rows_chunked_path(dtype, B, M, K):
    """17..64 rows with the one-pass kernel switched off: evenly split over ceil(B/16) launches; each row of the result equals
    what a single-chunk call on that row's chunk gives (bit for bit) and meets the GEMV bar against the float64 product."""
    c = case(M, K, seed=shape_seed(M, K))
    x_t, _, exact, scale = c.rows(B, B * 3 + K, dtype, bias=False)
    P, A = c.P, c.A
    hipabi.set_variant("gemm_wide", 0)
    try:
        y = hipabi.gemm_small(x_t, P, A, M, K, 64)
    finally:
        hipabi.set_variant("gemm_wide", -1)
    chunks = -(-B // 16)
    per = -(-B // chunks)
    parts = [hipabi.gemm_small(x_t[b0:b0 + per].contiguous(), P, A, M, K, 64) for b0 in range(0, B, per)]
    assert torch.equal(y, torch.cat(parts))
    assert_within_bar(y, exact, scale, dtype)


WIDE_SHAPES = [(4096, 4096), (1024, 4096), (300, 8192), (66, 2048), (33, 512), (130, 14336), (257, 1024), (130, 11008), (66, 768), (40, 1472)]


def wide_case(M, K):
    """One weight per shape (session-cached), with its exact f32 values as float64."""
    c = case(M, K, seed=shape_seed(M, K))
    return c.packed, c.am, c.w64()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cfg", [1, 2, 3, 4, 5])  # 16 / 32 / 64 / 128 rows per workgroup; 5: the barrier-free 16-row kernel (up to 32 rows)
@pytest.mark.parametrize("B", [17, 32, 33, 48, 49, 64])
@pytest.mark.parametrize("M,K", WIDE_SHAPES)
def test_wide_batch_one_pass_kernel(dtype, cfg, B, M, K):
    """17..64 activation rows in ONE pass over the weight (gemm16_wide_ring8_kernel / gemm16_wide_ring_kernel: every stream through
    LDS by LDS-DMA, 2..4 column tiles per decoded weight fragment), every workgroup shape forced at every size: ragged M (33, 66,
    130, 257, 300: clamped rows, last workgroup partly empty), B not a multiple of 16 (clamped columns), 1..56 steps per K slice
    (fewer steps than the weight ring is deep, exactly as many, more).  Every row of the result meets
    the GEMV bar against the float64 product (bias added before the one rounding, as F.linear does)."""
    c = case(M, K, seed=shape_seed(M, K))
    x_t, b_t, exact, scale = c.rows(B, B * 7 + K, dtype)  # (the float64 product is spot-checked against the C oracle's loop there)
    hipabi.set_variant("gemm_wide", cfg)
    try:
        y = hipabi.gemm_small(x_t, c.P, c.A, M, K, 64, bias=b_t)
    finally:
        hipabi.set_variant("gemm_wide", -1)
    assert_within_bar(y, exact, scale, dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B", [2, 16, 33, 48, 64, 100])
@pytest.mark.parametrize("M,K", [(4096, 14336), (4096, 11008), (300, 8192), (130, 13824), (2050, 8256), (16, 8192)])
def test_split_k_with_workspace(dtype, B, M, K):
    """fp4_hip_gemm_small_ws on and around the shapes its split-K path takes (33..64 rows, short weight, K >= 8192; ragged last K slice
    at 11008 and 8256, ragged rows, more tiles than waves and fewer): partial sums through the workspace + the reducing launch meet the GEMV bar
    against the float64 product; twice the same call gives the same bits (fixed summation order); without a workspace, or with one
    that is too small, the call is fp4_hip_gemm_small_fused and agrees with it bit for bit."""
    c = case(M, K, seed=shape_seed(M, K))
    x_t, b_t, exact, scale = c.rows(B, B * 13 + K, dtype)
    rng = np.random.default_rng(B * 17 + K)
    P, A = c.P, c.A
    y, want_bytes = hipabi.gemm_small_ws(x_t, P, A, M, K, 64, bias=b_t)
    chunk = B if B <= 64 else (B + 1) // 2  # 65..128 rows: two even chunks through the same workspace
    assert want_bytes == (-(-(K // 64) // 8) * chunk * M * 4 if (chunk >= 33 and 16 <= M < 6144) else 0)  # (256 CUs: short = M < 6144)
    assert_within_bar(y, exact, scale, dtype)
    y2, _ = hipabi.gemm_small_ws(x_t, P, A, M, K, 64, bias=b_t)
    assert torch.equal(y, y2)
    small = torch.empty(max(16, want_bytes // 2), dtype=torch.uint8, device=dev())
    y3, _ = hipabi.gemm_small_ws(x_t, P, A, M, K, 64, bias=b_t, workspace=small)
    assert torch.equal(y3, hipabi.gemm_small_fused(x_t, P, A, M, K, 64, b_t, None))
    if want_bytes == 0:
        assert torch.equal(y, y3)  # not a split-K shape: the workspace entry IS the plain one
    # epilogues through the reducing launch: residual bit-exact on top of the plain result, gate|up within one ulp of torch's ops
    r_t = torch_values(rng.standard_normal((B, M)).astype(np.float32), dtype)
    yr, _ = hipabi.gemm_small_ws(x_t, P, A, M, K, 64, bias=b_t, residual=r_t)
    want = o.linear_epilogue(y.float().cpu().numpy(), "float16" if dtype == torch.float16 else "bfloat16", None, r_t.float().cpu().numpy())
    assert np.array_equal(yr.float().cpu().numpy().view(np.uint32), np.asarray(want, np.float32).view(np.uint32))
    if M % 2 == 0:
        gu, _ = hipabi.gemm_small_ws(x_t, P, A, M, K, 64, bias=b_t, epilogue=hipabi.EPILOGUE_SILU_MUL_PAIRS)
        ref = torch.nn.functional.silu(y[:, 0::2]) * y[:, 1::2]
        d = (gu.view(torch.int16).int() - ref.view(torch.int16).int()).abs()
        assert int(d.max()) <= 1 and float((d == 0).float().mean()) >= 0.995


def test_small_batch_random_shapes_against_float64():
    """200 seeded random cases through fp4_hip_gemm_small: 1..128 rows, M in 1..200, K = 64 * (1..40) - fewer quant blocks than K
    slices, ragged last steps, single-row weights, every workgroup shape of the one-pass kernels forced or chosen - against the
    float64 product (the GEMV's bar).  Shapes no kernel covers must be refused, never computed wrongly."""
    rng = np.random.default_rng(20260)
    for case in range(200):
        K = 64 * int(rng.integers(1, 41))
        M = int(rng.integers(1, 201))
        B = int(rng.integers(1, 129))
        dtype = (torch.bfloat16, torch.float16)[case & 1]
        cfg = int(rng.integers(-1, 6))
        w = (rng.standard_normal(M * K) * 0.03).astype(np.float32)
        packed, am = c_oracle.quantize(w, 64)
        x_t = torch_values(rng.standard_normal((B, K)).astype(np.float32), dtype)
        b_t = torch_values(rng.standard_normal(M).astype(np.float32) * 0.1, dtype) if case % 3 else None
        hipabi.set_variant("gemm_wide", cfg)
        try:
            rc, y = hipabi.gemm_small(x_t, to_dev(packed), to_dev(am), M, K, 64, bias=b_t, expect_ok=None)
        finally:
            hipabi.set_variant("gemm_wide", -1)
        if rc != hipabi.OK:
            # refused: only legal where neither the 16-row kernels (K % 512, or <= 8 rows on short rows) nor the one-pass ones
            # (switched off by cfg 0 for > 16 rows) apply
            assert rc == hipabi.ERR_UNSUPPORTED and cfg == 0 and B > 8 and K % 512, (case, B, M, K, cfg, hipabi.last_error())
            continue
        wd = o.dequantize_f32(packed, am, 64, M * K).reshape(M, K).astype(np.float64)
        xv = x_t.float().cpu().numpy().astype(np.float64)
        bv = np.zeros(M) if b_t is None else b_t.float().cpu().numpy().astype(np.float64)
        exact = xv @ wd.T + bv
        scale = np.abs(xv) @ np.abs(wd).T + np.abs(bv)
        err = np.abs(y.float().cpu().numpy().astype(np.float64) - exact)
        assert (err <= HALF_ULP[dtype] * 1.01 * np.abs(exact) + 1e-5 * scale + 1e-30).all(), (case, B, M, K, cfg, float(err.max()))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_wide_batch_default_dispatch_and_row_limits(dtype):
    """Default dispatch: 17..64 rows take the one-pass kernel (the result differs from the 16-row launches only in summation
    order: both within the bar, so about one ulp apart at most); 65..128 rows are split into two even chunks of at most 64;
    129 rows are refused."""
    M, K = 4096, 4096
    packed, am, _ = make_case(M, K, seed=5)
    P, A = to_dev(packed), to_dev(am)
    rng = np.random.default_rng(11)
    x_t = torch_values(rng.standard_normal((128, K)).astype(np.float32), dtype)
    y48 = hipabi.gemm_small(x_t[:48].contiguous(), P, A, M, K, 64)
    hipabi.set_variant("gemm_wide", 0)
    try:
        y48_chunked = hipabi.gemm_small(x_t[:48].contiguous(), P, A, M, K, 64)
    finally:
        hipabi.set_variant("gemm_wide", -1)
    a, b = y48.float(), y48_chunked.float()
    assert float((a == b).float().mean()) > 0.9  # mostly the same bits; elsewhere one ulp (more only where the sum cancels)
    assert torch.allclose(a, b, rtol=4 * HALF_ULP[dtype], atol=1e-4 * float(a.abs().max()))
    y100 = hipabi.gemm_small(x_t[:100].contiguous(), P, A, M, K, 64)
    assert torch.equal(y100[:50], hipabi.gemm_small(x_t[:50].contiguous(), P, A, M, K, 64))
    assert torch.equal(y100[50:], hipabi.gemm_small(x_t[50:100].contiguous(), P, A, M, K, 64))
    too_many = torch.zeros(129, K, dtype=dtype, device=dev())
    assert hipabi.gemm_small(too_many, P, A, M, K, 64, expect_ok=False) == hipabi.ERR_INVALID
    # what stays outside every small-batch kernel: more than 8 rows on a blocksize other than 64 -> the caller's dequant + GEMM
    A128 = torch.ones(M * K // 128, device=dev())
    rc = hipabi.gemm_small(x_t[:12].contiguous(), P, A128, M, K, 128, expect_ok=False)
    assert rc == hipabi.ERR_UNSUPPORTED and "dequant + GEMM" in hipabi.last_error()


@pytest.mark.parametrize("dtype", DTYPES)
def test_unaligned_operands_take_the_generic_path(dtype):
    M, K = 48, 1024
    packed, am, x = make_case(M, K, seed=123)
    x_full = torch_values(np.concatenate([[0.0], x]), dtype)
    x_t = x_full[1:]  # element offset: not 16-byte aligned
    p_full = to_dev(np.concatenate([np.zeros(1, np.uint8), packed]))
    y = hipabi.gemv(x_t, p_full[1:], to_dev(am), M, K, 64)
    check(y, x_t, packed, am, M, K, 64, dtype)


def test_degenerate_sizes():
    P, A = torch.zeros(64, dtype=torch.uint8, device=dev()), torch.ones(2, device=dev())
    x = torch.ones(64, dtype=torch.bfloat16, device=dev())
    assert hipabi.gemv(x, P, A, 0, 64, 64).numel() == 0  # M = 0
    y = hipabi.gemv(x[:0], P, A, 2, 0, 64)  # K = 0: empty sum
    assert torch.equal(y.float(), torch.zeros(2, device=dev()))


def test_gemv_beyond_32bit_weight_indices():
    """M*K > 2^31 weights: the last rows of a 2^19 x 4608 weight must still be addressed correctly (64-bit chunk index)."""
    M, K, bs = 1 << 19, 4608, 64  # 2.4e9 weights, 1.2 GB packed
    gen = torch.Generator(device=dev()).manual_seed(4)
    packed = torch.randint(0, 256, (M * K // 2,), dtype=torch.uint8, device=dev(), generator=gen)
    absmax = torch.rand(M * K // bs, device=dev(), generator=gen) * 0.05 + 0.01
    x = torch.randn(K, device=dev(), generator=gen).to(torch.bfloat16)
    y = hipabi.gemv(x, packed, absmax, M, K, bs)
    torch.cuda.synchronize()
    xv = x.float().cpu().numpy().astype(np.float64)
    for r0 in (0, (1 << 31) // K - 2, M - 4):
        rows = 4
        p = packed[r0 * K // 2:(r0 + rows) * K // 2].cpu().numpy()
        a = absmax[r0 * K // bs:(r0 + rows) * K // bs].cpu().numpy()
        exact = c_oracle.gemv_f64(xv, p, a, rows, K, bs)
        scale = np.abs(o.dequantize_f32(p, a, bs, rows * K).reshape(rows, K).astype(np.float64)) @ np.abs(xv)
        got = y[r0:r0 + rows].float().cpu().numpy()
        assert (np.abs(got - exact) <= 2.0**-8 * 1.01 * np.abs(exact) + 1e-5 * scale).all(), r0
    del packed, absmax, y
    torch.cuda.empty_cache()


@pytest.mark.parametrize("M,K,B", [(4096, 4096, 2), (2048, 768, 8), (64, 2048, 5), (300, 1000, 3)])
def test_small_batch_f32_rows_are_the_f32_gemv_row_by_row(M, K, B):
    """f32 activations in fp4_hip_gemm_small / _fused: up to 8 rows, each computed by the f32 GEMV kernel - bit-identical to the
    single-row call (with bias, with residual), inside the float64 bar, FP4_ERR_UNSUPPORTED above 8 rows and for the gated epilogue."""
    c = case(M, K)
    x_t, b_t, exact, scale = c.rows(B, 91, torch.float32)
    got = hipabi.gemm_small(x_t, c.P, c.A, M, K, 64, b_t)
    assert got.shape == (B, M) and got.dtype == torch.float32
    for b in range(B):
        assert torch.equal(got[b], hipabi.gemv(x_t[b].contiguous(), c.P, c.A, M, K, 64, b_t)), b
    assert_within_bar(got, exact, scale, torch.float32)
    res = torch_values(np.random.default_rng(7).standard_normal((B, M)).astype(np.float32), torch.float32)
    fused = hipabi.gemm_small_fused(x_t, c.P, c.A, M, K, 64, b_t, res)
    for b in range(B):
        assert torch.equal(fused[b], hipabi.gemv_fused(x_t[b].contiguous(), c.P, c.A, M, K, 64, b_t, res[b].contiguous())), b
    nine = torch.zeros(9, K, dtype=torch.float32, device=dev())
    assert hipabi.gemm_small(nine, c.P, c.A, M, K, 64, expect_ok=False) == hipabi.ERR_UNSUPPORTED and "dequant + GEMM" in hipabi.last_error()
    if M % 2 == 0:
        assert hipabi.gemm_small_fused(x_t, c.P, c.A, M, K, 64, None, None, hipabi.EPILOGUE_SILU_MUL_PAIRS, expect_ok=False) == hipabi.ERR_UNSUPPORTED
