"""The dense GEMM behind the batch > 1 path (`qlinear*` = dequant + GEMM, reference csrc/torch_fp4.cpp:64-103): called on hipBLASLt directly with
cached plans (csrc/torch_ext.cpp, lt_linear) instead of through at::linear, whose ~18 us of host time per call made an eager small-batch
FP4 layer slower than the dense layer it replaces.  Same maths, so: both routes against the float64 product of the dequantised weight (rounded
to the activation dtype first, as the reference's dequant does), against each other, for every dtype / bias / rank / odd size, under HIP-graph
capture (first call of a process arriving under capture included), and the host-side saving itself."""
import time

import numpy as np
import pytest
import torch

from gpu_util import case, dev
from oracle import fp4_oracle as o

pytestmark = pytest.mark.gpu
BS = 64
TOL = {torch.float32: (2e-5, 2e-5), torch.float16: (2e-3, 2e-3), torch.bfloat16: (1.6e-2, 1.6e-2)}


def pkg():
    import torch_bnb_fp4

    return torch_bnb_fp4


@pytest.fixture(autouse=True)
def _default_route():
    yield
    pkg().ext.set_qlinear_gemm("hipblaslt")


def _want(c, x_t, bias_t, dtype, table):
    """float64 x @ W_T^T + b with W_T = the dequantised weight rounded to the activation dtype (what both routes multiply)."""
    w = o.dequantize(c.packed, c.am, BS, c.M * c.K, {torch.float32: "float32", torch.float16: "float16", torch.bfloat16: "bfloat16"}[dtype], table)
    if dtype == torch.float32:
        w64 = w.astype(np.float64)
    elif dtype == torch.float16:
        w64 = w.view(np.float16).astype(np.float64)
    else:
        w64 = o.bf16_bits_to_f32(w).astype(np.float64)
    want = x_t.double().cpu().numpy().reshape(-1, c.K) @ w64.reshape(c.M, c.K).T
    if bias_t is not None:
        want = want + bias_t.double().cpu().numpy()
    return want


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,K,shape", [(2048, 768, (2, 768)), (64, 2048, (6, 2048)), (4096, 4096, (3, 5, 4096)), (300, 1000, (7, 1000)),
                                      (1024, 4096, (300, 4096))])
def test_both_gemm_routes_against_float64(dtype, M, K, shape):
    P = pkg()
    c = case(M, K, seed=5)
    g = torch.Generator(device=dev()).manual_seed(M + K)
    x = torch.randn(*shape, device=dev(), generator=g).to(dtype)
    bias = (torch.randn(M, device=dev(), generator=g) * 0.1).to(dtype)
    code = P.ext.code_table("codebook").to(dev())
    A = c.P.view(-1, 1)
    rtol, atol = TOL[dtype]
    for table, ops in (("codebook", (lambda: P.ext.qlinear_codebook(x, A, c.A, code, M, K, BS), lambda: P.ext.qlinear_codebook_bias(x, A, c.A, code, M, K, BS, bias))),
                       ("tree", (lambda: P.ext.qlinear(x, A, c.A, M, K, BS), lambda: P.ext.qlinear_bias(x, A, c.A, M, K, BS, bias)))):
        for op, b in zip(ops, (None, bias)):
            want = _want(c, x, b, dtype, table)
            scale = np.abs(want).max()
            outs = {}
            for route in ("hipblaslt", "aten"):
                P.ext.set_qlinear_gemm(route)
                y = op()
                assert y.shape == (*shape[:-1], M) and y.dtype == dtype
                outs[route] = y.double().cpu().numpy().reshape(-1, M)
                err = np.abs(outs[route] - want).max()
                assert err <= rtol * scale + atol, (route, table, b is not None, err, scale)
            assert np.abs(outs["hipblaslt"] - outs["aten"]).max() <= rtol * scale + atol


def test_non_contiguous_activations_and_foreign_bias_dtype_still_work():
    P = pkg()
    M, K = 512, 1024
    c = case(M, K, seed=9)
    A = c.P.view(-1, 1)
    x = torch.randn(K, 4, device=dev()).to(torch.bfloat16).t()  # [4, K], not contiguous
    assert not x.is_contiguous()
    y = P.ext.qlinear(x, A, c.A, M, K, BS)
    P.ext.set_qlinear_gemm("aten")
    assert torch.allclose(y.float(), P.ext.qlinear(x, A, c.A, M, K, BS).float(), rtol=2e-2, atol=2e-2)
    P.ext.set_qlinear_gemm("hipblaslt")
    # a bias of another dtype is not the library path's business: at::linear decides (and raises, like the reference's F.linear would)
    with pytest.raises(RuntimeError):
        P.ext.qlinear_bias(x.contiguous(), A, c.A, M, K, BS, torch.zeros(M, device=dev(), dtype=torch.float32))
    with pytest.raises(RuntimeError):
        P.ext.set_qlinear_gemm("cublas")


def test_graph_capture_and_replay_of_the_direct_gemm():
    """A shape never seen before is planned (heuristic query, host only) during capture; replays reproduce the eager result bit for bit."""
    P = pkg()
    M, K = 768, 1536
    c = case(M, K, seed=2)
    A = c.P.view(-1, 1)
    x = torch.randn(5, K, device=dev()).to(torch.float16)
    bias = torch.randn(M, device=dev()).to(torch.float16)
    P.ext.qlinear(torch.randn(2, K, device=dev()).to(torch.float16), A, c.A, M, K, BS)  # the device's handle exists before capture
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            y = P.ext.qlinear_bias(x, A, c.A, M, K, BS, bias)
        torch.cuda.synchronize()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        got = y.clone()
    torch.cuda.current_stream().wait_stream(s)
    assert torch.equal(got, P.ext.qlinear_bias(x, A, c.A, M, K, BS, bias))


def test_the_direct_gemm_saves_host_time_in_the_layer_call():
    """What the change is for: the eager batch-2 call of an FP4 layer (dequant + GEMM) next to a dense nn.Linear of the same shape.
    The at::linear route costs a dense call plus the dequant op (measured 29.6 us against 21.5 dense on the round-5 box); the direct route
    must come out below the dense call (18.7 us there) and well below the at::linear route."""
    P = pkg()
    lin = torch.nn.Linear(2048, 2048).to(dev()).to(torch.bfloat16)
    fp4 = P.recursively_replace_with_fp4_linear(torch.nn.Linear(2048, 2048).to(torch.bfloat16), as_dtype=torch.bfloat16, device=dev())
    x = torch.randn(2, 2048, device=dev(), dtype=torch.bfloat16)

    def us(fn, n=1500):
        for _ in range(200):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6

    with torch.inference_mode():
        dense = us(lambda: lin(x))
        direct = us(lambda: fp4(x))
        P.ext.set_qlinear_gemm("aten")
        aten = us(lambda: fp4(x))
    assert direct < dense and direct < 0.8 * aten, (dense, direct, aten)
