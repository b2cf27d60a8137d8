"""GPU tests of the nn.Module layer: TorchFP4Linear / QuantData dispatch on real kernels, model
surgery, and the reference's own acceptance check (sanity_check.py:130-179)."""
import numpy as np
import pytest
import torch
from torch import nn

from gpu_util import NPDT, dev
from oracle import fp4_oracle as o

pytestmark = pytest.mark.gpu
DTYPES = [torch.float32, torch.float16, torch.bfloat16]


def pkg():
    import torch_bnb_fp4

    return torch_bnb_fp4


class TinyModel(nn.Module):  # sanity_check.py:29-35
    def __init__(self, i, o_):
        super().__init__()
        self.in_proj = nn.Linear(i, o_)

    def forward(self, x):
        return self.in_proj(x)


class TestModel(nn.Module):  # sanity_check.py:38-50 (note: ONE shared Linear in four slots)
    __test__ = False

    def __init__(self, in_dim, hidden, num_hidden, out_dim):
        super().__init__()
        self.in_proj = nn.Linear(in_dim, hidden)
        self.blocks = nn.Sequential(*([nn.GELU(), nn.Linear(hidden, hidden)] * num_hidden))
        self.out_proj = nn.Linear(hidden, out_dim)

    def forward(self, x):
        return self.out_proj(self.blocks(self.in_proj(x)))


@pytest.mark.parametrize("dtype", DTYPES)
def test_acceptance_statistic_on_gpu(dtype):
    """mean |nn.Linear - TorchFP4Linear| in [0.045, 0.065] for the three input shapes (README.md:90-91)."""
    P = pkg()
    torch.manual_seed(10)
    gen = torch.Generator(device="cuda").manual_seed(10)
    model = TinyModel(256, 256).to(dev()).type(dtype)
    hijack = TinyModel(256, 256).to(dev()).type(dtype)
    hijack.load_state_dict(model.state_dict())
    hijack = P.recursively_replace_with_fp4_linear(hijack, device=dev())
    assert isinstance(hijack.in_proj, P.TorchFP4Linear)
    with torch.inference_mode():
        for shape in ((1, 1, 256), (1, 256), (1, 2048, 256)):
            x = torch.randn(*shape, generator=gen, device=dev()).type(dtype)
            stat = (model(x) - hijack(x)).abs().mean().item()
            assert 0.045 <= stat <= 0.065, (dtype, shape, stat)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("codebook", [True, False])
def test_forward_paths_against_oracle(dtype, codebook):
    P = pkg()
    torch.manual_seed(0)
    M, K = 320, 512
    lin = nn.Linear(K, M).to(dev())
    fp4 = P.TorchFP4Linear(P.swap_linear_with_bnb_linear(lin).to(dev()), use_codebook_dequant=codebook)
    qd = fp4.quant_data
    packed, am = qd.A.cpu().numpy().reshape(-1), qd.absmax.cpu().numpy()
    # quantisation happened on the GPU from the fp16-cast weight, and equals the oracle's
    want_p, want_a = o.quantize_fp4(lin.weight.detach().half().float().cpu().numpy().reshape(-1), 64)
    assert np.array_equal(packed, want_p) and np.array_equal(am, want_a)
    wq = torch.from_numpy(o.dequantize_f32(packed, am, 64, M * K, "codebook" if codebook else "tree")).view(M, K).to(dev())
    bias = lin.bias.detach().to(dtype).float()
    tol = {torch.float32: 2e-5, torch.float16: 2e-3, torch.bfloat16: 1.6e-2}[dtype]
    for shape in ((1, K), (1, 1, K), (3, K), (2, 5, K), (1, 1, 1, K)):
        x = torch.randn(*shape, device=dev()).to(dtype)
        y = fp4(x)
        assert y.shape == shape[:-1] + (M,) and y.dtype == dtype
        ref = torch.nn.functional.linear(x.float(), wq, bias)
        assert (y.float() - ref).abs().max().item() <= tol * (1 + ref.abs().max().item()), (shape, dtype)
    assert fp4(torch.empty(0, K, device=dev(), dtype=dtype)).shape == (0, M)
    # the batch>1 path really is dequantise + F.linear with the dequantised weight bit-exact
    wd = qd.dequantize()
    want = o.dequantize(packed, am, 64, M * K, NPDT[dtype], "codebook" if codebook else "tree")
    from gpu_util import bits, np_bits

    assert np.array_equal(bits(wd).reshape(-1), np_bits(want))


def test_recursive_replacement_rules():
    P = pkg()
    torch.manual_seed(10)
    model = TestModel(768, 2048, 4, 64).to(dev()).half()
    x = torch.randn(1, 768, device=dev()).half()
    dense_out = model(x)
    model = P.recursively_replace_with_fp4_linear(model, as_dtype=torch.float16, device=dev())
    # named_children() dedupes the shared Linear: only slot 1 of `blocks` is swapped
    kinds = [type(m).__name__ for m in model.blocks]
    assert kinds == ["GELU", "TorchFP4Linear", "GELU", "Linear", "GELU", "Linear", "GELU", "Linear"]
    assert isinstance(model.in_proj, P.TorchFP4Linear) and isinstance(model.out_proj, P.TorchFP4Linear)
    with torch.inference_mode():
        y1 = model(x)  # GEMV path
        y2 = model(torch.cat([x, x]))  # dequant + GEMM path
    assert y1.shape == (1, 64) and torch.isfinite(y1).all()
    assert (y1 - y2[0]).abs().max().item() < 0.05 and (y1 - dense_out).abs().mean().item() < 0.2

    class LM(nn.Module):
        def __init__(self):
            super().__init__()
            self.body = nn.Sequential(nn.Linear(64, 64), nn.ReLU())
            self.lm_head = nn.Linear(64, 100)
            self.pooler = nn.Sequential(nn.Linear(64, 64))

    lm = P.recursively_replace_with_fp4_linear(LM().to(dev()), device=dev())
    assert isinstance(lm.body[0], P.TorchFP4Linear) and isinstance(lm.lm_head, nn.Linear) and not isinstance(lm.lm_head, P.TorchFP4Linear)
    lm2 = P.recursively_replace_with_fp4_linear(LM().to(dev()), device=dev(), ignore_layer_names=["lm_head", "pooler"])
    assert isinstance(lm2.pooler[0], nn.Linear) and not isinstance(lm2.pooler[0], P.TorchFP4Linear)
    # only_replace_bnb_layers: plain nn.Linear stays, an FP4 layer is wrapped
    mixed = nn.Sequential(nn.Linear(64, 64), P.swap_linear_with_bnb_linear(nn.Linear(64, 64))).to(dev())
    mixed = P.recursively_replace_with_fp4_linear(mixed, device=dev(), only_replace_bnb_layers=True)
    assert type(mixed[0]) is nn.Linear and isinstance(mixed[1], P.TorchFP4Linear)
    # a root that is itself a Linear is converted and returned
    root = P.recursively_replace_with_fp4_linear(nn.Linear(128, 64).to(dev()), device=dev())
    assert isinstance(root, P.TorchFP4Linear) and root(torch.randn(1, 128, device=dev())).shape == (1, 64)


def test_state_dict_and_device_move():
    P = pkg()
    lin = nn.Linear(256, 128).to(dev())
    fp4 = P.TorchFP4Linear(P.swap_linear_with_bnb_linear(lin).to(dev()))
    sd = fp4.state_dict()
    assert set(sd) == {"qweight", "absmax", "code", "bias"} and sd["qweight"].dtype == torch.uint8
    x = torch.randn(1, 256, device=dev(), dtype=torch.bfloat16)
    y = fp4(x)
    fp4.cpu()
    assert fp4.qweight.device.type == "cpu" and fp4.quant_data.A.device.type == "cpu"
    fp4.to(dev())
    assert torch.equal(fp4(x), y)
    assert "TorchFP4Linear(in_features=256, out_features=128, bias=True" in repr(fp4)


def test_bitsandbytes_format_roundtrip(tmp_path):
    """TorchFP4Linear -> bitsandbytes 4-bit state-dict layout -> safetensors -> TorchFP4Linear, bit-identical outputs."""
    P = pkg()
    torch.manual_seed(3)

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.a = nn.Linear(256, 512)
            self.norm = nn.LayerNorm(512)
            self.b = nn.Linear(512, 128, bias=False)

        def forward(self, x):
            return self.b(self.norm(self.a(x)))

    net = P.recursively_replace_with_fp4_linear(Net().to(dev()), device=dev())
    x = torch.randn(1, 256, device=dev(), dtype=torch.bfloat16)
    xb = torch.randn(4, 256, device=dev(), dtype=torch.bfloat16)
    net.norm.to(torch.bfloat16)
    y, yb = net(x), net(xb)
    entries = P.fp4_linear_to_bnb_state(net.a, "a.")
    assert set(entries) == {"a.weight", "a.weight.absmax", "a.weight.quant_map", "a.weight.quant_state.bitsandbytes__fp4", "a.bias"}
    assert entries["a.weight"].dtype == torch.uint8 and entries["a.weight"].shape == (512 * 256 // 2, 1)
    import json

    meta = json.loads(bytes(entries["a.weight.quant_state.bitsandbytes__fp4"].tolist()).decode())
    assert meta["quant_type"] == "fp4" and meta["blocksize"] == 64 and meta["shape"] == [512, 256]
    path = str(tmp_path / "model.safetensors")
    P.save_fp4_model(net, path)
    fresh = Net().to(dev())
    fresh.norm.to(torch.bfloat16)
    fresh = P.load_fp4_layers(fresh, path, device=dev())
    assert isinstance(fresh.a, P.TorchFP4Linear) and isinstance(fresh.b, P.TorchFP4Linear)
    assert torch.equal(fresh(x), y) and torch.equal(fresh(xb), yb)
    with pytest.raises(KeyError):
        P.fp4_linear_from_bnb_state({"w.weight": entries["a.weight"]}, "w.")
    bad = dict(entries)
    bad["a.weight.nested_absmax"] = torch.zeros(1)
    with pytest.raises(ValueError, match="nested"):
        P.fp4_linear_from_bnb_state(bad, "a.")


def test_hip_graph_capture_and_replay_of_the_linear_shell():
    """Every op on the path is capturable (no allocation in the C ABI, current-stream launches): a decode step
    captured once replays correctly on new inputs (the reference's legacy-stream launches could not be captured)."""
    P = pkg()
    torch.manual_seed(5)
    lin1, lin2 = nn.Linear(512, 1024).to(dev()), nn.Linear(1024, 256).to(dev())
    f1 = P.TorchFP4Linear(P.swap_linear_with_bnb_linear(lin1).to(dev()))
    f2 = P.TorchFP4Linear(P.swap_linear_with_bnb_linear(lin2).to(dev()))
    static_x = torch.randn(1, 512, device=dev(), dtype=torch.bfloat16)
    static_xb = torch.randn(4, 512, device=dev(), dtype=torch.bfloat16)

    def step(x):
        return f2(torch.nn.functional.gelu(f1(x)))

    with torch.inference_mode():
        eager = step(static_x).clone()
        eager_b = step(static_xb).clone()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            g = torch.cuda.CUDAGraph()
            step(static_x), step(static_xb)
            torch.cuda.synchronize()
            with torch.cuda.graph(g):
                out, out_b = step(static_x), step(static_xb)  # GEMV path and dequant+GEMM path
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, eager) and torch.equal(out_b, eager_b)
        new_x = torch.randn(1, 512, device=dev(), dtype=torch.bfloat16)
        static_x.copy_(new_x)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, step(new_x))


def test_fuse_rows_is_bit_identical_to_separate_layers():
    P = pkg()
    torch.manual_seed(11)
    q, k, v = (P.TorchFP4Linear(P.swap_linear_with_bnb_linear(nn.Linear(1024, m).to(dev())).to(dev())) for m in (1024, 256, 256))
    qkv = P.TorchFP4Linear.fuse([q, k, v], name="qkv")
    assert qkv.out_features == 1536 and qkv.in_features == 1024
    for dtype in (torch.bfloat16, torch.float16, torch.float32):
        x = torch.randn(1, 1, 1024, device=dev(), dtype=dtype)
        for lyr in (q, k, v, qkv):
            lyr.quant_data.compute_dtype_set = False  # let every layer follow the activation dtype of this round
        want = torch.cat([q(x), k(x), v(x)], dim=-1)
        assert torch.equal(qkv(x), want)
        xb = torch.randn(5, 1024, device=dev(), dtype=dtype)
        got, ref = qkv(xb), torch.cat([q(xb), k(xb), v(xb)], dim=-1)
        assert (got.float() - ref.float()).abs().max().item() <= 2e-2 * (1 + ref.float().abs().max().item())
    with pytest.raises(ValueError):
        P.TorchFP4Linear.fuse([q, P.TorchFP4Linear(P.swap_linear_with_bnb_linear(nn.Linear(512, 64).to(dev())).to(dev()))])


def test_set_small_batch_fused_switches_the_batch_path():
    """Same model, same batched input: the fused small-batch kernels and the reference dispatch (dequant + GEMM) agree
    to rounding, and the switch reaches every converted layer."""
    import torch_bnb_fp4 as pkg

    torch.manual_seed(3)
    model = torch.nn.Sequential(torch.nn.Linear(512, 1024), torch.nn.SiLU(), torch.nn.Linear(1024, 512)).to(torch.bfloat16)
    model = pkg.recursively_replace_with_fp4_linear(model, as_dtype=torch.bfloat16, device=dev())
    x = torch.randn(6, 512, device=dev(), dtype=torch.bfloat16)
    with torch.inference_mode():
        ref = model(x)
        assert pkg.set_small_batch_fused(model, True) == 2
        assert all(m.quant_data.small_batch_fused for m in model.modules() if isinstance(m, pkg.TorchFP4Linear))
        fused = model(x)
        assert pkg.set_small_batch_fused(model, False) == 2
        again = model(x)
    assert torch.equal(again, ref)
    assert torch.allclose(fused.float(), ref.float(), rtol=2e-2, atol=2e-2)

