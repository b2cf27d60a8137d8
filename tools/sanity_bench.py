#!/usr/bin/env python3
"""The reference's own harness, restated for this package (BASELINE config 3): speed of the 6-layer MLP
`TestModel(768, 2048, 4, 64)` (sanity_check.py:38-50,65-122; 3 FP4 layers + 3 dense applications + 4 GELU, see
SURVEY 0.2-11) for GEMV ([1,768]) and GEMM ([2,768]) inputs in fp32/fp16/bf16, dense vs torch-bnb-fp4, timed with
torch.utils.benchmark.Timer.adaptive_autorange like the reference (mean/median/iqr in us per forward), and the
accuracy check of sanity_check.py:130-171 (mean |dense - fp4| for three input shapes; band 0.045-0.065).
bitsandbytes is not available on this platform, so its column of the README table is absent."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "torch-bnb-fp4_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
from torch import nn  # noqa: E402
from torch.utils.benchmark import Timer  # noqa: E402

import torch_bnb_fp4 as pkg  # noqa: E402


class TinyModel(nn.Module):
    def __init__(self, i, o):
        super().__init__()
        self.in_proj = nn.Linear(i, o)

    def forward(self, x):
        return self.in_proj(x)


class TestModel(nn.Module):
    def __init__(self, in_dim, hidden, num_hidden, out_dim):
        super().__init__()
        self.in_proj = nn.Linear(in_dim, hidden)
        self.blocks = nn.Sequential(*([nn.GELU(), nn.Linear(hidden, hidden)] * num_hidden))
        self.out_proj = nn.Linear(hidden, out_dim)

    def forward(self, x):
        return self.out_proj(self.blocks(self.in_proj(x)))


def time_run(model, inputs):
    t = Timer("model(inputs)", globals={"model": model, "inputs": inputs})
    t.adaptive_autorange()  # discarded, like the reference (sanity_check.py:78)
    r1, r2 = t.adaptive_autorange(), t.adaptive_autorange()
    m = r1.merge([r1, r2])[0]
    return {"mean": m.mean * 1e6, "median": m.median * 1e6, "iqr": m.iqr * 1e6}


def check_speed(dtype, kind):
    torch.manual_seed(10)
    gen = torch.Generator("cuda").manual_seed(10)
    model = TestModel(768, 2048, 4, 64).cuda().type(dtype)
    x = torch.randn(1 if kind == "gemv" else 2, 768, generator=gen, device="cuda").type(dtype)
    with torch.inference_mode():
        dense = time_run(model, x)
        model = pkg.recursively_replace_with_fp4_linear(model, as_dtype=dtype, device=model.in_proj.weight.device)
        ours = time_run(model, x)
        g = torch.cuda.CUDAGraph()
        sx = x.clone()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            model(sx)
            torch.cuda.synchronize()
            with torch.cuda.graph(g):
                model(sx)
        torch.cuda.synchronize()
        graph = Timer("g.replay(); torch.cuda.synchronize()", globals={"g": g, "torch": torch}).adaptive_autorange()
    return {"dtype": str(dtype).replace("torch.", ""), "kind": kind, "pytorch_dense_us": dense, "torch_bnb_fp4_amd_us": ours,
            "torch_bnb_fp4_amd_graph_replay_us": graph.median * 1e6}


def check(dtype):
    torch.manual_seed(10)
    gen = torch.Generator("cuda").manual_seed(10)
    model = TinyModel(256, 256).cuda().type(dtype)
    hijack = TinyModel(256, 256).cuda().type(dtype)
    hijack.load_state_dict(model.state_dict())
    hijack = pkg.recursively_replace_with_fp4_linear(hijack)
    out = {}
    with torch.inference_mode():
        for name, shape in (("gemv_3dim", (1, 1, 256)), ("gemv_2dim", (1, 256)), ("gemm_3dim", (1, 2048, 256))):
            x = torch.randn(*shape, generator=gen, device="cuda").type(dtype)
            out[name] = round((model(x) - hijack(x)).abs().mean().item(), 5)
    return out


if __name__ == "__main__":
    for dt in (torch.float32, torch.float16, torch.bfloat16):
        for kind in ("gemv", "gemm"):
            r = check_speed(dt, kind)
            print(json.dumps({k: ({kk: round(vv, 2) for kk, vv in v.items()} if isinstance(v, dict) else (round(v, 2) if isinstance(v, float) else v))
                              for k, v in r.items()}), flush=True)
        print(json.dumps({"dtype": str(dt).replace("torch.", ""), "elementwise_diff_avg": check(dt), "accepted_band": [0.045, 0.065]}), flush=True)
