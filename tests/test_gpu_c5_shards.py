"""C5 at its real sharding, on one GPU: the eight ranks' shards of ONE Llama-3-8B decoder layer go through the HIP kernels
serially and the reassembled layer output is compared with the float64 oracle of the UNSHARDED layer.

Layout (SURVEY section 8e "C5 layout", Megatron pairing; hidden 4096, intermediate 14336, 32 query / 8 KV heads, G = 8):

* q (4096x4096), k and v (1024x4096): M-split -> 512 / 128 / 128 rows per rank, row-concatenated into one 768x4096 launch;
* gate and up (14336x4096 each): M-split -> 1792 rows each, row-INTERLEAVED into one 3584x4096 launch, silu(g)*u in the epilogue;
* o (4096x4096): K-split -> a re-packed 4096x512 shard per rank, f32 partials, summed in rank order, rounded once;
* down (4096x14336): K-split -> 4096x1792 per rank, same.

Bars: (1) the M-split parts equal the corresponding rows of the unsharded GEMV BIT FOR BIT (rows are independent; the
shard is a slice of the same bytes); (2) every reassembled output meets the GEMV's float64 half-ulp bar (test_gpu_gemv.py)
against the oracle of the unsharded weight; the K-split sum is additionally rebuilt on the host from the eight
fp4_hip_gemv_partial outputs (f32 adds in rank order, one rounding) - fp4_hip_allreduce_oneshot's arithmetic, which
test_gpu_allreduce_ranks.py runs on eight such o-projection partials with eight real ranks.  Only "RCCL over xGMI" stays
unmeasured after this: the transport, not the shard arithmetic.

The reference has no sharding code (its only device handling is /root/reference/csrc/torch_fp4.cpp:47)."""
import numpy as np
import pytest
import torch

import hipabi
from gpu_util import NPDT, bits, to_dev, torch_values
from oracle import c_oracle, fp4_oracle as o

pytestmark = pytest.mark.gpu
G, BS = 8, 64
H, I, KV = 4096, 14336, 1024
HALF_ULP = {torch.bfloat16: 2.0**-8, torch.float16: 2.0**-11}


@pytest.fixture(scope="module")
def layer():
    """One layer's seven FP4 weights (CPU oracle quantiser) + their device copies + the float64 |W| needed by the bars."""
    rng = np.random.default_rng(20240508)
    out = {}
    for name, (M, K) in {"q": (H, H), "k": (KV, H), "v": (KV, H), "gate": (I, H), "up": (I, H), "o": (H, H), "down": (H, I)}.items():
        w = (rng.standard_normal(M * K) * 0.02).astype(np.float32)
        packed, am = c_oracle.quantize(w, BS)
        out[name] = {"shape": (M, K), "packed": packed, "absmax": am, "P": to_dev(packed).view(-1, 1), "A": to_dev(am)}
    out["h"] = rng.standard_normal(H).astype(np.float32)
    out["a"] = rng.standard_normal(H).astype(np.float32)      # the attention output fed to o
    out["m"] = rng.standard_normal(I).astype(np.float32) * 0.5  # the gated product fed to down
    out["res"] = rng.standard_normal(H).astype(np.float32)
    return out


def exact_and_tol(w, x_t, dtype):
    M, K = w["shape"]
    xv = x_t.float().cpu().numpy().astype(np.float64)
    exact = c_oracle.gemv_f64(xv, w["packed"], w["absmax"], M, K, BS)
    scale = np.zeros(M)
    step = 512  # |W| @ |x| in row chunks: the dequantised 14336x4096 f64 matrix is 470 MB otherwise
    for r0 in range(0, M, step):
        r1 = min(M, r0 + step)
        wd = o.dequantize_f32(w["packed"][r0 * K // 2:r1 * K // 2], w["absmax"][r0 * K // BS:r1 * K // BS], BS, (r1 - r0) * K)
        scale[r0:r1] = np.abs(wd.reshape(r1 - r0, K).astype(np.float64)) @ np.abs(xv)
    return exact, HALF_ULP[dtype] * 1.01 * np.abs(exact) + 1e-5 * scale + 1e-30


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_m_split_qkv_and_gate_up_shards(layer, dtype):
    from torch_bnb_fp4 import parallel as par
    from torch_bnb_fp4.fused import interleave_rows

    x_t = torch_values(layer["h"], dtype)
    # unsharded launches (what one GPU would run)
    full = {n: hipabi.gemv(x_t, layer[n]["P"], layer[n]["A"], *layer[n]["shape"], BS) for n in ("q", "k", "v", "gate", "up")}
    got = {n: [] for n in ("q", "k", "v")}
    gated = []
    for rank in range(G):
        shards = [par.shard_rows(layer[n]["P"], layer[n]["A"], layer[n]["shape"], BS, rank, G) for n in ("q", "k", "v")]
        assert [s[2] for s in shards] == [(512, H), (128, H), (128, H)]
        P, A, (M, K) = par.concat_rows(shards, BS)
        assert (M, K) == (768, H)
        y = hipabi.gemv(x_t, P.contiguous(), A.contiguous(), M, K, BS)
        for n, part in zip(("q", "k", "v"), torch.split(y, [512, 128, 128])):
            got[n].append(part)
        # a k / v shard on its own (128 x 4096: the smallest launch of the layer) gives the same bits as inside the concatenation
        pk, ak, (mk, kk) = shards[1]
        assert torch.equal(hipabi.gemv(x_t, pk.contiguous(), ak.contiguous(), mk, kk, BS), got["k"][-1])
        sg = par.shard_rows(layer["gate"]["P"], layer["gate"]["A"], (I, H), BS, rank, G)
        su = par.shard_rows(layer["up"]["P"], layer["up"]["A"], (I, H), BS, rank, G)
        Pgu, Agu, (Mgu, Kgu) = interleave_rows((sg[0], sg[1]), (su[0], su[1]), sg[2], BS)
        assert (Mgu, Kgu) == (3584, H)
        gated.append(hipabi.gemv_fused(x_t, Pgu.contiguous(), Agu.contiguous(), Mgu, Kgu, BS, None, None, hipabi.EPILOGUE_SILU_MUL_PAIRS))
        # the interleaved shard WITHOUT the epilogue returns gate_i, up_i alternating: bit for bit the unsharded rows
        plain = hipabi.gemv(x_t, Pgu.contiguous(), Agu.contiguous(), Mgu, Kgu, BS)
        assert torch.equal(plain[0::2], full["gate"][rank * 1792:(rank + 1) * 1792])
        assert torch.equal(plain[1::2], full["up"][rank * 1792:(rank + 1) * 1792])
    for n in ("q", "k", "v"):
        whole = torch.cat(got[n])
        assert torch.equal(whole, full[n]), n  # bar 1: bit for bit the unsharded GEMV
        exact, tol = exact_and_tol(layer[n], x_t, dtype)  # bar 2: float64 oracle of the unsharded weight
        err = np.abs(whole.float().cpu().numpy().astype(np.float64) - exact)
        assert (err <= tol).all(), (n, float((err / tol).max()))
    # gate|up: the reassembled silu(g) * u against torch's own ops on the unsharded outputs (<= 1 ulp, >= 99.9 % identical: the
    # bar of test_gpu_fused.py) and against float64 end to end
    whole = torch.cat(gated)
    ref = torch.nn.functional.silu(full["gate"]) * full["up"]
    a, b = bits(whole).astype(np.int64), bits(ref).astype(np.int64)
    a, b = np.where(a & 0x8000, 0x8000 - a, a), np.where(b & 0x8000, 0x8000 - b, b)
    d = np.abs(a - b)
    assert d.max() <= 1 and (d == 0).mean() >= 0.999, (int(d.max()), float((d == 0).mean()))
    xv = x_t.float().cpu().numpy().astype(np.float64)
    g64 = c_oracle.gemv_f64(xv, layer["gate"]["packed"], layer["gate"]["absmax"], I, H, BS)
    u64 = c_oracle.gemv_f64(xv, layer["up"]["packed"], layer["up"]["absmax"], I, H, BS)
    want = g64 / (1.0 + np.exp(-g64)) * u64
    tol = HALF_ULP[dtype] * 1.02 * (np.abs(want) * 3 + 1.1 * np.abs(g64) * np.abs(u64)) + 1e-6
    assert (np.abs(whole.float().cpu().numpy().astype(np.float64) - want) <= tol).all()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("name,vec", [("o", "a"), ("down", "m")])
def test_k_split_o_and_down_shards(layer, dtype, name, vec):
    from torch_bnb_fp4 import parallel as par

    w = layer[name]
    M, K = w["shape"]
    ks = K // G
    assert ks == (512 if name == "o" else 1792)
    x_t = torch_values(layer[vec], dtype)
    res_t = torch_values(layer["res"], dtype)
    parts = []
    for rank in range(G):
        p, a, local = par.shard_cols(w["P"], w["A"], (M, K), BS, rank, G)
        assert local == (M, ks) and p.numel() == M * ks // 2 and a.numel() == M * ks // BS
        # the re-pack is a byte / scale permutation: the shard dequantises to exactly the column range of the full weight
        if rank in (0, 5):
            d_full = hipabi.dequantize(w["P"], w["A"], BS, M * K, torch.float32).view(M, K)[:, rank * ks:(rank + 1) * ks]
            d_shard = hipabi.dequantize(p, a, BS, M * ks, torch.float32).view(M, ks)
            assert torch.equal(d_full, d_shard)
        parts.append(hipabi.gemv_partial(x_t[rank * ks:(rank + 1) * ks].contiguous(), p, a, M, ks, BS))
    # what the all-reduce computes: f32 adds in rank order 0..G-1, one rounding to T, then the rounded residual add
    acc = torch.zeros(M, dtype=torch.float32, device=parts[0].device)
    for t in parts:
        acc = acc + t
    y = acc.to(dtype)
    exact, tol = exact_and_tol(w, x_t, dtype)
    err = np.abs(y.float().cpu().numpy().astype(np.float64) - exact)
    assert (err <= tol).all(), (name, float((err / tol).max()))
    # the unsharded GEMV meets the same bar, and the two agree to within one rounding step of T wherever both are in tolerance
    y_full = hipabi.gemv(x_t, w["P"], w["A"], M, K, BS)
    err_full = np.abs(y_full.float().cpu().numpy().astype(np.float64) - exact)
    assert (err_full <= tol).all()
    # residual epilogue of the reduction: T(T(sum) + residual), as the unsharded fused layer rounds it
    want = o.linear_epilogue(y.float().cpu().numpy(), NPDT[dtype], None, res_t.float().cpu().numpy())
    got = (y.float() + res_t.float()).to(dtype)
    assert np.array_equal(got.float().cpu().numpy().view(np.uint32), np.asarray(want, np.float32).view(np.uint32))
