import numpy as np
import torch

NPDT = {torch.float16: "float16", torch.float32: "float32", torch.bfloat16: "bfloat16"}


def dev():
    return torch.device("cuda", 0)


def to_dev(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def bits(t: torch.Tensor) -> np.ndarray:
    """Raw bit patterns of a float tensor as an integer numpy array."""
    t = t.detach().contiguous().cpu()
    if t.dtype == torch.float32:
        return t.view(torch.int32).numpy().view(np.uint32)
    return t.view(torch.int16).numpy().view(np.uint16)


def np_bits(a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a.view(np.uint16)


def torch_values(x: np.ndarray, dtype: torch.dtype) -> torch.Tensor:
    """float array -> device tensor of ``dtype`` (rounded by torch, RNE)."""
    return torch.from_numpy(np.asarray(x, np.float32)).to(dtype).to(dev())


# ---- session-wide cache of test weights and their float64 answers -------------------------------------------------------------------
# A parity case costs ~1 s of HOST time at the decode shapes (quantise 58 M weights with the C oracle, dequantise them again for the
# |W| |x| tolerance term, the float64 product) and the parametrised GPU tests repeat it per kernel variant and dtype: the oracle work
# is done once per (shape, seed) here and shared.  Nothing of the product is cached - kernels run in every test - and the oracle
# stays the checker: every float64 product taken through BLAS is spot-checked against the C oracle's own loop.
import collections

from oracle import c_oracle, fp4_oracle as _o

HALF_ULP = {torch.bfloat16: 2.0**-8, torch.float16: 2.0**-11, torch.float32: 0.0}
_CASES: "collections.OrderedDict" = collections.OrderedDict()
_W64: "collections.OrderedDict" = collections.OrderedDict()
_MAX_CASES, _MAX_W64_BYTES = 40, 1 << 30


class Case:
    """One FP4 weight [M, K] quantised by the CPU oracle from a seeded N(0, wscale) matrix, plus seeded activations: ``x`` [K],
    ``bias`` [M] (x 0.1) and ``res`` [M], drawn in that order."""

    def __init__(self, M, K, bs, seed, wscale):
        self.M, self.K, self.bs, self.key = M, K, bs, (M, K, bs, seed, wscale)
        rng = np.random.default_rng(seed)
        w = (rng.standard_normal(M * K) * wscale).astype(np.float32)
        self.packed, self.am = c_oracle.quantize(w, bs) if bs >= 2 else _o.quantize_fp4(w, bs)
        self.x = rng.standard_normal(K).astype(np.float32)
        self.bias = rng.standard_normal(M).astype(np.float32) * 0.1
        self.res = rng.standard_normal(M).astype(np.float32)
        self._dev = None
        self._exact = {}
        self._rows = {}

    # device copies of the packed bytes and the scales (made once; the kernels only read them)
    @property
    def P(self):
        if self._dev is None:
            self._dev = (to_dev(self.packed), to_dev(self.am))
        return self._dev[0]

    @property
    def A(self):
        self.P
        return self._dev[1]

    def w64(self):
        """The exact f32 weight values as a float64 matrix (oracle dequant), LRU-cached by bytes."""
        if self.key in _W64:
            _W64.move_to_end(self.key)
            return _W64[self.key]
        w = _o.dequantize_f32(self.packed, self.am, self.bs, self.M * self.K).reshape(self.M, self.K).astype(np.float64)
        _W64[self.key] = w
        while sum(v.nbytes for v in _W64.values()) > _MAX_W64_BYTES and len(_W64) > 1:
            _W64.popitem(last=False)
        return w

    def abs_w_times(self, xabs: np.ndarray) -> np.ndarray:
        """|W| @ |x| in float64 without holding the whole float64 matrix (row chunks)."""
        M, K, bs = self.M, self.K, self.bs
        if self.key in _W64 or M * K <= (1 << 22):
            return np.abs(self.w64()) @ xabs
        out = np.zeros(M)
        step = max(1, (1 << 22) // K)
        for r0 in range(0, M, step):
            r1 = min(M, r0 + step)
            if (r0 * K) % bs or (r1 * K) % bs:  # blocks straddle rows only for exotic block sizes: take the simple path
                return np.abs(self.w64()) @ xabs
            wd = _o.dequantize_f32(self.packed[r0 * K // 2:r1 * K // 2], self.am[r0 * K // bs:r1 * K // bs], bs, (r1 - r0) * K)
            out[r0:r1] = np.abs(wd.reshape(r1 - r0, K).astype(np.float64)) @ xabs
        return out

    def exact(self, x_t: torch.Tensor, tag=None):
        """(float64 product of the oracle, |W| |x|) for an activation vector; memoised under ``tag`` (e.g. the dtype of the
        case's own ``x``) when one is given."""
        if tag is not None and tag in self._exact:
            return self._exact[tag]
        xv = x_t.float().cpu().numpy().astype(np.float64).reshape(-1)
        M, K, bs = self.M, self.K, self.bs
        ex = c_oracle.gemv_f64(xv, self.packed, self.am, M, K, bs) if K % 2 == 0 else _o.gemv_exact(xv, self.packed, self.am, M, K, bs)
        out = (ex, self.abs_w_times(np.abs(xv)))
        if tag is not None:
            self._exact[tag] = out
        return out

    def rows(self, B: int, seed: int, dtype: torch.dtype, bias: bool = True):
        """A seeded batch of ``B`` activation rows (+ bias) rounded to ``dtype`` on the device, with the float64 product
        ``x @ W^T + bias`` of the exact f32 weights and the tolerance scale ``|x| @ |W|^T + |bias|`` - memoised.  The BLAS product
        is spot-checked against the C oracle's own loop on one row."""
        key = (B, seed, dtype, bias)
        if key not in self._rows:
            rng = np.random.default_rng(seed)
            x = rng.standard_normal((B, self.K)).astype(np.float32)
            bv32 = rng.standard_normal(self.M).astype(np.float32) * 0.1
            x_t = torch_values(x, dtype)
            b_t = torch_values(bv32, dtype) if bias else None
            xv = x_t.float().cpu().numpy().astype(np.float64)
            bv = b_t.float().cpu().numpy().astype(np.float64) if bias else np.zeros(self.M)
            w = self.w64()
            exact = xv @ w.T + bv
            spot = c_oracle.gemv_f64(xv[B - 1], self.packed, self.am, self.M, self.K, self.bs) + bv
            assert np.allclose(exact[B - 1], spot, rtol=1e-12, atol=1e-12)
            scale = np.abs(xv) @ np.abs(w).T + np.abs(bv)
            if len(self._rows) >= 24:
                self._rows.pop(next(iter(self._rows)))
            self._rows[key] = (x_t, b_t, exact, scale)
        return self._rows[key]


def case(M, K, bs=64, seed=0, wscale=0.02) -> Case:
    key = (M, K, bs, seed, wscale)
    if key in _CASES:
        _CASES.move_to_end(key)
        return _CASES[key]
    c = _CASES[key] = Case(M, K, bs, seed, wscale)
    while len(_CASES) > _MAX_CASES:
        old, _ = _CASES.popitem(last=False)
        _W64.pop(old, None)
    return c


def assert_within_bar(y: torch.Tensor, exact: np.ndarray, scale: np.ndarray, dtype: torch.dtype):
    """The GEMV bar: |y - y*| <= ulp_T(y*)/2 * 1.01 + 1e-5 * sum |x_k w_rk|."""
    got = y.float().cpu().numpy().astype(np.float64)
    tol = HALF_ULP[dtype] * 1.01 * np.abs(exact) + 1e-5 * scale + 1e-30
    err = np.abs(got - exact)
    worst = np.unravel_index(int(np.argmax(err - tol)), err.shape)
    assert (err <= tol).all(), (worst, got[worst], exact[worst], err[worst], tol[worst])
    return err
