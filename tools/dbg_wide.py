import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "tests"), os.path.join(REPO, "torch-bnb-fp4_amd"), REPO]
import numpy as np, torch
import hipabi
from oracle import fp4_oracle as o, c_oracle
M, K, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(1)
w = (rng.standard_normal(M * K) * 0.03).astype(np.float32)
packed, am = c_oracle.quantize(w, 64)
x = torch.from_numpy(rng.standard_normal((B, K)).astype(np.float32)).cuda().bfloat16()
P, A = torch.from_numpy(packed).cuda(), torch.from_numpy(am).cuda()
wd = o.dequantize_f32(packed, am, 64, M * K).reshape(M, K).astype(np.float64)
exact = x.float().cpu().numpy().astype(np.float64) @ wd.T
for cfg in [int(c) for c in sys.argv[4:]]:
    hipabi.set_variant("gemm_wide", cfg)
    y = hipabi.gemm_small(x, P, A, M, K, 64).float().cpu().numpy()
    err = np.abs(y - exact) / (np.abs(exact) + 1e-3)
    bad = err > 0.02
    print("cfg", cfg, "bad share", bad.mean(), "bad rows (weight rows) of batch 0:", np.nonzero(bad[0])[0][:40], "bad batch cols for row 0:", np.nonzero(bad[:, 0])[0][:40])
    print("   per 16-row tile bad share:", [round(float(bad[:, t * 16:(t + 1) * 16].mean()), 2) for t in range(min(8, M // 16))])
    if bad.any():
        bi, ri = np.nonzero(bad)
        print("   bad batch cols histogram:", np.bincount(bi, minlength=B))
        print("   bad row-in-tile histogram:", np.bincount(ri % 16, minlength=16), " tiles:", np.bincount(ri // 16))
        k = 0
        for b_, r_ in list(zip(bi, ri))[:6]:
            print("   ", b_, r_, "got", y[b_, r_], "exact", exact[b_, r_])
    if bad.any():
        xv = x.float().cpu().numpy().astype(np.float64)
        nblk = K // 64
        for b_, r_ in list(zip(bi, ri))[:3]:
            parts = np.array([sum(float(xv[b_, jb * 64:(jb + 1) * 64] @ wd[r_, jb * 64:(jb + 1) * 64]) for jb in range(wk, nblk, 8)) for wk in range(8)])
            print("   elem", b_, r_, "diff exact-got", exact[b_, r_] - y[b_, r_], "slice partials", np.round(parts, 4))
            # hypothesis: some slice used another activation row
            for b2 in range(B):
                alt = np.array([sum(float(xv[b2, jb * 64:(jb + 1) * 64] @ wd[r_, jb * 64:(jb + 1) * 64]) for jb in range(wk, nblk, 8)) for wk in range(8)])
                for wk in range(8):
                    if abs((exact[b_, r_] - parts[wk] + alt[wk]) - y[b_, r_]) < 6e-3 and b2 != b_:
                        print("      matches if slice", wk, "used activation row", b2)
    if bad.any():
        code = wd.reshape(M, nblk, 64) / am.reshape(M, nblk, 1).astype(np.float64)  # 12*code/12 values before the block scale
        amr = am.reshape(M, nblk).astype(np.float64)
        for b_, r_ in list(zip(bi, ri))[:4]:
            dots = np.array([xv[b_, jb * 64:(jb + 1) * 64] @ code[r_, jb] for jb in range(nblk)])
            print("   elem", b_, r_, "got", y[b_, r_], "exact", float(dots @ amr[r_]), "with scales of row-16:", float(dots @ amr[r_ - 16]),
                  "| using activation row b^1:", float(np.array([xv[b_ ^ 1, jb * 64:(jb + 1) * 64] @ code[r_, jb] for jb in range(nblk)]) @ amr[r_]))
            for r2 in range(M):
                if abs(float(dots @ amr[r2]) - y[b_, r_]) < 4e-3 * max(1, abs(y[b_, r_])):
                    print("      matches scales of row", r2)
            for b2 in range(B):
                d2 = np.array([xv[b2, jb * 64:(jb + 1) * 64] @ code[r_, jb] for jb in range(nblk)])
                for r2 in (r_, r_ - 16):
                    if abs(float(d2 @ amr[r2]) - y[b_, r_]) < 4e-3 * max(1, abs(y[b_, r_])):
                        print("      matches activation row", b2, "with scales of row", r2)
