#!/usr/bin/env python3
"""Round 5: HOST time of torch's GEMM entry points on this platform (linear / addmm / mm, with and without bias, both BLAS backends) next to a
trivial torch op - the measurement behind calling hipBLASLt directly in the qlinear* ops (csrc/torch_ext.cpp, lt_linear)."""
import time, torch
dev=torch.device("cuda",0)
def bench(name, fn, n=2000):
    for _ in range(200): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    ti=time.perf_counter()-t0; torch.cuda.synchronize(); ta=time.perf_counter()-t0
    print(f"{name:60s} host issue {ti/n*1e6:6.2f} us   with drain {ta/n*1e6:6.2f} us", flush=True)
for dt in (torch.bfloat16, torch.float32):
  for rows in (2, 64):
    x=torch.randn(rows,2048,device=dev,dtype=dt); w=torch.randn(2048,2048,device=dev,dtype=dt); b=torch.randn(2048,device=dev,dtype=dt)
    out=torch.empty(rows,2048,device=dev,dtype=dt)
    print(f"--- {dt} rows={rows}, preferred blas: {torch.backends.cuda.preferred_blas_library()}")
    with torch.inference_mode():
        bench("F.linear(x, w, b)", lambda: torch.nn.functional.linear(x,w,b))
        bench("F.linear(x, w) (no bias)", lambda: torch.nn.functional.linear(x,w))
        bench("torch.addmm(b, x, w.t())", lambda: torch.addmm(b,x,w.t()))
        bench("torch.mm(x, w.t())", lambda: torch.mm(x,w.t()))
        bench("torch.mm(x, w.t(), out=out)", lambda: torch.mm(x,w.t(),out=out))
        bench("torch.add(x, x) (one trivial op)", lambda: torch.add(x,x))
torch.backends.cuda.preferred_blas_library("cublas")
print("=== preferred blas library -> cublas (rocBLAS on ROCm)")
for dt in (torch.bfloat16,):
  for rows in (2,):
    x=torch.randn(rows,2048,device=dev,dtype=dt); w=torch.randn(2048,2048,device=dev,dtype=dt); b=torch.randn(2048,device=dev,dtype=dt)
    with torch.inference_mode():
        bench("F.linear(x, w, b)", lambda: torch.nn.functional.linear(x,w,b))
        bench("torch.mm(x, w.t())", lambda: torch.mm(x,w.t()))
