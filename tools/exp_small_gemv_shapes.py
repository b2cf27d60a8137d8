#!/usr/bin/env python3
"""Round 5: is there anything to gain on the GEMV at the SMALL shapes of BASELINE config 3 (2048x768, 2048x2048, 64x2048) and at k / v
(1024x4096)?  Default dispatch and the two LDS-x geometries, with bias, cache-hot, 64 identical launches per HIP-graph replay, next to
a trivial torch elementwise kernel (the launch boundary itself)."""
import os, sys, statistics
REPO="/root/repo"
sys.path[:0]=[REPO, REPO+"/torch-bnb-fp4_amd", REPO+"/tests"]
import torch, hipabi
dev=torch.device("cuda",0)
def capture(fn):
    fn(); torch.cuda.synchronize(); g=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): fn()
    torch.cuda.synchronize(); return g.replay
def timeit(rp,n,reps=9):
    ts=[]
    for _ in range(reps+3):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record(); rp(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b)*1e3/n)
    return statistics.median(ts[3:])
for dt in (torch.bfloat16, torch.float32):
  for M,K in ((2048,768),(2048,2048),(64,2048),(1024,4096),(4096,4096)):
    n=M*K
    P=torch.randint(0,256,(n//2,),dtype=torch.uint8,device=dev); A=torch.rand(n//64,device=dev)*0.1+0.01
    x=torch.randn(K,device=dev).to(dt); bias=torch.randn(M,device=dev).to(dt)
    # dependent chain like a model: output feeds nothing, but same stream -> serialized
    rp=capture(lambda:[hipabi.gemv(x,P,A,M,K,64,bias) for _ in range(64)])
    print(f"{str(dt):16s} {M}x{K}: {timeit(rp,64):.2f} us per launch (hot, graph of 64)", flush=True)
    if dt==torch.bfloat16:
        for v,name in (((1|(4<<8)|(2<<16)),"lds-x 1x4"),((1|(8<<8)|(2<<16)),"lds-x 1x8")):
            hipabi.set_variant("gemv", v)
            rp=capture(lambda:[hipabi.gemv(x,P,A,M,K,64,bias) for _ in range(64)])
            print(f"      variant {name}: {timeit(rp,64):.2f} us", flush=True)
        hipabi.set_variant("gemv",-1)
e=torch.empty(64,device=dev)
rp=capture(lambda:[e.add_(1.0) for _ in range(64)])
print(f"torch add_ on 64 floats: {timeit(rp,64):.2f} us per launch")
