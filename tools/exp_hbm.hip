// Practical HBM ceilings of this box for the dequant's traffic mix: write-only, read-only, copy and a 22 % read / 78 % write
// stream with the same launch structure as the bench (R launches of 43 MB each from a HIP graph, buffers rotating over > 2 GB).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp_hbm.hip -o /tmp/exp_hbm
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// each workgroup: 256 threads; per thread RD 16-byte nt loads and WR 16-byte nt stores, all loads first
template <int RD, int WR>
__global__ __launch_bounds__(256) void stream(const u32x4 *__restrict__ in, u32x4 *__restrict__ out) {
    const int64_t t = int64_t(blockIdx.x) * 256 + threadIdx.x;
    const int64_t nthreads = int64_t(gridDim.x) * 256;
    u32x4 v[RD > 0 ? RD : 1];
    u32x4 acc = {threadIdx.x, 1u, 2u, 3u};
#pragma unroll
    for (int i = 0; i < RD; ++i) v[i] = __builtin_nontemporal_load(in + t + i * nthreads);
#pragma unroll
    for (int i = 0; i < RD; ++i) acc ^= v[i];
#pragma unroll
    for (int i = 0; i < WR; ++i) {
        u32x4 o = acc;
        o.x += i;
        __builtin_nontemporal_store(o, out + t + i * nthreads);
    }
    if (WR == 0 && acc.x == 0x12345u) out[t] = acc;
}

int main() {
    const int R = 48;
    const size_t buf = 64u << 20;  // 64 MiB per buffer
    std::vector<void *> a(R), b(R);
    for (int i = 0; i < R; ++i) {
        CK(hipMalloc(&a[i], buf));
        CK(hipMalloc(&b[i], buf));
        CK(hipMemset(a[i], i + 1, buf));
    }
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto run = [&](const char *name, double bytes_per_launch, auto launch) {
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < R; ++i) launch(i);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int w = 0; w < 2; ++w) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        std::vector<float> t;
        for (int rep = 0; rep < 7; ++rep) {
            CK(hipEventRecord(e0, s));
            CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            t.push_back(ms * 1e3f / R);
        }
        std::sort(t.begin(), t.end());
        printf("%-58s %8.2f us/launch -> %7.1f GB/s\n", name, t[t.size() / 2], bytes_per_launch / t[t.size() / 2] / 1e3);
        fflush(stdout);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    };
    // 2048 workgroups of 256 threads, like the dequant kernel
    const int blocks = 2048;
    const double unit = double(blocks) * 256 * 16;  // bytes per "one access per thread" = 8 MiB
    run("write only, 4 x 16 B per thread (32 MiB per launch)", 4 * unit, [&](int i) { hipLaunchKernelGGL((stream<0, 4>), dim3(blocks), dim3(256), 0, s, (const u32x4 *)a[i], (u32x4 *)b[i]); });
    run("write only, 8 x 16 B per thread (64 MiB per launch)", 8 * unit, [&](int i) { hipLaunchKernelGGL((stream<0, 8>), dim3(blocks), dim3(256), 0, s, (const u32x4 *)a[i], (u32x4 *)b[i]); });
    run("read only, 4 x 16 B per thread (32 MiB per launch)", 4 * unit, [&](int i) { hipLaunchKernelGGL((stream<4, 0>), dim3(blocks), dim3(256), 0, s, (const u32x4 *)a[i], (u32x4 *)b[i]); });
    run("read only, 8 x 16 B per thread (64 MiB per launch)", 8 * unit, [&](int i) { hipLaunchKernelGGL((stream<8, 0>), dim3(blocks), dim3(256), 0, s, (const u32x4 *)a[i], (u32x4 *)b[i]); });
    run("copy, 4 + 4 x 16 B per thread (64 MiB per launch)", 8 * unit, [&](int i) { hipLaunchKernelGGL((stream<4, 4>), dim3(blocks), dim3(256), 0, s, (const u32x4 *)a[i], (u32x4 *)b[i]); });
    run("dequant mix, 1 read + 4 writes per thread (40 MiB per launch)", 5 * unit, [&](int i) { hipLaunchKernelGGL((stream<1, 4>), dim3(blocks), dim3(256), 0, s, (const u32x4 *)a[i], (u32x4 *)b[i]); });
    run("dequant mix x2, 2 reads + 8 writes per thread (80 MiB)", 10 * unit, [&](int i) { hipLaunchKernelGGL((stream<2, 8>), dim3(blocks), dim3(256), 0, s, (const u32x4 *)a[i], (u32x4 *)b[i]); });
    return 0;
}
