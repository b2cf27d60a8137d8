// What THIS box streams with the dequant kernel's own access geometry and no arithmetic: the same-run ceiling bench.py prints
// next to the 8 TB/s spec figure (roofline.box_stream_gbps / frac_of_box_stream; SURVEY section 8d asks for a measured figure
// beside the vendor peak).  Not part of the product and not behind its C ABI: a measuring stick, built into
// tools/libfp4_stream_probe.so by torch-bnb-fp4_amd/build.py and loaded by bench.py OUTSIDE the timed region.
//
// Geometry = dequant_tiles_kernel<bf16, 4 loads, nt> (csrc/dequant_fp4.hip): 256-thread workgroups, one workgroup per 16 KiB of
// output, a wave owns 4 KiB contiguous of it and each of its four store instructions writes 1 KiB contiguous (16 B per lane,
// non-temporal); the packed side is a quarter of that (4 B per lane and load).  tools/exp_hbm.hip, round 1, measured plain streams
// whose per-thread accesses are megabytes apart (4.7-5.0 TB/s): that is not this kernel's pattern and is not a ceiling for it.
//
//   mode 0  write only : 16 B x 4 per lane                         bytes = n
//   mode 1  read only  : 16 B x 4 per lane (same spans, loads)     bytes = n
//   mode 2  copy       : 16 B x 4 in, 16 B x 4 out                 bytes = 2 n
//   mode 3  dequant mix: 4 B x 4 in, 16 B x 4 out (1 read : 4 write, the kernel's own ratio without the scales)   bytes = 1.25 n
//   mode 4  the same mix with the kernel's own phase structure: all loads, one LDS hand-off + workgroup barrier, then the stores
//
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC tools/stream_probe.hip -o tools/libfp4_stream_probe.so
#include <hip/hip_runtime.h>

#include <cstdint>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int kThreads = 256;
constexpr int kLoads = 4;
constexpr int kTileBytes = kThreads * kLoads * 16;  // 16 KiB of 16-byte accesses per workgroup

template <int MODE>
__global__ __launch_bounds__(kThreads) void stream_probe_kernel(const void *__restrict__ in, void *__restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t word = int64_t(blockIdx.x) * (kThreads * kLoads) + wave * (64 * kLoads) + lane;  // 16-byte word of lane's 1st access
    u32x4 acc = {uint32_t(threadIdx.x), 1u, 2u, 3u};
    if constexpr (MODE == 1 || MODE == 2) {
        const u32x4 *src = reinterpret_cast<const u32x4 *>(in) + word;
        u32x4 v[kLoads];
#pragma unroll
        for (int j = 0; j < kLoads; ++j) v[j] = __builtin_nontemporal_load(src + j * 64);
        if constexpr (MODE == 2) {
            u32x4 *dst = reinterpret_cast<u32x4 *>(out) + word;
#pragma unroll
            for (int j = 0; j < kLoads; ++j) __builtin_nontemporal_store(v[j], dst + j * 64);
        } else {
#pragma unroll
            for (int j = 0; j < kLoads; ++j) acc ^= v[j];
            if (acc.x == 0x12345u && acc.y == 0x6789u && acc.z == 0xabcdu) reinterpret_cast<u32x4 *>(out)[word] = acc;  // keeps the loads alive
        }
    } else {
        uint32_t q[kLoads] = {0u, 0u, 0u, 0u};
        if constexpr (MODE == 4) {
            __shared__ uint32_t s_stage[kThreads];
            const uint32_t *src = reinterpret_cast<const uint32_t *>(in) + word;
#pragma unroll
            for (int j = 0; j < kLoads; ++j) q[j] = __builtin_nontemporal_load(src + j * 64);
            s_stage[threadIdx.x] = q[0];
            __syncthreads();
            acc.y = s_stage[threadIdx.x ^ 1];  // (consumed: word 1 of every store)
        }
        if constexpr (MODE == 3) {
            const uint32_t *src = reinterpret_cast<const uint32_t *>(in) + word;
#pragma unroll
            for (int j = 0; j < kLoads; ++j) q[j] = __builtin_nontemporal_load(src + j * 64);
        }
        u32x4 *dst = reinterpret_cast<u32x4 *>(out) + word;
#pragma unroll
        for (int j = 0; j < kLoads; ++j) {
            u32x4 o = acc;
            o.x += q[j] + j;
            __builtin_nontemporal_store(o, dst + j * 64);
        }
    }
}
}  // namespace

// n = bytes written (modes 0, 2, 3) or read (mode 1) by one launch; must be a multiple of 16 KiB.  `in` needs n bytes (mode 3: n / 4).
// Returns 0, or -1 for a bad argument, or the hipError_t of the launch.
extern "C" int fp4_probe_stream(int mode, const void *in, void *out, int64_t n, void *stream) {
    if (mode < 0 || mode > 4 || n <= 0 || (n % kTileBytes) || !out || (mode != 0 && !in)) return -1;
    const dim3 grid((unsigned)(n / kTileBytes)), block(kThreads);
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (mode) {
        case 0: hipLaunchKernelGGL(stream_probe_kernel<0>, grid, block, 0, s, in, out); break;
        case 1: hipLaunchKernelGGL(stream_probe_kernel<1>, grid, block, 0, s, in, out); break;
        case 2: hipLaunchKernelGGL(stream_probe_kernel<2>, grid, block, 0, s, in, out); break;
        case 4: hipLaunchKernelGGL(stream_probe_kernel<4>, grid, block, 0, s, in, out); break;
        default: hipLaunchKernelGGL(stream_probe_kernel<3>, grid, block, 0, s, in, out); break;
    }
    return (int)hipGetLastError();
}

extern "C" int64_t fp4_probe_bytes(int mode, int64_t n) { return mode == 2 ? 2 * n : (mode >= 3 ? n + n / 4 : n); }
