// Standalone GEMV experiment (no torch): every geometry of fp4_hip_gemv through the C ABI, next to
// "floor" kernels that only stream the same bytes, HBM-cold (R rotating weights) and cache-hot.
// Build on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Itorch-bnb-fp4_amd/csrc \
//         tools/exp_gemv.hip torch-bnb-fp4_amd/csrc/{capi,dequant_fp4,gemv_fp4,quantize_fp4}.hip -o /tmp/exp_gemv
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "torch_bnb_fp4_hip.h"
#ifdef FP4_EXP_STAMPS
extern "C" int fp4_exp_set_stamps(void *);  // csrc/gemv_fp4.hip, diagnostic build only
#endif

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// floor: every lane loads LOADS x 16 B of the packed stream (all issued up front), xors them, one dword store per wave
template <int LOADS, int THREADS, bool NT>
__global__ __launch_bounds__(THREADS) void stream_floor(const u32x4 *__restrict__ W, const float *__restrict__ absmax,
                                                        uint32_t *__restrict__ out, int nchunks) {
    const int64_t base = (int64_t(blockIdx.x) * THREADS + threadIdx.x);
    const int64_t stride = int64_t(gridDim.x) * THREADS;
    u32x4 v[LOADS];
    float a[LOADS];
#pragma unroll
    for (int j = 0; j < LOADS; ++j) {
        const int64_t c = base + j * stride;
        if (c < nchunks) {
            v[j] = NT ? __builtin_nontemporal_load(W + c) : W[c];
            a[j] = absmax[c >> 1];
        } else {
            v[j] = u32x4{0, 0, 0, 0};
            a[j] = 0.f;
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < LOADS; ++j) acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w ^ __builtin_bit_cast(uint32_t, a[j]);
    for (int m = 32; m >= 1; m >>= 1) acc ^= __shfl_xor(acc, m);
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * THREADS + threadIdx.x) >> 6] = acc;
}

// floor 2 (round 4): the bare read with the best geometry found for this chip (tools/stream_probe.hip, mode 1): 256-thread workgroups, a
// wave reads 4 KiB contiguous as four 1-KiB instructions, 16 B per lane, non-temporal; packed bytes AND scales as one byte stream
// (LOADS 16-byte loads per lane: 4 = the probe's tile of 16 KiB per workgroup; 2 / 1 = 8 / 4 KiB tiles, i.e. twice / four times the workgroups)
template <int LOADS>
__global__ __launch_bounds__(256) void stream_floor2(const u32x4 *__restrict__ W, const u32x4 *__restrict__ A, uint32_t *__restrict__ out,
                                                     int w_tiles, int a_tiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool second = (int)blockIdx.x >= w_tiles;
    if (second && (int)blockIdx.x - w_tiles >= a_tiles) return;
    const u32x4 *src = (second ? A + int64_t(blockIdx.x - w_tiles) * (256 * LOADS) : W + int64_t(blockIdx.x) * (256 * LOADS)) + wave * (64 * LOADS) + lane;
    u32x4 v[LOADS];
#pragma unroll
    for (int j = 0; j < LOADS; ++j) v[j] = __builtin_nontemporal_load(src + j * 64);
    u32x4 acc = v[0];
#pragma unroll
    for (int j = 1; j < LOADS; ++j) acc ^= v[j];
    if (acc.x == 0x12345u && acc.y == 0x6789u && acc.z == 0xabcdu) out[threadIdx.x] = acc.w;
}

__global__ void empty_kernel(uint32_t *out) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && out == nullptr) out[0] = 1;
}

static float bf16_to_f(uint16_t b) {
    uint32_t u = uint32_t(b) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static uint16_t f_to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

struct Timer {
    hipStream_t s;
    hipEvent_t e0, e1;
    template <typename F>
    void run(const char *name, int launches, double bytes, F &&record) {
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        record();
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        std::vector<float> t;
        for (int rep = 0; rep < 9; ++rep) {
            CK(hipEventRecord(e0, s));
            CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            t.push_back(ms * 1e3f / launches);
        }
        std::sort(t.begin(), t.end());
        printf("%-44s med %7.3f us  min %7.3f us  -> %7.1f GB/s\n", name, t[t.size() / 2], t[0], bytes / t[t.size() / 2] / 1e3);
        fflush(stdout);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
};

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 4096, K = argc > 2 ? atoi(argv[2]) : 4096;
    const bool quick = argc > 3 && !strcmp(argv[3], "quick");  // default geometry + one floor only (what the FP4_EXP_* builds are run with)
    const int64_t n = int64_t(M) * K;
    const int R = (int)std::max<int64_t>(8, std::min<int64_t>(64, (int64_t)(700e6 / (n * 0.5625))));
    std::vector<uint8_t> hp(n / 2);
    std::vector<float> ha(n / 64);
    std::vector<uint16_t> hx(K);
    srand(3);
    for (auto &b : hp) b = (uint8_t)rand();
    for (auto &f : ha) f = 0.01f + 0.1f * (rand() / (float)RAND_MAX);
    for (auto &v : hx) v = f_to_bf16((rand() / (float)RAND_MAX) * 2.f - 1.f);
    std::vector<uint8_t *> packed(R);
    std::vector<float *> absmax(R);
    std::vector<uint8_t> hdev = hp;  // what the device sees
#ifdef FP4_EXP_RELAID
    // load-time re-layout under test: per packed dword, byte 0 = (e0,e2), byte 1 = (e1,e3), byte 2 = (e4,e6), byte 3 = (e5,e7)
    for (int64_t d = 0; d + 3 < n / 2; d += 4) {
        uint8_t e[8];
        for (int b = 0; b < 4; ++b) e[2 * b] = hp[d + b] >> 4, e[2 * b + 1] = hp[d + b] & 15;
        hdev[d + 0] = uint8_t(e[0] << 4 | e[2]);
        hdev[d + 1] = uint8_t(e[1] << 4 | e[3]);
        hdev[d + 2] = uint8_t(e[4] << 4 | e[6]);
        hdev[d + 3] = uint8_t(e[5] << 4 | e[7]);
    }
#endif
    for (int i = 0; i < R; ++i) {
        CK(hipMalloc(&packed[i], n / 2));
        CK(hipMalloc(&absmax[i], n / 64 * 4));
        CK(hipMemcpy(packed[i], hdev.data(), n / 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(absmax[i], ha.data(), n / 64 * 4, hipMemcpyHostToDevice));
    }
    uint16_t *x, *y;
    uint32_t *scratch;
    CK(hipMalloc(&x, K * 2));
    CK(hipMalloc(&y, M * 2));
    CK(hipMalloc(&scratch, 1 << 20));
    CK(hipMemcpy(x, hx.data(), K * 2, hipMemcpyHostToDevice));
    Timer T;
    CK(hipStreamCreate(&T.s));
    CK(hipEventCreate(&T.e0));
    CK(hipEventCreate(&T.e1));
    const double bytes = n / 2.0 + n / 64.0 * 4 + (K + M) * 2.0;
    const int reps = 4;
    printf("M=%d K=%d R=%d  algorithmic bytes %.0f\n", M, K, R, bytes);

    T.run("empty kernel chain (launch boundary)", 256, 0.0, [&] {
        for (int i = 0; i < 256; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, T.s, scratch);
    });
    const int nchunks = (int)(n / 32);
#define FLOOR(L, TH, NT, COLD)                                                                                        \
    T.run("floor L" #L " T" #TH " nt=" #NT " cold=" #COLD, R * reps, bytes, [&] {                                     \
        for (int i = 0; i < R * reps; ++i) {                                                                          \
            const int w = COLD ? i % R : 0;                                                                           \
            const int blocks = (nchunks + L * TH - 1) / (L * TH);                                                     \
            hipLaunchKernelGGL((stream_floor<L, TH, NT>), dim3(blocks), dim3(TH), 0, T.s, (const u32x4 *)packed[w], \
                               absmax[w], scratch, nchunks);                                                          \
        }                                                                                                             \
    });
    {   // floor 2: whole 16-KiB tiles of the packed bytes and of the scales (both are multiples of 16 KiB at the shapes this is run on)
        const int w_tiles = (int)((n / 2) / 16384), a_tiles = (int)((n / 64 * 4) / 16384);
        const double b2 = double(w_tiles + a_tiles) * 16384.0;
        for (int cold = 1; cold >= 0; --cold)
            T.run(cold ? "floor2 probe geometry (bare read of weights + scales) cold" : "floor2 probe geometry (bare read of weights + scales) hot", R * reps, b2, [&] {
                for (int i = 0; i < R * reps; ++i) {
                    const int w = cold ? i % R : 0;
                    hipLaunchKernelGGL(stream_floor2<4>, dim3(w_tiles + a_tiles), dim3(256), 0, T.s, (const u32x4 *)packed[w], (const u32x4 *)absmax[w],
                                       scratch, w_tiles, a_tiles);
                }
            });
        T.run("floor2 with 2 loads per lane (twice the workgroups) cold", R * reps, b2, [&] {
            for (int i = 0; i < R * reps; ++i)
                hipLaunchKernelGGL(stream_floor2<2>, dim3(2 * (w_tiles + a_tiles)), dim3(256), 0, T.s, (const u32x4 *)packed[i % R], (const u32x4 *)absmax[i % R],
                                   scratch, 2 * w_tiles, 2 * a_tiles);
        });
        T.run("floor2 with 1 load per lane (four times the workgroups) cold", R * reps, b2, [&] {
            for (int i = 0; i < R * reps; ++i)
                hipLaunchKernelGGL(stream_floor2<1>, dim3(4 * (w_tiles + a_tiles)), dim3(256), 0, T.s, (const u32x4 *)packed[i % R], (const u32x4 *)absmax[i % R],
                                   scratch, 4 * w_tiles, 4 * a_tiles);
        });
    }
    if (!quick) {
        FLOOR(1, 256, false, true) FLOOR(2, 256, false, true) FLOOR(4, 256, false, true) FLOOR(8, 256, false, true)
        FLOOR(4, 256, true, true) FLOOR(2, 512, true, true) FLOOR(4, 512, true, true) FLOOR(2, 1024, true, true)
        FLOOR(4, 256, true, false)
    }
    FLOOR(2, 256, true, true) FLOOR(2, 256, true, false)

    // CPU reference for a handful of rows
    auto ref_row = [&](int r) {
        static const float mag[8] = {0.f, 0.0052083330f, 0.6666667f, 1.f, 0.333333f, 0.5f, 0.1666667f, 0.25f};
        double acc = 0;
        for (int k = 0; k < K; ++k) {
            const int64_t e = int64_t(r) * K + k;
            const uint8_t b = hp[e >> 1];
            const int nib = (e & 1) ? (b & 15) : (b >> 4);
            const float w = (nib & 8 ? -mag[nib & 7] : mag[nib & 7]) * ha[e / 64];
            acc += (double)w * bf16_to_f(hx[k]);
        }
        return acc;
    };
    const int check_rows[6] = {0, 1, 7, M / 2 + 3, M - 2, M - 1};
    double refs[6];
    for (int i = 0; i < 6; ++i) refs[i] = ref_row(check_rows[i]);

    std::vector<std::pair<const char *, int>> variants = {
        {"lds r1 w4 u2", 1 | (4 << 8) | (2 << 16)},   {"lds r1 w8 u2", 1 | (8 << 8) | (2 << 16)},
        {"regx it1", (1 << 24) | 1},                  {"regx it2", (1 << 24) | 2},
        {"regx it4", (1 << 24) | 4},                  {"regx 5 bands it2", (1 << 24) | (5 << 8) | 2},
        {"regx 6 bands it2", (1 << 24) | (6 << 8) | 2}, {"regx 7 bands", (1 << 24) | (7 << 8) | 2},
        {"regx 8 bands it2", (1 << 24) | (8 << 8) | 2},  // (band geometries apply to the row lengths they were built for; elsewhere = the standard split)
        {"default heuristic", -1},
    };
    if (quick) variants = {{"default heuristic", -1}};
#ifdef FP4_EXP_STAMPS
    {   // per-wave timeline of ONE HBM-cold launch of the default geometry (the last of a graph of R back-to-back launches)
        const size_t kMaxWaves = 1 << 17;
        unsigned long long *dstamps;
        CK(hipMalloc(&dstamps, kMaxWaves * 4 * 8));
        CK(hipMemset(dstamps, 0, kMaxWaves * 4 * 8));
        fp4_exp_set_stamps(dstamps);
        fp4_hip_set_variant("gemv", -1);
        for (int rep = 0; rep < 3; ++rep) {
            T.run("stamped gemv default cold", R, bytes, [&] {
                for (int i = 0; i < R; ++i) fp4_hip_gemv(x, packed[i % R], absmax[i % R], nullptr, y, M, K, 64, FP4_DTYPE_BF16, T.s);
            });
        }
        std::vector<unsigned long long> hs(kMaxWaves * 4);
        CK(hipMemcpy(hs.data(), dstamps, hs.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> t[4];
        unsigned long long base = ~0ull;
        size_t nw = 0;
        for (size_t w = 0; w < kMaxWaves; ++w)
            if (hs[4 * w]) base = std::min(base, hs[4 * w]), nw = w + 1;
        for (size_t w = 0; w < nw; ++w)
            if (hs[4 * w])
                for (int i = 0; i < 4; ++i) t[i].push_back((hs[4 * w + i] - base) * 0.01);  // 100 MHz ticks -> us
        const char *names[4] = {"entry", "loads issued", "last data consumed", "exit"};
        printf("per-wave stamps of one cold launch, us after the first wave's entry (%zu waves)\n", t[0].size());
        for (int i = 0; i < 4; ++i) {
            std::sort(t[i].begin(), t[i].end());
            auto q = [&](double f) { return t[i][std::min(t[i].size() - 1, size_t(f * t[i].size()))]; };
            printf("  %-20s min %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  p99 %6.2f  max %6.2f\n", names[i], q(0.0), q(0.10), q(0.50), q(0.90), q(0.99),
                   t[i].back());
        }
        // lifetime of a wave and its waiting share
        std::vector<double> life, wait;
        for (size_t w = 0; w < nw; ++w)
            if (hs[4 * w]) {
                life.push_back((hs[4 * w + 3] - hs[4 * w]) * 0.01);
                wait.push_back((hs[4 * w + 2] - hs[4 * w + 1]) * 0.01);
            }
        std::sort(life.begin(), life.end());
        std::sort(wait.begin(), wait.end());
        printf("  wave lifetime        p10 %6.2f  p50 %6.2f  p90 %6.2f   loads issued -> last data consumed  p10 %6.2f  p50 %6.2f  p90 %6.2f\n",
               life[life.size() / 10], life[life.size() / 2], life[life.size() * 9 / 10], wait[wait.size() / 10], wait[wait.size() / 2],
               wait[wait.size() * 9 / 10]);
        fp4_exp_set_stamps(nullptr);
        fflush(stdout);
    }
#endif
    for (auto &v : variants) {
        if (fp4_hip_set_variant("gemv", v.second)) {
            printf("set_variant failed\n");
            return 1;
        }
        CK(hipMemset(y, 0xFF, M * 2));
        int rc = fp4_hip_gemv(x, packed[0], absmax[0], nullptr, y, M, K, 64, FP4_DTYPE_BF16, T.s);
        CK(hipStreamSynchronize(T.s));
        if (rc) {
            printf("%s: rc=%d %s\n", v.first, rc, fp4_hip_last_error());
            continue;
        }
        std::vector<uint16_t> hy(M);
        CK(hipMemcpy(hy.data(), y, M * 2, hipMemcpyDeviceToHost));
        double maxrel = 0;
        for (int i = 0; i < 6; ++i) {
            const double got = bf16_to_f(hy[check_rows[i]]);
            maxrel = std::max(maxrel, std::fabs(got - refs[i]) / (std::fabs(refs[i]) + 1e-3));
        }
        char name[96];
        snprintf(name, sizeof name, "gemv %s cold (maxrel %.1e)", v.first, maxrel);
        T.run(name, R * reps, bytes, [&] {
            for (int i = 0; i < R * reps; ++i) fp4_hip_gemv(x, packed[i % R], absmax[i % R], nullptr, y, M, K, 64, FP4_DTYPE_BF16, T.s);
        });
        snprintf(name, sizeof name, "gemv %s hot", v.first);
        T.run(name, R * reps, bytes, [&] {
            for (int i = 0; i < R * reps; ++i) fp4_hip_gemv(x, packed[0], absmax[0], nullptr, y, M, K, 64, FP4_DTYPE_BF16, T.s);
        });
    }
    return 0;
}
