// Standalone geometry / cache-policy experiment for the bf16 dequant kernel (no torch).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/exp_dequant.hip -o gpurun_out/exp_dequant
// Times R distinct 4096x4096 weights back to back from a HIP graph (HBM-cold) for every variant.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                           \
    do {                                                                                \
        hipError_t e_ = (x);                                                            \
        if (e_ != hipSuccess) {                                                         \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                    \
        }                                                                               \
    } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ uint32_t nibble_of(uint32_t q, int i) { return (q >> (8 * (i >> 1) + ((i & 1) ? 0 : 4))) & 15u; }
__device__ __forceinline__ float lut_entry(int idx) {
    const int m = idx & 7;
    uint32_t b = 0u;
    b = m == 1 ? 0x3BAAAAAAu : b;
    b = m == 2 ? 0x3F2AAAABu : b;
    b = m == 3 ? 0x3F800000u : b;
    b = m == 4 ? 0x3EAAAA9Fu : b;
    b = m == 5 ? 0x3F000000u : b;
    b = m == 6 ? 0x3E2AAAADu : b;
    b = m == 7 ? 0x3E800000u : b;
    return __builtin_bit_cast(float, b | (uint32_t(idx & 8) << 28));
}

// STORE: 0 plain global store, 1 nontemporal builtin, 2.. buffer store with aux = STORE - 2 + ... see table in main
template <int LOADS, int THREADS, int STORE_AUX, bool USE_BUFFER, bool LOAD_NT, int ABSMODE, int REMAP = 0>
__global__ __launch_bounds__(THREADS) void dq(const uint8_t *__restrict__ packed, const float *__restrict__ absmax,
                                              void *__restrict__ out, int bs_shift, uint32_t out_bytes) {
    constexpr int kVals = 8;
    constexpr int kTile = THREADS * LOADS * kVals;
    constexpr int kMaxAbs = kTile / 32;
    __shared__ float s_lut[ABSMODE == 2 ? 16 * (THREADS / 64) : 16];
    __shared__ float s_abs[kMaxAbs];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // REMAP 1: each XCD (blocks b, b+8, ...) owns a contiguous eighth of the tensor; REMAP 2: pairs of XCD-neighbours
    unsigned bid = blockIdx.x;
    if constexpr (REMAP == 1) bid = (blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8;
    if constexpr (REMAP == 2) bid = (blockIdx.x % 8) * 2 + (blockIdx.x / 8 % 2) + (blockIdx.x / 16) * 16;
    const int64_t e_base = int64_t(bid) * kTile;
    const int n_abs = kTile >> bs_shift;
    const float *abs_src = absmax + (e_base >> bs_shift);
    constexpr int NA = (kMaxAbs + THREADS - 1) / THREADS;
    float am_reg[NA];
    if constexpr (ABSMODE == 0) {
#pragma unroll
        for (int r = 0; r < NA; ++r) {
            const int i = tid + r * THREADS;
            am_reg[r] = i < n_abs ? abs_src[i] : 0.f;
        }
    }
    const uint32_t *src = reinterpret_cast<const uint32_t *>(packed) + (e_base / kVals) + wave * (64 * LOADS) + lane;
    uint32_t q[LOADS];
#pragma unroll
    for (int j = 0; j < LOADS; ++j) q[j] = LOAD_NT ? __builtin_nontemporal_load(src + j * 64) : src[j * 64];
    float am_direct[LOADS];
    if constexpr (ABSMODE == 1) {  // no LDS staging: each lane loads its own scale per load-word
#pragma unroll
        for (int j = 0; j < LOADS; ++j) {
            const int word = wave * (64 * LOADS) + j * 64 + lane;
            am_direct[j] = abs_src[(word * kVals) >> bs_shift];
        }
    }
    if constexpr (ABSMODE == 0) {
#pragma unroll
        for (int r = 0; r < NA; ++r) {
            const int i = tid + r * THREADS;
            if (i < n_abs) s_abs[i] = am_reg[r];
        }
    }
    if constexpr (ABSMODE == 2) {  // each wave stages the scales of its own span; wave-level sync only (bs = 64)
        constexpr int per_wave = 64 * LOADS * kVals / 64;  // scales per wave
        static_assert(per_wave <= 64, "one scale per lane at most");
        float a = 0.f;
        if (lane < per_wave) a = abs_src[wave * per_wave + lane];
        if (lane < 16) s_lut[wave * 16 + lane] = lut_entry(lane);
        if (lane < per_wave) s_abs[wave * per_wave + lane] = a;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        if (tid < 16) s_lut[tid] = lut_entry(tid);
        __syncthreads();
    }
    __amdgpu_buffer_rsrc_t rsrc;
    if constexpr (USE_BUFFER) rsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, out_bytes, 0x00020000);
#pragma unroll
    for (int j = 0; j < LOADS; ++j) {
        const int word = wave * (64 * LOADS) + j * 64 + lane;
        const int e_local = word * kVals;
        const float am = ABSMODE == 1 ? am_direct[j] : s_abs[e_local >> bs_shift];
        const float *lut = ABSMODE == 2 ? s_lut + wave * 16 : s_lut;
        float v[kVals];
#pragma unroll
        for (int i = 0; i < kVals; ++i) v[i] = lut[nibble_of(q[j], i)] * am;
        u32x4 o = {pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
        if constexpr (USE_BUFFER) {
            __builtin_amdgcn_raw_buffer_store_b128(o, rsrc, (int)((e_base + e_local) * 2), 0, STORE_AUX);
        } else if constexpr (STORE_AUX == 2) {
            __builtin_nontemporal_store(o, reinterpret_cast<u32x4 *>(out) + (e_base + e_local) / 8);
        } else {
            reinterpret_cast<u32x4 *>(out)[(e_base + e_local) / 8] = o;
        }
    }
}


// Persistent variant (round 2, one bounded attempt): gridDim.x = k workgroups per CU, each walks tiles with stride
// gridDim.x; the NEXT tile's packed words and scales are requested before the current tile is decoded and stored, so a
// workgroup always has loads in flight behind its stores.  Scales are staged per wave (wave barrier only) into a
// double-buffered LDS slice; the LUT is staged once.
template <int LOADS, int THREADS>
__global__ __launch_bounds__(THREADS) void dqp(const uint8_t *__restrict__ packed, const float *__restrict__ absmax,
                                               void *__restrict__ out, int ntiles) {
    constexpr int kVals = 8, bs_shift = 6;
    constexpr int kTile = THREADS * LOADS * kVals;
    constexpr int per_wave = 64 * LOADS * kVals / 64;  // scales per wave and tile
    static_assert(per_wave <= 64, "one scale per lane at most");
    __shared__ float s_lut[16];
    __shared__ float s_abs[2][THREADS / 64][per_wave];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 16) s_lut[tid] = lut_entry(tid);
    __syncthreads();
    auto issue = [&](int tile, uint32_t (&q)[LOADS], float &a) {
        const int64_t e_base = int64_t(tile) * kTile;
        const uint32_t *src = reinterpret_cast<const uint32_t *>(packed) + (e_base / kVals) + wave * (64 * LOADS) + lane;
#pragma unroll
        for (int j = 0; j < LOADS; ++j) q[j] = __builtin_nontemporal_load(src + j * 64);
        a = absmax[(e_base >> bs_shift) + wave * per_wave + (lane < per_wave ? lane : per_wave - 1)];
    };
    uint32_t qa[LOADS], qb[LOADS];
    float aa, ab;
    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    issue(tile, qa, aa);
    int par = 0;
    auto body = [&](int t, const uint32_t (&q)[LOADS], float a) {
        if (lane < per_wave) s_abs[par][wave][lane] = a;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int64_t e_base = int64_t(t) * kTile;
#pragma unroll
        for (int j = 0; j < LOADS; ++j) {
            const int wl = j * 64 + lane;  // word inside this wave's span
            const float am = s_abs[par][wave][(wl * kVals) >> bs_shift];
            float v[kVals];
#pragma unroll
            for (int i = 0; i < kVals; ++i) v[i] = s_lut[nibble_of(q[j], i)] * am;
            u32x4 o = {pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
            __builtin_nontemporal_store(o, reinterpret_cast<u32x4 *>(out) + (e_base + (wave * (64 * LOADS) + wl) * kVals) / 8);
        }
        par ^= 1;
    };
    while (true) {
        int next = tile + gridDim.x;
        if (next < ntiles) issue(next, qb, ab);
        body(tile, qa, aa);
        if (next >= ntiles) break;
        tile = next;
        next = tile + gridDim.x;
        if (next < ntiles) issue(next, qa, aa);
        body(tile, qb, ab);
        if (next >= ntiles) break;
        tile = next;
    }
}

template <int LOADS, int THREADS, int WG_PER_CU>
void launch_p(const uint8_t *p, const float *a, void *o, int64_t n, hipStream_t s) {
    constexpr int tile = THREADS * LOADS * 8;
    const int ntiles = (int)(n / tile);
    const int grid = std::min(ntiles, WG_PER_CU * 256);
    hipLaunchKernelGGL((dqp<LOADS, THREADS>), dim3(grid), dim3(THREADS), 0, s, p, a, o, ntiles);
}

struct Variant {
    const char *name;
    void (*launch)(const uint8_t *, const float *, void *, int64_t, hipStream_t);
};

template <int LOADS, int THREADS, int AUX, bool BUF, bool LNT, int ABSMODE, int REMAP = 0>
void launch(const uint8_t *p, const float *a, void *o, int64_t n, hipStream_t s) {
    constexpr int tile = THREADS * LOADS * 8;
    hipLaunchKernelGGL((dq<LOADS, THREADS, AUX, BUF, LNT, ABSMODE, REMAP>), dim3((unsigned)(n / tile)), dim3(THREADS), 0, s, p, a, o, 6,
                       (uint32_t)(n * 2));
}

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 4096, K = argc > 2 ? atoi(argv[2]) : 4096;
    const int64_t n = int64_t(M) * K;
    const int R = 64;
    std::vector<uint8_t *> packed(R);
    std::vector<float *> absmax(R);
    std::vector<void *> outs(R);
    std::vector<uint8_t> hp(n / 2);
    std::vector<float> ha(n / 64);
    srand(1);
    for (auto &b : hp) b = (uint8_t)rand();
    for (auto &f : ha) f = 0.01f + 0.1f * (rand() / (float)RAND_MAX);
    for (int i = 0; i < R; ++i) {
        CK(hipMalloc(&packed[i], n / 2));
        CK(hipMalloc(&absmax[i], n / 64 * 4));
        CK(hipMalloc(&outs[i], n * 2));
        CK(hipMemcpy(packed[i], hp.data(), n / 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(absmax[i], ha.data(), n / 64 * 4, hipMemcpyHostToDevice));
    }
    hipStream_t s;
    CK(hipStreamCreate(&s));
    // aux bits (gfx940+): 1 = sc0, 2 = nt, 16 = sc1
    std::vector<Variant> vs = {
        {"PERSIST L4 T256 k=8 (one tile)", launch_p<4, 256, 8>},
        {"PERSIST L4 T256 k=4       ", launch_p<4, 256, 4>},
        {"PERSIST L4 T256 k=2       ", launch_p<4, 256, 2>},
        {"PERSIST L2 T256 k=8       ", launch_p<2, 256, 8>},
        {"PERSIST L2 T256 k=4       ", launch_p<2, 256, 4>},
        {"PERSIST L2 T256 k=2       ", launch_p<2, 256, 2>},
        {"PERSIST L1 T256 k=8       ", launch_p<1, 256, 8>},
        {"PERSIST L1 T256 k=4       ", launch_p<1, 256, 4>},
        {"PERSIST L4 T512 k=4       ", launch_p<4, 512, 4>},
        {"PERSIST L4 T512 k=2       ", launch_p<4, 512, 2>},
        {"PERSIST L2 T512 k=4       ", launch_p<2, 512, 4>},
        {"PERSIST L2 T512 k=2       ", launch_p<2, 512, 2>},
        {"PERSIST L2 T1024 k=2      ", launch_p<2, 1024, 2>},
        {"PERSIST L2 T1024 k=1      ", launch_p<2, 1024, 1>},
        {"PERSIST L1 T1024 k=2      ", launch_p<1, 1024, 2>},
        {"L4 T256 plain            ", launch<4, 256, 0, false, false, 0>},
        {"L4 T256 nt(builtin)      ", launch<4, 256, 2, false, false, 0>},
        {"L4 T256 buf aux0         ", launch<4, 256, 0, true, false, 0>},
        {"L4 T256 buf nt           ", launch<4, 256, 2, true, false, 0>},
        {"L4 T256 buf sc1          ", launch<4, 256, 16, true, false, 0>},
        {"L4 T256 buf sc0          ", launch<4, 256, 1, true, false, 0>},
        {"L4 T256 buf sc0 sc1      ", launch<4, 256, 17, true, false, 0>},
        {"L4 T256 buf nt sc1       ", launch<4, 256, 18, true, false, 0>},
        {"L4 T256 buf nt sc0       ", launch<4, 256, 3, true, false, 0>},
        {"L4 T256 buf nt sc0 sc1   ", launch<4, 256, 19, true, false, 0>},
        {"L4 T256 nt + nt loads    ", launch<4, 256, 2, false, true, 0>},
        {"L4 T256 nt ntld remap-xcd", launch<4, 256, 2, false, true, 0, 1>},
        {"L4 T256 nt ntld remap-pair", launch<4, 256, 2, false, true, 0, 2>},
        {"L2 T256 nt ntld remap-xcd", launch<2, 256, 2, false, true, 0, 1>},
        {"L8 T256 nt ntld remap-xcd", launch<8, 256, 2, false, true, 0, 1>},
        {"L4 T256 nt ntld wave-sync", launch<4, 256, 2, false, true, 2>},
        {"L8 T256 nt ntld wave-sync", launch<8, 256, 2, false, true, 2>},
        {"L4 T128 nt ntld wave-sync", launch<4, 128, 2, false, true, 2>},
        {"L4 T64 nt ntld wave-sync ", launch<4, 64, 2, false, true, 2>},
        {"L4 T512 nt ntld wave-sync", launch<4, 512, 2, false, true, 2>},
        {"L4 T256 nt absmax direct ", launch<4, 256, 2, false, false, 1>},
        {"L4 T256 nt ntld absdirect", launch<4, 256, 2, false, true, 1>},
        {"L2 T256 nt               ", launch<2, 256, 2, false, false, 0>},
        {"L8 T256 nt               ", launch<8, 256, 2, false, false, 0>},
        {"L2 T512 nt               ", launch<2, 512, 2, false, false, 0>},
        {"L4 T512 nt               ", launch<4, 512, 2, false, false, 0>},
        {"L8 T512 nt               ", launch<8, 512, 2, false, false, 0>},
        {"L2 T1024 nt              ", launch<2, 1024, 2, false, false, 0>},
        {"L4 T1024 nt              ", launch<4, 1024, 2, false, false, 0>},
        {"L1 T1024 nt              ", launch<1, 1024, 2, false, false, 0>},
        {"L2 T128 nt               ", launch<2, 128, 2, false, false, 0>},
        {"L4 T128 nt               ", launch<4, 128, 2, false, false, 0>},
        {"L8 T128 nt               ", launch<8, 128, 2, false, false, 0>},
        {"L4 T64 nt                ", launch<4, 64, 2, false, false, 0>},
        {"L8 T64 nt                ", launch<8, 64, 2, false, false, 0>},
        {"L16 T64 nt               ", launch<16, 64, 2, false, false, 0>},
    };
    const double bytes = n / 2.0 + n / 64.0 * 4 + n * 2.0;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (auto &v : vs) {
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < R; ++i) v.launch(packed[i], absmax[i], outs[i], n, s);
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
            t.push_back(ms * 1e3f / R);
        }
        std::sort(t.begin(), t.end());
        printf("%s  med %7.3f us  min %7.3f us  -> %7.1f GB/s\n", v.name, t[t.size() / 2], t[0], bytes / t[t.size() / 2] / 1e3);
        fflush(stdout);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    // persistent vs one-shot outputs, bit for bit (matrix 0): run one of each into two buffers and compare on the host
    {
        std::vector<uint16_t> ra(n), rb(n);
        CK(hipMemset(outs[0], 0, n * 2));
        CK(hipMemset(outs[1], 0, n * 2));
        launch<4, 256, 2, false, true, 0>(packed[0], absmax[0], outs[0], n, s);
        launch_p<2, 256, 2>(packed[0], absmax[0], outs[1], n, s);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(ra.data(), outs[0], n * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(rb.data(), outs[1], n * 2, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (int64_t i = 0; i < n; ++i) bad += ra[i] != rb[i];
        printf("persistent L2 T256 k=2 vs one-shot L4 T256: %zu mismatching elements of %lld\n", bad, (long long)n);
        launch_p<4, 512, 2>(packed[0], absmax[0], outs[1], n, s);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(rb.data(), outs[1], n * 2, hipMemcpyDeviceToHost));
        bad = 0;
        for (int64_t i = 0; i < n; ++i) bad += ra[i] != rb[i];
        printf("persistent L4 T512 k=2 vs one-shot L4 T256: %zu mismatching elements of %lld\n", bad, (long long)n);
    }
    // sanity: last variant's output of matrix 0, element checks on the host
    std::vector<uint16_t> ho(1024);
    CK(hipMemcpy(ho.data(), outs[0], 2048, hipMemcpyDeviceToHost));
    printf("out[0..3] bits %04x %04x %04x %04x  (packed byte0 %02x, absmax0 %g)\n", ho[0], ho[1], ho[2], ho[3], hp[0], ha[0]);
    return 0;
}
