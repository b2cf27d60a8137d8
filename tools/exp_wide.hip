// The one-pass 17..64-row kernels (csrc/gemm_wide_fp4.hip, included as is) without torch: us per launch for every workgroup shape.
// (The first version of that file also had a register-staged kernel with compile-time ablation switches - no x DMA, no weight loads;
// its numbers are in profiles/r02_wide_batch_17_to_128_rows.txt.)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Itorch-bnb-fp4_amd/csrc tools/exp_wide.hip -o build_tmp/exp/exp_wide
// Run: exp_wide M K B   -> HBM-cold rotation over R weights in a HIP graph
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../torch-bnb-fp4_amd/csrc/gemm_wide_fp4.hip"

namespace fp4 {
void set_error(const char *, ...) {}
int check_launch(const char *) { return hipGetLastError() == hipSuccess ? FP4_OK : FP4_ERR_LAUNCH; }
int device_cu_count() { return 256; }
}  // namespace fp4

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 4096, K = argc > 2 ? atoi(argv[2]) : 4096, B = argc > 3 ? atoi(argv[3]) : 64;
    const int64_t n = int64_t(M) * K;
    int R = int(1.0e9 / (n * 0.5625));
    R = R < 6 ? 6 : (R > 48 ? 48 : R);
    std::vector<uint8_t *> packed(R);
    std::vector<float *> absmax(R);
    std::vector<uint8_t> hp(n / 2);
    std::vector<float> ha(n / 64);
    srand(1);
    for (auto &b : hp) b = (uint8_t)rand();
    for (auto &f : ha) f = 0.01f + 0.1f * (rand() / (float)RAND_MAX);
    for (int i = 0; i < R; ++i) {
        CK(hipMalloc(&packed[i], n / 2));
        CK(hipMalloc(&absmax[i], n / 64 * 4));
        CK(hipMemcpy(packed[i], hp.data(), n / 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(absmax[i], ha.data(), n / 64 * 4, hipMemcpyHostToDevice));
    }
    uint16_t *x, *out;
    CK(hipMalloc(&x, size_t(B) * K * 2));
    CK(hipMalloc(&out, size_t(B) * M * 2));
    std::vector<uint16_t> hx(size_t(B) * K);
    for (auto &v : hx) v = 0x3C00 + (rand() & 0x3FF);
    CK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    printf("%dx%d, %d rows, us per launch by rows per workgroup:", M, K, B);
    for (int cfg : {1, 2, 3, 4, 5}) {
        fp4::set_wide_variant(cfg);
        auto pass = [&]() {
            for (int i = 0; i < R; ++i)
                if (fp4::gemm_wide_launch(FP4_DTYPE_BF16, x, packed[i], absmax[i], nullptr, nullptr, out, B, M, K, 0, true, s) != FP4_OK) exit(2);
        };
        pass();
        CK(hipStreamSynchronize(s));
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        pass();
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        float best = 1e9f;
        for (int rep = 0; rep < 7; ++rep) {
            CK(hipEventRecord(e0, s));
            CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        if (cfg == 5) printf("  auto16: %6.2f", best * 1e3f / R); else printf("  %d rows: %6.2f", 8 << cfg, best * 1e3f / R);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    printf("\n");
    return 0;
}
