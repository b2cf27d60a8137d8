// Status plumbing and the small host-only entry points of the C ABI (include/torch_bnb_fp4_hip.h).
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "fp4_common.h"

namespace fp4 {

namespace {
thread_local char g_last_error[512] = "";
}

void set_dequant_variant(int v);
void set_gemv_variant(int v);
void set_small_variant(int v);
void set_wide_variant(int v);
void set_quantize_variant(int v);

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}

// The reference only printf's launch failures (csrc/dequant_fp4_optimized.cu:48-53,
// csrc/gemv_fp4_optimized.cu:54-58); here they surface as a status the host layer raises on.
int check_launch(const char *what) {
    const hipError_t err = hipGetLastError();
    if (err == hipSuccess) return FP4_OK;
    set_error("%s: kernel launch failed: %s", what, hipGetErrorString(err));
    return FP4_ERR_LAUNCH;
}

// Compute units of the current device (256 on MI355X), cached per device ordinal; used to size persistent grids.
int device_cu_count() {
    static std::atomic<int> cached[64] = {};  // racing first calls both query and store the same value
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int cus = cached[dev].load(std::memory_order_relaxed);
    if (cus == 0) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        cached[dev].store(cus, std::memory_order_relaxed);
    }
    return cus;
}

}  // namespace fp4

extern "C" int fp4_hip_abi_version(void) { return FP4_HIP_ABI_VERSION; }

extern "C" const char *fp4_hip_last_error(void) { return fp4::g_last_error; }

extern "C" int fp4_hip_code_table(int table, float out16[16]) {
    if ((table != FP4_TABLE_CODEBOOK && table != FP4_TABLE_TREE) || !out16) {
        fp4::set_error("fp4_hip_code_table: bad argument");
        return FP4_ERR_INVALID_ARGUMENT;
    }
    const fp4::CodeTable t = fp4::make_table(table);
    std::memcpy(out16, t.bits, sizeof(t.bits));
    return FP4_OK;
}

extern "C" int fp4_hip_set_variant(const char *kernel, int variant) {
    if (kernel && !std::strcmp(kernel, "dequant")) {
        fp4::set_dequant_variant(variant);
        return FP4_OK;
    }
    if (kernel && !std::strcmp(kernel, "gemv")) {
        fp4::set_gemv_variant(variant);
        return FP4_OK;
    }
    if (kernel && !std::strcmp(kernel, "quantize")) {
        fp4::set_quantize_variant(variant);
        return FP4_OK;
    }
    if (kernel && !std::strcmp(kernel, "gemm_small")) {
        fp4::set_small_variant(variant);
        return FP4_OK;
    }
    if (kernel && !std::strcmp(kernel, "gemm_wide")) {
        fp4::set_wide_variant(variant);
        return FP4_OK;
    }
    fp4::set_error("fp4_hip_set_variant: unknown kernel '%s'", kernel ? kernel : "(null)");
    return FP4_ERR_INVALID_ARGUMENT;
}
