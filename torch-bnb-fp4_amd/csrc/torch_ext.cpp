// torch_bnb_fp4_ext -- the PyTorch-ROCm host layer above the C ABI (include/torch_bnb_fp4_hip.h).
//
// Re-exports the operator surface of the reference's pybind module (reference
// csrc/torch_fp4.cpp:125-139): ScalarType + dequantize_fp4, dequantize_fp4_codebook, gemv_fp4,
// qlinear, qlinear_bias, qlinear_codebook, qlinear_codebook_bias, with the same positional
// signatures, argument meaning and error behaviour (RuntimeError on a non-GPU / non-contiguous
// tensor, TypeError on a bad enum).  PyTorch is plumbing only: it owns device memory and the current
// stream; the batch>1 GEMM is a plain library GEMM on hipBLASLt, called directly with cached plans
// (lt_linear below; at::linear is its fallback); every FP4 kernel is behind the C ABI.
// Differences from the reference, all deliberate:
//   * launches go to the CURRENT torch stream under a device guard (the reference uses the legacy
//     default stream and no guard, csrc/dequant_fp4_optimized.cu:176, csrc/gemv_fp4_optimized.cu:266);
//   * launch / dtype failures raise instead of printf (csrc/dequant_fp4_optimized.cu:48-53,201-203);
//   * qlinear_codebook* dequantise all M*N elements (the reference passes the BYTE count,
//     csrc/torch_fp4.cpp:90,101, leaving half of the weight uninitialised).
// Extra exports (not in the reference): gemv_fp4_bias, gemv_fp4_fused, comm_* / allreduce_oneshot, gemm_small_fp4, gemv_fp4_partial, quantize_fp4, set_kernel_variant, set_qlinear_gemm, code_table.
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <hip/hip_runtime_api.h>
#include <hipblaslt/hipblaslt.h>
#include <torch/extension.h>

#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <stdexcept>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "torch_bnb_fp4_hip.h"

namespace {

enum class ScalarTypeEnum { float16 = FP4_DTYPE_F16, float32 = FP4_DTYPE_F32, bfloat16 = FP4_DTYPE_BF16 };

torch::ScalarType to_torch(ScalarTypeEnum t) {
    switch (t) {
        case ScalarTypeEnum::float16:
            return torch::kFloat16;
        case ScalarTypeEnum::float32:
            return torch::kFloat32;
        case ScalarTypeEnum::bfloat16:
            return torch::kBFloat16;
    }
    throw py::type_error("Unsupported scalar type");
}

int to_fp4_dtype(torch::ScalarType t, const char *what) {
    switch (t) {
        case torch::kFloat16:
            return FP4_DTYPE_F16;
        case torch::kFloat32:
            return FP4_DTYPE_F32;
        case torch::kBFloat16:
            return FP4_DTYPE_BF16;
        default:
            TORCH_CHECK(false, what, ": unsupported dtype ", t, " (need float16, bfloat16 or float32)");
    }
}

void check_gpu_contiguous(const torch::Tensor &t, const char *name) {
    // reference: CHECK_CUDA / CHECK_CONTIGUOUS (csrc/torch_fp4.cpp:19-20)
    TORCH_CHECK(t.is_cuda(), name, " must be a CUDA tensor");
    TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
}

void check_status(int rc) {
    if (rc == FP4_OK) return;
    const std::string msg = fp4_hip_last_error();
    if (rc == FP4_ERR_UNSUPPORTED && msg.find("dtype") != std::string::npos) throw std::runtime_error("Unsupported datatype: " + msg);
    TORCH_CHECK(false, msg);
}

void *current_stream(const torch::Tensor &t) { return c10::hip::getCurrentHIPStream(t.device().index()).stream(); }

// dequant of the first n elements of `out` (out is [M,N], n <= M*N)
void dequant_into(const torch::Tensor &A, const torch::Tensor &absmax, torch::Tensor &out, int64_t blocksize, int64_t n,
                  int table, int flags = FP4_DEQUANT_AUTO) {
    // reference: TORCH_CHECKs of csrc/dequant_fp4_optimized.cu:183-187,210-213
    TORCH_CHECK(A.dtype() == torch::kUInt8, "A must be uint8");
    TORCH_CHECK(absmax.dtype() == torch::kFloat32, "absmax must be float32");
    TORCH_CHECK(A.is_cuda(), "A must be cuda");
    TORCH_CHECK(absmax.is_cuda(), "absmax must be cuda");
    TORCH_CHECK(out.is_cuda(), "out must be cuda");
    TORCH_CHECK(absmax.device() == A.device() && out.device() == A.device(), "A, absmax and out must be on one device");
    TORCH_CHECK(n >= 0 && n <= out.numel(), "n = ", n, " does not fit the [M, N] output (", out.numel(), " elements)");
    TORCH_CHECK(blocksize >= 2 && blocksize % 2 == 0, "blocksize must be even and >= 2");
    TORCH_CHECK(A.numel() >= (n + 1) / 2, "packed tensor holds ", A.numel(), " bytes, ", (n + 1) / 2, " needed");
    TORCH_CHECK(absmax.numel() >= (n + blocksize - 1) / blocksize, "absmax holds ", absmax.numel(), " scales, ",
                (n + blocksize - 1) / blocksize, " needed");
    const int dt = to_fp4_dtype(out.scalar_type(), "dequantize");
    c10::DeviceGuard guard(A.device());
    check_status(fp4_hip_dequantize_blockwise(A.data_ptr<uint8_t>(), absmax.data_ptr<float>(), out.data_ptr(), (int)blocksize,
                                              n, dt, table, flags, current_stream(A)));
}

torch::Tensor dequantize_fp4(torch::Tensor A, torch::Tensor absmax, int blocksize, int M, int N, ScalarTypeEnum o_type) {
    check_gpu_contiguous(A, "A");
    check_gpu_contiguous(absmax, "absmax");
    torch::Tensor out = torch::empty({M, N}, torch::TensorOptions().dtype(to_torch(o_type)).device(A.device()));
    dequant_into(A, absmax, out, blocksize, int64_t(M) * N, FP4_TABLE_TREE);
    return out;
}

torch::Tensor dequantize_fp4_codebook(torch::Tensor A, torch::Tensor absmax, torch::Tensor codebook, int M, int N,
                                      int blocksize, int64_t n, ScalarTypeEnum dtype) {
    check_gpu_contiguous(A, "A");
    check_gpu_contiguous(absmax, "absmax");
    check_gpu_contiguous(codebook, "codebook");  // checked but unused, like the reference (its kernels use CODE_PARAM)
    torch::Tensor out = torch::empty({M, N}, torch::TensorOptions().dtype(to_torch(dtype)).device(A.device()));
    dequant_into(A, absmax, out, blocksize, n, FP4_TABLE_CODEBOOK);
    return out;
}

// ---- the dense GEMM of the batch > 1 path: hipBLASLt, called directly ------------------------------------------------------------
// The reference's batch > 1 path is dequant + torch::nn::functional::linear (csrc/torch_fp4.cpp:64-103).  On ROCm every torch GEMM
// entry point (linear, addmm, mm, with or without bias, either BLAS backend) costs ~18.5 us of HOST time per call on this platform
// (tools/exp_blas_host_cost.py; a trivial torch op: 4.3 us), most of it descriptor set-up and the heuristic query repeated on every
// call - and an eager small-batch forward is host-bound, so the dequant + GEMM layer came out slower than the dense layer it replaces
// (BASELINE config 3, `c3_sanity_mlp`).  The same library call with the descriptors, layouts and the chosen algorithm cached per
// (device, dtype, rows, M, K, bias) costs a few microseconds.  Same maths as at::linear: x [rows, K] row-major times W [M, K]^T,
// f32 accumulation (HIPBLAS_COMPUTE_32F, no reduced-precision f32 mode), bias through the library's epilogue, one rounding to T.
// Anything this path does not cover - other dtypes, a bias of another dtype, no algorithm returned, the first call of a device
// arriving under stream capture - goes to at::linear; FP4_QLINEAR_GEMM=aten (or set_qlinear_gemm("aten")) forces that route.
std::atomic<int> g_qlinear_gemm{-1};  // -1 = not decided yet (environment), 0 = at::linear, 1 = hipBLASLt direct

bool qlinear_gemm_direct() {
    int v = g_qlinear_gemm.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = std::getenv("FP4_QLINEAR_GEMM");
        v = (e && std::string(e) == "aten") ? 0 : 1;
        g_qlinear_gemm.store(v, std::memory_order_relaxed);
    }
    return v == 1;
}

std::string set_qlinear_gemm(const std::string &which) {
    TORCH_CHECK(which == "hipblaslt" || which == "aten", "set_qlinear_gemm: 'hipblaslt' or 'aten', got '", which, "'");
    const bool was = qlinear_gemm_direct();
    g_qlinear_gemm.store(which == "hipblaslt" ? 1 : 0, std::memory_order_relaxed);
    return was ? "hipblaslt" : "aten";
}

struct LtPlan {
    hipblasLtMatmulDesc_t desc = nullptr;
    hipblasLtMatrixLayout_t a = nullptr, b = nullptr, c = nullptr;
    hipblasLtMatmulAlgo_t algo{};
    size_t workspace = 0;
    bool usable = false;
    ~LtPlan() {
        if (a) hipblasLtMatrixLayoutDestroy(a);
        if (b) hipblasLtMatrixLayoutDestroy(b);
        if (c) hipblasLtMatrixLayoutDestroy(c);
        if (desc) hipblasLtMatmulDescDestroy(desc);
    }
};

struct LtKey {
    int device, dtype, bias;
    int64_t rows, M, K;
    bool operator==(const LtKey &o) const {
        return device == o.device && dtype == o.dtype && bias == o.bias && rows == o.rows && M == o.M && K == o.K;
    }
};
struct LtKeyHash {
    size_t operator()(const LtKey &k) const {
        size_t h = std::hash<int64_t>()(k.rows * 1000003 + k.M);
        h ^= std::hash<int64_t>()(k.K * 31 + k.dtype * 7 + k.bias * 3 + k.device) + 0x9e3779b97f4a7c15ULL + (h << 6) + (h >> 2);
        return h;
    }
};

// one handle per device, shared by every thread (hipblasLtMatmul is thread-safe on a handle as long as the workspaces differ)
hipblasLtHandle_t lt_handle(int device, bool may_create) {
    static std::mutex mu;
    static hipblasLtHandle_t handles[64] = {};
    if (device < 0 || device >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!handles[device] && may_create) {
        hipblasLtHandle_t h = nullptr;
        if (hipblasLtCreate(&h) == HIPBLAS_STATUS_SUCCESS) handles[device] = h;
    }
    return handles[device];
}

constexpr size_t kLtMaxWorkspace = size_t(76) << 20;  // what torch itself grants hipBLASLt on gfx94x / gfx95x: the heuristic then picks from the same algorithms

// Plans are per THREAD: the bias pointer is an attribute of the matmul descriptor and is set on every call, so a descriptor must not
// be shared by two threads launching at once (the concurrency promise of this boundary: tests/test_gpu_concurrency.py).
LtPlan *lt_plan(hipblasLtHandle_t handle, const LtKey &key, hipDataType dt) {
    thread_local std::unordered_map<LtKey, std::unique_ptr<LtPlan>, LtKeyHash> plans;
    auto it = plans.find(key);
    if (it != plans.end()) return it->second.get();
    // (a prefill loop over ever-changing sequence lengths must not grow this without bound: plans are cheap to rebuild, ~0.1 ms each)
    if (plans.size() >= 4096) plans.clear();
    auto plan = std::make_unique<LtPlan>();
    LtPlan *p = plan.get();
    plans.emplace(key, std::move(plan));  // kept even if unusable: the failure is remembered, not retried on every call
    // column-major view of the row-major operands: D[M, rows] = op_T(A = W as K x M, ld K) * (B = x as K x rows, ld K), ld(D) = M
    const hipblasOperation_t opT = HIPBLAS_OP_T, opN = HIPBLAS_OP_N;
    if (hipblasLtMatmulDescCreate(&p->desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) != HIPBLAS_STATUS_SUCCESS) return p;
    if (hipblasLtMatmulDescSetAttribute(p->desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opT, sizeof(opT)) != HIPBLAS_STATUS_SUCCESS ||
        hipblasLtMatmulDescSetAttribute(p->desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opN, sizeof(opN)) != HIPBLAS_STATUS_SUCCESS)
        return p;
    if (key.bias) {
        const hipblasLtEpilogue_t epi = HIPBLASLT_EPILOGUE_BIAS;
        const int32_t bias_type = static_cast<int32_t>(dt);
        if (hipblasLtMatmulDescSetAttribute(p->desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi)) != HIPBLAS_STATUS_SUCCESS ||
            hipblasLtMatmulDescSetAttribute(p->desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bias_type, sizeof(bias_type)) !=
                HIPBLAS_STATUS_SUCCESS)
            return p;
    }
    if (hipblasLtMatrixLayoutCreate(&p->a, dt, uint64_t(key.K), uint64_t(key.M), key.K) != HIPBLAS_STATUS_SUCCESS ||
        hipblasLtMatrixLayoutCreate(&p->b, dt, uint64_t(key.K), uint64_t(key.rows), key.K) != HIPBLAS_STATUS_SUCCESS ||
        hipblasLtMatrixLayoutCreate(&p->c, dt, uint64_t(key.M), uint64_t(key.rows), key.M) != HIPBLAS_STATUS_SUCCESS)
        return p;
    hipblasLtMatmulPreference_t pref = nullptr;
    if (hipblasLtMatmulPreferenceCreate(&pref) != HIPBLAS_STATUS_SUCCESS) return p;
    const uint64_t max_ws = kLtMaxWorkspace;
    hipblasLtMatmulHeuristicResult_t result{};
    int found = 0;
    if (hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &max_ws, sizeof(max_ws)) == HIPBLAS_STATUS_SUCCESS &&
        hipblasLtMatmulAlgoGetHeuristic(handle, p->desc, p->a, p->b, p->c, p->c, pref, 1, &result, &found) == HIPBLAS_STATUS_SUCCESS &&
        found > 0 && result.state == HIPBLAS_STATUS_SUCCESS && result.workspaceSize <= kLtMaxWorkspace) {
        p->algo = result.algo;
        p->workspace = result.workspaceSize;
        p->usable = true;
    }
    hipblasLtMatmulPreferenceDestroy(pref);
    return p;
}

// out [.., M] = x [.., K] @ weight[M, K]^T (+ bias) through the cached hipBLASLt plan; false = not covered (the caller uses at::linear)
bool lt_linear(const torch::Tensor &x, const torch::Tensor &weight, const c10::optional<torch::Tensor> &bias, torch::Tensor &out) {
    hipDataType dt;
    switch (x.scalar_type()) {
        case torch::kBFloat16: dt = HIP_R_16BF; break;
        case torch::kFloat16: dt = HIP_R_16F; break;
        case torch::kFloat32: dt = HIP_R_32F; break;
        default: return false;
    }
    const int64_t K = weight.size(1), M = weight.size(0);
    if (x.dim() < 1 || x.size(-1) != K || !x.is_cuda() || x.device() != weight.device() || weight.scalar_type() != x.scalar_type() ||
        K <= 0 || M <= 0)
        return false;
    const int64_t rows = x.numel() / K;
    if (rows <= 0 || rows > (int64_t(1) << 30)) return false;
    if (bias.has_value() && (!bias->is_cuda() || bias->device() != x.device() || bias->scalar_type() != x.scalar_type() ||
                             bias->numel() != M || !bias->is_contiguous()))
        return false;
    const int device = x.device().index();
    hipStream_t stream = c10::hip::getCurrentHIPStream(device).stream();
    hipblasLtHandle_t handle = lt_handle(device, false);
    if (!handle) {  // first GEMM on this device: creating the library handle is not permitted under stream capture - hand the call to
                    // at::linear, which (measured) fails there exactly as a dense torch GEMM does: warm up before capturing, as always
        hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(stream, &capturing);
        if (capturing != hipStreamCaptureStatusNone) return false;
        handle = lt_handle(device, true);
        if (!handle) return false;
    }
    LtPlan *plan = lt_plan(handle, LtKey{device, int(dt), bias.has_value() ? 1 : 0, rows, M, K}, dt);
    if (!plan->usable) return false;
    const torch::Tensor xc = x.is_contiguous() ? x : x.contiguous();
    auto shape = x.sizes().vec();
    shape.back() = M;
    out = torch::empty(shape, x.options());
    torch::Tensor ws;
    void *ws_ptr = nullptr;
    if (plan->workspace > 0) {  // from torch's caching allocator: stream-ordered, no synchronisation, fine under graph capture
        ws = torch::empty({int64_t(plan->workspace)}, x.options().dtype(torch::kUInt8));
        ws_ptr = ws.data_ptr();
    }
    if (bias.has_value()) {
        const void *bp = bias->data_ptr();
        if (hipblasLtMatmulDescSetAttribute(plan->desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bp, sizeof(bp)) != HIPBLAS_STATUS_SUCCESS) return false;
    }
    const float alpha = 1.0f, beta = 0.0f;
    const hipblasStatus_t st = hipblasLtMatmul(handle, plan->desc, &alpha, weight.data_ptr(), plan->a, xc.data_ptr(), plan->b, &beta,
                                               out.data_ptr(), plan->c, out.data_ptr(), plan->c, &plan->algo, ws_ptr, plan->workspace, stream);
    if (st != HIPBLAS_STATUS_SUCCESS) {
        plan->usable = false;  // do not try this shape again
        return false;
    }
    return true;
}

torch::Tensor qlinear_impl(const torch::Tensor &A_in, const torch::Tensor &A, const torch::Tensor &absmax, int M, int N,
                           int blocksize, int table, const c10::optional<torch::Tensor> &bias) {
    check_gpu_contiguous(A, "A");
    check_gpu_contiguous(absmax, "absmax");
    torch::Tensor weight = torch::empty({M, N}, A_in.options());
    // the GEMM below reads the weight straight back: keep it in L2 / Infinity Cache (plain stores)
    dequant_into(A, absmax, weight, blocksize, int64_t(M) * N, table, FP4_DEQUANT_KEEP_CACHED);
    if (qlinear_gemm_direct()) {
        c10::DeviceGuard guard(A_in.device());
        torch::Tensor out;
        if (lt_linear(A_in, weight, bias, out)) return out;
    }
    return bias.has_value() ? at::linear(A_in, weight, *bias) : at::linear(A_in, weight);
}

torch::Tensor qlinear(torch::Tensor A_in, torch::Tensor A, torch::Tensor absmax, int M, int N, int blocksize) {
    return qlinear_impl(A_in, A, absmax, M, N, blocksize, FP4_TABLE_TREE, c10::nullopt);
}
torch::Tensor qlinear_bias(torch::Tensor A_in, torch::Tensor A, torch::Tensor absmax, int M, int N, int blocksize,
                           torch::Tensor bias) {
    return qlinear_impl(A_in, A, absmax, M, N, blocksize, FP4_TABLE_TREE, bias);
}
torch::Tensor qlinear_codebook(torch::Tensor A_in, torch::Tensor A, torch::Tensor absmax, torch::Tensor codebook, int M, int N,
                               int blocksize) {
    (void)codebook;
    return qlinear_impl(A_in, A, absmax, M, N, blocksize, FP4_TABLE_CODEBOOK, c10::nullopt);
}
torch::Tensor qlinear_codebook_bias(torch::Tensor A_in, torch::Tensor A, torch::Tensor absmax, torch::Tensor codebook, int M,
                                    int N, int blocksize, torch::Tensor bias) {
    (void)codebook;
    return qlinear_impl(A_in, A, absmax, M, N, blocksize, FP4_TABLE_CODEBOOK, bias);
}

torch::Tensor gemv_impl(const torch::Tensor &A, const torch::Tensor &B, const torch::Tensor &absmax,
                        const torch::Tensor &datatype, int blocksize, ScalarTypeEnum dtype, const std::vector<uint32_t> &Bshape,
                        const c10::optional<torch::Tensor> &bias) {
    check_gpu_contiguous(A, "A");
    check_gpu_contiguous(B, "B");
    check_gpu_contiguous(absmax, "absmax");
    check_gpu_contiguous(datatype, "datatype");
    TORCH_CHECK(Bshape.size() == 2, "Bshape must be the [out_features, in_features] of the quantised weight");
    const int64_t m = Bshape[0], k = Bshape[1];
    const torch::ScalarType st = to_torch(dtype);
    // reference: per-dtype TORCH_CHECKs of csrc/gemv_fp4_optimized.cu:303-305,323-325,343-345
    TORCH_CHECK(A.scalar_type() == st, "gemv_fp4: dtype argument (", st, ") must equal the activation dtype (", A.scalar_type(), ")");
    TORCH_CHECK(absmax.scalar_type() == torch::kFloat32, "Only fp32 absmax is supported");
    TORCH_CHECK(datatype.scalar_type() == torch::kFloat32, "Only fp32 code is supported");
    TORCH_CHECK(B.dtype() == torch::kUInt8, "B must be uint8");
    TORCH_CHECK(A.dim() == 2 || A.dim() == 3, "gemv_fp4: activation must be [1, K] or [1, 1, K]");
    TORCH_CHECK(A.numel() == k && A.size(-1) == k, "gemv_fp4 is batch-1 only: activation has ", A.numel(),
                " elements, in_features is ", k);
    TORCH_CHECK(B.numel() * 2 >= m * k, "B holds ", B.numel(), " bytes, ", m * k / 2, " needed");
    TORCH_CHECK(absmax.numel() * int64_t(blocksize) >= m * k, "absmax too small for a ", m, "x", k, " weight");
    TORCH_CHECK(B.device() == A.device() && absmax.device() == A.device(), "all tensors must be on one device");
    // reference output shape: [A.size(0), m] or [A.size(0), A.size(1), m] (csrc/gemv_fp4_optimized.cu:296-299)
    torch::Tensor out = A.dim() == 3 ? torch::empty({A.size(0), A.size(1), m}, A.options()) : torch::empty({A.size(0), m}, A.options());
    const void *bias_ptr = nullptr;
    torch::Tensor bias_c;
    if (bias.has_value()) {
        TORCH_CHECK(bias->is_cuda() && bias->numel() == m && bias->scalar_type() == st, "bias must be a [", m, "] tensor of ", st);
        bias_c = bias->contiguous();
        bias_ptr = bias_c.data_ptr();
    }
    c10::DeviceGuard guard(A.device());
    check_status(fp4_hip_gemv(A.data_ptr(), B.data_ptr<uint8_t>(), absmax.data_ptr<float>(), bias_ptr, out.data_ptr(), m, k,
                              blocksize, (int)dtype, current_stream(A)));
    return out;
}

torch::Tensor gemv_fp4(torch::Tensor A, torch::Tensor B, torch::Tensor absmax, torch::Tensor datatype, int blocksize,
                       ScalarTypeEnum dtype, std::vector<uint32_t> Bshape) {
    return gemv_impl(A, B, absmax, datatype, blocksize, dtype, Bshape, c10::nullopt);
}
torch::Tensor gemv_fp4_bias(torch::Tensor A, torch::Tensor B, torch::Tensor absmax, torch::Tensor datatype, int blocksize,
                            ScalarTypeEnum dtype, std::vector<uint32_t> Bshape, torch::Tensor bias) {
    return gemv_impl(A, B, absmax, datatype, blocksize, dtype, Bshape, bias);
}

// GEMV with a fused epilogue (fp4_hip_gemv_fused): bias, residual add, and - for a weight whose rows interleave a gate and an
// up projection - silu(gate) * up.  A is [1, K] / [1, 1, K]; the result is [.., m] or, for the gated epilogue, [.., m / 2].
torch::Tensor gemv_fp4_fused(torch::Tensor A, torch::Tensor B, torch::Tensor absmax, int blocksize, std::vector<uint32_t> Bshape,
                             c10::optional<torch::Tensor> bias, c10::optional<torch::Tensor> residual, int epilogue) {
    check_gpu_contiguous(A, "A");
    check_gpu_contiguous(B, "B");
    check_gpu_contiguous(absmax, "absmax");
    TORCH_CHECK(Bshape.size() == 2, "Bshape must be the [out_features, in_features] of the quantised weight");
    const int64_t m = Bshape[0], k = Bshape[1];
    TORCH_CHECK(epilogue == FP4_EPILOGUE_NONE || epilogue == FP4_EPILOGUE_SILU_MUL_PAIRS, "gemv_fp4_fused: unknown epilogue ", epilogue);
    const int64_t m_out = epilogue == FP4_EPILOGUE_SILU_MUL_PAIRS ? m / 2 : m;
    const int dt = to_fp4_dtype(A.scalar_type(), "gemv_fp4_fused");
    TORCH_CHECK(absmax.scalar_type() == torch::kFloat32, "Only fp32 absmax is supported");
    TORCH_CHECK(B.dtype() == torch::kUInt8, "B must be uint8");
    TORCH_CHECK(A.dim() >= 1 && A.numel() == k && A.size(-1) == k, "gemv_fp4_fused is batch-1 only: activation has ", A.numel(),
                " elements, in_features is ", k);
    TORCH_CHECK(B.numel() * 2 >= m * k, "B holds ", B.numel(), " bytes, ", m * k / 2, " needed");
    TORCH_CHECK(absmax.numel() * int64_t(blocksize) >= m * k, "absmax too small for a ", m, "x", k, " weight");
    TORCH_CHECK(B.device() == A.device() && absmax.device() == A.device(), "all tensors must be on one device");
    auto shape = A.sizes().vec();
    shape.back() = m_out;
    torch::Tensor out = torch::empty(shape, A.options());
    const void *bias_ptr = nullptr, *res_ptr = nullptr;
    torch::Tensor bias_c, res_c;
    if (bias.has_value()) {
        TORCH_CHECK(bias->is_cuda() && bias->numel() == m && bias->scalar_type() == A.scalar_type(), "bias must be a [", m,
                    "] tensor of the activation dtype");
        bias_c = bias->contiguous();
        bias_ptr = bias_c.data_ptr();
    }
    if (residual.has_value()) {
        TORCH_CHECK(residual->is_cuda() && residual->numel() == m_out && residual->scalar_type() == A.scalar_type() &&
                        residual->device() == A.device(),
                    "residual must hold ", m_out, " elements of the activation dtype on the activation's device");
        res_c = residual->contiguous();
        res_ptr = res_c.data_ptr();
    }
    c10::DeviceGuard guard(A.device());
    check_status(fp4_hip_gemv_fused(A.data_ptr(), B.data_ptr<uint8_t>(), absmax.data_ptr<float>(), bias_ptr, res_ptr, out.data_ptr(), m, k,
                                    blocksize, dt, epilogue, current_stream(A)));
    return out;
}

// fused small-batch product: A [..., K] with 1..128 rows in total -> [..., m]; raises if the shape is not covered
torch::Tensor gemm_small_fp4(torch::Tensor A, torch::Tensor B, torch::Tensor absmax, int blocksize, std::vector<uint32_t> Bshape,
                             c10::optional<torch::Tensor> bias) {
    check_gpu_contiguous(A, "A");
    check_gpu_contiguous(B, "B");
    check_gpu_contiguous(absmax, "absmax");
    TORCH_CHECK(Bshape.size() == 2, "Bshape must be [out_features, in_features]");
    const int64_t m = Bshape[0], k = Bshape[1];
    TORCH_CHECK(A.dim() >= 1 && A.size(-1) == k, "gemm_small_fp4: last dim of the activation must be in_features = ", k);
    const int64_t rows = A.numel() / k;
    TORCH_CHECK(rows >= 1 && rows <= 128, "gemm_small_fp4 covers 1..128 activation rows, got ", rows);
    TORCH_CHECK(B.dtype() == torch::kUInt8 && B.numel() * 2 >= m * k, "B too small for a ", m, "x", k, " weight");
    TORCH_CHECK(absmax.scalar_type() == torch::kFloat32 && absmax.numel() * int64_t(blocksize) >= m * k, "absmax too small");
    const int dt = to_fp4_dtype(A.scalar_type(), "gemm_small_fp4");
    auto shape = A.sizes().vec();
    shape.back() = m;
    torch::Tensor out = torch::empty(shape, A.options());
    const void *bias_ptr = nullptr;
    torch::Tensor bias_c;
    if (bias.has_value()) {
        TORCH_CHECK(bias->is_cuda() && bias->numel() == m && bias->scalar_type() == A.scalar_type(), "bias must be a [", m,
                    "] tensor of the activation dtype");
        bias_c = bias->contiguous();
        bias_ptr = bias_c.data_ptr();
    }
    c10::DeviceGuard guard(A.device());
    // short weight x long rows: the split-K path wants a scratch buffer (torch's caching allocator: no sync, graph-capturable)
    const int64_t ws_bytes = fp4_hip_gemm_small_ws_bytes(rows, m, k, blocksize, dt);
    if (ws_bytes > 0) {
        torch::Tensor ws = torch::empty({ws_bytes}, A.options().dtype(torch::kUInt8));
        check_status(fp4_hip_gemm_small_ws(A.data_ptr(), B.data_ptr<uint8_t>(), absmax.data_ptr<float>(), bias_ptr, nullptr, out.data_ptr(),
                                           rows, m, k, blocksize, dt, FP4_EPILOGUE_NONE, ws.data_ptr(), ws_bytes, current_stream(A)));
        return out;
    }
    check_status(fp4_hip_gemm_small(A.data_ptr(), B.data_ptr<uint8_t>(), absmax.data_ptr<float>(), bias_ptr, out.data_ptr(), rows, m,
                                    k, blocksize, dt, current_stream(A)));
    return out;
}

// the small-batch product with the fused epilogues (fp4_hip_gemm_small_fused): A [..., K] with 1..128 rows -> [..., m] or [..., m / 2]
torch::Tensor gemm_small_fp4_fused(torch::Tensor A, torch::Tensor B, torch::Tensor absmax, int blocksize, std::vector<uint32_t> Bshape,
                                   c10::optional<torch::Tensor> bias, c10::optional<torch::Tensor> residual, int epilogue) {
    check_gpu_contiguous(A, "A");
    check_gpu_contiguous(B, "B");
    check_gpu_contiguous(absmax, "absmax");
    TORCH_CHECK(Bshape.size() == 2, "Bshape must be [out_features, in_features]");
    const int64_t m = Bshape[0], k = Bshape[1];
    TORCH_CHECK(epilogue == FP4_EPILOGUE_NONE || epilogue == FP4_EPILOGUE_SILU_MUL_PAIRS, "gemm_small_fp4_fused: unknown epilogue ", epilogue);
    const int64_t m_out = epilogue == FP4_EPILOGUE_SILU_MUL_PAIRS ? m / 2 : m;
    TORCH_CHECK(A.dim() >= 1 && A.size(-1) == k, "gemm_small_fp4_fused: last dim of the activation must be in_features = ", k);
    const int64_t rows = A.numel() / k;
    TORCH_CHECK(rows >= 1 && rows <= 128, "gemm_small_fp4_fused covers 1..128 activation rows, got ", rows);
    TORCH_CHECK(B.dtype() == torch::kUInt8 && B.numel() * 2 >= m * k, "B too small for a ", m, "x", k, " weight");
    TORCH_CHECK(absmax.scalar_type() == torch::kFloat32 && absmax.numel() * int64_t(blocksize) >= m * k, "absmax too small");
    const int dt = to_fp4_dtype(A.scalar_type(), "gemm_small_fp4_fused");
    auto shape = A.sizes().vec();
    shape.back() = m_out;
    torch::Tensor out = torch::empty(shape, A.options());
    const void *bias_ptr = nullptr, *res_ptr = nullptr;
    torch::Tensor bias_c, res_c;
    if (bias.has_value()) {
        TORCH_CHECK(bias->is_cuda() && bias->numel() == m && bias->scalar_type() == A.scalar_type(), "bias must be a [", m,
                    "] tensor of the activation dtype");
        bias_c = bias->contiguous();
        bias_ptr = bias_c.data_ptr();
    }
    if (residual.has_value()) {
        TORCH_CHECK(residual->is_cuda() && residual->numel() == rows * m_out && residual->scalar_type() == A.scalar_type() &&
                        residual->device() == A.device(),
                    "residual must hold ", rows * m_out, " elements of the activation dtype on the activation's device");
        res_c = residual->contiguous();
        res_ptr = res_c.data_ptr();
    }
    c10::DeviceGuard guard(A.device());
    const int64_t ws_bytes = fp4_hip_gemm_small_ws_bytes(rows, m, k, blocksize, dt);
    if (ws_bytes > 0) {
        torch::Tensor ws = torch::empty({ws_bytes}, A.options().dtype(torch::kUInt8));
        check_status(fp4_hip_gemm_small_ws(A.data_ptr(), B.data_ptr<uint8_t>(), absmax.data_ptr<float>(), bias_ptr, res_ptr, out.data_ptr(),
                                           rows, m, k, blocksize, dt, epilogue, ws.data_ptr(), ws_bytes, current_stream(A)));
        return out;
    }
    check_status(fp4_hip_gemm_small_fused(A.data_ptr(), B.data_ptr<uint8_t>(), absmax.data_ptr<float>(), bias_ptr, res_ptr, out.data_ptr(),
                                          rows, m, k, blocksize, dt, epilogue, current_stream(A)));
    return out;
}

// f32 partial sums of a K-split shard: [1, m] float32 (see fp4_hip_gemv_partial)
torch::Tensor gemv_fp4_partial(torch::Tensor A, torch::Tensor B, torch::Tensor absmax, int blocksize, std::vector<uint32_t> Bshape) {
    check_gpu_contiguous(A, "A");
    check_gpu_contiguous(B, "B");
    check_gpu_contiguous(absmax, "absmax");
    TORCH_CHECK(Bshape.size() == 2, "Bshape must be [out_features, in_features_of_this_shard]");
    const int64_t m = Bshape[0], k = Bshape[1];
    TORCH_CHECK(A.numel() == k, "gemv_fp4_partial is batch-1 only: activation has ", A.numel(), " elements, shard K is ", k);
    TORCH_CHECK(B.dtype() == torch::kUInt8 && B.numel() * 2 >= m * k, "B too small for a ", m, "x", k, " shard");
    TORCH_CHECK(absmax.scalar_type() == torch::kFloat32 && absmax.numel() * int64_t(blocksize) >= m * k, "absmax too small");
    const int dt = to_fp4_dtype(A.scalar_type(), "gemv_fp4_partial");
    torch::Tensor out = torch::empty({1, m}, A.options().dtype(torch::kFloat32));
    c10::DeviceGuard guard(A.device());
    check_status(fp4_hip_gemv_partial(A.data_ptr(), B.data_ptr<uint8_t>(), absmax.data_ptr<float>(), out.data_ptr<float>(), m, k,
                                      blocksize, dt, current_stream(A)));
    return out;
}

// bitsandbytes-format FP4 quantisation of a float tensor: returns (packed uint8[ceil(n/2), 1], absmax float32[ceil(n/bs)])
std::tuple<torch::Tensor, torch::Tensor> quantize_fp4(torch::Tensor W, int blocksize) {
    check_gpu_contiguous(W, "W");
    const int dt = to_fp4_dtype(W.scalar_type(), "quantize_fp4");
    const int64_t n = W.numel();
    TORCH_CHECK(blocksize > 0, "blocksize must be positive");
    torch::Tensor packed = torch::empty({(n + 1) / 2, 1}, torch::TensorOptions().dtype(torch::kUInt8).device(W.device()));
    torch::Tensor absmax = torch::empty({(n + blocksize - 1) / blocksize}, torch::TensorOptions().dtype(torch::kFloat32).device(W.device()));
    c10::DeviceGuard guard(W.device());
    check_status(fp4_hip_quantize_blockwise(W.data_ptr(), dt, packed.data_ptr<uint8_t>(), absmax.data_ptr<float>(), n, blocksize,
                                            current_stream(W)));
    return {packed, absmax};
}

// ---- one-shot all-reduce plumbing (fp4_hip_comm_* / fp4_hip_allreduce_oneshot) ---------------------------------------
// Buffers are identified by their device address (an int on the Python side); torch_bnb_fp4/comm.py owns their lifetime.
std::tuple<int64_t, py::bytes, int> comm_alloc(int world, int64_t capacity, int device) {
    const int64_t bytes = fp4_hip_comm_bytes(world, capacity);
    TORCH_CHECK(bytes > 0, "comm_alloc: bad world / capacity");
    c10::DeviceGuard guard(c10::Device(c10::kCUDA, (c10::DeviceIndex)device));
    void *p = nullptr;
    uint8_t handle[64];
    int kind = -1;
    check_status(fp4_hip_comm_alloc(bytes, &p, handle, &kind));
    return {reinterpret_cast<int64_t>(p), py::bytes(reinterpret_cast<const char *>(handle), 64), kind};
}

int64_t comm_open(const std::string &handle, int device) {
    TORCH_CHECK(handle.size() == 64, "comm_open: an IPC handle is 64 bytes");
    c10::DeviceGuard guard(c10::Device(c10::kCUDA, (c10::DeviceIndex)device));
    void *p = nullptr;
    check_status(fp4_hip_comm_open(reinterpret_cast<const uint8_t *>(handle.data()), &p));
    return reinterpret_cast<int64_t>(p);
}

void comm_close(int64_t ptr) { check_status(fp4_hip_comm_close(reinterpret_cast<void *>(ptr))); }
void comm_free(int64_t ptr) { check_status(fp4_hip_comm_free(reinterpret_cast<void *>(ptr))); }

std::vector<uint32_t> comm_status(int64_t own_ptr) {
    std::vector<uint32_t> out(4);
    check_status(fp4_hip_comm_status(reinterpret_cast<const void *>(own_ptr), out.data()));
    return out;
}

void comm_clear_status(int64_t own_ptr) { check_status(fp4_hip_comm_clear_status(reinterpret_cast<void *>(own_ptr))); }

// partial: f32 [.., M] on this rank's GPU; returns T [.., M] = T(sum over ranks) (+ bias) (+ residual)
torch::Tensor allreduce_oneshot(torch::Tensor partial, std::vector<int64_t> peers, int rank, int64_t capacity, ScalarTypeEnum dtype,
                                c10::optional<torch::Tensor> bias, c10::optional<torch::Tensor> residual, int64_t timeout_us) {
    check_gpu_contiguous(partial, "partial");
    TORCH_CHECK(partial.scalar_type() == torch::kFloat32, "allreduce_oneshot: the partial sums must be float32");
    const int world = (int)peers.size();
    const int64_t m = partial.numel();
    const torch::ScalarType st = to_torch(dtype);
    torch::Tensor out = torch::empty(partial.sizes(), partial.options().dtype(st));
    const void *bias_ptr = nullptr, *res_ptr = nullptr;
    torch::Tensor bias_c, res_c;
    if (bias.has_value()) {
        TORCH_CHECK(bias->is_cuda() && bias->numel() == m && bias->scalar_type() == st, "bias must hold ", m, " elements of ", st);
        bias_c = bias->contiguous();
        bias_ptr = bias_c.data_ptr();
    }
    if (residual.has_value()) {
        TORCH_CHECK(residual->is_cuda() && residual->numel() == m && residual->scalar_type() == st, "residual must hold ", m,
                    " elements of ", st);
        res_c = residual->contiguous();
        res_ptr = res_c.data_ptr();
    }
    std::vector<void *> bufs(world);
    for (int p = 0; p < world; ++p) bufs[p] = reinterpret_cast<void *>(peers[p]);
    c10::DeviceGuard guard(partial.device());
    check_status(fp4_hip_allreduce_oneshot(partial.data_ptr<float>(), bufs.data(), rank, world, m, capacity, bias_ptr, res_ptr,
                                           out.data_ptr(), (int)dtype, timeout_us, current_stream(partial)));
    return out;
}

torch::Tensor code_table(const std::string &name) {
    TORCH_CHECK(name == "codebook" || name == "tree", "code_table: name must be 'codebook' or 'tree'");
    torch::Tensor t = torch::empty({16}, torch::kFloat32);
    check_status(fp4_hip_code_table(name == "tree" ? FP4_TABLE_TREE : FP4_TABLE_CODEBOOK, t.data_ptr<float>()));
    return t;
}

void set_kernel_variant(const std::string &kernel, int variant) { check_status(fp4_hip_set_variant(kernel.c_str(), variant)); }

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "MI355X (gfx950) FP4 dequant / fused GEMV operators; same surface as aredden/torch-bnb-fp4's torch_bnb_fp4_ext";
    pybind11::enum_<ScalarTypeEnum>(m, "ScalarType")
        .value("bfloat16", ScalarTypeEnum::bfloat16)
        .value("float16", ScalarTypeEnum::float16)
        .value("float32", ScalarTypeEnum::float32)
        .export_values();

    m.def("dequantize_fp4", &dequantize_fp4, "FP4 -> T dequant, tree constants: (A, absmax, blocksize, M, N, o_type)");
    m.def("dequantize_fp4_codebook", &dequantize_fp4_codebook,
          "FP4 -> T dequant, CODE_PARAM table: (A, absmax, codebook, M, N, blocksize, n, dtype)");
    m.def("gemv_fp4", &gemv_fp4, "fused batch-1 FP4 GEMV: (A, B, absmax, datatype, blocksize, dtype, Bshape)");
    m.def("qlinear", &qlinear, "tree dequant + linear: (A_in, A, absmax, M, N, blocksize)");
    m.def("qlinear_bias", &qlinear_bias, "tree dequant + linear + bias");
    m.def("qlinear_codebook", &qlinear_codebook, "codebook dequant + linear: (A_in, A, absmax, codebook, M, N, blocksize)");
    m.def("qlinear_codebook_bias", &qlinear_codebook_bias, "codebook dequant + linear + bias");
    // extras
    m.def("gemv_fp4_bias", &gemv_fp4_bias, "gemv_fp4 with the bias add fused into the epilogue");
    m.def("gemv_fp4_fused", &gemv_fp4_fused,
          "GEMV with a fused epilogue: (A, B, absmax, blocksize, Bshape, bias|None, residual|None, epilogue) ; epilogue 0 = bias/residual, "
          "1 = silu(gate) * up over interleaved rows");
    m.def("gemm_small_fp4", &gemm_small_fp4, "fused FP4 product for 1..128 activation rows: (A, B, absmax, blocksize, Bshape, bias|None)");
    m.def("gemm_small_fp4_fused", &gemm_small_fp4_fused,
          "fused FP4 product for 1..128 rows with an epilogue: (A, B, absmax, blocksize, Bshape, bias|None, residual|None, epilogue)");
    m.def("gemv_fp4_partial", &gemv_fp4_partial, "f32 partial sums of a K-split shard: (A, B, absmax, blocksize, Bshape)");
    m.def("comm_alloc", &comm_alloc, "(world, capacity, device) -> (buffer address, 64-byte IPC handle, memory kind)");
    m.def("comm_open", &comm_open, "(handle, device) -> mapped address of a peer's buffer");
    m.def("comm_close", &comm_close, "unmap a peer's buffer");
    m.def("comm_free", &comm_free, "free the own buffer");
    m.def("comm_status", &comm_status, "(own buffer) -> [epoch, busy, status, lanes timed out]  (synchronous)");
    m.def("comm_clear_status", &comm_clear_status, "(own buffer): zero the sticky status word and the timed-out lane count (synchronous)");
    m.def("allreduce_oneshot", &allreduce_oneshot,
          "one-shot all-reduce of f32 partials over peer-mapped slots: (partial, peers, rank, capacity, dtype, bias|None, residual|None, timeout_us)");
    m.def("quantize_fp4", &quantize_fp4, "blockwise FP4 quantiser: (W, blocksize) -> (packed, absmax)");
    m.def("code_table", &code_table, "16-entry code table as a CPU float tensor");
    m.def("set_kernel_variant", &set_kernel_variant, "benchmark hook: select a kernel geometry");
    m.def("set_qlinear_gemm", &set_qlinear_gemm,
          "which dense GEMM the qlinear* ops call after the dequant: 'hipblaslt' (direct, cached plans; default) or 'aten' (at::linear); "
          "returns the previous setting");
    m.attr("EPILOGUE_NONE") = (int)FP4_EPILOGUE_NONE;
    m.attr("EPILOGUE_SILU_MUL_PAIRS") = (int)FP4_EPILOGUE_SILU_MUL_PAIRS;
    m.attr("abi_version") = fp4_hip_abi_version();
}
