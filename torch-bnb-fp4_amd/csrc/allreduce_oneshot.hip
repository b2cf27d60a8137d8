// One-shot all-reduce of the K-split (row-parallel) layers' f32 partial sums over peer-mapped buffers.
//
// The reference has no multi-GPU path.  In tensor-parallel decode the o- and down-projection are K-split: every GPU
// produces a full-length f32 partial (fp4_hip_gemv_partial, 16 KiB at M = 4096) and the G partials must be summed
// before the one rounding to T.  At this size a ring all-reduce is pure latency (2(G-1) serial hops over point-to-point
// xGMI links).  The MI355X-native form uses all 7 links at once, one hop:
//
//   * every rank owns a slot buffer  slots[2][G][capacity]  of 8-byte granules {tag = epoch, value = f32 bits} that all
//     its peers have mapped (hipIpcOpenMemHandle);
//   * rank r writes its partial as granules into slot r of EVERY rank's buffer (system-scope, write-through 8-byte
//     stores: the data is its own flag, no separate flag, no fence), then sweeps its own G slots until every tag equals
//     the epoch, sums them in rank order 0..G-1 (so every rank computes bit-identical results), adds bias / residual and
//     rounds once;
//   * the epoch is counted in device memory (kernel arguments are frozen under HIP-graph replay), slots are double-
//     buffered by epoch parity: a rank can only reach call n+1 after it has read every peer's call-n data, which those
//     peers wrote after finishing call n-1, so a slot of parity n is never overwritten while someone still reads it;
//   * polling is bounded by wall time (s_memrealtime): on a time-out the wave records {epoch, peer} in the header's status
//     word, writes NaN and leaves - a peer that is not co-scheduled can never hang the GPU; the host raises on the status.
//
// No RCCL call, no host synchronisation: the step is HIP-graph capturable.  Cross-GPU performance is UNMEASURED (the
// development box has one GPU); correctness is developed with two ranks sharing one device, where the hand-off still
// crosses XCDs (non-coherent L2s) and therefore exercises the same cache-bypassing accesses.
#include <cstdlib>
#include <cstring>

#include "fp4_common.h"

namespace fp4 {
namespace {

constexpr int kMaxRanks = 16;
constexpr int kHeaderBytes = 256;

struct CommHeader {      // first 256 bytes of a rank's buffer; touched by the owning rank only
    uint32_t epoch;      // number of completed calls
    uint32_t done;       // workgroups of the running call that have finished
    uint32_t status;     // 0 = ok; otherwise (epoch << 8) | (peer + 1) of the first time-out
    uint32_t timeouts;   // number of lanes that gave up
};

struct PeerTable {
    uint64_t *slots[kMaxRanks];  // peer p's granule array (its buffer + kHeaderBytes)
};

typedef __attribute__((address_space(1))) unsigned long long gu64;

__device__ __forceinline__ void store_granule(uint64_t *p, uint64_t v) {
    __hip_atomic_store((gu64 *)p, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // one 8-byte sc0 sc1 global store
}
__device__ __forceinline__ uint64_t load_granule(const uint64_t *p) {
    return __hip_atomic_load((const gu64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <int DT>
__global__ __launch_bounds__(256) void allreduce_oneshot_kernel(const float *__restrict__ partial, PeerTable peers,
                                                                CommHeader *hdr, int rank, int world, int M, int64_t capacity,
                                                                const void *__restrict__ biasv, const void *residualv, void *outv,
                                                                uint64_t timeout_ticks) {
    const uint32_t epoch = hdr->epoch + 1u;  // uniform; written back by the last workgroup to finish
    const int64_t par_off = int64_t(epoch & 1u) * world * capacity;
    const uint64_t tag = uint64_t(epoch) << 32;
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz, constant rate
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < M; e += gridDim.x * blockDim.x) {
        // 1. publish: this rank's value into slot `rank` of every rank's buffer (own buffer included)
        const uint64_t g = tag | __builtin_bit_cast(uint32_t, partial[e]);
        for (int p = 0; p < world; ++p) store_granule(peers.slots[p] + par_off + int64_t(rank) * capacity + e, g);
        // 2. gather: sweep the G granules of element e in the own buffer until every tag is this call's
        const uint64_t *mine = peers.slots[rank] + par_off + e;
        float sum = 0.0f;
        bool ok = true;
        for (int j = 0; j < world && ok; ++j) {
            uint64_t v = load_granule(mine + int64_t(j) * capacity);
            while ((v >> 32) != epoch) {
                if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) {
                    ok = false;
                    atomicCAS(&hdr->status, 0u, (epoch << 8) | uint32_t(j + 1));
                    atomicAdd(&hdr->timeouts, 1u);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
                v = load_granule(mine + int64_t(j) * capacity);
            }
            sum += __builtin_bit_cast(float, uint32_t(v));  // rank order 0..G-1 on every rank: bit-identical everywhere
        }
        if (!ok) sum = __builtin_nanf("");
        // 3. epilogue: one rounding of the full sum, then bias / residual as rounded adds (the unsharded layer's semantics)
        if constexpr (DT == FP4_DTYPE_F32) {
            const float *bias = reinterpret_cast<const float *>(biasv), *residual = reinterpret_cast<const float *>(residualv);
            float t = bias ? sum + bias[e] : sum;
            reinterpret_cast<float *>(outv)[e] = residual ? t + residual[e] : t;
        } else {
            const uint16_t *bias = reinterpret_cast<const uint16_t *>(biasv), *residual = reinterpret_cast<const uint16_t *>(residualv);
            uint16_t t = from_f32<DT>(sum);
            if (bias) t = from_f32<DT>(to_f32<DT>(t) + to_f32<DT>(bias[e]));
            if (residual) t = from_f32<DT>(to_f32<DT>(t) + to_f32<DT>(residual[e]));
            reinterpret_cast<uint16_t *>(outv)[e] = t;
        }
    }
    // the last workgroup to finish advances the epoch for the next call (every workgroup has read it by then)
    __syncthreads();
    if (threadIdx.x == 0) {
        if (atomicAdd(&hdr->done, 1u) == gridDim.x - 1) {
            hdr->done = 0u;
            hdr->epoch = epoch;
        }
    }
}

}  // namespace
}  // namespace fp4

extern "C" int64_t fp4_hip_comm_bytes(int world, int64_t capacity) {
    if (world < 1 || world > fp4::kMaxRanks || capacity < 1) return -1;
    return int64_t(fp4::kHeaderBytes) + int64_t(2) * world * capacity * 8;
}

extern "C" int fp4_hip_comm_alloc(int64_t bytes, void **ptr, uint8_t handle_out[64], int *kind_out) {
    using namespace fp4;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the ABI passes IPC handles as 64 opaque bytes");
    if (bytes < kHeaderBytes || !ptr || !handle_out) {
        set_error("fp4_hip_comm_alloc: bad argument");
        return FP4_ERR_INVALID_ARGUMENT;
    }
    // Peer-written memory must not be cached by the owner's L2 as if only the owner wrote it: ask for uncached device
    // memory first, fine-grained next; plain hipMalloc last (the kernel's accesses are system-scope either way).
    const char *want = std::getenv("FP4_COMM_ALLOC");
    const int first = !want ? 0 : (!std::strcmp(want, "finegrained") ? 1 : (!std::strcmp(want, "default") ? 2 : 0));
    for (int kind = first; kind < 3; ++kind) {
        void *p = nullptr;
        hipError_t e = kind == 0   ? hipExtMallocWithFlags(&p, size_t(bytes), hipDeviceMallocUncached)
                       : kind == 1 ? hipExtMallocWithFlags(&p, size_t(bytes), hipDeviceMallocFinegrained)
                                   : hipMalloc(&p, size_t(bytes));
        if (e != hipSuccess) {
            (void)hipGetLastError();
            continue;
        }
        hipIpcMemHandle_t h;
        if (hipMemset(p, 0, size_t(bytes)) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
            hipIpcGetMemHandle(&h, p) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(p);
            continue;
        }
        std::memcpy(handle_out, &h, 64);
        *ptr = p;
        if (kind_out) *kind_out = kind;
        return FP4_OK;
    }
    set_error("fp4_hip_comm_alloc: could not allocate and export %lld bytes of device memory", (long long)bytes);
    return FP4_ERR_LAUNCH;
}

extern "C" int fp4_hip_comm_open(const uint8_t handle[64], void **ptr) {
    using namespace fp4;
    if (!handle || !ptr) {
        set_error("fp4_hip_comm_open: bad argument");
        return FP4_ERR_INVALID_ARGUMENT;
    }
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle, 64);
    const hipError_t e = hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("fp4_hip_comm_open: hipIpcOpenMemHandle failed: %s", hipGetErrorString(e));
        return FP4_ERR_LAUNCH;
    }
    return FP4_OK;
}

extern "C" int fp4_hip_comm_close(void *ptr) {
    if (ptr && hipIpcCloseMemHandle(ptr) != hipSuccess) {
        (void)hipGetLastError();
        fp4::set_error("fp4_hip_comm_close: hipIpcCloseMemHandle failed");
        return FP4_ERR_LAUNCH;
    }
    return FP4_OK;
}

extern "C" int fp4_hip_comm_free(void *ptr) {
    if (ptr && hipFree(ptr) != hipSuccess) {
        (void)hipGetLastError();
        fp4::set_error("fp4_hip_comm_free: hipFree failed");
        return FP4_ERR_LAUNCH;
    }
    return FP4_OK;
}

extern "C" int fp4_hip_comm_status(const void *own_buffer, uint32_t out4[4]) {
    if (!own_buffer || !out4 || hipMemcpy(out4, own_buffer, 16, hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        fp4::set_error("fp4_hip_comm_status: cannot read the header");
        return FP4_ERR_INVALID_ARGUMENT;
    }
    return FP4_OK;
}

extern "C" int fp4_hip_comm_clear_status(void *own_buffer) {
    // status word + timed-out lane count (bytes 8..16 of the header); epoch and done counter stay: the call sequence goes on.
    // Synchronous, like fp4_hip_comm_status: to be called at a sync point, with no reduction of this rank in flight.
    if (!own_buffer) {
        fp4::set_error("fp4_hip_comm_clear_status: null buffer");
        return FP4_ERR_INVALID_ARGUMENT;
    }
    hipError_t err = hipDeviceSynchronize();
    if (err == hipSuccess) err = hipMemset(static_cast<uint8_t *>(own_buffer) + 8, 0, 8);
    if (err == hipSuccess) err = hipDeviceSynchronize();
    if (err != hipSuccess) {  // a runtime failure, not a caller mistake
        (void)hipGetLastError();
        fp4::set_error("fp4_hip_comm_clear_status: cannot reset the header: %s", hipGetErrorString(err));
        return FP4_ERR_LAUNCH;
    }
    return FP4_OK;
}

extern "C" int fp4_hip_allreduce_oneshot(const float *partial, void *const *peer_buffers, int rank, int world, int64_t M,
                                         int64_t capacity, const void *bias, const void *residual, void *out, int out_dtype,
                                         int64_t timeout_us, void *stream) {
    using namespace fp4;
    if (world < 1 || world > kMaxRanks || rank < 0 || rank >= world || M < 0 || capacity < 1 || M > capacity || M > (int64_t(1) << 30)) {
        set_error("fp4_hip_allreduce_oneshot: rank %d of %d, M=%lld, capacity=%lld (need world <= %d, 0 <= M <= capacity)", rank, world,
                  (long long)M, (long long)capacity, kMaxRanks);
        return FP4_ERR_INVALID_ARGUMENT;
    }
    if (out_dtype != FP4_DTYPE_F16 && out_dtype != FP4_DTYPE_BF16 && out_dtype != FP4_DTYPE_F32) {
        set_error("fp4_hip_allreduce_oneshot: unsupported dtype %d", out_dtype);
        return FP4_ERR_UNSUPPORTED;
    }
    if (M == 0) return FP4_OK;
    if (!partial || !peer_buffers || !out) {
        set_error("fp4_hip_allreduce_oneshot: null pointer");
        return FP4_ERR_INVALID_ARGUMENT;
    }
    PeerTable t;
    for (int p = 0; p < kMaxRanks; ++p) t.slots[p] = nullptr;
    for (int p = 0; p < world; ++p) {
        if (!peer_buffers[p]) {
            set_error("fp4_hip_allreduce_oneshot: peer buffer %d is null", p);
            return FP4_ERR_INVALID_ARGUMENT;
        }
        t.slots[p] = reinterpret_cast<uint64_t *>(static_cast<uint8_t *>(peer_buffers[p]) + kHeaderBytes);
    }
    CommHeader *hdr = static_cast<CommHeader *>(peer_buffers[rank]);
    const uint64_t ticks = uint64_t(timeout_us > 0 ? timeout_us : 2000000) * 100u;  // s_memrealtime runs at 100 MHz
    // few, small workgroups: the step is latency, not bandwidth; both ranks' grids must fit next to whatever else runs
    const unsigned blocks = (unsigned)((M + 255) / 256 < 64 ? (M + 255) / 256 : 64);
    hipStream_t s = static_cast<hipStream_t>(stream);
#define FP4_AR(DT)                                                                                                              \
    hipLaunchKernelGGL((allreduce_oneshot_kernel<DT>), dim3(blocks), dim3(256), 0, s, partial, t, hdr, rank, world, (int)M, capacity, \
                       bias, residual, out, ticks)
    if (out_dtype == FP4_DTYPE_F16)
        FP4_AR(FP4_DTYPE_F16);
    else if (out_dtype == FP4_DTYPE_BF16)
        FP4_AR(FP4_DTYPE_BF16);
    else
        FP4_AR(FP4_DTYPE_F32);
#undef FP4_AR
    return check_launch("fp4_hip_allreduce_oneshot");
}
