// Blockwise FP4 -> f16 / bf16 / f32 dequant for gfx950 (MI355X).
//
// Replaces the reference's dequantize_blockwise_kernel_fp4 /
// dequantize_blockwise_codebook_kernel_fp4 (csrc/dequant_fp4_optimized.cu:89-171) and their
// host launchers (:173-255).  Same arithmetic -- out = RN_T(f32(code[nibble]) * absmax) --
// but a different machine mapping:
//
//  * HBM-bound streaming kernel (0.5 B in + 4/blocksize B scale + sizeof(T) B out per element),
//    so no MFMA; the job is to keep ~8 MiB of loads in flight and write full cache lines.
//  * 256-thread workgroups (4 wave64).  Every wave-instruction touches one contiguous span:
//    loads are 4 B/lane (16-bit outputs) or 2 B/lane (f32 output) so that the 8 (or 4) values a
//    lane decodes from one load are exactly one 16-byte store, and the 64 lanes of a store
//    write 1 KiB contiguous.  (The reference's CUB warp-transpose through shared memory does
//    the same reshuffle with two barriers per tile; here the lane map makes it unnecessary.)
//  * The 16-entry code LUT and the tile's absmax slice are staged in LDS once per workgroup;
//    the LUT read is a conflict-free broadcast (16 consecutive dwords), the absmax read is an
//    8-lane broadcast.  The absmax load is issued first so the barrier's vmcnt wait leaves all
//    packed loads in flight.  Large outputs use non-temporal loads and stores (streamed once).
//  * One launch covers all whole tiles; a small generic kernel covers the ragged tail, odd
//    alignments and unusual block sizes with the reference's exact absmax-index rule.
#include <atomic>

#include "fp4_common.h"

namespace fp4 {

namespace {

constexpr int kThreads = 256;

template <int DT>
struct OutCfg;
template <>
struct OutCfg<FP4_DTYPE_BF16> {
    using load_t = uint32_t;
    static constexpr int kVals = 8;
};
template <>
struct OutCfg<FP4_DTYPE_F16> {
    using load_t = uint32_t;
    static constexpr int kVals = 8;
};
template <>
struct OutCfg<FP4_DTYPE_F32> {
    using load_t = uint16_t;
    static constexpr int kVals = 4;
};

// nibble of element i (0-based within the loaded word): high nibble of byte i/2 when i is even
__device__ __forceinline__ uint32_t nibble_of(uint32_t q, int i) { return (q >> (8 * (i >> 1) + ((i & 1) ? 0 : 4))) & 15u; }

template <bool NT, typename V>
__device__ __forceinline__ void store_vec(V *p, V v) {
    if constexpr (NT)
        __builtin_nontemporal_store(v, p);
    else
        *p = v;
}

// Whole tiles only: tile = kThreads * LOADS * kVals elements, aligned so that either the tile
// is a whole number of quant blocks or lies inside one.
template <int DT, int LOADS, bool NT>
__global__ __launch_bounds__(kThreads) void dequant_tiles_kernel(const uint8_t *__restrict__ packed,
                                                                  const float *__restrict__ absmax,
                                                                  void *__restrict__ out, int bs_shift, int which_table) {
    using Cfg = OutCfg<DT>;
    using load_t = typename Cfg::load_t;
    constexpr int kVals = Cfg::kVals;
    constexpr int kTileElems = kThreads * LOADS * kVals;
    constexpr int kMaxAbs = kTileElems / 32;  // smallest fast-path blocksize is 32

    __shared__ float s_lut[16];
    __shared__ float s_absmax[kMaxAbs];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int64_t e_base = int64_t(blockIdx.x) * kTileElems;

    // 1. absmax slice of this tile -> LDS (issued first: oldest entry of the vmcnt queue)
    const int n_abs = (kTileElems >> bs_shift) > 0 ? (kTileElems >> bs_shift) : 1;
    const float *abs_src = absmax + (e_base >> bs_shift);
    float am_reg[(kMaxAbs + kThreads - 1) / kThreads];
#pragma unroll
    for (int r = 0; r < (kMaxAbs + kThreads - 1) / kThreads; ++r) {
        const int i = tid + r * kThreads;
        am_reg[r] = i < n_abs ? abs_src[i] : 0.0f;
    }

    // 2. packed nibbles: LOADS coalesced loads per lane, all in flight together
    const load_t *src = reinterpret_cast<const load_t *>(packed) + (e_base / kVals) + wave * (64 * LOADS) + lane;
    load_t q[LOADS];
#pragma unroll
    for (int j = 0; j < LOADS; ++j) q[j] = NT ? __builtin_nontemporal_load(src + j * 64) : src[j * 64];

#pragma unroll
    for (int r = 0; r < (kMaxAbs + kThreads - 1) / kThreads; ++r) {
        const int i = tid + r * kThreads;
        if (i < n_abs) s_absmax[i] = am_reg[r];
    }
    if (tid < 16) s_lut[tid] = lut_entry(which_table, tid);
    __syncthreads();

    // 3. decode + store
#pragma unroll
    for (int j = 0; j < LOADS; ++j) {
        const int word = wave * (64 * LOADS) + j * 64 + lane;  // load-word index inside the tile
        const int e_local = word * kVals;
        const float am = s_absmax[e_local >> bs_shift];
        float v[kVals];
#pragma unroll
        for (int i = 0; i < kVals; ++i) v[i] = s_lut[nibble_of(uint32_t(q[j]), i)] * am;
        if constexpr (DT == FP4_DTYPE_F32) {
            f32x4 o = {v[0], v[1], v[2], v[3]};
            store_vec<NT>(reinterpret_cast<f32x4 *>(out) + (e_base + e_local) / 4, o);
        } else {
            u32x4 o = {pack2<DT>(v[0], v[1]), pack2<DT>(v[2], v[3]), pack2<DT>(v[4], v[5]), pack2<DT>(v[6], v[7])};
            store_vec<NT>(reinterpret_cast<u32x4 *>(out) + (e_base + e_local) / 8, o);
        }
    }
}

// Anything the tile kernel does not take: ragged tails, unaligned pointers, block sizes that are
// not a power of two >= 32.  One thread per packed byte.  The absmax index is the reference's
// literal rule -- one lookup per 8-byte thread group, taken at the group's first byte
// (csrc/dequant_fp4_optimized.cu:110,159,177) -- which equals e / blocksize when blocksize % 16 == 0.
template <int DT>
__global__ __launch_bounds__(kThreads) void dequant_generic_kernel(const uint8_t *__restrict__ packed,
                                                                    const float *__restrict__ absmax,
                                                                    void *__restrict__ out, int blocksize,
                                                                    int64_t e_start, int64_t n, CodeTable tbl) {
    const int64_t e = e_start + 2 * (int64_t(blockIdx.x) * kThreads + threadIdx.x);
    if (e >= n) return;
    const uint32_t b = packed[e >> 1];
    const float am = absmax[((e >> 4) << 3) / (blocksize >> 1)];
    const float v0 = __builtin_bit_cast(float, tbl.bits[b >> 4]) * am;
    const float v1 = __builtin_bit_cast(float, tbl.bits[b & 15u]) * am;
    if constexpr (DT == FP4_DTYPE_F32) {
        float *o = reinterpret_cast<float *>(out);
        o[e] = v0;
        if (e + 1 < n) o[e + 1] = v1;
    } else {
        uint16_t *o = reinterpret_cast<uint16_t *>(out);
        o[e] = from_f32<DT>(v0);
        if (e + 1 < n) o[e + 1] = from_f32<DT>(v1);
    }
}

std::atomic<int> g_dequant_variant{-1};  // sweep hook: LOADS | NT << 8, or -1 = heuristic (relaxed atomic, one snapshot per call)

template <int DT, int LOADS, bool NT>
void launch_tiles(const uint8_t *packed, const float *absmax, void *out, int bs_shift, int64_t tiles, int which_table,
                  hipStream_t stream) {
    hipLaunchKernelGGL((dequant_tiles_kernel<DT, LOADS, NT>), dim3((unsigned)tiles), dim3(kThreads), 0, stream, packed,
                       absmax, out, bs_shift, which_table);
}

template <int DT>
int64_t run_tiles(const uint8_t *packed, const float *absmax, void *out, int bs_shift, int64_t n, int which_table,
                  int flags, hipStream_t stream) {
    constexpr int kVals = OutCfg<DT>::kVals;
    int loads;
    bool nt;
    const int gv = g_dequant_variant.load(std::memory_order_relaxed);
    if (gv >= 0) {
        loads = gv & 0xFF;
        nt = (gv >> 8) & 1;
    } else {
        // Measured on MI355X at 4096x4096 (profiles/r01_*): 4 loads per lane (2048 workgroups = 8 per CU, one
        // resident round) with non-temporal loads AND stores is the fastest 16-bit geometry (7.5 us vs 9.7 us
        // with plain stores: the 32 MiB of output otherwise sits dirty in L2 until the end-of-kernel write-back);
        // f32 output wants 16 two-byte loads per lane.  Smaller problems shrink the tile to keep >= 1024 workgroups.
        const int64_t per_load = int64_t(kThreads) * kVals;
        loads = DT == FP4_DTYPE_F32 ? 16 : 4;
        while (loads > 1 && n / (per_load * loads) < 1024) loads >>= 1;
        nt = flags == FP4_DEQUANT_STREAM || (flags == FP4_DEQUANT_AUTO && n >= (int64_t(1) << 22));
        // a consumer that reads the weight back at once (dequant + GEMM) is better served by plain stores:
        // the output stays in L2 / Infinity Cache (measured: -2 us per 4096x4096 layer end to end)
        if (flags == FP4_DEQUANT_KEEP_CACHED && DT != FP4_DTYPE_F32 && n / (per_load * 8) >= 1024) loads = 8;
    }
    const int64_t tile = int64_t(kThreads) * loads * kVals;
    const int64_t tiles = n / tile;
    if (tiles == 0) return 0;
    // Built: f32 output - 1..16 loads per lane, streaming or plain stores; 16-bit output - 1, 2, 4 loads either way and 8 loads with
    // plain stores (KEEP_CACHED).  (Round 3 removed the 16-bit 8-load streaming and 16-load geometries: sweep-only, behind 4 loads
    // at every size, profiles/r01_a_sweep_4096x4096.txt.)
#define FP4_CASE(L)                                                                                   \
    case L:                                                                                           \
        if constexpr (DT == FP4_DTYPE_F32 || (L) <= 4) {                                              \
            if (nt)                                                                                   \
                launch_tiles<DT, L, true>(packed, absmax, out, bs_shift, tiles, which_table, stream);  \
            else                                                                                      \
                launch_tiles<DT, L, false>(packed, absmax, out, bs_shift, tiles, which_table, stream); \
        } else if constexpr ((L) == 8) {                                                              \
            if (nt) return -1;                                                                        \
            launch_tiles<DT, L, false>(packed, absmax, out, bs_shift, tiles, which_table, stream);     \
        } else {                                                                                      \
            return -1;                                                                                \
        }                                                                                             \
        break;
    switch (loads) {
        FP4_CASE(1)
        FP4_CASE(2)
        FP4_CASE(4)
        FP4_CASE(8)
        FP4_CASE(16)
        default:
            return -1;
    }
#undef FP4_CASE
    return tiles * tile;
}

template <int DT>
int run(const uint8_t *packed, const float *absmax, void *out, int blocksize, int64_t n, int which_table, int flags,
        hipStream_t stream) {
    const CodeTable tbl = make_table(which_table);
    const int bs_shift = ilog2_exact(blocksize);
    const uintptr_t align = reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(out);
    int64_t done = 0;
    if (bs_shift >= 5 && (align & 15u) == 0) {
        done = run_tiles<DT>(packed, absmax, out, bs_shift, n, which_table, flags, stream);
        if (done < 0) {
            set_error("fp4_hip_dequantize_blockwise: unknown kernel variant %d", g_dequant_variant.load(std::memory_order_relaxed));
            return FP4_ERR_INVALID_ARGUMENT;
        }
    }
    if (done < n) {
        const int64_t bytes = (n - done + 1) / 2;
        const int64_t blocks = (bytes + kThreads - 1) / kThreads;
        hipLaunchKernelGGL((dequant_generic_kernel<DT>), dim3((unsigned)blocks), dim3(kThreads), 0, stream, packed, absmax,
                           out, blocksize, done, n, tbl);
    }
    return check_launch("fp4_hip_dequantize_blockwise");
}

}  // namespace

void set_dequant_variant(int v) { g_dequant_variant.store(v, std::memory_order_relaxed); }

}  // namespace fp4

extern "C" int fp4_hip_dequantize_blockwise(const uint8_t *packed, const float *absmax, void *out, int blocksize, int64_t n,
                                            int out_dtype, int table, int flags, void *stream) {
    using namespace fp4;
    if (n < 0 || blocksize < 2 || (blocksize & 1)) {
        set_error("fp4_hip_dequantize_blockwise: n=%lld blocksize=%d (need n >= 0, even blocksize >= 2)", (long long)n,
                  blocksize);
        return FP4_ERR_INVALID_ARGUMENT;
    }
    if (table != FP4_TABLE_CODEBOOK && table != FP4_TABLE_TREE) {
        set_error("fp4_hip_dequantize_blockwise: unknown table %d", table);
        return FP4_ERR_INVALID_ARGUMENT;
    }
    if (flags != FP4_DEQUANT_AUTO && flags != FP4_DEQUANT_KEEP_CACHED && flags != FP4_DEQUANT_STREAM) {
        set_error("fp4_hip_dequantize_blockwise: unknown flags %d", flags);
        return FP4_ERR_INVALID_ARGUMENT;
    }
    if (n == 0) return FP4_OK;
    if (!packed || !absmax || !out) {
        set_error("fp4_hip_dequantize_blockwise: null pointer");
        return FP4_ERR_INVALID_ARGUMENT;
    }
    if (n > (int64_t(1) << 40)) {
        set_error("fp4_hip_dequantize_blockwise: n=%lld too large", (long long)n);
        return FP4_ERR_UNSUPPORTED;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (out_dtype) {
        case FP4_DTYPE_F16:
            return run<FP4_DTYPE_F16>(packed, absmax, out, blocksize, n, table, flags, s);
        case FP4_DTYPE_BF16:
            return run<FP4_DTYPE_BF16>(packed, absmax, out, blocksize, n, table, flags, s);
        case FP4_DTYPE_F32:
            return run<FP4_DTYPE_F32>(packed, absmax, out, blocksize, n, table, flags, s);
        default:
            // the reference prints "NO APPLICABLE DTYPE!" and returns garbage
            // (csrc/dequant_fp4_optimized.cu:201-203,250-252); here it is an error
            set_error("fp4_hip_dequantize_blockwise: unsupported output dtype %d", out_dtype);
            return FP4_ERR_UNSUPPORTED;
    }
}
