// Blockwise FP4 quantiser for gfx950 -- the producer side of the hot path.
//
// In the reference this step is bitsandbytes' (Params4bit.cuda() / BF.quantize_fp4, call sites
// torch_bnb_fp4/__init__.py:736-747,775-777,859-861); bitsandbytes is not part of the reference
// checkout, so the algorithm restated here is the published one (see oracle/fp4_oracle.py):
//   absmax = max|w| over the block;  x = w * (1/absmax) in f32;
//   magnitude rank = #{t in thresholds : |x| > t};  code = rankmap[rank] | (x < 0 ? 8 : 0);
//   even element -> high nibble.
// HBM-bound (2-4 B read, 0.5625 B written per element).  Each lane owns 8 consecutive elements
// (one 16-byte load for 16-bit inputs) and emits one packed dword, so loads and stores are both
// fully coalesced; the block maximum is a butterfly over the bs/8 lanes that share a block
// (cross-wave through LDS only for blocksize > 512).
#include "fp4_common.h"

namespace fp4 {

namespace {

constexpr int kQThreads = 512;  // 4096 elements per workgroup = the largest supported blocksize

__device__ __forceinline__ uint32_t quantize_one(float x) {
    const float a = __builtin_fabsf(x);
    // midpoints between neighbouring magnitudes of {0, 1/192, 1/6, 1/4, 1/3, 1/2, 2/3, 1}; strict '>'
    int rank = (a > 0.00260417f) + (a > 0.0859375f) + (a > 0.20833333f) + (a > 0.29166667f) + (a > 0.4166667f) +
               (a > 0.583333f) + (a > 0.8333333f);
    // rank -> 3-bit code {0,1,6,7,4,5,2,3}, one nibble each
    const uint32_t code = (0x32547610u >> (4 * rank)) & 7u;
    return code | (x < 0.0f ? 8u : 0u);
}

template <int DT>
__device__ __forceinline__ void load8(const void *w, int64_t e0, int64_t n, float (&v)[8]) {
    if (e0 + 8 <= n) {
        if constexpr (DT == FP4_DTYPE_F32) {
            const f32x4 a = reinterpret_cast<const f32x4 *>(w)[e0 / 4];
            const f32x4 b = reinterpret_cast<const f32x4 *>(w)[e0 / 4 + 1];
            v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
        } else {
            const u32x4 a = reinterpret_cast<const u32x4 *>(w)[e0 / 8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[2 * i] = to_f32<DT>(uint16_t(a[i] & 0xFFFFu));
                v[2 * i + 1] = to_f32<DT>(uint16_t(a[i] >> 16));
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float t = 0.0f;
            if (e0 + i < n) {
                if constexpr (DT == FP4_DTYPE_F32)
                    t = reinterpret_cast<const float *>(w)[e0 + i];
                else
                    t = to_f32<DT>(reinterpret_cast<const uint16_t *>(w)[e0 + i]);
            }
            v[i] = t;
        }
    }
}

template <int DT>
__global__ __launch_bounds__(kQThreads) void quantize_kernel(const void *__restrict__ w, uint8_t *__restrict__ packed,
                                                             float *__restrict__ absmax, int64_t n, int bs_shift) {
    __shared__ float s_wave_max[kQThreads / 64];
    const int tid = threadIdx.x;
    const int64_t e0 = (int64_t(blockIdx.x) * kQThreads + tid) * 8;
    float v[8];
    load8<DT>(w, e0, n, v);

    float m = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) m = __builtin_fmaxf(m, __builtin_fabsf(v[i]));
    const int lanes_per_block = 1 << (bs_shift - 3);
    for (int mask = 1; mask < lanes_per_block && mask < 64; mask <<= 1) m = __builtin_fmaxf(m, __shfl_xor(m, mask));
    if (lanes_per_block > 64) {  // uniform across the grid
        if ((tid & 63) == 0) s_wave_max[tid >> 6] = m;
        __syncthreads();
        const int waves_per_block = lanes_per_block >> 6;
        const int first = ((tid >> 6) / waves_per_block) * waves_per_block;
        for (int i = 0; i < waves_per_block; ++i) m = __builtin_fmaxf(m, s_wave_max[first + i]);
    }
    if (e0 >= n) return;
    if ((tid & (lanes_per_block - 1)) == 0) absmax[e0 >> bs_shift] = m;

    const float inv = 1.0f / m;
    uint32_t word = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t q = quantize_one(v[i] * inv);
        word |= q << (8 * (i >> 1) + ((i & 1) ? 0 : 4));
    }
    if (e0 + 8 <= n) {
        reinterpret_cast<uint32_t *>(packed)[e0 / 8] = word;
    } else {
        const int nbytes = int((n - e0 + 1) / 2);
        for (int b = 0; b < nbytes; ++b) packed[e0 / 2 + b] = uint8_t(word >> (8 * b));
    }
}

}  // namespace
}  // namespace fp4

extern "C" int fp4_hip_quantize_blockwise(const void *w, int w_dtype, uint8_t *packed, float *absmax, int64_t n,
                                          int blocksize, void *stream) {
    using namespace fp4;
    const int bs_shift = ilog2_exact(blocksize);
    if (n < 0) {
        set_error("fp4_hip_quantize_blockwise: n=%lld", (long long)n);
        return FP4_ERR_INVALID_ARGUMENT;
    }
    if (bs_shift < 5 || bs_shift > 12) {
        set_error("fp4_hip_quantize_blockwise: blocksize %d (need a power of two in 32..4096)", blocksize);
        return FP4_ERR_UNSUPPORTED;
    }
    if (n == 0) return FP4_OK;
    if (!w || !packed || !absmax) {
        set_error("fp4_hip_quantize_blockwise: null pointer");
        return FP4_ERR_INVALID_ARGUMENT;
    }
    if ((reinterpret_cast<uintptr_t>(w) & 15u) || (reinterpret_cast<uintptr_t>(packed) & 3u)) {
        set_error("fp4_hip_quantize_blockwise: w must be 16-byte and packed 4-byte aligned");
        return FP4_ERR_UNSUPPORTED;
    }
    const int64_t per_wg = int64_t(kQThreads) * 8;
    const unsigned blocks = (unsigned)((n + per_wg - 1) / per_wg);
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (w_dtype) {
        case FP4_DTYPE_F16:
            hipLaunchKernelGGL((quantize_kernel<FP4_DTYPE_F16>), dim3(blocks), dim3(kQThreads), 0, s, w, packed, absmax, n,
                               bs_shift);
            break;
        case FP4_DTYPE_BF16:
            hipLaunchKernelGGL((quantize_kernel<FP4_DTYPE_BF16>), dim3(blocks), dim3(kQThreads), 0, s, w, packed, absmax, n,
                               bs_shift);
            break;
        case FP4_DTYPE_F32:
            hipLaunchKernelGGL((quantize_kernel<FP4_DTYPE_F32>), dim3(blocks), dim3(kQThreads), 0, s, w, packed, absmax, n,
                               bs_shift);
            break;
        default:
            set_error("fp4_hip_quantize_blockwise: unsupported dtype %d", w_dtype);
            return FP4_ERR_UNSUPPORTED;
    }
    return check_launch("fp4_hip_quantize_blockwise");
}
