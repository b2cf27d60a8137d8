// Few activation rows x LONG weight rows on SHORT weights (the down projection of every decoder: 4096 x 14336, 4096 x 11008,
// 5120 x 13824) - x-stationary split-K over workgroups, with a caller-provided workspace.
// The one-pass kernels of gemm_wide_fp4.hip give every 16-row workgroup all of x: 256 workgroups x (rows x K x 2 B) through the L2s,
// 147 MB per launch at 16 rows x K = 14336 - the aggregate L2 read bandwidth (13-14 TB/s) is their wall, four times the weight's own
// bytes.  Here a workgroup owns a K SLICE of 512 columns: its slice of x (rows x 1 KB) is fetched ONCE into LDS and stays, and the
// workgroup streams that slice of many weight rows past it (every wave its own 16-row tiles, wave-private rings, no barrier after
// the first).  x traffic drops to (workgroups x rows x 1 KB), a few MB; the price is one float per (K slice, activation row, weight
// row) through a workspace and a second, tiny launch that adds the slices in a fixed order and applies the epilogue - deterministic,
// no atomics, no cross-workgroup synchronisation (stream order does it).
#include "gemv_common.h"

namespace fp4 {

namespace {

typedef __bf16 bf16x8s_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8s_t __attribute__((ext_vector_type(8)));

template <int DT>
__device__ __forceinline__ f32x4 mfma_xw_s(u32x4 xfrag, u32x4 wfrag, f32x4 c) {
    if constexpr (DT == FP4_DTYPE_F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8s_t, xfrag), __builtin_bit_cast(f16x8s_t, wfrag), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8s_t, xfrag), __builtin_bit_cast(bf16x8s_t, wfrag), c, 0, 0, 0);
}

template <int DT>
__device__ __forceinline__ u32x4 decode8_nat(uint32_t q) {  // (e0,e1) (e2,e3) (e4,e5) (e6,e7), as in gemm_wide_fp4.hip
    uint32_t P[4];
    decode8<DT>(q, P);
    u32x4 n;
    n.x = perm(P[2], P[0], 0x05040100u);
    n.y = perm(P[2], P[0], 0x07060302u);
    n.z = perm(P[3], P[1], 0x05040100u);
    n.w = perm(P[3], P[1], 0x07060302u);
    return n;
}

__device__ __forceinline__ void dma16(const uint8_t *src, uint8_t *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// grid (K slices, row chunks); a K slice is 8 quant blocks (512 columns; the last may be shorter); `rows_per_wg` % 128 == 0.
// partial[(slice * B + n) * M + row] = sum over the slice's columns, already times 1/12.
template <int DT, int NT>
__global__ __launch_bounds__(512) void gemm16_xstat_kernel(const uint16_t *__restrict__ x, const uint8_t *__restrict__ W,
                                                           const float *__restrict__ absmax, float *__restrict__ partial, int B, int M, int K,
                                                           int rows_per_wg) {
    constexpr int D = NT <= 3 ? 3 : 2;                    // weight ring depth per wave (LDS: NT*16 KB of x + 8 * D * 4.5 KB)
    constexpr int kXBytes = 8 * NT * 2048, kTile = 16 * 256 + 16 * 32, kWave = D * kTile;
    static_assert(kXBytes + 8 * kWave <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(1024))) uint8_t s_raw[kXBytes + 8 * kWave];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, kb = lane >> 4;
    const int nblk = K >> 6, jb0 = blockIdx.x * 8;  // (K slice = the fast grid index: neighbours in time read neighbouring bytes of the same rows)
    const int nb = nblk - jb0 < 8 ? nblk - jb0 : 8;      // quant blocks of this slice (>= 1)
    const int row_base = blockIdx.y * rows_per_wg;
    const int tiles = rows_per_wg >> 4;
    uint8_t *mine = s_raw + kXBytes + wave * kWave;

    // x slice: wave w fetches block w (if the slice has it): 2*NT DMAs of 8 columns x 128 B, swizzled as in gemm_wide_fp4.hip
    if (wave < nb) {
        const uint8_t *xb = reinterpret_cast<const uint8_t *>(x);
#pragma unroll
        for (int d = 0; d < 2 * NT; ++d) {
            const int n = 8 * d + (lane >> 3), sl = lane & 7;
            const int nn = n < B ? n : B - 1;
            dma16(xb + (uint32_t)nn * (uint32_t)K * 2u + (uint32_t)((sl ^ ((n >> 1) & 7)) * 16) + (uint32_t)(jb0 + wave) * 128u,
                  s_raw + wave * (NT * 2048) + d * 1024);
        }
    }
    // weight tile t of this wave = tile (wave + 8 t) of the workgroup: 4 DMAs of 4 rows x 256 B (piece ^ row), 2 of 8 rows x 8 scales
    const int wrl = lane >> 4, wp = lane & 15, srl = lane >> 3, sc = lane & 7;
    auto issue = [&](int t) {
        const int r0 = row_base + (wave + 8 * t) * 16;
        uint8_t *slot = mine + (t % D) * kTile;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rl = 4 * q + wrl, r = r0 + rl;
            const int p = wp ^ rl;
            dma16(W + (int64_t)(r < M ? r : M - 1) * (int64_t)(K >> 1) + jb0 * 32 + ((p >> 1) < nb ? p * 16 : (p & 1) * 16), slot + q * 1024);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r = r0 + 8 * q + srl;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(absmax + (int64_t)(r < M ? r : M - 1) * nblk + jb0 + (sc < nb ? sc : 0)),
                (__attribute__((address_space(3))) void *)(slot + 4096 + q * 256), 4, 0, 0);
        }
    };
    const int my_tiles = tiles > wave ? (tiles - wave + 7) >> 3 : 0;
    for (int t = 0; t < D - 1 && t < my_tiles; ++t) issue(t);
    wait_vm<0>();
    __syncthreads();  // the x slice is in LDS for everyone; from here on every wave runs on its own

    const int xrd0 = i * 128 + (((2 * kb) ^ (i >> 1)) * 16), xrd1 = i * 128 + (((2 * kb + 1) ^ (i >> 1)) * 16);
    for (int t = 0; t < my_tiles; ++t) {
        if (D > 2 && t + D - 2 < my_tiles)
            wait_vm<(D - 2) * 6>();
        else
            wait_vm<0>();
        asm volatile("" ::: "memory");
        if (t + D - 1 < my_tiles) issue(t + D - 1);  // into the slot tile t - 1 used
        const uint8_t *slot = mine + (t % D) * kTile;
        f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        auto block = [&](int b) {
            const u32x2 wq = *reinterpret_cast<const u32x2 *>(slot + i * 256 + (((2 * b + (kb >> 1)) ^ i) * 16) + (kb & 1) * 8);
            const float am = *reinterpret_cast<const float *>(slot + 4096 + i * 32 + b * 4);
            const uint8_t *xs = s_raw + b * (NT * 2048);
            f32x4 tile[NT];
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const u32x4 wf = decode8_nat<DT>(t2 == 0 ? wq.x : wq.y);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const u32x4 xf = *reinterpret_cast<const u32x4 *>(xs + nt * 2048 + (t2 == 0 ? xrd0 : xrd1));
                    tile[nt] = mfma_xw_s<DT>(xf, wf, t2 == 0 ? f32x4{0.0f, 0.0f, 0.0f, 0.0f} : tile[nt]);
                }
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                acc[nt].x = __builtin_fmaf(tile[nt].x, am, acc[nt].x);
                acc[nt].y = __builtin_fmaf(tile[nt].y, am, acc[nt].y);
                acc[nt].z = __builtin_fmaf(tile[nt].z, am, acc[nt].z);
                acc[nt].w = __builtin_fmaf(tile[nt].w, am, acc[nt].w);
            }
        };
        if (nb == 8) {  // the usual case, unrolled: eight independent ds_read -> decode -> MFMA chains for the scheduler to interleave
#pragma unroll
            for (int b = 0; b < 8; ++b) block(b);
        } else {
            for (int b = 0; b < nb; ++b) block(b);
        }
        // D layout: lane (j = i -> weight row of the tile, kb) register g -> activation row nt*16 + kb*4 + g
        const int row = row_base + (wave + 8 * t) * 16 + i;
        if (row < M) {
            float *dst = partial + (int64_t)blockIdx.x * B * M + row;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = nt * 16 + kb * 4;
                if (n + 0 < B) dst[(int64_t)(n + 0) * M] = acc[nt].x * (1.0f / 12.0f);
                if (n + 1 < B) dst[(int64_t)(n + 1) * M] = acc[nt].y * (1.0f / 12.0f);
                if (n + 2 < B) dst[(int64_t)(n + 2) * M] = acc[nt].z * (1.0f / 12.0f);
                if (n + 3 < B) dst[(int64_t)(n + 3) * M] = acc[nt].w * (1.0f / 12.0f);
            }
        }
    }
}

// out[n][row] = epilogue( sum over slices, in slice order, of partial[slice][n][row] ); one thread per output (pairs: per pair)
template <int DT>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float *__restrict__ partial, const uint16_t *__restrict__ bias,
                                                            const uint16_t *residual, uint16_t *out, int B, int M, int slices, int mode) {
    const bool pairs = (mode & kModeSiluMulPairs) != 0;
    const int width = pairs ? M >> 1 : M;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)B * width) return;
    const int n = (int)(idx / width), c = (int)(idx % width);
    const int64_t stride = (int64_t)B * M;
    // eight slices' loads are issued together, then added in slice order (a load per addition would serialise on memory latency)
    if (pairs) {
        const float *src = partial + (int64_t)n * M + 2 * c;
        float g = 0.0f, u = 0.0f;
        for (int s0 = 0; s0 < slices; s0 += 8) {
            f32x2 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = s0 + j < slices ? *reinterpret_cast<const f32x2 *>(src + (s0 + j) * stride) : f32x2{0.0f, 0.0f};
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (s0 + j < slices) g += v[j].x, u += v[j].y;
        }
        store_small_silu_mul<DT>(out, bias, residual, n, c, width, g, u);
    } else {
        const float *src = partial + (int64_t)n * M + c;
        float t = 0.0f;
        for (int s0 = 0; s0 < slices; s0 += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = s0 + j < slices ? src[(s0 + j) * stride] : 0.0f;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (s0 + j < slices) t += v[j];
        }
        store_small<DT>(out, bias, residual, n, c, M, t);
    }
}

template <int DT, int NT>
void launch_xstat(const void *x, const uint8_t *W, const float *absmax, float *partial, int B, int M, int K, int rows_per_wg, int slices,
                  hipStream_t stream) {
    const dim3 grid((unsigned)slices, (unsigned)((M + rows_per_wg - 1) / rows_per_wg));
    hipLaunchKernelGGL((gemm16_xstat_kernel<DT, NT>), grid, dim3(512), 0, stream, reinterpret_cast<const uint16_t *>(x), W, absmax, partial,
                       B, M, K, rows_per_wg);
}

template <int DT>
void launch_splitk(const void *x, const uint8_t *W, const float *absmax, const void *bias, const void *residual, void *out, float *ws,
                   int B, int M, int K, int mode, int rows_per_wg, int slices, hipStream_t stream) {
    const int nt = (B + 15) / 16;
    if (nt == 1)
        launch_xstat<DT, 1>(x, W, absmax, ws, B, M, K, rows_per_wg, slices, stream);
    else if (nt == 2)
        launch_xstat<DT, 2>(x, W, absmax, ws, B, M, K, rows_per_wg, slices, stream);
    else if (nt == 3)
        launch_xstat<DT, 3>(x, W, absmax, ws, B, M, K, rows_per_wg, slices, stream);
    else
        launch_xstat<DT, 4>(x, W, absmax, ws, B, M, K, rows_per_wg, slices, stream);
    const int width = (mode & kModeSiluMulPairs) ? M / 2 : M;
    const int64_t outs = (int64_t)B * width;
    hipLaunchKernelGGL((splitk_reduce_kernel<DT>), dim3((unsigned)((outs + 255) / 256)), dim3(256), 0, stream, ws,
                       reinterpret_cast<const uint16_t *>(bias), reinterpret_cast<const uint16_t *>(residual),
                       reinterpret_cast<uint16_t *>(out), B, M, slices, mode);
}

}  // namespace

// Workspace the split-K path wants for this shape, or 0 where it is not the faster path (then no workspace is needed).
// Measured (profiles/r02_wide_batch_17_to_128_rows.txt): two launches and the x slice's fetch cost ~16 us before the first row, so it
// pays from 33 activation rows on, on short weights (M below 24 rows per CU) with rows of 8192 columns and more
// (4096 x 14336 x 64 rows: 37.2 -> 27.6 us, 5120 x 13824 x 64 rows: 65 -> 40 us); below that the one-pass kernels are level or ahead.
int64_t gemm_splitk_workspace_bytes(int64_t B, int64_t M, int64_t K, int blocksize, int dtype) {
    if (B < 33 || B > 64 || blocksize != 64 || (K % 64) != 0 || K < 8192 || (dtype != FP4_DTYPE_F16 && dtype != FP4_DTYPE_BF16)) return 0;
    if (M < 16 || M >= 24 * int64_t(device_cu_count())) return 0;
    if (B * K * 2 >= (int64_t(1) << 32)) return 0;
    const int64_t slices = (K / 64 + 7) / 8;
    return slices * B * M * 4;
}

// FP4_OK after both launches; -1 if the shape is not one for this path or the workspace is too small.
int gemm_splitk_launch(int dtype, const void *x, const uint8_t *W, const float *absmax, const void *bias, const void *residual, void *out,
                       int B, int M, int K, int mode, void *workspace, int64_t workspace_bytes, hipStream_t stream) {
    const int64_t need = gemm_splitk_workspace_bytes(B, M, K, 64, dtype);
    if (need == 0 || !workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 15u)) return -1;
    const int slices = (K / 64 + 7) / 8;
    // every wave gets the same number of 16-row tiles, and slices x row chunks is ONE round of workgroups (one per CU: the x slice's
    // fetch and the first barrier are paid once per workgroup)
    const int64_t tiles = (int64_t)((M + 15) / 16) * slices, waves = 8 * int64_t(device_cu_count());
    int tpw = (int)((tiles + waves - 1) / waves);
    if (tpw < 1) tpw = 1;
    const int rows_per_wg = 128 * tpw;
    float *ws = static_cast<float *>(workspace);
    if (dtype == FP4_DTYPE_F16)
        launch_splitk<FP4_DTYPE_F16>(x, W, absmax, bias, residual, out, ws, B, M, K, mode, rows_per_wg, slices, stream);
    else
        launch_splitk<FP4_DTYPE_BF16>(x, W, absmax, bias, residual, out, ws, B, M, K, mode, rows_per_wg, slices, stream);
    return FP4_OK;
}

}  // namespace fp4
