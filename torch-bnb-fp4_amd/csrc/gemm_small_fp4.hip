// Small-batch companion of the fused FP4 GEMV for gfx950 (MI355X): 2..64 activation rows against one FP4 weight,
//     out[b][r] = T( sum_k x[b][k] * code[nib(r,k)] * absmax[(r*K+k)/bs] + bias[r] )
// The reference sends every batch > 1 through a full dequant + dense GEMM (torch_bnb_fp4/__init__.py:423-436,616-617);
// here the weight stream is decoded once and used for all rows, so traffic stays at the GEMV's.
//  * gemm16_small_kernel  - the register-x GEMV geometry with 2..8 activation rows on the VALU (fallback).
//  * gemm16_mfma_kernel / gemm16_mfma_persist_kernel - 2..16 activation rows on the matrix cores
//                           (v_mfma_f32_16x16x32): one-shot for any K % 512 == 0, persistent for K = 4096.
#include <atomic>

#include "gemv_common.h"

namespace fp4 {

namespace {

// sweep hooks (fp4_hip_set_variant("gemm_small", ...)); relaxed atomics: read on every launch, flipped only by sweeps
std::atomic<int> g_small_variant{-1};  // -1 heuristic, 0 VALU kernel, 1 matrix-core kernel

// ---- small batch (2..8 activation rows): the same register-x geometry with NB x-slices per lane ----------
// The reference sends every batch > 1 through a full dequant (43 MB written and re-read at 4096x4096) plus a dense
// GEMM (torch_bnb_fp4/__init__.py:616-617).  For a handful of rows the weight stream can instead be decoded once
// per nibble and dotted against NB activation rows: traffic stays at the GEMV's 9.45 MB.  One rounding at the end,
// bias added in f32 before it (what F.linear does for batch > 1).
template <int DT, int KSPLIT, int G, int ITERS, int NB>
__global__ __launch_bounds__(256) void gemm16_small_kernel(const uint16_t *__restrict__ x, const uint8_t *__restrict__ W,
                                                           const float *__restrict__ absmax,
                                                           const uint16_t *__restrict__ bias, const uint16_t *residual, uint16_t *out,
                                                           int B, int M, int K, int bs_shift, int mode) {
    constexpr int RG = 4 / KSPLIT;
    constexpr int kRowsPerBlock = 2 * RG * ITERS;
    __shared__ float s_part[kRowsPerBlock][KSPLIT][NB];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int kw = wave % KSPLIT, rw = wave / KSPLIT;
    const int half = lane >> 5, l32 = lane & 31;
    const int C = K >> 5;
    const int row_base = blockIdx.x * kRowsPerBlock;

    int cidx[G];
    bool live[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int c = g * (32 * KSPLIT) + kw * 32 + l32;
        live[g] = c < C;
        cidx[g] = live[g] ? c : C - 1;
    }
    // x first (L2-resident; vector-memory results return in issue order), then the whole weight stream of this lane - the GEMV's
    // structure (gemv16_regx_kernel), including its buffer-descriptor loads: one 32-bit offset per lane instead of 64-bit address
    // arithmetic on the VALU (round 3; the dispatcher only comes here while M * K < 2^32).  What keeps this kernel behind the GEMV
    // is not arithmetic but x: every workgroup (4 weight rows) pulls NB x rows through the texture path - at two rows twice the bytes
    // of its weights - which is what the matrix-core kernels' 16-row tiles and LDS x image amortise (measured with 8 rows per
    // workgroup as well: 28672 x 4096 x 2 rows 19.4 -> 16.3 us, level with the matrix-core kernel, nowhere ahead of it).
    u32x4 xd[NB][G][4];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int64_t xrow = int64_t(b < B ? b : B - 1) * (K >> 3);  // in 16-byte pieces; rows past B are computed, not stored
#pragma unroll
        for (int g = 0; g < G; ++g) {
#pragma unroll
            for (int q = 0; q < 4; ++q) xd[b][g][q] = reinterpret_cast<const u32x4 *>(x)[xrow + cidx[g] * 4 + q];
        }
    }
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(W), 0, int((uint32_t(M) * uint32_t(C)) << 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_a =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(absmax), 0, int(((uint32_t(M) * uint32_t(C)) >> (bs_shift - 5)) << 2), 0x00020000);
    u32x4 wq[ITERS][G];
    float am[ITERS][G];
    int rowi[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int local = 2 * (it * RG + rw) + half;
        const int row = row_base + local;
        rowi[it] = local;
        const int rclamp = row < M ? row : M - 1;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const uint32_t chunk = uint32_t(rclamp) * uint32_t(C) + uint32_t(cidx[g]);
            wq[it][g] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, int(chunk << 4), 0, 2));  // aux 2 = nt
            const float a = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_a, int((chunk >> (bs_shift - 5)) << 2), 0, 0));
            am[it][g] = live[g] ? a : 0.0f;
        }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u32x4 w = xd[b][g][q];
                xd[b][g][q].x = perm(w.y, w.x, 0x05040100u);
                xd[b][g][q].y = perm(w.w, w.z, 0x05040100u);
                xd[b][g][q].z = perm(w.y, w.x, 0x07060302u);
                xd[b][g][q].w = perm(w.w, w.z, 0x07060302u);
            }
        }
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        float p[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) p[b] = 0.0f;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float s[NB][2];  // two accumulation chains per row and 32-weight chunk, as in the GEMV
#pragma unroll
            for (int b = 0; b < NB; ++b) s[b][0] = s[b][1] = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t P[4];
                decode8<DT>(wq[it][g][q], P);  // decoded once, used by every activation row
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    s[b][q & 1] = dot2<DT>(P[0], xd[b][g][q].x, s[b][q & 1]);
                    s[b][q & 1] = dot2<DT>(P[1], xd[b][g][q].y, s[b][q & 1]);
                    s[b][q & 1] = dot2<DT>(P[2], xd[b][g][q].z, s[b][q & 1]);
                    s[b][q & 1] = dot2<DT>(P[3], xd[b][g][q].w, s[b][q & 1]);
                }
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) p[b] = __builtin_fmaf(s[b][0] + s[b][1], am[it][g], p[b]);
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            float v = p[b];
            v = dpp_add<0x128>(v);
            v = dpp_add<0x124>(v);
            v = dpp_add<0x122>(v);
            v = dpp_add<0x121>(v);
            v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));
            if (l32 == 0) s_part[rowi[it]][kw][b] = v;
        }
    }
    __syncthreads();
    for (int i = tid; i < kRowsPerBlock * NB; i += 256) {
        const int r = i / NB, b = i % NB;
        float t = 0.0f;
#pragma unroll
        for (int k = 0; k < KSPLIT; ++k) t += s_part[r][k][b];
        const int row = row_base + r;
        if (row < M && b < B) {
            if (mode & kModeSiluMulPairs) {  // rows (r, r + 1) of this workgroup = (gate, up) of one pair; row_base is even
                if (r & 1) continue;
                float u = 0.0f;
#pragma unroll
                for (int k = 0; k < KSPLIT; ++k) u += s_part[r + 1][k][b];
                store_small_silu_mul<DT>(out, bias, residual, b, row >> 1, M >> 1, t * (1.0f / 12.0f), u * (1.0f / 12.0f));
            } else {
                store_small<DT>(out, bias, residual, b, row, M, t * (1.0f / 12.0f));
            }
        }
    }
}

// ---- small batch on the matrix cores (2..16 activation rows) -----------------------------------------------------
// With up to 16 activation rows the product is GEMM-shaped enough for v_mfma_f32_16x16x32: M-dim = 16 weight rows,
// N-dim = the (up to 16) activation rows, K-dim = 32 weights.  The dot work moves from the VALU to the matrix pipe, so
// the cost no longer grows with the batch (the VALU small-batch kernel above pays 4 v_dot2 per row per 8 weights);
// the VALU only decodes (decode8: the four dwords it returns ARE the A fragment, with the k order of the 8-group
// permuted the same way on the B side).
//   * workgroup = 8 waves = one 16-row tile of W; the 8 waves split K (wave w owns quant blocks w*NBW.. of every pass);
//   * lane (r = l&15, kb = l>>4) supplies, per 64-weight quant block, the 8 packed bytes [8kb, 8kb+8) of row r: two
//     MFMAs per block (k-sets {16kb + 8t + j}), so one MFMA never straddles two scales; the block's partial tile is
//     scaled by absmax[row(reg), block] and added to the f32 accumulator (4 FMAs per 2 MFMAs);
//   * the B operand is x[n = l&15][64b + 16kb + 8t + j], loaded straight from L2 into VGPRs (32 B per block);
//   * the 8 partial 16x16 tiles meet in LDS; one rounding, bias added in f32 first (F.linear semantics).
// Needs blocksize 64 and K a multiple of 512; everything else is served by the kernels above or by dequant + GEMM.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

template <int DT>
__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
    if constexpr (DT == FP4_DTYPE_F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// 8 consecutive 16-bit activations (x0..x7) -> the dword order decode8's pairs multiply with: (x0,x2) (x4,x6) (x1,x3) (x5,x7)
__device__ __forceinline__ u32x4 pair_up8(u32x4 w) {
    u32x4 o;
    o.x = perm(w.y, w.x, 0x05040100u);
    o.y = perm(w.w, w.z, 0x05040100u);
    o.z = perm(w.y, w.x, 0x07060302u);
    o.w = perm(w.w, w.z, 0x07060302u);
    return o;
}

template <int DT, int NBW, int ROWT, bool STAGE, int XS = 0>
__global__ __launch_bounds__(512) void gemm16_mfma_kernel(const uint16_t *__restrict__ x, const uint8_t *__restrict__ W,
                                                          const float *__restrict__ absmax,
                                                          const uint16_t *__restrict__ bias, const uint16_t *residual, uint16_t *out,
                                                          int B, int M, int K, int mode) {
    // ROWT 16-row tiles per workgroup share one B fragment (x slice): x is re-read by every workgroup, so taller
    // workgroups cut that L2 traffic (B*K*2 bytes each) at the price of fewer workgroups
    // (STAGE = false - direct 8-byte fragment loads - lost every sweep and is no longer instantiated since round 3; the branch stays as
    //  the record of the alternative.)
    // STAGE: the A-fragment layout wants 8 bytes per lane from 16 different rows (32-byte segments per row per
    // instruction); instead each wave pulls its 16 x (32*NBW)-byte region with row-contiguous 16-byte loads and
    // re-reads it from a wave-private LDS image (no workgroup barrier).  Each image row is followed by that row's NBW
    // block scales: the accumulator layout needs the scales of 4 rows x NBW blocks per lane, the same values in 16 lanes,
    // so they are fetched once per wave (one 4- or 8-byte load per lane) and re-read from LDS as broadcasts instead of
    // 8 x 16-byte global loads per lane and tile.  Row stride 32*NBW + 32 bytes: the 8-byte fragment reads of 16 rows
    // x 4 k-groups fall on distinct banks per half-wave.
    // XS > 0 (batch <= XS, XS in {4, 8}): the B fragment wants 32 bytes per lane from 16 activation rows, of which only
    // `batch` are real - loaded straight from global every 16-row workgroup would pull 16 / batch times its share of x
    // through the texture path (4x the weight stream at batch 4).  Instead the wave copies the XS x (64*NBW) activations of
    // its K slice into a wave-private LDS image with row-contiguous 16-byte loads and reads the fragments from there
    // (lanes of the unused columns read a real row: broadcast, results never stored).
    constexpr int kStageStride = 32 * NBW + 32;
    constexpr int kXStride = 128 * NBW + 16;
    constexpr int kWImageBytes = STAGE ? 8 * ROWT * 16 * kStageStride : 0;
    constexpr int kXImageBytes = 8 * XS * kXStride;
    constexpr int kImageBytes = kWImageBytes + kXImageBytes;
    constexpr int kPartBytes = 8 * ROWT * 256 * 4;
    static_assert(XS == 0 || (STAGE && (XS * NBW) % 8 == 0 && (XS & (XS - 1)) == 0), "x staging: whole 16-byte units per lane");
    // the cross-wave partial sums reuse the images' storage after the K loop (one barrier in between): LDS per workgroup
    // decides how many of these 8-wave workgroups a CU holds
    __shared__ __attribute__((aligned(16))) uint8_t s_raw[kImageBytes > kPartBytes ? kImageBytes : kPartBytes];
    uint8_t *s_w = s_raw;
    uint8_t *s_x = s_raw + kWImageBytes;
    float (*s_part)[ROWT][256] = reinterpret_cast<float (*)[ROWT][256]>(s_raw);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r = lane & 15, kb = lane >> 4;
    const int row0 = blockIdx.x * (16 * ROWT);
    const int nblk = K >> 6;
    const int passes = nblk / (8 * NBW);
    const int64_t n_b = r < B ? r : B - 1;  // clamped rows / batch entries are computed, never stored
    int64_t row_a[ROWT], row_d[ROWT][4];
#pragma unroll
    for (int rt = 0; rt < ROWT; ++rt) {
        row_a[rt] = row0 + 16 * rt + r < M ? row0 + 16 * rt + r : M - 1;
#pragma unroll
        for (int g = 0; g < 4; ++g) row_d[rt][g] = row0 + 16 * rt + kb * 4 + g < M ? row0 + 16 * rt + kb * 4 + g : M - 1;
    }
    const u32x4 *x4 = reinterpret_cast<const u32x4 *>(x);
    const u32x2 *W2 = reinterpret_cast<const u32x2 *>(W);

    f32x4 acc[ROWT];
#pragma unroll
    for (int rt = 0; rt < ROWT; ++rt) acc[rt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    // With up to 4 blocks per wave and pass (K = 14336 and the like: many short passes) the staged operands - weights, their
    // scales, for XS > 0 the x slice - are loaded one pass AHEAD: the registers of pass p + 1 are requested as soon as pass p's
    // have been written to the LDS images and fly while pass p is decoded and multiplied.  With 8 blocks per pass the extra live
    // registers would cost the second workgroup per CU (128 -> 140..164 VGPRs), so there each pass loads its own.
    constexpr bool kPassAhead = STAGE && NBW <= 4;
    constexpr int kXUnits = XS ? XS * NBW / 8 : 1;  // 16-byte units of the x image per lane
    constexpr int kLanesPerRow = 2 * NBW, kRowsPerInstr = 64 / kLanesPerRow, kInstr = (16 + kRowsPerInstr - 1) / kRowsPerInstr;
    // 16 rows x NBW scales over 64 lanes (with NBW < 4 the upper lanes repeat rows: same address, same value)
    constexpr int kScalesPerLane = NBW > 4 ? NBW / 4 : 1;
    constexpr int kLanesPerScaleRow = NBW / kScalesPerLane;
    const int srow = (lane / kLanesPerScaleRow) & 15, sj0 = (lane % kLanesPerScaleRow) * kScalesPerLane;
    u32x4 xstage[kXUnits];
    u32x4 wstage[ROWT][kInstr];
    float amstage[ROWT][kScalesPerLane];
    auto issue_staged = [&](int pass) {  // x first (L2), then the weight stream (HBM), then the scales; all branch-free
        const int b0 = (pass * 8 + wave) * NBW;
        if constexpr (XS > 0) {
#pragma unroll
            for (int i = 0; i < kXUnits; ++i) {
                const int u = i * 64 + lane, n = u / (8 * NBW), c16 = u % (8 * NBW);
                const int64_t nn = n < B ? n : B - 1;
                xstage[i] = x4[((nn * K + 64 * b0) >> 3) + c16];
            }
        }
#pragma unroll
        for (int rt = 0; rt < ROWT; ++rt)
#pragma unroll
            for (int i = 0; i < kInstr; ++i) {
                const int rr = i * kRowsPerInstr + lane / kLanesPerRow;  // row of the tile this lane fetches
                const int64_t row = row0 + 16 * rt + (rr & 15) < M ? row0 + 16 * rt + (rr & 15) : M - 1;
                wstage[rt][i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(W) + ((row * K) >> 5) + 2 * b0 +
                                                           (lane % kLanesPerRow));
            }
#pragma unroll
        for (int rt = 0; rt < ROWT; ++rt) {
            const int64_t row = row0 + 16 * rt + srow < M ? row0 + 16 * rt + srow : M - 1;
            const float *src = absmax + row * nblk + b0 + sj0;
#pragma unroll
            for (int i = 0; i < kScalesPerLane; ++i) amstage[rt][i] = src[i];
        }
    };
    if constexpr (kPassAhead) issue_staged(0);
    for (int p = 0; p < passes; ++p) {
        const int b0 = (p * 8 + wave) * NBW;
        if constexpr (STAGE && !kPassAhead) issue_staged(p);
        u32x4 xr[XS ? 1 : NBW][2];
        if constexpr (XS == 0) {
#pragma unroll
            for (int j = 0; j < NBW; ++j) {
                const int64_t e = n_b * K + 64 * (b0 + j) + 16 * kb;
                xr[j][0] = x4[e >> 3];
                xr[j][1] = x4[(e >> 3) + 1];
            }
        }
        u32x2 wq[ROWT][NBW];
        float am[ROWT][4][NBW];
        if constexpr (!STAGE) {
#pragma unroll
            for (int rt = 0; rt < ROWT; ++rt)
#pragma unroll
                for (int j = 0; j < NBW; ++j)
                    wq[rt][j] = __builtin_nontemporal_load(W2 + ((row_a[rt] * K) >> 4) + 4 * (b0 + j) + kb);
#pragma unroll
            for (int rt = 0; rt < ROWT; ++rt) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float *src = absmax + row_d[rt][g] * nblk + b0;
                    if constexpr (NBW % 4 == 0) {
#pragma unroll
                        for (int j = 0; j < NBW; j += 4) {
                            const f32x4 v = *reinterpret_cast<const f32x4 *>(src + j);
                            am[rt][g][j] = v.x, am[rt][g][j + 1] = v.y, am[rt][g][j + 2] = v.z, am[rt][g][j + 3] = v.w;
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < NBW; ++j) am[rt][g][j] = src[j];
                    }
                }
            }
        }
        if constexpr (STAGE) {
            if (p > 0) __builtin_amdgcn_wave_barrier();  // the previous pass's reads are done before the image is rewritten
#pragma unroll
            for (int rt = 0; rt < ROWT; ++rt) {
                uint8_t *img = s_w + (wave * ROWT + rt) * 16 * kStageStride;
#pragma unroll
                for (int i = 0; i < kInstr; ++i) {
                    const int rr = i * kRowsPerInstr + lane / kLanesPerRow;
                    if (rr < 16) *reinterpret_cast<u32x4 *>(img + rr * kStageStride + 16 * (lane % kLanesPerRow)) = wstage[rt][i];
                }
                float *tail = reinterpret_cast<float *>(img + srow * kStageStride + 32 * NBW) + sj0;
#pragma unroll
                for (int i = 0; i < kScalesPerLane; ++i) tail[i] = amstage[rt][i];
            }
            if constexpr (XS > 0) {
                uint8_t *ximg = s_x + wave * XS * kXStride;
#pragma unroll
                for (int i = 0; i < kXUnits; ++i) {
                    const int u = i * 64 + lane, n = u / (8 * NBW), c16 = u % (8 * NBW);
                    *reinterpret_cast<u32x4 *>(ximg + n * kXStride + 16 * c16) = pair_up8(xstage[i]);  // once, not per tile
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if constexpr (kPassAhead)
                if (p + 1 < passes) issue_staged(p + 1);  // uniform
#pragma unroll
            for (int rt = 0; rt < ROWT; ++rt) {
                const uint8_t *img = s_w + (wave * ROWT + rt) * 16 * kStageStride;
#pragma unroll
                for (int j = 0; j < NBW; ++j) wq[rt][j] = *reinterpret_cast<const u32x2 *>(img + r * kStageStride + 32 * j + 8 * kb);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const uint8_t *src = img + (kb * 4 + g) * kStageStride + 32 * NBW;
                    if constexpr (NBW % 4 == 0) {
#pragma unroll
                        for (int j = 0; j < NBW; j += 4) {
                            const f32x4 v = reinterpret_cast<const f32x4 *>(src)[j >> 2];
                            am[rt][g][j] = v.x, am[rt][g][j + 1] = v.y, am[rt][g][j + 2] = v.z, am[rt][g][j + 3] = v.w;
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < NBW; ++j) am[rt][g][j] = reinterpret_cast<const float *>(src)[j];
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            u32x4 bfrag[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if constexpr (XS > 0)
                    bfrag[t] = *reinterpret_cast<const u32x4 *>(s_x + wave * XS * kXStride + (r & (XS - 1)) * kXStride + 128 * j + 32 * kb + 16 * t);
                else
                    bfrag[t] = pair_up8(xr[j][t]);
            }
#pragma unroll
            for (int rt = 0; rt < ROWT; ++rt) {
                f32x4 tile = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    uint32_t P[4];
                    decode8<DT>(t == 0 ? wq[rt][j].x : wq[rt][j].y, P);
                    const u32x4 afrag = {P[0], P[1], P[2], P[3]};
                    tile = mfma16<DT>(afrag, bfrag[t], tile);
                }
                acc[rt].x = __builtin_fmaf(tile.x, am[rt][0][j], acc[rt].x);
                acc[rt].y = __builtin_fmaf(tile.y, am[rt][1][j], acc[rt].y);
                acc[rt].z = __builtin_fmaf(tile.z, am[rt][2][j], acc[rt].z);
                acc[rt].w = __builtin_fmaf(tile.w, am[rt][3][j], acc[rt].w);
            }
        }
    }
    if constexpr (STAGE) __syncthreads();  // every wave is done with its image before the partials overwrite the storage
#pragma unroll
    for (int rt = 0; rt < ROWT; ++rt) *reinterpret_cast<f32x4 *>(&s_part[wave][rt][lane * 4]) = acc[rt];
    __syncthreads();
    for (int i = tid; i < ROWT * 256; i += 512) {
        const int rt = i >> 8, e = i & 255;
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += s_part[w][rt][e];
        const int l = e >> 2, reg = e & 3;  // D layout: col = l & 15 (activation row), row = (l >> 4) * 4 + reg (weight row)
        const int n = l & 15, row = row0 + 16 * rt + (l >> 4) * 4 + reg;
        if (row < M && n < B) {
            if (mode & kModeSiluMulPairs) {  // registers (0, 1) and (2, 3) of a lane hold the (gate, up) rows of one pair
                if (reg & 1) continue;
                float u = 0.0f;
#pragma unroll
                for (int w = 0; w < 8; ++w) u += s_part[w][rt][e + 1];
                store_small_silu_mul<DT>(out, bias, residual, n, row >> 1, M >> 1, t * (1.0f / 12.0f), u * (1.0f / 12.0f));
            } else {
                store_small<DT>(out, bias, residual, n, row, M, t * (1.0f / 12.0f));
            }
        }
    }
}

// ---- the same tile arithmetic as a PERSISTENT workgroup, for K = 4096 (one pass: 8 waves x 8 blocks) -------------
// On tall weights the one-shot kernel above runs as three to four generations of 8-wave workgroups, each of which has
// nothing in flight while it stages, decodes and reduces.  Here a workgroup walks 16-row tiles with stride gridDim.x:
//   * a wave keeps the same K slice for every tile, so its XS x (512) activations are staged ONCE into its LDS image;
//   * the next tile's weights and scales (4 + 1 loads per lane) are issued right after the current tile's registers have
//     been written to the weight image, and fly while the current tile is decoded, multiplied, reduced and stored.
template <int DT, int XS>
__global__ __launch_bounds__(512) void gemm16_mfma_persist_kernel(const uint16_t *__restrict__ x, const uint8_t *__restrict__ W,
                                                                  const float *__restrict__ absmax,
                                                                  const uint16_t *__restrict__ bias, const uint16_t *residual,
                                                                  uint16_t *out, int B, int M, int K, int ntiles, int mode) {
    constexpr int NBW = 8;
    constexpr int kStageStride = 32 * NBW + 32, kXStride = 128 * NBW + 16;
    constexpr int kXUnits = XS ? XS * NBW / 8 : 1;
    __shared__ __attribute__((aligned(16))) uint8_t s_w[8 * 16 * kStageStride];
    __shared__ __attribute__((aligned(16))) uint8_t s_x[XS ? 8 * XS * kXStride : 16];
    // Above 4 rows one workgroup owns the CU anyway (LDS), so the cross-wave partials are double-buffered there: tile t writes buffer
    // t & 1, its reducers read it behind the tile's one barrier, tile t + 2 overwrites it only behind tile t + 1's barrier - one
    // workgroup barrier per tile instead of two.  Up to 4 rows a second buffer would cost the second workgroup per CU.
    constexpr int NBUF = XS == 4 ? 1 : 2;
    __shared__ __attribute__((aligned(16))) float s_part[NBUF][8][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, kb = lane >> 4;
    const int nblk = K >> 6;
    const int b0 = wave * NBW;
    int par = 0;
    const u32x4 *x4 = reinterpret_cast<const u32x4 *>(x);
    uint8_t *img = s_w + wave * 16 * kStageStride;
    uint8_t *ximg = s_x + wave * XS * kXStride;

    // x slice of this wave, once: up to 8 rows as a paired-up LDS image; above that (XS == 0) the 16 B fragments of the slice
    // stay in registers for every tile (64 VGPRs: one workgroup per CU, but no x traffic at all after the first tile)
    u32x4 bregs[XS ? 1 : NBW][2];
    if constexpr (XS == 0) {
        const int64_t n_b = r < B ? r : B - 1;
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            const int64_t e = n_b * K + 64 * (b0 + j) + 16 * kb;
            bregs[j][0] = pair_up8(x4[e >> 3]);
            bregs[j][1] = pair_up8(x4[(e >> 3) + 1]);
        }
    } else {
        u32x4 xstage[kXUnits];
#pragma unroll
        for (int i = 0; i < kXUnits; ++i) {
            const int u = i * 64 + lane, n = u / (8 * NBW), c16 = u % (8 * NBW);
            const int64_t nn = n < B ? n : B - 1;
            xstage[i] = x4[((nn * K + 64 * b0) >> 3) + c16];
        }
#pragma unroll
        for (int i = 0; i < kXUnits; ++i) {
            const int u = i * 64 + lane, n = u / (8 * NBW), c16 = u % (8 * NBW);
            *reinterpret_cast<u32x4 *>(ximg + n * kXStride + 16 * c16) = pair_up8(xstage[i]);
        }
    }

    constexpr int kLanesPerRow = 2 * NBW, kRowsPerInstr = 64 / kLanesPerRow, kInstr = 16 / kRowsPerInstr;  // 16, 4, 4
    constexpr int kScalesPerLane = NBW / 4;
    const int srow = lane >> 2, sj0 = (lane & 3) * kScalesPerLane;
    u32x4 wstage[kInstr];
    float amstage[kScalesPerLane];
    auto issue = [&](int tile) {  // branch-free: rows past M are clamped (computed, never stored)
        const int row0 = tile * 16;
#pragma unroll
        for (int i = 0; i < kInstr; ++i) {
            const int rr = i * kRowsPerInstr + lane / kLanesPerRow;
            const int64_t row = row0 + rr < M ? row0 + rr : M - 1;
            wstage[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(W) + ((row * K) >> 5) + 2 * b0 + (lane % kLanesPerRow));
        }
        const int64_t row = row0 + srow < M ? row0 + srow : M - 1;
        const float *src = absmax + row * nblk + b0 + sj0;
#pragma unroll
        for (int i = 0; i < kScalesPerLane; ++i) amstage[i] = src[i];
    };

    int tile = blockIdx.x;
    issue(tile);
    while (true) {
        __builtin_amdgcn_wave_barrier();  // this wave's reads of the previous tile's image are done
#pragma unroll
        for (int i = 0; i < kInstr; ++i) {
            const int rr = i * kRowsPerInstr + lane / kLanesPerRow;
            *reinterpret_cast<u32x4 *>(img + rr * kStageStride + 16 * (lane % kLanesPerRow)) = wstage[i];
        }
        {
            float *tail = reinterpret_cast<float *>(img + srow * kStageStride + 32 * NBW) + sj0;
#pragma unroll
            for (int i = 0; i < kScalesPerLane; ++i) tail[i] = amstage[i];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        const int next = tile + gridDim.x;
        const bool has_next = next < ntiles;  // uniform
        if (has_next) issue(next);

        u32x2 wq[NBW];
        float am[4][NBW];
#pragma unroll
        for (int j = 0; j < NBW; ++j) wq[j] = *reinterpret_cast<const u32x2 *>(img + r * kStageStride + 32 * j + 8 * kb);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 *src = reinterpret_cast<const f32x4 *>(img + (kb * 4 + g) * kStageStride + 32 * NBW);
#pragma unroll
            for (int j = 0; j < NBW; j += 4) {
                const f32x4 v = src[j >> 2];
                am[g][j] = v.x, am[g][j + 1] = v.y, am[g][j + 2] = v.z, am[g][j + 3] = v.w;
            }
        }
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            f32x4 t16 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                u32x4 bfrag;
                if constexpr (XS == 0)
                    bfrag = bregs[j][t];
                else
                    bfrag = *reinterpret_cast<const u32x4 *>(ximg + (r & (XS - 1)) * kXStride + 128 * j + 32 * kb + 16 * t);
                uint32_t P[4];
                decode8<DT>(t == 0 ? wq[j].x : wq[j].y, P);
                const u32x4 afrag = {P[0], P[1], P[2], P[3]};
                t16 = mfma16<DT>(afrag, bfrag, t16);
            }
            acc.x = __builtin_fmaf(t16.x, am[0][j], acc.x);
            acc.y = __builtin_fmaf(t16.y, am[1][j], acc.y);
            acc.z = __builtin_fmaf(t16.z, am[2][j], acc.z);
            acc.w = __builtin_fmaf(t16.w, am[3][j], acc.w);
        }
        *reinterpret_cast<f32x4 *>(&s_part[par][wave][lane * 4]) = acc;
        __syncthreads();
        if (tid < 256) {
            float t = 0.0f;
#pragma unroll
            for (int w = 0; w < 8; ++w) t += s_part[par][w][tid];
            const int l = tid >> 2, reg = tid & 3;  // D layout: col = l & 15 (activation row), row = (l >> 4) * 4 + reg
            const int n = l & 15, row = tile * 16 + (l >> 4) * 4 + reg;
            if (row < M && n < B) {
                if (mode & kModeSiluMulPairs) {
                    if (!(reg & 1)) {
                        float u = 0.0f;
#pragma unroll
                        for (int w = 0; w < 8; ++w) u += s_part[par][w][tid + 1];
                        store_small_silu_mul<DT>(out, bias, residual, n, row >> 1, M >> 1, t * (1.0f / 12.0f), u * (1.0f / 12.0f));
                    }
                } else {
                    store_small<DT>(out, bias, residual, n, row, M, t * (1.0f / 12.0f));
                }
            }
        }
        if constexpr (NBUF == 1)
            __syncthreads();  // s_part is rewritten by the next tile
        else
            par ^= 1;
        if (!has_next) break;
        tile = next;
    }
}

std::atomic<int> g_small_wide{0};     // sweep hook: 1 = try the one-pass kernels of gemm_wide_fp4.hip first, for any row count
std::atomic<int> g_mfma_rowt{-1};     // force 1 or 2 row tiles per workgroup
std::atomic<int> g_mfma_xstage{-1};   // 0 = B fragments straight from global even for batch <= 8
std::atomic<int> g_mfma_persist{-1};  // 0 = never the persistent kernel, 1 = whenever it applies

template <int DT>
int dispatch_mfma(const void *x, const uint8_t *W, const float *absmax, const void *bias, const void *residual, void *out, int B, int M,
                  int K, int mode, hipStream_t stream) {
    if (K % 512) return -1;
    const int units = K / 512;  // quant blocks per wave over the whole K
    const int v_rowt = g_mfma_rowt.load(std::memory_order_relaxed),
              v_xstage = g_mfma_xstage.load(std::memory_order_relaxed), v_persist = g_mfma_persist.load(std::memory_order_relaxed);  // sweep hooks, one snapshot per call
    // Measured (profiles/r01_f_small_batch_shapes.txt): two row tiles per workgroup (x fetched once per 32 rows) only pay
    // for more than 4 activation rows on tall weights; at 188 VGPRs they leave one 8-wave workgroup per CU.
    const int rowt = v_rowt > 0 ? v_rowt : ((B > 4 && M >= 32 * 256) ? 2 : 1);
    const unsigned blocks = (unsigned)((M + 16 * rowt - 1) / (16 * rowt));
    // x staged per wave in LDS: always for <= 4 rows (33 KB, two workgroups per CU still fit); for 5..8 rows (66 KB, one
    // workgroup per CU) only while the grid is a single round anyway
    // Round 3 (profiles/r03_small_batch_dispatch.txt): with a single tile per workgroup the persistent form has nothing to prefetch
    // and its 8-block single pass is behind the one-shot kernel's two 4-block passes with pass-ahead loads (4096 x 4096: 4.96 -> 4.56 us
    // at 2..4 rows, 14336 x 4096: 12.0 -> 10.8); from 5 rows up on more than 4096 rows it is what keeps the x fragments out of the
    // texture path (14336 x 4096 x 8 rows: 11.0 vs 13.3 us), and on very tall weights it wins at every row count.
    const bool persist_auto = B <= 4 ? M >= 16384 : M > 4096;
    if (v_persist != 0 && v_rowt <= 0 && v_xstage != 0 && K == 4096 && (v_persist == 1 || persist_auto)) {
        // Up to 4 rows two workgroups fit a CU (78 KB of LDS each), above that one (111 KB of LDS / 166 VGPRs).
        const int ntiles = (M + 15) / 16, resident = (B <= 4 ? 2 : 1) * device_cu_count();
        // measured (profiles/r01_f_small_batch_shapes.txt): never slower than the one-shot kernel, level with it while
        // every workgroup has a single tile, up to 1.6x faster on tall weights
        {
            const dim3 grid(ntiles < resident ? ntiles : resident);
            if (B <= 4)
                hipLaunchKernelGGL((gemm16_mfma_persist_kernel<DT, 4>), grid, dim3(512), 0, stream,
                                   reinterpret_cast<const uint16_t *>(x), W, absmax, reinterpret_cast<const uint16_t *>(bias),
                                   reinterpret_cast<const uint16_t *>(residual), reinterpret_cast<uint16_t *>(out), B, M, K, ntiles, mode);
            else if (B <= 8)
                hipLaunchKernelGGL((gemm16_mfma_persist_kernel<DT, 8>), grid, dim3(512), 0, stream,
                                   reinterpret_cast<const uint16_t *>(x), W, absmax, reinterpret_cast<const uint16_t *>(bias),
                                   reinterpret_cast<const uint16_t *>(residual), reinterpret_cast<uint16_t *>(out), B, M, K, ntiles, mode);
            else
                hipLaunchKernelGGL((gemm16_mfma_persist_kernel<DT, 0>), grid, dim3(512), 0, stream,
                                   reinterpret_cast<const uint16_t *>(x), W, absmax, reinterpret_cast<const uint16_t *>(bias),
                                   reinterpret_cast<const uint16_t *>(residual), reinterpret_cast<uint16_t *>(out), B, M, K, ntiles, mode);
            return FP4_OK;
        }
    }
    const bool xs4 = v_xstage != 0 && B <= 4;
    const bool xs8 = v_xstage != 0 && !xs4 && B <= 8 && (int)blocks <= device_cu_count();
    // (round 3: the direct, un-staged 8-byte weight loads - a sweep switch that never won, profiles/r01_e_small_batch_valu_vs_mfma.txt -
    //  and the <= 4-row x image under two row tiles, which needs > 4 rows to be chosen, are no longer built)
#define FP4_MF(NBW, RT)                                                                                               \
    if constexpr ((NBW) >= 4) {                                                                                       \
        if constexpr ((RT) == 1) {                                                                                    \
            if (xs4) {                                                                                                \
                hipLaunchKernelGGL((gemm16_mfma_kernel<DT, NBW, RT, true, 4>), dim3(blocks), dim3(512), 0, stream,    \
                                   reinterpret_cast<const uint16_t *>(x), W, absmax,                                  \
                                   reinterpret_cast<const uint16_t *>(bias), reinterpret_cast<const uint16_t *>(residual), reinterpret_cast<uint16_t *>(out), B, M, K, mode); \
                return FP4_OK;                                                                                        \
            }                                                                                                         \
        }                                                                                                             \
        if (xs8) {                                                                                                    \
            hipLaunchKernelGGL((gemm16_mfma_kernel<DT, NBW, RT, true, 8>), dim3(blocks), dim3(512), 0, stream,        \
                               reinterpret_cast<const uint16_t *>(x), W, absmax,                                      \
                               reinterpret_cast<const uint16_t *>(bias), reinterpret_cast<const uint16_t *>(residual), reinterpret_cast<uint16_t *>(out), B, M, K, mode); \
            return FP4_OK;                                                                                            \
        }                                                                                                             \
    }                                                                                                                 \
    hipLaunchKernelGGL((gemm16_mfma_kernel<DT, NBW, RT, true>), dim3(blocks), dim3(512), 0, stream,                   \
                       reinterpret_cast<const uint16_t *>(x), W, absmax, reinterpret_cast<const uint16_t *>(bias),    \
                       reinterpret_cast<const uint16_t *>(residual), reinterpret_cast<uint16_t *>(out), B, M, K, mode);                                                   \
    return FP4_OK
#define FP4_MF_RT(NBW)          \
    if (rowt == 2) {            \
        FP4_MF(NBW, 2);         \
    } else {                    \
        FP4_MF(NBW, 1);         \
    }
    // At most 4 blocks per wave and pass, the next pass's loads in flight (8192 x 8192 x 4 rows: 12.4 -> 10.6 us, 28672 x 8192: 36 -> 33 us,
    // profiles/r01_f_small_batch_pass_ahead.txt).  Round 3: also where 8 blocks would be the whole K in one pass (K = 4096: 5.0 -> 4.6 us
    // at 4096 rows) - the 8-block instantiations (up to 208 VGPRs) are no longer built.
    if (units % 4 == 0) { FP4_MF_RT(4) }
    if (units % 2 == 0) { FP4_MF_RT(2) }
    FP4_MF_RT(1)
#undef FP4_MF_RT
#undef FP4_MF
}

template <int DT, int KSPLIT, int G, int ITERS, int NB>
int launch_small(const void *x, const uint8_t *W, const float *absmax, const void *bias, const void *residual, void *out, int B, int M,
                 int K, int bs_shift, int mode, hipStream_t stream) {
    constexpr int rows_per_block = 2 * (4 / KSPLIT) * ITERS;
    const unsigned blocks = (unsigned)((M + rows_per_block - 1) / rows_per_block);
    hipLaunchKernelGGL((gemm16_small_kernel<DT, KSPLIT, G, ITERS, NB>), dim3(blocks), dim3(256), 0, stream,
                       reinterpret_cast<const uint16_t *>(x), W, absmax, reinterpret_cast<const uint16_t *>(bias),
                       reinterpret_cast<const uint16_t *>(residual), reinterpret_cast<uint16_t *>(out), B, M, K, bs_shift, mode);
    return FP4_OK;
}

// returns -1 when the shape is outside what the register budget covers (caller falls back to dequant + GEMM)
template <int DT>
int dispatch_small(const void *x, const uint8_t *W, const float *absmax, const void *bias, const void *residual, void *out, int B, int M,
                   int K, int bs_shift, int mode, hipStream_t stream) {
    const int C = K >> 5;
    const int nb = B <= 2 ? 2 : (B <= 4 ? 4 : 8);
    if (int64_t(M) * K >= (int64_t(1) << 32)) return -1;  // 32-bit buffer offsets
#define FP4_SM(KS, GG, NBB) return launch_small<DT, KS, GG, 2, NBB>(x, W, absmax, bias, residual, out, B, M, K, bs_shift, mode, stream)
#define FP4_SM_NB(KS, GG)                                                                                                   \
    switch (nb) {                                                                                                           \
        case 2: FP4_SM(KS, GG, 2);                                                                                          \
        case 4: FP4_SM(KS, GG, 4);                                                                                          \
        default: FP4_SM(KS, GG, 8);                                                                                         \
    }
    if (C <= 32) {
        FP4_SM_NB(1, 1)
    } else if (C <= 64) {
        FP4_SM_NB(2, 1)
    } else if (C <= 128) {
        FP4_SM_NB(4, 1)
    } else if (C <= 256 && nb <= 4) {
        switch (nb) {
            case 2: FP4_SM(4, 2, 2);
            default: FP4_SM(4, 2, 4);
        }
    } else if (C <= 512 && nb <= 2) {
        FP4_SM(4, 4, 2);
    }
#undef FP4_SM_NB
#undef FP4_SM
    return -1;
}

}  // namespace

void set_small_variant(int v) {
    g_small_variant = v < 0 ? -1 : (v & 1);
    const int rowt = v < 0 ? 0 : ((v >> 4) & 3);  // bits 4-5: row tiles per workgroup of the matrix-core kernel (0 = auto)
    g_mfma_rowt = rowt == 0 ? -1 : rowt;
    g_mfma_xstage = v < 0 ? -1 : ((v >> 9) & 1 ? 0 : 1);  // bit 9: B fragments straight from global
    g_small_wide = v < 0 ? 0 : ((v >> 12) & 1);            // bit 12: the one-pass (wide) kernels first
    g_mfma_persist = v < 0 ? -1 : ((v >> 10) & 3) == 1 ? 0 : (((v >> 10) & 3) == 2 ? 1 : -1);  // bits 10-11: 1 = off, 2 = force
}

}  // namespace fp4

namespace fp4 {
int64_t gemm_splitk_workspace_bytes(int64_t B, int64_t M, int64_t K, int blocksize, int dtype);  // gemm_splitk_fp4.hip
int gemm_splitk_launch(int dtype, const void *x, const uint8_t *W, const float *absmax, const void *bias, const void *residual, void *out,
                       int B, int M, int K, int mode, void *workspace, int64_t workspace_bytes, hipStream_t stream);
int gemm_wide_launch(int dtype, const void *x, const uint8_t *W, const float *absmax, const void *bias, const void *residual, void *out,
                     int B, int M, int K, int mode, bool any_rows, hipStream_t stream);  // gemm_wide_fp4.hip
namespace {
int gemm_small_entry(const void *x, const uint8_t *packed, const float *absmax, const void *bias, const void *residual, void *out,
                     int64_t B, int64_t M, int64_t K, int blocksize, int dtype, int mode, void *stream) {
    if (B < 1 || B > 128 || M < 0 || K <= 0) {
        set_error("fp4_hip_gemm_small: B=%lld M=%lld K=%lld (need 1 <= B <= 128)", (long long)B, (long long)M, (long long)K);
        return FP4_ERR_INVALID_ARGUMENT;
    }
    if ((mode & kModeSiluMulPairs) && (M & 1)) {
        set_error("fp4_hip_gemm_small_fused: the gate|up epilogue needs an even row count, got M=%lld", (long long)M);
        return FP4_ERR_INVALID_ARGUMENT;
    }
    const int64_t M_out = (mode & kModeSiluMulPairs) ? M / 2 : M;
    if (dtype == FP4_DTYPE_F32) {
        // f32 activations (the reference's sanity harness; models run 16-bit): no matrix-core kernel and none needed - up to 8 rows run
        // as B launches of the f32 GEMV, each streaming the weight once (B x 9.45 MB against the 76 MB a dequantised 4096 x 4096 f32
        // weight costs to write and read back) and each bit-identical to the single-row call; bias and residual as the GEMV applies them
        // (in f32 "T(T(sum) + bias)" is the one f32 addition F.linear does).  More rows, or the gated epilogue: the caller's dequant + GEMM.
        if (B > 8 || (mode & kModeSiluMulPairs)) {
            set_error("fp4_hip_gemm_small: f32 activations are covered up to 8 rows without the gate|up epilogue (got B=%lld); "
                      "use dequant + GEMM", (long long)B);
            return FP4_ERR_UNSUPPORTED;
        }
        if (M == 0) return FP4_OK;
        if (!x || !out) {
            set_error("fp4_hip_gemm_small: null pointer");
            return FP4_ERR_INVALID_ARGUMENT;
        }
        for (int64_t b = 0; b < B; ++b) {
            const int rc = fp4_hip_gemv_fused(static_cast<const float *>(x) + size_t(b) * size_t(K), packed, absmax, bias,
                                              residual ? static_cast<const float *>(residual) + size_t(b) * size_t(M) : nullptr,
                                              static_cast<float *>(out) + size_t(b) * size_t(M), M, K, blocksize, FP4_DTYPE_F32,
                                              FP4_EPILOGUE_NONE, stream);
            if (rc != FP4_OK) return rc;
        }
        return FP4_OK;
    }
    if (B > 16) {
        if (dtype != FP4_DTYPE_F16 && dtype != FP4_DTYPE_BF16) {
            set_error("fp4_hip_gemm_small: more than 16 rows need a 16-bit dtype, got %d", dtype);
            return FP4_ERR_UNSUPPORTED;
        }
        // 17..64 rows: one pass over the weight with 2..4 column tiles per decoded fragment (gemm_wide_fp4.hip) where the shape
        // allows; otherwise, and above 64 rows, the rows are split evenly over several launches, each streaming the weight once -
        // still ahead of dequant + GEMM while launches x 9.45 MB < the 76 MB the dequantised weight costs to write and read back
        const uintptr_t al = reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(x);
        if (B <= 64 && M > 0 && x && packed && absmax && out && blocksize == 64 && (al & 15u) == 0 && M <= (int64_t(1) << 30) &&
            K <= (int64_t(1) << 24) &&
            gemm_wide_launch(dtype, x, packed, absmax, bias, residual, out, (int)B, (int)M, (int)K, mode, false, static_cast<hipStream_t>(stream)) == FP4_OK)
            return check_launch("fp4_hip_gemm_small");
        const int64_t unit = B > 64 ? 64 : 16;
        const int64_t chunks = (B + unit - 1) / unit, per = (B + chunks - 1) / chunks;
        for (int64_t b0 = 0; b0 < B; b0 += per) {
            const int64_t nb = B - b0 < per ? B - b0 : per;
            const int rc = gemm_small_entry(static_cast<const uint8_t *>(x) + size_t(b0) * size_t(K) * 2, packed, absmax, bias,
                                            residual ? static_cast<const uint8_t *>(residual) + size_t(b0) * size_t(M_out) * 2 : nullptr,
                                            out ? static_cast<uint8_t *>(out) + size_t(b0) * size_t(M_out) * 2 : nullptr, nb, M, K,
                                            blocksize, dtype, mode, stream);
            if (rc != FP4_OK) return rc;
        }
        return FP4_OK;
    }
    if (M == 0) return FP4_OK;
    if (!x || !packed || !absmax || !out) {
        set_error("fp4_hip_gemm_small: null pointer");
        return FP4_ERR_INVALID_ARGUMENT;
    }
    const int bs_shift = ilog2_exact(blocksize);
    const uintptr_t align = reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(x);
    const bool ok = (dtype == FP4_DTYPE_F16 || dtype == FP4_DTYPE_BF16) && (K % 32) == 0 && bs_shift >= 5 &&
                    (K % blocksize) == 0 && (align & 15u) == 0 && M <= (int64_t(1) << 30) && K <= (int64_t(1) << 24);
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc = -1;
    // matrix-core kernel: blocksize 64, K % 512 == 0; wins from 2 rows up (4.96 vs 5.11 us at 2, 5.0 vs 11.3 us at 8), mandatory above 8
    const bool mfma_ok = ok && blocksize == 64 && (K % 512) == 0;
    const int v_small = g_small_variant.load(std::memory_order_relaxed);
    // Short weights with one or two activation rows: the VALU kernel (the GEMV's geometry, no LDS staging, no workgroup barrier per tile)
    // is ahead of the matrix-core kernels' fixed cost (1024 x 4096: 3.2 vs 3.9-4.0 us, 2048 x 4096: 3.95 vs 4.25, 2048 x 2048: 3.15 vs
    // 3.35); from three rows (four x slices per lane) it falls behind, and everything else, a single row included (4096 x 14336: 9.0 vs
    // 14.8 us), is faster on the matrix cores (profiles/r03_small_batch_dispatch.txt)
    const bool short_few = M <= 2048 && B <= 2 && K <= 4096;
    const bool want_mfma = v_small == 1 || (v_small < 0 && !short_few);
    // Row lengths that leave the matrix-core kernel below with one or two quant blocks per wave and pass (K / 512 odd: 13824; K / 512
    // = 2 mod 4: 5120, 7168), and tall weights with long rows, are faster on the one-pass kernels of gemm_wide_fp4.hip with ONE
    // column tile (profiles/r02_wide_batch_17_to_128_rows.txt: 5120 x 13824 x 16 rows 49.2 -> 26.9 us, 13824 x 5120 21.7 -> 15.6 us,
    // 28672 x 8192 52.7 -> 38.8 us); up to 4 rows the old kernel holds where K / 512 is even.
    if (ok && blocksize == 64 && (K % 64) == 0 && g_small_wide.load(std::memory_order_relaxed))  // sweep hook
        rc = gemm_wide_launch(dtype, x, packed, absmax, bias, residual, out, (int)B, (int)M, (int)K, mode, true, s);
    if (rc == -1 && mfma_ok && v_small < 0 && !short_few) {
        const int64_t units = K / 512;
        // (round 3, profiles/r03_small_batch_dispatch.txt: K / 512 = 2 mod 4 also below 5 rows on tall weights - 13824 x 5120 x 2..4 rows
        //  17.4-18.4 -> 15.3-15.6 us, while 5120 x 5120 stays on the 16-row kernel, 9.0 vs 11.7; and 13..16 rows on a short weight with
        //  very long rows, where every workgroup re-reads all of x - 4096 x 14336 x 16 rows 18.1 -> 16.4 us)
        const int64_t cus = device_cu_count();
        if ((units & 1) || ((units % 4) != 0 && (B >= 5 || M >= 48 * cus)) || (M >= 96 * cus && K >= 8192 && B >= 5) ||
            (B >= 13 && K >= 12288 && M <= 16 * cus))
            rc = gemm_wide_launch(dtype, x, packed, absmax, bias, residual, out, (int)B, (int)M, (int)K, mode, true, s);
    }
    if (rc == -1 && mfma_ok && (want_mfma || B > 8))
        rc = dtype == FP4_DTYPE_F16 ? dispatch_mfma<FP4_DTYPE_F16>(x, packed, absmax, bias, residual, out, (int)B, (int)M, (int)K, mode, s)
                                    : dispatch_mfma<FP4_DTYPE_BF16>(x, packed, absmax, bias, residual, out, (int)B, (int)M, (int)K, mode, s);
    if (rc == -1 && ok && B <= 8 && K <= 16384)
        rc = dtype == FP4_DTYPE_F16
                 ? dispatch_small<FP4_DTYPE_F16>(x, packed, absmax, bias, residual, out, (int)B, (int)M, (int)K, bs_shift, mode, s)
                 : dispatch_small<FP4_DTYPE_BF16>(x, packed, absmax, bias, residual, out, (int)B, (int)M, (int)K, bs_shift, mode, s);
    if (rc == -1 && mfma_ok)
        rc = dtype == FP4_DTYPE_F16 ? dispatch_mfma<FP4_DTYPE_F16>(x, packed, absmax, bias, residual, out, (int)B, (int)M, (int)K, mode, s)
                                    : dispatch_mfma<FP4_DTYPE_BF16>(x, packed, absmax, bias, residual, out, (int)B, (int)M, (int)K, mode, s);
    // K % 512 != 0 beyond the VALU kernel's reach (e.g. Llama-2-7B's down projection, K = 11008): the one-pass kernels of
    // gemm_wide_fp4.hip take any K % 64 == 0 and 1..64 rows
    if (rc == -1 && ok && blocksize == 64 && (K % 64) == 0)
        rc = gemm_wide_launch(dtype, x, packed, absmax, bias, residual, out, (int)B, (int)M, (int)K, mode, true, s);
    if (rc == -1) {
        set_error("fp4_hip_gemm_small: shape B=%lld M=%lld K=%lld blocksize=%d dtype=%d is not covered; use dequant + GEMM",
                  (long long)B, (long long)M, (long long)K, blocksize, dtype);
        return FP4_ERR_UNSUPPORTED;
    }
    return check_launch("fp4_hip_gemm_small");
}
}  // namespace
}  // namespace fp4

extern "C" int fp4_hip_gemm_small(const void *x, const uint8_t *packed, const float *absmax, const void *bias, void *out,
                                  int64_t B, int64_t M, int64_t K, int blocksize, int dtype, void *stream) {
    return fp4::gemm_small_entry(x, packed, absmax, bias, nullptr, out, B, M, K, blocksize, dtype, 0, stream);
}

extern "C" int fp4_hip_gemm_small_fused(const void *x, const uint8_t *packed, const float *absmax, const void *bias,
                                        const void *residual, void *out, int64_t B, int64_t M, int64_t K, int blocksize, int dtype,
                                        int epilogue, void *stream) {
    if (epilogue != FP4_EPILOGUE_NONE && epilogue != FP4_EPILOGUE_SILU_MUL_PAIRS) {
        fp4::set_error("fp4_hip_gemm_small_fused: unknown epilogue %d", epilogue);
        return FP4_ERR_INVALID_ARGUMENT;
    }
    return fp4::gemm_small_entry(x, packed, absmax, bias, residual, out, B, M, K, blocksize, dtype,
                                 epilogue == FP4_EPILOGUE_SILU_MUL_PAIRS ? fp4::kModeSiluMulPairs : 0, stream);
}

extern "C" int64_t fp4_hip_gemm_small_ws_bytes(int64_t B, int64_t M, int64_t K, int blocksize, int dtype) {
    if (B > 64 && B <= 128) B = (B + 1) / 2;  // two even chunks, one after the other through the same workspace
    return fp4::gemm_splitk_workspace_bytes(B, M, K, blocksize, dtype);
}

extern "C" int fp4_hip_gemm_small_ws(const void *x, const uint8_t *packed, const float *absmax, const void *bias, const void *residual,
                                     void *out, int64_t B, int64_t M, int64_t K, int blocksize, int dtype, int epilogue, void *workspace,
                                     int64_t workspace_bytes, void *stream) {
    if (epilogue != FP4_EPILOGUE_NONE && epilogue != FP4_EPILOGUE_SILU_MUL_PAIRS) {
        fp4::set_error("fp4_hip_gemm_small_ws: unknown epilogue %d", epilogue);
        return FP4_ERR_INVALID_ARGUMENT;
    }
    const int mode = epilogue == FP4_EPILOGUE_SILU_MUL_PAIRS ? fp4::kModeSiluMulPairs : 0;
    if (B > 64 && B <= 128 && workspace && x && out && M > 0 && K > 0 && (dtype == FP4_DTYPE_F16 || dtype == FP4_DTYPE_BF16)) {
        const int64_t M_out = (mode & fp4::kModeSiluMulPairs) ? M / 2 : M, per = (B + 1) / 2;
        for (int64_t b0 = 0; b0 < B; b0 += per) {
            const int64_t nb = B - b0 < per ? B - b0 : per;
            const int rc = fp4_hip_gemm_small_ws(static_cast<const uint8_t *>(x) + size_t(b0) * size_t(K) * 2, packed, absmax, bias,
                                                 residual ? static_cast<const uint8_t *>(residual) + size_t(b0) * size_t(M_out) * 2 : nullptr,
                                                 static_cast<uint8_t *>(out) + size_t(b0) * size_t(M_out) * 2, nb, M, K, blocksize, dtype,
                                                 epilogue, workspace, workspace_bytes, stream);
            if (rc != FP4_OK) return rc;
        }
        return FP4_OK;
    }
    const uintptr_t al = reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(x);
    if (workspace && x && packed && absmax && out && (al & 15u) == 0 && B >= 33 && B <= 64 && M > 0 && M <= (int64_t(1) << 30) &&
        K <= (int64_t(1) << 24) && !((mode & fp4::kModeSiluMulPairs) && (M & 1)) &&
        fp4::gemm_splitk_launch(dtype, x, packed, absmax, bias, residual, out, (int)B, (int)M, (int)K, mode, workspace, workspace_bytes,
                                static_cast<hipStream_t>(stream)) == FP4_OK)
        return fp4::check_launch("fp4_hip_gemm_small_ws");
    return fp4::gemm_small_entry(x, packed, absmax, bias, residual, out, B, M, K, blocksize, dtype, mode, stream);
}
