// Device helpers shared by the fused FP4 GEMV (gemv_fp4.hip) and its small-batch companion (gemm_small_fp4.hip):
// the table-free nibble decode, v_dot2 wrappers, DPP reductions and the GEMV row epilogue.
#pragma once

#include "fp4_common.h"

namespace fp4 {
namespace {

// ---- byte tables for v_perm_b32: magnitude index 0..7 -> 12*|code| --------------------------
// fp16 patterns 0x0000 0x2C00 0x4800 0x4A00 0x4400 0x4600 0x4000 0x4200 (low byte always 0)
constexpr uint32_t kF16HiLo = 0x4A482C00u, kF16HiHi = 0x42404644u;
// OCP E4M3 patterns of the same eight values (exact): 0x00 0x18 0x50 0x54 0x48 0x4C 0x40 0x44 - the bf16 path widens them with
// v_cvt_scalef32_pk_bf16_fp8 (bf16 itself would need two byte planes: 0x0000 0x3D80 0x4100 0x4140 0x4080 0x40C0 0x4000 0x4040)
constexpr uint32_t kE4M3Lo = 0x54501800u, kE4M3Hi = 0x44404C48u;

__device__ __forceinline__ uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }

// One packed dword = 8 weights e0..e7 (byte b holds e_2b in its high nibble, e_2b+1 in its low
// nibble).  Produces four 16-bit pairs of 12*code: P0=(e0,e2) P1=(e4,e6) P2=(e1,e3) P3=(e5,e7).
// (FP4_EXP_* are experiment switches for tools/exp_gemv.hip only - never defined in the library build.  FP4_EXP_BITOP3: the two
// sign merges as full-rate v_bitop3_b32 with the mask in a VGPR instead of half-rate v_and_or_b32; FP4_EXP_RELAID: that, plus a
// load-time nibble permutation of every packed dword - byte 0 = (e0,e2), byte 1 = (e1,e3), byte 2 = (e4,e6), byte 3 = (e5,e7) - so
// that the decoded pairs are (e0,e1) (e4,e5) (e2,e3) (e6,e7) and x is used as loaded, without its 16 v_perm per group.)
template <int DT>
__device__ __forceinline__ void decode8(uint32_t q, uint32_t (&P)[4]) {
#if defined(FP4_EXP_BITOP3) || defined(FP4_EXP_RELAID)
    if constexpr (DT == FP4_DTYPE_BF16) {
        uint32_t m80;
        asm("v_mov_b32 %0, 0x80808080" : "=v"(m80));
        const uint32_t mhi = __builtin_amdgcn_bitop3_b32(q, m80, perm(kE4M3Hi, kE4M3Lo, (q >> 4) & 0x07070707u), 0xEA);
        const uint32_t mlo = __builtin_amdgcn_bitop3_b32(q << 4, m80, perm(kE4M3Hi, kE4M3Lo, q & 0x07070707u), 0xEA);
        P[0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(mhi, 1.0f, false));
        P[1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(mhi, 1.0f, true));
        P[2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(mlo, 1.0f, false));
        P[3] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(mlo, 1.0f, true));
        return;
    }
#endif
    const uint32_t lo_sel = q & 0x07070707u;         // magnitudes of e1,e3,e5,e7
    const uint32_t hi_sel = (q >> 4) & 0x07070707u;  // magnitudes of e0,e2,e4,e6
    const uint32_t lo_sgn = (q & 0x08080808u) << 4;  // sign -> bit 7 of each byte
    const uint32_t hi_sgn = q & 0x80808080u;
    if constexpr (DT == FP4_DTYPE_F16) {
        const uint32_t mhi = perm(kF16HiHi, kF16HiLo, hi_sel) | hi_sgn;
        const uint32_t mlo = perm(kF16HiHi, kF16HiLo, lo_sel) | lo_sgn;
        P[0] = perm(0u, mhi, 0x010C000Cu);
        P[1] = perm(0u, mhi, 0x030C020Cu);
        P[2] = perm(0u, mlo, 0x010C000Cu);
        P[3] = perm(0u, mlo, 0x030C020Cu);
    } else {
        // bf16 has no one-byte pattern for 12*|code| (the x1.5 values need mantissa bit 6), so a two-plane v_perm decode costs 8
        // v_perm per 8 weights.  gfx950's packed FP8 -> bf16 conversion does the widening instead: 12*|code| is exact in OCP E4M3,
        // one v_perm per nibble plane looks the E4M3 byte up, the sign is bit 7 there as well, and v_cvt_scalef32_pk_bf16_fp8
        // (scale 1.0, exact) turns bytes (0,1) / (2,3) of a plane into the same (e0,e2) (e4,e6) (e1,e3) (e5,e7) pairs:
        // 2 v_perm + 4 conversions (same issue rate as v_perm, profiles/r02_valu_cvt_rates.txt) instead of 8 v_perm.
        const uint32_t mhi = perm(kE4M3Hi, kE4M3Lo, hi_sel) | hi_sgn;
        const uint32_t mlo = perm(kE4M3Hi, kE4M3Lo, lo_sel) | lo_sgn;
        P[0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(mhi, 1.0f, false));
        P[1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(mhi, 1.0f, true));
        P[2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(mlo, 1.0f, false));
        P[3] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(mlo, 1.0f, true));
    }
}

template <int DT>
__device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
    if constexpr (DT == FP4_DTYPE_F16)
        return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b), c, false);
    else
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), c, false);
}

// ---- wave64 all-lanes sum without LDS storage -------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false);
    return v + __builtin_bit_cast(float, moved);
}
__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_add<0x128>(v);  // row_ror:8
    v = dpp_add<0x124>(v);  // row_ror:4
    v = dpp_add<0x122>(v);  // row_ror:2
    v = dpp_add<0x121>(v);  // row_ror:1   -> every lane holds its 16-lane row sum
    v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));  // lane ^ 16
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    return a + b;
}

// bit flags of the 16-bit GEMV kernels' `mode` argument
constexpr int kModeOutF32 = 1;        // K-split partial: the raw f32 accumulator goes out, summed across shards before any rounding
constexpr int kModeSiluMulPairs = 2;  // rows (2i, 2i+1) hold (gate_i, up_i): out[i] = silu(gate_i) * up_i  (fp4_hip_gemv_fused)

// Row epilogue.  Every step rounds to T exactly where the reference's separate torch ops would:
//   out = T(gemv); out += bias (torch_bnb_fp4/__init__.py:608-613); then the caller's `h + out` as one more rounded add.
// `residual` may alias `out` (the row is read before it is written, by the same lane).
template <int DT>
__device__ __forceinline__ void store_row(uint16_t *out, const uint16_t *bias, const uint16_t *residual, int row, float sum,
                                          int mode = 0) {
    if (mode & kModeOutF32) {
        reinterpret_cast<float *>(out)[row] = sum;
        return;
    }
    uint16_t t = from_f32<DT>(sum);
    if (bias) t = from_f32<DT>(to_f32<DT>(t) + to_f32<DT>(bias[row]));
    if (residual) t = from_f32<DT>(to_f32<DT>(t) + to_f32<DT>(residual[row]));
    out[row] = t;
}

// Gated-MLP epilogue for a weight whose rows interleave the gate and the up projection: what the model code does with
// three more launches - g = gate(h), u = up(h) (each rounded to T, bias added as above), silu(g) (torch: x / (1 + exp(-x)) in
// f32, rounded to T), then the product rounded to T - and optionally the residual add on top.
template <int DT>
__device__ __forceinline__ void store_silu_mul(uint16_t *out, const uint16_t *bias, const uint16_t *residual, int pair,
                                               float gate_sum, float up_sum) {
    uint16_t g = from_f32<DT>(gate_sum), u = from_f32<DT>(up_sum);
    if (bias) {
        g = from_f32<DT>(to_f32<DT>(g) + to_f32<DT>(bias[2 * pair]));
        u = from_f32<DT>(to_f32<DT>(u) + to_f32<DT>(bias[2 * pair + 1]));
    }
    const float gf = to_f32<DT>(g);
    const uint16_t s = from_f32<DT>(gf / (1.0f + expf(-gf)));
    uint16_t t = from_f32<DT>(to_f32<DT>(s) * to_f32<DT>(u));
    if (residual) t = from_f32<DT>(to_f32<DT>(t) + to_f32<DT>(residual[pair]));
    out[pair] = t;
}

// Small-batch (2..16 activation rows) epilogues: F.linear semantics - the bias is added to the f32 sum BEFORE the one rounding
// (torch_bnb_fp4/__init__.py:423-436 is dequant + F.linear(A, W, bias)) - then the same optional residual add / gated-MLP
// product as the GEMV's, each a rounded op of its own.  `sum` is the finished dot product (already times 1/12).
template <int DT>
__device__ __forceinline__ void store_small(uint16_t *out, const uint16_t *bias, const uint16_t *residual, int64_t b, int row, int M,
                                            float sum) {
    if (bias) sum += to_f32<DT>(bias[row]);
    uint16_t t = from_f32<DT>(sum);
    if (residual) t = from_f32<DT>(to_f32<DT>(t) + to_f32<DT>(residual[b * M + row]));
    out[b * M + row] = t;
}

template <int DT>
__device__ __forceinline__ void store_small_silu_mul(uint16_t *out, const uint16_t *bias, const uint16_t *residual, int64_t b, int pair,
                                                     int M_half, float gate_sum, float up_sum) {
    if (bias) gate_sum += to_f32<DT>(bias[2 * pair]), up_sum += to_f32<DT>(bias[2 * pair + 1]);
    const float gf = to_f32<DT>(from_f32<DT>(gate_sum));
    const uint16_t s = from_f32<DT>(gf / (1.0f + expf(-gf)));
    uint16_t t = from_f32<DT>(to_f32<DT>(s) * to_f32<DT>(from_f32<DT>(up_sum)));
    if (residual) t = from_f32<DT>(to_f32<DT>(t) + to_f32<DT>(residual[b * M_half + pair]));
    out[b * M_half + pair] = t;
}

}  // namespace
}  // namespace fp4
