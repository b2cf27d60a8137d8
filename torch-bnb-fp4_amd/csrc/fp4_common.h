// Shared device/host helpers for the gfx950 FP4 kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "torch_bnb_fp4_hip.h"

namespace fp4 {

// ---- code tables (binary32 bit patterns; nibble bit 3 = sign) -------------------
// CODE_PARAM of the reference (csrc/dequant_fp4_optimized.cu:28-46) and the constants
// of dequantize_fp4_tree (:55-76), as a C compiler rounds those decimal literals.
// tests/test_oracle.py re-derives both rows with gcc from the decimal spellings.
#define FP4_CODEBOOK_MAG_BITS 0x00000000u, 0x3BAAAAAAu, 0x3F2AAAABu, 0x3F800000u, 0x3EAAAA9Fu, 0x3F000000u, 0x3E2AAAADu, 0x3E800000u
#define FP4_TREE_MAG_BITS 0x00000000u, 0x3BAAAAABu, 0x3F2AAAABu, 0x3F800000u, 0x3EAAAAABu, 0x3F000000u, 0x3E2AAAABu, 0x3E800000u

static constexpr uint32_t kMagBits[2][8] = {{FP4_CODEBOOK_MAG_BITS}, {FP4_TREE_MAG_BITS}};

struct CodeTable {
    uint32_t bits[16];
};

inline CodeTable make_table(int which) {
    CodeTable t;
    for (int i = 0; i < 8; ++i) {
        t.bits[i] = kMagBits[which][i];
        t.bits[i + 8] = kMagBits[which][i] | 0x80000000u;
    }
    return t;
}

// One LUT entry from immediates only (select chain, no memory access): what the 16 staging
// lanes of a workgroup run once to fill the LDS table.
__device__ __forceinline__ float lut_entry(int which, int idx) {
    const int m = idx & 7;
    const bool tree = which == FP4_TABLE_TREE;
    uint32_t b = 0u;
    b = m == 1 ? (tree ? kMagBits[1][1] : kMagBits[0][1]) : b;
    b = m == 2 ? kMagBits[0][2] : b;
    b = m == 3 ? kMagBits[0][3] : b;
    b = m == 4 ? (tree ? kMagBits[1][4] : kMagBits[0][4]) : b;
    b = m == 5 ? kMagBits[0][5] : b;
    b = m == 6 ? (tree ? kMagBits[1][6] : kMagBits[0][6]) : b;
    b = m == 7 ? kMagBits[0][7] : b;
    return __builtin_bit_cast(float, b | (uint32_t(idx & 8) << 28));
}
static_assert(kMagBits[0][2] == kMagBits[1][2] && kMagBits[0][3] == kMagBits[1][3] && kMagBits[0][5] == kMagBits[1][5] &&
                  kMagBits[0][7] == kMagBits[1][7] && kMagBits[0][0] == 0 && kMagBits[1][0] == 0,
              "lut_entry assumes the two tables differ only at magnitudes 1, 4 and 6");

// ---- status plumbing ---------------------------------------------------------------
void set_error(const char *fmt, ...);
int check_launch(const char *what);
int device_cu_count();

inline int ilog2_exact(int64_t v) {  // log2(v) if v is a power of two, else -1
    if (v <= 0 || (v & (v - 1))) return -1;
    int s = 0;
    while ((int64_t(1) << s) < v) ++s;
    return s;
}

// ---- vector types ------------------------------------------------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// f32 pair -> packed 16-bit pair, round-to-nearest-even (never v_cvt_pkrtz).
// lo lands in bits 15:0.
template <int DT>
__device__ __forceinline__ uint32_t pack2(float lo, float hi);
template <>
__device__ __forceinline__ uint32_t pack2<FP4_DTYPE_BF16>(float lo, float hi) {
    f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
template <>
__device__ __forceinline__ uint32_t pack2<FP4_DTYPE_F16>(float lo, float hi) {
    f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}

template <int DT>
__device__ __forceinline__ float to_f32(uint16_t bits);
template <>
__device__ __forceinline__ float to_f32<FP4_DTYPE_BF16>(uint16_t bits) {
    return __builtin_bit_cast(float, uint32_t(bits) << 16);
}
template <>
__device__ __forceinline__ float to_f32<FP4_DTYPE_F16>(uint16_t bits) {
    return float(__builtin_bit_cast(_Float16, bits));
}

// f32 -> T of a value that was just produced by an f32 multiply/add.  The optimisation barrier
// keeps hipcc from folding "mul; cvt" into v_fma_mixlo_f16 (a*b + (+0.0)), which turns a -0.0
// product into +0.0 - the FP4 code has a -0 entry (nibble 8), so that is a visible bit.
__device__ __forceinline__ float opaque(float v) {
    asm volatile("" : "+v"(v));
    return v;
}

template <int DT>
__device__ __forceinline__ uint16_t from_f32(float v);
template <>
__device__ __forceinline__ uint16_t from_f32<FP4_DTYPE_BF16>(float v) {
    return __builtin_bit_cast(uint16_t, __bf16(opaque(v)));
}
template <>
__device__ __forceinline__ uint16_t from_f32<FP4_DTYPE_F16>(float v) {
    return __builtin_bit_cast(uint16_t, _Float16(opaque(v)));
}

}  // namespace fp4
