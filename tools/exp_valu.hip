// VALU issue-rate microbenchmark for the instructions the FP4 GEMV decode path is built from.
// Each kernel runs ITER x 16 independent instances of one instruction per lane; 256 CUs x 8 waves/SIMD.
// Prints lane-ops per clock per CU (128 = one wave64 instruction per 2 cycles per SIMD).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int ITER = 2048;

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed) {
    uint32_t a[16];
    float f[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        a[i] = seed * (i + 1) + threadIdx.x;
        f[i] = float(i) + seed;
    }
    const uint32_t b = seed ^ 0x3c003c00u, c = seed | 0x07030602u;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if constexpr (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(b));
            if constexpr (OP == 1) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if constexpr (OP == 2) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(f[i]) : "v"(a[i]), "v"(b));
            if constexpr (OP == 3) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(f[i]) : "v"(a[i]), "v"(b));
            if constexpr (OP == 12) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(f[i]) : "v"(a[i]), "v"(b));
            if constexpr (OP == 13) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(f[i]) : "v"(a[i]), "v"(b));
            if constexpr (OP == 14) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if constexpr (OP == 15) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(f[i]) : "v"(a[i]));
            if constexpr (OP == 4) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if constexpr (OP == 5) {  // v_fma_mix_f32: f16 lo * f16 lo + f32
                float r;
                asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0]" : "=v"(r) : "v"(a[i]), "v"(b), "v"(f[i]));
                f[i] = r;
            }
            if constexpr (OP == 6) {  // v_pk_fma_f32 on register pairs
                f32x2 v = {f[i], f[(i + 1) & 15]};
                f32x2 r;
                asm volatile("v_pk_fma_f32 %0, %1, %1, %1" : "=v"(r) : "v"(v));
                f[i] = r.x;
            }
            if constexpr (OP == 7) f[i] = __builtin_bit_cast(float, a[i] << 16) * f[i];  // shift + mul
            if constexpr (OP == 8) a[i] = __builtin_amdgcn_ubfe(a[i], 4, 4) + a[i];       // v_bfe_u32 + add
            if constexpr (OP == 9) {  // v_pk_fma_f16
                f16x2 r;
                asm volatile("v_pk_fma_f16 %0, %1, %2, %1" : "=v"(r) : "v"(a[i]), "v"(b));
                a[i] = __builtin_bit_cast(uint32_t, r);
            }
            if constexpr (OP == 10) {  // v_pk_mul_f16
                f16x2 r;
                asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(r) : "v"(a[i]), "v"(b));
                a[i] = __builtin_bit_cast(uint32_t, r);
            }
            if constexpr (OP == 20) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if constexpr (OP == 21) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a[i]));
            if constexpr (OP == 22) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if constexpr (OP == 23) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a[i]) : "v"(b));
            if constexpr (OP == 24) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i]) : "v"(b), "v"(c));
            if constexpr (OP == 25) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if constexpr (OP == 26) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(b), "v"(c));
            if constexpr (OP == 27) {
                f32x2 v = {f[i], f[(i + 1) & 15]};
                f32x2 r;
                asm volatile("v_pk_add_f32 %0, %1, %1" : "=v"(r) : "v"(v));
                f[i] = r.x;
            }
            if constexpr (OP == 28) asm volatile("v_lshl_or_b32 %0, %0, 4, %1" : "+v"(a[i]) : "v"(b));
            if constexpr (OP == 29) asm volatile("v_cmp_lt_u32 vcc, %1, %0\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
            if constexpr (OP == 30) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(b));
            if constexpr (OP == 31) asm volatile("v_pk_sub_u16 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if constexpr (OP == 32) asm volatile("v_pk_lshrrev_b16 %0, 15, %0" : "+v"(a[i]));
            if constexpr (OP == 33) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if constexpr (OP == 34) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if constexpr (OP == 35) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(a[i]) : "s"(seed));
            if constexpr (OP == 36) asm volatile("v_sub_u32 %0, 0x3e955555, %0" : "+v"(a[i]));
            if constexpr (OP == 37) asm volatile("v_max_f32 %0, |%0|, |%1|" : "+v"(f[i]) : "v"(b));
            if constexpr (OP == 38) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if constexpr (OP == 39) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            // round 2: gfx950 low-precision conversions and SDWA forms, candidates for a cheaper FP4 decode
            if constexpr (OP == 40) asm volatile("v_cvt_scalef32_pk_bf16_fp8 %0, %1, %2" : "=v"(a[i]) : "v"(a[(i + 1) & 15]), "v"(f[0]));
            if constexpr (OP == 41) asm volatile("v_cvt_scalef32_pk_f16_fp8 %0, %1, %2" : "=v"(a[i]) : "v"(a[(i + 1) & 15]), "v"(f[0]));
            if constexpr (OP == 42) asm volatile("v_cvt_scalef32_pk_bf16_fp4 %0, %1, %2" : "=v"(a[i]) : "v"(a[(i + 1) & 15]), "v"(f[0]));
            if constexpr (OP == 43) asm volatile("v_cvt_scalef32_pk_f32_fp4 %0, %1, %2" : "=v"(*reinterpret_cast<f32x2 *>(&f[i & 14])) : "v"(a[i]), "v"(f[0]));
            if constexpr (OP == 44) asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(a[i]) : "v"(b));
            if constexpr (OP == 45) asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2" : "+v"(a[i]) : "v"(b));
            if constexpr (OP == 46) asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(*reinterpret_cast<f32x2 *>(&f[i & 14])) : "v"(*reinterpret_cast<f32x2 *>(&f[(i + 2) & 14])));
            if constexpr (OP == 11) {  // v_cvt_f32_f16 (SDWA-free) + fma
                f[i] = __builtin_fmaf(float(__builtin_bit_cast(f16x2, a[i]).x), f[i], 1.0f);
            }
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) r ^= a[i] ^ __builtin_bit_cast(uint32_t, f[i]);
    if (r == 0x12345678u) out[threadIdx.x] = r;
}

template <int OP>
void run(const char *name, int instr_per_step, uint32_t *out) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int blocks = 256 * 8;  // 8 blocks of 4 waves per CU = 8 waves per SIMD
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 3u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 3u);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double lane_ops = double(blocks) * 256 * ITER * 16 * instr_per_step;
    printf("%-34s %8.3f ms  %6.1f lane-instr/ns/chip = %6.1f per clk per CU @2.4GHz\n", name, ms, lane_ops / (ms * 1e6),
           lane_ops / (ms * 1e6) / 256 / 2.4);
}

int main() {
    uint32_t *out;
    CK(hipMalloc(&out, 4096));
    if (getenv("EXP_VALU_INT")) {
        run<14>("v_and_b32", 1, out);
        run<20>("v_sub_u32", 1, out);
        run<35>("v_sub_u32 (sgpr operand)", 1, out);
        run<36>("v_sub_u32 (literal operand)", 1, out);
        run<21>("v_lshrrev_b32", 1, out);
        run<33>("v_xor_b32", 1, out);
        run<22>("v_add3_u32", 1, out);
        run<34>("v_or3_b32", 1, out);
        run<28>("v_lshl_or_b32", 1, out);
        run<23>("v_alignbit_b32", 1, out);
        run<24>("v_bitop3_b32", 1, out);
        run<38>("v_bfi_b32", 1, out);
        run<39>("v_mad_u32_u24", 1, out);
        run<25>("v_bcnt_u32_b32", 1, out);
        run<29>("v_cmp_lt_u32 + v_addc_co (2 instr)", 2, out);
        run<31>("v_pk_sub_u16", 1, out);
        run<32>("v_pk_lshrrev_b16", 1, out);
        run<30>("v_mul_f32", 1, out);
        run<37>("v_max_f32 |a|,|b|", 1, out);
        run<26>("v_max3_f32", 1, out);
        run<27>("v_pk_add_f32", 1, out);
        return 0;
    }
    if (getenv("EXP_VALU_CVT")) {
        run<1>("v_perm_b32", 1, out);
        run<2>("v_dot2c_f32_bf16", 1, out);
        run<14>("v_and_b32", 1, out);
        run<40>("v_cvt_scalef32_pk_bf16_fp8", 1, out);
        run<41>("v_cvt_scalef32_pk_f16_fp8", 1, out);
        run<42>("v_cvt_scalef32_pk_bf16_fp4", 1, out);
        run<43>("v_cvt_scalef32_pk_f32_fp4", 1, out);
        run<44>("v_lshlrev_b32_sdwa (BYTE_1)", 1, out);
        run<45>("v_mov_b32_sdwa (dst BYTE_1 preserve)", 1, out);
        run<46>("v_pk_mul_f32", 1, out);
        return 0;
    }
    run<0>("v_fma_f32", 1, out);
    run<1>("v_perm_b32", 1, out);
    run<2>("v_dot2c_f32_bf16", 1, out);
    run<3>("v_dot2c_f32_f16", 1, out);
    run<4>("v_and_or_b32", 1, out);
    run<5>("v_fma_mix_f32", 1, out);
    run<6>("v_pk_fma_f32", 1, out);
    run<7>("v_lshl + v_mul_f32 (2 instr)", 2, out);
    run<8>("v_bfe_u32 + v_add (2 instr)", 2, out);
    run<9>("v_pk_fma_f16", 1, out);
    run<10>("v_pk_mul_f16", 1, out);
    run<11>("v_cvt_f32_f16 + v_fma (2 instr)", 2, out);
    run<12>("v_dot2_f32_bf16 (VOP3P)", 1, out);
    run<13>("v_dot2_f32_f16 (VOP3P)", 1, out);
    run<14>("v_and_b32", 1, out);
    run<15>("v_cvt_f32_f16", 1, out);
    return 0;
}
