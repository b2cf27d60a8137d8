/*
 * CPU oracle for the FP4 dequant / fused-GEMV hot path -- plain C restatement.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check in
 * __graft_entry__.py and the cpu_baseline leg of bench.py may load it.  It is a
 * second, independent restatement next to oracle/fp4_oracle.py; the two are
 * cross-checked against each other in tests/test_oracle.py.
 *
 * Reference = aredden/torch-bnb-fp4 (paths relative to its checkout):
 *   tables            csrc/dequant_fp4_optimized.cu:28-46 (CODE_PARAM), :55-76 (tree)
 *   dequant indexing  csrc/dequant_fp4_optimized.cu:107-121, 156-169
 *   conversions       csrc/dequant_fp4_optimized.cu:78-87
 *   GEMV indexing     csrc/gemv_fp4_optimized.cu:99-156
 * Pinning: see the header of oracle/fp4_oracle.py (reference not buildable or
 * importable here; pinned by the reference's table literals and its published
 * acceptance statistic; the bitsandbytes quantiser is "parity unpinned").
 *
 * Build: make -C oracle   (gcc -O2 -fopenmp -shared -fPIC)
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

enum { FP4_ORACLE_F16 = 0, FP4_ORACLE_F32 = 1, FP4_ORACLE_BF16 = 2 }; /* order of ScalarTypeEnum, csrc/torch_fp4.cpp:22-26 */
enum { FP4_ORACLE_TABLE_CODEBOOK = 0, FP4_ORACLE_TABLE_TREE = 1 };

/* Decimal literals exactly as the reference spells them; the C compiler does the
 * decimal -> binary32 rounding, which is what pins the hex constants used by the
 * numpy oracle and by the HIP kernels. */
static const float kCodebookMag[8] = {0.00000f, 5.208333e-03f, 0.6666667f, 1.000000f,
                                      0.333333f, 0.500000f,    0.1666667f, 0.250000f};
static const float kTreeMag[8] = {0.00000000f, 5.208333333e-03f, 0.66666667f, 1.00000000f,
                                  0.33333333f, 0.50000000f,      0.16666667f, 0.25000000f};

void fp4_oracle_table(int which, float out[16]) {
    const float *m = which == FP4_ORACLE_TABLE_TREE ? kTreeMag : kCodebookMag;
    for (int i = 0; i < 8; ++i) {
        out[i] = m[i];
        out[i + 8] = -m[i]; /* nibble 8 is -0.0 */
    }
}

static inline uint32_t f32_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

/* binary32 -> bfloat16, round to nearest even (__float2bfloat16_rn) */
static inline uint16_t f32_to_bf16(float f) {
    uint32_t u = f32_bits(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x0040u);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

/* binary32 -> binary16, round to nearest even, subnormals kept (__float2half_rn) */
static inline uint16_t f32_to_f16(float f) {
    uint32_t u = f32_bits(f);
    uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
    uint32_t a = u & 0x7FFFFFFFu;
    if (a > 0x7F800000u) return (uint16_t)(sign | 0x7E00u);      /* NaN */
    if (a >= 0x47800000u) return (uint16_t)(sign | 0x7C00u);     /* >= 65536 or inf -> inf */
    if (a >= 0x38800000u) {                                      /* normal half range */
        uint32_t v = a - 0x38000000u;                            /* rebias 127 -> 15 */
        uint32_t r = v + 0x0FFFu + ((v >> 13) & 1u);
        return (uint16_t)(sign | (r >> 13));                     /* carry may reach inf: correct */
    }
    if (a < 0x33000000u) return sign;                            /* < 2^-25 -> 0 */
    /* subnormal half: value = mant * 2^(e-150); half ulp = 2^-24 */
    uint32_t e = a >> 23;
    uint32_t mant = (a & 0x7FFFFFu) | 0x800000u;
    uint32_t shift = 126u - e;                                   /* 14..24 */
    uint32_t q = mant >> shift;
    uint32_t rem = mant & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1u);
    if (rem > half || (rem == half && (q & 1u))) q++;
    return (uint16_t)(sign | q);
}

/*
 * Blockwise dequant.  Element e (0 <= e < n) comes from byte e/2, HIGH nibble when e
 * is even; its scale is absmax[(16*(e/16)/2) / (blocksize/2)] -- the reference's
 * one-lookup-per-8-byte-thread rule (csrc/dequant_fp4_optimized.cu:110,159), which
 * is absmax[e / blocksize] whenever blocksize % 16 == 0.
 */
void fp4_oracle_dequant(const uint8_t *packed, const float *absmax, void *out, int blocksize, int64_t n,
                        int dtype, int which_table) {
    float tab[16];
    fp4_oracle_table(which_table, tab);
    const int64_t half_bs = blocksize / 2;
#pragma omp parallel for schedule(static)
    for (int64_t g = 0; g < (n + 15) / 16; ++g) {
        const float am = absmax[(g * 8) / half_bs];
        const int64_t e_end = (g * 16 + 16 < n) ? g * 16 + 16 : n;
        for (int64_t e = g * 16; e < e_end; ++e) {
            const uint8_t b = packed[e >> 1];
            const uint8_t nib = (e & 1) ? (uint8_t)(b & 0x0F) : (uint8_t)(b >> 4);
            const float v = tab[nib] * am;
            if (dtype == FP4_ORACLE_F32)
                ((float *)out)[e] = v;
            else if (dtype == FP4_ORACLE_F16)
                ((uint16_t *)out)[e] = f32_to_f16(v);
            else
                ((uint16_t *)out)[e] = f32_to_bf16(v);
        }
    }
}

/* float64 accumulate of x @ dequant_f32(W)^T, x given as doubles (exact activations). */
void fp4_oracle_gemv_f64(const double *x, const uint8_t *packed, const float *absmax, double *out, int64_t M,
                         int64_t K, int blocksize, int which_table) {
    float tab[16];
    fp4_oracle_table(which_table, tab);
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < M; ++r) {
        double acc = 0.0;
        for (int64_t k = 0; k < K; ++k) {
            const int64_t e = r * K + k;
            const uint8_t b = packed[e >> 1];
            const uint8_t nib = (e & 1) ? (uint8_t)(b & 0x0F) : (uint8_t)(b >> 4);
            const float w = tab[nib] * absmax[e / blocksize];
            acc += (double)w * x[k];
        }
        out[r] = acc;
    }
}

/* bitsandbytes-style FP4 blockwise quantiser ("parity unpinned", see fp4_oracle.py). */
static inline uint8_t quantize_one(float x) {
    uint8_t sign = x < 0.0f ? 8 : 0;
    float a = fabsf(x);
    if (a > 0.29166667f) {
        if (a > 0.583333f) return (uint8_t)((a > 0.8333333f ? 3 : 2) | sign);
        return (uint8_t)((a > 0.4166667f ? 5 : 4) | sign);
    }
    if (a > 0.0859375f) return (uint8_t)((a > 0.20833333f ? 7 : 6) | sign);
    return (uint8_t)((a > 0.00260417f ? 1 : 0) | sign);
}

void fp4_oracle_quantize(const float *w, uint8_t *packed, float *absmax, int64_t n, int blocksize) {
    const int64_t nblocks = (n + blocksize - 1) / blocksize;
    memset(packed, 0, (size_t)((n + 1) / 2));
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < nblocks; ++b) {
        const int64_t lo = b * blocksize, hi = (lo + blocksize < n) ? lo + blocksize : n;
        float m = 0.0f;
        for (int64_t e = lo; e < hi; ++e) m = fmaxf(m, fabsf(w[e]));
        absmax[b] = m;
        const float inv = 1.0f / m;
        for (int64_t e = lo; e < hi; ++e) {
            const uint8_t q = quantize_one(w[e] * inv);
            /* blocksize is even, so two blocks never share a byte */
            packed[e >> 1] |= (e & 1) ? q : (uint8_t)(q << 4);
        }
    }
}
