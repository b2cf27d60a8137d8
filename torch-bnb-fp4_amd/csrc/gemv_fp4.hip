// Fused FP4 GEMV (and its small-batch companion) for gfx950 (MI355X):
//     out[r] = sum_k x[k] * code[nib(r,k)] * absmax[(r*K+k)/bs]      (+ bias[r])
//
// Replaces gemv_4bit_inference_kernel / gemv_4bit_inference_kernel_float and their host dispatcher
// (reference csrc/gemv_fp4_optimized.cu:60-368).  2 flops per 0.5625 B of weight stream: no MFMA.  What limits a
// wave64 GEMV on this chip is VALU / LDS issue and launch + HBM latency, not bandwidth: at 8 TB/s each CU must retire
// ~23 weights per clock, v_perm_b32 / v_dot2c / v_fma_mix issue at half rate (profiles/r01_b_exp_valu_issue_rates.txt),
// and one LDS table read per weight would saturate the LDS pipe.
//
// Shared building blocks
//  * decode8: nibble -> value WITHOUT a table in memory.  12*|code| = {0, 1/16, 8, 12, 4, 6, 2, 3} is exact in fp16 and
//    bf16 and its fp16 pattern fits one byte, so one v_perm_b32 whose 64-bit pool is the 8-entry byte table decodes four
//    nibbles (bf16: a second byte plane); signs are OR-ed in from nibble bit 3; v_perm_b32 again pairs the bytes up for
//    v_dot2_f32_f16 / v_dot2_f32_bf16 (f32 accumulate).  ~2.5 issue slots per weight.
//  * absmax is factored out of every 32-weight chunk (acc += absmax * sum x*12code); 1/12 is applied once per row.
//    Accumulation is f32 throughout (the reference accumulates per lane in half/bf16, csrc/gemv_fp4_optimized.cu:87,
//    146-148), so results are closer to the exact x @ dequant(W)^T than the reference's, not bit-identical to it.
//  * two ordering rules: x (L2-resident) is requested BEFORE the HBM weight stream, because vector-memory results return
//    in issue order; and load phases are branch-free (clamped index + zero scale), because a load under a divergent `if`
//    makes hipcc drain the queue with vmcnt(0) at the join.
//
// Kernels, in file order
//  * gemv16_kernel        - LDS geometry (the north-star mapping): one wave64 per row, x staged once per workgroup in a
//                           pre-permuted, bank-conflict-free LDS image; serves K > 16384 (other than 28672) and stays selectable for sweeps.
//  * gemv16_regx_kernel   - register-x geometry, the DEFAULT for 16-bit activations: a lane owns the same K-chunks for every
//                           row of its workgroup, so its slice of x lives in VGPRs; K split into 1..8 bands of 32 chunks,
//                           one wave per band, the band count chosen so that no lane of a band idles (default_variant16).
//  * gemv32_kernel / gemv32_regx_kernel - f32 activations (bit-faithful CODE_PARAM f32 table in LDS).
//  * gemv_generic_kernel  - any even K / blocksize, unaligned operands.
// (the 2..64-row small-batch kernels built on the same decode live in gemm_small_fp4.hip)
#include <atomic>

#include "gemv_common.h"

#ifdef FP4_EXP_STAMPS
// Diagnostic build only (tools/exp_gemv.hip -DFP4_EXP_STAMPS): per-wave wall-clock stamps (s_memrealtime, 100 MHz) of the register-x
// kernel - entry, loads issued, last data consumed, exit - written to a buffer of their own; no output value depends on them.
__device__ unsigned long long *fp4_exp_stamps = nullptr;
extern "C" int fp4_exp_set_stamps(void *buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(fp4_exp_stamps), &buf, sizeof(buf));
}
#define FP4_STAMP(i)                                                                                        \
    do {                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        const unsigned long long t_ = __builtin_amdgcn_s_memrealtime();                                     \
        if ((threadIdx.x & 63) == 0 && fp4_exp_stamps)                                                      \
            fp4_exp_stamps[(size_t(blockIdx.x) * WAVES + (threadIdx.x >> 6)) * 4 + (i)] = t_;               \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
    } while (0)
#else
#define FP4_STAMP(i)
#endif

namespace fp4 {

namespace {

// ---- 16-bit activations: the decode path -----------------------------------------------------
// LDS image of x: group g (8 values) of chunk c sits at 16-byte slot [g*C + c], holding the
// dwords (x0,x2) (x4,x6) (x1,x3) (x5,x7) that pair with decode8's P0..P3.
template <int ROWS, int UNROLL>
struct WTrip {  // one trip of a wave: UNROLL chunks x ROWS rows of packed weights + their scales
    u32x4 wq[UNROLL][ROWS];
    float am[UNROLL][ROWS];
};

// Branch-free on purpose: a load inside a divergent `if` makes hipcc drain the whole queue (vmcnt(0)) at the
// join, which would serialise "all weights arrived" -> "first FMA".  Out-of-range chunks are clamped to the
// last valid one and neutralised through a zero scale instead.
template <int ROWS, int UNROLL>
__device__ __forceinline__ void issue_trip(WTrip<ROWS, UNROLL> &t, const u32x4 *__restrict__ Wv,
                                           const float *__restrict__ absmax, const int (&rows)[ROWS], int c0, int C,
                                           int bs_shift) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const int c = c0 + 64 * u;
        const int cc = c < C ? c : C - 1;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int64_t chunk = int64_t(rows[r]) * C + cc;
            t.wq[u][r] = __builtin_nontemporal_load(Wv + chunk);
            const float a = absmax[(chunk << 5) >> bs_shift];
            t.am[u][r] = c < C ? a : 0.0f;
        }
    }
}

template <int DT, int ROWS, int UNROLL>
__device__ __forceinline__ void consume_trip(const WTrip<ROWS, UNROLL> &t, const u32x4 *s_x4, int c0, int C,
                                             float (&acc)[ROWS]) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const int c = c0 + 64 * u;
        const int cc = c < C ? c : C - 1;  // clamped chunks carry a zero scale
        u32x4 xd[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) xd[g] = s_x4[g * C + cc];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint32_t P[4];
                decode8<DT>(t.wq[u][r][g], P);
                s[g] = dot2<DT>(P[0], xd[g].x, s[g]);
                s[g] = dot2<DT>(P[1], xd[g].y, s[g]);
                s[g] = dot2<DT>(P[2], xd[g].z, s[g]);
                s[g] = dot2<DT>(P[3], xd[g].w, s[g]);
            }
            acc[r] = __builtin_fmaf((s[0] + s[1]) + (s[2] + s[3]), t.am[u][r], acc[r]);
        }
    }
}

template <int DT, int ROWS, int WAVES, int UNROLL>
__global__ __launch_bounds__(WAVES * 64) void gemv16_kernel(const uint16_t *__restrict__ x,
                                                            const uint8_t *__restrict__ W,
                                                            const float *__restrict__ absmax,
                                                            const uint16_t *__restrict__ bias,
                                                            const uint16_t *residual, uint16_t *out, int M, int K,
                                                            int bs_shift, int mode) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_x[];
    u32x4 *s_x4 = reinterpret_cast<u32x4 *>(s_x);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int C = K >> 5;  // 32-weight chunks per row
    constexpr int kStride = 64 * UNROLL;
    const int trips = (C + kStride - 1) / kStride;

    const int row0 = (blockIdx.x * WAVES + wave) * ROWS;
    int rows[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) rows[r] = (row0 + r < M) ? row0 + r : M - 1;  // clamped rows are computed, not stored
    const u32x4 *Wv = reinterpret_cast<const u32x4 *>(W);

    // Order matters: vector-memory results return in issue order (vmcnt), so the (L2-resident) x pieces are
    // requested FIRST and the HBM weight stream of the first trip right behind them; the LDS staging of x
    // then only waits for the x pieces while the weights are still in flight.
    constexpr int kXPre = 2;  // x pieces per thread requested ahead of the weights (covers K <= 4096 * WAVES / 4)
    u32x4 xpre[kXPre];
#pragma unroll
    for (int i = 0; i < kXPre; ++i) {
        const int p = tid + i * WAVES * 64;
        xpre[i] = reinterpret_cast<const u32x4 *>(x)[p < 4 * C ? p : 4 * C - 1];
    }
    WTrip<ROWS, UNROLL> ta, tb;
    issue_trip<ROWS, UNROLL>(ta, Wv, absmax, rows, lane, C, bs_shift);

    auto stage = [&](int p, const u32x4 w) {
        u32x4 d;
        d.x = perm(w.y, w.x, 0x05040100u);  // (x0,x2)
        d.y = perm(w.w, w.z, 0x05040100u);  // (x4,x6)
        d.z = perm(w.y, w.x, 0x07060302u);  // (x1,x3)
        d.w = perm(w.w, w.z, 0x07060302u);  // (x5,x7)
        s_x4[(p & 3) * C + (p >> 2)] = d;
    };
#pragma unroll
    for (int i = 0; i < kXPre; ++i) {
        const int p = tid + i * WAVES * 64;
        if (p < 4 * C) stage(p, xpre[i]);
    }
    for (int p = tid + kXPre * WAVES * 64; p < 4 * C; p += WAVES * 64) stage(p, reinterpret_cast<const u32x4 *>(x)[p]);
    __syncthreads();

    float acc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = 0.0f;

    // ping-pong the two register sets so the next trip's loads fly while this one is consumed
    for (int t = 0;;) {
        if (t + 1 < trips) issue_trip<ROWS, UNROLL>(tb, Wv, absmax, rows, lane + (t + 1) * kStride, C, bs_shift);
        consume_trip<DT, ROWS, UNROLL>(ta, s_x4, lane + t * kStride, C, acc);
        if (++t >= trips) break;
        if (t + 1 < trips) issue_trip<ROWS, UNROLL>(ta, Wv, absmax, rows, lane + (t + 1) * kStride, C, bs_shift);
        consume_trip<DT, ROWS, UNROLL>(tb, s_x4, lane + t * kStride, C, acc);
        if (++t >= trips) break;
    }

#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const float total = wave_sum(acc[r]) * (1.0f / 12.0f);
        if (lane == 0 && row0 + r < M) store_row<DT>(out, bias, residual, row0 + r, total, mode);
    }
}

// ---- 16-bit activations, second geometry: x in registers, K split across the waves of a workgroup ----
// For decode shapes the LDS staging of x (K*2 bytes per workgroup, a barrier before the first FMA, and
// as much L2->LDS traffic as the weight stream itself at 4 rows per workgroup) is the critical path,
// not HBM.  Here a lane owns the SAME K-chunks for every row its workgroup handles, so its slice of x is
// loaded from L2 once into VGPRs and never touches LDS:
//   * 4 waves; wave w = (kw, rw): kw in [0,KSPLIT) picks a 32-chunk column band, rw a row-pair group;
//     the two 32-lane halves of a wave work on two different rows (one 512-byte span of each per load);
//   * lane (kw, l&31) owns chunks  g*32*KSPLIT + kw*32 + (l&31),  g < G   (G*16 VGPRs of permuted x);
//   * a workgroup handles ITERS row-pairs per row group; every weight load of the workgroup is issued
//     before the first dot2; per (row, band) partial sums meet in LDS (KSPLIT floats per row), one
//     barrier at the very end; with KSPLIT == 1 nothing is shared at all.
// (a deep x slice, G = 4, is held to 128 VGPRs = 4 waves per SIMD: at 142 only three 4-wave workgroups fit a CU, the
// 1024 workgroups of a 4096-row layer no longer run as one resident round and a third of the chip idles in the second)
template <int DT, int KSPLIT, int G, int ITERS, int WAVES = 4>
__global__ __launch_bounds__(WAVES * 64, (G >= 4 ? 4 : 1)) void gemv16_regx_kernel(const uint16_t *__restrict__ x, const uint8_t *__restrict__ W,
                                                          const float *__restrict__ absmax,
                                                          const uint16_t *__restrict__ bias, const uint16_t *residual,
                                                          uint16_t *out, int M, int K, int bs_shift, int mode) {
    constexpr int RG = WAVES / KSPLIT;           // row-pair groups per workgroup
    constexpr int kRowsPerBlock = 2 * RG * ITERS;
    __shared__ float s_part[kRowsPerBlock][KSPLIT];
    FP4_STAMP(0);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int kw = wave % KSPLIT, rw = wave / KSPLIT;
    const int half = lane >> 5, l32 = lane & 31;
    const int C = K >> 5;
    const int row_base = blockIdx.x * kRowsPerBlock;

    // Everything below is branch-free (see issue_trip): dead lanes / rows are clamped and get a zero scale.
    int cidx[G];
    bool live[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int c = g * (32 * KSPLIT) + kw * 32 + l32;
        live[g] = c < C;
        cidx[g] = live[g] ? c : C - 1;
    }
    // 1. this lane's slice of x first: results return in issue order, and x (L2-resident, shared by every
    //    workgroup) must not queue behind the HBM weight stream.  With a deep slice (G > 1: long rows) the x loads
    //    alone keep the texture path busy for over a microsecond, so there the issue order is group-major - x of
    //    group g, then every row's weights of group g - and the HBM stream starts after a quarter of x.
    constexpr bool GMAJOR = G > 1;
    u32x4 xraw[G][4];
    u32x4 wq[ITERS][G];
    float am[ITERS][G];
    int rowi[ITERS], rclamp[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int local = 2 * (it * RG + rw) + half;
        const int row = row_base + local;
        rowi[it] = local;
        rclamp[it] = row < M ? row : M - 1;
    }
    // (FP4_ABL_* are ablation switches for tools/exp_gemv.hip only - never defined in the library build)
    auto load_x = [&](int g) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#ifdef FP4_ABL_NOX
            xraw[g][q] = u32x4{uint32_t(cidx[g]), 0x3c003c00u, uint32_t(q), 0x3c003c00u};
#else
            xraw[g][q] = reinterpret_cast<const u32x4 *>(x)[cidx[g] * 4 + q];
#endif
        }
    };
    // Weights and scales come through buffer descriptors: a wave-uniform base + one 32-bit offset per lane, instead of a 64-bit
    // multiply-add and two 64-bit shifts per load on the VALU - this kernel is vector-issue-bound in steady state
    // (profiles/r02_gemv_steady_state_sq_counters.txt).  The dispatcher only comes here while M * K < 2^32.
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(W), 0, int((uint32_t(M) * uint32_t(C)) << 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_a =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(absmax), 0, int(((uint32_t(M) * uint32_t(C)) >> (bs_shift - 5)) << 2), 0x00020000);
    auto load_w = [&](int it, int g) {
        const uint32_t chunk = uint32_t(rclamp[it]) * uint32_t(C) + uint32_t(cidx[g]);
        wq[it][g] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, int(chunk << 4), 0, 2));  // aux 2 = nt
#ifdef FP4_ABL_NOABSMAX
        const float a = 0.5f;
#else
        const float a = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_a, int((chunk >> (bs_shift - 5)) << 2), 0, 0));
#endif
        am[it][g] = live[g] ? a : 0.0f;
    };
    if constexpr (GMAJOR) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            load_x(g);
#pragma unroll
            for (int it = 0; it < ITERS; ++it) load_w(it, g);
        }
    } else {
#pragma unroll
        for (int g = 0; g < G; ++g) load_x(g);
        // 2. the whole weight stream of this lane, issued up front
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int g = 0; g < G; ++g) load_w(it, g);
        }
    }
    FP4_STAMP(1);
    // permute x to decode8's pairing: (x0,x2) (x4,x6) (x1,x3) (x5,x7) per group of 8
    u32x4 xd[G][4];
    auto permute_x = [&](int g) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const u32x4 w = xraw[g][q];
#ifdef FP4_EXP_RELAID
            xd[g][q] = u32x4{w.x, w.z, w.y, w.w};  // pairs (x0,x1) (x4,x5) (x2,x3) (x6,x7): register renaming only
#else
            xd[g][q].x = perm(w.y, w.x, 0x05040100u);
            xd[g][q].y = perm(w.w, w.z, 0x05040100u);
            xd[g][q].z = perm(w.y, w.x, 0x07060302u);
            xd[g][q].w = perm(w.w, w.z, 0x07060302u);
#endif
        }
    };
    if constexpr (!GMAJOR) {
#pragma unroll
        for (int g = 0; g < G; ++g) permute_x(g);
    }
    // 3. decode + dot, one row per half-wave per iteration (consumed in the order the loads were issued; the sum
    //    over g of each row is accumulated in the same order either way)
    float pacc[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) pacc[it] = 0.0f;
    auto consume = [&](int it, int g) {
        // two accumulation chains per 32-weight chunk (eight dependent v_dot2 each): with several waves per SIMD the chain latency
        // is covered, and one add per chunk replaces three
        float s[2] = {0.0f, 0.0f};
#ifdef FP4_ABL_NOCOMPUTE
#pragma unroll
        for (int q = 0; q < 4; ++q) s[q & 1] = __builtin_bit_cast(float, (wq[it][g][q] ^ xd[g][q].x) & 0x3fffffffu);
#else
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t P[4];
            decode8<DT>(wq[it][g][q], P);
            s[q & 1] = dot2<DT>(P[0], xd[g][q].x, s[q & 1]);
            s[q & 1] = dot2<DT>(P[1], xd[g][q].y, s[q & 1]);
            s[q & 1] = dot2<DT>(P[2], xd[g][q].z, s[q & 1]);
            s[q & 1] = dot2<DT>(P[3], xd[g][q].w, s[q & 1]);
        }
#endif
        pacc[it] = __builtin_fmaf(s[0] + s[1], am[it][g], pacc[it]);
    };
    if constexpr (GMAJOR) {
        // group by group, fenced: without the fence hipcc hoists every group's x permutes above the first dot2, i.e.
        // waits for almost the whole stream before it starts, and the load phase and the VALU phase of all the
        // (same-phase) waves of the chip serialise
#pragma unroll
        for (int g = 0; g < G; ++g) {
            permute_x(g);
#pragma unroll
            for (int it = 0; it < ITERS; ++it) consume(it, g);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int g = 0; g < G; ++g) consume(it, g);
        }
    }
    FP4_STAMP(2);
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        float p = pacc[it];
        // sum over the 32 lanes of this half-wave
        p = dpp_add<0x128>(p);
        p = dpp_add<0x124>(p);
        p = dpp_add<0x122>(p);
        p = dpp_add<0x121>(p);
        p += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, p), 0x401F));
        if constexpr (KSPLIT == 1) {
            const int row = row_base + rowi[it];
            if (mode & kModeSiluMulPairs) {
                // the two half-waves hold the gate row (even, lanes 0..31) and the up row (odd) of one pair; M is even
                const float up = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, p), 32));
                if (lane == 0 && row < M) store_silu_mul<DT>(out, bias, residual, row >> 1, p * (1.0f / 12.0f), up * (1.0f / 12.0f));
            } else if (l32 == 0 && row < M) {
                store_row<DT>(out, bias, residual, row, p * (1.0f / 12.0f), mode);
            }
        } else {
            if (l32 == 0) s_part[rowi[it]][kw] = p;
        }
    }
    if constexpr (KSPLIT > 1) {
        __syncthreads();
        if (mode & kModeSiluMulPairs) {
            if (tid < kRowsPerBlock / 2) {  // rows 2*tid (gate) and 2*tid + 1 (up) of this workgroup; row_base and M are even
                float g = 0.0f, u = 0.0f;
#pragma unroll
                for (int k = 0; k < KSPLIT; ++k) g += s_part[2 * tid][k], u += s_part[2 * tid + 1][k];
                const int row = row_base + 2 * tid;
                if (row < M) store_silu_mul<DT>(out, bias, residual, row >> 1, g * (1.0f / 12.0f), u * (1.0f / 12.0f));
            }
        } else if (tid < kRowsPerBlock) {
            float t = 0.0f;
#pragma unroll
            for (int k = 0; k < KSPLIT; ++k) t += s_part[tid][k];
            const int row = row_base + tid;
            if (row < M) store_row<DT>(out, bias, residual, row, t * (1.0f / 12.0f), mode);
        }
    }
    FP4_STAMP(3);
}

// ---- f32 activations: CODE_PARAM f32 table in LDS -----------------------------------------------
// LDS image of x: 4-float group g (0..7) of chunk c at 16-byte slot [g*C + c].
template <int ROWS, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void gemv32_kernel(const float *__restrict__ x, const uint8_t *__restrict__ W,
                                                            const float *__restrict__ absmax,
                                                            const float *__restrict__ bias, const float *residual, float *out,
                                                            int M, int K, int bs_shift) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_raw[];
    f32x4 *s_x4 = reinterpret_cast<f32x4 *>(s_raw);
    __shared__ float s_lut[16];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int C = K >> 5;

    if (tid < 16) s_lut[tid] = lut_entry(FP4_TABLE_CODEBOOK, tid);
    for (int p = tid; p < 8 * C; p += WAVES * 64) s_x4[(p & 7) * C + (p >> 3)] = reinterpret_cast<const f32x4 *>(x)[p];
    __syncthreads();

    const int row0 = (blockIdx.x * WAVES + wave) * ROWS;
    if (row0 >= M) return;
    int rows[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) rows[r] = (row0 + r < M) ? row0 + r : M - 1;

    const u32x4 *Wv = reinterpret_cast<const u32x4 *>(W);
    float acc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = 0.0f;

    for (int c = lane; c < C; c += 64) {
        u32x4 wq[ROWS];
        float am[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int64_t chunk = int64_t(rows[r]) * C + c;
            wq[r] = __builtin_nontemporal_load(Wv + chunk);
            am[r] = absmax[(chunk << 5) >> bs_shift];
        }
        float s[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) s[r] = 0.0f;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const f32x4 xv = s_x4[g * C + c];  // weights 4g..4g+3 of the chunk = bytes 2g, 2g+1
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                const uint32_t h = (wq[r][g >> 1] >> (16 * (g & 1))) & 0xFFFFu;  // two bytes
                s[r] = __builtin_fmaf(s_lut[(h >> 4) & 15u], xv.x, s[r]);
                s[r] = __builtin_fmaf(s_lut[h & 15u], xv.y, s[r]);
                s[r] = __builtin_fmaf(s_lut[(h >> 12) & 15u], xv.z, s[r]);
                s[r] = __builtin_fmaf(s_lut[(h >> 8) & 15u], xv.w, s[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) acc[r] = __builtin_fmaf(s[r], am[r], acc[r]);
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const float total = wave_sum(acc[r]);
        if (lane == 0 && row0 + r < M) {
            const float t = bias ? total + bias[row0 + r] : total;
            out[row0 + r] = residual ? t + residual[row0 + r] : t;
        }
    }
}

// ---- f32 activations, register-x geometry (same structure as gemv16_regx_kernel) -------------------------------
// x slice as 32 floats per chunk in VGPRs; weights decoded through the bit-faithful CODE_PARAM f32 table in LDS
// (16 consecutive dwords: every read is a conflict-free broadcast); (sum x*code) * absmax per chunk, f32 throughout.
// (A table-free variant - fp16 pairs of 12*code widened by v_fma_mix - was 10 % faster but outside parity bar 2 at f32; removed in round 3.)
template <int KSPLIT, int G, int ITERS>
__global__ __launch_bounds__(256) void gemv32_regx_kernel(const float *__restrict__ x, const uint8_t *__restrict__ W,
                                                          const float *__restrict__ absmax, const float *__restrict__ bias,
                                                          const float *residual, float *out, int M, int K, int bs_shift) {
    constexpr int RG = 4 / KSPLIT;
    constexpr int kRowsPerBlock = 2 * RG * ITERS;
    __shared__ float s_lut[16];
    __shared__ float s_part[kRowsPerBlock][KSPLIT];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int kw = wave % KSPLIT, rw = wave / KSPLIT;
    const int half = lane >> 5, l32 = lane & 31;
    const int C = K >> 5;
    const int row_base = blockIdx.x * kRowsPerBlock;
    const u32x4 *Wv = reinterpret_cast<const u32x4 *>(W);
    if (tid < 16) s_lut[tid] = lut_entry(FP4_TABLE_CODEBOOK, tid);

    int cidx[G];
    bool live[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int c = g * (32 * KSPLIT) + kw * 32 + l32;
        live[g] = c < C;
        cidx[g] = live[g] ? c : C - 1;
    }
    f32x4 xv[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int q = 0; q < 8; ++q) xv[g][q] = reinterpret_cast<const f32x4 *>(x)[cidx[g] * 8 + q];
    }
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
            const int64_t chunk = int64_t(rclamp) * C + cidx[g];
            wq[it][g] = __builtin_nontemporal_load(Wv + chunk);
            const float a = absmax[(chunk << 5) >> bs_shift];
            am[it][g] = live[g] ? a : 0.0f;
        }
    }
    __syncthreads();  // LUT visible
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        float p = 0.0f;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
            for (int q = 0; q < 8; ++q) {  // weights 4q..4q+3 of the chunk = bytes 2q, 2q+1
                const uint32_t h = (wq[it][g][q >> 1] >> (16 * (q & 1))) & 0xFFFFu;
                s0 = __builtin_fmaf(s_lut[(h >> 4) & 15u], xv[g][q].x, s0);
                s1 = __builtin_fmaf(s_lut[h & 15u], xv[g][q].y, s1);
                s0 = __builtin_fmaf(s_lut[(h >> 12) & 15u], xv[g][q].z, s0);
                s1 = __builtin_fmaf(s_lut[(h >> 8) & 15u], xv[g][q].w, s1);
            }
            p = __builtin_fmaf(s0 + s1, am[it][g], p);
        }
        p = dpp_add<0x128>(p);
        p = dpp_add<0x124>(p);
        p = dpp_add<0x122>(p);
        p = dpp_add<0x121>(p);
        p += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, p), 0x401F));
        if constexpr (KSPLIT == 1) {
            const int row = row_base + rowi[it];
            if (l32 == 0 && row < M) {
                const float t = bias ? p + bias[row] : p;
                out[row] = residual ? t + residual[row] : t;
            }
        } else {
            if (l32 == 0) s_part[rowi[it]][kw] = p;
        }
    }
    if constexpr (KSPLIT > 1) {
        __syncthreads();
        if (tid < kRowsPerBlock) {
            float t = 0.0f;
#pragma unroll
            for (int k = 0; k < KSPLIT; ++k) t += s_part[tid][k];
            const int row = row_base + tid;
            if (row < M) {
                t = bias ? t + bias[row] : t;
                out[row] = residual ? t + residual[row] : t;
            }
        }
    }
}

template <int KSPLIT, int G, int ITERS>
int launch32_regx(const void *x, const uint8_t *W, const float *absmax, const void *bias, const void *residual, void *out, int M,
                  int K, int bs_shift, hipStream_t stream) {
    constexpr int rows_per_block = 2 * (4 / KSPLIT) * ITERS;
    const unsigned blocks = (unsigned)((M + rows_per_block - 1) / rows_per_block);
    hipLaunchKernelGGL((gemv32_regx_kernel<KSPLIT, G, ITERS>), dim3(blocks), dim3(256), 0, stream,
                       reinterpret_cast<const float *>(x), W, absmax, reinterpret_cast<const float *>(bias),
                       reinterpret_cast<const float *>(residual), reinterpret_cast<float *>(out), M, K, bs_shift);
    return FP4_OK;
}

// returns -1 when K is too deep for a register-resident f32 x slice (K > 8192): use the LDS kernel
// (round 3: the table-free decode - 10 % faster, but outside parity bar 2 in f32 - and 8 row pairs per group are no longer built)
int dispatch32_regx(int iters, const void *x, const uint8_t *W, const float *absmax, const void *bias, const void *residual, void *out,
                    int M, int K, int bs_shift, hipStream_t stream) {
    const int C = K >> 5;
    const int ks = C <= 32 ? 1 : (C <= 64 ? 2 : 4);
    if (iters <= 0) {
        // the f32 x slice is 128 B per chunk per lane, re-read by every workgroup: amortise it over more rows than
        // the 16-bit kernel does, as long as >= ~512 workgroups remain
        // (measured: 4096x4096 5.1 us at 4 row pairs vs 9.3 us at 1; profiles/r01_d_*)
        iters = 1;
        while (iters < 4 && M / (2 * (4 / ks) * iters * 2) >= 256) iters *= 2;
    }
#define FP4_R32(KS, GG)                                                                                                   \
    switch (iters) {                                                                                                      \
        case 1: return launch32_regx<KS, GG, 1>(x, W, absmax, bias, residual, out, M, K, bs_shift, stream);        \
        case 2: return launch32_regx<KS, GG, 2>(x, W, absmax, bias, residual, out, M, K, bs_shift, stream);        \
        default: return launch32_regx<KS, GG, 4>(x, W, absmax, bias, residual, out, M, K, bs_shift, stream);       \
    }
    if (C <= 32) { FP4_R32(1, 1) }
    if (C <= 64) { FP4_R32(2, 1) }
    if (C <= 128) { FP4_R32(4, 1) }
    if (C <= 256) { FP4_R32(4, 2) }
#undef FP4_R32
    return -1;
}

// ---- generic: any even K, any even blocksize, no alignment assumptions ---------------------------
template <int DT>
__global__ __launch_bounds__(256) void gemv_generic_kernel(const void *__restrict__ xv, const uint8_t *__restrict__ W,
                                                           const float *__restrict__ absmax, const void *__restrict__ biasv,
                                                           const void *residualv, void *outv, int M, int64_t K, int blocksize,
                                                           CodeTable tbl, int mode) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int64_t e_row = int64_t(row) * K;
    float acc = 0.0f;
    for (int64_t b = lane; b < K / 2; b += 64) {
        const int64_t e = e_row + 2 * b;
        const uint32_t q = W[e >> 1];
        const float am = absmax[e / blocksize];
        float x0, x1;
        if constexpr (DT == FP4_DTYPE_F32) {
            x0 = reinterpret_cast<const float *>(xv)[2 * b];
            x1 = reinterpret_cast<const float *>(xv)[2 * b + 1];
        } else {
            x0 = to_f32<DT>(reinterpret_cast<const uint16_t *>(xv)[2 * b]);
            x1 = to_f32<DT>(reinterpret_cast<const uint16_t *>(xv)[2 * b + 1]);
        }
        acc = __builtin_fmaf(__builtin_bit_cast(float, tbl.bits[q >> 4]) * am, x0, acc);
        acc = __builtin_fmaf(__builtin_bit_cast(float, tbl.bits[q & 15u]) * am, x1, acc);
    }
    const float total = wave_sum(acc);
    if (lane == 0) {
        if constexpr (DT == FP4_DTYPE_F32) {
            const float *bias = reinterpret_cast<const float *>(biasv), *residual = reinterpret_cast<const float *>(residualv);
            const float t = bias ? total + bias[row] : total;
            reinterpret_cast<float *>(outv)[row] = residual ? t + residual[row] : t;
        } else {
            store_row<DT>(reinterpret_cast<uint16_t *>(outv), reinterpret_cast<const uint16_t *>(biasv),
                          reinterpret_cast<const uint16_t *>(residualv), row, total, mode);
        }
    }
}

// LDS geometry: ROWS | WAVES << 8 | UNROLL << 16; register-x geometry: 1 << 24 | bands (0, 5, 6, 7, 8) << 8 | ITERS; -1 = heuristic.
// Sweep hook only (fp4_hip_set_variant); relaxed atomic so that a server thread launching while a sweep flips it is not a data race.
std::atomic<int> g_gemv_variant{-1};

constexpr int kMaxLdsBytes = 160 * 1024 - 256;

template <typename Kern>
int ensure_lds(Kern kern, size_t lds_bytes) {
    if (lds_bytes <= 64 * 1024) return FP4_OK;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_bytes) != hipSuccess) {
        (void)hipGetLastError();
        set_error("fp4_hip_gemv: cannot reserve %zu bytes of LDS", lds_bytes);
        return FP4_ERR_UNSUPPORTED;
    }
    return FP4_OK;
}

// one 16-bit GEMV launch: operands, epilogue inputs and the mode flags of gemv_common.h
struct GemvArgs {
    const void *x;
    const uint8_t *W;
    const float *absmax;
    const void *bias, *residual;
    void *out;
    int M, K, bs_shift, mode;
    hipStream_t stream;
};

template <int DT, int ROWS, int WAVES, int UNROLL>
int launch16(const GemvArgs &a) {
    if (a.mode & kModeSiluMulPairs) return -2;  // a pair's rows sit in different waves here
    auto kern = gemv16_kernel<DT, ROWS, WAVES, UNROLL>;
    const int M = a.M, K = a.K;
    const size_t lds = size_t(K) * 2;
    if (int rc = ensure_lds(kern, lds)) return rc;
    const int rows_per_block = ROWS * WAVES;
    const unsigned blocks = (unsigned)((M + rows_per_block - 1) / rows_per_block);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(WAVES * 64), lds, a.stream, reinterpret_cast<const uint16_t *>(a.x), a.W, a.absmax,
                       reinterpret_cast<const uint16_t *>(a.bias), reinterpret_cast<const uint16_t *>(a.residual),
                       reinterpret_cast<uint16_t *>(a.out), M, K, a.bs_shift, a.mode);
    return FP4_OK;
}

template <int DT>
int dispatch16(int variant, const GemvArgs &a) {
    switch (variant) {
#define FP4_V(R, Wv, U)                  \
    case (R | (Wv << 8) | (U << 16)):    \
        return launch16<DT, R, Wv, U>(a);
        // (1, 8, 2) is what K > 16384 falls back to; (1, 4, 2) is the north-star mapping at the reference's 4 rows per block.
        // Round 3 removed (1,4,1) (2,4,2) (2,8,2) (1,16,2): sweep-only, behind the register-x geometry at every measured shape
        // (profiles/r01_b_exp_gemv_geometries.txt, profiles/r03_kernel_inventory.txt)
        FP4_V(1, 4, 2) FP4_V(1, 8, 2)
#undef FP4_V
        default:
            set_error("fp4_hip_gemv: unknown kernel variant 0x%x", variant);
            return FP4_ERR_INVALID_ARGUMENT;
    }
}

constexpr int kRegxFlag = 1 << 24;

template <int DT, int KSPLIT, int G, int ITERS, int WAVES = 4>
int launch_regx(const GemvArgs &a) {
    constexpr int rows_per_block = 2 * (WAVES / KSPLIT) * ITERS;
    const unsigned blocks = (unsigned)((a.M + rows_per_block - 1) / rows_per_block);
    hipLaunchKernelGGL((gemv16_regx_kernel<DT, KSPLIT, G, ITERS, WAVES>), dim3(blocks), dim3(WAVES * 64), 0, a.stream,
                       reinterpret_cast<const uint16_t *>(a.x), a.W, a.absmax, reinterpret_cast<const uint16_t *>(a.bias),
                       reinterpret_cast<const uint16_t *>(a.residual), reinterpret_cast<uint16_t *>(a.out), a.M, a.K, a.bs_shift,
                       a.mode);
    return FP4_OK;
}

// K decides the band split and the x-slice depth; ITERS (row pairs per group) is the tunable.  `ks_override` selects the
// 5 / 6 / 7 / 8-band geometries for the row lengths they were built for; G follows from ceil(C / (32 * KSPLIT)).
template <int DT>
int dispatch_regx(int iters, int ks_override, const GemvArgs &a) {
    const int C = a.K >> 5;
    if (int64_t(a.M) * a.K >= (int64_t(1) << 32)) return -1;  // 32-bit buffer offsets; the LDS geometry addresses with 64 bits
    // K = 7 * 1024 * {1, 2, 4} (7168, 14336, 28672 - the Llama-3 / Mistral intermediate sizes): seven waves split the row into
    // seven bands, so no lane of a 32-chunk band is idle (4 bands leave an eighth of the deepest slice dead at 14336) and
    // 28672 still fits the register-resident x slice
    // Only the (depth, row pairs) the heuristic picks are built (round 3: the other ITERS lost every sweep,
    // profiles/r01_f_gemv_seven_bands.txt, profiles/r02_gemv_regx_sweep.txt): 7168 -> 2 row pairs, 14336 -> 4, 28672 -> 2.
    if (ks_override == 7 && C % 224 == 0 && (C / 224 == 1 || C / 224 == 2 || C / 224 == 4)) {
        const int g7 = C / 224;
        if (g7 == 1) return launch_regx<DT, 7, 1, 2, 7>(a);
        if (g7 == 2) return launch_regx<DT, 7, 2, 4, 7>(a);
        return launch_regx<DT, 7, 4, 2, 7>(a);
    }
    // five / six bands for rows of exactly 5 or 6 (x 1, 2) band widths: K = 5120 / 10240 (Llama-2-13B hidden size) and 6144 / 12288
    if ((ks_override == 5 && C % 160 == 0 && C / 160 <= 2) || (ks_override == 6 && C == 192)) {
        const int gg = C / (32 * ks_override);
        if (iters > 4) iters = 4;
#define FP4_RXN(KS, GG)                                                                                               \
    switch (iters) {                                                                                                  \
        case 1: return launch_regx<DT, KS, GG, 1, KS>(a);      \
        case 2: return launch_regx<DT, KS, GG, 2, KS>(a);      \
        default: return launch_regx<DT, KS, GG, 4, KS>(a);     \
    }
        if (ks_override == 5 && gg == 1) { FP4_RXN(5, 1) }
        if (ks_override == 5 && gg == 2) { FP4_RXN(5, 2) }
        if (ks_override == 6 && gg == 1) { FP4_RXN(6, 1) }
#undef FP4_RXN
    }
    if (ks_override == 8 && C == 256) {  // K = 8192 split 8 ways over 8 waves (the heuristic's choice from 4096 rows up)
        if (iters >= 4) return launch_regx<DT, 8, 1, 4, 8>(a);
        return launch_regx<DT, 8, 1, 2, 8>(a);
    }
    const int ks = C <= 32 ? 1 : (C <= 64 ? 2 : 4);
    const int need = (C + 32 * ks - 1) / (32 * ks);
    // (a 3-deep slice only with the 4-way split: K = 11008, the Llama-2 down-projection, would waste a third of a 4-deep one)
    const int g = need <= 1 ? 1 : (need <= 2 ? 2 : ((need == 3 && ks == 4) ? 3 : (need <= 4 ? 4 : 0)));
    if (g == 0) return -1;  // the x slice no longer fits the register budget; use the LDS geometry
    if (iters > 4) iters = 4;
    if (g >= 3 && iters > 2) iters = 2;
    if (iters == 3) iters = 2;
    if (iters < 1) iters = 1;
#define FP4_RX(KS, GG, IT) return launch_regx<DT, KS, GG, IT>(a)
    // only the (KSPLIT, G, ITERS) triples the heuristic can reach are instantiated: the split follows K, G = 1, 2 -> 1, 2, 4 row pairs;
    // G >= 3 -> 1, 2 (round 3 removed 8 row pairs and the forced splits with deep slices: sweep-only, never ahead)
#define FP4_RX_IT(KS, GG)                                    \
    if (ks == KS && g == GG) {                               \
        switch (iters) {                                     \
            case 1: FP4_RX(KS, GG, 1);                       \
            case 2: FP4_RX(KS, GG, 2);                       \
            case 4:                                          \
                if constexpr ((GG) <= 2) { FP4_RX(KS, GG, 4); } \
                break;                                       \
            default: break;                                  \
        }                                                    \
    }
    FP4_RX_IT(1, 1) FP4_RX_IT(2, 1) FP4_RX_IT(4, 1) FP4_RX_IT(4, 2) FP4_RX_IT(4, 3) FP4_RX_IT(4, 4)
#undef FP4_RX_IT
#undef FP4_RX
    set_error("fp4_hip_gemv: unknown regx geometry (iters %d, ksplit %d, g %d)", iters, ks, g);
    return FP4_ERR_INVALID_ARGUMENT;
}

int default_variant16(int M, int K) {
    // Measured on MI355X (profiles/r01_*): the register-x geometry wins at every decode shape it covers;
    // rows per workgroup grow with M so that the grid stays at >= ~1024 workgroups (one resident round at
    // 4096 rows, ~2 at 14336).  K > 16384 (other than 28672) falls back to the LDS geometry inside dispatch.
    const int C = K >> 5;
    // seven bands (dispatch_regx): measured ahead of the 4-band / LDS geometries at 4096 x 7168 (5.3 vs 6.1 us), 4096 x 14336
    // (8.6 vs 8.9) and 8192 x 28672 (30.7 vs 31.8), behind at 5120 x 14336 (profiles/r01_f_gemv_seven_bands.txt)
    // five / six bands where the row is exactly that many band widths (x 1 or 2): K = 5120 (5.1 vs 6.6 us at 5120 x 5120,
    // 9.5 vs 12.0 at 13824 x 5120), 10240, 6144 (6.4 vs 7.5 at 6144 x 6144) - profiles/r01_f_gemv_five_six_bands.txt
    if (C == 160 || C == 320 || C == 192) {
        int it = 1;
        while (it * 2 <= 4 && M / (2 * it * 2) >= 1024) it *= 2;
        return kRegxFlag | ((C == 192 ? 6 : 5) << 8) | it;
    }
    // eight bands for K = 8192 on 4096 rows and more (8.9 vs 9.4 us at 8192 x 8192, 23.0 vs 24.0 at 28672 x 8192)
    if (C == 256 && M >= 4096) {
        int it = 1;
        while (it * 2 <= 4 && M / (2 * it * 2) >= 1024) it *= 2;
        return kRegxFlag | (8 << 8) | it;
    }
    if (C == 224) return kRegxFlag | (7 << 8) | 2;
    if (C == 448 && M <= 4096) return kRegxFlag | (7 << 8) | 4;
    if (C == 896) return kRegxFlag | (7 << 8) | 2;
    const int ksplit = C <= 32 ? 1 : (C <= 64 ? 2 : 4);
    const int rows_per_iter = 2 * (4 / ksplit);
    const int max_iters = C <= 128 ? 4 : (C <= 256 ? 4 : 2);
    // two row pairs per group once that still leaves >= 1024 workgroups, four only from 2048 workgroups up (round 3 sweep over M at
    // K = 4096 / 2048 / 1024, profiles/r03_gemv_rows_per_workgroup.txt: 8192..14336 x 4096 0.5-2.5 % faster at two, 32768 x 1024 5 %;
    // from 16384 x 4096 four is ahead)
    int iters = 1;
    if (max_iters >= 2 && M / (rows_per_iter * 2) >= 1024) iters = 2;
    if (max_iters >= 4 && M / (rows_per_iter * 4) >= 2048) iters = 4;
    return kRegxFlag | iters;
}

template <int DT>
int run_generic(const void *x, const uint8_t *W, const float *absmax, const void *bias, const void *residual, void *out, int64_t M,
                int64_t K, int blocksize, int mode, hipStream_t stream) {
    const CodeTable tbl = make_table(FP4_TABLE_CODEBOOK);
    hipLaunchKernelGGL((gemv_generic_kernel<DT>), dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream, x, W, absmax, bias,
                       residual, out, (int)M, K, blocksize, tbl, mode);
    return FP4_OK;
}

}  // namespace

void set_gemv_variant(int v) { g_gemv_variant.store(v, std::memory_order_relaxed); }

}  // namespace fp4

namespace fp4 {
namespace {
int gemv_entry(const void *x, const uint8_t *packed, const float *absmax, const void *bias, const void *residual, void *out,
               int64_t M, int64_t K, int blocksize, int dtype, int mode, void *stream) {
    if (M < 0 || K < 0 || (K & 1) || blocksize < 2 || (blocksize & 1)) {
        set_error("fp4_hip_gemv: M=%lld K=%lld blocksize=%d (need M,K >= 0, even K, even blocksize >= 2)", (long long)M,
                  (long long)K, blocksize);
        return FP4_ERR_INVALID_ARGUMENT;
    }
    if (dtype != FP4_DTYPE_F16 && dtype != FP4_DTYPE_BF16 && dtype != FP4_DTYPE_F32) {
        // reference: std::runtime_error("Unsupported datatype") (csrc/gemv_fp4_optimized.cu:362-363)
        set_error("fp4_hip_gemv: unsupported dtype %d", dtype);
        return FP4_ERR_UNSUPPORTED;
    }
    if ((mode & kModeSiluMulPairs) && (M & 1)) {
        set_error("fp4_hip_gemv_fused: the gate|up epilogue needs an even row count, got M=%lld", (long long)M);
        return FP4_ERR_INVALID_ARGUMENT;
    }
    if (M == 0) return FP4_OK;
    if (!out || (K > 0 && (!x || !packed || !absmax))) {
        set_error("fp4_hip_gemv: null pointer");
        return FP4_ERR_INVALID_ARGUMENT;
    }
    if (M > (int64_t(1) << 30) || K > (int64_t(1) << 30)) {
        set_error("fp4_hip_gemv: M=%lld K=%lld too large", (long long)M, (long long)K);
        return FP4_ERR_UNSUPPORTED;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int bs_shift = ilog2_exact(blocksize);
    const uintptr_t align = reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(x);
    const size_t esz = dtype == FP4_DTYPE_F32 ? 4 : 2;
    const bool fast = K > 0 && (K % 32) == 0 && bs_shift >= 5 && (K % blocksize) == 0 && (align & 15u) == 0 &&
                      size_t(K) * esz <= size_t(kMaxLdsBytes);
    int rc = FP4_OK;
    const int gv = g_gemv_variant.load(std::memory_order_relaxed);  // one snapshot per call
    if (fast && dtype != FP4_DTYPE_F32) {
        const GemvArgs a{x, packed, absmax, bias, residual, out, (int)M, (int)K, bs_shift, mode, s};
        int variant = gv >= 0 ? gv : default_variant16((int)M, (int)K);
        rc = -1;
        if (variant & kRegxFlag) {
            const int iters = variant & 0xFF, ks_override = (variant >> 8) & 0xF;
            rc = dtype == FP4_DTYPE_F16 ? dispatch_regx<FP4_DTYPE_F16>(iters, ks_override, a)
                                        : dispatch_regx<FP4_DTYPE_BF16>(iters, ks_override, a);
            if (rc == -1) variant = 1 | (8 << 8) | (2 << 16);  // K too large for register-resident x
        }
        if (rc == -1) rc = dtype == FP4_DTYPE_F16 ? dispatch16<FP4_DTYPE_F16>(variant, a) : dispatch16<FP4_DTYPE_BF16>(variant, a);
    } else if (mode & kModeSiluMulPairs) {
        rc = -2;  // f32 activations / irregular shapes: only the register-x geometry pairs rows up
    } else if (fast && gv != 0 &&
               // the bit-faithful CODE_PARAM f32 table (the all-f32 reference kernel is accurate to ~1e-7, so the table's 1e-6
               // deviations from k/12 are visible at f32)
               dispatch32_regx(gv < 0 ? 0 : (gv & 0xFF), x, packed, absmax, bias, residual, out, (int)M, (int)K, bs_shift, s) == FP4_OK) {
        rc = FP4_OK;  // f32 activations, register-x geometry (variant 0 forces the LDS kernel below, for sweeps)
    } else if (fast) {
        auto kern = gemv32_kernel<1, 4>;
        const size_t lds = size_t(K) * 4;
        rc = ensure_lds(kern, lds);
        if (rc == FP4_OK)
            hipLaunchKernelGGL(kern, dim3((unsigned)((M + 3) / 4)), dim3(256), lds, s, reinterpret_cast<const float *>(x),
                               packed, absmax, reinterpret_cast<const float *>(bias), reinterpret_cast<const float *>(residual),
                               reinterpret_cast<float *>(out), (int)M, (int)K, bs_shift);
    } else {
        switch (dtype) {
            case FP4_DTYPE_F16:
                rc = run_generic<FP4_DTYPE_F16>(x, packed, absmax, bias, residual, out, M, K, blocksize, mode, s);
                break;
            case FP4_DTYPE_BF16:
                rc = run_generic<FP4_DTYPE_BF16>(x, packed, absmax, bias, residual, out, M, K, blocksize, mode, s);
                break;
            default:
                rc = run_generic<FP4_DTYPE_F32>(x, packed, absmax, bias, residual, out, M, K, blocksize, mode, s);
                break;
        }
    }
    if (rc == -2) {
        set_error("fp4_hip_gemv_fused: the gate|up epilogue is not available for M=%lld K=%lld blocksize=%d dtype=%d "
                  "(needs a 16-bit dtype, K %% 32 == 0, K <= 16384 or 28672, a power-of-two blocksize >= 32 that divides K); run the plain GEMV and "
                  "apply the activation separately", (long long)M, (long long)K, blocksize, dtype);
        return FP4_ERR_UNSUPPORTED;
    }
    if (rc != FP4_OK) return rc;
    return check_launch("fp4_hip_gemv");
}
}  // namespace
}  // namespace fp4

extern "C" int fp4_hip_gemv(const void *x, const uint8_t *packed, const float *absmax, const void *bias, void *out,
                            int64_t M, int64_t K, int blocksize, int dtype, void *stream) {
    return fp4::gemv_entry(x, packed, absmax, bias, nullptr, out, M, K, blocksize, dtype, 0, stream);
}

extern "C" int fp4_hip_gemv_fused(const void *x, const uint8_t *packed, const float *absmax, const void *bias, const void *residual,
                                  void *out, int64_t M, int64_t K, int blocksize, int dtype, int epilogue, void *stream) {
    if (epilogue != FP4_EPILOGUE_NONE && epilogue != FP4_EPILOGUE_SILU_MUL_PAIRS) {
        fp4::set_error("fp4_hip_gemv_fused: unknown epilogue %d", epilogue);
        return FP4_ERR_INVALID_ARGUMENT;
    }
    return fp4::gemv_entry(x, packed, absmax, bias, residual, out, M, K, blocksize, dtype,
                           epilogue == FP4_EPILOGUE_SILU_MUL_PAIRS ? fp4::kModeSiluMulPairs : 0, stream);
}

extern "C" int fp4_hip_gemv_partial(const void *x, const uint8_t *packed, const float *absmax, float *out_f32, int64_t M,
                                    int64_t K, int blocksize, int x_dtype, void *stream) {
    return fp4::gemv_entry(x, packed, absmax, nullptr, nullptr, out_f32, M, K, blocksize, x_dtype, fp4::kModeOutF32, stream);
}
