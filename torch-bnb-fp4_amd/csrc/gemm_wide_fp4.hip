// 17..64 activation rows against one FP4 weight in ONE pass over the weight (gfx950 / MI355X):
//     out[b][r] = T( sum_k x[b][k] * code[nib(r,k)] * absmax[(r*K+k)/64] + bias[r] )
// The reference sends every batch > 1 through a full dequant + dense GEMM (torch_bnb_fp4/__init__.py:423-436,616-617): at
// 14336 x 4096 that is 150 MB written and read back for 33 MB of weight.  gemm_small_fp4.hip covers up to 16 rows per launch
// (one 16-column matrix-core tile, x in registers); above that it used to stream the weight once per 16 rows.  Here one launch
// multiplies every decoded weight fragment with NT = 2..4 column tiles of x (NT = 1 as well: 1..16 rows on K % 512 != 0, which
// the kernels of gemm_small_fp4.hip do not take):
//   * the compute waves of a workgroup are WR row groups x WK K slices; a wave owns RT 16-row tiles and every WK-th quant block;
//   * per step (one 64-column quant block per K slice) the x operand - 16*NT columns x 128 B, L2-resident, by far the larger
//     on-chip stream - goes into LDS in full 128-byte lines by LDS-DMA (global_load_lds_dwordx4, no VGPR staging), 2..4 ring
//     slots per K slice; the image is lane-linear, so the bank swizzle (16-byte unit ^ (column >> 1)) is applied to the SOURCE
//     address and again to the ds_read_b128 address;
//   * the weight and its block scales come by LDS-DMA as well, in full lines, into a ring several steps deep (see the three
//     kernels for who issues what); a first version loaded them fragment-shaped into registers one step ahead and spent more time on that
//     than on everything else (profiles/r02_wide_batch_17_to_128_rows.txt);
//   * waits are counted (s_waitcnt vmcnt(N) in front of a bare s_barrier), so younger steps stay in flight across the barrier;
//   * x is the A operand and the weight the B operand, so a lane's accumulators all belong to ONE weight row and the block scale is
//     one scalar per lane;
//   * decode8's (e0,e2)(e4,e6)(e1,e3)(e5,e7) pairs are put back into natural order (4 v_perm per 8 weights, amortised over the
//     NT column tiles), so the x image needs no re-pairing;
//   * any K % 64 == 0: in a ragged last step the K slices past the row's end skip their work, and DMA lanes past it re-read a
//     valid piece;
//   * K-slice partials meet in LDS (the ring's storage, after the loop) and are summed in a fixed order: deterministic.
#include <atomic>

#include "gemv_common.h"

namespace fp4 {

namespace {

typedef __bf16 bf16x8w_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8w_t __attribute__((ext_vector_type(8)));

template <int DT>
__device__ __forceinline__ f32x4 mfma_xw(u32x4 xfrag, u32x4 wfrag, f32x4 c) {
    if constexpr (DT == FP4_DTYPE_F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8w_t, xfrag), __builtin_bit_cast(f16x8w_t, wfrag), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8w_t, xfrag), __builtin_bit_cast(bf16x8w_t, wfrag), c, 0, 0, 0);
}

// 8 weights of one packed dword as 12*code in natural order: (e0,e1) (e2,e3) (e4,e5) (e6,e7)
template <int DT>
__device__ __forceinline__ u32x4 decode8_natural(uint32_t q) {
    uint32_t P[4];
    decode8<DT>(q, P);
    u32x4 n;
    n.x = perm(P[2], P[0], 0x05040100u);
    n.y = perm(P[2], P[0], 0x07060302u);
    n.z = perm(P[3], P[1], 0x05040100u);
    n.w = perm(P[3], P[1], 0x07060302u);
    return n;
}

__device__ __forceinline__ void lds_dma16(const uint8_t *src, uint8_t *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// K-slice partials -> LDS (the ring's storage; the caller has synchronised) -> fixed-order sum -> store.
// D layout: lane (j = lane & 15 -> weight row of the tile, lane >> 4) register g -> activation row nt*16 + (lane >> 4)*4 + g
template <int DT, int NT, int RT, int WR, int NTHREADS = 512>
__device__ __forceinline__ void wide_epilogue(uint8_t *s_raw, const f32x4 (&acc)[RT][NT], const uint16_t *bias, const uint16_t *residual,
                                              uint16_t *out, int B, int M, int row0, int mode) {
    constexpr int WK = 8 / WR, kTiles = WR * RT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave % WR, wk = wave / WR;
    float *s_part = reinterpret_cast<float *>(s_raw);  // [WK][kTiles][NT][256]
    if (NTHREADS == 512 || wave < 8) {  // (waves beyond the eighth only load)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                *reinterpret_cast<f32x4 *>(s_part + ((wk * kTiles + wr * RT + rt) * NT + nt) * 256 + lane * 4) = acc[rt][nt];
    }
    __syncthreads();
    const bool pairs = (mode & kModeSiluMulPairs) != 0;
    for (int o = tid; o < kTiles * NT * 256; o += NTHREADS) {
        // 16 consecutive threads store 16 consecutive weight rows of one activation row
        const int j = o & 15, nl = (o >> 4) & 15, nt = (o >> 8) % NT, tl = o / (256 * NT);
        const int e = (((nl >> 2) * 16 + j) << 2) + (nl & 3);
        const int n = nt * 16 + nl, row = row0 + tl * 16 + j;
        if (row >= M || n >= B) continue;
        const float *src = s_part + (tl * NT + nt) * 256 + e;
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < WK; ++w) t += src[w * kTiles * NT * 256];
        if (pairs) {  // rows (2p, 2p+1) = (gate, up): neighbouring lanes j, j+1 -> elements e, e+4
            if (j & 1) continue;
            float u = 0.0f;
#pragma unroll
            for (int w = 0; w < WK; ++w) u += src[w * kTiles * NT * 256 + 4];
            store_small_silu_mul<DT>(out, bias, residual, n, row >> 1, M >> 1, t * (1.0f / 12.0f), u * (1.0f / 12.0f));
        } else {
            store_small<DT>(out, bias, residual, n, row, M, t * (1.0f / 12.0f));
        }
    }
}

// ---- tall weights: 32*RT rows per workgroup -------------------------------------------------------------------------------------
// Fragment-shaped weight loads (8 bytes per lane from 16 different rows) keep the texture path busy for four cache-line lookups
// per quad, so the weight comes by LDS-DMA in full lines too: WR = 2, WK = 4, and per step the workgroup's 32*RT rows x 128 B of
// weight (the four K slices' blocks are one full line per row) and their 32*RT x 4 scales.
//   * ONE ring of D = 3..4 steps for x, weight and scales; every wave issues a share of each for step s + D - 1, so all eight waves
//     carry about the same number of LDS-DMA issues per step (a first form had waves 0..3 fetch x and waves 4..7 the weight into a
//     deeper ring of its own: 4-8 % slower - a step here is long enough for D - 1 steps to cover the HBM latency);
//   * a counted s_waitcnt vmcnt leaves D - 2 steps in flight across the barrier, which is the bare s_barrier (a __syncthreads()
//     fence would drain every wave's counter);
//   * weight image: [row][128 B] with 16-byte piece ^ (row >> 1 & 7), applied to the DMA's source and to the ds_read_b64.

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int DT, int NT, int RT>
__global__ __launch_bounds__(512) void gemm16_wide_ring_kernel(const uint16_t *__restrict__ x, const uint8_t *__restrict__ W,
                                                               const float *__restrict__ absmax, const uint16_t *__restrict__ bias,
                                                               const uint16_t *residual, uint16_t *out, int B, int M, int K, int mode) {
    constexpr int WR = 2, WK = 4, kTiles = WR * RT, kRows = 16 * kTiles;
    // ONE ring depth D for x and weight: every wave issues a share of both for step s + D - 1 (x: the K slice wave & 3, every second
    // 8-column piece; weight: 8 rows x 128 B; waves 4..7 also 16 (8) rows of scales)
    constexpr int kSlot = NT * 2048;
    constexpr int kWSlot = kRows * 128, kSSlot = kRows * 16;
    constexpr int kPerDepth = WK * kSlot + kWSlot + kSSlot;
    constexpr int D = (4 * kPerDepth <= 150 * 1024) ? 4 : 3;
    constexpr int kXRing = D * WK * kSlot;
    constexpr int kPart = WK * kTiles * NT * 1024;  // (the partials reuse the rings' storage after the loop)
    constexpr int kRings = kXRing + D * (kWSlot + kSSlot);
    static_assert((kRings > kPart ? kRings : kPart) <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(1024))) uint8_t s_raw[kRings > kPart ? kRings : kPart];
    uint8_t *s_w = s_raw + kXRing, *s_s = s_w + D * kWSlot;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave % WR, wk = wave / WR;
    const int i = lane & 15, kb = lane >> 4;
    const int row0 = blockIdx.x * kRows;
    const int nblk = K >> 6, steps = (nblk + WK - 1) / WK;  // K % 256 != 0: ragged last step, handled as in the 16-row kernel below

    const bool upper = wave >= 4;  // wave-uniform
    const int xs_slice = wave & 3;
    // x: waves w and w + 4 share K slice w: piece d = 2e + (wave >> 2) -> columns 8d .. 8d+7
    uint32_t xoff[NT];
#pragma unroll
    for (int e = 0; e < NT; ++e) {
        const int d = 2 * e + (wave >> 2);
        const int n = 8 * d + (lane >> 3), sl = lane & 7;
        const int nn = n < B ? n : B - 1;
        xoff[e] = (uint32_t)nn * (uint32_t)K * 2u + (uint32_t)((sl ^ ((n >> 1) & 7)) * 16);
    }
    const uint8_t *xb = reinterpret_cast<const uint8_t *>(x);
    // weight: 8 rows x 128 B per DMA; RT = 4: two per wave, RT = 2: one per wave, RT = 1: one per wave 0..3; lane -> (row, piece)
    constexpr int kWD = RT == 4 ? 2 : 1;
    constexpr bool kAllWavesLoadW = RT >= 2;
    const uint8_t *wsrc[kWD];
    int wpiece[kWD];
#pragma unroll
    for (int d = 0; d < kWD; ++d) {
        const int wrl = 8 * ((kAllWavesLoadW ? wave : (wave & 3)) * kWD + d) + (lane >> 3);
        wpiece[d] = (lane & 7) ^ ((wrl >> 1) & 7);
        wsrc[d] = W + (int64_t)(row0 + wrl < M ? row0 + wrl : M - 1) * (int64_t)(K >> 1);
    }
    // scales: waves 4..7, 16 rows x 4 scales per DMA (RT = 4: two, RT = 1: 8 rows with lanes 0..31); lane -> (row, K slice)
    constexpr int kSRows = kRows / 4, kSD = RT == 4 ? 2 : 1;
    const float *ssrc[kSD];
#pragma unroll
    for (int d = 0; d < kSD; ++d) {
        const int srow = row0 + (wave & 3) * kSRows + d * 16 + (lane >> 2);
        ssrc[d] = absmax + (int64_t)(srow < M ? srow : M - 1) * nblk;
    }
    const bool slane = (lane >> 2) < kSRows;

    // fragment reads
    const int xrd0 = i * 128 + (((2 * kb) ^ (i >> 1)) * 16), xrd1 = i * 128 + (((2 * kb + 1) ^ (i >> 1)) * 16);
    int wrd[RT], srd[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int rl = (wr * RT + rt) * 16 + i;
        wrd[rt] = rl * 128 + (((2 * wk + (kb >> 1)) ^ ((rl >> 1) & 7)) * 16) + (kb & 1) * 8;
        srd[rt] = rl * 16 + wk * 4;
    }

    f32x4 acc[RT][NT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[rt][nt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    // DMAs of this wave per step: x NT (none where the slice has no block: ragged last step), weight kWD (RT = 1: waves 0..3 only),
    // scales kSD (waves 4..7)
    constexpr int kGroupLo = NT + kWD, kGroupHi = NT + (kAllWavesLoadW ? kWD : 0) + kSD;
    auto issue = [&](int s) {
        const int ring = s % D, left = nblk - s * WK;
        if (s * WK + xs_slice < nblk) {  // wave-uniform
            uint8_t *slot = s_raw + (ring * WK + xs_slice) * kSlot;
#pragma unroll
            for (int e = 0; e < NT; ++e)
                lds_dma16(xb + xoff[e] + (uint32_t)(s * WK + xs_slice) * 128u, slot + (2 * e + (wave >> 2)) * 1024);
        }
        if (kAllWavesLoadW || !upper) {
#pragma unroll
            for (int d = 0; d < kWD; ++d)
                lds_dma16(wsrc[d] + s * 128 + ((wpiece[d] >> 1) < left ? wpiece[d] * 16 : 0),
                          s_w + ring * kWSlot + ((kAllWavesLoadW ? wave : (wave & 3)) * kWD + d) * 1024);
        }
        if (upper && slane) {
#pragma unroll
            for (int d = 0; d < kSD; ++d)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(ssrc[d] + s * 4 + ((lane & 3) < left ? (lane & 3) : 0)),
                    (__attribute__((address_space(3))) void *)(s_s + ring * kSSlot + ((wave & 3) * kSRows + d * 16) * 16), 4, 0, 0);
        }
    };
    for (int s = 0; s < D - 1 && s < steps; ++s) issue(s);
    for (int s = 0; s < steps; ++s) {
        // step s has landed; D - 2 younger steps may stay in flight (not near the end, where a ragged step issues fewer DMAs)
        if (s + D - 2 < steps - 1) {
            if (upper)
                wait_vmcnt<(D - 2) * kGroupHi>();
            else
                wait_vmcnt<(D - 2) * kGroupLo>();
        } else {
            wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + D - 1 < steps) issue(s + D - 1);  // into the slots step s - 1 used
        if (s * WK + wk >= nblk) continue;  // wave-uniform (ragged last step): no block for this K slice
        const uint8_t *xs = s_raw + ((s % D) * WK + wk) * kSlot;
        const uint8_t *ws = s_w + (s % D) * kWSlot;
        const uint8_t *ss = s_s + (s % D) * kSSlot;
        u32x2 wq[RT];
        float am[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            wq[rt] = *reinterpret_cast<const u32x2 *>(ws + wrd[rt]);
            am[rt] = *reinterpret_cast<const float *>(ss + srd[rt]);
        }
        f32x4 tile[RT][NT];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            u32x4 wf[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) wf[rt] = decode8_natural<DT>(t == 0 ? wq[rt].x : wq[rt].y);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const u32x4 xf = *reinterpret_cast<const u32x4 *>(xs + nt * 2048 + (t == 0 ? xrd0 : xrd1));
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    tile[rt][nt] = mfma_xw<DT>(xf, wf[rt], t == 0 ? f32x4{0.0f, 0.0f, 0.0f, 0.0f} : tile[rt][nt]);
            }
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                acc[rt][nt].x = __builtin_fmaf(tile[rt][nt].x, am[rt], acc[rt][nt].x);
                acc[rt][nt].y = __builtin_fmaf(tile[rt][nt].y, am[rt], acc[rt][nt].y);
                acc[rt][nt].z = __builtin_fmaf(tile[rt][nt].z, am[rt], acc[rt][nt].z);
                acc[rt][nt].w = __builtin_fmaf(tile[rt][nt].w, am[rt], acc[rt][nt].w);
            }
    }
    __syncthreads();  // every wave is done with the rings before the partials overwrite the x ring's storage
    wide_epilogue<DT, NT, RT, WR>(s_raw, acc, bias, residual, out, B, M, row0, mode);
}

// ---- short weights: 16 rows per workgroup, K split 8 ways, two DEDICATED loader waves ---------------------------------------------
// With 8 K slices every compute wave already fetches its own x slice; the weight stream gets two extra waves (8 and 9) that do
// nothing else: per step 16 rows x 256 B of weight (the eight slices' blocks: two full lines per row; 2 DMAs per loader wave) and
// 16 x 8 scales (1 DMA of 4 bytes per lane each) into a ring of DW steps, DW - 2 of them in flight across every barrier.
// Weight image: [row][256 B] with 16-byte piece ^ row on the DMA source and on the ds_read_b64 (conflict-free).
template <int DT, int NT>
__global__ __launch_bounds__(640) void gemm16_wide_ring8_kernel(const uint16_t *__restrict__ x, const uint8_t *__restrict__ W,
                                                                const float *__restrict__ absmax, const uint16_t *__restrict__ bias,
                                                                const uint16_t *residual, uint16_t *out, int B, int M, int K, int mode) {
    // ring depths: x DX slots per K slice (DX - 2 steps of x stay in flight across a barrier), weight DW steps; LDS decides:
    // NT = 1: 64 + 36 KB, NT = 2: 128 + 27 KB, NT = 3: 144 + 13.5 KB, NT = 4: 128 + 27 KB
    constexpr int WK = 8, DX = NT <= 2 ? 4 : (NT == 3 ? 3 : 2), DW = NT == 3 ? 3 : ((NT == 4 || NT == 2) ? 6 : 8);
    constexpr int kSlot = NT * 2048, kXRing = DX * WK * kSlot;
    constexpr int kWSlot = 16 * 256, kSSlot = 16 * 32;
    constexpr int kPart = WK * NT * 1024;
    constexpr int kXBytes = kXRing > kPart ? kXRing : kPart;
    constexpr int kPerStep = 3;
    __shared__ __attribute__((aligned(1024))) uint8_t s_raw[kXBytes + DW * (kWSlot + kSSlot)];
    uint8_t *s_w = s_raw + kXBytes, *s_s = s_w + DW * kWSlot;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave & 7;
    const int i = lane & 15, kb = lane >> 4;
    const int row0 = blockIdx.x * 16;
    // K % 512 != 0: the last step is ragged - K slices past the row's end skip their DMA and their arithmetic, and the loader
    // lanes whose bytes would lie past the row's end re-read the step's first piece instead (never used)
    const int nblk = K >> 6, steps = (nblk + WK - 1) / WK;
    const bool loader = wave >= 8;  // wave-uniform
    const int v = wave & 1;         // loader index

    uint32_t xoff[2 * NT];
#pragma unroll
    for (int d = 0; d < 2 * NT; ++d) {
        const int n = 8 * d + (lane >> 3), sl = lane & 7;
        const int nn = n < B ? n : B - 1;
        xoff[d] = (uint32_t)nn * (uint32_t)K * 2u + (uint32_t)((sl ^ ((n >> 1) & 7)) * 16);
    }
    const uint8_t *xb = reinterpret_cast<const uint8_t *>(x);
    // loader v, DMA d -> rows 4*(2v + d) .. +3; lane -> (row, 16-byte piece of the row's 256 B)
    const uint8_t *wsrc[2];
    int wpiece[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const int rl = 4 * (2 * v + d) + (lane >> 4), sl = lane & 15;
        const int r = row0 + rl;
        const int64_t row = r < M ? r : M - 1;
        wpiece[d] = sl ^ rl;
        wsrc[d] = W + row * (int64_t)(K >> 1);
    }
    // scales: loader v -> rows 8v .. 8v+7; lane -> (row, K slice)
    const int srl = 8 * v + (lane >> 3);
    const float *ssrc = absmax + (int64_t)(row0 + srl < M ? row0 + srl : M - 1) * nblk;

    const int xrd0 = i * 128 + (((2 * kb) ^ (i >> 1)) * 16), xrd1 = i * 128 + (((2 * kb + 1) ^ (i >> 1)) * 16);
    const int wrd = i * 256 + (((2 * wk + (kb >> 1)) ^ i) * 16) + (kb & 1) * 8;
    const int srd = i * 32 + wk * 4;

    f32x4 acc[1][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[0][nt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    auto issue_x = [&](int s) {  // compute wave: its own K slice
        if (s * WK + wk >= nblk) return;  // wave-uniform: no such block (ragged last step)
        uint8_t *slot = s_raw + ((s % DX) * WK + wk) * kSlot;
#pragma unroll
        for (int d = 0; d < 2 * NT; ++d) lds_dma16(xb + xoff[d] + (uint32_t)(s * WK + wk) * 128u, slot + d * 1024);
    };
    auto issue_w = [&](int s) {  // loader wave
        const int ring = s % DW, left = nblk - s * WK;  // blocks of this step that exist (>= 1)
#pragma unroll
        for (int d = 0; d < 2; ++d)
            lds_dma16(wsrc[d] + s * 256 + ((wpiece[d] >> 1) < left ? wpiece[d] * 16 : 0), s_w + ring * kWSlot + (2 * v + d) * 1024);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ssrc + s * 8 + ((lane & 7) < left ? (lane & 7) : 0)),
                                         (__attribute__((address_space(3))) void *)(s_s + ring * kSSlot + v * 256), 4, 0, 0);
    };
    if (loader) {
        for (int s = 0; s < DW - 1 && s < steps; ++s) issue_w(s);
    } else {
        for (int s = 0; s < DX - 1 && s < steps; ++s) issue_x(s);
    }
    for (int s = 0; s < steps; ++s) {
        // step s has landed; younger steps may stay in flight (not in the last steps, where a ragged step issues fewer DMAs)
        if (loader && s + DW - 2 < steps)
            wait_vmcnt<(DW - 2) * kPerStep>();
        else if (!loader && DX > 2 && s + DX - 2 < steps - 1)
            wait_vmcnt<(DX - 2) * 2 * NT>();
        else
            wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (loader) {
            if (s + DW - 1 < steps) issue_w(s + DW - 1);
            continue;  // loader waves take no part in the arithmetic
        }
        if (s + DX - 1 < steps) issue_x(s + DX - 1);  // into the slot step s - 1 used
        if (s * WK + wk >= nblk) continue;  // wave-uniform (ragged last step)
        const uint8_t *xs = s_raw + ((s % DX) * WK + wk) * kSlot;
        const u32x2 wq = *reinterpret_cast<const u32x2 *>(s_w + (s % DW) * kWSlot + wrd);
        const float am = *reinterpret_cast<const float *>(s_s + (s % DW) * kSSlot + srd);
        f32x4 tile[NT];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const u32x4 wf = decode8_natural<DT>(t == 0 ? wq.x : wq.y);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const u32x4 xf = *reinterpret_cast<const u32x4 *>(xs + nt * 2048 + (t == 0 ? xrd0 : xrd1));
                tile[nt] = mfma_xw<DT>(xf, wf, t == 0 ? f32x4{0.0f, 0.0f, 0.0f, 0.0f} : tile[nt]);
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            acc[0][nt].x = __builtin_fmaf(tile[nt].x, am, acc[0][nt].x);
            acc[0][nt].y = __builtin_fmaf(tile[nt].y, am, acc[0][nt].y);
            acc[0][nt].z = __builtin_fmaf(tile[nt].z, am, acc[0][nt].z);
            acc[0][nt].w = __builtin_fmaf(tile[nt].w, am, acc[0][nt].w);
        }
    }
    __syncthreads();
    wide_epilogue<DT, NT, 1, 1, 640>(s_raw, acc, bias, residual, out, B, M, row0, mode);
}

// ---- up to 32 rows on short weights: eight self-contained waves, no barrier in the K loop ------------------------------------------
// 16 rows per workgroup, K split 8 ways, and every wave fetches ITS OWN operands - x, weight and scales of two consecutive quant
// blocks per step - into wave-private ring slots: nothing another wave reads, so the loop needs no workgroup barrier, only the
// wave's own counted s_waitcnt (one queue, every step the same 4*NT + 2 DMAs, D - 2 steps left in flight).  A step's two blocks are
// 64 contiguous bytes per weight row (one DMA: 16 rows x 4 lanes) and give two independent ds_read -> decode -> MFMA -> scale chains;
// the ring kernels above pay a barrier and one such chain per block, which is what bounds them at one or two column tiles
// (4096 x 14336 x 16 rows: 0.6 us per step there).  Weight image: [row][64 B] with 16-byte piece ^ (row >> 2 & 3).
template <int DT, int NT>
__global__ __launch_bounds__(512) void gemm16_wide_auto_kernel(const uint16_t *__restrict__ x, const uint8_t *__restrict__ W,
                                                               const float *__restrict__ absmax, const uint16_t *__restrict__ bias,
                                                               const uint16_t *residual, uint16_t *out, int B, int M, int K, int mode) {
    constexpr int WK = 8, D = NT == 1 ? 3 : 2;
    constexpr int kXSlot = 2 * NT * 2048, kWaveSlot = kXSlot + 1024 + 256;  // x (2 blocks), weight (16 x 64 B), scales (16 x 2, padded)
    constexpr int kWaveBytes = D * kWaveSlot;
    constexpr int kPart = WK * NT * 1024;
    static_assert(8 * kWaveBytes <= 160 * 1024 && kPart <= 8 * kWaveBytes, "LDS");
    __shared__ __attribute__((aligned(1024))) uint8_t s_raw[8 * kWaveBytes];
    const int tid = threadIdx.x, lane = tid & 63, wk = tid >> 6;
    const int i = lane & 15, kb = lane >> 4;
    const int row0 = blockIdx.x * 16;
    const int nblk = K >> 6, steps = (nblk + 2 * WK - 1) / (2 * WK);
    uint8_t *mine = s_raw + wk * kWaveBytes;

    uint32_t xoff[2 * NT];
#pragma unroll
    for (int d = 0; d < 2 * NT; ++d) {
        const int n = 8 * d + (lane >> 3), sl = lane & 7;
        const int nn = n < B ? n : B - 1;
        xoff[d] = (uint32_t)nn * (uint32_t)K * 2u + (uint32_t)((sl ^ ((n >> 1) & 7)) * 16);
    }
    const uint8_t *xb = reinterpret_cast<const uint8_t *>(x);
    // weight: lane -> (row = lane >> 2, 16-byte piece of the step's 64 B); scales: lanes 0..31 -> (row = lane >> 1, block = lane & 1)
    const int wrl = lane >> 2, wp = (lane & 3) ^ ((wrl >> 2) & 3);
    const uint8_t *wsrc = W + (int64_t)(row0 + wrl < M ? row0 + wrl : M - 1) * (int64_t)(K >> 1);
    const int srl = (lane >> 1) & 15;
    const float *ssrc = absmax + (int64_t)(row0 + srl < M ? row0 + srl : M - 1) * nblk;

    const int xrd0 = i * 128 + (((2 * kb) ^ (i >> 1)) * 16), xrd1 = i * 128 + (((2 * kb + 1) ^ (i >> 1)) * 16);

    f32x4 acc[1][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[0][nt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    constexpr int kGroup = 4 * NT + 2;
    auto issue = [&](int t) {
        const int jb0 = (t * WK + wk) * 2, left = nblk - jb0;  // blocks of this wave's pair that exist: <= 0, 1 or >= 2 (wave-uniform)
        if (left <= 0) return;
        uint8_t *slot = mine + (t % D) * kWaveSlot;
#pragma unroll
        for (int b = 0; b < 2; ++b)
            if (b < left) {
#pragma unroll
                for (int d = 0; d < 2 * NT; ++d) lds_dma16(xb + xoff[d] + (uint32_t)(jb0 + b) * 128u, slot + b * (NT * 2048) + d * 1024);
            }
        lds_dma16(wsrc + jb0 * 32 + ((wp >> 1) < left ? wp * 16 : (wp & 1) * 16), slot + kXSlot);
        if (lane < 32)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ssrc + jb0 + ((lane & 1) < left ? (lane & 1) : 0)),
                                             (__attribute__((address_space(3))) void *)(slot + kXSlot + 1024), 4, 0, 0);
    };
    for (int t = 0; t < D - 1 && t < steps; ++t) issue(t);
    for (int t = 0; t < steps; ++t) {
        // this wave's step t has landed; D - 2 younger steps may stay in flight (not near the end, where groups may be smaller)
        if (D > 2 && t + D - 2 < steps - 1)
            wait_vmcnt<(D - 2) * kGroup>();
        else
            wait_vmcnt<0>();
        asm volatile("" ::: "memory");
        if (t + D - 1 < steps) issue(t + D - 1);  // into the slot step t - 1 used (its reads have been consumed)
        const int left = nblk - (t * WK + wk) * 2;
        const uint8_t *slot = mine + (t % D) * kWaveSlot;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if (b >= left) break;  // wave-uniform (ragged end)
            const u32x2 wq = *reinterpret_cast<const u32x2 *>(slot + kXSlot + i * 64 + (((2 * b + (kb >> 1)) ^ ((i >> 2) & 3)) * 16) + (kb & 1) * 8);
            const float am = *reinterpret_cast<const float *>(slot + kXSlot + 1024 + i * 8 + b * 4);
            const uint8_t *xs = slot + b * (NT * 2048);
            f32x4 tile[NT];
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const u32x4 wf = decode8_natural<DT>(t2 == 0 ? wq.x : wq.y);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const u32x4 xf = *reinterpret_cast<const u32x4 *>(xs + nt * 2048 + (t2 == 0 ? xrd0 : xrd1));
                    tile[nt] = mfma_xw<DT>(xf, wf, t2 == 0 ? f32x4{0.0f, 0.0f, 0.0f, 0.0f} : tile[nt]);
                }
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                acc[0][nt].x = __builtin_fmaf(tile[nt].x, am, acc[0][nt].x);
                acc[0][nt].y = __builtin_fmaf(tile[nt].y, am, acc[0][nt].y);
                acc[0][nt].z = __builtin_fmaf(tile[nt].z, am, acc[0][nt].z);
                acc[0][nt].w = __builtin_fmaf(tile[nt].w, am, acc[0][nt].w);
            }
        }
    }
    __syncthreads();  // every wave is done with its ring before the partials overwrite the storage
    wide_epilogue<DT, NT, 1, 1>(s_raw, acc, bias, residual, out, B, M, row0, mode);
}

std::atomic<int> g_wide_cfg{-1};  // -1 heuristic; 0 = never (16-row launches); 1 / 2 / 3 / 4 = 16 / 32 / 64 / 128 rows per workgroup;
                                  // 5 = 16 rows, self-contained waves (up to 32 activation rows, else as 1)

template <int DT, int NT>
int dispatch_wide_cfg(int cfg, const void *x, const uint8_t *W, const float *absmax, const void *bias, const void *residual, void *out,
                      int B, int M, int K, int mode, hipStream_t stream) {
    const unsigned rows = (cfg == 1 || cfg == 5) ? 16u : (cfg == 2 ? 32u : (cfg == 3 ? 64u : 128u));
    const dim3 grid(((unsigned)M + rows - 1) / rows);
#define FP4_WIDE_ARGS reinterpret_cast<const uint16_t *>(x), W, absmax, reinterpret_cast<const uint16_t *>(bias), \
                      reinterpret_cast<const uint16_t *>(residual), reinterpret_cast<uint16_t *>(out), B, M, K, mode
    // 16 rows per workgroup: one or two column tiles (up to 32 activation rows) always take the barrier-free form - 5-17 % faster,
    // so the ring form is only built for three and four tiles (round 3); 128 rows per workgroup needs more than one column tile.
    if constexpr (NT <= 2) {
        if (cfg == 5 || cfg == 1) {
            hipLaunchKernelGGL((gemm16_wide_auto_kernel<DT, NT>), dim3(((unsigned)M + 15u) / 16u), dim3(512), 0, stream, FP4_WIDE_ARGS);
            return FP4_OK;
        }
    } else {
        if (cfg == 5 || cfg == 1) {
            hipLaunchKernelGGL((gemm16_wide_ring8_kernel<DT, NT>), grid, dim3(640), 0, stream, FP4_WIDE_ARGS);
            return FP4_OK;
        }
    }
    if (cfg == 2) {
        hipLaunchKernelGGL((gemm16_wide_ring_kernel<DT, NT, 1>), grid, dim3(512), 0, stream, FP4_WIDE_ARGS);
    } else if (cfg == 3 || NT == 1) {
        hipLaunchKernelGGL((gemm16_wide_ring_kernel<DT, NT, 2>), dim3(((unsigned)M + 63u) / 64u), dim3(512), 0, stream, FP4_WIDE_ARGS);
    } else {
        if constexpr (NT >= 2) hipLaunchKernelGGL((gemm16_wide_ring_kernel<DT, NT, 4>), grid, dim3(512), 0, stream, FP4_WIDE_ARGS);
    }
#undef FP4_WIDE_ARGS
    return FP4_OK;
}

template <int DT>
int dispatch_wide(int cfg, const void *x, const uint8_t *W, const float *absmax, const void *bias, const void *residual, void *out, int B,
                  int M, int K, int mode, hipStream_t stream) {
    const int nt = (B + 15) / 16;
    if (nt == 1) return dispatch_wide_cfg<DT, 1>(cfg, x, W, absmax, bias, residual, out, B, M, K, mode, stream);
    if (nt == 2) return dispatch_wide_cfg<DT, 2>(cfg, x, W, absmax, bias, residual, out, B, M, K, mode, stream);
    if (nt == 3) return dispatch_wide_cfg<DT, 3>(cfg, x, W, absmax, bias, residual, out, B, M, K, mode, stream);
    return dispatch_wide_cfg<DT, 4>(cfg, x, W, absmax, bias, residual, out, B, M, K, mode, stream);
}

}  // namespace

void set_wide_variant(int v) { g_wide_cfg = v < 0 ? -1 : (v > 5 ? 5 : v); }

// 17..64 activation rows (any_rows: 1..64 - the caller's other kernels do not cover the shape), 16-bit dtype, blocksize 64,
// K % 64 == 0, 16-byte aligned operands.  Returns FP4_OK after the launch, or -1 when the shape is not covered / the path is switched
// off (the caller then streams the weight once per 16 rows).
int gemm_wide_launch(int dtype, const void *x, const uint8_t *W, const float *absmax, const void *bias, const void *residual, void *out,
                     int B, int M, int K, int mode, bool any_rows, hipStream_t stream) {
    int cfg = g_wide_cfg.load(std::memory_order_relaxed);
    if (cfg == 0 || B < 1 || (B <= 16 && !any_rows) || B > 64 || (K % 64) != 0 || M < 1) return -1;
    if ((uint64_t)B * (uint64_t)K * 2u >= (uint64_t(1) << 32)) return -1;  // 32-bit x offsets
    if (cfg < 0) {
        // Measured (profiles/r02_wide_batch_17_to_128_rows.txt, MI355X): the tallest workgroup that still fills three quarters of the
        // chip - 128 rows (four tiles per wave), 64, 32 - and below that 16 rows with two loader waves.
        const int cus = device_cu_count();
        if (B <= 16)  // one column tile: little x to share, 32 rows from 5120 rows on, never 128
            cfg = M >= 48 * cus ? 3 : (M >= 20 * cus ? 2 : 1);
        else
            cfg = M >= 96 * cus ? 4 : (M >= 48 * cus ? 3 : (M >= 24 * cus ? 2 : 1));
        if (cfg == 1 && B <= 32) cfg = 5;  // 16 rows per workgroup, one or two column tiles: the barrier-free form (5-17 % faster)
    }
    return dtype == FP4_DTYPE_F16 ? dispatch_wide<FP4_DTYPE_F16>(cfg, x, W, absmax, bias, residual, out, B, M, K, mode, stream)
                                  : dispatch_wide<FP4_DTYPE_BF16>(cfg, x, W, absmax, bias, residual, out, B, M, K, mode, stream);
}

}  // namespace fp4
