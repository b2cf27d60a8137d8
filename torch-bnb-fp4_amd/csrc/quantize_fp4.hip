// Blockwise FP4 quantiser for gfx950 -- the producer side of the hot path.
//
// In the reference this step is bitsandbytes' (Params4bit.cuda() / BF.quantize_fp4, call sites
// torch_bnb_fp4/__init__.py:736-747,775-777,859-861); bitsandbytes is not part of the reference
// checkout, so the algorithm restated here is the published one (see oracle/fp4_oracle.py):
//   absmax = max|w| over the block;  x = w * (1/absmax) in f32;
//   magnitude rank = #{t in thresholds : |x| > t};  code = rankmap[rank] | (x < 0 ? 8 : 0);
//   even element -> high nibble.
// 2-4 B read and 0.5625 B written per element.  Each lane owns 8 consecutive elements (one 16-byte load for 16-bit inputs) and
// emits one packed dword, so loads and stores are both fully coalesced; the block maximum is a butterfly over the bs/8 lanes
// that share a block (DPP up to 16 lanes, cross-wave through LDS only for blocksize > 512); the ranking costs 5 vector
// instructions per element (encode8_fast).  Two kernels share all of that and differ in memory structure only: the persistent
// quantize_kernel (next tile's load in flight behind the tile being ranked) and the one-shot quantize_tiles_kernel (the dequant
// kernel's geometry: every load of a lane up front); fp4_hip_quantize_blockwise picks by dtype and size from measurements.
#include <algorithm>
#include <atomic>

#include "fp4_common.h"

namespace fp4 {

namespace {

constexpr int kQThreads = 512;  // 4096 elements per workgroup = the largest supported blocksize

// Ranking.  Thresholds = midpoints between neighbouring magnitudes of {0, 1/192, 1/6, 1/4, 1/3, 1/2, 2/3, 1}, strict '>'
// (the published bitsandbytes rule, see oracle/fp4_oracle.py).  Done naively (seven float compares per element, ~70
// instructions with the SGPR-mask round trips) the kernel is VALU-bound at a third of the HBM rate, so the rank comes
// from a bucket table in LDS instead, exactly:
//   * for non-negative floats the unsigned order of the BIT PATTERNS is the numeric order;
//   * bucket = bits 30..20 (exponent + 3 mantissa bits); every threshold falls in a different bucket, so inside a bucket
//     the rank is `rank_lo`, plus one if the low 20 bits exceed that bucket's threshold;
//   * the constant table below holds, for the 71 buckets from the first threshold's to 1.0's, (7 - rank_lo) << 28 | threshold_low20
//     (0xFFFFF if the bucket has none): subtracting the element's low 20 bits borrows out of bit 28 exactly when they exceed the
//     threshold, leaving r = 7 - rank in bits 30..28;
//   * the LDS copy (fill in the kernels) covers EVERY bucket from 0 - everything below the first threshold's bucket is rank 0 -
//     and each entry has its own bucket number << 20 added, so that `entry - bits` needs no mask of the low 20 bits, no clamp of
//     |x| and - see encode8_fast - not even a separate sign merge.
// One v_alignbit pushes the nibble [sign r2 r1 r0] into the packed word and three bitwise ops on the finished word turn all
// eight r's into codes (rank -> code is {0,1,6,7,4,5,2,3}, i.e. code = [r2^r1, ~r1, ~r0]).  Issue rates: profiles/r01_f_exp_valu_int_rates.txt.
constexpr uint32_t kThresholdBits[7] = {
    __builtin_bit_cast(uint32_t, 0.00260417f), __builtin_bit_cast(uint32_t, 0.0859375f),
    __builtin_bit_cast(uint32_t, 0.20833333f), __builtin_bit_cast(uint32_t, 0.29166667f),
    __builtin_bit_cast(uint32_t, 0.4166667f),  __builtin_bit_cast(uint32_t, 0.583333f),
    __builtin_bit_cast(uint32_t, 0.8333333f)};
constexpr uint32_t kLutFirst = kThresholdBits[0] >> 20;                      // 0x3B2
constexpr uint32_t kLutLast = __builtin_bit_cast(uint32_t, 1.0f) >> 20;      // 0x3F8
constexpr int kLutSize = int(kLutLast - kLutFirst) + 1;                      // 71
constexpr float kLutFloor = __builtin_bit_cast(float, kLutFirst << 20);      // 2^-9 * 1.25

struct RankLut {
    uint32_t e[kLutSize];
};
constexpr RankLut make_rank_lut() {
    RankLut lut{};
    for (int i = 0; i < kLutSize; ++i) {
        const uint32_t bucket = kLutFirst + uint32_t(i);
        uint32_t below = 0, thr = 0xFFFFFu;
        for (int j = 0; j < 7; ++j) {
            if ((kThresholdBits[j] >> 20) < bucket) ++below;
            if ((kThresholdBits[j] >> 20) == bucket) thr = kThresholdBits[j] & 0xFFFFFu;
        }
        lut.e[i] = ((7u - below) << 28) | thr;
    }
    return lut;
}
constexpr bool thresholds_in_distinct_buckets() {
    for (int j = 1; j < 7; ++j)
        if ((kThresholdBits[j] >> 20) <= (kThresholdBits[j - 1] >> 20)) return false;
    return (kThresholdBits[6] >> 20) < kLutLast;
}
static_assert(thresholds_in_distinct_buckets(), "the bucket table needs at most one threshold per bucket");
__device__ const RankLut kRankLut = make_rank_lut();

// 8 scaled values -> one packed dword (byte j = element 2j in the high nibble, 2j+1 in the low nibble).  `lut` is indexed
// by the absolute bucket number.  GUARD is the rare path for a block whose absmax or 1/absmax is not finite (subnormal,
// inf or NaN weights): NaN products (0*inf, inf*0, NaN) must rank as 0 with no sign, which is what the reference's float
// compares do with a NaN, and inf must rank as 7.
template <bool GUARD>
__device__ __forceinline__ uint32_t encode8(const float (&v)[8], float inv, const uint32_t *lut) {
    constexpr int order[8] = {6, 7, 4, 5, 2, 3, 0, 1};  // the first nibble pushed ends up in bits 31..28
    uint32_t word = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float x = v[order[k]] * inv;
        if (GUARD) x = (x != x) ? 0.0f : x;
        x = x + 0.0f;  // -0.0 (a -0.0 weight or an underflowed product) -> +0.0: `x < 0` is false for it
        float mag = __builtin_fmaxf(__builtin_fabsf(x), kLutFloor);
        if (GUARD) mag = __builtin_fminf(mag, 1.0f);
        const uint32_t mb = __builtin_bit_cast(uint32_t, mag);
        const uint32_t r = lut[mb >> 20] - mb;  // bits 30..28 = 7 - rank (the entry carries its bucket number, see fill_lut)
        // bit 31 from x, the rest from r: (r & ~C) | (x & C)
        const uint32_t nib = __builtin_amdgcn_bitop3_b32(r, __builtin_bit_cast(uint32_t, x), 0x80000000u, 0xD8);
        word = __builtin_amdgcn_alignbit(word, nib, 28);  // (word << 4) | (nib >> 28)
    }
    // r -> code in all eight nibbles at once: flip r1 and r0, code bit 2 = r2 ^ r1
    return (word ^ 0x33333333u) ^ ((word << 1) & 0x44444444u);
}

// The hot path (every scale of the wave finite): 5 vector instructions per element instead of 9.
//   * x = w * (1/absmax), then x + 0.0, for TWO elements per instruction (v_pk_mul_f32, v_pk_add_f32).  NOT one fma: a product
//     that underflows to zero keeps the sign of the exact value through an fma (fma(-tiny, inv, +0) = -0.0), while the separate
//     multiply rounds to -0.0 first and the add then turns it into +0.0 - which is what `x < 0` in the reference sees
//     (caught by test_quantize_every_16bit_pattern: large-absmax blocks have a subnormal 1/absmax);
//   * no clamp of |x|: the table covers every bucket from 0 (all of them below the first threshold's are rank 0), and a finite
//     scale means |x| <= 1.0, the last bucket;
//   * no separate |x|, no mask, no sign merge: the bucket index is a bit-field extract that skips the sign bit, and
//       entry - bits(x)  =  ((7 - rank_lo) << 28) + thr_low20 - low20(x)  -  sign * 2^31        (mod 2^32),
//     where the first three terms lie in [0, 2^31) (rank_lo = 7 only in buckets above the last threshold, whose thr_low20 is
//     0xFFFFF), so bit 31 of the difference IS the sign of x and bits 30..28 are 7 - rank: the nibble [sign r2 r1 r0] as is.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t encode8_fast(const float (&v)[8], float inv, const uint32_t *lut) {
    const f32x2 inv2 = {inv, inv}, zero2 = {0.0f, 0.0f};
    uint32_t word = 0;
#pragma unroll
    for (int p = 3; p >= 0; --p) {  // pairs (6,7) (4,5) (2,3) (0,1): the first nibble pushed ends up in bits 31..28
        const f32x2 w2 = {v[2 * p], v[2 * p + 1]};
        const f32x2 x2 = w2 * inv2 + zero2;  // -ffp-contract=off (build.py): two instructions, two roundings
        const uint32_t xa = __builtin_bit_cast(uint32_t, float(x2.x)), xb = __builtin_bit_cast(uint32_t, float(x2.y));
        word = __builtin_amdgcn_alignbit(word, lut[(xa >> 20) & 0x7FFu] - xa, 28);
        word = __builtin_amdgcn_alignbit(word, lut[(xb >> 20) & 0x7FFu] - xb, 28);
    }
    return (word ^ 0x33333333u) ^ ((word << 1) & 0x44444444u);
}

// One lane's 8 elements of a full tile, still in their storage type (16 or 32 bytes), streamed past the caches.
template <int DT>
struct RawTile {
    uint32_t d[DT == FP4_DTYPE_F32 ? 8 : 4];
};

template <int DT>
__device__ __forceinline__ RawTile<DT> load_raw(const void *w, int64_t e0) {
    RawTile<DT> r;
#ifdef FP4_ABL_NOLOAD  // experiment builds only: values made up from the element index, nothing read
    for (int i = 0; i < (DT == FP4_DTYPE_F32 ? 8 : 4); ++i) r.d[i] = (uint32_t(e0) * 2654435761u + uint32_t(i) * 40503u) & 0x3FFF3FFFu;
    (void)w;
    return r;
#endif
    if constexpr (DT == FP4_DTYPE_F32) {
        const u32x4 lo = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(w) + e0 / 4);
        const u32x4 hi = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(w) + e0 / 4 + 1);
        r.d[0] = lo.x, r.d[1] = lo.y, r.d[2] = lo.z, r.d[3] = lo.w, r.d[4] = hi.x, r.d[5] = hi.y, r.d[6] = hi.z, r.d[7] = hi.w;
    } else {
        const u32x4 lo = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(w) + e0 / 8);
        r.d[0] = lo.x, r.d[1] = lo.y, r.d[2] = lo.z, r.d[3] = lo.w;
    }
    return r;
}

template <int DT>
__device__ __forceinline__ void unpack8(const RawTile<DT> &r, float (&v)[8]) {
    if constexpr (DT == FP4_DTYPE_F32) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = __builtin_bit_cast(float, r.d[i]);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = to_f32<DT>(uint16_t(r.d[i] & 0xFFFFu));
            v[2 * i + 1] = to_f32<DT>(uint16_t(r.d[i] >> 16));
        }
    }
}

// Bit pattern (as f32) of the largest |w| among one lane's 8 elements.  |w| orders like its bit pattern, and a 16-bit value's
// pattern orders like the f32 it converts to (NaN patterns above inf included), so the 16-bit types take the maximum on PAIRS of
// raw patterns (v_pk_max_u16: 10 instructions for 8 elements instead of 16) and widen the winner once.
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
template <int DT>
__device__ __forceinline__ uint32_t lane_max_bits(const RawTile<DT> &r, const float (&v)[8]) {
    if constexpr (DT == FP4_DTYPE_F32) {
        uint32_t mb = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) mb = max(mb, __builtin_bit_cast(uint32_t, v[i]) & 0x7FFFFFFFu);
        return mb;
    } else {
        u16x2 m = __builtin_bit_cast(u16x2, r.d[0] & 0x7FFF7FFFu);
#pragma unroll
        for (int i = 1; i < 4; ++i) m = __builtin_elementwise_max(m, __builtin_bit_cast(u16x2, r.d[i] & 0x7FFF7FFFu));
        const uint16_t top = m.x > m.y ? m.x : m.y;
        return __builtin_bit_cast(uint32_t, to_f32<DT>(top));
    }
}
__device__ __forceinline__ uint32_t lane_max_bits_f32(const float (&v)[8]) {
    uint32_t mb = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) mb = max(mb, __builtin_bit_cast(uint32_t, v[i]) & 0x7FFFFFFFu);
    return mb;
}

// The ragged last tile: element by element, zeros past n.
template <int DT>
__device__ __forceinline__ void load8_tail(const void *w, int64_t e0, int64_t n, float (&v)[8]) {
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

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_umax(uint32_t v) {
    const uint32_t moved = uint32_t(__builtin_amdgcn_update_dpp(0, int(v), CTRL, 0xF, 0xF, false));
    return v > moved ? v : moved;
}

// One tile of kQThreads * 8 elements: block maxima, scales, codes.  TAIL = the ragged last tile (lanes past n idle, partial
// dword stores); full tiles have no divergent memory operation at all, so the compiler can count what is in flight.
template <bool TAIL>
__device__ __forceinline__ void quantize_tile(const float (&v)[8], uint32_t lane_max, int64_t e0, int64_t n, int tid, int bs_shift,
                                              uint8_t *__restrict__ packed, float *__restrict__ absmax,
                                              const uint32_t *lut, uint32_t *s_wave_max) {
    // Block maximum on the BIT PATTERNS of |w| (same order as the values; a NaN weight, whose pattern is above inf,
    // propagates into absmax the way numpy's max does).  lanes_per_block is uniform across the grid, so these are scalar
    // branches; up to 16 lanes the butterfly is pure DPP.
    const int lanes_per_block = 1 << (bs_shift - 3);
    uint32_t mb = lane_max;
#ifdef FP4_ABL_NOABSMAX  // experiment builds only (tools/exp_quant_ablate.sh, never defined by build.py): results are meaningless
    mb = (__builtin_bit_cast(uint32_t, v[0]) & 0x007FFFFFu) | 0x40000000u;
    if (false) {
#endif
    mb = dpp_umax<0xB1>(mb);                                                        // quad_perm [1,0,3,2]
    mb = dpp_umax<0x4E>(mb);                                                        // quad_perm [2,3,0,1]
    if (lanes_per_block >= 8) mb = dpp_umax<0x141>(mb);                             // row_half_mirror
    if (lanes_per_block >= 16) mb = dpp_umax<0x140>(mb);                            // row_mirror
    if (lanes_per_block >= 32) mb = max(mb, uint32_t(__shfl_xor(int(mb), 16)));
    if (lanes_per_block >= 64) mb = max(mb, uint32_t(__shfl_xor(int(mb), 32)));
    if (lanes_per_block > 64) {
        if ((tid & 63) == 0) s_wave_max[tid >> 6] = mb;
        __syncthreads();
        const int waves_per_block = lanes_per_block >> 6;
        const int first = ((tid >> 6) / waves_per_block) * waves_per_block;
        for (int i = 0; i < waves_per_block; ++i) mb = max(mb, s_wave_max[first + i]);
        __syncthreads();  // s_wave_max is rewritten by the next tile
    }
#ifdef FP4_ABL_NOABSMAX
    }
#endif
    if (TAIL && e0 >= n) return;
    const float m = __builtin_bit_cast(float, mb);
    // every lane of a block holds the same maximum; in a full tile all of them store it (same address, same value) so that
    // the store is not under a divergent branch
#ifdef FP4_ABL_NOSTORE
    if (mb == 0x12345678u)
#endif
    if (!TAIL || (tid & (lanes_per_block - 1)) == 0) __builtin_nontemporal_store(m, absmax + (e0 >> bs_shift));

    // x = w * (1/absmax) as the reference computes it.  An all-zero block (1/0 = inf, 0*inf = NaN, every compare false)
    // encodes as all +0, which scaling by 0 reproduces without the NaN.
    const float inv = m > 0.0f ? 1.0f / m : 0.0f;
    const bool not_finite = !(inv <= 3.4028234664e38f && m <= 3.4028234664e38f);  // inf, NaN or subnormal absmax
    uint32_t word;
#ifdef FP4_ABL_NORANK
    word = __builtin_bit_cast(uint32_t, inv);
    for (int i = 0; i < 8; ++i) word ^= __builtin_bit_cast(uint32_t, v[i]);
    if (false)
#endif
    if (__builtin_amdgcn_ballot_w64(not_finite) != 0)  // wave-uniform: keeps the guard out of the hot path
        word = encode8<true>(v, inv, lut);
    else
#ifdef FP4_EXP_QUANT_OLD_ENCODE  // A/B builds only (tools/exp_quant_ablate.sh): the round-1..4 ranking, 9 instructions per element
        word = encode8<false>(v, inv, lut);
#else
        word = encode8_fast(v, inv, lut);
#endif
#ifdef FP4_ABL_NOSTORE
    if (word == 0x9ABCDEF1u && mb == 0x12345678u)
#endif
    if (!TAIL || e0 + 8 <= n) {
        __builtin_nontemporal_store(word, reinterpret_cast<uint32_t *>(packed) + e0 / 8);
    } else {
        const int nbytes = int((n - e0 + 1) / 2);
        for (int b = 0; b < nbytes; ++b) packed[e0 / 2 + b] = uint8_t(word >> (8 * b));
    }
}

// Persistent: a workgroup walks the full tiles with stride gridDim.x and keeps the NEXT tile's load in flight while it
// ranks the current one, so the VALU work of one tile overlaps the HBM latency of the next (a one-shot grid leaves the
// first and last generation of waves un-overlapped).  The ragged last tile, if any, is done after the loop by the
// workgroup whose turn it would have been.  Every condition on a tile index is uniform across the workgroup.
template <int DT>
__global__ __launch_bounds__(kQThreads) void quantize_kernel(const void *__restrict__ w, uint8_t *__restrict__ packed,
                                                             float *__restrict__ absmax, int64_t n, int bs_shift) {
    __shared__ uint32_t s_wave_max[kQThreads / 64];
    __shared__ uint32_t s_lut[kLutLast + 1];  // indexed by bucket number = bits 30..20 of |x|, from 0
    static_assert(kLutLast + 1 <= 2 * kQThreads && kLutFirst >= uint32_t(kQThreads), "fill_lut: two entries per thread, the table part in the second");
    constexpr int64_t kTile = int64_t(kQThreads) * 8;
    const int tid = threadIdx.x;
    const int64_t nfull = n / kTile;

    // The table entry is loaded first and unconditionally: vector-memory results return in issue order, and a load under
    // a divergent branch would be drained on its own, so this way its (L2) latency hides under the first weight load's.
    const uint32_t b_hi = uint32_t(tid) + uint32_t(kQThreads);  // this thread's second bucket; only those can hold a threshold
    const uint32_t lut_entry = kRankLut.e[b_hi < kLutFirst ? 0u : (b_hi > kLutLast ? uint32_t(kLutSize - 1) : b_hi - kLutFirst)];
    int64_t tile = blockIdx.x;
    RawTile<DT> cur{};
    if (tile < nfull) cur = load_raw<DT>(w, tile * kTile + tid * 8);
    // fill_lut: buckets below the first threshold's are rank 0 with no threshold inside (7 << 28 | 0xFFFFF); every entry gets its
    // bucket number << 20 added, so that `entry - bits(|x|)` = (7 - rank_lo) << 28 | thr_low20, minus the low 20 bits of |x|.
    constexpr uint32_t kRankZero = (7u << 28) | 0xFFFFFu;
    s_lut[tid] = kRankZero + (uint32_t(tid) << 20);
    if (b_hi <= kLutLast) s_lut[b_hi] = (b_hi < kLutFirst ? kRankZero : lut_entry) + (b_hi << 20);
    __syncthreads();

    float v[8];
    while (tile < nfull) {
        const int64_t next = tile + gridDim.x;
        RawTile<DT> nxt{};
        if (next < nfull) nxt = load_raw<DT>(w, next * kTile + tid * 8);
        unpack8<DT>(cur, v);
        quantize_tile<false>(v, lane_max_bits<DT>(cur, v), tile * kTile + tid * 8, n, tid, bs_shift, packed, absmax, s_lut, s_wave_max);
        cur = nxt;
        tile = next;
    }
    if (tile == nfull && nfull * kTile < n) {  // exactly one workgroup gets here with the ragged tile
        const int64_t e0 = tile * kTile + tid * 8;
        load8_tail<DT>(w, e0, n, v);
        quantize_tile<true>(v, lane_max_bits_f32(v), e0, n, tid, bs_shift, packed, absmax, s_lut, s_wave_max);
    }
}

// The same work with the DEQUANT kernel's memory geometry (round 5; tools/exp_quant_ablate.sh showed the persistent kernel above at
// 8.0 us per 4096 x 4096 in steady state with ALL arithmetic removed - its own one-tile-in-flight load / store structure - against
// 6.4 us for reading and writing the same bytes back to back at this box's bare rates).  One-shot grid, 256-thread workgroups,
// each lane issues its four 16-byte loads (f32 input: eight) before anything else, a wave owns 4 KiB contiguous of 16-bit input
// and every one of its load / store instructions touches one contiguous span (1 KiB in, 256 B of packed bytes out).  Full tiles
// of kTileElems only, blocks of at most 512 elements (the lanes of a block sit in one wave): the launcher sends everything else
// to the persistent kernel.
constexpr int kTThreads = 256;
template <int DT, int kTLoads>
__global__ __launch_bounds__(kTThreads) void quantize_tiles_kernel(const void *__restrict__ w, uint8_t *__restrict__ packed,
                                                                   float *__restrict__ absmax, int64_t n, int bs_shift) {
    __shared__ uint32_t s_lut[kLutLast + 1];
    static_assert(kLutLast + 1 <= 4 * kTThreads && kLutFirst >= uint32_t(3 * kTThreads), "fill: four entries per thread, the table part in the last");
    const int tid = threadIdx.x;
    const uint32_t b_hi = uint32_t(tid) + uint32_t(3 * kTThreads);
    const uint32_t lut_entry = kRankLut.e[b_hi < kLutFirst ? 0u : (b_hi > kLutLast ? uint32_t(kLutSize - 1) : b_hi - kLutFirst)];
    const int64_t e_lane = int64_t(blockIdx.x) * (int64_t(kTThreads) * kTLoads * 8) + (tid >> 6) * (64 * kTLoads * 8) + (tid & 63) * 8;
    RawTile<DT> raw[kTLoads];
#pragma unroll
    for (int j = 0; j < kTLoads; ++j) raw[j] = load_raw<DT>(w, e_lane + j * 512);
    constexpr uint32_t kRankZero = (7u << 28) | 0xFFFFFu;
#pragma unroll
    for (int i = 0; i < 3; ++i) s_lut[tid + i * kTThreads] = kRankZero + (uint32_t(tid + i * kTThreads) << 20);
    if (b_hi <= kLutLast) s_lut[b_hi] = (b_hi < kLutFirst ? kRankZero : lut_entry) + (b_hi << 20);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kTLoads; ++j) {
        float v[8];
        unpack8<DT>(raw[j], v);
        quantize_tile<false>(v, lane_max_bits<DT>(raw[j], v), e_lane + j * 512, n, tid, bs_shift, packed, absmax, s_lut, nullptr);
    }
}

}  // namespace
}  // namespace fp4

namespace fp4 {
namespace {
std::atomic<int> g_quant_wg_per_cu{0};  // 0 = default (tiles kernel where it applies); sweeps: fp4_hip_set_variant("quantize", k > 0) = persistent kernel, k workgroups per CU
}
void set_quantize_variant(int v) { g_quant_wg_per_cu.store(v > 0 ? v : 0, std::memory_order_relaxed); }
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
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int wg_override = g_quant_wg_per_cu.load(std::memory_order_relaxed);
    // Which kernel (profiles/r05_quantize_geometry.txt, MI355X, us per launch HBM-cold / per matrix over a stack):
    //   f32 input: the tiles kernel with one load pair per lane wins at every size (4096 x 4096: 14.9 / 13.9 vs 16.6 / 15.4 persistent);
    //   16-bit input: small weights (< 8 M elements) tiles with one load per lane (1024 x 4096: 4.2 vs 4.5 us per launch); large ones
    //   (>= 32 M) tiles with two (14336 x 4096: 27.3 / 25.2 vs 29.4 / 30.2); in between the persistent kernel, whose overlap of one
    //   tile's ranking with the next tile's load wins PER LAUNCH (4096 x 4096: 9.5 vs 10.5) although the tiles kernel streams faster
    //   in steady state (7.8 vs 8.7 per matrix over a stack).
    // fp4_hip_set_variant("quantize", v): 0 = this heuristic, 1..999 = persistent kernel with v workgroups per CU, 1001 / 1002 / 1004 =
    // tiles kernel with 1 / 2 / 4 loads per lane (falls back to the persistent kernel where the tiles kernel does not apply).
    int loads = 0;  // 0 = persistent
    if (wg_override >= 1000)
        loads = wg_override - 1000;
    else if (wg_override == 0)
        loads = w_dtype == FP4_DTYPE_F32 ? 1 : (n < (int64_t(8) << 20) ? 1 : (n >= (int64_t(32) << 20) ? 2 : 0));
    const int64_t tile_elems = int64_t(kTThreads) * (loads > 0 ? loads : 1) * 8;
    if ((loads == 1 || loads == 2 || loads == 4) && bs_shift <= 9 && n % tile_elems == 0) {
        // whole tiles and blocks inside one wave (every decoder weight at the usual block sizes): one-shot tiles kernel
        const dim3 grid((unsigned)(n / tile_elems)), block(kTThreads);
#define FP4_QT(DTV, L) hipLaunchKernelGGL((quantize_tiles_kernel<DTV, L>), grid, block, 0, s, w, packed, absmax, n, bs_shift)
#define FP4_QT_LOADS(DTV)            \
    do {                             \
        if (loads == 1) FP4_QT(DTV, 1); \
        else if (loads == 2) FP4_QT(DTV, 2); \
        else FP4_QT(DTV, 4);         \
    } while (0)
        switch (w_dtype) {
            case FP4_DTYPE_F16: FP4_QT_LOADS(FP4_DTYPE_F16); break;
            case FP4_DTYPE_BF16: FP4_QT_LOADS(FP4_DTYPE_BF16); break;
            case FP4_DTYPE_F32: FP4_QT_LOADS(FP4_DTYPE_F32); break;
            default:
                set_error("fp4_hip_quantize_blockwise: unsupported dtype %d", w_dtype);
                return FP4_ERR_UNSUPPORTED;
        }
#undef FP4_QT_LOADS
#undef FP4_QT
        return check_launch("fp4_hip_quantize_blockwise");
    }
    const int64_t per_wg = int64_t(kQThreads) * 8;
    const int64_t ntiles = (n + per_wg - 1) / per_wg;
    // 4 workgroups of 512 threads fill a CU's 2048 wave slots; each keeps one tile in flight while it ranks another
    const int wg_per_cu = (wg_override > 0 && wg_override < 1000) ? wg_override : 4;
    const unsigned blocks = (unsigned)std::min<int64_t>(ntiles, int64_t(device_cu_count()) * wg_per_cu);  // ntiles >= 1
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
