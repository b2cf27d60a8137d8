/*
 * torch_bnb_fp4_hip.h -- C ABI of libtorch_bnb_fp4_hip.so, the MI355X (gfx950) FP4
 * dequant / fused-GEMV kernels.
 *
 * This is the drop-in boundary of the hot path: plain pointers and sizes, no torch
 * types.  Every pointer is a DEVICE pointer valid on the HIP device that is current
 * on the calling thread; `stream` is a hipStream_t (NULL = the null stream).  The
 * compute entry points are asynchronous (they enqueue on `stream` and return) and
 * re-entrant: they keep no state between calls and may be called from any number of
 * threads at once.  The only process-wide state is the benchmark hook
 * fp4_hip_set_variant() (relaxed atomics, read once per launch; a production caller
 * never touches it) and a per-device cache of the compute-unit count.
 * Return value: 0 on success, otherwise an fp4_status code; fp4_hip_last_error() then
 * holds a thread-local message.  The compute entry points neither allocate nor
 * synchronise, so every one of them is HIP-graph capturable.
 *
 * Each entry point names the reference interface it replaces
 * (aredden/torch-bnb-fp4, paths relative to that checkout).  The reference has no
 * C ABI of its own -- its boundary is the pybind module csrc/torch_fp4.cpp:125-139,
 * which torch-bnb-fp4_amd/csrc/torch_ext.cpp re-exports 1:1 on top of this header
 * (see INTEGRATION.md for the binding a reference maintainer would add).
 */
#ifndef TORCH_BNB_FP4_HIP_H
#define TORCH_BNB_FP4_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FP4_HIP_ABI_VERSION 7
#define FP4_HIP_API __attribute__((visibility("default")))

/* Element types, numbered like the reference's ScalarTypeEnum (csrc/torch_fp4.cpp:22-26). */
enum fp4_dtype { FP4_DTYPE_F16 = 0, FP4_DTYPE_F32 = 1, FP4_DTYPE_BF16 = 2 };

/* Which 16-entry code table a dequant uses.
 * CODEBOOK = the CODE_PARAM literals (csrc/dequant_fp4_optimized.cu:28-46),
 * TREE     = the constants of dequantize_fp4_tree (csrc/dequant_fp4_optimized.cu:55-76);
 * they differ by 1-12 ulp in f32 for nibbles 1,4,6 (and 9,12,14). */
enum fp4_table { FP4_TABLE_CODEBOOK = 0, FP4_TABLE_TREE = 1 };

/* Cache policy of the dequant's output stream.  AUTO = non-temporal loads and stores for large outputs (fastest
 * when the result is not read back at once: +30 % at 4096x4096); KEEP_CACHED = plain stores, for a consumer that
 * reads the weight right away (the dequant + GEMM of the batch > 1 path: the GEMM then hits L2 / Infinity Cache,
 * -2 us per 4096x4096 layer end to end); STREAM = always non-temporal. */
enum fp4_dequant_flags { FP4_DEQUANT_AUTO = 0, FP4_DEQUANT_KEEP_CACHED = 1, FP4_DEQUANT_STREAM = 2 };

enum fp4_status {
    FP4_OK = 0,
    FP4_ERR_INVALID_ARGUMENT = 1, /* null pointer, negative size, unknown enum */
    FP4_ERR_UNSUPPORTED = 2,      /* shape/blocksize the kernels do not cover */
    FP4_ERR_LAUNCH = 3            /* the HIP runtime failed: hipGetLastError() != hipSuccess after a launch, or a memset / sync of a host-side helper */
};

FP4_HIP_API int fp4_hip_abi_version(void);
FP4_HIP_API const char *fp4_hip_last_error(void);

/* Host-side copy of a code table (16 floats; nibble bit 3 = sign). */
FP4_HIP_API int fp4_hip_code_table(int table, float out16[16]);

/*
 * Blockwise FP4 -> f16 / bf16 / f32 dequant of n elements:
 *   out[e] = RN_T( f32(code[nibble(e)]) * absmax[e / blocksize] ),  0 <= e < n,
 * nibble(e) = HIGH nibble of packed[e/2] when e is even, LOW nibble when odd.
 * One f32 multiply, then round-to-nearest-even to T (f16 subnormals kept).
 *   packed : uint8[(n+1)/2]      absmax : float[ceil(n/blocksize)]      out : T[n]
 * blocksize: any even value >= 2 (power-of-two >= 32 takes the fast path).
 * Replaces dequantize_blockwise_fp4 (table = TREE, csrc/dequant_fp4_optimized.cu:182-205,
 * kernel :89-123) and dequantize_blockwise_codebook_fp4 (table = CODEBOOK, :207-255,
 * kernel :125-171).  Unlike the reference an unsupported dtype is an error, not a
 * printf (:201-203,250-252).
 */
FP4_HIP_API int fp4_hip_dequantize_blockwise(const uint8_t *packed, const float *absmax, void *out, int blocksize, int64_t n,
                                 int out_dtype, int table, int flags, void *stream);

/*
 * Fused batch-1 GEMV over an FP4 weight W[M,K] (row r = packed bytes [r*K/2, (r+1)*K/2),
 * scales absmax[(r*K + k) / blocksize]):
 *   out[r] = T( sum_k x[k] * code[nibble(r,k)] * absmax[(r*K+k)/blocksize] (+ bias[r]) )
 * accumulated in f32.  x, out, bias are T[K], T[M], T[M]; bias may be NULL.
 * With a bias the result is T( f32(T(sum)) + f32(bias[r]) ), i.e. exactly the
 * reference's separate `out += bias` (torch_bnb_fp4/__init__.py:608-613) fused in.
 * K must be even; K % 32 == 0 with a power-of-two blocksize >= 32 dividing K takes
 * the fast path (the reference's own GEMV gate, torch_bnb_fp4/__init__.py:593).
 * Replaces gemv_4bit_inference (csrc/gemv_fp4_optimized.cu:277-368; kernels :60-157
 * half/bf16 and :159-259 float).  The code values are those of the reference's GEMV, which ignores
 * its `datatype` tensor and uses CODE_PARAM (csrc/gemv_fp4_optimized.cu:266,274): the f32 kernels
 * read the bit-faithful CODE_PARAM f32 table; the f16 / bf16 kernels decode the exact k/12 values
 * (12 * |code| is exact in 16 bits, and CODE_PARAM rounded to f16 / bf16 - what the reference's
 * 16-bit kernels load, :92-96 - equals k/12 rounded to that type, tests/test_oracle.py), so the
 * two are indistinguishable there.
 */
FP4_HIP_API int fp4_hip_gemv(const void *x, const uint8_t *packed, const float *absmax, const void *bias, void *out, int64_t M,
                 int64_t K, int blocksize, int dtype, void *stream);

/*
 * The same GEMV with the elementwise work that FOLLOWS a Linear in a decoder layer folded into its epilogue, so that a
 * decode step pays one kernel boundary where the reference's op surface pays two to four
 * (torch_bnb_fp4/__init__.py:603-613 is where the reference already pays a separate launch for the bias).
 * Every intermediate is rounded to T exactly where the separate torch ops would round it:
 *   EPILOGUE_NONE:            t = T(sum_r); if bias: t = T(t + bias[r]); if residual: t = T(t + residual[r]); out[r] = t
 *                             out : T[M].  `residual` may alias `out` (h = h + Linear(a) in place).
 *   EPILOGUE_SILU_MUL_PAIRS:  the weight's rows interleave a gate and an up projection (row 2i = gate_i, row 2i+1 = up_i;
 *                             a row permutation done once at load time - rows of an FP4 weight are independent);
 *                             g = T(sum_2i) (+bias), u = T(sum_2i+1) (+bias), s = T(g / (1 + exp(-g))) (torch's silu),
 *                             t = T(s * u); if residual: t = T(t + residual[i]); out[i] = t.   out : T[M/2], M even.
 * 16-bit dtypes with K % 32 == 0 and a power-of-two blocksize >= 32 dividing K (the decode fast path) support both
 * epilogues; other shapes / f32 support EPILOGUE_NONE only and return FP4_ERR_UNSUPPORTED for the gated one, so the
 * caller can run the plain GEMV and the separate ops.  Not in the reference.
 */
enum fp4_epilogue { FP4_EPILOGUE_NONE = 0, FP4_EPILOGUE_SILU_MUL_PAIRS = 1 };
FP4_HIP_API int fp4_hip_gemv_fused(const void *x, const uint8_t *packed, const float *absmax, const void *bias, const void *residual,
                       void *out, int64_t M, int64_t K, int blocksize, int dtype, int epilogue, void *stream);

/*
 * Small-batch companion of the GEMV (2..128 activation rows; also accepts 1):
 *   out[b][r] = T( sum_k x[b][k] * code[nibble(r,k)] * absmax[(r*K+k)/blocksize] + bias[r] )
 * x is T[B,K] row-major, out T[B,M]; f32 accumulation, ONE rounding, bias added in f32 first (what the
 * reference's batch > 1 path, dequant + F.linear, does - torch_bnb_fp4/__init__.py:423-436,616-617 - without
 * writing and re-reading the M*K dequantised weight).  16-bit dtypes; f32 activations are covered up to 8 rows as B launches of the f32
 * GEMV (each row bit-identical to fp4_hip_gemv; FP4_ERR_UNSUPPORTED above that and for the gated epilogue).  Two kernels: a matrix-core one
 * (v_mfma_f32_16x16x32; blocksize 64, K % 512 == 0; any B <= 16) and a VALU one (B <= 8; K % 32 == 0, K <= 16384,
 * less for larger B; power-of-two blocksize >= 32 dividing K).  Returns FP4_ERR_UNSUPPORTED for shapes neither
 * covers, so the caller can fall back to dequant + GEMM.  17..64 rows (blocksize 64, K % 64 == 0; and 1..16 rows where
 * K % 512 != 0 keeps the two kernels above out, e.g. K = 11008): ONE pass over the
 * weight on the matrix cores with 2..4 column tiles of x per decoded weight fragment (x staged through LDS by LDS-DMA);
 * where that kernel does not apply the rows are split evenly over ceil(B/16) launches, each streaming the weight once.
 * 65..128 rows: two even chunks of at most 64.  Measured against dequant + hipBLASLt GEMM on MI355X
 * (profiles/r02_wide_batch_17_to_128_rows.txt): 1.7-3.1x faster at 17..64 rows, 1.27-1.6x at 65..128.
 */
FP4_HIP_API int fp4_hip_gemm_small(const void *x, const uint8_t *packed, const float *absmax, const void *bias, void *out,
                                   int64_t B, int64_t M, int64_t K, int blocksize, int dtype, void *stream);

/*
 * The small-batch product with the fused decode epilogues of fp4_hip_gemv_fused, for 1..128 activation rows (batched decode):
 *   EPILOGUE_NONE:            t = T(sum[b][r] + bias[r]) (F.linear: bias in f32, one rounding); if residual: t = T(t + residual[b][r])
 *                             out : T[B, M], residual : T[B, M].
 *   EPILOGUE_SILU_MUL_PAIRS:  rows interleave gate and up (row 2i / 2i+1): g = T(sum_2i + bias_2i), u = T(sum_2i+1 + bias_2i+1),
 *                             t = T(T(silu(g)) * u); if residual: t = T(t + residual[b][i]);  out, residual : T[B, M/2], M even.
 * Same shape coverage and FP4_ERR_UNSUPPORTED behaviour as fp4_hip_gemm_small.  Not in the reference.
 */
FP4_HIP_API int fp4_hip_gemm_small_fused(const void *x, const uint8_t *packed, const float *absmax, const void *bias,
                             const void *residual, void *out, int64_t B, int64_t M, int64_t K, int blocksize, int dtype,
                             int epilogue, void *stream);

/*
 * fp4_hip_gemm_small_fused with a caller-provided scratch buffer.  On short weights with long rows (M below 24 rows per CU, K >= 8192:
 * the down projection of a decoder) 33..64 activation rows (65..128: two even chunks) run as x-stationary split-K over workgroups: partial sums of 512-column
 * slices go through `workspace` and a second small launch adds them in a fixed order and applies the epilogue (deterministic, no
 * atomics).  fp4_hip_gemm_small_ws_bytes returns the bytes that path wants for a shape, or 0 where it would not be used (then, or
 * with workspace == NULL or too small, the call is exactly fp4_hip_gemm_small_fused).  16-byte aligned workspace; it may be reused by
 * the next call on the same stream.  Not in the reference.
 */
FP4_HIP_API int64_t fp4_hip_gemm_small_ws_bytes(int64_t B, int64_t M, int64_t K, int blocksize, int dtype);
FP4_HIP_API int fp4_hip_gemm_small_ws(const void *x, const uint8_t *packed, const float *absmax, const void *bias, const void *residual,
                          void *out, int64_t B, int64_t M, int64_t K, int blocksize, int dtype, int epilogue, void *workspace,
                          int64_t workspace_bytes, void *stream);

/*
 * K-split (row-parallel) building block: the same GEMV, but the f32 accumulator is written as is
 * (out_f32 : float[M], no bias, no rounding to T), so that the partial sums of the column shards
 * of one weight can be added across GPUs (RCCL all-reduce) before the single final rounding.
 * Not in the reference (it has no multi-GPU path); x is T[K] of x_dtype.
 */
FP4_HIP_API int fp4_hip_gemv_partial(const void *x, const uint8_t *packed, const float *absmax, float *out_f32, int64_t M,
                                     int64_t K, int blocksize, int x_dtype, void *stream);

/*
 * One-shot all-reduce for the K-split partials (the reference has no multi-GPU path).
 *
 * Setup (host side, synchronous, once): every rank allocates one slot buffer of fp4_hip_comm_bytes(world, capacity) bytes
 * with fp4_hip_comm_alloc (zero-filled device memory, uncached / fine-grained where the runtime offers it; `kind_out`
 * reports 0 uncached, 1 fine-grained, 2 plain), sends the 64-byte IPC handle to its peers by whatever channel the host
 * program has (torch.distributed all_gather in torch_bnb_fp4/comm.py) and maps theirs with fp4_hip_comm_open.
 *
 * fp4_hip_allreduce_oneshot(partial, peer_buffers, rank, world, M, capacity, bias, residual, out, dtype, timeout_us, stream):
 *   out[e] = T( T( T( sum_{p < world} partial_p[e] ) + bias[e] ) + residual[e] ),   e < M <= capacity,
 * the sum taken in f32 in rank order (identical bits on every rank).  `peer_buffers` is a HOST array of `world` device
 * pointers, entry `rank` being this rank's own buffer.  Each rank writes its partial as 8-byte {epoch, value} granules
 * straight into its slot of every peer's buffer (one hop over all xGMI links at once) and sweeps its own buffer until
 * all `world` slots carry the call's epoch; the epoch lives in device memory and slots are double-buffered, so the call
 * needs no host involvement and is HIP-graph capturable.  All ranks must issue their calls on a communicator in the
 * same order, one stream per rank.  Polling is bounded: after `timeout_us` (<= 0: 2 s) without a peer's data a lane
 * records {epoch, peer} in the buffer's status word and writes NaN; fp4_hip_comm_status copies {epoch, busy, status,
 * lanes timed out} to the host (synchronous) so the caller can raise.  The status word is sticky (first time-out wins) until
 * fp4_hip_comm_clear_status zeroes it and the lane count (synchronous; the epoch is kept, the call sequence continues), so that a
 * transient time-out is reported once and later checks speak about later calls.
 * Limit of that recovery: slots are double-buffered by call parity, so a rank that gave up on call n and carries on rewrites the slot of
 * call n at call n + 2.  A peer that lags by LESS than one call still finds its data; one that lags further meets a newer epoch in
 * the slot, times out in turn and writes NaN (never a wrong finite value).  Only the rank that timed out sees a status word - its peer
 * may hold a valid result for the same call - so the decision to go on must be taken by the whole group: reduce the status words
 * (MAX) over the ranks at the sync point and treat a non-zero result as a failure on every rank (OneShotAllReduce.check_collective).
 */
FP4_HIP_API int64_t fp4_hip_comm_bytes(int world, int64_t capacity);
FP4_HIP_API int fp4_hip_comm_alloc(int64_t bytes, void **ptr, uint8_t handle_out[64], int *kind_out);
FP4_HIP_API int fp4_hip_comm_open(const uint8_t handle[64], void **ptr);
FP4_HIP_API int fp4_hip_comm_close(void *ptr);
FP4_HIP_API int fp4_hip_comm_free(void *ptr);
FP4_HIP_API int fp4_hip_comm_status(const void *own_buffer, uint32_t out4[4]);
FP4_HIP_API int fp4_hip_comm_clear_status(void *own_buffer);
FP4_HIP_API int fp4_hip_allreduce_oneshot(const float *partial, void *const *peer_buffers, int rank, int world, int64_t M,
                              int64_t capacity, const void *bias, const void *residual, void *out, int out_dtype,
                              int64_t timeout_us, void *stream);

/*
 * Blockwise FP4 quantiser (the producer side; bitsandbytes' quantize_fp4 as called at
 * torch_bnb_fp4/__init__.py:775 and inside Params4bit.cuda(), :861):
 * per block of `blocksize` elements absmax = max|w|, code = nearest FP4 magnitude of
 * w/absmax (midpoint thresholds, strict >), sign in bit 3, even element in the high
 * nibble.  w is T[n] (w_dtype), packed uint8[(n+1)/2], absmax float[ceil(n/blocksize)].
 * blocksize: power of two, 32..4096.
 */
FP4_HIP_API int fp4_hip_quantize_blockwise(const void *w, int w_dtype, uint8_t *packed, float *absmax, int64_t n, int blocksize,
                               void *stream);

/*
 * Tuning hook for benchmarks/sweeps: selects a kernel geometry by name
 * ("dequant", "gemv", "gemm_small", "gemm_wide" = rows per workgroup of the 17..64-row kernels (1 / 2 / 3 / 4 = 16 / 32 / 64 / 128, 5 = 16 with self-contained waves; 0 = off),
 * "quantize": 1..999 = the persistent kernel with that many workgroups per CU, 1001 / 1002 / 1004 = the one-shot tiles kernel with 1 / 2 / 4 loads per lane).  variant < 0 (quantize: 0) restores the built-in heuristic.
 * Process-wide (relaxed atomics: safe to flip while other threads launch, each launch
 * reads it once); for sweeps and tests only, not part of the reference surface.
 */
FP4_HIP_API int fp4_hip_set_variant(const char *kernel, int variant);

#ifdef __cplusplus
}
#endif
#endif /* TORCH_BNB_FP4_HIP_H */
