/*
 * mi355q.h -- C ABI of libmi355q.so: MI355X (gfx950) kernels for the AI Edge
 * Quantizer calibration + requantization hot path.
 *
 * This is the drop-in boundary. The reference has no native
 * code; each entry point below replaces a NumPy/SciPy routine of the reference
 * (cited as `ref:` relative to /root/reference/ai_edge_quantizer/). The Python
 * mirror of the reference's plugin interface (ai-edge-quantizer_amd/mi355q) binds
 * these symbols with ctypes; INTEGRATION.md shows the stub a reference maintainer
 * would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter name ends in _host;
 *   - the caller owns every buffer; the library never allocates, frees or retains
 *     caller memory and keeps no state besides a thread-local error string;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls
 *     only enqueue work, they never synchronize;
 *   - functions return 0 on success or a negative mi355q_status; a message is
 *     available from mi355q_last_error();
 *   - weights are row-major FP32, `rows x cols` with `cols` contiguous (LiteRT
 *     FULLY_CONNECTED / EMBEDDING_LOOKUP layout [out, in]);
 *   - integer outputs are bit-exact with the reference (IEEE division, round half
 *     to even, NaN -> 0); see DESIGN.md for the parity contract.
 */
#ifndef MI355Q_H_
#define MI355Q_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355Q_VERSION 100 /* 0.1.0 */

typedef enum mi355q_status {
  MI355Q_OK = 0,
  MI355Q_BAD_ARG = -1,     /* null pointer, negative size, bad enum value */
  MI355Q_BAD_SHAPE = -2,   /* e.g. cols not divisible by block size */
  MI355Q_UNSUPPORTED = -3, /* valid request this build has no kernel for */
  MI355Q_HIP_ERROR = -4,   /* launch / runtime failure; see mi355q_last_error() */
  MI355Q_RCCL_ERROR = -5,  /* RCCL missing or a collective / communicator call failed */
  MI355Q_IO_ERROR = -6     /* pread / pwrite on a model file failed or came up short */
} mi355q_status;

int32_t mi355q_version(void);
/* Thread-local message of the last failing call on this thread ("" if none). */
const char* mi355q_last_error(void);
/* Device facts the host side uses for launch geometry and bench reporting. */
int32_t mi355q_device_info(int32_t* cu_count_host, int32_t* wavefront_size_host,
                           char* arch_name_host, int32_t arch_name_len);

/* ------------------------------------------------------------------------
 * K1 -- weight min / max.
 * ref: algorithms/uniform_quantize/common_quantize.py:1311-1359 (init_tensor_min_max)
 *
 * x is viewed as [outer, channels, inner] (row-major); min/max are reduced over
 * `outer` and `inner` for every channel:
 *   TENSORWISE            outer=1, channels=1,            inner=numel
 *   CHANNELWISE dim 0     outer=1, channels=shape[0],     inner=numel/shape[0]
 *   CHANNELWISE last dim  outer=numel/shape[-1], channels=shape[-1], inner=1
 *   BLOCKWISE_b (dim 1)   outer=1, channels=rows*cols/b,  inner=b
 * NaN propagates (np.min / np.max semantics).
 * workspace: mi355q_minmax_workspace_bytes(...) bytes of scratch (may be NULL
 * when that returns 0).
 * ------------------------------------------------------------------------ */
size_t mi355q_minmax_workspace_bytes(int64_t outer, int64_t channels, int64_t inner);
int32_t mi355q_minmax_f32(const float* x, int64_t outer, int64_t channels, int64_t inner,
                          float* min_out, float* max_out, void* workspace,
                          size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * K1+K2+K3(+K4) fused -- symmetric min/max requantization of a weight buffer.
 * ref: naive_min_max_quantize.py:34-110 (get_tensor_quant_params) =
 *      common_quantize.py:1311-1359 (min/max)
 *    + uniform_quantize_tensor.py:492-586 (scale; zero point is 0)
 *    + uniform_quantize_tensor.py:273-362 (divide, rint, clip, cast)
 *    + transformations/transformation_utils.py:293-353 (pack_data, optional)
 *
 *   block == 0 : CHANNELWISE on dim 0 -- one scale per row: scale[rows]
 *   block  > 0 : BLOCKWISE along cols -- scale[rows, cols/block], rounded
 *                f32 -> bf16 (RNE) -> f16 -> f32 as the reference does; block must
 *                be a multiple of 4 and divide cols
 *   bits       : 8, 4 or 2 (signed; narrow range [-127,127] only for 8 bits)
 *   clip       : NULL, or per-scale absolute clipping constants (OCTAV) applied as
 *                bound = clip(max|x|, -c, c) (+ the f16 range cap when blockwise)
 *   q_out      : NULL, or int8[rows*cols] unpacked values (UniformQuantParams.quantized_data)
 *   packed_out : NULL, or the bytes quantize_tensor stores in the flatbuffer:
 *                bits=8 -> same as q_out; bits=4 -> rows*cols/2 bytes, element 0 in
 *                the low nibble; bits=2 -> rows*cols/4 bytes. Requires
 *                rows*cols % (8/bits) == 0 (use mi355q_pack_bits for ragged tails).
 *   scale_out  : float[n_scales] (required)
 *   scale_f16_out : NULL, or uint16[n_scales] IEEE half bit patterns of
 *                f16(bf16(scale)) -- what transformations/quantize_tensor.py:129-137
 *                stores in the `<name>_scales` tensor
 * One HBM read of x; scales, q and packed bytes are written once.
 * ------------------------------------------------------------------------ */
int32_t mi355q_requant_sym_f32(const float* x, int64_t rows, int64_t cols, int32_t block,
                               int32_t bits, const float* clip, int8_t* q_out,
                               uint8_t* packed_out, float* scale_out,
                               uint16_t* scale_f16_out, void* stream);

/* Same, over `count` equally shaped weight buffers in ONE launch (model-level
 * batching of ParamsGenerator's per-op loop, ref: params_generator.py:110-183).
 * The pointer tables themselves live in device memory. Any output table may be
 * NULL (that output is skipped for all tensors); clip is not supported here. */
int32_t mi355q_requant_sym_f32_batched(const float* const* x_ptrs, int32_t count,
                                       int64_t rows, int64_t cols, int32_t block,
                                       int32_t bits, int8_t* const* q_ptrs,
                                       uint8_t* const* packed_ptrs,
                                       float* const* scale_ptrs,
                                       uint16_t* const* scale_f16_ptrs, void* stream);

/* The same launch with the pointer tables in HOST memory, read before the call returns: the device pointers travel
 * inside the kernel arguments (16 buffers per dispatch; more than 16 leave as several dispatches), so a group of weights
 * needs no host-to-device copy of its tables in front of its launch -- on an in-order stream that copy's completion
 * signal, not its few hundred bytes, costs ~10 us per launch against 14 us of kernel per 4096 x 4096 buffer. What
 * requant_queue issues for the resident weights of ParamsGenerator's per-op loop (ref: params_generator.py:162-183).
 * Inputs must be 16-byte aligned (outputs as in the device-table form); entries of x / scale tables must not be NULL. */
int32_t mi355q_requant_sym_f32_batched_hostptrs(const float* const* x_ptrs_host, int32_t count,
                                                int64_t rows, int64_t cols, int32_t block,
                                                int32_t bits, int8_t* const* q_ptrs_host,
                                                uint8_t* const* packed_ptrs_host,
                                                float* const* scale_ptrs_host,
                                                uint16_t* const* scale_f16_ptrs_host, void* stream);

/* ------------------------------------------------------------------------
 * K3 -- uniform quantize with given parameters ("T1" path; also asymmetric,
 * tensorwise and channel-last layouts).
 * ref: uniform_quantize_tensor.py:273-362
 *
 * x is viewed as [outer, channels, inner]; scale/zero_point have `channels`
 * entries (1 for TENSORWISE; rows*cols/b with inner=b for BLOCKWISE).
 *   q = cast(clip(rint(x / scale[c] + zp[c]), lo, hi)),  lo = qmin + (narrow ? 1 : 0)
 *   scale_is_f64 : scale points at double[channels]; the division and the zero-point
 *                  add are then done in FP64 as NumPy does for a float64 scale
 *   zero_point   : NULL (all zero) or int32[channels]
 *   zp_via_f64   : the reference adds an int32/int64 zero point in FP64 and rounds
 *                  back to FP32 (np.add(f32, int32, out=f32)); int8/int16 zero points
 *                  are added in FP32. Pass 1 for the former.
 *   out_bits     : container of q_out: 8 -> int8, 16 -> int16, 32 -> int32
 * ------------------------------------------------------------------------ */
int32_t mi355q_quantize_f32(const float* x, int64_t outer, int64_t channels, int64_t inner,
                            const void* scale, int32_t scale_is_f64,
                            const int32_t* zero_point, int32_t zp_via_f64, int32_t bits,
                            int32_t narrow, int32_t out_bits, void* q_out, void* stream);

/* (q - zp) * scale. ref: uniform_quantize_tensor.py:365-409.
 * q is int8/int16/int32 per in_bits; same [outer, channels, inner] view.
 *   diff_bits  : width NumPy subtracts in = promoted type of (q, zero_point):
 *                8 when both are int8 (the difference wraps, as in the reference),
 *                16 / 32 otherwise
 *   out_is_f64 : 0 -> float32 out (int8/int16 times float32 scale);
 *                1 -> double out (NumPy promotes int32 * float32 to float64) */
int32_t mi355q_dequantize_f32(const void* q, int32_t in_bits, int64_t outer,
                              int64_t channels, int64_t inner, const float* scale,
                              const int32_t* zero_point, int32_t diff_bits,
                              int32_t out_is_f64, void* out, void* stream);

/* ------------------------------------------------------------------------
 * float_casting -- float32 weights stored as float16 (round to nearest even, overflow -> inf,
 * subnormals kept), i.e. `weight.astype(np.float16)`.
 * ref: algorithms/nonlinear_quantize/float_casting.py:157-160, 272-275
 * ------------------------------------------------------------------------ */
int32_t mi355q_cast_f32_to_f16(const float* x, int64_t n, uint16_t* out, void* stream);

/* ------------------------------------------------------------------------
 * K4 -- bit packing of int4 / int2 values held in int8 containers.
 * ref: transformations/transformation_utils.py:293-353 (pack_data)
 * out has ceil(n * bits / 8) bytes; element 0 sits in the lowest bits; the ragged
 * tail is zero padded. bits=8 copies.
 * ------------------------------------------------------------------------ */
int32_t mi355q_pack_bits(const int8_t* q, int64_t n, int32_t bits, uint8_t* out,
                         void* stream);
/* The inverse: n sign-extended int8 values from ceil(n * bits / 8) packed bytes. The fused
 * requantization of sub-byte targets writes only the packed bytes the model file stores
 * (4.5 instead of 5.5 bytes of HBM traffic per int4 weight); the int8 containers of
 * UniformQuantParams.quantized_data (ref: uniform_quantize_tensor.py:357-360 yields them,
 * transformations/quantize_tensor.py:176-194 packs them) are produced from those bytes only
 * if a caller actually reads them. */
int32_t mi355q_unpack_bits(const uint8_t* packed, int64_t n, int32_t bits, int8_t* q_out,
                           void* stream);

/* ------------------------------------------------------------------------
 * K7 -- activation statistics: scalar min over x > lo, max over x < hi, with the
 * all-masked fallback to the plain min / max.
 * ref: common_quantize.py:1362-1413 (get_activation_min_max)
 *
 * Processes `count` tensors (one calibration sample each, or any mix) in one
 * launch. `x_ptrs`, `numel` are device tables. minmax_out is float[count][2].
 * use_range == 0 disables the masks (integer-like / no valid range).
 * workspace: mi355q_act_minmax_workspace_bytes(count) bytes.
 * ------------------------------------------------------------------------ */
size_t mi355q_act_minmax_workspace_bytes(int32_t count);
int32_t mi355q_act_minmax_f32(const float* const* x_ptrs, const int64_t* numel,
                              int32_t count, float lo, float hi, int32_t use_range,
                              float* minmax_out, void* workspace, size_t workspace_bytes,
                              void* stream);

/* ------------------------------------------------------------------------
 * K5 -- OCTAV clipping constants, NumPy-order exact.
 * ref: algorithms/uniform_quantize/octav.py:30-112 (_guess_clipping_with_octav)
 *
 * x holds `units` contiguous reduction units of `unit_len` floats (rows for
 * CHANNELWISE on dim 0, blocks for BLOCKWISE_*, the whole tensor for TENSORWISE).
 *   c <- (sum_{x>=c} x - sum_{x<=-c} x) / (n_{|x|>=c} (1-s) + s N),  s = f32(4^-bits / divisor)
 * starting from c = 1 for at most max_iter steps; with early_stop the result is
 * the iterate at which np.allclose(old, new) first holds for ALL units.
 *   count_is_f64 : 1 when the reference reduces over an axis (N is np.int64 and
 *                  s*N is evaluated in float64), 0 for TENSORWISE (axis=None).
 * The float32 sums are accumulated in the order NumPy uses (8192-element chunks,
 * pairwise over runs of selected elements), so clip_out is bit-identical.
 *   clip_out  : float[units];  iters_out: NULL or int32[1] (iterations used)
 *   workspace : mi355q_octav_workspace_bytes(units, max_iter) bytes
 * ------------------------------------------------------------------------ */
size_t mi355q_octav_workspace_bytes(int64_t units, int32_t max_iter);
/* The same plus, for units of 1024 .. 16384 elements (weight rows), room for the hand-over of a
 * row's late iterations to a one-wave tail kernel (the few per cent of a row a converging guess still
 * selects are listed once and re-tested instead of the row): a caller that passes this much gets
 * that path, one that passes mi355q_octav_workspace_bytes() gets every iteration from the row's own
 * workgroup -- same bits either way. */
size_t mi355q_octav_rows_workspace_bytes(int64_t units, int64_t unit_len, int32_t max_iter);
int32_t mi355q_octav_clip_f32(const float* x, int64_t units, int64_t unit_len, int32_t bits,
                              int32_t max_iter, float exponent_divisor, int32_t early_stop,
                              int32_t count_is_f64, float* clip_out, int32_t* iters_out,
                              void* workspace, size_t workspace_bytes, void* stream);
/* K5, one read (opt-in, tolerance class T2; mi355q.ops.octav_mode("fast") / MI355Q_OCTAV_FAST=1).
 * ref: algorithms/uniform_quantize/octav.py:30-112, the same iteration and the same global early stop as
 * mi355q_octav_clip_f32 with an axis given (count_is_f64 = 1). A unit stays in registers as |x| for all iterations:
 * ONE pass over HBM instead of one per iteration. The masked sums are float32 partials per lane (<= 64 elements), a
 * float32 tree over a wave's lanes and a float64 sum over a unit's waves -- not NumPy's order: clip_out agrees with the
 * reference to ~1e-7 relative (scales within 1e-6, integers +-1 on <= 1e-5 of the elements: SURVEY 7's T2), not bit for
 * bit. unit_len: a multiple of 4 in [4, 65536]; x 16-byte aligned; anything else is UNSUPPORTED (use the exact entry).
 * workspace: mi355q_octav_workspace_bytes(units, max_iter). */
int32_t mi355q_octav_clip_fast_f32(const float* x, int64_t units, int64_t unit_len, int32_t bits,
                                   int32_t max_iter, float exponent_divisor, int32_t early_stop,
                                   float* clip_out, int32_t* iters_out, void* workspace,
                                   size_t workspace_bytes, void* stream);

/* General form: x is viewed as [outer, channels, inner] row-major and channel c's unit is the
 * `outer` segments x[o, c, :] (the reduction NumPy does when a middle or last axis is the
 * quantized dimension: DEPTHWISE_CONV_2D weights, BATCH_MATMUL right-hand sides; ref
 * octav.py:186-199 with common_utils.get_reduce_dims). NumPy keeps one running float32 total
 * per channel and adds each segment's runs to it in order; inner == 1 is a plain left-to-right
 * column sum, outer == 1 is the contiguous form above.
 * workspace: mi355q_octav_workspace_bytes(channels, max_iter). */
int32_t mi355q_octav_clip_nd_f32(const float* x, int64_t outer, int64_t channels, int64_t inner,
                                 int32_t bits, int32_t max_iter, float exponent_divisor,
                                 int32_t early_stop, float* clip_out, int32_t* iters_out,
                                 void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * a14 -- MSE scale: scale[u] = multiplier * sqrt(mean(x_u^2)), NumPy-order exact.
 * ref: algorithms/uniform_quantize/mse.py:100-109
 * ------------------------------------------------------------------------ */
int32_t mi355q_mse_scale_f32(const float* x, int64_t units, int64_t unit_len, float multiplier,
                             float* scale_out, void* stream);
/* [outer, channels, inner] view, one scale per channel (see mi355q_octav_clip_nd_f32). */
int32_t mi355q_mse_scale_nd_f32(const float* x, int64_t outer, int64_t channels, int64_t inner,
                                float multiplier, float* scale_out, void* stream);
/* a14 whole: the scale of every contiguous unit AND its integers, q = clip(rint(x / scale[u])) with a zero zero point
 * (ref: mse.py:100-128 -> uniform_quantize_tensor.py:273-362), int8 containers, lo = -2^(bits-1) + (narrow ? 1 : 0).
 * One kernel when a unit's pairwise tree is complete (every length 128 * 2^k, and 8192-multiples of those) and x is
 * 16-byte aligned: the wave that summed a unit quantizes it out of the L2; otherwise mi355q_mse_scale_f32 followed by
 * mi355q_quantize_f32. Same bits either way. */
int32_t mi355q_mse_requant_f32(const float* x, int64_t units, int64_t unit_len, float multiplier, int32_t bits,
                               int32_t narrow, float* scale_out, int8_t* q_out, void* stream);

/* ------------------------------------------------------------------------
 * K6 -- block-diagonal Hadamard rotation: out = reshape(x, (n_vec, h)) @ (H_h / sqrt(h)),
 * H_h the Sylvester matrix, h a power of two <= 16384. In-LDS fast Walsh-Hadamard
 * transform; FP32; out may alias x.
 * ref: algorithms/uniform_quantize/hadamard_rotation.py:48-134
 * ------------------------------------------------------------------------ */
int32_t mi355q_hadamard_rotate_f32(const float* x, int64_t n_vec, int32_t h, float* out,
                                   void* stream);

/* ------------------------------------------------------------------------
 * Dense contraction building blocks (MFMA): C = beta*C + alpha * A.B with
 * A(i,k) at A + i*a_i + k*a_k (element strides; likewise B(k,j), C(i,j)), so
 * transposes and sub-matrices need no copies. float -> v_mfma_f32_32x32x2_f32
 * (exact FP32 fmaf chains, sgemm-class numerics), double -> v_mfma_f64_16x16x4_f64.
 * lower_only writes only j <= i. Used by the GPTQ entry points below; replaces
 * the BLAS calls behind np.matmul / x.T.dot(x) on this path
 * (ref: gptq.py:106, 214; hadamard_rotation.py:129 uses the FWHT instead).
 * ------------------------------------------------------------------------ */
int32_t mi355q_gemm_f32(const float* A, int64_t a_i, int64_t a_k, const float* B, int64_t b_k,
                        int64_t b_j, float* C, int64_t c_i, int64_t c_j, int64_t M, int64_t N,
                        int64_t K, float alpha, float beta, int32_t lower_only, void* stream);
int32_t mi355q_gemm_f64(const double* A, int64_t a_i, int64_t a_k, const double* B, int64_t b_k,
                        int64_t b_j, double* C, int64_t c_i, int64_t c_j, int64_t M, int64_t N,
                        int64_t K, double alpha, double beta, int32_t lower_only, void* stream);

/* ------------------------------------------------------------------------
 * K8 -- GPTQ Hessian of one calibration tensor: hessian_out (FLOAT64 [d,d]) =
 * alpha * (X^T X), X = float32 [n, d] (tokens x channels), X^T X accumulated in
 * FP32 on the MFMA units, alpha = 2 / num_samples applied in FLOAT64 exactly as
 * `(2.0 / num_samples) * x.T.dot(x)` promotes in the reference.
 * ref: algorithms/uniform_quantize/gptq.py:100-107
 * When d is a multiple of 128 and n >= 1024 the float32 products are formed on the
 * bf16 matrix cores: each float32 is split exactly into three bfloat16 and six
 * exact bf16 products per pair are accumulated in FP32 (what is dropped is below
 * half an ulp of each product; csrc/xtx_bf16x3.hip). MI355Q_XTX_FP32_MFMA=1 in the
 * environment selects the FP32-MFMA product for those shapes too. Either way the
 * result is an FP32-accumulated sgemm whose addition order is the kernel's own
 * (tolerance class T2), symmetric bit for bit, and deterministic run to run.
 * workspace: mi355q_gptq_xtx_workspace_bytes(n, d) bytes (X^T X in FP32, plus the
 * bfloat16 planes of one slab of <= 16384 tokens or the split-K partial sums).
 * ------------------------------------------------------------------------ */
size_t mi355q_gptq_xtx_workspace_bytes(int64_t n, int64_t d);
int32_t mi355q_gptq_xtx_f32(const float* x, int64_t n, int64_t d, double alpha,
                            double* hessian_out, void* workspace, size_t workspace_bytes,
                            void* stream);

/* K8 in two steps, for statistics collected over many calibration samples: the sample-weighted
 * mean of (2 / n_i) X_i^T X_i over samples i (what chaining _gptq_merge_hessian yields, ref
 * utils/qsv_utils.py:71-102) is (2 / N) sum_i X_i^T X_i with N = sum n_i, so the float32 product
 * itself is accumulated -- product (float32 [d, d], lower-triangular part valid) (+)= X^T X, the
 * addition in float32 like the K loop's own -- and scaled into FLOAT64 once at the end:
 * hessian_out = alpha * product, mirrored to both triangles. accumulate == 0 starts a product.
 * workspace: mi355q_gptq_xtx_accum_workspace_bytes(n, d) (the bfloat16 planes of one slab of
 * <= 16384 tokens or the split-K partial sums; no room for the product, which is the caller's). */
size_t mi355q_gptq_xtx_accum_workspace_bytes(int64_t n, int64_t d);
int32_t mi355q_gptq_xtx_accum_f32(const float* x, int64_t n, int64_t d, float* product,
                                  int32_t accumulate, void* workspace, size_t workspace_bytes,
                                  void* stream);
int32_t mi355q_gptq_xtx_finish_f64(const float* product, int64_t d, double alpha,
                                   double* hessian_out, void* stream);

/* Sample-weighted running mean of two Hessians, FLOAT64:
 * h_out = (h_cur*n_cur + h_new*n_new) / (n_cur + n_new); h_out may alias an input.
 * ref: utils/qsv_utils.py:71-88 (_gptq_merge_hessian) */
int32_t mi355q_gptq_hessian_merge_f64(const double* h_cur, double n_cur, const double* h_new,
                                      double n_new, int64_t d, double* h_out, void* stream);

/* ------------------------------------------------------------------------
 * K9 -- damped Hessian inverse: diag zeros -> 1, diag += damp*mean(diag), blocked
 * FP64 Cholesky (MFMA trailing updates), triangular inverse, H^-1 = L^-T L^-1;
 * hinv_out is float32 [d,d] (both triangles). info_out (device int32): 0 on
 * success, j+1 if the leading minor of order j+1 is not positive definite
 * (np.linalg.cholesky would raise LinAlgError).
 * ref: algorithms/uniform_quantize/gptq.py:111-128 (_prepare_hessian_inverse)
 * workspace: mi355q_gptq_hinv_workspace_bytes(d) bytes.
 * ------------------------------------------------------------------------ */
size_t mi355q_gptq_hinv_workspace_bytes(int64_t d);
/* Releases the per-device side stream + events the blocked Cholesky creates on first use for
 * its look-ahead (the only state the library keeps between calls). Safe to call at any time. */
int32_t mi355q_shutdown(void);
/* Creates the current device's look-ahead stream and the eight lanes of the batched inverse NOW instead of on
 * the first inverse that needs them (~35 ms; a hardware queue costs 3 - 4 times as much to create later).
 * The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues, least used
 * first: created after an application's stream pools (PyTorch makes 64 streams at its first
 * torch.cuda.Stream()), the look-ahead stream can land on the caller's own hardware queue, where its
 * trailing updates run IN TURN with the panel chain instead of beside it -- 76 instead of 56 ms for a
 * d = 16384 inverse (tools/hinv_after_c5_probe.py; GPU_MAX_HW_QUEUES=16 has the same effect). The host
 * side calls this before its first kernel on a device. Safe to call any number of times. */
int32_t mi355q_prepare_device(void);
/* hipMalloc / hipFree for a workspace the host side wants to own itself (round 4). A fresh GiB of HBM costs its caller
 * ~30 ms, the workspace of a d = 16384 inverse is 5 GiB, and the framework's caching allocator holds its lock (and the
 * interpreter's) for that long; through these two a helper thread of the host side allocates while the thread that
 * feeds the GPU goes on (mi355q/ops.py: HinvWorkspace). mi355q_device_free(NULL) is a no-op; freeing waits for the
 * device like hipFree. */
/* Measurement aid (bench.py `clock_while_timed`): one wave counts shader clocks (clock64) over `seconds` of the constant
 * 100 MHz counter (wall_clock64) and writes {clocks, ticks} to the device int64[2] -- clocks / ticks * 100 = MHz sustained
 * while whatever else is running runs. Launch it on a stream of its own beside the kernel under test. */
int32_t mi355q_clock_probe(double seconds, int64_t* clocks_and_ticks_out, void* stream);
int32_t mi355q_device_alloc(size_t nbytes, void** out);
int32_t mi355q_device_free(void* p);
int32_t mi355q_gptq_hinv_f64(const double* hessian, int64_t d, double damp_factor, float* hinv_out,
                             int32_t* info_out, void* workspace, size_t workspace_bytes,
                             void* stream);

/* The same inverse of hessian = alpha * product, taken straight from the float32 product X^T X that
 * mi355q_gptq_xtx_accum_f32 collected (lower triangle valid): alpha * double(product) is formed where
 * the damped copy reads it, so the 2 GiB float64 Hessian of a d = 16384 layer is never materialized
 * (mi355q_gptq_xtx_finish_f64 makes it only when a caller reads the statistic). Bit-identical to
 * mi355q_gptq_hinv_f64 on the finished Hessian. Same workspace. */
int32_t mi355q_gptq_hinv_from_product_f32(const float* product, int64_t d, double alpha, double damp_factor,
                                          float* hinv_out, int32_t* info_out, void* workspace,
                                          size_t workspace_bytes, void* stream);

/* `count` independent inverses of equally sized Hessians in one call (a model has one per distinct
 * FULLY_CONNECTED input: 54 of order 2048 and 18 of order 16384 in a Gemma-2B; ref gptq.py:111-128 is
 * called once per weight, the calls share nothing). d < 4096, where one inverse is a chain of small
 * dependent kernels that leaves the chip idle: the matrices advance through every step together.
 * d >= 4096: one after the other on the caller's stream (a lone large inverse keeps the machine busy with its
 * own look-ahead update; MI355Q_HINV_PAIRS=1 keeps two in flight, each with a look-ahead stream of its own --
 * measured slower, profiles/r05_hinv_pairs.txt). Every matrix sees the launches of a single call in the
 * same order: results are bit-identical to `count` calls of mi355q_gptq_hinv_f64. The two pointer tables
 * are HOST arrays of device pointers; info_out is device int32[count].
 * workspace: mi355q_gptq_hinv_batched_workspace_bytes(count, d). */
size_t mi355q_gptq_hinv_batched_workspace_bytes(int32_t count, int64_t d);
int32_t mi355q_gptq_hinv_f64_batched(const double* const* hessians_host, int32_t count, int64_t d,
                                     double damp_factor, float* const* hinv_out_host, int32_t* info_out,
                                     void* workspace, size_t workspace_bytes, void* stream);

/* The same for Hessians handed over as hessian[i] = alphas_host[i] * products_host[i] (float32 X^T X, lower
 * triangle valid: see mi355q_gptq_hinv_from_product_f32) -- what calibration leaves behind for the large
 * layers. alphas_host is a HOST array of count doubles. Bit-identical to `count` single calls.
 * workspace: mi355q_gptq_hinv_from_product_batched_workspace_bytes(count, d). */
size_t mi355q_gptq_hinv_from_product_batched_workspace_bytes(int32_t count, int64_t d);
int32_t mi355q_gptq_hinv_from_product_f32_batched(const float* const* products_host, const double* alphas_host,
                                                  int32_t count, int64_t d, double damp_factor,
                                                  float* const* hinv_out_host, int32_t* info_out, void* workspace,
                                                  size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * K10 -- GPTQ weight update + quantization: for each 64-column block, quantize
 * column by column with the up-front scales, propagate err/Hinv[c,c] to the rest
 * of the block (rank-1, FP32, product rounded before the subtraction as NumPy's
 * np.outer does) and then to all later columns with MFMA GEMMs (lazy batch
 * updates: the later blocks of a group of four catch up inside their own
 * kernel, the columns beyond the group get one K = 256 product per group; for
 * rows and d multiples of 128 with d >= 4096 or rows >= 8192 that product runs on
 * the bf16 matrix cores on the exact three-way split of its float32 operands,
 * csrc/xtx_bf16x3.hip -- MI355Q_UPD_FP32_MFMA=1 keeps the FP32 MFMA product).
 * ref: algorithms/uniform_quantize/gptq.py:131-216 (_apply_gptq, blocksize 64)
 *   w [rows, d] float32 (not modified); hinv float32 [d, d]
 *   scale_mode 0: scale[1] (TENSORWISE); 1: scale[rows] (CHANNELWISE);
 *              2: scale[rows, d/block_size] (BLOCKWISE)
 *   scale_is_f64 / zp_via_f64 / diff_bits: NumPy promotion switches, see
 *              mi355q_quantize_f32 / mi355q_dequantize_f32
 *   q_out int8 [rows, d]; workspace: mi355q_gptq_apply_workspace_bytes(rows, d)
 * ------------------------------------------------------------------------ */
size_t mi355q_gptq_apply_workspace_bytes(int64_t rows, int64_t d);
int32_t mi355q_gptq_apply_f32(const float* w, int64_t rows, int64_t d, const float* hinv,
                              const void* scale, int32_t scale_is_f64, const int32_t* zero_point,
                              int32_t scale_mode, int32_t block_size, int32_t bits, int32_t narrow,
                              int32_t zp_via_f64, int32_t diff_bits, int8_t* q_out, void* workspace,
                              size_t workspace_bytes, void* stream);
/* The same sweep for targets of 9..32 bits (ref gptq.py:141-151: `_get_quantized_dtype` gives int16 / int32 containers;
 * the reference's policy admits 2-, 4- and 8-bit weights only, so only a direct caller of get_tensor_quant_params gets
 * here): q_out int32 [rows, d] -- for 9..16 bits the host narrows to the int16 the reference returns --, the general
 * 64-column block kernel (one launch per block: the symmetric fast path packs bytes), the far update per block as for
 * 8 bits; diff_bits 16 wraps q - zp like int16 - int16. From 17 bits on the container is int32 and the reference's
 * float32 arithmetic is kept as it is: clip bounds are the float32 nearest to the integer bounds (2^(bits-1) from 26 bits
 * on), a quotient of 2^31 or a NaN casts to INT32_MIN (x86), q - zp wraps in int32, (q - zp) * scale is a float64
 * product rounded once with the subtraction (NumPy: int32 x float32). Same workspace. */
int32_t mi355q_gptq_apply_wide_f32(const float* w, int64_t rows, int64_t d, const float* hinv,
                                   const void* scale, int32_t scale_is_f64, const int32_t* zero_point,
                                   int32_t scale_mode, int32_t block_size, int32_t bits, int32_t narrow,
                                   int32_t zp_via_f64, int32_t diff_bits, int32_t* q_out, void* workspace,
                                   size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * f4 -- OSCAR (activation-aware channel scaling + optimal clipping), FULLY_CONNECTED weights
 * w float32 [n, d] (n output channels, d input channels). All arithmetic FP64 as in the
 * reference; summation orders are NumPy's. The O(d) vector algebra between these calls
 * (geometric-mean normalisation, clamps) is host NumPy, as in the reference.
 * ref: algorithms/uniform_quantize/oscar.py
 * ------------------------------------------------------------------------ */
/* out[j] = sum_r x[r, j]^2 (rows added in order), divided by rows when mean != 0.
 *   mean=1: calibration statistic mu2 = np.mean(x*x, axis=0)            (ref oscar.py:318-324)
 *   mean=0: column energies (w*w).sum(0)                                 (ref oscar.py:224) */
int32_t mi355q_oscar_col_sumsq_f32(const float* x, int64_t rows, int64_t d, int32_t mean,
                                   double* out, void* stream);
/* Per (row, group of g consecutive columns; g == d or g in {32,64,128,256}):
 *   top = max_j |w[r,j]| * s[j], winner = first argmax, wsq = w[r, winner]^2;
 *   sums_out[k] = np.sum over rows of top^2 for group k (8192-chunk pairwise order).
 * The objective of ref oscar.py:175-194 is sum_k sums_out[k] * mass_k (host); winner / wsq
 * feed mi355q_oscar_winner_energy_f64 (ref oscar.py:236-246).
 *   top2_workspace double [d/g * n]; winner_out int32 [d/g, n]; wsq_out double [d/g, n] */
int32_t mi355q_oscar_group_terms_f32(const float* w, const double* s, int64_t n, int64_t d,
                                     int32_t g, double* top2_workspace, int32_t* winner_out,
                                     double* wsq_out, double* sums_out, void* stream);
/* eff_out[j] = sum over rows in order of wsq[j/g, r] where winner[j/g, r] == j
 * (np.add.at(a_eff, j_star, w[rows, j_star]**2), ref oscar.py:243-246). */
int32_t mi355q_oscar_winner_energy_f64(const int32_t* winner, const double* wsq, int64_t n,
                                       int64_t d, int32_t g, double* eff_out, void* stream);
/* Optimal clip bound of every segment of g consecutive elements of the flattened [n, d]
 * matrix (g divides d, or g == n*d for TENSORWISE): stable descending sort of |w|*s carrying
 * the masses m[j], sequential running sums, closed-form candidate per breakpoint interval,
 * first minimum (ref oscar.py:62-108). u[k] = M_k/(6 qmax^2), noise[k] = M_k/(12 qmax^2) with
 * M_k the group's total mass + 1e-12, k = column group (host FP64).
 * scale_out (optional) is tensor_zp_scale_from_min_max(-bound, bound) of the symmetric signed
 * target: max(bound, 1e-9) / qmax, and with blockwise_scale != 0 rounded FP64 -> float32 ->
 * bfloat16 -> float16 (ref uniform_quantize_tensor.py:553-581), stored as double.
 *   bounds_out / scale_out double [n*d/g], either may be NULL;
 *   workspace: mi355q_oscar_clip_workspace_bytes(n, d, g) (two (key, mass) slabs for the sort's merge passes + one byte per
 *   segment).
 * CHANNELWISE rows (g == d, 1024 <= g <= 16384, qmax >= 7) are answered from a sorted prefix of their largest magnitudes
 * where a convexity argument with an explicit rounding bound shows that no later breakpoint can hold the first minimum;
 * rows where it cannot take the full sort + scan. Same bits either way (csrc/oscar.hip: clip_prefix_kernel).
 * Environment: MI355Q_OSCAR_PREFIX=0 full sort + scan for every row; MI355Q_OSCAR_PREFIX_TARGET=<n> elements asked for. */
int32_t mi355q_oscar_clip_workspace_bytes(int64_t n, int64_t d, int64_t g, size_t* bytes_out);
int32_t mi355q_oscar_clip_bounds_f32(const float* w, const double* s, const double* m, int64_t n,
                                     int64_t d, int64_t g, const double* u, const double* noise,
                                     int32_t qmax, int32_t blockwise_scale, double* bounds_out,
                                     double* scale_out, void* workspace, size_t workspace_bytes,
                                     void* stream);
/* out = clip(rint((w * s[j]) / scale[segment]), qlo, qhi) with FP64 product and quotient
 * (uniform_quantize of the FP64 scaled weight, ref oscar.py:470-478). scale double [n*d/g]. */
int32_t mi355q_oscar_quantize_f32(const float* w, const double* s, const double* scale, int64_t n,
                                  int64_t d, int64_t g, int32_t qlo, int32_t qhi, int8_t* out,
                                  void* stream);

/* ------------------------------------------------------------------------
 * dequantized_weight_recovery -- scales of fake-quantized (QAT) weights and the recovery check.
 * ref: algorithms/uniform_quantize/dequantized_weight_recovery.py:48-61, 118-186, 30-45
 * ------------------------------------------------------------------------ */
/* scale_out[k] = smallest positive step between the sorted magnitudes (0 appended) of segment k
 * of g consecutive elements of w [n, d] (g divides d, or g == n*d), floored at 1e-9.
 *   rounded != 0: float32 arithmetic of the channel- / blockwise path (steps rounded to float32,
 *   only steps > float32(1e-9) count, floor float32(1e-9)); rounded == 0: the TENSORWISE path's
 *   float64. workspace: mi355q_oscar_clip_workspace_bytes(n, d, g). */
int32_t mi355q_dwr_scales_f32(const float* w, int64_t n, int64_t d, int64_t g, int32_t rounded,
                              double* scale_out, void* workspace, size_t workspace_bytes,
                              void* stream);
/* *max_out = max |q * scale[e / g] - w| in FP64 (NaN if any), the quantity
 * _validate_recovered_weights compares with its tolerance. */
int32_t mi355q_dwr_max_error_f32(const float* w, const int8_t* q, const double* scale, int64_t total,
                                 int64_t g, double* max_out, void* stream);

/* ------------------------------------------------------------------------
 * Multi-GPU exchange steps: one process per GPU, RCCL over xGMI. The
 * reference is single-process; these define what the sharded calibration adds and must
 * reproduce. `comm` is an opaque RCCL communicator (ncclComm_t) owned by the caller:
 *   rank 0: mi355q_comm_unique_id(id)  ->  the host broadcasts the 128 bytes over its own
 *   rendezvous  ->  every rank: mi355q_comm_init_rank(&comm, nranks, id, rank) with its GPU
 *   current. Collectives enqueue on `stream` and return; all ranks must call them in the same
 *   order. RCCL is bound at run time (the librccl.so.1 already in the process under
 *   PyTorch-ROCm); without it these calls -- and only these -- return MI355Q_RCCL_ERROR.
 * ------------------------------------------------------------------------ */
#define MI355Q_UNIQUE_ID_BYTES 128
int32_t mi355q_comm_unique_id(char* id_host /* [MI355Q_UNIQUE_ID_BYTES] */);
int32_t mi355q_comm_init_rank(void** comm_out_host, int32_t nranks, const char* id_host, int32_t rank);
int32_t mi355q_comm_info(void* comm, int32_t* nranks_host, int32_t* rank_host);
int32_t mi355q_comm_destroy(void* comm);
/* X1 -- per-sample activation statistics of every rank to every rank: out[r*n .. r*n+n) =
 * rank r's local[0..n) (float (min, max) pairs, n floats per rank, equal on all ranks: shards
 * are padded to the longest). The host then replays the order-dependent moving average
 * (ref: utils/qsv_utils.py:43-68, calibrator.py:395-421) over all samples in dataset order. */
int32_t mi355q_allgather_minmax(void* comm, const float* local, int64_t n, float* out, void* stream);
/* X1 fast path -- global ranges when the update rule is min_max_update (ref:
 * utils/qsv_utils.py:105-122; associative and commutative): in-place all-reduce(min) of mins[n]
 * and all-reduce(max) of maxs[n], issued as one RCCL group. */
int32_t mi355q_allreduce_minmax_f32(void* comm, float* mins, float* maxs, int64_t n, void* stream);
/* In-place all-reduce(sum) building blocks (OSCAR's sample-weighted second moments, counts). */
int32_t mi355q_allreduce_sum_f32(void* comm, float* buf, int64_t n, void* stream);
int32_t mi355q_allreduce_sum_f64(void* comm, double* buf, int64_t n, void* stream);
/* X2 -- GPTQ Hessian over sample-sharded calibration: hessian (FLOAT64 [d, d], this rank's
 * sample-weighted mean over its own samples, as chaining mi355q_gptq_hessian_merge_f64 yields) is
 * scaled by weight = n_rank / N and all-reduced(sum) in place; every rank ends with the mean over
 * all N samples, which is what chaining _gptq_merge_hessian over the whole dataset yields up to
 * FP64 rounding (ref: utils/qsv_utils.py:71-102). One collective of d*d*8 bytes per distinct
 * Hessian: 32 MiB at d = 2048, 2 GiB at d = 16384. A rank without samples passes zeros, weight 0. */
int32_t mi355q_allreduce_hessian_f64(void* comm, double* hessian, int64_t d, double weight, void* stream);
/* X2 at half the bytes, and to the one rank that needs it: the Hessian is symmetric, so only its
 * packed lower triangle (d (d + 1) / 2 doubles: 1 GiB at d = 16384) travels -- packed = weight *
 * lower(hessian) in `workspace`, one all-reduce(sum) (root < 0: every rank ends with the mean) or one
 * reduce(sum) to `root` (under GPTQ exactly one rank reads a given Hessian: the owner of the ops
 * whose input it belongs to; a ring reduce moves half of what a ring all-reduce does), then the
 * receiving ranks unpack to both triangles in place. On the other ranks `hessian` is left as it was.
 * The sum is the same FP64 sum of the same weighted entries as in mi355q_allreduce_hessian_f64.
 * workspace: mi355q_hessian_exchange_workspace_bytes(d) bytes. */
size_t mi355q_hessian_exchange_workspace_bytes(int64_t d);
int32_t mi355q_reduce_hessian_f64(void* comm, double* hessian, int64_t d, double weight, int32_t root,
                                  void* workspace, size_t workspace_bytes, void* stream);
/* X2 in the form the statistic is kept in: the mean of ref utils/qsv_utils.py:71-102 over all samples is
 * (2 / N) * sum over ranks of P_rank, P_rank = sum of X^T X over the rank's own samples -- the FLOAT32
 * product mi355q_gptq_xtx_accum_f32 accumulates (lower triangle valid). Only that triangle travels, as
 * float32: d (d + 1) / 2 * 4 bytes, 0.5 GiB at d = 16384 (a quarter of the float64 all-reduce, half of the
 * packed float64 reduce), summed across ranks in float32 -- the precision in which a single process adds
 * its own slabs' products -- by one reduce(sum) to `root` (root < 0: all-reduce). The receiving ranks end
 * with the sum in `product`'s lower triangle and keep the statistic in product form
 * (mi355q_gptq_hinv_from_product_f32 with alpha = 2 / N reads it as it is); no float64 d x d array is
 * made on any rank. product == NULL: this rank saw no sample (it contributes zeros and must not be `root`).
 * workspace: mi355q_product_exchange_workspace_bytes(d) bytes. */
size_t mi355q_product_exchange_workspace_bytes(int64_t d);
int32_t mi355q_reduce_product_f32(void* comm, float* product, int64_t d, int32_t root, void* workspace,
                                  size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------
 * Model file <-> HBM without a host copy of the weights (the file path's io ring).
 *   ref: utils/tfl_flatbuffer_utils.py:142-163  the model file is mapped / read whole into host memory;
 *        model_modifier.py:290-391              the serializers copy every quantized buffer into one host
 *                                               bytearray and write it out
 * mi355q_file_to_device: bytes [file_offset, file_offset + nbytes) of the open file `fd` -> `dst` (device):
 * pread() on the library's io threads into a ring of three pinned 8 MiB slots (made on first use, per
 * device), hipMemcpyAsync on `copy_stream` (the caller orders its compute stream behind it). Returns when the
 * last copy is ENQUEUED and every read is done. mi355q_device_to_file: `src` (device; the caller has ordered
 * `copy_stream` behind its producer) -> the file at file_offset: asynchronous copies into the ring, pwrite()
 * from there on the io threads; returns with the last writes possibly in flight -- mi355q_file_io_finish()
 * waits for them and reports the first write error. A short read (end of file) or a failing pread / pwrite is
 * MI355Q_IO_ERROR. One transfer enqueues at a time per process; mi355q_shutdown() releases ring and threads. */
int32_t mi355q_file_to_device(int32_t fd, int64_t file_offset, int64_t nbytes, void* dst, void* copy_stream);
int32_t mi355q_device_to_file(const void* src, int64_t nbytes, int32_t fd, int64_t file_offset, void* copy_stream);
int32_t mi355q_file_io_finish(void);

/* The same transfers without holding the caller (round 4). mi355q_file_to_device returns when the file has been READ --
 * a model's weights cross at the ring's rate while the calling thread, the one that walks the ops and launches the
 * kernels, stands still (0.62 s of a 3.6 s Gemma-2B GPTQ run; 30 of the 60 ms of a 1.4 GB file -> file run).
 *   mi355q_file_io_submit_upload    queues the transfer on the library's upload thread (first in, first out) and
 *                                   returns a ticket at once; `dst` and `fd` stay valid until the ticket is waited for.
 *   mi355q_file_io_wait             blocks until the ticket's last copy is ENQUEUED on its copy stream (an event the
 *                                   caller records on that stream afterwards lies behind all of them), frees the
 *                                   ticket and returns the transfer's status (a short or failing read is
 *                                   MI355Q_IO_ERROR here; nothing of a spoiled slot was copied).
 *   mi355q_file_io_submit_download  queues `src` -> file on the download thread (its own ring: uploads and
 *                                   downloads run at the same time). `ready_event`, when not null, is a hipEvent_t
 *                                   the caller has RECORDED behind the payload's producer: the download thread waits for it
 *                                   (hipEventSynchronize) before it enqueues the first copy -- for that producer, not
 *                                   for everything queued on the compute stream. No ticket:
 *                                   mi355q_file_io_finish waits for every submitted transfer of both directions
 *                                   and the writes behind them, and reports the first failure.
 * ref: the same two call sites (utils/tfl_flatbuffer_utils.py:142-163, model_modifier.py:290-391). */
int32_t mi355q_file_io_submit_upload(int32_t fd, int64_t file_offset, int64_t nbytes, void* dst, void* copy_stream,
                                     int64_t* ticket);
int32_t mi355q_file_io_wait(int64_t ticket);
int32_t mi355q_file_io_submit_download(const void* src, int64_t nbytes, int32_t fd, int64_t file_offset,
                                       void* copy_stream, void* ready_event);
/* The same download with memory as its destination (round 5): `dst` is the payload's place inside the OUTPUT FILE'S OWN
 * shared mapping, whose pages exist already (the host allocates them ahead of time, LiteRTLMFile.prepare_output). The io
 * threads then copy each staged piece into the mapping instead of pwrite()ing it: buffered writes of several threads to
 * one file take turns on its inode lock (11 GB/s into allocated pages with 4 threads, 7 into fresh ones), copies into
 * mapped pages do not (22 - 30 GB/s with 4 - 8 threads; tools/tmpfs_write_probe.py). Same queue, gate and completion
 * (mi355q_file_io_finish) as mi355q_file_io_submit_download; `dst` .. `dst + nbytes` must stay mapped until then.
 * ref: model_modifier.py:290-391 (the reference copies every quantized buffer into one host bytearray). */
int32_t mi355q_file_io_submit_download_mapped(const void* src, int64_t nbytes, void* dst, void* copy_stream,
                                              void* ready_event);

#ifdef __cplusplus
} /* extern "C" */
#endif
#endif /* MI355Q_H_ */
