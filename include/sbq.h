/*
 * sbq.h -- C ABI of libsbq.so: the MI355X (gfx950) fake-quantization hot path.
 *
 * This is the drop-in boundary for Sparsebit's Quantizer/Observer/masker path.
 * Every entry point takes raw DEVICE pointers for tensor data, plain sizes and a
 * hipStream_t (passed as void*); the few descriptor arrays (the item lists of the
 * model-wide launches, a list of calibration batches, a step's gradient pointers)
 * are HOST arrays and say so where they are declared.  The caller owns and
 * allocates every buffer (the library never allocates or frees, and keeps no
 * pointer it would dereference later) and all launches are asynchronous on
 * `stream`.  Every function returns an sbq_status (0 == OK) unless it is a size query.
 * Host-side state: none that changes results -- a cache of device attributes, a counter that
 * numbers selections, the address registry of the zero-contract workspaces (sbq_workspace_release
 * below) and the per-THREAD tuning knobs of section 6.
 *
 * Reference interfaces replaced (paths relative to the Sparsebit tree):
 *   - pybind module `fake_quant` (sparsebit/quantization/torch_extensions/
 *     export.cc:3-8, fake_quant_tensor.h:9-36): the four functions
 *     quant_pertensor_forward / quant_perchannel_forward /
 *     quant_pertensor_backward / quant_perchannel_backward.
 *   - the torch-op bodies of the observers
 *     (sparsebit/quantization/observers/{base,minmax,mse,percentile}.py),
 *     LSQ's init (sparsebit/quantization/quantizers/lsq.py:32-51) and the
 *     unstructured L1 masker (sparsebit/sparse/sparsers/l1norm.py:14-26),
 *     which have no native form in the reference.
 *   - pybind module `cuda_kernel` vecquant{4,3,2}matmul / vecgroupquant{4,3,2}matmul
 *     (large_language_models/llama/quantization/cuda/cuda_kernel.cpp:6-62).
 *
 * Tensor geometry.  A contiguous tensor quantized along `ch_axis` is described
 * as [outer, C, inner]: C = shape[ch_axis], inner = prod(shape[ch_axis+1:]),
 * outer = prod(shape[:ch_axis]).  Element i belongs to channel (i / inner) % C
 * (the indexing of fake_quant_tensor.cu:181).  Per-tensor == C 1.
 */
#ifndef SBQ_H_
#define SBQ_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

#define SBQ_VERSION 100 /* 0.1.0 */

/* element types of data tensors */
enum { SBQ_F32 = 0, SBQ_F16 = 1, SBQ_BF16 = 2 };
/* type of the optional integer output of the forward QDQ */
enum { SBQ_Q_NONE = 0, SBQ_Q_I8 = 1, SBQ_Q_I32 = 2, SBQ_Q_I4 = 3 };
/* rounding of x/scale: common.cuh:13-15,64-77 (Python always passes 0) */
enum { SBQ_ROUND_HALF_EVEN = 0, SBQ_ROUND_HALF_UP = 1, SBQ_ROUND_HALF_DOWN = 2 };

typedef enum {
  SBQ_OK = 0,
  SBQ_ERR_DTYPE = 1,     /* unsupported dtype (ref: ValueTypeException, common.cuh:45-49) */
  SBQ_ERR_EMPTY = 2,     /* numel == 0 (ref: InvalidValueException, common.cuh:50-54) */
  SBQ_ERR_NULL = 3,      /* required pointer is NULL */
  SBQ_ERR_ARG = 4,       /* inconsistent sizes / ranges */
  SBQ_ERR_WORKSPACE = 5, /* workspace too small or misaligned */
  SBQ_ERR_LAUNCH = 6,    /* hipGetLastError() != hipSuccess after launch */
  SBQ_ERR_ALIGN = 7,     /* pointer not aligned to its element size */
  SBQ_ERR_BUSY = 8       /* a zero-contract workspace is still in use by a call on another stream */
} sbq_status;

int sbq_version(void);
const char* sbq_strerror(int status);
/* name of the last HIP error seen by this thread's failed launch ("" if none) */
const char* sbq_last_hip_error(void);

/* number of MSE shrink candidates (observers/mse.py:46: `for i in range(80)`) */
#define SBQ_MSE_CANDIDATES 80
/* radix-select digit: 3 passes of 11+11+10 bits over the 32-bit key */
#define SBQ_RADIX_BINS 2048

/* ------------------------------------------------------------------ *
 * 1. Forward quantize-dequantize
 *    y = (clamp(round(x / s) + round(zp), qmin, qmax) - round(zp)) * s
 *    replaces fake_quant_tensor.cu:50-94 (per tensor), :170-224 (per channel);
 *    arithmetic follows the CPU path quant_tensor.py:182-184 (zp rounded
 *    half-to-even, everything in fp32).
 *
 *    x        data, dtype x_dtype, outer*C*inner elements
 *    y        dequantized output, y_dtype == SBQ_F32 or == x_dtype (RNE cast); may be NULL
 *             together with an integer output: quantize only (QuantizeLinear)
 *    q        optional integer tensor (NULL with SBQ_Q_NONE); SBQ_Q_I8 stores
 *             the low 8 bits (needs qmax - qmin <= 255), SBQ_Q_I32 an int32,
 *             SBQ_Q_I4 two levels per byte (element i in the low nibble of byte
 *             i/2 for even i, two's complement when qmin < 0; needs
 *             qmax - qmin <= 15, rows of whole 8-element packs, 16-byte aligned
 *             pointers, half-even rounding and no fused mask -- else SBQ_ERR_ARG)
 *    scale, zero_point   fp32, C elements (1 for per tensor)
 * ------------------------------------------------------------------ */
int sbq_quant_pertensor_forward(const void* x, int x_dtype, void* y, int y_dtype,
                                void* q, int q_type,
                                const float* scale, const float* zero_point,
                                int64_t numel, int qmin, int qmax, int rounding,
                                void* stream);

int sbq_quant_perchannel_forward(const void* x, int x_dtype, void* y, int y_dtype,
                                 void* q, int q_type,
                                 const float* scale, const float* zero_point,
                                 int64_t outer, int64_t C, int64_t inner,
                                 int qmin, int qmax, int rounding, void* stream);

/* DequantizeLinear of stored levels: y = (q - round(zp)) * s in fp32, cast to y_dtype.
 * q_type SBQ_Q_I8 (q_signed: int8, else uint8), SBQ_Q_I4 (two per byte as written by the
 * forward, q_signed: two's complement nibbles) or SBQ_Q_I32.  Bit-identical to the dequantized
 * output of sbq_quant_*_forward for the same levels (quant_tensor.py:184). */
int sbq_dequantize_linear(const void* q, int q_type, int q_signed, void* y, int y_dtype,
                          const float* scale, const float* zero_point,
                          int64_t outer, int64_t C, int64_t inner, void* stream);

/* Multi-tensor launch: the same per-channel QDQ over n_items tensors that share dtype,
 * geometry and integer range (the q/k/v/o projections of a layer, a stack of equal blocks)
 * in ONE kernel.  A 4096x4096 weight is a ~12 us kernel of which ~2 us are launch ramp, tail
 * and the stream boundary; batching amortises them (the reference issues one launch per
 * quantizer per forward, quantizers/base.py:55-64).  `table` is a DEVICE array of
 * n_items x 4 pointers {x, y, scale, zero_point}; results are identical to n_items calls of
 * sbq_quant_perchannel_forward.  Needs 16-byte aligned tensors and inner % 8 == 0. */
#define SBQ_MAX_BATCH 64
int sbq_quant_perchannel_forward_batched(const void* const* table, int n_items,
                                         int x_dtype, int y_dtype,
                                         int64_t outer, int64_t C, int64_t inner,
                                         int qmin, int qmax, void* stream);

/* Model-wide launch: the weights of a whole model -- any mix of [C, inner] shapes, per channel
 * (C > 1) or per tensor (C == 1), each with its own integer range and optionally its own
 * sparsity mask -- quantized by ONE grid.  The reference launches one kernel per layer per
 * forward (modules/conv.py:30-36 -> quantizers/base.py:55-64); for CNN / ViT weights every one
 * of those is pure launch latency.
 *   1. describe the tensors with sbq_group_item (host array);
 *   2. sbq_group_table_build(items, n, NULL, 0, &n_tiles, &bytes) sizes the table, a second call
 *      fills a HOST buffer of `bytes`; copy it to device memory (16-byte aligned) once;
 *   3. sbq_quant_group_forward(device_table, n, n_tiles, ...) as often as needed: pointers are
 *      captured, so tensors must stay where they are (in-place optimizer updates do).
 * Results are identical to per-tensor sbq_quant_per{channel,tensor}_forward / sbq_mask_quant_forward
 * calls.  Constraints per item: contiguous, channel axis 0, inner % 8 == 0, 16-byte aligned x / y,
 * fewer than 2^24 packs (134 M elements); a group is either all masked (1 byte per element) or
 * mask-free, and shares x / y dtypes.  SBQ_GROUP_LSQ applies LSQ's pre-ops to the item inside
 * the kernel (lsq.py:61-62: scale = |scale|, zero_point = clamp(zero_point, qmin, qmax)). */
#define SBQ_GROUP_LSQ 1u
/* the item's `y` is a byte offset (multiple of 16) from the y_base handed to the launch: the
 * outputs of a step live in ONE freshly allocated buffer while the table stays constant */
#define SBQ_GROUP_Y_OFFSET 2u
typedef struct {
  const void* x;
  void* y;
  const float* scale;      /* C values */
  const float* zero_point; /* C values */
  const uint8_t* mask;     /* NULL, or C*inner bytes */
  int64_t C, inner;
  int32_t qmin, qmax;
  uint32_t flags;
  uint32_t reserved;
} sbq_group_item;
int sbq_group_table_build(const sbq_group_item* items, int n_items,
                          void* host_table, size_t host_table_bytes,
                          uint32_t* n_tiles_out, size_t* bytes_needed_out);
int sbq_quant_group_forward(const void* device_table, int n_items, uint32_t n_tiles,
                            int x_dtype, int y_dtype, int has_mask,
                            void* y_base /* NULL: items hold absolute y pointers */, void* stream);

/* Model-wide STE backward: gx of every grouped weight (and the LSQ step-size gradient of those
 * that want one) in two launches -- replaces quant_per{tensor,channel}_backward per layer
 * (fake_quant_tensor.cu:97-132,227-270) and the gs_scaling / abs autograd nodes of
 * lsq.py:13-21,61-76.  gx = mask * STE'(mask * x); gs[c] = sum_row gy * dq/ds, then for
 * SBQ_GROUP_LSQ items gs *= gs_ratio * sign(scale).  Outputs live in caller-provided flat
 * buffers (gx_base: bytes, gs_base: floats) at the item's offsets, so a step allocates two
 * tensors whatever the number of layers.  `gy` is a HOST array of n_items device pointers
 * (contiguous, 16-byte aligned, dtype == x dtype): they change every step and travel as kernel
 * arguments, SBQ_GROUP_BWD_CHUNK items per launch.  Same per-item constraints as the forward. */
#define SBQ_GROUP_BWD_CHUNK 128
typedef struct {
  const void* x;
  const float* scale;
  const float* zero_point;
  const uint8_t* mask; /* NULL or C*inner bytes; all-or-none inside a group */
  uint64_t gx_offset;  /* bytes from gx_base, multiple of 16 */
  uint64_t gs_offset;  /* floats from gs_base (C values written) */
  int64_t C, inner;
  int32_t qmin, qmax;
  uint32_t flags;      /* SBQ_GROUP_LSQ */
  int32_t want_gs;
  float gs_ratio;      /* LSQ: 1/sqrt(inner * qmax) (lsq.py:68-71) */
  uint32_t reserved;
} sbq_group_bwd_item;
int sbq_group_bwd_table_build(const sbq_group_bwd_item* items, int n_items,
                              void* host_table, size_t host_table_bytes,
                              uint32_t* n_wgs_out, uint32_t* n_rows_out,
                              size_t* bytes_needed_out, size_t* workspace_bytes_out);
/* device_table: the built table copied to the GPU; host_table: the same bytes, still on the host */
int sbq_quant_group_backward(const void* device_table, const void* host_table, int n_items,
                             int x_dtype, int gx_dtype, int has_mask,
                             const void* const* gy, void* gx_base, float* gs_base,
                             void* workspace, size_t workspace_bytes, void* stream);

/* Model-wide CALIBRATION: the min-max (and MSE) observers + calc_qparams of every weight of a model in two (four)
 * launches -- replaces the per-layer loop of CalibrationRunner.run_weight_calibration (tools/calibration.py:117-135)
 * over observers/minmax.py:14-25 / observers/mse.py:28-63 and observers/base.py:63-79.
 *   1. describe the tensors with sbq_calib_item (HOST array): [C, inner] contiguous, 16-byte aligned, inner % 8 == 0,
 *      C == 1 for a per-tensor quantizer; out_offset = where the item's C results start in the flat output buffers;
 *   2. sbq_calib_table_build(items, n, NULL, 0, &n_rows, &bytes, &workspace_bytes) sizes the table, a second call fills
 *      a HOST buffer; copy it to device memory (16-byte aligned) once and keep the host copy;
 *   3. sbq_group_minmax_qparams: min_base / max_base (and, unless NULL, scale_base / zp_base) of every row of every
 *      tensor; sbq_group_mse_qparams: the 80-candidate search on those (min, max) -> scale / zero_point / index.
 * The min-max results are bit-identical to the per-tensor calls (sbq_channel_stats + sbq_qparams_from_minmax: same
 * device code, same summation order).  The MSE search sums each candidate's squared errors in a tree of its own since
 * round 6 (a lane per (row, candidate), fp64 across tiles of 1024 elements; knob 2 == 37 at table-build time: round 3's
 * per-tensor tree) and picks the candidate in the same launch: scale / zero_point / index equal those of
 * sbq_mse_accumulate + sbq_mse_select except where two candidates' losses tie to the rounding of an fp32 mean -- there
 * either kernel may name either candidate (as the reference's own answer depends on torch's summation order).  Rows of
 * more than 16 384 elements keep the per-tensor chunk form, at most 96 x 4096 elements (SBQ_ERR_ARG beyond: such tensors
 * go through the per-tensor entry points). */
#define SBQ_CALIB_SYMMETRIC 1u
typedef struct {
  const void* x;
  int64_t C, inner;
  uint64_t out_offset; /* floats from the *_base pointers; C values are written there */
  int32_t qmin, qmax;
  uint32_t flags;      /* SBQ_CALIB_SYMMETRIC */
  uint32_t reserved;
} sbq_calib_item;
int sbq_calib_table_build(const sbq_calib_item* items, int n_items, void* host_table, size_t host_table_bytes,
                          uint32_t* n_rows_out, size_t* bytes_needed_out, size_t* workspace_bytes_out);
/* device_table: the built table copied to the GPU; host_table: the same bytes, still on the host */
int sbq_group_minmax_qparams(const void* device_table, const void* host_table, int x_dtype,
                             float* min_base, float* max_base, float* scale_base, float* zp_base,
                             void* workspace, size_t workspace_bytes, void* stream);
int sbq_group_mse_qparams(const void* device_table, const void* host_table, int x_dtype,
                          const float* min_base, const float* max_base,
                          float* scale_base, float* zp_base, int32_t* index_base /* or NULL */,
                          void* workspace, size_t workspace_bytes, void* stream);

/* LSQ forward / backward on the RAW learnable parameters (quantizers/lsq.py:61-76): the kernels
 * apply scale = |scale| and zero_point = clamp(zero_point, qmin, qmax) themselves, and the backward
 * returns the step-size gradient already multiplied by gs_ratio (lsq.py:13-21,68-71) and
 * sign(scale) -- one launch forward, two backward, instead of the reference's abs / clamp /
 * gs_scaling tensor ops and autograd nodes around K1-K4.  mask: NULL or the sparse layer's
 * byte mask (forward only).  C == 1: per tensor. */
int sbq_quant_lsq_forward(const void* x, int x_dtype, void* y, int y_dtype, const uint8_t* mask,
                          const float* scale, const float* zero_point,
                          int64_t outer, int64_t C, int64_t inner, int qmin, int qmax, void* stream);
int sbq_quant_lsq_backward(const void* x, const void* gy, int x_dtype, void* gx, int gx_dtype,
                           float* gs /* [C] or NULL */, const float* scale, const float* zero_point,
                           int64_t outer, int64_t C, int64_t inner, int qmin, int qmax, float gs_ratio,
                           void* workspace, size_t workspace_bytes /* sbq_backward_workspace_bytes */,
                           void* stream);

/* Fused unstructured mask + QDQ: y = qdq(keep ? x : 0).
 * keep = mask[i] != 0 when `mask` (1 byte/elem, torch.bool) is given, else
 * keep = |x| > *thresh  (l1norm.py:24-25, strict).  Exactly one of mask/thresh
 * must be non-NULL.  Replaces `weight * w_mask` (sparse/modules/conv.py:40,
 * linear.py:31) followed by the weight quantizer. */
int sbq_mask_quant_forward(const void* x, int x_dtype, void* y, int y_dtype,
                           void* q, int q_type,
                           const uint8_t* mask, const float* thresh,
                           const float* scale, const float* zero_point,
                           int64_t outer, int64_t C, int64_t inner,
                           int qmin, int qmax, int rounding, void* stream);

/* ------------------------------------------------------------------ *
 * 2. Backward of the straight-through estimator (STE / LSQ)
 *    replaces fake_quant_tensor.cu:97-167 and :227-308; semantics follow
 *    MySTE.backward (quant_tensor.py:45-71) == kernel K3:
 *      v  = round(x/s) + round(zp)
 *      gx = qmin <= v <= qmax ? gy : 0
 *      gs[c]  = sum gy * (v<qmin ? qmin-zp : v>qmax ? qmax-zp : round(x/s)-x/s)
 *      gzp[c] = sum (qmin <= v <= qmax) ? 0 : -s*gy
 *    gs/gzp may be NULL (not needed); both are fp32 with C elements and are
 *    fully written (no pre-zeroing needed).  x, gy share x_dtype; gx has
 *    gx_dtype (F32 or x_dtype).  workspace: sbq_backward_workspace_bytes().
 * ------------------------------------------------------------------ */
size_t sbq_backward_workspace_bytes(int64_t outer, int64_t C, int64_t inner);

int sbq_quant_pertensor_backward(const void* x, const void* gy, int x_dtype,
                                 void* gx, int gx_dtype, float* gs, float* gzp,
                                 const float* scale, const float* zero_point,
                                 int64_t numel, int qmin, int qmax, int rounding,
                                 void* workspace, size_t workspace_bytes, void* stream);

int sbq_quant_perchannel_backward(const void* x, const void* gy, int x_dtype,
                                  void* gx, int gx_dtype, float* gs, float* gzp,
                                  const float* scale, const float* zero_point,
                                  int64_t outer, int64_t C, int64_t inner,
                                  int qmin, int qmax, int rounding,
                                  void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ *
 * 3. Observer reductions
 * ------------------------------------------------------------------ */

/* Per-channel statistics in one pass over x: min, max (NaN-propagating like
 * torch.min/max; observers/minmax.py:14-25) and sum|x| in fp64 (LSQ init,
 * lsq.py:44-47).  Any of the three outputs may be NULL.  C == 1: per tensor. */
size_t sbq_stats_workspace_bytes(int64_t outer, int64_t C, int64_t inner);
int sbq_channel_stats(const void* x, int x_dtype,
                      int64_t outer, int64_t C, int64_t inner,
                      float* min_out, float* max_out, double* abssum_out,
                      void* workspace, size_t workspace_bytes, void* stream);

/* The streaming per-tensor min-max observer (observers/minmax.py:14-25 over batches that arrive one by one,
 * tools/calibration.py:109-115): ONE launch per batch and nothing to fold.  `state` is uint32[SBQ_MINMAX_STATE_WORDS]
 * owned by the observer (64 slots of one 128-byte line: word 0 of a slot the largest order-preserving key so far,
 * word 1 the smallest; workgroup b updates slot b % 64 -- a thousand workgroups on one address pair serialise at the
 * memory side; a NaN anywhere pins both results to NaN like torch.min / torch.max): sbq_minmax_state_reset once,
 * sbq_minmax_accumulate per batch
 * (x 16-byte aligned, any length; every workgroup updates the state with one integer atomicMax / atomicMin pair --
 * order independent, hence exact and deterministic), sbq_minmax_state_read when calc_qparams wants (min, max). */
#define SBQ_MINMAX_STATE_WORDS 2048
int sbq_minmax_state_reset(uint32_t* state, void* stream);
int sbq_minmax_accumulate(const void* x, int x_dtype, int64_t numel, uint32_t* state, void* stream);
int sbq_minmax_state_read(uint32_t* state, float* min_out, float* max_out, void* stream);

/* Per-channel moments in one pass (fp64): sum x and sum x^2 -- LSQ+ init (mean / std,
 * quantizers/lsq_plus.py:33-38) and the ACIQ-Laplace mean (observers/aciq.py:67-72).  With
 * `center` (fp32 [C]) given, absdev_out receives sum |x - center[c]| instead (the Laplace
 * scale b, aciq.py:68,71) and sum/sumsq may be NULL.  Outputs are ADDED to (zero them first;
 * shards and ranks accumulate / all-reduce with SUM).  Workspace: sbq_stats_workspace_bytes. */
int sbq_channel_moments(const void* x, int x_dtype,
                        int64_t outer, int64_t C, int64_t inner,
                        const float* center, double* sum_out, double* sumsq_out, double* absdev_out,
                        void* workspace, size_t workspace_bytes, void* stream);

/* ACIQ clipping thresholds from the reduced statistics (observers/aciq.py:65-114), fp32 and
 * correctly rounded like the reference's CPU tensor ops (torch's own GPU division is not):
 *   gaus    (b == NULL): std = ((max - min) * gaus_const) / sqrt_2logn;  t = alpha * std
 *   laplace (b != NULL): t = alpha * b
 *   half_range ? (min_out, max_out) = (0, t) : (-t, t)                                      */
int sbq_aciq_thresholds(const float* min_val, const float* max_val, const float* b, int64_t C,
                        float alpha, float gaus_const, float sqrt_2logn, int half_range,
                        float* min_out, float* max_out, void* stream);

/* Exponential moving average over an ordered run of per-sample (min, max) pairs
 * (observers/moving_average.py:23-31): state = first sample if !has_state, then
 * state = ratio * state + (1 - ratio) * sample, in fp32, in order.  state = {min, max}. */
int sbq_ema_minmax(const float* sample_min, const float* sample_max, int64_t n,
                   float ratio, float one_minus_ratio, float* state, int has_state, void* stream);

/* scale / zero_point from min / max, observers/base.py:63-79:
 *   symmetric: s = max(max(-min(min,0), max(max,0)) * 2 / (qmax-qmin), 1e-6), zp = 0
 *   affine:    s = max((max(max,0) - min(min,0)) / (qmax-qmin), 1e-6),
 *              zp = round(-min(min,0) / s)                                   */
int sbq_qparams_from_minmax(const float* min_val, const float* max_val, int64_t C,
                            int qmin, int qmax, int symmetric,
                            float* scale_out, float* zero_point_out, void* stream);

/* Fused min-max observer + qparams + QDQ of a per-channel weight [C, inner] in ONE read of x (4 bytes per element for
 * bf16 in / out instead of 2 + 4 in two launches): observers/minmax.py:14-25 -> observers/base.py:63-79 ->
 * quantizers/base.py:55-64.  Writes y, scale_out / zero_point_out [C] and the observer's min_out / max_out [C].
 * Bit-identical to sbq_channel_stats + sbq_qparams_from_minmax + sbq_quant_perchannel_forward, which is also what
 * it runs for geometries the fused kernel does not take (rows other than 2048 / 4096 elements, unaligned pointers).
 * workspace: as for sbq_channel_stats (sbq_stats_workspace_bytes). */
int sbq_observe_quant_perchannel_forward(const void* x, int x_dtype, void* y, int y_dtype,
                                         float* scale_out, float* zero_point_out, float* min_out, float* max_out,
                                         int64_t C, int64_t inner, int qmin, int qmax, int symmetric,
                                         void* workspace, size_t workspace_bytes, void* stream);

/* Wire format of the cross-GPU min/max exchange: ONE MAX all-reduce over
 * buf[4C] = { max or -inf if NaN, -min or -inf if NaN, isnan(max), isnan(min) }.
 * pack builds it from a rank's local statistics, unpack restores (min, max) with the NaNs
 * (torch.min/max propagate NaN locally; a MAX collective's NaN behaviour is backend defined). */
int sbq_minmax_pack(const float* min_val, const float* max_val, int64_t C, float* buf, void* stream);
int sbq_minmax_unpack(const float* buf, int64_t C, float* min_out, float* max_out, void* stream);

/* LSQ init: scale[c] = 2 * (abssum[c] / count) / sqrt(qmax)   (lsq.py:44-47) */
int sbq_lsq_init_scale(const double* abssum, int64_t C, double count, int qmax,
                       float* scale_out, void* stream);

/* MSE observer (observers/mse.py:28-63).  Step 1 accumulates, for each channel
 * and each of the 80 shrink candidates i, the sum of squared QDQ errors of this
 * rank's data into sse[C][80] (fp64, ADDED to what is there: zero it first;
 * all-reduce it across ranks with SUM between the two steps).  Step 2 picks,
 * per channel, the first candidate with strictly smaller loss (fp32 loss =
 * sse / count) and writes its scale / zero_point / index. */
size_t sbq_mse_workspace_bytes(int64_t outer, int64_t C, int64_t inner);
int sbq_mse_accumulate(const void* x, int x_dtype,
                       int64_t outer, int64_t C, int64_t inner,
                       const float* min_val, const float* max_val,
                       int qmin, int qmax, int symmetric,
                       double* sse, void* workspace, size_t workspace_bytes,
                       void* stream);
int sbq_mse_select(const double* sse, double count_per_channel,
                   const float* min_val, const float* max_val, int64_t C,
                   int qmin, int qmax, int symmetric,
                   float* scale_out, float* zero_point_out, int32_t* best_index_out,
                   void* stream);
/* The same with the count read from DEVICE memory (one double): sharded calibration carries each rank's element
 * count in the buffer of the table's SUM all-reduce, so no host round trip separates the collective from the pick. */
int sbq_mse_select_devcount(const double* sse, const double* count_per_channel_dev,
                            const float* min_val, const float* max_val, int64_t C,
                            int qmin, int qmax, int symmetric,
                            float* scale_out, float* zero_point_out, int32_t* best_index_out,
                            void* stream);

/* ------------------------------------------------------------------ *
 * 4. Order statistics (percentile observer, unstructured-mask threshold)
 *
 * Keys: a float is mapped to a uint32 whose unsigned order equals the float
 * order (-0 == +0, NaN largest like torch.kthvalue / torch.sort); with
 * use_abs != 0 the key is that of |x|.
 * ------------------------------------------------------------------ */

/* Fast path, rows resident on chip: x is [C, inner] (inner <= SBQ_ROWSEL_MAX),
 * one workgroup per row.  observers/percentile.py:16-46:
 *   pos = count(x >= 0), neg = count(x < 0)
 *   max[c] = pos ? kth(row, inner - max(round(pos*alpha), 0)) : 0
 *   min[c] = neg ? kth(row, max(round(neg*alpha), 1))         : 0
 * (k is 1-indexed k-th smallest; round == Python round, half to even).
 * Rows of at most 4096 elements whose ranks lie within 9 of either end (round(inner * alpha) + 1 <= 9: the
 * reference's default alpha = 1e-3) are taken a WAVE per row: the row in registers once, the few largest / smallest
 * keys of every lane in sorted registers (packed two per register for 16-bit inputs), one head popped per rank --
 * 9.3 us for a 4096 x 4096 bf16 weight, 17.9 us for fp32. */
#define SBQ_ROWSEL_MAX 16384
int sbq_percentile_rows(const void* x, int x_dtype, int64_t C, int64_t inner,
                        double alpha, float* min_out, float* max_out, void* stream);

/* General path (any size, any [outer, C, inner] geometry, shardable):
 * three-pass radix select.  Per pass the caller
 *   1. zeroes hist (int64 [C][n_sel][SBQ_RADIX_BINS]),
 *   2. calls sbq_radix_histogram on its shard (adds into hist),
 *   3. (multi-GPU) all-reduces hist with SUM,
 *   4. calls sbq_radix_advance, which picks each selector's bin and updates its
 *      state {prefix, remaining k}.
 * State is int64 [C][n_sel][2] = {prefix key bits found so far, k still to
 * skip (1-indexed within the prefix bucket)}; initialise prefix = 0, k = rank.
 * After pass 2 the prefix is the full 32-bit key; sbq_radix_finish converts it
 * back to float.  n_sel selectors share one read of the data per pass (2 for
 * percentile min+max, 1 for the mask threshold). */
int sbq_radix_histogram(const void* x, int x_dtype,
                        int64_t outer, int64_t C, int64_t inner,
                        int use_abs, int pass, int n_sel, const int64_t* state,
                        int64_t* hist, void* stream);
int sbq_radix_advance(const int64_t* hist, int64_t C, int pass, int n_sel,
                      int64_t* state, void* stream);
int sbq_radix_finish(const int64_t* state, int64_t C, int n_sel, int use_abs,
                     float* values_out /* [C][n_sel] */, void* stream);

/* The whole protocol above enqueued by ONE call, for a single process (nothing to all-reduce
 * between the passes): sbq_percentile_select is the percentile observer over a list of cached
 * batches (`shards`: HOST array of n_shards device pointers, shard i shaped [outers[i], C,
 * inner]; ranks from the first histogram as in sbq_percentile_ranks; min / max [C] with the
 * "no negative -> 0" rule of percentile.py:30-43); sbq_kth_value is the 1-indexed k-th smallest
 * of x (of |x| with use_abs) -- the L1 masker's threshold, l1norm.py:21-24. */
/* Workspace contract of these two calls (as for the mat-vec's arrival counters, section 5): the engine's part of the
 * workspace is a zero-contract region (section 5b): the library zeroes it the first time it sees it, every call
 * leaves it reusable by the next one, and a call on another stream while the last one's stream is still busy is
 * refused with SBQ_ERR_BUSY.  A whole-tensor selection (C == 1) of a 16-bit
 * tensor is ONE launch: every workgroup brackets the wanted ranks from the same 2048-pack sample, sweeps its slabs
 * once, and the last workgroup to arrive resolves the ranks (fp32: one such launch per 11 key bits that remain).
 * Layout: [whole-tensor engine: the part that must start zero][fixed-digit passes (C > 1): self-initialising], so
 * one workspace may serve per-channel and whole-tensor selections in any order.
 * Resident rounds: when the first windows cannot resolve a rank in one sweep -- an extreme rank (k = 1, alpha =
 * 1e-5), a bracket wider than 2048 values (fp16 around zero), half of the data one value -- every workgroup derives
 * that from the same sample, none of them leaves, and the launch sweeps again as a whole (+15-40 us instead of one
 * workgroup sweeping the tensor alone, 0.8-7 ms).  The launch is at most one workgroup per compute unit; waiting
 * workgroups poll a verdict word and hold their compute unit meanwhile, so two such launches running CONCURRENTLY on
 * one device (two streams with their own workspaces, two processes sharing a GPU) can each hold units the other
 * needs.  The wait is therefore bounded: a workgroup that has waited 100 us plus four times its own time to arrival
 * resigns and leaves, the round's last arriver tells the others how many are left, and they share the next sweep by
 * ticket -- in the worst case it sweeps alone.  Concurrent selections are slower, never stuck, always exact
 * (tests/test_gpu_r03.py::test_concurrent_resident_selections_neither_hang_nor_differ). */
size_t sbq_radix_select_workspace_bytes(int64_t C, int n_sel);
int sbq_percentile_select(const void* const* shards, const int64_t* outers, int n_shards, int x_dtype,
                          int64_t C, int64_t inner, double alpha, float* min_out, float* max_out,
                          void* workspace, size_t workspace_bytes, void* stream);
int sbq_kth_value(const void* x, int x_dtype, int64_t numel, int use_abs, int64_t k, float* value_out,
                  void* workspace, size_t workspace_bytes, void* stream);

/* Many whole-tensor selections in ONE launch (fp32 too since round 6: the keys inside each first window stay in LDS for
 * the later rounds): the L1 mask thresholds of a whole model
 * (sparse/sparse_model.py:107-113 sorts every layer's weight on its own).  items: HOST array; item i is the 1-indexed
 * k-th smallest of x_i (of |x_i| with use_abs), written to values_out[i].  Every item runs the one-launch engine of
 * sbq_kth_value on its own share of the grid and its own region of the workspace, so the results are those of
 * n_items sbq_kth_value calls.  Tensors must be 16-byte aligned and hold at least 8 elements (others: sbq_kth_value).
 * Workspace: ZERO before its first use, left reusable by every call, not shared by concurrent calls. */
typedef struct {
  const void* x;
  int64_t numel;
  int64_t k;
} sbq_kth_item;
size_t sbq_group_kth_workspace_bytes(int n_items);
int sbq_group_kth_value(const sbq_kth_item* items, int n_items, int x_dtype, int use_abs, float* values_out,
                        void* workspace, size_t workspace_bytes, void* stream);

/* The percentile observer's two ranks per channel, straight from the pass-0 histogram
 * (after any cross-rank SUM): neg / pos counts are sums over its lower / upper half, then
 * k_min = max(round(neg*alpha), 1), k_max = n - max(round(pos*alpha), 0) with Python's
 * half-to-even round on the fp64 product (percentile.py:27-43).  Writes state
 * [C][2][2] = {prefix 0, rank} for selectors {min, max} and counts_out [2][C] = {neg, pos};
 * no host round trip, no separate sign-count pass.  hist must come from use_abs == 0. */
int sbq_percentile_ranks(const int64_t* hist, int64_t C, int n_sel /* 2 */, double alpha,
                         int64_t* state, int64_t* counts_out, void* stream);

/* counts per channel: neg = count(x < 0), pos = count(x >= 0) (int64 [C] each,
 * ADDED into the outputs) -- percentile.py:27-28 */
int sbq_sign_counts(const void* x, int x_dtype, int64_t outer, int64_t C, int64_t inner,
                    int64_t* neg_out, int64_t* pos_out, void* stream);

/* mask[i] = |x[i]| > *thresh  (1 byte per element; l1norm.py:24-25) */
int sbq_mask_from_threshold(const void* x, int x_dtype, int64_t numel,
                            const float* thresh, uint8_t* mask_out, void* stream);

/* ------------------------------------------------------------------ *
 * 5. GPTQ 4- / 3- / 2-bit grouped mat-vec (config 4)
 *    out[b,n] += sum_k (scales[n,g(k)] * lvl(k,n) - zeros[n,g(k)]) * x[b,k]
 *    replaces vecquant{4,3,2}matmul_cuda and their vecgroupquant* twins
 *    (cuda_kernel.cpp:6-62; cuda_kernel_4bit.cu:36-180, cuda_kernel_3bit.cu:29-196,
 *    cuda_kernel_2bit.cu:29-150).
 *    qweight int32 [rows, out]: each column is a little-endian bit stream over the
 *    rows with `bits` bits per input channel, as written by QuantLinear.pack
 *    (quant.py:187-260) -- rows = ceil(in/8) (4-bit), ceil(in/16) (2-bit),
 *    3*ceil(in/32) (3-bit: 32 levels per 3 words).  scales/zeros fp32 [out, groups],
 *    x fp32 [batch, in], out fp32 [batch, out] pre-filled with the bias
 *    (quant.py:285-289) and accumulated in place.  group_size 0 == one group
 *    (cuda_kernel.cpp:10-16); otherwise a multiple of 128 (4-, 3-bit) or 64 (2-bit).
 *    Deterministic: fixed summation order (the persistent-strip kernel of HBM-sized matrices adds each output
 *    element with ONE float atomic -- a single addend, so still order independent).
 * ------------------------------------------------------------------ */
/* GPTQ's find_params grid search (quant.py:86-104, `mse=True`): for every row of x [rows, inner] (an output channel,
 * or one quantization group of it) the first of n_candidates shrink factors p = 1 - i / grid with the strictly
 * smallest sum |quantize(x; p * xmin, p * xmax) - x|^norm.  xmin / xmax: the row's adjusted extrema (quant.py:71-84);
 * scale_io / zero_io hold the un-shrunk parameters on entry and the chosen ones on return; index_out (or NULL) the
 * chosen i (-1: none was finite).  symmetric: zero stays (maxq + 1) / 2.
 * Rows longer than 16 Ki elements (perchannel=False flattens the whole weight into one) are cut into slices across
 * workgroups with fp64 partial sums per candidate in `workspace` (sbq_gptq_mse_search_workspace_bytes; 0 = none
 * needed, NULL is then fine), folded in slice order; n_candidates <= 128 there. */
size_t sbq_gptq_mse_search_workspace_bytes(int64_t rows, int64_t inner, int n_candidates);
int sbq_gptq_mse_search(const void* x, int x_dtype, int64_t rows, int64_t inner, const float* xmin, const float* xmax,
                        int maxq, int symmetric, float norm, int grid, int n_candidates,
                        float* scale_io, float* zero_io, int32_t* index_out,
                        void* workspace, size_t workspace_bytes, void* stream);

/* Workspace contract: its first SBQ_GPTQ_COUNTER_BYTES bytes are arrival counters of the
 * single-launch K-split fold: a zero-contract region (section 5b) -- zeroed by the library the first
 * time it sees the workspace, left zero by every call, and refused with SBQ_ERR_BUSY when a call on
 * another stream may still be using it. */
#define SBQ_GPTQ_COUNTER_BYTES 262144
size_t sbq_gptq_workspace_bytes(int64_t batch, int64_t in_features, int64_t out_features);
int sbq_vecquant4matmul(const float* x, const int32_t* qweight, float* out,
                        const float* scales, const float* zeros,
                        int64_t batch, int64_t in_features, int64_t out_features,
                        int64_t group_size,
                        void* workspace, size_t workspace_bytes, void* stream);
int sbq_vecquant3matmul(const float* x, const int32_t* qweight, float* out,
                        const float* scales, const float* zeros,
                        int64_t batch, int64_t in_features, int64_t out_features,
                        int64_t group_size,
                        void* workspace, size_t workspace_bytes, void* stream);
int sbq_vecquant2matmul(const float* x, const int32_t* qweight, float* out,
                        const float* scales, const float* zeros,
                        int64_t batch, int64_t in_features, int64_t out_features,
                        int64_t group_size,
                        void* workspace, size_t workspace_bytes, void* stream);

/* Several quantized matrices that share the activation vector -- the q / k / v projections of a decoder layer, the
 * gate and up projections of its MLP -- in ONE launch: the reference issues one mat-vec per QuantLinear
 * (quant.py:262-278), seven launches of 6-9 us per LLaMA-7B decoder layer for 4-8 us of bandwidth work.  All arrays
 * are HOST arrays of n_mats (<= 4) entries; matrix m is [rows(in_features), out_features[m]] with its own scales /
 * zeros / pre-filled out.  Results are those of n_mats single calls up to the fp32 summation order across K blocks
 * (the K split is chosen for the whole launch; equal splits give bit-identical results) -- and single calls are what
 * runs for anything the strip kernel does not take.  Workspace: sbq_vecquantmatmul_multi_workspace_bytes (the larger
 * of the joint launch's need and any single matrix's), same zero-counter contract. */
size_t sbq_vecquantmatmul_multi_workspace_bytes(int64_t batch, int64_t in_features, int n_mats, const int64_t* out_features);
int sbq_vecquantmatmul_multi(int bits, const float* x, int n_mats, const int32_t* const* qweights, float* const* outs,
                             const float* const* scales, const float* const* zeros, const int64_t* out_features,
                             int64_t batch, int64_t in_features, int64_t group_size,
                             void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ *
 * 4b. The same order statistics over data that is spread over RANKS (one process per GPU)
 *
 *  observers/percentile.py:27-43 / sparse/sparsers/l1norm.py:21-24 on the UNION of the ranks' calibration batches,
 *  exact, with ONE read of a 16-bit tensor and two small SUM all-reduces (the library never communicates: the
 *  caller all-reduces the two int64 records, e.g. with RCCL through torch.distributed -- sparsebit_amd/select.py):
 *
 *    sbq_dist_select_sample   this rank's sample histogram + element count  -> int64[SBQ_DIST_SAMPLE_WORDS]
 *    -- SUM all-reduce --
 *    sbq_dist_select_plan     windows from the reduced sample (identical on every rank); clears the workspace
 *    per round (one for 16-bit data, two for fp32; more only when a window missed its rank):
 *      sbq_dist_select_sweep    counts of this rank's shards             -> int64[SBQ_DIST_ROUND_WORDS]
 *      -- SUM all-reduce --
 *      sbq_dist_select_advance  places the ranks on the reduced record; done_out[s] = 1 once selector s is resolved
 *                               (its value is then in out0 / out1).  Rounds after the last needed one are harmless.
 *
 *  percentile != 0: two selectors, ranks from alpha and the union's sign counts (count_signs = 1 in the FIRST sweep),
 *  out0[0] = min side, out1[0] = max side.  percentile == 0: explicit global ranks k0 (k1), 1-based, out0[s].
 *  Shards: flat tensors of counts[i] elements, any alignment, counts[i] == 0 allowed (a rank without data takes
 *  part with zero records).  Workspace: sbq_dist_select_workspace_bytes(), owned by ONE selection from its plan to
 *  its last advance; no zero contract (the plan clears it).
 * ------------------------------------------------------------------ */
#define SBQ_DIST_SAMPLE_WORDS 8193
#define SBQ_DIST_ROUND_WORDS 4100
size_t sbq_dist_select_workspace_bytes(void);
int sbq_dist_select_sample(const void* const* shards, const int64_t* counts, int n_shards, int x_dtype, int use_abs,
                           int64_t* sample_out, void* stream);
int sbq_dist_select_plan(const int64_t* sample, int x_dtype, int n_sel, int percentile, double alpha, int64_t k0, int64_t k1,
                         void* workspace, size_t workspace_bytes, void* stream);
int sbq_dist_select_sweep(const void* const* shards, const int64_t* counts, int n_shards, int x_dtype, int use_abs, int n_sel,
                          int count_signs, void* workspace, size_t workspace_bytes, int64_t* round_out, void* stream);
int sbq_dist_select_advance(const int64_t* round_record, int x_dtype, int n_sel, int percentile, double alpha,
                            void* workspace, size_t workspace_bytes, float* out0, float* out1, int32_t* done_out,
                            void* stream);

/* ------------------------------------------------------------------ *
 * 5b. Zero-contract workspaces
 *
 *  The workspaces of sbq_percentile_select / sbq_kth_value / sbq_group_kth_value (whole-tensor engine) and of the
 *  GPTQ mat-vecs hold arrival counters and histogram copies that every call expects zero and leaves zero.  The
 *  library keeps a registry of the region ADDRESSES it has been handed (it owns nothing):
 *    - first sight of a region: the library zeroes it itself on the call's stream (no hipMemset by the caller);
 *    - a region belongs to the stream of its last call; a call on another stream while that stream still has work
 *      returns SBQ_ERR_BUSY instead of racing for the counters (use one workspace per stream);
 *    - sbq_workspace_release(ws, bytes): forget every region inside [ws, ws + bytes) -- call it before freeing the
 *      memory, or after anything else wrote to it; the next call treats it as new and zeroes it again.
 * ------------------------------------------------------------------ */
int sbq_workspace_release(const void* workspace, size_t workspace_bytes);

/* ------------------------------------------------------------------ *
 * 6. Launch tuning (benchmarks only; defaults are chosen per shape).  Knobs are per calling THREAD.
 * ------------------------------------------------------------------ */
/* knob 0: forward-QDQ variant override (-1 = auto). knob 1: grid cap (0 = auto; in the GPTQ
 * mat-vec: K split when < 128, number of persistent workers when >= 128).
 * knob 2: A/B switches that never change results (3 = IEEE division in the headline QDQ
 * kernel; 1 / 2 = 128 / 64 channels per K lane, 4 = byte converts instead of the e4m3 decode,
 * 9 = two-launch path in the GPTQ mat-vec, 6 = no LDS-DMA prefetch, 8 = plain strip order,
 * 5 = (strip, K block) grid instead of the persistent strip workers on HBM-sized matrices, 13 = round 2's K-split
 * rule --
 * and, in sbq_percentile_rows, the general row kernel instead of the small-rank extraction;
 * 7 = fixed-digit radix engine for whole-tensor selections, 12 = the multi-launch windowed protocol (plan / sweep /
 * advance / fallback launches) instead of the one-launch engine, 15 = an fp32 whole-tensor selection as ONE launch of resident
 * rounds instead of one launch per sweep, 16 = every whole-tensor selection waits for its verdict (resident) even when its plan expects
 * one sweep: +1.5-2 us, measured, 17 = the extraction kernel instead of the sorted lists in sbq_percentile_rows, 11 = general statistics kernel for a min-max-only call;
 * 34 = the grouped fp32 selection without its candidate store (two launches, each sweeps the tensors), 37 (read by
 * sbq_calib_table_build) = round 3's wave-per-row form of the model-wide MSE search,
 * 31 / 32 / 33 = TEST hook of the whole-tensor selections' resident rounds: a waiting workgroup resigns after half a microsecond
 * from round 1 / 2 / 3 on and never before -- results must not change). knob 3: resident schedule of the forward QDQ
 * (0 = auto: tensors that fit the chip's registers in one sitting, 1 = never, 2 = always).
 * Scope: a setting belongs to the CALLING THREAD (two threads' experiments cannot disturb each other).  OR the knob
 * with SBQ_TUNING_PROCESS to set the process-wide default instead -- what every thread that never set the knob itself
 * sees, e.g. autograd's backward threads when the setting is made on the Python main thread. */
#define SBQ_TUNING_PROCESS 0x100
int sbq_set_tuning(int knob, int value);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* SBQ_H_ */
