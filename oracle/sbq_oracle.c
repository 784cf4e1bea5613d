/*
 * sbq_oracle.c -- CPU restatement of Sparsebit's fake-quantization hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call
 * it, and only as the checker / the timed CPU baseline -- never as a product
 * path.  sparsebit_amd/ must not import anything from oracle/.
 *
 * Pinning: the reference holds no golden vectors for this path;
 * this restatement is pinned against outputs of the reference's own CPU
 * implementation run in the authoring container (the .npz files under tests/golden/, generated
 * by tests/golden/gen_golden.py) and against the hand-checked KATs of
 * SURVEY.md 8c (tests/test_oracle_golden.py).
 *
 * Every function cites the reference lines it restates (paths relative to the
 * Sparsebit tree).  Plain scalar C, strict IEEE fp32: build with
 * -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* torch.minimum / torch.maximum / torch.min / torch.max propagate NaN */
static float nan_min(float a, float b) { return (a != a || a < b) ? a : ((b != b) ? b : (b < a ? b : a)); }
static float nan_max(float a, float b) { return (a != a || a > b) ? a : ((b != b) ? b : (b > a ? b : a)); }

/* torch.clamp(v, lo, hi) propagates NaN */
static float clampf(float v, float lo, float hi) {
  if (v != v) return v;
  return v < lo ? lo : (v > hi ? hi : v);
}

/* ------------------------------------------------------------------------
 * ort_fake_quant, CPU branch: sparsebit/quantization/quantizers/quant_tensor.py:182-184
 *     zp  = zero_point.round()
 *     x_q = torch.clamp((x_f / scale).round() + zp, qmin, qmax)
 *     x_dq = (x_q - zp) * scale
 * channel of element i: (i / inner) % C  (fake_quant_tensor.cu:181; the broadcast
 * shape produced by Quantizer._broadcast_qparams, quantizers/base.py:97-100).
 * q_out (optional) receives x_q as int32.
 * ------------------------------------------------------------------------ */
ORC_API void orc_qdq(const float* x, int64_t outer, int64_t C, int64_t inner, const float* scale,
                     const float* zero_point, int qmin, int qmax, float* dq_out, int32_t* q_out) {
  const int64_t n = outer * C * inner;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t c = (i / inner) % C;
    const float s = scale[c];
    const float zp = rintf(zero_point[c]);
    const float t = x[i] / s;
    const float xq = clampf(rintf(t) + zp, (float)qmin, (float)qmax);
    dq_out[i] = (xq - zp) * s;
    if (q_out) q_out[i] = (xq != xq) ? 0 : (int32_t)xq;
  }
}

/* `weight * w_mask` (sparsebit/sparse/modules/conv.py:40, linear.py:31) followed by
 * the weight quantizer: qdq(x * mask), mask a torch.bool tensor (1 byte/element). */
ORC_API void orc_mask_qdq(const float* x, const uint8_t* mask, int64_t outer, int64_t C, int64_t inner,
                          const float* scale, const float* zero_point, int qmin, int qmax,
                          float* dq_out, int32_t* q_out) {
  const int64_t n = outer * C * inner;
  float* tmp = (float*)malloc((size_t)n * sizeof(float));
  for (int64_t i = 0; i < n; ++i) tmp[i] = x[i] * (mask[i] ? 1.0f : 0.0f);
  orc_qdq(tmp, outer, C, inner, scale, zero_point, qmin, qmax, dq_out, q_out);
  free(tmp);
}

/* ------------------------------------------------------------------------
 * Observer.calc_qparams_with_minmax: sparsebit/quantization/observers/base.py:63-79
 * ------------------------------------------------------------------------ */
static void qparams_one(float mn, float mx, int qmin, int qmax, int symmetric, float* scale, float* zp) {
  const float min_neg = nan_min(mn, 0.0f);
  float max_pos = nan_max(mx, 0.0f);
  const float qrange = (float)(qmax - qmin);
  if (symmetric) {
    max_pos = nan_max(-min_neg, max_pos);
    *scale = nan_max(max_pos * 2.0f / qrange, 1e-6f);
    *zp = 0.0f;
  } else {
    *scale = nan_max((max_pos - min_neg) / qrange, 1e-6f);
    *zp = rintf(-min_neg / *scale);
  }
}

ORC_API void orc_qparams_from_minmax(const float* min_val, const float* max_val, int64_t C, int qmin,
                                     int qmax, int symmetric, float* scale_out, float* zp_out) {
  for (int64_t c = 0; c < C; ++c)
    qparams_one(min_val[c], max_val[c], qmin, qmax, symmetric, &scale_out[c], &zp_out[c]);
}

/* ------------------------------------------------------------------------
 * minmax observer: sparsebit/quantization/observers/minmax.py:14-25 over the
 * channel-first view built by DataCache.get_data_for_calibration
 * (observers/base.py:21-36).  C == 1: per tensor.
 * ------------------------------------------------------------------------ */
ORC_API void orc_minmax(const float* x, int64_t outer, int64_t C, int64_t inner, float* min_out,
                        float* max_out) {
  for (int64_t c = 0; c < C; ++c) {
    float mn = INFINITY, mx = -INFINITY;
    for (int64_t o = 0; o < outer; ++o) {
      const float* row = x + (o * C + c) * inner;
      for (int64_t i = 0; i < inner; ++i) {
        mn = nan_min(mn, row[i]);
        mx = nan_max(mx, row[i]);
      }
    }
    min_out[c] = mn;
    max_out[c] = mx;
  }
}

/* ------------------------------------------------------------------------
 * LSQ init: sparsebit/quantization/quantizers/lsq.py:44-47
 *     scale = 2 * x_oc.abs().mean(axis=1) / math.sqrt(qmax)
 * The mean is accumulated in fp64 here (torch sums fp32 in a blocked order that
 * is not worth restating); the comparison tolerance is 1e-6 relative.
 * ------------------------------------------------------------------------ */
ORC_API void orc_lsq_init_scale(const float* x, int64_t outer, int64_t C, int64_t inner, int qmax,
                                float* scale_out) {
  const float sq = (float)sqrt((double)qmax);
  for (int64_t c = 0; c < C; ++c) {
    double acc = 0.0;
    for (int64_t o = 0; o < outer; ++o) {
      const float* row = x + (o * C + c) * inner;
      for (int64_t i = 0; i < inner; ++i) acc += fabs((double)row[i]);
    }
    const float mean = (float)(acc / (double)(outer * inner));
    scale_out[c] = (2.0f * mean) / sq;
  }
}

/* ------------------------------------------------------------------------
 * MSE observer: sparsebit/quantization/observers/mse.py:28-63 + observers/utils.py:1-5.
 * For i in [0, 80): shrink (min,max) by the fp32 factor (1 - 0.01 i), derive
 * scale/zp, fake-quantize, loss = mean((x - dq)^2); keep the FIRST strictly
 * smaller loss.  Per channel the scale is applied per row (the CUDA kernel's
 * indexing, fake_quant_tensor.cu:181-186) -- the reference's CPU broadcast of a
 * flat [C] scale is a bug and is not reproduced.
 * The squared error is accumulated in fp64 and the loss compared in fp32.
 * Outputs: scale/zp/best index per channel and (optional) all losses [C][80].
 * ------------------------------------------------------------------------ */
ORC_API void orc_mse(const float* x, int64_t outer, int64_t C, int64_t inner, int qmin, int qmax,
                     int symmetric, float* scale_out, float* zp_out, int32_t* best_out,
                     double* sse_out /* [C][80] or NULL */) {
  float* mn = (float*)malloc((size_t)C * sizeof(float));
  float* mx = (float*)malloc((size_t)C * sizeof(float));
  orc_minmax(x, outer, C, inner, mn, mx);
  const double count = (double)(outer * inner);
  for (int64_t c = 0; c < C; ++c) {
    float loss_min = 1e10f, best_s = 1.0f, best_z = 0.0f;
    int best = -1;
    for (int i = 0; i < 80; ++i) {
      const float f = (float)(1.0 - (double)i * 0.01);
      float s, z;
      qparams_one(mn[c] * f, mx[c] * f, qmin, qmax, symmetric, &s, &z);
      double sse = 0.0;
      for (int64_t o = 0; o < outer; ++o) {
        const float* row = x + (o * C + c) * inner;
        for (int64_t k = 0; k < inner; ++k) {
          const float xq = clampf(rintf(row[k] / s) + z, (float)qmin, (float)qmax);
          const float d = row[k] - (xq - z) * s;
          sse += (double)(d * d);
        }
      }
      if (sse_out) sse_out[c * 80 + i] = sse;
      const float loss = (float)(sse / count);
      if (loss < loss_min) {
        loss_min = loss;
        best_s = s;
        best_z = z;
        best = i;
      }
    }
    scale_out[c] = best_s;
    zp_out[c] = best_z;
    if (best_out) best_out[c] = best;
  }
  free(mn);
  free(mx);
}

/* ------------------------------------------------------------------------
 * percentile observer: sparsebit/quantization/observers/percentile.py:16-46.
 * rows = C channel-first rows of n elements (per tensor: one row).
 *   pos = count(x >= 0), neg = count(x < 0)
 *   max = pos ? kthvalue(row, n - max(round(pos*alpha), 0)) : 0
 *   min = neg ? kthvalue(row, max(round(neg*alpha), 1))      : 0
 * kthvalue = k-th smallest, 1-indexed; round() is Python's (half to even on the
 * double product) == rint().
 * ------------------------------------------------------------------------ */
static int cmp_float(const void* a, const void* b) {
  const float x = *(const float*)a, y = *(const float*)b;
  /* NaN sorts last, like torch.kthvalue / torch.sort */
  if (x != x) return (y != y) ? 0 : 1;
  if (y != y) return -1;
  return (x > y) - (x < y);
}

ORC_API void orc_percentile(const float* x, int64_t C, int64_t n, double alpha, float* min_out,
                            float* max_out) {
  float* tmp = (float*)malloc((size_t)n * sizeof(float));
  for (int64_t c = 0; c < C; ++c) {
    const float* row = x + c * n;
    int64_t pos = 0, neg = 0;
    for (int64_t i = 0; i < n; ++i) {
      pos += row[i] >= 0.0f;
      neg += row[i] < 0.0f;
    }
    memcpy(tmp, row, (size_t)n * sizeof(float));
    qsort(tmp, (size_t)n, sizeof(float), cmp_float);
    float mx = 0.0f, mn = 0.0f;
    if (pos > 0) {
      double r = rint((double)pos * alpha);
      if (r < 0.0) r = 0.0;
      int64_t k = n - (int64_t)r;
      if (k < 1) k = 1;
      mx = tmp[k - 1];
    }
    if (neg > 0) {
      double r = rint((double)neg * alpha);
      if (r < 1.0) r = 1.0;
      int64_t k = (int64_t)r;
      if (k > n) k = n;
      mn = tmp[k - 1];
    }
    min_out[c] = mn;
    max_out[c] = mx;
  }
  free(tmp);
}

/* ------------------------------------------------------------------------
 * unstructured L1 mask: sparsebit/sparse/sparsers/l1norm.py:18-26
 *     thresh = sort(abs(x).flatten())[min(int(n*ratio), n-1)];  mask = abs(x) > thresh
 * `thresh_idx` is the already computed index (Python evaluates int(n*ratio)).
 * Returns the threshold.
 * ------------------------------------------------------------------------ */
ORC_API float orc_l1_mask(const float* x, int64_t n, int64_t thresh_idx, uint8_t* mask_out) {
  float* tmp = (float*)malloc((size_t)n * sizeof(float));
  for (int64_t i = 0; i < n; ++i) tmp[i] = fabsf(x[i]);
  qsort(tmp, (size_t)n, sizeof(float), cmp_float);
  if (thresh_idx > n - 1) thresh_idx = n - 1;
  const float thresh = tmp[thresh_idx];
  free(tmp);
  if (mask_out)
    for (int64_t i = 0; i < n; ++i) mask_out[i] = fabsf(x[i]) > thresh;
  return thresh;
}

/* ------------------------------------------------------------------------
 * STE / LSQ backward: MySTE.backward (quant_tensor.py:45-71) reduced to the
 * parameter shapes like the per-tensor CUDA kernel (fake_quant_tensor.cu:97-132):
 *     v   = round(x/s) + round(zp)
 *     gx  = (qmin <= v <= qmax) ? gy : 0
 *     gs[c]  = sum gy * (v<qmin ? qmin-zp : v>qmax ? qmax-zp : round(x/s) - x/s)
 *     gzp[c] = sum (qmin <= v <= qmax) ? 0 : -s*gy
 * (the per-channel CUDA kernel's `vq < qmax` at :264 is an off-by-one and is not
 * reproduced).  Sums in fp64, returned as fp32.
 * ------------------------------------------------------------------------ */
ORC_API void orc_ste_backward(const float* x, const float* gy, int64_t outer, int64_t C, int64_t inner,
                              const float* scale, const float* zero_point, int qmin, int qmax,
                              float* gx_out, float* gs_out, float* gzp_out) {
  double* gs = (double*)calloc((size_t)C, sizeof(double));
  double* gz = (double*)calloc((size_t)C, sizeof(double));
  const int64_t n = outer * C * inner;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t c = (i / inner) % C;
    const float s = scale[c];
    const float zp = rintf(zero_point[c]);
    const float t = x[i] / s;
    const float r = rintf(t);
    const float v = r + zp;
    const int below = v < (float)qmin, above = v > (float)qmax;
    const int inside = !(below || above);
    float pgs = (r - t) * gy[i];
    if (above) pgs = ((float)qmax - zp) * gy[i];
    if (below) pgs = ((float)qmin - zp) * gy[i];
    gs[c] += (double)pgs;
    gz[c] += inside ? 0.0 : (double)(-s * gy[i]);
    gx_out[i] = inside ? gy[i] : 0.0f;
  }
  for (int64_t c = 0; c < C; ++c) {
    if (gs_out) gs_out[c] = (float)gs[c];
    if (gzp_out) gzp_out[c] = (float)gz[c];
  }
  free(gs);
  free(gz);
}

/* ------------------------------------------------------------------------
 * GPTQ 4-bit mat-vec: Quant4Matmul.forward + VecQuant4MatMulKernel
 * (large_language_models/llama/quantization/utils/quant.py:281-307,
 *  cuda/cuda_kernel_4bit.cu:88-180):
 *   out[b,n] += sum_k (scales[n,g]*nib(k,n) - zeros[n,g]) * x[b,k],  g = k / group_size
 * qweight int32 [rows, out] (4-bit: 8 input-channel nibbles per word, low first,
 * QuantLinear.pack, quant.py:224-229; 2- and 3-bit: see orc_stream_level);
 * scales/zeros [out, groups]; `out` pre-filled with the bias.  The reference accumulates fp32 with atomics in an
 * unspecified order (its own test tolerance is 1e-5); the oracle accumulates in
 * fp64.
 * ------------------------------------------------------------------------ */
/* level k of column n: bits [bits*k, bits*k + bits) of the column's little-endian bit stream
 * over the qweight rows.  For 4 and 2 bits that is (word >> bits*(k % per_word)) & mask
 * (cuda_kernel_4bit.cu:133-156, cuda_kernel_2bit.cu:139-143); for 3 bits it is the layout
 * QuantLinear.pack builds with its two split levels per 32 (quant.py:230-257), which
 * cuda_kernel_3bit.cu:126-189 unpicks word by word. */
static uint32_t orc_stream_level(const int32_t* qweight, int64_t out_features, int64_t n, int bits, int64_t k) {
  const int64_t bit = (int64_t)bits * k, idx = bit >> 5;
  const int sh = (int)(bit & 31);
  uint32_t v = (uint32_t)qweight[idx * out_features + n] >> sh;
  if (sh + bits > 32) v |= (uint32_t)qweight[(idx + 1) * out_features + n] << (32 - sh);
  return v & ((1u << bits) - 1u);
}

ORC_API void orc_vecquantmatmul(int bits, const float* x, const int32_t* qweight, float* out, const float* scales,
                                const float* zeros, int64_t batch, int64_t in_features,
                                int64_t out_features, int64_t group_size) {
  if (group_size == 0) group_size = in_features;
  const int64_t groups = (in_features + group_size - 1) / group_size;
  for (int64_t b = 0; b < batch; ++b) {
    for (int64_t n = 0; n < out_features; ++n) {
      double acc = 0.0;
      for (int64_t k = 0; k < in_features; ++k) {
        const float lvl = (float)orc_stream_level(qweight, out_features, n, bits, k);
        const int64_t g = k / group_size;
        const float w = scales[n * groups + g] * lvl - zeros[n * groups + g];
        acc += (double)w * (double)x[b * in_features + k];
      }
      out[b * out_features + n] = (float)((double)out[b * out_features + n] + acc);
    }
  }
}

ORC_API void orc_vecquant4matmul(const float* x, const int32_t* qweight, float* out, const float* scales,
                                 const float* zeros, int64_t batch, int64_t in_features,
                                 int64_t out_features, int64_t group_size) {
  orc_vecquantmatmul(4, x, qweight, out, scales, zeros, batch, in_features, out_features, group_size);
}
