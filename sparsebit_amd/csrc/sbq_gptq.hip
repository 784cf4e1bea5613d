// sbq_gptq.hip -- GPTQ 4-bit (grouped) weight mat-vec for gfx950.
//
// Replaces VecQuant4MatMulKernel / vecquant4matmul_cuda
// (large_language_models/llama/quantization/cuda/cuda_kernel_4bit.cu:36-180):
//   out[b,n] += sum_k (scales[n,g(k)] * nib(k,n) - zeros[n,g(k)]) * x[b,k]
// with qweight int32 [ceil(in/8), out] holding 8 consecutive input-channel
// nibbles per word, low nibble first (QuantLinear.pack, utils/quant.py:187-260),
// scales / zeros fp32 [out, groups] and `out` pre-filled with the bias.
//
// The op is a weight stream (8.4 MB for 4096x4096, ~3.5 flop/byte): HBM/latency
// bound, no MFMA.  Design:
//   * a lane owns 4 adjacent output columns and reads them as one 16-byte word
//     per qweight row, so a wave streams 1 KiB of a row per load; all rows of a
//     workgroup's K slice are requested before any is consumed;
//   * K is split across workgroups in slices of 128 input channels (= one
//     quantization group, so scale/zero are loaded once per lane per slice) and
//     across the 4 waves of a workgroup; the activation slice sits in LDS and is
//     read with broadcast ds_read_b128;
//   * no float atomics (the reference's per-block atomicAdd makes the sum order,
//     hence the result, run-to-run dependent): every workgroup writes its
//     partial tile, a second kernel adds the K slices in a fixed order.
#include "sbq_common.hpp"

namespace sbq {
namespace {

constexpr int kSliceRows = 16;            // qweight rows per K slice
constexpr int kSliceK = kSliceRows * 8;   // 128 input channels

struct GptqGeom {
  int64_t in_features, out_features, batch;
  int32_t H;           // qweight rows
  int32_t groups;
  int32_t group_size;  // in input channels
  int32_t slices;      // K slices
  int32_t slices_per_block;
  int32_t kblocks;     // ceil(slices / slices_per_block)
};

// COLS = 4: 16-byte loads (out_features % 4 == 0, aligned); COLS = 1: any shape.
// kBT: batch rows per register tile -- a mat-VEC (batch 1) must not pay 8 FMAs per weight.
template <int COLS, int kBT>
__global__ __launch_bounds__(kBlock) void gptq4_partial_kernel(
    const float* __restrict__ x, const int32_t* __restrict__ qw, const float* __restrict__ scales,
    const float* __restrict__ zeros, float* __restrict__ part, const GptqGeom g) {
  __shared__ __attribute__((aligned(16))) float xs[kBT][kSliceK];
  __shared__ float red[kWavesPerBlock][kBT][kWave * COLS];
  const int lane = threadIdx.x & (kWave - 1);
  const int wid = threadIdx.x / kWave;
  const int64_t col0 = (static_cast<int64_t>(blockIdx.x) * kWave + lane) * COLS;
  const bool col_ok = col0 < g.out_features;  // COLS == 4 implies out % 4 == 0
  const int kb = blockIdx.y;

  for (int64_t b0 = 0; b0 < g.batch; b0 += kBT) {
    float acc[COLS][kBT];
#pragma unroll
    for (int j = 0; j < COLS; ++j)
#pragma unroll
      for (int b = 0; b < kBT; ++b) acc[j][b] = 0.0f;

    for (int sl = kb * g.slices_per_block; sl < (kb + 1) * g.slices_per_block && sl < g.slices; ++sl) {
      const int row0 = sl * kSliceRows;
      const int64_t k0 = static_cast<int64_t>(row0) * 8;
      // this wave's rows of the slice: row0 + wid, +4, +8, +12 -- request them all first
      uint32_t w[kSliceRows / kWavesPerBlock][COLS];
#pragma unroll
      for (int i = 0; i < kSliceRows / kWavesPerBlock; ++i) {
        const int r = row0 + wid + i * kWavesPerBlock;
        const bool ok = col_ok && r < g.H;
        if constexpr (COLS == 4) {
          u32x4 t = {0, 0, 0, 0};
          if (ok) t = ld16<true>(qw + static_cast<int64_t>(r) * g.out_features + col0);
#pragma unroll
          for (int j = 0; j < 4; ++j) w[i][j] = t[j];
        } else {
          w[i][0] = ok ? static_cast<uint32_t>(__builtin_nontemporal_load(qw + static_cast<int64_t>(r) * g.out_features + col0)) : 0u;
        }
      }
      // quantization group of this slice (group_size % 128 == 0, so it is unique)
      const int grp = static_cast<int>(k0 / g.group_size);
      float sc[COLS], zr[COLS];
#pragma unroll
      for (int j = 0; j < COLS; ++j) {
        const bool ok = col0 + j < g.out_features;
        sc[j] = ok ? scales[(col0 + j) * g.groups + grp] : 0.0f;
        zr[j] = ok ? zeros[(col0 + j) * g.groups + grp] : 0.0f;
      }
      // activations of the slice -> LDS (zero beyond in_features / batch)
      __syncthreads();
      for (int i = threadIdx.x; i < kBT * kSliceK; i += kBlock) {
        const int b = i / kSliceK, kk = i - b * kSliceK;
        const int64_t k = k0 + kk;
        xs[b][kk] = (b0 + b < g.batch && k < g.in_features) ? x[(b0 + b) * g.in_features + k] : 0.0f;
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < kSliceRows / kWavesPerBlock; ++i) {
        const int kk0 = (wid + i * kWavesPerBlock) * 8;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          f32x4 xv[kBT];
#pragma unroll
          for (int b = 0; b < kBT; ++b) xv[b] = *reinterpret_cast<const f32x4*>(&xs[b][kk0 + 4 * h]);
#pragma unroll
          for (int j = 0; j < COLS; ++j) {
#pragma unroll
            for (int n = 0; n < 4; ++n) {
              const float nib = static_cast<float>((w[i][j] >> (4 * (4 * h + n))) & 0xfu);
              const float wt = __builtin_fmaf(sc[j], nib, -zr[j]);
#pragma unroll
              for (int b = 0; b < kBT; ++b) acc[j][b] = __builtin_fmaf(wt, xv[b][n], acc[j][b]);
            }
          }
        }
      }
    }
    // fold the 4 waves (fixed order) and store this K block's partial tile
    __syncthreads();
#pragma unroll
    for (int b = 0; b < kBT; ++b)
#pragma unroll
      for (int j = 0; j < COLS; ++j) red[wid][b][lane * COLS + j] = acc[j][b];
    __syncthreads();
    for (int i = threadIdx.x; i < kBT * kWave * COLS; i += kBlock) {
      const int b = i / (kWave * COLS), cc = i - b * (kWave * COLS);
      const int64_t col = static_cast<int64_t>(blockIdx.x) * kWave * COLS + cc;
      if (b0 + b < g.batch && col < g.out_features) {
        const float t = ((red[0][b][cc] + red[1][b][cc]) + red[2][b][cc]) + red[3][b][cc];
        part[(static_cast<int64_t>(kb) * g.batch + (b0 + b)) * g.out_features + col] = t;
      }
    }
  }
}

// ---- single-launch "strip" kernel (mat-VEC: batch 1 or 2) --------------------------------
// For decode-sized batches no cross-workgroup reduction is needed at all: a workgroup owns a
// 32-column strip of the output for ALL of K.  Its 256 lanes form an 8 (column quads) x 32
// (K lanes) grid; a K lane takes 16 consecutive qweight rows = 128 input channels = exactly
// one quantization group, so scale / zero are loaded once per lane, all 16 of its 16-byte
// weight words are requested back to back (256 B per lane in flight; a wave's load covers
// full 128-byte lines of 8 rows), and the dequantization factors out of the inner loop:
//     sum_k (s*nib_k - z) * x_k  =  s * sum_k nib_k*x_k  -  z * sum_k x_k
// (one convert + one fma per weight instead of two fmas; sum_k x_k is shared by the lane's 4
// columns).  The activations of a pass (4096 floats per batch row) are staged in LDS with a
// 4-float skew per K lane so that the broadcast ds_read_b128 of the 8 K lanes of a wave hit
// different banks.  The 32 K lanes are folded through LDS in a fixed order and the strip is
// written once: one kernel, deterministic, no atomics, no partial-tile traffic.
constexpr int kStripCols = 32;
constexpr int kStripKLanes = kBlock / (kStripCols / 4);       // 32
constexpr int kStripRowsPerPass = kStripKLanes * kSliceRows;  // 512 qweight rows = 4096 channels
constexpr int kStripXStride = kSliceK + 4;                    // skewed LDS row of one K lane

template <int kBT>
__global__ __launch_bounds__(kBlock) void gptq4_strip_kernel(
    const float* __restrict__ x, const int32_t* __restrict__ qw, const float* __restrict__ scales,
    const float* __restrict__ zeros, float* __restrict__ out, const GptqGeom g) {
  __shared__ __attribute__((aligned(16))) float xs[kBT][kStripKLanes * kStripXStride];
  __shared__ float red[kStripKLanes][kBT][kStripCols + 1];
  const int cl = threadIdx.x & 7;   // column quad inside the strip
  const int kl = threadIdx.x >> 3;  // K lane
  const int64_t col0 = static_cast<int64_t>(blockIdx.x) * kStripCols + cl * 4;
  // 16-byte loads of x need aligned rows
  const bool x_vec = (reinterpret_cast<uintptr_t>(x) & 15u) == 0 && (g.in_features & 3) == 0;

  float acc[4][kBT];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int b = 0; b < kBT; ++b) acc[j][b] = 0.0f;

  for (int pass0 = 0; pass0 < g.H; pass0 += kStripRowsPerPass) {
    const int row0 = pass0 + kl * kSliceRows;
    const bool live = row0 < g.H;
    // every global load of the pass is issued before anything waits: this lane's share of the
    // activations (4 x 16 bytes per batch row, coalesced), its 16 weight words, its scale / zero
    const int64_t kbase = static_cast<int64_t>(pass0) * 8;
    f32x4 xg[kBT][4];
#pragma unroll
    for (int b = 0; b < kBT; ++b)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = (j * kBlock + threadIdx.x) * 4;  // element of the pass, multiple of 4
        const int64_t k = kbase + e;
        xg[b][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (b < g.batch) {
          const float* xr = x + b * g.in_features + k;
          if (x_vec && k + 4 <= g.in_features) {
            xg[b][j] = *reinterpret_cast<const f32x4*>(xr);
          } else {
#pragma unroll
            for (int n = 0; n < 4; ++n)
              if (k + n < g.in_features) xg[b][j][n] = xr[n];
          }
        }
      }
    u32x4 w[kSliceRows];
#pragma unroll
    for (int i = 0; i < kSliceRows; ++i) {
      const int r = row0 + i;
      w[i] = u32x4{0, 0, 0, 0};
      if (live && r < g.H) w[i] = ld16<true>(qw + static_cast<int64_t>(r) * g.out_features + col0);
    }
    const int64_t k0 = static_cast<int64_t>(row0) * 8;
    float sc[4] = {0, 0, 0, 0}, zr[4] = {0, 0, 0, 0};
    if (live) {
      const int grp = static_cast<int>(k0 / g.group_size);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        sc[j] = scales[(col0 + j) * g.groups + grp];
        zr[j] = zeros[(col0 + j) * g.groups + grp];
      }
    }
    __syncthreads();  // previous pass done with xs
#pragma unroll
    for (int b = 0; b < kBT; ++b)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = (j * kBlock + threadIdx.x) * 4;
        const int lane_k = e / kSliceK, off = e - lane_k * kSliceK;
        *reinterpret_cast<f32x4*>(&xs[b][lane_k * kStripXStride + off]) = xg[b][j];
      }
    __syncthreads();
    if (live) {
      float dot[4][kBT], xsum[kBT];
#pragma unroll
      for (int b = 0; b < kBT; ++b) {
        xsum[b] = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) dot[j][b] = 0.0f;
      }
#pragma unroll
      for (int i = 0; i < kSliceRows; ++i) {
        f32x4 xv[kBT][2];
#pragma unroll
        for (int b = 0; b < kBT; ++b) {
          const float* xr = &xs[b][kl * kStripXStride + i * 8];
          xv[b][0] = *reinterpret_cast<const f32x4*>(xr);
          xv[b][1] = *reinterpret_cast<const f32x4*>(xr + 4);
#pragma unroll
          for (int n = 0; n < 8; ++n) xsum[b] += xv[b][n >> 2][n & 3];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t word = w[i][j];
          const uint32_t even = word & 0x0f0f0f0fu;         // nibbles 0,2,4,6 as bytes
          const uint32_t odd = (word >> 4) & 0x0f0f0f0fu;   // nibbles 1,3,5,7 as bytes
#pragma unroll
          for (int n = 0; n < 8; ++n) {
            const uint32_t src = (n & 1) ? odd : even;
            const float nib = static_cast<float>((src >> (8 * (n >> 1))) & 0xffu);  // v_cvt_f32_ubyteN
#pragma unroll
            for (int b = 0; b < kBT; ++b) dot[j][b] = __builtin_fmaf(nib, xv[b][n >> 2][n & 3], dot[j][b]);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int b = 0; b < kBT; ++b) acc[j][b] += __builtin_fmaf(sc[j], dot[j][b], -(zr[j] * xsum[b]));
    }
  }
  // fold the 32 K lanes in ascending order, add to out (pre-filled with the bias)
#pragma unroll
  for (int b = 0; b < kBT; ++b)
#pragma unroll
    for (int j = 0; j < 4; ++j) red[kl][b][cl * 4 + j] = acc[j][b];
  __syncthreads();
  for (int i = threadIdx.x; i < kBT * kStripCols; i += kBlock) {
    const int b = i / kStripCols, cc = i - b * kStripCols;
    if (b < g.batch) {
      float t = 0.0f;
#pragma unroll
      for (int q = 0; q < kStripKLanes; ++q) t += red[q][b][cc];
      out[b * g.out_features + static_cast<int64_t>(blockIdx.x) * kStripCols + cc] += t;
    }
  }
}

// out[b,n] += sum over K blocks, ascending
__global__ __launch_bounds__(kBlock) void gptq4_fold_kernel(const float* __restrict__ part,
                                                            float* __restrict__ out, int64_t bn,
                                                            int kblocks) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= bn) return;
  float t = 0.0f;
  for (int kb = 0; kb < kblocks; ++kb) t += part[static_cast<int64_t>(kb) * bn + i];
  out[i] += t;
}

bool gptq_geom(int64_t batch, int64_t in_f, int64_t out_f, int64_t group_size, GptqGeom& g) {
  if (batch <= 0 || in_f <= 0 || out_f <= 0) return false;
  if (in_f >= (1ll << 31) || out_f >= (1ll << 31)) return false;
  if (group_size == 0) group_size = in_f;
  if (group_size < 0 || group_size >= (1ll << 31)) return false;
  g.in_features = in_f;
  g.out_features = out_f;
  g.batch = batch;
  g.H = static_cast<int32_t>(ceil_div(in_f, 8));
  g.group_size = static_cast<int32_t>(group_size);
  g.groups = static_cast<int32_t>(ceil_div(in_f, group_size));
  g.slices = static_cast<int32_t>(ceil_div(g.H, kSliceRows));
  // enough workgroups to fill 256 CUs a few times over, but not more K blocks than
  // that needs: each K block costs batch*out floats of partial traffic
  const int64_t colblocks = ceil_div(out_f, kWave * 4);
  int64_t want = ceil_div(1024, colblocks);
  if (want < 1) want = 1;
  if (want > g.slices) want = g.slices;
  g.slices_per_block = static_cast<int32_t>(ceil_div(g.slices, want));
  g.kblocks = static_cast<int32_t>(ceil_div(g.slices, g.slices_per_block));
  return true;
}

}  // namespace
}  // namespace sbq

extern "C" {

size_t sbq_gptq_workspace_bytes(int64_t batch, int64_t in_features, int64_t out_features) {
  using namespace sbq;
  GptqGeom g;
  if (!gptq_geom(batch, in_features, out_features, 0, g)) return 0;
  return static_cast<size_t>(g.kblocks) * batch * out_features * sizeof(float);
}

int sbq_vecquant4matmul(const float* x, const int32_t* qweight, float* out, const float* scales,
                        const float* zeros, int64_t batch, int64_t in_features, int64_t out_features,
                        int64_t group_size, void* workspace, size_t workspace_bytes, void* stream) {
  using namespace sbq;
  if (batch < 0 || in_features < 0 || out_features < 0) return SBQ_ERR_ARG;
  if (batch == 0 || in_features == 0 || out_features == 0) return SBQ_ERR_EMPTY;
  if (!x || !qweight || !out || !scales || !zeros || !workspace) return SBQ_ERR_NULL;
  // cuda_kernel_4bit.cu:58-61: group size must be a multiple of 128 (0 = one group)
  if (group_size != 0 && group_size % 128 != 0) return SBQ_ERR_ARG;
  GptqGeom g;
  if (!gptq_geom(batch, in_features, out_features, group_size, g)) return SBQ_ERR_ARG;
  const size_t need = static_cast<size_t>(g.kblocks) * batch * out_features * sizeof(float);
  if (workspace_bytes < need || !aligned16(workspace)) return SBQ_ERR_WORKSPACE;
  if ((reinterpret_cast<uintptr_t>(x) & 3u) || (reinterpret_cast<uintptr_t>(qweight) & 3u) ||
      (reinterpret_cast<uintptr_t>(out) & 3u))
    return SBQ_ERR_ALIGN;
  hipStream_t st = as_stream(stream);
  float* part = static_cast<float*>(workspace);
  const bool vec = (out_features % 4 == 0) && aligned16(qweight);
  const int bt = batch >= 8 ? 8 : (batch >= 3 ? 4 : (batch == 2 ? 2 : 1));
  // single-launch strip kernel: mat-vec sized batches, whole 32-column strips
  if (vec && out_features % kStripCols == 0 && batch <= 2 && knob(2) != 9) {
    const uint32_t grid = static_cast<uint32_t>(out_features / kStripCols);
    if (batch == 2) gptq4_strip_kernel<2><<<grid, kBlock, 0, st>>>(x, qweight, scales, zeros, out, g);
    else gptq4_strip_kernel<1><<<grid, kBlock, 0, st>>>(x, qweight, scales, zeros, out, g);
    return check_launch();
  }
#define SBQ_GPTQ(COLS)                                                                              \
  do {                                                                                              \
    dim3 grid(static_cast<uint32_t>(ceil_div(out_features, kWave * COLS)), static_cast<uint32_t>(g.kblocks)); \
    if (bt == 8) gptq4_partial_kernel<COLS, 8><<<grid, kBlock, 0, st>>>(x, qweight, scales, zeros, part, g); \
    else if (bt == 4) gptq4_partial_kernel<COLS, 4><<<grid, kBlock, 0, st>>>(x, qweight, scales, zeros, part, g); \
    else if (bt == 2) gptq4_partial_kernel<COLS, 2><<<grid, kBlock, 0, st>>>(x, qweight, scales, zeros, part, g); \
    else gptq4_partial_kernel<COLS, 1><<<grid, kBlock, 0, st>>>(x, qweight, scales, zeros, part, g);  \
  } while (0)
  if (vec) SBQ_GPTQ(4);
  else SBQ_GPTQ(1);
#undef SBQ_GPTQ
  int rc = check_launch();
  if (rc != SBQ_OK) return rc;
  const int64_t bn = batch * out_features;
  gptq4_fold_kernel<<<static_cast<uint32_t>(ceil_div(bn, kBlock)), kBlock, 0, st>>>(part, out, bn, g.kblocks);
  return check_launch();
}

}  // extern "C"
