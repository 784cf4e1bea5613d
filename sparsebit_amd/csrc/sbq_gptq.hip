// sbq_gptq.hip -- GPTQ 4- / 3- / 2-bit (grouped) weight mat-vec for gfx950.
//
// Replaces VecQuant{4,3,2}MatMulKernel / vecquant{4,3,2}matmul_cuda
// (large_language_models/llama/quantization/cuda/cuda_kernel_4bit.cu:36-180,
//  cuda_kernel_3bit.cu:29-196, cuda_kernel_2bit.cu:29-150):
//   out[b,n] += sum_k (scales[n,g(k)] * lvl(k,n) - zeros[n,g(k)]) * x[b,k]
// with qweight int32 [rows, out], scales / zeros fp32 [out, groups] and `out`
// pre-filled with the bias.  QuantLinear.pack (utils/quant.py:187-260) lays a column's
// levels out as ONE little-endian bit stream over its rows, BITS bits per input channel:
// 8 nibbles per word (4-bit), 16 crumbs per word (2-bit), and for 3-bit 32 levels per three
// words -- its "<< 30 / >> 2 & 1" and "<< 31 / >> 1 & 3" split cases are exactly the two
// levels of a 96-bit stream that straddle a word boundary.  lvl(k) = bits [BITS*k, BITS*k+BITS).
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
#include <climits>
#include <type_traits>

#include "sbq_common.hpp"

namespace sbq {
namespace {

constexpr int kSliceK = 128;  // input channels per K slice (64 only for 2-bit with 64-channel groups)

// level k of a column whose stream words are w[0..]: all indices are compile-time constants
// at every call site (fully unrolled), so this is a v_bfe_u32 (plus one v_alignbit for the two
// straddling 3-bit levels of every 32)
template <int BITS, typename W>
__device__ __forceinline__ uint32_t stream_level(const W& w, int k) {
  const int bit = BITS * k, idx = bit >> 5, sh = bit & 31;
  uint32_t v = w(idx) >> sh;
  if (sh + BITS > 32) v |= w(idx + 1) << (32 - sh);
  return v & ((1u << BITS) - 1u);
}

struct GptqGeom {
  int64_t in_features, out_features, batch;
  int32_t H;           // qweight rows
  int32_t groups;
  int32_t group_size;  // in input channels
  int32_t slices;      // K slices
  int32_t slices_per_block;
  int32_t kblocks;     // ceil(slices / slices_per_block)
  int32_t xcd_swizzle; // strip kernels: XCD-aware strip order (strips % 8 == 0)
  int32_t grid_x, grid_y;  // LEAN strip kernels: the launch grid (read with the other arguments instead of from
                           // the hidden ones at the far end of the argument block)
};

// COLS = 4: 16-byte loads (out_features % 4 == 0, aligned); COLS = 1: any shape.
// kBT: batch rows per register tile -- a mat-VEC (batch 1) must not pay 8 FMAs per weight.
// A slice of SK channels is SK*BITS/32 stream words; a wave owns whole "units" of the slice
// (a unit = the fewest words holding whole levels: 1 word, or 3 for 3-bit).
template <int BITS, int SK, int COLS, int kBT>
__global__ __launch_bounds__(kBlock) void gptq_partial_kernel(
    const float* __restrict__ x, const int32_t* __restrict__ qw, const float* __restrict__ scales,
    const float* __restrict__ zeros, float* __restrict__ out, float* __restrict__ part, const GptqGeom g) {
  constexpr int kUnitRows = BITS == 3 ? 3 : 1;
  constexpr int kUnitCh = 32 * kUnitRows / BITS;   // 8 / 32 / 16 channels
  constexpr int kUnits = SK / kUnitCh;             // per slice
  constexpr int kUPW = kUnits / kWavesPerBlock;    // units per wave
  constexpr int kSliceRows = SK * BITS / 32;
  static_assert(kUnits % kWavesPerBlock == 0 && kUPW >= 1, "slice must split evenly over the waves");
  __shared__ __attribute__((aligned(16))) float xs[kBT][SK];
  // the four waves' accumulators are folded through LDS eight batch rows at a time (kBT = 32 would need 128 KB)
  constexpr int kRedB = kBT < 8 ? kBT : 8;
  __shared__ float red[kWavesPerBlock][kRedB][kWave * COLS];
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
      const int64_t k0 = static_cast<int64_t>(sl) * SK;
      // this wave's units of the slice: wid, wid + 4, ... -- request all their words first
      uint32_t w[kUPW][kUnitRows][COLS];
#pragma unroll
      for (int i = 0; i < kUPW; ++i)
#pragma unroll
        for (int rr = 0; rr < kUnitRows; ++rr) {
          const int r = row0 + (wid + i * kWavesPerBlock) * kUnitRows + rr;
          const bool ok = col_ok && r < g.H;
          if constexpr (COLS == 4) {
            u32x4 t = {0, 0, 0, 0};
            if (ok) t = ld16<true>(qw + static_cast<int64_t>(r) * g.out_features + col0);
#pragma unroll
            for (int j = 0; j < 4; ++j) w[i][rr][j] = t[j];
          } else {
            w[i][rr][0] = ok ? static_cast<uint32_t>(__builtin_nontemporal_load(qw + static_cast<int64_t>(r) * g.out_features + col0)) : 0u;
          }
        }
      // quantization group of this slice (group_size % SK == 0, so it is unique)
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
      for (int i = threadIdx.x; i < kBT * SK; i += kBlock) {
        const int b = i / SK, kk = i - b * SK;
        const int64_t k = k0 + kk;
        xs[b][kk] = (b0 + b < g.batch && k < g.in_features) ? x[(b0 + b) * g.in_features + k] : 0.0f;
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < kUPW; ++i) {
        const int kk0 = (wid + i * kWavesPerBlock) * kUnitCh;
#pragma unroll
        for (int h = 0; h < kUnitCh / 4; ++h) {
          // dequantize the 4 channels x COLS columns once, then stream the batch rows over them: the weights stay
          // in registers for every batch row of the tile (one read of the matrix for up to 32 rows)
          float wt[COLS][4];
#pragma unroll
          for (int j = 0; j < COLS; ++j)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
              const float lvl = static_cast<float>(
                  stream_level<BITS>([&](int idx) { return w[i][idx][j]; }, 4 * h + n));
              wt[j][n] = __builtin_fmaf(sc[j], lvl, -zr[j]);
            }
#pragma unroll
          for (int b = 0; b < kBT; ++b) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(&xs[b][kk0 + 4 * h]);
#pragma unroll
            for (int j = 0; j < COLS; ++j)
#pragma unroll
              for (int n = 0; n < 4; ++n) acc[j][b] = __builtin_fmaf(wt[j][n], xv[n], acc[j][b]);
          }
        }
      }
    }
    // fold the 4 waves (fixed order) and store this K block's partial tile -- or, when this workgroup saw all
    // of K, add it to `out` (pre-filled with the bias) directly
#pragma unroll
    for (int bb = 0; bb < kBT; bb += kRedB) {
      __syncthreads();
#pragma unroll
      for (int b = 0; b < kRedB; ++b)
#pragma unroll
        for (int j = 0; j < COLS; ++j) red[wid][b][lane * COLS + j] = acc[j][bb + b];
      __syncthreads();
      for (int i = threadIdx.x; i < kRedB * kWave * COLS; i += kBlock) {
        const int b = i / (kWave * COLS), cc = i - b * (kWave * COLS);
        const int64_t col = static_cast<int64_t>(blockIdx.x) * kWave * COLS + cc;
        const int64_t row = b0 + bb + b;
        if (row < g.batch && col < g.out_features) {
          const float t = ((red[0][b][cc] + red[1][b][cc]) + red[2][b][cc]) + red[3][b][cc];
          if (g.kblocks == 1) out[row * g.out_features + col] += t;
          else part[(static_cast<int64_t>(kb) * g.batch + row) * g.out_features + col] = t;
        }
      }
    }
  }
}

// ---- single-launch "strip" kernel (mat-VEC: batch 1 or 2) --------------------------------
// A workgroup owns a 32-column strip of the output for a block of K.  Its lanes form an
// 8 (column quads) x KL (K lanes) grid; a K lane takes CH = 128 or 64 consecutive input channels
// (CH*BITS/32 qweight rows) inside one quantization group, so scale / zero are loaded once per lane, all
// of its 16-byte weight words are requested back to back (up to 256 B per lane in flight; a
// wave's load covers full 128-byte lines of 8 rows), and the dequantization factors out of the
// inner loop:
//     sum_k (s*lvl_k - z) * x_k  =  s * sum_k lvl_k*x_k  -  z * sum_k x_k
// (one convert + one fma per weight instead of two fmas; sum_k x_k is shared by the lane's 4
// columns).  The activations of a pass (KL*128 floats per batch row) are staged in LDS with a
// 4-float skew per K lane so that the broadcast ds_read_b128 of the 8 K lanes of a wave hit
// different banks.
// A mat-vec is a latency problem before it is a bandwidth problem (9.4 MB for 4096x4096 is
// 1.2 us of HBM time): what matters is that the WHOLE weight matrix is in flight at once on
// all 256 CUs.  So CH and the K split S = gridDim.y are chosen per shape to put a few workgroups
// on every CU, each of which issues every one of its loads before it waits for any.
// Cross-workgroup fold without atomics on floats and without a second launch: every workgroup
// writes its 32-column partial, then bumps the strip's arrival counter; whichever workgroup
// arrives last adds the S partials IN INDEX ORDER (so the sum does not depend on who was last)
// and resets the counter.  One kernel, deterministic, nobody ever waits on another workgroup.
constexpr int kStripCols = 32;

// -DSBQ_GPTQ_STAMPS=1 (tools/lab/gptq_stamps.py builds such a library next to the product one): thread 0 of every
// workgroup of gptq_strip_kernel writes s_memrealtime (100 MHz) stamps through a device-global pointer
#ifndef SBQ_GPTQ_STAMPS
#define SBQ_GPTQ_STAMPS 0
#endif
#if SBQ_GPTQ_STAMPS != 0
__constant__ unsigned long long* g_gptq_stamps = nullptr;
#define GPTQ_STAMP_INIT() \
  unsigned long long* const stamp_base = g_gptq_stamps ? g_gptq_stamps + (blockIdx.y * gridDim.x + blockIdx.x) * 8 : nullptr
#define GPTQ_STAMP(i)                                                                     \
  do {                                                                                    \
    if (threadIdx.x == 0 && stamp_base) stamp_base[(i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define GPTQ_STAMP_INIT() \
  do {                    \
  } while (0)
#define GPTQ_STAMP(i) \
  do {                \
  } while (0)
#endif
constexpr int kStripMaxSplit = 16;          // upper bound of the K split S
constexpr size_t kCounterBytes = SBQ_GPTQ_COUNTER_BYTES;  // fixed region at the head of the workspace
constexpr int64_t kMaxStrips = static_cast<int64_t>(kCounterBytes / sizeof(uint32_t));

// level k (0..127) of column j of a K lane's words as a float.  4- and 2-bit levels never
// straddle a word: mask them into byte lanes first so that the convert is v_cvt_f32_ubyteN
// (the masks are shared by all levels of a word after unrolling).
template <int BITS, int ROWS>
__device__ __forceinline__ float strip_level(const u32x4 (&w)[ROWS], int j, int k) {
  if constexpr (BITS == 4) {
    const uint32_t word = w[k >> 3][j];
    const uint32_t src = (k & 1) ? (word >> 4) & 0x0f0f0f0fu : word & 0x0f0f0f0fu;
    return static_cast<float>((src >> (8 * ((k & 7) >> 1))) & 0xffu);
  } else if constexpr (BITS == 2) {
    const uint32_t word = w[k >> 4][j];
    const uint32_t src = (word >> (2 * (k & 3))) & 0x03030303u;  // crumbs s, s+4, s+8, s+12
    return static_cast<float>((src >> (8 * ((k & 15) >> 2))) & 0xffu);
  } else {
    return static_cast<float>(stream_level<BITS>([&](int idx) { return w[idx][j]; }, k));
  }
}

// DEC8 (4- and 2-bit): a level sitting alone in a byte IS the OCP e4m3 encoding of n * 2^-9
// (subnormals m * 2^-9 for n < 8, then (8 + m) * 2^-9), so v_cvt_pk_f32_fp8 turns two levels into two
// exact floats per instruction and its result pair feeds v_pk_fma_f32 directly; the 2^9 is folded back
// into the scale (a power of two: the products are the same reals).  The activations are staged in LDS
// in the matching pair order (x0, x2, x1, x3).
// One K lane's share of a pass: its CH channels x 4 columns against kBT activation rows (staged in LDS, row stride
// XS floats), dequantization factored out of the inner loop (see the kernel's header comment).
// XSE: the sums of the activations over the lane's channels come from the caller (`xsum_in`) instead of being
// accumulated here (the in-loop adds are then dead code).
template <int BITS, int kBT, int CH, bool DEC8, int XS, bool XSE = false>
__device__ __forceinline__ void strip_compute(const u32x4 (&w)[CH * BITS / 32], const float* __restrict__ xs, int kl,
                                              const float (&sc)[4], const float (&zr)[4], float (&acc)[4][kBT],
                                              const float* xsum_in = nullptr) {
  constexpr int kRows = CH * BITS / 32;
  constexpr int kXStride = CH + 4;
  float dot[4][kBT], xsum[kBT];
#pragma unroll
  for (int b = 0; b < kBT; ++b) {
    xsum[b] = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) dot[j][b] = 0.0f;
  }
  if constexpr (DEC8 && BITS == 2) {
    f32x2 dot2[4][kBT];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int b = 0; b < kBT; ++b) dot2[j][b] = f32x2{0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < CH / 16; ++i) {  // one weight word = 16 channels at a time
      f32x2 xa[kBT][4], xb[kBT][4];      // (x_s, x_s+4) and (x_s+8, x_s+12), s = 0..3
#pragma unroll
      for (int b = 0; b < kBT; ++b) {
        const float* xr = &xs[b * XS + kl * kXStride + i * 16];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          const f32x4 q = *reinterpret_cast<const f32x4*>(xr + 4 * h);
          if (h < 2) {
            xa[b][2 * h] = f32x2{q[0], q[1]};
            xa[b][2 * h + 1] = f32x2{q[2], q[3]};
          } else {
            xb[b][2 * (h - 2)] = f32x2{q[0], q[1]};
            xb[b][2 * (h - 2) + 1] = f32x2{q[2], q[3]};
          }
#pragma unroll
          for (int n = 0; n < 4; ++n) xsum[b] += q[n];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t word = w[i][j];
#pragma unroll
        for (int sft = 0; sft < 4; ++sft) {
          const uint32_t m = (word >> (2 * sft)) & 0x03030303u;  // crumbs s, s+4, s+8, s+12 alone in bytes
          const f32x2 la = __builtin_amdgcn_cvt_pk_f32_fp8(m, false);
          const f32x2 lb = __builtin_amdgcn_cvt_pk_f32_fp8(m, true);
#pragma unroll
          for (int b = 0; b < kBT; ++b) {
            dot2[j][b] = __builtin_elementwise_fma(la, xa[b][sft], dot2[j][b]);
            dot2[j][b] = __builtin_elementwise_fma(lb, xb[b][sft], dot2[j][b]);
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int b = 0; b < kBT; ++b) dot[j][b] = (dot2[j][b][0] + dot2[j][b][1]) * 512.0f;  // exact: 2^9
  } else if constexpr (DEC8) {
    static_assert(!DEC8 || BITS == 4 || BITS == 2, "the e4m3 decode needs a level alone in a byte");
    f32x2 dot2[4][kBT];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int b = 0; b < kBT; ++b) dot2[j][b] = f32x2{0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < CH / 8; ++i) {  // one weight word = 8 channels at a time
      f32x2 xp[kBT][4];                 // (x0,x2) (x1,x3) (x4,x6) (x5,x7)
#pragma unroll
      for (int b = 0; b < kBT; ++b) {
        const float* xr = &xs[b * XS + kl * kXStride + i * 8];
        const f32x4 lo = *reinterpret_cast<const f32x4*>(xr), hi = *reinterpret_cast<const f32x4*>(xr + 4);
        xp[b][0] = f32x2{lo[0], lo[1]};
        xp[b][1] = f32x2{lo[2], lo[3]};
        xp[b][2] = f32x2{hi[0], hi[1]};
        xp[b][3] = f32x2{hi[2], hi[3]};
#pragma unroll
        for (int n = 0; n < 4; ++n) xsum[b] += lo[n];
#pragma unroll
        for (int n = 0; n < 4; ++n) xsum[b] += hi[n];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t word = w[i][j];
        const uint32_t even = word & 0x0f0f0f0fu, odd = (word >> 4) & 0x0f0f0f0fu;
        const f32x2 l02 = __builtin_amdgcn_cvt_pk_f32_fp8(even, false);
        const f32x2 l13 = __builtin_amdgcn_cvt_pk_f32_fp8(odd, false);
        const f32x2 l46 = __builtin_amdgcn_cvt_pk_f32_fp8(even, true);
        const f32x2 l57 = __builtin_amdgcn_cvt_pk_f32_fp8(odd, true);
#pragma unroll
        for (int b = 0; b < kBT; ++b) {
          dot2[j][b] = __builtin_elementwise_fma(l02, xp[b][0], dot2[j][b]);
          dot2[j][b] = __builtin_elementwise_fma(l13, xp[b][1], dot2[j][b]);
          dot2[j][b] = __builtin_elementwise_fma(l46, xp[b][2], dot2[j][b]);
          dot2[j][b] = __builtin_elementwise_fma(l57, xp[b][3], dot2[j][b]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int b = 0; b < kBT; ++b) dot[j][b] = (dot2[j][b][0] + dot2[j][b][1]) * 512.0f;  // exact: 2^9
  } else {
#pragma unroll
    for (int i = 0; i < CH / 8; ++i) {  // 8 channels at a time
      f32x4 xv[kBT][2];
  #pragma unroll
      for (int b = 0; b < kBT; ++b) {
        const float* xr = &xs[b * XS + kl * kXStride + i * 8];
        xv[b][0] = *reinterpret_cast<const f32x4*>(xr);
        xv[b][1] = *reinterpret_cast<const f32x4*>(xr + 4);
  #pragma unroll
        for (int n = 0; n < 8; ++n) xsum[b] += xv[b][n >> 2][n & 3];
      }
  #pragma unroll
      for (int j = 0; j < 4; ++j) {
  #pragma unroll
        for (int n = 0; n < 8; ++n) {
          const float lvl = strip_level<BITS, kRows>(w, j, i * 8 + n);
  #pragma unroll
          for (int b = 0; b < kBT; ++b) dot[j][b] = __builtin_fmaf(lvl, xv[b][n >> 2][n & 3], dot[j][b]);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int b = 0; b < kBT; ++b) acc[j][b] += __builtin_fmaf(sc[j], dot[j][b], -(zr[j] * (XSE ? xsum_in[b] : xsum[b])));
}

// LDS-DMA of one 16-byte word per lane: global -> LDS without passing through (or occupying) VGPRs.  The wave's
// 64 words land at m0 + lane * 16.
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// ... the same with a uniform base (SGPR pair) + a 32-bit byte offset per lane: half the address registers and none
// of the 64-bit address arithmetic
__device__ __forceinline__ void glds16s(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// The cross-workgroup half of the K split (shared by the strip kernels): publish, count the arrival, and the last
// workgroup of the strip folds the S partials in index order.
template <int kThreads, int kStripCols = 32>
__device__ __forceinline__ void strip_fold_partials(uint32_t strip, int split, const GptqGeom& g,
                                                    float* __restrict__ part, float* __restrict__ out,
                                                    uint32_t* __restrict__ arrivals, uint32_t& s_prev) {
  // publish the partials, count the arrival; the last one folds all S partials in index order.
  // No agent-scope fence anywhere: on gfx950 such a fence writes back / invalidates the whole
  // per-XCD L2 (measured: +15..35 us per launch with ~1000 workgroups doing it).  Instead every
  // access of the protocol individually goes to the device-coherent level -- agent-scope atomic
  // store / load / add carry sc1 and bypass the non-coherent L2 -- and the only ordering needed
  // is "my partial stores are acknowledged before my arrival is counted": s_waitcnt vmcnt(0)
  // (a workgroup-scope release fence) + the barrier.  The fold's loads depend on the counter
  // value through LDS and the barrier, so they are issued after the add has returned.
  // tests/test_gpu_gptq_stress.py: 10^4 calls on two streams under load, bit-equal, counters back at zero.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);  // belt and braces: all counters drained
  __syncthreads();
  if (threadIdx.x == 0)
    s_prev = __hip_atomic_fetch_add(&arrivals[strip], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (s_prev != static_cast<uint32_t>(split - 1)) return;
  // (row, column) outputs of the strip, four per thread at a time so that four independent loads are in flight
  // per partial index instead of one
  const int64_t n_out = g.batch * kStripCols;
  for (int64_t i0 = threadIdx.x; i0 < n_out; i0 += 4 * kThreads) {
    float total[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    int64_t addr[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = i0 + static_cast<int64_t>(u) * kThreads;
      ok[u] = i < n_out;
      const int64_t row = ok[u] ? i / kStripCols : 0, cc = ok[u] ? i - row * kStripCols : 0;
      addr[u] = row * g.out_features + static_cast<int64_t>(strip) * kStripCols + cc;
    }
    for (int sidx = 0; sidx < split; ++sidx) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        total[u] += __hip_atomic_load(&part[static_cast<int64_t>(sidx) * g.batch * g.out_features + addr[u]],
                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (ok[u]) out[addr[u]] += total[u];
  }
  if (threadIdx.x == 0)  // leave the counter as we found it: the workspace stays reusable
    __hip_atomic_store(&arrivals[strip], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// PF (HBM-sized matrices, several passes per workgroup): the weights of pass p+1 travel global -> LDS by DMA while
// pass p is being computed, and are picked up from LDS at the top of pass p+1.  A wave only ever reads back the
// words its own lanes requested, so the weight ring needs no barrier; in-flight bytes cost LDS, not registers,
// so three workgroups per CU keep ~96 KB of weight reads outstanding ALL the time instead of 4 x 32 KB half of it.
// Several matrices that share the activation vector (the q / k / v projections of a decoder layer; gate + up of its
// MLP: quant.py:262-278 issues one mat-vec per QuantLinear) as ONE launch: the strips of all of them form one grid,
// a workgroup looks up its matrix (uniform scalar loads out of the kernel arguments) and proceeds as for a single
// matrix.  n == 0: the plain pointer arguments.
constexpr int kMaxMulti = 4;
struct GptqMulti {
  int32_t n;
  int32_t strip_begin[kMaxMulti + 1];
  const int32_t* qw[kMaxMulti];
  const float* scales[kMaxMulti];
  const float* zeros[kMaxMulti];
  float* out[kMaxMulti];
  int64_t out_features[kMaxMulti];
  int64_t part_off[kMaxMulti];  // floats: where the matrix's partial tiles start
};

// LEAN (host-checked: x rows 16-byte aligned, in_features % 4 == 0): EVERY global load of a pass is unconditional --
// channels, batch rows, weight rows and groups are clamped to valid ones and what must not count is zeroed by a
// mask afterwards.  A load under a branch makes the compiler wait for everything in flight at the join: the
// timeline of the branchy version (tools/lab/gptq_stamps.py, profiles/r04_gptq_timeline.txt) showed the weight
// words of the first pass being REQUESTED 2.3 us after the workgroup started -- behind two full memory round trips
// (the value added to, then the activations) that the joins had serialised.
// LEAN also means a lean PROLOGUE (the timeline again: 1.4 us from a workgroup's start to its first load): the grid
// shape travels with the geometry, the XCD swizzle is arithmetic instead of a branch, the matrix lookup of a
// multi-matrix launch (MULTI) is three comparisons instead of a loop over the argument block -- so every kernel
// argument is fetched by ONE group of scalar loads instead of five dependent ones -- and every address is a uniform
// base + a 32-bit byte offset (host-checked: each tensor is below 4 GB).
template <int BITS, int kBT, int KL, int CH, bool DEC8 = false, bool PF = false, bool LEAN = false, bool MULTI = false>
__global__ __launch_bounds__(8 * KL) __attribute__((amdgpu_waves_per_eu(CH == 64 && kBT == 1 ? (PF ? 3 : 4) : 1, 8))) void gptq_strip_kernel(
    // the first 14 argument dwords arrive in SGPRs with the wave (kernarg preload, sparsebit_amd/build.py): the four
    // pointers the loads of a pass go through and the geometry in 32 bits -- a LEAN kernel issues its first loads
    // without having waited for any scalar load.  (host side: SBQ_STRIP_LEAD)
    const int32_t* __restrict__ qw_a, const float* __restrict__ x, const float* __restrict__ scales_a,
    const float* __restrict__ zeros_a, int32_t out32_a, int32_t H_a, uint32_t grid_batch_a, int32_t groups_a,
    int32_t group_size_a, int32_t in32_a,  // <- 14 dwords: what fits beside the system SGPRs
    float* __restrict__ out_a, float* __restrict__ part_a, uint32_t* __restrict__ arrivals_a, const GptqGeom g_a,
    const GptqMulti mm) {
  GptqGeom g;
  if constexpr (LEAN) {  // from the preloaded scalars; g_a is never read
    g.in_features = in32_a;
    g.out_features = out32_a;
    g.batch = static_cast<int32_t>(grid_batch_a >> 26);
    g.H = H_a;
    g.groups = groups_a;
    g.group_size = group_size_a;
    g.slices = g.slices_per_block = g.kblocks = 0;
    g.xcd_swizzle = static_cast<int32_t>((grid_batch_a >> 25) & 1u);
    g.grid_x = static_cast<int32_t>(grid_batch_a & 0xfffffu);
    g.grid_y = static_cast<int32_t>((grid_batch_a >> 20) & 31u);
  } else {
    g = g_a;
  }
  const int32_t* __restrict__ qw = qw_a;
  const float* __restrict__ scales = scales_a;
  const float* __restrict__ zeros = zeros_a;
  float* __restrict__ out = out_a;
  float* __restrict__ part = part_a;
  uint32_t* __restrict__ arrivals = arrivals_a;
  GPTQ_STAMP_INIT();
  GPTQ_STAMP(0);
  constexpr int kThreads = 8 * KL;
  constexpr int kRows = CH * BITS / 32;       // qweight rows of one K lane: 16 / 12 / 8 for CH = 128
  constexpr int kRowsPerPass = KL * kRows;    // = KL * CH input channels
  constexpr int kXLoads = (KL * CH) / (kThreads * 4);  // 16-byte x loads per thread per pass
  constexpr int kXStride = CH + 4;            // skewed LDS row of one K lane
  __shared__ __attribute__((aligned(16))) float xs[kBT][KL * kXStride];
  // (a mat-vec folds a wave's 8 K lanes by shuffles and hands one partial per WAVE through LDS; more batch rows: one
  // per K lane -- the shuffles' registers cost the two-row 512-thread kernel its second workgroup per CU)
  constexpr bool kShuffleFold = kBT == 1;
  constexpr int kRedRows = kShuffleFold ? kThreads / kWave : KL;
  __shared__ float red[kRedRows][kBT][kStripCols + 1];
  __shared__ uint32_t s_prev;
  const int cl = threadIdx.x & 7;   // column quad inside the strip
  const int kl = threadIdx.x >> 3;  // K lane
  // XCD-aware strip order: consecutive workgroup ids go round-robin over the 8 XCDs, so with strip = blockIdx.x
  // the neighbouring 128-byte pieces of a qweight row are requested by eight different L2s.  Giving XCD i the
  // i-th contiguous eighth of the strips makes the workgroups that run side by side on one XCD read adjacent
  // pieces of the same rows (knob 2 == 8: plain order, for A/B runs).
  static_assert(!MULTI || LEAN, "several matrices per launch: the LEAN kernels only");
  uint32_t strip = blockIdx.x;
  int split;
  if constexpr (LEAN) {
    const uint32_t sw = (blockIdx.x & 7u) * (static_cast<uint32_t>(g.grid_x) >> 3) + (blockIdx.x >> 3);
    strip = blockIdx.x + (sw - blockIdx.x) * static_cast<uint32_t>(g.xcd_swizzle);  // xcd_swizzle is 0 or 1
    split = g.grid_y;
  } else {
    if (g.xcd_swizzle) strip = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    split = gridDim.y;
  }
  if constexpr (MULTI) {  // strip_begin[m] for m >= n holds INT32_MAX (host): uniform, no loop
    const int m = static_cast<int>(strip >= static_cast<uint32_t>(mm.strip_begin[1])) +
                  static_cast<int>(strip >= static_cast<uint32_t>(mm.strip_begin[2])) +
                  static_cast<int>(strip >= static_cast<uint32_t>(mm.strip_begin[3]));
    qw = mm.qw[m];
    scales = mm.scales[m];
    zeros = mm.zeros[m];
    out = mm.out[m];
    g.out_features = mm.out_features[m];
    part = part_a + mm.part_off[m];
    arrivals = arrivals_a + mm.strip_begin[m];
    strip -= static_cast<uint32_t>(mm.strip_begin[m]);
  }
  const int64_t col0 = static_cast<int64_t>(strip) * kStripCols + cl * 4;
  // LEAN: byte offsets in 32 bits
  const uint32_t out32 = static_cast<uint32_t>(g.out_features), col32 = strip * kStripCols + cl * 4u;
  const uint32_t wstride = out32 * 4u;  // bytes from one qweight row to the next
  auto at32 = [](const auto* base, uint32_t byte_off) {
    return reinterpret_cast<decltype(base)>(reinterpret_cast<const char*>(base) + byte_off);
  };
  // 16-byte loads of x need aligned rows
  const bool x_vec = LEAN || ((reinterpret_cast<uintptr_t>(x) & 15u) == 0 && (g.in_features & 3) == 0);
  // LEAN: 32-bit channel arithmetic (in_features, batch * in_features < 2^31: host)
  const int in32 = static_cast<int>(g.in_features), batch32 = static_cast<int>(g.batch);
  auto lean_x = [&](int brow, int k) -> f32x4 {  // 4 channels from k of batch row brow, zeros when out of range
    const int kc = k < in32 ? k : in32 - 4;
    const int rc = brow < batch32 ? brow : batch32 - 1;
    const f32x4 t = *reinterpret_cast<const f32x4*>(at32(x, static_cast<uint32_t>(rc * in32 + kc) * 4u));
    // zeroed by a bit MASK: a select (of a vector, on a scalar condition) becomes a branch and the load sinks into it;
    // a multiply by 0 / 1 would turn an inf or NaN of the clamped element into a NaN of a channel that does not exist
    const uint32_t keep = ((k < in32) & (brow < batch32)) ? 0xffffffffu : 0u;
    return __builtin_bit_cast(f32x4, __builtin_bit_cast(u32x4, t) & keep);
  };
  auto lean_group = [&](int k0) -> int {
    const int grp = static_cast<int>(static_cast<uint32_t>(k0) / static_cast<uint32_t>(g.group_size));
    return grp < g.groups ? grp : g.groups - 1;
  };
  auto lean_sz = [&](int grp, float (&sc)[4], float (&zr)[4]) {
    const uint32_t o = (col32 * static_cast<uint32_t>(g.groups) + static_cast<uint32_t>(grp)) * 4u;
    const uint32_t step = static_cast<uint32_t>(g.groups) * 4u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sc[j] = *at32(scales, o + static_cast<uint32_t>(j) * step);
      zr[j] = *at32(zeros, o + static_cast<uint32_t>(j) * step);
    }
  };
  // byte offsets of kRows consecutive qweight rows from row0, rows past the end clamped to the last one
  auto lean_rows = [&](int64_t row0, uint32_t (&off)[CH * BITS / 32]) {
    const int last = g.H - 1;
    const int r0 = row0 < last ? static_cast<int>(row0) : last;
    off[0] = (static_cast<uint32_t>(r0) * out32 + col32) * 4u;
#pragma unroll
    for (int i = 1; i < CH * BITS / 32; ++i) off[i] = off[i - 1] + (r0 + i <= last ? wstride : 0u);
  };

  const bool owner = threadIdx.x < kBT * kStripCols;  // one thread per (batch row of the tile, column)
  const int ob = threadIdx.x / kStripCols, occ = threadIdx.x - ob * kStripCols;
  const int64_t ocol = static_cast<int64_t>(strip) * kStripCols + occ;
  // batch rows in tiles of kBT (a mat-VEC is one tile; up to 32 rows re-read the strip's weights -- out of L2 /
  // Infinity Cache from the second tile on -- inside the same launch)
  for (int64_t b0 = 0; b0 < g.batch; b0 += kBT) {
  float acc[4][kBT];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int b = 0; b < kBT; ++b) acc[j][b] = 0.0f;
  // An unsplit workgroup adds its strip's sums to `out` (pre-filled with the bias) itself: the value it adds to is
  // requested HERE, with the first weight words, instead of after the fold -- where the read was one more exposed
  // memory round trip (~1.3 us of a 7.7 us launch at 4096 x 4096) at the very end of the kernel.
  // (unconditionally, every thread, clamped to a valid element: a load behind a branch costs the compiler its count of
  // the loads in flight, and it then waits for all of them at the join)
  // (LEAN: `out` is the one pointer of the prologue that is not preloaded -- its load is placed BEHIND the first pass's
  // loads, so that waiting for the argument does not hold them up)
  float out_prev = 0.0f;
  auto lean_out_prev = [&]() -> float {
    const uint32_t orow = static_cast<uint32_t>(b0) + ((ob < kBT && static_cast<int>(b0) + ob < batch32) ? ob : 0);
    return *at32(out, (orow * out32 + strip * kStripCols + static_cast<uint32_t>(occ)) * 4u);
  };
  if constexpr (!LEAN) out_prev = out[(b0 + ((ob < kBT && b0 + ob < g.batch) ? ob : 0)) * g.out_features + ocol];

  if constexpr (PF) {
    __shared__ __attribute__((aligned(1024))) uint8_t wring[kThreads * 16 * kRows];  // [row i][lane] 16-byte words
    const uint32_t wave_base = __builtin_amdgcn_readfirstlane(
        static_cast<uint32_t>(reinterpret_cast<uintptr_t>(wring)) + (threadIdx.x & ~63u) * 16u);
    const int64_t step = static_cast<int64_t>(split) * kRowsPerPass;
    auto load_small = [&](int64_t pass0, f32x4 (&xg)[kBT][kXLoads], float (&sc)[4], float (&zr)[4]) {
      const int64_t kbase = (pass0 / kRows) * CH;
      if constexpr (LEAN) {
        const int kb32 = static_cast<int>(kbase);
#pragma unroll
        for (int b = 0; b < kBT; ++b)
#pragma unroll
          for (int j = 0; j < kXLoads; ++j)
            xg[b][j] = lean_x(static_cast<int>(b0) + b, kb32 + (j * kThreads + static_cast<int>(threadIdx.x)) * 4);
        lean_sz(lean_group(kb32 + kl * CH), sc, zr);
        return;
      }
#pragma unroll
      for (int b = 0; b < kBT; ++b)
#pragma unroll
        for (int j = 0; j < kXLoads; ++j) {
          const int e = (j * kThreads + threadIdx.x) * 4;
          const int64_t k = kbase + e;
          xg[b][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
          if (b0 + b < g.batch) {
            const float* xr = x + (b0 + b) * g.in_features + k;
            if (x_vec && k + 4 <= g.in_features) {
              xg[b][j] = *reinterpret_cast<const f32x4*>(xr);
            } else {
#pragma unroll
              for (int n = 0; n < 4; ++n)
                if (k + n < g.in_features) xg[b][j][n] = xr[n];
            }
          }
        }
      const int64_t k0 = kbase + static_cast<int64_t>(kl) * CH;
#pragma unroll
      for (int j = 0; j < 4; ++j) sc[j] = zr[j] = 0.0f;
      if (pass0 + kl * kRows < g.H) {
        const int grp = static_cast<int>(k0 / g.group_size);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          sc[j] = scales[(col0 + j) * g.groups + grp];
          zr[j] = zeros[(col0 + j) * g.groups + grp];
        }
      }
    };
    auto dma_weights = [&](int64_t pass0) {
      const int64_t row0 = pass0 + kl * kRows;
      if constexpr (LEAN) {
        uint32_t off[kRows];
        lean_rows(row0, off);
#pragma unroll
        for (int i = 0; i < kRows; ++i) glds16s(qw, off[i], wave_base + static_cast<uint32_t>(i) * (kThreads * 16u));
        return;
      }
#pragma unroll
      for (int i = 0; i < kRows; ++i) {
        int64_t r = row0 + i;
        if (r >= g.H) r = g.H - 1;  // valid address; a dead K lane never uses the words
        glds16(qw + r * g.out_features + col0, wave_base + static_cast<uint32_t>(i) * (kThreads * 16u));
      }
    };
    f32x4 xg_n[kBT][kXLoads];
    float sc_n[4], zr_n[4];
    const int64_t first = static_cast<int64_t>(blockIdx.y) * kRowsPerPass;
    if (first < g.H) {
      load_small(first, xg_n, sc_n, zr_n);
      dma_weights(first);
    }
    if constexpr (LEAN) out_prev = lean_out_prev();
    GPTQ_STAMP(1);
    for (int64_t pass0 = first; pass0 < g.H; pass0 += step) {
      const bool live = pass0 + kl * kRows < g.H;
      // this pass's weights (DMA) and small loads have landed
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (pass0 == first) GPTQ_STAMP(2);
      else GPTQ_STAMP(3);
      f32x4 xg[kBT][kXLoads];
      float sc[4], zr[4];
#pragma unroll
      for (int b = 0; b < kBT; ++b)
#pragma unroll
        for (int j = 0; j < kXLoads; ++j) xg[b][j] = xg_n[b][j];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        sc[j] = sc_n[j];
        zr[j] = zr_n[j];
      }
      u32x4 w[kRows];
#pragma unroll
      for (int i = 0; i < kRows; ++i)
        w[i] = *reinterpret_cast<const u32x4*>(wring + static_cast<size_t>(i) * (kThreads * 16) + threadIdx.x * 16);
      __syncthreads();  // previous pass done with xs
#pragma unroll
      for (int b = 0; b < kBT; ++b)
#pragma unroll
        for (int j = 0; j < kXLoads; ++j) {
          const int e = (j * kThreads + threadIdx.x) * 4;
          const int lane_k = e / CH, off = e - lane_k * CH;
          f32x4 t = xg[b][j];
          if constexpr (DEC8 && BITS == 2) {
            const int k = (off >> 2) & 3;
            float* dst = &xs[b][lane_k * kXStride + (off & ~15) + 8 * (k >> 1) + (k & 1)];
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[2 * r] = t[r];
          } else {
            if constexpr (DEC8) t = f32x4{t[0], t[2], t[1], t[3]};
            *reinterpret_cast<f32x4*>(&xs[b][lane_k * kXStride + off]) = t;
          }
        }
      // the weight words are in registers: the ring is free for the next pass, whose loads now overlap this
      // pass's arithmetic
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (pass0 + step < g.H) {
        load_small(pass0 + step, xg_n, sc_n, zr_n);
        dma_weights(pass0 + step);
      }
      __syncthreads();
      // (LEAN: unconditionally -- a dead K lane's activations are zeros, it adds exact zeros; under a branch the compiler
      // SINKS the weight loads into it, behind the barriers, one more exposed memory round trip)
      if (LEAN || live) strip_compute<BITS, kBT, CH, DEC8, KL * kXStride>(w, &xs[0][0], kl, sc, zr, acc);
    }
  } else {
  // passes y, y + S, y + 2S, ... of the K dimension
  for (int64_t pass0 = static_cast<int64_t>(blockIdx.y) * kRowsPerPass; pass0 < g.H;
       pass0 += static_cast<int64_t>(split) * kRowsPerPass) {
    const int64_t row0 = pass0 + kl * kRows;
    const bool live = row0 < g.H;
    // every global load of the pass is issued before anything waits: this lane's share of the
    // activations (4 x 16 bytes per batch row, coalesced), its weight words, its scale / zero
    const int64_t kbase = (pass0 / kRows) * CH;
    f32x4 xg[kBT][kXLoads];
    u32x4 w[kRows];
    const int64_t k0 = kbase + static_cast<int64_t>(kl) * CH;
    float sc[4] = {0, 0, 0, 0}, zr[4] = {0, 0, 0, 0};
    if constexpr (LEAN) {
      const int kb32 = static_cast<int>(kbase);
#pragma unroll
      for (int b = 0; b < kBT; ++b)
#pragma unroll
        for (int j = 0; j < kXLoads; ++j)
          xg[b][j] = lean_x(static_cast<int>(b0) + b, kb32 + (j * kThreads + static_cast<int>(threadIdx.x)) * 4);
      uint32_t off[kRows];
      lean_rows(row0, off);
#pragma unroll
      for (int i = 0; i < kRows; ++i) {  // rows past the end: a valid address, words zeroed
        const u32x4 t = ld16<true>(at32(qw, off[i]));
        w[i] = t & (row0 + i < g.H ? 0xffffffffu : 0u);
      }
      lean_sz(lean_group(static_cast<int>(k0)), sc, zr);
      out_prev = lean_out_prev();  // every pass (one dword, the same element): unconditional, behind the pass's loads
    } else {
#pragma unroll
    for (int b = 0; b < kBT; ++b)
#pragma unroll
      for (int j = 0; j < kXLoads; ++j) {
        const int e = (j * kThreads + threadIdx.x) * 4;  // element of the pass, multiple of 4
        const int64_t k = kbase + e;
        xg[b][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (b0 + b < g.batch) {
          const float* xr = x + (b0 + b) * g.in_features + k;
          if (x_vec && k + 4 <= g.in_features) {
            xg[b][j] = *reinterpret_cast<const f32x4*>(xr);
          } else {
#pragma unroll
            for (int n = 0; n < 4; ++n)
              if (k + n < g.in_features) xg[b][j][n] = xr[n];
          }
        }
      }
#pragma unroll
    for (int i = 0; i < kRows; ++i) {
      const int64_t r = row0 + i;
      w[i] = u32x4{0, 0, 0, 0};
      if (live && r < g.H) w[i] = ld16<true>(qw + r * g.out_features + col0);
    }
    if (live) {
      const int grp = static_cast<int>(k0 / g.group_size);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        sc[j] = scales[(col0 + j) * g.groups + grp];
        zr[j] = zeros[(col0 + j) * g.groups + grp];
      }
    }
    }
    GPTQ_STAMP(1);
    __syncthreads();  // previous pass done with xs
    GPTQ_STAMP(2);
#pragma unroll
    for (int b = 0; b < kBT; ++b)
#pragma unroll
      for (int j = 0; j < kXLoads; ++j) {
        const int e = (j * kThreads + threadIdx.x) * 4;
        const int lane_k = e / CH, off = e - lane_k * CH;
        f32x4 t = xg[b][j];
        if constexpr (DEC8 && BITS == 2) {
          // pair order of the 2-bit converts inside a 16-channel word: (x_s, x_s+4) for s = 0..3, then
          // (x_s+8, x_s+12): channel c = 4k + r of the word goes to slot 8*(k/2) + 2*r + (k&1)
          const int k = (off >> 2) & 3;
          float* dst = &xs[b][lane_k * kXStride + (off & ~15) + 8 * (k >> 1) + (k & 1)];
#pragma unroll
          for (int r = 0; r < 4; ++r) dst[2 * r] = t[r];
        } else {
          if constexpr (DEC8) t = f32x4{t[0], t[2], t[1], t[3]};  // 4-bit: (x0,x2) (x1,x3) within each 4
          *reinterpret_cast<f32x4*>(&xs[b][lane_k * kXStride + off]) = t;
        }
      }
    __syncthreads();
    GPTQ_STAMP(3);
    // (LEAN: unconditionally -- a dead K lane's activations are zeros, it adds exact zeros; under a branch the compiler
      // SINKS the weight loads into it, behind the barriers, one more exposed memory round trip)
      if (LEAN || live) strip_compute<BITS, kBT, CH, DEC8, KL * kXStride>(w, &xs[0][0], kl, sc, zr, acc);
  }
  }
  // fold the K lanes, in a fixed order.  Mat-vec: a wave holds 8 K lanes x 8 column quads (lane = 8 * (K lane % 8) +
  // quad), so its K lanes meet through three butterfly shuffles (lanes 8, 16, 32 apart: ((k0 + k1) + (k2 + k3)) + ...),
  // and only one partial per wave goes through LDS -- 4 or 8 values per column for the final ascending sum instead of
  // 32 or 64 read one after the other (0.6 us of the 512-thread kernel's 5.3, tools/lab/gptq_stamps.py).  Otherwise:
  // every K lane's partial through LDS, summed in ascending order.
  GPTQ_STAMP(4);
  if constexpr (kShuffleFold) {
#pragma unroll
    for (int b = 0; b < kBT; ++b)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = acc[j][b];
        v += __shfl_xor(v, 8);
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        acc[j][b] = v;
      }
  }
  __syncthreads();  // the previous tile's readers are done with `red`
  GPTQ_STAMP(5);
  if (!kShuffleFold || (threadIdx.x & (kWave - 1)) < 8) {
#pragma unroll
    for (int b = 0; b < kBT; ++b)
#pragma unroll
      for (int j = 0; j < 4; ++j) red[kShuffleFold ? threadIdx.x / kWave : kl][b][cl * 4 + j] = acc[j][b];
  }
  __syncthreads();
  if (owner && b0 + ob < g.batch) {
    float t = 0.0f;
#pragma unroll
    for (int q = 0; q < kRedRows; ++q) t += red[q][ob][occ];
    if (split == 1)  // this workgroup saw all of K: add to out (pre-filled with the bias)
      out[(b0 + ob) * g.out_features + ocol] = out_prev + t;
    else
      __hip_atomic_store(&part[(static_cast<int64_t>(blockIdx.y) * g.batch + b0 + ob) * g.out_features + ocol], t,
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  }  // batch tiles
  GPTQ_STAMP(6);
  if (split == 1) return;
  strip_fold_partials<kThreads>(strip, split, g, part, out, arrivals, s_prev);
  GPTQ_STAMP(7);
}

// HBM-sized matrices, B <= 2: PERSISTENT strip workers.  A 134 MB matrix is 1024 strips x 4 passes: as a grid of
// (strip, K block) workgroups every workgroup lives for two passes, and its start-up (first loads with the full
// memory latency exposed) and its fold dominate -- the strip kernel reaches 3.9 TB/s where a pure read of the same
// 128-byte pieces gets 6.4 (tools/lab/strip_read.hip).  Here the grid is what the chip holds at once (two
// workgroups per CU); each workgroup walks whole strips (all of K: no partial tiles, no arrival protocol), strip
// after strip, and the weight words of the next NS - 1 passes are always in flight -- across strip boundaries too.
// NS register sets of weight words per K lane, addressed statically (the pass loop is unrolled NS times; rotating
// by copies would wait for the loads being moved).  Every load in the loop is unconditional -- rows, channels and
// groups are clamped, dead K lanes skip the arithmetic, a worker past its last strip re-reads a valid one: a load
// under a branch makes the compiler wait for everything in flight at the join.  The result is added to `out` by a
// no-return float atomic (one add per element and call: deterministic), because reading `out` back would drain
// the queue of prefetched loads.
// Geometry requirements (host): in_features a multiple of CH, x rows 16-byte aligned, batch == kBT.
template <int N>
struct StaticFor {
  template <typename F>
  static __device__ __forceinline__ bool run(F&& f) {  // stops at the first false
    if constexpr (N > 0) {
      if (!StaticFor<N - 1>::run(f)) return false;
      return f(std::integral_constant<int, N - 1>{});
    } else {
      return true;
    }
  }
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t gptq_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

// Addressing: raw buffer loads.  Everything that changes from pass to pass is wave-uniform and lives in the SGPR
// offset (computed on the scalar unit); a thread's own offsets (its K lane's rows, its column quad, its slice of
// the activations, its columns' scale rows) are computed ONCE.  The range check of a raw buffer covers the sum of
// both offsets on gfx950 (tools/lab/buf_oob.hip), so rows past the end of the matrix and channels past the end of x
// read as zero -- no clamps, no selects (they belong to dead K lanes; a group index past the end of a scale row
// reads the next column's, which nobody uses either).
// Before this, 64-bit address arithmetic and clamps were 170 of the ~800 vector instructions of a pass, and the
// in-loop sums of the activations (now taken once per worker: `xsum_h`) another 96: the decode itself needs ~380,
// and at 800 the kernel was bound by the vector ALU, not by memory.
template <int BITS, int kBT, int NS, bool DEC8, int DW = NS - 1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gptq_stream_kernel(
    const float* __restrict__ x, const int32_t* __restrict__ qw, const float* __restrict__ scales,
    const float* __restrict__ zeros, float* __restrict__ out, const GptqGeom g, uint32_t strips) {
  constexpr int KL = 32, CH = 64, kThreads = 256;
  constexpr int kRows = CH * BITS / 32;
  constexpr int kRowsPerPass = KL * kRows;
  constexpr int kXLoads = (KL * CH) / (kThreads * 4);
  constexpr int kXStride = CH + 4;
  constexpr int kMaxHalfGroups = 512;  // in_features <= 32768
  __shared__ __attribute__((aligned(16))) float xs[kBT][KL * kXStride];
  __shared__ float red[KL][kBT][kStripCols + 1];
  __shared__ float xsum_h[kBT][kMaxHalfGroups];  // sum of x over channels [64 h, 64 h + 64)
  __shared__ __attribute__((aligned(16))) float szs[2][KL * CH / 128][kStripCols];  // [scale | zero][group of the pass][column]
  const int cl = threadIdx.x & 7, kl = threadIdx.x >> 3;
  // workers that run side by side on one XCD take adjacent strips (see gptq_strip_kernel)
  uint32_t worker = blockIdx.x;
  if (g.xcd_swizzle && (gridDim.x & 7u) == 0) worker = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const uint32_t row_bytes = static_cast<uint32_t>(g.out_features) * 4u;
  const uint32_t w_bytes = static_cast<uint32_t>(g.H) * row_bytes;
  const uint32_t x_bytes = static_cast<uint32_t>(g.in_features) * 4u;
  const uint32_t s_bytes = static_cast<uint32_t>(g.out_features * g.groups) * 4u;
  const __amdgpu_buffer_rsrc_t rw = gptq_rsrc(qw, w_bytes), rs = gptq_rsrc(scales, s_bytes), rz = gptq_rsrc(zeros, s_bytes);
  // per-thread offsets, fixed for the whole kernel
  uint32_t voff_w[kRows], voff_x[kXLoads];
#pragma unroll
  for (int i = 0; i < kRows; ++i) voff_w[i] = static_cast<uint32_t>(kl * kRows + i) * row_bytes + cl * 16u;
#pragma unroll
  for (int j = 0; j < kXLoads; ++j) voff_x[j] = static_cast<uint32_t>(j * kThreads + threadIdx.x) * 16u;
  // scales / zeros of a pass: 32 columns x 16 groups each (group size 128: host), fetched as ONE 16-byte load per
  // thread -- threads 0..127 the scales, 128..255 the zeros; thread (column c, quarter q) takes groups 4q..4q+3 of
  // its column's row -- and handed to the K lanes through LDS like the activations.  (Eight scattered 4-byte loads
  // per thread and pass, the obvious way, cost 14 of 84 us on 12288 x 49152.)
  const int sz_half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 7);
  const int sz_c = (threadIdx.x & 127) >> 2, sz_q = threadIdx.x & 3;
  const uint32_t voff_sz = (static_cast<uint32_t>(sz_c) * static_cast<uint32_t>(g.groups) + sz_q * 4u) * 4u;
  const __amdgpu_buffer_rsrc_t rsz = sz_half ? rz : rs;

  // a position in this worker's sequence of passes: strips worker, worker + G, ... , each from K = 0 to the end
  struct Cursor {
    uint32_t strip;
    uint32_t pass0;
  };
  const uint32_t H = static_cast<uint32_t>(g.H);
  auto advance = [&](Cursor& c) {
    c.pass0 += kRowsPerPass;
    if (c.pass0 >= H) {
      c.pass0 = 0;
      c.strip += gridDim.x;
    }
  };
  struct Small {
    f32x4 xg[kBT][kXLoads];
    f32x4 sz;  // four groups of one column's scales (threads 0..127) or zeros (128..255)
  };
  auto load_w = [&](const Cursor& c, u32x4 (&w)[kRows]) {
    const uint32_t st = c.strip < strips ? c.strip : strips - 1;  // a worker past its last strip: any valid one
    const uint32_t soff = c.pass0 * row_bytes + st * (kStripCols * 4u);
#pragma unroll
    for (int i = 0; i < kRows; ++i) w[i] = __builtin_amdgcn_raw_buffer_load_b128(rw, voff_w[i], soff, 2);
  };
  auto load_small = [&](const Cursor& c, Small& sm) {
    const uint32_t st = c.strip < strips ? c.strip : strips - 1;
    const uint32_t kbase = (c.pass0 / kRows) * CH;
#pragma unroll
    for (int b = 0; b < kBT; ++b) {
      const uint32_t soff = kbase * 4u;
      const __amdgpu_buffer_rsrc_t r = gptq_rsrc(x + static_cast<int64_t>(b) * g.in_features, x_bytes);
#pragma unroll
      for (int j = 0; j < kXLoads; ++j) {
        const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(r, voff_x[j], soff, 0);
        sm.xg[b][j] = __builtin_bit_cast(f32x4, t);
      }
    }
    const uint32_t soff = (st * kStripCols * static_cast<uint32_t>(g.groups) + (kbase >> 7)) * 4u;
    sm.sz = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsz, voff_sz, soff, 0));
  };

  float acc[4][kBT];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int b = 0; b < kBT; ++b) acc[j][b] = 0.0f;

  u32x4 w[NS][kRows];
  Small sm[NS];
  Cursor cur{worker, 0}, nxt{worker, 0}, far{worker, 0};
  // prologue: the first NS - 1 passes' weights, the first pass's activations / scales
  static_assert(NS >= 3, "the activations / scales run two passes ahead");
  load_small(nxt, sm[0]);
  advance(nxt);
  load_small(nxt, sm[1]);
  advance(nxt);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = 0; s < DW; ++s) {
    load_w(far, w[s]);
    advance(far);
  }
  // ... and, while they travel, the sums of the activations over every half group (the zero-point term of a K
  // lane's 64 channels): once per worker instead of 64 adds per thread and pass
  {
    const int halves = static_cast<int>(g.in_features / CH);
    for (int h = threadIdx.x; h < halves * kBT; h += kThreads) {
      const int b = h / halves, hh = h - b * halves;
      const f32x4* src = reinterpret_cast<const f32x4*>(x + static_cast<int64_t>(b) * g.in_features + hh * CH);
      f32x4 t[CH / 4];
#pragma unroll
      for (int q = 0; q < CH / 4; ++q) t[q] = src[q];
      float sum = 0.0f;
#pragma unroll
      for (int q = 0; q < CH / 4; ++q) sum += (t[q][0] + t[q][1]) + (t[q][2] + t[q][3]);
      xsum_h[b][hh] = sum;
    }
  }
  const bool owner = threadIdx.x < kBT * kStripCols;
  const int ob = threadIdx.x / kStripCols, occ = threadIdx.x - ob * kStripCols;
  bool more = cur.strip < strips;
  while (more) {
    more = StaticFor<NS>::run([&](auto tag) {
      constexpr int s = decltype(tag)::value;
      constexpr int s_far = (s + DW) % NS, s_next = (s + 2) % NS;
      // vector-memory loads return IN ORDER (vmcnt): waiting for the next pass's activations / scales also waits
      // for every weight word requested before them.  So they run two passes ahead and go out before this pass's
      // weight request: the words of the last two requests stay in flight across that wait.
      load_small(nxt, sm[s_next]);
      advance(nxt);
      __builtin_amdgcn_sched_barrier(0);
      load_w(far, w[s_far]);
      advance(far);
      __builtin_amdgcn_sched_barrier(0);  // the requests leave before anything below waits
      const bool live = cur.pass0 + kl * kRows < H;
      __syncthreads();  // previous pass done with xs (first pass: xsum_h complete)
#pragma unroll
      for (int b = 0; b < kBT; ++b)
#pragma unroll
        for (int j = 0; j < kXLoads; ++j) {
          const int e = (j * kThreads + threadIdx.x) * 4;
          const int lane_k = e / CH, off = e - lane_k * CH;
          f32x4 t = sm[s].xg[b][j];
          if constexpr (DEC8 && BITS == 2) {
            const int k = (off >> 2) & 3;
            float* dst = &xs[b][lane_k * kXStride + (off & ~15) + 8 * (k >> 1) + (k & 1)];
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[2 * r] = t[r];
          } else {
            if constexpr (DEC8) t = f32x4{t[0], t[2], t[1], t[3]};
            *reinterpret_cast<f32x4*>(&xs[b][lane_k * kXStride + off]) = t;
          }
        }
#pragma unroll
      for (int n = 0; n < 4; ++n) szs[sz_half][sz_q * 4 + n][sz_c] = sm[s].sz[n];
      __syncthreads();
      if (live) {
        float xsum[kBT];
        const uint32_t hh = (cur.pass0 / kRows) + kl;  // half group of this lane's channels
#pragma unroll
        for (int b = 0; b < kBT; ++b) xsum[b] = xsum_h[b][hh];
        const f32x4 sc4 = *reinterpret_cast<const f32x4*>(&szs[0][kl >> 1][cl * 4]);
        const f32x4 zr4 = *reinterpret_cast<const f32x4*>(&szs[1][kl >> 1][cl * 4]);
        const float sc[4] = {sc4[0], sc4[1], sc4[2], sc4[3]}, zr[4] = {zr4[0], zr4[1], zr4[2], zr4[3]};
        strip_compute<BITS, kBT, CH, DEC8, KL * kXStride, true>(w[s], &xs[0][0], kl, sc, zr, acc, xsum);
      }
      if (cur.pass0 + kRowsPerPass >= H) {
        // the strip is done: fold the K lanes in ascending order, add to `out` (pre-filled with the bias)
#pragma unroll
        for (int b = 0; b < kBT; ++b)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            red[kl][b][cl * 4 + j] = acc[j][b];
            acc[j][b] = 0.0f;
          }
        __syncthreads();
        if (owner) {
          float t = 0.0f;
#pragma unroll
          for (int q = 0; q < KL; ++q) t += red[q][ob][occ];
          __hip_atomic_fetch_add(&out[ob * g.out_features + static_cast<int64_t>(cur.strip) * kStripCols + occ], t,
                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // (the next pass's first barrier separates these reads of `red` from the next strip's writes)
      }
      advance(cur);
      return cur.strip < strips;
    });
  }
}

// out[b,n] += sum over K blocks, ascending
// 16-byte accesses at the device-coherent level (sc1: write-through / L2-bypassing, what the agent-scope atomic
// dword accesses of strip_fold_partials carry); the load is asynchronous -- the caller waits (s_waitcnt vmcnt) before
// it touches the result
__device__ __forceinline__ void st16_sc1(float* p, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ f32x4 ld16_sc1(const float* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// ---- batched mat-mul, 5 <= B <= 32, 4-bit: the batch rows through the matrix cores ----------------------------------
// The reference's kernel takes any batch in one launch (cuda_kernel_4bit.cu:36-81; test_cuda_kernel.py:81-126 runs
// B = 8 .. 32).  With B rows per weight the op stops being a weight STREAM: B = 32 on 4096 x 4096 is 1.07 GFLOP against
// 8.4 MB -- 6.8 us at the fp32 peak, 1.2 us of HBM.  The strip kernels tile the batch four rows at a time and re-read and
// re-decode the strip per tile, and every lane fetches its activations from LDS (two ds_read_b128 per sixteen packed
// FMAs: the LDS pipe is as busy as the vector ALU): 17 us at B = 8, 55 us at B = 32 (0.10 - 0.17 of the fp32 peak).
// Here the contraction runs on v_mfma_f32_16x16x4_f32 -- TRUE fp32 operands and accumulation (the contract is fp32 FMA
// on int nibbles; the fp32 matrix peak equals the vector peak, what the MFMA buys is operand delivery): a lane
// supplies ONE activation and ONE dequantized weight per instruction and the 16 x 16 x 4 products happen inside the
// core, so nothing is broadcast through LDS and the vector ALU is left with the decode (one cvt per two levels and one
// FMA per weight -- a quarter of the MFMA pipe's time).
//   wave tile   64 output columns x 16 (MT = 1) or 32 (MT = 2) batch rows; K in chunks of BPC 128-channel blocks.
//   lane (kg = lane / 16, j = lane % 16) loads ONE 16-byte qweight word per 4 rows: row (4 s + kg), columns n0 + 4 j ..
//   + 3 -- sixteen lanes read 256 contiguous bytes of a row -- i.e. 8 channels of 4 columns = 32 weights = 32 MFMAs:
//   step i of column t multiplies A[m = j][k = kg] = x[row m][channel 8 row + i] with B[k = kg][n = j] = w[that
//   channel][column n0 + 4 j + t].  C[m = 4 kg + c][n = j] of column tile t sits in acc[t][c]: the lane's four
//   accumulators of one batch row are four ADJACENT output columns (one 16-byte store).
//   The four waves of a workgroup take four K chunks of the same columns and meet through LDS in wave order; K blocks
//   beyond that (gridDim.y) publish partial tiles and the last arriver folds them in index order (strip_fold_partials:
//   deterministic, no float atomics).
// Requirements (host): out_features % 64 == 0, in_features % 128 == 0, 16-byte aligned qweight / x rows.
// BITS (round 6): 3- and 2-bit levels through the same kernel.  A lane's unit of work is the fewest qweight rows that hold
// whole levels -- one row of 8 (4 bit) or 16 (2 bit) levels, three rows of 32 (3 bit: a 96-bit stream, quant.py:230-257)
// -- so a 128-channel block is 16 / 8 / 4 units, lane kg takes unit 4 s + kg (s < 4 / 2 / 1) and its MFMA step i
// multiplies x[row m][channel kW unit + i] with level i of the unit; the levels are decoded EIGHT at a time per column
// (32 registers), then 32 MT MFMAs issue back to back, as before.  4 bit: v_cvt_pk_f32_fp8 of a nibble alone in a byte
// (level 2^-9, exact); 2 bit: the same on crumbs ((word >> 2 k) & 0x03030303: byte b is level 4 b + k); 3 bit:
// v_bfe_u32 / v_alignbit (stream_level) + v_cvt_f32_u32.
template <int MT, int BITS>
__global__ __launch_bounds__(256) void gptq_mfma_kernel(const float* __restrict__ x, const int32_t* __restrict__ qw,
                                                        const float* __restrict__ scales, const float* __restrict__ zeros,
                                                        float* __restrict__ out, float* __restrict__ part,
                                                        uint32_t* __restrict__ arrivals, const GptqGeom g, int bpc) {
  constexpr int kTileCols = 64;
  constexpr int kW = BITS == 4 ? 8 : (BITS == 2 ? 16 : 32);  // levels (channels) per unit
  constexpr int kR = BITS == 3 ? 3 : 1;                      // qweight rows per unit
  constexpr int kUPB = 128 / kW;                             // units per 128-channel block
  constexpr int kS = kUPB / 4;                               // units per lane and block
  constexpr int kXV = kW / 4;                                // 16-byte activation loads per unit
  __shared__ __attribute__((aligned(16))) f32x4 red[4][MT][4][kWave];  // [wave][batch tile][c][lane]: 16 KB x MT
  __shared__ uint32_t s_prev;
  GPTQ_STAMP_INIT();
  GPTQ_STAMP(0);
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  const int kg = lane >> 4, j = lane & 15;
  // XCD-aware tile order (as the strip kernels): the workgroups that run side by side on one XCD read adjacent
  // 256-byte pieces of the same rows
  uint32_t tile = blockIdx.x;
  if (g.xcd_swizzle) tile = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const uint32_t out32 = static_cast<uint32_t>(g.out_features), in32 = static_cast<uint32_t>(g.in_features);
  const uint32_t col = tile * kTileCols + 4u * j;  // this lane's four columns
  const int nblk = static_cast<int>(in32 / 128u);  // 128-channel blocks (16 qweight rows) of K
  const int chunk = static_cast<int>(blockIdx.y) * 4 + wid;
  const int blk_begin = chunk * bpc, blk_end = blk_begin + bpc < nblk ? blk_begin + bpc : nblk;
  // batch rows of this lane's A operand, clamped (rows >= batch are computed on a valid row and never stored).
  // 32-bit element offsets throughout (host-checked: every tensor below 4 GB)
  uint32_t xrow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int b = mt * 16 + j;
    xrow[mt] = static_cast<uint32_t>(b < g.batch ? b : static_cast<int>(g.batch) - 1) * in32;
  }
  const uint32_t groups32 = static_cast<uint32_t>(g.groups);
  const uint32_t col_lane = (tile * kTileCols + static_cast<uint32_t>(lane)) * groups32;  // scale / zero: ONE column per lane
  f32x4 tot[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < 4; ++t) tot[mt][t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

  // One 128-channel block = 4 weight words + 4 x 8 activations per batch tile per lane, and the group's scale / zero'
  // of ONE column per lane (lane L: column L of the tile; the four columns a lane works on come by shuffle -- a load of
  // four scattered dwords per lane touches 64 cache lines per instruction, eight such instructions per block were
  // a third of the 3 us this kernel took to ISSUE its first loads).
  // Every load is unconditional (a block index past the chunk's end is clamped to its last block and the result
  // ignored: a load under a branch makes the compiler wait for everything in flight at the join) and the NEXT block's
  // loads are issued before the current block's arithmetic (two register sets, the loop unrolled by two).
  struct Blk {
    u32x4 w4[kS][kR];
    f32x4 xa[kS][MT][kXV];
    float s_lane, z_lane;
  };
  const int blk_last = blk_end - 1;
  auto at32 = [](const auto* base, uint32_t byte_off) {
    return reinterpret_cast<decltype(base)>(reinterpret_cast<const char*>(base) + byte_off);
  };
  auto load_blk = [&](int blk_in, Blk& r) {
    const uint32_t blk = static_cast<uint32_t>(blk_in < blk_last ? blk_in : blk_last);
#pragma unroll
    for (int s = 0; s < kS; ++s) {
      // (uniform base + 32-bit BYTE offset per lane: one address register and no 64-bit arithmetic per load)
      const uint32_t unit = blk * static_cast<uint32_t>(kUPB) + 4u * s + kg;
#pragma unroll
      for (int rr = 0; rr < kR; ++rr) r.w4[s][rr] = ld16<true>(at32(qw, ((unit * kR + rr) * out32 + col) * 4u));
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const float* xp = at32(x, (xrow[mt] + unit * static_cast<uint32_t>(kW)) * 4u);
#pragma unroll
        for (int v = 0; v < kXV; ++v) r.xa[s][mt][v] = *reinterpret_cast<const f32x4*>(xp + 4 * v);
      }
    }
    const uint32_t grp = (blk * 128u) / static_cast<uint32_t>(g.group_size);
    r.s_lane = *at32(scales, (col_lane + grp) * 4u);
    r.z_lane = *at32(zeros, (col_lane + grp) * 4u);
  };
  // The block's arithmetic, dequantization factored out of the contraction (as in the strip kernels):
  //     sum_k (s * lvl_k - z) * x_k  =  s * sum_k lvl_k * x_k  -  z * sum_k x_k
  // so the B operand of an MFMA is the decoded LEVEL itself -- the output of v_cvt_pk_f32_fp8, no per-weight FMA
  // in front of the matrix core -- and scale / zero' meet the 16 x 16 tile once per block.  Per word set: the 32 levels
  // are decoded into 32 DISTINCT registers first, then the 32 x MT MFMAs issue back to back (a VALU write to a
  // register an MFMA in flight still reads stalls the pipe: the first version of this kernel, one re-used operand
  // register per MFMA, ran 65 cycles per instruction instead of 32).
  // live == false (the odd block of a chunk with an odd block count): the block's sums are computed and dropped by a
  // select -- not branched around (loads that only a conditional block uses are sunk into it and waited for one by
  // one there), and not multiplied by zero (an inf / NaN of the clamped block must not leak).
  auto compute_blk = [&](const Blk& r, bool live) {
    float xsum[MT][4], sc[4], zr[4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      f32x4 v4 = r.xa[0][mt][0] + r.xa[0][mt][1];
#pragma unroll
      for (int v = 2; v < kXV; ++v) v4 += r.xa[0][mt][v];
#pragma unroll
      for (int s = 1; s < kS; ++s) {
        f32x4 u4 = r.xa[s][mt][0] + r.xa[s][mt][1];
#pragma unroll
        for (int v = 2; v < kXV; ++v) u4 += r.xa[s][mt][v];
        v4 += u4;
      }
      float xs = (v4[0] + v4[1]) + (v4[2] + v4[3]);  // batch row j, this lane's 32 channels of the block
      xs += __shfl_xor(xs, 16);
      xs += __shfl_xor(xs, 32);  // ... all 128 channels (the four K sub-lanes kg)
#pragma unroll
      for (int c = 0; c < 4; ++c) xsum[mt][c] = __shfl(xs, 4 * kg + c);  // batch row 4 kg + c: where this lane's C values live
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      // (4 / 2 bit: the e4m3 decode yields level * 2^-9, exact: 2^9 into the scale)
      sc[t] = __shfl(r.s_lane, 4 * j + t) * (BITS == 3 ? 1.0f : 512.0f);
      zr[t] = __shfl(r.z_lane, 4 * j + t);
    }
    f32x4 acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[mt][t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int s = 0; s < kS; ++s) {
      // (decode and MFMAs in separate phases: interleaving the NEXT set's decode into the MFMA stream -- one VALU
      // instruction per MFMA issue slot via sched_group_barrier -- was measured slower, 11.6 vs 10.9 us at B = 8 and
      // 18.9 vs 18.0 us at B = 32 on 4096 x 4096: a filler beside every fp32 MFMA costs more than the 28-instruction
      // decode phase it hides)
#pragma unroll
      for (int ch = 0; ch < kW / 8; ++ch) {  // eight levels of the unit at a time
        f32x2 lp[4][4];  // column t, levels 8 ch + ...: pairs (0, 2) (1, 3) (4, 6) (5, 7) for 4 bit; see lvl() below
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if constexpr (BITS == 4) {
            const uint32_t word = r.w4[s][0][t];
            const uint32_t even = word & 0x0f0f0f0fu, odd = (word >> 4) & 0x0f0f0f0fu;
            lp[t][0] = __builtin_amdgcn_cvt_pk_f32_fp8(even, false);
            lp[t][1] = __builtin_amdgcn_cvt_pk_f32_fp8(odd, false);
            lp[t][2] = __builtin_amdgcn_cvt_pk_f32_fp8(even, true);
            lp[t][3] = __builtin_amdgcn_cvt_pk_f32_fp8(odd, true);
          } else if constexpr (BITS == 2) {
            // crumb k of every byte: byte b of ((word >> 2 k) & 0x03030303) is level 4 b + k; this chunk's bytes are
            // 2 ch and 2 ch + 1: lp[t][k] = levels (8 ch + k, 8 ch + 4 + k)
            const uint32_t word = r.w4[s][0][t];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const uint32_t m = (word >> (2 * k)) & 0x03030303u;
              if (ch == 0) lp[t][k] = __builtin_amdgcn_cvt_pk_f32_fp8(m, false);  // (the selector is an immediate)
              else lp[t][k] = __builtin_amdgcn_cvt_pk_f32_fp8(m, true);
            }
          } else {
            // lp[t][p] = levels (8 ch + 2 p, 8 ch + 2 p + 1) of the 96-bit stream
#pragma unroll
            for (int p2 = 0; p2 < 4; ++p2) {
              const auto wd = [&](int idx) { return r.w4[s][idx][t]; };
              lp[t][p2] = f32x2{static_cast<float>(stream_level<3>(wd, 8 * ch + 2 * p2)),
                                static_cast<float>(stream_level<3>(wd, 8 * ch + 2 * p2 + 1))};
            }
          }
#pragma unroll
          for (int p2 = 0; p2 < 4; ++p2) asm volatile("" : "+v"(lp[t][p2]));  // the decoded pair EXISTS here, in registers of its own
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            // level 8 ch + i of the unit
            float lv;
            if constexpr (BITS == 4) lv = lp[t][(i >> 2) * 2 + (i & 1)][(i >> 1) & 1];  // pair (i >> 2) * 2 + (i & 1), half (i >> 1) & 1
            else if constexpr (BITS == 2) lv = lp[t][i & 3][i >> 2];
            else lv = lp[t][i >> 1][i & 1];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              const int e = 8 * ch + i;  // channel of the unit
              const float xe = r.xa[s][mt][e >> 2][e & 3];
              acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xe, lv, acc[mt][t], 0, 0, 0);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float v = tot[mt][t][c] + __builtin_fmaf(sc[t], acc[mt][t][c], -(zr[t] * xsum[mt][c]));
          tot[mt][t][c] = live ? v : tot[mt][t][c];
        }
  };
  if (blk_begin < blk_end) {  // (uniform per wave; a wave past the end of K only takes part in the fold)
    Blk ra, rb;
    load_blk(blk_begin, ra);
    int blk = blk_begin;
    // steady state (three or more blocks left): both prefetches are real
    for (; blk + 2 < blk_end; blk += 2) {
      load_blk(blk + 1, rb);
      __builtin_amdgcn_sched_barrier(0);
      compute_blk(ra, true);
      __builtin_amdgcn_sched_barrier(0);
      load_blk(blk + 2, ra);
      __builtin_amdgcn_sched_barrier(0);
      compute_blk(rb, true);
      __builtin_amdgcn_sched_barrier(0);
    }
    // the last one or two blocks: nothing is requested that nobody will use (the clamped re-load of the last block
    // cost the two-block chunks of 4096 x 4096 twenty load instructions -- ~1 us of issue -- in front of their
    // second block)
    load_blk(blk + 1, rb);  // (clamped to the last block when only one is left: ignored then)
    __builtin_amdgcn_sched_barrier(0);
    GPTQ_STAMP(1);
    compute_blk(ra, true);
    __builtin_amdgcn_sched_barrier(0);
    GPTQ_STAMP(2);
    compute_blk(rb, blk + 1 < blk_end);
    __builtin_amdgcn_sched_barrier(0);
  }
  GPTQ_STAMP(3);
  // the four waves' K chunks, in wave order.  tot[mt][t][c] = row (mt * 16 + 4 kg + c), column (col + t): transposed to
  // one 16-byte value per (row, lane) on the way into LDS
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int c = 0; c < 4; ++c) red[wid][mt][c][lane] = f32x4{tot[mt][0][c], tot[mt][1][c], tot[mt][2][c], tot[mt][3][c]};
  __syncthreads();
  const int split = static_cast<int>(gridDim.y);
  // wave `wid` finishes c = wid of every batch tile
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const f32x4 t = ((red[0][mt][wid][lane] + red[1][mt][wid][lane]) + red[2][mt][wid][lane]) + red[3][mt][wid][lane];
    const int b = mt * 16 + 4 * kg + wid;
    if (b < g.batch) {
      if (split == 1) {
        float* o = out + static_cast<size_t>(b) * out32 + col;
        const f32x4 prev = *reinterpret_cast<const f32x4*>(o);
        *reinterpret_cast<f32x4*>(o) = prev + t;
      } else {
        // publish write-through (sc1): the reader is a workgroup of another XCD, behind another L2
        st16_sc1(part + (static_cast<size_t>(blockIdx.y) * g.batch + b) * out32 + col, t);
      }
    }
  }
  GPTQ_STAMP(4);
  if (split == 1) return;
  // arrival protocol of strip_fold_partials (see there: no agent-scope fence; every access of the protocol goes to the
  // device-coherent level by itself); the fold below differs: 16-byte accesses, and the partials of FOUR K blocks are
  // requested before any is added -- one memory round trip per four instead of one per K block (the scalar fold spent
  // ~1 us per partial: 4 of this kernel's 15 us at 4096 x 4096)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (threadIdx.x == 0) s_prev = __hip_atomic_fetch_add(&arrivals[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  GPTQ_STAMP(5);
  if (s_prev != static_cast<uint32_t>(split - 1)) return;
  const size_t slab = static_cast<size_t>(g.batch) * out32;  // floats per K block
  for (int e = threadIdx.x; e < static_cast<int>(g.batch) * (kTileCols / 4); e += 256) {
    const size_t o = static_cast<size_t>(e >> 4) * out32 + tile * kTileCols + 4u * (e & 15);
    const f32x4 prev = *reinterpret_cast<const f32x4*>(out + o);
    f32x4 total = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int s0 = 0; s0 < split; s0 += 4) {  // (uniform)
      f32x4 p[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) p[u] = ld16_sc1(part + static_cast<size_t>(s0 + u < split ? s0 + u : split - 1) * slab + o);
      // (the loaded values flow THROUGH the wait: nothing that uses them can be scheduled above it)
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3])::"memory");
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (s0 + u < split) total += p[u];  // index order: the sum does not depend on who arrived last
    }
    *reinterpret_cast<f32x4*>(out + o) = prev + total;
  }
  GPTQ_STAMP(6);
  if (threadIdx.x == 0) __hip_atomic_store(&arrivals[tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(kBlock) void gptq_fold_kernel(const float* __restrict__ part,
                                                            float* __restrict__ out, int64_t bn,
                                                            int kblocks) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= bn) return;
  float t = 0.0f;
  for (int kb = 0; kb < kblocks; ++kb) t += part[static_cast<int64_t>(kb) * bn + i];
  out[i] += t;
}

// rows of qweight for `in_f` input channels (QuantLinear.__init__, quant.py:171-183)
int64_t gptq_rows(int bits, int64_t in_f) {
  return bits == 3 ? ceil_div(in_f, 32) * 3 : ceil_div(in_f * bits, 32);
}

bool gptq_geom(int bits, int slice_k, int64_t batch, int64_t in_f, int64_t out_f, int64_t group_size,
               GptqGeom& g) {
  if (batch <= 0 || in_f <= 0 || out_f <= 0) return false;
  if (in_f >= (1ll << 31) || out_f >= (1ll << 31)) return false;
  if (group_size == 0) group_size = in_f;
  if (group_size < 0 || group_size >= (1ll << 31)) return false;
  g.in_features = in_f;
  g.out_features = out_f;
  g.batch = batch;
  g.H = static_cast<int32_t>(gptq_rows(bits, in_f));
  g.group_size = static_cast<int32_t>(group_size);
  g.groups = static_cast<int32_t>(ceil_div(in_f, group_size));
  g.slices = static_cast<int32_t>(ceil_div(in_f, slice_k));
  // enough workgroups to fill 256 CUs a few times over, but not more K blocks than
  // that needs: each K block costs batch*out floats of partial traffic
  const int64_t colblocks = ceil_div(out_f, kWave * 4);
  int64_t want = ceil_div(1024, colblocks);
  if (want < 1) want = 1;
  if (want > g.slices) want = g.slices;
  g.slices_per_block = static_cast<int32_t>(ceil_div(g.slices, want));
  g.kblocks = static_cast<int32_t>(ceil_div(g.slices, g.slices_per_block));
  g.xcd_swizzle = ((out_f / 32) % 8 == 0 && knob(2) != 8) ? 1 : 0;
  g.grid_x = g.grid_y = 0;
  return true;
}

// upper bound of kblocks over every bit width / slice size (the workspace query does not
// take the bit width)
int64_t gptq_max_kblocks(int64_t in_f, int64_t out_f) {
  const int64_t colblocks = ceil_div(out_f, kWave * 4);
  int64_t want = ceil_div(1024, colblocks);
  const int64_t slices = ceil_div(in_f, 64);
  if (want > slices) want = slices;
  return want < 1 ? 1 : want;
}

// General path (any shape, any batch): K blocks write partial tiles, a second kernel adds them in a fixed order.
template <int BITS, int SK>
int gptq_launch_partial(const float* x, const int32_t* qweight, float* out, const float* scales,
                        const float* zeros, float* part, const GptqGeom& g, bool vec, hipStream_t st) {
  const int64_t batch = g.batch;
  // batch rows per register tile: the matrix is read once per tile, so up to 32 rows share one read
  const int bt = batch > 16 ? 32 : (batch > 8 ? 16 : (batch > 4 ? 8 : (batch >= 3 ? 4 : (batch == 2 ? 2 : 1))));
#define SBQ_GPTQ_K(COLS, BT) \
  gptq_partial_kernel<BITS, SK, COLS, BT><<<grid, kBlock, 0, st>>>(x, qweight, scales, zeros, out, part, g)
#define SBQ_GPTQ(COLS)                                                                                          \
  do {                                                                                                          \
    const dim3 grid(static_cast<uint32_t>(ceil_div(g.out_features, kWave * COLS)), static_cast<uint32_t>(g.kblocks)); \
    if (bt == 32) SBQ_GPTQ_K(COLS, 32);                                                                         \
    else if (bt == 16) SBQ_GPTQ_K(COLS, 16);                                                                    \
    else if (bt == 8) SBQ_GPTQ_K(COLS, 8);                                                                      \
    else if (bt == 4) SBQ_GPTQ_K(COLS, 4);                                                                      \
    else if (bt == 2) SBQ_GPTQ_K(COLS, 2);                                                                      \
    else SBQ_GPTQ_K(COLS, 1);                                                                                   \
  } while (0)
  if (vec) SBQ_GPTQ(4);
  else SBQ_GPTQ(1);
#undef SBQ_GPTQ
#undef SBQ_GPTQ_K
  int rc = check_launch();
  if (rc != SBQ_OK || g.kblocks == 1) return rc;
  const int64_t bn = batch * g.out_features;
  gptq_fold_kernel<<<static_cast<uint32_t>(ceil_div(bn, kBlock)), kBlock, 0, st>>>(part, out, bn, g.kblocks);
  return check_launch();
}

// the 32-bit geometry scalars of gptq_strip_kernel's argument list (six dwords after the four pointers): out_features,
// H, {grid x: 20 bits | grid y: 5 | XCD swizzle: 1 | batch: 6}, groups, group size, in_features
#define SBQ_STRIP_LEAD(G)                                                                                  \
  static_cast<int32_t>((G).out_features), (G).H,                                                           \
      (static_cast<uint32_t>((G).grid_x) & 0xfffffu) | ((static_cast<uint32_t>((G).grid_y) & 31u) << 20) |  \
          ((static_cast<uint32_t>((G).xcd_swizzle) & 1u) << 25) | (static_cast<uint32_t>((G).batch) << 26), \
      (G).groups, (G).group_size, static_cast<int32_t>((G).in_features)

// the LEAN strip kernels address every tensor as base + 32-bit byte offset
bool lean_sizes(int64_t H, int64_t batch, int64_t in_f, int64_t out_f, int64_t groups) {
  const int64_t lim = 1ll << 32;
  return H * out_f * 4 < lim && out_f * groups * 4 < lim && batch * in_f * 4 < (lim >> 1) && batch * out_f * 4 < lim;
}

template <int BITS>
int gptq_matmul(const float* x, const int32_t* qweight, float* out, const float* scales,
                const float* zeros, int64_t batch, int64_t in_features, int64_t out_features,
                int64_t group_size, void* workspace, size_t workspace_bytes, void* stream) {
  if (batch < 0 || in_features < 0 || out_features < 0) return SBQ_ERR_ARG;
  if (batch == 0 || in_features == 0 || out_features == 0) return SBQ_ERR_EMPTY;
  if (!x || !qweight || !out || !scales || !zeros || !workspace) return SBQ_ERR_NULL;
  // cuda_kernel_4bit.cu:58-61, cuda_kernel_3bit.cu:56-59: group size must be a multiple of 128
  // (0 = one group); cuda_kernel_2bit.cu:56-59: of 64
  constexpr int kMinGroup = BITS == 2 ? 64 : 128;
  if (group_size != 0 && group_size % kMinGroup != 0) return SBQ_ERR_ARG;
  const bool half_slices = group_size % kSliceK != 0;  // 2-bit with 64-channel group granularity
  GptqGeom g;
  if (!gptq_geom(BITS, half_slices ? 64 : kSliceK, batch, in_features, out_features, group_size, g))
    return SBQ_ERR_ARG;
  const int64_t tiles = g.kblocks > kStripMaxSplit ? g.kblocks : kStripMaxSplit;
  const size_t need = kCounterBytes + static_cast<size_t>(tiles) * batch * out_features * sizeof(float);
  if (workspace_bytes < need || !aligned16(workspace)) return SBQ_ERR_WORKSPACE;
  if ((reinterpret_cast<uintptr_t>(x) & 3u) || (reinterpret_cast<uintptr_t>(qweight) & 3u) ||
      (reinterpret_cast<uintptr_t>(out) & 3u))
    return SBQ_ERR_ALIGN;
  hipStream_t st = as_stream(stream);
  // workspace = [arrival counters (fixed size, zero between calls) | partial tiles]
  {
    const int rc = workspace_guard(workspace, kCounterBytes, st);  // zeroed at first sight, bound to its stream
    if (rc != SBQ_OK) return rc;
  }
  uint32_t* arrivals = static_cast<uint32_t*>(workspace);
  float* part = reinterpret_cast<float*>(static_cast<char*>(workspace) + kCounterBytes);
  const bool vec = (out_features % 4 == 0) && aligned16(qweight);
  // single-launch strip kernel: mat-vec sized batches, whole 32-column strips
  const int64_t strips = out_features / kStripCols;
  // batches of 3 and 4 (a handful of concurrent decode streams) take the same single launch with four
  // batch rows per register tile, half-group K lanes only (register budget)
  // ... and so do batches up to 32 (the reference's kernel takes any batch in one launch,
  // cuda_kernel_4bit.cu:36-81): tiles of four rows, the strip's weights re-read per tile out of the caches
  {
    // 5 <= B <= 32: the batch rows through the fp32 matrix cores (gptq_mfma_kernel; knob 2 == 26: the strip tiles of four
    // rows instead, for A/B runs).  3 / 2 bit since round 6 (the reference's multi-batch cases run all three widths,
    // test_cuda_kernel.py:81-109; before, they took B strip passes)
    // (any batch >= 5: more than 32 rows go through the kernel 32 at a time -- the launches of one call follow each other
    // on the stream and share the partial-tile workspace; the weights of the later tiles come out of L2 / Infinity Cache)
    if (vec && batch >= 5 && out_features % 64 == 0 && in_features % 128 == 0 && group_size % 128 == 0 &&
        aligned16(x) && in_features % 4 == 0 && out_features / 64 <= kMaxStrips && g.H * out_features * 4 < (1ll << 32) &&
        out_features * g.groups < (1ll << 30) && 32 * in_features < (1ll << 30) && knob(2) != 26 && knob(2) != 9) {
      const int64_t tiles64 = out_features / 64, nblk = in_features / 128;
      // Blocks per wave (profiles/r05_gptq_batch_bpc_sweep.log).  The kernel is bound by the matrix cores, four waves
      // keep a CU's four busy, and a launch ends with its slowest workgroup: best is ONE round of equal workgroups --
      // tiles x K blocks just under the CU count (4096 x 4096: 64 tiles x 4 K blocks of 2-block waves; 11008 -> 4096:
      // 64 x 4 of 6-block waves, 20.9 us at B = 8 against 27.2 with 4-block waves = 384 workgroups = one and a half
      // rounds).  Where no block count gives that (4096 -> 11008: 172 tiles), many SMALL workgroups balance
      // dynamically: 2-block waves (688 workgroups, 21.4 us) beat 4-block waves (344: a second workgroup on a third of
      // the CUs, 26.5 us).  At most kStripMaxSplit K blocks (the partial tiles' workspace).
      const int64_t cus = cu_count();
      int64_t bpc = 0, best_wgs = 0;
      for (int64_t cand = 1; cand <= nblk; cand += cand == 1 ? 1 : 2) {  // 1, 2, 4, 6, ...
        const int64_t ks = ceil_div(ceil_div(nblk, cand), 4);
        if (ks > kStripMaxSplit) continue;
        const int64_t wgs = tiles64 * ks;
        if (wgs <= cus && 4 * wgs >= 3 * cus && wgs > best_wgs) {
          best_wgs = wgs;
          bpc = cand;
        }
      }
      if (bpc == 0) {
        bpc = nblk >= 2 ? 2 : 1;
        while (ceil_div(ceil_div(nblk, bpc), 4) > kStripMaxSplit) bpc += 2;
      }
      if (knob(1) > 0 && knob(1) < 128 && ceil_div(ceil_div(nblk, static_cast<int64_t>(knob(1))), 4) <= kStripMaxSplit)
        bpc = knob(1);  // dev override (shares the strip kernels' K-split knob)
      const int64_t ksplit = ceil_div(ceil_div(nblk, bpc), 4);
      g.xcd_swizzle = (tiles64 % 8 == 0 && knob(2) != 8) ? 1 : 0;
      const dim3 grid(static_cast<uint32_t>(tiles64), static_cast<uint32_t>(ksplit));
      for (int64_t b0 = 0; b0 < batch; b0 += 32) {
        GptqGeom gt = g;
        gt.batch = batch - b0 < 32 ? batch - b0 : 32;
        const float* xt = x + b0 * in_features;
        float* ot = out + b0 * out_features;
        if (gt.batch <= 16)
          gptq_mfma_kernel<1, BITS><<<grid, 256, 0, st>>>(xt, qweight, scales, zeros, ot, part, arrivals, gt, static_cast<int>(bpc));
        else
          gptq_mfma_kernel<2, BITS><<<grid, 256, 0, st>>>(xt, qweight, scales, zeros, ot, part, arrivals, gt, static_cast<int>(bpc));
        const int rc = check_launch();
        if (rc != SBQ_OK) return rc;
      }
      return SBQ_OK;
    }
  }
  if (vec && out_features % kStripCols == 0 && batch >= 3 && batch <= 32 && !half_slices && strips <= kMaxStrips &&
      knob(2) != 9) {
    int64_t split = ceil_div(in_features, 32 * (kSliceK / 2));
    if (split > kStripMaxSplit) split = kStripMaxSplit;
    const dim3 grid(static_cast<uint32_t>(strips), static_cast<uint32_t>(split));
    gptq_strip_kernel<BITS, 4, 32, kSliceK / 2, BITS != 3><<<grid, 256, 0, st>>>(qweight, x, scales, zeros,
                                                                                SBQ_STRIP_LEAD(g), out, part, arrivals, g,
                                                                                GptqMulti{});
    return check_launch();
  }
  if (vec && out_features % kStripCols == 0 && batch <= 2 && !half_slices && strips <= kMaxStrips &&
      knob(2) != 9) {
    // 256-thread workgroups (32 K lanes) measured best throughout.  A K lane takes a whole
    // 128-channel group when that alone fills the chip; otherwise half a group, which doubles
    // the workgroup count and halves each one's serial decode (decode shapes: 4096x4096 is 128
    // strips on 256 CUs).  The K split S then covers the K lanes a single pass does not.
    // Half-group K lanes throughout since round 2: with 128 VGPRs four workgroups share a CU (a whole-group lane
    // needs 195: two), which is worth 7-20 % on the HBM-sized shapes and nothing less on the decode shapes.
    int ch = kSliceK / 2;
    if (knob(2) == 1) ch = kSliceK;      // dev overrides
    if (knob(2) == 2) ch = kSliceK / 2;
    // K split: as many K blocks as fill the chip a few times over (about 8 workgroups per CU in total), no more --
    // every extra block is another partial tile, another arrival and another workgroup start-up for the same bytes.
    // A matrix with >= 2048 strips needs none: each workgroup walks all of K and adds to `out` directly.
    const int64_t passes = ceil_div(in_features, 32 * ch);
    // Round 3 (tools/r03_gptq_probe.py, HBM-cold, B = 1): the split's price -- partial tiles, the arrival add and the
    // last arriver's fold are three dependent memory round trips -- exceeds its gain whenever a workgroup's own K walk
    // is short.  4096 -> {4096, 11008, 12288, 22016} (two passes): 7.3 / 11.3 / 11.3 / 15.9 us unsplit against 8.7 /
    // 13.8 / 14.3 / 22.6 with any split; 11008 -> 4096 (six passes, 128 strips): 15.7 unsplit, 12.2 with two K blocks,
    // 15.2 with six.  So: one K block per three passes, more only to put a workgroup on every other CU.
    int64_t split = passes >= 3 ? ceil_div(passes, 3) : 1;
    while (strips * split * 2 < static_cast<int64_t>(cu_count()) && split < passes) ++split;
    if (knob(2) == 13) split = ceil_div(static_cast<int64_t>(cu_count()) * 8, strips);  // round 2's rule, for A/B runs
    if (knob(1) > 0 && knob(1) < 128) split = knob(1);  // dev override (shares the grid-cap knob)
    if (split > passes) split = passes;
    if (split > kStripMaxSplit) split = kStripMaxSplit;
    if (split < 1) split = 1;
    const dim3 grid(static_cast<uint32_t>(strips), static_cast<uint32_t>(split));
#define SBQ_STRIP(CH, D8)                                                                                  \
  do {                                                                                                     \
    if (batch == 2)                                                                                        \
      gptq_strip_kernel<BITS, 2, 32, CH, D8><<<grid, 256, 0, st>>>(qweight, x, scales, zeros, SBQ_STRIP_LEAD(g), out, part, arrivals, g, GptqMulti{}); \
    else                                                                                                   \
      gptq_strip_kernel<BITS, 1, 32, CH, D8><<<grid, 256, 0, st>>>(qweight, x, scales, zeros, SBQ_STRIP_LEAD(g), out, part, arrivals, g, GptqMulti{}); \
  } while (0)
    // half-group K lanes (the default), x aligned: the branch-free loads (knob 2 == 23: the branchy ones, for A/B runs)
#define SBQ_STRIP_LEAN(D8, PFV)                                                                            \
  do {                                                                                                     \
    if (batch == 2)                                                                                        \
      gptq_strip_kernel<BITS, 2, 32, 64, D8, PFV, true><<<grid, 256, 0, st>>>(qweight, x, scales, zeros, SBQ_STRIP_LEAD(g), out, part, arrivals, g, GptqMulti{}); \
    else                                                                                                   \
      gptq_strip_kernel<BITS, 1, 32, 64, D8, PFV, true><<<grid, 256, 0, st>>>(qweight, x, scales, zeros, SBQ_STRIP_LEAD(g), out, part, arrivals, g, GptqMulti{}); \
  } while (0)
    const bool lean = ch == kSliceK / 2 && aligned16(x) && in_features % 4 == 0 &&
                      lean_sizes(g.H, batch, in_features, out_features, g.groups) && knob(2) != 23;
    g.grid_x = static_cast<int32_t>(strips);
    g.grid_y = static_cast<int32_t>(split);
    // several passes per workgroup (HBM-sized matrices): the next pass's weights are prefetched by LDS-DMA
    // (knob 2 == 6: off, for A/B runs)
    const bool prefetch = ch == kSliceK / 2 && passes >= 2 * split && knob(2) != 6;
    {
      // (4- and 2-bit decode through the fp8 converter, 3-bit through byte converts; knob 2 == 4 -- byte converts
      // for A/B runs -- keeps the (strip, K block) grid)
      constexpr bool kDec8 = BITS != 3;
      // HBM-sized: persistent strip workers (knob 2 == 5: the (strip, K block) grid instead, for A/B runs).  Whole
      // strips only, so the workers' shares must come out even: two workers per CU when the strip count is a
      // multiple of that, else one per CU (measured: 4 % slower per strip, but 1152 strips are 4.5 per worker
      // instead of 2.25), and the (strip, K block) grid when even that leaves more than a fifth of the chip idle
      const int64_t cus = cu_count();
      const int64_t rounds2 = ceil_div(strips, 2 * cus), rounds1 = ceil_div(strips, cus);
      int64_t workers = 2.0 * static_cast<double>(rounds2) <= 1.04 * static_cast<double>(rounds1) ? 2 * cus : cus;
      if (knob(1) >= 128) workers = knob(1);  // dev override
      const bool even = static_cast<double>(strips) >= 0.8 * static_cast<double>(workers * ceil_div(strips, workers));
      const bool stream = passes >= 3 && strips >= workers && (even || knob(1) >= 128) &&
                          in_features % 64 == 0 && in_features <= 32768 && aligned16(x) && (cus % 8) == 0 &&
                          g.group_size == 128 &&
                          g.H * out_features * 4 < (1ll << 32) && out_features * g.groups * 4 < (1ll << 32) &&
                          (BITS == 4 || batch == 1) &&  // 3- / 2-bit with two batch rows: over the register budget
                          knob(2) != 5 && knob(2) != 4;
      if (stream) {
        const uint32_t n_strips = static_cast<uint32_t>(strips);
        // weight words ONE pass ahead: with 128-byte row pieces more requests in flight make the memory system
        // slower, not faster (tools/lab/strip_read.hip: a bare read by 512 persistent workers gets 5.3 TB/s with
        // one set in flight, 4.3 with three, 4.1 with five; this kernel 0.62 / 0.60 / 0.54 of peak on
        // 12288 x 49152 with one / two / three)
        if constexpr (BITS == 4) {
          if (batch == 2) {
            gptq_stream_kernel<4, 2, 3, true, 1><<<static_cast<uint32_t>(workers), 256, 0, st>>>(x, qweight, scales, zeros, out, g, n_strips);
            return check_launch();
          }
        }
        if (batch == 1)
          gptq_stream_kernel<BITS, 1, 3, kDec8, 1><<<static_cast<uint32_t>(workers), 256, 0, st>>>(x, qweight, scales, zeros, out, g, n_strips);
        return check_launch();
      }
    }
    // WIDE workgroups (64 K lanes, 512 threads): all of a 4096-channel K in ONE pass -- every weight word of the
    // strip requested at once, one memory round trip instead of two dependent ones, and the arithmetic on eight
    // waves instead of four.  Measured (tools/lab/r04_gptq_narrow.py, profiles/r04_gptq_lean_wide.log, HBM-cold, us,
    // 256-thread lean kernel -> wide):  B = 1, 4096 -> 11008: 4-bit 10.8 -> 10.1, 3-bit 18.4 -> 11.6, 2-bit 11.2 -> 8.0;
    // 4096 -> 4096: 6.5 -> 6.5 / 12.3 -> 8.1 / 7.5 -> 5.7; 11008 -> 4096 (three wide passes, one K block each):
    // 11.3 -> 11.7 / 17.7 -> 13.8 / 11.3 -> 9.6.  B = 2: a gain for one pass (4-bit 12.8 -> 11.5, 3-bit 14.5 -> 13.3),
    // a loss for 2-bit (11.0 -> 12.3) and for several passes.
    // So: one wide pass -> wide (2-bit: B = 1 only); several -> wide for the 3- and 2-bit mat-VEC only, one K block
    // per pass from three passes on.  knob 2 == 24 / 25: always / never, for A/B runs.
    {
      const int64_t wpasses = ceil_div(in_features, 64 * ch);
      bool wide = lean && (wpasses == 1 ? (BITS != 2 || batch == 1) : (BITS != 4 && batch == 1));
      if (knob(2) == 24) wide = lean;
      if (knob(2) == 25) wide = false;
      if (wide) {
        int64_t wsplit = wpasses >= 3 ? wpasses : 1;
        if (knob(1) > 0 && knob(1) < 128) wsplit = knob(1);
        if (wsplit > wpasses) wsplit = wpasses;
        if (wsplit > kStripMaxSplit) wsplit = kStripMaxSplit;
        const dim3 wgrid(static_cast<uint32_t>(strips), static_cast<uint32_t>(wsplit));
        g.grid_y = static_cast<int32_t>(wsplit);
        constexpr bool kD8 = BITS != 3;
        if (batch == 2)
          gptq_strip_kernel<BITS, 2, 64, 64, kD8, false, true><<<wgrid, 512, 0, st>>>(qweight, x, scales, zeros, SBQ_STRIP_LEAD(g), out, part, arrivals, g, GptqMulti{});
        else
          gptq_strip_kernel<BITS, 1, 64, 64, kD8, false, true><<<wgrid, 512, 0, st>>>(qweight, x, scales, zeros, SBQ_STRIP_LEAD(g), out, part, arrivals, g, GptqMulti{});
        return check_launch();
      }
    }
    if constexpr (BITS == 4 || BITS == 2) {
      if (knob(2) != 4) {  // packed e4m3 decode (knob 2 == 4: byte converts, for A/B runs)
        if (lean) {
          if (prefetch) SBQ_STRIP_LEAN(true, true);
          else SBQ_STRIP_LEAN(true, false);
          return check_launch();
        }
        if (prefetch) {
          if (batch == 2)
            gptq_strip_kernel<BITS, 2, 32, 64, true, true><<<grid, 256, 0, st>>>(qweight, x, scales, zeros, SBQ_STRIP_LEAD(g), out, part, arrivals, g, GptqMulti{});
          else
            gptq_strip_kernel<BITS, 1, 32, 64, true, true><<<grid, 256, 0, st>>>(qweight, x, scales, zeros, SBQ_STRIP_LEAD(g), out, part, arrivals, g, GptqMulti{});
          return check_launch();
        }
        if (ch == kSliceK) SBQ_STRIP(128, true);
        else SBQ_STRIP(64, true);
        return check_launch();
      }
    }
    if (lean && BITS == 3) {
      if (prefetch) SBQ_STRIP_LEAN(false, true);
      else SBQ_STRIP_LEAN(false, false);
      return check_launch();
    }
    if (prefetch && BITS == 3) {
      if (batch == 2)
        gptq_strip_kernel<BITS, 2, 32, 64, false, true><<<grid, 256, 0, st>>>(qweight, x, scales, zeros, SBQ_STRIP_LEAD(g), out, part, arrivals, g, GptqMulti{});
      else
        gptq_strip_kernel<BITS, 1, 32, 64, false, true><<<grid, 256, 0, st>>>(qweight, x, scales, zeros, SBQ_STRIP_LEAD(g), out, part, arrivals, g, GptqMulti{});
      return check_launch();
    }
    if (ch == kSliceK) SBQ_STRIP(128, false);
    else SBQ_STRIP(64, false);
#undef SBQ_STRIP
#undef SBQ_STRIP_LEAN
    return check_launch();
  }
  if constexpr (BITS == 2) {
    if (half_slices) return gptq_launch_partial<2, 64>(x, qweight, out, scales, zeros, part, g, vec, st);
  }
  return gptq_launch_partial<BITS, kSliceK>(x, qweight, out, scales, zeros, part, g, vec, st);
}

// Several matrices, one activation vector, ONE launch (see GptqMulti).  Falls back to one launch per matrix -- same
// results -- for anything the strip kernel does not take.
template <int BITS>
int gptq_matmul_multi(const float* x, int n_mats, const int32_t* const* qweights, float* const* outs,
                      const float* const* scales, const float* const* zeros, const int64_t* out_features, int64_t batch,
                      int64_t in_features, int64_t group_size, void* workspace, size_t workspace_bytes, void* stream) {
  if (n_mats < 1 || n_mats > kMaxMulti) return SBQ_ERR_ARG;
  if (!x || !qweights || !outs || !scales || !zeros || !out_features || !workspace) return SBQ_ERR_NULL;
  int64_t total_out = 0;
  // (the multi-matrix launch exists for the LEAN kernels only: x 16-byte aligned, in_features % 4 == 0, every
  // tensor below 4 GB; anything else takes one launch per matrix)
  bool strip_ok = batch >= 1 && batch <= 2 && group_size != 0 && group_size % kSliceK == 0 && knob(2) != 9 &&
                  aligned16(x) && in_features % 4 == 0 && knob(2) != 23;
  for (int m = 0; m < n_mats; ++m) {
    if (!qweights[m] || !outs[m] || !scales[m] || !zeros[m]) return SBQ_ERR_NULL;
    if (out_features[m] <= 0) return SBQ_ERR_ARG;
    total_out += out_features[m];
    strip_ok = strip_ok && out_features[m] % kStripCols == 0 && aligned16(qweights[m]) &&
               lean_sizes(gptq_rows(BITS, in_features), batch, in_features, out_features[m],
                          group_size > 0 ? ceil_div(in_features, group_size) : 1);
  }
  const int64_t strips = total_out / kStripCols;
  strip_ok = strip_ok && strips <= kMaxStrips && in_features > 0 && in_features < (1ll << 31);
  if (!strip_ok || n_mats == 1) {
    for (int m = 0; m < n_mats; ++m) {
      int rc = gptq_matmul<BITS>(x, qweights[m], outs[m], scales[m], zeros[m], batch, in_features, out_features[m], group_size,
                                 workspace, workspace_bytes, stream);
      if (rc != SBQ_OK) return rc;
    }
    return SBQ_OK;
  }
  constexpr int kMinGroup = BITS == 2 ? 64 : 128;
  if (group_size % kMinGroup != 0) return SBQ_ERR_ARG;
  GptqGeom g;
  if (!gptq_geom(BITS, kSliceK, batch, in_features, total_out, group_size, g)) return SBQ_ERR_ARG;
  const size_t need = kCounterBytes + static_cast<size_t>(kStripMaxSplit) * batch * total_out * sizeof(float);
  if (workspace_bytes < need || !aligned16(workspace)) return SBQ_ERR_WORKSPACE;
  if (reinterpret_cast<uintptr_t>(x) & 3u) return SBQ_ERR_ALIGN;
  hipStream_t st = as_stream(stream);
  {
    const int rc = workspace_guard(workspace, kCounterBytes, st);
    if (rc != SBQ_OK) return rc;
  }
  uint32_t* arrivals = static_cast<uint32_t*>(workspace);
  float* part = reinterpret_cast<float*>(static_cast<char*>(workspace) + kCounterBytes);
  constexpr int ch = kSliceK / 2;
  const int64_t passes = ceil_div(in_features, 32 * ch);
  int64_t split = passes >= 3 ? ceil_div(passes, 3) : 1;
  while (strips * split * 2 < static_cast<int64_t>(cu_count()) && split < passes) ++split;
  if (knob(1) > 0 && knob(1) < 128) split = knob(1);
  if (split > passes) split = passes;
  if (split > kStripMaxSplit) split = kStripMaxSplit;
  GptqMulti mm{};
  mm.n = n_mats;
  int64_t sb = 0, po = 0;
  for (int m = 0; m < n_mats; ++m) {
    mm.strip_begin[m] = static_cast<int32_t>(sb);
    mm.qw[m] = qweights[m];
    mm.scales[m] = scales[m];
    mm.zeros[m] = zeros[m];
    mm.out[m] = outs[m];
    mm.out_features[m] = out_features[m];
    mm.part_off[m] = po;
    sb += out_features[m] / kStripCols;
    po += split * batch * out_features[m];
  }
  mm.strip_begin[n_mats] = static_cast<int32_t>(sb);
  for (int m = n_mats + 1; m <= kMaxMulti; ++m) mm.strip_begin[m] = INT32_MAX;  // the kernel's lookup compares, no loop
  g.grid_x = static_cast<int32_t>(strips);
  g.grid_y = static_cast<int32_t>(split);
  g.xcd_swizzle = (strips % 8 == 0 && knob(2) != 8) ? 1 : 0;
  constexpr bool kDec8 = BITS != 3;
  {  // the wide workgroups, by the rule of the single-matrix launch (same kernel, same K split: same bits)
    const int64_t wpasses = ceil_div(in_features, 64 * ch);
    bool wide = wpasses == 1 ? (BITS != 2 || batch == 1) : (BITS != 4 && batch == 1);
    if (knob(2) == 24) wide = true;
    if (knob(2) == 25) wide = false;
    if (wide) {
      int64_t wsplit = wpasses >= 3 ? wpasses : 1;
      if (knob(1) > 0 && knob(1) < 128) wsplit = knob(1);
      if (wsplit > wpasses) wsplit = wpasses;
      if (wsplit > kStripMaxSplit) wsplit = kStripMaxSplit;
      po = 0;
      for (int m = 0; m < n_mats; ++m) {
        mm.part_off[m] = po;
        po += wsplit * batch * out_features[m];
      }
      g.grid_y = static_cast<int32_t>(wsplit);
      const dim3 wgrid(static_cast<uint32_t>(strips), static_cast<uint32_t>(wsplit));
      if (batch == 2)
        gptq_strip_kernel<BITS, 2, 64, 64, kDec8, false, true, true><<<wgrid, 512, 0, st>>>(nullptr, x, nullptr, nullptr, SBQ_STRIP_LEAD(g), nullptr, part, arrivals, g, mm);
      else
        gptq_strip_kernel<BITS, 1, 64, 64, kDec8, false, true, true><<<wgrid, 512, 0, st>>>(nullptr, x, nullptr, nullptr, SBQ_STRIP_LEAD(g), nullptr, part, arrivals, g, mm);
      return check_launch();
    }
  }
  const dim3 grid(static_cast<uint32_t>(strips), static_cast<uint32_t>(split));
  const bool prefetch = passes >= 2 * split && knob(2) != 6;
#define SBQ_MULTI_LEAN(BT, PFV) \
  gptq_strip_kernel<BITS, BT, 32, 64, kDec8, PFV, true, true><<<grid, 256, 0, st>>>(nullptr, x, nullptr, nullptr, SBQ_STRIP_LEAD(g), nullptr, part, arrivals, g, mm)
  if (prefetch) {
    if (batch == 2) SBQ_MULTI_LEAN(2, true);
    else SBQ_MULTI_LEAN(1, true);
  } else {
    if (batch == 2) SBQ_MULTI_LEAN(2, false);
    else SBQ_MULTI_LEAN(1, false);
  }
#undef SBQ_MULTI_LEAN
  return check_launch();
}

}  // namespace
}  // namespace sbq

extern "C" {
#if SBQ_GPTQ_STAMPS != 0
__attribute__((visibility("default"))) int sbq_debug_gptq_stamps(void* buffer) {  // development build only: where gptq_strip_kernel writes its stamps
  unsigned long long* p = static_cast<unsigned long long*>(buffer);
  return hipMemcpyToSymbol(HIP_SYMBOL(sbq::g_gptq_stamps), &p, sizeof(p)) == hipSuccess ? SBQ_OK : SBQ_ERR_ARG;
}
#endif

int sbq_vecquantmatmul_multi(int bits, const float* x, int n_mats, const int32_t* const* qweights, float* const* outs,
                             const float* const* scales, const float* const* zeros, const int64_t* out_features,
                             int64_t batch, int64_t in_features, int64_t group_size, void* workspace,
                             size_t workspace_bytes, void* stream) {
  if (bits == 4)
    return sbq::gptq_matmul_multi<4>(x, n_mats, qweights, outs, scales, zeros, out_features, batch, in_features, group_size,
                                     workspace, workspace_bytes, stream);
  if (bits == 3)
    return sbq::gptq_matmul_multi<3>(x, n_mats, qweights, outs, scales, zeros, out_features, batch, in_features, group_size,
                                     workspace, workspace_bytes, stream);
  if (bits == 2)
    return sbq::gptq_matmul_multi<2>(x, n_mats, qweights, outs, scales, zeros, out_features, batch, in_features, group_size,
                                     workspace, workspace_bytes, stream);
  return SBQ_ERR_ARG;
}

size_t sbq_gptq_workspace_bytes(int64_t batch, int64_t in_features, int64_t out_features) {
  using namespace sbq;
  if (batch <= 0 || in_features <= 0 || out_features <= 0) return 0;
  if (in_features >= (1ll << 31) || out_features >= (1ll << 31)) return 0;
  int64_t tiles = gptq_max_kblocks(in_features, out_features);
  if (tiles < kStripMaxSplit) tiles = kStripMaxSplit;
  return kCounterBytes + static_cast<size_t>(tiles) * batch * out_features * sizeof(float);
}

// What sbq_vecquantmatmul_multi may need: the joint launch sizes its K-split partials by the summed width; the
// per-matrix route (a width that is not a multiple of 32, one matrix, no groups) by each matrix's own width with a
// floor on the block count -- a narrow matrix of a very deep layer can need more than the sum suggests.
size_t sbq_vecquantmatmul_multi_workspace_bytes(int64_t batch, int64_t in_features, int n_mats, const int64_t* out_features) {
  if (n_mats <= 0 || !out_features) return 0;
  int64_t total = 0;
  size_t need = 0;
  for (int m = 0; m < n_mats; ++m) {
    if (out_features[m] <= 0) return 0;
    total += out_features[m];
    const size_t one = sbq_gptq_workspace_bytes(batch, in_features, out_features[m]);
    need = one > need ? one : need;
  }
  const size_t all = sbq_gptq_workspace_bytes(batch, in_features, total);
  return all > need ? all : need;
}

int sbq_vecquant4matmul(const float* x, const int32_t* qweight, float* out, const float* scales,
                        const float* zeros, int64_t batch, int64_t in_features, int64_t out_features,
                        int64_t group_size, void* workspace, size_t workspace_bytes, void* stream) {
  return sbq::gptq_matmul<4>(x, qweight, out, scales, zeros, batch, in_features, out_features, group_size,
                             workspace, workspace_bytes, stream);
}

int sbq_vecquant3matmul(const float* x, const int32_t* qweight, float* out, const float* scales,
                        const float* zeros, int64_t batch, int64_t in_features, int64_t out_features,
                        int64_t group_size, void* workspace, size_t workspace_bytes, void* stream) {
  return sbq::gptq_matmul<3>(x, qweight, out, scales, zeros, batch, in_features, out_features, group_size,
                             workspace, workspace_bytes, stream);
}

int sbq_vecquant2matmul(const float* x, const int32_t* qweight, float* out, const float* scales,
                        const float* zeros, int64_t batch, int64_t in_features, int64_t out_features,
                        int64_t group_size, void* workspace, size_t workspace_bytes, void* stream) {
  return sbq::gptq_matmul<2>(x, qweight, out, scales, zeros, batch, in_features, out_features, group_size,
                             workspace, workspace_bytes, stream);
}

}  // extern "C"
