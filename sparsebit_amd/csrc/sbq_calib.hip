// sbq_calib.hip -- model-wide calibration launches: the min-max (and MSE) observers + calc_qparams of EVERY weight of a
// model in two (four) launches.
//
// The reference calibrates weights layer by layer: CalibrationRunner.run_weight_calibration loops over the graph
// (sparsebit/quantization/tools/calibration.py:117-135), each layer's observer runs its own torch reductions
// (observers/minmax.py:14-25, observers/mse.py:28-63) and calc_qparams_with_minmax (observers/base.py:63-79).  On the
// device that is 3-5 launches per layer of ~5 us each for tensors of a few hundred kilobytes: ResNet-50's 53 weights
// are ~1 ms of launch latency for 100 MB of reads.  Here the tensors of a model -- any mix of [C, inner] shapes, per
// channel or per tensor, each with its own integer range and scheme -- are described once in a device table
// (sbq_calib_table_build) and
//   sbq_group_minmax_qparams   statistics of every row of every tensor by ONE grid (a wave per <= 4096-element
//                              segment of a row, the per-tensor kernels' reductions: sbq_observe_body.hpp), then one
//                              small launch that folds a row's segments and applies calc_qparams_with_minmax;
//   sbq_group_mse_qparams      the 80-candidate search of every row by one grid (a workgroup per 4096-element chunk:
//                              mse_chunk_body, the per-tensor kernel's body), then one launch that folds a row's
//                              chunks in the per-tensor path's order and keeps the first strictly better candidate.
// Same device code and the same summation order as the per-tensor entry points => bit-identical results
// (tests/test_gpu_r03.py::test_group_calibration_equals_per_tensor).
#include "sbq_observe_body.hpp"

namespace sbq {
namespace {

struct CalibItemDev {  // 64 bytes, read through the scalar unit
  const void* x;
  uint64_t out_off;        // floats from the output bases
  uint32_t C;
  uint32_t inner;          // elements per row (< 2^32: checked)
  uint32_t segs_per_row;   // statistics: ceil(inner / 4096)
  uint32_t seg_wg_begin;   // first statistics workgroup of the item (its segments are padded to whole workgroups)
  uint32_t seg_begin;      // first statistics partial of the item
  uint32_t row_begin;      // first row of the item in the model-wide row numbering
  uint32_t chunk_begin;    // first MSE chunk (== workgroup) of the item
  float qlo, qhi;
  uint32_t flags;
  uint32_t part_begin;     // first MSE partial record of the item
  uint32_t mse_waves;      // 0: a workgroup per 4096-element chunk; 1 / 2 / 4: a WAVE per row of <= 512 / 1024 / 2048
                           // elements, holding that many of the per-tensor kernel's waves' shares (round 3; knob 2 == 37);
                           // 8: a LANE per (row, candidate), four rows per workgroup (round 6: calib_mse_lanes_kernel)
};
static_assert(sizeof(CalibItemDev) == 64, "device table layout");

struct CalibHeader {  // 64 bytes
  uint32_t n_items, n_seg_wgs, n_rows, n_chunks, n_segs, max_chunks_per_row, n_parts;
  uint32_t n_lane_chunks;  // the first n_lane_chunks MSE chunks are workgroups of calib_mse_lanes_kernel (round 6)
  uint32_t pad[8];
};
static_assert(sizeof(CalibHeader) == 64, "device table layout");

template <typename V>
__device__ __forceinline__ V uread(const V* p) {  // uniform (scalar) load
  typedef const V __attribute__((address_space(4))) * cptr;
  return *reinterpret_cast<cptr>(reinterpret_cast<uintptr_t>(p));
}

__device__ __forceinline__ uint32_t udiv24(uint32_t a, uint32_t b) {  // a < 2^24: exact through one fp32 estimate
  uint32_t r = static_cast<uint32_t>(static_cast<float>(a) * (1.0f / static_cast<float>(b)));
  const int32_t rem = static_cast<int32_t>(a - r * b);
  if (rem < 0) --r;
  else if (rem >= static_cast<int32_t>(b)) ++r;
  return r;
}

// ---- statistics: a wave per segment ---------------------------------------------------------------------
// (the reductions of stats_minmax_kernel: three packed integer operations per dword of a 16-bit tensor,
// v_minimum3 / v_maximum3 for fp32 -- NaN-propagating like torch.min / max)
template <typename T>
__global__ __launch_bounds__(kBlock) void calib_stats_kernel(const CalibItemDev* __restrict__ items,
                                                             const uint32_t* __restrict__ wg_item,
                                                             StatPartial* __restrict__ part) {
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t idx = uread(wg_item + blockIdx.x);
  const CalibItemDev* it = items + idx;
  const uint32_t spr = uread(&it->segs_per_row), C = uread(&it->C), inner = uread(&it->inner);
  const uint32_t seg = (blockIdx.x - uread(&it->seg_wg_begin)) * kWavesPerBlock +
                       __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  if (seg >= C * spr) return;  // padding wave of the item's last workgroup
  const uint32_t row = spr == 1 ? seg : udiv24(seg, spr);
  const uint32_t k = seg - row * spr;
  const uint32_t begin = k * kStatsChunk;
  const uint32_t len = inner - begin < kStatsChunk ? inner - begin : kStatsChunk;  // a multiple of 8
  const int64_t first = static_cast<int64_t>(row) * inner + begin;
  const void* x = uread(&it->x);
  constexpr int U = 8;
  float mn, mx;
  if constexpr (T::id == SBQ_F32) {
    mn = __builtin_inff();
    mx = -__builtin_inff();
    // rounds past the end of a short row (a 1 x 1 convolution's 64 ... 2048 elements) are skipped: a model's rows are
    // mostly shorter than a segment, and eight clamped loads per lane would re-read each one's last pack 64-fold
    const int n_u = static_cast<int>((len + kWave * kPack - 1) / (kWave * kPack));  // wave-uniform
    RawPack<T> raw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (u >= n_u) break;
      const uint32_t eA = u * kWave * kPack + 4 * lane, eB = eA + kWave * kPack / 2;
      raw[u] = load_raw2<T, true>(x, first + (eA < len ? eA : len - 4), first + (eB < len ? eB : len - 4));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (u >= n_u) break;
      float v[kPack];
      unpack_raw<T>(raw[u], v);
#pragma unroll
      for (int q = 0; q < kPack; q += 2) {
        mn = __builtin_elementwise_minimum(__builtin_elementwise_minimum(mn, v[q]), v[q + 1]);
        mx = __builtin_elementwise_maximum(__builtin_elementwise_maximum(mx, v[q]), v[q + 1]);
      }
    }
    mn = wave_reduce(mn, [](float a, float b) { return __builtin_elementwise_minimum(a, b); });
    mx = wave_reduce(mx, [](float a, float b) { return __builtin_elementwise_maximum(a, b); });
  } else {
    Stat16 s = kStat16Identity;
    const int n_u = static_cast<int>((len + kWave * kPack - 1) / (kWave * kPack));  // wave-uniform
    RawPack<T> raw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (u >= n_u) break;
      uint32_t e = (u * kWave + lane) * kPack;
      if (e >= len) e = len - kPack;
      raw[u] = load_raw<T, true>(x, first + e);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (u >= n_u) break;
#pragma unroll
      for (int q = 0; q < 4; ++q) stat16_fold(s, raw[u].d[0][q]);
    }
    s = stat16_wave(s);
    stat16_decode<T>(s, mn, mx);
  }
  if (lane == 0) part[uread(&it->seg_begin) + seg] = StatPartial{mn, mx, 0.0};
}

__device__ __forceinline__ const CalibItemDev& item_of_row(const CalibItemDev* __restrict__ items, uint32_t n_items,
                                                           uint32_t r) {
  uint32_t lo = 0, hi = n_items - 1;  // last item with row_begin <= r
  while (lo < hi) {
    const uint32_t mid = (lo + hi + 1) >> 1;
    if (items[mid].row_begin <= r) lo = mid;
    else hi = mid - 1;
  }
  return items[lo];
}

// a thread per row: fold the row's segments (torch.min / max semantics: NaN wins), then observers/base.py:63-79
__global__ __launch_bounds__(kBlock) void calib_stats_fold_kernel(const CalibItemDev* __restrict__ items, uint32_t n_items,
                                                                  uint32_t n_rows, const StatPartial* __restrict__ part,
                                                                  float* __restrict__ min_base, float* __restrict__ max_base,
                                                                  float* __restrict__ scale_base, float* __restrict__ zp_base) {
  const uint32_t r = blockIdx.x * kBlock + threadIdx.x;
  if (r >= n_rows) return;
  const CalibItemDev& it = item_of_row(items, n_items, r);
  const uint32_t row = r - it.row_begin;
  const StatPartial* p = part + it.seg_begin + static_cast<size_t>(row) * it.segs_per_row;
  NanMin nmin;
  NanMax nmax;
  float mn = p[0].mn, mx = p[0].mx;
  for (uint32_t k = 1; k < it.segs_per_row; ++k) {
    mn = nmin(mn, p[k].mn);
    mx = nmax(mx, p[k].mx);
  }
  min_base[it.out_off + row] = mn;
  max_base[it.out_off + row] = mx;
  if (scale_base) {
    float s, z;
    qparams_from_minmax(mn, mx, it.qhi - it.qlo, (it.flags & SBQ_CALIB_SYMMETRIC) != 0, s, z);
    scale_base[it.out_off + row] = s;
    zp_base[it.out_off + row] = z;
  }
}

// ---- MSE ------------------------------------------------------------------------------------------------
// A model's rows are mostly SHORT (a 1 x 1 convolution: 64 ... 2048 elements): a workgroup per row would walk the 80
// candidates over 4096 register slots of which a few per cent hold data (967 us for ResNet-50's 25.5 M elements,
// four times the per-tensor kernel's rate).  Rows of at most 2048 elements are therefore taken by ONE WAVE each, four
// rows per workgroup.  To stay bit-identical with the per-tensor kernel the wave reproduces its summation tree: there
// pack p of a row sits in lane p % 64 of wave (p % 256) / 64, each wave reduces its lanes' fp32 sums by the
// wave sum (wave_sum_f32) and the four wave sums are added in fp64 -- here lane l holds packs l, 64 + l, 128 + l, 192 + l with one
// accumulator per "virtual wave", reduces each by the same wave sum (wave_sum_f32) and adds them in the same order.
template <typename T, int NV>
__device__ __forceinline__ void mse_wave_rows(MseLds (&lds)[kWavesPerBlock], const CalibItemDev* it, uint32_t row,
                                              const float* __restrict__ min_base, const float* __restrict__ max_base,
                                              double* __restrict__ part) {
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  MseLds& L = lds[wid];
  const uint32_t inner = uread(&it->inner), C = uread(&it->C);
  const bool live = row < C;  // wave-uniform
  const uint64_t off = uread(&it->out_off) + (live ? row : 0);
  const float qlo = uread(&it->qlo), qhi = uread(&it->qhi);
  const bool symmetric = (uread(&it->flags) & SBQ_CALIB_SYMMETRIC) != 0;
  const float mn = min_base[off], mx = max_base[off];
  for (int i = lane; i < SBQ_MSE_CANDIDATES; i += kWave) {
    float s, z;
    mse_candidate(mn, mx, i, qhi - qlo, symmetric, s, z);
    L.scale[i] = s;
    L.zp[i] = z;
    L.rcp[i] = fast_div_ok(s) ? 1.0f / s : 0.0f;
  }
  float v[NV][kPack];
  const void* x = uread(&it->x);
  const int64_t row_base = static_cast<int64_t>(live ? row : 0) * inner;
#pragma unroll
  for (int w = 0; w < NV; ++w) {
    const uint32_t e = (w * kWave + lane) * kPack;
    const bool in = e + kPack <= inner;
    float t[kPack];
    load_pack<T, true>(x, row_base + (in ? e : 0), t);
#pragma unroll
    for (int q = 0; q < kPack; ++q) v[w][q] = in ? t[q] : 0.0f;
  }
  // (the candidates were written and are read by this wave alone: LDS operations of a wave execute in order)
  for (int i = 0; i < SBQ_MSE_CANDIDATES; ++i) {
    const float s = L.scale[i], z = L.zp[i], y = L.rcp[i];
    float acc[NV];
#pragma unroll
    for (int w = 0; w < NV; ++w) {
      float a = 0.0f;
      if (y != 0.0f) {
        if (z == 0.0f) {
#pragma unroll
          for (int q = 0; q < kPack; ++q) {
            const float lv = __builtin_amdgcn_fmed3f(__builtin_rintf(v[w][q] * y), qlo, qhi);
            const float d = __builtin_fmaf(-lv, s, v[w][q]);
            a = __builtin_fmaf(d, d, a);
          }
        } else {
#pragma unroll
          for (int q = 0; q < kPack; ++q) {
            const float lv = __builtin_amdgcn_fmed3f(__builtin_rintf(v[w][q] * y) + z, qlo, qhi);
            const float d = __builtin_fmaf(-(lv - z), s, v[w][q]);
            a = __builtin_fmaf(d, d, a);
          }
        }
      } else {
#pragma unroll
        for (int q = 0; q < kPack; ++q) {
          const float lv = quant_level<SBQ_ROUND_HALF_EVEN>(v[w][q], s, z, qlo, qhi);
          const float d = v[w][q] - dequant_level(lv, s, z);
          a += d * d;
        }
      }
      acc[w] = wave_sum_f32(a);
    }
    if (lane == 0 && live) {
      double t = 0.0;
#pragma unroll
      for (int w = 0; w < kWavesPerBlock; ++w) t += w < NV ? static_cast<double>(acc[w < NV ? w : 0]) : 0.0;
      part[(static_cast<size_t>(uread(&it->part_begin)) + row) * SBQ_MSE_CANDIDATES + i] = t;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void calib_mse_kernel(const CalibItemDev* __restrict__ items,
                                                           const uint32_t* __restrict__ chunk_item,
                                                           const float* __restrict__ min_base,
                                                           const float* __restrict__ max_base, double* __restrict__ part,
                                                           const uint32_t first_chunk) {
  __shared__ MseLds lds_w[kWavesPerBlock];
  MseLds& lds = lds_w[0];
  const uint32_t bid = blockIdx.x + first_chunk;  // (the table's chunk list starts with the lane kernel's workgroups)
  const uint32_t idx = uread(chunk_item + bid);
  const CalibItemDev* it = items + idx;
  const uint32_t mode = uread(&it->mse_waves);
  if (mode != 0) {  // workgroup-uniform: a wave per row
    const uint32_t row = (bid - uread(&it->chunk_begin)) * kWavesPerBlock + threadIdx.x / kWave;
    if (mode == 1) mse_wave_rows<T, 1>(lds_w, it, row, min_base, max_base, part);
    else if (mode == 2) mse_wave_rows<T, 2>(lds_w, it, row, min_base, max_base, part);
    else mse_wave_rows<T, 4>(lds_w, it, row, min_base, max_base, part);
    return;
  }
  const uint32_t inner = uread(&it->inner);
  const uint32_t cpr = (inner + kMseChunk - 1) / kMseChunk;
  const uint32_t ch = bid - uread(&it->chunk_begin);
  const uint32_t row = cpr == 1 ? ch : udiv24(ch, cpr);
  const uint32_t k = ch - row * cpr;
  const int64_t begin = static_cast<int64_t>(k) * kMseChunk;
  const int64_t end = begin + kMseChunk < inner ? begin + kMseChunk : inner;
  const uint64_t off = uread(&it->out_off) + row;
  const float qlo = uread(&it->qlo), qhi = uread(&it->qhi);
  const double t = mse_chunk_body<T, true>(lds, uread(&it->x), static_cast<int64_t>(row) * inner, begin, end, min_base[off],
                                           max_base[off], qhi - qlo, qlo, qhi, (uread(&it->flags) & SBQ_CALIB_SYMMETRIC) != 0);
  if (threadIdx.x < SBQ_MSE_CANDIDATES)
    part[(static_cast<size_t>(uread(&it->part_begin)) + ch) * SBQ_MSE_CANDIDATES + threadIdx.x] = t;
}

// ---- round 6: a LANE per (row, candidate) ----------------------------------------------------------------------
// The wave-per-row form above spends 9.8 wave instructions per 64 evaluations (the per-tensor kernel: 5.8): a wave
// reduction, an exec-masked fp64 store, three LDS reads and two branches PER CANDIDATE for 8 ... 32 evaluations per lane,
// and a 64-element row still occupies a whole wave.  Here the roles are turned round: four rows per workgroup of 320
// threads, thread t = (row t / 80, candidate t % 80) -- five full waves, no idle lane whatever the row length -- and
// every thread walks ITS candidate over ITS row's elements, which the workgroup stages in LDS tile by tile (a wave reads
// at most two rows: the reads are broadcasts).  Per four elements: one ds_read_b128 and 4 x (mul, rndne, med3, fma,
// fma) -- 5.25 instructions per evaluation, no cross-lane traffic at all, nothing per candidate.  The sum of squares
// runs in four interleaved fp32 accumulators per tile of 1024 elements and in fp64 across tiles: a different (and
// tighter) summation tree than the per-tensor kernel's, so the two may name neighbouring candidates where their losses
// tie to fp32 rounding -- the gate is the argmin index with such ties allowed (tests/test_gpu_r06.py), as SURVEY 7
// prescribes, not bit-identity of the sums.
constexpr int kLaneRows = 4;
constexpr int kLaneBlock = kLaneRows * SBQ_MSE_CANDIDATES;  // 320
constexpr uint32_t kLaneTile = 1024;                        // elements of each row staged at a time (16 KB)
constexpr uint32_t kLaneMaxInner = 16384;                   // longer rows (a per-tensor item): the chunk form above
static_assert(kLaneBlock % kWave == 0, "whole waves");
template <typename T>
__global__ __launch_bounds__(kLaneBlock) void calib_mse_lanes_kernel(const CalibItemDev* __restrict__ items,
                                                                    const uint32_t* __restrict__ chunk_item,
                                                                    const float* __restrict__ min_base,
                                                                    const float* __restrict__ max_base,
                                                                    float* __restrict__ scale_base, float* __restrict__ zp_base,
                                                                    int32_t* __restrict__ index_base) {
  __shared__ float s_loss[kLaneRows][SBQ_MSE_CANDIDATES];
  // (+ 4 floats per row: the rows a wave straddles sit in different banks -- 4 KB apart they would share every bank)
  __shared__ __attribute__((aligned(16))) float xs[kLaneRows][kLaneTile + 4];
  const uint32_t idx = uread(chunk_item + blockIdx.x);
  const CalibItemDev* it = items + idx;
  const uint32_t inner = uread(&it->inner), C = uread(&it->C);
  const uint32_t row0 = (blockIdx.x - uread(&it->chunk_begin)) * kLaneRows;
  const uint32_t r = threadIdx.x / SBQ_MSE_CANDIDATES, cand = threadIdx.x - r * SBQ_MSE_CANDIDATES;
  const uint32_t row = row0 + r;
  const bool live = row < C;
  const uint64_t off = uread(&it->out_off) + (live ? row : row0);
  const float qlo = uread(&it->qlo), qhi = uread(&it->qhi);
  float s, z;
  mse_candidate(min_base[off], max_base[off], static_cast<int>(cand), qhi - qlo, (uread(&it->flags) & SBQ_CALIB_SYMMETRIC) != 0, s, z);
  const bool fast = fast_div_ok(s);
  const float y = fast ? 1.0f / s : 0.0f;
  // every candidate of the workgroup's rows on the fast division with zero point 0 (any symmetric scheme on ordinary
  // data): the branch-free loop.  (mse_chunk_body's three forms of one element's error, operation for operation.)
  const bool plain = __syncthreads_and(fast && z == 0.0f) != 0;
  const void* x = uread(&it->x);
  const float* xr = xs[r];
  double total = 0.0;
  for (uint32_t t0 = 0; t0 < inner; t0 += kLaneTile) {  // (uniform)
    const uint32_t n = inner - t0 < kLaneTile ? inner - t0 : kLaneTile;  // a multiple of 8 (the host admits whole packs)
    const uint32_t ppr = n / kPack;                                      // packs per row in this tile
    for (uint32_t p = threadIdx.x; p < kLaneRows * ppr; p += kLaneBlock) {
      const uint32_t rr = p / ppr, pp = p - rr * ppr;
      const uint32_t src_row = row0 + rr < C ? row0 + rr : row0;  // (rows past the item's last: a copy of row0, never stored)
      float v[kPack];
      load_pack<T, true>(x, static_cast<int64_t>(src_row) * inner + t0 + pp * kPack, v);
      float4* dst = reinterpret_cast<float4*>(&xs[rr][pp * kPack]);
      dst[0] = float4{v[0], v[1], v[2], v[3]};
      dst[1] = float4{v[4], v[5], v[6], v[7]};
    }
    __syncthreads();
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    if (plain) {
#pragma unroll 2
      for (uint32_t j = 0; j < n; j += 4) {
        const float4 v = *reinterpret_cast<const float4*>(xr + j);
        const float l0 = __builtin_amdgcn_fmed3f(__builtin_rintf(v.x * y), qlo, qhi);
        const float l1 = __builtin_amdgcn_fmed3f(__builtin_rintf(v.y * y), qlo, qhi);
        const float l2 = __builtin_amdgcn_fmed3f(__builtin_rintf(v.z * y), qlo, qhi);
        const float l3 = __builtin_amdgcn_fmed3f(__builtin_rintf(v.w * y), qlo, qhi);
        const float d0 = __builtin_fmaf(-l0, s, v.x), d1 = __builtin_fmaf(-l1, s, v.y);
        const float d2 = __builtin_fmaf(-l2, s, v.z), d3 = __builtin_fmaf(-l3, s, v.w);
        a0 = __builtin_fmaf(d0, d0, a0);
        a1 = __builtin_fmaf(d1, d1, a1);
        a2 = __builtin_fmaf(d2, d2, a2);
        a3 = __builtin_fmaf(d3, d3, a3);
      }
    } else {
      for (uint32_t j = 0; j < n; j += 4) {
        const float4 v4 = *reinterpret_cast<const float4*>(xr + j);
        const float v[4] = {v4.x, v4.y, v4.z, v4.w};
        float e[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (fast) {
            if (z == 0.0f) {
              const float lv = __builtin_amdgcn_fmed3f(__builtin_rintf(v[q] * y), qlo, qhi);
              e[q] = __builtin_fmaf(-lv, s, v[q]);
            } else {
              const float lv = __builtin_amdgcn_fmed3f(__builtin_rintf(v[q] * y) + z, qlo, qhi);
              e[q] = __builtin_fmaf(-(lv - z), s, v[q]);
            }
          } else {
            const float lv = quant_level<SBQ_ROUND_HALF_EVEN>(v[q], s, z, qlo, qhi);
            e[q] = v[q] - dequant_level(lv, s, z);
          }
        }
        a0 = __builtin_fmaf(e[0], e[0], a0);
        a1 = __builtin_fmaf(e[1], e[1], a1);
        a2 = __builtin_fmaf(e[2], e[2], a2);
        a3 = __builtin_fmaf(e[3], e[3], a3);
      }
    }
    total += static_cast<double>((a0 + a1) + (a2 + a3));
    __syncthreads();  // (the tile is read; the next one may be staged)
  }
  // The pick, in the same launch (round 6: a select launch of its own spent 25 us walking 80 losses per row with one
  // thread, and the table of partial sums went out to memory and came back): mse.py:51-61 keeps the first candidate
  // whose fp32 loss is strictly smaller than 1e10 and than every earlier one -- the smallest loss, the lowest index
  // among equals; NaN never wins.  Wave w < 4 takes row w: lane l holds candidates l and 64 + l.
  s_loss[r][cand] = static_cast<float>(total / static_cast<double>(inner));
  __syncthreads();
  const uint32_t wv = threadIdx.x / kWave, ln = threadIdx.x & (kWave - 1);
  if (wv < static_cast<uint32_t>(kLaneRows) && row0 + wv < C) {  // (wave-uniform)
    auto key = [](float loss) {  // order-preserving for losses >= 0; everything that cannot win (NaN, >= 1e10) last
      return loss < 1e10f ? __builtin_bit_cast(uint32_t, loss + 0.0f) : 0xffffffffu;
    };
    const uint32_t ka = key(s_loss[wv][ln]);
    const uint32_t kb = ln + kWave < static_cast<uint32_t>(SBQ_MSE_CANDIDATES) ? key(s_loss[wv][(ln + kWave) % SBQ_MSE_CANDIDATES]) : 0xffffffffu;
    const uint32_t mine = kb < ka ? kb : ka;
    const uint32_t mine_i = kb < ka ? ln + kWave : ln;
    const uint32_t best_k = dpp_reduce_u32(mine, 0xffffffffu, [](uint32_t p, uint32_t q) { return p < q ? p : q; });
    const uint32_t best_i = dpp_reduce_u32(mine == best_k ? mine_i : 0xffffffffu, 0xffffffffu, [](uint32_t p, uint32_t q) { return p < q ? p : q; });
    if (ln == 0) {
      const uint64_t o = uread(&it->out_off) + row0 + wv;
      const int best = best_k == 0xffffffffu ? -1 : static_cast<int>(best_i);
      float bs = 1.0f, bz = 0.0f;  // mse.py:34-39 initial values
      if (best >= 0) mse_candidate(min_base[o], max_base[o], best, qhi - qlo, (uread(&it->flags) & SBQ_CALIB_SYMMETRIC) != 0, bs, bz);
      scale_base[o] = bs;
      zp_base[o] = bz;
      if (index_base) index_base[o] = best;
    }
  }
}

// Three rows per workgroup, a thread per (row, candidate): the row's chunks are summed in the per-tensor path's
// order -- a single chunk is the sum itself (sse += t onto zero), several go through mse_fold_kernel's three
// interleaved running sums ((s0 + s1) + s2, chunk j in sum j % 3) -- then mse_select_kernel's rule: the first
// candidate whose fp32 loss is strictly smaller (observers/mse.py:51-61).
constexpr int kRowsPerFoldWg = kBlock / SBQ_MSE_CANDIDATES;  // 3
constexpr uint32_t kMaxGroupMseChunks = 96;  // one level of the per-tensor fold (kFoldFan)

__global__ __launch_bounds__(kBlock) void calib_mse_select_kernel(const CalibItemDev* __restrict__ items, uint32_t n_items,
                                                                  uint32_t n_rows, const double* __restrict__ part,
                                                                  const float* __restrict__ min_base,
                                                                  const float* __restrict__ max_base,
                                                                  float* __restrict__ scale_base, float* __restrict__ zp_base,
                                                                  int32_t* __restrict__ index_base) {
  __shared__ float s_loss[kRowsPerFoldWg][SBQ_MSE_CANDIDATES];
  const uint32_t sub = threadIdx.x / SBQ_MSE_CANDIDATES, i = threadIdx.x % SBQ_MSE_CANDIDATES;
  const uint32_t r = blockIdx.x * kRowsPerFoldWg + sub;
  bool live = sub < static_cast<uint32_t>(kRowsPerFoldWg) && r < n_rows;
  const CalibItemDev* it = nullptr;
  uint32_t row = 0;
  if (live) {
    it = &item_of_row(items, n_items, r);
    row = r - it->row_begin;
  }
  if (live && it->mse_waves == 8u) live = false;  // (picked inside calib_mse_lanes_kernel; uniform per (row, sub): no barrier depends on it)
  if (live) {
    const uint32_t cpr = it->mse_waves ? 1u : (it->inner + kMseChunk - 1) / kMseChunk;
    const double* p = part + (static_cast<size_t>(it->part_begin) + static_cast<size_t>(row) * cpr) * SBQ_MSE_CANDIDATES + i;
    double t;
    if (cpr == 1) {
      t = 0.0 + p[0];
    } else {
      double s3[3] = {0.0, 0.0, 0.0};
      for (uint32_t j = 0; j < cpr; ++j) s3[j % 3] += p[static_cast<size_t>(j) * SBQ_MSE_CANDIDATES];
      t = 0.0 + ((s3[0] + s3[1]) + s3[2]);
    }
    s_loss[sub][i] = static_cast<float>(t / static_cast<double>(it->inner));
  }
  __syncthreads();
  if (live && i == 0) {
    float loss_min = 1e10f;
    int best = -1;
    for (int c = 0; c < SBQ_MSE_CANDIDATES; ++c) {
      const float loss = s_loss[sub][c];
      if (loss < loss_min) {
        loss_min = loss;
        best = c;
      }
    }
    const uint64_t off = it->out_off + row;
    float s = 1.0f, z = 0.0f;  // mse.py:34-39 initial values
    if (best >= 0)
      mse_candidate(min_base[off], max_base[off], best, it->qhi - it->qlo, (it->flags & SBQ_CALIB_SYMMETRIC) != 0, s, z);
    scale_base[off] = s;
    zp_base[off] = z;
    if (index_base) index_base[off] = best;
  }
}

// ---- GPTQ's grid search (large_language_models/llama/quantization/utils/quant.py:86-104) ----------------------
// find_params(mse=True): for every row (output channel, or one group of it) the shrink factor p = 1 - i / grid whose
// parameters give the smallest sum |quantize(x) - x|^norm; the first strictly smaller error wins.  The reference
// makes int(maxshrink * grid) = 80 passes of six tensor ops over the whole weight; here a wave keeps walking its
// row (L1 / L2 resident after the first pass) and only scale / zero / index leave the kernel.  fp32 operation for
// operation like the tensor ops (the Python scalar p is rounded to fp32 where torch multiplies a fp32 tensor by
// it); the error sums are fp32 lane partials + a wave reduction, where torch.sum has its own order -- candidates
// whose errors tie to the last bits may swap, which the tests allow for (tests/test_gpu_r03.py).
template <typename T>
__global__ __launch_bounds__(kBlock) void gptq_mse_kernel(const void* __restrict__ x, int64_t rows, int64_t inner,
                                                          const float* __restrict__ xmin_v, const float* __restrict__ xmax_v,
                                                          float maxq, int symmetric, float zero_sym, float norm, int grid,
                                                          int n_cand, float* __restrict__ scale_io,
                                                          float* __restrict__ zero_io, int32_t* __restrict__ index_out) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t row = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + threadIdx.x / kWave;
  if (row >= rows) return;
  const float xmin = xmin_v[row], xmax = xmax_v[row];
  float best = __builtin_inff();
  float best_s = scale_io[row], best_z = zero_io[row];
  int best_i = -1;
  for (int i = 0; i < n_cand; ++i) {
    const float p = static_cast<float>(1.0 - static_cast<double>(i) / static_cast<double>(grid));
    const float xmin1 = p * xmin, xmax1 = p * xmax;
    const float scale1 = (xmax1 - xmin1) / maxq;
    const float zero1 = symmetric ? zero_sym : __builtin_rintf(-xmin1 / scale1);
    float err = 0.0f;
    for (int64_t e = lane; e < inner; e += kWave) {
      const float v = Elem<T>::load1(x, row * inner + e);
      float q = __builtin_rintf(v / scale1) + zero1;
      const float qc = __builtin_fminf(__builtin_fmaxf(q, 0.0f), maxq);  // torch.clamp(.., 0, maxq) ...
      q = (q != q) ? q : qc;                                               // ... which propagates NaN
      const float d = __builtin_fabsf(scale1 * (q - zero1) - v);
      err += __builtin_powf(d, norm);
    }
    err = wave_reduce(err, Sum());
    if (err < best) {
      best = err;
      best_s = scale1;
      best_z = zero1;
      best_i = i;
    }
  }
  if (lane == 0) {
    scale_io[row] = best_s;
    zero_io[row] = best_z;
    if (index_out) index_out[row] = best_i;
  }
}

// Long rows (find_params(perchannel=False) flattens the whole weight into ONE row: 16.7 M elements for a 4096 x 4096
// layer -- a single wave would walk it 80 times).  A row is cut into slices of kGptqSlice elements, one workgroup per
// slice: the slice sits in registers (16 elements per thread), every candidate's error over it is reduced to ONE fp64
// partial (fp32 lane sums over 16 elements, fp64 across lanes / waves), and a second kernel adds a row's partials in
// ascending slice order in fp64 and picks the first strictly smallest error: deterministic, and closer to the exact
// sums than torch's fp32 reduction (ties to the last bits may still swap, as in the wave-per-row kernel).
constexpr int kGptqSlice = kBlock * 16;
constexpr int64_t kGptqSplitFrom = 4 * kGptqSlice;  // rows longer than this take the split path
constexpr int kGptqMaxCand = 128;
template <typename T>
__global__ __launch_bounds__(kBlock) void gptq_mse_slice_kernel(const void* __restrict__ x, int64_t inner, uint32_t slices,
                                                                const float* __restrict__ xmin_v, const float* __restrict__ xmax_v,
                                                                float maxq, int symmetric, float zero_sym, float norm, int grid,
                                                                int n_cand, double* __restrict__ part) {
  __shared__ double s_wave[kWavesPerBlock];
  const int64_t row = blockIdx.x / slices;
  const uint32_t sl = blockIdx.x % slices;
  const float xmin = xmin_v[row], xmax = xmax_v[row];
  const int64_t begin = static_cast<int64_t>(sl) * kGptqSlice;
  float v[16];
  bool ok[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int64_t e = begin + j * kBlock + threadIdx.x;
    ok[j] = e < inner;
    v[j] = ok[j] ? Elem<T>::load1(x, row * inner + e) : 0.0f;
  }
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  double* out = part + (static_cast<size_t>(row) * slices + sl) * n_cand;
  for (int i = 0; i < n_cand; ++i) {
    const float p = static_cast<float>(1.0 - static_cast<double>(i) / static_cast<double>(grid));
    const float xmin1 = p * xmin, xmax1 = p * xmax;
    const float scale1 = (xmax1 - xmin1) / maxq;
    const float zero1 = symmetric ? zero_sym : __builtin_rintf(-xmin1 / scale1);
    float err = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      float q = __builtin_rintf(v[j] / scale1) + zero1;
      const float qc = __builtin_fminf(__builtin_fmaxf(q, 0.0f), maxq);
      q = (q != q) ? q : qc;
      const float d = __builtin_fabsf(scale1 * (q - zero1) - v[j]);
      err += ok[j] ? __builtin_powf(d, norm) : 0.0f;
    }
    double e64 = static_cast<double>(err);
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) e64 += __shfl_xor(e64, d, kWave);
    if (lane == 0) s_wave[wid] = e64;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0.0;
      for (int w = 0; w < kWavesPerBlock; ++w) t += s_wave[w];
      out[i] = t;
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(kGptqMaxCand) void gptq_mse_pick_kernel(const double* __restrict__ part, uint32_t slices,
                                                                     const float* __restrict__ xmin_v, const float* __restrict__ xmax_v,
                                                                     float maxq, int symmetric, float zero_sym, int grid, int n_cand,
                                                                     float* __restrict__ scale_io, float* __restrict__ zero_io,
                                                                     int32_t* __restrict__ index_out) {
  __shared__ float s_err[kGptqMaxCand];
  const int64_t row = blockIdx.x;
  const int i = threadIdx.x;
  if (i < n_cand) {
    const double* p = part + static_cast<size_t>(row) * slices * n_cand + i;
    double t = 0.0;
    for (uint32_t j = 0; j < slices; ++j) t += p[static_cast<size_t>(j) * n_cand];
    s_err[i] = static_cast<float>(t);
  }
  __syncthreads();
  if (i != 0) return;
  float best = __builtin_inff();
  int best_i = -1;
  for (int c = 0; c < n_cand; ++c) {
    if (s_err[c] < best) {
      best = s_err[c];
      best_i = c;
    }
  }
  if (best_i >= 0) {
    const float p = static_cast<float>(1.0 - static_cast<double>(best_i) / static_cast<double>(grid));
    const float xmin1 = p * xmin_v[row], xmax1 = p * xmax_v[row];
    const float scale1 = (xmax1 - xmin1) / maxq;
    scale_io[row] = scale1;
    zero_io[row] = symmetric ? zero_sym : __builtin_rintf(-xmin1 / scale1);
  }
  if (index_out) index_out[row] = best_i;
}

}  // namespace
}  // namespace sbq

extern "C" {

size_t sbq_gptq_mse_search_workspace_bytes(int64_t rows, int64_t inner, int n_candidates) {
  using namespace sbq;
  if (rows <= 0 || inner <= kGptqSplitFrom || n_candidates <= 0) return 0;  // the wave-per-row kernel needs none
  const int64_t slices = ceil_div(inner, static_cast<int64_t>(kGptqSlice));
  return static_cast<size_t>(rows) * slices * n_candidates * sizeof(double);
}

int sbq_gptq_mse_search(const void* x, int x_dtype, int64_t rows, int64_t inner, const float* xmin, const float* xmax,
                        int maxq, int symmetric, float norm, int grid, int n_candidates, float* scale_io,
                        float* zero_io, int32_t* index_out, void* workspace, size_t workspace_bytes, void* stream) {
  using namespace sbq;
  if (!valid_dtype(x_dtype)) return SBQ_ERR_DTYPE;
  if (rows < 0 || inner < 0) return SBQ_ERR_ARG;
  if (rows == 0 || inner == 0) return SBQ_ERR_EMPTY;
  if (!x || !xmin || !xmax || !scale_io || !zero_io) return SBQ_ERR_NULL;
  if (maxq < 1 || grid < 1 || n_candidates < 0 || rows >= (1ll << 33)) return SBQ_ERR_ARG;
  if (reinterpret_cast<uintptr_t>(x) % dtype_size(x_dtype)) return SBQ_ERR_ALIGN;
  hipStream_t st = as_stream(stream);
  const float zero_sym_ = static_cast<float>((maxq + 1) / 2.0);
  if (inner > kGptqSplitFrom && n_candidates > 0) {
    // long rows: slices across workgroups, fp64 partials, fixed-order fold (see gptq_mse_slice_kernel)
    if (n_candidates > kGptqMaxCand) return SBQ_ERR_ARG;
    const size_t need = sbq_gptq_mse_search_workspace_bytes(rows, inner, n_candidates);
    if (!workspace) return SBQ_ERR_NULL;
    if (workspace_bytes < need || reinterpret_cast<uintptr_t>(workspace) % 8) return SBQ_ERR_WORKSPACE;
    const int64_t slices = ceil_div(inner, static_cast<int64_t>(kGptqSlice));
    if (rows * slices >= (1ll << 31)) return SBQ_ERR_ARG;
    double* part = static_cast<double*>(workspace);
    int rc2 = dispatch_dtype(x_dtype, [&](auto tag) {
      using T = decltype(tag);
      gptq_mse_slice_kernel<T><<<static_cast<uint32_t>(rows * slices), kBlock, 0, st>>>(
          x, inner, static_cast<uint32_t>(slices), xmin, xmax, static_cast<float>(maxq), symmetric, zero_sym_, norm, grid,
          n_candidates, part);
    });
    if (rc2 != SBQ_OK) return rc2;
    gptq_mse_pick_kernel<<<static_cast<uint32_t>(rows), kGptqMaxCand, 0, st>>>(part, static_cast<uint32_t>(slices), xmin, xmax,
                                                                             static_cast<float>(maxq), symmetric, zero_sym_,
                                                                             grid, n_candidates, scale_io, zero_io, index_out);
    return check_launch();
  }
  const uint32_t gridx = static_cast<uint32_t>(ceil_div(rows, static_cast<int64_t>(kWavesPerBlock)));
  const float zero_sym = static_cast<float>((maxq + 1) / 2.0);
  int rc = dispatch_dtype(x_dtype, [&](auto tag) {
    using T = decltype(tag);
    gptq_mse_kernel<T><<<gridx, kBlock, 0, st>>>(x, rows, inner, xmin, xmax, static_cast<float>(maxq), symmetric, zero_sym,
                                               norm, grid, n_candidates, scale_io, zero_io, index_out);
  });
  if (rc != SBQ_OK) return rc;
  return check_launch();
}

// table = [header | items | statistics workgroup -> item | MSE chunk -> item]
int sbq_calib_table_build(const sbq_calib_item* items, int n_items, void* host_table, size_t host_table_bytes,
                          uint32_t* n_rows_out, size_t* bytes_needed_out, size_t* workspace_bytes_out) {
  using namespace sbq;
  if (n_items < 0) return SBQ_ERR_ARG;
  if (n_items == 0) return SBQ_ERR_EMPTY;
  if (!items) return SBQ_ERR_NULL;
  uint64_t seg_wgs = 0, segs = 0, rows = 0, chunks = 0, max_cpr = 0, parts = 0;
  // 8: a lane per (row, candidate) (round 6; rows of at most kLaneMaxInner elements); knob 2 == 37: round 3's forms --
  // 1 / 2 / 4 virtual waves of a wave-per-row item, 0: workgroup per chunk
  const bool lanes = knob(2) != 37;
  auto mse_mode = [lanes](int64_t inner) -> uint32_t {
    if (lanes && inner <= static_cast<int64_t>(kLaneMaxInner)) return 8u;
    return inner <= 512 ? 1u : (inner <= 1024 ? 2u : (inner <= 2048 ? 4u : 0u));
  };
  auto mse_chunks = [&](const sbq_calib_item& it, uint32_t mode) -> uint64_t {
    if (mode == 8u) return ceil_div(it.C, static_cast<int64_t>(kLaneRows));
    if (mode != 0u) return ceil_div(it.C, static_cast<int64_t>(kWavesPerBlock));
    return it.C * ceil_div(it.inner, static_cast<int64_t>(kMseChunk));
  };
  uint64_t lane_chunks = 0;
  for (int i = 0; i < n_items; ++i) {
    const sbq_calib_item& it = items[i];
    if (!it.x) return SBQ_ERR_NULL;
    if (it.C < 0 || it.inner < 0) return SBQ_ERR_ARG;
    if (it.C == 0 || it.inner == 0) return SBQ_ERR_EMPTY;
    if (it.qmin >= it.qmax || it.inner % kPack != 0 || it.inner >= (1ll << 32) || it.C >= (1ll << 24)) return SBQ_ERR_ARG;
    if (!aligned16(it.x)) return SBQ_ERR_ALIGN;
    const uint64_t spr = ceil_div(it.inner, static_cast<int64_t>(kStatsChunk));
    const uint64_t cpr = ceil_div(it.inner, static_cast<int64_t>(kMseChunk));
    if (static_cast<uint64_t>(it.C) * spr >= (1ull << 24) || static_cast<uint64_t>(it.C) * cpr >= (1ull << 24)) return SBQ_ERR_ARG;
    seg_wgs += ceil_div(static_cast<int64_t>(it.C * spr), static_cast<int64_t>(kWavesPerBlock));
    segs += it.C * spr;
    rows += it.C;
    const uint32_t mode = mse_mode(it.inner);
    chunks += mse_chunks(it, mode);
    if (mode == 8u) lane_chunks += mse_chunks(it, mode);
    parts += mode ? it.C : it.C * cpr;
    if (mode == 0u) max_cpr = cpr > max_cpr ? cpr : max_cpr;
  }
  if (seg_wgs >= (1ull << 31) || chunks >= (1ull << 31) || rows >= (1ull << 31)) return SBQ_ERR_ARG;
  const size_t bytes = sizeof(CalibHeader) + static_cast<size_t>(n_items) * sizeof(CalibItemDev) + (seg_wgs + chunks) * 4;
  if (bytes_needed_out) *bytes_needed_out = bytes;
  if (n_rows_out) *n_rows_out = static_cast<uint32_t>(rows);
  const size_t ws_stats = segs * sizeof(StatPartial), ws_mse = parts * SBQ_MSE_CANDIDATES * sizeof(double);
  if (workspace_bytes_out) *workspace_bytes_out = ws_stats > ws_mse ? ws_stats : ws_mse;
  if (!host_table) return SBQ_OK;  // size query
  if (host_table_bytes < bytes) return SBQ_ERR_WORKSPACE;
  char* base = static_cast<char*>(host_table);
  CalibHeader* h = reinterpret_cast<CalibHeader*>(base);
  CalibItemDev* dev = reinterpret_cast<CalibItemDev*>(base + sizeof(CalibHeader));
  uint32_t* wg_item = reinterpret_cast<uint32_t*>(dev + n_items);
  uint32_t* chunk_item = wg_item + seg_wgs;
  *h = CalibHeader{};
  h->n_items = static_cast<uint32_t>(n_items);
  h->n_seg_wgs = static_cast<uint32_t>(seg_wgs);
  h->n_rows = static_cast<uint32_t>(rows);
  h->n_chunks = static_cast<uint32_t>(chunks);
  h->n_segs = static_cast<uint32_t>(segs);
  h->max_chunks_per_row = static_cast<uint32_t>(max_cpr);
  h->n_parts = static_cast<uint32_t>(parts);
  h->n_lane_chunks = static_cast<uint32_t>(lane_chunks);
  // (the MSE chunk list: the lane kernel's workgroups first, then the chunk kernel's -- two launches, two block sizes)
  uint32_t wg = 0, sg = 0, rw = 0, ck_lane = 0, ck_rest = static_cast<uint32_t>(lane_chunks), pt = 0;
  for (int i = 0; i < n_items; ++i) {
    const sbq_calib_item& it = items[i];
    CalibItemDev d{};
    d.x = it.x;
    d.out_off = it.out_offset;
    d.C = static_cast<uint32_t>(it.C);
    d.inner = static_cast<uint32_t>(it.inner);
    d.segs_per_row = static_cast<uint32_t>(ceil_div(it.inner, static_cast<int64_t>(kStatsChunk)));
    d.seg_wg_begin = wg;
    d.seg_begin = sg;
    d.row_begin = rw;
    d.mse_waves = mse_mode(it.inner);
    uint32_t& ck = d.mse_waves == 8u ? ck_lane : ck_rest;
    d.chunk_begin = ck;
    d.part_begin = pt;
    d.qlo = static_cast<float>(it.qmin);
    d.qhi = static_cast<float>(it.qmax);
    d.flags = it.flags;
    dev[i] = d;
    const uint32_t n_seg = d.C * d.segs_per_row;
    const uint32_t n_wg = (n_seg + kWavesPerBlock - 1) / kWavesPerBlock;
    for (uint32_t k = 0; k < n_wg; ++k) wg_item[wg + k] = static_cast<uint32_t>(i);
    const uint32_t cpr = static_cast<uint32_t>(ceil_div(it.inner, static_cast<int64_t>(kMseChunk)));
    const uint32_t n_ck = static_cast<uint32_t>(mse_chunks(it, d.mse_waves));
    for (uint32_t k = 0; k < n_ck; ++k) chunk_item[ck + k] = static_cast<uint32_t>(i);
    wg += n_wg;
    sg += n_seg;
    rw += d.C;
    ck += n_ck;
    pt += d.mse_waves ? d.C : d.C * cpr;
  }
  return SBQ_OK;
}

static int calib_header(const void* host_table, sbq::CalibHeader& h) {
  if (!host_table) return SBQ_ERR_NULL;
  h = *static_cast<const sbq::CalibHeader*>(host_table);
  if (h.n_items == 0) return SBQ_ERR_EMPTY;
  return SBQ_OK;
}

int sbq_group_minmax_qparams(const void* device_table, const void* host_table, int x_dtype, float* min_base,
                             float* max_base, float* scale_base, float* zp_base, void* workspace,
                             size_t workspace_bytes, void* stream) {
  using namespace sbq;
  if (!valid_dtype(x_dtype)) return SBQ_ERR_DTYPE;
  CalibHeader h;
  int rc = calib_header(host_table, h);
  if (rc != SBQ_OK) return rc;
  if (!device_table || !min_base || !max_base || !workspace) return SBQ_ERR_NULL;
  if ((scale_base == nullptr) != (zp_base == nullptr)) return SBQ_ERR_NULL;
  if (!aligned16(device_table) || !aligned16(workspace)) return SBQ_ERR_ALIGN;
  if (workspace_bytes < static_cast<size_t>(h.n_segs) * sizeof(StatPartial)) return SBQ_ERR_WORKSPACE;
  hipStream_t st = as_stream(stream);
  const char* base = static_cast<const char*>(device_table);
  const CalibItemDev* items = reinterpret_cast<const CalibItemDev*>(base + sizeof(CalibHeader));
  const uint32_t* wg_item = reinterpret_cast<const uint32_t*>(items + h.n_items);
  StatPartial* part = static_cast<StatPartial*>(workspace);
  rc = dispatch_dtype(x_dtype, [&](auto tag) {
    using T = decltype(tag);
    calib_stats_kernel<T><<<h.n_seg_wgs, kBlock, 0, st>>>(items, wg_item, part);
  });
  if (rc != SBQ_OK) return rc;
  rc = check_launch();
  if (rc != SBQ_OK) return rc;
  calib_stats_fold_kernel<<<(h.n_rows + kBlock - 1) / kBlock, kBlock, 0, st>>>(items, h.n_items, h.n_rows, part, min_base,
                                                                             max_base, scale_base, zp_base);
  return check_launch();
}

int sbq_group_mse_qparams(const void* device_table, const void* host_table, int x_dtype, const float* min_base,
                          const float* max_base, float* scale_base, float* zp_base, int32_t* index_base,
                          void* workspace, size_t workspace_bytes, void* stream) {
  using namespace sbq;
  if (!valid_dtype(x_dtype)) return SBQ_ERR_DTYPE;
  CalibHeader h;
  int rc = calib_header(host_table, h);
  if (rc != SBQ_OK) return rc;
  if (!device_table || !min_base || !max_base || !scale_base || !zp_base || !workspace) return SBQ_ERR_NULL;
  if (!aligned16(device_table) || !aligned16(workspace)) return SBQ_ERR_ALIGN;
  if (h.max_chunks_per_row > kMaxGroupMseChunks) return SBQ_ERR_ARG;  // rows beyond one fold level: per-tensor path
  if (workspace_bytes < static_cast<size_t>(h.n_parts) * SBQ_MSE_CANDIDATES * sizeof(double)) return SBQ_ERR_WORKSPACE;
  hipStream_t st = as_stream(stream);
  const char* base = static_cast<const char*>(device_table);
  const CalibItemDev* items = reinterpret_cast<const CalibItemDev*>(base + sizeof(CalibHeader));
  const uint32_t* chunk_item = reinterpret_cast<const uint32_t*>(items + h.n_items) + h.n_seg_wgs;
  double* part = static_cast<double*>(workspace);
  rc = dispatch_dtype(x_dtype, [&](auto tag) {
    using T = decltype(tag);
    if (h.n_lane_chunks)
      calib_mse_lanes_kernel<T><<<h.n_lane_chunks, kLaneBlock, 0, st>>>(items, chunk_item, min_base, max_base, scale_base, zp_base,
                                                                       index_base);
    if (h.n_chunks > h.n_lane_chunks)
      calib_mse_kernel<T><<<h.n_chunks - h.n_lane_chunks, kBlock, 0, st>>>(items, chunk_item, min_base, max_base, part,
                                                                          h.n_lane_chunks);
  });
  if (rc != SBQ_OK) return rc;
  rc = check_launch();
  if (rc != SBQ_OK) return rc;
  if (h.n_chunks > h.n_lane_chunks)  // (rows of the chunk / wave forms; the lane kernel picks its own)
    calib_mse_select_kernel<<<(h.n_rows + kRowsPerFoldWg - 1) / kRowsPerFoldWg, kBlock, 0, st>>>(
        items, h.n_items, h.n_rows, part, min_base, max_base, scale_base, zp_base, index_base);
  return check_launch();
}

}  // extern "C"
