// sbq_observe.hip -- observer reductions for gfx950: per-channel min / max /
// sum|x| in one pass, scale/zero-point from min/max, LSQ scale init and the
// 80-candidate MSE search.
//
// Replaces the torch-op bodies of
//   sparsebit/quantization/observers/minmax.py:14-25     (min / max)
//   sparsebit/quantization/observers/base.py:63-79       (calc_qparams_with_minmax)
//   sparsebit/quantization/observers/mse.py:28-63        (MSE search)
//   sparsebit/quantization/quantizers/lsq.py:44-47       (LSQ init)
// and removes DataCache's torch.cat/transpose copy (observers/base.py:21-36):
// the kernels index the original [outer, C, inner] tensor in place.
//
// Structure: HBM-bound read-once reductions.  A workgroup reduces one "chunk"
// (a run of packs inside one channel row) with 16-byte loads, 64-lane shuffle
// reductions and a 4-entry LDS cross-wave step, and writes ONE partial record;
// a second tiny kernel folds the partials of a channel in a fixed order, so the
// results are deterministic (no float atomics).
#include "sbq_observe_body.hpp"

namespace sbq {
namespace {

// ---- stage 1: one partial {min, max, sum|x|} per chunk ---------------------------
// A chunk (<= 4096 elements of one channel row) belongs to ONE WAVE: 8 packs per lane are
// requested back to back (128 B per lane in flight) and the four accumulators are folded
// with shuffle trees only -- no LDS, no barrier, nothing shared between the 4 waves of a
// workgroup.  min/max use the NaN-dropping v_min/v_max plus a separate "saw a NaN" flag,
// which reproduces torch's NaN-propagating result at 1 op per element instead of 5.  When a
// channel is a single chunk (a [C, inner <= 4096] weight) the result is final and written
// directly: no second kernel.
struct StatAcc {
  float mn, mx, as;
  int nan;
};

// min / max of one chunk (<= 4096 elements starting at element `row_base`, whole packs up to `vlen`, `len` in all) by
// one wave: 16-bit inputs as raw bit patterns (Stat16), fp32 with v_minimum3 / v_maximum3 -- NaN propagates either way.
template <typename T>
__device__ __forceinline__ void minmax_chunk(const void* __restrict__ x, const int64_t row_base, const uint32_t len,
                                             const int lane, float& mn, float& mx) {
  const uint32_t vlen = len & ~static_cast<uint32_t>(kPack - 1), end = len;
  const uint32_t begin = 0, vend = vlen;
  constexpr int U = 8;
  if constexpr (T::id == SBQ_F32) {
    mn = __builtin_inff();
    mx = -__builtin_inff();
    if (vend > begin) {
      RawPack<T> raw[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        // two 16-byte runs 256 elements apart (see load_raw2); a run past the end folds the chunk's last one again
        const uint32_t eA = u * kWave * kPack + 4 * lane;
        const uint32_t eB = eA + kWave * kPack / 2;
        raw[u] = load_raw2<T, true>(x, row_base + (eA < vend ? eA : vend - 4), row_base + (eB < vend ? eB : vend - 4));
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float v[kPack];
        unpack_raw<T>(raw[u], v);
#pragma unroll
        for (int q = 0; q < kPack; q += 2) {
          mn = __builtin_elementwise_minimum(__builtin_elementwise_minimum(mn, v[q]), v[q + 1]);
          mx = __builtin_elementwise_maximum(__builtin_elementwise_maximum(mx, v[q]), v[q + 1]);
        }
      }
    }
    for (uint32_t e = vend + lane; e < end; e += kWave) {  // ragged tail of a per-tensor row (< 8 elements)
      const float f = Elem<T>::load1(x, row_base + e);
      mn = __builtin_elementwise_minimum(mn, f);
      mx = __builtin_elementwise_maximum(mx, f);
    }
    mn = wave_reduce(mn, [](float a, float b) { return __builtin_elementwise_minimum(a, b); });
    mx = wave_reduce(mx, [](float a, float b) { return __builtin_elementwise_maximum(a, b); });
  } else {
    Stat16 s = kStat16Identity;
    if (vend > begin) {
      RawPack<T> raw[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        uint32_t e = (u * kWave + lane) * kPack;
        if (e >= vend) e = vend - kPack;
        raw[u] = load_raw<T, true>(x, row_base + e);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int q = 0; q < 4; ++q) stat16_fold(s, raw[u].d[0][q]);
      }
    }
    for (uint32_t e = vend + lane; e < end; e += kWave) {
      const uint32_t w = static_cast<const uint16_t*>(x)[row_base + e];
      stat16_fold(s, w | (w << 16));
    }
    s = stat16_wave(s);
    stat16_decode<T>(s, mn, mx);
  }
}

template <typename T, bool FINAL>
__global__ __launch_bounds__(kBlock) void stats_minmax_kernel(const void* __restrict__ x, StatPartial* __restrict__ part,
                                                              float* __restrict__ min_out, float* __restrict__ max_out,
                                                              uint32_t n_chunks, int64_t inner, const ChunkGeom g) {
  // (n_chunks and inner in front of the struct: with the pointers they are the argument dwords that arrive preloaded
  // in SGPRs, and a weight's launch (FINAL) needs nothing else before its first load)
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t cid = blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);  // wave-uniform
  if (cid >= n_chunks) return;
  // FINAL == one chunk per channel == a [C, inner <= 4096] weight: the chunk is row `cid`, no divisions in front of
  // the first load.  Offsets inside a chunk are 32-bit (a chunk holds at most 4096 elements).
  uint32_t c_out = cid;
  int64_t first = static_cast<int64_t>(cid) * inner;  // element index of the chunk's first element
  uint32_t len = static_cast<uint32_t>(inner);
  if constexpr (!FINAL) {
    const ChunkPos cp = chunk_pos(g, cid);
    c_out = cp.c;
    first = cp.row_base + cp.begin;
    len = static_cast<uint32_t>(cp.end - cp.begin);
  }
  float mn, mx;
  minmax_chunk<T>(x, first, len, lane, mn, mx);
  if (lane == 0) {
    if constexpr (FINAL) {
      if (min_out) min_out[c_out] = mn;
      if (max_out) max_out[c_out] = mx;
    } else {
      part[cid] = StatPartial{mn, mx, 0.0};
    }
  }
}

// ---- the streaming per-tensor observer: one launch per calibration batch, nothing to fold ---------------------------
// observers/minmax.py:14-25 per tensor over batches that arrive one by one (tools/calibration.py:109-115).  As
// sbq_channel_stats a batch cost two launches (chunk partials, then their fold) plus the torch.minimum / maximum that
// joined it to the running statistic: a 51 MB batch spent 6 of its 16 us outside its read.  min / max are order
// independent, so every workgroup folds its four chunks and updates the observer's running state directly with one
// integer atomicMax / atomicMin pair -- exact, deterministic, no second launch, and the state IS the running
// statistic.  state: uint32[2048] = 64 slots of one 128-byte line, word 0 of a slot the largest key so far, word 1 the smallest;
// key = the usual order-preserving map of the float's bits, with NaN sent to the top of the max word and to the
// bottom of the min word, so that a NaN anywhere makes both results NaN like torch.min / torch.max.
__device__ __forceinline__ uint32_t ordered_key(float f) {
  const uint32_t b = __builtin_bit_cast(uint32_t, f);
  return b ^ ((b & 0x80000000u) ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ float ordered_value(uint32_t k) {
  return __builtin_bit_cast(float, k ^ ((k & 0x80000000u) ? 0x80000000u : 0xffffffffu));
}
// (64 slots, a 128-byte line each: word 0 of a slot the largest key, word 1 the smallest.  Workgroup b updates slot
// b % 64: a thousand workgroups updating ONE pair of addresses queued up at the memory side -- 8 ns per atomic, 16 us for
// a 16.7 M-element tensor whose read takes 7)
constexpr int kMmSlots = 64, kMmSlotWords = 32;
template <typename T>
__global__ __launch_bounds__(kBlock) void minmax_accumulate_kernel(const void* __restrict__ x, int64_t n, uint32_t n_chunks,
                                                                   uint32_t* __restrict__ state) {
  __shared__ float s_mn[kWavesPerBlock], s_mx[kWavesPerBlock];
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t wid = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const uint32_t cid = blockIdx.x * kWavesPerBlock + wid;
  float mn = __builtin_inff(), mx = -__builtin_inff();
  if (cid < n_chunks) {
    const int64_t first = static_cast<int64_t>(cid) * kStatsChunk;
    const int64_t left = n - first;
    minmax_chunk<T>(x, first, static_cast<uint32_t>(left < kStatsChunk ? left : kStatsChunk), lane, mn, mx);
  }
  if (lane == 0) {
    s_mn[wid] = mn;
    s_mx[wid] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 1; w < kWavesPerBlock; ++w) {
      mn = __builtin_elementwise_minimum(mn, s_mn[w]);
      mx = __builtin_elementwise_maximum(mx, s_mx[w]);
    }
    const uint32_t kmax = mx != mx ? 0xffffffffu : ordered_key(mx);
    const uint32_t kmin = mn != mn ? 0u : ordered_key(mn);
    uint32_t* slot = state + (blockIdx.x % kMmSlots) * kMmSlotWords;
    atomicMax(slot, kmax);
    atomicMin(slot + 1, kmin);
  }
}
__global__ void minmax_state_kernel(uint32_t* __restrict__ state, float* __restrict__ min_out, float* __restrict__ max_out) {
  const int t = threadIdx.x;  // one wave: lane t owns slot t
  if (!min_out) {  // reset: the identities of max / min
    state[t * kMmSlotWords] = 0u;
    state[t * kMmSlotWords + 1] = 0xffffffffu;
    return;
  }
  uint32_t kmax = state[t * kMmSlotWords], kmin = state[t * kMmSlotWords + 1];
  kmax = dpp_reduce_u32(kmax, 0u, [](uint32_t a, uint32_t b) { return a > b ? a : b; });
  kmin = dpp_reduce_u32(kmin, 0xffffffffu, [](uint32_t a, uint32_t b) { return a < b ? a : b; });
  if (t != 0) return;
  const bool nan = kmax == 0xffffffffu || kmin == 0u;
  const float qnan = __builtin_nanf("");
  min_out[0] = nan ? qnan : ordered_value(kmin);
  max_out[0] = nan ? qnan : ordered_value(kmax);
}

template <typename T, bool VEC, bool FINAL>
__global__ __launch_bounds__(kBlock) void stats_partial_kernel(const void* __restrict__ x,
                                                               StatPartial* __restrict__ part,
                                                               float* __restrict__ min_out,
                                                               float* __restrict__ max_out,
                                                               double* __restrict__ abssum_out,
                                                               const ChunkGeom g, uint32_t n_chunks) {
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t cid = blockIdx.x * kWavesPerBlock + threadIdx.x / kWave;  // wave-uniform
  if (cid >= n_chunks) return;
  const ChunkPos cp = chunk_pos(g, cid);
  const int64_t row_base = cp.row_base, begin = cp.begin, end = cp.end;

  StatAcc a{__builtin_inff(), -__builtin_inff(), 0.0f, 0};
  auto visit = [&](float f) {
    a.mn = __builtin_fminf(a.mn, f);
    a.mx = __builtin_fmaxf(a.mx, f);
    a.nan |= (f != f);
    a.as += __builtin_fabsf(f);
  };
  if constexpr (VEC) {
    const int64_t vend = begin + ((end - begin) / kPack) * kPack;
    constexpr int U = 8;
    if (vend > begin) {
      RawPack<T> raw[U];
      bool ok[U], okB[U];
      // fp32: the lane's two 16-byte loads are 256 elements apart, so each is part of a contiguous
      // 1 KiB wave access (sbq_common.hpp: load_raw2) instead of a stride-32-byte one
      constexpr bool SPLIT = T::id == SBQ_F32;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if constexpr (SPLIT) {
          const int64_t eA = begin + static_cast<int64_t>(u) * kWave * kPack + 4 * lane;
          const int64_t eB = eA + kWave * kPack / 2;
          ok[u] = eA < vend;
          okB[u] = eB < vend;
          raw[u] = load_raw2<T, true>(x, row_base + (ok[u] ? eA : vend - 4), row_base + (okB[u] ? eB : vend - 4));
        } else {
          int64_t e = begin + (static_cast<int64_t>(u) * kWave + lane) * kPack;
          ok[u] = okB[u] = e < vend;
          if (!ok[u]) e = vend - kPack;
          raw[u] = load_raw<T, true>(x, row_base + e);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float v[kPack];
        unpack_raw<T>(raw[u], v);
        if (ok[u]) {
#pragma unroll
          for (int q = 0; q < 4; ++q) visit(v[q]);
        }
        if (okB[u]) {
#pragma unroll
          for (int q = 4; q < kPack; ++q) visit(v[q]);
        }
      }
    }
    // ragged tail of a per-tensor row (< 8 elements)
    for (int64_t e = vend + lane; e < end; e += kWave) visit(Elem<T>::load1(x, row_base + e));
  } else {
    for (int64_t e = begin + lane; e < end; e += kWave) visit(Elem<T>::load1(x, row_base + e));
  }
  float mn = wave_reduce(a.mn, MinF());
  float mx = wave_reduce(a.mx, MaxF());
  const int nan = wave_reduce(a.nan, OrI());
  const double as = wave_reduce(static_cast<double>(a.as), Sum());
  if (lane == 0) {
    if (nan) mn = mx = __builtin_nanf("");
    if constexpr (FINAL) {
      if (min_out) min_out[cp.c] = mn;
      if (max_out) max_out[cp.c] = mx;
      if (abssum_out) abssum_out[cp.c] = as;
    } else {
      part[cid] = StatPartial{mn, mx, as};
    }
  }
}

// ---- moments: sum x, sum x^2 (fp64) or sum |x - center[c]| -------------------------
// Same wave-per-chunk structure as the min/max pass.  Accumulation is fp64 per element
// (the variance formula cancels; fp64 vector rate is ample for a read-once kernel).
struct MomentPartial {
  double s1, s2;
};

template <typename T, bool VEC, bool ABSDEV>
__global__ __launch_bounds__(kBlock) void moments_partial_kernel(const void* __restrict__ x,
                                                                 const float* __restrict__ center,
                                                                 MomentPartial* __restrict__ part,
                                                                 const ChunkGeom g, uint32_t n_chunks) {
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t cid = blockIdx.x * kWavesPerBlock + threadIdx.x / kWave;
  if (cid >= n_chunks) return;
  const ChunkPos cp = chunk_pos(g, cid);
  const int64_t row_base = cp.row_base, begin = cp.begin, end = cp.end;
  float m = 0.0f;
  if constexpr (ABSDEV) m = center[cp.c];
  double a1 = 0.0, a2 = 0.0;
  auto visit = [&](float f) {
    if constexpr (ABSDEV) {
      a1 += static_cast<double>(__builtin_fabsf(f - m));  // fp32 subtract like torch.abs(data - mean)
    } else {
      const double d = static_cast<double>(f);
      a1 += d;
      a2 += d * d;
    }
  };
  if constexpr (VEC) {
    const int64_t vend = begin + ((end - begin) / kPack) * kPack;
    constexpr int U = 8;
    if (vend > begin) {
      RawPack<T> raw[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        int64_t e = begin + (static_cast<int64_t>(u) * kWave + lane) * kPack;
        ok[u] = e < vend;
        if (!ok[u]) e = vend - kPack;
        raw[u] = load_raw<T, true>(x, row_base + e);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float v[kPack];
        unpack_raw<T>(raw[u], v);
        if (ok[u]) {
#pragma unroll
          for (int q = 0; q < kPack; ++q) visit(v[q]);
        }
      }
    }
    for (int64_t e = vend + lane; e < end; e += kWave) visit(Elem<T>::load1(x, row_base + e));
  } else {
    for (int64_t e = begin + lane; e < end; e += kWave) visit(Elem<T>::load1(x, row_base + e));
  }
  a1 = wave_reduce(a1, Sum());
  a2 = wave_reduce(a2, Sum());
  if (lane == 0) part[cid] = MomentPartial{a1, a2};
}

__global__ __launch_bounds__(kBlock) void moments_finish_kernel(const MomentPartial* __restrict__ part,
                                                                uint32_t chunks_per_chan,
                                                                double* __restrict__ out1,
                                                                double* __restrict__ out2) {
  __shared__ double s_d[kWavesPerBlock];
  const uint32_t c = blockIdx.x;
  const MomentPartial* p = part + static_cast<size_t>(c) * chunks_per_chan;
  double a1 = 0.0, a2 = 0.0;
  constexpr int kBatch = 8;  // eight records per lane in flight (see stats_finish_kernel); same summation order
  for (uint32_t i0 = threadIdx.x; i0 < chunks_per_chan; i0 += kBlock * kBatch) {
    MomentPartial r[kBatch];
#pragma unroll
    for (int b = 0; b < kBatch; ++b) {
      const uint32_t i = i0 + b * kBlock;
      r[b] = i < chunks_per_chan ? p[i] : MomentPartial{0.0, 0.0};
    }
#pragma unroll
    for (int b = 0; b < kBatch; ++b) {
      a1 += r[b].s1;
      a2 += r[b].s2;
    }
  }
  a1 = block_reduce(a1, Sum(), s_d);
  a2 = block_reduce(a2, Sum(), s_d);
  if (threadIdx.x == 0) {
    if (out1) out1[c] += a1;
    if (out2) out2[c] += a2;
  }
}

__global__ void aciq_kernel(const float* __restrict__ mn, const float* __restrict__ mx,
                            const float* __restrict__ b, int64_t C, float alpha, float gaus_const,
                            float sqrt_2logn, int half_range, float* __restrict__ min_out,
                            float* __restrict__ max_out) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= C) return;
  float t;
  if (b) {
    t = alpha * b[i];
  } else {
    const float std = ((mx[i] - mn[i]) * gaus_const) / sqrt_2logn;
    t = alpha * std;
  }
  max_out[i] = t;
  min_out[i] = half_range ? 0.0f : -t;
}

__global__ void minmax_pack_kernel(const float* __restrict__ mn, const float* __restrict__ mx, int64_t C,
                                   float* __restrict__ buf) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= C) return;
  const float a = mx[i], b = mn[i];
  const bool na = a != a, nb = b != b;
  buf[i] = na ? -__builtin_inff() : a;
  buf[C + i] = nb ? -__builtin_inff() : -b;
  buf[2 * C + i] = na ? 1.0f : 0.0f;
  buf[3 * C + i] = nb ? 1.0f : 0.0f;
}

__global__ void minmax_unpack_kernel(const float* __restrict__ buf, int64_t C, float* __restrict__ mn,
                                     float* __restrict__ mx) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= C) return;
  const float nan = __builtin_nanf("");
  mx[i] = buf[2 * C + i] > 0.0f ? nan : buf[i];
  mn[i] = buf[3 * C + i] > 0.0f ? nan : -buf[C + i];
}

// moving_average.py:23-31, one thread: the recurrence is sequential by definition
__global__ void ema_minmax_kernel(const float* __restrict__ smin, const float* __restrict__ smax, int64_t n,
                                  float ratio, float one_minus, float* __restrict__ state, int has_state) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float mn = state[0], mx = state[1];
  int64_t i = 0;
  if (!has_state) {
    mn = smin[0];
    mx = smax[0];
    i = 1;
  }
  for (; i < n; ++i) {
    mx = ratio * mx + one_minus * smax[i];
    mn = ratio * mn + one_minus * smin[i];
  }
  state[0] = mn;
  state[1] = mx;
}

// ---- channels-last (inner == 1): [rows, C] reduced over rows, per column -----------------
// The NLC activation layout observed per channel.  A lane owns 8 adjacent channels (one pack
// per row), a workgroup 2048 channels, and blockIdx.y strides over the rows; partials land in
// the same part[c][split] layout, so stats_finish_kernel folds them unchanged.
template <typename T>
__global__ __launch_bounds__(kBlock) void stats_clast_kernel(const void* __restrict__ x,
                                                             StatPartial* __restrict__ part, uint32_t rows,
                                                             uint32_t C, uint32_t splits) {
  const uint32_t c0 = (blockIdx.x * kBlock + threadIdx.x) * kPack;
  if (c0 >= C) return;
  float mn[kPack], mx[kPack], as[kPack];
  int nan = 0;  // bit j: channel c0 + j saw a NaN
#pragma unroll
  for (int j = 0; j < kPack; ++j) {
    mn[j] = __builtin_inff();
    mx[j] = -__builtin_inff();
    as[j] = 0.0f;
  }
  double asd[kPack];
#pragma unroll
  for (int j = 0; j < kPack; ++j) asd[j] = 0.0;
  uint32_t since_flush = 0;
  for (uint32_t r = blockIdx.y; r < rows; r += splits) {
    float v[kPack];
    load_pack<T, true>(x, static_cast<int64_t>(r) * C + c0, v);
#pragma unroll
    for (int j = 0; j < kPack; ++j) {
      mn[j] = __builtin_fminf(mn[j], v[j]);
      mx[j] = __builtin_fmaxf(mx[j], v[j]);
      nan |= (v[j] != v[j]) << j;
      as[j] += __builtin_fabsf(v[j]);
    }
    if (++since_flush == 64) {  // bound the fp32 run length like the row kernels do
#pragma unroll
      for (int j = 0; j < kPack; ++j) {
        asd[j] += static_cast<double>(as[j]);
        as[j] = 0.0f;
      }
      since_flush = 0;
    }
  }
#pragma unroll
  for (int j = 0; j < kPack; ++j) {
    const bool bad = (nan >> j) & 1;
    StatPartial p;
    p.mn = bad ? __builtin_nanf("") : mn[j];
    p.mx = bad ? __builtin_nanf("") : mx[j];
    p.abssum = asd[j] + static_cast<double>(as[j]);
    part[static_cast<size_t>(c0 + j) * splits + blockIdx.y] = p;
  }
}

// ---- stage 2: fold a channel's partials (fixed order => deterministic) ------------
__global__ __launch_bounds__(kBlock) void stats_finish_kernel(const StatPartial* __restrict__ part,
                                                              uint32_t chunks_per_chan,
                                                              float* __restrict__ min_out,
                                                              float* __restrict__ max_out,
                                                              double* __restrict__ abssum_out) {
  __shared__ float s_f[kWavesPerBlock];
  __shared__ double s_d[kWavesPerBlock];
  const uint32_t c = blockIdx.x;
  const StatPartial* p = part + static_cast<size_t>(c) * chunks_per_chan;
  float mn = __builtin_inff(), mx = -__builtin_inff();
  double as = 0.0;
  NanMin nmin;
  NanMax nmax;
  // eight 16-byte records per lane requested before the first is folded (a per-tensor call has one
  // channel with thousands of partials: a dependent load per iteration made this tiny kernel as long as
  // the pass over the data it follows)
  constexpr int kBatch = 8;
  for (uint32_t i0 = threadIdx.x; i0 < chunks_per_chan; i0 += kBlock * kBatch) {
    StatPartial r[kBatch];
#pragma unroll
    for (int b = 0; b < kBatch; ++b) {
      const uint32_t i = i0 + b * kBlock;
      r[b] = i < chunks_per_chan ? p[i] : StatPartial{__builtin_inff(), -__builtin_inff(), 0.0};
    }
#pragma unroll
    for (int b = 0; b < kBatch; ++b) {
      mn = nmin(mn, r[b].mn);
      mx = nmax(mx, r[b].mx);
      as += r[b].abssum;
    }
  }
  mn = block_reduce(mn, nmin, s_f);
  mx = block_reduce(mx, nmax, s_f);
  as = block_reduce(as, Sum(), s_d);
  if (threadIdx.x == 0) {
    if (min_out) min_out[c] = mn;
    if (max_out) max_out[c] = mx;
    if (abssum_out) abssum_out[c] = as;
  }
}

__global__ void qparams_kernel(const float* __restrict__ mn, const float* __restrict__ mx, int64_t C,
                               float qrange, int symmetric, float* __restrict__ scale,
                               float* __restrict__ zp) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= C) return;
  float s, z;
  qparams_from_minmax(mn[i], mx[i], qrange, symmetric != 0, s, z);
  scale[i] = s;
  zp[i] = z;
}

// lsq.py:44-47: scale = 2 * mean(|x|) / sqrt(qmax); torch computes the mean and
// both scalings in fp32 (the Python scalar sqrt(qmax) is rounded to fp32).
__global__ void lsq_init_kernel(const double* __restrict__ abssum, int64_t C, double count,
                                float sqrt_qmax, float* __restrict__ scale) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= C) return;
  const float mean = static_cast<float>(abssum[i] / count);
  scale[i] = (2.0f * mean) / sqrt_qmax;
}

// One workgroup keeps a 4096-element chunk of one channel in registers and walks
// the 80 candidates over it (mse_chunk_body, sbq_observe_body.hpp): x is read from HBM once, not 80 times (the
// reference makes 80 x 4 full passes).  Output: part[bid][80] fp64 partial sums of (x-dq)^2.
template <typename T, bool VEC>
__global__ __launch_bounds__(kBlock) void mse_partial_kernel(
    const void* __restrict__ x, const float* __restrict__ min_val, const float* __restrict__ max_val,
    double* __restrict__ part, double* __restrict__ sse, const ChunkGeom g, float qrange, float qlo,
    float qhi, int symmetric) {
  __shared__ MseLds lds;
  const uint32_t bid = blockIdx.x;
  const ChunkPos cp = chunk_pos(g, bid);
  const uint32_t c = cp.c;
  const double t = mse_chunk_body<T, VEC>(lds, x, cp.row_base, cp.begin, cp.end, min_val[c], max_val[c], qrange, qlo, qhi,
                                          symmetric != 0);
  if (threadIdx.x < SBQ_MSE_CANDIDATES) {
    if (g.chunks_per_chan == 1)  // the chunk IS the channel: accumulate in place, no fold kernel
      sse[static_cast<size_t>(c) * SBQ_MSE_CANDIDATES + threadIdx.x] += t;
    else
      part[static_cast<size_t>(bid) * SBQ_MSE_CANDIDATES + threadIdx.x] = t;
  }
}

// ---- the MSE observer of a 16-bit tensor taken as a WHOLE: a histogram instead of 80 evaluations per element -------
// observers/mse.py:46-61 per tensor evaluates 80 candidates on every element: 80 x N quantize-dequantize-square
// operations, the vector ALU's whole capacity for 150 us at N = 16.7 M.  A bf16 / fp16 tensor holds at most 65 536
// distinct VALUES: the loss of a candidate is sum over values of count(v) * err(v)^2.  So the tensor is read ONCE into
// an exact histogram of its 16-bit keys -- each workgroup counts its 65 536 elements in LDS (two 16-bit counts per
// dword, +-0 per lane: the counting of sbq_select_win.hip's h16_select_kernel) and adds its occupied bins to one of 8
// global copies -- and the 80 candidates are evaluated on the 65 536 values, weighted by their counts: HBM-bound
// instead of VALU-bound, and the per-value error is computed with the very operations of mse_chunk_body, so a
// candidate's loss is the same sum with its equal terms collected.  fp64 from the first multiplication on.
constexpr int kHistCopies = 8;
constexpr uint32_t kHistKeys = 65536;
constexpr size_t kHist16Bytes = static_cast<size_t>(kHistCopies) * kHistKeys * 4;
constexpr int kH16Threads = 1024;
constexpr uint32_t kH16Elems = 65536;  // per workgroup: 8 packs of 8 per thread
constexpr uint32_t kEvalBlocks = kHistKeys / kBlock;  // 256 workgroups of 256 keys

template <typename T>
__global__ __launch_bounds__(kH16Threads) void hist16_kernel(const void* __restrict__ x, uint32_t n, uint32_t* __restrict__ ghist) {
  extern __shared__ __attribute__((aligned(16))) uint32_t hist[];  // 32 Ki dwords
  __shared__ uint32_t zero_word[kH16Threads];
  __shared__ uint32_t s_tot[kH16Threads / kWave];
  __shared__ uint32_t s_first;
  const uint32_t wg = blockIdx.x, nwg = gridDim.x;
  const uint32_t n_packs = n / kPack;
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  u32x4 raw[8];
  uint32_t okmask = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    // slab (16 Ki elements) j / 2 of this workgroup, half j % 2: every wave instruction reads 1 KiB
    const uint32_t pk = (wg + static_cast<uint32_t>(j / 2) * nwg) * 2048u + (j % 2) * kH16Threads + threadIdx.x;
    const bool there = pk < n_packs;
    okmask |= there ? 1u << j : 0u;
    // (a pack that is not there re-reads the last one that is; the host never launches with n < 8: sbq_mse_accumulate folds
    // a rest shorter than one pack into the launch before it -- the clamp below only keeps the address inside x)
    raw[j] = load_raw<T, true>(x, static_cast<int64_t>((there ? pk : (n_packs ? n_packs - 1u : 0u)) * static_cast<uint32_t>(kPack))).d[0];
  }
  __builtin_amdgcn_sched_barrier(0);
  {
    u32x4* h4 = reinterpret_cast<u32x4*>(hist);
#pragma unroll
    for (int i = 0; i < 8; ++i) h4[i * kH16Threads + threadIdx.x] = u32x4{0, 0, 0, 0};
    zero_word[threadIdx.x] = 0;
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  constexpr uint32_t kZero16 = Key16<T>::kZero >> 16;
  typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
  auto pkmin = [](uint32_t p, uint32_t q) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, p), __builtin_bit_cast(u16x2, q)));
  };
  uint32_t first_key = 0;
  bool zero_hot = false;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const bool there = (okmask & (1u << j)) != 0;
    const u32x4 r = raw[j];
    if (j == 0) {
      const uint32_t m2 = pkmin(pkmin(r[0] & 0x7fff7fffu, r[1] & 0x7fff7fffu), pkmin(r[2] & 0x7fff7fffu, r[3] & 0x7fff7fffu));
      zero_hot = __builtin_amdgcn_ballot_w64(there && ((m2 & 0xffffu) == 0u || (m2 >> 16) == 0u)) != 0;
    }
    if (!there) continue;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t k2 = Key16<T>::pack2(r[q], 0xffffffffu);
      if (j == 0 && q == 0) first_key = k2 & 0xffffu;
      if (!zero_hot) {
        atomicAdd(&hist[(k2 & 0xffffu) >> 1], 1u + (k2 & 1u) * 0xffffu);
        atomicAdd(&hist[k2 >> 17], 1u + ((k2 >> 16) & 1u) * 0xffffu);
      } else {
#pragma unroll
        for (int hsel = 0; hsel < 2; ++hsel) {
          const uint32_t k = hsel == 0 ? (k2 & 0xffffu) : (k2 >> 16);
          const bool z = (k - kZero16) <= 1u;
          atomicAdd(z ? &zero_word[threadIdx.x] : &hist[k >> 1], (k & 1u) ? 0x10000u : 1u);
        }
      }
    }
  }
  if (wg == 0 && wid == 1) {
    // the launch's last n % 8 elements: straight into the global histogram (copy 0), NOT into this workgroup's LDS
    // counts -- the single-key carry fallback below credits everything the workgroup counted to one key, and must
    // therefore cover whole packs only
    const uint32_t e = n_packs * kPack + lane;
    if (e < n) {
      const uint32_t k = Key16<T>::pack2(static_cast<const uint16_t*>(x)[e], 0xffffffffu) & 0xffffu;
      atomicAdd(&ghist[k], 1u);
    }
  }
  if (threadIdx.x == 0) s_first = first_key;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  // this workgroup's elements: a dword decodes wrong only when ONE key took all 65 536 of them (carry)
  uint32_t n_wg = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t p0 = (wg + static_cast<uint32_t>(j) * nwg) * 2048u, p1 = p0 + 2048u;
    if (p0 < n_packs) n_wg += ((p1 < n_packs ? p1 : n_packs) - p0) * kPack;
  }
  uint32_t wv[32], total = 0;
#pragma unroll
  for (int m = 0; m < 32; ++m) {
    wv[m] = hist[m * kH16Threads + threadIdx.x];
    total += (wv[m] & 0xffffu) + (wv[m] >> 16);
  }
  const uint32_t zw = zero_word[threadIdx.x];
  total += (zw & 0xffffu) + (zw >> 16);
  total = dpp_reduce_u32(total, 0u, [](uint32_t p, uint32_t q) { return p + q; });
  if (lane == 0) s_tot[wid] = total;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  uint32_t all = 0;
#pragma unroll
  for (int w = 0; w < kH16Threads / kWave; ++w) all += s_tot[w];
  uint32_t* g = ghist + static_cast<size_t>(wg % kHistCopies) * kHistKeys;
  if (all != n_wg) {  // (uniform) every element of this workgroup is the key thread 0 saw first
    if (threadIdx.x == 0) atomicAdd(&g[s_first], n_wg);
    return;
  }
#pragma unroll
  for (int m = 0; m < 32; ++m) {
    if (__builtin_amdgcn_ballot_w64(wv[m] != 0u) == 0) continue;
    // the two keys of a dword are neighbours in the global histogram too: ONE 64-bit add for both counters (their sums
    // stay below 2^32: no carry from the low counter into the high one)
    const uint32_t key0 = (m * kH16Threads + threadIdx.x) * 2u;
    if (wv[m])
      atomicAdd(reinterpret_cast<unsigned long long*>(&g[key0]),
                static_cast<unsigned long long>(wv[m] & 0xffffu) | (static_cast<unsigned long long>(wv[m] >> 16) << 32));
  }
  // the zeros: one add per WAVE and sign of zero
  const uint32_t zn = dpp_reduce_u32(zw & 0xffffu, 0u, [](uint32_t p, uint32_t q) { return p + q; });
  const uint32_t zp = dpp_reduce_u32(zw >> 16, 0u, [](uint32_t p, uint32_t q) { return p + q; });
  if (lane == 0) {
    if (zn) atomicAdd(&g[kZero16], zn);
    if (zp) atomicAdd(&g[kZero16 + 1u], zp);
  }
}

// 80 candidates x 65 536 values: a workgroup owns 256 keys.  Its occupied keys (a tensor's values fill a few per cent of
// the key space) are compacted into an LDS list of (value, count);
// thread (candidate i, third r) then walks its third of the list and adds count * err(value; candidate i)^2 in fp64 --
// the per-value error with the very operations of mse_chunk_body.  (One value per thread and a wave reduction per
// candidate -- 80 x 12 cross-lane moves of 64-bit values -- was 20 us on the workgroups at the centre of the
// distribution.)  part[b][80] = the workgroup's share of every candidate's squared error, folded in workgroup order
// by mse16_fold_kernel.
template <typename T>
__global__ __launch_bounds__(kBlock) void mse16_eval_kernel(const uint32_t* __restrict__ ghist, const float* __restrict__ min_val,
                                                            const float* __restrict__ max_val, float qrange, float qlo,
                                                            float qhi, int symmetric, double* __restrict__ part) {
  __shared__ float s_val[kBlock];
  __shared__ uint32_t s_cnt[kBlock];
  __shared__ uint32_t s_wave_n[kWavesPerBlock];
  __shared__ double s_acc[3][SBQ_MSE_CANDIDATES];
  // Keys are dealt to the workgroups in runs of 16 (64 bytes of counters: still whole sectors per quarter wave):
  // workgroup b owns runs b, b + 256, b + 512, ...  A tensor's values occupy a few contiguous stretches of the key
  // space; as 256 consecutive keys per workgroup a dozen workgroups had 256-entry lists and 240 had none.
  const uint32_t key = 16u * (blockIdx.x + kEvalBlocks * (threadIdx.x / 16u)) + (threadIdx.x & 15u);
  uint32_t c = 0;
#pragma unroll
  for (int k = 0; k < kHistCopies; ++k) c += ghist[static_cast<size_t>(k) * kHistKeys + key];
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  const uint64_t mask = __builtin_amdgcn_ballot_w64(c != 0u);
  if (lane == 0) s_wave_n[wid] = static_cast<uint32_t>(__builtin_popcountll(mask));
  __syncthreads();
  uint32_t base = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kWavesPerBlock; ++w) {
    base += w < wid ? s_wave_n[w] : 0u;
    total += s_wave_n[w];
  }
  if (c != 0u) {
    const uint32_t pos = base + static_cast<uint32_t>(__builtin_popcountll(mask & ((1ull << lane) - 1ull)));
    s_val[pos] = Key16<T>::value(key << 16);
    s_cnt[pos] = c;
  }
  __syncthreads();
  const uint32_t i = threadIdx.x % SBQ_MSE_CANDIDATES, r = threadIdx.x / SBQ_MSE_CANDIDATES;
  if (r < 3) {
    double acc = 0.0;
    if (total != 0u) {  // (uniform)
      float s, z;
      mse_candidate(min_val[0], max_val[0], static_cast<int>(i), qrange, symmetric != 0, s, z);
      const float y = fast_div_ok(s) ? 1.0f / s : 0.0f;
      for (uint32_t e = r; e < total; e += 3) {
        const float v = s_val[e];
        float d;
        // (mse_chunk_body's three forms of one element's error, operation for operation)
        if (y != 0.0f) {
          if (z == 0.0f) {
            const float lv = __builtin_amdgcn_fmed3f(__builtin_rintf(v * y), qlo, qhi);
            d = __builtin_fmaf(-lv, s, v);
          } else {
            const float lv = __builtin_amdgcn_fmed3f(__builtin_rintf(v * y) + z, qlo, qhi);
            d = __builtin_fmaf(-(lv - z), s, v);
          }
        } else {
          const float lv = quant_level<SBQ_ROUND_HALF_EVEN>(v, s, z, qlo, qhi);
          d = v - dequant_level(lv, s, z);
        }
        acc += static_cast<double>(s_cnt[e]) * (static_cast<double>(d) * static_cast<double>(d));
      }
    }
    s_acc[r][i] = acc;
  }
  __syncthreads();
  if (threadIdx.x < SBQ_MSE_CANDIDATES)
    part[static_cast<size_t>(blockIdx.x) * SBQ_MSE_CANDIDATES + threadIdx.x] =
        (s_acc[0][threadIdx.x] + s_acc[1][threadIdx.x]) + s_acc[2][threadIdx.x];
}
__global__ __launch_bounds__(kBlock) void mse16_fold_kernel(const double* __restrict__ part, double* __restrict__ sse) {
  // workgroup i folds candidate i: thread b fetches workgroup b's share (ONE load per thread -- as a serial walk of 85
  // dependent loads per thread this fold took 22 us), then a fixed-order tree in LDS
  static_assert(kEvalBlocks == kBlock, "one partial table per thread");
  __shared__ double s_d[kBlock];
  const uint32_t i = blockIdx.x;
  s_d[threadIdx.x] = part[static_cast<size_t>(threadIdx.x) * SBQ_MSE_CANDIDATES + i];
  __syncthreads();
#pragma unroll
  for (int w = kBlock / 2; w > 0; w >>= 1) {
    if (static_cast<int>(threadIdx.x) < w) s_d[threadIdx.x] += s_d[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) sse[i] += s_d[0];
}

// does sbq_mse_accumulate take the histogram route?  (per tensor, 16-bit, whole 16-byte packs from an aligned base,
// large enough that three small launches and a 2 MB clear beat the evaluation of every element; knob 2 == 19: never)
bool mse16_eligible(const void* x, int x_dtype, int64_t outer, int64_t C, int64_t inner) {
  const int64_t numel = outer * C * inner;
  return C == 1 && x_dtype != SBQ_F32 && numel >= (1ll << 22) && numel < (1ll << 31) && aligned16(x) && knob(2) != 19;
}
constexpr size_t kMse16Bytes = kHist16Bytes + static_cast<size_t>(kEvalBlocks) * SBQ_MSE_CANDIDATES * sizeof(double);

// Fold the per-chunk tables of a channel in a fixed order, kFoldFan chunks per workgroup
// and level: in[c][n_in][80] -> out[c][ceil(n_in / kFoldFan)][80]; the last level (one
// group left) ADDS into sse[c][80] instead.  Thread t owns candidate t % 80 and every third
// chunk of its group, so consecutive lanes read consecutive doubles.
constexpr uint32_t kFoldFan = 96;

__global__ __launch_bounds__(kBlock) void mse_fold_kernel(const double* __restrict__ in, uint32_t n_in,
                                                          double* __restrict__ out, uint32_t n_out,
                                                          int accumulate) {
  __shared__ double s_d[3][SBQ_MSE_CANDIDATES];
  const uint32_t c = blockIdx.y;
  const uint32_t grp = blockIdx.x;
  const uint32_t first = grp * kFoldFan;
  uint32_t last = first + kFoldFan;
  if (last > n_in) last = n_in;
  const double* p = in + static_cast<size_t>(c) * n_in * SBQ_MSE_CANDIDATES;
  const uint32_t i = threadIdx.x % SBQ_MSE_CANDIDATES;
  const uint32_t lane3 = threadIdx.x / SBQ_MSE_CANDIDATES;  // 0..3; 3 idles (256 = 3*80 + 16)
  if (lane3 < 3) {
    double t = 0.0;
    for (uint32_t j = first + lane3; j < last; j += 3) t += p[static_cast<size_t>(j) * SBQ_MSE_CANDIDATES + i];
    s_d[lane3][i] = t;
  }
  __syncthreads();
  if (threadIdx.x < SBQ_MSE_CANDIDATES) {
    const double t = (s_d[0][i] + s_d[1][i]) + s_d[2][i];
    double* o = out + (static_cast<size_t>(c) * n_out + grp) * SBQ_MSE_CANDIDATES + i;
    if (accumulate) *o += t;
    else *o = t;
  }
}

// workspace layout: level-0 tables, then the (geometrically shrinking) fold levels
size_t mse_workspace_doubles(const ChunkGeom& g) {
  size_t total = 0;
  uint32_t n = g.chunks_per_chan;
  while (true) {
    total += static_cast<size_t>(n) * g.C * SBQ_MSE_CANDIDATES;
    if (n <= 1) break;
    n = (n + kFoldFan - 1) / kFoldFan;
  }
  return total;
}

// mse.py:51-61: keep the first candidate whose fp32 loss is strictly smaller.
__global__ void mse_select_kernel(const double* __restrict__ sse, double count, const double* __restrict__ count_dev,
                                  const float* __restrict__ min_val, const float* __restrict__ max_val,
                                  int64_t C, float qrange, int symmetric, float* __restrict__ scale,
                                  float* __restrict__ zp, int32_t* __restrict__ best_index) {
  const int64_t c = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (count_dev) count = count_dev[0];  // sharded calibration: the count travelled in the all-reduced table's buffer
  float loss_min = 1e10f;
  int best = -1;
  for (int i = 0; i < SBQ_MSE_CANDIDATES; ++i) {
    const float loss = static_cast<float>(sse[c * SBQ_MSE_CANDIDATES + i] / count);
    if (loss < loss_min) {
      loss_min = loss;
      best = i;
    }
  }
  float s = 1.0f, z = 0.0f;  // mse.py:34-39 initial values
  if (best >= 0) mse_candidate(min_val[c], max_val[c], best, qrange, symmetric != 0, s, z);
  scale[c] = s;
  zp[c] = z;
  if (best_index) best_index[c] = best;
}

}  // namespace
}  // namespace sbq

extern "C" {

size_t sbq_stats_workspace_bytes(int64_t outer, int64_t C, int64_t inner) {
  using namespace sbq;
  if (!geom_ok(outer, C, inner, kStatsChunk)) return 0;
  const ChunkGeom g = make_geom(outer, C, inner, kStatsChunk);
  return static_cast<size_t>(g.chunks_per_chan) * g.C * sizeof(StatPartial);
}

int sbq_channel_stats(const void* x, int x_dtype, int64_t outer, int64_t C, int64_t inner,
                      float* min_out, float* max_out, double* abssum_out,
                      void* workspace, size_t workspace_bytes, void* stream) {
  using namespace sbq;
  if (!valid_dtype(x_dtype)) return SBQ_ERR_DTYPE;
  if (outer < 0 || C < 0 || inner < 0) return SBQ_ERR_ARG;
  if (outer == 0 || C == 0 || inner == 0) return SBQ_ERR_EMPTY;
  if (!x || !workspace) return SBQ_ERR_NULL;
  if (!geom_ok(outer, C, inner, kStatsChunk)) return SBQ_ERR_ARG;
  if (reinterpret_cast<uintptr_t>(x) % dtype_size(x_dtype)) return SBQ_ERR_ALIGN;
  const ChunkGeom g = make_geom(outer, C, inner, kStatsChunk);
  const size_t need = static_cast<size_t>(g.chunks_per_chan) * g.C * sizeof(StatPartial);
  if (workspace_bytes < need || !aligned16(workspace)) return SBQ_ERR_WORKSPACE;
  hipStream_t st = as_stream(stream);
  StatPartial* part = static_cast<StatPartial*>(workspace);
  if (inner == 1 && C % kPack == 0 && outer > 1 && aligned16(x)) {
    // channels-last: reduce [outer, C] over rows per column; `outer` partials per channel are
    // provisioned, at most that many row splits are used
    const uint32_t gx = static_cast<uint32_t>(ceil_div(C / kPack, kBlock));
    uint32_t splits = 2048u / gx;
    if (splits < 1) splits = 1;
    if (splits > static_cast<uint32_t>(outer)) splits = static_cast<uint32_t>(outer);
    if (splits > 65535u) splits = 65535u;
    int rc = dispatch_dtype(x_dtype, [&](auto tag) {
      using T = decltype(tag);
      stats_clast_kernel<T><<<dim3(gx, splits), kBlock, 0, st>>>(x, part, static_cast<uint32_t>(outer),
                                                               static_cast<uint32_t>(C), splits);
    });
    if (rc != SBQ_OK) return rc;
    rc = check_launch();
    if (rc != SBQ_OK) return rc;
    stats_finish_kernel<<<g.C, kBlock, 0, st>>>(part, splits, min_out, max_out, abssum_out);
    return check_launch();
  }
  const uint32_t n_chunks = g.chunks_per_chan * g.C;
  const uint32_t grid = (n_chunks + kWavesPerBlock - 1) / kWavesPerBlock;  // one wave per chunk
  const bool vec = pack_friendly(x, C, outer, inner);
  const bool final_ = g.chunks_per_chan == 1;
  int rc = dispatch_dtype(x_dtype, [&](auto tag) {
    using T = decltype(tag);
#define SBQ_STATS(V, F) \
  stats_partial_kernel<T, V, F><<<grid, kBlock, 0, st>>>(x, part, min_out, max_out, abssum_out, g, n_chunks)
    if (vec && !abssum_out && knob(2) != 11) {
      // the min-max observer: integer / minimum3 reductions (knob 2 == 11: the general kernel, for A/B runs)
      if (final_) stats_minmax_kernel<T, true><<<grid, kBlock, 0, st>>>(x, part, min_out, max_out, n_chunks, g.inner, g);
      else stats_minmax_kernel<T, false><<<grid, kBlock, 0, st>>>(x, part, min_out, max_out, n_chunks, g.inner, g);
    } else if (vec) { if (final_) SBQ_STATS(true, true); else SBQ_STATS(true, false); }
    else { if (final_) SBQ_STATS(false, true); else SBQ_STATS(false, false); }
#undef SBQ_STATS
  });
  if (rc != SBQ_OK) return rc;
  rc = check_launch();
  if (rc != SBQ_OK || final_) return rc;
  stats_finish_kernel<<<g.C, kBlock, 0, st>>>(part, g.chunks_per_chan, min_out, max_out, abssum_out);
  return check_launch();
}

int sbq_minmax_state_reset(uint32_t* state, void* stream) {
  using namespace sbq;
  if (!state) return SBQ_ERR_NULL;
  minmax_state_kernel<<<1, kWave, 0, as_stream(stream)>>>(state, nullptr, nullptr);
  return check_launch();
}

int sbq_minmax_accumulate(const void* x, int x_dtype, int64_t numel, uint32_t* state, void* stream) {
  using namespace sbq;
  if (!valid_dtype(x_dtype)) return SBQ_ERR_DTYPE;
  if (numel < 0) return SBQ_ERR_ARG;
  if (numel == 0) return SBQ_ERR_EMPTY;
  if (!x || !state) return SBQ_ERR_NULL;
  if (!aligned16(x)) return SBQ_ERR_ALIGN;  // (whole 16-byte packs: what sbq_channel_stats' fast kernel takes too)
  const int64_t chunks = ceil_div(numel, static_cast<int64_t>(kStatsChunk));
  if (chunks >= (1ll << 31)) return SBQ_ERR_ARG;
  const uint32_t grid = static_cast<uint32_t>(ceil_div(chunks, static_cast<int64_t>(kWavesPerBlock)));
  hipStream_t st = as_stream(stream);
  int rc = dispatch_dtype(x_dtype, [&](auto tag) {
    using T = decltype(tag);
    minmax_accumulate_kernel<T><<<grid, kBlock, 0, st>>>(x, numel, static_cast<uint32_t>(chunks), state);
  });
  if (rc != SBQ_OK) return rc;
  return check_launch();
}

int sbq_minmax_state_read(uint32_t* state, float* min_out, float* max_out, void* stream) {
  using namespace sbq;
  if (!state || !min_out || !max_out) return SBQ_ERR_NULL;
  minmax_state_kernel<<<1, kWave, 0, as_stream(stream)>>>(state, min_out, max_out);
  return check_launch();
}

int sbq_channel_moments(const void* x, int x_dtype, int64_t outer, int64_t C, int64_t inner,
                        const float* center, double* sum_out, double* sumsq_out, double* absdev_out,
                        void* workspace, size_t workspace_bytes, void* stream) {
  using namespace sbq;
  if (!valid_dtype(x_dtype)) return SBQ_ERR_DTYPE;
  if (outer < 0 || C < 0 || inner < 0) return SBQ_ERR_ARG;
  if (outer == 0 || C == 0 || inner == 0) return SBQ_ERR_EMPTY;
  if (!x || !workspace) return SBQ_ERR_NULL;
  if (center ? !absdev_out : (!sum_out && !sumsq_out)) return SBQ_ERR_NULL;
  if (!geom_ok(outer, C, inner, kStatsChunk)) return SBQ_ERR_ARG;
  if (reinterpret_cast<uintptr_t>(x) % dtype_size(x_dtype)) return SBQ_ERR_ALIGN;
  const ChunkGeom g = make_geom(outer, C, inner, kStatsChunk);
  static_assert(sizeof(MomentPartial) == sizeof(StatPartial), "shared workspace query");
  const size_t need = static_cast<size_t>(g.chunks_per_chan) * g.C * sizeof(MomentPartial);
  if (workspace_bytes < need || !aligned16(workspace)) return SBQ_ERR_WORKSPACE;
  hipStream_t st = as_stream(stream);
  MomentPartial* part = static_cast<MomentPartial*>(workspace);
  const uint32_t n_chunks = g.chunks_per_chan * g.C;
  const uint32_t grid = (n_chunks + kWavesPerBlock - 1) / kWavesPerBlock;
  const bool vec = pack_friendly(x, C, outer, inner);
  int rc = dispatch_dtype(x_dtype, [&](auto tag) {
    using T = decltype(tag);
    if (center) {
      if (vec) moments_partial_kernel<T, true, true><<<grid, kBlock, 0, st>>>(x, center, part, g, n_chunks);
      else moments_partial_kernel<T, false, true><<<grid, kBlock, 0, st>>>(x, center, part, g, n_chunks);
    } else {
      if (vec) moments_partial_kernel<T, true, false><<<grid, kBlock, 0, st>>>(x, center, part, g, n_chunks);
      else moments_partial_kernel<T, false, false><<<grid, kBlock, 0, st>>>(x, center, part, g, n_chunks);
    }
  });
  if (rc != SBQ_OK) return rc;
  rc = check_launch();
  if (rc != SBQ_OK) return rc;
  moments_finish_kernel<<<g.C, kBlock, 0, st>>>(part, g.chunks_per_chan, center ? absdev_out : sum_out,
                                                center ? nullptr : sumsq_out);
  return check_launch();
}

int sbq_aciq_thresholds(const float* min_val, const float* max_val, const float* b, int64_t C, float alpha,
                        float gaus_const, float sqrt_2logn, int half_range, float* min_out, float* max_out,
                        void* stream) {
  using namespace sbq;
  if (C < 0) return SBQ_ERR_ARG;
  if (C == 0) return SBQ_ERR_EMPTY;
  if (!min_out || !max_out || (!b && (!min_val || !max_val))) return SBQ_ERR_NULL;
  aciq_kernel<<<static_cast<uint32_t>(ceil_div(C, kBlock)), kBlock, 0, as_stream(stream)>>>(
      min_val, max_val, b, C, alpha, gaus_const, sqrt_2logn, half_range, min_out, max_out);
  return check_launch();
}

int sbq_minmax_pack(const float* min_val, const float* max_val, int64_t C, float* buf, void* stream) {
  using namespace sbq;
  if (C < 0) return SBQ_ERR_ARG;
  if (C == 0) return SBQ_ERR_EMPTY;
  if (!min_val || !max_val || !buf) return SBQ_ERR_NULL;
  minmax_pack_kernel<<<static_cast<uint32_t>(ceil_div(C, kBlock)), kBlock, 0, as_stream(stream)>>>(min_val, max_val, C, buf);
  return check_launch();
}

int sbq_minmax_unpack(const float* buf, int64_t C, float* min_out, float* max_out, void* stream) {
  using namespace sbq;
  if (C < 0) return SBQ_ERR_ARG;
  if (C == 0) return SBQ_ERR_EMPTY;
  if (!buf || !min_out || !max_out) return SBQ_ERR_NULL;
  minmax_unpack_kernel<<<static_cast<uint32_t>(ceil_div(C, kBlock)), kBlock, 0, as_stream(stream)>>>(buf, C, min_out, max_out);
  return check_launch();
}

int sbq_ema_minmax(const float* sample_min, const float* sample_max, int64_t n, float ratio,
                   float one_minus_ratio, float* state, int has_state, void* stream) {
  using namespace sbq;
  if (n < 0) return SBQ_ERR_ARG;
  if (n == 0) return SBQ_ERR_EMPTY;
  if (!sample_min || !sample_max || !state) return SBQ_ERR_NULL;
  ema_minmax_kernel<<<1, kWave, 0, as_stream(stream)>>>(sample_min, sample_max, n, ratio, one_minus_ratio, state,
                                                        has_state);
  return check_launch();
}

int sbq_qparams_from_minmax(const float* min_val, const float* max_val, int64_t C, int qmin, int qmax,
                            int symmetric, float* scale_out, float* zero_point_out, void* stream) {
  using namespace sbq;
  if (C < 0) return SBQ_ERR_ARG;
  if (C == 0) return SBQ_ERR_EMPTY;
  if (!min_val || !max_val || !scale_out || !zero_point_out) return SBQ_ERR_NULL;
  if (qmin >= qmax) return SBQ_ERR_ARG;
  const float qrange = static_cast<float>(qmax - qmin);
  qparams_kernel<<<static_cast<uint32_t>(ceil_div(C, kBlock)), kBlock, 0, as_stream(stream)>>>(
      min_val, max_val, C, qrange, symmetric, scale_out, zero_point_out);
  return check_launch();
}

int sbq_lsq_init_scale(const double* abssum, int64_t C, double count, int qmax, float* scale_out,
                       void* stream) {
  using namespace sbq;
  if (C < 0 || count <= 0 || qmax <= 0) return SBQ_ERR_ARG;
  if (C == 0) return SBQ_ERR_EMPTY;
  if (!abssum || !scale_out) return SBQ_ERR_NULL;
  const float sq = static_cast<float>(__builtin_sqrt(static_cast<double>(qmax)));
  lsq_init_kernel<<<static_cast<uint32_t>(ceil_div(C, kBlock)), kBlock, 0, as_stream(stream)>>>(
      abssum, C, count, sq, scale_out);
  return check_launch();
}

size_t sbq_mse_workspace_bytes(int64_t outer, int64_t C, int64_t inner) {
  using namespace sbq;
  if (!geom_ok(outer, C, inner, kMseChunk)) return 0;
  const ChunkGeom g = make_geom(outer, C, inner, kMseChunk);
  const size_t chunks = mse_workspace_doubles(g) * sizeof(double);
  // (a per-tensor call may take the histogram route of 16-bit inputs: 8 copies of 65 536 counters + 256 partial tables)
  return C == 1 && chunks < kMse16Bytes ? kMse16Bytes : chunks;
}

int sbq_mse_accumulate(const void* x, int x_dtype, int64_t outer, int64_t C, int64_t inner,
                       const float* min_val, const float* max_val, int qmin, int qmax, int symmetric,
                       double* sse, void* workspace, size_t workspace_bytes, void* stream) {
  using namespace sbq;
  if (!valid_dtype(x_dtype)) return SBQ_ERR_DTYPE;
  if (outer < 0 || C < 0 || inner < 0) return SBQ_ERR_ARG;
  if (outer == 0 || C == 0 || inner == 0) return SBQ_ERR_EMPTY;
  if (!x || !min_val || !max_val || !sse || !workspace) return SBQ_ERR_NULL;
  if (qmin >= qmax) return SBQ_ERR_ARG;
  if (!geom_ok(outer, C, inner, kMseChunk)) return SBQ_ERR_ARG;
  if (reinterpret_cast<uintptr_t>(x) % dtype_size(x_dtype)) return SBQ_ERR_ALIGN;
  const ChunkGeom g = make_geom(outer, C, inner, kMseChunk);
  const size_t need = mse_workspace_doubles(g) * sizeof(double);
  if (workspace_bytes < need || !aligned16(workspace)) return SBQ_ERR_WORKSPACE;
  if (g.C > 65535) return SBQ_ERR_ARG;  // fold grid.y
  hipStream_t st = as_stream(stream);
  double* part = static_cast<double*>(workspace);
  const uint32_t grid = g.chunks_per_chan * g.C;
  const bool vec = aligned16(x) && inner % kPack == 0;
  const float qrange = static_cast<float>(qmax - qmin);
  const float qlo = static_cast<float>(qmin), qhi = static_cast<float>(qmax);
  if (mse16_eligible(x, x_dtype, outer, C, inner) && workspace_bytes >= kMse16Bytes) {
    // one read of the tensor into an exact histogram of its 16-bit values, then 80 candidates x 65 536 values
    uint32_t* ghist = static_cast<uint32_t*>(workspace);
    double* part16 = reinterpret_cast<double*>(static_cast<char*>(workspace) + kHist16Bytes);
    if (hipMemsetAsync(ghist, 0, kHist16Bytes, st) != hipSuccess) return check_launch();
    const int64_t numel = outer * C * inner;
    const int64_t per_launch = static_cast<int64_t>(kH16Elems) * cu_count();
    int rc16 = SBQ_OK;
    for (int64_t done = 0; done < numel && rc16 == SBQ_OK;) {
      int64_t cnt = numel - done < per_launch ? numel - done : per_launch;
      // a rest shorter than one 8-element pack rides with this launch (one more workgroup's worth of slab indices, no
      // launch of its own: hist16_kernel reads whole packs and takes n % 8 elements as the ragged end of workgroup 0)
      if (numel - done - cnt < kPack) cnt = numel - done;
      const uint32_t wgs = static_cast<uint32_t>(ceil_div(cnt, static_cast<int64_t>(kH16Elems)));
      const void* xp = static_cast<const char*>(x) + done * 2;
      rc16 = dispatch_dtype(x_dtype, [&](auto tag) {
        using T = decltype(tag);
        if constexpr (T::id != SBQ_F32) {
          static bool once = [] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(hist16_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
            return true;
          }();
          (void)once;
          hist16_kernel<T><<<wgs, kH16Threads, 131072, st>>>(xp, static_cast<uint32_t>(cnt), ghist);
        }
      });
      done += cnt;
    }
    if (rc16 != SBQ_OK) return rc16;
    rc16 = dispatch_dtype(x_dtype, [&](auto tag) {
      using T = decltype(tag);
      if constexpr (T::id != SBQ_F32)
        mse16_eval_kernel<T><<<kEvalBlocks, kBlock, 0, st>>>(ghist, min_val, max_val, qrange, qlo, qhi, symmetric, part16);
    });
    if (rc16 != SBQ_OK) return rc16;
    mse16_fold_kernel<<<SBQ_MSE_CANDIDATES, kBlock, 0, st>>>(part16, sse);
    return check_launch();
  }
  int rc = dispatch_dtype(x_dtype, [&](auto tag) {
    using T = decltype(tag);
    if (vec)
      mse_partial_kernel<T, true><<<grid, kBlock, 0, st>>>(x, min_val, max_val, part, sse, g, qrange, qlo, qhi, symmetric);
    else
      mse_partial_kernel<T, false><<<grid, kBlock, 0, st>>>(x, min_val, max_val, part, sse, g, qrange, qlo, qhi, symmetric);
  });
  if (rc != SBQ_OK) return rc;
  rc = check_launch();
  if (rc != SBQ_OK || g.chunks_per_chan == 1) return rc;
  const double* in = part;
  uint32_t n_in = g.chunks_per_chan;
  while (true) {
    const uint32_t n_out = (n_in + kFoldFan - 1) / kFoldFan;
    const bool last = n_out == 1;
    double* out = last ? sse : const_cast<double*>(in) + static_cast<size_t>(n_in) * g.C * SBQ_MSE_CANDIDATES;
    mse_fold_kernel<<<dim3(n_out, g.C), kBlock, 0, st>>>(in, n_in, out, n_out, last ? 1 : 0);
    rc = check_launch();
    if (rc != SBQ_OK || last) return rc;
    in = out;
    n_in = n_out;
  }
}

int sbq_mse_select(const double* sse, double count_per_channel, const float* min_val,
                   const float* max_val, int64_t C, int qmin, int qmax, int symmetric,
                   float* scale_out, float* zero_point_out, int32_t* best_index_out, void* stream) {
  using namespace sbq;
  if (C < 0 || count_per_channel <= 0) return SBQ_ERR_ARG;
  if (C == 0) return SBQ_ERR_EMPTY;
  if (!sse || !min_val || !max_val || !scale_out || !zero_point_out) return SBQ_ERR_NULL;
  if (qmin >= qmax) return SBQ_ERR_ARG;
  const float qrange = static_cast<float>(qmax - qmin);
  mse_select_kernel<<<static_cast<uint32_t>(ceil_div(C, kWave)), kWave, 0, as_stream(stream)>>>(
      sse, count_per_channel, nullptr, min_val, max_val, C, qrange, symmetric, scale_out, zero_point_out,
      best_index_out);
  return check_launch();
}

int sbq_mse_select_devcount(const double* sse, const double* count_per_channel_dev, const float* min_val,
                            const float* max_val, int64_t C, int qmin, int qmax, int symmetric,
                            float* scale_out, float* zero_point_out, int32_t* best_index_out, void* stream) {
  using namespace sbq;
  if (C < 0) return SBQ_ERR_ARG;
  if (C == 0) return SBQ_ERR_EMPTY;
  if (!sse || !count_per_channel_dev || !min_val || !max_val || !scale_out || !zero_point_out) return SBQ_ERR_NULL;
  if (qmin >= qmax) return SBQ_ERR_ARG;
  const float qrange = static_cast<float>(qmax - qmin);
  mse_select_kernel<<<static_cast<uint32_t>(ceil_div(C, kWave)), kWave, 0, as_stream(stream)>>>(
      sse, 1.0, count_per_channel_dev, min_val, max_val, C, qrange, symmetric, scale_out, zero_point_out,
      best_index_out);
  return check_launch();
}

}  // extern "C"
