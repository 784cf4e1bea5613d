// sbq_select_win.hip -- exact per-tensor order statistics in (typically) ONE sweep of the data: the percentile
// observer's k-th values over the cached calibration batches and the unstructured-mask threshold.
//
// Replaces, for a tensor selected as a whole (C == 1) on one device,
//   sparsebit/quantization/observers/percentile.py:16-46  (two torch.kthvalue calls on the concatenated data),
//   sparsebit/sparse/sparsers/l1norm.py:18-26             (a full torch.sort for one order statistic).
//
// The fixed-digit radix engine of sbq_select.hip sweeps the data three times and its first sweep is bound by
// LDS atomics: the top key bits of real tensors fall into a handful of bins (sign + exponent), so every element
// of a wave updates one of a few addresses.  Here the first window is chosen from a SAMPLE instead:
//   1. plan:   one workgroup reads 2048 strided packs of 8 elements, histograms the top 13 key bits in LDS and
//      brackets each wanted rank between two sample quantiles (+- 6 sigma of the rank error of a clustered
//      sample): a key window holding a few per cent of the data;
//   2. pass:   ONE sweep counts, per selector, the keys below the window in registers and histograms the keys
//      inside it (2048 LDS bins: only the few elements inside the window touch LDS);
//   3. advance: if the rank is inside the window (it is, unless the sample lied) the bin that holds it is the
//      next window, 2048 times narrower; a 16-bit input is resolved after one sweep, fp32 after two.  Otherwise
//      the window becomes everything below / above the first one and the protocol simply continues -- exact for
//      any data, just with more sweeps (the launches of the later rounds are enqueued anyway and exit at once
//      when every selector is done).
// Cross-workgroup accumulation: a window a few keys wide means every workgroup adds to the SAME few bins, and
// same-line device atomics serialise (~3 ns each: 512 workgroups x 18 bins cost 20 us -- measured, it was the
// whole kernel).  So the workgroups spread over kCopies copies of the histogram and 64 lines of counters; the
// advance kernel sums them.
// Keys, NaN / -0 handling, rank formulas: exactly those of sbq_select.hip (float_key; percentile_ranks_kernel).
// Histograms are integer counts (order independent => deterministic).
#include "sbq_common.hpp"

namespace sbq {
namespace {

constexpr int kWinBins = 2048;  // histogram bins per selector
constexpr int kWinLog = 11;
constexpr int kWinSel = 2;      // selectors (percentile: min side, max side)
constexpr int kCopies = 8;      // copies of the global histogram (workgroup b adds to copy b % kCopies)
constexpr int kSlots = 64;      // counter lines (workgroup b adds to line b % kSlots)
constexpr int kPlanBins = 8192;  // plan: top 13 key bits (32 KB of LDS)
constexpr int kPlanShift = 19;
constexpr int kPlanPacks = 2048;  // sampled packs of 8 consecutive elements
constexpr int kMaxShards = 64;
constexpr int kAdvBlock = 512;

struct WinSel {
  uint32_t lo;     // first key of the window
  uint32_t shift;  // bin = (key - lo) >> shift, kWinBins bins
  int64_t k;       // rank (1-based): absolute while `fresh`, relative to the window afterwards
  uint32_t done;   // key `lo` is the answer
  uint32_t fresh;  // window came from the sample: the sweep also counts the keys below it
};
struct WinState {
  WinSel sel[kWinSel];
  int64_t n;  // elements in all shards
  long long pad[3];
};
struct WinSlot {  // one 128-byte line
  unsigned long long below[kWinSel];
  unsigned long long neg, nan;
  unsigned long long pad[12];
};
struct ShardTable {
  const void* ptr[kMaxShards];
  int64_t count[kMaxShards];
};

// Order-preserving key of this engine: the usual sign transform, then rotated down by 2^23 so that the keys of
// NEGATIVE NaNs (which the transform puts first) wrap around to the top, above +inf and the positive NaNs: every
// NaN sorts last, as in torch.sort / kthvalue, without a per-element NaN test.  Three integer operations.
// (-0 keeps its own key just below +0: equal values, adjacent keys.)
constexpr uint32_t kRot = 0x007fffffu;      // key(-inf): -inf becomes key 0, the negative NaNs below it wrap to the top
constexpr uint32_t kKeyZero = 0x7f800000u;            // key(-0): keys below this are x < 0 (NaNs excluded)
constexpr uint32_t kKeyInf = 0xff800000u - kRot;      // key(+inf): keys above this are NaN
__device__ __forceinline__ uint32_t win_key(uint32_t bits, bool use_abs) {
  if (use_abs) return ((bits & 0x7fffffffu) | 0x80000000u) - kRot;
  const uint32_t m = static_cast<uint32_t>(static_cast<int32_t>(bits) >> 31) | 0x80000000u;
  return (bits ^ m) - kRot;
}
__device__ __forceinline__ float win_key_float(uint32_t k) {
  k += kRot;
  const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __builtin_bit_cast(float, u);
}

struct SumL { __device__ __forceinline__ unsigned long long operator()(unsigned long long a, unsigned long long b) const { return a + b; } };

__device__ __forceinline__ uint32_t shift_for(uint64_t width, uint32_t min_shift) {
  // smallest shift >= min_shift with ceil(width / 2^shift) <= kWinBins  (width in keys, up to 2^32)
  uint32_t s = min_shift;
  while (s < 32 && ((width + ((1ull << s) - 1)) >> s) > static_cast<uint64_t>(kWinBins)) ++s;
  return s;
}

__global__ __launch_bounds__(kBlock) void win_init_kernel(int64_t* __restrict__ base, size_t words) {
  for (size_t i = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x; i < words;
       i += static_cast<size_t>(gridDim.x) * kBlock)
    base[i] = 0;
}

// One workgroup of 1024: sample, histogram in LDS, bracket each selector's rank, write the first windows.
// mode 0: explicit ranks k0 (k1); mode 1: percentile (ranks from the sample's own sign counts; the exact ones
// follow from the first sweep).
template <typename T>
__global__ __launch_bounds__(1024) void win_plan_kernel(const ShardTable tab, int n_shards, WinState* __restrict__ st,
                                                        int mode, int n_sel, int use_abs, int64_t k0, int64_t k1,
                                                        int64_t n, double alpha, uint32_t min_shift) {
  constexpr int kT = 1024, kPer = kPlanBins / kT;  // 8 bins per thread
  __shared__ uint32_t hist[kPlanBins];
  __shared__ uint32_t wave_tot[kT / kWave];
  __shared__ uint32_t s_total, s_neg, s_first, s_last;
  __shared__ int64_t r_lo[kWinSel], r_hi[kWinSel];
  __shared__ double r_mid[kWinSel];
  __shared__ uint32_t b_lo[kWinSel], b_hi[kWinSel];
  for (int i = threadIdx.x; i < kPlanBins; i += kT) hist[i] = 0;
  if (threadIdx.x == 0) {
    s_first = kPlanBins - 1;
    s_last = 0;
  }
  __syncthreads();
  // sample: pack p of n_packs starts at element p * floor(n / n_packs) of the concatenated shards
  const int64_t n_packs = n / kPack < kPlanPacks ? (n / kPack > 0 ? n / kPack : 1) : kPlanPacks;
  for (int64_t p = threadIdx.x; p < n_packs; p += kT) {
    // which shard: the table lives in the kernel arguments, so it is walked with a UNIFORM index (scalar loads) and
    // the lane keeps its own pointer / count by selects -- a per-lane index would spill the table to scratch
    int64_t e = p * (n / n_packs);
    const void* base = tab.ptr[0];
    int64_t cnt = tab.count[0];
    bool found = false;
    for (int i = 0; i < n_shards; ++i) {
      const int64_t c = tab.count[i];
      const bool here = !found && (e < c || i + 1 == n_shards);
      base = here ? tab.ptr[i] : base;
      cnt = here ? c : cnt;
      e = (found || here) ? e : e - c;
      found |= here;
    }
    float v[kPack];
    e &= ~static_cast<int64_t>(kPack - 1);  // whole packs: one 16-byte load (two for fp32) when the shard allows it
    if ((reinterpret_cast<uintptr_t>(base) & 15u) == 0 && e + kPack <= cnt) {
      load_pack<T, false>(base, e, v);
    } else {
#pragma unroll
      for (int j = 0; j < kPack; ++j) v[j] = Elem<T>::load1(base, e + j < cnt ? e + j : cnt - 1);
    }
#pragma unroll
    for (int j = 0; j < kPack; ++j)
      if (e + j < cnt) atomicAdd(&hist[win_key(__builtin_bit_cast(uint32_t, v[j]), use_abs != 0) >> kPlanShift], 1u);
  }
  __syncthreads();
  uint32_t bins[kPer];
  uint32_t t = 0;
#pragma unroll
  for (int i = 0; i < kPer; ++i) {
    bins[i] = hist[threadIdx.x * kPer + i];
    t += bins[i];
  }
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  uint32_t incl = t;
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const uint32_t up = __shfl_up(incl, d, kWave);
    if (lane >= d) incl += up;
  }
  if (lane == kWave - 1) wave_tot[wid] = incl;
  __syncthreads();
  uint32_t off = 0;
  for (int w = 0; w < wid; ++w) off += wave_tot[w];
  incl += off;
  const uint32_t excl = incl - t;
  if (threadIdx.x == kT - 1) s_total = incl;
  if (threadIdx.x == (kKeyZero >> kPlanShift) / kPer) s_neg = excl;  // bins below kKeyZero: keys of x < 0
  if (t) {                                  // first / last occupied bin of the sample
    uint32_t f = 0, l = 0;
    for (int i = 0; i < kPer; ++i)
      if (bins[i]) { f = i; break; }
    for (int i = kPer - 1; i >= 0; --i)
      if (bins[i]) { l = i; break; }
    atomicMin(&s_first, threadIdx.x * kPer + f);
    atomicMax(&s_last, threadIdx.x * kPer + l);
  }
  if (threadIdx.x < kWinSel) {
    b_lo[threadIdx.x] = 0;
    b_hi[threadIdx.x] = kPlanBins - 1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double S = static_cast<double>(s_total);
    const double scale = n > 0 ? S / static_cast<double>(n) : 0.0;
    for (int s = 0; s < n_sel; ++s) {
      double r;  // expected rank of the target inside the sample (1-based, fractional)
      if (mode == 0) {
        r = static_cast<double>(s == 0 ? k0 : k1) * scale;
      } else {
        // percentile.py:36-43 on the sample's own counts (the last bin holds the NaNs, and nothing else that
        // matters: +inf and the largest finite values share it)
        const double neg = static_cast<double>(s_neg);
        const double pos = S - neg;
        r = s == 0 ? __builtin_fmax(neg * alpha, 1.0 * scale) : S - pos * alpha;
      }
      // rank error of the sample: sigma_iid = sqrt(r (1 - r/S)) for independent draws; the 8 neighbours of a pack
      // are correlated (design effect 1 + 7 rho), so twice 6 sigma_iid + slack.  Only the cost of a miss (one more
      // round) depends on this, never the result.
      const double q = S > 0 ? r / S : 0.0;
      const double var = __builtin_fmax(r * (1.0 - (q < 1.0 ? q : 1.0)), 1.0);
      const double m = 2.0 * 6.0 * __builtin_sqrt(var) + 16.0;
      r_mid[s] = r;
      r_lo[s] = static_cast<int64_t>(__builtin_floor(r - m));
      r_hi[s] = static_cast<int64_t>(__builtin_ceil(r + m));
    }
  }
  __syncthreads();
  // the thread whose bins hold sample rank r (excl < r <= incl) names the bin
  for (int s = 0; s < n_sel; ++s) {
    for (int side = 0; side < 2; ++side) {
      const int64_t r = side == 0 ? r_lo[s] : r_hi[s];
      if (r >= 1 && r > static_cast<int64_t>(excl) && r <= static_cast<int64_t>(incl)) {
        int64_t kk = r - excl;
        uint32_t b = 0;
        for (int i = 0; i < kPer; ++i) {
          if (kk > static_cast<int64_t>(bins[i])) kk -= bins[i];
          else { b = i; break; }
        }
        (side == 0 ? b_lo : b_hi)[s] = threadIdx.x * kPer + b;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    st->n = n;
    const double S = static_cast<double>(s_total);
    for (int s = 0; s < n_sel; ++s) {
      // A bracket that runs off the sample: the window starts at the first key instead -- or, when the target is
      // at least 8 sample ranks away from that end (the sample's own extreme is then beyond it with probability
      // 1 - e^-8), at the sample's extreme bin, which keeps a tail quantile's window a few bins wide.  Keys outside
      // the window are counted, so a wrong guess only costs another round.
      uint32_t a = b_lo[s], b = b_hi[s];
      if (r_lo[s] < 1) a = r_mid[s] >= 8.0 ? s_first : 0u;
      if (r_hi[s] > static_cast<int64_t>(s_total)) b = S - r_mid[s] >= 8.0 ? s_last : kPlanBins - 1;
      if (b < a) b = a;
      const uint32_t lo = a << kPlanShift;
      const uint64_t width = (static_cast<uint64_t>(b - a) + 1) << kPlanShift;
      WinSel w;
      w.lo = lo;
      w.shift = shift_for(width, min_shift);
      w.k = mode == 0 ? (s == 0 ? k0 : k1) : 0;
      w.done = 0;
      w.fresh = 1;
      st->sel[s] = w;
    }
  }
}

// The sweep.  A workgroup walks slabs of 8 Ki elements (grid-stride), so the LDS histograms are cleared and flushed
// once per workgroup.  Per element and selector: one subtraction and one unsigned compare decide "inside the
// window"; a fresh window also counts the keys below it (registers).  Only elements inside a window touch LDS,
// and a wave-wide vote per element skips the LDS instruction when no lane has one.
constexpr uint32_t kWinSlab = kBlock * kPack * 4;

template <typename T, bool VEC, int NSEL, bool SIGNS>
__global__ __launch_bounds__(kBlock) void win_pass_kernel(const void* __restrict__ x, int64_t n,
                                                          const WinState* __restrict__ st, WinSlot* __restrict__ slots,
                                                          uint32_t* __restrict__ hist, int use_abs) {
  __shared__ uint32_t lh[NSEL][kWinBins];
  __shared__ unsigned long long red[kWavesPerBlock];
  // every selector resolved: nothing to do (the later rounds of a protocol that needed only one)
  bool live = false;
  uint32_t lo[NSEL], sh[NSEL], wm1[NSEL];
  bool act[NSEL], fresh[NSEL];
#pragma unroll
  for (int s = 0; s < NSEL; ++s) {
    act[s] = st->sel[s].done == 0;
    lo[s] = st->sel[s].lo;
    sh[s] = st->sel[s].shift;
    // last in-window offset, cut at the last key: then `key - lo <= wm1` (unsigned) is the whole window test,
    // keys below lo wrap to offsets beyond it.  A finished selector gets an empty window (offset test never true).
    uint64_t w = (static_cast<uint64_t>(kWinBins) << sh[s]) - 1;
    const uint64_t room = 0xffffffffull - lo[s];
    if (w > room) w = room;
    wm1[s] = static_cast<uint32_t>(w);
    if (!act[s]) {
      lo[s] = 0xffffffffu;
      wm1[s] = 0;  // only key 0xffffffff would pass; `act` masks it below
    }
    fresh[s] = act[s] && st->sel[s].fresh != 0;
    live |= act[s];
  }
  if (!live) return;
  for (uint32_t i = threadIdx.x; i < NSEL * kWinBins; i += kBlock) (&lh[0][0])[i] = 0;
  __syncthreads();
  uint32_t lt[NSEL];
#pragma unroll
  for (int s = 0; s < NSEL; ++s) lt[s] = 0;
  uint32_t neg = 0, nan = 0;
  const bool ab = use_abs != 0;
  auto visit = [&](uint32_t kk, bool valid) {
    if constexpr (SIGNS) {
      neg += valid && kk < kKeyZero;
      nan += valid && kk > kKeyInf;
    }
#pragma unroll
    for (int s = 0; s < NSEL; ++s) {
      const uint32_t d = kk - lo[s];
      lt[s] += valid && kk < lo[s];
      const bool in = valid && act[s] && d <= wm1[s];
      if (__builtin_amdgcn_ballot_w64(in) != 0) {
        if (in) atomicAdd(&lh[s][d >> sh[s]], 1u);
      }
    }
  };
  const int64_t n_slabs = (n + kWinSlab - 1) / kWinSlab;
  for (int64_t slab = blockIdx.x; slab < n_slabs; slab += gridDim.x) {
    const int64_t begin = slab * kWinSlab;
    const int64_t end = begin + kWinSlab < n ? begin + kWinSlab : n;
    if constexpr (VEC) {
      const int64_t vend = begin + ((end - begin) / kPack) * kPack;
      constexpr int U = kWinSlab / (kBlock * kPack);
      if (vend > begin) {
        RawPack<T> raw[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          int64_t e = begin + (static_cast<int64_t>(u) * kBlock + threadIdx.x) * kPack;
          ok[u] = e < vend;
          if (!ok[u]) e = vend - kPack;
          raw[u] = load_raw<T, true>(x, e);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if constexpr (T::id == SBQ_F32) {
#pragma unroll
            for (int q = 0; q < 4; ++q) visit(win_key(raw[u].d[0][q], ab), ok[u]);
#pragma unroll
            for (int q = 0; q < 4; ++q) visit(win_key(raw[u].d[1][q], ab), ok[u]);
          } else if constexpr (T::id == SBQ_BF16) {
            // bf16 -> fp32 bits is a shift / a mask: no conversion
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const uint32_t w = raw[u].d[0][q];
              visit(win_key(w << 16, ab), ok[u]);
              visit(win_key(w & 0xffff0000u, ab), ok[u]);
            }
          } else {
            float v[kPack];
            unpack_raw<T>(raw[u], v);
#pragma unroll
            for (int q = 0; q < kPack; ++q) visit(win_key(__builtin_bit_cast(uint32_t, v[q]), ab), ok[u]);
          }
        }
      }
      for (int64_t e = vend + threadIdx.x; e < end; e += kBlock)
        visit(win_key(__builtin_bit_cast(uint32_t, Elem<T>::load1(x, e)), ab), true);
    } else {
      for (int64_t e = begin + threadIdx.x; e < end; e += kBlock)
        visit(win_key(__builtin_bit_cast(uint32_t, Elem<T>::load1(x, e)), ab), true);
    }
  }
  __syncthreads();
  WinSlot* slot = slots + (blockIdx.x % kSlots);
#pragma unroll
  for (int s = 0; s < NSEL; ++s) {
    if (!act[s]) continue;
    uint32_t* gh = hist + (static_cast<size_t>(blockIdx.x % kCopies) * kWinSel + s) * kWinBins;
    for (uint32_t i = threadIdx.x; i < static_cast<uint32_t>(kWinBins); i += kBlock) {
      const uint32_t v = lh[s][i];
      if (v) atomicAdd(&gh[i], v);
    }
    if (fresh[s]) {
      const unsigned long long b = block_reduce(static_cast<unsigned long long>(lt[s]), SumL(), red);
      if (threadIdx.x == 0 && b) atomicAdd(&slot->below[s], b);
    }
  }
  if constexpr (SIGNS) {
    const unsigned long long a = block_reduce(static_cast<unsigned long long>(neg), SumL(), red);
    const unsigned long long b = block_reduce(static_cast<unsigned long long>(nan), SumL(), red);
    if (threadIdx.x == 0) {
      if (a) atomicAdd(&slot->neg, a);
      if (b) atomicAdd(&slot->nan, b);
    }
  }
}

// One workgroup of 1024 per selector: sum the copies, place the rank, write the result when it is final.
// Leaves the selector's histogram copies and its `below` counters zeroed.
__global__ __launch_bounds__(kAdvBlock) void win_advance_kernel(uint32_t* __restrict__ hist,
                                                                WinState* __restrict__ st, WinSlot* __restrict__ slots,
                                                                int percentile, double alpha, uint32_t min_shift,
                                                                float* __restrict__ out0, float* __restrict__ out1) {
  __shared__ unsigned long long wave_tot[kAdvBlock / kWave];
  __shared__ unsigned long long s_total, s_below, s_neg, s_nan;
  const int s = blockIdx.x;
  WinSel w = st->sel[s];
  if (w.done) return;
  constexpr int kPer = kWinBins / kAdvBlock;  // 4 bins per thread: one 16-byte load per copy
  unsigned long long bins[kPer] = {0, 0, 0, 0};
  for (int c = 0; c < kCopies; ++c) {
    u32x4* p = reinterpret_cast<u32x4*>(hist + (static_cast<size_t>(c) * kWinSel + s) * kWinBins) + threadIdx.x;
    const u32x4 v = *p;
    *p = u32x4{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < kPer; ++i) bins[i] += v[i];
  }
  unsigned long long t = 0;
#pragma unroll
  for (int i = 0; i < kPer; ++i) t += bins[i];
  // counters: thread i < kSlots reads line i
  unsigned long long c_below = 0, c_neg = 0, c_nan = 0;
  if (threadIdx.x < kSlots) {
    c_below = slots[threadIdx.x].below[s];
    c_neg = slots[threadIdx.x].neg;
    c_nan = slots[threadIdx.x].nan;
    slots[threadIdx.x].below[s] = 0;
  }
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  unsigned long long incl = t;
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const unsigned long long up = __shfl_up(incl, d, kWave);
    if (lane >= d) incl += up;
  }
  if (lane == kWave - 1) wave_tot[wid] = incl;
  if (wid == 0) {  // the 64 counter lines live in wave 0
    c_below = wave_reduce(c_below, SumL());
    c_neg = wave_reduce(c_neg, SumL());
    c_nan = wave_reduce(c_nan, SumL());
    if (lane == 0) {
      s_below = c_below;
      s_neg = c_neg;
      s_nan = c_nan;
    }
  }
  __syncthreads();
  unsigned long long off = 0;
  for (int v = 0; v < wid; ++v) off += wave_tot[v];
  incl += off;
  const unsigned long long excl = incl - t;
  if (threadIdx.x == kAdvBlock - 1) s_total = incl;
  __syncthreads();
  const unsigned long long total = s_total;
  const int64_t n = st->n;
  const int64_t neg = static_cast<int64_t>(s_neg), nan = static_cast<int64_t>(s_nan);
  const int64_t pos = n - neg - nan;
  int64_t k = w.k;
  if (w.fresh) {
    if (percentile) {
      // percentile.py:36-43 with the exact counts of the first sweep (Python round == rint on a double)
      if (s == 0) k = static_cast<int64_t>(__builtin_fmax(__builtin_rint(static_cast<double>(neg) * alpha), 1.0));
      else k = n - static_cast<int64_t>(__builtin_fmax(__builtin_rint(static_cast<double>(pos) * alpha), 0.0));
      k = k < 1 ? 1 : (k > n ? n : k);
    }
    const unsigned long long below = s_below;
    const uint64_t hi = static_cast<uint64_t>(w.lo) + (static_cast<uint64_t>(kWinBins) << w.shift);  // exclusive
    if (static_cast<unsigned long long>(k) <= below) {
      // the sample lied: the rank is below the window.  New window: every key below it.
      if (threadIdx.x == 0) {
        WinSel nw = w;
        nw.lo = 0;
        nw.shift = shift_for(w.lo, min_shift);
        nw.k = k;
        nw.fresh = 0;
        st->sel[s] = nw;
      }
      return;
    }
    if (static_cast<unsigned long long>(k) > below + total) {
      // ... or above it.  New window: every key from its end on (hi < 2^32 here: a window reaching the last key
      // holds every element that is not below it)
      if (threadIdx.x == 0) {
        WinSel nw = w;
        nw.lo = static_cast<uint32_t>(hi);
        nw.shift = shift_for((1ull << 32) - hi, min_shift);
        nw.k = k - static_cast<int64_t>(below + total);
        nw.fresh = 0;
        st->sel[s] = nw;
      }
      return;
    }
    k -= static_cast<int64_t>(below);
  }
  // the rank lies in (excl, incl] of exactly one thread's bins
  const unsigned long long uk = static_cast<unsigned long long>(k);
  if (uk > excl && uk <= incl) {
    unsigned long long kk = uk - excl;
    int b = 0;
#pragma unroll
    for (int i = 0; i < kPer - 1; ++i) {
      if (b == i && kk > bins[i]) {
        kk -= bins[i];
        ++b;
      }
    }
    WinSel nw = w;
    nw.lo = w.lo + (static_cast<uint32_t>(threadIdx.x * kPer + b) << w.shift);
    nw.k = static_cast<int64_t>(kk);
    nw.fresh = 0;
    if (w.shift <= min_shift) {
      nw.done = 1;  // a bin is one representable value: of a 16-bit input's 2^min_shift keys in it, the real one has
      // low bits 0 for x < 0 (~bits ends in ones, minus the rotation) and 1 for x >= 0 (zeros minus the rotation)
      if (min_shift > 0 && ((nw.lo + kRot) & 0x80000000u)) nw.lo |= 1u;
      if (percentile) {
        // percentile.py:30-43: without negative (non-negative) elements min (max) stays 0
        if (s == 0) out0[0] = neg > 0 ? win_key_float(nw.lo) : 0.0f;
        else out1[0] = pos > 0 ? win_key_float(nw.lo) : 0.0f;
      } else {
        out0[s] = win_key_float(nw.lo);
      }
    } else {
      nw.shift = w.shift > min_shift + kWinLog ? w.shift - kWinLog : min_shift;
    }
    st->sel[s] = nw;
  }
}

constexpr size_t kStateBytes = 256;
constexpr size_t kSlotBytes = sizeof(WinSlot) * kSlots;
constexpr size_t kHistBytes = static_cast<size_t>(kCopies) * kWinSel * kWinBins * 4;

}  // namespace

size_t win_select_workspace_bytes() { return kStateBytes + kSlotBytes + kHistBytes + 256; }

// shards: flat tensors of counts[i] elements each (a per-tensor selection over the cached batches).
int win_select_run(const void* const* shards, const int64_t* counts, int n_shards, int x_dtype, int use_abs, int n_sel,
                   bool percentile, double alpha, int64_t k0, int64_t k1, float* out0, float* out1, void* workspace,
                   size_t workspace_bytes, hipStream_t st) {
  static_assert(sizeof(WinState) <= kStateBytes && sizeof(WinSlot) == 128, "workspace layout");
  if (workspace_bytes < win_select_workspace_bytes() || !aligned16(workspace)) return SBQ_ERR_WORKSPACE;
  if (n_shards > kMaxShards) return SBQ_ERR_ARG;
  char* ws = static_cast<char*>(workspace);
  WinState* state = reinterpret_cast<WinState*>(ws);
  WinSlot* slots = reinterpret_cast<WinSlot*>(ws + kStateBytes);
  uint32_t* hist = reinterpret_cast<uint32_t*>(ws + kStateBytes + kSlotBytes);
  ShardTable tab{};
  int64_t n = 0;
  for (int i = 0; i < n_shards; ++i) {
    tab.ptr[i] = shards[i];
    tab.count[i] = counts[i];
    if (counts[i] >= (1ll << 32)) return SBQ_ERR_ARG;  // 32-bit per-workgroup and histogram-copy counters
    n += counts[i];
  }
  const uint32_t min_shift = x_dtype == SBQ_BF16 ? 16u : (x_dtype == SBQ_F16 ? 13u : 0u);
  win_init_kernel<<<64, kBlock, 0, st>>>(reinterpret_cast<int64_t*>(ws), (kStateBytes + kSlotBytes + kHistBytes) / 8);
  int rc = dispatch_dtype(x_dtype, [&](auto tag) {
    using T = decltype(tag);
    win_plan_kernel<T><<<1, 1024, 0, st>>>(tab, n_shards, state, percentile ? 1 : 0, n_sel, use_abs, k0, k1, n, alpha,
                                           min_shift);
  });
  if (rc != SBQ_OK) return rc;
  // rounds: one resolves a 16-bit input, two an fp32 one -- when the first window holds the rank; a missed window
  // costs up to ceil((32 - min_shift) / 11) more.  All are enqueued; rounds after the last needed one exit at once.
  const int rounds = 1 + static_cast<int>((32 - min_shift + kWinLog - 1) / kWinLog);
  for (int r = 0; r < rounds; ++r) {
    for (int i = 0; i < n_shards && rc == SBQ_OK; ++i) {
      const bool vec = aligned16(shards[i]);
      const int64_t slabs = ceil_div(counts[i], static_cast<int64_t>(kWinSlab));
      const uint32_t grid = static_cast<uint32_t>(slabs < 1024 ? slabs : 1024);
      const bool signs = r == 0 && percentile;
      rc = dispatch_dtype(x_dtype, [&](auto tag) {
        using T = decltype(tag);
#define SBQ_WIN(V, NS, SG) \
  win_pass_kernel<T, V, NS, SG><<<grid, kBlock, 0, st>>>(shards[i], counts[i], state, slots, hist, use_abs)
        if (n_sel == 1) {
          if (vec) SBQ_WIN(true, 1, false);
          else SBQ_WIN(false, 1, false);
        } else if (signs) {
          if (vec) SBQ_WIN(true, 2, true);
          else SBQ_WIN(false, 2, true);
        } else {
          if (vec) SBQ_WIN(true, 2, false);
          else SBQ_WIN(false, 2, false);
        }
#undef SBQ_WIN
      });
    }
    if (rc != SBQ_OK) return rc;
    // the advance of the round that resolves a selector also writes its result
    win_advance_kernel<<<n_sel, kAdvBlock, 0, st>>>(hist, state, slots, percentile ? 1 : 0, alpha, min_shift, out0,
                                                    out1);
  }
  return check_launch();
}

}  // namespace sbq
