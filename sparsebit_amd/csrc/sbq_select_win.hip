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
  uint32_t span;   // last in-window offset: the window is [lo, lo + span], at most kWinBins << shift keys
  uint32_t side;   // what a fresh window's sweep counts besides the histogram: 0 = the keys below it, 1 = above it
  int64_t k;       // rank (1-based): absolute while `fresh`, relative to the window afterwards
  uint32_t done;   // key `lo` is the answer
  uint32_t fresh;  // window came from the sample: the sweep also counts the keys below it
};
struct WinState {
  WinSel sel[kWinSel];
  int64_t n;  // elements in all shards
  uint32_t arrivals;  // win_fallback_kernel: workgroups of the running sweep that have flushed (zero between launches)
  uint32_t pad1;
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

// One workgroup of 1024: sample, histogram in LDS, bracket each selector's rank, write the first windows.
// mode 0: explicit ranks k0 (k1); mode 1: percentile (ranks from the sample's own sign counts; the exact ones
// follow from the first sweep).
template <typename T>
__global__ __launch_bounds__(1024) void win_plan_kernel(const ShardTable tab, int n_shards, WinState* __restrict__ st,
                                                        int mode, int n_sel, int use_abs, int64_t k0, int64_t k1,
                                                        int64_t n, double alpha, uint32_t min_shift,
                                                        u32x4* __restrict__ scratch, uint32_t scratch_vecs) {
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
  // every thread's packs are located first and requested together: one memory round trip for the whole sample
  constexpr int kMine = kPlanPacks / kT;
  const void* base[kMine];
  int64_t e[kMine], cnt[kMine];
  float v[kMine][kPack];
#pragma unroll
  for (int m = 0; m < kMine; ++m) {
    // which shard: the table lives in the kernel arguments, so it is walked with a UNIFORM index (scalar loads) and
    // the lane keeps its own pointer / count by selects -- a per-lane index would spill the table to scratch
    const int64_t p = static_cast<int64_t>(threadIdx.x) + m * kT;
    e[m] = (p < n_packs ? p : 0) * (n / n_packs);
    base[m] = tab.ptr[0];
    cnt[m] = tab.count[0];
    bool found = false;
    for (int i = 0; i < n_shards; ++i) {
      const int64_t c = tab.count[i];
      const bool here = !found && (e[m] < c || i + 1 == n_shards);
      base[m] = here ? tab.ptr[i] : base[m];
      cnt[m] = here ? c : cnt[m];
      e[m] = (found || here) ? e[m] : e[m] - c;
      found |= here;
    }
    e[m] &= ~static_cast<int64_t>(kPack - 1);  // whole packs: one 16-byte load (two for fp32) when the shard allows it
  }
#pragma unroll
  for (int m = 0; m < kMine; ++m) {
    if ((reinterpret_cast<uintptr_t>(base[m]) & 15u) == 0 && e[m] + kPack <= cnt[m]) {
      load_pack<T, false>(base[m], e[m], v[m]);
    } else {
#pragma unroll
      for (int j = 0; j < kPack; ++j) v[m][j] = Elem<T>::load1(base[m], e[m] + j < cnt[m] ? e[m] + j : cnt[m] - 1);
    }
  }
  // the counter lines and histogram copies the sweeps add to (the advance leaves them clean, a first call or an
  // abandoned one does not): 136 KB of stores, queued BEHIND the sample's loads (the vector-memory path of the one
  // CU this kernel runs on is in order)
  __builtin_amdgcn_sched_barrier(0);
  for (uint32_t i = threadIdx.x; i < scratch_vecs; i += kT) scratch[i] = u32x4{0, 0, 0, 0};
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int m = 0; m < kMine; ++m) {
    if (static_cast<int64_t>(threadIdx.x) + m * kT >= n_packs) continue;
#pragma unroll
    for (int j = 0; j < kPack; ++j)
      if (e[m] + j < cnt[m])
        atomicAdd(&hist[win_key(__builtin_bit_cast(uint32_t, v[m][j]), use_abs != 0) >> kPlanShift], 1u);
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
    st->arrivals = 0;
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
      // only the bracket itself is histogrammed: with 16-bit inputs the 2048 bins of the smallest shift span 16
      // binades -- half of a weight tensor -- while the bracket is a few dozen values wide
      const uint64_t room = 0xffffffffull - lo;
      w.span = static_cast<uint32_t>(width - 1 < room ? width - 1 : room);
      // percentile: the min side's window sits at the bottom of the data, the max side's at the top -- the sweep
      // tests the near end first and counts what lies beyond it (a handful of keys) instead of what lies before
      w.side = mode == 1 && s == 1 ? 1u : 0u;
      w.k = mode == 0 ? (s == 0 ? k0 : k1) : 0;
      w.done = 0;
      w.fresh = 1;
      st->sel[s] = w;
    }
  }
}

// The advance of one selector: sum the copies of its histogram, place the rank, name the next window -- or write the
// result when the window is one value wide.  Leaves the selector's histogram copies and its `below` counters zeroed.
// COHERENT: run by the last workgroup of a sweep INSIDE a kernel (win_fallback_kernel) -- everything other
// workgroups produced or will read goes through agent-scope atomics (sc1: the device-coherent level, not this
// XCD's L2).  Otherwise it is its own launch and plain accesses do.
struct AdvShared {
  unsigned long long wave_tot[1024 / kWave];
  unsigned long long total, below, neg, nan;
};
template <bool COHERENT, typename V>
__device__ __forceinline__ V win_ld(const V* p) {
  if constexpr (COHERENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}
template <bool COHERENT, typename V>
__device__ __forceinline__ void win_st(V* p, V v) {
  if constexpr (COHERENT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
template <bool COHERENT>
__device__ __forceinline__ WinSel win_read_sel(const WinState* st, int s) {
  static_assert(sizeof(WinSel) == 32, "four 8-byte words");
  unsigned long long q[4];
  const unsigned long long* src = reinterpret_cast<const unsigned long long*>(&st->sel[s]);
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] = win_ld<COHERENT>(src + i);
  return __builtin_bit_cast(WinSel, q);
}

template <int BLOCK, bool COHERENT>
__device__ void win_advance(const int s, uint32_t* __restrict__ hist, WinState* __restrict__ st,
                            WinSlot* __restrict__ slots, int percentile, double alpha, uint32_t min_shift,
                            float* __restrict__ out0, float* __restrict__ out1, AdvShared& sh) {
  const WinSel w = win_read_sel<COHERENT>(st, s);
  if (w.done) return;
  auto write_sel = [&](const WinSel& nw) {
    struct Q { unsigned long long q[4]; };
    const Q q = __builtin_bit_cast(Q, nw);
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(&st->sel[s]);
#pragma unroll
    for (int i = 0; i < 4; ++i) win_st<COHERENT>(dst + i, q.q[i]);
  };
  constexpr int kPer = kWinBins / BLOCK;  // bins per thread
  unsigned long long bins[kPer];
#pragma unroll
  for (int i = 0; i < kPer; ++i) bins[i] = 0;
  {
    // all copies requested together, cleared afterwards (a store between two loads would order them)
    uint32_t v[kCopies][kPer];
#pragma unroll
    for (int c = 0; c < kCopies; ++c) {
      const uint32_t* src = hist + (static_cast<size_t>(c) * kWinSel + s) * kWinBins + threadIdx.x * kPer;
      if constexpr (!COHERENT && kPer == 4) {
        const u32x4 q = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src));
#pragma unroll
        for (int i = 0; i < kPer; ++i) v[c][i] = q[i];
      } else {
#pragma unroll
        for (int i = 0; i < kPer; ++i) v[c][i] = win_ld<COHERENT>(src + i);
      }
    }
#pragma unroll
    for (int c = 0; c < kCopies; ++c) {
      uint32_t* dst = hist + (static_cast<size_t>(c) * kWinSel + s) * kWinBins + threadIdx.x * kPer;
#pragma unroll
      for (int i = 0; i < kPer; ++i) bins[i] += v[c][i];
      if constexpr (!COHERENT && kPer == 4) {
        *reinterpret_cast<u32x4*>(dst) = u32x4{0, 0, 0, 0};
      } else {
#pragma unroll
        for (int i = 0; i < kPer; ++i)
          if (v[c][i]) win_st<COHERENT>(dst + i, 0u);
      }
    }
  }
  unsigned long long t = 0;
#pragma unroll
  for (int i = 0; i < kPer; ++i) t += bins[i];
  // counters: thread i < kSlots reads line i
  unsigned long long c_below = 0, c_neg = 0, c_nan = 0;
  if (threadIdx.x < kSlots) {
    c_below = win_ld<COHERENT>(&slots[threadIdx.x].below[s]);
    c_neg = win_ld<COHERENT>(&slots[threadIdx.x].neg);
    c_nan = win_ld<COHERENT>(&slots[threadIdx.x].nan);
    if (c_below) win_st<COHERENT>(&slots[threadIdx.x].below[s], 0ull);
  }
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  unsigned long long incl = t;
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const unsigned long long up = __shfl_up(incl, d, kWave);
    if (lane >= d) incl += up;
  }
  if (lane == kWave - 1) sh.wave_tot[wid] = incl;
  if (wid == 0) {  // the 64 counter lines live in wave 0
    c_below = wave_reduce(c_below, SumL());
    c_neg = wave_reduce(c_neg, SumL());
    c_nan = wave_reduce(c_nan, SumL());
    if (lane == 0) {
      sh.below = c_below;
      sh.neg = c_neg;
      sh.nan = c_nan;
    }
  }
  __syncthreads();
  unsigned long long off = 0;
  for (int v = 0; v < wid; ++v) off += sh.wave_tot[v];
  incl += off;
  const unsigned long long excl = incl - t;
  if (threadIdx.x == BLOCK - 1) sh.total = incl;
  __syncthreads();
  const unsigned long long total = sh.total;
  const int64_t n = st->n;
  const int64_t neg = static_cast<int64_t>(sh.neg), nan = static_cast<int64_t>(sh.nan);
  const int64_t pos = n - neg - nan;
  int64_t k = w.k;
  if (w.fresh) {
    if (percentile) {
      // percentile.py:36-43 with the exact counts of the first sweep (Python round == rint on a double)
      if (s == 0) k = static_cast<int64_t>(__builtin_fmax(__builtin_rint(static_cast<double>(neg) * alpha), 1.0));
      else k = n - static_cast<int64_t>(__builtin_fmax(__builtin_rint(static_cast<double>(pos) * alpha), 0.0));
      k = k < 1 ? 1 : (k > n ? n : k);
    }
    // (side 1: the counter holds the keys ABOVE the window -- NaNs included, they sort last)
    const unsigned long long below = w.side ? static_cast<unsigned long long>(n) - total - sh.below : sh.below;
    const uint64_t hi = static_cast<uint64_t>(w.lo) + w.span + 1;  // exclusive
    if (static_cast<unsigned long long>(k) <= below) {
      // the sample lied: the rank is below the window.  New window: every key below it.
      if (threadIdx.x == 0) {
        WinSel nw = w;
        nw.lo = 0;
        nw.shift = shift_for(w.lo, min_shift);
        nw.span = w.lo - 1;  // k <= below: there are keys below lo, so lo > 0
        nw.k = k;
        nw.fresh = 0;
        write_sel(nw);
      }
      __syncthreads();
      return;
    }
    if (static_cast<unsigned long long>(k) > below + total) {
      // ... or above it.  New window: every key from its end on (hi < 2^32 here: a window reaching the last key
      // holds every element that is not below it)
      if (threadIdx.x == 0) {
        WinSel nw = w;
        nw.lo = static_cast<uint32_t>(hi);
        nw.shift = shift_for((1ull << 32) - hi, min_shift);
        nw.span = 0xffffffffu - static_cast<uint32_t>(hi);
        nw.k = k - static_cast<int64_t>(below + total);
        nw.fresh = 0;
        write_sel(nw);
      }
      __syncthreads();
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
      // the bin, cut at the end of its parent window
      const uint32_t bin_last = (w.shift < 32 ? (1u << w.shift) : 0u) - 1u;
      const uint32_t left = w.lo + w.span - nw.lo;
      nw.span = bin_last < left ? bin_last : left;
    }
    write_sel(nw);
  }
  __syncthreads();  // `sh` is reused by the next selector
}

// ... as its own launch: one workgroup per selector.
__global__ __launch_bounds__(kAdvBlock) void win_advance_kernel(uint32_t* __restrict__ hist,
                                                                WinState* __restrict__ st, WinSlot* __restrict__ slots,
                                                                int percentile, double alpha, uint32_t min_shift,
                                                                float* __restrict__ out0, float* __restrict__ out1) {
  __shared__ AdvShared sh;
  win_advance<kAdvBlock, false>(blockIdx.x, hist, st, slots, percentile, alpha, min_shift, out0, out1, sh);
}

// The sweep.  A workgroup walks slabs of 32 elements per thread (grid-stride), so the LDS histograms are cleared
// and flushed once per workgroup -- and the workgroups are BIG (1024 threads, one per CU) whenever the tensor has
// two slabs per CU: every workgroup flushes the same few dozen non-empty bins, and same-line device atomics
// serialise.  Whole slabs take the lean path, written so that each element costs, per selector,
//   v_cmp  (key < lo)  -> SGPR mask -> s_bcnt1 / s_add on the scalar unit: the keys below the window,
//   v_sub, v_cmp       offset inside the window?   (keys below lo wrap to offsets beyond span)
//   exec-masked shift / address / ds_add_u32 for the few per cent of elements inside it.
// The ragged last slab and unaligned shards take the per-element path with validity flags.
// ONE launch sweeps every shard of the selection (the cached calibration batches): the slabs of all shards form one
// list, walked grid-stride; the table lives in the kernel arguments and is indexed uniformly (scalar loads).  The
// loads of a workgroup's next slab are issued before it processes the current one.
template <int BLOCK>
struct WinGeom {
  static constexpr int kU = 2;  // packs per thread and slab
  static constexpr uint32_t kSlab = BLOCK * kPack * kU;
};
struct PassTable {
  const void* ptr[kMaxShards];
  int64_t count[kMaxShards];
  // two lists over all shards: the whole slabs of 16-byte aligned shards (lean path), and the rest (a ragged last
  // slab; every slab of an unaligned shard).  Entry i = first list index of shard i.
  uint32_t lean_first[kMaxShards + 1];
  uint32_t rag_first[kMaxShards + 1];
};

// (the body of a sweep, shared by win_pass_kernel and win_fallback_kernel; `load_state` fetches the selectors AFTER
// the first slab has been requested; returns false when every selector is resolved already)
template <typename T, int NSEL, bool SIGNS, int BLOCK, bool EARLY, typename LoadState>
__device__ __forceinline__ bool win_sweep(const PassTable& tab, int n_shards, LoadState&& load_state,
                                          WinSlot* __restrict__ slots, uint32_t* __restrict__ hist, int use_abs) {
  constexpr uint32_t kSlab = WinGeom<BLOCK>::kSlab;
  constexpr int U = WinGeom<BLOCK>::kU;
  constexpr int kWaves = BLOCK / kWave;
  constexpr int kCounters = NSEL + (SIGNS ? 2 : 0);
  __shared__ uint32_t lh[NSEL][kWinBins];
  __shared__ unsigned long long red[kCounters][kWaves];
  auto locate = [&](const uint32_t (&first)[kMaxShards + 1], uint32_t g, int& shard, uint32_t& local) {  // uniform
    int i = 0;
    while (i + 1 < n_shards && g >= first[i + 1]) ++i;
    shard = i;
    local = g - first[i];
  };
  // Request lean slab g.  ALWAYS issues its loads -- past the end of the list they all read the first 16 bytes of
  // the last slab: a conditional issue would make the compiler wait for everything in flight at the join.
  const uint32_t n_lean = tab.lean_first[n_shards];
  auto issue = [&](uint32_t g, RawPack<T> (&raw)[U]) {
    const bool real = g < n_lean;
    int shard;
    uint32_t local;
    locate(tab.lean_first, real ? g : n_lean - 1, shard, local);
    const void* x = tab.ptr[shard];
    const int64_t begin = static_cast<int64_t>(local) * kSlab;
    const uint32_t stride = real ? kPack : 0u;
#pragma unroll
    for (int u = 0; u < U; ++u)
      raw[u] = load_raw<T, true>(x, begin + static_cast<int64_t>((u * BLOCK + threadIdx.x) * stride));
    __builtin_amdgcn_sched_barrier(0);  // nothing that waits for the other buffer moves above these loads
  };
  // the first slab is requested before anything else: the selector state below comes from a cold scalar load, the
  // LDS histograms want clearing -- a memory round trip that overlaps both
  // (EARLY == false, the fallback launch: it usually finds nothing to do, so it looks at the state first)
  RawPack<T> buf_a[U], buf_b[U];
  if (EARLY && n_lean > 0) issue(blockIdx.x, buf_a);
  // every selector resolved: nothing to do (the later rounds of a protocol that needed only one)
  bool live = false;
  uint32_t lo[NSEL], lom1[NSEL], sh[NSEL], span[NSEL];
  bool act[NSEL], fresh[NSEL];
  WinSel sel[NSEL];
  load_state(sel);
#pragma unroll
  for (int s = 0; s < NSEL; ++s) {
    act[s] = sel[s].done == 0;
    lo[s] = sel[s].lo;
    sh[s] = sel[s].shift;
    span[s] = sel[s].span;
    if (!act[s]) {  // a finished selector gets an empty window: only key 0xffffffff passes, and nobody reads its bins
      lo[s] = 0xffffffffu;
      span[s] = 0;
    }
    fresh[s] = act[s] && sel[s].fresh != 0;
    lom1[s] = lo[s] - 1u;
    live |= act[s];
  }
  if (!live) return false;
  if (!EARLY && n_lean > 0) issue(blockIdx.x, buf_a);
  for (uint32_t i = threadIdx.x; i < NSEL * kWinBins; i += BLOCK) (&lh[0][0])[i] = 0;
  __syncthreads();
  // per-lane counters (ragged path) and wave-uniform ones (lean path)
  uint32_t lt[NSEL];
#pragma unroll
  for (int s = 0; s < NSEL; ++s) lt[s] = 0;
  uint32_t neg = 0, nan = 0;
  const bool lane0 = (threadIdx.x & (kWave - 1)) == 0;
  // |x|: clear the sign first; then the same transform (the sign fill of a non-negative word is 0)
  const uint32_t amask = use_abs ? 0x7fffffffu : 0xffffffffu;
  auto key_of = [&](uint32_t bits) {
    const uint32_t b = bits & amask;
    const uint32_t m = static_cast<uint32_t>(static_cast<int32_t>(b) >> 31) | 0x80000000u;
    return (b ^ m) - kRot;
  };
  // The percentile's first sweep (two selectors + sign counts): selector 0's window sits at the bottom of the data,
  // selector 1's at the top (WinSel::side = 0 / 1).  One compare against the window's NEAR end settles all but a
  // few per cent of the elements; only those go on to the window test, and the ones beyond the far end are what
  // the sweep counts (side 1: the keys ABOVE the window; the advance turns that into the keys below).
  constexpr bool ONESIDED = SIGNS && NSEL == 2;
  auto visit = [&](uint32_t kk, bool valid) {
    if constexpr (SIGNS) {
      neg += valid && kk < kKeyZero;
      nan += valid && kk > kKeyInf;
    }
#pragma unroll
    for (int s = 0; s < NSEL; ++s) {
      const uint32_t d = kk - lo[s];
      if (ONESIDED && s == 1) lt[s] += valid && kk > lo[s] + span[s];  // lo + span <= 0xffffffff by construction
      else lt[s] += valid && kk < lo[s];
      if (valid && d <= span[s]) atomicAdd(&lh[s][d >> sh[s]], 1u);
    }
  };
  // count the lanes of a compare on the scalar unit, HERE: as plain C++ (popcount of a ballot, added to a uniform
  // counter) the adds are sunk to the end of the slab and the 128 masks waiting for them spill into VGPR lanes
  auto count = [](uint32_t& acc, bool p) {
    const uint64_t mask = __builtin_amdgcn_ballot_w64(p);
    uint32_t c;
    asm volatile("s_bcnt1_i32_b64 %1, %2\n\ts_add_u32 %0, %0, %1" : "+s"(acc), "=&s"(c) : "s"(mask) : "scc");
  };
  auto lean = [&](uint32_t kk, uint32_t (&w_lt)[NSEL], uint32_t& w_neg, uint32_t& w_nan) {
    if constexpr (SIGNS) {
      count(w_neg, kk < kKeyZero);
      count(w_nan, kk > kKeyInf);
    }
    if constexpr (ONESIDED) {
      if (kk <= lo[0] + span[0]) {  // at or below the top of the bottom window: rare
        const uint32_t d = kk - lo[0];
        if (d <= span[0]) atomicAdd(&lh[0][d >> sh[0]], 1u);
        else ++lt[0];  // wrapped: below the window
      }
      if (kk >= lo[1]) {  // at or above the bottom of the top window: rare
        const uint32_t d = kk - lo[1];
        if (d <= span[1]) atomicAdd(&lh[1][d >> sh[1]], 1u);
        else ++lt[1];  // above the window
      }
      return;
    }
#pragma unroll
    for (int s = 0; s < NSEL; ++s) {
      // `kk <= lo - 1`, not `kk < lo`: the latter is folded into the borrow of the subtraction below, which costs
      // two more VALU operations to turn back into a lane mask (lo == 0: counts everything, dropped at the end)
      count(w_lt[s], kk <= lom1[s]);
      const uint32_t d = kk - lo[s];
      if (d <= span[s]) atomicAdd(&lh[s][d >> sh[s]], 1u);
    }
  };
  auto sweep_lean = [&](const RawPack<T> (&raw)[U]) {
    // wave-uniform counters of this slab (SGPRs), folded into lane 0's counters at its end
    uint32_t w_lt[NSEL], w_neg = 0, w_nan = 0;
#pragma unroll
    for (int s = 0; s < NSEL; ++s) w_lt[s] = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if constexpr (T::id == SBQ_F32) {
#pragma unroll
        for (int q = 0; q < 4; ++q) lean(key_of(raw[u].d[0][q]), w_lt, w_neg, w_nan);
#pragma unroll
        for (int q = 0; q < 4; ++q) lean(key_of(raw[u].d[1][q]), w_lt, w_neg, w_nan);
      } else if constexpr (T::id == SBQ_BF16) {
        // bf16 -> fp32 bits is a shift / a mask: no conversion
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t w = raw[u].d[0][q];
          lean(key_of(w << 16), w_lt, w_neg, w_nan);
          lean(key_of(w & 0xffff0000u), w_lt, w_neg, w_nan);
        }
      } else {
        float v[kPack];
        unpack_raw<T>(raw[u], v);
#pragma unroll
        for (int q = 0; q < kPack; ++q) lean(key_of(__builtin_bit_cast(uint32_t, v[q])), w_lt, w_neg, w_nan);
      }
    }
#pragma unroll
    for (int s = 0; s < NSEL; ++s) lt[s] += !ONESIDED && lane0 && lo[s] != 0 ? w_lt[s] : 0u;
    if constexpr (SIGNS) {
      neg += lane0 ? w_neg : 0u;
      nan += lane0 ? w_nan : 0u;
    }
  };
  if (n_lean > 0) {
    // two slab buffers, alternating (a copy `current = next` would have to wait for the loads it is meant to hide)
    uint32_t g = blockIdx.x;
    while (g < n_lean) {
      issue(g + gridDim.x, buf_b);
      sweep_lean(buf_a);
      g += gridDim.x;
      if (g >= n_lean) break;
      issue(g + gridDim.x, buf_a);
      sweep_lean(buf_b);
      g += gridDim.x;
    }
  }
  // the ragged last slab of a shard, and every slab of an unaligned one
  const uint32_t n_rag = tab.rag_first[n_shards];
  for (uint32_t r = blockIdx.x; r < n_rag; r += gridDim.x) {
    int shard;
    uint32_t local;
    locate(tab.rag_first, r, shard, local);
    local += tab.lean_first[shard + 1] - tab.lean_first[shard];
    const void* x = tab.ptr[shard];
    const int64_t n = tab.count[shard];
    const int64_t begin = static_cast<int64_t>(local) * kSlab;
    const int64_t end = begin + kSlab < n ? begin + kSlab : n;
    int64_t vend = begin;
    if ((reinterpret_cast<uintptr_t>(x) & 15u) == 0) {
      vend = begin + ((end - begin) / kPack) * kPack;
      for (int64_t e = begin + static_cast<int64_t>(threadIdx.x) * kPack; e < vend; e += static_cast<int64_t>(BLOCK) * kPack) {
        float v[kPack];
        load_pack<T, true>(x, e, v);
#pragma unroll
        for (int q = 0; q < kPack; ++q) visit(key_of(__builtin_bit_cast(uint32_t, v[q])), true);
      }
    }
    for (int64_t e = vend + threadIdx.x; e < end; e += BLOCK)
      visit(key_of(__builtin_bit_cast(uint32_t, Elem<T>::load1(x, e))), true);
  }
  // counters: lanes -> wave -> workgroup -> one of the 64 counter lines
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  unsigned long long tot[kCounters];
#pragma unroll
  for (int s = 0; s < NSEL; ++s) tot[s] = wave_reduce(static_cast<unsigned long long>(lt[s]), SumL());
  if constexpr (SIGNS) {
    tot[NSEL] = wave_reduce(static_cast<unsigned long long>(neg), SumL());
    tot[NSEL + 1] = wave_reduce(static_cast<unsigned long long>(nan), SumL());
  }
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < kCounters; ++c) red[c][wid] = tot[c];
  }
  __syncthreads();
  WinSlot* slot = slots + (blockIdx.x % kSlots);
  if (threadIdx.x < kCounters) {
    unsigned long long t = 0;
    for (int w = 0; w < kWaves; ++w) t += red[threadIdx.x][w];
    if (t) {
      if (static_cast<int>(threadIdx.x) < NSEL) {
        bool mine = false;
#pragma unroll
        for (int s = 0; s < NSEL; ++s) mine |= static_cast<int>(threadIdx.x) == s && fresh[s];
        if (mine) atomicAdd(&slot->below[threadIdx.x], t);
      } else {
        atomicAdd(threadIdx.x == NSEL ? &slot->neg : &slot->nan, t);
      }
    }
  }
#pragma unroll
  for (int s = 0; s < NSEL; ++s) {
    if (!act[s]) continue;
    uint32_t* gh = hist + (static_cast<size_t>(blockIdx.x % kCopies) * kWinSel + s) * kWinBins;
    for (uint32_t i = threadIdx.x; i < static_cast<uint32_t>(kWinBins); i += BLOCK) {
      const uint32_t v = lh[s][i];
      if (v) atomicAdd(&gh[i], v);
    }
  }
  return true;
}

template <typename T, int NSEL, bool SIGNS, int BLOCK>
__global__ __launch_bounds__(BLOCK) void win_pass_kernel(const PassTable tab, int n_shards,
                                                         const WinState* __restrict__ st, WinSlot* __restrict__ slots,
                                                         uint32_t* __restrict__ hist, int use_abs) {
  win_sweep<T, NSEL, SIGNS, BLOCK, true>(tab, n_shards, [&](WinSel (&sel)[NSEL]) {
#pragma unroll
    for (int s = 0; s < NSEL; ++s) sel[s] = st->sel[s];
  }, slots, hist, use_abs);
}

// A round after the expected ones: sweep and advance in ONE launch.  Such rounds are needed only when the sample
// lied about a window; as separate (sweep, advance) launches they cost ~3 us each just to find every selector
// resolved.  Here a small grid looks at the state first (and leaves at once when nothing is left to do), sweeps, and
// the last workgroup to arrive advances the selectors.  Nobody ever WAITS for another workgroup -- a version that
// ran all the remaining rounds in one launch behind a grid-wide wait was 3 us faster and could deadlock when more
// such launches are live at once than the chip holds workgroups.  No agent-scope fences (see the GPTQ strip
// kernels): the histogram / counter adds are agent-scope atomics, the advance reads and writes them -- and the
// selector state -- with agent-scope atomic loads / stores, and a workgroup's adds are acknowledged (vmcnt(0))
// before its arrival is counted.
template <typename T, int NSEL, int BLOCK>
__global__ __launch_bounds__(BLOCK) void win_fallback_kernel(const PassTable tab, int n_shards, WinState* __restrict__ st,
                                                             WinSlot* __restrict__ slots, uint32_t* __restrict__ hist,
                                                             int use_abs, int percentile, double alpha, uint32_t min_shift,
                                                             float* __restrict__ out0, float* __restrict__ out1) {
  __shared__ AdvShared adv;
  __shared__ uint32_t s_last;
  const bool live = win_sweep<T, NSEL, false, BLOCK, false>(tab, n_shards, [&](WinSel (&sel)[NSEL]) {
#pragma unroll
    for (int s = 0; s < NSEL; ++s) sel[s] = st->sel[s];
  }, slots, hist, use_abs);
  if (!live) return;  // uniform over the grid: every workgroup read the same state
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (threadIdx.x == 0)
    s_last = __hip_atomic_fetch_add(&st->arrivals, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
  __syncthreads();
  if (!s_last) return;
#pragma unroll
  for (int s = 0; s < NSEL; ++s)
    win_advance<BLOCK, true>(s, hist, st, slots, percentile, alpha, min_shift, out0, out1, adv);
  if (threadIdx.x == 0) win_st<true>(&st->arrivals, 0u);
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
  const uint32_t cus = cu_count();
  int rc = dispatch_dtype(x_dtype, [&](auto tag) {
    using T = decltype(tag);
    win_plan_kernel<T><<<1, 1024, 0, st>>>(tab, n_shards, state, percentile ? 1 : 0, n_sel, use_abs, k0, k1, n, alpha,
                                           min_shift, reinterpret_cast<u32x4*>(ws + kStateBytes),
                                           static_cast<uint32_t>((kSlotBytes + kHistBytes) / 16));
  });
  if (rc != SBQ_OK) return rc;
  // rounds: one resolves a 16-bit input, two an fp32 one -- when the first window holds the rank; a missed window
  // costs up to ceil((32 - min_shift) / 11) more.  All are enqueued; rounds after the last needed one exit at once.
  const int rounds = 1 + static_cast<int>((32 - min_shift + kWinLog - 1) / kWinLog);
  // the slab list: 1024-thread workgroups (one per CU: 4x fewer histogram flushes) when that gives every CU two slabs
  // of 16 Ki elements, else 256-thread workgroups and 4 Ki slabs
  int64_t big_slabs = 0;
  for (int i = 0; i < n_shards; ++i) big_slabs += ceil_div(counts[i], static_cast<int64_t>(WinGeom<1024>::kSlab));
  const bool big = big_slabs >= 2 * static_cast<int64_t>(cus) && knob(2) != 8;
  const int64_t slab = big ? WinGeom<1024>::kSlab : WinGeom<kBlock>::kSlab;
  auto make_table = [&](int64_t slab_elems, PassTable& t, int64_t& total) {
    int64_t n_lean = 0;
    total = 0;
    for (int i = 0; i < n_shards; ++i) {
      t.ptr[i] = shards[i];
      t.count[i] = counts[i];
      const int64_t all = ceil_div(counts[i], slab_elems), lean = aligned16(shards[i]) ? counts[i] / slab_elems : 0;
      t.lean_first[i] = static_cast<uint32_t>(n_lean);
      t.rag_first[i] = static_cast<uint32_t>(total - n_lean);
      n_lean += lean;
      total += all;
    }
    t.lean_first[n_shards] = static_cast<uint32_t>(n_lean);
    t.rag_first[n_shards] = static_cast<uint32_t>(total - n_lean);
  };
  PassTable pt{};
  int64_t total_slabs = 0;
  make_table(slab, pt, total_slabs);
  if (total_slabs >= (1ll << 31)) return SBQ_ERR_ARG;
  // rounds beyond the expected ones (one sweep for 16-bit inputs, up to three for fp32) almost always find every
  // selector resolved and exit at once: launch them small -- a miss of the first window just sweeps slower
  const int expected = min_shift > 0 ? 1 : 3;
  for (int r = 0; r < rounds && r < expected; ++r) {
    const int64_t cap = r >= expected ? 64 : (big ? cus : 4 * static_cast<int64_t>(cus));
    const uint32_t grid = static_cast<uint32_t>(total_slabs < cap ? (total_slabs > 0 ? total_slabs : 1) : cap);
    const bool signs = r == 0 && percentile;
    rc = dispatch_dtype(x_dtype, [&](auto tag) {
      using T = decltype(tag);
#define SBQ_WIN2(NS, SG, B) win_pass_kernel<T, NS, SG, B><<<grid, B, 0, st>>>(pt, n_shards, state, slots, hist, use_abs)
#define SBQ_WIN(NS, SG)         \
  do {                          \
    if (big) SBQ_WIN2(NS, SG, 1024); \
    else SBQ_WIN2(NS, SG, kBlock);   \
  } while (0)
      if (n_sel == 1) SBQ_WIN(1, false);
      else if (signs) SBQ_WIN(2, true);
      else SBQ_WIN(2, false);
#undef SBQ_WIN
#undef SBQ_WIN2
    });
    if (rc != SBQ_OK) return rc;
    // the advance of the round that resolves a selector also writes its result
    win_advance_kernel<<<n_sel, kAdvBlock, 0, st>>>(hist, state, slots, percentile ? 1 : 0, alpha, min_shift, out0,
                                                    out1);
  }
  if (rounds > expected) {
    // (its workgroups are 1024 threads whatever the sweeps above used: their own slab list)
    PassTable pf{};
    int64_t fb_slabs = 0;
    make_table(WinGeom<1024>::kSlab, pf, fb_slabs);
    const int64_t cap = cus >= 2 ? cus / 2 : 1;  // a miss sweeps at half speed; an idle launch exits sooner
    const uint32_t grid = static_cast<uint32_t>(fb_slabs < cap ? (fb_slabs > 0 ? fb_slabs : 1) : cap);
    for (int r = expected; r < rounds && rc == SBQ_OK; ++r) {
      rc = dispatch_dtype(x_dtype, [&](auto tag) {
        using T = decltype(tag);
        if (n_sel == 1)
          win_fallback_kernel<T, 1, 1024><<<grid, 1024, 0, st>>>(pf, n_shards, state, slots, hist, use_abs,
                                                              percentile ? 1 : 0, alpha, min_shift, out0, out1);
        else
          win_fallback_kernel<T, 2, 1024><<<grid, 1024, 0, st>>>(pf, n_shards, state, slots, hist, use_abs,
                                                              percentile ? 1 : 0, alpha, min_shift, out0, out1);
      });
    }
    if (rc != SBQ_OK) return rc;
  }
  return check_launch();
}

}  // namespace sbq
