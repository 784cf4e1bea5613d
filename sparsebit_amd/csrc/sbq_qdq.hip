// sbq_qdq.hip -- forward quantize-dequantize for gfx950 (MI355X).
//
// Replaces QuantizePerTensorForwardCUDA / QuantizePerChannelForwardCUDA
// (sparsebit/quantization/torch_extensions/fake_quant_tensor.cu:50-66,170-188)
// and, fused, the `weight * w_mask` multiply of sparsebit/sparse/modules/
// conv.py:40.  Arithmetic follows the reference CPU path quant_tensor.py:182-184.
//
// Design (HBM-bound streaming, no MFMA):
//   * a lane moves one "pack" of 8 consecutive elements: a single 16-byte
//     global_load_dwordx4 for bf16/fp16 (two for fp32), so a wave covers 1 KiB
//     of contiguous input per load instruction;
//   * ROWS variant: a workgroup owns a tile of one channel row, so scale /
//     zero_point are wave-uniform and live in SGPRs (s_load), no per-lane gather;
//   * FLAT variant (short rows): packs are numbered across the whole tensor and
//     the channel is recomputed per pack, keeping every lane busy;
//   * U independent packs per lane are loaded before any is used (memory-level
//     parallelism), streamed with non-temporal hints: the data is touched once, and
//     the loads of the next tile are issued before the current one is processed;
//   * x / s is exact but cheap: the row's reciprocal is computed once and each
//     element pays a multiply and two fma refinements (sbq_common.hpp, fast_div);
//   * anything not 16-byte friendly (inner % 8 != 0, odd pointers, the exotic
//     rounding modes) goes through a scalar kernel with identical arithmetic.
#include "sbq_qdq_math.hpp"

namespace sbq {
namespace {


// the four levels lv[0..3] of a lane's 4-element run at element index i (SPLIT mapping)
template <int QT>
__device__ __forceinline__ void store_q_half(void* q, int64_t i, const float* lv) {
  if constexpr (QT == SBQ_Q_I8) {
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc |= (static_cast<uint32_t>(static_cast<int>(lv[j])) & 0xffu) << (8 * j);
    st4<true>(static_cast<char*>(q) + i, acc);
  } else if constexpr (QT == SBQ_Q_I32) {
    u32x4 a;
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = static_cast<uint32_t>(static_cast<int>(lv[j]));
    st16<true>(static_cast<char*>(q) + i * 4, a);
  }
}

template <int QT>
__device__ __forceinline__ void store_q_pack(void* q, int64_t i, const float (&lv)[kPack]) {
  if constexpr (QT == SBQ_Q_I8) {
    u32x2 w;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      uint32_t acc = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc |= (static_cast<uint32_t>(static_cast<int>(lv[4 * h + j])) & 0xffu) << (8 * j);
      w[h] = acc;
    }
    st8<true>(static_cast<char*>(q) + i, w);
  } else if constexpr (QT == SBQ_Q_I32) {
    u32x4 a, b;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a[j] = static_cast<uint32_t>(static_cast<int>(lv[j]));
      b[j] = static_cast<uint32_t>(static_cast<int>(lv[4 + j]));
    }
    char* p = static_cast<char*>(q) + i * 4;
    st16<true>(p, a);
    st16<true>(p + 16, b);
  } else if constexpr (QT == SBQ_Q_I4) {
    // eight levels -> one dword: element i in the low nibble of byte i/2 (two's complement for a
    // signed range).  i is a multiple of 8, so a wave writes 256 contiguous bytes.
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < kPack; ++j) acc |= (static_cast<uint32_t>(static_cast<int>(lv[j])) & 0xfu) << (4 * j);
    __builtin_nontemporal_store(acc, reinterpret_cast<uint32_t*>(static_cast<char*>(q) + (i >> 1)));
  }
}

struct QdqGeom {
  int64_t inner;           // elements per channel row
  uint32_t C;
  uint32_t packs_per_row;  // inner / 8
  uint32_t slabs_per_row;  // ROWS: ceil(packs_per_row / kBlock); a slab = kBlock packs of one row
  uint32_t n_slabs;        // ROWS: rows * slabs_per_row
  uint32_t n_tiles;        // ceil(n_slabs / U) (ROWS) or ceil(total_packs / (kBlock*U)) (FLAT)
  uint32_t total_packs;    // FLAT only
  float qlo, qhi;
  uint32_t lsq;            // LSQ pre-ops on the raw parameters: s = |s|, zp = clamp(zp, qmin, qmax) (lsq.py:61-62)
};

struct QdqPtrs {
  const void* x;
  void* y;
  void* q;
  const uint8_t* mask;
  const float* thresh;
  const float* scale;
  const float* zp;
};

// One tile = U packs per lane.
template <int U>
struct Tile {
  int64_t elem[U];
  bool ok[U];
  float s[U], z[U];
  int64_t elemB[U];  // SPLIT mapping only: the lane's second 4-element run
  bool okB[U];
};

// ROWS: the tensor is cut into slabs of kBlock packs (2048 elements) that never straddle a
// row; a tile is U consecutive slabs, so each of its U loads has a block-uniform channel
// (scale / zero_point through s_load) and a wave reads 1 KiB of contiguous HBM per load
// instruction.  FLAT (short rows): packs are numbered across the tensor, channel per lane.
// Out-of-range lanes point at the last valid pack (loads are never predicated, only stores).
// Position of a ROWS tile's first slab.  Tiles of a workgroup are visited in steps of
// gridDim.x tiles, so after one division at the start the cursor only adds a precomputed
// (rows, slabs) stride -- no per-tile integer division on the scalar unit in front of the loads.
struct RowCursor {
  uint32_t sl, row, col, c;          // first slab of the current tile
  uint32_t d_sl, d_row, d_col, d_c;  // stride of one step (gridDim.x tiles)
};

// (position and stride separately: the stride needs the grid size, a hidden kernel argument that is not preloaded
// -- it is computed after the first tile's loads have been issued)
template <int U>
__device__ __forceinline__ RowCursor make_cursor(const QdqGeom& g, uint32_t tile) {
  RowCursor k;
  k.sl = tile * U;
  k.row = k.sl / g.slabs_per_row;
  k.col = k.sl - k.row * g.slabs_per_row;
  k.c = k.row % g.C;
  k.d_sl = k.d_row = k.d_col = k.d_c = 0;
  return k;
}
template <int U>
__device__ __forceinline__ void set_stride(const QdqGeom& g, RowCursor& k, uint32_t step_tiles) {
  k.d_sl = step_tiles * U;
  k.d_row = k.d_sl / g.slabs_per_row;
  k.d_col = k.d_sl - k.d_row * g.slabs_per_row;
  k.d_c = k.d_row % g.C;
}

__device__ __forceinline__ void advance(const QdqGeom& g, RowCursor& k) {
  k.sl += k.d_sl;
  k.col += k.d_col;
  k.row += k.d_row;
  k.c += k.d_c;
  const bool carry = k.col >= g.slabs_per_row;  // (selects: see locate)
  k.col -= carry ? g.slabs_per_row : 0u;
  k.row += carry ? 1u : 0u;
  k.c += carry ? 1u : 0u;
  // c < C and d_c < C before, plus a carry of at most 1: c < 2C, one subtraction suffices
  k.c -= k.c >= g.C ? g.C : 0u;
}

template <bool FLAT, int U, bool SPLIT = false>
__device__ __forceinline__ void locate(const QdqGeom& g, uint32_t tile, const RowCursor& k,
                                       const float* __restrict__ scale,
                                       const float* __restrict__ zero_point, Tile<U>& t) {
  if constexpr (!FLAT) {
    uint32_t sl = k.sl, row = k.row, col = k.col, c = k.c;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool slab_ok = sl < g.n_slabs;
      if constexpr (SPLIT) {
        // lane t of the slab: elements [4t, 4t+4) and [1024 + 4t, 1024 + 4t + 4) of its 2048
        constexpr int kHalf = kBlock * kPack / 2;
        const int64_t eA = static_cast<int64_t>(col) * (kBlock * kPack) + 4 * threadIdx.x;
        const int64_t eB = eA + kHalf;
        t.ok[u] = slab_ok && eA < g.inner;  // rows are whole 8-element packs: a started run is a whole run
        t.okB[u] = slab_ok && eB < g.inner;
        const int64_t rb = static_cast<int64_t>(row) * g.inner;
        t.elem[u] = rb + (eA < g.inner ? eA : g.inner - 4);
        t.elemB[u] = rb + (eB < g.inner ? eB : g.inner - 4);
      } else {
      const uint32_t pk = col * kBlock + threadIdx.x;
      t.ok[u] = slab_ok && pk < g.packs_per_row;
      const uint32_t pkc = pk < g.packs_per_row ? pk : g.packs_per_row - 1;
      t.elem[u] = static_cast<int64_t>(row) * g.inner + static_cast<int64_t>(pkc) * kPack;
      }
      // (requested here, used by settle() AFTER the tile's vector loads have been issued: finishing them here made
      // every tile's loads wait for a scalar round trip first)
      t.s[u] = uniform_load(scale, c);
      t.z[u] = uniform_load(zero_point, c);
      // advance to the next slab without dividing; past the end stay on the last one.  SELECTS, not branches: with
      // branches every slab's parameter loads sit in a basic block of their own and the compiler drains the scalar
      // loads at each join -- four dependent scalar round trips (~1 us) in front of the tile's first vector load.
      const bool more = sl + 1 < g.n_slabs;
      const bool wrap = more && col + 1 == g.slabs_per_row;
      col = more ? (wrap ? 0u : col + 1u) : col;
      row += wrap ? 1u : 0u;
      c = wrap ? (c + 1u == g.C ? 0u : c + 1u) : c;
      sl = more ? sl + 1u : g.n_slabs;
    }
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t pk = (tile * U + u) * kBlock + threadIdx.x;
      t.ok[u] = pk < g.total_packs;
      const uint32_t pkc = t.ok[u] ? pk : g.total_packs - 1;
      t.elem[u] = static_cast<int64_t>(pkc) * kPack;
      uint32_t c = 0;
      if (g.C != 1) c = (pkc / g.packs_per_row) % g.C;  // per tensor: no division
      t.s[u] = scale[c];
      t.z[u] = zero_point[c];
    }
  }
}

// the tile's raw scale / zero point -> what the arithmetic uses (LSQ: |s|, clamped zero point; zero point to the
// nearest integer, half to even).  Called after issue_loads: the wait for the parameters overlaps the data's flight.
template <int U>
__device__ __forceinline__ void settle(const QdqGeom& g, Tile<U>& t) {
#pragma unroll
  for (int u = 0; u < U; ++u) {
    float s_ = t.s[u], z_ = t.z[u];
    if (g.lsq) {
      s_ = __builtin_fabsf(s_);
      z_ = __builtin_amdgcn_fmed3f(z_, g.qlo, g.qhi);
    }
    t.s[u] = s_;
    t.z[u] = __builtin_rintf(z_);
  }
}

template <typename Tin, int MASK, bool NT, int U, bool SPLIT = false>
__device__ __forceinline__ void issue_loads(const void* __restrict__ x, const uint8_t* __restrict__ mask,
                                            const Tile<U>& t, RawPack<Tin> (&raw)[U], u32x2 (&mk)[U]) {
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if constexpr (SPLIT) raw[u] = load_raw2<Tin, NT>(x, t.elem[u], t.elemB[u]);
    else raw[u] = load_raw<Tin, NT>(x, t.elem[u]);
    if constexpr (MASK == MASK_BYTES) {
      if constexpr (SPLIT) mk[u] = u32x2{ld4<NT>(mask + t.elem[u]), ld4<NT>(mask + t.elemB[u])};
      else mk[u] = ld8<NT>(mask + t.elem[u]);
    }
  }
}

template <typename Tin, typename Tout, int QT, int MASK, bool FLAT, bool NT, int U, int MATH, bool SPLIT = false>
__device__ __forceinline__ void finish_tile(void* __restrict__ y, void* __restrict__ q, const Tile<U>& t,
                                            const RawPack<Tin> (&raw)[U], const u32x2 (&mk)[U], float thr,
                                            float qlo, float qhi) {
#pragma unroll
  for (int u = 0; u < U; ++u) {
    float v[kPack], lv[kPack], dq[kPack];
    unpack_raw<Tin>(raw[u], v);
    quantize_pack<MASK, FLAT, MATH>(v, mk[u], thr, t.s[u], t.z[u], qlo, qhi, lv, dq);
    if constexpr (SPLIT) {
      if (QT == SBQ_Q_NONE || y) {  // y == nullptr: quantize only (block-uniform)
        if (t.ok[u]) store_half_f32<NT>(y, t.elem[u], dq);
        if (t.okB[u]) store_half_f32<NT>(y, t.elemB[u], dq + 4);
      }
      if constexpr (QT != SBQ_Q_NONE) {
        if (t.ok[u]) store_q_half<QT>(q, t.elem[u], lv);
        if (t.okB[u]) store_q_half<QT>(q, t.elemB[u], lv + 4);
      }
    } else if (t.ok[u]) {
      if constexpr (QT != SBQ_Q_NONE) {
        if (y) store_pack<Tout, NT>(y, t.elem[u], dq);  // y == nullptr: quantize only (block-uniform)
        store_q_pack<QT>(q, t.elem[u], lv);
      } else {
        store_pack<Tout, NT>(y, t.elem[u], dq);
      }
    }
  }
}

// Pointers are separate __restrict__ kernel parameters (not struct members) so that the
// compiler keeps the block-uniform scale / zero_point loads on the scalar unit.
//
// The loop is software pipelined: the (still packed) loads of tile i+1 are in flight while
// tile i is converted, quantized and stored, so the VALU work (~15 ops per element) hides
// under HBM latency instead of adding to it -- the kernel is short (one 4096x4096 weight is
// ~12 us), there is no steady state to amortise a load->compute->store serialisation.
//
// Parameter order matters: the library is built with -amdgpu-kernarg-preload-count=16 (14 fit), so the
// first 14 dwords of the kernarg segment (x, the tile geometry, scale, zero_point) arrive
// in SGPRs with the wave instead of through a cold scalar load -- every launch gets a fresh
// kernarg block, and for a ~12 us kernel one more dependent HBM round trip in front of the
// first data load is measurable.  The rarely used arguments follow and are loaded normally.
template <typename Tin, typename Tout, int QT, int MASK, bool FLAT, bool NT, int U, int MATH, bool NTS = NT>
__global__ __launch_bounds__(kBlock) void qdq_pack_kernel(
    const void* __restrict__ x, uint32_t n_tiles, uint32_t slabs_per_row, uint32_t packs_per_row,
    uint32_t n_channels, int64_t inner, uint32_t n_slabs, uint32_t total_packs, const float* __restrict__ scale,
    const float* __restrict__ zero_point,
    // ---- not preloaded (nothing the first tile's loads need, the mask pointer of the masked variants excepted) ----
    void* __restrict__ y, void* __restrict__ q, const uint8_t* __restrict__ mask, const float* __restrict__ thresh,
    float qlo, float qhi, uint32_t lsq) {
  QdqGeom g;
  g.lsq = lsq;
  g.inner = inner;
  g.C = n_channels;
  g.packs_per_row = packs_per_row;
  g.slabs_per_row = slabs_per_row;
  g.n_slabs = n_slabs;
  g.n_tiles = n_tiles;
  g.total_packs = total_packs;
  g.qlo = qlo;
  g.qhi = qhi;
  float thr = 0.0f;
  if constexpr (MASK == MASK_THRESH) thr = *thresh;

  // fp32 outputs: two 4-element runs per lane, half a slab apart (sbq_common.hpp: load_raw2); the packed-int4
  // output needs 8 consecutive levels per dword and keeps the contiguous pack
  constexpr bool SPLIT = !FLAT && QT != SBQ_Q_I4 && Tout::id == SBQ_F32;
  uint32_t tile = blockIdx.x;
  if (tile >= g.n_tiles) return;
  Tile<U> ta, tb;
  RawPack<Tin> ra[U], rb[U];
  u32x2 ma[U], mb[U];
  RowCursor cur{};
  if constexpr (!FLAT) cur = make_cursor<U>(g, tile);
  // fetches happen in tile order (tile, tile+G, tile+2G, ...): one cursor, stepped after each
#define SBQ_FETCH(T, R, M, IDX)                                     \
  locate<FLAT, U, SPLIT>(g, (IDX), cur, scale, zero_point, T);      \
  issue_loads<Tin, MASK, NT, U, SPLIT>(x, mask, T, R, M);           \
  __builtin_amdgcn_sched_barrier(0); /* nothing of the following FINISH (whose first use waits for the */ \
  /* PREVIOUS tile's loads) may be scheduled above these loads: that would serialise the pipeline */   \
  settle<U>(g, T);                                                  \
  if constexpr (!FLAT) advance(g, cur)
#define SBQ_FINISH(T, R, M) \
  finish_tile<Tin, Tout, QT, MASK, FLAT, NTS, U, MATH, SPLIT>(y, q, T, R, M, thr, g.qlo, g.qhi)
  // the first tile: as SBQ_FETCH, with the cursor's stride computed behind the loads
  locate<FLAT, U, SPLIT>(g, tile, cur, scale, zero_point, ta);
  issue_loads<Tin, MASK, NT, U, SPLIT>(x, mask, ta, ra, ma);
  __builtin_amdgcn_sched_barrier(0);
  settle<U>(g, ta);
  const uint32_t G = gridDim.x;
  if constexpr (!FLAT) {
    set_stride<U>(g, cur, G);
    advance(g, cur);
  }
  // Steady state: both prefetches are unconditional, so the compiler's vmcnt bookkeeping
  // stays exact (a conditional prefetch merges two scoreboard states at the join and makes
  // every wait drain the prefetched tile too).
  while (static_cast<uint64_t>(tile) + 2ull * G < g.n_tiles) {
    SBQ_FETCH(tb, rb, mb, tile + G);
    SBQ_FINISH(ta, ra, ma);
    SBQ_FETCH(ta, ra, ma, tile + 2 * G);
    SBQ_FINISH(tb, rb, mb);
    tile += 2 * G;
  }
  if (static_cast<uint64_t>(tile) + G < g.n_tiles) {  // exactly one more tile after this one
    SBQ_FETCH(tb, rb, mb, tile + G);
    SBQ_FINISH(ta, ra, ma);
    SBQ_FINISH(tb, rb, mb);
  } else {
    SBQ_FINISH(ta, ra, ma);
  }
#undef SBQ_FETCH
#undef SBQ_FINISH
}

// Multi-tensor variant of the ROWS kernel: tile -> (item, tile inside the item); the four
// pointers of the item come from a device table through scalar loads (block-uniform index).
// Same helpers, same arithmetic, same two-stage software pipeline.
template <typename Tin, typename Tout, int U>
__global__ __launch_bounds__(kBlock) void qdq_batched_kernel(const void* const* __restrict__ table,
                                                             uint32_t n_items, uint32_t tiles_per_item,
                                                             uint32_t n_tiles_total, const QdqGeom g) {
  constexpr int MASK = MASK_NONE;
  struct Ptrs {
    const void* x;
    void* y;
    const float* scale;
    const float* zp;
  };
  // Fetch order is tile, tile+G, tile+2G, ...: (item, tile-in-item) and the row cursor are
  // stepped, not recomputed; the table (a dependent scalar-load chain in front of the data
  // loads) is read only when the item changes.
  uint32_t tile = blockIdx.x;
  const uint32_t G = gridDim.x;
  if (tile >= n_tiles_total) return;
  uint32_t item = tile / tiles_per_item;
  uint32_t lt = tile - item * tiles_per_item;
  Ptrs cur{};
  RowCursor rc{};
  auto load_item = [&]() {
    const void* const* e = table + static_cast<size_t>(item) * 4;
    cur = Ptrs{e[0], const_cast<void*>(e[1]), static_cast<const float*>(e[2]), static_cast<const float*>(e[3])};
    rc = make_cursor<U>(g, lt);
    set_stride<U>(g, rc, G);
  };
  load_item();
  auto step_item = [&]() {  // move to the tile G further on
    lt += G;
    if (lt >= tiles_per_item) {
      do {
        lt -= tiles_per_item;
        ++item;
      } while (lt >= tiles_per_item);
      if (item < n_items) load_item();
    } else {
      advance(g, rc);
    }
  };
  Tile<U> ta, tb;
  RawPack<Tin> ra[U], rb[U];
  u32x2 ma[U], mb[U];
  Ptrs pa, pb;
  constexpr bool SPLIT = Tout::id == SBQ_F32;  // fp32 outputs: the split lane mapping of qdq_pack_kernel
#define SBQ_FETCH(P, T, R, M, IDX)                                 \
  P = cur;                                                         \
  locate<false, U, SPLIT>(g, lt, rc, P.scale, P.zp, T);            \
  issue_loads<Tin, MASK, true, U, SPLIT>(P.x, nullptr, T, R, M);   \
  step_item();                                                     \
  __builtin_amdgcn_sched_barrier(0); /* nothing of the following FINISH above this tile's loads */ \
  settle<U>(g, T)
#define SBQ_FINISH(P, T, R, M) \
  finish_tile<Tin, Tout, SBQ_Q_NONE, MASK, false, true, U, MATH_FAST, SPLIT>(P.y, nullptr, T, R, M, 0.0f, g.qlo, g.qhi)
  SBQ_FETCH(pa, ta, ra, ma, tile);
  while (static_cast<uint64_t>(tile) + 2ull * G < n_tiles_total) {
    SBQ_FETCH(pb, tb, rb, mb, tile + G);
    SBQ_FINISH(pa, ta, ra, ma);
    SBQ_FETCH(pa, ta, ra, ma, tile + 2 * G);
    SBQ_FINISH(pb, tb, rb, mb);
    tile += 2 * G;
  }
  if (static_cast<uint64_t>(tile) + G < n_tiles_total) {
    SBQ_FETCH(pb, tb, rb, mb, tile + G);
    SBQ_FINISH(pa, ta, ra, ma);
    SBQ_FINISH(pb, tb, rb, mb);
  } else {
    SBQ_FINISH(pa, ta, ra, ma);
  }
#undef SBQ_FETCH
#undef SBQ_FINISH
}

// Channels-last variant (inner == 1, C % 8 == 0): the NLC activation layout quantized per
// channel (quant_descriptor.py:36-47, ch_axis = 2).  Consecutive elements are consecutive
// channels, so a pack needs the 8 scales / zero points of channels c0..c0+7: two 16-byte loads
// each from vectors that stay in L1/L2.  The divisor differs per element: IEEE division.
template <typename Tin, typename Tout, int QT>
__global__ __launch_bounds__(kBlock) void qdq_clast_kernel(const void* __restrict__ x, void* __restrict__ y,
                                                           void* __restrict__ q, const float* __restrict__ scale,
                                                           const float* __restrict__ zero_point,
                                                           uint32_t total_packs, uint32_t C, float qlo, float qhi) {
  constexpr int U = 2;
  const uint32_t stride = gridDim.x * kBlock * U;
  for (uint32_t base = blockIdx.x * kBlock * U; base < total_packs; base += stride) {
    RawPack<Tin> raw[U];
    float s[U][kPack], z[U][kPack];
    uint32_t pk[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      pk[u] = base + u * kBlock + threadIdx.x;
      ok[u] = pk[u] < total_packs;
      if (!ok[u]) pk[u] = total_packs - 1;
      raw[u] = load_raw<Tin, true>(x, static_cast<int64_t>(pk[u]) * kPack);
      const uint32_t c0 = static_cast<uint32_t>((static_cast<uint64_t>(pk[u]) * kPack) % C);
      const u32x4 s0 = ld16<false>(scale + c0), s1 = ld16<false>(scale + c0 + 4);
      const u32x4 z0 = ld16<false>(zero_point + c0), z1 = ld16<false>(zero_point + c0 + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t a = s0[j], b = s1[j], c = z0[j], d = z1[j];
        s[u][j] = __builtin_bit_cast(float, a);
        s[u][4 + j] = __builtin_bit_cast(float, b);
        z[u][j] = __builtin_rintf(__builtin_bit_cast(float, c));
        z[u][4 + j] = __builtin_rintf(__builtin_bit_cast(float, d));
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float v[kPack], lv[kPack], dq[kPack];
      unpack_raw<Tin>(raw[u], v);
#pragma unroll
      for (int j = 0; j < kPack; ++j) {
        lv[j] = quant_level<SBQ_ROUND_HALF_EVEN>(v[j], s[u][j], z[u][j], qlo, qhi);
        dq[j] = dequant_level(lv[j], s[u][j], z[u][j]);
      }
      if (ok[u]) {
        const int64_t e = static_cast<int64_t>(pk[u]) * kPack;
        if (QT == SBQ_Q_NONE || y) store_pack<Tout, true>(y, e, dq);
        if constexpr (QT != SBQ_Q_NONE) store_q_pack<QT>(q, e, lv);
      }
    }
  }
}

// Scalar path: any geometry, any alignment, all rounding modes, runtime dtypes.
struct ScalarArgs {
  const void* x;
  void* y;
  void* q;
  const uint8_t* mask;
  const float* thresh;
  const float* scale;
  const float* zp;
  int64_t begin, end;  // element range handled
  int64_t inner;
  int64_t C;
  int x_dtype, y_dtype, q_type, rounding;
  float qlo, qhi;
  int lsq;
};

__device__ __forceinline__ float load_any(const void* p, int dt, int64_t i) {
  if (dt == SBQ_F32) return Elem<F32>::load1(p, i);
  if (dt == SBQ_F16) return Elem<F16>::load1(p, i);
  return Elem<BF16>::load1(p, i);
}
__device__ __forceinline__ void store_any(void* p, int dt, int64_t i, float v) {
  if (dt == SBQ_F32) Elem<F32>::store1(p, i, v);
  else if (dt == SBQ_F16) Elem<F16>::store1(p, i, v);
  else Elem<BF16>::store1(p, i, v);
}

__global__ __launch_bounds__(kBlock) void qdq_scalar_kernel(const ScalarArgs a) {
  float thr = 0.0f;
  if (a.thresh) thr = *a.thresh;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t i = a.begin + static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < a.end;
       i += stride) {
    const int64_t c = (i / a.inner) % a.C;
    float s = a.scale[c], z = a.zp[c];
    if (a.lsq) {
      s = __builtin_fabsf(s);
      z = __builtin_amdgcn_fmed3f(z, a.qlo, a.qhi);
    }
    z = __builtin_rintf(z);
    float xv = load_any(a.x, a.x_dtype, i);
    if (a.mask) xv = a.mask[i] ? xv : 0.0f;
    else if (a.thresh) xv = (__builtin_fabsf(xv) > thr) ? xv : 0.0f;
    float lv;
    if (a.rounding == SBQ_ROUND_HALF_EVEN)
      lv = quant_level<SBQ_ROUND_HALF_EVEN>(xv, s, z, a.qlo, a.qhi);
    else if (a.rounding == SBQ_ROUND_HALF_UP)
      lv = quant_level<SBQ_ROUND_HALF_UP>(xv, s, z, a.qlo, a.qhi);
    else
      lv = quant_level<SBQ_ROUND_HALF_DOWN>(xv, s, z, a.qlo, a.qhi);
    if (a.y) store_any(a.y, a.y_dtype, i, dequant_level(lv, s, z));
    if (a.q_type == SBQ_Q_I8) static_cast<int8_t*>(a.q)[i] = static_cast<int8_t>(static_cast<int>(lv));
    else if (a.q_type == SBQ_Q_I32) static_cast<int32_t*>(a.q)[i] = static_cast<int>(lv);
  }
}

// ---- host dispatch -------------------------------------------------------------------
// Grid policy (tools/qdq_sweep.py on 4096x4096, 11008x4096 and 32768x4096 bf16 weights):
//  * up to 2048 tiles everything is resident at once (8 workgroups of 256 per CU): one tile each;
//  * beyond that every workgroup gets at least two tiles, so that the second one's loads are in
//    flight under the first one's arithmetic, and the grid grows up to 8192 workgroups: only 2048
//    are resident, the rest are handed out by the dispatcher as workgroups retire, which evens
//    out the 10-15 % spread in how fast individual CUs get served (a fixed 4-per-CU grid left a
//    ~1.3 us straggler tail on the headline shape).
constexpr uint32_t kResidentWorkgroups = 256 * 8;
constexpr uint32_t kMaxGrid = 8192;

inline uint32_t auto_grid(uint32_t n_tiles) {
  if (knob(1) > 0) return n_tiles < static_cast<uint32_t>(knob(1)) ? n_tiles : static_cast<uint32_t>(knob(1));
  if (n_tiles <= kResidentWorkgroups) return n_tiles;
  // never fewer workgroups than fit at once: between one and two residencies' worth of tiles, a full
  // residency with one or two tiles each beats half-empty CUs with two each (measured on 2.4 k - 3.7 k tiles)
  uint32_t half = (n_tiles + 1) / 2;
  if (half < kResidentWorkgroups) half = kResidentWorkgroups;
  return half < kMaxGrid ? half : kMaxGrid;
}

struct QdqCall {
  QdqPtrs p;
  QdqGeom g;
  uint32_t rows;
};

template <typename Tin, typename Tout, int QT, int MASK, bool FLAT, bool NT, int U, int MATH = MATH_FAST, bool NTS = NT>
void launch_pack(const QdqCall& c, hipStream_t st) {
  const uint32_t grid = auto_grid(c.g.n_tiles);
  qdq_pack_kernel<Tin, Tout, QT, MASK, FLAT, NT, U, MATH, NTS><<<grid, kBlock, 0, st>>>(
      c.p.x, c.g.n_tiles, c.g.slabs_per_row, c.g.packs_per_row, c.g.C, c.g.inner, c.g.n_slabs, c.g.total_packs, c.p.scale,
      c.p.zp, c.p.y, c.p.q, c.p.mask, c.p.thresh, c.g.qlo, c.g.qhi, c.g.lsq);
}

// variant id (knob 0):  bit0-1: log2(U) (0..2) ; bit2: NT off ; -1 auto
template <typename Tin, typename Tout, int QT, int MASK, bool FLAT>
void launch_variant(QdqCall c, int variant, hipStream_t st) {
  int lu;
  bool nt = true;
  if (variant >= 0) {
    lu = variant & 3;
    if (lu > 2) lu = 2;
    nt = !(variant & 4);
  } else {
    // auto: one pack per lane per tile; two for very large tensors (>= 64 Mi elements), where the
    // longer per-workgroup loop amortises the wider tile
    lu = (!FLAT && c.g.n_slabs >= 32768u) ? 1 : 0;
  }
  const uint32_t U = 1u << lu;
  if (FLAT) c.g.n_tiles = (c.g.total_packs + kBlock * U - 1) / (kBlock * U);
  else c.g.n_tiles = (c.g.n_slabs + U - 1) / U;
#define SBQ_LAUNCH(NTV, UV) launch_pack<Tin, Tout, QT, MASK, FLAT, NTV, UV>(c, st)
  if constexpr (QT == SBQ_Q_NONE && MASK == MASK_NONE && Tin::id == SBQ_BF16 && Tout::id == SBQ_BF16 && !FLAT) {
    // development variants for the headline shape only (A/B measurements)
    const int math = knob(2);
    if (math == MATH_IEEE) {
#define SBQ_LAUNCH_M(UV) launch_pack<Tin, Tout, QT, MASK, FLAT, true, UV, MATH_IEEE>(c, st)
      if (U == 1) SBQ_LAUNCH_M(1);
      else if (U == 2) SBQ_LAUNCH_M(2);
      else SBQ_LAUNCH_M(4);
#undef SBQ_LAUNCH_M
      return;
    }
    if (variant >= 0 && (variant & 8)) {  // mixed cache policy: bit3 set -> loads nt, stores cached;
      if (variant & 4) launch_pack<Tin, Tout, QT, MASK, FLAT, false, 1, MATH_FAST, true>(c, st);  // + bit2 -> loads cached, stores nt
      else launch_pack<Tin, Tout, QT, MASK, FLAT, true, 1, MATH_FAST, false>(c, st);
      return;
    }
    if (!nt) {
      if (U == 1) SBQ_LAUNCH(false, 1);
      else if (U == 2) SBQ_LAUNCH(false, 2);
      else SBQ_LAUNCH(false, 4);
      return;
    }
  }
  if (U == 1) SBQ_LAUNCH(true, 1);
  else if (U == 2) SBQ_LAUNCH(true, 2);
  else SBQ_LAUNCH(true, 4);
#undef SBQ_LAUNCH
}

template <typename Tin, typename Tout, int QT, int MASK>
void launch_geom(const QdqCall& c, bool flat, hipStream_t st) {
  const int variant = knob(0);
  if constexpr (QT == SBQ_Q_NONE) {
    if (!flat && variant < 0) {
      const ResidentCall r{c.p.x, c.p.y, c.p.mask, c.p.thresh, c.p.scale, c.p.zp, c.g.packs_per_row, c.g.slabs_per_row,
                           c.g.n_slabs, c.rows, c.g.C, c.g.qlo, c.g.qhi, c.g.lsq, Tin::id, Tout::id};
      if (qdq_try_resident(r, st)) return;
    }
  }
  if (flat) launch_variant<Tin, Tout, QT, MASK, true>(c, variant, st);
  else launch_variant<Tin, Tout, QT, MASK, false>(c, variant, st);
}

template <typename Tin, typename Tout, int QT>
void launch_mask(const QdqCall& c, bool flat, hipStream_t st) {
  if (c.p.mask) launch_geom<Tin, Tout, QT, MASK_BYTES>(c, flat, st);
  else if (c.p.thresh) launch_geom<Tin, Tout, QT, MASK_THRESH>(c, flat, st);
  else launch_geom<Tin, Tout, QT, MASK_NONE>(c, flat, st);
}

template <typename Tin, typename Tout>
void launch_q(const QdqCall& c, int q_type, bool flat, hipStream_t st) {
  if (q_type == SBQ_Q_I8) launch_mask<Tin, Tout, SBQ_Q_I8>(c, flat, st);
  else if (q_type == SBQ_Q_I4) launch_geom<Tin, Tout, SBQ_Q_I4, MASK_NONE>(c, flat, st);  // no mask variants
  else if (q_type == SBQ_Q_I32) launch_mask<Tin, Tout, SBQ_Q_I32>(c, flat, st);
  else launch_mask<Tin, Tout, SBQ_Q_NONE>(c, flat, st);
}

void launch_scalar(ScalarArgs a, hipStream_t st) {
  const int64_t n = a.end - a.begin;
  if (n <= 0) return;
  int64_t blocks = ceil_div(n, kBlock);
  if (blocks > static_cast<int64_t>(kMaxGrid)) blocks = kMaxGrid;
  qdq_scalar_kernel<<<static_cast<uint32_t>(blocks), kBlock, 0, st>>>(a);
}

int qdq_forward(const void* x, int x_dtype, void* y, int y_dtype, void* q, int q_type,
                const uint8_t* mask, const float* thresh, const float* scale, const float* zp,
                int64_t outer, int64_t C, int64_t inner, int qmin, int qmax, int rounding,
                void* stream, int lsq = 0) {
  if (!valid_dtype(x_dtype) || !valid_dtype(y_dtype)) return SBQ_ERR_DTYPE;
  if (y_dtype != SBQ_F32 && y_dtype != x_dtype) return SBQ_ERR_DTYPE;
  if (q_type != SBQ_Q_NONE && q_type != SBQ_Q_I8 && q_type != SBQ_Q_I32 && q_type != SBQ_Q_I4) return SBQ_ERR_DTYPE;
  if (outer < 0 || C < 0 || inner < 0) return SBQ_ERR_ARG;
  if (outer == 0 || C == 0 || inner == 0) return SBQ_ERR_EMPTY;
  if (!x || !scale || !zp) return SBQ_ERR_NULL;
  if (q_type != SBQ_Q_NONE && !q) return SBQ_ERR_NULL;
  if (!y && q_type == SBQ_Q_NONE) return SBQ_ERR_NULL;  // y may be NULL only in quantize-only mode
  if (qmin > qmax) return SBQ_ERR_ARG;
  if (q_type == SBQ_Q_I8 && static_cast<int64_t>(qmax) - qmin > 255) return SBQ_ERR_ARG;
  // packed int4: 16 levels, whole 8-element packs only (two elements share a byte: no scalar
  // path), the even-rounding pack kernels, no fused mask
  if (q_type == SBQ_Q_I4 && (static_cast<int64_t>(qmax) - qmin > 15 || mask || thresh ||
                             rounding != SBQ_ROUND_HALF_EVEN))
    return SBQ_ERR_ARG;
  if (rounding < 0 || rounding > 2) return SBQ_ERR_ARG;
  if (mask && thresh) return SBQ_ERR_ARG;
  if (C > 0x7fffffff) return SBQ_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(x) % dtype_size(x_dtype)) ||
      (reinterpret_cast<uintptr_t>(y) % dtype_size(y_dtype)))
    return SBQ_ERR_ALIGN;

  hipStream_t st = as_stream(stream);
  const int64_t rows = outer * C;
  const int64_t numel = rows * inner;

  ScalarArgs sa{x, y, q, mask, thresh, scale, zp, 0, numel, inner, C,
                x_dtype, y_dtype, q_type, rounding, static_cast<float>(qmin), static_cast<float>(qmax), lsq};

  const bool ptr_ok = aligned16(x) && aligned16(y) && (q_type == SBQ_Q_NONE || aligned16(q)) &&
                      (!mask || (reinterpret_cast<uintptr_t>(mask) & 7u) == 0);
  // channels-last per-channel (NLC activations): inner == 1, whole packs of channels
  if (inner == 1 && C > 1 && C % kPack == 0 && rounding == SBQ_ROUND_HALF_EVEN && ptr_ok && !mask && !thresh && !lsq &&
      aligned16(scale) && aligned16(zp) && numel / kPack < (1ll << 31)) {
    const uint32_t packs = static_cast<uint32_t>(numel / kPack);
    uint32_t grid = (packs + kBlock * 2 - 1) / (kBlock * 2);
    if (grid > kMaxGrid) grid = kMaxGrid;
    const float qlo = static_cast<float>(qmin), qhi = static_cast<float>(qmax);
#define SBQ_CL(TI, TO)                                                                                         \
  do {                                                                                                         \
    if (q_type == SBQ_Q_I8) qdq_clast_kernel<TI, TO, SBQ_Q_I8><<<grid, kBlock, 0, st>>>(x, y, q, scale, zp, packs, static_cast<uint32_t>(C), qlo, qhi); \
    else if (q_type == SBQ_Q_I32) qdq_clast_kernel<TI, TO, SBQ_Q_I32><<<grid, kBlock, 0, st>>>(x, y, q, scale, zp, packs, static_cast<uint32_t>(C), qlo, qhi); \
    else if (q_type == SBQ_Q_I4) qdq_clast_kernel<TI, TO, SBQ_Q_I4><<<grid, kBlock, 0, st>>>(x, y, q, scale, zp, packs, static_cast<uint32_t>(C), qlo, qhi); \
    else qdq_clast_kernel<TI, TO, SBQ_Q_NONE><<<grid, kBlock, 0, st>>>(x, y, q, scale, zp, packs, static_cast<uint32_t>(C), qlo, qhi); \
  } while (0)
    if (x_dtype == SBQ_F32) SBQ_CL(F32, F32);
    else if (x_dtype == SBQ_F16) { if (y_dtype == SBQ_F32) SBQ_CL(F16, F32); else SBQ_CL(F16, F16); }
    else { if (y_dtype == SBQ_F32) SBQ_CL(BF16, F32); else SBQ_CL(BF16, BF16); }
#undef SBQ_CL
    return check_launch();
  }

  // The pack kernels need rows made of whole 8-element packs.  A per-tensor
  // call (C == 1) is one long row, so only its last numel % 8 elements are ragged.
  const int64_t body = (C == 1) ? (numel / kPack) * kPack : (inner % kPack == 0 ? numel : 0);
  const uint64_t total_packs = static_cast<uint64_t>(body / kPack);
  if (q_type == SBQ_Q_I4 && body != numel) return SBQ_ERR_ARG;  // no ragged tail in packed int4
  // 32-bit pack / tile arithmetic in the kernels: < 2^31 packs (16 Gi elements)
  if (rounding != SBQ_ROUND_HALF_EVEN || !ptr_ok || body == 0 || total_packs >= (1ull << 31) ||
      rows >= (1ll << 31)) {
    if (q_type == SBQ_Q_I4) return ptr_ok ? SBQ_ERR_ARG : SBQ_ERR_ALIGN;
    launch_scalar(sa, st);
    return check_launch();
  }

  QdqCall c{};
  c.g.lsq = lsq ? 1u : 0u;
  c.p = QdqPtrs{x, y, q, mask, thresh, scale, zp};
  c.g.C = static_cast<uint32_t>(C);
  c.g.qlo = static_cast<float>(qmin);
  c.g.qhi = static_cast<float>(qmax);
  c.g.total_packs = static_cast<uint32_t>(total_packs);
  bool flat;
  if (C == 1) {  // per tensor: one row of `body` elements, cut into slabs like any other row
    c.g.inner = body;
    c.g.packs_per_row = c.g.total_packs;
    c.rows = 1;
    flat = false;
  } else {
    c.g.inner = inner;
    c.g.packs_per_row = static_cast<uint32_t>(inner / kPack);
    c.rows = static_cast<uint32_t>(rows);
    flat = c.g.packs_per_row < static_cast<uint32_t>(kBlock);  // short rows: keep lanes busy
  }
  c.g.slabs_per_row = (c.g.packs_per_row + kBlock - 1) / kBlock;
  if (!flat && static_cast<uint64_t>(c.rows) * c.g.slabs_per_row >= (1ull << 31)) flat = true;
  c.g.n_slabs = flat ? 0 : c.rows * c.g.slabs_per_row;

#define SBQ_DISPATCH(TI, TO) launch_q<TI, TO>(c, q_type, flat, st)
  if (x_dtype == SBQ_F32) SBQ_DISPATCH(F32, F32);
  else if (x_dtype == SBQ_F16) { if (y_dtype == SBQ_F32) SBQ_DISPATCH(F16, F32); else SBQ_DISPATCH(F16, F16); }
  else { if (y_dtype == SBQ_F32) SBQ_DISPATCH(BF16, F32); else SBQ_DISPATCH(BF16, BF16); }
#undef SBQ_DISPATCH
  int rc = check_launch();
  if (rc != SBQ_OK) return rc;
  if (body < numel) {  // ragged per-tensor tail (< 8 elements)
    sa.begin = body;
    launch_scalar(sa, st);
    rc = check_launch();
  }
  return rc;
}

int qdq_forward_batched(const void* const* table, int n_items, int x_dtype, int y_dtype, int64_t outer,
                        int64_t C, int64_t inner, int qmin, int qmax, void* stream) {
  if (!valid_dtype(x_dtype) || !valid_dtype(y_dtype)) return SBQ_ERR_DTYPE;
  if (y_dtype != SBQ_F32 && y_dtype != x_dtype) return SBQ_ERR_DTYPE;
  if (outer < 0 || C < 0 || inner < 0 || n_items < 0) return SBQ_ERR_ARG;
  if (outer == 0 || C == 0 || inner == 0 || n_items == 0) return SBQ_ERR_EMPTY;
  if (!table) return SBQ_ERR_NULL;
  if (n_items > SBQ_MAX_BATCH || qmin > qmax || C > 0x7fffffff) return SBQ_ERR_ARG;
  if (inner % kPack != 0) return SBQ_ERR_ARG;  // pack kernel only; ragged tensors go one by one
  const int64_t rows = outer * C;
  const uint64_t packs = static_cast<uint64_t>(rows) * (inner / kPack);
  if (rows >= (1ll << 31) || packs * n_items >= (1ull << 31)) return SBQ_ERR_ARG;
  QdqGeom g{};
  g.inner = inner;
  g.C = static_cast<uint32_t>(C);
  g.packs_per_row = static_cast<uint32_t>(inner / kPack);
  g.slabs_per_row = (g.packs_per_row + kBlock - 1) / kBlock;
  g.n_slabs = static_cast<uint32_t>(rows) * g.slabs_per_row;
  g.total_packs = static_cast<uint32_t>(packs);
  g.qlo = static_cast<float>(qmin);
  g.qhi = static_cast<float>(qmax);
  constexpr uint32_t U = 2;
  const uint32_t tiles_per_item = (g.n_slabs + U - 1) / U;
  g.n_tiles = tiles_per_item;
  const uint32_t total = tiles_per_item * static_cast<uint32_t>(n_items);
  const uint32_t grid = auto_grid(total);
  hipStream_t st = as_stream(stream);
#define SBQ_B(TI, TO) \
  qdq_batched_kernel<TI, TO, U><<<grid, kBlock, 0, st>>>(table, static_cast<uint32_t>(n_items), tiles_per_item, total, g)
  if (x_dtype == SBQ_F32) SBQ_B(F32, F32);
  else if (x_dtype == SBQ_F16) { if (y_dtype == SBQ_F32) SBQ_B(F16, F32); else SBQ_B(F16, F16); }
  else { if (y_dtype == SBQ_F32) SBQ_B(BF16, F32); else SBQ_B(BF16, BF16); }
#undef SBQ_B
  return check_launch();
}

}  // namespace
}  // namespace sbq

extern "C" {

int sbq_quant_perchannel_forward_batched(const void* const* table, int n_items, int x_dtype, int y_dtype,
                                         int64_t outer, int64_t C, int64_t inner, int qmin, int qmax,
                                         void* stream) {
  return sbq::qdq_forward_batched(table, n_items, x_dtype, y_dtype, outer, C, inner, qmin, qmax, stream);
}

int sbq_quant_pertensor_forward(const void* x, int x_dtype, void* y, int y_dtype, void* q, int q_type,
                                const float* scale, const float* zero_point, int64_t numel,
                                int qmin, int qmax, int rounding, void* stream) {
  return sbq::qdq_forward(x, x_dtype, y, y_dtype, q, q_type, nullptr, nullptr, scale, zero_point,
                          1, 1, numel, qmin, qmax, rounding, stream);
}

int sbq_quant_perchannel_forward(const void* x, int x_dtype, void* y, int y_dtype, void* q, int q_type,
                                 const float* scale, const float* zero_point,
                                 int64_t outer, int64_t C, int64_t inner,
                                 int qmin, int qmax, int rounding, void* stream) {
  return sbq::qdq_forward(x, x_dtype, y, y_dtype, q, q_type, nullptr, nullptr, scale, zero_point,
                          outer, C, inner, qmin, qmax, rounding, stream);
}

int sbq_quant_lsq_forward(const void* x, int x_dtype, void* y, int y_dtype, const uint8_t* mask,
                          const float* scale, const float* zero_point, int64_t outer, int64_t C,
                          int64_t inner, int qmin, int qmax, void* stream) {
  return sbq::qdq_forward(x, x_dtype, y, y_dtype, nullptr, SBQ_Q_NONE, mask, nullptr, scale, zero_point, outer, C,
                          inner, qmin, qmax, SBQ_ROUND_HALF_EVEN, stream, 1);
}

int sbq_mask_quant_forward(const void* x, int x_dtype, void* y, int y_dtype, void* q, int q_type,
                           const uint8_t* mask, const float* thresh,
                           const float* scale, const float* zero_point,
                           int64_t outer, int64_t C, int64_t inner,
                           int qmin, int qmax, int rounding, void* stream) {
  if ((mask == nullptr) == (thresh == nullptr)) return SBQ_ERR_ARG;
  return sbq::qdq_forward(x, x_dtype, y, y_dtype, q, q_type, mask, thresh, scale, zero_point,
                          outer, C, inner, qmin, qmax, rounding, stream);
}

}  // extern "C"
