// sbq_group.hip -- one launch for the fake-quant forward of a whole MODEL's weights.
//
// The reference quantizes every layer's weight with its own kernel launch
// (QuantOpr.forward -> weight_quantizer(weight), e.g. modules/conv.py:30-36; K1/K2 of
// fake_quant_tensor.cu).  For the CNN / ViT configs the weights are small (ResNet-50: 54
// tensors, 2.4 M elements at most) and every launch is latency: ~4 us of launch + a kernel
// that is over before the chip has filled.  A QAT step re-quantizes all of them.
// Here the tensors of a model -- any mix of [C, inner] shapes, per-channel or per-tensor,
// each with its own integer range -- are described once in a device table and quantized by
// ONE grid: tile -> item through a flat index array, item descriptor through scalar loads,
// then the flat pack mapping (256 packs of 8 elements per tile, channel per lane).
// Arithmetic is the single-tensor kernels': exact x/s by reciprocal + two fma refinements
// with the wave-vote IEEE fallback (sbq_common.hpp), half-to-even, fp32 throughout.
#include "sbq_common.hpp"

namespace sbq {
namespace {

struct GroupItemDev {  // 80 bytes, read through the scalar unit
  const void* x;
  void* y;
  const float* scale;
  const float* zp;
  const uint8_t* mask;
  uint32_t C;
  uint32_t packs_per_row;
  uint32_t total_packs;
  uint32_t tile_begin;
  float qlo, qhi;
  uint32_t flags;
  uint32_t pad[3];
};
static_assert(sizeof(GroupItemDev) == 80, "device table layout");

constexpr size_t kGroupHeaderBytes = 64;

struct GroupLane {
  const GroupItemDev* it;  // block-uniform
  int64_t elem;
  bool ok;
  float s, z;
};

template <typename T>
__device__ __forceinline__ T uniform_read(const T* p) {
  typedef const T __attribute__((address_space(4))) * cptr;
  return *reinterpret_cast<cptr>(reinterpret_cast<uintptr_t>(p));
}

__device__ __forceinline__ GroupLane group_locate(const GroupItemDev* __restrict__ items,
                                                  const uint32_t* __restrict__ tile_item, uint32_t tile) {
  GroupLane L;
  const uint32_t idx = uniform_read(tile_item + tile);
  const GroupItemDev* it = items + idx;
  L.it = it;
  const uint32_t total = uniform_read(&it->total_packs);
  const uint32_t ppr = uniform_read(&it->packs_per_row);
  const uint32_t C = uniform_read(&it->C);
  const uint32_t pk = (tile - uniform_read(&it->tile_begin)) * kBlock + threadIdx.x;
  L.ok = pk < total;
  const uint32_t pkc = L.ok ? pk : total - 1;  // loads are never predicated, only stores
  L.elem = static_cast<int64_t>(pkc) * kPack;
  uint32_t c = 0;
  if (C != 1) {
    // row = pkc / ppr without an integer division: pkc < 2^24 is exact in fp32, the quotient
    // estimate is off by at most one
    const float inv = 1.0f / static_cast<float>(ppr);
    uint32_t r = static_cast<uint32_t>(static_cast<float>(pkc) * inv);
    const int32_t rem = static_cast<int32_t>(pkc - r * ppr);
    if (rem < 0) --r;
    else if (rem >= static_cast<int32_t>(ppr)) ++r;
    c = r;  // weights: outer == 1, the row IS the channel
  }
  const float* sc = uniform_read(&it->scale);
  const float* zp = uniform_read(&it->zp);
  float s = sc[c], z = zp[c];
  const float qlo = uniform_read(&it->qlo), qhi = uniform_read(&it->qhi);
  if (uniform_read(&it->flags) & SBQ_GROUP_LSQ) {  // lsq.py:61-62: s = |s|, zp = clamp(zp, qmin, qmax)
    s = __builtin_fabsf(s);
    z = __builtin_amdgcn_fmed3f(z, qlo, qhi);
  }
  L.s = s;
  L.z = __builtin_rintf(z);
  return L;
}

template <typename Tin, typename Tout, bool HAS_MASK>
__device__ __forceinline__ void group_finish(const GroupLane& L, const RawPack<Tin>& raw, const u32x2& mk,
                                             char* y_base) {
  float v[kPack], dq[kPack];
  unpack_raw<Tin>(raw, v);
  if constexpr (HAS_MASK) {
#pragma unroll
    for (int j = 0; j < kPack; ++j) {
      const uint32_t byte = (mk[j >> 2] >> (8 * (j & 3))) & 0xffu;
      v[j] = byte ? v[j] : 0.0f;
    }
  }
  const float s = L.s, z = L.z;
  const float qlo = uniform_read(&L.it->qlo), qhi = uniform_read(&L.it->qhi);
  // the scale differs per lane here: one IEEE reciprocal per pack, then the exact fma refinement
  // per element; any lane outside its range sends the whole wave through IEEE division
  const float yr = 1.0f / s;
  const float bound = s * 0x1p40f;
  bool odd = !fast_div_ok(s);
#pragma unroll
  for (int j = 0; j < kPack; ++j) odd |= !(__builtin_fabsf(v[j]) < bound);
  if (__builtin_amdgcn_ballot_w64(odd) == 0) {
#pragma unroll
    for (int j = 0; j < kPack; j += 2) {  // the fma chain on packed fp32 ops: two quotients per instruction
      const f32x2 t = fast_div2(f32x2{v[j], v[j + 1]}, s, yr);
      dq[j] = dequant_level(__builtin_amdgcn_fmed3f(__builtin_rintf(t[0]) + z, qlo, qhi), s, z);
      dq[j + 1] = dequant_level(__builtin_amdgcn_fmed3f(__builtin_rintf(t[1]) + z, qlo, qhi), s, z);
    }
  } else {
#pragma unroll
    for (int j = 0; j < kPack; ++j)
      dq[j] = dequant_level(quant_level<SBQ_ROUND_HALF_EVEN>(v[j], s, z, qlo, qhi), s, z);
  }
  // y_base != nullptr: the item's y is a byte offset into one flat output buffer
  char* y = y_base + reinterpret_cast<uintptr_t>(uniform_read(&L.it->y));
  if (L.ok) store_pack<Tout, false>(y, L.elem, dq);
}

// Cached (not nontemporal) accesses on purpose: in a training step the weights were just
// written by the optimizer and the quantized copies are consumed by the very next GEMM /
// convolution, and a whole CNN's weights fit the 256 MB Infinity Cache.
template <typename Tin, typename Tout, bool HAS_MASK>
__global__ __launch_bounds__(kBlock) void qdq_group_kernel(const GroupItemDev* __restrict__ items,
                                                           const uint32_t* __restrict__ tile_item,
                                                           uint32_t n_tiles, char* y_base) {
  uint32_t tile = blockIdx.x;
  const uint32_t G = gridDim.x;
  if (tile >= n_tiles) return;
  GroupLane la, lb;
  RawPack<Tin> ra, rb;
  u32x2 ma = {0, 0}, mb = {0, 0};
#define SBQ_FETCH(L, R, M, IDX)                                                  \
  L = group_locate(items, tile_item, (IDX));                                     \
  R = load_raw<Tin, false>(uniform_read(&L.it->x), L.elem);                      \
  if constexpr (HAS_MASK) M = ld8<false>(uniform_read(&L.it->mask) + L.elem)
  SBQ_FETCH(la, ra, ma, tile);
  while (static_cast<uint64_t>(tile) + 2ull * G < n_tiles) {  // two-stage software pipeline
    SBQ_FETCH(lb, rb, mb, tile + G);
    group_finish<Tin, Tout, HAS_MASK>(la, ra, ma, y_base);
    SBQ_FETCH(la, ra, ma, tile + 2 * G);
    group_finish<Tin, Tout, HAS_MASK>(lb, rb, mb, y_base);
    tile += 2 * G;
  }
  if (static_cast<uint64_t>(tile) + G < n_tiles) {
    SBQ_FETCH(lb, rb, mb, tile + G);
    group_finish<Tin, Tout, HAS_MASK>(la, ra, ma, y_base);
    group_finish<Tin, Tout, HAS_MASK>(lb, rb, mb, y_base);
  } else {
    group_finish<Tin, Tout, HAS_MASK>(la, ra, ma, y_base);
  }
#undef SBQ_FETCH
}

int group_check_item(const sbq_group_item& it, bool want_mask) {
  if (!it.x || !it.scale || !it.zero_point) return SBQ_ERR_NULL;
  if (!it.y && !(it.flags & SBQ_GROUP_Y_OFFSET)) return SBQ_ERR_NULL;  // offset 0 is a valid offset
  if (want_mask != (it.mask != nullptr)) return SBQ_ERR_ARG;  // a group is all-masked or mask-free
  if (it.C < 0 || it.inner < 0) return SBQ_ERR_ARG;
  if (it.C == 0 || it.inner == 0) return SBQ_ERR_EMPTY;
  if (it.qmin > it.qmax || it.C > 0x7fffffff) return SBQ_ERR_ARG;
  if (it.inner % kPack != 0) return SBQ_ERR_ARG;  // whole 8-element packs per row
  if (static_cast<uint64_t>(it.C) * static_cast<uint64_t>(it.inner / kPack) >= (1ull << 24)) return SBQ_ERR_ARG;
  if (!aligned16(it.x) || !aligned16(it.y) || (reinterpret_cast<uintptr_t>(it.scale) & 3u) ||
      (reinterpret_cast<uintptr_t>(it.zero_point) & 3u) || (reinterpret_cast<uintptr_t>(it.mask) & 7u))
    return SBQ_ERR_ALIGN;
  return SBQ_OK;
}

// ---- model-wide STE backward ------------------------------------------------------------------
// gx and the LSQ step-size gradients of every weight quantizer of a model in TWO launches
// (the reference: QuantizePer{Tensor,Channel}BackwardCUDA per layer, fake_quant_tensor.cu:97-132,
// 227-270, plus the gs_scaling / abs autograd nodes of lsq.py:13-21,61-76).  Semantics are
// sbq_backward.hip's (K3 == MySTE.backward), with the mask of a sparse layer applied on both
// sides: gx = mask * STE'(mask * w).
// Work unit: a WAVE takes one segment = up to 64 packs (512 elements) of one channel row, so the
// reduction of gs never crosses a row and needs no LDS; every segment writes one fp64 partial and
// a second tiny kernel folds the segments of a row in ascending order (deterministic) and applies
// LSQ's gradient scaling and sign(scale).  Workgroups (4 segments) never straddle items: each
// item's segment count is padded to a multiple of 4.
struct GroupBwdItemDev {  // 96 bytes
  const void* x;
  uint64_t gx_off;   // bytes from gx_base
  const float* scale;
  const float* zp;
  const uint8_t* mask;
  uint64_t gs_off;   // floats from gs_base; ~0 = this item wants no scale gradient
  uint32_t C, packs_per_row, segs_per_row, wg_begin;
  uint32_t seg_begin, row_begin;
  float qlo, qhi;
  float ratio;
  uint32_t flags;
  uint32_t pad[2];
};
static_assert(sizeof(GroupBwdItemDev) == 96, "device table layout");

constexpr int kSegPacks = kWave;  // packs per segment
constexpr int kMaxGroupGy = SBQ_GROUP_BWD_CHUNK;

struct GyPtrs {
  const void* p[kMaxGroupGy];
};

__device__ __forceinline__ uint32_t exact_udiv(uint32_t a, uint32_t b) {  // a < 2^24
  uint32_t r = static_cast<uint32_t>(static_cast<float>(a) * (1.0f / static_cast<float>(b)));
  const int32_t rem = static_cast<int32_t>(a - r * b);
  if (rem < 0) --r;
  else if (rem >= static_cast<int32_t>(b)) ++r;
  return r;
}

template <typename T, typename Tg, bool HAS_MASK>
__global__ __launch_bounds__(kBlock) void group_bwd_kernel(const GroupBwdItemDev* __restrict__ items,
                                                           const uint32_t* __restrict__ wg_item,
                                                           const GyPtrs gys, uint32_t item0, uint32_t wg0,
                                                           char* gx_base, double* __restrict__ part) {
  const uint32_t wg = wg0 + blockIdx.x;
  const uint32_t idx = uniform_read(wg_item + wg);
  const GroupBwdItemDev* it = items + idx;
  const uint32_t C = uniform_read(&it->C), ppr = uniform_read(&it->packs_per_row);
  const uint32_t spr = uniform_read(&it->segs_per_row);
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t seg = (wg - uniform_read(&it->wg_begin)) * kWavesPerBlock + threadIdx.x / kWave;
  if (seg >= C * spr) return;  // padding wave of the item's last workgroup
  const uint32_t row = exact_udiv(seg, spr);
  const uint32_t pack = (seg - row * spr) * kSegPacks + lane;
  const bool ok = pack < ppr;
  const uint32_t pkc = ok ? pack : ppr - 1;
  const int64_t elem = (static_cast<int64_t>(row) * ppr + pkc) * kPack;
  const float* sc = uniform_read(&it->scale);
  const float* zpp = uniform_read(&it->zp);
  float s = sc[row], z = zpp[row];
  const float qlo = uniform_read(&it->qlo), qhi = uniform_read(&it->qhi);
  if (uniform_read(&it->flags) & SBQ_GROUP_LSQ) {
    s = __builtin_fabsf(s);
    z = __builtin_amdgcn_fmed3f(z, qlo, qhi);
  }
  z = __builtin_rintf(z);
  float xv[kPack], gv[kPack], o[kPack];
  load_pack<T, false>(uniform_read(&it->x), elem, xv);
  load_pack<T, false>(gys.p[idx - item0], elem, gv);
  u32x2 mk = {0, 0};
  if constexpr (HAS_MASK) mk = ld8<false>(uniform_read(&it->mask) + elem);
  if constexpr (HAS_MASK) {
#pragma unroll
    for (int j = 0; j < kPack; ++j) {
      const uint32_t byte = (mk[j >> 2] >> (8 * (j & 3))) & 0xffu;
      xv[j] = byte ? xv[j] : 0.0f;
    }
  }
  // the row's scale is wave-uniform: exact quotient by reciprocal + fma refinement, IEEE division
  // for the whole wave if any element is outside the refinement's range
  const float yr = 1.0f / s;
  const float bound = s * 0x1p40f;
  bool odd = !fast_div_ok(s);
#pragma unroll
  for (int j = 0; j < kPack; ++j) odd |= !(__builtin_fabsf(xv[j]) < bound);
  float gs = 0.0f;
  auto one = [&](float t, float gyv) -> float {  // t = x / s, correctly rounded
    const float r = __builtin_rintf(t);
    const float v = r + z;
    const bool below = v < qlo, above = v > qhi;
    float pgs = (r - t) * gyv;
    if (above) pgs = (qhi - z) * gyv;
    if (below) pgs = (qlo - z) * gyv;
    gs += pgs;
    return (below || above) ? 0.0f : gyv;  // NaN counts as inside, like sbq_backward.hip
  };
  if (__builtin_amdgcn_ballot_w64(odd) == 0) {
#pragma unroll
    for (int j = 0; j < kPack; ++j) o[j] = one(fast_div(xv[j], s, yr), gv[j]);
  } else {
#pragma unroll
    for (int j = 0; j < kPack; ++j) o[j] = one(xv[j] / s, gv[j]);
  }
  if constexpr (HAS_MASK) {
#pragma unroll
    for (int j = 0; j < kPack; ++j) {
      const uint32_t byte = (mk[j >> 2] >> (8 * (j & 3))) & 0xffu;
      o[j] = byte ? o[j] : 0.0f;
    }
  }
  if (ok) store_pack<Tg, false>(gx_base + uniform_read(&it->gx_off), elem, o);
  if (uniform_read(&it->gs_off) != ~0ull) {
    const double tot = wave_reduce(ok ? static_cast<double>(gs) : 0.0, Sum());
    if (lane == 0) part[uniform_read(&it->seg_begin) + seg] = tot;
  }
}

__global__ __launch_bounds__(kBlock) void group_bwd_fold_kernel(const GroupBwdItemDev* __restrict__ items,
                                                                uint32_t n_items, uint32_t n_rows,
                                                                const double* __restrict__ part,
                                                                float* __restrict__ gs_base) {
  const uint32_t r = blockIdx.x * kBlock + threadIdx.x;
  if (r >= n_rows) return;
  uint32_t lo = 0, hi = n_items - 1;  // last item with row_begin <= r
  while (lo < hi) {
    const uint32_t mid = (lo + hi + 1) >> 1;
    if (items[mid].row_begin <= r) lo = mid;
    else hi = mid - 1;
  }
  const GroupBwdItemDev& it = items[lo];
  if (it.gs_off == ~0ull) return;
  const uint32_t row = r - it.row_begin;
  const double* p = part + it.seg_begin + static_cast<size_t>(row) * it.segs_per_row;
  double a = 0.0;
  for (uint32_t k = 0; k < it.segs_per_row; ++k) a += p[k];
  float g = static_cast<float>(a);
  if (it.flags & SBQ_GROUP_LSQ) {  // lsq.py:13-21,61: d|s|/ds = sign(s); gs_scaling multiplies by ratio
    const float sraw = it.scale[row];
    const float sign = sraw > 0.0f ? 1.0f : (sraw < 0.0f ? -1.0f : 0.0f);
    g = (g * it.ratio) * sign;
  }
  gs_base[it.gs_off + row] = g;
}

}  // namespace
}  // namespace sbq

extern "C" {

int sbq_group_table_build(const sbq_group_item* items, int n_items, void* host_table, size_t host_table_bytes,
                          uint32_t* n_tiles_out, size_t* bytes_needed_out) {
  using namespace sbq;
  if (n_items < 0) return SBQ_ERR_ARG;
  if (n_items == 0) return SBQ_ERR_EMPTY;
  if (!items) return SBQ_ERR_NULL;
  const bool want_mask = items[0].mask != nullptr;
  uint64_t tiles = 0;
  for (int i = 0; i < n_items; ++i) {
    const int rc = group_check_item(items[i], want_mask);
    if (rc != SBQ_OK) return rc;
    const uint64_t packs = static_cast<uint64_t>(items[i].C) * static_cast<uint64_t>(items[i].inner / kPack);
    tiles += (packs + kBlock - 1) / kBlock;
  }
  if (tiles >= (1ull << 31)) return SBQ_ERR_ARG;
  const size_t need = kGroupHeaderBytes + static_cast<size_t>(n_items) * sizeof(GroupItemDev) +
                      static_cast<size_t>(tiles) * sizeof(uint32_t);
  if (n_tiles_out) *n_tiles_out = static_cast<uint32_t>(tiles);
  if (bytes_needed_out) *bytes_needed_out = need;
  if (!host_table) return SBQ_OK;  // size query
  if (host_table_bytes < need) return SBQ_ERR_WORKSPACE;
  char* base = static_cast<char*>(host_table);
  uint32_t* header = reinterpret_cast<uint32_t*>(base);
  for (size_t i = 0; i < kGroupHeaderBytes / 4; ++i) header[i] = 0;
  header[0] = static_cast<uint32_t>(n_items);
  header[1] = static_cast<uint32_t>(tiles);
  GroupItemDev* dev = reinterpret_cast<GroupItemDev*>(base + kGroupHeaderBytes);
  uint32_t* tile_item = reinterpret_cast<uint32_t*>(base + kGroupHeaderBytes + static_cast<size_t>(n_items) * sizeof(GroupItemDev));
  uint32_t t = 0;
  for (int i = 0; i < n_items; ++i) {
    const sbq_group_item& it = items[i];
    GroupItemDev d{};
    d.x = it.x;
    d.y = it.y;
    d.scale = it.scale;
    d.zp = it.zero_point;
    d.mask = it.mask;
    d.C = static_cast<uint32_t>(it.C);
    d.packs_per_row = static_cast<uint32_t>(it.inner / kPack);
    d.total_packs = d.C * d.packs_per_row;
    d.tile_begin = t;
    d.qlo = static_cast<float>(it.qmin);
    d.qhi = static_cast<float>(it.qmax);
    d.flags = it.flags;
    dev[i] = d;
    const uint32_t nt = (d.total_packs + kBlock - 1) / kBlock;
    for (uint32_t k = 0; k < nt; ++k) tile_item[t + k] = static_cast<uint32_t>(i);
    t += nt;
  }
  return SBQ_OK;
}

int sbq_quant_group_forward(const void* device_table, int n_items, uint32_t n_tiles, int x_dtype, int y_dtype,
                            int has_mask, void* y_base, void* stream) {
  using namespace sbq;
  if (!valid_dtype(x_dtype) || !valid_dtype(y_dtype)) return SBQ_ERR_DTYPE;
  if (y_dtype != SBQ_F32 && y_dtype != x_dtype) return SBQ_ERR_DTYPE;
  if (n_items < 0) return SBQ_ERR_ARG;
  if (n_items == 0 || n_tiles == 0) return SBQ_ERR_EMPTY;
  if (!device_table) return SBQ_ERR_NULL;
  if (!aligned16(device_table) || n_tiles >= (1u << 31)) return SBQ_ERR_ARG;
  if (!aligned16(y_base)) return SBQ_ERR_ALIGN;
  char* yb = static_cast<char*>(y_base);
  const char* base = static_cast<const char*>(device_table);
  const GroupItemDev* items = reinterpret_cast<const GroupItemDev*>(base + kGroupHeaderBytes);
  const uint32_t* tile_item = reinterpret_cast<const uint32_t*>(base + kGroupHeaderBytes + static_cast<size_t>(n_items) * sizeof(GroupItemDev));
  // one tile per workgroup while everything is resident at once, then two (second one's loads
  // in flight while the first is finished) -- same policy as the single-tensor kernels
  uint32_t grid = n_tiles <= 2048 ? n_tiles : (n_tiles + 1) / 2;
  if (n_tiles > 2048 && grid < 2048) grid = 2048;  // never fewer workgroups than fit at once
  if (grid > 8192) grid = 8192;
  hipStream_t st = as_stream(stream);
#define SBQ_G(TI, TO)                                                                              \
  do {                                                                                             \
    if (has_mask) qdq_group_kernel<TI, TO, true><<<grid, kBlock, 0, st>>>(items, tile_item, n_tiles, yb);  \
    else qdq_group_kernel<TI, TO, false><<<grid, kBlock, 0, st>>>(items, tile_item, n_tiles, yb);     \
  } while (0)
  if (x_dtype == SBQ_F32) SBQ_G(F32, F32);
  else if (x_dtype == SBQ_F16) { if (y_dtype == SBQ_F32) SBQ_G(F16, F32); else SBQ_G(F16, F16); }
  else { if (y_dtype == SBQ_F32) SBQ_G(BF16, F32); else SBQ_G(BF16, BF16); }
#undef SBQ_G
  return check_launch();
}

int sbq_group_bwd_table_build(const sbq_group_bwd_item* items, int n_items, void* host_table,
                              size_t host_table_bytes, uint32_t* n_wgs_out, uint32_t* n_rows_out,
                              size_t* bytes_needed_out, size_t* workspace_bytes_out) {
  using namespace sbq;
  if (n_items < 0) return SBQ_ERR_ARG;
  if (n_items == 0) return SBQ_ERR_EMPTY;
  if (!items) return SBQ_ERR_NULL;
  const bool want_mask = items[0].mask != nullptr;
  uint64_t wgs = 0, rows = 0, segs = 0;
  for (int i = 0; i < n_items; ++i) {
    const sbq_group_bwd_item& it = items[i];
    if (!it.x || !it.scale || !it.zero_point) return SBQ_ERR_NULL;
    if (want_mask != (it.mask != nullptr)) return SBQ_ERR_ARG;
    if (it.C < 0 || it.inner < 0) return SBQ_ERR_ARG;
    if (it.C == 0 || it.inner == 0) return SBQ_ERR_EMPTY;
    if (it.qmin > it.qmax || it.C > 0x7fffffff || it.inner % kPack != 0) return SBQ_ERR_ARG;
    const uint64_t ppr = static_cast<uint64_t>(it.inner / kPack);
    const uint64_t spr = (ppr + kSegPacks - 1) / kSegPacks;
    if (static_cast<uint64_t>(it.C) * ppr >= (1ull << 24) || static_cast<uint64_t>(it.C) * spr >= (1ull << 24)) return SBQ_ERR_ARG;
    if (!aligned16(it.x) || (it.gx_offset & 15u) || (reinterpret_cast<uintptr_t>(it.mask) & 7u) ||
        (reinterpret_cast<uintptr_t>(it.scale) & 3u) || (reinterpret_cast<uintptr_t>(it.zero_point) & 3u))
      return SBQ_ERR_ALIGN;
    segs += static_cast<uint64_t>(it.C) * spr;
    wgs += (static_cast<uint64_t>(it.C) * spr + kWavesPerBlock - 1) / kWavesPerBlock;
    rows += static_cast<uint64_t>(it.C);
  }
  if (wgs >= (1ull << 31) || rows >= (1ull << 31) || segs >= (1ull << 31)) return SBQ_ERR_ARG;
  const size_t need = kGroupHeaderBytes + static_cast<size_t>(n_items) * sizeof(GroupBwdItemDev) +
                      static_cast<size_t>(wgs) * sizeof(uint32_t);
  if (n_wgs_out) *n_wgs_out = static_cast<uint32_t>(wgs);
  if (n_rows_out) *n_rows_out = static_cast<uint32_t>(rows);
  if (bytes_needed_out) *bytes_needed_out = need;
  if (workspace_bytes_out) *workspace_bytes_out = static_cast<size_t>(segs) * sizeof(double) + 16;
  if (!host_table) return SBQ_OK;
  if (host_table_bytes < need) return SBQ_ERR_WORKSPACE;
  char* base = static_cast<char*>(host_table);
  uint32_t* header = reinterpret_cast<uint32_t*>(base);
  for (size_t i = 0; i < kGroupHeaderBytes / 4; ++i) header[i] = 0;
  header[0] = static_cast<uint32_t>(n_items);
  header[1] = static_cast<uint32_t>(wgs);
  header[2] = static_cast<uint32_t>(rows);
  GroupBwdItemDev* dev = reinterpret_cast<GroupBwdItemDev*>(base + kGroupHeaderBytes);
  uint32_t* wg_item = reinterpret_cast<uint32_t*>(base + kGroupHeaderBytes + static_cast<size_t>(n_items) * sizeof(GroupBwdItemDev));
  uint32_t w = 0, r = 0, sg = 0;
  for (int i = 0; i < n_items; ++i) {
    const sbq_group_bwd_item& it = items[i];
    GroupBwdItemDev d{};
    d.x = it.x;
    d.gx_off = it.gx_offset;
    d.scale = it.scale;
    d.zp = it.zero_point;
    d.mask = it.mask;
    d.gs_off = it.want_gs ? it.gs_offset : ~0ull;
    d.C = static_cast<uint32_t>(it.C);
    d.packs_per_row = static_cast<uint32_t>(it.inner / kPack);
    d.segs_per_row = (d.packs_per_row + kSegPacks - 1) / kSegPacks;
    d.wg_begin = w;
    d.seg_begin = sg;
    d.row_begin = r;
    d.qlo = static_cast<float>(it.qmin);
    d.qhi = static_cast<float>(it.qmax);
    d.ratio = it.gs_ratio;
    d.flags = it.flags;
    dev[i] = d;
    const uint32_t nseg = d.C * d.segs_per_row;
    const uint32_t nwg = (nseg + kWavesPerBlock - 1) / kWavesPerBlock;
    for (uint32_t k = 0; k < nwg; ++k) wg_item[w + k] = static_cast<uint32_t>(i);
    w += nwg;
    sg += nseg;
    r += d.C;
  }
  return SBQ_OK;
}

int sbq_quant_group_backward(const void* device_table, const void* host_table, int n_items, int x_dtype,
                             int gx_dtype, int has_mask, const void* const* gy, void* gx_base, float* gs_base,
                             void* workspace, size_t workspace_bytes, void* stream) {
  using namespace sbq;
  if (!valid_dtype(x_dtype) || !valid_dtype(gx_dtype)) return SBQ_ERR_DTYPE;
  if (gx_dtype != SBQ_F32 && gx_dtype != x_dtype) return SBQ_ERR_DTYPE;
  if (n_items < 0) return SBQ_ERR_ARG;
  if (n_items == 0) return SBQ_ERR_EMPTY;
  if (!device_table || !host_table || !gy || !gx_base || !workspace) return SBQ_ERR_NULL;
  if (!aligned16(device_table) || !aligned16(gx_base) || !aligned16(workspace)) return SBQ_ERR_ALIGN;
  // the host copy of the table tells where each item's workgroups start (the device one is not readable here)
  const char* hb = static_cast<const char*>(host_table);
  const uint32_t* header = reinterpret_cast<const uint32_t*>(hb);
  if (header[0] != static_cast<uint32_t>(n_items)) return SBQ_ERR_ARG;
  const uint32_t n_wgs = header[1], n_rows = header[2];
  const GroupBwdItemDev* hitems = reinterpret_cast<const GroupBwdItemDev*>(hb + kGroupHeaderBytes);
  uint64_t segs = 0;
  bool any_gs = false;
  for (int i = 0; i < n_items; ++i) {
    segs += static_cast<uint64_t>(hitems[i].C) * hitems[i].segs_per_row;
    any_gs |= hitems[i].gs_off != ~0ull;
    if (!gy[i]) return SBQ_ERR_NULL;
    if (!aligned16(gy[i])) return SBQ_ERR_ALIGN;
  }
  if (any_gs && !gs_base) return SBQ_ERR_NULL;
  if (workspace_bytes < segs * sizeof(double)) return SBQ_ERR_WORKSPACE;
  const char* base = static_cast<const char*>(device_table);
  const GroupBwdItemDev* items = reinterpret_cast<const GroupBwdItemDev*>(base + kGroupHeaderBytes);
  const uint32_t* wg_item = reinterpret_cast<const uint32_t*>(base + kGroupHeaderBytes + static_cast<size_t>(n_items) * sizeof(GroupBwdItemDev));
  double* part = static_cast<double*>(workspace);
  char* gxb = static_cast<char*>(gx_base);
  hipStream_t st = as_stream(stream);
  // the gy pointers of a step travel as kernel arguments (they change every step; the table does
  // not): SBQ_GROUP_BWD_CHUNK items per launch
  for (int i0 = 0; i0 < n_items; i0 += kMaxGroupGy) {
    const int i1 = i0 + kMaxGroupGy < n_items ? i0 + kMaxGroupGy : n_items;
    GyPtrs gp{};
    for (int i = i0; i < i1; ++i) gp.p[i - i0] = gy[i];
    const uint32_t wg0 = hitems[i0].wg_begin;
    const uint32_t wg1 = i1 < n_items ? hitems[i1].wg_begin : n_wgs;
    const uint32_t grid = wg1 - wg0;
#define SBQ_GB(T, TG)                                                                                          \
  do {                                                                                                         \
    if (has_mask) group_bwd_kernel<T, TG, true><<<grid, kBlock, 0, st>>>(items, wg_item, gp, static_cast<uint32_t>(i0), wg0, gxb, part); \
    else group_bwd_kernel<T, TG, false><<<grid, kBlock, 0, st>>>(items, wg_item, gp, static_cast<uint32_t>(i0), wg0, gxb, part); \
  } while (0)
    if (x_dtype == SBQ_F32) SBQ_GB(F32, F32);
    else if (x_dtype == SBQ_F16) { if (gx_dtype == SBQ_F32) SBQ_GB(F16, F32); else SBQ_GB(F16, F16); }
    else { if (gx_dtype == SBQ_F32) SBQ_GB(BF16, F32); else SBQ_GB(BF16, BF16); }
#undef SBQ_GB
    const int rc = check_launch();
    if (rc != SBQ_OK) return rc;
  }
  if (any_gs) {
    group_bwd_fold_kernel<<<(n_rows + kBlock - 1) / kBlock, kBlock, 0, st>>>(items, static_cast<uint32_t>(n_items), n_rows, part, gs_base);
    return check_launch();
  }
  return SBQ_OK;
}

}  // extern "C"
