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
    for (int j = 0; j < kPack; ++j)
      dq[j] = dequant_level(__builtin_amdgcn_fmed3f(__builtin_rintf(fast_div(v[j], s, yr)) + z, qlo, qhi), s, z);
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

}  // extern "C"
