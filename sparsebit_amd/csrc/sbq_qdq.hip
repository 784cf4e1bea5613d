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
//     parallelism), streamed with non-temporal hints: the data is touched once;
//   * anything not 16-byte friendly (inner % 8 != 0, odd pointers, the exotic
//     rounding modes) goes through a scalar kernel with identical arithmetic.
#include "sbq_common.hpp"

namespace sbq {
namespace {

enum { MASK_NONE = 0, MASK_BYTES = 1, MASK_THRESH = 2 };

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

// development-only arithmetic variants (knob 2), used to locate the bottleneck:
// MATH_EXACT is the product; the others are NOT parity-correct.
enum { MATH_EXACT = 0, MATH_RCP = 1, MATH_COPY = 2 };

// Pointers are separate __restrict__ kernel parameters (not struct members) so
// that the compiler may keep the wave-uniform scale / zero_point loads on the
// scalar unit.  Loads are never predicated: out-of-range lanes re-read the last
// valid pack (clamped index) and only the stores are masked, which keeps all U
// loads of a lane in flight together.
//
// ROWS: the tensor is cut into slabs of kBlock packs (2048 elements) that never
// straddle a row; a workgroup takes U consecutive slabs per iteration, so each
// of its U loads has a block-uniform channel (scale/zp via s_load) and a wave
// reads 1 KiB of contiguous HBM per load instruction.
template <typename Tin, typename Tout, int QT, int MASK, bool FLAT, bool NT, int U, int MATH>
__global__ __launch_bounds__(kBlock) void qdq_pack_kernel(
    const void* __restrict__ x, void* __restrict__ y, void* __restrict__ q,
    const uint8_t* __restrict__ mask, const float* __restrict__ thresh,
    const float* __restrict__ scale, const float* __restrict__ zero_point, const QdqGeom g) {
  float thr = 0.0f;
  if constexpr (MASK == MASK_THRESH) thr = *thresh;

  for (uint32_t tile = blockIdx.x; tile < g.n_tiles; tile += gridDim.x) {
    int64_t elem[U];
    bool ok[U];
    float s[U], z[U];
    if constexpr (!FLAT) {
      uint32_t sl = tile * U;
      uint32_t row = sl / g.slabs_per_row;  // scalar unit; once per tile
      uint32_t col = sl - row * g.slabs_per_row;
      uint32_t c = row % g.C;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool slab_ok = sl < g.n_slabs;
        const uint32_t pk = col * kBlock + threadIdx.x;
        ok[u] = slab_ok && pk < g.packs_per_row;
        const uint32_t pkc = pk < g.packs_per_row ? pk : g.packs_per_row - 1;
        elem[u] = static_cast<int64_t>(row) * g.inner + static_cast<int64_t>(pkc) * kPack;
        s[u] = scale[c];
        z[u] = __builtin_rintf(zero_point[c]);
        // advance to the next slab without dividing; past the end stay on the last one
        if (sl + 1 < g.n_slabs) {
          ++sl;
          if (++col == g.slabs_per_row) {
            col = 0;
            ++row;
            if (++c == g.C) c = 0;
          }
        } else {
          sl = g.n_slabs;
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t pk = (tile * U + u) * kBlock + threadIdx.x;
        ok[u] = pk < g.total_packs;
        const uint32_t pkc = ok[u] ? pk : g.total_packs - 1;
        elem[u] = static_cast<int64_t>(pkc) * kPack;
        uint32_t c = 0;
        if (g.C != 1) c = (pkc / g.packs_per_row) % g.C;  // per tensor: no division
        s[u] = scale[c];
        z[u] = __builtin_rintf(zero_point[c]);
      }
    }

    float v[U][kPack];
    u32x2 mk[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      load_pack<Tin, NT>(x, elem[u], v[u]);
      if constexpr (MASK == MASK_BYTES) mk[u] = ld8<NT>(mask + elem[u]);
    }

#pragma unroll
    for (int u = 0; u < U; ++u) {
      float lv[kPack], dq[kPack];
      float rs = 0.0f;
      if constexpr (MATH == MATH_RCP) rs = 1.0f / s[u];
#pragma unroll
      for (int j = 0; j < kPack; ++j) {
        float xv = v[u][j];
        if constexpr (MASK == MASK_BYTES) {
          const uint32_t byte = (mk[u][j >> 2] >> (8 * (j & 3))) & 0xffu;
          xv = byte ? xv : 0.0f;
        } else if constexpr (MASK == MASK_THRESH) {
          xv = (__builtin_fabsf(xv) > thr) ? xv : 0.0f;
        }
        if constexpr (MATH == MATH_EXACT) {
          lv[j] = quant_level<SBQ_ROUND_HALF_EVEN>(xv, s[u], z[u], g.qlo, g.qhi);
          dq[j] = dequant_level(lv[j], s[u], z[u]);
        } else if constexpr (MATH == MATH_RCP) {
          float t = __builtin_rintf(xv * rs) + z[u];
          lv[j] = __builtin_fminf(__builtin_fmaxf(t, g.qlo), g.qhi);
          dq[j] = dequant_level(lv[j], s[u], z[u]);
        } else {
          lv[j] = xv;
          dq[j] = xv;
        }
      }
      if (ok[u]) {
        store_pack<Tout, NT>(y, elem[u], dq);
        if constexpr (QT != SBQ_Q_NONE) store_q_pack<QT>(q, elem[u], lv);
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
    const float s = a.scale[c];
    const float z = __builtin_rintf(a.zp[c]);
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
    store_any(a.y, a.y_dtype, i, dequant_level(lv, s, z));
    if (a.q_type == SBQ_Q_I8) static_cast<int8_t*>(a.q)[i] = static_cast<int8_t>(static_cast<int>(lv));
    else if (a.q_type == SBQ_Q_I32) static_cast<int32_t*>(a.q)[i] = static_cast<int>(lv);
  }
}

// ---- host dispatch -------------------------------------------------------------------
constexpr uint32_t kDefaultGridCap = 256 * 8;  // 8 resident workgroups of 256 on each of 256 CUs

struct QdqCall {
  QdqPtrs p;
  QdqGeom g;
  uint32_t rows;
};

template <typename Tin, typename Tout, int QT, int MASK, bool FLAT, bool NT, int U, int MATH = MATH_EXACT>
void launch_pack(const QdqCall& c, hipStream_t st) {
  const uint32_t cap = knob(1) > 0 ? static_cast<uint32_t>(knob(1)) : kDefaultGridCap;
  const uint32_t grid = c.g.n_tiles < cap ? c.g.n_tiles : cap;
  qdq_pack_kernel<Tin, Tout, QT, MASK, FLAT, NT, U, MATH><<<grid, kBlock, 0, st>>>(
      c.p.x, c.p.y, c.p.q, c.p.mask, c.p.thresh, c.p.scale, c.p.zp, c.g);
}

// variant id (knob 0):  bit0-1: log2(U) (0..3) ; bit2: NT off ; -1 auto
template <typename Tin, typename Tout, int QT, int MASK, bool FLAT>
void launch_variant(QdqCall c, int variant, hipStream_t st) {
  int lu;
  bool nt = true;
  if (variant >= 0) {
    lu = variant & 3;
    nt = !(variant & 4);
  } else {
    // auto: enough packs in flight per lane to cover HBM latency, but keep >= 2
    // workgroups per CU busy
    const uint32_t units = FLAT ? (c.g.total_packs + kBlock - 1) / kBlock : c.g.n_slabs;
    lu = units >= 8u * 512u ? 2 : (units >= 4u * 512u ? 1 : 0);
  }
  const uint32_t U = 1u << lu;
  if (FLAT) c.g.n_tiles = (c.g.total_packs + kBlock * U - 1) / (kBlock * U);
  else c.g.n_tiles = (c.g.n_slabs + U - 1) / U;
#define SBQ_LAUNCH(NTV, UV) launch_pack<Tin, Tout, QT, MASK, FLAT, NTV, UV>(c, st)
  if constexpr (QT == SBQ_Q_NONE && MASK == MASK_NONE && Tin::id == SBQ_BF16 && Tout::id == SBQ_BF16 && !FLAT) {
    // development variants for the headline shape only (A/B measurements)
    const int math = knob(2);
    if (math == MATH_RCP) {
      if (U == 4) launch_pack<Tin, Tout, QT, MASK, FLAT, true, 4, MATH_RCP>(c, st);
      else launch_pack<Tin, Tout, QT, MASK, FLAT, true, 8, MATH_RCP>(c, st);
      return;
    }
    if (math == MATH_COPY) {
      if (U == 4) launch_pack<Tin, Tout, QT, MASK, FLAT, true, 4, MATH_COPY>(c, st);
      else launch_pack<Tin, Tout, QT, MASK, FLAT, true, 8, MATH_COPY>(c, st);
      return;
    }
    if (!nt) {
      if (U == 1) SBQ_LAUNCH(false, 1);
      else if (U == 2) SBQ_LAUNCH(false, 2);
      else if (U == 4) SBQ_LAUNCH(false, 4);
      else SBQ_LAUNCH(false, 8);
      return;
    }
  }
  if (U == 1) SBQ_LAUNCH(true, 1);
  else if (U == 2) SBQ_LAUNCH(true, 2);
  else if (U == 4) SBQ_LAUNCH(true, 4);
  else SBQ_LAUNCH(true, 8);
#undef SBQ_LAUNCH
}

template <typename Tin, typename Tout, int QT, int MASK>
void launch_geom(const QdqCall& c, bool flat, hipStream_t st) {
  const int variant = knob(0);
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
  else if (q_type == SBQ_Q_I32) launch_mask<Tin, Tout, SBQ_Q_I32>(c, flat, st);
  else launch_mask<Tin, Tout, SBQ_Q_NONE>(c, flat, st);
}

void launch_scalar(ScalarArgs a, hipStream_t st) {
  const int64_t n = a.end - a.begin;
  if (n <= 0) return;
  int64_t blocks = ceil_div(n, kBlock);
  if (blocks > static_cast<int64_t>(kDefaultGridCap)) blocks = kDefaultGridCap;
  qdq_scalar_kernel<<<static_cast<uint32_t>(blocks), kBlock, 0, st>>>(a);
}

int qdq_forward(const void* x, int x_dtype, void* y, int y_dtype, void* q, int q_type,
                const uint8_t* mask, const float* thresh, const float* scale, const float* zp,
                int64_t outer, int64_t C, int64_t inner, int qmin, int qmax, int rounding,
                void* stream) {
  if (!valid_dtype(x_dtype) || !valid_dtype(y_dtype)) return SBQ_ERR_DTYPE;
  if (y_dtype != SBQ_F32 && y_dtype != x_dtype) return SBQ_ERR_DTYPE;
  if (q_type != SBQ_Q_NONE && q_type != SBQ_Q_I8 && q_type != SBQ_Q_I32) return SBQ_ERR_DTYPE;
  if (outer < 0 || C < 0 || inner < 0) return SBQ_ERR_ARG;
  if (outer == 0 || C == 0 || inner == 0) return SBQ_ERR_EMPTY;
  if (!x || !y || !scale || !zp) return SBQ_ERR_NULL;
  if (q_type != SBQ_Q_NONE && !q) return SBQ_ERR_NULL;
  if (qmin > qmax) return SBQ_ERR_ARG;
  if (q_type == SBQ_Q_I8 && static_cast<int64_t>(qmax) - qmin > 255) return SBQ_ERR_ARG;
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
                x_dtype, y_dtype, q_type, rounding, static_cast<float>(qmin), static_cast<float>(qmax)};

  const bool ptr_ok = aligned16(x) && aligned16(y) && (q_type == SBQ_Q_NONE || aligned16(q)) &&
                      (!mask || (reinterpret_cast<uintptr_t>(mask) & 7u) == 0);
  // The pack kernels need rows made of whole 8-element packs.  A per-tensor
  // call (C == 1) is one long row, so only its last numel % 8 elements are ragged.
  const int64_t body = (C == 1) ? (numel / kPack) * kPack : (inner % kPack == 0 ? numel : 0);
  const uint64_t total_packs = static_cast<uint64_t>(body / kPack);
  // 32-bit pack / tile arithmetic in the kernels: < 2^31 packs (16 Gi elements)
  if (rounding != SBQ_ROUND_HALF_EVEN || !ptr_ok || body == 0 || total_packs >= (1ull << 31) ||
      rows >= (1ll << 31)) {
    launch_scalar(sa, st);
    return check_launch();
  }

  QdqCall c{};
  c.p = QdqPtrs{x, y, q, mask, thresh, scale, zp};
  c.g.C = static_cast<uint32_t>(C);
  c.g.qlo = static_cast<float>(qmin);
  c.g.qhi = static_cast<float>(qmax);
  c.g.total_packs = static_cast<uint32_t>(total_packs);
  bool flat;
  if (C == 1) {  // one row of `body` elements
    c.g.inner = body;
    c.g.packs_per_row = c.g.total_packs;
    c.rows = 1;
    flat = true;
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

}  // namespace
}  // namespace sbq

extern "C" {

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
