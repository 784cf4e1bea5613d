// sbq_dequant.hip -- DequantizeLinear for the integer tensors the QDQ kernels emit.
//
//   y = (q - round(zp)) * s      in fp32, then the RNE cast to y's dtype
//
// This is the second half of the reference's fake-quant formula (quant_tensor.py:184,
// fake_quant_tensor.cu:70-75 `(q - zp) * scale`) applied to STORED levels: what a runtime
// does with the QuantizeLinear / DequantizeLinear pair the reference exports
// (quant_model.py:222-324), and bit-identical to the dequantized output of
// sbq_quant_*_forward for the same q, scale and zero point.  HBM-bound: 1 (int8) or 0.5
// (packed int4) bytes in, 2 or 4 bytes out per element.
#include "sbq_common.hpp"

namespace sbq {
namespace {

constexpr uint32_t kDqChunk = kBlock * kPack * 4;  // 8192 elements per workgroup

template <int QT>
__device__ __forceinline__ void load_levels(const void* q, int64_t i, bool is_signed, float (&lv)[kPack]) {
  if constexpr (QT == SBQ_Q_I8) {
    const u32x2 w = ld8<true>(static_cast<const char*>(q) + i);
#pragma unroll
    for (int j = 0; j < kPack; ++j) {
      const uint32_t b = (w[j >> 2] >> (8 * (j & 3))) & 0xffu;
      lv[j] = is_signed ? static_cast<float>(static_cast<int8_t>(b)) : static_cast<float>(b);
    }
  } else if constexpr (QT == SBQ_Q_I4) {
    const uint32_t w = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(static_cast<const char*>(q) + (i >> 1)));
#pragma unroll
    for (int j = 0; j < kPack; ++j) {
      const int32_t n = static_cast<int32_t>((w >> (4 * j)) & 0xfu);
      lv[j] = static_cast<float>(is_signed ? ((n ^ 8) - 8) : n);  // sign-extend the nibble
    }
  } else {
    const char* p = static_cast<const char*>(q) + i * 4;
    const u32x4 a = ld16<true>(p), b = ld16<true>(p + 16);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      lv[j] = static_cast<float>(static_cast<int32_t>(a[j]));
      lv[4 + j] = static_cast<float>(static_cast<int32_t>(b[j]));
    }
  }
}

// four levels starting at element i (a lane's 4-element run of the split mapping)
template <int QT>
__device__ __forceinline__ void load_levels_half(const void* q, int64_t i, bool is_signed, float* lv) {
  if constexpr (QT == SBQ_Q_I8) {
    const uint32_t w = ld4<true>(static_cast<const char*>(q) + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t b = (w >> (8 * j)) & 0xffu;
      lv[j] = is_signed ? static_cast<float>(static_cast<int8_t>(b)) : static_cast<float>(b);
    }
  } else if constexpr (QT == SBQ_Q_I4) {
    const uint32_t w = __builtin_nontemporal_load(reinterpret_cast<const uint16_t*>(static_cast<const char*>(q) + (i >> 1)));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int32_t n = static_cast<int32_t>((w >> (4 * j)) & 0xfu);
      lv[j] = static_cast<float>(is_signed ? ((n ^ 8) - 8) : n);
    }
  } else {
    const u32x4 a = ld16<true>(static_cast<const char*>(q) + i * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) lv[j] = static_cast<float>(static_cast<int32_t>(a[j]));
  }
}

__device__ __forceinline__ float load_level1(const void* q, int q_type, int64_t i, bool is_signed) {
  if (q_type == SBQ_Q_I8) {
    const uint8_t b = static_cast<const uint8_t*>(q)[i];
    return is_signed ? static_cast<float>(static_cast<int8_t>(b)) : static_cast<float>(b);
  }
  if (q_type == SBQ_Q_I4) {
    const int32_t n = (static_cast<const uint8_t*>(q)[i >> 1] >> (4 * (i & 1))) & 0xf;
    return static_cast<float>(is_signed ? ((n ^ 8) - 8) : n);
  }
  return static_cast<float>(static_cast<const int32_t*>(q)[i]);
}

template <typename Tout, int QT, bool VEC>
__global__ __launch_bounds__(kBlock) void dequant_kernel(const void* __restrict__ q, void* __restrict__ y,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ zero_point, const ChunkGeom g,
                                                         int is_signed) {
  const ChunkPos cp = chunk_pos(g, blockIdx.x);
  const float s = scale[cp.c];
  const float z = __builtin_rintf(zero_point[cp.c]);
  if constexpr (VEC && Tout::id == SBQ_F32) {
    // fp32 output: two 4-element runs per lane half a 2048-element block apart -- every 16-byte store is
    // part of a contiguous 1 KiB wave access (sbq_common.hpp: load_raw2)
    constexpr int64_t kBlk = static_cast<int64_t>(kBlock) * kPack;
    for (int64_t b0 = cp.begin; b0 < cp.end; b0 += kBlk) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int64_t e = b0 + h * (kBlk / 2) + 4 * threadIdx.x;
        if (e < cp.end) {  // rows are whole 8-element packs: a started run is a whole run
          float lv[4], o[4];
          load_levels_half<QT>(q, cp.row_base + e, is_signed != 0, lv);
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = dequant_level(lv[j], s, z);
          store_half<Tout, true>(y, cp.row_base + e, o);
        }
      }
    }
  } else if constexpr (VEC) {
    for (int64_t e = cp.begin + static_cast<int64_t>(threadIdx.x) * kPack; e < cp.end;
         e += static_cast<int64_t>(kBlock) * kPack) {
      float lv[kPack], o[kPack];
      load_levels<QT>(q, cp.row_base + e, is_signed != 0, lv);
#pragma unroll
      for (int j = 0; j < kPack; ++j) o[j] = dequant_level(lv[j], s, z);
      store_pack<Tout, true>(y, cp.row_base + e, o);
    }
  } else {
    for (int64_t e = cp.begin + threadIdx.x; e < cp.end; e += kBlock)
      Elem<Tout>::store1(y, cp.row_base + e, dequant_level(load_level1(q, QT, cp.row_base + e, is_signed != 0), s, z));
  }
}

}  // namespace
}  // namespace sbq

extern "C" {

int sbq_dequantize_linear(const void* q, int q_type, int q_signed, void* y, int y_dtype, const float* scale,
                          const float* zero_point, int64_t outer, int64_t C, int64_t inner, void* stream) {
  using namespace sbq;
  if (!valid_dtype(y_dtype)) return SBQ_ERR_DTYPE;
  if (q_type != SBQ_Q_I8 && q_type != SBQ_Q_I32 && q_type != SBQ_Q_I4) return SBQ_ERR_DTYPE;
  if (outer < 0 || C < 0 || inner < 0) return SBQ_ERR_ARG;
  if (outer == 0 || C == 0 || inner == 0) return SBQ_ERR_EMPTY;
  if (!q || !y || !scale || !zero_point) return SBQ_ERR_NULL;
  if (!geom_ok(outer, C, inner, kDqChunk)) return SBQ_ERR_ARG;
  if (reinterpret_cast<uintptr_t>(y) % dtype_size(y_dtype)) return SBQ_ERR_ALIGN;
  if (q_type == SBQ_Q_I32 && (reinterpret_cast<uintptr_t>(q) & 3u)) return SBQ_ERR_ALIGN;
  const bool vec = inner % kPack == 0 && aligned16(y) && aligned16(q);
  // two elements share a byte: rows must start on a byte boundary for the element-wise path too
  if (q_type == SBQ_Q_I4 && !vec && (inner & 1)) return SBQ_ERR_ARG;
  const ChunkGeom g = make_geom(outer, C, inner, kDqChunk);
  const uint32_t grid = g.chunks_per_chan * g.C;
  hipStream_t st = as_stream(stream);
  const int rc = dispatch_dtype(y_dtype, [&](auto tag) {
    using T = decltype(tag);
#define SBQ_DQ(QTV)                                                                                     \
  do {                                                                                                  \
    if (vec) dequant_kernel<T, QTV, true><<<grid, kBlock, 0, st>>>(q, y, scale, zero_point, g, q_signed);   \
    else dequant_kernel<T, QTV, false><<<grid, kBlock, 0, st>>>(q, y, scale, zero_point, g, q_signed);  \
  } while (0)
    if (q_type == SBQ_Q_I8) SBQ_DQ(SBQ_Q_I8);
    else if (q_type == SBQ_Q_I4) SBQ_DQ(SBQ_Q_I4);
    else SBQ_DQ(SBQ_Q_I32);
#undef SBQ_DQ
  });
  if (rc != SBQ_OK) return rc;
  return check_launch();
}

}  // extern "C"
