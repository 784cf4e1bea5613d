// sbq_qdq_math.hpp -- the per-pack arithmetic of the forward QDQ, shared by the pipelined kernels
// (sbq_qdq.hip) and the resident schedule (sbq_qdq_resident.hip).  Arithmetic follows the reference CPU
// path sparsebit/quantization/quantizers/quant_tensor.py:182-184.
#pragma once
#include "sbq_common.hpp"

namespace sbq {

enum { MASK_NONE = 0, MASK_BYTES = 1, MASK_THRESH = 2 };

// Arithmetic of the pack kernels: both are exact.  MATH_IEEE (knob 2 == 3, headline shape only)
// is kept for A/B measurements of what the reciprocal + fma refinement buys; the two
// non-parity probes used to locate the bottleneck (reciprocal multiply, plain copy; numbers
// in DESIGN.md 7) are gone from the product library.
enum { MATH_FAST = 0, MATH_IEEE = 3 };

// Block-uniform read of a quantization parameter through the scalar unit.  scale/zero_point
// are never written by these kernels, so they may be read through the constant address
// space; without this a pointer that itself came from memory (the batched kernel's table)
// is not provably alias-free and the load degrades to a per-lane VMEM broadcast, doubling
// the number of vector-memory instructions per tile.
__device__ __forceinline__ float uniform_load(const float* p, uint32_t i) {
  typedef const float __attribute__((address_space(4))) * cptr;
  return reinterpret_cast<cptr>(reinterpret_cast<uintptr_t>(p))[i];
}

// The arithmetic of one pack: v[] (already unpacked) -> masked -> levels lv[] and dequantized dq[].
template <int MASK, bool FLAT, int MATH>
__device__ __forceinline__ void quantize_pack(float (&v)[kPack], const u32x2 mk, float thr, float s, float z,
                                              float qlo, float qhi, float (&lv)[kPack], float (&dq)[kPack]) {
#pragma unroll
  for (int j = 0; j < kPack; ++j) {
    if constexpr (MASK == MASK_BYTES) {
      const uint32_t byte = (mk[j >> 2] >> (8 * (j & 3))) & 0xffu;
      v[j] = byte ? v[j] : 0.0f;
    } else if constexpr (MASK == MASK_THRESH) {
      v[j] = (__builtin_fabsf(v[j]) > thr) ? v[j] : 0.0f;
    }
  }
  // ROWS: `s` is block-uniform, so the choice below is a scalar branch and y = 1/s is one
  // division per slab instead of one per element.
  // NaN, +-inf and |x| >= s * 2^40 leave the range in which the fma refinement is exact.
  // One compare per element feeds a wave-wide vote; a wave that holds any such value
  // (never, on real weights) redoes the pack with IEEE division instead of every element
  // paying a clamp and a NaN restore.
  bool fast = (MATH == MATH_FAST) && !FLAT && fast_div_ok(s);
  if (fast) {
    const float bound = s * 0x1p40f;
    bool odd = false;
#pragma unroll
    for (int j = 0; j < kPack; ++j) odd |= !(__builtin_fabsf(v[j]) < bound);
    fast = __builtin_amdgcn_ballot_w64(odd) == 0;
  }
  if (fast) {
    const float yr = 1.0f / s;
    if (z == 0.0f) {
      // zero point 0 (every symmetric scheme; block-uniform): no zero-point add / subtract; the product
      // goes through fma(lv, s, +0) so that a level of -0 still dequantizes to +0 like (lv - 0) * s does
#pragma unroll
      for (int j = 0; j < kPack; j += 2) {
        const f32x2 t = fast_div2(f32x2{v[j], v[j + 1]}, s, yr);
        lv[j] = __builtin_amdgcn_fmed3f(__builtin_rintf(t[0]), qlo, qhi);
        lv[j + 1] = __builtin_amdgcn_fmed3f(__builtin_rintf(t[1]), qlo, qhi);
        const f32x2 d = __builtin_elementwise_fma(f32x2{lv[j], lv[j + 1]}, f32x2{s, s}, f32x2{0.0f, 0.0f});
        dq[j] = d[0];
        dq[j + 1] = d[1];
      }
    } else {
#pragma unroll
      for (int j = 0; j < kPack; j += 2) {
        const f32x2 t = fast_div2(f32x2{v[j], v[j + 1]}, s, yr);
        lv[j] = __builtin_amdgcn_fmed3f(__builtin_rintf(t[0]) + z, qlo, qhi);
        lv[j + 1] = __builtin_amdgcn_fmed3f(__builtin_rintf(t[1]) + z, qlo, qhi);
        dq[j] = dequant_level(lv[j], s, z);
        dq[j + 1] = dequant_level(lv[j + 1], s, z);
      }
    }
  } else {
    // The IEEE form exists ONCE and starts with a statement the optimiser may not execute speculatively:
    // hoisted above the branch (it has been: ten VALU operations per element on every pack, a 40 % slower
    // kernel with identical results) it turns the memory-bound kernel into a VALU-bound one.
    asm volatile("; IEEE division path");  // (no memory clobber: that would cost the hot path its cached loads)
#pragma unroll
    for (int j = 0; j < kPack; ++j) {
      lv[j] = quant_level<SBQ_ROUND_HALF_EVEN>(v[j], s, z, qlo, qhi);
      dq[j] = dequant_level(lv[j], s, z);
    }
  }
}

// ---- resident schedule (sbq_qdq_resident.hip) -------------------------------------------------------
struct ResidentCall {
  const void* x;
  void* y;
  const uint8_t* mask;
  const float* thresh;
  const float* scale;
  const float* zp;
  uint32_t packs_per_row, slabs_per_row, n_slabs, rows, C;
  float qlo, qhi;
  uint32_t lsq;
  int x_dtype, y_dtype;
};
// launches the resident kernel and returns true when the geometry is eligible (and knob 3 allows it)
bool qdq_try_resident(const ResidentCall& c, hipStream_t st);

}  // namespace sbq
