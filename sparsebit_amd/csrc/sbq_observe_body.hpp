// sbq_observe_body.hpp -- device code shared by the per-tensor observer kernels (sbq_observe.hip) and the model-wide
// calibration launches (sbq_calib.hip): the packed 16-bit min / max reduction, the MSE candidate and the body that
// walks the 80 candidates over one 4096-element chunk.  Same code in both => bit-identical results.
#pragma once

#include "sbq_common.hpp"

namespace sbq {
namespace {

constexpr uint32_t kStatsChunk = kWave * kPack * 8;   // 4096 elements: ONE WAVE, 8 packs per lane
constexpr uint32_t kMseChunk = kBlock * kPack * 2;    // 4096 elements, 2 packs per lane (registers)

struct StatPartial {
  float mn, mx;
  double abssum;
};

struct MinF { __device__ __forceinline__ float operator()(float a, float b) const { return __builtin_fminf(a, b); } };
struct MaxF { __device__ __forceinline__ float operator()(float a, float b) const { return __builtin_fmaxf(a, b); } };
struct OrI { __device__ __forceinline__ int operator()(int a, int b) const { return a | b; } };

// ---- min / max only (the min-max observer; abssum_out == NULL): half the vector work or less -------------
// 16-bit inputs never become floats.  Two raw elements per dword go through THREE packed integer operations:
//   A = v_pk_max_u16   B = v_pk_min_u16   C = v_pk_max_i16        (1.5 operations per element, no unpack)
// and the floats come out of (A, B, C) once per wave.  For a sign-magnitude format, as unsigned 16-bit numbers the
// non-negative values sort upwards from +0 to +NaN and the negative ones follow them, -0 first, -NaN last:
//   any negative?      A >= 0x8000        its most negative value (or a -NaN) IS A
//   any non-negative?  B <  0x8000        its largest value (or a +NaN) is C, the signed maximum
//   min = any negative ? A : B            max = any non-negative ? C : B
//   NaN present  <=>  (any non-negative && C > +inf)  ||  (any negative && A > -inf)     -> min = max = NaN (torch)
// fp32 inputs use gfx950's NaN-propagating v_minimum3_f32 / v_maximum3_f32 (IEEE-754-2019 minimum / maximum:
// torch.min / max semantics in one instruction per two elements, no NaN flag).  Both are idempotent, so the lanes
// past the end of a short chunk simply fold a valid pack of the same chunk again: no validity flags.
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
typedef int16_t i16x2 __attribute__((ext_vector_type(2)));
struct Stat16 {
  uint32_t a, b, c;  // packed pairs: max_u16, min_u16, max_i16
};
constexpr Stat16 kStat16Identity{0x00000000u, 0xffffffffu, 0x80008000u};
__device__ __forceinline__ void stat16_fold(Stat16& s, uint32_t w) {
  s.a = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, s.a), __builtin_bit_cast(u16x2, w)));
  s.b = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, s.b), __builtin_bit_cast(u16x2, w)));
  s.c = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(i16x2, s.c), __builtin_bit_cast(i16x2, w)));
}
__device__ __forceinline__ Stat16 stat16_merge(const Stat16& x, const Stat16& y) {
  Stat16 r = x;
  r.a = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, x.a), __builtin_bit_cast(u16x2, y.a)));
  r.b = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, x.b), __builtin_bit_cast(u16x2, y.b)));
  r.c = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(i16x2, x.c), __builtin_bit_cast(i16x2, y.c)));
  return r;
}
// both halves of every lane -> one (A, B, C) for the wave (valid in every lane, in the low half); the cross-lane part
// on DPP (sbq_common.hpp), not through the LDS crossbar
__device__ __forceinline__ Stat16 stat16_wave(Stat16 s) {
  s = stat16_merge(s, Stat16{s.a >> 16, s.b >> 16, static_cast<uint32_t>(static_cast<int32_t>(s.c) >> 16)});
  s.a = dpp_reduce_u32(s.a, 0x00000000u, [](uint32_t x, uint32_t y) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, x), __builtin_bit_cast(u16x2, y)));
  });
  s.b = dpp_reduce_u32(s.b, 0xffffffffu, [](uint32_t x, uint32_t y) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, x), __builtin_bit_cast(u16x2, y)));
  });
  s.c = dpp_reduce_u32(s.c, 0x80008000u, [](uint32_t x, uint32_t y) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(i16x2, x), __builtin_bit_cast(i16x2, y)));
  });
  return s;
}
template <typename T>
__device__ __forceinline__ void stat16_decode(const Stat16& s, float& mn, float& mx) {
  constexpr uint32_t kInf = T::id == SBQ_BF16 ? 0x7f80u : 0x7c00u;
  const uint32_t a = s.a & 0xffffu, b = s.b & 0xffffu, c = s.c & 0xffffu;
  const bool any_neg = a >= 0x8000u, any_pos = b < 0x8000u;
  const bool nan = (any_pos && c > kInf) || (any_neg && (a & 0x7fffu) > kInf);
  mn = Elem<T>::from_bits(static_cast<uint16_t>(any_neg ? a : b));
  mx = Elem<T>::from_bits(static_cast<uint16_t>(any_pos ? c : b));
  if (nan) mn = mx = __builtin_nanf("");
}

// ---- MSE search -------------------------------------------------------------------
// mse.py:46-49: candidate i shrinks (min, max) by the fp32 factor (1 - 0.01 i).
__device__ __forceinline__ void mse_candidate(float mn, float mx, int i, float qrange, bool symmetric,
                                              float& s, float& z) {
  const float f = static_cast<float>(1.0 - static_cast<double>(i) * 0.01);
  qparams_from_minmax(mn * f, mx * f, qrange, symmetric, s, z);
}

struct MseLds {
  float scale[SBQ_MSE_CANDIDATES];
  float zp[SBQ_MSE_CANDIDATES];
  float rcp[SBQ_MSE_CANDIDATES];  // RN(1/scale) when the exact fast division applies, else 0
  float acc[SBQ_MSE_CANDIDATES][kWavesPerBlock];
};

// One workgroup (256 threads), one chunk = elements [begin, end) of the row that starts at row_base (end - begin <=
// 4096; whole packs when VEC): the 80 candidates of (mn, mx) walked over the chunk held in registers.  Returns, in
// thread i < 80, the chunk's sum of squared QDQ errors of candidate i (fp64 sum of the four waves' fp32 sums).
template <typename T, bool VEC>
__device__ __forceinline__ double mse_chunk_body(MseLds& lds, const void* __restrict__ x, int64_t row_base, int64_t begin,
                                                 int64_t end, float mn, float mx, float qrange, float qlo, float qhi,
                                                 bool symmetric) {
  if (threadIdx.x < SBQ_MSE_CANDIDATES) {
    float s, z;
    mse_candidate(mn, mx, threadIdx.x, qrange, symmetric, s, z);
    lds.scale[threadIdx.x] = s;
    lds.zp[threadIdx.x] = z;  // already integral (rint) or 0
    lds.rcp[threadIdx.x] = fast_div_ok(s) ? 1.0f / s : 0.0f;
  }

  // Lanes past the end of the chunk hold x = 0: its QDQ is exactly 0 for every candidate
  // (zp lies inside [qmin, qmax]), so they add exactly 0 to every sum -- no masking needed.
  constexpr int E = 2 * kPack;  // elements per lane
  float v[E];
  if constexpr (VEC) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      int64_t e = begin + (static_cast<int64_t>(u) * kBlock + threadIdx.x) * kPack;
      const bool in = e + kPack <= end;  // chunk and inner are multiples of 8 here
      if (!in) e = begin;
      float t[kPack];
      load_pack<T, true>(x, row_base + e, t);
#pragma unroll
      for (int q = 0; q < kPack; ++q) v[u * kPack + q] = in ? t[q] : 0.0f;
    }
  } else {
#pragma unroll
    for (int q = 0; q < E; ++q) {
      const int64_t e = begin + static_cast<int64_t>(q) * kBlock + threadIdx.x;
      v[q] = e < end ? Elem<T>::load1(x, row_base + e) : 0.0f;
    }
  }
  __syncthreads();

  const int lane = threadIdx.x & (kWave - 1);
  const int wid = threadIdx.x / kWave;
  // Every candidate of the chunk on the fast division with zero point 0 (any symmetric scheme on ordinary data)?
  // Then the loop below has no branch per candidate and takes the candidates two at a time: two independent
  // dependency chains over the same 16 registers, one loop-carried LDS access pattern -- tools/lab/mse_lab.hip,
  // warm clocks: 130.1 us against 136.3 us for the one-at-a-time loop on 4096 x 4096 bf16 (packed fp32 operations:
  // 144 us -- v_pk_mul / v_pk_fma issue at 6.2 cycles).  Per candidate the operations and their order are the same:
  // bit-identical sums.
  bool plain = true;
  {
    const int i0 = lane, i1 = lane + kWave;
    const bool bad0 = lds.rcp[i0] == 0.0f || lds.zp[i0] != 0.0f;
    const bool bad1 = i1 < SBQ_MSE_CANDIDATES && (lds.rcp[i1 < SBQ_MSE_CANDIDATES ? i1 : 0] == 0.0f ||
                                                   lds.zp[i1 < SBQ_MSE_CANDIDATES ? i1 : 0] != 0.0f);
    plain = __builtin_amdgcn_ballot_w64(bad0 || bad1) == 0;  // the same answer in every wave
  }
  if (plain) {
    static_assert(SBQ_MSE_CANDIDATES % 2 == 0, "candidates in pairs");
    for (int i = 0; i < SBQ_MSE_CANDIDATES; i += 2) {
      const float s0 = lds.scale[i], y0 = lds.rcp[i], s1 = lds.scale[i + 1], y1 = lds.rcp[i + 1];
      float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
      for (int q = 0; q < E; ++q) {
        const float l0 = __builtin_amdgcn_fmed3f(__builtin_rintf(v[q] * y0), qlo, qhi);
        const float l1 = __builtin_amdgcn_fmed3f(__builtin_rintf(v[q] * y1), qlo, qhi);
        const float d0 = __builtin_fmaf(-l0, s0, v[q]);
        const float d1 = __builtin_fmaf(-l1, s1, v[q]);
        a0 = __builtin_fmaf(d0, d0, a0);
        a1 = __builtin_fmaf(d1, d1, a1);
      }
      // (wave sums on DPP: the butterfly's twelve ds_bpermute round trips per pair were a fifth of this loop)
      a0 = wave_sum_f32(a0);
      a1 = wave_sum_f32(a1);
      if (lane == 0) {
        lds.acc[i][wid] = a0;
        lds.acc[i + 1][wid] = a1;
      }
    }
  } else
  for (int i = 0; i < SBQ_MSE_CANDIDATES; ++i) {
    const float s = lds.scale[i];
    const float z = lds.zp[i];
    const float y = lds.rcp[i];
    float acc = 0.0f;
    if (y != 0.0f) {  // block-uniform: the candidate's scale is shared by the whole chunk
      // The level is taken from x * RN(1/s) instead of the correctly rounded x / s: the two can
      // only differ within ~1e-7 (relative) of a rounding tie, and AT a tie both neighbouring
      // levels are equally far from x, so the squared error -- the only thing this kernel
      // produces -- is unchanged to ~1e-7 of one element's term.  (The forward QDQ kernels keep
      // the exact quotient: there the level itself is the output.)  NaN / inf inputs poison the loss
      // either way.
      // This loop is the kernel (80 x 16.7 M evaluations, VALU-bound): the residual and its square are
      // contracted into fmas (x - lv*s and acc + d*d, each with ONE rounding -- closer to the exact loss than
      // the reference's separately rounded tensor ops; only the argmin is compared), and a candidate with
      // zero_point 0 (every symmetric scheme) skips the two zero-point operations: 5 ops instead of 9.
      if (z == 0.0f) {  // block-uniform
#pragma unroll
        for (int q = 0; q < E; ++q) {
          const float lv = __builtin_amdgcn_fmed3f(__builtin_rintf(v[q] * y), qlo, qhi);
          const float d = __builtin_fmaf(-lv, s, v[q]);
          acc = __builtin_fmaf(d, d, acc);
        }
      } else {
#pragma unroll
        for (int q = 0; q < E; ++q) {
          const float lv = __builtin_amdgcn_fmed3f(__builtin_rintf(v[q] * y) + z, qlo, qhi);
          const float d = __builtin_fmaf(-(lv - z), s, v[q]);
          acc = __builtin_fmaf(d, d, acc);
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < E; ++q) {
        const float lv = quant_level<SBQ_ROUND_HALF_EVEN>(v[q], s, z, qlo, qhi);
        const float d = v[q] - dequant_level(lv, s, z);
        acc += d * d;
      }
    }
    acc = wave_sum_f32(acc);
    if (lane == 0) lds.acc[i][wid] = acc;
  }
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x < SBQ_MSE_CANDIDATES) {
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w) t += static_cast<double>(lds.acc[threadIdx.x][w]);
  }
  return t;
}

}  // namespace
}  // namespace sbq
