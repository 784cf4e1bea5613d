// sbq_common.hpp -- shared device/host helpers for libsbq (gfx950 only).
//
// Everything here is written for CDNA4: 64-lane wavefronts, 16-byte per-lane
// global accesses, DPP/shuffle wave reductions and LDS cross-wave reductions.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sbq.h"

namespace sbq {

constexpr int kWave = 64;       // gfx950 wavefront
constexpr int kBlock = 256;     // 4 waves: one per SIMD of a CU
constexpr int kWavesPerBlock = kBlock / kWave;

// ---- 16-byte vector payloads -------------------------------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Element tags.  Storage is raw bits; conversion is explicit so that the
// arithmetic is always IEEE fp32 exactly like the reference's CPU path.
struct F32 { using storage = float;    static constexpr int id = SBQ_F32;  };
struct F16 { using storage = uint16_t; static constexpr int id = SBQ_F16;  };
struct BF16 { using storage = uint16_t; static constexpr int id = SBQ_BF16; };

template <typename T> struct Elem;

template <> struct Elem<F32> {
  static constexpr int kVec = 4;  // elements per 16 bytes
  static __device__ __forceinline__ float load1(const void* p, int64_t i) {
    return static_cast<const float*>(p)[i];
  }
  static __device__ __forceinline__ void store1(void* p, int64_t i, float v) {
    static_cast<float*>(p)[i] = v;
  }
};

template <> struct Elem<F16> {
  static constexpr int kVec = 8;
  static __device__ __forceinline__ float from_bits(uint16_t b) {
    _Float16 h;
    __builtin_memcpy(&h, &b, 2);
    return static_cast<float>(h);
  }
  static __device__ __forceinline__ uint16_t to_bits(float v) {
    _Float16 h = static_cast<_Float16>(v);  // v_cvt_f16_f32, RNE
    uint16_t b;
    __builtin_memcpy(&b, &h, 2);
    return b;
  }
  static __device__ __forceinline__ float load1(const void* p, int64_t i) {
    return from_bits(static_cast<const uint16_t*>(p)[i]);
  }
  static __device__ __forceinline__ void store1(void* p, int64_t i, float v) {
    static_cast<uint16_t*>(p)[i] = to_bits(v);
  }
};

template <> struct Elem<BF16> {
  static constexpr int kVec = 8;
  static __device__ __forceinline__ float from_bits(uint16_t b) {
    return __builtin_bit_cast(float, static_cast<uint32_t>(b) << 16);
  }
  static __device__ __forceinline__ uint16_t to_bits(float v) {
    __bf16 h = static_cast<__bf16>(v);  // v_cvt_pk_bf16_f32 on gfx950, RNE
    return __builtin_bit_cast(uint16_t, h);
  }
  static __device__ __forceinline__ float load1(const void* p, int64_t i) {
    return from_bits(static_cast<const uint16_t*>(p)[i]);
  }
  static __device__ __forceinline__ void store1(void* p, int64_t i, float v) {
    static_cast<uint16_t*>(p)[i] = to_bits(v);
  }
};

// A "pack" is the per-lane unit of work of the streaming kernels: 8 elements.
// 16-bit types: one 16-byte access; fp32: two.
constexpr int kPack = 8;

template <bool NT>
__device__ __forceinline__ u32x4 ld16(const void* p) {
  if constexpr (NT) return __builtin_nontemporal_load(static_cast<const u32x4*>(p));
  else return *static_cast<const u32x4*>(p);
}
template <bool NT>
__device__ __forceinline__ void st16(void* p, u32x4 v) {
  if constexpr (NT) __builtin_nontemporal_store(v, static_cast<u32x4*>(p));
  else *static_cast<u32x4*>(p) = v;
}
template <bool NT>
__device__ __forceinline__ u32x2 ld8(const void* p) {
  if constexpr (NT) return __builtin_nontemporal_load(static_cast<const u32x2*>(p));
  else return *static_cast<const u32x2*>(p);
}
template <bool NT>
__device__ __forceinline__ void st8(void* p, u32x2 v) {
  if constexpr (NT) __builtin_nontemporal_store(v, static_cast<u32x2*>(p));
  else *static_cast<u32x2*>(p) = v;
}

template <bool NT>
__device__ __forceinline__ uint32_t ld4(const void* p) {
  if constexpr (NT) return __builtin_nontemporal_load(static_cast<const uint32_t*>(p));
  else return *static_cast<const uint32_t*>(p);
}
template <bool NT>
__device__ __forceinline__ void st4(void* p, uint32_t v) {
  if constexpr (NT) __builtin_nontemporal_store(v, static_cast<uint32_t*>(p));
  else *static_cast<uint32_t*>(p) = v;
}

// load 8 consecutive elements starting at element index i (i % 8 == 0, base
// 16-byte aligned) as fp32
template <typename T, bool NT>
__device__ __forceinline__ void load_pack(const void* base, int64_t i, float (&v)[kPack]) {
  if constexpr (T::id == SBQ_F32) {
    const char* p = static_cast<const char*>(base) + i * 4;
    u32x4 a = ld16<NT>(p), b = ld16<NT>(p + 16);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // copy the lane out first: __builtin_bit_cast on an ext-vector ELEMENT lvalue reads
      // element 0 for every j (clang 22 / ROCm 7.2)
      const uint32_t aj = a[j], bj = b[j];
      v[j] = __builtin_bit_cast(float, aj);
      v[4 + j] = __builtin_bit_cast(float, bj);
    }
  } else {
    const char* p = static_cast<const char*>(base) + i * 2;
    u32x4 a = ld16<NT>(p);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[2 * j] = Elem<T>::from_bits(static_cast<uint16_t>(a[j] & 0xffffu));
      v[2 * j + 1] = Elem<T>::from_bits(static_cast<uint16_t>(a[j] >> 16));
    }
  }
}

// Raw (still packed) form of a pack: lets a kernel keep a second tile in flight in a
// quarter of the registers the unpacked floats would take.
template <typename T>
struct RawPack {
  u32x4 d[T::id == SBQ_F32 ? 2 : 1];
};

template <typename T, bool NT>
__device__ __forceinline__ RawPack<T> load_raw(const void* base, int64_t i) {
  RawPack<T> r;
  if constexpr (T::id == SBQ_F32) {
    const char* p = static_cast<const char*>(base) + i * 4;
    r.d[0] = ld16<NT>(p);
    r.d[1] = ld16<NT>(p + 16);
  } else {
    r.d[0] = ld16<NT>(static_cast<const char*>(base) + i * 2);
  }
  return r;
}

// Split form: the pack's first four elements start at iA, the last four at iB (fp32-output
// kernels give a lane two 4-element runs half a slab apart, so that each of its two 16-byte
// stores -- and fp32 loads -- is part of a fully contiguous 1 KiB wave access instead of a
// stride-32-byte one; half-written 32-byte sectors were 13 % write amplification)
template <typename T, bool NT>
__device__ __forceinline__ RawPack<T> load_raw2(const void* base, int64_t iA, int64_t iB) {
  RawPack<T> r;
  if constexpr (T::id == SBQ_F32) {
    r.d[0] = ld16<NT>(static_cast<const char*>(base) + iA * 4);
    r.d[1] = ld16<NT>(static_cast<const char*>(base) + iB * 4);
  } else {
    const u32x2 a = ld8<NT>(static_cast<const char*>(base) + iA * 2);
    const u32x2 b = ld8<NT>(static_cast<const char*>(base) + iB * 2);
    r.d[0] = u32x4{a[0], a[1], b[0], b[1]};
  }
  return r;
}

// four elements at i (fp32: 16 bytes, 16-bit types: 8 bytes)
template <typename T, bool NT>
__device__ __forceinline__ void store_half(void* base, int64_t i, const float* v) {
  if constexpr (T::id == SBQ_F32) {
    u32x4 a;
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = __builtin_bit_cast(uint32_t, v[j]);
    st16<NT>(static_cast<char*>(base) + i * 4, a);
  } else {
    u32x2 a;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint32_t lo = Elem<T>::to_bits(v[2 * j]), hi = Elem<T>::to_bits(v[2 * j + 1]);
      a[j] = lo | (hi << 16);
    }
    st8<NT>(static_cast<char*>(base) + i * 2, a);
  }
}

template <bool NT>
__device__ __forceinline__ void store_half_f32(void* base, int64_t i, const float* v) {
  u32x4 a;
#pragma unroll
  for (int j = 0; j < 4; ++j) a[j] = __builtin_bit_cast(uint32_t, v[j]);
  st16<NT>(static_cast<char*>(base) + i * 4, a);
}

template <typename T>
__device__ __forceinline__ void unpack_raw(const RawPack<T>& r, float (&v)[kPack]) {
  if constexpr (T::id == SBQ_F32) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t aj = r.d[0][j], bj = r.d[1][j];  // see load_pack: no bit_cast on a vector lane
      v[j] = __builtin_bit_cast(float, aj);
      v[4 + j] = __builtin_bit_cast(float, bj);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t w = r.d[0][j];
      v[2 * j] = Elem<T>::from_bits(static_cast<uint16_t>(w & 0xffffu));
      v[2 * j + 1] = Elem<T>::from_bits(static_cast<uint16_t>(w >> 16));
    }
  }
}

template <typename T, bool NT>
__device__ __forceinline__ void load_pack2(const void* base, int64_t iA, int64_t iB, float (&v)[kPack]) {
  const RawPack<T> r = load_raw2<T, NT>(base, iA, iB);
  unpack_raw<T>(r, v);
}

template <typename T, bool NT>
__device__ __forceinline__ void store_pack(void* base, int64_t i, const float (&v)[kPack]) {
  if constexpr (T::id == SBQ_F32) {
    char* p = static_cast<char*>(base) + i * 4;
    u32x4 a, b;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a[j] = __builtin_bit_cast(uint32_t, v[j]);
      b[j] = __builtin_bit_cast(uint32_t, v[4 + j]);
    }
    st16<NT>(p, a);
    st16<NT>(p + 16, b);
  } else {
    char* p = static_cast<char*>(base) + i * 2;
    u32x4 a;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t lo = Elem<T>::to_bits(v[2 * j]);
      uint32_t hi = Elem<T>::to_bits(v[2 * j + 1]);
      a[j] = lo | (hi << 16);
    }
    st16<NT>(p, a);
  }
}

// ---- wave / block reductions ---------------------------------------------------
template <typename V, typename Op>
__device__ __forceinline__ V wave_reduce(V v, Op op) {
#pragma unroll
  for (int m = kWave / 2; m > 0; m >>= 1) v = op(v, __shfl_xor(v, m, kWave));
  return v;
}

// Reduce across the 4 waves of a 256-thread block.  `slot` is LDS scratch with
// at least kWavesPerBlock entries; result valid in every thread.
template <typename V, typename Op>
__device__ __forceinline__ V block_reduce(V v, Op op, V* slot) {
  v = wave_reduce(v, op);
  const int lane = threadIdx.x & (kWave - 1);
  const int wid = threadIdx.x / kWave;
  __syncthreads();  // protect slot reuse
  if (lane == 0) slot[wid] = v;
  __syncthreads();
  V r = slot[0];
#pragma unroll
  for (int w = 1; w < kWavesPerBlock; ++w) r = op(r, slot[w]);
  return r;
}

// Wave-level inclusive scan / reductions of a 32-bit value on DPP (row shifts inside the 16-lane rows, then the two row
// broadcasts of gfx9): ~8 vector instructions instead of six ds_bpermute round trips through the LDS crossbar
// (~100 cycles each) -- the plan runs a scan and three reductions on the critical path of every workgroup.
template <typename Op>
__device__ __forceinline__ uint32_t dpp_scan_u32(uint32_t v, uint32_t identity, Op op) {
  const int id = static_cast<int>(identity);
#define SBQ_DPP(CTRL, ROWMASK) \
  v = op(v, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(id, static_cast<int>(v), CTRL, ROWMASK, 0xf, false)))
  SBQ_DPP(0x111, 0xf);  // row_shr:1
  SBQ_DPP(0x112, 0xf);  // row_shr:2
  SBQ_DPP(0x114, 0xf);  // row_shr:4
  SBQ_DPP(0x118, 0xf);  // row_shr:8
  SBQ_DPP(0x142, 0xa);  // row_bcast:15 into rows 1 and 3
  SBQ_DPP(0x143, 0xc);  // row_bcast:31 into rows 2 and 3
#undef SBQ_DPP
  return v;
}
template <typename Op>
__device__ __forceinline__ uint32_t dpp_reduce_u32(uint32_t v, uint32_t identity, Op op) {  // result in every lane
  v = dpp_scan_u32(v, identity, op);
  return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), kWave - 1));
}

// Sum of an fp32 value over the wave on DPP, the result in every lane: six dependent vector adds instead of six
// ds_bpermute round trips.  The association is the scan's (fixed, the same on every call): callers that must agree
// bit for bit (the MSE observer's per-tensor and model-wide kernels) all use THIS function.
__device__ __forceinline__ float wave_sum_f32(float a) {
  const uint32_t r = dpp_reduce_u32(__builtin_bit_cast(uint32_t, a), 0u, [](uint32_t x, uint32_t y) {
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(float, x) + __builtin_bit_cast(float, y));
  });
  return __builtin_bit_cast(float, r);
}

// A 16-bit tensor has 65 536 values: its keys are its own bit patterns through the same sign transform in 16 bits,
//   key16(b) = (b ^ (b < 0 ? 0xffff : 0x8000)) - key16'(-inf)      (mod 2^16; negative NaNs wrap to the top)
// and the engine works on key32 = key16 << 16 with min_shift 16, so plan, windows (always 2^16 aligned) and advance
// are untouched.  Both keys of a dword come out of five PACKED operations (and, arithmetic shift, or, xor, subtract:
// 2.5 per element) plus one shift / mask each to feed the window tests -- 3.5 operations per element instead of 6.
template <typename T>
struct Key16 {
  static constexpr uint32_t kNegInf = T::id == SBQ_BF16 ? 0xff80u : 0xfc00u;
  static constexpr uint32_t kRot = (~kNegInf) & 0xffffu;                       // key16'(-inf) before the rotation
  static constexpr uint32_t kZero = ((0x7fffu - kRot) & 0xffffu) << 16;        // key32(-0): keys below are x < 0
  static constexpr uint32_t kInf = ((((kNegInf & 0x7fffu) | 0x8000u) - kRot) & 0xffffu) << 16;  // key32(+inf)
  // two keys, packed like the two elements of the dword; amask2 = 0x7fff7fff for |x|, else all ones
  static __device__ __forceinline__ uint32_t pack2(uint32_t w, uint32_t amask2) {
    typedef int16_t i16x2 __attribute__((ext_vector_type(2)));
    typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
    w &= amask2;
    const uint32_t m = __builtin_bit_cast(uint32_t, __builtin_bit_cast(i16x2, w) >> static_cast<int16_t>(15));
    const uint32_t t = w ^ (m | 0x80008000u);
    const u16x2 rot = {static_cast<uint16_t>(kRot), static_cast<uint16_t>(kRot)};
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, t) - rot);
  }
  static __device__ __forceinline__ uint32_t one(uint32_t b16, bool use_abs) {  // key32 of one raw element
    return pack2(b16, use_abs ? 0x7fff7fffu : 0xffffffffu) << 16;
  }
  static __device__ __forceinline__ float value(uint32_t key32) {
    const uint32_t t = ((key32 >> 16) + kRot) & 0xffffu;
    const uint32_t b = (t & 0x8000u) ? (t & 0x7fffu) : (~t & 0xffffu);
    return Elem<T>::from_bits(static_cast<uint16_t>(b));
  }
};
// torch.min / torch.max semantics: NaN wins.
struct NanMin {
  __device__ __forceinline__ float operator()(float a, float b) const {
    return (a != a || a < b) ? a : ((b != b) ? b : (b < a ? b : a));
  }
};
struct NanMax {
  __device__ __forceinline__ float operator()(float a, float b) const {
    return (a != a || a > b) ? a : ((b != b) ? b : (b > a ? b : a));
  }
};
struct Sum {
  template <typename V>
  __device__ __forceinline__ V operator()(V a, V b) const { return a + b; }
};

// ---- the QDQ scalar core ---------------------------------------------------------
// quant_tensor.py:182-184:  zp = round(zp); q = clamp(round(x/s) + zp, qmin, qmax);
// dq = (q - zp) * s -- every step an individually rounded IEEE fp32 operation
// (build with -ffp-contract=off).  `zp` is already rounded, qlo/qhi are floats.
template <int ROUND>
__device__ __forceinline__ float round_mode(float t) {
  if constexpr (ROUND == SBQ_ROUND_HALF_EVEN) return __builtin_rintf(t);  // v_rndne_f32
  else if constexpr (ROUND == SBQ_ROUND_HALF_UP) return __builtin_floorf(t + 0.5f);
  else return __builtin_ceilf(t - 0.5f);
}

template <int ROUND>
__device__ __forceinline__ float quant_level(float x, float s, float zp, float qlo, float qhi) {
  float t = x / s;  // IEEE-correct division (v_div_scale/fmas/fixup)
  float v = round_mode<ROUND>(t) + zp;
  // torch.clamp propagates NaN; med3/min/max would drop it
  float c = __builtin_fminf(__builtin_fmaxf(v, qlo), qhi);
  return (v != v) ? v : c;
}

__device__ __forceinline__ float dequant_level(float q, float s, float zp) {
  return (q - zp) * s;
}

// ---- exact x/s without the IEEE division sequence ------------------------------------
// When the divisor is uniform over many elements (one scale per channel row) the
// 10-instruction v_div_scale/rcp/fma/div_fmas/div_fixup expansion is replaced by
//     q0 = x * y            with y = RN(1/s), computed ONCE per row by a true division
//     r0 = fma(-q0, s, x);  q1 = fma(r0, y, q0)      -- q1 is within (1/2 + 2^-23) ulp of x/s
//     r1 = fma(-q1, s, x);  q2 = fma(r1, y, q1)      -- q2 == RN(x/s)
// The last step is Markstein's theorem (Muller et al., Handbook of Floating-Point
// Arithmetic, Thm "Markstein"): if y approximates 1/s with relative error < 2^-24 (true
// for y = RN(1/s)), q1 is a faithful rounding of x/s and r1 = x - s*q1 is computed exactly
// (it is, for a faithful q1, as long as nothing underflows), then RN(q1 + r1*y) is the
// correctly rounded quotient.  q0 alone may be 2 ulp off, hence the first refinement.
// Range conditions are enforced by the caller: s in [2^-60, 2^60] (else the IEEE path is
// taken for that row) and |x| < s*2^40 (NaN, inf or larger magnitudes send the whole wave's
// pack through the IEEE path -- a wave-wide vote on one compare per element -- or, in
// quant_level_fast, are clamped / restored per element); for |x/s| < 1/4 the residuals may
// underflow, which cannot move the quotient across 0.5.  tests/test_gpu_parity.py::test_fast_division_equals_ieee
// compares this against the IEEE path bit for bit on adversarial data.
__device__ __forceinline__ bool fast_div_ok(float s) {
  return s >= 0x1p-60f && s <= 0x1p60f;  // false for NaN, subnormal, inf
}

__device__ __forceinline__ float fast_div(float x, float s, float y) {
  float q = x * y;
  float r = __builtin_fmaf(-q, s, x);
  q = __builtin_fmaf(r, y, q);
  r = __builtin_fmaf(-q, s, x);
  return __builtin_fmaf(r, y, q);
}

// two quotients at a time: the same five operations as fast_div on v_pk_mul_f32 / v_pk_fma_f32
// (each lane of a packed op is an individually rounded IEEE operation: identical results)
__device__ __forceinline__ f32x2 fast_div2(f32x2 x, float s, float y) {
  const f32x2 yv = {y, y}, ns = {-s, -s};
  f32x2 q = x * yv;
  f32x2 r = __builtin_elementwise_fma(q, ns, x);
  q = __builtin_elementwise_fma(r, yv, q);
  r = __builtin_elementwise_fma(q, ns, x);
  return __builtin_elementwise_fma(r, yv, q);
}

// quant_level with the fast division; `bound` = s * 2^40, `y` = 1/s (IEEE)
__device__ __forceinline__ float quant_level_fast(float x, float s, float y, float bound, float zp,
                                                  float qlo, float qhi) {
  const float xc = __builtin_amdgcn_fmed3f(x, -bound, bound);
  const float v = __builtin_rintf(fast_div(xc, s, y)) + zp;
  const float c = __builtin_amdgcn_fmed3f(v, qlo, qhi);
  return (x != x) ? x : c;  // torch.clamp propagates NaN
}

// observers/base.py:63-79 in fp32, op for op.
__device__ __forceinline__ void qparams_from_minmax(float mn, float mx, float qrange,
                                                    bool symmetric, float& scale, float& zp) {
  NanMin nmin;
  NanMax nmax;
  float mn_neg = nmin(mn, 0.0f);  // torch.minimum: NaN propagates
  float mx_pos = nmax(mx, 0.0f);
  if (symmetric) {
    mx_pos = nmax(-mn_neg, mx_pos);
    float s = (mx_pos * 2.0f) / qrange;
    scale = nmax(s, 1e-6f);
    zp = 0.0f;
  } else {
    float s = (mx_pos - mn_neg) / qrange;
    scale = nmax(s, 1e-6f);
    zp = __builtin_rintf(-mn_neg / scale);
  }
}

// ---- order-preserving float <-> uint32 key --------------------------------------
__device__ __forceinline__ uint32_t float_key(float f, bool use_abs) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  if (use_abs) u &= 0x7fffffffu;
  if (f != f) return 0xffffffffu;          // NaN sorts last (torch.sort / kthvalue)
  if (u == 0x80000000u) u = 0;             // -0 == +0
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_float(uint32_t k) {
  if (k == 0xffffffffu) return __builtin_nanf("");
  uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __builtin_bit_cast(float, u);
}

// ---- host side ---------------------------------------------------------------------
int check_launch();  // hipGetLastError -> sbq_status, records the error string
int knob(int which);
// zero-contract workspace regions (sbq_core.hip): zeroes a region the library has not seen before, refuses one that a
// call on another stream may still be using (SBQ_ERR_BUSY)
int workspace_guard(void* region, size_t bytes, hipStream_t st);
uint32_t cu_count();  // compute units of the current device
// sample-guided windowed selection of a whole tensor (sbq_select_win.hip)
size_t win_select_workspace_bytes();
int win_select_run(const void* const* shards, const int64_t* counts, int n_shards, int x_dtype, int use_abs, int n_sel,
                   bool percentile, double alpha, int64_t k0, int64_t k1, float* out0, float* out1, void* workspace,
                   size_t workspace_bytes, hipStream_t st);
inline hipStream_t as_stream(void* s) { return static_cast<hipStream_t>(s); }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline size_t dtype_size(int dt) { return dt == SBQ_F32 ? 4 : 2; }
inline bool valid_dtype(int dt) { return dt == SBQ_F32 || dt == SBQ_F16 || dt == SBQ_BF16; }
inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }


// ---- chunk geometry shared by the reductions ------------------------------------
// A row is one (outer index, channel) run of `inner` contiguous elements.  It is
// cut into chunks of `chunk_elems` elements; a workgroup handles one chunk and the
// linear block id enumerates [channel][outer][chunk-in-row].
struct ChunkGeom {
  int64_t inner;
  uint32_t C;
  uint32_t outer;
  uint32_t chunks_per_row;
  uint32_t chunk_elems;      // elements per chunk (multiple of 8)
  uint32_t chunks_per_chan;  // outer * chunks_per_row
};

inline bool geom_ok(int64_t outer, int64_t C, int64_t inner, uint32_t chunk) {
  if (outer <= 0 || C <= 0 || inner <= 0) return false;
  if (outer >= (1ll << 31) || C >= (1ll << 31)) return false;
  const int64_t cpr = ceil_div(inner, chunk);
  if (cpr * outer >= (1ll << 31)) return false;
  if (cpr * outer * C >= (1ll << 31)) return false;
  return true;
}

inline ChunkGeom make_geom(int64_t outer, int64_t C, int64_t inner, uint32_t chunk) {
  ChunkGeom g;
  g.inner = inner;
  g.C = static_cast<uint32_t>(C);
  g.outer = static_cast<uint32_t>(outer);
  g.chunk_elems = chunk;
  g.chunks_per_row = static_cast<uint32_t>(ceil_div(inner, chunk));
  g.chunks_per_chan = g.outer * g.chunks_per_row;
  return g;
}

// decode blockIdx.x -> (channel, first element of the chunk, one-past-last)
struct ChunkPos {
  uint32_t c;
  int64_t row_base, begin, end;
};
__device__ __forceinline__ ChunkPos chunk_pos(const ChunkGeom& g, uint32_t bid) {
  ChunkPos p;
  p.c = bid / g.chunks_per_chan;
  const uint32_t j = bid - p.c * g.chunks_per_chan;
  const uint32_t o = j / g.chunks_per_row;
  const uint32_t k = j - o * g.chunks_per_row;
  p.row_base = (static_cast<int64_t>(o) * g.C + p.c) * g.inner;
  p.begin = static_cast<int64_t>(k) * g.chunk_elems;
  p.end = p.begin + g.chunk_elems;
  if (p.end > g.inner) p.end = g.inner;
  return p;
}

template <typename F>
inline int dispatch_dtype(int dt, F&& f) {
  if (dt == SBQ_F32) f(F32());
  else if (dt == SBQ_F16) f(F16());
  else if (dt == SBQ_BF16) f(BF16());
  else return SBQ_ERR_DTYPE;
  return SBQ_OK;
}

// can the rows be read as whole 16-byte packs?  (a single row may have a ragged
// tail, which the kernels read element-wise)
inline bool pack_friendly(const void* x, int64_t C, int64_t outer, int64_t inner) {
  if (!aligned16(x)) return false;
  if (inner % kPack == 0) return true;
  return C == 1 && outer == 1;
}

}  // namespace sbq
