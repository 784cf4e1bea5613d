// sbq_qdq_resident.hip -- forward quantize-dequantize of a tensor that fits the chip's register files in one
// sitting (the headline 4096x4096 weight).  Same arithmetic as sbq_qdq.hip (sbq_qdq_math.hpp), different
// schedule.  Replaces QuantizePerChannelForwardCUDA / QuantizePerTensorForwardCUDA
// (sparsebit/quantization/torch_extensions/fake_quant_tensor.cu:50-66,170-188) for those shapes.
#include <utility>

#include "sbq_observe_body.hpp"
#include "sbq_qdq_math.hpp"

namespace sbq {
namespace {

// ---- resident schedule ---------------------------------------------------------------------------
// A tensor that fits the chip's register files in ONE sitting (n_slabs <= CUs * kResSub * U: the
// headline 4096x4096 weight is exactly 256 * 2 * 16 slabs) is not pipelined at all: one workgroup of 512
// per CU, every wave issues the loads of all its U slabs up front, converts each pack as it lands
// (progressive vmcnt waits) keeping the result in registers, and only after its last conversion issues
// its U stores back to back.  Chip-wide the launch is a read burst followed by a write burst instead of
// a read/write mix for its whole duration, and the arithmetic hides under the tail of the read burst.
// Measured (tools/lab/qdq_lab.hip, profiles/r02_lab_*.log; launch to launch, 4096x4096 bf16, rotating
// buffers): 11.1 us against 12.5 us for the best pipelined variant and 12.2 us for a pipelined COPY.
// Workgroup = kResSub sub-blocks of 256 lanes, each one slab wide: slab(u) = tile*kResSub*U + u*kResSub + sub,
// so one load instruction of the workgroup covers kResSub adjacent slabs.
constexpr int kResBlock = 512;
constexpr int kResSub = kResBlock / kBlock;

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>)
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

template <typename Tout>
struct OutPack {
  u32x4 d[Tout::id == SBQ_F32 ? 2 : 1];
};

// Every access is a raw buffer access `buffer_load/store ... v_lane_offset, s[descriptor], s_slab_offset offen nt`:
// one descriptor per tensor, the slab's byte offset in an SGPR, the lane's offset in ONE VGPR shared by all slabs --
// no per-slab 64-bit VGPR address, and (unlike hand-issued asm loads) the compiler keeps the vmcnt bookkeeping, so
// each slab's conversion waits exactly for its own loads.  aux 2 = nt (streamed once).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 bld16(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 2);
}
__device__ __forceinline__ u32x2 bld8(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
  return __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 2);
}
__device__ __forceinline__ uint32_t bld4(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
  return __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 2);
}
// The s_nop: a buffer store of more than 64 bits reads its data registers a cycle after issue; a VALU write to
// them in the very next slot corrupts the stored value.  The compiler inserts that wait state only when the store
// has NO SGPR soffset (GCNHazardRecognizer::createsVALUHazard) -- with one, gfx950 showed the hazard all the same:
// qdq_observe_kernel<BF16, F32> stored 2^40 (the next slab's range constant, moved into a data register right
// behind the store) in a few lanes of a few rows.  One wait state restores it; tests/test_gpu_observe_fused.py.
__device__ __forceinline__ void bst16(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, u32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, 2);
  asm volatile("s_nop 1");
}
// two fp32 -> one dword of two 16-bit values (RNE), as ONE packed convert
template <typename T>
__device__ __forceinline__ uint32_t pack2(float a, float b) {
  if constexpr (T::id == SBQ_BF16) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2));
  } else {
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, f16x2));
  }
}

// Eligible geometry (try_resident): rows made of whole slabs (inner % 2048 == 0), so slab sl simply starts at
// element sl * 2048 and its channel is sl / slabs_per_row (outer == 1 or per tensor): the address arithmetic in
// front of the first load is one scalar multiply-add, not a cursor.
//
// Code layout.  The arithmetic has three forms (sbq_qdq_math.hpp: reciprocal + fma refinement with zero point 0,
// the same with a zero point, IEEE division for scales outside [2^-60, 2^60] or elements that are NaN / inf /
// >= s * 2^40).  Unrolled over 16 slabs with all three forms inline, the kernel is 50 KB of code whose executed
// lines are scattered over all of it -- instruction fetches that queue behind the read burst.  So the choice is
// made ONCE per wave: both fast forms exist as compact straight-line blocks (no branch inside), they only
// accumulate an "odd element" flag, and a wave that saw one -- or holds a row the fast division does not cover --
// redoes its tile in a rolled loop through the generic quantize_pack (cold code at the end of the kernel).
// (yr = 1 / s, correctly rounded: the callers divide ONCE for all of a wave's slabs -- lane u for slab u -- instead of
// every lane repeating the 12-instruction IEEE sequence for every slab: a sixth of the kernel's vector instructions)
// (PER_SLAB: the round-5 form -- the division and eight compares inside every slab's block.  The fp32-output kernels
// keep it: with the shorter form they measured 17.55 -> 18.6 us on the same box, their stores being the longer phase.)
template <int MASK, bool ZP, bool PER_SLAB = false>
__device__ __forceinline__ void fast_pack(float (&v)[kPack], const u32x2 mk, float thr, float s, float yr, float z, float qlo,
                                          float qhi, float (&dq)[kPack], bool& odd) {
#pragma unroll
  for (int j = 0; j < kPack; ++j) {
    if constexpr (MASK == MASK_BYTES) {
      const uint32_t byte = (mk[j >> 2] >> (8 * (j & 3))) & 0xffu;
      v[j] = byte ? v[j] : 0.0f;
    } else if constexpr (MASK == MASK_THRESH) {
      v[j] = (__builtin_fabsf(v[j]) > thr) ? v[j] : 0.0f;
    }
  }
  const float bound = s * 0x1p40f;
  if constexpr (PER_SLAB) {
    yr = 1.0f / s;
#pragma unroll
    for (int j = 0; j < kPack; ++j) odd |= !(__builtin_fabsf(v[j]) < bound);
  } else {
    // NaN, +-inf or |x| >= s * 2^40 anywhere in the pack: one compare on the NaN-propagating maximum of the eight
    // magnitudes (v_maximum3_f32) instead of eight compares
    float m = __builtin_fabsf(v[0]);
#pragma unroll
    for (int j = 1; j < kPack; ++j) m = __builtin_elementwise_maximum(m, __builtin_fabsf(v[j]));
    odd |= !(m < bound);
  }
#pragma unroll
  for (int j = 0; j < kPack; j += 2) {
    const f32x2 t = fast_div2(f32x2{v[j], v[j + 1]}, s, yr);
    if constexpr (ZP) {
      const float l0 = __builtin_amdgcn_fmed3f(__builtin_rintf(t[0]) + z, qlo, qhi);
      const float l1 = __builtin_amdgcn_fmed3f(__builtin_rintf(t[1]) + z, qlo, qhi);
      dq[j] = dequant_level(l0, s, z);
      dq[j + 1] = dequant_level(l1, s, z);
    } else {
      // fma(lv, s, +0): a level of -0 still dequantizes to +0 like (lv - 0) * s does
      const float l0 = __builtin_amdgcn_fmed3f(__builtin_rintf(t[0]), qlo, qhi);
      const float l1 = __builtin_amdgcn_fmed3f(__builtin_rintf(t[1]), qlo, qhi);
      const f32x2 d = __builtin_elementwise_fma(f32x2{l0, l1}, f32x2{s, s}, f32x2{0.0f, 0.0f});
      dq[j] = d[0];
      dq[j + 1] = d[1];
    }
  }
}

template <typename Tout>
__device__ __forceinline__ void pack_out(const float (&dq)[kPack], OutPack<Tout>& o) {
  if constexpr (Tout::id == SBQ_F32) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o.d[0][j] = __builtin_bit_cast(uint32_t, dq[j]);
      o.d[1][j] = __builtin_bit_cast(uint32_t, dq[4 + j]);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) o.d[0][j] = pack2<Tout>(dq[2 * j], dq[2 * j + 1]);
  }
}

template <typename Tin, typename Tout, int MASK, int U>
__global__ __launch_bounds__(kResBlock) void qdq_resident_kernel(
    const void* __restrict__ x, uint32_t n_slabs, uint32_t row_inv, uint32_t lsq, uint32_t slabs_per_row, float qlo,
    float qhi, void* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ zero_point,
    // ---- not preloaded ----
    const uint8_t* __restrict__ mask, const float* __restrict__ thresh) {
  constexpr bool SPLIT = Tout::id == SBQ_F32;  // same lane mapping as qdq_pack_kernel
  constexpr uint32_t kIn = Tin::id == SBQ_F32 ? 4 : 2, kOut = Tout::id == SBQ_F32 ? 4 : 2;
  constexpr uint32_t kSlabElems = kBlock * kPack;
  // Everything from here to the choice of the arithmetic form is ONE basic block (selects, no branches): the
  // scheduling barriers below then pin the order "all data loads, then the scale loads, then anything else".
  // (With a branch in between, the loads are sunk below it -- behind a chain of dependent scalar loads.)
  const uint32_t sub = __builtin_amdgcn_readfirstlane(threadIdx.x / kBlock);
  const uint32_t tid = threadIdx.x % kBlock;
  // the lane's run(s) inside a slab, in elements
  const uint32_t laneA = SPLIT ? 4 * tid : kPack * tid;
  const uint32_t laneB = laneA + kSlabElems / 2;  // SPLIT only
  // one tile per workgroup (grid == n_tiles): no loop, nothing for the compiler to hoist in front of the loads
  const uint32_t sl0 = blockIdx.x * (kResSub * U) + sub;  // slab(u) = sl0 + u * kResSub
  RawPack<Tin> raw[U];
  u32x2 mk[U];
  const uint32_t inA = laneA * kIn, inB = laneB * kIn;
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, n_slabs * (kSlabElems * kIn));
  const __amdgpu_buffer_rsrc_t ry = make_rsrc(y, n_slabs * (kSlabElems * kOut));
  __amdgpu_buffer_rsrc_t rm = rx;
  if constexpr (MASK == MASK_BYTES) rm = make_rsrc(mask, n_slabs * kSlabElems);
  uint32_t slc[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    // slabs past the end read slab 0 (valid memory) and store nothing
    const uint32_t sl = sl0 + u * kResSub;
    slc[u] = sl < n_slabs ? sl : 0u;
    const uint32_t so = slc[u] * (kSlabElems * kIn);
    if constexpr (SPLIT) {
      if constexpr (Tin::id == SBQ_F32) {
        raw[u].d[0] = bld16(rx, inA, so);
        raw[u].d[1] = bld16(rx, inB, so);
      } else {
        const u32x2 a = bld8(rx, inA, so), b = bld8(rx, inB, so);
        raw[u].d[0] = u32x4{a[0], a[1], b[0], b[1]};
      }
      if constexpr (MASK == MASK_BYTES) mk[u] = u32x2{bld4(rm, laneA, slc[u] * kSlabElems), bld4(rm, laneB, slc[u] * kSlabElems)};
    } else {
      raw[u].d[0] = bld16(rx, inA, so);
      if constexpr (MASK == MASK_BYTES) mk[u] = bld8(rm, laneA, slc[u] * kSlabElems);
    }
  }
  __builtin_amdgcn_sched_barrier(0);  // all data loads are in flight before any other work
  // channel of slab sl = sl / slabs_per_row = (2 sl) / (2 slabs_per_row), as one multiply-high by the host's
  // reciprocal of 2 * slabs_per_row (the doubling keeps the reciprocal in 32 bits for one slab per row; exact while
  // 4 * sl * slabs_per_row < 2^32, i.e. for every slab count try_resident admits).  Per tensor the host passes
  // slabs_per_row = n_slabs: channel 0 throughout.  No select, no branch: see the note on basic blocks above.
  float sc[U], zp[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint32_t c = __builtin_amdgcn_readfirstlane(__umulhi(slc[u] * 2u, row_inv));
    sc[u] = uniform_load(scale, c);
    zp[u] = uniform_load(zero_point, c);
  }
  float thr = 0.0f;
  if constexpr (MASK == MASK_THRESH) thr = *thresh;
  __builtin_amdgcn_sched_barrier(0);  // ... and so are all scale / zero-point loads
  bool all_fast = true, all_zero = true;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    // LSQ pre-ops on the raw parameters (lsq.py:61-62), as selects
    const float s_ = lsq ? __builtin_fabsf(sc[u]) : sc[u];
    const float z_ = lsq ? __builtin_amdgcn_fmed3f(zp[u], qlo, qhi) : zp[u];
    sc[u] = s_;
    zp[u] = __builtin_rintf(z_);
    all_fast &= fast_div_ok(sc[u]);
    all_zero &= zp[u] == 0.0f;
  }
  // the reciprocals of all U scales in one division: lane u holds slab u's
  constexpr bool PER_SLAB = Tout::id == SBQ_F32;
  float yr[U];
  if constexpr (PER_SLAB) {
#pragma unroll
    for (int u = 0; u < U; ++u) yr[u] = 0.0f;  // (fast_pack divides)
  } else {
    float sv = 1.0f;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      // sv[lane u] = sc[u]: one select under a constant lane mask (sc[u] is the same in every lane)
      const unsigned long long only_u = 1ull << u;
      asm("v_cndmask_b32 %0, %0, %1, %2" : "+v"(sv) : "v"(sc[u]), "s"(only_u));
    }
    const float yv = 1.0f / sv;
#pragma unroll
    for (int u = 0; u < U; ++u) yr[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, yv), u));
  }
  __builtin_amdgcn_sched_barrier(0);
  if (all_fast) {  // wave-uniform
    // same-width input and output: the result overwrites the slab's input registers (an explicit alias: left to
    // the register allocator the fp32 -> fp32 kernel keeps both and spills)
    constexpr bool INPLACE = (Tin::id == SBQ_F32) == (Tout::id == SBQ_F32);
    OutPack<Tout> out_sep[INPLACE ? 1 : U];
    auto out_d0 = [&](int u) -> u32x4& {
      if constexpr (INPLACE) return raw[u].d[0];
      else return out_sep[u].d[0];
    };
    auto out_d1 = [&](int u) -> u32x4& {  // fp32 output only
      if constexpr (INPLACE) return raw[u].d[Tin::id == SBQ_F32 ? 1 : 0];
      else return out_sep[u].d[Tout::id == SBQ_F32 ? 1 : 0];
    };
    bool odd = false;
    auto convert = [&](auto zp_tag) {
      constexpr bool ZP = decltype(zp_tag)::value;
      // the two instances start with DIFFERENT statements: identical leading code (unpacking slab after slab) would
      // be hoisted into the common predecessor, i.e. every slab held unpacked in registers at once
      static_for<U>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        // The slab's registers pass through a statement that differs between the two forms: code that is identical
        // in both (unpacking slab after slab) would otherwise be hoisted into their common predecessor -- every slab
        // unpacked and held in registers at once.  It is also where the compiler places this slab's vmcnt wait.
        if constexpr (ZP) {
          asm volatile("; zero-point form" : "+v"(raw[u].d[0]));
          if constexpr (Tin::id == SBQ_F32) asm volatile("; zero-point form" : "+v"(raw[u].d[1]));
        } else {
          asm volatile("; symmetric form" : "+v"(raw[u].d[0]));
          if constexpr (Tin::id == SBQ_F32) asm volatile("; symmetric form" : "+v"(raw[u].d[1]));
        }
        float v[kPack], dq[kPack];
        unpack_raw<Tin>(raw[u], v);
        fast_pack<MASK, ZP, PER_SLAB>(v, mk[u], thr, sc[u], yr[u], zp[u], qlo, qhi, dq, odd);
        OutPack<Tout> o;
        pack_out<Tout>(dq, o);
        out_d0(u) = o.d[0];
        if constexpr (Tout::id == SBQ_F32) out_d1(u) = o.d[1];
        // the packed result exists HERE (otherwise the converts sink into the store phase)
        asm volatile("" : "+v"(out_d0(u)));
        if constexpr (Tout::id == SBQ_F32) asm volatile("" : "+v"(out_d1(u)));
      });
    };
    if (all_zero) convert(std::false_type{});
    else convert(std::true_type{});
    __builtin_amdgcn_sched_barrier(0);  // the first store is issued after the last conversion
    if (__builtin_amdgcn_ballot_w64(odd) == 0) {
      uint32_t n_end = n_slabs;
      asm volatile("" : "+s"(n_end));  // a fresh compare per store instead of 16 saved (and spilled) lane masks
      const uint32_t outA = laneA * kOut, outB = laneB * kOut;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t sl = sl0 + u * kResSub;
        if (sl < n_end) {  // workgroup-uniform
          bst16(ry, outA, sl * (kSlabElems * kOut), out_d0(u));
          if constexpr (SPLIT) bst16(ry, outB, sl * (kSlabElems * kOut), out_d1(u));
        }
      }
      return;
    }
  }
  // ---- cold path: this wave's tile again, slab by slab, through the generic arithmetic -----------------------
  // (It "uses" the registers of the up-front loads: values consumed on the hot side only would be SUNK into it by
  // the optimiser, i.e. the data loads would be issued behind the scale loads and the branch that depends on them.)
#pragma unroll
  for (int u = 0; u < U; ++u) {
    asm volatile("" : : "v"(raw[u].d[0]));
    if constexpr (Tin::id == SBQ_F32) asm volatile("" : : "v"(raw[u].d[1]));
    if constexpr (MASK == MASK_BYTES) asm volatile("" : : "v"(mk[u]));
  }
  {
#pragma nounroll
    for (uint32_t u = 0; u < static_cast<uint32_t>(U); ++u) {
      const uint32_t sl = sl0 + u * kResSub;
      if (sl >= n_slabs) break;
      const uint32_t c = sl / slabs_per_row;
      float s_ = uniform_load(scale, c), z_ = uniform_load(zero_point, c);
      if (lsq) {
        s_ = __builtin_fabsf(s_);
        z_ = __builtin_amdgcn_fmed3f(z_, qlo, qhi);
      }
      z_ = __builtin_rintf(z_);
      const int64_t e0 = static_cast<int64_t>(sl) * kSlabElems;
      RawPack<Tin> r;
      u32x2 m{};
      if constexpr (SPLIT) {
        r = load_raw2<Tin, true>(x, e0 + laneA, e0 + laneB);
        if constexpr (MASK == MASK_BYTES) m = u32x2{ld4<true>(mask + e0 + laneA), ld4<true>(mask + e0 + laneB)};
      } else {
        r = load_raw<Tin, true>(x, e0 + laneA);
        if constexpr (MASK == MASK_BYTES) m = ld8<true>(mask + e0 + laneA);
      }
      float v[kPack], lv[kPack], dq[kPack];
      unpack_raw<Tin>(r, v);
      quantize_pack<MASK, false, MATH_FAST>(v, m, thr, s_, z_, qlo, qhi, lv, dq);
      if constexpr (SPLIT) {
        store_half_f32<true>(y, e0 + laneA, dq);
        store_half_f32<true>(y, e0 + laneB, dq + 4);
      } else {
        store_pack<Tout, true>(y, e0 + laneA, dq);
      }
    }
  }
}

// ---- fused observe + QDQ of a weight --------------------------------------------------
// min-max observer -> scale / zero point -> quantize-dequantize in ONE read of the weight: 4 bytes per element
// (bf16 in and out) instead of 2 (statistics) + 4 (QDQ) in two launches.  Replaces, for a per-channel weight,
//   observers/minmax.py:14-25 + observers/base.py:63-79 (calc_qparams_with_minmax) + quantizers/base.py:55-64.
// Same workgroup shape as the resident kernel: 512 threads = 2 sub-blocks of 256 lanes, a wave holds U slabs in
// registers.  A row is 1 or 2 whole slabs (inner = 2048 or 4096), so with slab(u) = sl0 + 2u + sub a row's
// elements sit in ONE workgroup at ONE u: its extrema are a wave reduction, an LDS exchange between the 4 (8)
// waves and a single barrier for all U rows; every lane then derives the row's scale / zero point itself
// (qparams_from_minmax: the reference's fp32 operations) and converts its slab in place.  NaN propagates like
// torch.min / max: v_min / v_max drop it, a separate flag restores it (as in stats_partial_kernel).
// Statistics of a slab, reduced over the wave for all U slabs AT ONCE: per-lane accumulators (16-bit inputs: the three
// packed integer words of sbq_observe_body.hpp -- no unpack, 1.5 operations per element; fp32: v_minimum3 /
// v_maximum3, NaN-propagating), then a transposing butterfly -- at each of the first log2(U) exchange steps a lane
// keeps half of its slabs and hands the other half to its partner, so U slabs cost U - 1 + (6 - log2 U) shuffles per
// word instead of 6 U.  Afterwards lane l holds the wave's result of slab
//   U == 8: (l >> 5 & 1) * 4 + (l >> 4 & 1) * 2 + (l >> 3 & 1)        U == 4: (l >> 5 & 1) * 2 + (l >> 4 & 1)
// in every lane of its group (round 2: 2 wave reductions + a ballot per slab, after unpacking it: 16 reductions of
// 6 shuffles each for U = 8).
struct SlabAccF32 {
  float mn, mx;
  __device__ __forceinline__ static SlabAccF32 of(const RawPack<F32>& r) {
    float v[kPack];
    unpack_raw<F32>(r, v);
    SlabAccF32 a{v[0], v[0]};
#pragma unroll
    for (int j = 1; j < kPack; j += 2) {
      const float w = j + 1 < kPack ? v[j + 1] : v[j];
      a.mn = __builtin_elementwise_minimum(__builtin_elementwise_minimum(a.mn, v[j]), w);
      a.mx = __builtin_elementwise_maximum(__builtin_elementwise_maximum(a.mx, v[j]), w);
    }
    return a;
  }
  __device__ __forceinline__ SlabAccF32 merged(const SlabAccF32& o) const {
    return SlabAccF32{__builtin_elementwise_minimum(mn, o.mn), __builtin_elementwise_maximum(mx, o.mx)};
  }
  __device__ __forceinline__ SlabAccF32 shuffled(int m) const { return SlabAccF32{__shfl_xor(mn, m, kWave), __shfl_xor(mx, m, kWave)}; }
  __device__ __forceinline__ void decode(float& lo, float& hi) const {
    lo = mn;
    hi = mx;
  }
};
template <typename T16>
struct SlabAcc16 {
  Stat16 s;
  __device__ __forceinline__ static SlabAcc16 of(const RawPack<T16>& r) {
    SlabAcc16 a{kStat16Identity};
#pragma unroll
    for (int q = 0; q < 4; ++q) stat16_fold(a.s, r.d[0][q]);
    // both 16-bit halves into the low one (the high one keeps a copy: harmless for max / min)
    a.s = stat16_merge(a.s, Stat16{a.s.a >> 16, a.s.b >> 16, static_cast<uint32_t>(static_cast<int32_t>(a.s.c) >> 16)});
    return a;
  }
  __device__ __forceinline__ SlabAcc16 merged(const SlabAcc16& o) const { return SlabAcc16{stat16_merge(s, o.s)}; }
  __device__ __forceinline__ SlabAcc16 shuffled(int m) const {
    return SlabAcc16{Stat16{static_cast<uint32_t>(__shfl_xor(static_cast<int>(s.a), m, kWave)),
                            static_cast<uint32_t>(__shfl_xor(static_cast<int>(s.b), m, kWave)),
                            static_cast<uint32_t>(__shfl_xor(static_cast<int>(s.c), m, kWave))}};
  }
  __device__ __forceinline__ void decode(float& lo, float& hi) const { stat16_decode<T16>(s, lo, hi); }
};
template <typename Tin>
struct SlabAccSel { using type = SlabAcc16<Tin>; };
template <>
struct SlabAccSel<F32> { using type = SlabAccF32; };

// the transposing butterfly: acc[0 .. U) per lane -> one accumulator per lane (its slab: see above)
template <typename A, int U>
__device__ __forceinline__ A slab_butterfly(A (&acc)[U], uint32_t lane) {
  static_assert(U == 4 || U == 8, "slabs per wave");
  if constexpr (U == 8) {
    A h4[4];
    const bool up = lane & 32u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const A keep = up ? acc[4 + k] : acc[k], give = up ? acc[k] : acc[4 + k];
      h4[k] = keep.merged(give.shuffled(32));
    }
    A h2[2];
    const bool up2 = lane & 16u;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const A keep = up2 ? h4[2 + k] : h4[k], give = up2 ? h4[k] : h4[2 + k];
      h2[k] = keep.merged(give.shuffled(16));
    }
    const bool up3 = lane & 8u;
    const A keep = up3 ? h2[1] : h2[0], give = up3 ? h2[0] : h2[1];
    A r = keep.merged(give.shuffled(8));
    r = r.merged(r.shuffled(4));
    r = r.merged(r.shuffled(2));
    return r.merged(r.shuffled(1));
  } else {
    A h2[2];
    const bool up = lane & 32u;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const A keep = up ? acc[2 + k] : acc[k], give = up ? acc[k] : acc[2 + k];
      h2[k] = keep.merged(give.shuffled(32));
    }
    const bool up2 = lane & 16u;
    const A keep = up2 ? h2[1] : h2[0], give = up2 ? h2[0] : h2[1];
    A r = keep.merged(give.shuffled(16));
    r = r.merged(r.shuffled(8));
    r = r.merged(r.shuffled(4));
    r = r.merged(r.shuffled(2));
    return r.merged(r.shuffled(1));
  }
}

template <typename Tin, typename Tout, int U>
__global__ __launch_bounds__(kResBlock) void qdq_observe_kernel(
    const void* __restrict__ x, uint32_t n_slabs, uint32_t slabs_per_row, uint32_t symmetric, float qlo, float qhi,
    void* __restrict__ y, float* __restrict__ scale_out, float* __restrict__ zp_out,
    // ---- not preloaded ----
    float* __restrict__ min_out, float* __restrict__ max_out) {
  constexpr bool SPLIT = Tout::id == SBQ_F32;
  constexpr uint32_t kIn = Tin::id == SBQ_F32 ? 4 : 2, kOut = Tout::id == SBQ_F32 ? 4 : 2;
  constexpr uint32_t kSlabElems = kBlock * kPack;
  constexpr int kWaves = kResBlock / kWave;
  using Acc = typename SlabAccSel<Tin>::type;
  __shared__ Acc part[U][kWaves];
  const uint32_t sub = __builtin_amdgcn_readfirstlane(threadIdx.x / kBlock);
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const uint32_t tid = threadIdx.x % kBlock;
  const uint32_t lane = threadIdx.x & (kWave - 1);
  const uint32_t laneA = SPLIT ? 4 * tid : kPack * tid;
  const uint32_t laneB = laneA + kSlabElems / 2;
  const uint32_t sl0 = blockIdx.x * (kResSub * U) + sub;
  RawPack<Tin> raw[U];
  const uint32_t inA = laneA * kIn, inB = laneB * kIn;
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, n_slabs * (kSlabElems * kIn));
  const __amdgpu_buffer_rsrc_t ry = make_rsrc(y, n_slabs * (kSlabElems * kOut));
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint32_t sl = sl0 + u * kResSub;
    const uint32_t so = (sl < n_slabs ? sl : 0u) * (kSlabElems * kIn);
    if constexpr (SPLIT) {
      if constexpr (Tin::id == SBQ_F32) {
        raw[u].d[0] = bld16(rx, inA, so);
        raw[u].d[1] = bld16(rx, inB, so);
      } else {
        const u32x2 a = bld8(rx, inA, so), b = bld8(rx, inB, so);
        raw[u].d[0] = u32x4{a[0], a[1], b[0], b[1]};
      }
    } else {
      raw[u].d[0] = bld16(rx, inA, so);
    }
  }
  __builtin_amdgcn_sched_barrier(0);  // all loads in flight first
  // phase A: per slab, as it lands: the lane's accumulator; then ONE butterfly for all U slabs, one LDS record per
  // (slab, wave) from the lane group that ends up holding it
  Acc acc[U];
#pragma unroll
  for (int u = 0; u < U; ++u) acc[u] = Acc::of(raw[u]);
  {
    const Acc mine = slab_butterfly<Acc, U>(acc, lane);
    const uint32_t slab_of_lane = U == 8 ? ((lane >> 5 & 1u) * 4 + (lane >> 4 & 1u) * 2 + (lane >> 3 & 1u))
                                         : ((lane >> 5 & 1u) * 2 + (lane >> 4 & 1u));
    if ((lane & (U == 8 ? 7u : 15u)) == 0) part[slab_of_lane][wave] = mine;
  }
  __syncthreads();
  // phase B: lane u of every wave turns slab u's records into the row's scale / zero point (the reference's fp32
  // operations, qparams_from_minmax) -- U lanes work, once, instead of every lane U times -- and the results are
  // handed to the whole wave as scalars (v_readlane)
  const bool two = slabs_per_row == 2;  // row = both sub-blocks of this u; else each sub-block has its own row
  const float qrange = qhi - qlo;
  float my_sc = 1.0f, my_zp = 0.0f;
  if (lane < static_cast<uint32_t>(U)) {
    const int w0 = two ? 0 : static_cast<int>(sub) * (kWaves / 2);
    const int nw = two ? kWaves : kWaves / 2;
    Acc r = part[lane][w0];
    for (int w = 1; w < nw; ++w) r = r.merged(part[lane][w0 + w]);
    float mn, mx;
    r.decode(mn, mx);  // (NaN in the row: both NaN, like torch.min / max)
    qparams_from_minmax(mn, mx, qrange, symmetric != 0, my_sc, my_zp);
    const uint32_t sl = sl0 + lane * kResSub;
    // one wave per row publishes the observer's results
    if (sl < n_slabs && (wave & (kWaves / 2 - 1)) == 0 && (!two || sub == 0)) {
      const uint32_t row = two ? sl / 2 : sl;
      scale_out[row] = my_sc;
      zp_out[row] = my_zp;
      if (min_out) min_out[row] = mn;
      if (max_out) max_out[row] = mx;
    }
  }
  float sc[U], zp[U], yr[U];
  bool all_fast = true;
  const float my_yr = 1.0f / my_sc;  // (one division for the wave's U slabs: lane u holds slab u's scale)
#pragma unroll
  for (int u = 0; u < U; ++u) {
    sc[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_sc), u));
    zp[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_zp), u));
    yr[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_yr), u));
    all_fast &= fast_div_ok(sc[u]);
  }
  __builtin_amdgcn_sched_barrier(0);
  // phase C: convert and store slab after slab (every load has landed: stores now keep the memory system busy
  // while the next slab is converted).  A wave that meets a value the fast division does not cover (NaN, inf,
  // |x| >= s * 2^40, or a row whose scale is NaN) redoes its tile through the generic arithmetic afterwards:
  // the same bytes are written twice, the second time with the right values.
  bool odd = false;
  const uint32_t outA = laneA * kOut, outB = laneB * kOut;
  if (all_fast) {
    auto convert = [&](auto zp_tag) {
      constexpr bool ZP = decltype(zp_tag)::value;
      static_for<U>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        if constexpr (ZP) asm volatile("; observe: affine" : "+v"(raw[u].d[0]));
        else asm volatile("; observe: symmetric" : "+v"(raw[u].d[0]));
        float v[kPack], dq[kPack];
        unpack_raw<Tin>(raw[u], v);
        fast_pack<MASK_NONE, ZP>(v, u32x2{0, 0}, 0.0f, sc[u], yr[u], zp[u], qlo, qhi, dq, odd);
        OutPack<Tout> o;
        pack_out<Tout>(dq, o);
        const uint32_t sl = sl0 + u * kResSub;
        if (sl < n_slabs) {
          bst16(ry, outA, sl * (kSlabElems * kOut), o.d[0]);
          if constexpr (SPLIT) bst16(ry, outB, sl * (kSlabElems * kOut), o.d[Tout::id == SBQ_F32 ? 1 : 0]);
        }
      });
    };
    if (symmetric) convert(std::false_type{});
    else convert(std::true_type{});
    if (__builtin_amdgcn_ballot_w64(odd) == 0) return;
  }
  // cold path (a row whose scale the fast division does not cover -- NaN, or an element that is NaN / inf):
  // the tile again from memory, slab by slab, through the generic arithmetic
#pragma nounroll
  for (uint32_t u = 0; u < static_cast<uint32_t>(U); ++u) {
    const uint32_t sl = sl0 + u * kResSub;
    if (sl >= n_slabs) break;
    float s_ = 0.0f, z_ = 0.0f;
#pragma unroll
    for (int k = 0; k < U; ++k)
      if (static_cast<uint32_t>(k) == u) {
        s_ = sc[k];
        z_ = zp[k];
      }
    const int64_t e0 = static_cast<int64_t>(sl) * kSlabElems;
    RawPack<Tin> r;
    if constexpr (SPLIT) r = load_raw2<Tin, true>(x, e0 + laneA, e0 + laneB);
    else r = load_raw<Tin, true>(x, e0 + laneA);
    float v[kPack], lv[kPack], dq[kPack];
    unpack_raw<Tin>(r, v);
    quantize_pack<MASK_NONE, false, MATH_FAST>(v, u32x2{0, 0}, 0.0f, s_, z_, qlo, qhi, lv, dq);
    if constexpr (SPLIT) {
      store_half_f32<true>(y, e0 + laneA, dq);
      store_half_f32<true>(y, e0 + laneB, dq + 4);
    } else {
      store_pack<Tout, true>(y, e0 + laneA, dq);
    }
  }
}

// Resident schedule (see qdq_resident_kernel): chosen when the whole tensor is one sitting of the chip.
// knob 3: 0 auto, 1 never, 2 always (whatever the size: more workgroups than CUs).
template <typename Tin, typename Tout, int MASK>
bool try_resident(const ResidentCall& c, hipStream_t st) {
  const int mode = knob(3);
  if (mode == 1) return false;
  // whole slabs only, and the channel of a row is the row itself (outer == 1) or 0 (per tensor)
  if (c.packs_per_row % kBlock != 0) return false;
  if (c.C != 1 && c.rows != c.C) return false;
  if (c.n_slabs >= 32768u) return false;  // multiply-high channel index (see the kernel), 32-bit byte offsets
  const uint32_t cus = cu_count();
  const uint32_t cap16 = cus * kResSub * 16u, cap8 = cap16 / 2, cap4 = cap16 / 4;
  if (mode != 2) {
    // Where the schedule wins (tools/resident_sweep.py, rows x 4096): 16-bit input, from a quarter of a residency
    // of 16-slab waves up to a whole one.  fp32 input never: 128 data registers per wave leave no room, the
    // pipelined kernel is faster.
    if (Tin::id == SBQ_F32) return false;
    if (c.n_slabs <= cap4 / 2 || c.n_slabs > cap16) return false;
  }
  // slabs per wave: the smallest of 4 / 8 / 16 that holds the tensor in one sitting (more workgroups, shorter waves)
  const int U = c.n_slabs <= cap4 ? 4 : (c.n_slabs <= cap8 ? 8 : 16);
  const uint32_t n_tiles = (c.n_slabs + kResSub * U - 1) / (kResSub * U);
  // per tensor: one "row" of n_slabs slabs
  const uint32_t spr = c.C == 1 ? c.n_slabs : c.slabs_per_row;
  const uint32_t inv = static_cast<uint32_t>((1ull << 32) / (2ull * spr) + 1);
#define SBQ_RES(UV)                                                                                                  \
  qdq_resident_kernel<Tin, Tout, MASK, UV><<<n_tiles, kResBlock, 0, st>>>(                                           \
      c.x, c.n_slabs, inv, c.lsq, spr, c.qlo, c.qhi, c.y, c.scale, c.zp, c.mask, c.thresh)
  if (U == 16) SBQ_RES(16);
  else if (U == 8) SBQ_RES(8);
  else SBQ_RES(4);
#undef SBQ_RES
  return true;
}

template <typename Tin, typename Tout>
bool resident_mask(const ResidentCall& c, hipStream_t st) {
  if (c.mask) return try_resident<Tin, Tout, MASK_BYTES>(c, st);
  if (c.thresh) return try_resident<Tin, Tout, MASK_THRESH>(c, st);
  return try_resident<Tin, Tout, MASK_NONE>(c, st);
}

}  // namespace

template <typename Tin, typename Tout>
void launch_observe(const void* x, void* y, float* scale, float* zp, float* mn, float* mx, uint32_t n_slabs, uint32_t spr,
                    int symmetric, float qlo, float qhi, hipStream_t st) {
  const uint32_t cus = cu_count();
  const uint32_t cap16 = cus * kResSub * 16u;
  // 8 slabs per wave at most: measured on 4096x4096 bf16, 16 slabs per wave (one workgroup per CU) take 18.0 us
  // against 14.4 us -- the statistics and conversion phases of a lone workgroup leave the memory system idle
  const int U = n_slabs <= cap16 / 4 ? 4 : 8;
#define SBQ_OBS(UV)                                                                                               \
  qdq_observe_kernel<Tin, Tout, UV><<<(n_slabs + kResSub * UV - 1) / (kResSub * UV), kResBlock, 0, st>>>(          \
      x, n_slabs, spr, symmetric ? 1u : 0u, qlo, qhi, y, scale, zp, mn, mx)
  if (U == 8) SBQ_OBS(8);
  else SBQ_OBS(4);
#undef SBQ_OBS
}

bool qdq_try_resident(const ResidentCall& c, hipStream_t st) {
  if (c.x_dtype == SBQ_F32) return resident_mask<F32, F32>(c, st);
  if (c.x_dtype == SBQ_F16) return c.y_dtype == SBQ_F32 ? resident_mask<F16, F32>(c, st) : resident_mask<F16, F16>(c, st);
  return c.y_dtype == SBQ_F32 ? resident_mask<BF16, F32>(c, st) : resident_mask<BF16, BF16>(c, st);
}

}  // namespace sbq

extern "C" int sbq_observe_quant_perchannel_forward(const void* x, int x_dtype, void* y, int y_dtype, float* scale_out,
                                                    float* zero_point_out, float* min_out, float* max_out, int64_t C,
                                                    int64_t inner, int qmin, int qmax, int symmetric, void* workspace,
                                                    size_t workspace_bytes, void* stream) {
  using namespace sbq;
  if (!valid_dtype(x_dtype) || !valid_dtype(y_dtype)) return SBQ_ERR_DTYPE;
  if (y_dtype != SBQ_F32 && y_dtype != x_dtype) return SBQ_ERR_DTYPE;
  if (C < 0 || inner < 0) return SBQ_ERR_ARG;
  if (C == 0 || inner == 0) return SBQ_ERR_EMPTY;
  if (!x || !y || !scale_out || !zero_point_out || !min_out || !max_out) return SBQ_ERR_NULL;
  if (qmin >= qmax) return SBQ_ERR_ARG;
  hipStream_t st = as_stream(stream);
  const int64_t slab = kBlock * kPack;
  const bool fused = (inner == slab || inner == 2 * slab) && aligned16(x) && aligned16(y) &&
                     C * (inner / slab) <= (1ll << 18) && knob(3) != 1;
  if (fused) {
    const uint32_t spr = static_cast<uint32_t>(inner / slab), n_slabs = static_cast<uint32_t>(C) * spr;
    const float qlo = static_cast<float>(qmin), qhi = static_cast<float>(qmax);
#define SBQ_O(TI, TO) launch_observe<TI, TO>(x, y, scale_out, zero_point_out, min_out, max_out, n_slabs, spr, symmetric, qlo, qhi, st)
    if (x_dtype == SBQ_F32) SBQ_O(F32, F32);
    else if (x_dtype == SBQ_F16) { if (y_dtype == SBQ_F32) SBQ_O(F16, F32); else SBQ_O(F16, F16); }
    else { if (y_dtype == SBQ_F32) SBQ_O(BF16, F32); else SBQ_O(BF16, BF16); }
#undef SBQ_O
    return check_launch();
  }
  // any other geometry: the three steps one after the other (same results: the fused kernel is these, fused)
  int rc = sbq_channel_stats(x, x_dtype, 1, C, inner, min_out, max_out, nullptr, workspace, workspace_bytes, stream);
  if (rc != SBQ_OK) return rc;
  rc = sbq_qparams_from_minmax(min_out, max_out, C, qmin, qmax, symmetric, scale_out, zero_point_out, stream);
  if (rc != SBQ_OK) return rc;
  return sbq_quant_perchannel_forward(x, x_dtype, y, y_dtype, nullptr, SBQ_Q_NONE, scale_out, zero_point_out, 1, C,
                                      inner, qmin, qmax, SBQ_ROUND_HALF_EVEN, stream);
}

