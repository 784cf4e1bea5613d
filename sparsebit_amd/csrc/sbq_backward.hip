// sbq_backward.hip -- backward of the fake-quant straight-through estimator
// (STE / LSQ) for gfx950.
//
// Replaces QuantizePerTensorBackwardCUDA / QuantizePerChannelBackwardCUDA
// (sparsebit/quantization/torch_extensions/fake_quant_tensor.cu:97-132,227-270).
// Semantics are those of the per-tensor kernel K3 == the pure-torch restatement
// MySTE.backward (quant_tensor.py:45-71):
//     v   = round(x/s) + round(zp)
//     gx  = (qmin <= v <= qmax) ? gy : 0
//     gs += gy * (v < qmin ? qmin - zp : v > qmax ? qmax - zp : round(x/s) - x/s)
//     gzp += (qmin <= v <= qmax) ? 0 : -s * gy
// Deliberate differences from the reference CUDA code, which is not a usable
// oracle here: (1) its BlockReduceSum sits inside divergent grid-stride loops and
// assumes 32-lane warps (UB on any GPU, wrong on wave64); (2) the per-channel
// kernel tests `vq < qmax` for gzp (off by one vs the per-tensor `<=`,
// fake_quant_tensor.cu:264 vs :127) -- we use `<=` in both; (3) it accumulates
// with float atomicAdd onto one scalar (non-deterministic) -- here every chunk
// writes one fp64 partial and a second kernel folds them in a fixed order.
#include "sbq_common.hpp"
#include "sbq_qdq_math.hpp"

namespace sbq {
namespace {

constexpr uint32_t kBwdChunk = kBlock * kPack * 4;  // 8192 elements per workgroup

struct GradPartial {
  double gs, gzp;
};

template <typename T, typename Tg, bool VEC>
__global__ __launch_bounds__(kBlock) void ste_backward_kernel(
    const void* __restrict__ x, const void* __restrict__ gy, void* __restrict__ gx,
    const float* __restrict__ scale, const float* __restrict__ zero_point,
    GradPartial* __restrict__ part, const ChunkGeom g, float qlo, float qhi, int rounding, int lsq,
    // a channel that is ONE chunk (every [C, inner <= 8192] weight) writes its gradients itself: no fold launch
    float* __restrict__ gs_final, float* __restrict__ gzp_final, float gs_ratio, int final_) {
  __shared__ double s_d[kWavesPerBlock];
  const uint32_t bid = blockIdx.x;
  const ChunkPos cp = chunk_pos(g, bid);
  const float s_raw = scale[cp.c];
  float s = s_raw, zp = zero_point[cp.c];
  if (lsq) {  // raw LSQ parameters: s = |s|, zp = clamp(zp, qmin, qmax)  (lsq.py:61-62)
    s = __builtin_fabsf(s);
    zp = __builtin_amdgcn_fmed3f(zp, qlo, qhi);
  }
  zp = __builtin_rintf(zp);
  float gs = 0.0f, gz = 0.0f;

  // the chunk's scale is block-uniform: exact quotient by reciprocal + two fma refinements
  // (sbq_common.hpp: fast_div) whenever the scale and the element are in its range
  const bool fast_s = fast_div_ok(s) && rounding == SBQ_ROUND_HALF_EVEN;
  const float yr = fast_s ? 1.0f / s : 0.0f;
  const float bound = s * 0x1p40f;
  auto one_t = [&](float t, float gyv) -> float {  // t = x / s, correctly rounded
    float r;
    if (rounding == SBQ_ROUND_HALF_EVEN) r = __builtin_rintf(t);
    else if (rounding == SBQ_ROUND_HALF_UP) r = __builtin_floorf(t + 0.5f);
    else r = __builtin_ceilf(t - 0.5f);
    const float v = r + zp;
    const bool below = v < qlo, above = v > qhi;
    const bool inside = !(below || above);  // NaN counts as inside, like the reference's int compare of 0
    float pgs = (r - t) * gyv;
    if (above) pgs = (qhi - zp) * gyv;
    if (below) pgs = (qlo - zp) * gyv;
    gs += pgs;
    gz += inside ? 0.0f : (-s * gyv);
    return inside ? gyv : 0.0f;
  };
  auto one = [&](float xv, float gyv) -> float { return one_t(xv / s, gyv); };

  if constexpr (VEC) {
    const int64_t vend = cp.begin + ((cp.end - cp.begin) / kPack) * kPack;
    // fp32 tensors: a lane takes two 4-element runs half a 2048-element block apart, so that every 16-byte
    // access is part of a contiguous 1 KiB wave access (sbq_common.hpp: load_raw2); 16-bit tensors keep
    // their one 16-byte pack per lane
    constexpr bool SPLIT = T::id == SBQ_F32 || Tg::id == SBQ_F32;
    constexpr int64_t kBlk = static_cast<int64_t>(kBlock) * kPack;
    for (int64_t b0 = cp.begin; b0 < vend; b0 += kBlk) {
      int64_t eA, eB;
      bool okA, okB;
      if constexpr (SPLIT) {
        eA = b0 + 4 * threadIdx.x;
        eB = eA + kBlk / 2;
        okA = eA < vend;  // vend - begin is a multiple of 8: a started run is a whole run
        okB = eB < vend;
      } else {
        eA = b0 + static_cast<int64_t>(threadIdx.x) * kPack;
        eB = eA + 4;
        okA = okB = eA < vend;
      }
      if (!okA && !okB) continue;
      const int64_t cA = okA ? eA : vend - 4, cB = okB ? eB : vend - 4;
      float xv[kPack], gv[kPack], o[kPack];
      if constexpr (SPLIT) {
        load_pack2<T, true>(x, cp.row_base + cA, cp.row_base + cB, xv);
        load_pack2<T, true>(gy, cp.row_base + cA, cp.row_base + cB, gv);
      } else {
        load_pack<T, true>(x, cp.row_base + eA, xv);
        load_pack<T, true>(gy, cp.row_base + eA, gv);
      }
      if (!okA) {
#pragma unroll
        for (int q = 0; q < 4; ++q) gv[q] = 0.0f, xv[q] = 0.0f;  // a clamped run contributes nothing
      }
      if (!okB) {
#pragma unroll
        for (int q = 4; q < kPack; ++q) gv[q] = 0.0f, xv[q] = 0.0f;
      }
      // same wave-wide vote as the forward kernel: NaN / inf / huge inputs send the pack
      // through IEEE division, everything else through the exact fma refinement
      bool odd = false;
#pragma unroll
      for (int q = 0; q < kPack; ++q) odd |= !(__builtin_fabsf(xv[q]) < bound);
      if (fast_s && __builtin_amdgcn_ballot_w64(odd) == 0) {
#pragma unroll
        for (int q = 0; q < kPack; ++q) o[q] = one_t(fast_div(xv[q], s, yr), gv[q]);
      } else {
#pragma unroll
        for (int q = 0; q < kPack; ++q) o[q] = one(xv[q], gv[q]);
      }
      if constexpr (SPLIT) {
        if (okA) store_half<Tg, true>(gx, cp.row_base + eA, o);
        if (okB) store_half<Tg, true>(gx, cp.row_base + eB, o + 4);
      } else {
        store_pack<Tg, true>(gx, cp.row_base + eA, o);
      }
    }
    for (int64_t e = vend + threadIdx.x; e < cp.end; e += kBlock) {
      const float o = one(Elem<T>::load1(x, cp.row_base + e), Elem<T>::load1(gy, cp.row_base + e));
      Elem<Tg>::store1(gx, cp.row_base + e, o);
    }
  } else {
    for (int64_t e = cp.begin + threadIdx.x; e < cp.end; e += kBlock) {
      const float o = one(Elem<T>::load1(x, cp.row_base + e), Elem<T>::load1(gy, cp.row_base + e));
      Elem<Tg>::store1(gx, cp.row_base + e, o);
    }
  }
  if (part) {
    const double a = block_reduce(static_cast<double>(gs), Sum(), s_d);
    const double b = block_reduce(static_cast<double>(gz), Sum(), s_d);
    if (threadIdx.x == 0) {
      if (final_) {  // exactly what ste_fold_kernel does with a single partial
        float gsv = static_cast<float>(a);
        if (lsq) gsv = (gsv * gs_ratio) * (s_raw > 0.0f ? 1.0f : (s_raw < 0.0f ? -1.0f : 0.0f));
        if (gs_final) gs_final[cp.c] = gsv;
        if (gzp_final) gzp_final[cp.c] = static_cast<float>(b);
      } else {
        part[bid] = GradPartial{a, b};
      }
    }
  }
}

__global__ __launch_bounds__(kBlock) void ste_fold_kernel(const GradPartial* __restrict__ part,
                                                          uint32_t chunks_per_chan,
                                                          float* __restrict__ gs_out,
                                                          float* __restrict__ gzp_out,
                                                          const float* __restrict__ raw_scale, float gs_ratio) {
  __shared__ double s_d[kWavesPerBlock];
  const uint32_t c = blockIdx.x;
  const GradPartial* p = part + static_cast<size_t>(c) * chunks_per_chan;
  double a = 0.0, b = 0.0;
  constexpr int kBatch = 8;  // eight records per lane in flight; same summation order as one at a time
  for (uint32_t i0 = threadIdx.x; i0 < chunks_per_chan; i0 += kBlock * kBatch) {
    GradPartial r[kBatch];
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
      const uint32_t i = i0 + k * kBlock;
      r[k] = i < chunks_per_chan ? p[i] : GradPartial{0.0, 0.0};
    }
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
      a += r[k].gs;
      b += r[k].gzp;
    }
  }
  a = block_reduce(a, Sum(), s_d);
  b = block_reduce(b, Sum(), s_d);
  if (threadIdx.x == 0) {
    float gsv = static_cast<float>(a);
    if (raw_scale) {  // LSQ: the gs_scaling and abs() autograd nodes of lsq.py:13-21,61 folded in
      const float r = raw_scale[c];
      gsv = (gsv * gs_ratio) * (r > 0.0f ? 1.0f : (r < 0.0f ? -1.0f : 0.0f));
    }
    if (gs_out) gs_out[c] = gsv;
    if (gzp_out) gzp_out[c] = static_cast<float>(b);
  }
}

// ---- resident schedule for the 16-bit backward (the forward's: sbq_qdq_resident.hip) ---------------------------------
// x, gy and gx of the headline weight are 100 MB -- three streams.  The chunked kernel above mixes them for its whole
// duration (every workgroup: load x, load gy, compute, store gx, four times over): 20.3 us = 4.95 TB/s for
// 4096 x 4096 bf16.  Here a tensor that fits the chip's registers in one sitting is read in one burst and written in
// another: 512-thread workgroups (two sub-blocks of 256 lanes, one 2048-element slab wide), two per CU; a wave
// requests the x and gy packs of all its U = 8 slabs up front, overwrites the gy registers with gx as each pair lands,
// and issues its 8 stores only after its last conversion.  The per-slab gradient sums leave as fp64 partials
// (part[slab], folded per channel by ste_fold_kernel in slab order): lane sums over 8 elements in fp32, then fp64
// through a transposing butterfly -- at each of the first three exchange steps a lane keeps half of its slabs and
// hands the other half to its partner, so 8 slabs cost 7 + 3 exchanges instead of 8 x 6 -- and across the four waves
// of a sub-block through LDS, in wave order.  Deterministic; the summation tree differs from the chunked kernel's
// (gs agrees to fp32 rounding of the total, the tests' 1e-5).
constexpr int kBwdResBlock = 512, kBwdResSub = 2, kBwdResU = 8;
constexpr uint32_t kBwdSlab = kBlock * kPack;  // 2048 elements

__device__ __forceinline__ __amdgpu_buffer_rsrc_t bwd_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

template <typename T, bool WANT_GZP>
__global__ __launch_bounds__(kBwdResBlock, 2) void ste_backward_resident_kernel(
    const void* __restrict__ x, const void* __restrict__ gy, void* __restrict__ gx, uint32_t n_slabs, uint32_t row_inv,
    uint32_t lsq, float qlo, float qhi, const float* __restrict__ scale, const float* __restrict__ zero_point,
    GradPartial* __restrict__ part) {
  constexpr int U = kBwdResU;
  __shared__ double s_red[kBwdResSub][U][kWavesPerBlock][2];
  const uint32_t sub = __builtin_amdgcn_readfirstlane(threadIdx.x / kBlock);
  const uint32_t tid = threadIdx.x % kBlock;
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t wsub = __builtin_amdgcn_readfirstlane(tid / kWave);  // wave inside the sub-block
  const uint32_t voff = tid * (kPack * 2u);                          // the lane's 16-byte pack inside a slab
  const uint32_t sl0 = blockIdx.x * (kBwdResSub * U) + sub;          // slab(u) = sl0 + u * kBwdResSub
  const uint32_t bytes = n_slabs * (kBwdSlab * 2u);
  const __amdgpu_buffer_rsrc_t rx = bwd_rsrc(x, bytes), rg = bwd_rsrc(gy, bytes), ro = bwd_rsrc(gx, bytes);
  RawPack<T> xr[U], gr[U];
  uint32_t slc[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint32_t sl = sl0 + u * kBwdResSub;
    slc[u] = sl < n_slabs ? sl : 0u;  // slabs past the end read slab 0 (valid memory) and store nothing
    const uint32_t so = slc[u] * (kBwdSlab * 2u);
    xr[u].d[0] = __builtin_amdgcn_raw_buffer_load_b128(rx, voff, so, 2);
    gr[u].d[0] = __builtin_amdgcn_raw_buffer_load_b128(rg, voff, so, 2);
  }
  __builtin_amdgcn_sched_barrier(0);  // all data loads are in flight before any other work
  float zp[U], sraw[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint32_t c = __builtin_amdgcn_readfirstlane(__umulhi(slc[u] * 2u, row_inv));  // channel of the slab
    sraw[u] = uniform_load(scale, c);
    zp[u] = uniform_load(zero_point, c);
  }
  __builtin_amdgcn_sched_barrier(0);
  float gs[U], gz[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    float s = sraw[u], z = zp[u];
    if (lsq) {  // raw LSQ parameters (lsq.py:61-62)
      s = __builtin_fabsf(s);
      z = __builtin_amdgcn_fmed3f(z, qlo, qhi);
    }
    z = __builtin_rintf(z);
    asm volatile("; slab pair" : "+v"(xr[u].d[0]), "+v"(gr[u].d[0]));  // (this slab's vmcnt wait sits here)
    float xv[kPack], gv[kPack], o[kPack];
    unpack_raw<T>(xr[u], xv);
    unpack_raw<T>(gr[u], gv);
    const bool fast_s = fast_div_ok(s);
    const float yr = fast_s ? 1.0f / s : 0.0f;
    const float bound = s * 0x1p40f;
    bool odd = false;
#pragma unroll
    for (int q = 0; q < kPack; ++q) odd |= !(__builtin_fabsf(xv[q]) < bound);
    const bool fast = fast_s && __builtin_amdgcn_ballot_w64(odd) == 0;  // the chunked kernel's wave-wide vote
    float a = 0.0f, b = 0.0f;
    auto elem = [&](float t, int q) {  // t = x / s, correctly rounded
      const float r = __builtin_rintf(t);
      const float v = r + z;
      const bool below = v < qlo, above = v > qhi;
      const bool inside = !(below || above);  // NaN counts as inside, like the reference's int compare of 0
      float pgs = (r - t) * gv[q];
      if (above) pgs = (qhi - z) * gv[q];
      if (below) pgs = (qlo - z) * gv[q];
      a += pgs;
      if constexpr (WANT_GZP) b += inside ? 0.0f : (-s * gv[q]);
      o[q] = inside ? gv[q] : 0.0f;
    };
    if (fast) {  // wave-uniform
#pragma unroll
      for (int q = 0; q < kPack; ++q) elem(fast_div(xv[q], s, yr), q);
    } else {
#pragma unroll
      for (int q = 0; q < kPack; ++q) elem(xv[q] / s, q);
    }
    gs[u] = a;
    gz[u] = b;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if constexpr (T::id == SBQ_BF16) {
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        gr[u].d[0][q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{o[2 * q], o[2 * q + 1]}, bf16x2));
      } else {
        typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
        gr[u].d[0][q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{o[2 * q], o[2 * q + 1]}, f16x2));
      }
    }
    asm volatile("" : "+v"(gr[u].d[0]));  // the packed result exists HERE (not in the store phase)
  }
  __builtin_amdgcn_sched_barrier(0);  // the first store is issued after the last conversion
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint32_t sl = sl0 + u * kBwdResSub;
    if (sl < n_slabs) {  // workgroup-uniform
      __builtin_amdgcn_raw_buffer_store_b128(gr[u].d[0], ro, voff, sl * (kBwdSlab * 2u), 2);
      asm volatile("s_nop 1");  // (see bst16 in sbq_qdq_resident.hip: the store reads its data a cycle after issue)
    }
  }
  if (!part) return;
  // 8 slabs' sums through the wave at once: after the three halving steps lane L holds slab (L >> 3) & 7
  auto fold8 = [&](const float (&v)[U]) -> double {
    double d[U];
#pragma unroll
    for (int u = 0; u < U; ++u) d[u] = static_cast<double>(v[u]);
    double k4[4], k2[2];
    {
      const bool hi = (lane & 32) != 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const double keep = hi ? d[i + 4] : d[i], send = hi ? d[i] : d[i + 4];
        k4[i] = keep + __shfl_xor(send, 32, kWave);
      }
    }
    {
      const bool hi = (lane & 16) != 0;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const double keep = hi ? k4[i + 2] : k4[i], send = hi ? k4[i] : k4[i + 2];
        k2[i] = keep + __shfl_xor(send, 16, kWave);
      }
    }
    const bool hi = (lane & 8) != 0;
    double r = (hi ? k2[1] : k2[0]) + __shfl_xor(hi ? k2[0] : k2[1], 8, kWave);
    r += __shfl_xor(r, 4, kWave);
    r += __shfl_xor(r, 2, kWave);
    r += __shfl_xor(r, 1, kWave);
    return r;
  };
  const double ra = fold8(gs);
  double rb = 0.0;
  if constexpr (WANT_GZP) rb = fold8(gz);
  if ((lane & 7) == 0) {
    const int u = (lane >> 3) & 7;
    s_red[sub][u][wsub][0] = ra;
    s_red[sub][u][wsub][1] = rb;
  }
  __syncthreads();
  if (tid < static_cast<uint32_t>(U)) {
    const uint32_t sl = sl0 + tid * kBwdResSub;
    if (sl < n_slabs) {
      double a = s_red[sub][tid][0][0], b = s_red[sub][tid][0][1];
#pragma unroll
      for (int w = 1; w < kWavesPerBlock; ++w) {
        a += s_red[sub][tid][w][0];
        b += s_red[sub][tid][w][1];
      }
      part[sl] = GradPartial{a, b};
    }
  }
}

// -> true when the resident kernel took the call (whole 2048-element slabs, rows == channels or per tensor, a tensor
// between a quarter of and one whole residency of two 512-thread workgroups per CU; knob 3 == 1: never)
template <typename T>
bool ste_backward_try_resident(const void* x, const void* gy, void* gx, float* gs, float* gzp, const float* scale,
                               const float* zp, int64_t outer, int64_t C, int64_t inner, float qlo, float qhi, int lsq,
                               float gs_ratio, void* workspace, size_t workspace_bytes, hipStream_t st, int* rc_out) {
  if (knob(3) == 1) return false;
  if (inner % kBwdSlab != 0 || (C != 1 && outer != 1)) return false;
  const int64_t n_slabs64 = outer * C * inner / kBwdSlab;
  const int64_t cap = static_cast<int64_t>(cu_count()) * 2 * kBwdResSub * kBwdResU;
  if (n_slabs64 > cap || n_slabs64 * 4 < cap || n_slabs64 >= 32768) return false;
  if (!aligned16(x) || !aligned16(gy) || !aligned16(gx)) return false;
  const uint32_t n_slabs = static_cast<uint32_t>(n_slabs64);
  const bool want = gs || gzp;
  if (want && (workspace_bytes < static_cast<size_t>(n_slabs) * sizeof(GradPartial) || !workspace || !aligned16(workspace)))
    return false;  // (a workspace sized for the chunked kernel: take that one)
  const uint32_t spr = C == 1 ? n_slabs : static_cast<uint32_t>(inner / kBwdSlab);
  const uint32_t inv = static_cast<uint32_t>((1ull << 32) / (2ull * spr) + 1);
  const uint32_t n_tiles = (n_slabs + kBwdResSub * kBwdResU - 1) / (kBwdResSub * kBwdResU);
  GradPartial* part = want ? static_cast<GradPartial*>(workspace) : nullptr;
  if (gzp)
    ste_backward_resident_kernel<T, true><<<n_tiles, kBwdResBlock, 0, st>>>(x, gy, gx, n_slabs, inv, lsq, qlo, qhi, scale, zp, part);
  else
    ste_backward_resident_kernel<T, false><<<n_tiles, kBwdResBlock, 0, st>>>(x, gy, gx, n_slabs, inv, lsq, qlo, qhi, scale, zp, part);
  *rc_out = check_launch();
  if (*rc_out != SBQ_OK || !want) return true;
  ste_fold_kernel<<<static_cast<uint32_t>(C), kBlock, 0, st>>>(part, spr, gs, gzp, lsq ? scale : nullptr, gs_ratio);
  *rc_out = check_launch();
  return true;
}

int ste_backward(const void* x, const void* gy, int x_dtype, void* gx, int gx_dtype, float* gs,
                 float* gzp, const float* scale, const float* zp, int64_t outer, int64_t C,
                 int64_t inner, int qmin, int qmax, int rounding, void* workspace,
                 size_t workspace_bytes, void* stream, int lsq = 0, float gs_ratio = 1.0f) {
  if (!valid_dtype(x_dtype) || !valid_dtype(gx_dtype)) return SBQ_ERR_DTYPE;
  if (gx_dtype != SBQ_F32 && gx_dtype != x_dtype) return SBQ_ERR_DTYPE;
  if (outer < 0 || C < 0 || inner < 0) return SBQ_ERR_ARG;
  if (outer == 0 || C == 0 || inner == 0) return SBQ_ERR_EMPTY;
  if (!x || !gy || !gx || !scale || !zp) return SBQ_ERR_NULL;
  if (qmin > qmax || rounding < 0 || rounding > 2) return SBQ_ERR_ARG;
  if (!geom_ok(outer, C, inner, kBwdChunk)) return SBQ_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(x) % dtype_size(x_dtype)) ||
      (reinterpret_cast<uintptr_t>(gy) % dtype_size(x_dtype)) ||
      (reinterpret_cast<uintptr_t>(gx) % dtype_size(gx_dtype)))
    return SBQ_ERR_ALIGN;
  const ChunkGeom g = make_geom(outer, C, inner, kBwdChunk);
  const bool want_param_grads = gs || gzp;
  GradPartial* part = nullptr;
  if (want_param_grads) {
    const size_t need = static_cast<size_t>(g.chunks_per_chan) * g.C * sizeof(GradPartial);
    if (!workspace) return SBQ_ERR_NULL;
    if (workspace_bytes < need || !aligned16(workspace)) return SBQ_ERR_WORKSPACE;
    part = static_cast<GradPartial*>(workspace);
  }
  hipStream_t st = as_stream(stream);
  if (x_dtype != SBQ_F32 && gx_dtype == x_dtype && rounding == SBQ_ROUND_HALF_EVEN) {
    int rc_res = SBQ_OK;
    const float qlo_r = static_cast<float>(qmin), qhi_r = static_cast<float>(qmax);
    const bool took = x_dtype == SBQ_BF16
                          ? ste_backward_try_resident<BF16>(x, gy, gx, gs, gzp, scale, zp, outer, C, inner, qlo_r, qhi_r, lsq,
                                                            gs_ratio, workspace, workspace_bytes, st, &rc_res)
                          : ste_backward_try_resident<F16>(x, gy, gx, gs, gzp, scale, zp, outer, C, inner, qlo_r, qhi_r, lsq,
                                                           gs_ratio, workspace, workspace_bytes, st, &rc_res);
    if (took) return rc_res;
  }
  const uint32_t grid = g.chunks_per_chan * g.C;
  const bool final_ = want_param_grads && g.chunks_per_chan == 1;
  const bool vec = pack_friendly(x, C, outer, inner) && aligned16(gy) && aligned16(gx);
  const float qlo = static_cast<float>(qmin), qhi = static_cast<float>(qmax);
  int rc = dispatch_dtype(x_dtype, [&](auto tag) {
    using T = decltype(tag);
    auto go = [&](auto gtag) {
      using Tg = decltype(gtag);
      if (vec)
        ste_backward_kernel<T, Tg, true><<<grid, kBlock, 0, st>>>(x, gy, gx, scale, zp, part, g, qlo, qhi, rounding, lsq,
                                                               gs, gzp, gs_ratio, final_ ? 1 : 0);
      else
        ste_backward_kernel<T, Tg, false><<<grid, kBlock, 0, st>>>(x, gy, gx, scale, zp, part, g, qlo, qhi, rounding, lsq,
                                                                gs, gzp, gs_ratio, final_ ? 1 : 0);
    };
    if (gx_dtype == SBQ_F32) go(F32());
    else go(T());
  });
  if (rc != SBQ_OK) return rc;
  rc = check_launch();
  if (rc != SBQ_OK || !want_param_grads || final_) return rc;
  ste_fold_kernel<<<g.C, kBlock, 0, st>>>(part, g.chunks_per_chan, gs, gzp, lsq ? scale : nullptr, gs_ratio);
  return check_launch();
}

}  // namespace
}  // namespace sbq

extern "C" {

size_t sbq_backward_workspace_bytes(int64_t outer, int64_t C, int64_t inner) {
  using namespace sbq;
  if (!geom_ok(outer, C, inner, kBwdChunk)) return 0;
  const ChunkGeom g = make_geom(outer, C, inner, kBwdChunk);
  const size_t chunked = static_cast<size_t>(g.chunks_per_chan) * g.C * sizeof(GradPartial);
  // (the resident schedule keeps one partial per 2048-element slab)
  const size_t resident = static_cast<size_t>(outer * C * inner / kBwdSlab + 1) * sizeof(GradPartial);
  return chunked > resident ? chunked : resident;
}

int sbq_quant_pertensor_backward(const void* x, const void* gy, int x_dtype, void* gx, int gx_dtype,
                                 float* gs, float* gzp, const float* scale, const float* zero_point,
                                 int64_t numel, int qmin, int qmax, int rounding, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  return sbq::ste_backward(x, gy, x_dtype, gx, gx_dtype, gs, gzp, scale, zero_point, 1, 1, numel,
                           qmin, qmax, rounding, workspace, workspace_bytes, stream);
}

int sbq_quant_perchannel_backward(const void* x, const void* gy, int x_dtype, void* gx, int gx_dtype,
                                  float* gs, float* gzp, const float* scale, const float* zero_point,
                                  int64_t outer, int64_t C, int64_t inner, int qmin, int qmax,
                                  int rounding, void* workspace, size_t workspace_bytes, void* stream) {
  return sbq::ste_backward(x, gy, x_dtype, gx, gx_dtype, gs, gzp, scale, zero_point, outer, C, inner,
                           qmin, qmax, rounding, workspace, workspace_bytes, stream);
}

int sbq_quant_lsq_backward(const void* x, const void* gy, int x_dtype, void* gx, int gx_dtype, float* gs,
                           const float* scale, const float* zero_point, int64_t outer, int64_t C, int64_t inner,
                           int qmin, int qmax, float gs_ratio, void* workspace, size_t workspace_bytes,
                           void* stream) {
  return sbq::ste_backward(x, gy, x_dtype, gx, gx_dtype, gs, nullptr, scale, zero_point, outer, C, inner, qmin, qmax,
                           SBQ_ROUND_HALF_EVEN, workspace, workspace_bytes, stream, 1, gs_ratio);
}

}  // extern "C"
