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
  return static_cast<size_t>(g.chunks_per_chan) * g.C * sizeof(GradPartial);
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
