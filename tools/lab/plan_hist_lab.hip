// lab (round 6): what does the plan's LDS histogram of the sample cost, and do replicas help?
// 1024 threads x 16 fp32 keys (|N(0,1)|), bins = key >> 19 (8192 bins), like plan_compute of sbq_select_win.hip.
//   hipcc --offload-arch=gfx950 -O3 -o tools/lab/plan_hist_lab tools/lab/plan_hist_lab.hip && tools/lab/plan_hist_lab
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>
constexpr int kBins = 8192, kT = 1024, kKeys = 16;
__device__ __forceinline__ uint32_t key_abs(uint32_t b) { return (b & 0x7fffffffu) + 0x80000000u; }
template <int REP, int MODE>
__global__ __launch_bounds__(kT) void k(const uint32_t* x, unsigned long long* cyc, uint32_t* out) {
  extern __shared__ uint32_t hist[];
  constexpr int kStride = kBins + (REP > 1 ? 8 : 0);  // (replica r starts 8 r banks further)
  for (int i = threadIdx.x; i < REP * kStride; i += kT) hist[i] = 0;
  uint32_t v[kKeys];
  const uint4* p = reinterpret_cast<const uint4*>(x + (static_cast<size_t>(blockIdx.x) * kT + threadIdx.x) * kKeys);
#pragma unroll
  for (int i = 0; i < kKeys / 4; ++i) {
    const uint4 q = p[i];
    v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w;
  }
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  uint32_t* h = hist + (threadIdx.x & (REP - 1)) * kStride;
  if constexpr (MODE == 0) {
#pragma unroll
    for (int j = 0; j < kKeys; ++j) atomicAdd(&h[key_abs(v[j]) >> 19], 1u);
  } else if constexpr (MODE == 1) {
    // keys of one instruction differ per lane group: lane l takes its keys in the order (j + l) % 16 -- no change in
    // conflicts expected (independent draws); control
#pragma unroll
    for (int j = 0; j < kKeys; ++j) atomicAdd(&h[key_abs(v[(j + 5) & 15]) >> 19], 1u);
  } else if constexpr (MODE == 2) {
    // 16-bit counters, two bins per dword?  no: same-dword conflicts double.  Instead: coarse bins (>> 21: 2048 bins)
#pragma unroll
    for (int j = 0; j < kKeys; ++j) atomicAdd(&h[key_abs(v[j]) >> 21], 1u);
  } else if constexpr (MODE == 3) {
    // returning atomics (ds_add_rtn): for comparison
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < kKeys; ++j) s += atomicAdd(&h[key_abs(v[j]) >> 19], 1u);
    if (s == 0xffffffffu) out[0] = s;
  } else if constexpr (MODE == 4) {
    // plain stores (no atomic): the LDS cost floor of 16 scattered accesses
#pragma unroll
    for (int j = 0; j < kKeys; ++j) h[key_abs(v[j]) >> 19] = 1u;
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  uint32_t s = 0;
  for (int i = threadIdx.x; i < REP * kStride; i += kT) s += hist[i];
  if (s == 0xdeadbeefu) out[1] = s;
}
template <int REP, int MODE>
void run(const char* name, const uint32_t* x, unsigned long long* cyc, uint32_t* out, int grid) {
  const size_t lds = sizeof(uint32_t) * REP * (kBins + 8);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<REP, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
  std::vector<unsigned long long> h(grid);
  for (int rep = 0; rep < 3; ++rep) {
    k<REP, MODE><<<grid, kT, lds>>>(x, cyc, out);
    hipDeviceSynchronize();
  }
  hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  printf("%-44s replicas %d: cycles median %llu  min %llu  max %llu\n", name, REP, h[grid / 2], h[0], h[grid - 1]);
}
int main() {
  const int grid = 236;
  std::vector<float> hx(static_cast<size_t>(grid) * kT * kKeys);
  std::mt19937 g(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  for (auto& f : hx) f = nd(g);
  uint32_t *x, *out;
  unsigned long long* cyc;
  hipMalloc(&x, hx.size() * 4);
  hipMalloc(&cyc, grid * 8);
  hipMalloc(&out, 64);
  hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  run<1, 0>("atomic add, 8192 bins", x, cyc, out, grid);
  run<2, 0>("atomic add, 8192 bins", x, cyc, out, grid);
  run<4, 0>("atomic add, 8192 bins", x, cyc, out, grid);
  run<1, 1>("atomic add, rotated key order (control)", x, cyc, out, grid);
  run<1, 2>("atomic add, 2048 bins (>> 21)", x, cyc, out, grid);
  run<1, 3>("returning atomic add", x, cyc, out, grid);
  run<1, 4>("plain store (floor)", x, cyc, out, grid);
  run<4, 4>("plain store (floor)", x, cyc, out, grid);
  return 0;
}
