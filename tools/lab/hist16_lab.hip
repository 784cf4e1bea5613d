// Dev tool (round 4): can a whole-tensor selection of a 16-bit tensor be made INDEPENDENT of the sample-derived plan?
// Every workgroup (1024 threads, one per CU) builds the FULL histogram of its 65 536 elements' 16-bit keys in LDS
// (65 536 bins x 16-bit counts packed two per dword = 128 KB of the 160 KB) while its slabs arrive; what the plan
// contributes afterwards is a window, and "count below / histogram inside" become LDS reads instead of a sweep that
// can only start once the plan is known (csrc/sbq_select_win.hip: plan done at 9.4 us, sweep done at 13-15 us).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab/hist16_lab.hip -o tools/lab/hist16_lab
// Variants: 0 = read only (the arrival floor), 1 = packed u16 histogram + below-count extraction, 2 = the same with
// +-0 counted in registers (ReLU data: half of the elements would hit one LDS word), 3 = |x| keys (15 bits) in 32 K
// u32 bins, 4 = variant 1 without the extraction (histogram alone).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int kT = 1024, kSlab = kT * 8 * 2;  // 16 Ki elements per slab, 4 slabs per workgroup
constexpr int kBufs = 9;
constexpr uint32_t kRot = (~0xff80u) & 0xffffu;  // bf16 Key16 rotation
constexpr uint32_t kZeroKey = (0x7fffu - kRot) & 0xffffu;  // key16(-0); key16(+0) = kZeroKey + 1

__device__ __forceinline__ uint32_t pack2(uint32_t w, uint32_t amask2) {
  typedef int16_t i16x2 __attribute__((ext_vector_type(2)));
  typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
  w &= amask2;
  const uint32_t m = __builtin_bit_cast(uint32_t, __builtin_bit_cast(i16x2, w) >> static_cast<int16_t>(15));
  const uint32_t t = w ^ (m | 0x80008000u);
  const u16x2 rot = {static_cast<uint16_t>(kRot), static_cast<uint16_t>(kRot)};
  return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, t) - rot);
}
__host__ uint32_t key16_host(uint16_t b, bool use_abs) {
  uint32_t w = use_abs ? (b & 0x7fffu) : b;
  const uint32_t m = (w & 0x8000u) ? 0xffffu : 0u;
  const uint32_t t = (w ^ (m | 0x8000u)) & 0xffffu;
  return (t - kRot) & 0xffffu;
}
__device__ __forceinline__ void lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int V>
__global__ __launch_bounds__(kT) void hist_k(const uint16_t* __restrict__ x, uint32_t n_slabs, uint32_t lo_win,
                                             unsigned long long* __restrict__ out, unsigned long long* __restrict__ stamps) {
  extern __shared__ uint32_t lds[];  // 32 Ki dwords
  __shared__ unsigned long long red[kT / 64];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  // all four slabs requested up front: 8 x dwordx4 per thread
  u32x4 raw[4][2];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t g = blockIdx.x + j * gridDim.x;
    const uint16_t* base = x + static_cast<size_t>(g < n_slabs ? g : 0) * kSlab;
#pragma unroll
    for (int u = 0; u < 2; ++u)
      raw[j][u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + (u * kT + threadIdx.x) * 8));
  }
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (V != 0) {
    u32x4* l4 = reinterpret_cast<u32x4*>(lds);
#pragma unroll
    for (int i = 0; i < 8; ++i) l4[i * kT + threadIdx.x] = u32x4{0, 0, 0, 0};
    lds_sync();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  uint32_t acc = 0, zeros = 0;
  const uint32_t amask2 = V == 3 ? 0x7fff7fffu : 0xffffffffu;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t w = raw[j][u][q];
        if constexpr (V == 0) {
          acc ^= w;
        } else {
          const uint32_t k2 = pack2(w, amask2);
          const uint32_t ka = k2 & 0xffffu, kb = k2 >> 16;
          if constexpr (V == 3) {
            // |x|: 15-bit keys (the sign transform leaves bit 15 set: drop it), one u32 bin per key
            atomicAdd(&lds[ka & 0x7fffu], 1u);
            atomicAdd(&lds[kb & 0x7fffu], 1u);
          } else if constexpr (V == 2) {
            // +-0 (keys kZeroKey, kZeroKey + 1) stay in a register
            const bool za = (ka - kZeroKey) <= 1u, zb = (kb - kZeroKey) <= 1u;
            zeros += za;
            zeros += zb;
            if (!za) atomicAdd(&lds[ka >> 1], (ka & 1u) ? 0x10000u : 1u);
            if (!zb) atomicAdd(&lds[kb >> 1], (kb & 1u) ? 0x10000u : 1u);
          } else {
            atomicAdd(&lds[ka >> 1], (ka & 1u) ? 0x10000u : 1u);
            atomicAdd(&lds[kb >> 1], (kb & 1u) ? 0x10000u : 1u);
          }
        }
      }
    }
  }
  if constexpr (V != 0) lds_sync();
  const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
  unsigned long long below = acc;
  if constexpr (V == 1 || V == 2) {
    // the keys below lo_win: thread t owns dwords [32 t, 32 t + 32) = keys [64 t, 64 t + 64)
    const u32x4* l4 = reinterpret_cast<const u32x4*>(lds);
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const u32x4 v = l4[threadIdx.x * 8 + i];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t key0 = (threadIdx.x * 32 + i * 4 + q) * 2;
        c += key0 < lo_win ? (v[q] & 0xffffu) : 0u;
        c += key0 + 1 < lo_win ? (v[q] >> 16) : 0u;
      }
    }
    if constexpr (V == 2) c += kZeroKey < lo_win ? zeros : 0u;  // (both zeros on one side of the window: a lab)
    below = c;
  } else if constexpr (V == 3) {
    const u32x4* l4 = reinterpret_cast<const u32x4*>(lds);
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const u32x4 v = l4[threadIdx.x * 8 + i];
#pragma unroll
      for (int q = 0; q < 4; ++q) c += (threadIdx.x * 32 + i * 4 + q) < (lo_win & 0x7fffu) ? v[q] : 0u;
    }
    below = c;
  } else if constexpr (V == 4) {
    below = lds[threadIdx.x];
  }
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) below += __shfl_xor(below, m, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = below;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long s = 0;
    for (int w = 0; w < kT / 64; ++w) s += red[w];
    out[blockIdx.x] = s;
    const unsigned long long t3 = __builtin_amdgcn_s_memrealtime();
    if (stamps) {
      stamps[blockIdx.x * 4 + 0] = t0;
      stamps[blockIdx.x * 4 + 1] = t1;
      stamps[blockIdx.x * 4 + 2] = t2;
      stamps[blockIdx.x * 4 + 3] = t3;
    }
  }
}

static uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return static_cast<uint16_t>(u >> 16);
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int V>
static void run(const char* name, uint16_t* const* bufs, uint32_t n_slabs, uint32_t lo_win, unsigned long long want,
                unsigned long long* d_out, unsigned long long* d_st) {
  const uint32_t grid = n_slabs / 4;
  const size_t shmem = V == 0 ? 0 : 131072;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(hist_k<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  for (int i = 0; i < 20; ++i) hist_k<V><<<grid, kT, shmem>>>(bufs[i % kBufs], n_slabs, lo_win, d_out, nullptr);
  CK(hipDeviceSynchronize());
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(a));
    for (int i = 0; i < 200; ++i) hist_k<V><<<grid, kT, shmem>>>(bufs[i % kBufs], n_slabs, lo_win, d_out, nullptr);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    best = std::min(best, ms * 1000.0f / 200);
  }
  // one stamped launch on buffer 0
  hist_k<V><<<grid, kT, shmem>>>(bufs[0], n_slabs, lo_win, d_out, d_st);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> out(grid), st(grid * 4);
  CK(hipMemcpy(out.data(), d_out, grid * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(st.data(), d_st, grid * 32, hipMemcpyDeviceToHost));
  unsigned long long sum = 0, first = ~0ull;
  for (uint32_t g = 0; g < grid; ++g) { sum += out[g]; first = std::min(first, st[g * 4]); }
  double med[4];
  for (int k = 0; k < 4; ++k) {
    std::vector<double> v(grid);
    for (uint32_t g = 0; g < grid; ++g) v[g] = (st[g * 4 + k] - first) / 100.0;  // 100 MHz -> us
    std::sort(v.begin(), v.end());
    med[k] = v[grid / 2];
    if (k == 3) printf("    (last workgroup's end %.2f us)\n", v[grid - 1]);
  }
  printf("%-28s %7.2f us/launch  %6.2f TB/s   stamps(us, median over wgs): start %.2f  loads+clear %.2f  hist %.2f  end %.2f   %s\n",
         name, best, n_slabs * (double)kSlab * 2 / best / 1e6, med[0], med[1], med[2], med[3],
         V == 0 || V == 4 ? "" : (sum == want ? "count OK" : "COUNT MISMATCH"));
  if (!(V == 0 || V == 4) && sum != want) printf("    got %llu want %llu\n", sum, want);
}

int main(int argc, char** argv) {
  const uint32_t rows = 4096, cols = 4096;
  const size_t n = static_cast<size_t>(rows) * cols;
  const uint32_t n_slabs = n / kSlab;  // 1024
  for (int data = 0; data < 2; ++data) {
    std::vector<uint16_t> h(n);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (s >> 11) * (1.0 / 9007199254740992.0); };
    for (uint32_t r = 0; r < rows; ++r) {
      const float sc = powf(10.0f, -2.0f + 3.0f * r / (rows - 1));
      for (uint32_t c = 0; c < cols; c += 2) {
        const double u1 = rnd() + 1e-12, u2 = rnd();
        const double m = sqrt(-2.0 * log(u1));
        float a = static_cast<float>(m * cos(6.283185307179586 * u2)) * sc, b = static_cast<float>(m * sin(6.283185307179586 * u2)) * sc;
        if (data == 1) { a = a > 0 ? a : 0.0f; b = b > 0 ? b : 0.0f; }  // ReLU: half zeros
        h[static_cast<size_t>(r) * cols + c] = f2bf(a);
        h[static_cast<size_t>(r) * cols + c + 1] = f2bf(b);
      }
    }
    // window start: the key of the median-ish value
    const uint32_t lo_win = key16_host(f2bf(0.05f), false), lo_abs = key16_host(f2bf(0.05f), true);
    unsigned long long want = 0, want_abs = 0;
    for (size_t i = 0; i < n; ++i) {
      want += key16_host(h[i], false) < lo_win;
      want_abs += (key16_host(h[i], true) & 0x7fffu) < (lo_abs & 0x7fffu);
    }
    uint16_t* bufs[kBufs];
    for (int i = 0; i < kBufs; ++i) {
      CK(hipMalloc(&bufs[i], n * 2));
      CK(hipMemcpy(bufs[i], h.data(), n * 2, hipMemcpyHostToDevice));
    }
    unsigned long long *d_out, *d_st;
    CK(hipMalloc(&d_out, 1024 * 8));
    CK(hipMalloc(&d_st, 1024 * 32));
    printf("---- data %d (%s), 16.7 M bf16, %d rotating buffers, 256 workgroups x 1024 threads ----\n", data,
           data == 0 ? "randn x logspace row scales" : "ReLU of the same: half zeros", kBufs);
    run<0>("0 read only", bufs, n_slabs, lo_win, want, d_out, d_st);
    run<4>("4 u16 histogram only", bufs, n_slabs, lo_win, want, d_out, d_st);
    run<1>("1 u16 histogram + below", bufs, n_slabs, lo_win, want, d_out, d_st);
    run<2>("2 same, zeros in registers", bufs, n_slabs, lo_win, want, d_out, d_st);
    run<3>("3 |x| keys, u32 bins + below", bufs, n_slabs, lo_abs, want_abs, d_out, d_st);
    for (int i = 0; i < kBufs; ++i) CK(hipFree(bufs[i]));
    CK(hipFree(d_out));
    CK(hipFree(d_st));
  }
  return 0;
}
