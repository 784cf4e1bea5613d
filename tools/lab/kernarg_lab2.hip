// Dev lab (round 4): what does a wave wait for before its first global load comes back?  Stamps by s_memrealtime
// (100 MHz) taken with volatile asm so that nothing moves across them:
//   t0 entry, t1 a scalar load from the argument block has returned, t2 a second one from ANOTHER 64-byte line of the block,
//   t3 a vector load through the pointer (rotating 64 MB buffers: HBM-cold) has returned.
// Built WITHOUT kernarg preload.  256 workgroups x 256 threads, back-to-back launches.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

struct Pad { unsigned v[40]; };

__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

__global__ __launch_bounds__(256) void probe(unsigned long long* stamps, const unsigned* data, Pad pad, unsigned tail) {
  const unsigned long long t0 = now();
  const unsigned* kp = (const unsigned*)__builtin_amdgcn_kernarg_segment_ptr();
  unsigned a, b;
  asm volatile("s_load_dword %0, %1, 0x8\n\ts_waitcnt lgkmcnt(0)" : "=s"(a) : "s"(kp) : "memory");
  const unsigned long long t1 = now();
  asm volatile("s_load_dword %0, %1, 0xb0\n\ts_waitcnt lgkmcnt(0)" : "=s"(b) : "s"(kp) : "memory");
  const unsigned long long t2 = now();
  unsigned v;
  const unsigned* p = data + blockIdx.x * 4096 + threadIdx.x * 4;
  asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  const unsigned long long t3 = now();
  if (threadIdx.x == 0) {
    stamps[blockIdx.x * 4 + 0] = t0;
    stamps[blockIdx.x * 4 + 1] = t1;
    stamps[blockIdx.x * 4 + 2] = t2;
    stamps[blockIdx.x * 4 + 3] = t3 + ((a + b + v + tail) == 0xdeadbeefu ? 1 : 0);
  }
}

int main() {
  const int wgs = 256;
  unsigned long long* stamps;
  unsigned* data;
  const size_t per = (size_t)wgs * 4096 * 4;  // 4 MB per launch
  const int rot = 96;                            // 384 MB: beyond L2 and the Infinity Cache
  hipMalloc(&stamps, wgs * 32);
  hipMalloc(&data, per * rot);
  hipMemset(data, 0, per * rot);
  hipStream_t st;
  hipStreamCreate(&st);
  std::vector<unsigned long long> h(wgs * 4);
  Pad pad{};
  for (int rep = 0; rep < 3; ++rep) {
    for (int i = 0; i < 500; ++i) probe<<<wgs, 256, 0, st>>>(stamps, data + (size_t)(i % rot) * (per / 4), pad, i);
    hipStreamSynchronize(st);
    hipMemcpy(h.data(), stamps, wgs * 32, hipMemcpyDeviceToHost);
    unsigned long long first = ~0ull;
    for (int i = 0; i < wgs; ++i) first = std::min(first, h[4 * i]);
    const char* names[4] = {"entry", "first argument line back", "second argument line back", "HBM-cold vector load back"};
    for (int k = 0; k < 4; ++k) {
      std::vector<double> d;
      for (int i = 0; i < wgs; ++i) d.push_back((h[4 * i + k] - first) * 0.01);
      std::sort(d.begin(), d.end());
      printf("  %-28s min %.2f median %.2f max %.2f us after the first workgroup's entry\n", names[k], d.front(), d[wgs / 2], d.back());
    }
    printf("\n");
  }
  return 0;
}
