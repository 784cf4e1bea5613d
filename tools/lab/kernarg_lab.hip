// Dev lab (round 4): how long does a wave wait for its kernel arguments?
// Each workgroup stamps s_memrealtime at entry and again once a value loaded through a pointer ARGUMENT has come back;
// a second kernel does the same with the pointer baked into a __device__ global (no kernarg read on the path? still one
// for the stamp buffer -- so that variant reads the stamp pointer AFTER the first stamp is taken into a register).
// Run plain, with HIP_FORCE_DEV_KERNARG=1, and built with -mllvm -amdgpu-kernarg-preload-count=N.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(256) void probe(unsigned long long* stamps, const unsigned* data, unsigned* sink, int n) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  // the first use of any argument
  const unsigned v = data[blockIdx.x * 256 + threadIdx.x];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (v == 0xdeadbeefu) sink[0] = v + n;
  if (threadIdx.x == 0) {
    stamps[blockIdx.x * 2] = t0;
    stamps[blockIdx.x * 2 + 1] = t1;
  }
}

__global__ void empty_kernel(int) {}

int main(int argc, char** argv) {
  const int wgs = argc > 1 ? atoi(argv[1]) : 256;
  unsigned long long* stamps;
  unsigned *data, *sink;
  hipMalloc(&stamps, wgs * 16);
  hipMalloc(&data, wgs * 1024 * 64);
  hipMalloc(&sink, 64);
  hipMemset(data, 0, wgs * 1024 * 64);
  hipStream_t st;
  hipStreamCreate(&st);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  std::vector<unsigned long long> h(wgs * 2);
  // back-to-back intervals
  for (int rep = 0; rep < 3; ++rep) {
    for (int i = 0; i < 200; ++i) empty_kernel<<<1, 64, 0, st>>>(i);
    hipEventRecord(a, st);
    for (int i = 0; i < 2000; ++i) empty_kernel<<<1, 64, 0, st>>>(i);
    hipEventRecord(b, st);
    hipStreamSynchronize(st);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("empty kernel, back to back: %.2f us per launch\n", ms * 1e3 / 2000);
  }
  for (int rep = 0; rep < 3; ++rep) {
    for (int i = 0; i < 200; ++i) probe<<<wgs, 256, 0, st>>>(stamps, data + (i % 64) * wgs * 256, sink, i);
    hipEventRecord(a, st);
    for (int i = 0; i < 2000; ++i) probe<<<wgs, 256, 0, st>>>(stamps, data + (i % 64) * wgs * 256, sink, i);
    hipEventRecord(b, st);
    hipStreamSynchronize(st);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    hipMemcpy(h.data(), stamps, wgs * 16, hipMemcpyDeviceToHost);
    unsigned long long first = ~0ull;
    for (int i = 0; i < wgs; ++i) first = std::min(first, h[2 * i]);
    std::vector<double> d, e;
    for (int i = 0; i < wgs; ++i) {
      d.push_back((h[2 * i + 1] - h[2 * i]) * 0.01);
      e.push_back((h[2 * i + 1] - first) * 0.01);
    }
    std::sort(d.begin(), d.end());
    std::sort(e.begin(), e.end());
    printf("probe %d wgs: %.2f us per launch; entry -> first argument-dependent load back: min %.2f median %.2f max %.2f us;"
           " first entry -> last value back %.2f us\n",
           wgs, ms * 1e3 / 2000, d.front(), d[wgs / 2], d.back(), e.back());
  }
  return 0;
}
