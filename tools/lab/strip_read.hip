// Dev tool: what does the access pattern of the GPTQ strip kernel cost by itself?  Reads a [rows, cols] int32
// matrix with workgroups that own a strip of W bytes per row and a K range, every thread keeping R 16-byte loads
// in flight -- no decode, no activations -- and reports TB/s per (W, R, split, workgroups/CU hint).
//   hipcc --offload-arch=gfx950 -O3 -o strip_read strip_read.hip && ./strip_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int WQ, int R, bool SWZ>  // WQ = 16-byte words per row piece (8 = 128 B), R = rows in flight per thread
__global__ __launch_bounds__(256) void strip_read(const uint32_t* __restrict__ qw, int64_t rows, int64_t cols,
                                                  uint32_t* __restrict__ sink) {
  constexpr int KL = 256 / WQ;
  const int cl = threadIdx.x % WQ, kl = threadIdx.x / WQ;
  uint32_t strip = blockIdx.x;
  if (SWZ) strip = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int64_t col0 = static_cast<int64_t>(strip) * WQ * 4 + cl * 4;
  const int64_t step = static_cast<int64_t>(gridDim.y) * KL * R;
  u32x4 acc = {0, 0, 0, 0};
  for (int64_t pass0 = static_cast<int64_t>(blockIdx.y) * KL * R; pass0 < rows; pass0 += step) {
    u32x4 w[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      int64_t r = pass0 + kl * R + i;
      if (r >= rows) r = rows - 1;
      w[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(qw + r * cols + col0));
    }
#pragma unroll
    for (int i = 0; i < R; ++i) acc ^= w[i];
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

static int g_copies = 1;
template <int WQ, int R, bool SWZ>
void run(const uint32_t* qw0, int64_t rows, int64_t cols, uint32_t* sink, int split) {
  int call = 0;
#define qw (qw0 + static_cast<int64_t>((call++) % g_copies) * rows * cols)
  dim3 grid(static_cast<uint32_t>(cols / (WQ * 4)), split);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    for (int i = 0; i < 3; ++i) strip_read<WQ, R, SWZ><<<grid, 256>>>(qw, rows, cols, sink);
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) strip_read<WQ, R, SWZ><<<grid, 256>>>(qw, rows, cols, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (ms / 10 < best) best = ms / 10;
  }
  const double bytes = static_cast<double>(rows) * cols * 4;
  printf("W=%4d B  R=%2d swz=%d split=%2d grid=%6u: %7.1f us  %5.2f TB/s\n", WQ * 16, R, SWZ ? 1 : 0, split,
         grid.x * grid.y, best * 1e3, bytes / best / 1e9);
#undef qw
}


// persistent workers as in gptq_stream_kernel: G workgroups walk strips w, w + G, ...; NS register sets of 8 rows
// each, NS - 1 in flight while one is consumed
#include <type_traits>
template <int N> struct SFor {
  template <typename F> static __device__ __forceinline__ bool run(F&& f) {
    if constexpr (N > 0) { if (!SFor<N - 1>::run(f)) return false; return f(std::integral_constant<int, N - 1>{}); } else return true;
  }
};
template <int NS>
__global__ __launch_bounds__(256) void strip_persist(const uint32_t* __restrict__ qw, uint32_t rows, uint32_t cols, uint32_t* __restrict__ sink) {
  const int cl = threadIdx.x & 7, kl = threadIdx.x >> 3;
  const uint32_t strips = cols / 32;
  uint32_t worker = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  struct Cur { uint32_t strip, pass0; };
  auto adv = [&](Cur& c) { c.pass0 += 256; if (c.pass0 >= rows) { c.pass0 = 0; c.strip += gridDim.x; } };
  auto load = [&](const Cur& c, u32x4 (&w)[8]) {
    const uint32_t st = c.strip < strips ? c.strip : strips - 1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      uint32_t r = c.pass0 + kl * 8 + i; r = r < rows ? r : rows - 1;
      w[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(qw + static_cast<int64_t>(r) * cols + st * 32 + cl * 4));
    }
  };
  u32x4 w[NS][8], acc = {0, 0, 0, 0};
  Cur cur{worker, 0}, far{worker, 0};
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) { load(far, w[s]); adv(far); }
  bool more = cur.strip < strips;
  while (more) {
    more = SFor<NS>::run([&](auto tag) {
      constexpr int s = decltype(tag)::value;
      load(far, w[(s + NS - 1) % NS]); adv(far);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc ^= w[s][i];
      adv(cur);
      return cur.strip < strips;
    });
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}
template <int NS>
void run_persist(const uint32_t* qw0, int64_t rows, int64_t cols, uint32_t* sink, int grid) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e9; int call = 0;
  for (int rep = 0; rep < 3; ++rep) {
    for (int i = 0; i < 3; ++i) strip_persist<NS><<<grid, 256>>>(qw0 + static_cast<int64_t>((call++) % g_copies) * rows * cols, rows, cols, sink);
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) strip_persist<NS><<<grid, 256>>>(qw0 + static_cast<int64_t>((call++) % g_copies) * rows * cols, rows, cols, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (ms / 10 < best) best = ms / 10;
  }
  printf("persistent  NS=%d workers=%4d: %7.1f us  %5.2f TB/s\n", NS, grid, best * 1e3, static_cast<double>(rows) * cols * 4 / best / 1e9);
}

int main() {
  const int64_t shapes[2][2] = {{8192 / 8, 32768}, {12288 / 8, 49152}};
  for (auto& sh : shapes) {
    const int64_t rows = sh[0], cols = sh[1];
    uint32_t *qw, *sink;
    g_copies = static_cast<int>(6e8 / (rows * cols * 4)) + 1;  // rotate over > 256 MiB: no Infinity Cache hits
    hipMalloc(&qw, rows * cols * 4 * g_copies);
    hipMalloc(&sink, 4);
    hipMemset(qw, 1, rows * cols * 4 * g_copies);
    printf("qweight %lld x %lld (%.1f MB x %d copies)\n", (long long)rows, (long long)cols, rows * cols * 4 / 1e6, g_copies);
    run_persist<2>(qw, rows, cols, sink, 512); run_persist<3>(qw, rows, cols, sink, 512); run_persist<4>(qw, rows, cols, sink, 512);
    run_persist<6>(qw, rows, cols, sink, 512); run_persist<4>(qw, rows, cols, sink, 1024); run_persist<3>(qw, rows, cols, sink, 1536); run_persist<2>(qw, rows, cols, sink, 2048);
    for (int split : {2}) {
      run<8, 8, true>(qw, rows, cols, sink, split);
      run<8, 8, false>(qw, rows, cols, sink, split);
      run<8, 16, true>(qw, rows, cols, sink, split);
      run<16, 8, true>(qw, rows, cols, sink, split);
      run<32, 8, true>(qw, rows, cols, sink, split);
      run<32, 8, false>(qw, rows, cols, sink, split);
      run<64, 8, true>(qw, rows, cols, sink, split);
      run<64, 4, true>(qw, rows, cols, sink, split);
      run<256, 4, false>(qw, rows, cols, sink, split);
    }
    hipFree(qw); hipFree(sink);
  }
  return 0;
}
