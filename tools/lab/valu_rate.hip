// Dev tool: issue rate of the vector instructions of the GPTQ decode on gfx950 (cycles per wave64 instruction per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) void rate(uint32_t* out, int iters) {
  uint32_t a[8];
  f32x2 f[8];
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 2654435761u + i; f[i] = f32x2{1.0f + i, 2.0f}; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) asm volatile("v_and_b32 %0, 0x0f0f0f0f, %0" : "+v"(a[i]));
      if (OP == 1) asm volatile("v_cvt_pk_f32_fp8 %0, %1" : "=v"(f[i]) : "v"(a[i]));
      if (OP == 2) asm volatile("v_cvt_pk_f32_fp8_sdwa %0, %1 src0_sel:WORD_1" : "=v"(f[i]) : "v"(a[i]));
      if (OP == 3) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(f[i]));
      if (OP == 4) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
      if (OP == 5) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(f[i][0]) : "v"(a[i]));
      if (OP == 6) asm volatile("v_lshrrev_b32 %0, 4, %0" : "+v"(a[i]));
    }
  }
  uint32_t s = 0;
  for (int i = 0; i < 8; ++i) s += a[i] + __builtin_bit_cast(uint32_t, f[i][0]) + __builtin_bit_cast(uint32_t, f[i][1]);
  if (s == 0x12345) out[0] = s;
}
template <int OP>
void run(const char* name, uint32_t* out) {
  const int iters = 4096;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  rate<OP><<<256 * 2, 256>>>(out, iters);  // 2 workgroups per CU: 2 waves per SIMD
  hipEventRecord(a);
  rate<OP><<<256 * 2, 256>>>(out, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  // per SIMD: 2 waves x iters x 8 instructions
  const double cyc = ms * 1e-3 * 2.4e9 / (2.0 * iters * 8);
  printf("%-28s %6.2f cycles per wave instruction (2.4 GHz assumed)\n", name, cyc);
}
int main() {
  uint32_t* out; hipMalloc(&out, 4);
  run<0>("v_and_b32", out); run<6>("v_lshrrev_b32", out); run<1>("v_cvt_pk_f32_fp8", out); run<2>("v_cvt_pk_f32_fp8_sdwa", out);
  run<3>("v_pk_fma_f32", out); run<4>("v_fma_f32", out); run<5>("v_cvt_f32_ubyte0", out);
  return 0;
}
