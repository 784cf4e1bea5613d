// Dev tool: does the raw-buffer range check of gfx950 include the SGPR offset?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(const uint32_t* p, uint32_t* o, uint32_t so) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(p), 0, 512, 0x00020000);
  o[0] = __builtin_amdgcn_raw_buffer_load_b32(r, 0, so, 0);        // voffset 0, soffset 768: beyond num_records only via soffset
  o[1] = __builtin_amdgcn_raw_buffer_load_b32(r, 600, 0, 0);       // voffset beyond num_records
  o[2] = __builtin_amdgcn_raw_buffer_load_b32(r, 256, so / 3, 0);  // 256 + 256 = 512: sum reaches num_records
  o[3] = __builtin_amdgcn_raw_buffer_load_b32(r, 256, 0, 0);       // in range
}
int main() {
  uint32_t *p, *o, h[256], ho[4];
  for (int i = 0; i < 256; ++i) h[i] = 1000 + i;
  hipMalloc(&p, 1024); hipMalloc(&o, 16);
  hipMemcpy(p, h, 1024, hipMemcpyHostToDevice);
  k<<<1, 1>>>(p, o, 768);
  hipMemcpy(ho, o, 16, hipMemcpyDeviceToHost);
  printf("soffset-only beyond: %u (1192 = soffset NOT checked, 0 = checked)\nvoffset beyond: %u (expect 0)\nsum reaches: %u (1128 = soffset not checked)\nin range: %u (expect 1064)\n", ho[0], ho[1], ho[2], ho[3]);
  return 0;
}
