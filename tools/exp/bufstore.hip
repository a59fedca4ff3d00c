#include <hip/hip_runtime.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(double* out, int n) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out + blockIdx.x * 128, 0, n * 8, 0x00020000);
  u32x4 v; v[0] = threadIdx.x; v[1] = 1; v[2] = 2; v[3] = 3;
  int off = (threadIdx.x < 32) ? threadIdx.x * 16 : 0x80000000;
  __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 0);
  u32x2 w; w[0] = 5; w[1] = 6;
  __builtin_amdgcn_raw_buffer_store_b64(w, r, off, 0, 0);
  __builtin_amdgcn_raw_buffer_store_b32(7u, r, off, 0, 0);
}
#include <cstdio>
int main() {
  double* d; (void)hipMalloc(&d, 2 * 128 * 8); (void)hipMemset(d, 0, 2 * 128 * 8);
  k<<<2, 64>>>(d, 40);  // 40 doubles = 320 bytes window: lanes with off+16 > 320 (lane >= 20) dropped
  unsigned h[2 * 256]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int b = 0; b < 2; ++b) { for (int i = 0; i < 100; ++i) printf("%u ", h[b * 256 + i]); printf("\n"); }
  return 0;
}
