#include <hip/hip_runtime.h>
#include <cstdio>
template <int BM>
__global__ void k(double* out) {
  double v = (double)threadIdx.x, one = 1.0, acc = -1.0;
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:%3" : "+v"(acc) : "v"(v), "v"(one), "n"(BM));
  out[threadIdx.x] = acc;
}
template <int BM>
__global__ void km(double* out) {
  double v = (double)threadIdx.x, acc = -1.0;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:%2" : "+v"(acc) : "v"(v), "n"(BM));
  out[threadIdx.x] = acc;
}
template <int BM> void run(double* d, bool mov) {
  if (mov) km<BM><<<1, 64>>>(d); else k<BM><<<1, 64>>>(d);
  double h[64];
  (void)hipMemcpy(h, d, 64 * 8, hipMemcpyDeviceToHost);
  printf("%s bank_mask %x: ", mov ? "mov " : "fmac", BM);
  for (int i = 0; i < 16; ++i) printf("%g ", h[i]);
  printf("\n");
}
int main() {
  double* d; (void)hipMalloc(&d, 64 * 8);
  run<1>(d, 0); run<2>(d, 0); run<4>(d, 0); run<8>(d, 0); run<3>(d, 0); run<12>(d, 0); run<15>(d, 0);
  run<1>(d, 1); run<2>(d, 1); run<3>(d, 1); run<12>(d, 1); run<15>(d, 1);
  return 0;
}
