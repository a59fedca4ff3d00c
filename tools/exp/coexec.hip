// Do f32-input MFMA (v_mfma_f32_32x32x2_f32) and plain f32 VALU work from ANOTHER wave of the same SIMD overlap?
// 512-thread blocks (two waves per SIMD): waves 0-3 run an MFMA chain, waves 4-7 a v_fma_f32 / v_fmac_f32_dpp chain.
// hipcc --offload-arch=gfx950 -O3 tools/exp/coexec.hip -o coexec && ./coexec
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = float __attribute__((ext_vector_type(16)));

using bf16x8 = __bf16 __attribute__((ext_vector_type(8)));

template <int MODE>  // bit0: MFMA waves work, bit1: VALU waves work, bit2: VALU waves use DPP fmac, bit3: bf16 MFMA instead of f32
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  const int wave = threadIdx.x >> 6;
  float x = threadIdx.x * 1e-3f, y = 1.0001f;
  if (wave < 4) {
    if (MODE & 1) {
      f32x16 acc = {0};
      if (MODE & 8) {
        bf16x8 a8, b8;
#pragma unroll
        for (int u = 0; u < 8; ++u) { a8[u] = (__bf16)(x + u); b8[u] = (__bf16)(y + u); }
        for (int i = 0; i < iters; ++i) {
#pragma unroll
          for (int u = 0; u < 32; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc, 0, 0, 0);  // 32 x 32 cycles = the time of 16 f32 MFMAs
        }
      } else {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
          for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc, 0, 0, 0);
        }
      }
      out[blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[5];
    }
  } else {
    if (MODE & 2) {
      float a[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) a[u] = x + u;
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {  // 16 x 16 = 256 VALU instructions = 1024 issue cycles, the time of 16 MFMAs
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            if (MODE & 4)
              asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[u]) : "v"(y), "v"(x));
            else
              asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[u]) : "v"(y), "v"(x));
          }
        }
      }
      float s = 0;
#pragma unroll
      for (int u = 0; u < 16; ++u) s += a[u];
      out[blockIdx.x * 512 + threadIdx.x] = s;
    }
  }
}

template <int MODE>
float run(float* out, int iters) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  k<MODE><<<256, 512>>>(out, 10);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<MODE><<<256, 512>>>(out, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * 4);
  const int it = 4000;
  printf("MFMA waves only           : %.3f ms\n", run<1>(out, it));
  printf("VALU waves only (fmac)    : %.3f ms\n", run<2>(out, it));
  printf("both (fmac)               : %.3f ms\n", run<3>(out, it));
  printf("VALU waves only (dpp fmac): %.3f ms\n", run<6>(out, it));
  printf("both (dpp fmac)           : %.3f ms\n", run<7>(out, it));
  printf("bf16 MFMA waves only      : %.3f ms\n", run<9>(out, it));
  printf("bf16 MFMA + VALU (fmac)   : %.3f ms\n", run<11>(out, it));
  printf("bf16 MFMA + VALU (dpp)    : %.3f ms\n", run<15>(out, it));
  return 0;
}
