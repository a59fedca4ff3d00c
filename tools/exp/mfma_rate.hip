// issue rate of the fp64 / fp32 16x16x4 MFMA on gfx950 (sets the matrix-core roofline of K3a / K4)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <typename T> struct V;
template <> struct V<double> { using t = f64x4; static __device__ t run(double a, double b, t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); } };
template <> struct V<float> { using t = f32x4; static __device__ t run(float a, float b, t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); } };
template <typename T>
__global__ __launch_bounds__(256) void k(T* out, int iters) {
  typename V<T>::t c0 = {}, c1 = {}, c2 = {}, c3 = {};
  T a = (T)threadIdx.x * (T)1e-3, b = (T)blockIdx.x * (T)1e-3;
  for (int i = 0; i < iters; ++i) {
    c0 = V<T>::run(a, b, c0); c1 = V<T>::run(a, b, c1); c2 = V<T>::run(a, b, c2); c3 = V<T>::run(a, b, c3);
  }
  out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
template <typename T> void run(const char* name, int waves_per_simd) {
  const int blocks = 256 * waves_per_simd, iters = 20000;
  T* d; (void)hipMalloc(&d, (size_t)blocks * 256 * sizeof(T));
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<T><<<blocks, 256>>>(d, 100);
  (void)hipEventRecord(e0); k<T><<<blocks, 256>>>(d, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)blocks * 4 * iters * 4;  // wave-level MFMAs
  printf("%s, %d wave(s)/SIMD: %.2f ms, %.1f TFLOP/s, %.1f ns per MFMA per SIMD\n", name, waves_per_simd, ms,
         n * 2048 / ms / 1e9, ms * 1e6 / (n / 1024));
  (void)hipFree(d);
}
int main() { run<double>("f64 16x16x4", 1); run<double>("f64 16x16x4", 4); run<float>("f32 16x16x4", 1); run<float>("f32 16x16x4", 4); return 0; }
