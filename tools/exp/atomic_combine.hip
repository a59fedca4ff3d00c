// cost of the "every block adds its partial statistics to the same addresses" epilogue (K4, K3): N blocks, each
// thread adds to `per_thread` of `naddr` shared addresses with no-return float / double atomics
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T>
__global__ void k(T* out, int naddr) {
  for (int e = threadIdx.x; e < naddr; e += blockDim.x) atomicAdd(&out[e], (T)1);
}
template <typename T>
__global__ void k_rot(T* out, int naddr) {  // same, but every block starts at its own offset
  const int r = (blockIdx.x * 97) % naddr;
  for (int e = threadIdx.x; e < naddr; e += blockDim.x) atomicAdd(&out[(e + r) % naddr], (T)1);
}
template <typename T> void run(const char* name, int blocks, int threads, int naddr, bool rot) {
  T* d; (void)hipMalloc(&d, naddr * sizeof(T)); (void)hipMemset(d, 0, naddr * sizeof(T));
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  if (rot) k_rot<T><<<blocks, threads>>>(d, naddr); else k<T><<<blocks, threads>>>(d, naddr);
  (void)hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) { if (rot) k_rot<T><<<blocks, threads>>>(d, naddr); else k<T><<<blocks, threads>>>(d, naddr); }
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%s blocks=%d threads=%d naddr=%d %s: %.1f us per launch (%.1f ns per block)\n", name, blocks, threads, naddr,
         rot ? "rotated" : "aligned", ms * 100, ms * 1e5 / blocks);
  (void)hipFree(d);
}
int main() {
  for (int blocks : {256, 1024, 4096}) {
    run<float>("f32", blocks, 256, 1088, false);
    run<float>("f32", blocks, 256, 1088, true);
    run<double>("f64", blocks, 256, 1088, false);
  }
  run<float>("f32", 1024, 256, 64, false);
  run<float>("f32", 1024, 256, 16384, false);
  return 0;
}
