// HBM ceilings of THIS box for the traffic mixes of the bench kernels: pure read, pure write, copy (1 read : 1 write) and the
// 1 : 2 mix of K2 (reads SExx, writes invU and U).  16 bytes per lane, grid-stride, 2 GB per stream like the headline.
//   hipcc -O3 --offload-arch=gfx950 tools/exp/stream_mix.hip -o tools/exp/stream_mix && tools/exp/stream_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NR, int NW>
__global__ __launch_bounds__(256) void k_mix(const f4* __restrict__ r0, f4* __restrict__ w0, f4* __restrict__ w1, size_t n, float* sink) {
  f4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    f4 v = {1, 2, 3, 4};
    if (NR >= 1) v = r0[i];
    if (NW >= 1) w0[i] = v;
    if (NW >= 2) w1[i] = v * 2.0f;
    if (NW == 0) acc += v;
  }
  if (NW == 0 && acc.x == 123.456f) *sink = acc.y;
}

template <int NR, int NW>
static void run(const char* name, f4* a, f4* b, f4* c, size_t n, float* sink, int blocks) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> ts;
  for (int it = 0; it < 12; ++it) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_mix<NR, NW>), dim3(blocks), dim3(256), 0, 0, a, b, c, n, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (it >= 2) ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  const double bytes = (double)(NR + NW) * n * 16;
  printf("%-22s blocks %6d: median %.4f ms -> %.0f GB/s (%.1f %% of 8 TB/s)\n", name, blocks, ts[ts.size() / 2],
         bytes / ts[ts.size() / 2] / 1e6, bytes / ts[ts.size() / 2] / 1e6 / 80.0);
}

int main() {
  const size_t n = (size_t)2048 * 1000 * 1000 / 16;  // 2.048 GB per stream = 1e6 x 16 x 16 doubles
  f4 *a, *b, *c; float* sink;
  CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMalloc(&c, n * 16)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(a, 1, n * 16));
  for (int blocks : {256 * 8, 256 * 32, (int)((n + 255) / 256)}) {
    run<1, 0>("read", a, b, c, n, sink, blocks);
    run<0, 1>("write", a, b, c, n, sink, blocks);
    run<1, 1>("copy 1:1", a, b, c, n, sink, blocks);
    run<1, 2>("mix 1 read : 2 writes", a, b, c, n, sink, blocks);
  }
  return 0;
}
