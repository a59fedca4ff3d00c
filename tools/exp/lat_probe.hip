// Latencies that bound K9's block form, measured the way it runs: 4 waves per block (one per SIMD), few blocks.
//   hipcc -O3 --offload-arch=gfx950 tools/exp/lat_probe.hip -o tools/exp/lat_probe && tools/exp/lat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define N 512
__global__ __launch_bounds__(256) void k_fma(double* o, double a, double b) {
  double x = o[threadIdx.x];
#pragma unroll 16
  for (int i = 0; i < N; ++i) x = __builtin_fma(x, a, b);
  o[threadIdx.x] = x;
}
__global__ __launch_bounds__(256) void k_fma4(double* o, double a, double b) {  // four independent chains
  double x0 = o[threadIdx.x], x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
#pragma unroll 16
  for (int i = 0; i < N; ++i) {
    x0 = __builtin_fma(x0, a, b); x1 = __builtin_fma(x1, a, b); x2 = __builtin_fma(x2, a, b); x3 = __builtin_fma(x3, a, b);
  }
  o[threadIdx.x] = x0 + x1 + x2 + x3;
}
// scalar instructions beside vector ones: do they take issue slots of a lone wave?  NS s_mul_i32 per v_fma_f64 (4 independent chains)
template <int NS>
__global__ __launch_bounds__(256) void k_fma4_salu(double* o, double a, double b, int seed) {
  double x0 = o[threadIdx.x], x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  int s0 = seed, s1 = seed + 1, s2 = seed + 2, s3 = seed + 3;
#pragma unroll 4
  for (int i = 0; i < N; ++i) {
    x0 = __builtin_fma(x0, a, b);
    if (NS >= 1) asm volatile("s_mul_i32 %0, %0, %1" : "+s"(s0) : "s"(seed) : "scc");
    if (NS >= 2) asm volatile("s_mul_i32 %0, %0, %1" : "+s"(s1) : "s"(seed) : "scc");
    x1 = __builtin_fma(x1, a, b);
    if (NS >= 3) asm volatile("s_mul_i32 %0, %0, %1" : "+s"(s2) : "s"(seed) : "scc");
    if (NS >= 4) asm volatile("s_mul_i32 %0, %0, %1" : "+s"(s3) : "s"(seed) : "scc");
    x2 = __builtin_fma(x2, a, b);
    if (NS >= 1) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(seed) : "scc");
    if (NS >= 2) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s1) : "s"(seed) : "scc");
    x3 = __builtin_fma(x3, a, b);
    if (NS >= 3) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s2) : "s"(seed) : "scc");
    if (NS >= 4) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s3) : "s"(seed) : "scc");
  }
  o[threadIdx.x] = x0 + x1 + x2 + x3 + (double)(s0 + s1 + s2 + s3);
}
__global__ __launch_bounds__(256) void k_rcp(double* o) {
  double x = o[threadIdx.x];
#pragma unroll 8
  for (int i = 0; i < N; ++i) x = __builtin_amdgcn_rcp(x);
  o[threadIdx.x] = x;
}
__global__ __launch_bounds__(256) void k_fmaf(float* o, float a, float b) {
  float x = o[threadIdx.x];
#pragma unroll 16
  for (int i = 0; i < N; ++i) x = __builtin_fmaf(x, a, b);
  o[threadIdx.x] = x;
}
__global__ __launch_bounds__(256) void k_lds(double* o) {  // write -> wave sync -> read of another lane's value
  __shared__ double buf[256];
  double x = o[threadIdx.x];
  const int other = threadIdx.x ^ 1;
#pragma unroll 4
  for (int i = 0; i < N; ++i) {
    buf[threadIdx.x] = x;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    x = buf[other] + 1.0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
  }
  o[threadIdx.x] = x;
}
__global__ __launch_bounds__(256) void k_bar(double* o) {  // LDS exchange across waves with a block barrier
  __shared__ double buf[2][256];
  double x = o[threadIdx.x];
  const int other = (threadIdx.x + 64) & 255;
#pragma unroll 4
  for (int i = 0; i < N; ++i) {
    buf[i & 1][threadIdx.x] = x;
    __syncthreads();
    x = buf[i & 1][other] + 1.0;
  }
  o[threadIdx.x] = x;
}
__global__ __launch_bounds__(256) void k_mfma_dep(double* o) {
  d4 acc = {0, 0, 0, 0};
  double a = o[threadIdx.x], b = o[threadIdx.x + 256];
#pragma unroll 8
  for (int i = 0; i < N; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  o[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
__global__ __launch_bounds__(256) void k_mfma_ind(double* o) {
  d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  double a = o[threadIdx.x], b = o[threadIdx.x + 256];
#pragma unroll 4
  for (int i = 0; i < N / 4; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  o[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
__global__ __launch_bounds__(256) void k_mfma_use(double* o) {  // MFMA -> VALU use of its result -> next MFMA operand
  d4 acc = {0, 0, 0, 0};
  double a = o[threadIdx.x], b = o[threadIdx.x + 256];
#pragma unroll 8
  for (int i = 0; i < N; ++i) {
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    a = acc[0] * 0.5;
  }
  o[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + a;
}
template <typename F>
static void run(const char* name, F launch, int nblk) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(nblk);
  hipDeviceSynchronize();
  float best = 1e9;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(e0);
    launch(nblk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  printf("%-34s %4d blocks: %8.1f ns per step (%.0f cycles at 2.4 GHz)\n", name, nblk, best * 1e6 / N, best * 1e6 / N * 2.4);
}
int main() {
  double* d; float* f;
  hipMalloc(&d, 1 << 20); hipMalloc(&f, 1 << 20);
  std::vector<double> h(1 << 17, 1.0000001);
  hipMemcpy(d, h.data(), 1 << 20, hipMemcpyHostToDevice);
  hipMemset(f, 0, 1 << 20);
  for (int nblk : {40, 256}) {
    run("dependent v_fma_f64", [&](int n) { hipLaunchKernelGGL(k_fma, dim3(n), dim3(256), 0, 0, d, 1.0000001, 1e-9); }, nblk);
    run("4 independent v_fma_f64 chains", [&](int n) { hipLaunchKernelGGL(k_fma4, dim3(n), dim3(256), 0, 0, d, 1.0000001, 1e-9); }, nblk);
    run("4 fma chains + 4 salu per 4 fma", [&](int n) { hipLaunchKernelGGL(k_fma4_salu<2>, dim3(n), dim3(256), 0, 0, d, 1.0000001, 1e-9, 3); }, nblk);
    run("4 fma chains + 8 salu per 4 fma", [&](int n) { hipLaunchKernelGGL(k_fma4_salu<4>, dim3(n), dim3(256), 0, 0, d, 1.0000001, 1e-9, 3); }, nblk);
    run("dependent v_fma_f32", [&](int n) { hipLaunchKernelGGL(k_fmaf, dim3(n), dim3(256), 0, 0, f, 1.0000001f, 1e-9f); }, nblk);
    run("dependent v_rcp_f64", [&](int n) { hipLaunchKernelGGL(k_rcp, dim3(n), dim3(256), 0, 0, d); }, nblk);
    run("LDS write/sync/read (wave)", [&](int n) { hipLaunchKernelGGL(k_lds, dim3(n), dim3(256), 0, 0, d); }, nblk);
    run("LDS write/barrier/read (block)", [&](int n) { hipLaunchKernelGGL(k_bar, dim3(n), dim3(256), 0, 0, d); }, nblk);
    run("dependent mfma_f64_16x16x4", [&](int n) { hipLaunchKernelGGL(k_mfma_dep, dim3(n), dim3(256), 0, 0, d); }, nblk);
    run("independent mfma_f64_16x16x4", [&](int n) { hipLaunchKernelGGL(k_mfma_ind, dim3(n), dim3(256), 0, 0, d); }, nblk);
    run("mfma -> valu -> mfma", [&](int n) { hipLaunchKernelGGL(k_mfma_use, dim3(n), dim3(256), 0, 0, d); }, nblk);
  }
  return 0;
}
