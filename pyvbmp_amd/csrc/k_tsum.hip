// K10: time-integrated cross moments for the LDS M-step statistics
// (ref models/LinearDynamicalSystems.py:173-190: SE_x_x, SE_xp_x, SE_x_u, SE_x_y, SE_y_y, ...):
//     out[s,i,j] = sum_t a[t,s,i] * b[t,s,j]  (+ sum_t M[t,s,i,j])
// One thread per output element, time loop in registers; a / b / M are addressed with (time, series)
// strides in elements so that time-shifted views (mu[:-1] vs mu[1:]) and inputs that are constant over
// time or series (stride 0) need no copies.  No (T,S,da,db) temporary ever exists.  gfx950 only.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vbmp_dispatch.h"
#include "../../include/vbmp_hip.h"

namespace vbmp {

template <typename T>
__global__ __launch_bounds__(256) void k_tsum_outer(const T* __restrict__ a, int64_t sa_t, int64_t sa_s, int da,
                                                    const T* __restrict__ b, int64_t sb_t, int64_t sb_s, int db,
                                                    const T* __restrict__ M, int64_t sM_t, int64_t sM_s, int64_t Tn,
                                                    int64_t Tc, int64_t S, T* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int dd = da * db;
  if (idx >= S * dd) return;
  const int64_t s = idx / dd;
  const int e = (int)(idx - s * dd);
  const int i = e / db, j = e - i * db;
  // the time axis is cut into gridDim.y chunks when the outputs alone cannot fill the chip (a few thousand series of a
  // 6 x 6 moment are two blocks per CU, each thread walking all T steps with four loads in flight)
  const int64_t tb = (int64_t)blockIdx.y * Tc;
  const int64_t te = (tb + Tc < Tn) ? tb + Tc : Tn;
  const T* pa = a + s * sa_s + i;
  const T* pb = b + s * sb_s + j;
  T acc0 = T(0), acc1 = T(0), acc2 = T(0), acc3 = T(0);
  int64_t t = tb;
  for (; t + 4 <= te; t += 4) {
    acc0 = __builtin_fma(pa[(t + 0) * sa_t], pb[(t + 0) * sb_t], acc0);
    acc1 = __builtin_fma(pa[(t + 1) * sa_t], pb[(t + 1) * sb_t], acc1);
    acc2 = __builtin_fma(pa[(t + 2) * sa_t], pb[(t + 2) * sb_t], acc2);
    acc3 = __builtin_fma(pa[(t + 3) * sa_t], pb[(t + 3) * sb_t], acc3);
  }
  for (; t < te; ++t) acc0 = __builtin_fma(pa[t * sa_t], pb[t * sb_t], acc0);
  T acc = (acc0 + acc1) + (acc2 + acc3);
  if (M) {
    const T* pm = M + s * sM_s + e;
    T m0 = T(0), m1 = T(0);
    int64_t u = tb;
    for (; u + 2 <= te; u += 2) {
      m0 += pm[(u + 0) * sM_t];
      m1 += pm[(u + 1) * sM_t];
    }
    for (; u < te; ++u) m0 += pm[u * sM_t];
    acc += m0 + m1;
  }
  if (gridDim.y == 1)
    out[idx] = acc;
  else
    atomicAdd(&out[idx], acc);  // out is zeroed by the dispatcher
}

template <typename T>
static int tsum_dispatch(const T* a, int64_t sa_t, int64_t sa_s, int da, const T* b, int64_t sb_t, int64_t sb_s, int db,
                         const T* M, int64_t sM_t, int64_t sM_s, int64_t Tn, int64_t S, T* out, void* stream) {
  if (S == 0 || da == 0 || db == 0) return 0;
  if (!a || !b || !out || Tn < 0 || S < 0 || da < 1 || db < 1) return VBMP_ERR_ARG;
  const int64_t n = S * da * db;
  const int64_t bx = (n + 255) / 256;
  // aim at >= 16 blocks per CU; at least 64 time steps per chunk
  int64_t by = (bx >= 4096 || Tn < 128) ? 1 : (4096 + bx - 1) / bx;
  if (by > Tn / 64) by = Tn / 64 > 0 ? Tn / 64 : 1;
  if (by > 1024) by = 1024;
  const int64_t Tc = (Tn + by - 1) / by;
  by = (Tn + Tc - 1) / Tc;
  if (by > 1 && hipMemsetAsync(out, 0, (size_t)n * sizeof(T), (hipStream_t)stream) != hipSuccess) return VBMP_ERR_LAUNCH;
  hipLaunchKernelGGL((k_tsum_outer<T>), dim3((unsigned)bx, (unsigned)by), dim3(256), 0, (hipStream_t)stream, a, sa_t, sa_s,
                     da, b, sb_t, sb_s, db, M, sM_t, sM_s, Tn, Tc, S, out);
  return hipGetLastError() == hipSuccess ? 0 : VBMP_ERR_LAUNCH;
}

}  // namespace vbmp

extern "C" {
int vbmp_tsum_outer_f64(const double* a, int64_t sa_t, int64_t sa_s, int da, const double* b, int64_t sb_t, int64_t sb_s,
                        int db, const double* M, int64_t sM_t, int64_t sM_s, int64_t Tn, int64_t S, double* out,
                        void* stream) {
  return vbmp::tsum_dispatch<double>(a, sa_t, sa_s, da, b, sb_t, sb_s, db, M, sM_t, sM_s, Tn, S, out, stream);
}
int vbmp_tsum_outer_f32(const float* a, int64_t sa_t, int64_t sa_s, int da, const float* b, int64_t sb_t, int64_t sb_s,
                        int db, const float* M, int64_t sM_t, int64_t sM_s, int64_t Tn, int64_t S, float* out,
                        void* stream) {
  return vbmp::tsum_dispatch<float>(a, sa_t, sa_s, da, b, sb_t, sb_s, db, M, sM_t, sM_s, Tn, S, out, stream);
}
}
