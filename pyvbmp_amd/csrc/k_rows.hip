// K12: one small shared matrix applied to very many rows,  out[s, :] = M x[s, :] + c   (s < S; M n x k, k, n <= 64):
// the natural-parameter message eta_t = E[A' R] y_t of every observation in MatrixNormalWishart.Elog_like_X
// (transforms/MatrixNormalWishart.py:251-261), called by the LDS E-step on (T * series) rows.  The library GEMM for
// this shape (4 096 000 x 6 by 6 x 6, fp64) runs at 0.7 TB/s of HBM traffic; the product is pure streaming: a row per
// thread, M and c in LDS (every lane reads the same word: a broadcast), rows read and written once.  gfx950 only.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vbmp_dispatch.h"
#include "../../include/vbmp_hip.h"

namespace vbmp {

template <typename T, int KP>
__global__ __launch_bounds__(256) void k_rows_affine(const T* __restrict__ X, int64_t S, int k, const T* __restrict__ M,
                                                     const T* __restrict__ c, int n, T* __restrict__ out) {
  __shared__ T Ms[64 * KP + 64];
  T* cs = Ms + 64 * KP;
  for (int e = threadIdx.x; e < n * KP; e += 256) {
    const int j = e / KP, i = e - j * KP;
    Ms[e] = (i < k) ? M[j * k + i] : T(0);
  }
  for (int j = threadIdx.x; j < n; j += 256) cs[j] = c ? c[j] : T(0);
  __syncthreads();
  for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < S; s += (int64_t)gridDim.x * 256) {
    T x[KP];
    const T* xr = X + s * k;
#pragma unroll
    for (int i = 0; i < KP; ++i) x[i] = (i < k) ? xr[i] : T(0);
    T* o = out + s * n;
    for (int j = 0; j < n; ++j) {
      T acc = cs[j];
#pragma unroll
      for (int i = 0; i < KP; ++i) acc = __builtin_fma(Ms[j * KP + i], x[i], acc);
      o[j] = acc;
    }
  }
}

template <typename T, int KP>
static int launch_rows(const T* X, int64_t S, int k, const T* M, const T* c, int n, T* out, hipStream_t st) {
  int64_t blocks = (S + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;  // grid-stride beyond 32 blocks per CU: M is staged once per block
  hipLaunchKernelGGL((k_rows_affine<T, KP>), dim3((unsigned)blocks), dim3(256), 0, st, X, S, k, M, c, n, out);
  return hipGetLastError() == hipSuccess ? 0 : VBMP_ERR_LAUNCH;
}

template <typename T>
static int rows_dispatch(const T* X, int64_t S, int k, const T* M, const T* c, int n, T* out, void* stream) {
  if (S == 0) return 0;
  if (!X || !M || !out || S < 0 || k < 1 || n < 1 || k > VBMP_ROWS_MAX_DIM || n > VBMP_ROWS_MAX_DIM) return VBMP_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (k <= 4) return launch_rows<T, 4>(X, S, k, M, c, n, out, st);
  if (k <= 8) return launch_rows<T, 8>(X, S, k, M, c, n, out, st);
  if (k <= 16) return launch_rows<T, 16>(X, S, k, M, c, n, out, st);
  if (k <= 32) return launch_rows<T, 32>(X, S, k, M, c, n, out, st);
  return launch_rows<T, 64>(X, S, k, M, c, n, out, st);
}

}  // namespace vbmp

extern "C" {
int vbmp_rows_affine_f64(const double* X, int64_t S, int k, const double* M, const double* c, int n, double* out,
                         void* stream) {
  return vbmp::rows_dispatch<double>(X, S, k, M, c, n, out, stream);
}
int vbmp_rows_affine_f32(const float* X, int64_t S, int k, const float* M, const float* c, int n, float* out,
                         void* stream) {
  return vbmp::rows_dispatch<float>(X, S, k, M, c, n, out, stream);
}
}
