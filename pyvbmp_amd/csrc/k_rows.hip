// K12: one small shared matrix applied to very many rows,  out[s, :] = M x[s, :] + c   (s < S; M n x k, k, n <= 64):
// the natural-parameter message eta_t = E[A' R] y_t of every observation in MatrixNormalWishart.Elog_like_X
// (transforms/MatrixNormalWishart.py:251-261), called by the LDS E-step on (T * series) rows.  The library GEMM for
// this shape (4 096 000 x 6 by 6 x 6, fp64) runs at 0.7 TB/s of HBM traffic; the product is pure streaming: a row per
// thread, M and c in LDS (every lane reads the same word: a broadcast), rows read and written once.  gfx950 only.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vbmp_dispatch.h"
#include "vbmp_flags.h"
extern "C" int g_vbmp_flags;
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

// K12 with the likelihood's scalar riding along: q[s] = -1/2 x' P x + b' x + c0 from the row the thread already holds (the
// observation message of the LDS E-step needs both, LinearDynamicalSystems.log_likelihood_function :244-266: one pass over
// the observations instead of K12 + K3a).  P (k x k), b (k), c0 (1 element, device memory).
template <typename T, int KP>
__global__ __launch_bounds__(256) void k_rows_affine_quad(const T* __restrict__ X, int64_t S, int k, const T* __restrict__ M,
                                                          const T* __restrict__ c, int n, T* __restrict__ out,
                                                          const T* __restrict__ P, const T* __restrict__ b,
                                                          const T* __restrict__ c0, T* __restrict__ q) {
  __shared__ T Ms[64 * KP + 64];
  __shared__ T Ps[KP * KP + KP];
  T* cs = Ms + 64 * KP;
  T* bs = Ps + KP * KP;
  for (int e = threadIdx.x; e < n * KP; e += 256) {
    const int j = e / KP, i = e - j * KP;
    Ms[e] = (i < k) ? M[j * k + i] : T(0);
  }
  for (int e = threadIdx.x; e < KP * KP; e += 256) {
    const int j = e / KP, i = e - j * KP;
    Ps[e] = (i < k && j < k) ? P[j * k + i] : T(0);
  }
  for (int j = threadIdx.x; j < n; j += 256) cs[j] = c ? c[j] : T(0);
  for (int j = threadIdx.x; j < KP; j += 256) bs[j] = (b && j < k) ? b[j] : T(0);
  const T cc = c0 ? c0[0] : T(0);
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
    T quad = T(0), lin = T(0);
#pragma unroll
    for (int j = 0; j < KP; ++j) {
      T t = T(0);
#pragma unroll
      for (int i = 0; i < KP; ++i) t = __builtin_fma(Ps[j * KP + i], x[i], t);
      quad = __builtin_fma(x[j], t, quad);
      lin = __builtin_fma(bs[j], x[j], lin);
    }
    q[s] = (T(-0.5) * quad + lin) + cc;
  }
}

// The same with the rows staged through LDS (round 3; taken for k > 8, see launch_rows_quad): a thread that reads ITS row straight
// from global memory issues k loads of one element each at a stride of k elements across the lanes, and stores likewise.  Here the block's 256 rows come in as ONE contiguous run, element e on lane e % 256
// (fully coalesced), are handed to their threads through an LDS image with an odd row stride (conflict-free), and the n outputs per
// row leave the same way.  Dynamic LDS: 256 x (k | 1) + 256 x (n | 1) elements.
template <typename T, int KP>
__global__ __launch_bounds__(256) void k_rows_affine_quad_staged(const T* __restrict__ X, int64_t S, int k, const T* __restrict__ M,
                                                                 const T* __restrict__ c, int n, T* __restrict__ out,
                                                                 const T* __restrict__ P, const T* __restrict__ b,
                                                                 const T* __restrict__ c0, T* __restrict__ q) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rows_smem[];
  __shared__ T Ms[64 * KP + 64];
  __shared__ T Ps[KP * KP + KP];
  T* cs = Ms + 64 * KP;
  T* bs = Ps + KP * KP;
  const int KS = k | 1, NS = n | 1;
  T* xin = reinterpret_cast<T*>(rows_smem);  // 256 x KS
  T* oout = xin + 256 * KS;                  // 256 x NS
  for (int e = threadIdx.x; e < n * KP; e += 256) {
    const int j = e / KP, i = e - j * KP;
    Ms[e] = (i < k) ? M[j * k + i] : T(0);
  }
  for (int e = threadIdx.x; e < KP * KP; e += 256) {
    const int j = e / KP, i = e - j * KP;
    Ps[e] = (i < k && j < k) ? P[j * k + i] : T(0);
  }
  for (int j = threadIdx.x; j < n; j += 256) cs[j] = c ? c[j] : T(0);
  for (int j = threadIdx.x; j < KP; j += 256) bs[j] = (b && j < k) ? b[j] : T(0);
  const T cc = c0 ? c0[0] : T(0);
  __syncthreads();
  for (int64_t s0 = (int64_t)blockIdx.x * 256; s0 < S; s0 += (int64_t)gridDim.x * 256) {
    const int rows = (int)((S - s0) < 256 ? (S - s0) : 256);
    const T* xb = X + s0 * k;
    for (int e = threadIdx.x; e < rows * k; e += 256) {
      const int r = e / k;
      xin[r * KS + (e - r * k)] = xb[e];
    }
    __syncthreads();
    if ((int)threadIdx.x < rows) {
      T x[KP];
#pragma unroll
      for (int i = 0; i < KP; ++i) x[i] = (i < k) ? xin[threadIdx.x * KS + i] : T(0);
      for (int j = 0; j < n; ++j) {
        T acc = cs[j];
#pragma unroll
        for (int i = 0; i < KP; ++i) acc = __builtin_fma(Ms[j * KP + i], x[i], acc);
        oout[threadIdx.x * NS + j] = acc;
      }
      T quad = T(0), lin = T(0);
#pragma unroll
      for (int j = 0; j < KP; ++j) {
        T t = T(0);
#pragma unroll
        for (int i = 0; i < KP; ++i) t = __builtin_fma(Ps[j * KP + i], x[i], t);
        quad = __builtin_fma(x[j], t, quad);
        lin = __builtin_fma(bs[j], x[j], lin);
      }
      q[s0 + threadIdx.x] = (T(-0.5) * quad + lin) + cc;
    }
    __syncthreads();
    T* ob = out + s0 * n;
    for (int e = threadIdx.x; e < rows * n; e += 256) {
      const int r = e / n;
      ob[e] = oout[r * NS + (e - r * n)];
    }
    // (the next chunk's loads into xin are ordered behind this chunk's reads of it by the barrier above; oout by the one below)
    __syncthreads();
  }
}

template <typename T, int KP>
static int launch_rows_quad(const T* X, int64_t S, int k, const T* M, const T* c, int n, T* out, const T* P, const T* b,
                            const T* c0, T* q, hipStream_t st) {
  int64_t blocks = (S + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  const size_t smem = (size_t)256 * ((k | 1) + (n | 1)) * sizeof(T);
  // rows through LDS for k > 8: measured 0.083 against 0.197 ms (fp64) / 0.080 against 0.463 ms (fp32) at 1e6 rows of k = 12, n = 6,
  // where a thread's own 12 strided loads crawl; at k <= 8 the row-per-thread form is as fast or faster (k = n = 6 fp64: 0.129
  // against 0.146 ms; tools/exp/rows_quad_ab.py), the strided rows are served from L1
  if (k > 8 && smem <= 40 * 1024 && S >= 4096 && !(g_vbmp_flags & VBMP_DBG_ROWS_DIRECT))
    hipLaunchKernelGGL((k_rows_affine_quad_staged<T, KP>), dim3((unsigned)blocks), dim3(256), smem, st, X, S, k, M, c, n, out, P, b,
                       c0, q);
  else
    hipLaunchKernelGGL((k_rows_affine_quad<T, KP>), dim3((unsigned)blocks), dim3(256), 0, st, X, S, k, M, c, n, out, P, b, c0, q);
  return hipGetLastError() == hipSuccess ? 0 : VBMP_ERR_LAUNCH;
}

template <typename T>
static int rows_quad_dispatch(const T* X, int64_t S, int k, const T* M, const T* c, int n, T* out, const T* P, const T* b,
                              const T* c0, T* q, void* stream) {
  if (S == 0) return 0;
  if (!X || !M || !out || !P || !q || S < 0 || k < 1 || n < 1 || k > VBMP_ROWS_QUAD_MAX_K || n > VBMP_ROWS_MAX_DIM)
    return VBMP_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (k <= 4) return launch_rows_quad<T, 4>(X, S, k, M, c, n, out, P, b, c0, q, st);
  if (k <= 8) return launch_rows_quad<T, 8>(X, S, k, M, c, n, out, P, b, c0, q, st);
  return launch_rows_quad<T, 16>(X, S, k, M, c, n, out, P, b, c0, q, st);
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
int vbmp_rows_affine_quad_f64(const double* X, int64_t S, int k, const double* M, const double* c, int n, double* out,
                              const double* P, const double* b, const double* c0, double* q, void* stream) {
  return vbmp::rows_quad_dispatch<double>(X, S, k, M, c, n, out, P, b, c0, q, stream);
}
int vbmp_rows_affine_quad_f32(const float* X, int64_t S, int k, const float* M, const float* c, int n, float* out,
                              const float* P, const float* b, const float* c0, float* q, void* stream) {
  return vbmp::rows_quad_dispatch<float>(X, S, k, M, c, n, out, P, b, c0, q, stream);
}
}
