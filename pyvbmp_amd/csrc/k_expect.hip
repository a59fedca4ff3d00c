// K13: parameter expectations of a conjugate family in ONE launch.  The mixture E-step needs, per component k, the quadratic
// form  -1/2 x' P_k x + x' b_k + c_k = E log N(x | mu_k, Sigma_k) + E log pi_k  of a Normal-inverse-Wishart posterior
// (ref dists/NormalInverseWishart.py:91-97 with :107-132, dists/Wishart.py:67-83, dists/Dirichlet.py:52-53):
//     P_k = U_k nu_k,   b_k = P_k mu_k,
//     c_k = -1/2 (mu_k' P_k mu_k + D / lambda_k) + 1/2 (D log 2 - logdet_invU_k + sum_{i<D} psi((nu_k - i) / 2)) - D/2 log 2 pi
//           [+ psi(alpha_k) - psi(sum_j alpha_j)]
// Composed from the classes' getters this is ~25 launches of KB-sized torch kernels in front of a 0.17 ms E-step kernel; here one
// wave per component does all of it.  gfx950 only.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vbmp_dispatch.h"
#include "../../include/vbmp_hip.h"

namespace vbmp {

// digamma for x > 0: recurrence up to x >= 10, then the asymptotic series (next term < 4e-17 relative there)
__device__ __forceinline__ double digamma_pos(double x) {
  double r = 0.0;
  while (x < 10.0) {
    r -= 1.0 / x;
    x += 1.0;
  }
  const double f = 1.0 / (x * x);
  const double t = f * (-1.0 / 12.0 + f * (1.0 / 120.0 + f * (-1.0 / 252.0 + f * (1.0 / 240.0 + f * (-1.0 / 132.0 +
                   f * (691.0 / 32760.0 + f * (-1.0 / 12.0)))))));
  return r + log(x) - 0.5 / x + t;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <typename T>
__global__ __launch_bounds__(64) void k_niw_estep_params(const T* __restrict__ U, const T* __restrict__ nu, const T* __restrict__ mu,
                                                         const T* __restrict__ lam, const T* __restrict__ logdet_invU,
                                                         const T* __restrict__ alpha, int64_t K, int D, T* __restrict__ P,
                                                         T* __restrict__ b, T* __restrict__ c) {
  const int64_t k = blockIdx.x;
  const int lane = threadIdx.x;
  const T nuk = nu[k];
  const T* Uk = U + k * (int64_t)D * D;
  T* Pk = P + k * (int64_t)D * D;
  // P = U nu ;  b = P mu (lane r owns rows r, r + 64, ...) ;  q = mu' b
  double q = 0.0;
  for (int r = lane; r < D; r += 64) {
    T br = T(0);
    for (int j = 0; j < D; ++j) {
      const T pv = Uk[r * D + j] * nuk;
      Pk[r * D + j] = pv;
      br += pv * mu[k * D + j];
    }
    b[k * D + r] = br;
    q += (double)br * (double)mu[k * D + r];
  }
  q = wave_sum(q);
  // sum_i psi((nu - i) / 2), i < D
  double ps = 0.0;
  for (int i = lane; i < D; i += 64) ps += digamma_pos(0.5 * ((double)nuk - (double)i));
  ps = wave_sum(ps);
  // E log pi_k = psi(alpha_k) - psi(sum alpha)
  double lp = 0.0;
  if (alpha) {
    double tot = 0.0;
    for (int64_t j = lane; j < K; j += 64) tot += (double)alpha[j];
    tot = wave_sum(tot);
    lp = digamma_pos((double)alpha[k]) - digamma_pos(tot);
  }
  if (lane == 0) {
    const double LOG2 = 0.693147180559945309417232121458, LOG2PI = 1.8378770664093454835606594728112;
    const double ck = -0.5 * (q + (double)D / (double)lam[k]) + 0.5 * ((double)D * LOG2 - (double)logdet_invU[k] + ps) -
                      0.5 * (double)D * LOG2PI + lp;
    c[k] = (T)ck;
  }
}

// K14: the expectations of a MatrixNormalWishart posterior that its messages and likelihoods read
// (ref transforms/MatrixNormalWishart.py:419-471 with dists/Wishart.py:67-83), for NB batch elements of an (n x p) transform:
//     R = E[Sigma^-1] = U nu                    (n x n)        EinvSigma
//     G = R mu                                  (n x p)        EinvUX        (EXTinvU is its transpose)
//     H = n V + mu' R mu = n V + mu' G          (p x p)        EXTinvUX
//     El = sum_{i<n} psi((nu - i) / 2) + n log 2 - logdet_invU                ElogdetinvSigma
// One block per batch element; composed from the getters this is 14 launches wherever a caller asks for them.
template <typename T>
__global__ __launch_bounds__(256) void k_mnw_expectations(const T* __restrict__ mu, const T* __restrict__ U, const T* __restrict__ nu,
                                                          const T* __restrict__ V, const T* __restrict__ logdet_invU, int n, int p,
                                                          T* __restrict__ R, T* __restrict__ G, T* __restrict__ H, T* __restrict__ El) {
  const int64_t b = blockIdx.x;
  const int tid = threadIdx.x;
  const T nub = nu[b];
  const T* mub = mu + b * (int64_t)n * p;
  const T* Ub = U + b * (int64_t)n * n;
  const T* Vb = V + b * (int64_t)p * p;
  T* Rb = R + b * (int64_t)n * n;
  T* Gb = G + b * (int64_t)n * p;
  T* Hb = H + b * (int64_t)p * p;
  for (int e = tid; e < n * n; e += 256) Rb[e] = Ub[e] * nub;
  for (int e = tid; e < n * p; e += 256) {
    const int i = e / p, c = e - i * p;
    T acc = T(0);
    for (int k = 0; k < n; ++k) acc += (Ub[i * n + k] * nub) * mub[k * p + c];
    Gb[e] = acc;
  }
  __syncthreads();  // G of this block is complete (and visible to the block)
  for (int e = tid; e < p * p; e += 256) {
    const int a = e / p, c = e - a * p;
    T acc = T(0);
    for (int i = 0; i < n; ++i) acc += mub[i * p + a] * Gb[i * p + c];
    Hb[e] = (T)n * Vb[e] + acc;
  }
  if (tid < 64) {
    double ps = 0.0;
    for (int i = tid; i < n; i += 64) ps += digamma_pos(0.5 * ((double)nub - (double)i));
    ps = wave_sum(ps);
    if (tid == 0) El[b] = (T)(ps + (double)n * 0.693147180559945309417232121458 - (double)logdet_invU[b]);
  }
}

template <typename T>
static int mnw_expectations_dispatch(const T* mu, const T* U, const T* nu, const T* V, const T* logdet_invU, int64_t NB, int n, int p,
                                     T* R, T* G, T* H, T* El, void* stream) {
  if (NB == 0) return 0;
  if (!mu || !U || !nu || !V || !logdet_invU || !R || !G || !H || !El || NB < 0 || NB > 0x7fffffff || n < 1 || p < 1) return VBMP_ERR_ARG;
  hipLaunchKernelGGL((k_mnw_expectations<T>), dim3((unsigned)NB), dim3(256), 0, (hipStream_t)stream, mu, U, nu, V, logdet_invU, n, p, R, G,
                     H, El);
  return hipGetLastError() == hipSuccess ? 0 : VBMP_ERR_LAUNCH;
}

template <typename T>
static int niw_estep_params_dispatch(const T* U, const T* nu, const T* mu, const T* lam, const T* logdet_invU, const T* alpha,
                                     int64_t K, int D, T* P, T* b, T* c, void* stream) {
  if (K == 0) return 0;
  if (!U || !nu || !mu || !lam || !logdet_invU || !P || !b || !c || K < 0 || D < 1 || K > 0x7fffffff) return VBMP_ERR_ARG;
  hipLaunchKernelGGL((k_niw_estep_params<T>), dim3((unsigned)K), dim3(64), 0, (hipStream_t)stream, U, nu, mu, lam, logdet_invU,
                     alpha, K, D, P, b, c);
  return hipGetLastError() == hipSuccess ? 0 : VBMP_ERR_LAUNCH;
}

}  // namespace vbmp

extern "C" {
int vbmp_niw_estep_params_f64(const double* U, const double* nu, const double* mu, const double* lam, const double* logdet_invU,
                              const double* alpha, int64_t K, int D, double* P, double* b, double* c, void* stream) {
  return vbmp::niw_estep_params_dispatch<double>(U, nu, mu, lam, logdet_invU, alpha, K, D, P, b, c, stream);
}
int vbmp_mnw_expectations_f64(const double* mu, const double* U, const double* nu, const double* V, const double* logdet_invU,
                              int64_t NB, int n, int p, double* R, double* G, double* H, double* El, void* stream) {
  return vbmp::mnw_expectations_dispatch<double>(mu, U, nu, V, logdet_invU, NB, n, p, R, G, H, El, stream);
}
int vbmp_mnw_expectations_f32(const float* mu, const float* U, const float* nu, const float* V, const float* logdet_invU, int64_t NB,
                              int n, int p, float* R, float* G, float* H, float* El, void* stream) {
  return vbmp::mnw_expectations_dispatch<float>(mu, U, nu, V, logdet_invU, NB, n, p, R, G, H, El, stream);
}
int vbmp_niw_estep_params_f32(const float* U, const float* nu, const float* mu, const float* lam, const float* logdet_invU,
                              const float* alpha, int64_t K, int D, float* P, float* b, float* c, void* stream) {
  return vbmp::niw_estep_params_dispatch<float>(U, nu, mu, lam, logdet_invU, alpha, K, D, P, b, c, stream);
}
}
