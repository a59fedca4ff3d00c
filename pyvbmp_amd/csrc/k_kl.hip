// K15: KL(q || prior) of the conjugate families, one launch per family.  The evidence (ELBO) of every model sums these
// parameter-sized terms once per VB iteration; composed from torch element-wise kernels, lgamma / digamma calls and reductions
// they are 12-40 launches EACH (217 launches, 0.6 ms, in one iteration of the flocking DMBD: tools/exp/dmbd_launch_census.py).
//   Dirichlet  (ref dists/Dirichlet.py:73-86)        Gamma (ref dists/Gamma.py:66-72)
//   Wishart    (ref dists/Wishart.py:85-95)          + the Normal part of a NormalInverseWishart (ref dists/NormalInverseWishart.py:134-141)
//   the matrix-normal part of MatrixNormalWishart / MatrixNormalGamma (ref transforms/MatrixNormalWishart.py:206-215,
//   transforms/MatrixNormalGamma.py:203-214)
// One wave (block) per batch element; prior operands carry a batch stride (0 = one prior shared by the batch, the reference's
// expanded priors).  Sums in fp64 whatever the storage type.  gfx950 only.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vbmp_dispatch.h"
#include "../../include/vbmp_hip.h"

namespace vbmp {

__device__ __forceinline__ double kl_digamma_pos(double x) {  // x > 0: recurrence up to x >= 10, then the asymptotic series
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
__device__ __forceinline__ double kl_wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// KL = lgamma(sum a) - sum lgamma(a) - lgamma(sum a0) + sum lgamma(a0) + sum (a - a0) (psi(a) - psi(sum a)); the +/-inf of lgamma /
// digamma at a structural zero (masked transition entries) count as 0, as in the reference's KL_lgamma / KL_digamma
template <typename T>
__global__ __launch_bounds__(64) void k_dirichlet_kl(const T* __restrict__ alpha, const T* __restrict__ alpha0, int64_t s0, int K,
                                                     T* __restrict__ out) {
  const int64_t b = blockIdx.x;
  const int lane = threadIdx.x;
  const T* a = alpha + b * (int64_t)K;
  const T* a0 = alpha0 + b * s0;
  double sa = 0, sa0 = 0, sl = 0, sl0 = 0;
  for (int k = lane; k < K; k += 64) {
    const double x = (double)a[k], x0 = (double)a0[k];
    sa += x;
    sa0 += x0;
    sl += x > 0.0 ? lgamma(x) : 0.0;
    sl0 += x0 > 0.0 ? lgamma(x0) : 0.0;
  }
  sa = kl_wave_sum(sa), sa0 = kl_wave_sum(sa0), sl = kl_wave_sum(sl), sl0 = kl_wave_sum(sl0);
  const double pt = kl_digamma_pos(sa);
  double cr = 0;
  for (int k = lane; k < K; k += 64) {
    const double x = (double)a[k], x0 = (double)a0[k];
    cr += (x - x0) * ((x > 0.0 ? kl_digamma_pos(x) : 0.0) - pt);
  }
  cr = kl_wave_sum(cr);
  if (lane == 0) out[b] = (T)((lgamma(sa) - sl) - (lgamma(sa0) - sl0) + cr);
}

// sum over the K entries of an event of (a - a0) psi(a) - lgamma(a) + lgamma(a0) + a0 (log b - log b0) + a (b0 / b - 1)
template <typename T>
__global__ __launch_bounds__(64) void k_gamma_kl(const T* __restrict__ alpha, const T* __restrict__ beta, const T* __restrict__ alpha0,
                                                 const T* __restrict__ beta0, int64_t sa0, int64_t sb0, int K, T* __restrict__ out) {
  const int64_t b = blockIdx.x;
  const int lane = threadIdx.x;
  double acc = 0;
  for (int k = lane; k < K; k += 64) {
    const double a = (double)alpha[b * K + k], be = (double)beta[b * K + k];
    const double a0 = (double)alpha0[b * sa0 + k], b0 = (double)beta0[b * sb0 + k];
    acc += (a - a0) * kl_digamma_pos(a) - lgamma(a) + lgamma(a0) + a0 * (log(be) - log(b0)) + a * (b0 / be - 1.0);
  }
  acc = kl_wave_sum(acc);
  if (lane == 0) out[b] = (T)acc;
}

// Wishart: 1/2 nu0 (ld - ld0) + 1/2 nu (tr(invU0 U) - n) + sum_i [lgamma((nu0 - i)/2) - lgamma((nu - i)/2)] + 1/2 (nu - nu0) sum_i psi((nu - i)/2)
// [+ the Normal part of a NormalInverseWishart when mu is given: 1/2 n (lam0/lam - 1 + log(lam/lam0)) + 1/2 lam0 nu d' U d, d = mu - mu0]
template <typename T>
__global__ __launch_bounds__(64) void k_wishart_kl(const T* __restrict__ invU0, int64_t sm0, const T* __restrict__ U,
                                                   const T* __restrict__ nu, const T* __restrict__ nu0, int64_t sn0,
                                                   const T* __restrict__ ld, const T* __restrict__ ld0, int64_t sl0, int n,
                                                   const T* __restrict__ mu, const T* __restrict__ mu0, int64_t smu0,
                                                   const T* __restrict__ lam, const T* __restrict__ lam0, int64_t slam0,
                                                   T* __restrict__ out) {
  const int64_t b = blockIdx.x;
  const int lane = threadIdx.x;
  const T* Ub = U + b * (int64_t)n * n;
  const T* I0 = invU0 + b * sm0;
  const double v = (double)nu[b], v0 = (double)nu0[b * sn0];
  double tr = 0, quad = 0;
  for (int e = lane; e < n * n; e += 64) {
    tr += (double)I0[e] * (double)Ub[e];
    if (mu) {
      const int i = e / n, j = e - i * n;
      quad += ((double)mu[b * n + i] - (double)mu0[b * smu0 + i]) * (double)Ub[e] * ((double)mu[b * n + j] - (double)mu0[b * smu0 + j]);
    }
  }
  double gl = 0, ps = 0;
  for (int i = lane; i < n; i += 64) {
    gl += lgamma(0.5 * (v0 - i)) - lgamma(0.5 * (v - i));
    ps += kl_digamma_pos(0.5 * (v - i));
  }
  tr = kl_wave_sum(tr), gl = kl_wave_sum(gl), ps = kl_wave_sum(ps);
  if (mu) quad = kl_wave_sum(quad);
  if (lane == 0) {
    double kl = 0.5 * v0 * ((double)ld[b] - (double)ld0[b * sl0]) + 0.5 * v * (tr - n) + gl + 0.5 * (v - v0) * ps;
    if (mu) {
      const double l = (double)lam[b], l0 = (double)lam0[b * slam0];
      kl += 0.5 * n * (l0 / l - 1.0 + log(l / l0)) + 0.5 * l0 * v * quad;
    }
    out[b] = (T)kl;
  }
}

// the matrix-normal part of MatrixNormalWishart / MatrixNormalGamma.KLqprior for an (n x p) transform:
//   n/2 (ldV - ldV0) - n p / 2 [+ n/2 ldV0 xm] + n/2 tr(invV0 V) + 1/2 tr(invV0 d' R d),   d = mu - mu0, R = E[invSigma] (n x n)
// (xm: the number of set entries of X_mask per batch element, NULL without one).  LDS: E = d invV0 (n x p).
template <typename T>
__global__ __launch_bounds__(1024) void k_mn_kl(const T* __restrict__ mu, const T* __restrict__ mu0, int64_t smu0,
                                                const T* __restrict__ invV0, int64_t sv0, const T* __restrict__ V,
                                                const T* __restrict__ R, const T* __restrict__ ldV, const T* __restrict__ ldV0,
                                                int64_t sl0, const T* __restrict__ xm, int64_t sxm, int n, int p, T* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char kl_smem[];
  double* Dm = reinterpret_cast<double*>(kl_smem);  // d = mu - mu0 (n x p)
  double* Bm = Dm + n * p;                          // invV0 (p x p), then E = d invV0 (n x p)
  __shared__ double red[16];
  const int64_t b = blockIdx.x;
  const int tid = threadIdx.x;
  const T* m = mu + b * (int64_t)n * p;
  const T* m0 = mu0 + b * smu0;
  const T* I0 = invV0 + b * sv0;
  const T* Vb = V + b * (int64_t)p * p;
  const T* Rb = R + b * (int64_t)n * n;
  double part = 0;
  for (int e = tid; e < n * p; e += 1024) Dm[e] = (double)m[e] - (double)m0[e];
  for (int e = tid; e < p * p; e += 1024) {
    const double i0 = (double)I0[e];
    Bm[e] = i0;
    part += 0.5 * n * i0 * (double)Vb[e];
  }
  __syncthreads();
  constexpr int NE = 8;  // n p <= 8192: at most 8 entries of E per thread
  double ev[NE];
#pragma unroll
  for (int u = 0; u < NE; ++u) {
    const int e = tid + 1024 * u;
    double acc = 0;
    if (e < n * p) {
      const int i = e / p, c = e - i * p;
      for (int a = 0; a < p; ++a) acc += Dm[i * p + a] * Bm[a * p + c];
    }
    ev[u] = acc;
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < NE; ++u) {
    const int e = tid + 1024 * u;
    if (e < n * p) Bm[e] = ev[u];
  }
  __syncthreads();
  for (int e = tid; e < n * n; e += 1024) {  // sum_{i,k} R[i][k] (E d')[i][k]
    const int i = e / n, k = e - i * n;
    double acc = 0;
    for (int c = 0; c < p; ++c) acc += Bm[i * p + c] * Dm[k * p + c];
    part += 0.5 * (double)Rb[e] * acc;
  }
  part = kl_wave_sum(part);
  if ((tid & 63) == 0) red[tid >> 6] = part;
  __syncthreads();
  if (tid == 0) {
    const double l0 = (double)ldV0[b * sl0];
    double tot = 0;
    for (int w = 0; w < 16; ++w) tot += red[w];
    out[b] = (T)(tot + 0.5 * n * ((double)ldV[b] - l0) - 0.5 * n * p + 0.5 * n * l0 * (xm ? (double)xm[b * sxm] : 0.0));
  }
}

template <typename T>
static int launch_ok() { return hipGetLastError() == hipSuccess ? 0 : VBMP_ERR_LAUNCH; }

}  // namespace vbmp

using namespace vbmp;
#define VBMP_KL_ENTRIES(T, SUF)                                                                                                     \
  extern "C" int vbmp_dirichlet_kl_##SUF(const T* alpha, const T* alpha0, int64_t s0, int64_t NB, int K, T* out, void* stream) {    \
    if (NB == 0) return 0;                                                                                                          \
    if (!alpha || !alpha0 || !out || NB < 0 || NB > 0x7fffffff || K < 1) return VBMP_ERR_ARG;                                      \
    hipLaunchKernelGGL((k_dirichlet_kl<T>), dim3((unsigned)NB), dim3(64), 0, (hipStream_t)stream, alpha, alpha0, s0, K, out);       \
    return launch_ok<T>();                                                                                                          \
  }                                                                                                                                 \
  extern "C" int vbmp_gamma_kl_##SUF(const T* alpha, const T* beta, const T* alpha0, const T* beta0, int64_t sa0, int64_t sb0,      \
                                     int64_t NB, int K, T* out, void* stream) {                                                     \
    if (NB == 0) return 0;                                                                                                          \
    if (!alpha || !beta || !alpha0 || !beta0 || !out || NB < 0 || NB > 0x7fffffff || K < 1) return VBMP_ERR_ARG;                   \
    hipLaunchKernelGGL((k_gamma_kl<T>), dim3((unsigned)NB), dim3(64), 0, (hipStream_t)stream, alpha, beta, alpha0, beta0, sa0, sb0, \
                       K, out);                                                                                                     \
    return launch_ok<T>();                                                                                                          \
  }                                                                                                                                 \
  extern "C" int vbmp_wishart_kl_##SUF(const T* invU0, int64_t sm0, const T* U, const T* nu, const T* nu0, int64_t sn0,             \
                                       const T* ld, const T* ld0, int64_t sl0, const T* mu, const T* mu0, int64_t smu0,            \
                                       const T* lam, const T* lam0, int64_t slam0, int64_t NB, int n, T* out, void* stream) {       \
    if (NB == 0) return 0;                                                                                                          \
    if (!invU0 || !U || !nu || !nu0 || !ld || !ld0 || !out || NB < 0 || NB > 0x7fffffff || n < 1) return VBMP_ERR_ARG;             \
    if (mu && (!mu0 || !lam || !lam0)) return VBMP_ERR_ARG;                                                                         \
    hipLaunchKernelGGL((k_wishart_kl<T>), dim3((unsigned)NB), dim3(64), 0, (hipStream_t)stream, invU0, sm0, U, nu, nu0, sn0, ld,    \
                       ld0, sl0, n, mu, mu0, smu0, lam, lam0, slam0, out);                                                          \
    return launch_ok<T>();                                                                                                          \
  }                                                                                                                                 \
  extern "C" int vbmp_mn_kl_##SUF(const T* mu, const T* mu0, int64_t smu0, const T* invV0, int64_t sv0, const T* V, const T* R,     \
                                  const T* ldV, const T* ldV0, int64_t sl0, const T* xm, int64_t sxm, int64_t NB, int n, int p,   \
                                  T* out, void* stream) {                                                                                   \
    if (NB == 0) return 0;                                                                                                          \
    if (!mu || !mu0 || !invV0 || !V || !R || !ldV || !ldV0 || !out || NB < 0 || NB > 0x7fffffff || n < 1 || p < 1)                 \
      return VBMP_ERR_ARG;                                                                                                          \
    const size_t smem = ((size_t)n * p + (size_t)p * (n > p ? n : p)) * sizeof(double);                                             \
    if (smem > 64 * 1024 || (size_t)n * p > 8192) return VBMP_ERR_ARG; /* beyond it the caller composes the term */                \
    hipLaunchKernelGGL((k_mn_kl<T>), dim3((unsigned)NB), dim3(1024), smem, (hipStream_t)stream, mu, mu0, smu0, invV0, sv0, V, R,     \
                       ldV, ldV0, sl0, xm, sxm, n, p, out);                                                                              \
    return launch_ok<T>();                                                                                                          \
  }
VBMP_KL_ENTRIES(double, f64)
VBMP_KL_ENTRIES(float, f32)
