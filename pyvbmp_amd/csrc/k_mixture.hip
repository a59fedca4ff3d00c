// K3 / K4: quadratic expected log-likelihood, fused mixture E-step, weighted moment reduction.
// gfx950 only.  C-ABI contract: include/vbmp_hip.h.
#include "vbmp_device.h"
#include "vbmp_dispatch.h"
#include "../../include/vbmp_hip.h"

namespace vbmp {

// one sample row in registers (zero padded to Dp)
template <typename T, int Dp>
__device__ __forceinline__ void load_row(const T* __restrict__ x, int D, T (&r)[Dp]) {
  constexpr int V = 16 / sizeof(T);
  if (D == Dp && Dp % V == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    using vec_t = T __attribute__((ext_vector_type(V)));
#pragma unroll
    for (int c = 0; c < Dp / V; ++c) {
      vec_t v = reinterpret_cast<const vec_t*>(x)[c];
#pragma unroll
      for (int u = 0; u < V; ++u) r[c * V + u] = v[u];
    }
  } else {
#pragma unroll
    for (int j = 0; j < Dp; ++j) r[j] = (j < D) ? x[j] : T(0);
  }
}

// -0.5 x^T P x + x^T b + c for one (x, component)
template <typename T, int Dp>
__device__ __forceinline__ T quadform(const T (&x)[Dp], int D, const T* __restrict__ P, const T* __restrict__ b, T c) {
  T acc = T(0);
#pragma unroll
  for (int i = 0; i < Dp; ++i) {
    if (i < D) {
      T row = T(0);
#pragma unroll
      for (int j = 0; j < Dp; ++j)
        if (j < D) row = xfma(P[i * D + j], x[j], row);
      acc = xfma(x[i], xfma(T(-0.5), row, b[i]), acc);
    }
  }
  return acc + c;
}

// ------------------------------------------------------------------------------------ K3a
// out[s, bo, bi] = -1/2 x^T P x + x^T b + c,  x = X[s, bi, :],  (P, b, c)[bo, bi]
template <typename T, int Dp>
__global__ __launch_bounds__(256) void k_quadform(const T* __restrict__ X, int64_t S, int64_t Bo, int64_t Bi, int D,
                                                  const T* __restrict__ P, const T* __restrict__ b,
                                                  const T* __restrict__ c, T* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over (s, bi)
  if (idx >= S * Bi) return;
  const int64_t s = idx / Bi, bi = idx - s * Bi;
  T x[Dp];
  load_row<T, Dp>(X + idx * D, D, x);
  for (int64_t bo = 0; bo < Bo; ++bo) {
    const int64_t comp = bo * Bi + bi;
    out[(s * Bo + bo) * Bi + bi] = quadform<T, Dp>(x, D, P + comp * D * D, b + comp * D, c[comp]);
  }
}

// ------------------------------------------------------------------------------------ K3
// Fused mixture E-step (ref dists/Mixture.py:38-45 with dists/NormalInverseWishart.py:91-97 inlined):
//   l[s,k] = quadform_k(x_s) (c_k already contains E log pi_k); lse_s = logsumexp_k l[s,k];
//   p[s,k] = exp(l[s,k] - lse_s); NA[k] += p[s,k]; logZ += lse_s.
// The p buffer doubles as the scratch for l.  NA / logZ must be zeroed by the caller.
template <typename T, int Dp>
__global__ __launch_bounds__(256) void k_mixture_estep(const T* __restrict__ X, int64_t S, int K, int D,
                                                       const T* __restrict__ P, const T* __restrict__ b,
                                                       const T* __restrict__ c, T* __restrict__ p,
                                                       T* __restrict__ NA, T* __restrict__ logZ) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* sNA = reinterpret_cast<T*>(smem_raw);  // K + 1 block partials
  for (int k = threadIdx.x; k <= K; k += 256) sNA[k] = T(0);
  __syncthreads();
  const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = s < S;
  T lse = T(0);
  if (live) {
    T x[Dp];
    load_row<T, Dp>(X + s * D, D, x);
    T* prow = p + s * K;
    T mx = -INFINITY, sum = T(0);
    for (int k = 0; k < K; ++k) {
      const T l = quadform<T, Dp>(x, D, P + (int64_t)k * D * D, b + (int64_t)k * D, c[k]);
      prow[k] = l;
      if (l > mx) {
        sum = sum * exp(mx - l) + T(1);
        mx = l;
      } else {
        sum += exp(l - mx);
      }
    }
    lse = mx + log(sum);
  }
  // second pass: normalise in place and reduce the responsibilities over the wave, then the block
  const int lane = threadIdx.x & 63;
  for (int k = 0; k < K; ++k) {
    T v = T(0);
    if (live) {
      T* prow = p + s * K;
      v = exp(prow[k] - lse);
      prow[k] = v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) atomicAdd(&sNA[k], v);
  }
  T z = live ? lse : T(0);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) z += __shfl_xor(z, off, 64);
  if (lane == 0) atomicAdd(&sNA[K], z);
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += 256) atomicAdd(&NA[k], sNA[k]);
  if (threadIdx.x == 0) atomicAdd(logZ, sNA[K]);
}

// ------------------------------------------------------------------------------------ K4
// Weighted moments (ref dists/NormalInverseWishart.py:72-84, dists/MultivariateNormal.py:93-99):
//   Nk[bo,bi] = sum_s w ; SEx[bo,bi,:] = sum_s w x ; SExx[bo,bi,:,:] = sum_s w x x^T,
//   x = X[s,bi,:], w = p[s,bo,bi] (1 when p == NULL).  Outputs must be zeroed by the caller.
// Block = 256 threads owns CH samples of one bi; thread t owns entries e = t, t+256, .. of the
// (D*D + D + 1)-long statistic vector [xx^T | x | 1] for every bo.
template <typename T>
__global__ __launch_bounds__(256) void k_weighted_moments(const T* __restrict__ X, const T* __restrict__ p, int64_t S,
                                                          int64_t Bo, int64_t Bi, int D, int CH, T* __restrict__ Nk,
                                                          T* __restrict__ SEx, T* __restrict__ SExx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* sx = reinterpret_cast<T*>(smem_raw);  // CH x (D+1), last column = 1
  T* sw = sx + CH * (D + 1);               // CH weights of the current bo
  const int64_t bi = blockIdx.y;
  const int64_t s0 = (int64_t)blockIdx.x * CH;
  const int ns = (int)((S - s0) < CH ? (S - s0) : CH);
  const int D1 = D + 1;
  for (int e = threadIdx.x; e < ns * D1; e += 256) {
    const int s = e / D1, j = e - s * D1;
    sx[e] = (j < D) ? X[((s0 + s) * Bi + bi) * D + j] : T(1);
  }
  const int nent = D1 * D1;  // augmented outer product [x;1][x;1]^T: xx^T, x and the count in one sweep
  for (int64_t bo = 0; bo < Bo; ++bo) {
    __syncthreads();
    for (int s = threadIdx.x; s < ns; s += 256) sw[s] = p ? p[((s0 + s) * Bo + bo) * Bi + bi] : T(1);
    __syncthreads();
    const int64_t comp = bo * Bi + bi;
    for (int e = threadIdx.x; e < nent; e += 256) {
      const int i = e / D1, j = e - i * D1;
      if (i == D && j < D) continue;  // mirror of the x column
      T acc = T(0);
      for (int s = 0; s < ns; ++s) acc = xfma(sw[s] * sx[s * D1 + i], sx[s * D1 + j], acc);
      if (i < D && j < D)
        atomicAdd(&SExx[(comp * D + i) * D + j], acc);
      else if (i < D)
        atomicAdd(&SEx[comp * D + i], acc);
      else
        atomicAdd(&Nk[comp], acc);
    }
  }
}

// ------------------------------------------------------------------------------------ K4 on the matrix cores
// fp32 weighted second moments  SExx[k] = sum_s w[s,k] x_s x_s^T  as an MFMA contraction over the SAMPLE axis
// (the one place on this path where the contraction is genuinely dense; north star).  D <= 64, Bi == 1, Bo <= 4.
// v_mfma_f32_32x32x2_f32 consumes two samples per instruction: A[i][k] = w_s x_s[i], B[k][j] = x_s[j] with
// k = lane>>5 selecting the sample and lane&31 the feature, so every load is two 128-byte row segments.
// The 64 x 64 result is a 2 x 2 grid of 32 x 32 accumulator tiles per component; sum w x and sum w ride along
// on the VALU.  Each wave owns a contiguous slab of samples and adds its partial with float atomics
// (2 x 128-byte segments per atomic instruction: the layout the memory-side atomics run at full rate for).
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NT, int BO>  // NT = feature tiles of 32 (1 or 2), BO = components
__global__ __launch_bounds__(256) void k_wmom_mfma_f32(const float* __restrict__ X, const float* __restrict__ p,
                                                        int64_t S, int D, float* __restrict__ Nk,
                                                        float* __restrict__ SEx, float* __restrict__ SExx) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, f = lane & 31;
  const int64_t nw = (int64_t)gridDim.x * 4, w = (int64_t)blockIdx.x * 4 + wave;
  // contiguous slab of sample pairs per wave
  const int64_t pairs = (S + 1) / 2, per = (pairs + nw - 1) / nw;
  const int64_t p0 = w * per, p1 = (p0 + per < pairs) ? p0 + per : pairs;
  f32x16 acc[BO][NT][NT];
  float sx[BO][NT], sn[BO];
#pragma unroll
  for (int k = 0; k < BO; ++k) {
    sn[k] = 0.f;
#pragma unroll
    for (int a = 0; a < NT; ++a) {
      sx[k][a] = 0.f;
#pragma unroll
      for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][a][b][r] = 0.f;
    }
  }
#pragma unroll 4
  for (int64_t pr = p0; pr < p1; ++pr) {
    const int64_t s = 2 * pr + half;
    const bool ok = s < S;
    float x[NT];
#pragma unroll
    for (int a = 0; a < NT; ++a) x[a] = (ok && 32 * a + f < D) ? X[s * D + 32 * a + f] : 0.f;
#pragma unroll
    for (int k = 0; k < BO; ++k) {
      const float wk = ok ? (p ? p[s * BO + k] : 1.f) : 0.f;
      sn[k] += wk;
      float wx[NT];
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        wx[a] = wk * x[a];
        sx[k][a] += wx[a];
      }
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[k][a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(wx[a], x[b], acc[k][a][b], 0, 0, 0);
    }
  }
  // The four waves of the block first combine in LDS (C layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)),
  // so that one set of global float atomics leaves per BLOCK, not per wave (the chip-wide atomic rate is ~1.3 TB/s).
  constexpr int DT = 32 * NT;
  __shared__ float red[BO * (DT * DT + DT + 1)];
  for (int e = threadIdx.x; e < BO * (DT * DT + DT + 1); e += 256) red[e] = 0.f;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < BO; ++k) {
    float* rk = red + k * (DT * DT + DT + 1);
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = 32 * a + (r & 3) + 8 * (r >> 2) + 4 * half, j = 32 * b + f;
          atomicAdd(&rk[i * DT + j], acc[k][a][b][r]);
        }
#pragma unroll
    for (int a = 0; a < NT; ++a) {
      const float t = sx[k][a] + __shfl_xor(sx[k][a], 32, 64);
      if (half == 0) atomicAdd(&rk[DT * DT + 32 * a + f], t);
    }
    const float tn = sn[k] + __shfl_xor(sn[k], 32, 64);
    if (lane == 0) atomicAdd(&rk[DT * DT + DT], tn);
  }
  __syncthreads();
  for (int k = 0; k < BO; ++k) {
    const float* rk = red + k * (DT * DT + DT + 1);
    for (int e = threadIdx.x; e < DT * DT; e += 256) {
      const int i = e / DT, j = e % DT;
      if (i < D && j < D) atomicAdd(&SExx[((int64_t)k * D + i) * D + j], rk[e]);
    }
    for (int e = threadIdx.x; e < DT; e += 256)
      if (e < D) atomicAdd(&SEx[(int64_t)k * D + e], rk[DT * DT + e]);
    if (threadIdx.x == 0) atomicAdd(&Nk[k], rk[DT * DT + DT]);
  }
}

static int wmom_mfma_f32(const float* X, const float* p, int64_t S, int64_t Bo, int D, float* Nk, float* SEx, float* SExx,
                         hipStream_t st) {
  int64_t pairs = (S + 1) / 2;
  int64_t blocks = (pairs + 4 * 64 - 1) / (4 * 64);  // >= 64 sample pairs per wave
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  const dim3 g((unsigned)blocks), b(256);
#define VBMP_WM(NT, BO) hipLaunchKernelGGL((k_wmom_mfma_f32<NT, BO>), g, b, 0, st, X, p, S, D, Nk, SEx, SExx)
  const int nt = D <= 32 ? 1 : 2;
  if (nt == 1) {
    if (Bo == 1) VBMP_WM(1, 1); else if (Bo == 2) VBMP_WM(1, 2); else if (Bo == 3) VBMP_WM(1, 3); else VBMP_WM(1, 4);
  } else {
    if (Bo == 1) VBMP_WM(2, 1); else if (Bo == 2) VBMP_WM(2, 2); else if (Bo == 3) VBMP_WM(2, 3); else VBMP_WM(2, 4);
  }
#undef VBMP_WM
  return hipGetLastError() == hipSuccess ? 0 : VBMP_ERR_LAUNCH;
}

template <typename T>
static int quadform_dispatch(const T* X, int64_t S, int64_t Bo, int64_t Bi, int D, const T* P, const T* b, const T* c,
                             T* out, void* stream) {
  if (S == 0 || Bo == 0 || Bi == 0) return 0;
  if (!X || !P || !b || !c || !out || S < 0 || Bo < 0 || Bi < 0 || D < 1 || D > VBMP_MAX_DIM) return VBMP_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int64_t blocks = (S * Bi + 255) / 256;
  VBMP_DISPATCH_DIM(T, D, {
    hipLaunchKernelGGL((k_quadform<T, DP>), dim3((unsigned)blocks), dim3(256), 0, st, X, S, Bo, Bi, D, P, b, c, out);
    return hipGetLastError() == hipSuccess ? 0 : VBMP_ERR_LAUNCH;
  });
  return VBMP_ERR_ARG;
}

template <typename T>
static int estep_dispatch(const T* X, int64_t S, int K, int D, const T* P, const T* b, const T* c, T* p, T* NA,
                          T* logZ, void* stream) {
  if (S == 0) return 0;
  if (!X || !P || !b || !c || !p || !NA || !logZ || S < 0 || K < 1 || D < 1 || D > VBMP_MAX_DIM) return VBMP_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int64_t blocks = (S + 255) / 256;
  const size_t smem = (size_t)(K + 1) * sizeof(T);
  VBMP_DISPATCH_DIM(T, D, {
    hipLaunchKernelGGL((k_mixture_estep<T, DP>), dim3((unsigned)blocks), dim3(256), smem, st, X, S, K, D, P, b, c, p,
                       NA, logZ);
    return hipGetLastError() == hipSuccess ? 0 : VBMP_ERR_LAUNCH;
  });
  return VBMP_ERR_ARG;
}

template <typename T>
static int wmom_dispatch(const T* X, const T* p, int64_t S, int64_t Bo, int64_t Bi, int D, T* Nk, T* SEx, T* SExx,
                         void* stream) {
  if (S == 0 || Bo == 0 || Bi == 0) return 0;
  if (!X || !Nk || !SEx || !SExx || S < 0 || Bo < 0 || Bi < 0 || D < 1 || D > 2 * VBMP_MAX_DIM || Bi > 65535)
    return VBMP_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if constexpr (sizeof(T) == 4) {
    // dense contraction over many samples: matrix cores (see k_wmom_mfma_f32)
    if (Bi == 1 && Bo <= 4 && D <= 64 && S >= 4096)
      return wmom_mfma_f32(reinterpret_cast<const float*>(X), reinterpret_cast<const float*>(p), S, Bo, D,
                           reinterpret_cast<float*>(Nk), reinterpret_cast<float*>(SEx), reinterpret_cast<float*>(SExx), st);
  }
  // samples per block: as many as fit a 48 KiB LDS image of [x | 1 | w] rows (at most 256)
  int CH = (int)((48 * 1024) / ((size_t)(D + 2) * sizeof(T)));
  CH = CH > 256 ? 256 : CH;
  const int64_t bx = (S + CH - 1) / CH;
  const size_t smem = (size_t)(CH * (D + 1) + CH) * sizeof(T);
  hipLaunchKernelGGL((k_weighted_moments<T>), dim3((unsigned)bx, (unsigned)Bi), dim3(256), smem, st, X, p, S, Bo, Bi, D,
                     CH, Nk, SEx, SExx);
  return hipGetLastError() == hipSuccess ? 0 : VBMP_ERR_LAUNCH;
}

}  // namespace vbmp

using namespace vbmp;

extern "C" {
#define VBMP_DEF_MIX(SUF, T)                                                                                         \
  int vbmp_quadform_loglike_##SUF(const T* X, int64_t S, int64_t Bo, int64_t Bi, int D, const T* P, const T* b,      \
                                  const T* c, T* out, void* stream) {                                                \
    return quadform_dispatch<T>(X, S, Bo, Bi, D, P, b, c, out, stream);                                              \
  }                                                                                                                  \
  int vbmp_mixture_estep_##SUF(const T* X, int64_t S, int K, int D, const T* P, const T* b, const T* c, T* p, T* NA, \
                               T* logZ, void* stream) {                                                              \
    return estep_dispatch<T>(X, S, K, D, P, b, c, p, NA, logZ, stream);                                              \
  }                                                                                                                  \
  int vbmp_weighted_moments_##SUF(const T* X, const T* p, int64_t S, int64_t Bo, int64_t Bi, int D, T* Nk, T* SEx,   \
                                  T* SExx, void* stream) {                                                           \
    return wmom_dispatch<T>(X, p, S, Bo, Bi, D, Nk, SEx, SExx, stream);                                              \
  }
VBMP_DEF_MIX(f64, double)
VBMP_DEF_MIX(f32, float)
}  // extern "C"
