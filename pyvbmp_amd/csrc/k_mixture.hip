// K3 / K4: quadratic expected log-likelihood, fused mixture E-step, weighted moment reduction.
// gfx950 only.  C-ABI contract: include/vbmp_hip.h.
#include "vbmp_device.h"
#include "vbmp_dispatch.h"
#include "../../include/vbmp_hip.h"

#include "vbmp_flags.h"
extern "C" int g_vbmp_flags;
extern "C" int g_vbmp_blocks_per_cu;

namespace vbmp {

// e / K by multiply-high with kinv = floor((2^32 - 1) / K) + 1: exact for e < 2^16 and 2 <= K; for K == 1 the
// reciprocal does not fit 32 bits (kinv wraps to 0), so that case is the identity
__device__ __forceinline__ int div_small(int e, unsigned int kinv, int K) {
  return K == 1 ? e : (int)__umulhi((unsigned int)e, kinv);
}


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

// -0.5 x^T P x + x^T b + c for one (x, component).  FULL (D == Dp): compile-time offsets, so a wave-uniform P
// arrives through wide scalar loads; otherwise row guards only, columns clamped in-bounds against the zero padding of x.
template <typename T, int Dp, bool FULL>
__device__ __forceinline__ T quadform(const T (&x)[Dp], int D, const T* __restrict__ P, const T* __restrict__ b, T c) {
  T acc = T(0);
  if constexpr (FULL) {
#pragma unroll
    for (int i = 0; i < Dp; ++i) {
      T row = T(0);
#pragma unroll
      for (int j = 0; j < Dp; ++j) row = xfma(P[i * Dp + j], x[j], row);
      acc = xfma(x[i], xfma(T(-0.5), row, b[i]), acc);
    }
  } else {
#pragma unroll
    for (int i = 0; i < Dp; ++i) {
      if (i < D) {
        const T* Pi = P + i * D;
        T row = T(0);
#pragma unroll
        for (int j = 0; j < Dp; ++j) row = xfma(Pi[j < D ? j : D - 1], x[j], row);
        acc = xfma(x[i], xfma(T(-0.5), row, b[i]), acc);
      }
    }
  }
  return acc + c;
}

// ------------------------------------------------------------------------------------ K3a
// out[s, bo, bi] = -1/2 x^T P x + x^T b + c,  x = X[s, bi, :],  (P, b, c)[bo, bi]
template <typename T, int Dp, bool FULL>
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
    out[(s * Bo + bo) * Bi + bi] = quadform<T, Dp, FULL>(x, D, P + comp * D * D, b + comp * D, c[comp]);
  }
}

// ------------------------------------------------------------------------------------ K3
// Fused mixture E-step (ref dists/Mixture.py:38-45 with dists/NormalInverseWishart.py:91-97 inlined):
//   l[s,k] = quadform_k(x_s) (c_k already contains E log pi_k); lse_s = logsumexp_k l[s,k];
//   p[s,k] = exp(l[s,k] - lse_s); NA[k] += p[s,k]; logZ += lse_s.
// The p buffer doubles as the scratch for l.  NA / logZ must be zeroed by the caller.
template <typename T, int Dp, bool FULL>
__global__ __launch_bounds__(256) void k_mixture_estep(const T* __restrict__ X, int64_t S, int K, int D,
                                                       const T* __restrict__ P, const T* __restrict__ b,
                                                       const T* __restrict__ c, T* __restrict__ p,
                                                       T* __restrict__ NA, T* __restrict__ logZ, int stage) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* sNA = reinterpret_cast<T*>(smem_raw);  // K + 1 block partials
  // stage != 0: the block's 256 x K responsibilities live in LDS (row stride K+1) between the two passes and leave
  // as one contiguous coalesced store; otherwise the p buffer doubles as the scratch for l.
  T* sL = sNA + (K + 1);
  const int KS = K + 1;
  for (int k = threadIdx.x; k <= K; k += 256) sNA[k] = T(0);
  if (stage)
    for (int k = 0; k <= K; ++k) sL[256 * KS + threadIdx.x * KS + k] = T(0);  // this thread's running sums
  const unsigned int kinv = 0xFFFFFFFFu / (unsigned int)K + 1u;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  // grid-stride over samples: the same-address global atomics at the end are per block, so the grid is capped
  for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s - threadIdx.x < S; s += (int64_t)gridDim.x * 256) {
    const bool live = s < S;
    T* lrow = stage ? sL + threadIdx.x * KS : p + s * K;
    T lse = T(0), inv = T(0);
    if (live) {
      T x[Dp];
      load_row<T, Dp>(X + s * D, D, x);
      // one exp per (sample, component): l_k, then e_k = exp(l_k - max), then p_k = e_k / sum  (the fp64 exp is ~50
      // instructions; an online log-sum-exp plus exp(l - lse) costs 2K + 1 of them per sample, this K + 1)
      T mx = -INFINITY;
      for (int k = 0; k < K; ++k) {
        const T l = quadform<T, Dp, FULL>(x, D, P + (int64_t)k * D * D, b + (int64_t)k * D, c[k]);
        if (stage) sL[threadIdx.x * KS + k] = l; else lrow[k] = l;
        mx = l > mx ? l : mx;
      }
      T sum = T(0);
      for (int k = 0; k < K; ++k) {
        const T e = exp((stage ? sL[threadIdx.x * KS + k] : lrow[k]) - mx);
        if (stage) sL[threadIdx.x * KS + k] = e; else lrow[k] = e;
        sum += e;
      }
      lse = mx + log(sum);
      inv = T(1) / sum;
    }
    // last pass: normalise in place and sum the responsibilities
    if (stage) {
      // staged form: every thread keeps running sums in its own LDS slots (one block-level reduction at the very end
      // instead of K + 1 wave butterflies per 256 samples)
      T* acc = sL + 256 * KS + threadIdx.x * KS;
      if (live) {
        for (int k = 0; k < K; ++k) {
          const T v = sL[threadIdx.x * KS + k] * inv;
          sL[threadIdx.x * KS + k] = v;
          acc[k] += v;
        }
        acc[K] += lse;
      }
      __syncthreads();
      const int64_t s0 = s - threadIdx.x;
      const int64_t n = ((S - s0) < 256 ? (S - s0) : 256) * K;
      T* dst = p + s0 * K;
      for (int e = threadIdx.x; e < n; e += 256) {
        const int r = div_small(e, kinv, K), k = e - r * K;  // e / K, exact for e < 2^16
        dst[e] = sL[r * KS + k];
      }
      __syncthreads();
    } else {
      for (int k = 0; k < K; ++k) {
        T v = T(0);
        if (live) {
          v = lrow[k] * inv;
          lrow[k] = v;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) atomicAdd(&sNA[k], v);
      }
      T z = live ? lse : T(0);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) z += __shfl_xor(z, off, 64);
      if (lane == 0) atomicAdd(&sNA[K], z);
    }
  }
  __syncthreads();
  if (stage) {
    for (int k = threadIdx.x; k <= K; k += 256) {
      T tot = T(0);
      for (int r = 0; r < 256; ++r) tot += sL[256 * KS + r * KS + k];
      if (k < K)
        atomicAdd(&NA[k], tot);
      else
        atomicAdd(logZ, tot);
    }
  } else {
    for (int k = threadIdx.x; k < K; k += 256) atomicAdd(&NA[k], sNA[k]);
    if (threadIdx.x == 0) atomicAdd(logZ, sNA[K]);
  }
}

// ------------------------------------------------------------------------------------ K3, symmetric-packed VALU form
// x' P x = sum_i x_i (Q_ii x_i + sum_{j>i} Q_ij x_j) with Q_ii = P_ii, Q_ij = P_ij + P_ji (exact for any P): D(D+1)/2 + D FMAs per
// (sample, component) instead of D^2 + D, the packed Q rows wave-uniform, i.e. scalar operands of the FMAs fetched through the
// scalar cache.  The plain VALU form above is bound by that cache (8 bytes of P per FMA of every SIMD: measured 2.5x its FMA
// time at K = 4, D = 16); here a thread owns NS samples, so one scalar fetch feeds NS FMAs, and the packing halves the fetches.
// Q: (K, Dp (Dp + 1) / 2) rows i = 0.. with entries j = i..Dp-1; b (K, Dp); D == Dp.  Softmax, NA / logZ as in k_mixture_estep
// (staged form).  Dynamic LDS: (K + 1) block partials | 256 NS rows of K + 1 | 256 rows of K + 1 running sums.
template <typename T, int Dp, int NS>
__global__ __launch_bounds__(256) void k_estep_sym(const T* __restrict__ X, int64_t S, int K, const T* __restrict__ Q,
                                                   const T* __restrict__ b, const T* __restrict__ c, T* __restrict__ p,
                                                   T* __restrict__ NA, T* __restrict__ logZ, T* __restrict__ lse_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int TRI = Dp * (Dp + 1) / 2;
  constexpr int ROWS = 256 * NS;
  T* sNA = reinterpret_cast<T*>(smem_raw);
  const int KS = K + 1;
  T* sL = sNA + KS;            // ROWS x KS: log-likelihoods, then responsibilities of the chunk
  T* sAcc = sL + ROWS * KS;    // 256 x KS: this thread's running sums
  for (int k = 0; k <= K; ++k) sAcc[threadIdx.x * KS + k] = T(0);
  const unsigned int kinv = 0xFFFFFFFFu / (unsigned int)K + 1u;
  __syncthreads();
  for (int64_t s0 = (int64_t)blockIdx.x * ROWS; s0 < S; s0 += (int64_t)gridDim.x * ROWS) {
    T x[NS][Dp];
    bool live[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int64_t sidx = s0 + threadIdx.x + 256 * u;
      live[u] = sidx < S;
      load_row<T, Dp>(X + (live[u] ? sidx : s0) * Dp, Dp, x[u]);
    }
    for (int k = 0; k < K; ++k) {
      const T* __restrict__ Qk = Q + (int64_t)k * TRI;
      const T* __restrict__ bk = b + (int64_t)k * Dp;
      T l[NS];
      const T ck = c[k];
#pragma unroll
      for (int u = 0; u < NS; ++u) l[u] = T(0);
      int q = 0;
#pragma unroll
      for (int i = 0; i < Dp; ++i) {
        T row[NS];
#pragma unroll
        for (int u = 0; u < NS; ++u) row[u] = T(0);
#pragma unroll
        for (int j = i; j < Dp; ++j) {
          const T qij = Qk[q++];
#pragma unroll
          for (int u = 0; u < NS; ++u) row[u] = xfma(qij, x[u][j], row[u]);
        }
        const T bi = bk[i];
#pragma unroll
        for (int u = 0; u < NS; ++u) l[u] = xfma(x[u][i], xfma(T(-0.5), row[u], bi), l[u]);
      }
#pragma unroll
      for (int u = 0; u < NS; ++u) sL[(threadIdx.x + 256 * u) * KS + k] = l[u] + ck;
    }
    // softmax of my rows (only this thread touches them), running sums in my LDS slots
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      if (live[u]) {
        T* row = sL + (threadIdx.x + 256 * u) * KS;
        T mx = -INFINITY;
        for (int k = 0; k < K; ++k) mx = row[k] > mx ? row[k] : mx;
        T sum = T(0);
        for (int k = 0; k < K; ++k) {
          const T e = exp(row[k] - mx);
          row[k] = e;
          sum += e;
        }
        const T inv = T(1) / sum;
        T* acc = sAcc + threadIdx.x * KS;
        for (int k = 0; k < K; ++k) {
          const T v = row[k] * inv;
          row[k] = v;
          acc[k] += v;
        }
        const T lse = mx + log(sum);
        acc[K] += lse;
        if (lse_out) lse_out[s0 + threadIdx.x + 256 * u] = lse;  // per-sample evidence (mixtures of experts keep it)
      }
    }
    __syncthreads();
    const int64_t nrow = (S - s0) < ROWS ? (S - s0) : ROWS;
    const int n = (int)nrow * K;
    T* dst = p + s0 * K;
    for (int e = threadIdx.x; e < n; e += 256) {
      const int r = div_small(e, kinv, K), k = e - r * K;  // e / K, exact for e < 2^16
      dst[e] = sL[r * KS + k];
    }
    __syncthreads();
  }
  for (int k = threadIdx.x; k <= K; k += 256) {
    T tot = T(0);
    for (int r = 0; r < 256; ++r) tot += sAcc[r * KS + k];
    if (k < K)
      atomicAdd(&NA[k], tot);
    else
      atomicAdd(logZ, tot);
  }
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
// Weighted second moments  SExx[k] = sum_s w[s,k] x_s x_s^T  as an MFMA contraction over the SAMPLE axis (the one
// place on this path where the contraction is genuinely dense).  A = (w x)^T, B = x, both ONE value per lane:
//   16-tile forms (v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32): lane -> (sample slot g = lane>>4, feature f = lane&15)
//   32-tile form  (v_mfma_f32_32x32x2_f32):                          lane -> (g = lane>>5, f = lane&31)
// so a wave consumes 64/TILE consecutive samples per instruction and the X loads are one contiguous run per wave.
// Loads of U steps are issued branch-free ahead of the MFMAs (clamped addresses, zeroed weights past the end).
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <typename T, int TILE> struct MomMfma;
template <> struct MomMfma<double, 16> {
  using acc_t = f64x4;
  static constexpr int R = 4;
  static __device__ __forceinline__ acc_t run(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int r, int g) { return g + 4 * r; }
};
template <> struct MomMfma<float, 16> {
  using acc_t = f32x4;
  static constexpr int R = 4;
  static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int r, int g) { return 4 * g + r; }
};
template <> struct MomMfma<float, 32> {
  using acc_t = f32x16;
  static constexpr int R = 16;
  static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int r, int g) { return (r & 3) + 8 * (r >> 2) + 4 * g; }
};

template <typename T, int TILE, int NT, int BO>  // NT feature tiles of TILE, BO <= 4 components per pass
__global__ __launch_bounds__(256) void k_wmom_mfma(const T* __restrict__ X, const T* __restrict__ p_all, int64_t S, int D,
                                                    int ps, T* __restrict__ Nk_all, T* __restrict__ SEx_all,
                                                    T* __restrict__ SExx_all) {
  // blockIdx.y = group of BO components: the passes over X for the groups of a wide statistic run side by side
  // instead of as one launch after the other (25 roles of a 57-wide moment: 25 launches of 24 blocks each)
  const int64_t k0 = (int64_t)blockIdx.y * BO;
  const T* __restrict__ p = p_all ? p_all + k0 : p_all;
  T* __restrict__ Nk = Nk_all + k0;
  T* __restrict__ SEx = SEx_all + k0 * D;
  T* __restrict__ SExx = SExx_all + k0 * D * D;
  using M = MomMfma<T, TILE>;
  using acc_t = typename M::acc_t;
  constexpr int SPS = 64 / TILE, U = 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / TILE, f = lane % TILE;
  const int64_t nw = (int64_t)gridDim.x * 4, w = (int64_t)blockIdx.x * 4 + wave;
  // contiguous slab of steps per wave
  const int64_t steps = (S + SPS - 1) / SPS, per = (steps + nw - 1) / nw;
  const int64_t t0 = w * per, t1 = (t0 + per < steps) ? t0 + per : steps;
  acc_t acc[BO][NT][NT];
  T sx[BO][NT], sn[BO];
#pragma unroll
  for (int k = 0; k < BO; ++k) {
    sn[k] = T(0);
#pragma unroll
    for (int a = 0; a < NT; ++a) {
      sx[k][a] = T(0);
#pragma unroll
      for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int r = 0; r < M::R; ++r) acc[k][a][b][r] = T(0);
    }
  }
  // every 16-lane DPP row loads the BO weights of its sample in its first lanes; row_newbcast hands them out
  const T* pp = p ? p : X;
  const int pst = p ? ps : 0;
  const int fk = p ? ((f & 15) < BO ? (f & 15) : BO - 1) : 0;
  for (int64_t t = t0; t < t1; t += U) {
    T x[U][NT], wp[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t s = SPS * (t + u) + g;
      ok[u] = (t + u < t1) && (s < S);
      const int64_t sc = ok[u] ? s : 0;
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        const int col = TILE * a + f;
        x[u][a] = X[sc * D + (col < D ? col : 0)];
      }
      wp[u] = pp[sc * pst + fk];
    }
    // all (always in-bounds) loads are in flight before the first use: the empty asm pins them outside the selects
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int a = 0; a < NT; ++a) asm volatile("" : "+v"(x[u][a]));
      asm volatile("" : "+v"(wp[u]));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int a = 0; a < NT; ++a) x[u][a] = (ok[u] && TILE * a + f < D) ? x[u][a] : T(0);
      wp[u] = ok[u] ? (p ? wp[u] : T(1)) : T(0);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      static_for<0, BO>([&](auto KC) {
        constexpr int k = decltype(KC)::value;
        const T wk = bcast<16, k>(wp[u]);
        sn[k] += wk;
        T wx[NT];
#pragma unroll
        for (int a = 0; a < NT; ++a) {
          wx[a] = wk * x[u][a];
          sx[k][a] += wx[a];
        }
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
          for (int b = 0; b < NT; ++b) acc[k][a][b] = M::run(wx[a], x[u][b], acc[k][a][b]);
      });
    }
  }
  // The four waves of the block first combine in LDS, so that one set of global atomics leaves per BLOCK, not per wave
  // (same-address float atomics serialise in L2; the chip-wide atomic rate is ~1.3 TB/s).
  constexpr int DT = TILE * NT;
  __shared__ T red[BO * (DT * DT + DT + 1)];
  for (int e = threadIdx.x; e < BO * (DT * DT + DT + 1); e += 256) red[e] = T(0);
  __syncthreads();
#pragma unroll
  for (int k = 0; k < BO; ++k) {
    T* rk = red + k * (DT * DT + DT + 1);
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int r = 0; r < M::R; ++r)
          atomicAdd(&rk[(TILE * a + M::row(r, g)) * DT + TILE * b + f], acc[k][a][b][r]);
#pragma unroll
    for (int a = 0; a < NT; ++a) {
      T tx = sx[k][a];
#pragma unroll
      for (int off = TILE; off < 64; off <<= 1) tx += __shfl_xor(tx, off, 64);
      if (g == 0) atomicAdd(&rk[DT * DT + TILE * a + f], tx);
    }
    T tn = sn[k];
#pragma unroll
    for (int off = TILE; off < 64; off <<= 1) tn += __shfl_xor(tn, off, 64);
    if (lane == 0) atomicAdd(&rk[DT * DT + DT], tn);
  }
  __syncthreads();
  for (int k = 0; k < BO; ++k) {
    const T* rk = red + k * (DT * DT + DT + 1);
    for (int e = threadIdx.x; e < DT * DT; e += 256) {
      const int i = e / DT, j = e % DT;
      if (i < D && j < D) atomicAdd(&SExx[((int64_t)k * D + i) * D + j], rk[e]);
    }
    for (int e = threadIdx.x; e < DT; e += 256)
      if (e < D) atomicAdd(&SEx[(int64_t)k * D + e], rk[DT * DT + e]);
    if (threadIdx.x == 0) atomicAdd(&Nk[k], rk[DT * DT + DT]);
  }
}

template <typename T, int TILE>
static int wmom_mfma(const T* X, const T* p, int64_t S, int64_t Bo, int D, T* Nk, T* SEx, T* SExx, hipStream_t st) {
  constexpr int SPS = 64 / TILE;
  const int64_t steps = (S + SPS - 1) / SPS;
  int64_t blocks = (steps + 4 * 64 - 1) / (4 * 64);  // >= 64 steps per wave
  const int64_t cap = g_vbmp_blocks_per_cu > 0 ? 256 * (int64_t)g_vbmp_blocks_per_cu : 1024;  // debug override
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  const dim3 b(256);
#define VBMP_WM(NT, BO)                                                                                              \
  hipLaunchKernelGGL((k_wmom_mfma<T, TILE, NT, BO>), dim3((unsigned)blocks, (unsigned)ngroups), b, 0, st, X,          \
                     p ? p + k0 : p, S, D, (int)Bo, Nk + k0, SEx + k0 * D, SExx + k0 * D * D)
  const int nt = (D + TILE - 1) / TILE;
  // components in groups per pass over X (accumulators live in registers: BO * NT^2 tiles, so wide statistics take
  // fewer components per pass: 4 for NT <= 2, 2 for NT = 3, 1 for NT = 4); all full groups go out as ONE launch
  // (grid.y), a last partial group as a second one
  const int64_t group = nt <= 2 ? 4 : (nt == 3 ? 2 : 1);
  const int64_t full = Bo / group, rem = Bo - full * group;
  bool tail_done = (rem == 0);
  for (int64_t done = 0; done < full || !tail_done;) {
    const bool tail = done == full;  // the partial group, after all full ones (grid.y holds at most 65535 of those)
    const int64_t k0 = done * group;
    const int64_t nb = tail ? rem : group;
    const int64_t ngroups = tail ? 1 : (full - done < 65535 ? full - done : 65535);
    if (tail)
      tail_done = true;
    else
      done += ngroups;
    if (nt == 1) {
      if (nb == 1) VBMP_WM(1, 1); else if (nb == 2) VBMP_WM(1, 2); else if (nb == 3) VBMP_WM(1, 3); else VBMP_WM(1, 4);
    } else if (nt == 2) {
      if (nb == 1) VBMP_WM(2, 1); else if (nb == 2) VBMP_WM(2, 2); else if (nb == 3) VBMP_WM(2, 3); else VBMP_WM(2, 4);
    } else if constexpr (sizeof(T) == 8 && TILE == 16) {  // fp64 up to D = 64 (fp32 uses the 32-wide tile there)
      if (nt == 3) {
        if (nb == 1) VBMP_WM(3, 1); else VBMP_WM(3, 2);
      } else {
        VBMP_WM(4, 1);
      }
    }
  }
#undef VBMP_WM
  return hipGetLastError() == hipSuccess ? 0 : VBMP_ERR_LAUNCH;
}

// ------------------------------------------------------------------------------------ K4 for tiny dimensions
// D <= 4 (the two-moons / two-cluster shapes of BASELINE configs[0]): a 16-wide MFMA tile would be >= 94 % padding,
// and the job is pure streaming (16 + 8 K bytes per fp64 sample at D = 2).  One lane per sample (grid-stride), the
// D(D+1)/2 + D + 1 distinct statistics of up to four components in registers, a wave-level butterfly at the end,
// then one LDS combine per block and one set of global atomics per block.
template <typename T, int D, int BO>
__global__ __launch_bounds__(256) void k_wmom_small(const T* __restrict__ X, const T* __restrict__ p, int64_t S,
                                                     int ps, T* __restrict__ Nk, T* __restrict__ SEx,
                                                     T* __restrict__ SExx) {
  constexpr int NS = D * (D + 1) / 2 + D + 1;  // xx^T upper triangle | x | 1
  T acc[BO][NS];
#pragma unroll
  for (int k = 0; k < BO; ++k)
#pragma unroll
    for (int e = 0; e < NS; ++e) acc[k][e] = T(0);
  for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < S; s += (int64_t)gridDim.x * 256) {
    T x[D], w[BO];
#pragma unroll
    for (int i = 0; i < D; ++i) x[i] = X[s * D + i];
#pragma unroll
    for (int k = 0; k < BO; ++k) w[k] = p ? p[s * ps + k] : T(1);
#pragma unroll
    for (int k = 0; k < BO; ++k) {
      int e = 0;
#pragma unroll
      for (int i = 0; i < D; ++i) {
        const T wx = w[k] * x[i];
#pragma unroll
        for (int j = i; j < D; ++j) acc[k][e++] = xfma(wx, x[j], acc[k][e]);
        acc[k][D * (D + 1) / 2 + i] += wx;
      }
      acc[k][NS - 1] += w[k];
    }
  }
  __shared__ T red[BO * NS];
  for (int e = threadIdx.x; e < BO * NS; e += 256) red[e] = T(0);
  __syncthreads();
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < BO; ++k)
#pragma unroll
    for (int e = 0; e < NS; ++e) {
      T v = acc[k][e];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
      if (lane == 0) atomicAdd(&red[k * NS + e], v);
    }
  __syncthreads();
  for (int idx = threadIdx.x; idx < BO * NS; idx += 256) {
    const int k = idx / NS, e = idx - k * NS;
    const T v = red[idx];
    if (e == NS - 1) {
      atomicAdd(&Nk[k], v);
    } else if (e >= D * (D + 1) / 2) {
      atomicAdd(&SEx[k * D + (e - D * (D + 1) / 2)], v);
    } else {
      int i = 0, r = e;  // e -> (i, j) of the upper triangle, row-major
      while (r >= D - i) {
        r -= D - i;
        ++i;
      }
      const int j = i + r;
      atomicAdd(&SExx[(k * D + i) * D + j], v);
      if (j != i) atomicAdd(&SExx[(k * D + j) * D + i], v);
    }
  }
}

template <typename T, int D>
static int wmom_small(const T* X, const T* p, int64_t S, int64_t Bo, T* Nk, T* SEx, T* SExx, hipStream_t st) {
  int64_t blocks = (S + 255) / 256;
  if (blocks > 256 * 8) blocks = 256 * 8;
  const dim3 g((unsigned)blocks), b(256);
#define VBMP_WS(BO)                                                                                                   \
  hipLaunchKernelGGL((k_wmom_small<T, D, BO>), g, b, 0, st, X, p ? p + k0 : p, S, (int)Bo, Nk + k0, SEx + k0 * D,      \
                     SExx + k0 * D * D)
  for (int64_t k0 = 0; k0 < Bo; k0 += 4) {
    const int64_t nb = Bo - k0 < 4 ? Bo - k0 : 4;
    if (nb == 1) VBMP_WS(1); else if (nb == 2) VBMP_WS(2); else if (nb == 3) VBMP_WS(3); else VBMP_WS(4);
  }
#undef VBMP_WS
  return hipGetLastError() == hipSuccess ? 0 : VBMP_ERR_LAUNCH;
}

// ------------------------------------------------------------------------------------ K3a / K3 on the matrix cores
// out[s,bo,bi] = -1/2 x'Px + x'b + c for 16 samples at a time:  (P x_j)_i = sum_f P[i][f] x_j[f]  is the 16x16x4 MFMA
// with A = P (rows i = feature out, LDS resident in operand order), B = X^T (k = feature in, column j = sample), so
// C[i][j] lands with rows = features and columns = samples, and the C rows a lane holds are exactly the features
// whose x values the lane already carries as its B operands (the k-slot -> feature map FEAT is chosen per dtype to
// make that true: f64 C rows are q + 4r, f32 C rows are 4q + r, q = lane >> 4).  The quadratic form is then a
// 4*DT-term dot product per lane plus two cross-lane adds.  C starts at -2b, so  l = -1/2 x'(Px - 2b) + c.
// Grid: x = groups of sample tiles, y = bi (inner component axis: its own X slice and its own parameters),
// z = chunks of BC outer components whose P fit in LDS together.
template <typename T> struct QfMfma;
template <> struct QfMfma<double> {
  using acc_t = f64x4;
  static __device__ __forceinline__ acc_t run(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __host__ __forceinline__ int feat(int q, int t) { return q + 4 * t; }
};
template <> struct QfMfma<float> {
  using acc_t = f32x4;
  static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __host__ __forceinline__ int feat(int q, int t) { return 4 * q + t; }
};

// KG > 0: the fused mixture E-step -- all Bo = K <= 4 KG components of a sample are in this block's chunk (Bi == 1, one
// chunk), so the responsibilities are normalised right here (the K values of a sample sit on its four lane groups,
// KG per lane) and p, NA and logZ leave the kernel: no (samples x K) round trip of the log-likelihoods, no second launch
template <typename T, int DT, int KG = 0>  // DT = ceil(D / 16) feature blocks
__global__ __launch_bounds__(256) void k_quadform_mfma(const T* __restrict__ X, int64_t S, int64_t Bo, int64_t Bi, int D,
                                                       const T* __restrict__ P, const T* __restrict__ b,
                                                       const T* __restrict__ c, T* __restrict__ out, int BC,
                                                       int64_t tiles_per_block, T* __restrict__ NA = nullptr,
                                                       T* __restrict__ logZ = nullptr) {
  using M = QfMfma<T>;
  using acc_t = typename M::acc_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* sP = reinterpret_cast<T*>(smem_raw);      // [e][a][b][t][lane]
  T* sB = sP + (size_t)BC * DT * DT * 4 * 64;  // [e][a][lane][r] = -2 b[feat(q,r) + 16a]
  T* sC = sB + (size_t)BC * DT * 64 * 4;       // [e]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, j = lane & 15;
  const int64_t bi = blockIdx.y;
  const int bo0 = blockIdx.z * BC;
  const int nb = (Bo - bo0) < BC ? (int)(Bo - bo0) : BC;
  constexpr int PE = DT * DT * 4 * 64;
  for (int idx = threadIdx.x; idx < nb * PE; idx += 256) {
    const int e = idx / PE, rem = idx - e * PE;
    const int l = rem & 63, t = (rem >> 6) & 3, bb = (rem >> 8) % DT, a = (rem >> 8) / DT;
    const int i = (l & 15) + 16 * a, f = M::feat(l >> 4, t) + 16 * bb;
    const int64_t comp = (int64_t)(bo0 + e) * Bi + bi;
    sP[idx] = (i < D && f < D) ? P[comp * D * D + (int64_t)i * D + f] : T(0);
  }
  for (int idx = threadIdx.x; idx < nb * DT * 256; idx += 256) {
    const int e = idx / (DT * 256), rem = idx - e * (DT * 256);
    const int r = rem & 3, l = (rem >> 2) & 63, a = rem >> 8;
    const int i = M::feat(l >> 4, r) + 16 * a;
    const int64_t comp = (int64_t)(bo0 + e) * Bi + bi;
    sB[idx] = (i < D) ? T(-2) * b[comp * D + i] : T(0);
  }
  for (int e = threadIdx.x; e < nb; e += 256) sC[e] = c[(int64_t)(bo0 + e) * Bi + bi];
  __syncthreads();
  const int64_t ntiles = (S + 15) / 16;
  const int64_t t_lo = (int64_t)blockIdx.x * tiles_per_block;
  const int64_t t_hi = (t_lo + tiles_per_block < ntiles) ? t_lo + tiles_per_block : ntiles;
  [[maybe_unused]] T na_acc[KG > 0 ? KG : 1];  // running sum of p[s, 4g + q] over this lane's samples
  [[maybe_unused]] T lz_acc = T(0);
#pragma unroll
  for (int gI = 0; gI < (KG > 0 ? KG : 1); ++gI) na_acc[gI] = T(0);
  for (int64_t tile = t_lo + wave; tile < t_hi; tile += 4) {
    const int64_t s = tile * 16 + j;
    const bool ok = s < S;
    const T* xrow = X + ((ok ? s : S - 1) * Bi + bi) * D;
    T x[DT][4];
#pragma unroll
    for (int bb = 0; bb < DT; ++bb)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int f = M::feat(q, t) + 16 * bb;
        x[bb][t] = xrow[f < D ? f : 0];
      }
#pragma unroll
    for (int bb = 0; bb < DT; ++bb)
#pragma unroll
      for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(x[bb][t]));  // loads stay unconditional and batched
#pragma unroll
    for (int bb = 0; bb < DT; ++bb)
#pragma unroll
      for (int t = 0; t < 4; ++t) x[bb][t] = (M::feat(q, t) + 16 * bb < D) ? x[bb][t] : T(0);
    [[maybe_unused]] T kv[KG > 0 ? KG : 1];
    // four components per round: lane group q keeps the value of component 4g + q, so that one store instruction
    // writes 16 samples x 4 adjacent components
    for (int e0 = 0; e0 < (KG > 0 ? 4 * KG : nb); e0 += 4) {
      T keep = T(0);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u;
        if (e < nb) {  // wave-uniform
          const T* pe = sP + (size_t)e * PE + lane;
          const acc_t* be = reinterpret_cast<const acc_t*>(sB + ((size_t)e * DT * 64 + lane) * 4);
          T v = T(0);
#pragma unroll
          for (int a = 0; a < DT; ++a) {
            acc_t acc = be[a * 64];
#pragma unroll
            for (int bb = 0; bb < DT; ++bb)
#pragma unroll
              for (int t = 0; t < 4; ++t) acc = M::run(pe[((a * DT + bb) * 4 + t) * 64], x[bb][t], acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) v = xfma(acc[r], x[a][r], v);
          }
          v += __shfl_xor(v, 16, 64);
          v += __shfl_xor(v, 32, 64);
          const T val = T(-0.5) * v + sC[e];
          keep = (q == u) ? val : keep;
        }
      }
      const int eq = e0 + q;
      if constexpr (KG > 0) {
#pragma unroll
        for (int gI = 0; gI < KG; ++gI)
          if (4 * gI == e0) kv[gI] = (eq < nb) ? keep : -INFINITY;  // e0 steps through compile-time multiples of 4
      } else {
        if (ok && eq < nb) out[(s * Bo + bo0 + eq) * Bi + bi] = keep;
      }
    }
    if constexpr (KG > 0) {
      // softmax over the K = nb components of sample j: KG values on each of its four lanes (j, j+16, j+32, j+48)
      T mx = kv[0];
#pragma unroll
      for (int gI = 1; gI < KG; ++gI) mx = kv[gI] > mx ? kv[gI] : mx;
      T o = __shfl_xor(mx, 16, 64);
      mx = o > mx ? o : mx;
      o = __shfl_xor(mx, 32, 64);
      mx = o > mx ? o : mx;
      T sum = T(0);
#pragma unroll
      for (int gI = 0; gI < KG; ++gI) {
        kv[gI] = exp(kv[gI] - mx);  // one exp per entry (components beyond K: exp(-inf) = 0)
        sum += kv[gI];
      }
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      const T inv = T(1) / sum;
#pragma unroll
      for (int gI = 0; gI < KG; ++gI) {
        const T pv = kv[gI] * inv;
        const int eq = 4 * gI + q;
        if (ok && eq < nb) {
          out[s * Bo + eq] = pv;
          na_acc[gI] += pv;
        }
      }
      if (ok && q == 0) lz_acc += mx + log(sum);
    }
  }
  if constexpr (KG > 0) {
    // block-level combine: over the 16 samples of a lane group (butterfly), then over the four waves through LDS (the
    // parameter image is dead by now), then ONE atomic per component and block
    __syncthreads();
    T* red = reinterpret_cast<T*>(smem_raw);  // [wave][4 KG + 1]
#pragma unroll
    for (int gI = 0; gI < KG; ++gI) {
      T v = na_acc[gI];
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
      if (j == 0) red[wave * (4 * KG + 1) + 4 * gI + q] = v;
    }
    {
      T v = lz_acc;
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
      if (lane == 0) red[wave * (4 * KG + 1) + 4 * KG] = v;
    }
    __syncthreads();
    if (threadIdx.x <= 4 * KG) {
      const int k = threadIdx.x;
      const T tot = red[k] + red[(4 * KG + 1) + k] + red[2 * (4 * KG + 1) + k] + red[3 * (4 * KG + 1) + k];
      if (k == 4 * KG)
        atomicAdd(logZ, tot);
      else if (k < nb)
        atomicAdd(&NA[k], tot);
    }
  }
}

// second half of the mixture E-step when the log-likelihoods were produced by k_quadform_mfma:
// p[s,:] <- softmax(p[s,:]) in place, NA[k] += p[s,k], logZ += logsumexp.  One lane per sample, rows staged
// through LDS so that the global traffic is coalesced.
template <typename T>
__global__ __launch_bounds__(256) void k_estep_softmax(T* __restrict__ p, int64_t S, int K, T* __restrict__ NA,
                                                       T* __restrict__ logZ) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int KS = K + 1;
  T* sL = reinterpret_cast<T*>(smem_raw);  // 256 x (K + 1): the rows of the current chunk (odd-ish stride)
  T* sAcc = sL + 256 * KS;                 // 256 x (K + 1): per-thread running sums of p[s, k] (slot K: log-sum-exp)
  T* acc = sAcc + threadIdx.x * KS;
  for (int k = 0; k <= K; ++k) acc[k] = T(0);
  // e / K for e < 256 K <= 2^16 as a multiply-high (exact for these ranges): the re-layout loops below run it per element
  const unsigned int kinv = 0xFFFFFFFFu / (unsigned int)K + 1u;
  for (int64_t s0 = (int64_t)blockIdx.x * 256; s0 < S; s0 += (int64_t)gridDim.x * 256) {
    const int64_t n = ((S - s0) < 256 ? (S - s0) : 256) * K;
    T* src = p + s0 * K;
    __syncthreads();
    for (int e = threadIdx.x; e < n; e += 256) {
      const int r = div_small(e, kinv, K), k = e - r * K;
      sL[r * KS + k] = src[e];
    }
    __syncthreads();
    if (s0 + threadIdx.x < S) {
      T* row = sL + threadIdx.x * KS;
      T mx = -INFINITY;
      for (int k = 0; k < K; ++k) mx = row[k] > mx ? row[k] : mx;
      T sum = T(0);
      for (int k = 0; k < K; ++k) {
        const T e = exp(row[k] - mx);  // one exp per entry; normalised by a multiply below
        row[k] = e;
        sum += e;
      }
      const T inv = T(1) / sum;
      for (int k = 0; k < K; ++k) {
        const T v = row[k] * inv;
        row[k] = v;
        acc[k] += v;  // own slot: no exchange per chunk, one block-level sum at the very end
      }
      acc[K] += mx + log(sum);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < n; e += 256) {
      const int r = div_small(e, kinv, K), k = e - r * K;
      src[e] = sL[r * KS + k];
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k <= K; k += 256) {
    T tot = T(0);
    for (int r = 0; r < 256; ++r) tot += sAcc[r * KS + k];
    if (k < K)
      atomicAdd(&NA[k], tot);
    else
      atomicAdd(logZ, tot);
  }
}

// launch of the MFMA quadratic form; returns false when the shape is not served (caller falls back to k_quadform).
// With NA / logZ given (mixture E-step, Bi == 1): the fused form when all Bo = K components fit one chunk and K <= 32;
// *fused tells the caller whether p already holds the responsibilities.
template <typename T>
static bool quadform_mfma_launch(const T* X, int64_t S, int64_t Bo, int64_t Bi, int D, const T* P, const T* b,
                                 const T* c, T* out, hipStream_t st, T* NA = nullptr, T* logZ = nullptr,
                                 bool* fused = nullptr) {
  if (D < 8 || D > 64 || S * Bi < 2048 || Bi > 65535) return false;
  const int DT = (D + 15) / 16;
  const size_t per = ((size_t)DT * DT * 256 + (size_t)DT * 256 + 1) * sizeof(T);
  int BC = (int)((size_t)(56 * 1024) / per);
  if (BC < 1) return false;
  if (BC > 32) BC = 32;
  if (BC > Bo) BC = (int)Bo;
  const int64_t nz = (Bo + BC - 1) / BC;
  if (nz > 65535) return false;
  const int64_t ntiles = (S + 15) / 16;
  // enough blocks to fill the chip, but every block pays for staging its P chunk: at least 32 tiles per block
  int64_t want = (256 * 8) / (Bi * nz);
  if (want < 1) want = 1;
  int64_t tpb = (ntiles + want - 1) / want;
  if (tpb < 32) tpb = 32;
  const int64_t gx = (ntiles + tpb - 1) / tpb;
  const dim3 g((unsigned)gx, (unsigned)Bi, (unsigned)nz), blk(256);
  const size_t smem = per * BC;
  if (fused) *fused = false;
  if (NA && logZ && Bi == 1 && nz == 1 && Bo <= 32 && !(g_vbmp_flags & VBMP_DBG_ESTEP_2KERNEL)) {  // debug flag: two-kernel form (A/B, tests)
    const int KG = Bo <= 4 ? 1 : Bo <= 8 ? 2 : Bo <= 16 ? 4 : 8;
#define VBMP_QFF(DTV, KGV) \
  hipLaunchKernelGGL((k_quadform_mfma<T, DTV, KGV>), g, blk, smem, st, X, S, Bo, Bi, D, P, b, c, out, BC, tpb, NA, logZ)
#define VBMP_QFK(DTV) \
  do { if (KG == 1) VBMP_QFF(DTV, 1); else if (KG == 2) VBMP_QFF(DTV, 2); else if (KG == 4) VBMP_QFF(DTV, 4); else VBMP_QFF(DTV, 8); } while (0)
    if (DT == 1) VBMP_QFK(1); else if (DT == 2) VBMP_QFK(2); else if (DT == 3) VBMP_QFK(3); else VBMP_QFK(4);
#undef VBMP_QFK
#undef VBMP_QFF
    if (fused) *fused = true;
    return true;
  }
#define VBMP_QF(DTV) hipLaunchKernelGGL((k_quadform_mfma<T, DTV>), g, blk, smem, st, X, S, Bo, Bi, D, P, b, c, out, BC, tpb)
  if (DT == 1) VBMP_QF(1); else if (DT == 2) VBMP_QF(2); else if (DT == 3) VBMP_QF(3); else VBMP_QF(4);
#undef VBMP_QF
  return true;
}

template <typename T>
static int quadform_dispatch(const T* X, int64_t S, int64_t Bo, int64_t Bi, int D, const T* P, const T* b, const T* c,
                             T* out, void* stream) {
  if (S == 0 || Bo == 0 || Bi == 0) return 0;
  if (!X || !P || !b || !c || !out || S < 0 || Bo < 0 || Bi < 0 || D < 1 || D > VBMP_MAX_DIM) return VBMP_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (!(g_vbmp_flags & VBMP_DBG_QF_VALU) && quadform_mfma_launch<T>(X, S, Bo, Bi, D, P, b, c, out, st))  // debug flag: VALU form only
    return hipGetLastError() == hipSuccess ? 0 : VBMP_ERR_LAUNCH;
  const int64_t blocks = (S * Bi + 255) / 256;
  VBMP_DISPATCH_DIM(T, D, {
    if (D == DP)
      hipLaunchKernelGGL((k_quadform<T, DP, true>), dim3((unsigned)blocks), dim3(256), 0, st, X, S, Bo, Bi, D, P, b, c,
                         out);
    else
      hipLaunchKernelGGL((k_quadform<T, DP, false>), dim3((unsigned)blocks), dim3(256), 0, st, X, S, Bo, Bi, D, P, b, c,
                         out);
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
  int64_t blocks = (S + 255) / 256;
  if (blocks > 256 * 8) blocks = 256 * 8;
  // D >= 8: log-likelihoods on the matrix cores into the p buffer, then the in-place softmax pass
  const size_t sm_smem = (size_t)512 * (K + 1) * sizeof(T);  // rows of a chunk + per-thread running sums
  bool fused = false;
  if (!(g_vbmp_flags & VBMP_DBG_QF_VALU) && sm_smem <= 150 * 1024 && quadform_mfma_launch<T>(X, S, K, 1, D, P, b, c, p, st, NA, logZ, &fused)) {
    if (fused) return hipGetLastError() == hipSuccess ? 0 : VBMP_ERR_LAUNCH;  // p, NA, logZ are complete
    if (sm_smem > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_estep_softmax<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)sm_smem) != hipSuccess)
      return VBMP_ERR_LAUNCH;
    hipLaunchKernelGGL((k_estep_softmax<T>), dim3((unsigned)blocks), dim3(256), sm_smem, st, p, S, K, NA, logZ);
    return hipGetLastError() == hipSuccess ? 0 : VBMP_ERR_LAUNCH;
  }
  size_t smem = (size_t)(K + 1) * sizeof(T);
  const size_t staged = (size_t)513 * (K + 1) * sizeof(T);  // block partials | 256 rows of a chunk | 256 rows of running sums
  const int stage = staged <= 40 * 1024;
  if (stage) smem = staged;
  VBMP_DISPATCH_DIM(T, D, {
    if (D == DP)
      hipLaunchKernelGGL((k_mixture_estep<T, DP, true>), dim3((unsigned)blocks), dim3(256), smem, st, X, S, K, D, P, b,
                         c, p, NA, logZ, stage);
    else
      hipLaunchKernelGGL((k_mixture_estep<T, DP, false>), dim3((unsigned)blocks), dim3(256), smem, st, X, S, K, D, P, b,
                         c, p, NA, logZ, stage);
    return hipGetLastError() == hipSuccess ? 0 : VBMP_ERR_LAUNCH;
  });
  return VBMP_ERR_ARG;
}

template <typename T>
static int estep_sym_dispatch(const T* X, int64_t S, int K, int D, const T* Q, const T* b, const T* c, T* p, T* NA, T* logZ,
                              T* lse, void* stream) {
  if (S == 0) return 0;
  if (!X || !Q || !b || !c || !p || !NA || !logZ || S < 0 || K < 1 || K > VBMP_ESTEP_SYM_MAX_K) return VBMP_ERR_ARG;
  if (D != 4 && D != 8 && D != 16 && D != 32) return VBMP_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
#ifndef VBMP_ES_NS64
#define VBMP_ES_NS64 2
#endif
#ifndef VBMP_ES_NS32
#define VBMP_ES_NS32 2
#endif
  constexpr int NS = sizeof(T) == 8 ? VBMP_ES_NS64 : VBMP_ES_NS32;  // samples per thread (A/B: tools/exp/build_variant.sh)
  // samples per thread: NS while the block's LDS rows leave several blocks per CU, else one (measured: K = 8 prefers 1, K = 4
  // prefers 2 in fp64, tools/exp/estep_sym_ab.py); D = 32 always one (registers)
  auto smem_of = [&](int ns) { return (size_t)((K + 1) + (256 * ns + 256) * (K + 1)) * sizeof(T); };
  int ns = (D == 32) ? 1 : NS;
  if (ns > 1 && smem_of(ns) > 40 * 1024) ns = 1;
  const size_t smem = smem_of(ns);
  if (smem > 150 * 1024) return VBMP_ERR_ARG;
  int64_t blocks = (S + 256 * ns - 1) / (256 * ns);
  if (blocks > 256 * 8) blocks = 256 * 8;  // same-address atomics at the end are per block
#define VBMP_ES(DPV, NSV)                                                                                                \
  do {                                                                                                                   \
    if (smem > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(&k_estep_sym<T, DPV, NSV>),                \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)   \
      return VBMP_ERR_LAUNCH;                                                                                            \
    hipLaunchKernelGGL((k_estep_sym<T, DPV, NSV>), dim3((unsigned)blocks), dim3(256), smem, st, X, S, K, Q, b, c, p, NA, \
                       logZ, lse);                                                                                       \
  } while (0)
  if (D == 32) VBMP_ES(32, 1);
  else if (ns == 1) {
    if (D == 4) VBMP_ES(4, 1);
    else if (D == 8) VBMP_ES(8, 1);
    else VBMP_ES(16, 1);
  } else {
    if (D == 4) VBMP_ES(4, NS);
    else if (D == 8) VBMP_ES(8, NS);
    else VBMP_ES(16, NS);
  }
#undef VBMP_ES
  return hipGetLastError() == hipSuccess ? 0 : VBMP_ERR_LAUNCH;
}

template <typename T>
static int wmom_dispatch(const T* X, const T* p, int64_t S, int64_t Bo, int64_t Bi, int D, T* Nk, T* SEx, T* SExx,
                         void* stream) {
  if (S == 0 || Bo == 0 || Bi == 0) return 0;
  if (!X || !Nk || !SEx || !SExx || S < 0 || Bo < 0 || Bi < 0 || D < 1 || D > 2 * VBMP_MAX_DIM || Bi > 65535)
    return VBMP_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  // dense contraction over many samples: matrix cores (see k_wmom_mfma)
  if (Bi == 1 && Bo <= 64 && S >= 4096) {
    if (D == 1) return wmom_small<T, 1>(X, p, S, Bo, Nk, SEx, SExx, st);
    if (D == 2) return wmom_small<T, 2>(X, p, S, Bo, Nk, SEx, SExx, st);
    if (D == 3) return wmom_small<T, 3>(X, p, S, Bo, Nk, SEx, SExx, st);
    if (D == 4) return wmom_small<T, 4>(X, p, S, Bo, Nk, SEx, SExx, st);
    if constexpr (sizeof(T) == 4) {
      if (D <= 16) return wmom_mfma<float, 16>(X, p, S, Bo, D, Nk, SEx, SExx, st);
      if (D <= 64) return wmom_mfma<float, 32>(X, p, S, Bo, D, Nk, SEx, SExx, st);
    } else {
      if (D <= 64) return wmom_mfma<double, 16>(X, p, S, Bo, D, Nk, SEx, SExx, st);
    }
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
  int vbmp_mixture_estep_sym_##SUF(const T* X, int64_t S, int K, int D, const T* Q, const T* b, const T* c, T* p,    \
                                   T* NA, T* logZ, T* lse, void* stream) {                                           \
    return estep_sym_dispatch<T>(X, S, K, D, Q, b, c, p, NA, logZ, lse, stream);                                     \
  }                                                                                                                  \
  int vbmp_weighted_moments_##SUF(const T* X, const T* p, int64_t S, int64_t Bo, int64_t Bi, int D, T* Nk, T* SEx,   \
                                  T* SExx, void* stream) {                                                           \
    return wmom_dispatch<T>(X, p, S, Bo, Bi, D, Nk, SEx, SExx, stream);                                              \
  }
VBMP_DEF_MIX(f64, double)
VBMP_DEF_MIX(f32, float)
}  // extern "C"
