// K5b: streaming weighted sum of per-sample matrices,  out[e] += sum_s w[s] * C[s, e]   (e < E = d*d),
// the covariance part of MatrixNormalWishart.update (sum_s p_s Sigma_s, transforms/MatrixNormalWishart.py:153-155).
// Pure HBM streaming: each block owns a slab of samples, every lane keeps 16 bytes of the row in registers and
// the rows are read with full-line 16-byte-per-lane loads; partials are combined with float atomics.  gfx950 only.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vbmp_dispatch.h"
#include "vbmp_flags.h"
extern "C" int g_vbmp_flags;
#include "../../include/vbmp_hip.h"

namespace vbmp {

template <typename T>
__global__ __launch_bounds__(256) void k_weighted_matsum(const T* __restrict__ C, const T* __restrict__ w, int64_t S,
                                                         int64_t E, int64_t chunk, T* __restrict__ out) {
  constexpr int V = 16 / sizeof(T);
  using vec_t = T __attribute__((ext_vector_type(V)));
  const int64_t s0 = (int64_t)blockIdx.y * chunk;
  const int64_t s1 = (s0 + chunk < S) ? s0 + chunk : S;
  const bool vec_ok = (E % V == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
  if (vec_ok) {
    const int64_t ev = E / V;
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < ev; c += (int64_t)gridDim.x * 256) {
      // eight independent 16-byte loads in flight per lane (the slab is streamed once: non-temporal)
      constexpr int U = 8;
      vec_t acc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) acc[u] = vec_t{};
      const vec_t* base = reinterpret_cast<const vec_t*>(C) + c;
      int64_t s = s0;
      for (; s + U <= s1; s += U) {
        vec_t v[U];
        T ww[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          v[u] = __builtin_nontemporal_load(&base[(s + u) * ev]);
          ww[u] = w ? w[s + u] : T(1);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] += ww[u] * v[u];
      }
      for (; s < s1; ++s) acc[0] += (w ? w[s] : T(1)) * base[s * ev];
#pragma unroll
      for (int u = 1; u < U; ++u) acc[0] += acc[u];
      const vec_t a0 = acc[0];
#pragma unroll
      for (int u = 0; u < V; ++u) atomicAdd(&out[c * V + u], a0[u]);
    }
  } else {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < E; e += (int64_t)gridDim.x * 256) {
      T a = T(0);
      for (int64_t s = s0; s < s1; ++s) a += (w ? w[s] : T(1)) * C[s * E + e];
      atomicAdd(&out[e], a);
    }
  }
}

// Several weight columns at once:  out[b, e] += sum_s W[s, b] * C[s, e],  b < NB  -- the covariance part of the update
// when every expert / role weighs the SAME per-sample matrices (the latent message all roles of a DMBD observation
// see).  As a library GEMM this is W^T (NB x S) @ C (S x E) with the sample axis as the inner dimension, which rocBLAS
// runs on a handful of workgroups (0.83 ms for 24000 x 2809 by 25 columns); here C is streamed once, each lane keeps
// NB 16-byte accumulators, and the weights of a sample are wave-uniform (scalar loads).
template <typename T, int NBP, int V>  // V elements per lane and load: 16 bytes, or 1 when E or the base address do not allow it
__global__ __launch_bounds__(256) void k_weighted_matsum_cols(const T* __restrict__ C, const T* __restrict__ W, int64_t S,
                                                              int64_t E, int NB, int64_t chunk, T* __restrict__ out) {
  using vec_t = T __attribute__((ext_vector_type(V)));
  const int64_t s0 = (int64_t)blockIdx.y * chunk;
  const int64_t s1 = (s0 + chunk < S) ? s0 + chunk : S;
  const int64_t ev = E / V;
  // the weights of SUB samples at a time sit in LDS, rows padded to NBP with zeros: every lane reads the same words (a
  // broadcast) and the inner loop needs no `b < NB` guard.  (Read straight from global memory they are NB scalar loads
  // per sample and wave, which bound the first version of this kernel at 1.2 TB/s.)
  constexpr int SUB = 64;
  __shared__ __attribute__((aligned(16))) T wsm[SUB * NBP];
  for (int e = threadIdx.x; e < SUB * NBP; e += 256) wsm[e] = T(0);
  for (int64_t cg = blockIdx.x; cg * 256 < ev; cg += gridDim.x) {  // block-uniform: the barriers below are reached by all
    const int64_t c = cg * 256 + threadIdx.x;
    const bool active = c < ev;
    vec_t acc[NBP];
#pragma unroll
    for (int b = 0; b < NBP; ++b) acc[b] = vec_t{};
    const vec_t* base = reinterpret_cast<const vec_t*>(C) + (active ? c : 0);
    for (int64_t sb = s0; sb < s1; sb += SUB) {
      const int n = (int)((s1 - sb) < SUB ? (s1 - sb) : SUB);
      __syncthreads();  // the previous sub-chunk's readers are done
      for (int e = threadIdx.x; e < n * NB; e += 256) {
        const int i = e / NB;
        wsm[i * NBP + (e - i * NB)] = W[sb * NB + e];
      }
      __syncthreads();
      if (active) {
        constexpr int U = 4;
        int i = 0;
        for (; i + U <= n; i += U) {
          vec_t v[U];
#pragma unroll
          for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(&base[(sb + i + u) * ev]);
#pragma unroll
          for (int u = 0; u < U; ++u)
#pragma unroll
            for (int b = 0; b < NBP; ++b) acc[b] += wsm[(i + u) * NBP + b] * v[u];
        }
        for (; i < n; ++i) {
          const vec_t v = base[(sb + i) * ev];
#pragma unroll
          for (int b = 0; b < NBP; ++b) acc[b] += wsm[i * NBP + b] * v;
        }
      }
    }
    if (active) {
#pragma unroll
      for (int b = 0; b < NBP; ++b) {
        if (b < NB) {
          const vec_t a = acc[b];
#pragma unroll
          for (int u = 0; u < V; ++u) atomicAdd(&out[(int64_t)b * E + c * V + u], a[u]);
        }
      }
    }
  }
}

// The same sum on the matrix cores, for MANY weight columns (round 3).  With 25 role columns on 52 x 52 matrices (the flocking
// DMBD) the form above spends 64 FMAs and 32 LDS broadcast reads per 16 bytes it loads and runs at 1.0 TB/s; as a product
// out (NB x E) = W' (NB x S) C (S x E) over the samples it is 16x16x4 MFMA steps: A[i][k] = W[s0 + k][b0 + i], B[k][j] = C[s0 + k][e(j)].
// A lane loads 16 bytes of a sample's row -- V consecutive elements -- and feeds V column tiles with them: tile u's column j IS
// element e0 + V j + u (any bijection serves, the epilogue stores with the same map), so the row is read with full 16-byte lanes.
// Wave w of a block owns 64 elements (NL = 64 / (16 V) loads per 4 samples); NBT = 1 or 2 row tiles of 16 weight columns.
typedef double ms_f64x4 __attribute__((ext_vector_type(4)));
typedef float ms_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ ms_f64x4 ms_mfma(double a, double b, ms_f64x4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ ms_f32x4 ms_mfma(float a, float b, ms_f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
template <typename T> struct MsAcc;
template <> struct MsAcc<double> { using type = ms_f64x4; static __device__ __forceinline__ int row(int r, int g) { return g + 4 * r; } };
template <> struct MsAcc<float> { using type = ms_f32x4; static __device__ __forceinline__ int row(int r, int g) { return 4 * g + r; } };

template <typename T, int NBT>
__global__ __launch_bounds__(256) void k_weighted_matsum_cols_mfma(const T* __restrict__ C, const T* __restrict__ W, int64_t S,
                                                                   int64_t E, int NB, int64_t chunk, T* __restrict__ out) {
  constexpr int V = 16 / sizeof(T);   // elements per lane and load = column tiles per load
  constexpr int NL = 64 / (16 * V);   // loads per wave and k step (64 elements per wave)
  using vec_t = T __attribute__((ext_vector_type(V)));
  using acc_t = typename MsAcc<T>::type;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, f = lane & 15;
  const int64_t s0 = (int64_t)blockIdx.y * chunk;
  const int64_t s1 = (s0 + chunk < S) ? s0 + chunk : S;
  const int64_t e_wave = ((int64_t)blockIdx.x * 4 + wave) * 64;  // first element of this wave
  if (e_wave >= E) return;                                      // (no block barrier below)
  // my 16-byte piece of load n: elements e_wave + 16 V n + V f .. + V - 1 (clamped into the row: stores are guarded)
  int64_t eo[NL];
#pragma unroll
  for (int n = 0; n < NL; ++n) {
    const int64_t e = e_wave + 16 * V * n + V * f;
    eo[n] = e + V <= E ? e : E - V;
  }
  acc_t acc[NBT][NL * V];
#pragma unroll
  for (int bt = 0; bt < NBT; ++bt)
#pragma unroll
    for (int t = 0; t < NL * V; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[bt][t][r] = T(0);
  int bcol[NBT];
  bool bok[NBT];
#pragma unroll
  for (int bt = 0; bt < NBT; ++bt) {
    bok[bt] = 16 * bt + f < NB;
    bcol[bt] = bok[bt] ? 16 * bt + f : 0;
  }
  auto load = [&](int64_t sb, vec_t (&bv)[NL], T (&av)[NBT]) {
    const int64_t s = sb + q;
    const bool sok = s < s1;
    const int64_t sc = sok ? s : s1 - 1;
#pragma unroll
    for (int n = 0; n < NL; ++n) bv[n] = __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(C + sc * E + eo[n]));
#pragma unroll
    for (int bt = 0; bt < NBT; ++bt) {
      const T w = W[sc * NB + bcol[bt]];
      av[bt] = (sok && bok[bt]) ? w : T(0);  // a sample past the slab or a column past NB contributes nothing
    }
  };
  constexpr int U = 2;  // k steps in flight
  vec_t bv[U][NL];
  T av[U][NBT];
  int64_t sb = s0;
#pragma unroll
  for (int u = 0; u < U; ++u) load(sb + 4 * u, bv[u], av[u]);
  for (; sb < s1; sb += 4 * U) {
    vec_t bc[U][NL];
    T ac[U][NBT];
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int n = 0; n < NL; ++n) bc[u][n] = bv[u][n];
#pragma unroll
      for (int bt = 0; bt < NBT; ++bt) ac[u][bt] = av[u][bt];
    }
    if (sb + 4 * U < s1) {  // wave-uniform: the next U steps' operands while these multiply
#pragma unroll
      for (int u = 0; u < U; ++u) load(sb + 4 * U + 4 * u, bv[u], av[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int bt = 0; bt < NBT; ++bt)
#pragma unroll
        for (int n = 0; n < NL; ++n)
#pragma unroll
          for (int v = 0; v < V; ++v) acc[bt][n * V + v] = ms_mfma(ac[u][bt], bc[u][n][v], acc[bt][n * V + v]);
  }
  // C tile (bt, n, v): column j = f is element e_wave + 16 V n + V f + v, rows i = MsAcc::row(r, q) are weight columns 16 bt + i
#pragma unroll
  for (int bt = 0; bt < NBT; ++bt)
#pragma unroll
    for (int n = 0; n < NL; ++n)
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const int64_t e = e_wave + 16 * V * n + V * f + v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int b = 16 * bt + MsAcc<T>::row(r, q);
          if (b < NB && e < E) atomicAdd(&out[(int64_t)b * E + e], acc[bt][n * V + v][r]);
        }
      }
}

template <typename T>
static int matsum_cols_mfma_launch(const T* C, const T* W, int64_t S, int64_t E, int NB, T* out, hipStream_t st) {
  const int64_t bx = (E + 255) / 256;
  // sample slabs: enough blocks to fill the chip, few enough that the final atomics (NB x 256 per block) stay short
  int64_t by = (768 + bx - 1) / bx;
  int64_t chunk = (S + by - 1) / by;
  chunk = (chunk + 7) / 8 * 8;
  if (chunk < 64) chunk = 64;
  by = (S + chunk - 1) / chunk;
  if (by > 65535) {
    by = 65535;
    chunk = ((S + by - 1) / by + 7) / 8 * 8;
    by = (S + chunk - 1) / chunk;
  }
  const dim3 grid((unsigned)bx, (unsigned)by);
  if (NB <= 16)
    hipLaunchKernelGGL((k_weighted_matsum_cols_mfma<T, 1>), grid, dim3(256), 0, st, C, W, S, E, NB, chunk, out);
  else
    hipLaunchKernelGGL((k_weighted_matsum_cols_mfma<T, 2>), grid, dim3(256), 0, st, C, W, S, E, NB, chunk, out);
  return hipGetLastError() == hipSuccess ? 0 : VBMP_ERR_LAUNCH;
}

template <typename T, int V>
static int matsum_cols_launch(const T* C, const T* W, int64_t S, int64_t E, int NB, T* out, hipStream_t st) {
  int64_t bx = (E / V + 255) / 256;
  if (bx > 64) bx = 64;
  int64_t by = (1024 + bx - 1) / bx;
  int64_t chunk = (S + by - 1) / by;
  if (chunk < 32) chunk = 32;
  by = (S + chunk - 1) / chunk;
  if (by > 65535) {
    by = 65535;
    chunk = (S + by - 1) / by;
  }
  const dim3 grid((unsigned)bx, (unsigned)by);
  if (NB <= 4)
    hipLaunchKernelGGL((k_weighted_matsum_cols<T, 4, V>), grid, dim3(256), 0, st, C, W, S, E, NB, chunk, out);
  else if (NB <= 8)
    hipLaunchKernelGGL((k_weighted_matsum_cols<T, 8, V>), grid, dim3(256), 0, st, C, W, S, E, NB, chunk, out);
  else if (NB <= 16)
    hipLaunchKernelGGL((k_weighted_matsum_cols<T, 16, V>), grid, dim3(256), 0, st, C, W, S, E, NB, chunk, out);
  else
    hipLaunchKernelGGL((k_weighted_matsum_cols<T, 32, V>), grid, dim3(256), 0, st, C, W, S, E, NB, chunk, out);
  return hipGetLastError() == hipSuccess ? 0 : VBMP_ERR_LAUNCH;
}

template <typename T>
static int matsum_cols_dispatch(const T* C, const T* W, int64_t S, int64_t E, int NB, T* out, void* stream) {
  if (S == 0 || E == 0 || NB == 0) return 0;
  if (!C || !W || !out || S < 0 || E < 0 || NB < 1 || NB > VBMP_MATSUM_MAX_COLS) return VBMP_ERR_ARG;
  constexpr int V = 16 / sizeof(T);
  hipStream_t st = (hipStream_t)stream;
  const bool vec_ok = E % V == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0;
  // many columns: the matrix-core form (the VALU form is HBM-bound up to 8 columns and FMA / LDS-bound beyond);
  // VBMP_DBG_MATSUM_VALU: the VALU form (tests, A/B timing)
  if (vec_ok && NB > 8 && E >= V && S >= 64 && !(g_vbmp_flags & VBMP_DBG_MATSUM_VALU)) return matsum_cols_mfma_launch<T>(C, W, S, E, NB, out, st);
  if (vec_ok) return matsum_cols_launch<T, V>(C, W, S, E, NB, out, st);
  return matsum_cols_launch<T, 1>(C, W, S, E, NB, out, st);  // odd d: d*d elements per matrix, one element per lane
}

template <typename T>
static int matsum_dispatch(const T* C, const T* w, int64_t S, int64_t E, T* out, void* stream) {
  if (S == 0 || E == 0) return 0;
  if (!C || !out || S < 0 || E < 0) return VBMP_ERR_ARG;
  constexpr int V = 16 / sizeof(T);
  int64_t bx = (E / V + 255) / 256;
  if (bx < 1) bx = 1;
  if (bx > 64) bx = 64;
  // enough sample slabs to fill the chip (>= 1024 blocks of 256 lanes x 8 loads in flight), at least 32 samples
  // each; few slabs keep the same-address atomics of the final combine short
  int64_t by = (1024 + bx - 1) / bx;
  int64_t chunk = (S + by - 1) / by;
  if (chunk < 32) chunk = 32;
  by = (S + chunk - 1) / chunk;
  if (by > 65535) {
    by = 65535;
    chunk = (S + by - 1) / by;
  }
  hipLaunchKernelGGL((k_weighted_matsum<T>), dim3((unsigned)bx, (unsigned)by), dim3(256), 0, (hipStream_t)stream, C, w, S, E,
                     chunk, out);
  return hipGetLastError() == hipSuccess ? 0 : VBMP_ERR_LAUNCH;
}

}  // namespace vbmp

extern "C" {
int vbmp_weighted_matsum_f64(const double* C, const double* w, int64_t S, int64_t E, double* out, void* stream) {
  return vbmp::matsum_dispatch<double>(C, w, S, E, out, stream);
}
int vbmp_weighted_matsum_f32(const float* C, const float* w, int64_t S, int64_t E, float* out, void* stream) {
  return vbmp::matsum_dispatch<float>(C, w, S, E, out, stream);
}
int vbmp_weighted_matsum_cols_f64(const double* C, const double* W, int64_t S, int64_t E, int NB, double* out,
                                  void* stream) {
  return vbmp::matsum_cols_dispatch<double>(C, W, S, E, NB, out, stream);
}
int vbmp_weighted_matsum_cols_f32(const float* C, const float* W, int64_t S, int64_t E, int NB, float* out,
                                  void* stream) {
  return vbmp::matsum_cols_dispatch<float>(C, W, S, E, NB, out, stream);
}
}
