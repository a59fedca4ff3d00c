// K5b: streaming weighted sum of per-sample matrices,  out[e] += sum_s w[s] * C[s, e]   (e < E = d*d),
// the covariance part of MatrixNormalWishart.update (sum_s p_s Sigma_s, transforms/MatrixNormalWishart.py:153-155).
// Pure HBM streaming: each block owns a slab of samples, every lane keeps 16 bytes of the row in registers and
// the rows are read with full-line 16-byte-per-lane loads; partials are combined with float atomics.  gfx950 only.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vbmp_dispatch.h"
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
  if (E % V == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0) return matsum_cols_launch<T, V>(C, W, S, E, NB, out, st);
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
