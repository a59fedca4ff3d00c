// K1 / K2: batched SPD inverse + logdet, fused Wishart / Normal-inverse-Wishart ss_update.
// gfx950 only.  See include/vbmp_hip.h for the C-ABI contract and DESIGN.md for the layout.
#include "vbmp_device.h"
#include "vbmp_dispatch.h"
#include "../../include/vbmp_hip.h"

namespace vbmp {

// ------------------------------------------------------------------------------------ K1
// Ainv = A^-1, logdet = log det A for B independent D x D (SPD) matrices.
template <typename T, int Dp, int G, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_spd_inv_logdet(const T* __restrict__ A, int64_t sA, T* __restrict__ Ainv,
                                                        T* __restrict__ logdet, int64_t B, int D,
                                                        int* __restrict__ nonspd) {
  using TL = Tile<T, Dp, G>;
  __shared__ __attribute__((aligned(16))) T lds_all[WPB * TL::LDS_ELEMS];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  T* lds = lds_all + wave * TL::LDS_ELEMS;
  const int64_t tile = (int64_t)blockIdx.x * WPB + wave;
  const int64_t m0 = tile * TL::MPW;
  if (m0 >= B) return;
  const int nm = (int)((B - m0) < TL::MPW ? (B - m0) : TL::MPW);
  const int mloc = lane / G, lig = lane % G;
  const bool live = mloc < nm;

  tile_g2lds<T, Dp, G>(A + m0 * sA, sA, nm, D, lds, lane);
  wave_lds_sync();
  T a[TL::R][Dp];
  tile_lds2rows<T, Dp, G, TL::R>(lds, live ? mloc : 0, lig, D, a);
  LogDet<T> ld;
  gj_inverse<T, Dp, G, TL::R>(a, lig, ld);
  wave_lds_sync();
  tile_rows2lds<T, Dp, G, TL::R>(lds, mloc, lig, D, a);
  wave_lds_sync();
  tile_lds2g<T, Dp, G>(Ainv + m0 * (int64_t)D * D, nm, D, lds, lane);
  if (live && lig == 0) {
    if (logdet) logdet[m0 + mloc] = ld.value();
    if (nonspd && (ld.neg || ld.bad)) atomicAdd(nonspd, 1);
  }
}

// ------------------------------------------------------------------------------------ K2
// Fused conjugate update of B independent posteriors (ref dists/NormalInverseWishart.py:61-68 +
// dists/Wishart.py:53-56).  MEAN=false gives the plain Wishart update (no rank-1 terms).
template <typename T>
struct NiwArgs {
  const T* SExx; int64_t sSExx;   // (B,D,D)
  const T* SEx;  int64_t sSEx;    // (B,D)      [MEAN]
  const T* N;    int64_t sN;      // (B)
  const T* lam0; int64_t slam0;   // prior (stride 0 = shared)  [MEAN]
  const T* mu0;  int64_t smu0;    //                              [MEAN]
  const T* invU0; int64_t sinvU0;
  const T* nu0;  int64_t snu0;
  const T* lam_old; int64_t slam_old;  // previous posterior, read only when lr != 1
  const T* mu_old;  int64_t smu_old;
  const T* invU_old; int64_t sinvU_old;
  const T* nu_old;  int64_t snu_old;
  T lr;
  T* lam; T* mu; T* invU; T* nu; T* U; T* logdet;  // outputs, contiguous; lam/mu nullable when !MEAN
  int64_t B; int D; int fixed_precision; int* nonspd;
};

template <typename T, int Dp, int G, int WPB, bool MEAN>
__global__ __launch_bounds__(64 * WPB) void k_niw_ss_update(NiwArgs<T> p) {
  using TL = Tile<T, Dp, G>;
  __shared__ __attribute__((aligned(16))) T lds_all[WPB * TL::LDS_ELEMS];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  T* lds = lds_all + wave * TL::LDS_ELEMS;
  const int64_t tile = (int64_t)blockIdx.x * WPB + wave;
  const int64_t m0 = tile * TL::MPW;
  const int D = p.D;
  if (m0 >= p.B) return;
  const int nm = (int)((p.B - m0) < TL::MPW ? (p.B - m0) : TL::MPW);
  const int mloc = lane / G, lig = lane % G;
  const bool live = mloc < nm;
  const int64_t m = m0 + (live ? mloc : 0);  // dead lanes shadow matrix m0 (never stored)
  const bool blend = (p.lr != T(1));
  const T lr = p.lr, olr = T(1) - p.lr;

  // stream the tile's SExx through LDS (coalesced) while the small per-element operands load
  tile_g2lds<T, Dp, G>(p.SExx + m0 * p.sSExx, p.sSExx, nm, D, lds, lane);

  const T Nn = p.N[m * p.sN];
  T lam = T(0), lam0 = T(0);
  T mu[Dp], mu0v[Dp];
  if constexpr (MEAN) {
    lam0 = p.lam0[m * p.slam0];
    lam = lam0 + Nn;
    const T rl = rcp_nr(lam);
    const T* sx = p.SEx + m * p.sSEx;
    const T* m0p = p.mu0 + m * p.smu0;
#pragma unroll
    for (int j = 0; j < Dp; ++j) {
      const bool in = j < D;
      mu0v[j] = in ? m0p[j] : T(0);
      mu[j] = in ? (lam0 * mu0v[j] + sx[j]) * rl : T(0);
    }
  }
  wave_lds_sync();
  T a[TL::R][Dp];
  tile_lds2rows<T, Dp, G, TL::R>(lds, live ? mloc : 0, lig, D, a);

  if (!p.fixed_precision) {
    // invU <- lr*(invU_0 + SExx + lam0 mu0 mu0^T - lam mu mu^T) + (1-lr)*invU_old
#pragma unroll
    for (int q = 0; q < TL::R; ++q) {
      const int row = lig + G * q;
      if (row < D) {
        const T* pri = p.invU0 + m * p.sinvU0 + (int64_t)row * D;
        const T* old = blend ? p.invU_old + m * p.sinvU_old + (int64_t)row * D : nullptr;
        T c0 = T(0), c1 = T(0);
        if constexpr (MEAN) {
          // row-th entries of mu0 / mu (dynamic row, static array: pick with a compare chain)
#pragma unroll
          for (int j = 0; j < Dp; ++j) {
            c0 = (j == row) ? mu0v[j] : c0;
            c1 = (j == row) ? mu[j] : c1;
          }
          c0 *= lam0;
          c1 *= lam;
        }
#pragma unroll
        for (int j = 0; j < Dp; ++j) {
          if (j < D) {
            T v = a[q][j];
            if constexpr (MEAN) v = v + c0 * mu0v[j] - c1 * mu[j];
            v = pri[j] + v;
            if (blend) v = lr * v + olr * old[j];
            a[q][j] = v;
          }
        }
      }
    }
    // posterior invU goes out through the LDS image (coalesced), then invert in registers
    wave_lds_sync();
    tile_rows2lds<T, Dp, G, TL::R>(lds, mloc, lig, D, a);
    wave_lds_sync();
    tile_lds2g<T, Dp, G>(p.invU + m0 * (int64_t)D * D, nm, D, lds, lane);
    LogDet<T> ld;
    gj_inverse<T, Dp, G, TL::R>(a, lig, ld);
    wave_lds_sync();
    tile_rows2lds<T, Dp, G, TL::R>(lds, mloc, lig, D, a);
    wave_lds_sync();
    tile_lds2g<T, Dp, G>(p.U + m0 * (int64_t)D * D, nm, D, lds, lane);
    if (live && lig == 0) {
      T nu = p.nu0[m * p.snu0] + Nn;
      if (blend) nu = lr * nu + olr * p.nu_old[m * p.snu_old];
      p.nu[m] = nu;
      p.logdet[m] = ld.value();
      if (p.nonspd && (ld.neg || ld.bad)) atomicAdd(p.nonspd, 1);
    }
  }
  if constexpr (MEAN) {
    if (live) {
      if (lig == 0) p.lam[m] = blend ? lr * lam + olr * p.lam_old[m * p.slam_old] : lam;
      // mean: lane lig writes entries lig, lig+G, ...
#pragma unroll
      for (int j = 0; j < Dp; ++j) {
        if (j < D && (j % G) == lig) {
          T v = mu[j];
          if (blend) v = lr * v + olr * p.mu_old[m * p.smu_old + j];
          p.mu[m * D + j] = v;
        }
      }
    }
  }
}

template <typename T, int Dp, int G, int WPB>
static int launch_spd_inv(const T* A, int64_t sA, T* Ainv, T* logdet, int64_t B, int D, int* nonspd,
                          hipStream_t st) {
  using TL = Tile<T, Dp, G>;
  const int64_t tiles = (B + TL::MPW - 1) / TL::MPW;
  const int64_t blocks = (tiles + WPB - 1) / WPB;
  hipLaunchKernelGGL((k_spd_inv_logdet<T, Dp, G, WPB>), dim3((unsigned)blocks), dim3(64 * WPB), 0, st, A, sA, Ainv, logdet, B, D,
                     nonspd);
  return hipGetLastError() == hipSuccess ? 0 : VBMP_ERR_LAUNCH;
}

template <typename T, int Dp, int G, int WPB>
static int launch_niw(const NiwArgs<T>& a, bool mean, hipStream_t st) {
  using TL = Tile<T, Dp, G>;
  const int64_t tiles = (a.B + TL::MPW - 1) / TL::MPW;
  const int64_t blocks = (tiles + WPB - 1) / WPB;
  if (mean)
    hipLaunchKernelGGL((k_niw_ss_update<T, Dp, G, WPB, true>), dim3((unsigned)blocks), dim3(64 * WPB), 0, st, a);
  else
    hipLaunchKernelGGL((k_niw_ss_update<T, Dp, G, WPB, false>), dim3((unsigned)blocks), dim3(64 * WPB), 0, st, a);
  return hipGetLastError() == hipSuccess ? 0 : VBMP_ERR_LAUNCH;
}

template <typename T>
static int spd_inv_dispatch(const T* A, int64_t sA, T* Ainv, T* logdet, int64_t B, int D, int* nonspd, void* stream) {
  if (B == 0) return 0;
  if (!A || !Ainv || B < 0 || D < 1 || D > VBMP_MAX_DIM || sA < (int64_t)D * D) return VBMP_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  VBMP_DISPATCH_DIM(T, D, return (launch_spd_inv<T, DP, GG, WW>(A, sA, Ainv, logdet, B, D, nonspd, st)));
  return VBMP_ERR_ARG;
}

template <typename T>
static int niw_dispatch(NiwArgs<T> a, bool mean, void* stream) {
  if (a.B == 0) return 0;
  if (a.B < 0 || a.D < 1 || a.D > VBMP_MAX_DIM) return VBMP_ERR_ARG;
  if (!a.SExx || !a.N || !a.invU0 || !a.nu0) return VBMP_ERR_ARG;
  if (!a.fixed_precision && (!a.invU || !a.nu || !a.U || !a.logdet)) return VBMP_ERR_ARG;
  if (mean && (!a.SEx || !a.lam0 || !a.mu0 || !a.lam || !a.mu)) return VBMP_ERR_ARG;
  if (a.lr != T(1)) {
    if (!a.fixed_precision && (!a.invU_old || !a.nu_old)) return VBMP_ERR_ARG;
    if (mean && (!a.lam_old || !a.mu_old)) return VBMP_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  VBMP_DISPATCH_DIM(T, a.D, return (launch_niw<T, DP, GG, WW>(a, mean, st)));
  return VBMP_ERR_ARG;
}

}  // namespace vbmp

using namespace vbmp;

extern "C" {

int vbmp_spd_inv_logdet_f64(const double* A, int64_t sA, double* Ainv, double* logdet, int64_t B, int D, int* nonspd,
                            void* stream) {
  return spd_inv_dispatch<double>(A, sA, Ainv, logdet, B, D, nonspd, stream);
}
int vbmp_spd_inv_logdet_f32(const float* A, int64_t sA, float* Ainv, float* logdet, int64_t B, int D, int* nonspd,
                            void* stream) {
  return spd_inv_dispatch<float>(A, sA, Ainv, logdet, B, D, nonspd, stream);
}

#define VBMP_DEF_NIW(SUF, T)                                                                                        \
  int vbmp_niw_ss_update_##SUF(const T* SExx, int64_t sSExx, const T* SEx, int64_t sSEx, const T* N, int64_t sN,    \
                               const T* lam0, int64_t slam0, const T* mu0, int64_t smu0, const T* invU0,            \
                               int64_t sinvU0, const T* nu0, int64_t snu0, const T* lam_old, int64_t slam_old,      \
                               const T* mu_old, int64_t smu_old, const T* invU_old, int64_t sinvU_old,              \
                               const T* nu_old, int64_t snu_old, T lr, T* lam, T* mu, T* invU, T* nu, T* U,         \
                               T* logdet, int64_t B, int D, int fixed_precision, int* nonspd, void* stream) {       \
    NiwArgs<T> a{SExx, sSExx, SEx, sSEx, N, sN, lam0, slam0, mu0, smu0, invU0, sinvU0, nu0, snu0,                   \
                 lam_old, slam_old, mu_old, smu_old, invU_old, sinvU_old, nu_old, snu_old, lr,                      \
                 lam, mu, invU, nu, U, logdet, B, D, fixed_precision, nonspd};                                      \
    return niw_dispatch<T>(a, true, stream);                                                                        \
  }                                                                                                                 \
  int vbmp_wishart_ss_update_##SUF(const T* SExx, int64_t sSExx, const T* N, int64_t sN, const T* invU0,            \
                                   int64_t sinvU0, const T* nu0, int64_t snu0, const T* invU_old,                   \
                                   int64_t sinvU_old, const T* nu_old, int64_t snu_old, T lr, T* invU, T* nu, T* U, \
                                   T* logdet, int64_t B, int D, int* nonspd, void* stream) {                        \
    NiwArgs<T> a{SExx, sSExx, nullptr, 0, N, sN, nullptr, 0, nullptr, 0, invU0, sinvU0, nu0, snu0,                  \
                 nullptr, 0, nullptr, 0, invU_old, sinvU_old, nu_old, snu_old, lr,                                  \
                 nullptr, nullptr, invU, nu, U, logdet, B, D, 0, nonspd};                                           \
    return niw_dispatch<T>(a, false, stream);                                                                       \
  }

VBMP_DEF_NIW(f64, double)
VBMP_DEF_NIW(f32, float)

}  // extern "C"
