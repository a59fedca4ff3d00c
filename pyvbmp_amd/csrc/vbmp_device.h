// Device building blocks for the conjugate-update kernels (gfx950 / CDNA4 only).
//
// Layout idea ("row-per-lane tiles"): a 64-lane wavefront works on a TILE of MPW = 64/G
// consecutive batch elements.  Each D x D matrix (D <= Dp, Dp a power of two) is owned by a group
// of G lanes; lane `lig` of the group keeps rows lig, lig+G, ... (R = Dp/G of them) entirely in
// registers.  Global memory is always touched with linear, 16-byte-per-lane, full-cache-line
// accesses of the tile's contiguous bytes; the (coalesced <-> row-per-lane) transposition goes
// through a padded, per-wave LDS image.  Dense factor/inverse work then runs out of registers with
// the pivot row broadcast inside the lane group (DPP row_newbcast for G=16, quad_perm for G=4,
// v_readlane for G=64, nothing for G=1).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vbmp {

// ---------------------------------------------------------------- compile-time loop
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// ---------------------------------------------------------------- lane-group broadcast
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  // the backend keeps a 64-bit row_newbcast as one DP-rate DPP move (v_mov_b64_dpp) on gfx950
  return __builtin_amdgcn_update_dpp(0.0, v, CTRL, 0xF, 0xF, false);
}
__device__ __forceinline__ float readlane_any(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ double readlane_any(double v, int lane) {
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
  u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
  return u.d;
}

// value of `v` held by lane SRC of this lane's G-lane group
template <int G, int SRC, typename T>
__device__ __forceinline__ T bcast(T v) {
  static_assert(SRC >= 0 && SRC < G, "source lane outside the group");
  if constexpr (G == 1) {
    return v;
  } else if constexpr (G == 4) {
    return dpp_mov<(SRC) | (SRC << 2) | (SRC << 4) | (SRC << 6)>(v);  // quad_perm:[SRC,SRC,SRC,SRC]
  } else if constexpr (G == 16) {
    return dpp_mov<0x150 + SRC>(v);  // row_newbcast:SRC
  } else if constexpr (G == 64) {
    return readlane_any(v, SRC);
  } else {
    return __shfl(v, SRC, G);
  }
}

// ---------------------------------------------------------------- scalar helpers
// reciprocal to ~1 ulp: hardware seed + Newton steps (the pivot reciprocal sits on the critical
// path of every elimination step; a full IEEE divide is ~3x longer).
__device__ __forceinline__ double rcp_nr(double d) {
  double r = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  return r;
}
__device__ __forceinline__ float rcp_nr(float d) {
  float r = __builtin_amdgcn_rcpf(d);
  float e = __builtin_fmaf(-d, r, 1.0f);
  return __builtin_fmaf(r, e, r);
}
__device__ __forceinline__ double xfma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float xfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// running log|det| kept as (mantissa product, exponent sum, sign): one log per matrix at the end
template <typename T>
struct LogDet {
  T mant = T(1);
  int expo = 0;
  bool neg = false;
  bool bad = false;  // a zero / NaN pivot
  __device__ __forceinline__ void mul(T d) {
    bad |= !(d != T(0)) || !(d == d);
    neg ^= (d < T(0));
    int e;
    T m = frexp(d < T(0) ? -d : d, &e);
    mant *= m;  // m in [0.5,1): at most 64 factors, cannot underflow
    expo += e;
  }
  // log(det) with the reference's Tensor.logdet() conventions: NaN when det < 0, -inf when det == 0
  __device__ __forceinline__ T value() const {
    T v = log(mant) + T(expo) * T(0.693147180559945309417232121458);
    if (neg) v = __builtin_nan("");
    return v;
  }
};

// ---------------------------------------------------------------- tile geometry
template <typename T, int Dp, int G>
struct Tile {
  static_assert(Dp % G == 0 && 64 % G == 0, "lane group must divide the padded dim and the wave");
  static constexpr int MPW = 64 / G;            // matrices per wave
  static constexpr int R = Dp / G;              // rows per lane
  static constexpr int V = 16 / sizeof(T);      // elements per 16-byte chunk
  static constexpr int RS = Dp + V;             // padded LDS row stride (elements): conflict-free row reads
  static constexpr int MS = Dp * RS;            // LDS matrix stride
  static constexpr int LDS_ELEMS = MPW * MS;    // per-wave LDS image
  using vec_t = T __attribute__((ext_vector_type(V)));
};

// wave-level ordering of LDS traffic (one wave owns its LDS image; no block barrier needed)
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// global (linear, coalesced) -> per-wave LDS image [m][row][col] with padded strides.
// g points at the first matrix of the tile; matrices are `gstride` elements apart; `nm` valid.
template <typename T, int Dp, int G>
__device__ __forceinline__ void tile_g2lds(const T* __restrict__ g, int64_t gstride, int nm, int D, T* __restrict__ lds,
                                           int lane) {
  using TL = Tile<T, Dp, G>;
  const int DD = D * D;
  const bool contiguous = (gstride == DD);
  const bool vec_ok = contiguous && (D % TL::V == 0) && ((reinterpret_cast<uintptr_t>(g) & 15) == 0);
  if (vec_ok) {
    const int nchunk = nm * DD / TL::V;  // chunks never straddle a row because D % V == 0
    const int cpr = D / TL::V;           // chunks per row
    const typename TL::vec_t* gv = reinterpret_cast<const typename TL::vec_t*>(g);
    for (int c = lane; c < nchunk; c += 64) {
      typename TL::vec_t v = gv[c];
      int row_all = c / cpr;             // row index across the tile
      int cc = c - row_all * cpr;
      int m = row_all / D;
      int row = row_all - m * D;
      *reinterpret_cast<typename TL::vec_t*>(&lds[m * TL::MS + row * TL::RS + cc * TL::V]) = v;
    }
  } else {
    const int n = nm * DD;
    for (int e = lane; e < n; e += 64) {
      int m = e / DD;
      int rem = e - m * DD;
      int row = rem / D;
      int col = rem - row * D;
      lds[m * TL::MS + row * TL::RS + col] = g[(int64_t)m * gstride + rem];
    }
  }
}

// per-wave LDS image -> global (linear, coalesced)
template <typename T, int Dp, int G>
__device__ __forceinline__ void tile_lds2g(T* __restrict__ g, int nm, int D, const T* __restrict__ lds, int lane) {
  using TL = Tile<T, Dp, G>;
  const int DD = D * D;
  const bool vec_ok = (D % TL::V == 0) && ((reinterpret_cast<uintptr_t>(g) & 15) == 0);
  if (vec_ok) {
    const int nchunk = nm * DD / TL::V;
    const int cpr = D / TL::V;
    typename TL::vec_t* gv = reinterpret_cast<typename TL::vec_t*>(g);
    for (int c = lane; c < nchunk; c += 64) {
      int row_all = c / cpr;
      int cc = c - row_all * cpr;
      int m = row_all / D;
      int row = row_all - m * D;
      gv[c] = *reinterpret_cast<const typename TL::vec_t*>(&lds[m * TL::MS + row * TL::RS + cc * TL::V]);
    }
  } else {
    const int n = nm * DD;
    for (int e = lane; e < n; e += 64) {
      int m = e / DD;
      int rem = e - m * DD;
      int row = rem / D;
      int col = rem - row * D;
      g[e] = lds[m * TL::MS + row * TL::RS + col];
    }
  }
}

// LDS image -> this lane's rows.  Rows/cols >= D are padded with the identity.
template <typename T, int Dp, int G, int R>
__device__ __forceinline__ void tile_lds2rows(const T* __restrict__ lds, int mloc, int lig, int D, T (&a)[R][Dp]) {
  using TL = Tile<T, Dp, G>;
#pragma unroll
  for (int q = 0; q < TL::R; ++q) {
    const int row = lig + G * q;
    const T* src = &lds[mloc * TL::MS + row * TL::RS];
    if (D == Dp) {
#pragma unroll
      for (int c = 0; c < Dp / TL::V; ++c) {
        typename TL::vec_t v = *reinterpret_cast<const typename TL::vec_t*>(&src[c * TL::V]);
#pragma unroll
        for (int u = 0; u < TL::V; ++u) a[q][c * TL::V + u] = v[u];
      }
      if constexpr (Dp % TL::V != 0) {
#pragma unroll
        for (int j = (Dp / TL::V) * TL::V; j < Dp; ++j) a[q][j] = src[j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < Dp; ++j) {
        T v = (row < D && j < D) ? src[j] : T(0);
        a[q][j] = (row >= D && j == row) ? T(1) : v;
      }
    }
  }
}

// this lane's rows -> LDS image (only the D x D part)
template <typename T, int Dp, int G, int R>
__device__ __forceinline__ void tile_rows2lds(T* __restrict__ lds, int mloc, int lig, int D, const T (&a)[R][Dp]) {
  using TL = Tile<T, Dp, G>;
#pragma unroll
  for (int q = 0; q < TL::R; ++q) {
    const int row = lig + G * q;
    T* dst = &lds[mloc * TL::MS + row * TL::RS];
    if (D == Dp) {
#pragma unroll
      for (int c = 0; c < Dp / TL::V; ++c) {
        typename TL::vec_t v;
#pragma unroll
        for (int u = 0; u < TL::V; ++u) v[u] = a[q][c * TL::V + u];
        *reinterpret_cast<typename TL::vec_t*>(&dst[c * TL::V]) = v;
      }
      if constexpr (Dp % TL::V != 0) {
#pragma unroll
        for (int j = (Dp / TL::V) * TL::V; j < Dp; ++j) dst[j] = a[q][j];
      }
    } else if (row < D) {
#pragma unroll
      for (int j = 0; j < Dp; ++j)
        if (j < D) dst[j] = a[q][j];
    }
  }
}

// ---------------------------------------------------------------- in-register Gauss-Jordan
// In-place inverse of the (symmetric, normally SPD) matrix whose rows lig+G*q live in a[q][*].
// No pivoting (SPD => pivots are the positive Schur complements).  `ld` accumulates log|det| and
// the determinant's sign so that the caller can mirror Tensor.logdet() (NaN for det < 0).
template <typename T, int Dp, int G, int R>
__device__ __forceinline__ void gj_inverse(T (&a)[R][Dp], int lig, LogDet<T>& ld) {
  static_assert(R * G == Dp, "rows per lane x lanes per matrix must cover the padded dim");
  static_for<0, Dp>([&](auto K) {
    constexpr int k = decltype(K)::value;
    constexpr int src = k % G;   // lane (in group) that owns the pivot row
    constexpr int slot = k / G;  // ... and the slot it sits in
    const bool owner = (G == 1) || (lig == src);
    const T d = bcast<G, src>(a[slot][k]);
    ld.mul(d);
    const T p = rcp_nr(d);
    const T s = owner ? p : T(1);
    // the owner scales its pivot row (x1 elsewhere: exact)
#pragma unroll
    for (int j = 0; j < Dp; ++j)
      if (j != k) a[slot][j] *= s;
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const bool is_piv = owner && (q == slot);
      const T f = a[q][k];
      const T nf = is_piv ? T(0) : -f;
#pragma unroll
      for (int j = 0; j < Dp; ++j)
        if (j != k) a[q][j] = xfma(nf, bcast<G, src>(a[slot][j]), a[q][j]);
      a[q][k] = is_piv ? p : nf * p;
    }
  });
}

}  // namespace vbmp
