// Device building blocks for the conjugate-update kernels (gfx950 / CDNA4 only).
//
// Layout idea ("row-per-lane tiles"): a 64-lane wavefront works on a TILE of MPW = 64/G
// consecutive batch elements.  Each D x D matrix (D <= Dp, Dp a power of two) is owned by a group
// of G lanes; lane `lig` of the group keeps rows lig, lig+G, ... (R = Dp/G of them) entirely in
// registers.  Global memory is always touched with linear, 16-byte-per-lane, full-cache-line
// accesses of the tile's contiguous bytes; the (coalesced <-> row-per-lane) transposition goes
// through a padded, per-wave LDS image.  Dense elimination then runs out of registers; the pivot
// row reaches the other lanes of the group INSIDE the FMA (v_fmac_*_dpp row_newbcast for G=16,
// quad_perm for G=4), via v_readlane for G=64, and is simply there for G=1.
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
// All DPP instructions are issued from inline asm that begins with `s_nop 1`: a DPP operand read
// needs 2 wait states after the VALU write of that register, and hipcc pads nothing around asm.
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

// hazard pad in front of a GROUP of DPP FMAs (a whole row update).  Empty: the rows a group reads through its DPP operand
// were written at least a pivot step earlier; that no VALU write of a source register lands within 2 wait states of its
// DPP read is proven on the final ISA by tools/check_dpp_hazards.py (tests/test_dpp_hazards.py), not assumed.
#ifndef VBMP_GROUP_PAD
#define VBMP_GROUP_PAD ""
#endif
// ... in front of a group that reads ITS OWN destination row through the DPP operand (the compiler has been seen to
// place a register copy of that row right in front of such a group), and in front of a single DPP FMA
#ifndef VBMP_SELF_PAD
#define VBMP_SELF_PAD VBMP_GROUP_PAD
#endif
#ifndef VBMP_SINGLE_PAD
#define VBMP_SINGLE_PAD "s_nop 1\n\t"
#endif
#define VBMP_DPP16 "row_newbcast:%c[src] row_mask:0xf bank_mask:0xf"
#define VBMP_DPP4 "quad_perm:[%c[src],%c[src],%c[src],%c[src]] row_mask:0xf bank_mask:0xf"

// value of `v` held by lane SRC of this lane's G-lane group
template <int G, int SRC>
__device__ __forceinline__ double bcast(double v) {
  static_assert(SRC >= 0 && SRC < G, "source lane outside the group");
  if constexpr (G == 1) {
    return v;
  } else if constexpr (G == 16) {
    double r;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %[r], %[v] " VBMP_DPP16 : [r] "=&v"(r) : [v] "v"(v), [src] "n"(SRC));
    return r;
  } else if constexpr (G == 4) {
    // no 64-bit quad_perm: move the two halves
    union { double d; int i[2]; } a, b;
    a.d = v;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %[r0], %[v0] " VBMP_DPP4 "\n\tv_mov_b32_dpp %[r1], %[v1] " VBMP_DPP4
                 : [r0] "=&v"(b.i[0]), [r1] "=&v"(b.i[1])
                 : [v0] "v"(a.i[0]), [v1] "v"(a.i[1]), [src] "n"(SRC));
    return b.d;
  } else if constexpr (G == 64) {
    return readlane_any(v, SRC);
  } else {
    return __shfl(v, SRC, G);
  }
}
template <int G, int SRC>
__device__ __forceinline__ float bcast(float v) {
  static_assert(SRC >= 0 && SRC < G, "source lane outside the group");
  if constexpr (G == 1) {
    return v;
  } else if constexpr (G == 16) {
    float r;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %[r], %[v] " VBMP_DPP16 : [r] "=&v"(r) : [v] "v"(v), [src] "n"(SRC));
    return r;
  } else if constexpr (G == 4) {
    float r;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %[r], %[v] " VBMP_DPP4 : [r] "=&v"(r) : [v] "v"(v), [src] "n"(SRC));
    return r;
  } else if constexpr (G == 64) {
    return readlane_any(v, SRC);
  } else {
    return __shfl(v, SRC, G);
  }
}

__device__ __forceinline__ double xfma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float xfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// acc += f * (value of `v` on lane SRC of the group): one instruction for G in {4,16}
template <int G, int SRC>
__device__ __forceinline__ void fmac_bcast(double& acc, double v, double f) {
  if constexpr (G == 16) {
    asm volatile(VBMP_SINGLE_PAD "v_fmac_f64_dpp %[a], %[v], %[f] " VBMP_DPP16 : [a] "+v"(acc) : [v] "v"(v), [f] "v"(f), [src] "n"(SRC));
  } else {
    acc = xfma(bcast<G, SRC>(v), f, acc);
  }
}
template <int G, int SRC>
__device__ __forceinline__ void fmac_bcast(float& acc, float v, float f) {
  if constexpr (G == 16) {
    asm volatile(VBMP_SINGLE_PAD "v_fmac_f32_dpp %[a], %[v], %[f] " VBMP_DPP16 : [a] "+v"(acc) : [v] "v"(v), [f] "v"(f), [src] "n"(SRC));
  } else if constexpr (G == 4) {
    asm volatile(VBMP_SINGLE_PAD "v_fmac_f32_dpp %[a], %[v], %[f] " VBMP_DPP4 : [a] "+v"(acc) : [v] "v"(v), [f] "v"(f), [src] "n"(SRC));
  } else {
    acc = xfma(bcast<G, SRC>(v), f, acc);
  }
}

// acc[j] += f * (piv[j] on lane SRC), j = 0..3: one s_nop + four DPP FMAs
template <int G, int SRC>
__device__ __forceinline__ void fmac_bcast4(double* acc, const double* piv, double f) {
  if constexpr (G == 16) {
    asm volatile(VBMP_GROUP_PAD
                 "v_fmac_f64_dpp %[a0], %[p0], %[f] " VBMP_DPP16 "\n\t"
                 "v_fmac_f64_dpp %[a1], %[p1], %[f] " VBMP_DPP16 "\n\t"
                 "v_fmac_f64_dpp %[a2], %[p2], %[f] " VBMP_DPP16 "\n\t"
                 "v_fmac_f64_dpp %[a3], %[p3], %[f] " VBMP_DPP16
                 : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3])
                 : [p0] "v"(piv[0]), [p1] "v"(piv[1]), [p2] "v"(piv[2]), [p3] "v"(piv[3]), [f] "v"(f), [src] "n"(SRC));
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = xfma(bcast<G, SRC>(piv[j]), f, acc[j]);
  }
}
template <int G, int SRC>
__device__ __forceinline__ void fmac_bcast4(float* acc, const float* piv, float f) {
  if constexpr (G == 16) {
    asm volatile(VBMP_GROUP_PAD
                 "v_fmac_f32_dpp %[a0], %[p0], %[f] " VBMP_DPP16 "\n\t"
                 "v_fmac_f32_dpp %[a1], %[p1], %[f] " VBMP_DPP16 "\n\t"
                 "v_fmac_f32_dpp %[a2], %[p2], %[f] " VBMP_DPP16 "\n\t"
                 "v_fmac_f32_dpp %[a3], %[p3], %[f] " VBMP_DPP16
                 : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3])
                 : [p0] "v"(piv[0]), [p1] "v"(piv[1]), [p2] "v"(piv[2]), [p3] "v"(piv[3]), [f] "v"(f), [src] "n"(SRC));
  } else if constexpr (G == 4) {
    asm volatile(VBMP_GROUP_PAD
                 "v_fmac_f32_dpp %[a0], %[p0], %[f] " VBMP_DPP4 "\n\t"
                 "v_fmac_f32_dpp %[a1], %[p1], %[f] " VBMP_DPP4 "\n\t"
                 "v_fmac_f32_dpp %[a2], %[p2], %[f] " VBMP_DPP4 "\n\t"
                 "v_fmac_f32_dpp %[a3], %[p3], %[f] " VBMP_DPP4
                 : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3])
                 : [p0] "v"(piv[0]), [p1] "v"(piv[1]), [p2] "v"(piv[2]), [p3] "v"(piv[3]), [f] "v"(f), [src] "n"(SRC));
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = xfma(bcast<G, SRC>(piv[j]), f, acc[j]);
  }
}

// acc[j] += f * (acc[j] on lane SRC), j = 0..3  (the row is its own pivot slot: the DPP source and the
// destination are the same register, so it must be ONE asm operand or the compiler copies it first)
template <int G, int SRC>
__device__ __forceinline__ void fmac_self4(double* acc, double f) {
  if constexpr (G == 16) {
    asm volatile(VBMP_SELF_PAD
                 "v_fmac_f64_dpp %[a0], %[a0], %[f] " VBMP_DPP16 "\n\t"
                 "v_fmac_f64_dpp %[a1], %[a1], %[f] " VBMP_DPP16 "\n\t"
                 "v_fmac_f64_dpp %[a2], %[a2], %[f] " VBMP_DPP16 "\n\t"
                 "v_fmac_f64_dpp %[a3], %[a3], %[f] " VBMP_DPP16
                 : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3])
                 : [f] "v"(f), [src] "n"(SRC));
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = xfma(bcast<G, SRC>(acc[j]), f, acc[j]);
  }
}
template <int G, int SRC>
__device__ __forceinline__ void fmac_self4(float* acc, float f) {
  if constexpr (G == 16) {
    asm volatile(VBMP_SELF_PAD
                 "v_fmac_f32_dpp %[a0], %[a0], %[f] " VBMP_DPP16 "\n\t"
                 "v_fmac_f32_dpp %[a1], %[a1], %[f] " VBMP_DPP16 "\n\t"
                 "v_fmac_f32_dpp %[a2], %[a2], %[f] " VBMP_DPP16 "\n\t"
                 "v_fmac_f32_dpp %[a3], %[a3], %[f] " VBMP_DPP16
                 : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3])
                 : [f] "v"(f), [src] "n"(SRC));
  } else if constexpr (G == 4) {
    asm volatile(VBMP_SELF_PAD
                 "v_fmac_f32_dpp %[a0], %[a0], %[f] " VBMP_DPP4 "\n\t"
                 "v_fmac_f32_dpp %[a1], %[a1], %[f] " VBMP_DPP4 "\n\t"
                 "v_fmac_f32_dpp %[a2], %[a2], %[f] " VBMP_DPP4 "\n\t"
                 "v_fmac_f32_dpp %[a3], %[a3], %[f] " VBMP_DPP4
                 : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3])
                 : [f] "v"(f), [src] "n"(SRC));
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = xfma(bcast<G, SRC>(acc[j]), f, acc[j]);
  }
}
// eight at a time (18 asm operands): halves the number of hazard pads per row
template <int G, int SRC>
__device__ __forceinline__ void fmac_self8(double* acc, double f) {
  asm volatile(VBMP_SELF_PAD
               "v_fmac_f64_dpp %[a0], %[a0], %[f] " VBMP_DPP16 "\n\t"
               "v_fmac_f64_dpp %[a1], %[a1], %[f] " VBMP_DPP16 "\n\t"
               "v_fmac_f64_dpp %[a2], %[a2], %[f] " VBMP_DPP16 "\n\t"
               "v_fmac_f64_dpp %[a3], %[a3], %[f] " VBMP_DPP16 "\n\t"
               "v_fmac_f64_dpp %[a4], %[a4], %[f] " VBMP_DPP16 "\n\t"
               "v_fmac_f64_dpp %[a5], %[a5], %[f] " VBMP_DPP16 "\n\t"
               "v_fmac_f64_dpp %[a6], %[a6], %[f] " VBMP_DPP16 "\n\t"
               "v_fmac_f64_dpp %[a7], %[a7], %[f] " VBMP_DPP16
               : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3]), [a4] "+v"(acc[4]),
                 [a5] "+v"(acc[5]), [a6] "+v"(acc[6]), [a7] "+v"(acc[7])
               : [f] "v"(f), [src] "n"(SRC));
}
template <int G, int SRC>
__device__ __forceinline__ void fmac_self8(float* acc, float f) {
  asm volatile(VBMP_SELF_PAD
               "v_fmac_f32_dpp %[a0], %[a0], %[f] " VBMP_DPP16 "\n\t"
               "v_fmac_f32_dpp %[a1], %[a1], %[f] " VBMP_DPP16 "\n\t"
               "v_fmac_f32_dpp %[a2], %[a2], %[f] " VBMP_DPP16 "\n\t"
               "v_fmac_f32_dpp %[a3], %[a3], %[f] " VBMP_DPP16 "\n\t"
               "v_fmac_f32_dpp %[a4], %[a4], %[f] " VBMP_DPP16 "\n\t"
               "v_fmac_f32_dpp %[a5], %[a5], %[f] " VBMP_DPP16 "\n\t"
               "v_fmac_f32_dpp %[a6], %[a6], %[f] " VBMP_DPP16 "\n\t"
               "v_fmac_f32_dpp %[a7], %[a7], %[f] " VBMP_DPP16
               : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3]), [a4] "+v"(acc[4]),
                 [a5] "+v"(acc[5]), [a6] "+v"(acc[6]), [a7] "+v"(acc[7])
               : [f] "v"(f), [src] "n"(SRC));
}
// (NC: the leading columns to update -- Dp, or the true size of an unpadded compile-time instance: the identity padding beyond it
// never changes)
template <int G, int SRC, int Dp, typename T, int NC = Dp>
__device__ __forceinline__ void fmac_self_row(T (&acc)[Dp], T f) {
  constexpr int N8 = (G == 16) ? NC / 8 : 0;  // groups of eight, then of four, then single columns
  constexpr int N4 = (NC - 8 * N8) / 4;
#pragma unroll
  for (int c = 0; c < N8; ++c) fmac_self8<G, SRC>(&acc[8 * c], f);
#pragma unroll
  for (int c = 0; c < N4; ++c) fmac_self4<G, SRC>(&acc[8 * N8 + 4 * c], f);
#pragma unroll
  for (int j = 8 * N8 + 4 * N4; j < NC; ++j) acc[j] = xfma(bcast<G, SRC>(acc[j]), f, acc[j]);
}

// acc[0..NC) += f * (piv[0..NC) on lane SRC)
template <int G, int SRC, int Dp, typename T, int NC = Dp>
__device__ __forceinline__ void fmac_bcast_row(T (&acc)[Dp], const T (&piv)[Dp], T f) {
  constexpr int N4 = NC / 4;
#pragma unroll
  for (int c = 0; c < N4; ++c) fmac_bcast4<G, SRC>(&acc[4 * c], &piv[4 * c], f);
#pragma unroll
  for (int j = 4 * N4; j < NC; ++j) acc[j] = xfma(bcast<G, SRC>(piv[j]), f, acc[j]);
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
__device__ __forceinline__ double frexp_m(double d) { return __builtin_amdgcn_frexp_mant(d); }
__device__ __forceinline__ float frexp_m(float d) { return __builtin_amdgcn_frexp_mantf(d); }
__device__ __forceinline__ int frexp_e(double d) { return __builtin_amdgcn_frexp_exp(d); }
__device__ __forceinline__ int frexp_e(float d) { return __builtin_amdgcn_frexp_expf(d); }

// running det kept as (signed mantissa product, exponent sum): one log per matrix at the end.
// log of a negative product is NaN and of zero is -inf, which is exactly Tensor.logdet().
template <typename T>
struct LogDet {
  T mant = T(1);
  int expo = 0;
  bool allpos = true;  // every pivot > 0  <=>  the matrix is positive definite
  __device__ __forceinline__ void mul(T d) {
    allpos = allpos && (d > T(0));
    mant *= frexp_m(d);  // |m| in [0.5,1): at most 64 factors, cannot underflow
    expo += frexp_e(d);
  }
  __device__ __forceinline__ T value() const {
    return log(mant) + T(expo) * T(0.693147180559945309417232121458);
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

// wave-level ordering of LDS traffic (one wave owns its LDS image; no block barrier needed).
// The fences name the LDS address space only, so they cost an lgkmcnt wait and do NOT drain the
// global loads/stores that are still in flight.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

// LDS image offset (in elements) of the 16-byte chunk c = lane + 64 i of a full tile, split into a per-lane base and a
// compile-time constant per i -- written as (lane + 64 i) / cpr ... the compiler keeps one address register PER i (16 of
// them for a 32 x 32 fp32 tile, all loop invariants, all spilled).  A wave instruction covers RPI = 64 / cpr rows.
template <typename T, int Dp, int G>
struct TileChunk {
  using TL = Tile<T, Dp, G>;
  static constexpr int cpr = Dp / TL::V > 0 ? Dp / TL::V : 1;  // chunks per row
  static constexpr int RPI = 64 / cpr > 0 ? 64 / cpr : 1;      // rows per wave instruction
  static __device__ __forceinline__ int base(int lane) {
    const int r0 = lane / cpr, cc = lane % cpr;
    if constexpr (Dp % RPI == 0)
      return r0 * TL::RS + cc * TL::V;
    else  // RPI is a multiple of Dp: an instruction covers RPI / Dp whole matrices
      return (r0 / Dp) * TL::MS + (r0 % Dp) * TL::RS + cc * TL::V;
  }
  static constexpr int off(int i) {
    if constexpr (Dp % RPI == 0)
      return ((RPI * i) / Dp) * TL::MS + ((RPI * i) % Dp) * TL::RS;
    else
      return (RPI / Dp) * i * TL::MS;
  }
};

// global (linear, coalesced) -> per-wave LDS image [m][row][col] with padded strides.
// g points at the first matrix of the tile; matrices are `gstride` elements apart; `nm` valid.
// FULL: D == Dp, the tile is contiguous and 16-byte aligned (checked on the host) -> all index
// arithmetic folds to shifts and the 16-byte path is unconditional.
template <typename T, int Dp, int G, bool FULL>
__device__ __forceinline__ void tile_g2lds(const T* __restrict__ g, int64_t gstride, int nm, int Drt,
                                           T* __restrict__ lds, int lane, bool NT = false) {
  using TL = Tile<T, Dp, G>;
  const int D = FULL ? Dp : Drt;
  const int DD = D * D;
  if constexpr (FULL && (Dp % TL::V == 0)) {
    constexpr int cpr = Dp / TL::V;                                // chunks per row
    constexpr int NIT = (TL::MPW * Dp * Dp / TL::V + 63) / 64;     // full-tile trip count
    const int nchunk = nm * (Dp * Dp / TL::V);
    const typename TL::vec_t* gv = reinterpret_cast<const typename TL::vec_t*>(g);
    // branch-free: a partial last tile re-reads its final chunk and fills LDS slots nobody consumes
    typename TL::vec_t v[NIT];
    const typename TL::vec_t* gl = gv + lane;  // one per-lane pointer, constant offsets 64 i from it
    if (nm == TL::MPW) {  // wave-uniform: constant offsets from one base address; streamed once -> nt
#pragma unroll
      for (int i = 0; i < NIT; ++i) v[i] = NT ? __builtin_nontemporal_load(&gl[64 * i]) : gl[64 * i];
    } else {
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        const int c = lane + 64 * i;
        v[i] = gv[c < nchunk ? c : nchunk - 1];
      }
    }
    using TC = TileChunk<T, Dp, G>;
    T* lb = lds + TC::base(lane);
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      if (lane + 64 * i < TL::MPW * Dp * Dp / TL::V) *reinterpret_cast<typename TL::vec_t*>(lb + TC::off(i)) = v[i];
    }
  } else {
    // generic D: same shape as the FULL path -- a compile-time trip count (that of the padded tile, in groups of at
    // most 16), clamped loads issued ahead of the LDS writes, and the divisions by the runtime D done in float
    // (exact here: indices < 4096, divisors <= 64)
    const float inv_D = 1.0f / (float)D;
    const bool vec_ok = (gstride == DD) && (D % TL::V == 0) && ((reinterpret_cast<uintptr_t>(g) & 15) == 0);
    if (vec_ok) {
      constexpr int NIT = (TL::MPW * Dp * Dp / TL::V + 63) / 64;
      constexpr int GRP = NIT < 16 ? NIT : 16;
      const int nchunk = nm * DD / TL::V;  // chunks never straddle a row because D % V == 0
      const int cpr = D / TL::V;
      const float inv_cpr = 1.0f / (float)cpr;
      const typename TL::vec_t* gv = reinterpret_cast<const typename TL::vec_t*>(g);
#pragma unroll
      for (int i0 = 0; i0 < NIT; i0 += GRP) {
        if (64 * i0 >= nchunk) break;  // wave-uniform
        typename TL::vec_t v[GRP];
#pragma unroll
        for (int i = 0; i < GRP; ++i) {
          const int c = lane + 64 * (i0 + i);
          v[i] = gv[c < nchunk ? c : nchunk - 1];
        }
#pragma unroll
        for (int i = 0; i < GRP; ++i) {
          const int c = lane + 64 * (i0 + i);
          const int row_all = (int)(((float)c + 0.5f) * inv_cpr);
          const int cc = c - row_all * cpr;
          const int m = (int)(((float)row_all + 0.5f) * inv_D);
          const int row = row_all - m * D;
          if (c < nchunk) *reinterpret_cast<typename TL::vec_t*>(&lds[m * TL::MS + row * TL::RS + cc * TL::V]) = v[i];
        }
      }
    } else {
      constexpr int NIT = (TL::MPW * Dp * Dp + 63) / 64;
      constexpr int GRP = NIT < 16 ? NIT : 16;
      const int n = nm * DD;
      const float inv_DD = 1.0f / (float)DD;
#pragma unroll
      for (int i0 = 0; i0 < NIT; i0 += GRP) {
        if (64 * i0 >= n) break;  // wave-uniform
        T v[GRP];
#pragma unroll
        for (int i = 0; i < GRP; ++i) {
          const int e = lane + 64 * (i0 + i);
          const int ec = e < n ? e : n - 1;
          const int m = (int)(((float)ec + 0.5f) * inv_DD);
          v[i] = g[(int64_t)m * gstride + (ec - m * DD)];
        }
#pragma unroll
        for (int i = 0; i < GRP; ++i) {
          const int e = lane + 64 * (i0 + i);
          const int m = (int)(((float)e + 0.5f) * inv_DD);
          const int rem = e - m * DD;
          const int row = (int)(((float)rem + 0.5f) * inv_D);
          const int col = rem - row * D;
          if (e < n) lds[m * TL::MS + row * TL::RS + col] = v[i];
        }
      }
    }
  }
}

// per-wave LDS image -> global (linear, coalesced)
template <typename T, int Dp, int G, bool FULL>
__device__ __forceinline__ void tile_lds2g(T* __restrict__ g, int nm, int Drt, const T* __restrict__ lds, int lane,
                                           bool NT = false) {
  using TL = Tile<T, Dp, G>;
  const int D = FULL ? Dp : Drt;
  const int DD = D * D;
  if constexpr (FULL && (Dp % TL::V == 0)) {
    constexpr int cpr = Dp / TL::V;
    constexpr int NIT = (TL::MPW * Dp * Dp / TL::V + 63) / 64;
    const int nchunk = nm * (Dp * Dp / TL::V);
    typename TL::vec_t* gv = reinterpret_cast<typename TL::vec_t*>(g);
    typename TL::vec_t v[NIT];
    using TC = TileChunk<T, Dp, G>;
    const T* lb = lds + TC::base(lane);
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      if (lane + 64 * i < TL::MPW * Dp * Dp / TL::V) v[i] = *reinterpret_cast<const typename TL::vec_t*>(lb + TC::off(i));
    }
    typename TL::vec_t* gl = gv + lane;
    if (nm == TL::MPW) {  // wave-uniform: unpredicated stores, constant offsets; written once -> nt
#pragma unroll
      for (int i = 0; i < NIT; ++i)
        if (lane + 64 * i < TL::MPW * Dp * Dp / TL::V) {
          if (NT) __builtin_nontemporal_store(v[i], &gl[64 * i]);
          else gl[64 * i] = v[i];
        }
    } else {
#pragma unroll
      for (int i = 0; i < NIT; ++i)
        if (lane + 64 * i < nchunk) gv[lane + 64 * i] = v[i];
    }
  } else {
    const float inv_D = 1.0f / (float)D;
    const bool vec_ok = (D % TL::V == 0) && ((reinterpret_cast<uintptr_t>(g) & 15) == 0);
    if (vec_ok) {
      constexpr int NIT = (TL::MPW * Dp * Dp / TL::V + 63) / 64;
      const int nchunk = nm * DD / TL::V;
      const int cpr = D / TL::V;
      const float inv_cpr = 1.0f / (float)cpr;
      typename TL::vec_t* gv = reinterpret_cast<typename TL::vec_t*>(g);
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        const int c = lane + 64 * i;
        if (64 * i >= nchunk) break;  // wave-uniform
        const int row_all = (int)(((float)c + 0.5f) * inv_cpr);
        const int cc = c - row_all * cpr;
        const int m = (int)(((float)row_all + 0.5f) * inv_D);
        const int row = row_all - m * D;
        if (c < nchunk)
          gv[c] = *reinterpret_cast<const typename TL::vec_t*>(&lds[m * TL::MS + row * TL::RS + cc * TL::V]);
      }
    } else {
      constexpr int NIT = (TL::MPW * Dp * Dp + 63) / 64;
      const int n = nm * DD;
      const float inv_DD = 1.0f / (float)DD;
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        const int e = lane + 64 * i;
        if (64 * i >= n) break;  // wave-uniform
        const int m = (int)(((float)e + 0.5f) * inv_DD);
        const int rem = e - m * DD;
        const int row = (int)(((float)rem + 0.5f) * inv_D);
        const int col = rem - row * D;
        if (e < n) g[e] = lds[m * TL::MS + row * TL::RS + col];
      }
    }
  }
}

// LDS image -> this lane's rows.  Rows/cols >= D are padded with the identity.
template <typename T, int Dp, int G, int R, bool FULL>
__device__ __forceinline__ void tile_lds2rows(const T* __restrict__ lds, int mloc, int lig, int Drt, T (&a)[R][Dp]) {
  using TL = Tile<T, Dp, G>;
  const int D = FULL ? Dp : Drt;
#pragma unroll
  for (int q = 0; q < R; ++q) {
    const int row = lig + G * q;
    const T* src = &lds[mloc * TL::MS + row * TL::RS];
    if (FULL) {
      if constexpr (Dp % TL::V == 0) {
#pragma unroll
        for (int c = 0; c < Dp / TL::V; ++c) {
          typename TL::vec_t v = *reinterpret_cast<const typename TL::vec_t*>(&src[c * TL::V]);
#pragma unroll
          for (int u = 0; u < TL::V; ++u) a[q][c * TL::V + u] = v[u];
        }
      } else {
#pragma unroll
        for (int j = 0; j < Dp; ++j) a[q][j] = src[j];
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
template <typename T, int Dp, int G, int R, bool FULL>
__device__ __forceinline__ void tile_rows2lds(T* __restrict__ lds, int mloc, int lig, int Drt, const T (&a)[R][Dp]) {
  using TL = Tile<T, Dp, G>;
  const int D = FULL ? Dp : Drt;
#pragma unroll
  for (int q = 0; q < R; ++q) {
    const int row = lig + G * q;
    T* dst = &lds[mloc * TL::MS + row * TL::RS];
    if (FULL) {
      if constexpr (Dp % TL::V == 0) {
#pragma unroll
        for (int c = 0; c < Dp / TL::V; ++c) {
          typename TL::vec_t v;
#pragma unroll
          for (int u = 0; u < TL::V; ++u) v[u] = a[q][c * TL::V + u];
          *reinterpret_cast<typename TL::vec_t*>(&dst[c * TL::V]) = v;
        }
      } else {
#pragma unroll
        for (int j = 0; j < Dp; ++j) dst[j] = a[q][j];
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
// No pivoting (SPD => the pivots are the positive Schur complements).  The pivot row is never
// scaled in place: row r keeps the factor 1/pivot_r aside (`sc`) and every later update is linear
// in the row, so the scale is applied once at the end -- this removes Dp multiplies per step and
// lets every other row consume the UNSCALED pivot row straight out of the owner's registers
// through the DPP operand of the FMA.
// DX > 0: the matrix is DX x DX (compile-time) inside its identity-padded Dp x Dp image: only the DX pivots, the DX leading columns and
// the row slots that hold a row < DX are worked on -- the padding stays the identity by itself (its multipliers are exact zeros), and
// the DX x DX block comes out bit for bit as from the full elimination.  D = 20 in a 32-image: 800 instead of 2 048 FMAs per tile.
// Drt >= 0 (run-time size of an unpadded matrix, wave-uniform; the run-time-D instances): the pivots k >= Drt are skipped -- each is a
// pivot of 1 on a unit row, a step that changes nothing -- at the price of one scalar branch per pivot.
template <typename T, int Dp, int G, int R, int DX = 0>
__device__ __forceinline__ void gj_inverse(T (&a)[R][Dp], int lig, LogDet<T>& ld, int Drt = -1) {
  static_assert(R * G == Dp, "rows per lane x lanes per matrix must cover the padded dim");
  constexpr int DN = DX > 0 ? DX : Dp;
  T sc[R];
#pragma unroll
  for (int q = 0; q < R; ++q) sc[q] = T(1);
  static_for<0, DN>([&](auto K) {
    constexpr int k = decltype(K)::value;
    if (Drt >= 0 && k >= Drt) return;  // (folds away where Drt is the literal -1)
    constexpr int src = k % G;   // lane (in group) that owns the pivot row
    constexpr int slot = k / G;  // ... and the slot it sits in
    const bool owner = (G == 1) || (lig == src);
    const T d = bcast<G, src>(a[slot][k]);
    ld.mul(d);
    const T p = rcp_nr(d);
    sc[slot] = owner ? p : sc[slot];
    // rows in the pivot slot go last: their destination registers are this step's DPP sources
    static_for<0, R>([&](auto Q) {
      constexpr int q = (decltype(Q)::value + slot + 1) % R;
      if constexpr (q * G < DN) {  // (a slot of padding rows only: nothing to do)
        const bool is_piv = owner && (q == slot);
        const T nf = is_piv ? T(0) : -(a[q][k] * p);
        if constexpr (q == slot)
          fmac_self_row<G, src, Dp, T, DN>(a[q], nf);
        else
          fmac_bcast_row<G, src, Dp, T, DN>(a[q], a[slot], nf);
        a[q][k] = is_piv ? T(1) : nf;
      }
    });
  });
#pragma unroll
  for (int q = 0; q < R; ++q)
    if (q * G < DN) {
#pragma unroll
      for (int j = 0; j < DN; ++j) a[q][j] *= sc[q];
    }
}

// ---------------------------------------------------------------- quadratic form + logdet only
// q = e^T A^-1 e and log det A by forward elimination (LDL^T) with e carried as an extra column:
// no back-substitution, no inverse, and at step k only the columns j > k are touched, i.e. about half of
// the FMAs of the full Gauss-Jordan.  `a` and `e` are destroyed.  e is distributed like the rows
// (entry row on the lane that owns the row).
template <typename T, int Dp, int G, int R>
__device__ __forceinline__ T elim_quad(T (&a)[R][Dp], T (&e)[R], int lig, LogDet<T>& ld) {
  static_assert(R * G == Dp, "rows per lane x lanes per matrix must cover the padded dim");
  T qacc = T(0);
  static_for<0, Dp>([&](auto K) {
    constexpr int k = decltype(K)::value;
    constexpr int src = k % G, slot = k / G;
    const bool owner = (G == 1) || (lig == src);
    const T d = bcast<G, src>(a[slot][k]);
    ld.mul(d);
    const T p = rcp_nr(d);
    const T ek = e[slot];
    qacc += owner ? ek * ek * p : T(0);
    static_for<0, R>([&](auto Q) {
      constexpr int q = (decltype(Q)::value + slot + 1) % R;  // rows of the pivot slot last (DPP read-after-write)
      const int row = lig + G * q;
      const T nf = (row > k) ? -(a[q][k] * p) : T(0);
      fmac_bcast<G, src>(e[q], e[slot], nf);
      constexpr int c0 = (k + 1) / 4;  // 4-column chunks that still hold a column j > k
      if constexpr (Dp % 4 == 0) {
#pragma unroll
        for (int c = c0; c < Dp / 4; ++c) {
          if constexpr (q == slot)
            fmac_self4<G, src>(&a[q][4 * c], nf);
          else
            fmac_bcast4<G, src>(&a[q][4 * c], &a[slot][4 * c], nf);
        }
      } else {
#pragma unroll
        for (int j = k + 1; j < Dp; ++j) a[q][j] = xfma(bcast<G, src>(a[slot][j]), nf, a[q][j]);
      }
    });
  });
  // sum of the owners' contributions over the lane group
  if constexpr (G > 1) {
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) qacc += __shfl_xor(qacc, off, G);
  }
  return qacc;
}

// the same elimination carrying TWO right-hand sides: returns e^T A^-1 e and writes f^T A^-1 f to qf
template <typename T, int Dp, int G, int R>
__device__ __forceinline__ T elim_quad2(T (&a)[R][Dp], T (&e)[R], T (&f)[R], int lig, LogDet<T>& ld, T& qf) {
  static_assert(R * G == Dp, "rows per lane x lanes per matrix must cover the padded dim");
  T qe = T(0), qq = T(0);
  static_for<0, Dp>([&](auto K) {
    constexpr int k = decltype(K)::value;
    constexpr int src = k % G, slot = k / G;
    const bool owner = (G == 1) || (lig == src);
    const T d = bcast<G, src>(a[slot][k]);
    ld.mul(d);
    const T p = rcp_nr(d);
    const T ek = e[slot], fk = f[slot];
    qe += owner ? ek * ek * p : T(0);
    qq += owner ? fk * fk * p : T(0);
    static_for<0, R>([&](auto Q) {
      constexpr int q = (decltype(Q)::value + slot + 1) % R;  // rows of the pivot slot last (DPP read-after-write)
      const int row = lig + G * q;
      const T nf = (row > k) ? -(a[q][k] * p) : T(0);
      fmac_bcast<G, src>(e[q], e[slot], nf);
      fmac_bcast<G, src>(f[q], f[slot], nf);
      constexpr int c0 = (k + 1) / 4;
      if constexpr (Dp % 4 == 0) {
#pragma unroll
        for (int c = c0; c < Dp / 4; ++c) {
          if constexpr (q == slot)
            fmac_self4<G, src>(&a[q][4 * c], nf);
          else
            fmac_bcast4<G, src>(&a[q][4 * c], &a[slot][4 * c], nf);
        }
      } else {
#pragma unroll
        for (int j = k + 1; j < Dp; ++j) a[q][j] = xfma(bcast<G, src>(a[slot][j]), nf, a[q][j]);
      }
    });
  });
  if constexpr (G > 1) {
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) {
      qe += __shfl_xor(qe, off, G);
      qq += __shfl_xor(qq, off, G);
    }
  }
  qf = qq;
  return qe;
}

}  // namespace vbmp
