// K11: log-space forward-backward of a discrete HMM (ref models/HMM.py:72-105), the role chain of
// DynamicMarkovBlanketDiscovery.  One launch replaces the reference's two Python loops over T.
// A chain (one series x observable) is owned by Kp lanes (Kp = K padded to a power of two); lane j keeps
// column j of the log transition matrix, of the pair-statistic accumulator SEzz and entry j of the message in
// registers; the K-vectors / K x K pair logits that must cross lanes go through a small per-chain LDS buffer.
// Time is sequential inside the kernel.  gfx950 only.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vbmp_dispatch.h"
#include "../../include/vbmp_hip.h"

namespace vbmp {

template <typename T>
__device__ __forceinline__ T neg_inf() { return -INFINITY; }

// wave-level LDS ordering (LDS address space only)
__device__ __forceinline__ void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

// log-sum-exp of a K-vector held one entry per lane of the chain (lanes j >= K pass -inf): every lane exponentiates
// its OWN entry once and the sum is read back from LDS (the fp64 exp is ~50 instructions; K of them per lane was the
// bulk of this kernel).  buf: Kp words of the chain's LDS.  Same value on every lane; -inf safe.
template <typename T>
__device__ __forceinline__ T lse_lanes(T mine, T* buf, int K, int j) {
  buf[j] = mine;
  wsync();
  T m = neg_inf<T>();
  for (int i = 0; i < K; ++i) m = buf[i] > m ? buf[i] : m;
  wsync();
  if (!(m > neg_inf<T>())) return m;
  buf[j] = (j < K) ? exp(mine - m) : T(0);
  wsync();
  T s = T(0);
  for (int i = 0; i < K; ++i) s += buf[i];
  wsync();
  return m + log(s);
}

template <typename T, int Kp>
__global__ __launch_bounds__(64) void k_hmm_fb(const T* __restrict__ logits, const T* __restrict__ trans,
                                               const T* __restrict__ init, int64_t Tn, int64_t C, int64_t NB, int K,
                                               T ptemp, T* __restrict__ p, T* __restrict__ SEzz, T* __restrict__ SEz0,
                                               T* __restrict__ logZ) {
  constexpr int CPW = 64 / Kp;  // chains per wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  const int lane = threadIdx.x, cl = lane / Kp, j = lane % Kp;
  const int64_t c = (int64_t)blockIdx.x * CPW + cl;
  const bool live = (c < C) && (j < K);
  const int64_t cc = c < C ? c : C - 1;
  const int64_t b = cc % NB;
  T* vec = smem + cl * (2 * Kp + Kp * Kp);  // [Kp] message exchange, [Kp] second vector, [Kp*Kp] pair logits
  T* vec2 = vec + Kp;
  T* mat = vec2 + Kp;
  const T NI = neg_inf<T>();
  // column j of the log transition matrix and entry j of the initial log probabilities
  T tr[Kp], acc[Kp];
#pragma unroll
  for (int i = 0; i < Kp; ++i) {
    tr[i] = (i < K && j < K) ? trans[(b * K + i) * K + j] : NI;
    acc[i] = T(0);
  }
  const T in_j = (j < K) ? init[b * K + j] : NI;
  const T* lg = logits + cc * K + (j < K ? j : 0);  // element (t, c, j) at lg[t*C*K]
  T* pj = p + cc * K + (j < K ? j : 0);
  const int64_t ts = C * K;

  // ---------------------------------------------------------------- forward (:77-80)
  T prev = in_j;
  for (int64_t t = 0; t < Tn; ++t) {
    vec[j] = prev;
    wsync();
    T m = NI;
#pragma unroll
    for (int i = 0; i < Kp; ++i) {
      const T v = vec[i] + tr[i];
      m = (i < K && v > m) ? v : m;
    }
    T s = T(0);
#pragma unroll
    for (int i = 0; i < Kp; ++i)
      if (i < K) s += (m > NI) ? exp(vec[i] + tr[i] - m) : T(0);
    prev = ((m > NI) ? m + log(s) : NI) + (j < K ? lg[t * ts] : T(0));
    if (j >= K) prev = NI;
    if (live) pj[t * ts] = prev;  // the p buffer holds the filtered logits until the backward sweep
    wsync();
  }
  // logZ and normalisation (:81-83)
  const T lz = lse_lanes<T>(prev, vec, K, j);
  T nxt = prev - lz;  // smoothed (= filtered) message at T-1

  // softmax with temperature of one message held one entry per lane (:100-101)
  auto emit = [&](int64_t t, T mine) {
    vec2[j] = mine;
    wsync();
    T mx = NI;
    for (int i = 0; i < K; ++i) mx = vec2[i] > mx ? vec2[i] : mx;
    wsync();
    const T e = (j < K) ? exp((mine - mx) / ptemp) : T(0);
    vec2[j] = e;
    wsync();
    T den = T(0);
    for (int i = 0; i < K; ++i) den += vec2[i];
    if (live) pj[t * ts] = e / den;
    wsync();
  };
  emit(Tn - 1, nxt);

  // ---------------------------------------------------------------- backward smoothing (:85-98)
  for (int64_t t = Tn - 2; t >= -1; --t) {
    // source message: filtered logits at t (normalised by logZ), or the initial distribution for the last step
    const T src = (t >= 0) ? ((live ? pj[t * ts] : NI) - lz) : in_j;
    vec[j] = (j < K) ? src : NI;
    vec2[j] = nxt;
    wsync();
    // column j: temp[i] = src[i] + tr[i][j]; xi[i][j] = temp[i] - lse_i temp + nxt[j]
    T m = NI;
#pragma unroll
    for (int i = 0; i < Kp; ++i) {
      const T v = vec[i] + tr[i];
      m = (i < K && v > m) ? v : m;
    }
    T s = T(0);
#pragma unroll
    for (int i = 0; i < Kp; ++i)
      if (i < K) s += (m > NI) ? exp(vec[i] + tr[i] - m) : T(0);
    const T cn = (m > NI) ? m + log(s) : NI;
    T xi[Kp];
#pragma unroll
    for (int i = 0; i < Kp; ++i) {
      const T v = vec[i] + tr[i];
      xi[i] = (i < K && j < K && v > NI && cn > NI) ? (v - cn) + nxt : NI;
    }
    // ONE exponentiation per pair: e_ij = exp(xi_ij - G) with G the largest pair logit of the step.  Row sums of e
    // give the new message, their total the normaliser, e / total the pair posterior (the reference's three
    // log-sum-exps over the same K x K logits, :88-95, exponentiate each pair three times).
    T cm = NI;
#pragma unroll
    for (int i = 0; i < Kp; ++i) cm = (i < K && xi[i] > cm) ? xi[i] : cm;
    vec[j] = (j < K) ? cm : NI;
    wsync();
    T G = NI;
    for (int i = 0; i < K; ++i) G = vec[i] > G ? vec[i] : G;
    T e[Kp];
#pragma unroll
    for (int i = 0; i < Kp; ++i) {
      e[i] = (i < K && j < K && xi[i] > NI) ? exp(xi[i] - G) : T(0);
      mat[i * Kp + j] = e[i];
    }
    wsync();
    T rs = T(0);  // row j of e: unnormalised posterior of state j at the source time
    for (int jj = 0; jj < K; ++jj) rs += mat[j * Kp + jj];
    const T newmsg = (j < K && rs > T(0)) ? G + log(rs) : NI;
    vec2[j] = (j < K) ? rs : T(0);
    wsync();
    T total = T(0);
    for (int i = 0; i < K; ++i) total += vec2[i];
    const bool any = total > T(0);  // false only when every pair is forbidden
    const T all = any ? G + log(total) : NI;
    const T inv_total = any ? T(1) / total : T(0);
#pragma unroll
    for (int i = 0; i < Kp; ++i)
      if (i < K) acc[i] += e[i] * inv_total;
    wsync();
    if (t >= 0) {
      nxt = newmsg;
      emit(t, newmsg);
    } else if (live) {
      SEz0[cc * K + j] = exp(newmsg - all);  // softmax of the initial-state message (:96-97)
    }
  }
  if (live) {
#pragma unroll
    for (int i = 0; i < Kp; ++i)
      if (i < K) SEzz[(cc * K + i) * K + j] = acc[i];
    if (j == 0) logZ[cc] = lz;
  }
}

template <typename T, int Kp>
static int launch_hmm(const T* logits, const T* trans, const T* init, int64_t Tn, int64_t C, int64_t NB, int K, T ptemp,
                      T* p, T* SEzz, T* SEz0, T* logZ, hipStream_t st) {
  constexpr int CPW = 64 / Kp;
  const int64_t blocks = (C + CPW - 1) / CPW;
  const size_t smem = (size_t)CPW * (2 * Kp + Kp * Kp) * sizeof(T);
  hipLaunchKernelGGL((k_hmm_fb<T, Kp>), dim3((unsigned)blocks), dim3(64), smem, st, logits, trans, init, Tn, C, NB, K,
                     ptemp, p, SEzz, SEz0, logZ);
  return hipGetLastError() == hipSuccess ? 0 : VBMP_ERR_LAUNCH;
}

template <typename T>
static int hmm_dispatch(const T* logits, const T* trans, const T* init, int64_t Tn, int64_t C, int64_t NB, int K,
                        T ptemp, T* p, T* SEzz, T* SEz0, T* logZ, void* stream) {
  if (C == 0) return 0;
  if (!logits || !trans || !init || !p || !SEzz || !SEz0 || !logZ || Tn < 1 || C < 0 || NB < 1 || K < 1 ||
      K > VBMP_HMM_MAX_K)
    return VBMP_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (K <= 2) return launch_hmm<T, 2>(logits, trans, init, Tn, C, NB, K, ptemp, p, SEzz, SEz0, logZ, st);
  if (K <= 4) return launch_hmm<T, 4>(logits, trans, init, Tn, C, NB, K, ptemp, p, SEzz, SEz0, logZ, st);
  if (K <= 8) return launch_hmm<T, 8>(logits, trans, init, Tn, C, NB, K, ptemp, p, SEzz, SEz0, logZ, st);
  if (K <= 16) return launch_hmm<T, 16>(logits, trans, init, Tn, C, NB, K, ptemp, p, SEzz, SEz0, logZ, st);
  if (K <= 32) return launch_hmm<T, 32>(logits, trans, init, Tn, C, NB, K, ptemp, p, SEzz, SEz0, logZ, st);
  return launch_hmm<T, 64>(logits, trans, init, Tn, C, NB, K, ptemp, p, SEzz, SEz0, logZ, st);
}

}  // namespace vbmp

extern "C" {
int vbmp_hmm_forward_backward_f64(const double* logits, const double* trans, const double* init, int64_t Tn, int64_t C,
                                  int64_t NB, int K, double ptemp, double* p, double* SEzz, double* SEz0, double* logZ,
                                  void* stream) {
  return vbmp::hmm_dispatch<double>(logits, trans, init, Tn, C, NB, K, ptemp, p, SEzz, SEz0, logZ, stream);
}
int vbmp_hmm_forward_backward_f32(const float* logits, const float* trans, const float* init, int64_t Tn, int64_t C,
                                  int64_t NB, int K, float ptemp, float* p, float* SEzz, float* SEz0, float* logZ,
                                  void* stream) {
  return vbmp::hmm_dispatch<float>(logits, trans, init, Tn, C, NB, K, ptemp, p, SEzz, SEz0, logZ, stream);
}
}
