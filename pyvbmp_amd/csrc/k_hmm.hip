// K11: forward-backward of a discrete HMM with log-space inputs and outputs (ref models/HMM.py:72-105), the role chain
// of DynamicMarkovBlanketDiscovery.  One launch replaces the reference's two Python loops over T.
// A chain (one series x observable) is owned by Kp lanes (Kp = K padded to a power of two); lane j keeps column j of
// the transition matrix, of the pair-statistic accumulator SEzz and entry j of the message in registers; K-vectors
// that every lane must see and the K x K pair weights (written by columns, summed by rows) go through a small
// per-chain LDS buffer, scalar reductions through DPP butterflies.  Time is sequential inside the kernel.  gfx950 only.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vbmp_dispatch.h"
#include "../../include/vbmp_hip.h"

namespace vbmp {

template <typename T>
__device__ __forceinline__ T neg_inf() { return -INFINITY; }
template <typename T>
__device__ __forceinline__ T nan_of() { return T(NAN); }

// wave-level LDS ordering (LDS address space only)
__device__ __forceinline__ void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

// max / sum over the Kp lanes of a chain as an xor butterfly (log2(Kp) steps instead of a K-step loop of dependent
// LDS reads; every pair is combined symmetrically, so all lanes of the chain end with the SAME bits).  The four steps
// inside a 16-lane row are DPP moves (quad_perm, row_half_mirror, row_mirror: ~10 cycles each); only the steps across
// rows (Kp = 32, 64) go through the LDS crossbar (ds_bpermute, ~10x that).
template <int CTRL>
__device__ __forceinline__ float dpp_xchg(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_xchg(double v) {
  union { double d; int i[2]; } a, r;
  a.d = v;
  r.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], CTRL, 0xf, 0xf, true);
  r.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], CTRL, 0xf, 0xf, true);
  return r.d;
}
template <int Kp, typename T, typename F>
__device__ __forceinline__ T grp_butterfly(T v, F op) {
  if constexpr (Kp >= 2) v = op(v, dpp_xchg<0xB1>(v));   // quad_perm:[1,0,3,2]  (lane ^ 1)
  if constexpr (Kp >= 4) v = op(v, dpp_xchg<0x4E>(v));   // quad_perm:[2,3,0,1]  (lane ^ 2)
  if constexpr (Kp >= 8) v = op(v, dpp_xchg<0x141>(v));  // row_half_mirror: 7 - lane, = lane ^ 4 once quads agree
  if constexpr (Kp >= 16) v = op(v, dpp_xchg<0x140>(v)); // row_mirror: 15 - lane, = lane ^ 8 once octets agree
  if constexpr (Kp >= 32) v = op(v, __shfl_xor(v, 16, 64));
  if constexpr (Kp >= 64) v = op(v, __shfl_xor(v, 32, 64));
  return v;
}
template <int Kp, typename T>
__device__ __forceinline__ T grp_max(T v) {
  return grp_butterfly<Kp>(v, [](T a, T b) { return b > a ? b : a; });
}
template <int Kp, typename T>
__device__ __forceinline__ T grp_sum(T v) {
  return grp_butterfly<Kp>(v, [](T a, T b) { return a + b; });
}

// log-sum-exp of a K-vector held one entry per lane of the chain (lanes j >= K pass -inf): every lane exponentiates
// its OWN entry once.  Same value on every lane; -inf safe.
template <int Kp, typename T>
__device__ __forceinline__ T lse_lanes(T mine) {
  const T m = grp_max<Kp>(mine);
  if (!(m > neg_inf<T>())) return m;
  return m + log(grp_sum<Kp>(mine > neg_inf<T>() ? exp(mine - m) : T(0)));
}

// smallest exp(transition) the probability-space step accepts as a factor (below it the product a_i A_ij would lose bits)
template <typename T> __device__ __forceinline__ T hmm_tiny();
template <> __device__ __forceinline__ double hmm_tiny<double>() { return 1e-290; }
template <> __device__ __forceinline__ float hmm_tiny<float>() { return 1e-30f; }

// Every step needs, for each target state j, L_j = log sum_i exp(x_i + tr_ij) over the K source states.  Taken
// literally that is K exponentials per lane and step (K^2 per chain), which was ~90 % of this kernel in fp64.  Here a
// step factors the sum in probability space with A = exp(tr) kept in registers and ONE exponential per lane -- in an
// EXTENDED-RANGE form, so that it survives sharp posteriors (round 3; round 2's form scaled by one exp(M) per chain and
// fell back to the literal step whenever the sources that reach some column all sat > 300 decades below the chain's
// maximum, which after the first DMBD iteration was most steps): with M = max_i x_i,
//     exp(x_i - M) = f_i 2^{n_i},   n_i = floor((x_i - M) log2 e) (an integer, however negative),  f_i = exp((x_i - M) - n_i ln 2) in [1, 2)
// (ln 2 split in two parts so that the reduction is exact to rounding), exchanged through LDS as (f_i, n_i); column j takes
// its OWN reference exponent E_j = max {n_i : A_ij > 0} and sums s_j = sum_i ldexp(f_i, n_i - E_j) A_ij, whose largest term
// is >= A_ij -- no underflow that matters, terms 2^-1000 below the column's largest flush to zero exactly as exp() does in
// the literal form -- and L_j = M + E_j ln 2 + log s_j.  The literal log-space step remains for chains whose transition
// matrix itself underflows (0 < exp(tr_ij) < 1e-290 / 1e-30).
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
  constexpr int LDM = Kp + 1;  // odd row stride: lane j reads ROW j of the pair matrix, a stride of Kp words would put all lanes on one bank
  T* vec = smem + cl * (Kp + Kp * LDM + Kp);  // [Kp] message exchange, [Kp][LDM] pair weights, [Kp] exponents
  T* mat = vec + Kp;
  const T NI = neg_inf<T>();
  // column j of the transition matrix (A = exp(log transition); the log form is re-read from global memory by the rare
  // log-space steps) and entry j of the initial log probabilities
  auto TR = [&](int i) -> T { return (i < K && j < K) ? trans[(b * K + i) * K + j] : NI; };
  T A[Kp], acc[Kp];
#pragma unroll
  for (int i = 0; i < Kp; ++i) {
    A[i] = exp(TR(i));
    acc[i] = T(0);
  }
  // do all lanes of my chain agree?
  const unsigned long long grp = (Kp == 64) ? ~0ull : (((1ull << (Kp & 63)) - 1ull) << (cl * Kp));
  auto chain_all = [&](bool ok) -> bool { return (__ballot(ok) & grp) == grp; };
  // does exp(tr) represent every finite transition of my chain's matrix?  (wave-level vote per chain, once)
  bool a_ok = true;
#pragma unroll
  for (int i = 0; i < Kp; ++i) a_ok = a_ok && !(TR(i) > NI && !(A[i] >= hmm_tiny<T>()));
  const bool prob_space = chain_all(a_ok);
  int* vecn = reinterpret_cast<int*>(vec + Kp + Kp * LDM);  // [Kp] exponents n_i
  constexpr int NEG = -(1 << 29);
  // Column sums in extended-range probability space for the message x (entry j on lane j; lanes j >= K pass -inf).
  // Leaves (f_i, n_i) in vec / vecn and returns M, E_j, s_j and (w != nullptr) the terms w_i = f_i 2^{n_i - E_j} A_ij of s_j.
  auto col_sums = [&](T x, T& M, int& Ej, T& sj, T* w) {
    M = grp_max<Kp>(x);
    const bool fin = x > NI;  // (M > -inf whenever some x is)
    const T d = fin ? x - M : T(0);
    T nf = floor(d * T(1.4426950408889634073599246810019));
    nf = nf < T(NEG) ? T(NEG) : nf;
    // d - n ln2 with ln2 = hi + lo, hi holding 32 (fp64) / 12 (fp32) significant bits: n * hi is exact for |n| < 2^20 / 2^11;
    // beyond that the term is 2^-2048 below the maximum and its mantissa does not matter
    T r;
    if constexpr (sizeof(T) == 8) {
      r = __builtin_fma(-nf, T(6.93147180369123816490e-01), d);
      r = __builtin_fma(-nf, T(1.90821492927058770002e-10), r);
    } else {
      r = __builtin_fmaf(-nf, T(6.9314575195e-01f), d);
      r = __builtin_fmaf(-nf, T(1.4286067653e-06f), r);
    }
    const T fj = fin ? exp(r) : T(0);
    const int nj = fin ? (int)nf : NEG;
    wsync();  // the previous readers of vec are done
    vec[j] = fj;
    vecn[j] = nj;
    wsync();
    int nn[Kp], E = NEG;
#pragma unroll
    for (int i = 0; i < Kp; ++i) {
      nn[i] = vecn[i];
      E = (A[i] > T(0) && nn[i] > E) ? nn[i] : E;  // (padding sources carry NEG, forbidden ones A = 0)
    }
    sj = T(0);
#pragma unroll
    for (int i = 0; i < Kp; ++i) {
      // shift <= 0 for every source that reaches j; the others may sit above E, but their A is 0 (v_ldexp takes any int)
      const int sh = nn[i] - E;
      const T wi = ldexp(vec[i], sh > 0 ? 0 : sh) * A[i];
      if (w) w[i] = wi;  // (compile-time: the pair weights of the backward step reuse the terms)
      sj += wi;
    }
    Ej = E;
  };
  const T LN2 = T(0.693147180559945309417232121458);
  const T in_j = (j < K) ? init[b * K + j] : NI;
  const T* lg = logits + cc * K + (j < K ? j : 0);  // element (t, c, j) at lg[t*C*K]
  T* pj = p + cc * K + (j < K ? j : 0);
  const int64_t ts = C * K;

  // ---------------------------------------------------------------- forward (:77-80)
  T prev = in_j;
  // a wave has few neighbours on its SIMD to hide a memory round trip behind (a few hundred chains make a few dozen
  // waves): the observation logits of step t+1 are requested at the top of step t
  T lg_next = lg[0];
  for (int64_t t = 0; t < Tn; ++t) {
    const T lg_t = lg_next;
    lg_next = lg[(t + 1 < Tn ? t + 1 : t) * ts];
    if (prob_space) {
      T M, sj;
      int Ej;
      col_sums(prev, M, Ej, sj, nullptr);
      prev = ((sj > T(0)) ? (M + T(Ej) * LN2) + log(sj) : NI) + (j < K ? lg_t : T(0));
    } else {
      // literal log-space step
      wsync();
      vec[j] = prev;
      wsync();
      T m = NI;
      for (int i = 0; i < K; ++i) {
        const T v = vec[i] + TR(i);
        m = v > m ? v : m;
      }
      T s = T(0);
      for (int i = 0; i < K; ++i) s += (m > NI) ? exp(vec[i] + TR(i) - m) : T(0);
      prev = ((m > NI) ? m + log(s) : NI) + (j < K ? lg_t : T(0));
    }
    if (j >= K) prev = NI;
    if (live) pj[t * ts] = prev;  // the p buffer holds the filtered logits until the backward sweep
    wsync();
  }
  // logZ and normalisation (:81-83)
  const T lz = lse_lanes<Kp>(prev);
  T nxt = prev - lz;  // smoothed (= filtered) message at T-1

  // softmax with temperature of one message held one entry per lane (:100-101)
  auto emit = [&](int64_t t, T mine) {
    const T mx = grp_max<Kp>(mine);
    const T e = (j < K) ? exp((mine - mx) / ptemp) : T(0);
    const T den = grp_sum<Kp>(e);
    if (live) pj[t * ts] = e / den;
  };
  emit(Tn - 1, nxt);
  T nxt_max = grp_max<Kp>(nxt);  // max_j nxt_j, carried along the backward sweep

  // ---------------------------------------------------------------- backward smoothing (:85-98)
  // the filtered logits of step t-1 are requested at the top of step t (the step writes slot t of p only)
  T flt_next = (Tn >= 2) ? pj[(Tn - 2) * ts] : T(0);
  for (int64_t t = Tn - 2; t >= -1; --t) {
    // source message: filtered logits at t (normalised by logZ), or the initial distribution for the last step
    const T flt = flt_next;
    flt_next = pj[(t >= 1 ? t - 1 : 0) * ts];
    const T src = (t >= 0) ? ((live ? flt : NI) - lz) : in_j;
    const T xs = (j < K) ? src : NI;
    // column j of the pair logits: xi[i][j] = (src[i] + tr[i][j] - lse_i(src[i] + tr[i][j])) + nxt[j]; the step needs
    // e_ij = exp(xi_ij - Gs) for some common Gs: row sums of e give the new message, their total the normaliser,
    // e / total the pair posterior (the reference's three log-sum-exps over the same K x K logits, :88-95).
    T e[Kp], Gs;
    if (prob_space) {
      // probability space: exp(src_i + tr_ij - lse) = a_i A_ij / (2^{E_j} s_j), and Gs = max_j nxt_j keeps every e_ij <= 1
      T M, sj;
      int Ej;
      col_sums(xs, M, Ej, sj, e);
      Gs = nxt_max;
      const T u = (j < K && nxt > NI && sj > T(0)) ? exp(nxt - Gs) / sj : T(0);
#pragma unroll
      for (int i = 0; i < Kp; ++i) e[i] *= u;
    } else {
      // literal log-space step, ONE exponentiation per pair with Gs the largest pair logit
      wsync();
      vec[j] = xs;
      wsync();
      T m = NI;
      for (int i = 0; i < K; ++i) {
        const T v = vec[i] + TR(i);
        m = v > m ? v : m;
      }
      T s = T(0);
      for (int i = 0; i < K; ++i) s += (m > NI) ? exp(vec[i] + TR(i) - m) : T(0);
      const T cn = (m > NI) ? m + log(s) : NI;
      T xi[Kp], cm = NI;
#pragma unroll
      for (int i = 0; i < Kp; ++i) {
        const T v = (i < K) ? vec[i] + TR(i) : NI;
        xi[i] = (i < K && j < K && v > NI && cn > NI) ? (v - cn) + nxt : NI;
        cm = xi[i] > cm ? xi[i] : cm;
      }
      Gs = grp_max<Kp>((j < K) ? cm : NI);
#pragma unroll
      for (int i = 0; i < Kp; ++i) e[i] = (xi[i] > NI) ? exp(xi[i] - Gs) : T(0);
    }
#pragma unroll
    for (int i = 0; i < Kp; ++i) mat[i * LDM + j] = e[i];
    wsync();
    T rs = T(0);  // row j of e: unnormalised posterior of state j at the source time
#pragma unroll
    for (int jj = 0; jj < Kp; ++jj) rs += mat[j * LDM + jj];  // padding entries are exact zeros
    // exp(newmsg_j - max newmsg) = rs_j / max rs and sum_j of it = total / max rs: the softmax of the new message is
    // rs / total, and max_j newmsg_j (the next step's Gs) is Gs + log(max rs) -- no further exponentials or reductions
    const T rsv = (j < K) ? rs : T(0);
    const T total = grp_sum<Kp>(rsv);
    const T rsmax = grp_max<Kp>(rsv);
    const bool any = total > T(0);  // false only when every pair is forbidden
    const T inv_total = any ? T(1) / total : T(0);
#pragma unroll
    for (int i = 0; i < Kp; ++i) acc[i] += e[i] * inv_total;
    const T soft = any ? rsv * inv_total : nan_of<T>();  // softmax of the new message (0/0 in the reference, too)
    if (t >= 0) {
      nxt = (rsv > T(0)) ? Gs + log(rsv) : NI;
      nxt_max = any ? Gs + log(rsmax) : NI;
      if (ptemp == T(1)) {
        if (live) pj[t * ts] = soft;
      } else {  // tempered softmax (:100-101)
        const T et = (j < K) ? exp((nxt - nxt_max) / ptemp) : T(0);
        const T den = grp_sum<Kp>(et);
        if (live) pj[t * ts] = et / den;
      }
    } else if (live) {
      SEz0[cc * K + j] = soft;  // softmax of the initial-state message (:96-97)
    }
    wsync();
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
  const size_t smem = (size_t)CPW * (Kp + Kp * (Kp + 1) + Kp) * sizeof(T);  // message, pair weights, exponents (int, in T-sized slots)
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
