/* libvbmp_hip.so -- C ABI of the MI355X (gfx950) conjugate-update kernels.
 *
 * This is the drop-in boundary for the hot path of bayesianempirimancer/pyVBMP (reference paths are
 * relative to the reference repo root).  The reference has no FFI: its boundary is the duck-typed
 * Python method surface (ss_update / raw_update / Elog_like / forward / backward ...).  Each entry
 * point below replaces the tensor arithmetic inside one of those methods; the Python classes in
 * pyvbmp_amd/ keep the reference's names and signatures and call these through ctypes
 * (see INTEGRATION.md for the stub a reference maintainer would add).
 *
 * Conventions (all entry points)
 *   - every pointer is a DEVICE pointer; `stream` is a hipStream_t (NULL = default stream);
 *   - the call only enqueues work on `stream`: no allocation, no synchronisation, no global
 *     state, re-entrant; the caller owns every buffer and keeps it alive until the stream passes;
 *   - batch layout: B independent problems; a D x D matrix is row-major contiguous, a D-vector is
 *     contiguous; `s*` arguments are the distance in ELEMENTS between consecutive batch entries of
 *     that operand.  Stride 0 is allowed for read-only operands (one shared value, the
 *     reference's stride-0 expanded priors, dists/Wishart.py:17-18);
 *   - outputs are dense (stride D*D, D or 1);
 *   - 1 <= D <= VBMP_MAX_DIM (64); `_f64` = double, `_f32` = float;
 *   - `nonspd` (nullable) is a device int32 that is atomically incremented once per matrix whose
 *     elimination met a non-positive pivot.  Outputs for such a matrix follow the reference's
 *     silent behaviour: logdet is NaN when det < 0 (Tensor.logdet), the inverse is still the
 *     algebraic inverse;
 *   - return value: 0 = enqueued, VBMP_ERR_ARG (-1) = bad argument, VBMP_ERR_LAUNCH (-2) = HIP
 *     launch failure.
 */
#ifndef VBMP_HIP_H
#define VBMP_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VBMP_ABI_VERSION 8
int vbmp_abi_version(void);

/* K1 -- Ainv = A^-1 and logdet = log det A of B symmetric positive definite matrices.
 * Replaces every `X.inverse()` + `X.logdet()` pair on the path: dists/Wishart.py:20-24,55-56;
 * dists/MultivariateNormal.py:35-59; dists/MultivariateNormal_vector_format.py:79-107;
 * utils/matrix_utils.py:11-55; transforms/MatrixNormalWishart.py:46-48,134-135.
 * logdet may be NULL. */
int vbmp_spd_inv_logdet_f64(const double* A, int64_t sA, double* Ainv, double* logdet, int64_t B, int D,
                            int* nonspd, void* stream);
int vbmp_spd_inv_logdet_f32(const float* A, int64_t sA, float* Ainv, float* logdet, int64_t B, int D,
                            int* nonspd, void* stream);

/* K2a -- Wishart.ss_update (dists/Wishart.py:53-56), beta handled by the caller:
 *   invU = lr*(invU0 + SExx) + (1-lr)*invU_old ; nu = lr*(nu0 + N) + (1-lr)*nu_old ;
 *   U = invU^-1 ; logdet = log det invU.
 * invU_old / nu_old are read only when lr != 1 (may be NULL otherwise) and may alias nothing
 * that is written. */
int vbmp_wishart_ss_update_f64(const double* SExx, int64_t sSExx, const double* N, int64_t sN, const double* invU0,
                               int64_t sinvU0, const double* nu0, int64_t snu0, const double* invU_old,
                               int64_t sinvU_old, const double* nu_old, int64_t snu_old, double lr, double* invU,
                               double* nu, double* U, double* logdet, int64_t B, int D, int* nonspd, void* stream);
int vbmp_wishart_ss_update_f32(const float* SExx, int64_t sSExx, const float* N, int64_t sN, const float* invU0,
                               int64_t sinvU0, const float* nu0, int64_t snu0, const float* invU_old,
                               int64_t sinvU_old, const float* nu_old, int64_t snu_old, float lr, float* invU,
                               float* nu, float* U, float* logdet, int64_t B, int D, int* nonspd, void* stream);

/* K2 -- NormalInverseWishart.ss_update (dists/NormalInverseWishart.py:61-68 fused with
 * dists/Wishart.py:53-56); the headline operation (BASELINE.json config 2).  Per batch element:
 *   lam* = lam0 + N ; mu* = (lam0*mu0 + SEx)/lam* ;
 *   W    = SExx + lam0 mu0 mu0^T - lam* mu* mu*^T            (un-blended lam*, mu*: NIW.py:61-63)
 *   lam  = lr*lam* + (1-lr)*lam_old ; mu = lr*mu* + (1-lr)*mu_old
 *   invU = lr*(invU0 + W) + (1-lr)*invU_old ; nu = lr*(nu0 + N) + (1-lr)*nu_old
 *   U = invU^-1 ; logdet = log det invU          (skipped when fixed_precision != 0, NIW.py:67)
 * *_old are read only when lr != 1. */
int vbmp_niw_ss_update_f64(const double* SExx, int64_t sSExx, const double* SEx, int64_t sSEx, const double* N,
                           int64_t sN, const double* lam0, int64_t slam0, const double* mu0, int64_t smu0,
                           const double* invU0, int64_t sinvU0, const double* nu0, int64_t snu0,
                           const double* lam_old, int64_t slam_old, const double* mu_old, int64_t smu_old,
                           const double* invU_old, int64_t sinvU_old, const double* nu_old, int64_t snu_old,
                           double lr, double* lam, double* mu, double* invU, double* nu, double* U, double* logdet,
                           int64_t B, int D, int fixed_precision, int* nonspd, void* stream);
int vbmp_niw_ss_update_f32(const float* SExx, int64_t sSExx, const float* SEx, int64_t sSEx, const float* N,
                           int64_t sN, const float* lam0, int64_t slam0, const float* mu0, int64_t smu0,
                           const float* invU0, int64_t sinvU0, const float* nu0, int64_t snu0, const float* lam_old,
                           int64_t slam_old, const float* mu_old, int64_t smu_old, const float* invU_old,
                           int64_t sinvU_old, const float* nu_old, int64_t snu_old, float lr, float* lam, float* mu,
                           float* invU, float* nu, float* U, float* logdet, int64_t B, int D, int fixed_precision,
                           int* nonspd, void* stream);

/* K3a -- expected Gaussian log-likelihood as a quadratic form (the body of
 * NormalInverseWishart.Elog_like, dists/NormalInverseWishart.py:91-97, without the (N,K,D,D)
 * temporary):   out[s,bo,bi] = -1/2 x^T P x + x^T b + c,   x = X[s,bi,:],  (P,b,c) = component (bo,bi).
 * X is dense (S,Bi,D); P (Bo*Bi,D,D), b (Bo*Bi,D), c (Bo*Bi) dense; out dense (S,Bo,Bi).
 * Bi = 1 is the mixture case (every component sees the same sample). */
int vbmp_quadform_loglike_f64(const double* X, int64_t S, int64_t Bo, int64_t Bi, int D, const double* P,
                              const double* b, const double* c, double* out, void* stream);
int vbmp_quadform_loglike_f32(const float* X, int64_t S, int64_t Bo, int64_t Bi, int D, const float* P,
                              const float* b, const float* c, float* out, void* stream);

/* K3 -- fused mixture E-step (Mixture.update_assignments, dists/Mixture.py:38-45, with the NIW
 * likelihood inlined): l[s,k] = quadratic form above with c[k] already holding E log pi_k;
 * p[s,k] = softmax_k l[s,k];  NA[k] += sum_s p[s,k];  logZ[0] += sum_s logsumexp_k l[s,k].
 * X dense (S,D); p dense (S,K) (also used as scratch); NA (K) and logZ (1) MUST be zeroed by the caller
 * (they are accumulated with atomics, so their last bits depend on arrival order). */
int vbmp_mixture_estep_f64(const double* X, int64_t S, int K, int D, const double* P, const double* b,
                           const double* c, double* p, double* NA, double* logZ, void* stream);
int vbmp_mixture_estep_f32(const float* X, int64_t S, int K, int D, const float* P, const float* b, const float* c,
                           float* p, float* NA, float* logZ, void* stream);
/* K3, symmetric-packed form: the same E-step with the precision of component k passed as the packed upper triangle
 * Q[k] = (Q_00, Q_01, .., Q_0,D-1, Q_11, ..), Q_ii = P_ii, Q_ij = P_ij + P_ji (j > i), so that x' P x costs D (D + 1) / 2 + D
 * fused multiply-adds per (sample, component) with Q as wave-uniform scalar operands.  D in {4, 8, 16, 32} (no padding),
 * 1 <= K <= VBMP_ESTEP_SYM_MAX_K.  lse (S) or NULL: the per-sample evidence logsumexp_k l[s,k] (the mixtures of linear
 * transforms keep it, transforms/MixtureofLinearTransforms.py:36-43).  Other arguments as vbmp_mixture_estep. */
#define VBMP_ESTEP_SYM_MAX_K 8
int vbmp_mixture_estep_sym_f64(const double* X, int64_t S, int K, int D, const double* Q, const double* b,
                               const double* c, double* p, double* NA, double* logZ, double* lse, void* stream);
int vbmp_mixture_estep_sym_f32(const float* X, int64_t S, int K, int D, const float* Q, const float* b, const float* c,
                               float* p, float* NA, float* logZ, float* lse, void* stream);


/* K4 -- weighted sufficient statistics (NormalInverseWishart.raw_update, dists/NormalInverseWishart.py:72-84;
 * MultivariateNormal.raw_update, dists/MultivariateNormal.py:93-99) without the (N,K,D,D) temporary:
 *   Nk[bo,bi] += sum_s w;  SEx[bo,bi,:] += sum_s w x;  SExx[bo,bi,:,:] += sum_s w x x^T,
 *   x = X[s,bi,:] (dense (S,Bi,D)),  w = p[s,bo,bi] (dense (S,Bo,Bi); NULL = all ones).
 * Outputs MUST be zeroed by the caller (atomic accumulation).  Bi <= 65535. */
int vbmp_weighted_moments_f64(const double* X, const double* p, int64_t S, int64_t Bo, int64_t Bi, int D, double* Nk,
                              double* SEx, double* SExx, void* stream);
int vbmp_weighted_moments_f32(const float* X, const float* p, int64_t S, int64_t Bo, int64_t Bi, int D, float* Nk,
                              float* SEx, float* SExx, void* stream);

/* K9 -- linear-dynamical-system E-step: information filter + smoother over T time steps for S independent
 * series in ONE persistent launch (LinearDynamicalSystems.forward_backward_loop with forward_step /
 * backward_step / forward_backward_combiner, models/LinearDynamicalSystems.py:268-383, including the
 * reference's slot convention: Sigma_t_tp1[T-1] holds the (x0, x_0) cross term, and its elementwise `*` at :372).
 * Hidden dimension 1 <= H <= VBMP_LDS_MAX_H.  Series s belongs to batch element b = s % NB (sample axes lead).
 * Per-batch system parameters are dense (NB, ...).  Per-step inputs are addressed as
 *     base + t*st_t + (s / NB)*st_s + (s % NB)*st_b          (strides in ELEMENTS, 0 = shared),
 * so a likelihood precision that does not depend on time or sample is passed once (st_t = st_s = 0).
 * cu1 = QA_xp_u U[t], cu2 = ATQA_x_u U[t] (H-vectors) and cu3 = U[t]' ATQA_u_u U[t] carry the control input.
 * x0_res = -1/2 EXTinvUX + 1/2 ElogdetinvSigma - H/2 log 2pi of the initial-state prior (:349).
 * Outputs are dense: invSigma/Sigma/Sigma_t_tp1 (T,S,H,H), invSigmamu/mu (T,S,H), logZ (T,S),
 * Sigma_x0_x0 (S,H,H), mu_x0 (S,H).  All blocks must be 16-byte aligned.
 * sum_xx / sum_xpx (nullable, dense (S,H,H)) receive the time-integrated latent statistics of update_latents
 * (models/LinearDynamicalSystems.py:173-178) that the backward sweep already holds in registers:
 *     sum_xx[s]  = sum_{t<T}   Sigma[t,s] + mu[t,s] mu[t,s]'
 *     sum_xpx[s] = sum_{t<T-1} Sigma_t_tp1[t,s] + mu[t,s] mu[t+1,s]'
 * so that the two largest outputs are not read back from HBM for their time sums.
 * flags & VBMP_LDS_CROSS_WORK: the caller only wants slot T-1 of Sigma_t_tp1 (the x0 cross term, which is all that
 * update_latents reads next to sum_xpx, :178); the other slots are then a work buffer of the sweeps (still (T,S,H,H)) whose
 * final contents are unspecified, and the kernel may skip their stores.
 * flags & VBMP_LDS_LOGZ_SUM: logZ is (1,S) instead of (T,S) and receives sum_t logZ[t,s] (update_latents keeps nothing else
 * of it, :216), accumulated in time order.
 * Fixed-point shortcut (row-per-lane form, likelihood precision independent of time, lP_t == 0): the matrix half of the
 * recursion is then a Riccati iteration with constant coefficients; once it has stopped moving the kernel runs only the mean
 * recursion on the frozen matrices (the reference recomputes the same matrices at every step, :268-330).  Accuracy contract:
 *   default                          stop at a bitwise repeat of the filtered precision, OR after 8 consecutive steps that each moved
 *                                    it by <= 4 ulp of its row's largest entry, moved it by <= 4 ulp in total, and only while the
 *                                    recursion was seen to contract by >= 2x per 8 steps: outputs equal the literal recursion's to
 *                                    its own last-bit wander (<= 1e-13 fp64 / 2e-6 fp32 normwise per tensor in the tests)
 *   flags & VBMP_LDS_FIXED_POINT_EXACT   stop only at a bitwise repeat: every output is bit for bit that of the literal recursion
 *   flags & VBMP_LDS_FIXED_POINT_OFF     never stop: the literal recursion at every step
 * H <= VBMP_LDS_MAX_H: two register-resident device forms, chosen by S: one series per 16-lane DPP row (S <= 32768: 4 series per wave, so that few
 * thousand series already cover every SIMD) and one series per lane (more series). */
#define VBMP_LDS_CROSS_WORK 1
#define VBMP_LDS_LOGZ_SUM 2
#define VBMP_LDS_FIXED_POINT_EXACT 4
#define VBMP_LDS_FIXED_POINT_OFF 8
#define VBMP_LDS_MAX_H 8        /* register-resident forms */
#define VBMP_LDS_MAX_H_BLOCK 64 /* block-per-series form with LDS-resident matrices (8 < H; needs 5 H^2 words of LDS:
                                   fp64 up to H = 61); larger H return VBMP_ERR_ARG and the caller composes the recursion */
#define VBMP_DECL_LDS_ARGS(SUF, REAL)                                                                         \
  typedef struct vbmp_lds_args_##SUF {                                                                     \
    int64_t T, S, NB;                                                                                      \
    int H;                                                                                                 \
    int flags; /* VBMP_LDS_CROSS_WORK | VBMP_LDS_LOGZ_SUM | VBMP_LDS_FIXED_POINT_*: see above */              \
    const REAL *invQ, *ATQA_xx, *QA_xp_x, *A_Elogdet; /* (NB,H,H) x3, (NB) */                                 \
    const REAL *x0_P, *x0_eta, *x0_res;                /* (NB,H,H), (NB,H), (NB) */                           \
    const REAL* like_P;   int64_t lP_t, lP_s, lP_b;                                                           \
    const REAL* like_eta; int64_t le_t, le_s, le_b;                                                           \
    const REAL* like_res; int64_t lr_t, lr_s, lr_b;                                                           \
    const REAL* cu1;      int64_t c1_t, c1_s, c1_b;                                                           \
    const REAL* cu2;      int64_t c2_t, c2_s, c2_b;                                                           \
    const REAL* cu3;      int64_t c3_t, c3_s, c3_b;                                                           \
    REAL *invSigma, *invSigmamu, *Sigma, *mu, *Sigma_t_tp1, *logZ, *Sigma_x0_x0, *mu_x0;                      \
    REAL *sum_xx, *sum_xpx; /* nullable (S,H,H): time-integrated second moments, see above */                  \
    /* optional first-moment sums (only where vbmp_lds_smoother_caps_* reports VBMP_LDS_CAP_OBS_SUMS, NULL otherwise): */ \
    const REAL* y; int64_t y_t, y_s, y_b; /* observations (.., nobs), addressed like the per-step inputs; nullable */ \
    int nobs, reserved_;                  /* 1 <= nobs <= 16 */                                                \
    REAL *sum_mu, *sum_xy; /* nullable (S,H) = sum_t mu[t,s];  (S,H,nobs) = sum_t mu[t,s] y[t,s]' (needs y) */  \
  } vbmp_lds_args_##SUF;
VBMP_DECL_LDS_ARGS(f64, double)
VBMP_DECL_LDS_ARGS(f32, float)
int vbmp_lds_smoother_f64(const vbmp_lds_args_f64* args, void* stream);
int vbmp_lds_smoother_f32(const vbmp_lds_args_f32* args, void* stream);
/* Which optional outputs the launch of `args` would fill (it depends on the device form chosen from H and S; only T, S, NB, H
 * are read).  VBMP_LDS_CAP_OBS_SUMS: sum_mu / sum_xy -- the sums over time that update_latents (:179-190) otherwise forms with
 * one more pass over mu and y -- are accumulated by the backward sweep (row-per-lane form); elsewhere they must be NULL. */
#define VBMP_LDS_CAP_OBS_SUMS 1
int vbmp_lds_smoother_caps_f64(const vbmp_lds_args_f64* args);
int vbmp_lds_smoother_caps_f32(const vbmp_lds_args_f32* args);

/* K10 -- time-integrated cross moments of the LDS statistics (LinearDynamicalSystems.update_latents,
 * models/LinearDynamicalSystems.py:173-190):  out[s,i,j] = sum_{t<Tn} a[t,s,i]*b[t,s,j] (+ sum_t M[t,s,i,j]).
 * a: element (t,s,i) at a + t*sa_t + s*sa_s + i (strides in elements, 0 = constant along that axis); b alike
 * with db entries; M (nullable): (t,s,i,j) at M + t*sM_t + s*sM_s + i*db + j.  out dense (S,da,db).
 * Time-shifted operands (mu[:-1] against mu[1:]) are expressed by offsetting the base pointers.
 * With few outputs and many steps the time axis is cut into chunks that are combined with atomics (the last bits of
 * out then depend on arrival order). */
int vbmp_tsum_outer_f64(const double* a, int64_t sa_t, int64_t sa_s, int da, const double* b, int64_t sb_t, int64_t sb_s,
                        int db, const double* M, int64_t sM_t, int64_t sM_s, int64_t Tn, int64_t S, double* out,
                        void* stream);
int vbmp_tsum_outer_f32(const float* a, int64_t sa_t, int64_t sa_s, int da, const float* b, int64_t sb_t, int64_t sb_s,
                        int db, const float* M, int64_t sM_t, int64_t sM_s, int64_t Tn, int64_t S, float* out,
                        void* stream);

/* K7 / K8 -- fused MatrixNormalWishart messages: forward (transforms/MatrixNormalWishart.py:303-328) and
 * backward (:352-375 with utils/matrix_utils.py:31-46) are the same "sandwich" computation.  Per message
 * (sample s < S, expert b < NB), with P a d x d SPD matrix:
 *     q1 = e1' P^-1 e1                ld1 = logdet P
 *     A  = P + Add1_b ;  Sm = A^-1 ;  v = Sm e2 ;   q2 = e2' v ;   ld2 = logdet A
 *     q3 = e3' (P + Add2_b)^-1 e3     ld3 = logdet(P + Add2_b)            (only when Add2 != NULL)
 *     ovec = M_b v   (m) ;            omat = C_b + sign * M_b Sm M_b^T   (m x m)
 *     q4 = w' omat^-1 w, ld4 = logdet omat,  w = ovec + cvec_b            (only when cvec != NULL)
 *   Schur mode (e3 != NULL and cvec != NULL with Add2 == NULL): q3 and ld3 refer to the matrix
 *     P + Add1_b + sign * M_b' C_b^-1 M_b  (the other Schur complement of the joint matrix [[A, .], [., C]]) and are
 *     obtained WITHOUT eliminating it:  q3 = e3' Sm e3 + t' omat^-1 t  with t = M_b Sm e3,  and scal[5] = ld2 + ld4
 *     = ld3 + logdet C_b -- the caller subtracts the per-expert constant logdet C_b.
 *   forward : P = P_x, e1 = eta_x, e2 = shifted eta, Add1 = n V, M = E[A], C = invEinvSigma, sign = +1 (d = p, m = n)
 *             Res = -q1/2 + q2/2 - (ld2 - ld1)/2
 *   backward: P = P_y, e1 = eta_y, e2 = j_y, Add1 = E[R], e3 = eta_y + G H^-1 j_x, M = G' = E[RA]', C = H = E[A'RA],
 *             sign = -1, cvec = j_x  (d = n, m = p):  omat = invSigma_x, ovec + cvec = invSigmamu_x; the q3 / ld3 terms of
 *             the marginal precision E[R] - G H^-1 G' + P_y either in Schur mode (Add2 = NULL, one elimination fewer) or
 *             with that matrix's constant part passed as Add2.
 * P, e1, e2, e3: element (s,b) at base + s*st_s + b*st_b (strides in elements, 0 = shared).  Add1, Add2 (NB,d,d),
 * M (NB,m,d), C (NB,m,m), cvec (NB,m) dense.  Outputs dense: ovec (S,NB,m), omat (S,NB,m,m),
 * scal (S,NB,8) = [q1, ld1, q2, ld2, q3, ld3, q4, ld4].
 * Needs padded(m) <= padded(d) <= VBMP_MNW_MAX_DIM (padding to 1,2,4,8,16,32), NB <= 65535. */
#define VBMP_MNW_MAX_DIM 32
int vbmp_mnw_message_f64(const double* P, int64_t sP_s, int64_t sP_b, const double* e1, int64_t s1_s, int64_t s1_b,
                         const double* e2, int64_t s2_s, int64_t s2_b, const double* e3, int64_t s3_s, int64_t s3_b,
                         const double* Add1, const double* Add2, const double* M, const double* C, const double* cvec,
                         double sign, double* ovec, double* omat, double* scal, int64_t S, int64_t NB, int m, int d,
                         void* stream);
int vbmp_mnw_message_f32(const float* P, int64_t sP_s, int64_t sP_b, const float* e1, int64_t s1_s, int64_t s1_b,
                         const float* e2, int64_t s2_s, int64_t s2_b, const float* e3, int64_t s3_s, int64_t s3_b,
                         const float* Add1, const float* Add2, const float* M, const float* C, const float* cvec,
                         float sign, float* ovec, float* omat, float* scal, int64_t S, int64_t NB, int m, int d,
                         void* stream);
/* The same launch with the caller's residual formed in the kernel: res[s,b] = res_c[b] + sum_k res_w[k] * scal[s,b,k]
 * (forward: Res = -q1/2 + q2/2 - (ld2 - ld1)/2, backward: transforms/MatrixNormalWishart.py:366-375 -- otherwise ~10 element-wise
 * passes over strided slices of scal), and, with add_cvec != 0, ovec + cvec_b stored instead of ovec (the backward message's
 * invSigmamu_x).  res_w: 8 weights in HOST memory; res_c: (NB) device memory or NULL; res: dense (S,NB) or NULL. */
int vbmp_mnw_message_res_f64(const double* P, int64_t sP_s, int64_t sP_b, const double* e1, int64_t s1_s, int64_t s1_b,
                             const double* e2, int64_t s2_s, int64_t s2_b, const double* e3, int64_t s3_s, int64_t s3_b,
                             const double* Add1, const double* Add2, const double* M, const double* C, const double* cvec,
                             double sign, double* ovec, double* omat, double* scal, int64_t S, int64_t NB, int m, int d,
                             const double* res_w, const double* res_c, double* res, int add_cvec, void* stream);
int vbmp_mnw_message_res_f32(const float* P, int64_t sP_s, int64_t sP_b, const float* e1, int64_t s1_s, int64_t s1_b,
                             const float* e2, int64_t s2_s, int64_t s2_b, const float* e3, int64_t s3_s, int64_t s3_b,
                             const float* Add1, const float* Add2, const float* M, const float* C, const float* cvec,
                             float sign, float* ovec, float* omat, float* scal, int64_t S, int64_t NB, int m, int d,
                             const float* res_w, const float* res_c, float* res, int add_cvec, void* stream);


/* K11 -- discrete HMM forward-backward in log space (HMM.forward_backward_logits, models/HMM.py:72-105), the role
 * chain of DynamicMarkovBlanketDiscovery.  C independent chains of length Tn over K states; chain c uses the
 * parameters of batch element c % NB.  logits (Tn,C,K): observation log-likelihoods; trans (NB,K,K): E log
 * transition (row = from, column = to; -inf = forbidden); init (NB,K): E log initial.  Outputs dense:
 * p (Tn,C,K) = softmax of the smoothed messages with temperature ptemp (:100-101), SEzz (C,K,K) = sum over time of
 * the pair posteriors incl. the initial step, SEz0 (C,K), logZ (C).  1 <= K <= VBMP_HMM_MAX_K. */
#define VBMP_HMM_MAX_K 64
int vbmp_hmm_forward_backward_f64(const double* logits, const double* trans, const double* init, int64_t Tn, int64_t C,
                                  int64_t NB, int K, double ptemp, double* p, double* SEzz, double* SEz0, double* logZ,
                                  void* stream);
int vbmp_hmm_forward_backward_f32(const float* logits, const float* trans, const float* init, int64_t Tn, int64_t C,
                                  int64_t NB, int K, float ptemp, float* p, float* SEzz, float* SEz0, float* logZ,
                                  void* stream);

/* K5b -- streaming weighted sum of per-sample matrices: out[e] += sum_{s<S} w[s] * C[s,e], e < E (= d*d): the
 * covariance part of MatrixNormalWishart.update (sum_s p_s Sigma_s, transforms/MatrixNormalWishart.py:153-155).
 * C dense (S,E); w (S) or NULL (unit weights); out (E) MUST be zeroed by the caller (atomic accumulation). */
int vbmp_weighted_matsum_f64(const double* C, const double* w, int64_t S, int64_t E, double* out, void* stream);
int vbmp_weighted_matsum_f32(const float* C, const float* w, int64_t S, int64_t E, float* out, void* stream);

/* K5b with several weight columns: out[b,e] += sum_{s<S} W[s,b] * C[s,e], b < NB <= VBMP_MATSUM_MAX_COLS: the same
 * covariance term when all NB experts / roles weigh the SAME per-sample matrices (a DMBD observation's latent message).
 * C dense (S,E); W dense (S,NB); out (NB,E) MUST be zeroed by the caller (atomic accumulation). */
#define VBMP_MATSUM_MAX_COLS 32
int vbmp_weighted_matsum_cols_f64(const double* C, const double* W, int64_t S, int64_t E, int NB, double* out,
                                  void* stream);
int vbmp_weighted_matsum_cols_f32(const float* C, const float* W, int64_t S, int64_t E, int NB, float* out,
                                  void* stream);

/* K12 -- one shared small matrix applied to many rows: out[s,:] = M x[s,:] + c, s < S (the likelihood message
 * eta = E[A' R] y of every observation, MatrixNormalWishart.Elog_like_X, transforms/MatrixNormalWishart.py:251-261, on
 * the (time x series) rows of the LDS E-step).  X dense (S,k); M dense (n,k); c (n) or NULL; out dense (S,n).
 * 1 <= k, n <= VBMP_ROWS_MAX_DIM. */
#define VBMP_ROWS_MAX_DIM 64
int vbmp_rows_affine_f64(const double* X, int64_t S, int k, const double* M, const double* c, int n, double* out,
                         void* stream);
int vbmp_rows_affine_f32(const float* X, int64_t S, int k, const float* M, const float* c, int n, float* out,
                         void* stream);
/* K13 -- the parameter expectations a mixture E-step reads from a Normal-inverse-Wishart posterior, in one launch (replaces the
 * getters NormalInverseWishart.EinvSigma / EinvSigmamu / EXTinvUX / ElogdetinvSigma, dists/NormalInverseWishart.py:107-132 as used
 * by Elog_like :91-97, Wishart.ElogdetinvSigma dists/Wishart.py:82-83 and Dirichlet.loggeomean dists/Dirichlet.py:52-53):
 *     P[k] = U[k] nu[k],  b[k] = P[k] mu[k],
 *     c[k] = -1/2 (mu' P mu + D / lambda) + 1/2 (D log 2 - logdet_invU + sum_{i<D} psi((nu - i) / 2)) - D/2 log 2 pi
 *            + (alpha ? psi(alpha[k]) - psi(sum_j alpha[j]) : 0)
 * so that  -1/2 x' P x + x' b + c  is E log N(x | component k) (+ E log pi_k): the operands of vbmp_mixture_estep.  All dense:
 * U (K,D,D), nu / lam / logdet_invU / alpha (K), mu (K,D); outputs P (K,D,D), b (K,D), c (K).  alpha nullable.  D >= 1. */
int vbmp_niw_estep_params_f64(const double* U, const double* nu, const double* mu, const double* lam, const double* logdet_invU,
                              const double* alpha, int64_t K, int D, double* P, double* b, double* c, void* stream);
int vbmp_niw_estep_params_f32(const float* U, const float* nu, const float* mu, const float* lam, const float* logdet_invU,
                              const float* alpha, int64_t K, int D, float* P, float* b, float* c, void* stream);
/* K14 -- the expectations of a MatrixNormalWishart posterior that its messages and likelihoods read, in one launch (the getters
 * MatrixNormalWishart.EinvSigma / EinvUX / EXTinvU / EXTinvUX / ElogdetinvSigma, transforms/MatrixNormalWishart.py:419-471 with
 * dists/Wishart.py:67-83), for NB batch elements of an (n x p) transform:
 *     R = U nu (n,n);  G = R mu (n,p) [EXTinvU = G'];  H = n V + mu' G (p,p);  El = sum_{i<n} psi((nu - i)/2) + n log 2 - logdet_invU.
 * All dense: mu (NB,n,p), U (NB,n,n), nu / logdet_invU (NB), V (NB,p,p); outputs R (NB,n,n), G (NB,n,p), H (NB,p,p), El (NB). */
int vbmp_mnw_expectations_f64(const double* mu, const double* U, const double* nu, const double* V, const double* logdet_invU,
                              int64_t NB, int n, int p, double* R, double* G, double* H, double* El, void* stream);
int vbmp_mnw_expectations_f32(const float* mu, const float* U, const float* nu, const float* V, const float* logdet_invU, int64_t NB,
                              int n, int p, float* R, float* G, float* H, float* El, void* stream);
/* K15 -- KL(q || prior) of the conjugate families, one launch each (12-40 launches each when composed from element-wise kernels,
 * lgamma / digamma and reductions; every model's ELBO sums them once per VB iteration).  One result per batch element, NB of them;
 * posterior operands dense; every PRIOR operand comes with its batch stride in elements (0 = one prior shared by the batch, the
 * reference's expanded priors).
 *   dirichlet_kl : Dirichlet.KLqprior (dists/Dirichlet.py:73-86), K = entries per event; structural zeros count as 0 like the
 *                  reference's KL_lgamma / KL_digamma.
 *   gamma_kl     : Gamma.KLqprior (dists/Gamma.py:66-72) summed over the K entries of an event (DiagonalWishart.KLqprior).
 *   wishart_kl   : Wishart.KLqprior (dists/Wishart.py:85-95) for n x n; with mu != NULL also the Normal part of
 *                  NormalInverseWishart.KLqprior (dists/NormalInverseWishart.py:134-141): mu / mu0 (NB, n), lam / lam0 (NB).
 *   mn_kl        : the matrix-normal part of MatrixNormalWishart / MatrixNormalGamma.KLqprior (transforms/MatrixNormalWishart.py:
 *                  206-215, MatrixNormalGamma.py:203-214) for an (n x p) transform: R = E[invSigma] (NB, n, n), xm = number of set
 *                  entries of X_mask per batch element (a prior-like operand with its stride; NULL without a mask); n p + p max(n, p) <= 8192. */
int vbmp_dirichlet_kl_f64(const double* alpha, const double* alpha0, int64_t s0, int64_t NB, int K, double* out, void* stream);
int vbmp_dirichlet_kl_f32(const float* alpha, const float* alpha0, int64_t s0, int64_t NB, int K, float* out, void* stream);
int vbmp_gamma_kl_f64(const double* alpha, const double* beta, const double* alpha0, const double* beta0, int64_t sa0, int64_t sb0,
                      int64_t NB, int K, double* out, void* stream);
int vbmp_gamma_kl_f32(const float* alpha, const float* beta, const float* alpha0, const float* beta0, int64_t sa0, int64_t sb0,
                      int64_t NB, int K, float* out, void* stream);
int vbmp_wishart_kl_f64(const double* invU0, int64_t sm0, const double* U, const double* nu, const double* nu0, int64_t sn0,
                        const double* ld, const double* ld0, int64_t sl0, const double* mu, const double* mu0, int64_t smu0,
                        const double* lam, const double* lam0, int64_t slam0, int64_t NB, int n, double* out, void* stream);
int vbmp_wishart_kl_f32(const float* invU0, int64_t sm0, const float* U, const float* nu, const float* nu0, int64_t sn0,
                        const float* ld, const float* ld0, int64_t sl0, const float* mu, const float* mu0, int64_t smu0,
                        const float* lam, const float* lam0, int64_t slam0, int64_t NB, int n, float* out, void* stream);
int vbmp_mn_kl_f64(const double* mu, const double* mu0, int64_t smu0, const double* invV0, int64_t sv0, const double* V,
                   const double* R, const double* ldV, const double* ldV0, int64_t sl0, const double* xm, int64_t sxm, int64_t NB,
                   int n, int p, double* out, void* stream);
int vbmp_mn_kl_f32(const float* mu, const float* mu0, int64_t smu0, const float* invV0, int64_t sv0, const float* V, const float* R,
                   const float* ldV, const float* ldV0, int64_t sl0, const float* xm, int64_t sxm, int64_t NB, int n, int p, float* out,
                   void* stream);
/* K12 with the observation likelihood's scalar in the same pass: additionally q[s] = -1/2 x' P x + b' x + c0[0]
 * (LinearDynamicalSystems.log_likelihood_function, models/LinearDynamicalSystems.py:244-266: invSigmamu_t and Residual of
 * every (time, series) from ONE read of the observations).  P dense (k,k); b (k) or NULL; c0 one element in device memory or
 * NULL; q dense (S).  1 <= k <= VBMP_ROWS_QUAD_MAX_K. */
#define VBMP_ROWS_QUAD_MAX_K 16
int vbmp_rows_affine_quad_f64(const double* X, int64_t S, int k, const double* M, const double* c, int n, double* out,
                              const double* P, const double* b, const double* c0, double* q, void* stream);
int vbmp_rows_affine_quad_f32(const float* X, int64_t S, int k, const float* M, const float* c, int n, float* out,
                              const float* P, const float* b, const float* c0, float* q, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VBMP_HIP_H */
