"""ctypes binding of libvbmp_hip.so (the C-ABI declared in include/vbmp_hip.h).

There is deliberately no fallback: if the shared library is missing or a tensor does not live on a
HIP device, the call raises.  Build the library with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C pyvbmp_amd/csrc`.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvbmp_hip.so")

_c_i64 = ctypes.c_int64
_c_int = ctypes.c_int
_c_ptr = ctypes.c_void_p
_c_double = ctypes.c_double

ABI_VERSION = 8

_lib = None


class VbmpHipError(RuntimeError):
    pass


def _sig_spd(T):
    return [_c_ptr, _c_i64, _c_ptr, _c_ptr, _c_i64, _c_int, _c_ptr, _c_ptr]


def _sig_wishart(T):
    return [_c_ptr, _c_i64] * 6 + [T] + [_c_ptr] * 4 + [_c_i64, _c_int, _c_ptr, _c_ptr]


def _sig_niw(T):
    return [_c_ptr, _c_i64] * 11 + [T] + [_c_ptr] * 6 + [_c_i64, _c_int, _c_int, _c_ptr, _c_ptr]


def _sig_quadform(T):
    # X, S, Bo, Bi, D, P, b, c, out, stream
    return [_c_ptr, _c_i64, _c_i64, _c_i64, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr]


def _sig_estep(T):
    # X, S, K, D, P, b, c, p, NA, logZ, stream
    return [_c_ptr, _c_i64, _c_int, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr]


def _sig_wmom(T):
    # X, p, S, Bo, Bi, D, Nk, SEx, SExx, stream
    return [_c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_ptr]


def lds_args_struct(cT):
    """ctypes mirror of vbmp_lds_args_{f64,f32} (include/vbmp_hip.h)."""
    P = _c_ptr
    fields = [("T", _c_i64), ("S", _c_i64), ("NB", _c_i64), ("H", _c_int), ("flags", _c_int)]
    fields += [(n, P) for n in ("invQ", "ATQA_xx", "QA_xp_x", "A_Elogdet", "x0_P", "x0_eta", "x0_res")]
    for n, pre in (("like_P", "lP"), ("like_eta", "le"), ("like_res", "lr"), ("cu1", "c1"), ("cu2", "c2"), ("cu3", "c3")):
        fields += [(n, P), (pre + "_t", _c_i64), (pre + "_s", _c_i64), (pre + "_b", _c_i64)]
    fields += [(n, P) for n in ("invSigma", "invSigmamu", "Sigma", "mu", "Sigma_t_tp1", "logZ", "Sigma_x0_x0", "mu_x0",
                                "sum_xx", "sum_xpx")]
    fields += [("y", P), ("y_t", _c_i64), ("y_s", _c_i64), ("y_b", _c_i64), ("nobs", _c_int), ("reserved_", _c_int),
               ("sum_mu", P), ("sum_xy", P)]
    return type("vbmp_lds_args", (ctypes.Structure,), {"_fields_": fields})


LDS_ARGS = {"f64": lds_args_struct(ctypes.c_double), "f32": lds_args_struct(ctypes.c_float)}
LDS_CROSS_WORK = 1     # vbmp_lds_args.flags: only slot T-1 of Sigma_t_tp1 is wanted
LDS_LOGZ_SUM = 2       # logZ is (1, S) and receives its sum over time
# fixed-point shortcut of the smoother (VBMP_LDS_FIXED_POINT_EXACT / _OFF; accuracy contract in include/vbmp_hip.h)
LDS_FIXED_POINT = {"auto": 0, "exact": 4, "off": 8}
LDS_CAP_OBS_SUMS = 1   # vbmp_lds_smoother_caps_*: sum_mu / sum_xy are filled by the launch
LDS_MAX_H = 8          # register-resident smoother forms
LDS_MAX_H_BLOCK = 64   # block-per-series form (LDS-resident matrices); also bounded by lds_block_fits()


def lds_block_fits(h, itemsize):
    """does the block-per-series smoother's LDS image (five h x h matrices + vectors) fit the CU's 160 KB?"""
    return (5 * h * (h | 1) + 24 * 72 + 8) * itemsize <= 160 * 1024


def _sig_lds(T):
    return [_c_ptr, _c_ptr]  # const vbmp_lds_args_*: passed byref, stream


def _sig_tsum(T):
    # a, sa_t, sa_s, da, b, sb_t, sb_s, db, M, sM_t, sM_s, Tn, S, out, stream
    return [_c_ptr, _c_i64, _c_i64, _c_int, _c_ptr, _c_i64, _c_i64, _c_int, _c_ptr, _c_i64, _c_i64, _c_i64, _c_i64,
            _c_ptr, _c_ptr]


def _sig_mnw_msg(T):
    # (P, e1, e2, e3 each with 2 strides), Add1, Add2, M, C, cvec, sign, ovec, omat, scal, S, NB, m, d, stream
    return [_c_ptr, _c_i64, _c_i64] * 4 + [_c_ptr] * 5 + [T] + [_c_ptr] * 3 + [_c_i64, _c_i64, _c_int, _c_int, _c_ptr]


MNW_MAX_DIM = 32
MAX_DIM = 64  # VBMP_MAX_DIM: K1 / K2 / K3a matrix size per wave


def _sig_hmm(T):
    # logits, trans, init, Tn, C, NB, K, ptemp, p, SEzz, SEz0, logZ, stream
    return [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_int, T, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr]


HMM_MAX_K = 64


def _sig_matsum(T):
    # C, w, S, E, out, stream
    return [_c_ptr, _c_ptr, _c_i64, _c_i64, _c_ptr, _c_ptr]


def _sig_matsum_cols(T):
    # C, W, S, E, NB, out, stream
    return [_c_ptr, _c_ptr, _c_i64, _c_i64, _c_int, _c_ptr, _c_ptr]


def _sig_rows(T):
    # X, S, k, M, c, n, out, stream
    return [_c_ptr, _c_i64, _c_int, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr]


# symbol -> argtypes builder.  Every symbol declared in include/vbmp_hip.h appears here
# (tests/test_cabi.py cross-checks the header against this table and against the .so).
SYMBOLS = {
    "vbmp_spd_inv_logdet": _sig_spd,
    "vbmp_wishart_ss_update": _sig_wishart,
    "vbmp_niw_ss_update": _sig_niw,
    "vbmp_quadform_loglike": _sig_quadform,
    "vbmp_mixture_estep": _sig_estep,
    "vbmp_mixture_estep_sym": lambda T: _sig_estep(T)[:-1] + [_c_ptr, _c_ptr],  # ... logZ, lse, stream
    "vbmp_weighted_moments": _sig_wmom,
    "vbmp_lds_smoother": _sig_lds,
    "vbmp_lds_smoother_caps": lambda T: [_c_ptr],
    "vbmp_tsum_outer": _sig_tsum,
    "vbmp_mnw_message": _sig_mnw_msg,
    # ... + res_w (host, 8 values), res_c, res, add_cvec before the stream
    "vbmp_mnw_message_res": lambda T: _sig_mnw_msg(T)[:-1] + [_c_ptr, _c_ptr, _c_ptr, _c_int, _c_ptr],
    "vbmp_hmm_forward_backward": _sig_hmm,
    "vbmp_weighted_matsum": _sig_matsum,
    "vbmp_weighted_matsum_cols": _sig_matsum_cols,
    "vbmp_rows_affine": _sig_rows,
    # U, nu, mu, lam, logdet_invU, alpha, K, D, P, b, c, stream
    "vbmp_niw_estep_params": lambda T: [_c_ptr] * 6 + [_c_i64, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_ptr],
    # mu, U, nu, V, logdet_invU, NB, n, p, R, G, H, El, stream
    "vbmp_mnw_expectations": lambda T: [_c_ptr] * 5 + [_c_i64, _c_int, _c_int] + [_c_ptr] * 5,
    # K15: alpha, alpha0, s0, NB, K, out, stream
    "vbmp_dirichlet_kl": lambda T: [_c_ptr, _c_ptr, _c_i64, _c_i64, _c_int, _c_ptr],
    # alpha, beta, alpha0, beta0, sa0, sb0, NB, K, out, stream
    "vbmp_gamma_kl": lambda T: [_c_ptr] * 4 + [_c_i64, _c_i64, _c_i64, _c_int, _c_ptr],
    # invU0, sm0, U, nu, nu0, sn0, ld, ld0, sl0, mu, mu0, smu0, lam, lam0, slam0, NB, n, out, stream
    "vbmp_wishart_kl": lambda T: [_c_ptr, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_i64,
                                  _c_ptr, _c_ptr, _c_i64, _c_i64, _c_int, _c_ptr],
    # mu, mu0, smu0, invV0, sv0, V, R, ldV, ldV0, sl0, xm, sxm, NB, n, p, out, stream
    "vbmp_mn_kl": lambda T: [_c_ptr, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_i64,
                             _c_int, _c_int, _c_ptr],
    # X, S, k, M, c, n, out, P, b, c0, q, stream
    "vbmp_rows_affine_quad": lambda T: [_c_ptr, _c_i64, _c_int, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr],
}
DTYPES = {"f64": (torch.float64, ctypes.c_double), "f32": (torch.float32, ctypes.c_float)}


def load():
    """Load (once) and return the ctypes library; raise loudly when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VbmpHipError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `make -C {os.path.join(_HERE, 'csrc')}` "
            "(hipcc --offload-arch=gfx950). pyvbmp_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    lib.vbmp_abi_version.restype = _c_int
    lib.vbmp_abi_version.argtypes = []
    v = lib.vbmp_abi_version()
    if v != ABI_VERSION:
        raise VbmpHipError(f"libvbmp_hip.so ABI {v} != expected {ABI_VERSION}; rebuild it")
    for base, sig in SYMBOLS.items():
        for suf, (_, cT) in DTYPES.items():
            fn = getattr(lib, f"{base}_{suf}")
            fn.restype = _c_int
            fn.argtypes = sig(cT)
    _lib = lib
    return lib


def suffix(dtype):
    if dtype == torch.float64:
        return "f64"
    if dtype == torch.float32:
        return "f32"
    raise VbmpHipError(f"unsupported dtype {dtype}: the HIP path computes in float32 or float64")


def require_device(*tensors):
    """All tensors must be on the same HIP device (torch calls it 'cuda')."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise VbmpHipError("pyvbmp_amd needs tensors on a HIP device (got a CPU tensor); "
                               "there is no CPU path - use torch.set_default_device('cuda') or device='cuda'")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise VbmpHipError(f"tensors on different devices: {dev} vs {t.device}")
    return dev


def stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


# Optional (before, after) callables run immediately around every kernel enqueue.  bench.py uses them to
# record HIP events right at the launch, so that the event pair brackets the kernel and not the
# Python-side marshalling in front of it.
launch_hooks = None


def call(fn, name, *args):
    """Enqueue one C-ABI kernel call (with the optional timing hooks) and raise on a non-zero code."""
    hooks = launch_hooks
    if hooks is not None:
        hooks[0](name)
    rc = fn(*args)
    if hooks is not None:
        hooks[1](name)
    check(rc, name)


def check(rc, name):
    if rc != 0:
        why = {-1: "bad argument: a null pointer, a negative size, or a size beyond the kernel's limit -- matrices D <= 64 "
                   "(K1 / K2 / K3a), message dims <= 32 (K7 / K8), <= 64 states (K11), <= 64 row entries (K12), <= 65535 "
                   "experts / components per launch; see INTEGRATION.md, section Limits",
               -2: "HIP launch failure"}.get(rc, "unknown")
        raise VbmpHipError(f"{name} failed with code {rc} ({why})")
