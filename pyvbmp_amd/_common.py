"""Small host-side helpers shared by the node classes."""
import torch


# Derived quantities cached on a parameter object (MatrixNormalWishart's expectations, K14) are keyed by the identity and version
# of the tensors they were computed from -- and by this epoch, which graph.GraphedStep advances whenever a replayed graph has
# rewritten state tensors in place behind Python's back (a replay bumps no tensor version).
state_epoch = [0]


def derived_key(*tensors):
    cap = tensors[0].is_cuda and torch.cuda.is_current_stream_capturing()
    return (state_epoch[0], cap) + tuple((id(t), t._version) for t in tensors)


def blend(new, old, lr):
    """lr * new + (1 - lr) * old, the natural-parameter step of every ss_update; at lr == 1.0 (the default, and what every VB loop
    of the reference's examples runs) it is `new` itself -- the same values (1.0 x + 0.0 old) without three launches per
    parameter.  (The one difference: a non-finite OLD value no longer poisons the new one through 0.0 * inf.)"""
    if isinstance(lr, (int, float)) and lr == 1.0:
        return new
    return lr * new + (1.0 - lr) * old


def resolve(device=None, dtype=None):
    """Device / dtype for freshly created state.  The reference has no device plumbing (it relies on
    torch defaults), so the defaults follow torch.set_default_device / set_default_dtype."""
    if device is None:
        device = torch.get_default_device() if hasattr(torch, "get_default_device") else torch.device("cpu")
    if dtype is None:
        dtype = torch.get_default_dtype()
    return torch.device(device), dtype


def as_param(v, device, dtype):
    """scalar / tensor prior parameter -> tensor on (device, dtype)"""
    if isinstance(v, torch.Tensor):
        return v.to(device=device, dtype=dtype)
    return torch.as_tensor(v, dtype=dtype, device=device)  # python scalars: no detour through float32


def trailing(v, k):
    """append k singleton axes"""
    return v.reshape(tuple(v.shape) + (1,) * k)


def collapse_to(t, shape):
    """Undo a broadcast: keep index 0 along every axis where `shape` has size 1."""
    shape = tuple(shape)
    assert t.ndim == len(shape), (t.shape, shape)
    for d, s in enumerate(shape):
        if s == 1 and t.shape[d] != 1:
            t = t.narrow(d, 0, 1)
    return t


def _k12_pays(rows, k, n):
    """K12 against the library GEMM, measured on MI355X (tools/exp/rows_time.py): 5-6x faster for k, n <= 9 at 1e5..4e6
    rows in both precisions (3.9-4.7 TB/s against 0.6-0.8), even at 32 x 32 in fp64, 2.3x slower at 32 x 32 in fp32"""
    return rows >= 16384 and k * n <= 256


def shared_matvec(G, Y, bias=None):
    """G @ Y for a stack of SHARED small matrices G (batch + (n, k), no sample axes) and per-sample vectors
    Y (sample + (1,)*len(batch) + (k, 1)): one (samples, k) x (k, batch*n) library GEMM instead of the
    samples*batch tiny matrix-vector products a broadcasting `@` is lowered to (600 000 of them, 2 ms, for the role
    emissions of the flocking DMBD).  Anything that does not have this shape goes to `@` unchanged."""
    if bias is not None:
        # bias: (n,) added to every product -- fused into K12 where that kernel runs, a broadcast add elsewhere
        if G.dim() == 2 and Y.dim() > 2 and Y.shape[-1] == 1 and Y.is_cuda and G.shape[-1] > 0 and Y.numel() > 0 \
                and _k12_pays(Y.numel() // G.shape[-1], G.shape[-1], G.shape[-2]):
            from . import ops
            return ops.rows_affine(Y.reshape(-1, G.shape[-1]), G, bias).reshape(tuple(Y.shape[:-2]) + (G.shape[-2], 1))
        return shared_matvec(G, Y) + bias.unsqueeze(-1)
    nb = G.dim() - 2
    if G.shape[-1] == 0 or Y.numel() == 0:
        return G @ Y  # empty contraction (e.g. no regressors: regression_dim = -1 in DynamicMarkovBlanketDiscovery)
    if nb == 0 and Y.dim() > 2 and Y.shape[-1] == 1 and Y.is_cuda and _k12_pays(Y.numel() // G.shape[-1], G.shape[-1], G.shape[-2]):
        # ONE matrix, very many vectors (the observation messages of an LDS E-step: 4e6 rows of 6): the library GEMM for
        # such a shape runs at a tenth of the memory bandwidth; K12 streams the rows once
        from . import ops
        return ops.rows_affine(Y.reshape(-1, G.shape[-1]), G).reshape(tuple(Y.shape[:-2]) + (G.shape[-2], 1))
    if nb == 0 and Y.dim() > 2 and Y.shape[-1] == 1:
        return (Y.squeeze(-1) @ G.transpose(0, 1)).unsqueeze(-1)  # fewer rows: one row-major GEMM
    if nb < 1 or Y.dim() < G.dim() or Y.shape[-1] != 1 or any(s != 1 for s in Y.shape[-2 - nb:-2]):
        return G @ Y
    sample = tuple(Y.shape[:-2 - nb])
    if len(sample) == 0:
        return G @ Y
    batch, n, k = tuple(G.shape[:-2]), G.shape[-2], G.shape[-1]
    out = Y.reshape(-1, k) @ G.reshape(-1, k).transpose(0, 1)
    return out.reshape(sample + batch + (n, 1))


def shared_weighted_sum(P, w):
    """sum_r w[..., r] * P[r] for SHARED matrices P ((R, a, b), no sample axes) and per-sample weights w
    (sample + (R,)): one (samples, R) x (R, a*b) GEMM instead of the (samples, R, a, b) product tensor (13 GB for the
    role-averaged likelihood precision of the flocking DMBD)."""
    R, a, b = P.shape
    return (w.reshape(-1, R) @ P.reshape(R, a * b)).reshape(tuple(w.shape[:-1]) + (a, b))


def shared_weighted_matvec(A, v, w):
    """sum_r w[..., r] * (A[r] @ v[...]) for SHARED matrices A ((R, n, k), no sample axes), per-sample vectors v (sample + (k, 1)) and
    per-sample weights w (sample + (R,)): ONE (samples, R k) x (R k, n) GEMM on the features w (x) v, instead of the per-state products
    (samples, R, n) and their weighted sum (250 MB each way for the role-averaged message of the flocking DMBD).  Returns sample + (n, 1)."""
    R, n, k = A.shape
    lead = tuple(torch.broadcast_shapes(v.shape[:-2], w.shape[:-1]))
    f = (w.expand(lead + (R,)).unsqueeze(-1) * v.expand(lead + (k, 1)).squeeze(-1).unsqueeze(-2)).reshape(-1, R * k)
    out = f @ A.transpose(-2, -1).reshape(R * k, n)
    return out.reshape(lead + (n, 1))


def rows_matmul(X, W):
    """X @ W for X = lead + (k,) with very many rows and ONE small W (k, n): K12 (one streaming pass) on the device when
    the library GEMM would be the tall-skinny case it handles poorly (_k12_pays); `@` otherwise."""
    if W.dim() == 2 and X.numel() > 0 and W.numel() > 0 and X.is_cuda and X.dim() >= 2 and _k12_pays(X.numel() // W.shape[0], W.shape[0], W.shape[1]):
        from . import ops
        return ops.rows_affine(X.reshape(-1, W.shape[0]), W.transpose(0, 1)).reshape(tuple(X.shape[:-1]) + (W.shape[1],))
    return X @ W
