#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING the reference.

This script is the only place where the reference (pyVBMP @ /root/reference) is
imported.  It runs only in the build container (the reference never travels to
the GPU box) and writes nothing but DATA: the inputs that were fed to a
reference method and the tensors that the reference returned / left in its
attributes.  No reference source text is stored.

    python tools/gen_golden.py            # all groups
    python tools/gen_golden.py niw mnw    # selected groups

Each group becomes tests/golden/<group>.npz with keys "<case>/<field>".
fp64 everywhere (torch.set_default_dtype BEFORE importing the reference, because
the reference builds its default prior tensors at import time).
"""
import os
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch

torch.set_default_dtype(torch.float64)
REF = os.environ.get("VBMP_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

import dists  # noqa: E402  (reference)
import transforms  # noqa: E402
import utils  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


class Book:
    """Collects tensors under "<case>/<field>" keys."""

    def __init__(self):
        self.d = {}
        self.case = None

    def begin(self, case):
        self.case = case

    def put(self, name, value):
        if value is None:
            return
        if isinstance(value, torch.Tensor):
            value = value.detach().clone().cpu().numpy()
        self.d[f"{self.case}/{name}"] = np.asarray(value)

    def save(self, group):
        os.makedirs(OUT, exist_ok=True)
        path = os.path.join(OUT, group + ".npz")
        np.savez_compressed(path, **self.d)
        print(f"{group}: {len(self.d)} arrays, {os.path.getsize(path)/1024:.1f} KiB")


def spd_stats(batch, D, n, gen):
    """Sufficient statistics of n pseudo-samples per batch element."""
    A = torch.randn(batch + (D, n), generator=gen)
    SExx = A @ A.transpose(-2, -1)
    SEx = A.sum(-1)
    return SExx, SEx


def rand_spd(shape, D, gen, jitter=0.5):
    A = torch.randn(shape + (D, D + 2), generator=gen)
    return A @ A.transpose(-2, -1) / (D + 2) + jitter * torch.eye(D)


# ------------------------------------------------------------------ Wishart
def snap_wishart(b, w, prefix=""):
    for f in ("invU", "U", "nu", "logdet_invU"):
        b.put(prefix + f, getattr(w, f))


def snap_wishart_expect(b, w, prefix=""):
    for f in ("mean", "meaninv", "ESigma", "EinvSigma", "invEinvSigma", "ElogdetinvSigma",
              "logdetEinvSigma", "KLqprior", "logZ"):
        b.put(prefix + f, getattr(w, f)())


def gen_wishart():
    b = Book()
    gen = torch.Generator().manual_seed(101)
    for D in (2, 6, 16):
        for lr in (1.0, 0.5):
            for beta in (None, 0.9):
                b.begin(f"w_d{D}_lr{lr}_beta{beta}")
                scale = 0.7
                w = dists.Wishart(event_shape=(D, D), batch_shape=(6,), scale=torch.tensor(scale))
                b.put("scale", scale)
                b.put("lr", lr)
                b.put("beta", -1.0 if beta is None else beta)
                snap_wishart(b, w, "init_")
                b.put("invU_0", w.invU_0)
                b.put("nu_0", w.nu_0)
                b.put("logdet_invU_0", w.logdet_invU_0)
                for step in (1, 2):
                    SExx, _ = spd_stats((6,), D, 2 * D + step, gen)
                    N = torch.full((6,), 2.0 * D + step) + torch.rand(6, generator=gen)
                    b.put(f"SExx{step}", SExx)
                    b.put(f"N{step}", N)
                    w.ss_update(SExx, N, lr=lr, beta=beta)
                    snap_wishart(b, w, f"s{step}_")
                snap_wishart_expect(b, w)
    # extra event dims + to_event
    b.begin("w_event322")
    w = dists.Wishart(event_shape=(3, 2, 2), batch_shape=(5, 6), scale=torch.tensor(1.3))
    SExx, _ = spd_stats((5, 6, 3), 2, 7, gen)
    N = 7.0 + torch.rand(5, 6, 3, generator=gen)
    b.put("SExx1", SExx)
    b.put("N1", N)
    w.ss_update(SExx, N, lr=0.8)
    snap_wishart(b, w, "s1_")
    snap_wishart_expect(b, w)
    w.to_event(1)
    b.put("KLqprior_to_event1", w.KLqprior())
    b.save("wishart")


# ---------------------------------------------------------------------- NIW
def snap_niw(b, q, prefix=""):
    b.put(prefix + "lambda_mu", q.lambda_mu)
    b.put(prefix + "mu", q.mu)
    snap_wishart(b, q.invU, prefix)


def snap_niw_expect(b, q, prefix=""):
    for f in ("mean", "EX", "EXXT", "ESigma", "ElogdetinvSigma", "EinvSigmamu", "EinvSigma", "EinvUX",
              "EXTinvUX", "KLqprior"):
        b.put(prefix + f, getattr(q, f)())


def gen_niw():
    b = Book()
    gen = torch.Generator().manual_seed(202)

    # --- D=16 flat batch: the headline layout (config 2, small)
    for lr in (1.0, 0.5):
        b.begin(f"niw_d16_lr{lr}")
        q = dists.NormalInverseWishart(event_shape=(16,), batch_shape=(8,))
        b.put("lr", lr)
        snap_niw(b, q, "init_")
        for step in (1, 2):
            SExx, SEx = spd_stats((8,), 16, 32, gen)
            N = torch.full((8,), 32.0)
            b.put(f"SExx{step}", SExx)
            b.put(f"SEx{step}", SEx)
            b.put(f"N{step}", N)
            q.ss_update(SExx, SEx, N, lr=lr)  # beta defaults to 0.0
            snap_niw(b, q, f"s{step}_")
        snap_niw_expect(b, q)
        X1 = torch.randn(7, 1, 16, generator=gen)
        X2 = torch.randn(7, 8, 16, generator=gen)
        b.put("X_bcast", X1)
        b.put("X_full", X2)
        b.put("Elog_like_bcast", q.Elog_like(X1))
        b.put("Elog_like_full", q.Elog_like(X2))

    # --- forgetting factor
    b.begin("niw_beta0.9")
    q = dists.NormalInverseWishart(event_shape=(6,), batch_shape=(8,), scale=torch.tensor(0.5))
    snap_niw(b, q, "init_")
    for step in (1, 2, 3):
        SExx, SEx = spd_stats((8,), 6, 10, gen)
        N = 10.0 + torch.rand(8, generator=gen)
        b.put(f"SExx{step}", SExx)
        b.put(f"SEx{step}", SEx)
        b.put(f"N{step}", N)
        q.ss_update(SExx, SEx, N, lr=0.7, beta=0.9)
        snap_niw(b, q, f"s{step}_")
    snap_niw_expect(b, q)

    # --- raw_update, with and without responsibilities
    b.begin("niw_raw")
    q = dists.NormalInverseWishart(event_shape=(5,), batch_shape=(4,))
    snap_niw(b, q, "init_")
    X = torch.randn(50, 1, 5, generator=gen) * 2.0 + 1.0
    p = torch.softmax(torch.randn(50, 4, generator=gen), -1)
    b.put("X", X)
    b.put("p", p)
    q.raw_update(X, p, lr=1.0)
    snap_niw(b, q, "p_")
    q.raw_update(X, p, lr=0.3)
    snap_niw(b, q, "p2_")
    Xf = torch.randn(50, 4, 5, generator=gen)
    b.put("X_full", Xf)
    q.raw_update(Xf, None, lr=1.0)
    snap_niw(b, q, "nop_")
    snap_niw_expect(b, q)

    # --- extra event dims (3,2) and batch (5,6)
    b.begin("niw_e32_b56")
    q = dists.NormalInverseWishart(event_shape=(3, 2), batch_shape=(5, 6), scale=torch.tensor(0.8))
    snap_niw(b, q, "init_")
    SExx, SEx = spd_stats((5, 6, 3), 2, 9, gen)
    N = 9.0 + torch.rand(5, 6, 1, generator=gen)
    b.put("SExx1", SExx)
    b.put("SEx1", SEx)
    b.put("N1", N)
    q.ss_update(SExx, SEx, N, lr=1.0)
    snap_niw(b, q, "s1_")
    X = torch.randn(9, 1, 1, 3, 2, generator=gen)
    p = torch.softmax(torch.randn(9, 5 * 6, generator=gen), -1).view(9, 5, 6)
    b.put("X", X)
    b.put("p", p)
    b.put("Elog_like", q.Elog_like(X))
    q.raw_update(X, p, lr=0.6)
    snap_niw(b, q, "raw_")
    snap_niw_expect(b, q)

    # --- to_event(1): event (6,2), batch (5,)
    b.begin("niw_toevent")
    q = dists.NormalInverseWishart(event_shape=(2,), batch_shape=(5, 6))
    snap_niw(b, q, "init_")
    SExx, SEx = spd_stats((5, 6), 2, 6, gen)
    N = 6.0 + torch.rand(5, 6, generator=gen)
    b.put("SExx1", SExx)
    b.put("SEx1", SEx)
    b.put("N1", N)
    q.ss_update(SExx, SEx, N)
    q.to_event(1)
    X = torch.randn(7, 1, 6, 2, generator=gen)
    b.put("X", X)
    b.put("Elog_like", q.Elog_like(X))
    b.put("KLqprior", q.KLqprior())

    # --- fixed precision
    b.begin("niw_fixed_precision")
    q = dists.NormalInverseWishart(event_shape=(4,), batch_shape=(3,), fixed_precision=True)
    snap_niw(b, q, "init_")
    SExx, SEx = spd_stats((3,), 4, 8, gen)
    N = torch.full((3,), 8.0)
    b.put("SExx1", SExx)
    b.put("SEx1", SEx)
    b.put("N1", N)
    q.ss_update(SExx, SEx, N, lr=0.9)
    snap_niw(b, q, "s1_")

    # --- user prior
    b.begin("niw_prior")
    prior = {"lambda_mu": torch.tensor(2.0), "mu": torch.tensor(0.5),
             "nu": torch.full((3,), 9.0), "invU": rand_spd((3,), 4, gen)}
    for k, v in prior.items():
        b.put("prior_" + k, v)
    q = dists.NormalInverseWishart(event_shape=(4,), batch_shape=(3,), prior_parms=prior)
    snap_niw(b, q, "init_")
    SExx, SEx = spd_stats((3,), 4, 8, gen)
    N = torch.full((3,), 8.0)
    b.put("SExx1", SExx)
    b.put("SEx1", SEx)
    b.put("N1", N)
    q.ss_update(SExx, SEx, N)
    snap_niw(b, q, "s1_")
    snap_niw_expect(b, q)
    b.save("niw")


# ---------------------------------------------------------------------- MVN
def gen_mvn():
    b = Book()
    gen = torch.Generator().manual_seed(303)
    D = 5
    # plain format
    b.begin("mvn_from_moments")
    mu = torch.randn(4, 3, D, generator=gen)
    Sigma = rand_spd((4, 3), D, gen)
    b.put("mu", mu)
    b.put("Sigma", Sigma)
    q = dists.MultivariateNormal(mu=mu, Sigma=Sigma)
    for f in ("EinvSigma", "EinvSigmamu", "ElogdetinvSigma", "EXXT", "EXTX", "EX", "mean", "ESigma"):
        b.put(f, getattr(q, f)())
    X = torch.randn(6, 4, 3, D, generator=gen)
    b.put("X", X)
    b.put("Elog_like", q.Elog_like(X))

    b.begin("mvn_from_natural")
    invSigma = rand_spd((4, 3), D, gen)
    invSigmamu = torch.randn(4, 3, D, generator=gen)
    b.put("invSigma", invSigma)
    b.put("invSigmamu", invSigmamu)
    q = dists.MultivariateNormal(invSigma=invSigma, invSigmamu=invSigmamu)
    for f in ("mean", "ESigma", "ElogdetinvSigma", "EXXT", "EXTX"):
        b.put(f, getattr(q, f)())

    b.begin("mvn_updates")
    q = dists.MultivariateNormal(mu=torch.zeros(3, D), Sigma=torch.eye(D).expand(3, D, D))
    X = torch.randn(40, 1, D, generator=gen)
    p = torch.softmax(torch.randn(40, 3, generator=gen), -1)
    b.put("X", X)
    b.put("p", p)
    q.raw_update(X, p)
    b.put("p_mu", q.mu)
    b.put("p_Sigma", q.Sigma)
    Xf = torch.randn(40, 3, D, generator=gen)
    b.put("X_full", Xf)
    q.raw_update(Xf)
    b.put("nop_mu", q.mu)
    b.put("nop_Sigma", q.Sigma)
    b.put("nop_EinvSigma", q.EinvSigma())
    b.put("nop_EinvSigmamu", q.EinvSigmamu())

    # vector format
    b.begin("vf_from_moments")
    mu = torch.randn(4, 3, D, 1, generator=gen)
    Sigma = rand_spd((4, 3), D, gen)
    b.put("mu", mu)
    b.put("Sigma", Sigma)
    q = dists.MultivariateNormal_vector_format(mu=mu, Sigma=Sigma)
    for f in ("EinvSigma", "EinvSigmamu", "ElogdetinvSigma", "EXXT", "EXTX", "Res", "mean", "ESigma"):
        b.put(f, getattr(q, f)())
    X = torch.randn(6, 4, 3, D, 1, generator=gen)
    b.put("X", X)
    b.put("Elog_like", q.Elog_like(X))

    b.begin("vf_from_natural")
    invSigma = rand_spd((4, 3), D, gen)
    invSigmamu = torch.randn(4, 3, D, 1, generator=gen)
    b.put("invSigma", invSigma)
    b.put("invSigmamu", invSigmamu)
    q = dists.MultivariateNormal_vector_format(invSigma=invSigma, invSigmamu=invSigmamu)
    for f in ("mean", "ESigma", "ElogdetinvSigma", "EXXT", "EXTX", "Res"):
        b.put(f, getattr(q, f)())
    other_P = rand_spd((4, 3), D, gen)
    other_eta = torch.randn(4, 3, D, 1, generator=gen)
    b.put("other_invSigma", other_P)
    b.put("other_invSigmamu", other_eta)
    q.nat_combiner(other_P, other_eta)
    b.put("nat_invSigma", q.invSigma)
    b.put("nat_invSigmamu", q.invSigmamu)
    b.put("nat_mean", q.mean())
    b.put("nat_Res", q.Res())
    o = dists.MultivariateNormal_vector_format(invSigma=other_P, invSigmamu=other_eta)
    q.combiner(o)
    b.put("comb_invSigma", q.invSigma)
    b.put("comb_invSigmamu", q.invSigmamu)
    b.put("comb_ESigma", q.ESigma())
    u = q.unsqueeze(-3)
    b.put("unsq_invSigma_shape", np.array(u.invSigma.shape))
    b.put("unsq_batch_shape", np.array(u.batch_shape))

    b.begin("vf_updates")
    q = dists.MultivariateNormal_vector_format(mu=torch.zeros(3, D, 1), Sigma=torch.eye(D).expand(3, D, D))
    X = torch.randn(40, 1, D, 1, generator=gen)
    p = torch.softmax(torch.randn(40, 3, generator=gen), -1)
    b.put("X", X)
    b.put("p", p)
    q.raw_update(X, p)
    b.put("p_mu", q.mu)
    b.put("p_Sigma", q.Sigma)
    Xf = torch.randn(40, 3, D, 1, generator=gen)
    b.put("X_full", Xf)
    q.raw_update(Xf)
    b.put("nop_mu", q.mu)
    b.put("nop_Sigma", q.Sigma)
    b.save("mvn")


# ------------------------------------------------------------- matrix_utils
def gen_matrix_utils():
    b = Book()
    gen = torch.Generator().manual_seed(404)
    mu_ = utils.matrix_utils
    for name, (na, nd, batch) in {"mu_4_3": (4, 3, (7,)), "mu_6_6": (6, 6, (2, 5)), "mu_16_8": (16, 8, (3,))}.items():
        b.begin(name)
        J = rand_spd(batch, na + nd, gen, jitter=1.0)
        A, B = J[..., :na, :na], J[..., :na, na:]
        C, D = J[..., na:, :na], J[..., na:, na:]
        for k, v in (("A", A), ("B", B), ("C", C), ("D", D)):
            b.put(k, v)
        b.put("block_diag", mu_.block_diag_matrix_builder(A, D))
        b.put("block_build", mu_.block_matrix_builder(A, B, C, D))
        for form in ("left", "right", "True"):
            out = mu_.block_matrix_inverse(A, B, C, D, block_form=form)
            for i, o in enumerate(out):
                b.put(f"inv_{form}_{i}", o)
        b.put("inv_full", mu_.block_matrix_inverse(A, B, C, D, block_form=False))
        b.put("inv_default", mu_.block_matrix_inverse(A, B, C, D))
        out = mu_.block_precision_marginalizer(A, B, C, D)
        for i, o in enumerate(out):
            b.put(f"marg_{i}", o)
        b.put("logdet", mu_.block_matrix_logdet(A, B, C, D))
        b.put("logdet_A", mu_.block_matrix_logdet(A, B, C, D, singular="A"))
        b.put("logdet_D", mu_.block_matrix_logdet(A, B, C, D, singular="D"))
    b.save("matrix_utils")


# ---------------------------------------------------------------------- MNW
def snap_mnw(b, m, prefix=""):
    for f in ("mu", "invV", "V", "logdetinvV"):
        b.put(prefix + f, getattr(m, f))
    snap_wishart(b, m.invU, prefix + "invU_")


def snap_mnw_expect(b, m, prefix=""):
    for f in ("EinvUX", "EXTinvU", "EXTinvUX", "EXinvVXT", "EXmMUTinvUXmMU", "EXmMUinvVXmMUT", "ElogdetinvU",
              "logdetEinvSigma", "ElogdetinvSigma", "EinvSigma", "invEinvSigma", "ESigma", "KLqprior", "mean",
              "weights", "var"):
        b.put(prefix + f, getattr(m, f)())
    if m.batch_dim == 0 and m.event_dim == 2:
        b.put(prefix + "EXTX", m.EXTX())
        b.put(prefix + "EXXT", m.EXXT())
        A = torch.eye(m.p) * 0.5 + 0.1
        An = torch.eye(m.n) * 0.5 + 0.1
        b.put(prefix + "EXTAX", m.EXTAX(An))
        b.put(prefix + "EXAXT", m.EXAXT(A))


def mnw_case(b, name, n, p, batch, pad_X, gen, mask=None, X_mask=None, N=24, lr2=0.5):
    b.begin(name)
    m = transforms.MatrixNormalWishart(event_shape=(n, p), batch_shape=batch, pad_X=pad_X, mask=mask, X_mask=X_mask)
    b.put("n", n)
    b.put("p", p)
    b.put("pad_X", int(pad_X))
    b.put("batch_shape", np.array(batch, dtype=np.int64))
    if mask is not None:
        b.put("mask", mask)
    if X_mask is not None:
        b.put("X_mask", X_mask)
    snap_mnw(b, m, "init_")
    nb = len(batch)
    W = torch.randn(batch + (n, p), generator=gen)
    X = torch.randn((N,) + (1,) * nb + (p, 1), generator=gen)
    Y = W @ X + 0.3 * torch.randn((N,) + batch + (n, 1), generator=gen)
    if nb:
        pr = torch.softmax(torch.randn((N,) + batch, generator=gen), -1)
    else:
        pr = None
    b.put("X", X)
    b.put("Y", Y)
    b.put("p_resp", pr)
    # raw_update (data), lr=1 then lr2
    Xe = X.expand((N,) + batch + (p, 1))
    m.raw_update(Xe, Y, p=pr, lr=1.0)
    snap_mnw(b, m, "raw1_")
    m.raw_update(Xe, Y, p=pr, lr=lr2)
    snap_mnw(b, m, "raw2_")
    snap_mnw_expect(b, m, "raw2_")
    # likelihoods
    b.put("Elog_like", m.Elog_like(X, Y))
    iSx, iSmx, R = m.Elog_like_X(Y)
    b.put("ELX_invSigma", iSx)
    b.put("ELX_invSigmamu", iSmx)
    b.put("ELX_Res", R)
    pY, R = m.predict(X)
    b.put("predict_invSigma", pY.invSigma)
    b.put("predict_invSigmamu", pY.invSigmamu)
    b.put("predict_Res", R)
    pX, R = m.postdict(Y)
    b.put("postdict_invSigma", pX.invSigma)
    b.put("postdict_invSigmamu", pX.invSigmamu)
    b.put("postdict_Res", R)

    # messages: per-sample precision (config-3 shape) and shared precision
    px_P = rand_spd((N,) + (1,) * nb, p, gen)
    px_eta = torch.randn((N,) + (1,) * nb + (p, 1), generator=gen)
    b.put("fw_in_invSigma", px_P)
    b.put("fw_in_invSigmamu", px_eta)
    pXm = dists.MultivariateNormal_vector_format(invSigma=px_P.clone(), invSigmamu=px_eta.clone())
    pYm, Res = m.forward(pXm)
    b.put("fw_mu", pYm.mu)
    b.put("fw_Sigma", pYm.Sigma)
    b.put("fw_Res", Res)
    shared_P = rand_spd((), p, gen)
    b.put("fws_in_invSigma", shared_P)
    pXs = dists.MultivariateNormal_vector_format(invSigma=shared_P.clone(), invSigmamu=px_eta.clone())
    pYs, Res = m.forward(pXs)
    b.put("fws_mu", pYs.mu)
    b.put("fws_Sigma", pYs.Sigma)
    b.put("fws_Res", Res)

    py_P = rand_spd((N,) + (1,) * nb, n, gen)
    py_eta = torch.randn((N,) + (1,) * nb + (n, 1), generator=gen)
    b.put("bw_in_invSigma", py_P)
    b.put("bw_in_invSigmamu", py_eta)
    pYb = dists.MultivariateNormal_vector_format(invSigma=py_P.clone(), invSigmamu=py_eta.clone())
    pXb, Res = m.backward(pYb)
    b.put("bw_invSigma", pXb.invSigma)
    b.put("bw_invSigmamu", pXb.invSigmamu)
    b.put("bw_Res", Res)
    shared_Py = rand_spd((), n, gen)
    b.put("bws_in_invSigma", shared_Py)
    pYbs = dists.MultivariateNormal_vector_format(invSigma=shared_Py.clone(), invSigmamu=py_eta.clone())
    pXbs, Res = m.backward(pYbs, Res=0.25)
    b.put("bws_invSigma", pXbs.invSigma)
    b.put("bws_invSigmamu", pXbs.invSigmamu)
    b.put("bws_Res", Res)

    pYq = dists.MultivariateNormal_vector_format(invSigma=py_P.clone(), invSigmamu=py_eta.clone())
    pxo, Res = m.Elog_like_X_given_pY(pYq)
    b.put("ELXpY_invSigma", pxo.invSigma)
    b.put("ELXpY_invSigmamu", pxo.invSigmamu)
    b.put("ELXpY_mu", pxo.mu)
    b.put("ELXpY_Sigma", pxo.Sigma)
    b.put("ELXpY_Res", Res)

    # update(pX, pY, p) with distributions: pX = MVN_vf (per-sample cov), pY = Delta(Y)
    ux_Sigma = rand_spd((N,) + (1,) * nb, p - (1 if pad_X else 0) if False else m.p - (1 if pad_X else 0), gen)
    ux_mu = torch.randn((N,) + (1,) * nb + (m.p - (1 if pad_X else 0), 1), generator=gen)
    b.put("upd_x_mu", ux_mu)
    b.put("upd_x_Sigma", ux_Sigma)
    pXu = dists.MultivariateNormal_vector_format(mu=ux_mu.expand((N,) + batch + ux_mu.shape[-2:]).clone(),
                                                 Sigma=ux_Sigma.expand((N,) + batch + ux_Sigma.shape[-2:]).clone())
    b.put("ELpXpY", m.Elog_like_given_pX_pY(pXu, dists.Delta(Y)))
    m.update(pXu, dists.Delta(Y), p=pr, lr=0.8)
    snap_mnw(b, m, "upd_")
    # also a Gaussian pY
    uy_Sigma = rand_spd((N,) + (1,) * nb, n, gen) * 0.1
    b.put("upd_y_Sigma", uy_Sigma)
    pYu = dists.MultivariateNormal_vector_format(mu=Y.clone(), Sigma=uy_Sigma.expand((N,) + batch + (n, n)).clone())
    m.update(pXu, pYu, p=pr, lr=1.0, beta=0.5)
    snap_mnw(b, m, "upd2_")
    m.update(pXu, pYu, p=pr, lr=1.0, beta=0.5)
    snap_mnw(b, m, "upd3_")
    b.put("KLqprior_end", m.KLqprior())


def gen_mnw():
    b = Book()
    gen = torch.Generator().manual_seed(505)
    mnw_case(b, "mnw_4x3_b5", 4, 3, (5,), False, gen)
    mnw_case(b, "mnw_4x3_b5_pad", 4, 3, (5,), True, gen)
    mnw_case(b, "mnw_4x3_nobatch", 4, 3, (), False, gen)
    mnw_case(b, "mnw_32x32", 32, 32, (), False, gen, N=6)
    mnw_case(b, "mnw_6x7_b2_pad", 6, 7, (2,), True, gen)
    # masks (shared by the batch)
    X_mask = (torch.rand(1, 3, generator=gen) > 0.3)
    X_mask[..., 0] = True
    mnw_mask_case(b, "mnw_Xmask", 4, 3, (5,), False, gen, X_mask=X_mask)
    mask = torch.rand(4, 3, generator=gen) > 0.3
    mask[0, 0] = True
    mnw_mask_case(b, "mnw_mask", 4, 3, (5,), False, gen, mask=mask)
    mnw_mask_case(b, "mnw_mask_pad", 4, 3, (5,), True, gen, mask=mask)
    b.save("mnw")


def mnw_mask_case(b, name, n, p, batch, pad_X, gen, mask=None, X_mask=None, N=24):
    b.begin(name)
    m = transforms.MatrixNormalWishart(event_shape=(n, p), batch_shape=batch, pad_X=pad_X,
                                       mask=None if mask is None else mask.clone(),
                                       X_mask=None if X_mask is None else X_mask.clone())
    b.put("n", n)
    b.put("p", p)
    b.put("pad_X", int(pad_X))
    b.put("batch_shape", np.array(batch, dtype=np.int64))
    if mask is not None:
        b.put("mask", mask)
    if X_mask is not None:
        b.put("X_mask", X_mask)
    snap_mnw(b, m, "init_")
    b.put("init_mu_0", m.mu_0)
    W = torch.randn(batch + (n, p), generator=gen)
    X = torch.randn((N,) + (1,) * len(batch) + (p, 1), generator=gen)
    Y = W @ X + 0.3 * torch.randn((N,) + batch + (n, 1), generator=gen)
    pr = torch.softmax(torch.randn((N,) + batch, generator=gen), -1)
    b.put("X", X)
    b.put("Y", Y)
    b.put("p_resp", pr)
    Xe = X.expand((N,) + batch + (p, 1))
    m.raw_update(Xe, Y, p=pr, lr=1.0)
    snap_mnw(b, m, "raw1_")
    m.raw_update(Xe, Y, p=pr, lr=0.5)
    snap_mnw(b, m, "raw2_")
    b.put("KLqprior", m.KLqprior())
    b.put("Elog_like", m.Elog_like(X, Y))


# ---------------------------------------------------------------------- GMM
def two_moons(n, gen, noise=0.08):
    """Our own two-interleaved-half-circles generator (shape of examples/two_moons.py data)."""
    t = torch.rand(n, generator=gen) * np.pi
    half = torch.arange(n) % 2
    x = torch.where(half == 0, torch.cos(t), 1.0 - torch.cos(t))
    y = torch.where(half == 0, torch.sin(t), 0.5 - torch.sin(t))
    return torch.stack((x, y), -1) + noise * torch.randn(n, 2, generator=gen)


def gen_gmm():
    import models  # reference

    b = Book()
    gen = torch.Generator().manual_seed(606)
    b.begin("gmm_k4_d2")
    data = two_moons(400, gen)
    b.put("data", data)
    torch.manual_seed(7)
    g = models.GaussianMixtureModel(4, 2)
    idx = torch.randint(400, (4,), generator=gen)
    g.dist.mu = data[idx, :].clone()
    b.put("init_mu", g.dist.mu)
    b.put("init_alpha", g.pi.alpha)
    b.put("alpha_0", g.pi.alpha_0)
    b.put("init_invU_0", g.dist.invU.invU_0)
    for it in range(1, 21):
        g.update(data, iters=1, lr=1.0, verbose=False)
        if it in (1, 2, 5, 20):
            pre = f"it{it}_"
            b.put(pre + "p", g.p)
            b.put(pre + "NA", g.NA)
            b.put(pre + "logZ", g.logZ)
            b.put(pre + "ELBO", g.ELBO_last)
            b.put(pre + "alpha", g.pi.alpha)
            snap_niw(b, g.dist, pre)
    b.put("final_assignment", g.assignment())
    b.put("final_KLqprior", g.KLqprior())

    # generic Mixture with non-trivial batch/event shapes (reference tests/test_dists.py shape)
    b.begin("mixture_b3_k6_e32")
    torch.manual_seed(8)
    niw = dists.NormalInverseWishart(event_shape=(3, 2), batch_shape=(3, 6))
    mix = dists.Mixture(niw, event_shape=(6,))
    X = torch.randn(50, 3, 3, 2, generator=gen) + torch.randn(1, 3, 3, 2, generator=gen)
    b.put("X", X)
    b.put("init_mu", niw.mu)
    b.put("init_alpha", mix.pi.alpha)
    for it in (1, 2, 3):
        mix.update(X, iters=1, lr=0.9)
        pre = f"it{it}_"
        b.put(pre + "p", mix.p)
        b.put(pre + "NA", mix.NA)
        b.put(pre + "logZ", mix.logZ)
        b.put(pre + "ELBO", mix.ELBO_last)
        b.put(pre + "alpha", mix.pi.alpha)
        snap_niw(b, mix.dist, pre)
    b.save("gmm")


GROUPS = {"wishart": gen_wishart, "niw": gen_niw, "mvn": gen_mvn, "matrix_utils": gen_matrix_utils,
          "mnw": gen_mnw, "gmm": gen_gmm}



# ---------------------------------------------------------------------- LDS
def snap_lds_state(b, m, pre):
    snap_niw(b, m.x0, pre + "x0_")
    snap_mnw(b, m.A, pre + "A_")
    snap_mnw(b, m.obs_model, pre + "obs_")


def lds_case(b, name, T, S, obs_shape, hidden, batch, gen, control=0, regression=0, iters=2, lr=1.0):
    import contextlib
    import io

    import models  # reference
    b.begin(name)
    with contextlib.redirect_stdout(io.StringIO()):
        m = models.LinearDynamicalSystems(obs_shape, hidden, control_dim=control, regression_dim=regression,
                                          latent_noise='shared', batch_shape=batch)
    b.put("T", T)
    b.put("S", S)
    b.put("hidden", hidden)
    b.put("obs_shape", np.array(obs_shape, dtype=np.int64))
    b.put("batch_shape", np.array(batch, dtype=np.int64))
    b.put("control", control)
    b.put("regression", regression)
    b.put("lr", lr)
    snap_lds_state(b, m, "init_")
    # a smooth latent trajectory observed through a random map
    nb = len(batch)
    tt = torch.arange(T, dtype=torch.float64).reshape(T, 1, 1)
    lat = torch.cat([torch.sin(0.2 * tt * (k + 1) + torch.rand(1, S, 1, generator=gen) * 6) for k in range(hidden)], -1)
    W = torch.randn(obs_shape + (hidden,), generator=gen)
    y = (W @ lat.reshape((T, S) + (1,) * (len(obs_shape) - 1) + (hidden, 1))).squeeze(-1)
    y = y + 0.1 * torch.randn(y.shape, generator=gen)
    u = torch.randn(T, S, control, generator=gen) if control else None
    r = torch.randn((T, S) + obs_shape[:-1] + (regression,), generator=gen) if regression else None
    b.put("y", y)
    b.put("u", u)
    b.put("r", r)
    m.expand_to_batch = nb > 0
    yy, uu, rr = m.reshape_inputs(y, u, r)
    for it in range(1, iters + 1):
        m.update_latents(yy, uu, rr)
        pre = f"it{it}_"
        for f in ("mu", "Sigma", "invSigma", "invSigmamu"):
            b.put(pre + "px_" + f, getattr(m.px, f))
        for f in ("SE_x_x", "SE_x0_x0", "SE_x0", "SE_y_xr", "SE_y_y", "SE_xpu_xpu", "SE_x_xpu", "SE_xr_xr", "T", "N",
                  "logZ"):
            b.put(pre + f, getattr(m, f))
        b.put(pre + "ELBO", m.ELBO())
        m.ss_update(p=None, lr=lr)
        m.obs_model.ss_update(m.SE_xr_xr, m.SE_y_xr, m.SE_y_y, m.T, lr)
        snap_lds_state(b, m, pre)
        for f in ("invQ", "ATQA_x_x", "invATQA_x_x", "logdetATQA_x_x", "ATQA_x_u", "ATQA_u_u", "QA_xp_x", "QA_xp_u"):
            b.put(pre + f, getattr(m, f))
    b.put("KLqprior", m.KLqprior())


def gen_lds():
    b = Book()
    gen = torch.Generator().manual_seed(707)
    torch.manual_seed(17)
    lds_case(b, "lds_h6_o6", 30, 3, (6,), 6, (), gen, iters=3)
    lds_case(b, "lds_h6_o6_lr", 25, 4, (6,), 6, (), gen, iters=2, lr=0.6)
    lds_case(b, "lds_h3_o5_ctrl_reg", 20, 3, (5,), 3, (), gen, control=2, regression=2, iters=2)
    lds_case(b, "lds_h2_o32", 16, 2, (3, 2), 2, (), gen, iters=2)
    lds_case(b, "lds_h4_o5_batch2", 18, 3, (5,), 4, (2,), gen, iters=2)
    lds_case(b, "lds_h8_o4", 12, 2, (4,), 8, (), gen, control=1, iters=2)
    b.save("lds")


GROUPS["lds"] = gen_lds


# ---------------------------------------------------------------------- MatrixNormalGamma
def snap_mng(b, m, prefix=""):
    for f in ("mu", "invV", "V", "logdetinvV"):
        b.put(prefix + f, getattr(m, f))
    b.put(prefix + "alpha", m.invU.gamma.alpha)
    b.put(prefix + "beta", m.invU.gamma.beta)


def mng_case(b, name, n, p, batch, pad_X, gen, mask=None, N=24):
    b.begin(name)
    m = transforms.MatrixNormalGamma(event_shape=(n, p), batch_shape=batch, pad_X=pad_X,
                                     mask=None if mask is None else mask.clone())
    b.put("n", n)
    b.put("p", p)
    b.put("pad_X", int(pad_X))
    b.put("batch_shape", np.array(batch, dtype=np.int64))
    if mask is not None:
        b.put("mask", mask)
    snap_mng(b, m, "init_")
    nb = len(batch)
    W = torch.randn(batch + (n, p), generator=gen)
    X = torch.randn((N,) + (1,) * nb + (p, 1), generator=gen)
    Y = W @ X + 0.3 * torch.randn((N,) + batch + (n, 1), generator=gen)
    pr = torch.softmax(torch.randn((N,) + batch, generator=gen), -1) if nb else None
    b.put("X", X)
    b.put("Y", Y)
    b.put("p_resp", pr)
    Xe = X.expand((N,) + batch + (p, 1))
    m.raw_update(Xe, Y, p=pr, lr=1.0)
    snap_mng(b, m, "raw1_")
    m.raw_update(Xe, Y, p=pr, lr=0.5)
    snap_mng(b, m, "raw2_")
    for f in ("EinvUX", "EXTinvU", "EXTinvUX", "EXinvVXT", "ElogdetinvU", "ElogdetinvSigma", "EinvSigma", "ESigma",
              "KLqprior", "mean", "weights", "var"):
        b.put("raw2_" + f, getattr(m, f)())
    b.put("Elog_like", m.Elog_like(X, Y))
    iSx, iSmx, R = m.Elog_like_X(Y)
    b.put("ELX_invSigma", iSx)
    b.put("ELX_invSigmamu", iSmx)
    b.put("ELX_Res", R)
    pY, R = m.predict(X)
    b.put("predict_invSigma", pY.invSigma)
    b.put("predict_invSigmamu", pY.invSigmamu)
    b.put("predict_Res", R)
    px_P = rand_spd((N,) + (1,) * nb, p, gen)
    px_eta = torch.randn((N,) + (1,) * nb + (p, 1), generator=gen)
    b.put("fw_in_invSigma", px_P)
    b.put("fw_in_invSigmamu", px_eta)
    pYm = m.forward(dists.MultivariateNormal_vector_format(invSigma=px_P.clone(), invSigmamu=px_eta.clone()))
    b.put("fw_invSigma", pYm.invSigma)
    b.put("fw_invSigmamu", pYm.invSigmamu)
    py_P = rand_spd((N,) + (1,) * nb, n, gen)
    py_eta = torch.randn((N,) + (1,) * nb + (n, 1), generator=gen)
    b.put("bw_in_invSigma", py_P)
    b.put("bw_in_invSigmamu", py_eta)
    pXb, Res = m.backward(dists.MultivariateNormal_vector_format(invSigma=py_P.clone(), invSigmamu=py_eta.clone()))
    b.put("bw_invSigma", pXb.invSigma)
    b.put("bw_invSigmamu", pXb.invSigmamu)
    b.put("bw_Res", Res)
    pd = m.p - (1 if pad_X else 0)
    ux_Sigma = rand_spd((N,) + (1,) * nb, int(pd), gen)
    ux_mu = torch.randn((N,) + (1,) * nb + (int(pd), 1), generator=gen)
    b.put("upd_x_mu", ux_mu)
    b.put("upd_x_Sigma", ux_Sigma)
    pXu = dists.MultivariateNormal_vector_format(mu=ux_mu.expand((N,) + batch + ux_mu.shape[-2:]).clone(),
                                                 Sigma=ux_Sigma.expand((N,) + batch + ux_Sigma.shape[-2:]).clone())
    b.put("ELpXpY", m.Elog_like_given_pX_pY(pXu, dists.Delta(Y)))
    m.update(pXu, dists.Delta(Y), p=pr, lr=0.8)
    snap_mng(b, m, "upd_")
    b.put("KLqprior_end", m.KLqprior())


def gen_mng():
    b = Book()
    gen = torch.Generator().manual_seed(808)
    torch.manual_seed(18)
    mng_case(b, "mng_4x3_b5", 4, 3, (5,), False, gen)
    mng_case(b, "mng_4x3_b5_pad", 4, 3, (5,), True, gen)
    mng_case(b, "mng_6x7_nobatch", 6, 7, (), False, gen)
    mask = torch.rand(4, 3, generator=gen) > 0.3
    mask[0, 0] = True
    mng_case(b, "mng_mask", 4, 3, (5,), False, gen, mask=mask)
    b.save("mng")


def snap_lds_state_mng(b, m, pre):
    snap_niw(b, m.x0, pre + "x0_")
    snap_mng(b, m.A, pre + "A_")
    snap_mnw(b, m.obs_model, pre + "obs_")


def lds_mng_case(b, name, T, S, obs_shape, hidden, batch, gen, control=0, regression=0, iters=2, lr=1.0):
    """LinearDynamicalSystems with the reference's DEFAULT transition (MatrixNormalGamma)."""
    import contextlib
    import io

    import models  # reference
    b.begin(name)
    with contextlib.redirect_stdout(io.StringIO()):
        m = models.LinearDynamicalSystems(obs_shape, hidden, control_dim=control, regression_dim=regression,
                                          batch_shape=batch)
    for k, v in (("T", T), ("S", S), ("hidden", hidden), ("control", control), ("regression", regression), ("lr", lr)):
        b.put(k, v)
    b.put("obs_shape", np.array(obs_shape, dtype=np.int64))
    b.put("batch_shape", np.array(batch, dtype=np.int64))
    snap_lds_state_mng(b, m, "init_")
    tt = torch.arange(T, dtype=torch.float64).reshape(T, 1, 1)
    lat = torch.cat([torch.sin(0.2 * tt * (k + 1) + torch.rand(1, S, 1, generator=gen) * 6) for k in range(hidden)], -1)
    W = torch.randn(obs_shape + (hidden,), generator=gen)
    y = (W @ lat.reshape((T, S) + (1,) * (len(obs_shape) - 1) + (hidden, 1))).squeeze(-1)
    y = y + 0.1 * torch.randn(y.shape, generator=gen)
    u = torch.randn(T, S, control, generator=gen) if control else None
    r = torch.randn((T, S) + obs_shape[:-1] + (regression,), generator=gen) if regression else None
    b.put("y", y)
    b.put("u", u)
    b.put("r", r)
    m.expand_to_batch = len(batch) > 0
    yy, uu, rr = m.reshape_inputs(y, u, r)
    for it in range(1, iters + 1):
        m.update_latents(yy, uu, rr)
        pre = f"it{it}_"
        for f in ("mu", "Sigma", "invSigma", "invSigmamu"):
            b.put(pre + "px_" + f, getattr(m.px, f))
        for f in ("SE_x_x", "SE_x0_x0", "SE_x0", "SE_y_xr", "SE_y_y", "SE_xpu_xpu", "SE_x_xpu", "SE_xr_xr", "T", "N",
                  "logZ"):
            b.put(pre + f, getattr(m, f))
        b.put(pre + "ELBO", m.ELBO())
        m.ss_update(p=None, lr=lr)
        m.obs_model.ss_update(m.SE_xr_xr, m.SE_y_xr, m.SE_y_y, m.T, lr)
        snap_lds_state_mng(b, m, pre)
    b.put("KLqprior", m.KLqprior())


def gen_lds_mng():
    b = Book()
    gen = torch.Generator().manual_seed(909)
    torch.manual_seed(19)
    lds_mng_case(b, "ldsg_h6_o6", 24, 3, (6,), 6, (), gen, iters=3)
    lds_mng_case(b, "ldsg_h3_o5_ctrl_reg", 18, 3, (5,), 3, (), gen, control=2, regression=2, iters=2, lr=0.7)
    lds_mng_case(b, "ldsg_h4_o5_batch2", 16, 3, (5,), 4, (2,), gen, iters=2)
    b.save("lds_mng")


GROUPS["mng"] = gen_mng
GROUPS["lds_mng"] = gen_lds_mng


# ---------------------------------------------------------------------- DMBD
def snap_dmbd(b, m, pre):
    snap_niw(b, m.x0, pre + "x0_")
    snap_mng(b, m.A, pre + "A_")
    snap_mnw(b, m.B, pre + "B_")
    b.put(pre + "trans_alpha", m.obs_model.transition.alpha)
    b.put(pre + "init_alpha", m.obs_model.initial.alpha)


def dmbd_case(b, name, T, S, n_obs, obs_dim, role_dims, hidden_dims, gen, number_of_objects=1, iters=3, latent_iters=1):
    import contextlib
    import io

    import models  # reference
    b.begin(name)
    with contextlib.redirect_stdout(io.StringIO()):
        m = models.DynamicMarkovBlanketDiscovery(obs_shape=(n_obs, obs_dim), role_dims=role_dims, hidden_dims=hidden_dims,
                                                 batch_shape=(), regression_dim=0, control_dim=0,
                                                 number_of_objects=number_of_objects)
    for k, v in (("T", T), ("S", S), ("n_obs", n_obs), ("obs_dim", obs_dim), ("number_of_objects", number_of_objects),
                 ("latent_iters", latent_iters)):
        b.put(k, v)
    b.put("role_dims", np.array(role_dims, dtype=np.int64))
    b.put("hidden_dims", np.array(hidden_dims, dtype=np.int64))
    b.put("A_mask", m.A.mask)
    b.put("B_X_mask", m.B.X_mask)
    b.put("role_mask", m.obs_model.transition_mask)
    snap_dmbd(b, m, "init_")
    b.put("init_B_invU_0", m.B.invU.invU_0)
    b.put("init_B_logdet_invU_0", m.B.invU.logdet_invU_0)
    # a few noisy oscillators seen by the observables
    tt = torch.arange(T, dtype=torch.float64).reshape(T, 1, 1, 1)
    ph = torch.rand(1, S, n_obs, obs_dim, generator=gen) * 6
    y = torch.sin(0.25 * tt + ph) + 0.5 * torch.cos(0.11 * tt * (1 + torch.arange(n_obs).reshape(1, 1, n_obs, 1)) + ph)
    y = y + 0.1 * torch.randn(y.shape, generator=gen)
    b.put("y", y)
    for it in range(1, iters + 1):
        with contextlib.redirect_stdout(io.StringIO()):
            m.update(y, None, None, iters=1, latent_iters=latent_iters, lr=1.0)
        pre = f"it{it}_"
        b.put(pre + "p", m.obs_model.p)
        b.put(pre + "NA", m.NA)
        b.put(pre + "SEzz", m.SEzz)
        b.put(pre + "SEz0", m.SEz0)
        b.put(pre + "px_mu", m.px.mu)
        b.put(pre + "px_Sigma", m.px.Sigma)
        b.put(pre + "logZ", m.logZ)
        b.put(pre + "ELBO", m.ELBO_last)
        snap_dmbd(b, m, pre)
    b.put("assignment_pr", m.assignment_pr())
    b.put("particular_assignment_pr", m.particular_assignment_pr())


def gen_dmbd():
    b = Book()
    gen = torch.Generator().manual_seed(1010)
    torch.manual_seed(20)
    dmbd_case(b, "dmbd_lorenz_like", 24, 3, 4, 2, (1, 2, 1), (2, 2, 2), gen, iters=3)
    dmbd_case(b, "dmbd_latent2", 16, 2, 3, 2, (2, 1, 1), (2, 1, 2), gen, iters=2, latent_iters=2)
    dmbd_case(b, "dmbd_two_objects", 14, 2, 5, 2, (1, 1, 1), (2, 1, 1), gen, number_of_objects=2, iters=2)
    b.save("dmbd")


GROUPS["dmbd"] = gen_dmbd

# ---------------------------------------------------------------------- mixture of LDS
def mixlds_case(b, name, K, T, S, obs_shape, hidden, gen, control=0, regression=0, iters=3, lr=1.0):
    import contextlib
    import io

    import models  # reference
    b.begin(name)
    with contextlib.redirect_stdout(io.StringIO()):
        m = models.MixtureofLinearDynamicalSystems(K, obs_shape, hidden, control, regression)
    for k, v in (("K", K), ("T", T), ("S", S), ("hidden", hidden), ("control", control), ("regression", regression),
                 ("lr", lr)):
        b.put(k, v)
    b.put("obs_shape", np.array(obs_shape, dtype=np.int64))
    snap_lds_state_mng(b, m.lds, "init_")
    b.put("init_pi_alpha", m.pi.alpha)
    # two families of series (different oscillation frequencies) so that the systems specialise
    tt = torch.arange(T, dtype=torch.float64).reshape(T, 1, 1)
    fam = (torch.arange(S) % 2).reshape(1, S, 1).to(torch.float64)
    lat = torch.cat([torch.sin((0.15 + 0.25 * fam) * tt * (k + 1) + torch.rand(1, S, 1, generator=gen) * 6)
                     for k in range(hidden)], -1)
    W = torch.randn(obs_shape + (hidden,), generator=gen)
    y = (W @ lat.reshape((T, S) + (1,) * (len(obs_shape) - 1) + (hidden, 1))).squeeze(-1)
    y = y + 0.1 * torch.randn(y.shape, generator=gen)
    u = torch.randn(T, S, control, generator=gen) if control else None
    r = torch.randn((T, S) + obs_shape[:-1] + (regression,), generator=gen) if regression else None
    b.put("y", y)
    b.put("u", u)
    b.put("r", r)
    for it in range(1, iters + 1):
        with contextlib.redirect_stdout(io.StringIO()):
            m.update(y, u, r, iters=1, lr=lr)
        pre = f"it{it}_"
        b.put(pre + "p", m.p)
        b.put(pre + "NA", m.NA)
        b.put(pre + "logZ", m.logZ)
        b.put(pre + "lds_logZ", m.lds.logZ)
        b.put(pre + "pi_alpha", m.pi.alpha)
        snap_lds_state_mng(b, m.lds, pre)
    b.put("KLqprior", m.KLqprior())
    b.put("assignment", m.assignment())


def gen_mixlds():
    b = Book()
    gen = torch.Generator().manual_seed(4242)
    torch.manual_seed(23)
    mixlds_case(b, "mix3_h3_o5", 3, 20, 6, (5,), 3, gen, iters=3)
    mixlds_case(b, "mix2_h2_o4_ctrl_reg", 2, 16, 5, (4,), 2, gen, control=2, regression=1, iters=2, lr=0.6)
    b.save("mixlds")


GROUPS["mixlds"] = gen_mixlds

# ---------------------------------------------------------------------- mixture of linear transforms
def mixlt_case(b, name, n, p, dim, N, gen, pad_X=True, kind='Wishart', iters=3, lr=1.0):
    import contextlib
    import io

    import transforms  # reference
    from dists import MultivariateNormal_vector_format as VF
    b.begin(name)
    with contextlib.redirect_stdout(io.StringIO()):
        m = transforms.MixtureofLinearTransforms(n, p, dim, pad_X=pad_X, type=kind)
    for k, v in (("n", n), ("p", p), ("dim", dim), ("N", N), ("pad_X", int(pad_X)), ("lr", lr),
                 ("gamma", int(kind == 'Gamma'))):
        b.put(k, v)
    b.put("init_W_mu", m.W.mu)
    b.put("init_pi_alpha", m.pi.alpha)
    if kind == 'Gamma':
        b.put("init_W_alpha", m.W.invU.gamma.alpha)
        b.put("init_W_beta", m.W.invU.gamma.beta)
    # piecewise-linear data: each sample follows one of `dim` random linear maps
    X = torch.randn(N, p, 1, generator=gen)
    Ws = torch.randn(dim, n, p, generator=gen)
    z = torch.randint(dim, (N,), generator=gen)
    Y = Ws[z] @ X + 0.5 * torch.randn(dim, n, 1, generator=gen)[z] + 0.1 * torch.randn(N, n, 1, generator=gen)
    b.put("X", X)
    b.put("Y", Y)
    for it in range(1, iters + 1):
        m.raw_update(X, Y, iters=1, lr=lr)
        pre = f"it{it}_"
        b.put(pre + "p", m.p)
        b.put(pre + "logZ", m.logZ)
        b.put(pre + "ELBO", m.ELBO_last)
        b.put(pre + "pi_alpha", m.pi.alpha)
        b.put(pre + "W_mu", m.W.mu)
        b.put(pre + "W_invV", m.W.invV)
        if kind == 'Gamma':
            b.put(pre + "W_alpha", m.W.invU.gamma.alpha)
            b.put(pre + "W_beta", m.W.invU.gamma.beta)
        else:
            b.put(pre + "W_invU", m.W.invU.invU)
            b.put(pre + "W_nu", m.W.invU.nu)
    b.put("KLqprior", m.KLqprior())
    pY, pr = m.predict(X[:7])
    b.put("pred_mu", pY.mean())
    b.put("pred_Sigma", pY.ESigma())
    b.put("pred_p", pr)
    b.put("EinvUX", m.EinvUX())
    b.put("EXTinvUX", m.EXTinvUX())
    b.put("EinvSigma", m.EinvSigma())
    b.put("ElogdetinvSigma", m.ElogdetinvSigma())
    # distributions in: update(pX, pY)
    A = torch.randn(N, p, p + 2, generator=gen)
    SigX = A @ A.transpose(-2, -1) / (p + 2) * 0.05
    pX = VF(mu=X, Sigma=SigX)
    Bm = torch.randn(N, n, n + 2, generator=gen)
    pYd = VF(mu=Y, Sigma=Bm @ Bm.transpose(-2, -1) / (n + 2) * 0.05)
    b.put("upd_SigX", SigX)
    b.put("upd_SigY", pYd.Sigma)
    m.update(pX, pYd, iters=1, lr=lr)
    b.put("upd_p", m.p)
    b.put("upd_logZ", m.logZ)
    b.put("upd_W_mu", m.W.mu)
    b.put("upd_pi_alpha", m.pi.alpha)
    b.put("upd_ELBO", m.ELBO_last)


def gen_mixlt():
    b = Book()
    gen = torch.Generator().manual_seed(777)
    torch.manual_seed(31)
    mixlt_case(b, "mixlt_w_n3_p4_k3", 3, 4, 3, 60, gen)
    mixlt_case(b, "mixlt_w_nopad_lr", 2, 3, 4, 50, gen, pad_X=False, iters=2, lr=0.7)
    mixlt_case(b, "mixlt_g_n3_p2_k2", 3, 2, 2, 40, gen, kind='Gamma', iters=2)
    b.save("mixlt")


GROUPS["mixlt"] = gen_mixlt

# ---------------------------------------------------------------------- logistic gate + gated mixture
def snap_ard(b, q, pre):
    for f in ("mu", "invSigma", "invSigmamu", "Sigma", "logdetinvSigma"):
        b.put(pre + f, getattr(q, f))
    b.put(pre + "alpha", q.alpha.alpha)
    b.put(pre + "beta", q.alpha.beta)


def quiet():
    import contextlib
    import io
    return contextlib.redirect_stdout(io.StringIO())


def mnlr_case(b, name, ncls, p, N, gen, pad_X=True, lr=1.0):
    import transforms  # reference
    from dists import MultivariateNormal_vector_format as VF
    b.begin(name)
    with quiet():
        m = transforms.MultiNomialLogisticRegression(ncls, p, pad_X=pad_X)
    for k, v in (("ncls", ncls), ("p", p), ("N", N), ("pad_X", int(pad_X)), ("lr", lr)):
        b.put(k, v)
    snap_ard(b, m.beta, "init_")
    X = torch.randn(N, p, generator=gen)
    Wt = torch.randn(ncls, p, generator=gen) * 1.5
    z = torch.distributions.Categorical(logits=X @ Wt.T).sample() if False else (X @ Wt.T + torch.randn(N, ncls, generator=gen)).argmax(-1)
    Y = torch.nn.functional.one_hot(z, ncls).to(torch.float64)
    wts = torch.rand(N, generator=gen) + 0.5
    b.put("X", X)
    b.put("Y", Y)
    b.put("w", wts)
    with quiet():
        m.raw_update(X, Y, iters=2, lr=lr)
    snap_ard(b, m.beta, "r1_")
    with quiet():
        m.raw_update(X, Y, iters=3, p=wts, lr=lr)
    snap_ard(b, m.beta, "r2_")
    b.put("KLqprior", m.KLqprior())
    b.put("Elog_like", m.Elog_like(X, Y))
    b.put("log_predict", m.log_predict(X[:9]))
    b.put("predict", m.predict(X[:9]))
    b.put("log_predict_1", m.log_predict_1(X[:9]))
    b.put("log_predict_2", m.log_predict_2(X[:9]))
    if pad_X:
        b.put("weights", m.weights())
    A = torch.randn(N, p, p + 2, generator=gen)
    SigX = A @ A.transpose(-2, -1) / (p + 2) * 0.1
    pX = VF(mu=X.unsqueeze(-1), Sigma=SigX)
    b.put("SigX", SigX)
    b.put("ELpXpY", m.Elog_like_given_pX_pY(pX, Y))
    b.put("log_forward", m.log_forward(VF(mu=X[:9].unsqueeze(-1), Sigma=SigX[:9])))
    pYs = torch.softmax(torch.randn(6, ncls, generator=gen), -1)
    b.put("bw_pY", pYs)
    with quiet():
        px, Res = m.backward(pYs)
    b.put("bw_invSigma", px.invSigma)
    b.put("bw_invSigmamu", px.invSigmamu)
    b.put("bw_mu", px.mu)
    b.put("bw_Res", Res)
    with quiet():
        m.update(pX, Y, iters=2, lr=lr)
    snap_ard(b, m.beta, "u1_")


def dmix_case(b, name, n, p, mix, N, gen, kind='Wishart', iters=3, lr=1.0):
    import transforms  # reference
    from dists import MultivariateNormal_vector_format as VF
    b.begin(name)
    with quiet():
        m = transforms.dMixtureofLinearTransforms(n, p, mix, pad_X=True, type=kind)
    for k, v in (("n", n), ("p", p), ("mix", mix), ("N", N), ("lr", lr), ("gamma", int(kind == 'Gamma'))):
        b.put(k, v)
    b.put("init_A_mu", m.A.mu)
    if kind == 'Gamma':
        b.put("init_A_alpha", m.A.invU.gamma.alpha)
        b.put("init_A_beta", m.A.invU.gamma.beta)
    snap_ard(b, m.pi.beta, "init_pi_")
    # piecewise-linear map: the active piece depends on the input (what the gate has to learn)
    X = torch.randn(N, p, generator=gen)
    z = (X[:, 0] > 0.3).long() + (X[:, 1] > 0.0).long() * (mix > 2)
    Ws = torch.randn(mix, n, p, generator=gen)
    Y = (Ws[z] @ X.unsqueeze(-1)).squeeze(-1) + 0.5 * torch.randn(mix, n, generator=gen)[z] + 0.1 * torch.randn(N, n, generator=gen)
    b.put("X", X)
    b.put("Y", Y)
    for it in range(1, iters + 1):
        with quiet():
            m.raw_update(X, Y, iters=1, lr=lr)
        pre = f"it{it}_"
        b.put(pre + "A_mu", m.A.mu)
        b.put(pre + "A_invV", m.A.invV)
        snap_ard(b, m.pi.beta, pre + "pi_")
    b.put("KLqprior", m.KLqprior())
    b.put("Elog_like", m.Elog_like(X, Y))
    pY, pr = m.predict(X[:7])
    b.put("pred_mu", pY.mean())
    b.put("pred_Sigma", pY.ESigma())
    b.put("pred_p", pr)
    A = torch.randn(N, p, p + 2, generator=gen)
    SigX = A @ A.transpose(-2, -1) / (p + 2) * 0.05
    Bm = torch.randn(N, n, n + 2, generator=gen)
    SigY = Bm @ Bm.transpose(-2, -1) / (n + 2) * 0.05
    pX, pYd = VF(mu=X.unsqueeze(-1), Sigma=SigX), VF(mu=Y.unsqueeze(-1), Sigma=SigY)
    b.put("SigX", SigX)
    b.put("SigY", SigY)
    b.put("ELpXpY", m.Elog_like_given_pX_pY(pX, pYd))
    try:  # the reference's forward indexes the expert message as a tuple: fails for Gamma experts (no residual there)
        with quiet():
            fw = m.forward(VF(mu=X[:5].unsqueeze(-1), Sigma=SigX[:5]))
        b.put("fw_mu", fw.mean())
        b.put("fw_Sigma", fw.ESigma())
    except TypeError:
        pass
    with quiet():
        px, logZ, pp = m.postdict(Y[:5])
    b.put("post_invSigma", px.invSigma)
    b.put("post_invSigmamu", px.invSigmamu)
    b.put("post_logZ", logZ)
    b.put("post_p", pp)
    with quiet():
        m.update(pX, pYd, iters=1, lr=lr)
    b.put("upd_logZ", m.logZ)
    b.put("upd_NA", m.NA)
    b.put("upd_ELBO", m.ELBO_last)
    b.put("upd_A_mu", m.A.mu)
    snap_ard(b, m.pi.beta, "upd_pi_")


def gen_dmix():
    b = Book()
    gen = torch.Generator().manual_seed(1234)
    torch.manual_seed(41)
    mnlr_case(b, "mnlr_c4_p3", 4, 3, 80, gen)
    mnlr_case(b, "mnlr_c3_p9_lr", 3, 9, 120, gen, lr=0.7)
    dmix_case(b, "dmix_w_n2_p3_k3", 2, 3, 3, 70, gen)
    dmix_case(b, "dmix_g_n3_p2_k2", 3, 2, 2, 60, gen, kind='Gamma', iters=2, lr=0.8)
    b.save("dmix")


GROUPS["dmix"] = gen_dmix

# ---------------------------------------------------------------------- config 3: MNW messages, one precision per message
def mnwmsg_case(b, name, n, p, batch, pad_X, gen, N=64, Nfit=96):
    """BASELINE configs[2] shape (forward / backward with a precision PER MESSAGE) from a fitted transform: the stored
    post-update state is the test's input, so that a float32 run of the product starts from the same (rounded) state."""
    b.begin(name)
    m = transforms.MatrixNormalWishart(event_shape=(n, p), batch_shape=batch, pad_X=pad_X)
    nb = len(batch)
    W = torch.randn(batch + (n, p), generator=gen) / p ** 0.5
    X = torch.randn((Nfit,) + (1,) * nb + (p, 1), generator=gen)
    Y = W @ X + 0.3 * torch.randn((Nfit,) + batch + (n, 1), generator=gen)
    m.raw_update(X.expand((Nfit,) + batch + (p, 1)), Y, lr=1.0)
    b.put("n", n)
    b.put("p", p)
    b.put("pad_X", int(pad_X))
    b.put("batch_shape", np.array(batch, dtype=np.int64))
    snap_mnw(b, m, "state_")
    # message inputs are float32-representable (stored as float32: half the fixture, and a float32 run reads them exactly)
    f32 = lambda t: t.float().double()
    px_P = f32(rand_spd((N,) + (1,) * nb, p, gen))
    px_eta = f32(torch.randn((N,) + (1,) * nb + (p, 1), generator=gen))
    b.put("fw_in_invSigma", px_P.float())
    b.put("fw_in_invSigmamu", px_eta.float())
    pYm, Res = m.forward(dists.MultivariateNormal_vector_format(invSigma=px_P.clone(), invSigmamu=px_eta.clone()))
    b.put("fw_mu", pYm.mu)
    b.put("fw_Sigma", pYm.Sigma)
    b.put("fw_Res", Res)
    py_P = f32(rand_spd((N,) + (1,) * nb, n, gen))
    py_eta = f32(torch.randn((N,) + (1,) * nb + (n, 1), generator=gen))
    b.put("bw_in_invSigma", py_P.float())
    b.put("bw_in_invSigmamu", py_eta.float())
    Res_in = f32(torch.randn((N,) + batch, generator=gen))
    b.put("bw_in_Res", Res_in.float())
    pXb, Res = m.backward(dists.MultivariateNormal_vector_format(invSigma=py_P.clone(), invSigmamu=py_eta.clone()), Res=Res_in)
    b.put("bw_invSigma", pXb.invSigma)
    b.put("bw_invSigmamu", pXb.invSigmamu)
    b.put("bw_Res", Res)


def gen_mnwmsg():
    b = Book()
    gen = torch.Generator().manual_seed(3232)
    torch.manual_seed(32)
    mnwmsg_case(b, "msg_32x32", 32, 32, (), False, gen)            # configs[2] itself
    mnwmsg_case(b, "msg_32x31_pad", 32, 31, (), True, gen, N=16)         # bias column: internal p = 32, message dim 31
    mnwmsg_case(b, "msg_32x32_pad", 32, 32, (), True, gen, N=8)   # internal p = 33, message dim 32
    mnwmsg_case(b, "msg_16x16_b3", 16, 16, (3,), False, gen, N=24)
    mnwmsg_case(b, "msg_24x32", 24, 32, (), False, gen, N=12)      # n != p, padded output rows
    mnwmsg_case(b, "msg_32x20", 32, 20, (), False, gen, N=12)
    mnwmsg_case(b, "msg_8x40", 8, 40, (), False, gen, N=6)         # p = 40: beyond the fused kernel (composed K1 + GEMM route)
    b.save("mnwmsg")


GROUPS["mnwmsg"] = gen_mnwmsg

# ---------------------------------------------------------------------- HMM forward-backward (models/HMM.py:72-105)
def hmm_case(b, name, K, T, lead, batch, ptemp, gen, mask=None, scale=2.0):
    import models  # reference
    b.begin(name)
    obs = dists.NormalInverseWishart(event_shape=(2,), batch_shape=batch + (K,))  # only its batch shape is used here
    h = models.HMM(obs, transition_mask=mask, ptemp=ptemp)
    h.transition.alpha = h.transition.alpha_0 + 3.0 * torch.rand(batch + (K, K), generator=gen) * (1.0 if mask is None else mask)
    h.initial.alpha = h.initial.alpha_0 + 2.0 * torch.rand(batch + (K,), generator=gen)
    logits = scale * torch.randn((T,) + lead + batch + (K,), generator=gen)
    for k, v in (("K", K), ("T", T), ("ptemp", ptemp)):
        b.put(k, v)
    b.put("batch_shape", np.array(batch, dtype=np.int64))
    b.put("logits", logits)
    b.put("trans", h.transition.loggeomean())
    b.put("init", h.initial.loggeomean())
    if mask is not None:
        b.put("mask", mask)
    p, SEzz, SEz0, logZ = h.forward_backward_logits(logits.clone())
    b.put("p", p)
    b.put("SEzz", SEzz)
    b.put("SEz0", SEz0)
    b.put("logZ", logZ)


def gen_hmm():
    import contextlib
    import io

    import models  # reference
    b = Book()
    gen = torch.Generator().manual_seed(2525)
    torch.manual_seed(25)
    with contextlib.redirect_stdout(io.StringIO()):  # the role chain of the flocking DMBD: 25 roles, masked transitions
        d = models.DynamicMarkovBlanketDiscovery(obs_shape=(12, 4), role_dims=(1, 2, 2), hidden_dims=(4, 4, 4),
                                                 regression_dim=-1, control_dim=0, number_of_objects=6)
    role_mask = d.obs_model.transition_mask
    hmm_case(b, "hmm_k25_roles", 25, 30, (6,), (), 1.0, gen, mask=role_mask)
    hmm_case(b, "hmm_k25_roles_ptemp", 25, 12, (2, 3), (), 3.0, gen, mask=role_mask, scale=6.0)
    hmm_case(b, "hmm_k4", 4, 40, (5, 3), (), 1.0, gen)
    hmm_case(b, "hmm_k3_b2", 3, 17, (4,), (2,), 2.0, gen)
    hmm_case(b, "hmm_k9_T1", 9, 1, (3,), (), 1.0, gen)
    hmm_case(b, "hmm_k2_T2", 2, 2, (1,), (), 1.0, gen)
    b.save("hmm")


GROUPS["hmm"] = gen_hmm

# ---------------------------------------------------------------------- config 5: DMBD at the Flocking_example hyper-parameters
def gen_dmbd_flock():
    """examples/Flocking_example.py:38 exactly: role_dims=(1,2,2), hidden_dims=(4,4,4), number_of_objects=6,
    regression_dim=-1, control_dim=0 (hidden 52, 25 roles) on the repo's boids generator (the reference's data file is
    absent) at T=20, 2 runs, 12 birds.  Only the quantities the test compares are stored (fixture < 1 MB)."""
    import contextlib
    import importlib.util
    import io

    import models  # reference
    spec = importlib.util.spec_from_file_location("vbmp_synth", os.path.join(os.path.dirname(os.path.abspath(__file__)), "synth.py"))
    synth = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(synth)
    b = Book()
    gen = torch.Generator().manual_seed(66)
    torch.manual_seed(6)
    T, S, n_obs, obs_dim = 20, 2, 12, 4
    b.begin("dmbd_flocking")
    with contextlib.redirect_stdout(io.StringIO()):
        m = models.DynamicMarkovBlanketDiscovery(obs_shape=(n_obs, obs_dim), role_dims=(1, 2, 2), hidden_dims=(4, 4, 4),
                                                 regression_dim=-1, control_dim=0, number_of_objects=6, unique_obs=False)
    y = synth.boids(T, S, n_obs, gen, device="cpu")
    b.put("y", y)
    b.put("A_mask", m.A.mask)
    b.put("B_X_mask", m.B.X_mask)
    b.put("role_mask", m.obs_model.transition_mask)
    b.put("init_x0_mu", m.x0.mu)
    b.put("init_A_mu", m.A.mu)
    b.put("init_A_alpha", m.A.invU.gamma.alpha)
    b.put("init_A_beta", m.A.invU.gamma.beta)
    b.put("init_B_mu", m.B.mu)
    b.put("init_B_invU_0", m.B.invU.invU_0)
    b.put("init_trans_alpha", m.obs_model.transition.alpha)
    b.put("init_init_alpha", m.obs_model.initial.alpha)
    for it in (1, 2):
        with contextlib.redirect_stdout(io.StringIO()):
            m.update(y, None, None, iters=1, latent_iters=1, lr=1.0)
        pre = f"it{it}_"
        b.put(pre + "p", m.obs_model.p)
        b.put(pre + "NA", m.NA)
        b.put(pre + "SEzz", m.SEzz)
        b.put(pre + "SEz0", m.SEz0)
        b.put(pre + "logZ", m.logZ)
        b.put(pre + "ELBO", m.ELBO_last)
        b.put(pre + "px_mu", m.px.mu)
        b.put(pre + "A_mu", m.A.mu)
        b.put(pre + "A_beta", m.A.invU.gamma.beta)
        b.put(pre + "B_mu", m.B.mu)
        b.put(pre + "B_invU_invU", m.B.invU.invU)
        b.put(pre + "x0_mu", m.x0.mu)
        b.put(pre + "trans_alpha", m.obs_model.transition.alpha)
    b.save("dmbd_flock")


GROUPS["dmbd_flock"] = gen_dmbd_flock


if __name__ == "__main__":
    want = sys.argv[1:] or list(GROUPS)
    for g in want:
        GROUPS[g]()
