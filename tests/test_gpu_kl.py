"""K15: the KL(q || prior) terms of the conjugate families from one launch each == the terms composed from torch operations the way
the reference writes them (dists/Dirichlet.py:73-86, Gamma.py:66-72, Wishart.py:85-95, NormalInverseWishart.py:134-141,
transforms/MatrixNormalWishart.py:206-215, MatrixNormalGamma.py:203-214); the composed forms themselves are pinned to the
reference by the golden fixtures of the models' ELBOs (tests/test_gpu_gmm.py, test_gpu_lds.py, test_gpu_dmbd.py, ...)."""
import pytest
import torch

from tests.helpers import assert_close

pytestmark = pytest.mark.gpu
DT = [(torch.float64, 1e-12), (torch.float32, 2e-5)]


@pytest.mark.parametrize("dtype, tol", DT)
@pytest.mark.parametrize("event, batch, to_event", [((5,), (), 0), ((7,), (3,), 0), ((4,), (6, 3), 1), ((130,), (2,), 0)])
def test_dirichlet_kl(event, batch, to_event, dtype, tol):
    from pyvbmp_amd.dists.Dirichlet import Dirichlet
    torch.manual_seed(0)
    d = Dirichlet(event, batch, device="cuda", dtype=dtype)
    d.ss_update(torch.rand(batch + event, device="cuda", dtype=dtype) * 20)
    d.to_event(to_event)
    got, want = d.KLqprior(), d._KLqprior_composed()
    assert got.shape == want.shape
    assert_close(got, want, tol=tol)


def test_dirichlet_kl_structural_zeros():
    """a masked transition matrix: zero concentrations in prior and posterior count as 0 (ref KL_lgamma / KL_digamma)"""
    from pyvbmp_amd.dists.Dirichlet import Dirichlet
    torch.manual_seed(1)
    mask = (torch.rand(6, 6, device="cuda") > 0.4) | torch.eye(6, device="cuda", dtype=torch.bool)
    d = Dirichlet((6,), (6,), prior_parms={'alpha': 0.5 * mask.double()}, device="cuda", dtype=torch.float64)
    d.alpha = d.alpha * mask
    d.ss_update(torch.rand(6, 6, device="cuda", dtype=torch.float64) * 9 * mask)
    assert_close(d.KLqprior(), d._KLqprior_composed(), tol=1e-12)
    assert torch.isfinite(d.KLqprior()).all()


@pytest.mark.parametrize("dtype, tol", DT)
@pytest.mark.parametrize("event, batch", [((4,), ()), ((6,), (5,)), ((3, 70), (2,))])
def test_gamma_kl(event, batch, dtype, tol):
    from pyvbmp_amd.dists.Gamma import Gamma
    torch.manual_seed(2)
    g = Gamma(event, batch, device="cuda", dtype=dtype)
    g.ss_update(torch.rand(batch + event, device="cuda", dtype=dtype) * 30, torch.rand(batch + event, device="cuda", dtype=dtype) * 10)
    got, want = g.KLqprior(), g._KLqprior_composed()
    assert got.shape == want.shape
    assert_close(got, want, tol=tol)


def _spd(shape, dtype):
    A = torch.randn(shape, device="cuda", dtype=dtype)
    return A @ A.mT + shape[-1] * torch.eye(shape[-1], device="cuda", dtype=dtype)


@pytest.mark.parametrize("dtype, tol", DT)
@pytest.mark.parametrize("event, batch", [((3, 3), ()), ((6, 6), (4,)), ((5, 2, 2), (3,)), ((40, 40), (2,))])
def test_wishart_kl(event, batch, dtype, tol):
    from pyvbmp_amd.dists.Wishart import Wishart
    torch.manual_seed(3)
    w = Wishart(event, batch, device="cuda", dtype=dtype)
    lead = batch + event[:-2]
    w.ss_update(_spd(lead + event[-2:], dtype) * 7, torch.full(lead, 11.0, device="cuda", dtype=dtype) + torch.rand(lead, device="cuda", dtype=dtype))
    got, want = w.KLqprior(), w._KLqprior_composed()
    assert got.shape == want.shape
    assert_close(got, want, tol=tol)


@pytest.mark.parametrize("dtype, tol", DT)
@pytest.mark.parametrize("event, batch", [((2,), (4,)), ((16,), (3, 2)), ((3, 5), (2,)), ((52,), ())])
def test_niw_kl(event, batch, dtype, tol):
    from pyvbmp_amd.dists.NormalInverseWishart import NormalInverseWishart
    torch.manual_seed(4)
    m = NormalInverseWishart(event, batch, device="cuda", dtype=dtype)
    X = torch.randn((64,) + batch + event, device="cuda", dtype=dtype) * 2 + 1
    m.raw_update(X, lr=0.8)
    got, want = m.KLqprior(), m._KLqprior_composed()
    assert got.shape == want.shape
    assert_close(got, want, tol=tol)


@pytest.mark.parametrize("dtype, tol", DT)
@pytest.mark.parametrize("cls", ["wishart", "gamma"])
@pytest.mark.parametrize("event, batch, pad, xmask", [((3, 2), (), False, False), ((4, 5), (6,), True, False),
                                                      ((4, 52), (25,), True, True), ((3, 6), (4,), False, True), ((2, 3, 4), (5,), False, False),
                                                      ((52, 52), (), True, False), ((64, 64), (2,), False, False)])
def test_matrix_normal_kl(event, batch, pad, xmask, cls, dtype, tol):
    from pyvbmp_amd.transforms.MatrixNormalGamma import MatrixNormalGamma
    from pyvbmp_amd.transforms.MatrixNormalWishart import MatrixNormalWishart
    torch.manual_seed(5)
    C = MatrixNormalWishart if cls == "wishart" else MatrixNormalGamma
    n, px = event[-2], event[-1]
    X_mask = None
    if xmask:  # one mask per batch element, as DynamicMarkovBlanketDiscovery builds for its emission model
        X_mask = (torch.rand(batch + (1, px), device="cuda") > 0.3)
    m = C(event, batch, pad_X=pad, X_mask=X_mask, device="cuda", dtype=dtype)
    lead = batch + tuple(event[:-2])
    X = torch.randn((80,) + (1,) * len(lead) + (px, 1), device="cuda", dtype=dtype)
    Y = torch.randn((80,) + lead + (n, 1), device="cuda", dtype=dtype)
    m.raw_update(X, Y, lr=0.9)
    got, want = m.KLqprior(), m._KLqprior_composed()
    assert got.shape == want.shape
    assert_close(got, want, tol=tol)
    if cls == "gamma":
        m.uniform_precision = True
        assert_close(m.KLqprior(), m._KLqprior_composed(), tol=tol)
