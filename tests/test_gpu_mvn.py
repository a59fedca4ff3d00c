"""GPU parity of MultivariateNormal, MultivariateNormal_vector_format and matrix_utils (all K1-backed)
against golden fixtures captured from the reference."""
import pytest
import torch

from tests.helpers import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _cls(vf):
    from pyvbmp_amd.dists import MultivariateNormal, MultivariateNormal_vector_format
    return MultivariateNormal_vector_format if vf else MultivariateNormal


@pytest.mark.parametrize("vf", [False, True])
def test_mvn_from_moments_golden(golden, vf):
    c = golden("mvn")["vf_from_moments" if vf else "mvn_from_moments"]
    q = _cls(vf)(mu=c["mu"].to(DEV), Sigma=c["Sigma"].to(DEV))
    fields = ["EinvSigma", "EinvSigmamu", "ElogdetinvSigma", "EXXT", "EXTX", "mean", "ESigma"] + (["Res"] if vf else ["EX"])
    for f in fields:
        assert_close(getattr(q, f)(), c[f], what=f)
    assert_close(q.Elog_like(c["X"].to(DEV)), c["Elog_like"], what="Elog_like")


@pytest.mark.parametrize("vf", [False, True])
def test_mvn_from_natural_golden(golden, vf):
    c = golden("mvn")["vf_from_natural" if vf else "mvn_from_natural"]
    q = _cls(vf)(invSigma=c["invSigma"].to(DEV), invSigmamu=c["invSigmamu"].to(DEV))
    for f in ["mean", "ESigma", "ElogdetinvSigma", "EXXT", "EXTX"] + (["Res"] if vf else []):
        assert_close(getattr(q, f)(), c[f], what=f)
    if vf:
        q = _cls(vf)(invSigma=c["invSigma"].to(DEV), invSigmamu=c["invSigmamu"].to(DEV))
        q.mean(), q.ESigma()  # the reference had cached moments when it combined
        q.nat_combiner(c["other_invSigma"].to(DEV), c["other_invSigmamu"].to(DEV))
        assert q.mu is None and q.Sigma is None
        assert_close(q.invSigma, c["nat_invSigma"])
        assert_close(q.invSigmamu, c["nat_invSigmamu"])
        assert_close(q.mean(), c["nat_mean"])
        assert_close(q.Res(), c["nat_Res"])
        o = _cls(vf)(invSigma=c["other_invSigma"].to(DEV), invSigmamu=c["other_invSigmamu"].to(DEV))
        q.combiner(o)
        assert_close(q.invSigma, c["comb_invSigma"])
        assert_close(q.invSigmamu, c["comb_invSigmamu"])
        assert_close(q.ESigma(), c["comb_ESigma"])
        u = q.unsqueeze(-3)
        assert list(u.invSigma.shape) == c["unsq_invSigma_shape"].tolist()
        assert list(u.batch_shape) == c["unsq_batch_shape"].tolist()


@pytest.mark.parametrize("vf", [False, True])
def test_mvn_updates_golden(golden, vf):
    c = golden("mvn")["vf_updates" if vf else "mvn_updates"]
    D = 5
    mu0 = torch.zeros((3, D, 1) if vf else (3, D), dtype=torch.float64, device=DEV)
    q = _cls(vf)(mu=mu0, Sigma=torch.eye(D, dtype=torch.float64, device=DEV).expand(3, D, D))
    q.raw_update(c["X"].to(DEV), c["p"].to(DEV))
    assert_close(q.mu, c["p_mu"], what="p mu")
    assert_close(q.Sigma, c["p_Sigma"], what="p Sigma")
    q.raw_update(c["X_full"].to(DEV))
    assert_close(q.mu, c["nop_mu"], what="nop mu")
    assert_close(q.Sigma, c["nop_Sigma"], what="nop Sigma")
    if not vf:
        assert_close(q.EinvSigma(), c["nop_EinvSigma"])
        assert_close(q.EinvSigmamu(), c["nop_EinvSigmamu"])


@pytest.mark.parametrize("case", ["mu_4_3", "mu_6_6", "mu_16_8"])
def test_matrix_utils_golden(golden, case):
    from pyvbmp_amd.utils import matrix_utils as mu_
    c = golden("matrix_utils")[case]
    A, B, C, D = (c[k].to(DEV) for k in "ABCD")
    assert_close(mu_.block_diag_matrix_builder(A, D), c["block_diag"])
    assert_close(mu_.block_matrix_builder(A, B, C, D), c["block_build"])
    for form in ("left", "right", "True"):
        for i, o in enumerate(mu_.block_matrix_inverse(A, B, C, D, block_form=form)):
            assert_close(o, c[f"inv_{form}_{i}"], what=f"{form}[{i}]")
    assert_close(mu_.block_matrix_inverse(A, B, C, D, block_form=False), c["inv_full"])
    assert_close(mu_.block_matrix_inverse(A, B, C, D), c["inv_default"])
    for i, o in enumerate(mu_.block_precision_marginalizer(A, B, C, D)):
        assert_close(o, c[f"marg_{i}"], what=f"marg[{i}]")
    assert_close(mu_.block_matrix_logdet(A, B, C, D), c["logdet"])
    assert_close(mu_.block_matrix_logdet(A, B, C, D, singular="A"), c["logdet_A"])
    assert_close(mu_.block_matrix_logdet(A, B, C, D, singular="D"), c["logdet_D"])
