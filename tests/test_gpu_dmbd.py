"""GPU parity of DynamicMarkovBlanketDiscovery (role HMM + masked LDS on the HIP kernels) against golden
fixtures captured from the reference."""
import pytest
import torch

from tests.helpers import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"
CASES = ["dmbd_lorenz_like", "dmbd_latent2", "dmbd_two_objects"]


def _n_iters(c):
    return max(int(k[2]) for k in c if k.startswith("it") and k[2].isdigit())


@pytest.mark.parametrize("case", CASES)
def test_dmbd_golden(golden, case):
    from pyvbmp_amd.models import DynamicMarkovBlanketDiscovery
    c = golden("dmbd")[case]
    m = DynamicMarkovBlanketDiscovery(obs_shape=(int(c["n_obs"]), int(c["obs_dim"])),
                                      role_dims=tuple(int(v) for v in c["role_dims"]),
                                      hidden_dims=tuple(int(v) for v in c["hidden_dims"]), batch_shape=(), regression_dim=0,
                                      control_dim=0, number_of_objects=int(c["number_of_objects"]), device=DEV,
                                      dtype=torch.float64)
    # the masks are deterministic functions of the dims
    assert torch.equal(m.A.mask.cpu(), c["A_mask"])
    assert torch.equal(m.B.X_mask.cpu(), c["B_X_mask"])
    assert torch.equal(m.obs_model.transition_mask.cpu(), c["role_mask"])
    assert_close(m.B.invU.invU_0, c["init_B_invU_0"])
    assert_close(m.B.invU.logdet_invU_0, c["init_B_logdet_invU_0"])
    # replay the reference's random initial state
    m.x0.mu = c["init_x0_mu"].to(DEV)
    m.A.mu = c["init_A_mu"].to(DEV)
    m.A.invU.gamma.alpha = c["init_A_alpha"].to(DEV)
    m.A.invU.gamma.beta = c["init_A_beta"].to(DEV)
    m.B.mu = c["init_B_mu"].to(DEV)
    m.obs_model.transition.alpha = c["init_trans_alpha"].to(DEV)
    m.obs_model.initial.alpha = c["init_init_alpha"].to(DEV)
    m.set_latent_parms()
    y = c["y"].to(DEV)
    li = int(c["latent_iters"])
    for it in range(1, _n_iters(c) + 1):
        pre = f"it{it}_"
        tol = 1e-8 if it == 1 else 1e-6  # alternating E-steps amplify rounding differences
        m.update(y, None, None, iters=1, latent_iters=li, lr=1.0)
        assert_close(m.obs_model.p, c[pre + "p"], tol, what=pre + "p")
        assert_close(m.NA, c[pre + "NA"], tol, what=pre + "NA")
        assert_close(m.SEzz, c[pre + "SEzz"], tol, what=pre + "SEzz")
        assert_close(m.SEz0, c[pre + "SEz0"], tol, what=pre + "SEz0")
        assert_close(m.px.mu, c[pre + "px_mu"], tol, what=pre + "px mu")
        assert_close(m.px.Sigma, c[pre + "px_Sigma"], tol, what=pre + "px Sigma")
        assert_close(m.logZ, c[pre + "logZ"], tol, what=pre + "logZ")
        assert_close(m.ELBO_last, c[pre + "ELBO"], tol, what=pre + "ELBO")
        assert_close(m.A.mu, c[pre + "A_mu"], tol, what=pre + "A mu")
        assert_close(m.A.invU.gamma.beta, c[pre + "A_beta"], tol)
        assert_close(m.B.mu, c[pre + "B_mu"], tol, what=pre + "B mu")
        assert_close(m.B.invU.invU, c[pre + "B_invU_invU"], tol)
        assert_close(m.obs_model.transition.alpha, c[pre + "trans_alpha"], tol)
        assert_close(m.x0.mu, c[pre + "x0_mu"], tol)
    assert_close(m.assignment_pr(), c["assignment_pr"], 1e-6)
    assert_close(m.particular_assignment_pr(), c["particular_assignment_pr"], 1e-6)
