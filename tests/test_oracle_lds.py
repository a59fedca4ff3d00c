"""Pin the LDS oracle (information filter / smoother E-step + M-step glue) to the reference's golden
outputs (tests/golden/lds.npz, latent_noise='shared').  CPU only."""
import pytest

from oracle import lds as olds
from oracle import mnw as omnw
from oracle import niw as oniw
from tests.helpers import assert_close

LDS_CASES = ["lds_h6_o6", "lds_h6_o6_lr", "lds_h3_o5_ctrl_reg", "lds_h2_o32", "lds_h4_o5_batch2", "lds_h8_o4"]


def lds_oracle_states(c):
    h = int(c["hidden"])
    obs_shape = tuple(int(v) for v in c["obs_shape"])
    batch = tuple(int(v) for v in c["batch_shape"])
    cd, rd = int(c["control"]) + 1, int(c["regression"]) + 1
    offset = (1,) * (len(obs_shape) - 1)
    x0 = oniw.niw_new(offset + (h,), batch, mu_init=c["init_x0_mu"])
    A = omnw.mnw_new(offset + (h, h + cd), batch, mu_init=c["init_A_mu"])
    obs = omnw.mnw_new(obs_shape + (h + rd,), batch, mu_init=c["init_obs_mu"])
    return x0, A, obs, h, obs_shape, batch, cd, rd


def n_iters(c):
    return max(int(k[2]) for k in c if k.startswith("it") and k[2].isdigit())


@pytest.mark.parametrize("case", LDS_CASES)
def test_lds_oracle_golden(golden, case):
    c = golden("lds")[case]
    x0, A, obs, h, obs_shape, batch, cd, rd = lds_oracle_states(c)
    nx = len(obs_shape) - 1
    lr = float(c["lr"])
    y, u, r = olds.reshape_inputs(c["y"], c.get("u"), c.get("r"), obs_shape, cd, rd, batch, len(batch) > 0)
    for it in range(1, n_iters(c) + 1):
        pre = f"it{it}_"
        lp = olds.latent_parms(A, h)
        sm = olds.smoother(lp, x0, h, y, u, r, obs, nx)
        for f in ("mu", "Sigma", "invSigma", "invSigmamu"):
            assert_close(sm[f], c[pre + "px_" + f], 1e-9, what=pre + f)
        st = olds.latent_stats(sm, y, u, r, obs_shape, cd, rd, batch, nx)
        for f in ("SE_x_x", "SE_x0_x0", "SE_x0", "SE_y_xr", "SE_y_y", "SE_xpu_xpu", "SE_x_xpu", "SE_xr_xr", "T", "N", "logZ"):
            assert_close(st[f], c[pre + f], 1e-9, what=pre + f)
        lz = st["logZ"]
        while lz.ndim > len(batch):
            lz = lz.sum(0)
        kl = oniw.niw_kl(x0) + omnw.mnw_kl(A)
        for _ in range(nx):
            kl = kl.squeeze(-1)
        assert_close(lz - (kl + omnw.mnw_kl(obs)), c[pre + "ELBO"], 1e-9, what=pre + "ELBO")
        st = olds.reduce_stats(st, len(batch), nx)
        x0 = oniw.niw_ss_update(x0, st["SE_x0_x0"], st["SE_x0"].squeeze(-1), st["N"], lr)
        A = omnw.mnw_ss_update(A, st["SE_xpu_xpu"], st["SE_x_xpu"], st["SE_x_x"], st["T"], lr)
        obs = omnw.mnw_ss_update(obs, st["SE_xr_xr"], st["SE_y_xr"], st["SE_y_y"], st["T"], lr)
        assert_close(x0["mu"], c[pre + "x0_mu"], 1e-9)
        assert_close(x0["W"]["invU"], c[pre + "x0_invU"], 1e-9)
        assert_close(A["mu"], c[pre + "A_mu"], 1e-9)
        assert_close(A["invV"], c[pre + "A_invV"], 1e-9)
        assert_close(A["W"]["invU"], c[pre + "A_invU_invU"], 1e-9)
        assert_close(obs["mu"], c[pre + "obs_mu"], 1e-9)
        assert_close(obs["W"]["U"], c[pre + "obs_invU_U"], 1e-9)
        lp2 = olds.latent_parms(A, h)
        for f in ("invQ", "ATQA_x_x", "invATQA_x_x", "logdetATQA_x_x", "ATQA_x_u", "ATQA_u_u", "QA_xp_x", "QA_xp_u"):
            assert_close(lp2[f], c[pre + f], 1e-9, what=pre + f)
