"""Block-matrix helpers (surface of the reference's utils/matrix_utils.py:2-55).

Every inverse / logdet is the batched K1 kernel (blocks on this path are precision blocks of joint
Gaussians, i.e. symmetric positive definite); products are plain batched GEMMs.
"""
import torch

from .. import ops


def _inv(M):
    return ops.spd_inverse(M)


def _logdet(M):
    return ops.spd_inv_logdet(M)[1]


class matrix_utils():

    def block_diag_matrix_builder(A, B):
        n1, n2 = A.shape[-1], B.shape[-1]
        t = tuple(A.shape[:-2])
        top = torch.cat((A, A.new_zeros(t + (n1, n2))), -1)
        bot = torch.cat((A.new_zeros(t + (n2, n1)), B), -1)
        return torch.cat((top, bot), -2)

    def block_matrix_inverse(A, B, C, D, block_form=True):
        invA, invD = _inv(A), _inv(D)
        Ainv = _inv(A - B @ invD @ C)
        Dinv = _inv(D - C @ invA @ B)
        # the reference compares against the strings 'left' / 'right' / 'True' (:18-25); a boolean
        # True therefore selects the assembled full inverse, like any other value
        if block_form == 'left':
            return Ainv, -B @ invD, -C @ invA, Dinv
        elif block_form == 'right':
            return Ainv, -invA @ B, -invD @ C, Dinv
        elif block_form == 'True':
            return Ainv, -Ainv @ B @ Dinv, -invD @ C @ invA, Dinv
        return matrix_utils.block_matrix_builder(Ainv, -invA @ B @ Dinv, -invD @ C @ Ainv, Dinv)

    def block_matrix_builder(A, B, C, D):
        return torch.cat((torch.cat((A, B), -1), torch.cat((C, D), -1)), -2)

    def block_precision_marginalizer(A, B, C, D):
        invA, invD = _inv(A), _inv(D)
        nBiD = -(B @ invD)
        nCiA = -(C @ invA)
        return A + nBiD @ C, nBiD, nCiA, D + nCiA @ B

    def block_matrix_logdet(A, B, C, D, singular=False):
        if singular == 'D':
            iA, ldA = ops.spd_inv_logdet(A)
            return ldA + _logdet(D - C @ iA @ B)
        iD, ldD = ops.spd_inv_logdet(D)
        return ldD + _logdet(A - B @ iD @ C)
