"""Oracle: block-matrix helpers (torch CPU, LU route).  TEST INFRASTRUCTURE ONLY.

Reference: utils/matrix_utils.py.
"""
import torch

inv = torch.linalg.inv


def block_diag(A, B):
    """ref utils/matrix_utils.py:4-9"""
    t = A.shape[:-2]
    n1, n2 = A.shape[-1], B.shape[-1]
    top = torch.cat((A, A.new_zeros(t + (n1, n2))), -1)
    bot = torch.cat((A.new_zeros(t + (n2, n1)), B), -1)
    return torch.cat((top, bot), -2)


def block_build(A, B, C, D):
    """ref utils/matrix_utils.py:27-29"""
    return torch.cat((torch.cat((A, B), -1), torch.cat((C, D), -1)), -2)


def block_inverse(A, B, C, D, block_form=True):
    """ref utils/matrix_utils.py:11-25.  Note the reference compares block_form with the *string* 'True',
    so the default (bool True) lands in the final branch and returns the assembled full inverse."""
    iA, iD = inv(A), inv(D)
    SA = inv(A - B @ iD @ C)
    SD = inv(D - C @ iA @ B)
    if block_form == "left":
        return SA, -B @ iD, -C @ iA, SD
    if block_form == "right":
        return SA, -iA @ B, -iD @ C, SD
    if block_form == "True":
        return SA, -SA @ B @ SD, -iD @ C @ iA, SD
    return block_build(SA, -iA @ B @ SD, -iD @ C @ SA, SD)


def precision_marginalizer(A, B, C, D):
    """ref utils/matrix_utils.py:31-46"""
    iA, iD = inv(A), inv(D)
    return A - B @ iD @ C, -B @ iD, -C @ iA, D - C @ iA @ B


def block_logdet(A, B, C, D, singular=False):
    """ref utils/matrix_utils.py:49-55"""
    if singular == "D":
        return torch.logdet(A) + torch.logdet(D - C @ inv(A) @ B)
    return torch.logdet(D) + torch.logdet(A - B @ inv(D) @ C)
