"""Small host-side helpers shared by the node classes."""
import torch


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
