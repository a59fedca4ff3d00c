"""Synthetic inputs shared by the benches, the golden-fixture generator and the tests' full-size cases.

The reference's flocking data (./data/flocking.pt, examples/Flocking_example.py:33) is git-ignored and absent, and the
tree has no simulator for it: `boids` is the stand-in with the same layout."""
import torch


def boids(T, S, n, gen, dt=0.1, device=None, dtype=torch.float64):
    """small Couzin/boids-style flock in 2-D: (T, S, n, 4) = positions and velocities of n birds in S independent runs,
    standardised like the reference's preprocessing (examples/Flocking_example.py:26: data / data.std)"""
    kw = {"generator": gen, "device": device if device is not None else gen.device, "dtype": dtype}
    pos = torch.randn(S, n, 2, **kw)
    vel = torch.randn(S, n, 2, **kw) * 0.5
    out = []
    for _ in range(T):
        com = pos.mean(1, keepdim=True)
        d = pos.unsqueeze(2) - pos.unsqueeze(1)                      # (S,n,n,2)
        rep = (d / (d.pow(2).sum(-1, keepdim=True) + 0.1)).sum(2)
        align = vel.mean(1, keepdim=True) - vel
        vel = vel + dt * (0.5 * (com - pos) + 0.3 * rep + 0.4 * align) + 0.05 * torch.randn(vel.shape, **kw)
        vel = vel / vel.norm(dim=-1, keepdim=True).clamp_min(0.3)
        pos = pos + dt * vel
        out.append(torch.cat((pos, vel), -1))
    y = torch.stack(out)
    return (y - y.mean((0, 1, 2), keepdim=True)) / y.std()
