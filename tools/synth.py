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


def lorenz(T, S, gen, dt=0.01, stride=5, device=None, dtype=torch.float64):
    """Euler-integrated Lorenz-63 trajectories with finite-difference velocities, (T, S, 6), centred: the same kind of
    data as the reference's simulations/Lorenz.py:16-58 concatenated to 6 observables (examples/Lorenz_example.py:23-25)"""
    dev = device if device is not None else gen.device
    x = torch.randn(S, 3, generator=gen, device=dev, dtype=torch.float64) * 5 + torch.tensor([0.0, 0.0, 25.0], device=dev,
                                                                                             dtype=torch.float64)
    out = []
    for i in range(T * stride):
        dx = torch.stack((10.0 * (x[:, 1] - x[:, 0]), x[:, 0] * (28.0 - x[:, 2]) - x[:, 1],
                          x[:, 0] * x[:, 1] - 8.0 / 3.0 * x[:, 2]), -1)
        x = x + dt * dx
        if i % stride == 0:
            out.append(x.clone())
    d = torch.stack(out)  # (T, S, 3)
    v = torch.cat((d[1:] - d[:-1], d[-1:] - d[-2:-1]), 0) / dt / 20.0
    z = torch.cat((d / 10.0, v), -1)
    return (z - z.mean((0, 1), keepdim=True)).to(dtype)
