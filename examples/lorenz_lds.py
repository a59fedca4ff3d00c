#!/usr/bin/env python3
"""Linear dynamical system fitted to Lorenz-63 trajectories (positions and velocities as six observables, as the
reference's examples/Lorenz_example.py feeds them; trajectories from our own Euler integrator).

    python examples/lorenz_lds.py [--series 256] [--steps 400] [--hidden 6] [--iters 10]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pyvbmp_amd.models import LinearDynamicalSystems  # noqa: E402


def lorenz(T, S, gen, device, dt=0.01, stride=5):
    x = torch.randn(S, 3, generator=gen, device=device, dtype=torch.float64) * 5
    x[:, 2] += 25.0
    out = []
    for i in range(T * stride):
        dx = torch.stack((10.0 * (x[:, 1] - x[:, 0]), x[:, 0] * (28.0 - x[:, 2]) - x[:, 1],
                          x[:, 0] * x[:, 1] - 8.0 / 3.0 * x[:, 2]), -1)
        x = x + dt * dx
        if i % stride == 0:
            out.append(x.clone())
    d = torch.stack(out)
    v = torch.cat((d[1:] - d[:-1], d[-1:] - d[-2:-1]), 0) / dt / 20.0
    z = torch.cat((d / 10.0, v), -1)
    return z - z.mean((0, 1), keepdim=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--series", type=int, default=256)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--hidden", type=int, default=6)
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    dev = "cuda"
    gen = torch.Generator(device=dev).manual_seed(0)
    y = lorenz(args.steps, args.series, gen, dev)
    torch.manual_seed(0)
    model = LinearDynamicalSystems((6,), args.hidden, device=dev, dtype=torch.float64)
    trace = []
    for it in range(args.iters):
        model.update(y, iters=1)
        trace.append(float(model.ELBO_last))
    print(f"Lorenz LDS, T={args.steps}, {args.series} series, hidden {args.hidden}: ELBO per (t, series) "
          f"{trace[0] / y.shape[0] / y.shape[1]:.3f} -> {trace[-1] / y.shape[0] / y.shape[1]:.3f} in {args.iters} iterations")
    return trace


if __name__ == "__main__":
    main()
