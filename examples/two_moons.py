#!/usr/bin/env python3
"""Two interleaved half-moons learned as a gated mixture of linear maps x -> y
(the model class of the reference's examples/two_moons.py: dMixtureofLinearTransforms; data from our own generator).

    python examples/two_moons.py [--n 2000] [--experts 6] [--iters 30]
"""
import argparse
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pyvbmp_amd.transforms import dMixtureofLinearTransforms  # noqa: E402


def two_moons(n, noise, gen, device, dtype):
    t = torch.rand(n, generator=gen, device=device, dtype=dtype) * math.pi
    upper = torch.rand(n, generator=gen, device=device, dtype=dtype) < 0.5
    x = torch.where(upper, torch.cos(t), 1.0 - torch.cos(t))
    y = torch.where(upper, torch.sin(t), 0.5 - torch.sin(t))
    xy = torch.stack((x, y), -1) + noise * torch.randn(n, 2, generator=gen, device=device, dtype=dtype)
    return xy[:, :1], xy[:, 1:], upper


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2000)
    ap.add_argument("--experts", type=int, default=6)
    ap.add_argument("--iters", type=int, default=30)
    args = ap.parse_args()
    dev, dt = "cuda", torch.float64
    gen = torch.Generator(device=dev).manual_seed(0)
    X, Y, _ = two_moons(args.n, 0.05, gen, dev, dt)
    torch.manual_seed(0)
    model = dMixtureofLinearTransforms(1, 1, args.experts, device=dev, dtype=dt)
    ll0 = float(model.Elog_like(X, Y).mean())
    model.raw_update(X, Y, iters=args.iters, lr=1.0)
    ll1 = float(model.Elog_like(X, Y).mean())
    pY, gate = model.predict(X)
    rmse = float((pY.mean().squeeze(-1) - Y).pow(2).mean().sqrt())
    print(f"two moons, {args.n} points, {args.experts} experts: mean log-likelihood bound {ll0:.3f} -> {ll1:.3f}; "
          f"predictive RMSE {rmse:.3f}; experts in use {int((gate.mean(0) > 0.02).sum())}")
    return ll0, ll1


if __name__ == "__main__":
    main()
