#!/usr/bin/env python3
"""Gaussian mixture on synthetic clusters; the iteration is replayed as one HIP graph when the problem is small.

    python examples/gmm.py [--n 100000] [--dim 8] [--components 6] [--iters 25]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pyvbmp_amd.models import GaussianMixtureModel  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100000)
    ap.add_argument("--dim", type=int, default=8)
    ap.add_argument("--components", type=int, default=6)
    ap.add_argument("--iters", type=int, default=25)
    args = ap.parse_args()
    dev, dt = "cuda", torch.float64
    gen = torch.Generator(device=dev).manual_seed(0)
    centers = 4.0 * torch.randn(args.components, args.dim, generator=gen, device=dev, dtype=dt)
    z = torch.randint(args.components, (args.n,), generator=gen, device=dev)
    X = centers[z] + torch.randn(args.n, args.dim, generator=gen, device=dev, dtype=dt)
    torch.manual_seed(0)
    model = GaussianMixtureModel(args.components, args.dim, device=dev, dtype=dt)
    model.initialize(X)
    model.update(X, iters=args.iters, graphed=args.n <= 20000)
    # cluster purity against the generating labels
    conf = torch.zeros(args.components, args.components, device=dev)
    conf.index_put_((z, model.assignment()), torch.ones(args.n, device=dev), accumulate=True)
    purity = float(conf.max(0)[0].sum() / args.n)
    print(f"GMM, {args.n} points in {args.dim}-D, {args.components} components: ELBO per point "
          f"{float(model.ELBO_last) / args.n:.3f}, purity {purity:.3f}")
    return purity


if __name__ == "__main__":
    main()
