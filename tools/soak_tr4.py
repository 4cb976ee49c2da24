#!/usr/bin/env python3
"""Soak test of round 4's fp32 training kernels — the pooled layer with resident activations (pinned epilogue steps, private LDS
candidate rows, prefetch across groups), the linear kernels with reduction epilogues, the narrow-layer and FC-tail kernels:
many launches at the step's shapes, every output compared bit for bit with the first launch.
  python tools/soak_tr4.py [--launches 300]"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hip = importlib.import_module("3dal_pytorch_amd._hip")
train = importlib.import_module("3dal_pytorch_amd.train")


def soak(name, run, n):
    first = [t.clone() for t in run()]
    bad = 0
    for _ in range(n):
        out = run()
        bad += 0 if all(torch.equal(a, b) for a, b in zip(out, first)) else 1
    print(f"{name}: {n} launches, {bad} mismatching", flush=True)
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--launches", type=int, default=300)
    n = ap.parse_args().launches
    g = torch.Generator(device="cuda").manual_seed(4)
    rnd = lambda *s, k=1.0: torch.randn(s, device="cuda", generator=g) * k                       # noqa: E731
    bad = 0
    for M, ci, co, seg in ((262144, 128, 1024, 4096), (76800, 128, 256, 192), (73728, 128, 128, 4096)):
        a, W, b = rnd(M, ci), rnd(co, ci, k=ci ** -0.5), rnd(co, k=0.1)
        sc, sh = torch.rand(ci, device="cuda", generator=g) + 0.5, rnd(ci, k=0.3)

        class BN:
            scale = torch.rand(co, device="cuda", generator=g) - 0.3
            shift = rnd(co, k=0.3)
        bad += soak(f"pooled layer {M} x {ci} -> {co}, segments of {seg}", lambda: train._linear_pool(a, (sc, sh, True), W, b, BN, seg), n)
    for M, ci, co in ((262144, 512, 256), (262144, 64, 512), (262144, 128, 128)):
        a, W, b = rnd(M, ci), rnd(co, ci, k=ci ** -0.5), rnd(co, k=0.1)
        sc, sh = torch.rand(ci, device="cuda", generator=g) + 0.5, rnd(ci, k=0.3)
        gamma, beta = torch.rand(co, device="cuda", generator=g) + 0.5, rnd(co, k=0.1)
        pk = train._prepack([(W, ci, co, False, M, 0, False, True)], a.device)[0]

        def fwd():
            z, bn = train._linear_bn(a, W, ci, co, (sc, sh, True), b, 0, pk, gamma, beta, None, M)
            return z, bn.mu, bn.rstd
        bad += soak(f"linear + statistics {M} x {ci} -> {co}", fwd, n)
    for M, K, C in ((262144, 256, 512), (262144, 128, 128)):
        dz, W, bz = rnd(M, K, k=1e-3), rnd(K, C, k=K ** -0.5), rnd(M, C)
        bn = train._BN(bz, torch.ones(C, device="cuda"), torch.zeros(C, device="cuda"), None, None)
        pk = train._prepack([(W, K, C, True, M, 0, False, False)], W.device)[0]
        bad += soak(f"dgrad + backward sums {M} x {K} -> {C}", lambda: bn.dgrad_with_sums(bz, dz, W, C, K, pk), n)
    for B, ci, co in ((64, 512, 512), (256, 512, 256), (37, 384, 512)):
        x, W, b = rnd(B, ci).abs(), rnd(co, ci, k=ci ** -0.5), rnd(co, k=0.1)
        gamma, beta = torch.rand(co, device="cuda", generator=g) + 0.5, rnd(co, k=0.1)
        da = rnd(B, co)

        def fc():
            z, bn = train._fc_forward(x, None, W, b, co, bn=(gamma, beta, None, None))
            dz, dW, db, dgam, dbet = train._fc_backward_w(da, z, bn, x, None, W.shape)
            return z, bn.mu, bn.rstd, dz, dW, dgam, dbet, train._fc_forward(dz, None, W, None, ci, transpose=True)
        bad += soak(f"FC layer {B} x {ci} -> {co}: forward, backward", fc, n)
    print("soak: " + ("0 mismatches" if bad == 0 else f"{bad} MISMATCHES"))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
