#!/usr/bin/env python3
"""Soak test of the f16x3 training kernels (hand-counted vmcnt waits, LDS-DMA stages, transposed LDS reads): many launches at
several shapes, every output compared bit for bit with the first launch.   python tools/soak_trx.py [--launches 300]"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hip = importlib.import_module("3dal_pytorch_amd._hip")
train = importlib.import_module("3dal_pytorch_amd.train")
lib = hip.lib()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--launches", type=int, default=300)
    n = ap.parse_args().launches
    g = torch.Generator(device="cuda").manual_seed(11)
    bad = 0
    for M, ci, co, seg in ((262144, 512, 256, 0), (65536, 64, 512, 4096), (24576, 128, 768, 256), (4096, 2048, 256, 0)):
        a = torch.randn((M, ci), device="cuda", generator=g)
        W = torch.randn((co, ci), device="cuda", generator=g) / ci ** 0.5
        sc, sh = torch.rand(ci, device="cuda", generator=g) + 0.5, torch.randn(ci, device="cuda", generator=g) * 0.3
        bias = torch.randn((M // seg if seg else 1, co), device="cuda", generator=g)
        pk = torch.empty(lib.dal3_tr_linear_workspace_bytes(ci, co), dtype=torch.uint8, device="cuda")
        item = (hip.PackItem * 1)(hip.PackItem(hip.ptr(W), W.stride(0), 0, co, ci, 0x108, hip.ptr(pk)))
        hip.check(lib.dal3_tr_pack_many(item, 1, hip.stream()))
        run = lambda: train._linear(a, W, ci, ci, co, act=(sc, sh, True), bias=bias, seg=seg, packed=train._X3Image(pk))   # noqa: E731
        first = run().clone()
        nb = sum(0 if torch.equal(run(), first) else 1 for _ in range(n))
        bad += nb
        print(f"linear {M} x {ci} -> {co}: {n} launches, {nb} mismatching", flush=True)
    for M, ci, co, seg in ((262144, 128, 1024, 4096), (32768, 256, 512, 512)):
        a = torch.randn((M, ci), device="cuda", generator=g)
        W = torch.randn((co, ci), device="cuda", generator=g) / ci ** 0.5
        b = torch.randn(co, device="cuda", generator=g) * 0.1
        sc, sh = torch.rand(ci, device="cuda", generator=g) + 0.5, torch.randn(ci, device="cuda", generator=g) * 0.3

        class BN:
            scale = torch.rand(co, device="cuda", generator=g) + 0.5
            shift = torch.randn(co, device="cuda", generator=g) * 0.3
        with train.arithmetic("f16x3"):
            g0, a0 = (t.clone() for t in train._linear_pool(a, (sc, sh, True), W, b, BN, seg))
            nb = 0
            for _ in range(n):
                g1, a1 = train._linear_pool(a, (sc, sh, True), W, b, BN, seg)
                nb += 0 if (torch.equal(g1, g0) and torch.equal(a1, a0)) else 1
        bad += nb
        print(f"pool {M} x {ci} -> {co}: {n} launches, {nb} mismatching", flush=True)
    for M, co, ci in ((262144, 256, 512), (262144, 128, 256), (65536 + 32, 128, 128), (32768, 512, 64)):
        dz = torch.randn((M, co), device="cuda", generator=g) * 3e-6
        a = torch.randn((M, ci), device="cuda", generator=g)
        sc, sh = torch.rand(ci, device="cuda", generator=g) + 0.5, torch.randn(ci, device="cuda", generator=g) * 0.3
        amax = torch.zeros(64, dtype=torch.int32, device="cuda")
        amax[9] = dz.abs().max().reshape(1).view(torch.int32)[0]
        run = lambda: train._wgrad(dz, a, co, ci, (sc, sh, True), amax=amax)   # noqa: E731
        first = run().clone()
        nb = sum(0 if torch.equal(run(), first) else 1 for _ in range(n))
        bad += nb
        print(f"wgrad {M}: {co} x {ci}: {n} launches, {nb} mismatching", flush=True)
    # round 6: the BatchNorm-backward apply pass that also takes max |dz| — the four waves of a workgroup meet in LDS behind a
    # barrier and ONE integer atomicMax leaves per workgroup: dz and the maximum bit for bit, every launch
    for M, C in ((262144, 256), (262144, 128), (65536 + 96, 512), (8192, 64)):
        z = torch.randn((M, C), device="cuda", generator=g)
        bn = train._BN(z, torch.rand(C, device="cuda", generator=g) + 0.5, torch.randn(C, device="cuda", generator=g), None, None, rows=M)
        da = torch.randn((M, C), device="cuda", generator=g) * 1e-6

        def run():
            words = torch.zeros(64, dtype=torch.int32, device="cuda")
            dz = bn.backward(z, da=da, amax=words)[0]
            return dz, words
        dz0, w0 = run()
        assert int(w0.max()) == int(dz0.abs().max().view(torch.int32))
        nb = 0
        for _ in range(n):
            dz1, w1 = run()
            nb += 0 if (torch.equal(dz1, dz0) and torch.equal(w1, w0)) else 1
        bad += nb
        print(f"apply+amax {M} x {C}: {n} launches, {nb} mismatching", flush=True)
    print("total mismatches", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
