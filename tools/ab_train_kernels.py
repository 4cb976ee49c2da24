#!/usr/bin/env python3
"""A/B timing of the training GEMM kernels of library builds in ONE process (boxes differ by several per cent):
  python tools/ab_train_kernels.py build_a.so build_b.so [--M 262144] [--rounds 5]
interleaved rounds of dal3_tr_linear (forward, dgrad) and dal3_tr_wgrad at the static train step's layer shapes."""
import argparse
import ctypes as C
import importlib
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hip = importlib.import_module("3dal_pytorch_amd._hip")
ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--M", type=int, default=64 * 4096)
ap.add_argument("--rounds", type=int, default=5)
args = ap.parse_args()
dev = torch.device("cuda", 0)
M = args.M


def load(path):
    h = C.CDLL(os.path.abspath(path))
    for name, (res, a) in hip.SIGNATURES.items():
        if hasattr(h, name):
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, a
    return h


libs = [(os.path.basename(p), load(p)) for p in args.libs]
st = hip.stream()
shapes = [(32, 64), (64, 64), (64, 128), (64, 512), (512, 256), (256, 128), (128, 128), (128, 32)]
if M <= 1024:                                               # the per-item FC tails (rows = items)
    shapes = [(512, 512), (512, 256), (256, 64), (384, 512)]
ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)


def timed(fn, iters=5):
    fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def pool_case():
    """dal3_tr_linear_pool at conv5's shape (128 -> 1024, 64 x 4096 points)"""
    ci, co, seg = 128, 1024, 4096
    a = torch.randn((M, ci), device=dev)
    W = torch.randn((co, ci), device=dev) * 0.1
    b = torch.randn(co, device=dev)
    sc, sh = torch.rand(ci, device=dev) + 0.5, torch.randn(ci, device=dev) * 0.1
    osc, osh = torch.rand(co, device=dev) + 0.5, torch.randn(co, device=dev) * 0.1
    n_seg = M // seg
    out = []
    res = {n: [] for n, _ in libs}
    for r in range(args.rounds):
        for n, lib in libs:
            g = torch.empty((n_seg, co), device=dev)
            arg = torch.empty((n_seg, co), dtype=torch.int32, device=dev)
            need = lib.dal3_tr_linear_pool_workspace_bytes(ci, co, n_seg)
            w2 = torch.empty(need, dtype=torch.uint8, device=dev)
            f = lambda: lib.dal3_tr_linear_pool(hip.ptr(a), M, ci, ci, hip.ptr(sc), hip.ptr(sh), 1, hip.ptr(W), ci, hip.ptr(b),
                                                hip.ptr(osc), hip.ptr(osh), seg, co, hip.ptr(g), hip.ptr(arg), hip.ptr(w2), need, st)
            res[n].append(timed(f))
            if r == 0:
                out.append((g.clone(), arg.clone()))
    same = all(torch.equal(out[0][0], o[0]) and torch.equal(out[0][1], o[1]) for o in out)
    print("linear_pool 128 -> 1024 + max: " + "  ".join(f"[{n}] {statistics.median(v):7.1f} us" for n, v in res.items()) + f"  (same bits: {same})")


if M % 4096 == 0 and M >= 8192:
    pool_case()
for ci, co in shapes:
    a = torch.randn((M, ci), device=dev)
    W = torch.randn((co, ci), device=dev) * 0.1
    b = torch.randn(co, device=dev)
    sc, sh = torch.rand(ci, device=dev) + 0.5, torch.randn(ci, device=dev) * 0.1
    dz = torch.randn((M, co), device=dev)
    z = torch.empty((M, co), device=dev)
    da = torch.empty((M, ci), device=dev)
    dW = [torch.empty((co, ci), device=dev) for _ in libs]
    res = {(n, k): [] for n, _ in libs for k in ("fwd", "dgrad", "wgrad")}
    for r in range(args.rounds):
        for i, (n, lib) in enumerate(libs):
            res[(n, "fwd")].append(timed(lambda: lib.dal3_tr_linear(hip.ptr(a), M, ci, ci, hip.ptr(sc), hip.ptr(sh), 1, hip.ptr(W), ci, 0,
                                                                    hip.ptr(b), 0, co, hip.ptr(z), co, 0, hip.ptr(ws), ws.numel(), st)))
            res[(n, "dgrad")].append(timed(lambda: lib.dal3_tr_linear(hip.ptr(dz), M, co, co, None, None, 0, hip.ptr(W), ci, 1, None, 0, ci,
                                                                      hip.ptr(da), ci, 0, hip.ptr(ws), ws.numel(), st)))
            res[(n, "wgrad")].append(timed(lambda: lib.dal3_tr_wgrad(hip.ptr(dz), co, hip.ptr(a), ci, hip.ptr(sc), hip.ptr(sh), 1, M, co, ci,
                                                                     hip.ptr(ws), ws.numel(), hip.ptr(dW[i]), st)))
    torch.cuda.synchronize()
    ref = dz.double().t() @ torch.relu(a.double() * sc.double() + sh.double())
    zref = torch.relu(a.double() * sc.double() + sh.double()) @ W.double().t() + b.double()
    zerr = float((z.double() - zref).abs().max() / zref.abs().max())      # (z: the LAST library's forward)
    daerr = float((da.double() - dz.double() @ W.double()).abs().max() / (dz.double() @ W.double()).abs().max())
    line = f"{ci:4d} -> {co:4d} (z err {zerr:.0e}, da err {daerr:.0e}): "
    for i, (n, _) in enumerate(libs):
        err = float((dW[i].double() - ref).abs().max() / ref.abs().max())
        line += f"[{n}] fwd {statistics.median(res[(n, 'fwd')]):7.1f} dgrad {statistics.median(res[(n, 'dgrad')]):7.1f} " \
                f"wgrad {statistics.median(res[(n, 'wgrad')]):7.1f} us (dW err {err:.1e})  "
    print(line)
