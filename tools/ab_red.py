#!/usr/bin/env python3
"""A/B of the linear kernels with a fused reduction epilogue (dal3_tr_linear_bn_stats / dal3_tr_linear_bnbwd_sums) between
library builds in ONE process, at the training step's shapes (262,144 rows):  python tools/ab_red.py a.so b.so"""
import ctypes as C
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hip = importlib.import_module("3dal_pytorch_amd._hip")
train = importlib.import_module("3dal_pytorch_amd.train")
M = 64 * 4096


def load(path):
    h = C.CDLL(os.path.abspath(path))
    for name, (res, a) in hip.SIGNATURES.items():
        if hasattr(h, name):
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, a
    return h


def timed(fn, iters=10):
    fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / iters * 1e3


libs = [(os.path.basename(p), load(p)) for p in sys.argv[1:]]
gen = torch.Generator(device="cuda").manual_seed(1)
for K, Cc in ((256, 512), (128, 256), (128, 128), (64, 128)):
    dz = torch.randn((M, K), device="cuda", generator=gen) * 1e-3
    W = torch.randn((K, Cc), device="cuda", generator=gen) / K ** 0.5
    bz = torch.randn((M, Cc), device="cuda", generator=gen)
    bn = train._BN(bz, torch.ones(Cc, device="cuda"), torch.zeros(Cc, device="cuda"), None, None)
    pk = train._prepack([(W, K, Cc, True, M, 0, False, False)], W.device)[0]
    da = torch.empty((M, Cc), device="cuda")
    co = torch.empty((5, Cc), device="cuda")
    z = torch.empty((M, Cc), device="cuda")
    a_in = torch.randn((M, K), device="cuda", generator=gen)
    Wf = torch.randn((Cc, K), device="cuda", generator=gen) / K ** 0.5
    pkf = train._prepack([(Wf, K, Cc, False, M, 0, False, False)], W.device)[0]
    st = torch.empty((4, Cc), device="cuda")
    fns = []
    for name, lib in libs:
        need = lib.dal3_tr_linear_red_workspace_bytes(M, Cc)
        ws = torch.empty(need, dtype=torch.uint8, device="cuda")

        def bwd(lib=lib, ws=ws, need=need):
            lib.dal3_tr_linear_bnbwd_sums(hip.ptr(dz), M, K, K, hip.ptr(W), Cc, Cc, hip.ptr(da), Cc, hip.ptr(pk), M, hip.ptr(bz), Cc,
                                          hip.ptr(bn.scale), hip.ptr(bn.shift), hip.ptr(bn.mu), hip.ptr(bn.rstd), hip.ptr(bn.gamma),
                                          hip.ptr(co[0]), hip.ptr(co[1]), hip.ptr(co[2]), hip.ptr(co[3]), hip.ptr(co[4]), hip.ptr(ws), need,
                                          hip.stream())

        def fwd(lib=lib, ws=ws, need=need):
            lib.dal3_tr_linear_bn_stats(hip.ptr(a_in), M, K, K, None, None, 0, hip.ptr(Wf), K, None, 0, Cc, hip.ptr(z), Cc, hip.ptr(pkf), M,
                                        hip.ptr(bn.gamma), hip.ptr(bn.gamma), None, None, 0.1, 1e-5, hip.ptr(st[0]), hip.ptr(st[1]),
                                        hip.ptr(st[2]), hip.ptr(st[3]), hip.ptr(ws), need, hip.stream())

        def plain(lib=lib):
            lib.dal3_tr_linear_prepacked(hip.ptr(dz), M, K, K, None, None, 0, hip.ptr(W), Cc, 1, None, 0, Cc, hip.ptr(da), Cc, 0, hip.ptr(pk),
                                         hip.stream())
        fns.append((name, {"dgrad+sums": bwd, "fwd+stats": fwd, "plain dgrad": plain}))
    best = {(n, k): 1e9 for n, f in fns for k in f}
    for _ in range(4):                                      # interleaved rounds, minimum (the first thing timed in a process runs slower)
        for n, f in fns:
            for k, fn in f.items():
                best[n, k] = min(best[n, k], timed(fn))
    print(f"{K:4d} -> {Cc:4d}:" + "".join(f"  [{n}] " + "  ".join(f"{k} {best[n, k]:7.1f}" for k in f) for n, f in fns))
