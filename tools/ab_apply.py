#!/usr/bin/env python3
"""A/B of the BatchNorm-backward apply pass (dal3_tr_bnbwd_apply / _segsum) between library builds in ONE process, at the
training step's shapes (262,144 rows):  python tools/ab_apply.py a.so b.so"""
import ctypes as C
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hip = importlib.import_module("3dal_pytorch_amd._hip")
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
for Cc in (512, 256, 128, 64):
    z = torch.randn((M, Cc), device="cuda", generator=gen)
    da = torch.randn((M, Cc), device="cuda", generator=gen)
    v = [torch.rand(Cc, device="cuda", generator=gen) + 0.5 for _ in range(7)]
    outs, fns = {}, []
    for name, lib in libs:
        dz = torch.empty((M, Cc), device="cuda")
        need = lib.dal3_tr_bnbwd_apply_segsum_workspace_bytes(M, Cc)
        ws = torch.empty(need, dtype=torch.uint8, device="cuda")
        seg = torch.empty((M // 4096, Cc), device="cuda")

        def plain(lib=lib, dz=dz):
            hip.check(lib.dal3_tr_bnbwd_apply(hip.ptr(z), M, Cc, Cc, hip.ptr(da), Cc, None, None, 0, *[hip.ptr(t) for t in v], hip.ptr(dz), Cc,
                                              hip.stream()))

        def segsum(lib=lib, dz=dz, ws=ws, seg=seg, need=need):
            hip.check(lib.dal3_tr_bnbwd_apply_segsum(hip.ptr(z), M, Cc, Cc, hip.ptr(da), Cc, *[hip.ptr(t) for t in v], hip.ptr(dz), Cc, 4096,
                                                     hip.ptr(seg), hip.ptr(ws), need, hip.stream()))
        fns.append((name, {"apply": plain, "apply+segsum": segsum}))
        segsum()
        outs[name] = (dz.clone(), seg.clone())
    best = {(n, k): 1e9 for n, f in fns for k in f}
    for _ in range(4):
        for n, f in fns:
            for k, fn in f.items():
                best[n, k] = min(best[n, k], timed(fn))
    n0 = libs[0][0]
    same = all(torch.equal(outs[n0][0], o[0]) for o in outs.values())
    segerr = max(float((outs[n0][1] - o[1]).abs().max() / outs[n0][1].abs().max()) for o in outs.values())
    gb = 12.0 * M * Cc / 1e9
    print(f"C {Cc:4d}:" + "".join(f"  [{n}] " + "  ".join(f"{k} {best[n, k]:6.1f} us ({gb / best[n, k] * 1e3:4.2f} TB/s)" for k in f) for n, f in fns)
          + f"  same dz: {same}, seg sums differ by {segerr:.1e}")
