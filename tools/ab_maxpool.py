#!/usr/bin/env python3
"""A/B of the standalone max-pool kernel between library builds in one process: python tools/ab_maxpool.py a.so b.so"""
import ctypes as C, importlib, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hip = importlib.import_module("3dal_pytorch_amd._hip")
dev = torch.device("cuda:0")
B, Cc, N = 4096, 1024, 1024
x = torch.randn((B * Cc, N), device=dev)
out = torch.empty(B * Cc, device=dev)
st = hip.stream()
libs = []
for p in sys.argv[1:]:
    h = C.CDLL(os.path.abspath(p))
    for name, (r, a) in hip.SIGNATURES.items():
        if hasattr(h, name):
            fn = getattr(h, name); fn.restype, fn.argtypes = r, a
    libs.append((os.path.basename(p), h))
res = {n: [] for n, _ in libs}
for r in range(7):
    for n, h in libs:
        h.dal3_maxpool_n(hip.ptr(x), B * Cc, N, hip.ptr(out), st)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            h.dal3_maxpool_n(hip.ptr(x), B * Cc, N, hip.ptr(out), st)
        b.record(); b.synchronize()
        res[n].append(a.elapsed_time(b) / 5)
for n in res:
    ms = statistics.median(res[n])
    print(f"{n:16s} {ms:.3f} ms  {x.numel() * 4 / ms / 1e9:.2f} TB/s")
