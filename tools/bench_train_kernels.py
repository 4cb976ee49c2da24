#!/usr/bin/env python3
"""Per-kernel rates of the training kernels at the shapes of one static train step (64 x 4096 points):
  python tools/bench_train_kernels.py            # linear / dgrad / wgrad in TFLOP/s, reductions and BN backward in GB/s"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
train = importlib.import_module("3dal_pytorch_amd.train")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 64 * 4096
dev = torch.device("cuda", 0)


def timed(fn, iters=10):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


shapes = [(32, 64), (64, 64), (64, 128), (128, 1024), (64, 512), (512, 256), (256, 128), (128, 128), (128, 32)]
print(f"M = {M} rows")
for ci, co in shapes:
    a = torch.randn((M, ci), device=dev)
    W = torch.randn((co, ci), device=dev) * 0.1
    b = torch.randn(co, device=dev)
    sc, sh = torch.rand(ci, device=dev) + 0.5, torch.randn(ci, device=dev) * 0.1
    dz = torch.randn((M, co), device=dev)
    fl = 2.0 * M * ci * co
    t_f = timed(lambda: train._linear(a, W, ci, ci, co, act=(sc, sh, True), bias=b))
    t_d = timed(lambda: train._linear(dz, W, ci, co, ci, transpose=True))
    t_w = timed(lambda: train._wgrad(dz, a, co, ci, act=(sc, sh, True)))
    z = dz
    t_r = timed(lambda: train._colred(z, 0))
    by_f = 4.0 * M * (ci + co)
    print(f"{ci:5d} -> {co:5d}: fwd {t_f * 1e6:7.1f} us {fl / t_f / 1e12:6.1f} TF/s ({by_f / t_f / 1e12:5.2f} TB/s) | "
          f"dgrad {t_d * 1e6:7.1f} us {fl / t_d / 1e12:6.1f} TF/s | wgrad {t_w * 1e6:7.1f} us {fl / t_w / 1e12:6.1f} TF/s | "
          f"colred(z {co}) {t_r * 1e6:6.1f} us {4.0 * M * co / t_r / 1e12:5.2f} TB/s")
    gamma, beta = torch.rand(co, device=dev) + 0.5, torch.randn(co, device=dev) * 0.1
    bn = train._BN(z, gamma, beta, None, None)
    da = torch.randn((M, co), device=dev)
    t_b = timed(lambda: bn.backward(z, da=da))
    print(f"               BN+ReLU backward (colred mode 1 + apply) on (M,{co}): {t_b * 1e6:7.1f} us, "
          f"{4.0 * M * co * 5 / t_b / 1e12:5.2f} TB/s of 5 passes (z, da read twice; dz written)")
