#!/usr/bin/env python3
"""f16x3 wgrad (dal3_tr_wgrad_x3) against the fp32-MFMA one (dal3_tr_wgrad) and float64 per layer shape: error relative to the
result's range, launch times. dz at gradient magnitudes (1e-6, six decades of spread between points) with its amax words."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
hip = importlib.import_module("3dal_pytorch_amd._hip")
train = importlib.import_module("3dal_pytorch_amd.train")
lib = hip.lib()


def events_ms(fn, iters=10):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / iters


def case(M, co, ci, act=True):
    g = torch.Generator(device="cuda").manual_seed(co * 7 + ci)
    spread = torch.exp(torch.rand((M, 1), device="cuda", generator=g) * -12.0)
    dz = torch.randn((M, co), device="cuda", generator=g) * spread * 3e-6
    a = torch.randn((M, ci), device="cuda", generator=g) * 1.5
    sc = torch.rand(ci, device="cuda", generator=g) + 0.5
    sh = torch.randn(ci, device="cuda", generator=g) * 0.3
    x = torch.relu(a.double() * sc.double() + sh.double()) if act else a.double()
    ref = dz.double().t() @ x
    rng = ref.abs().max().item()
    w32 = train._wgrad(dz, a, co, ci, (sc, sh, True) if act else None)
    need = lib.dal3_tr_wgrad_x3_workspace_bytes(M, co, ci)
    assert need, (M, co, ci)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    amax = torch.zeros(64, dtype=torch.int32, device="cuda")
    amax[3] = dz.abs().max().reshape(1).view(torch.int32)[0]
    wx = torch.empty((co, ci), device="cuda")

    def run_x3():
        hip.check(lib.dal3_tr_wgrad_x3(hip.ptr(dz), dz.stride(0), hip.ptr(a), a.stride(0), hip.ptr(sc) if act else None,
                                       hip.ptr(sh) if act else None, 1, hip.ptr(amax), M, co, ci, hip.ptr(ws), need, hip.ptr(wx),
                                       hip.stream()))
    run_x3()
    torch.cuda.synchronize()
    e32 = (w32.double() - ref).abs().max().item() / rng
    ex = (wx.double() - ref).abs().max().item() / rng
    t32 = events_ms(lambda: train._wgrad(dz, a, co, ci, (sc, sh, True) if act else None))
    tx = events_ms(run_x3)
    print(f"wgrad M={M} {co:4d} x {ci:4d}: err vs f64  fp32 {e32:.2e}  f16x3 {ex:.2e};  ms fp32 {t32:.3f}  f16x3 {tx:.3f}  x{t32 / tx:.2f}", flush=True)


if __name__ == "__main__":
    M = 64 * 4096
    case(M, 256, 512)
    case(M, 128, 256)
    case(M, 128, 128)
    case(M, 512, 64)
    case(32768, 128, 128)
