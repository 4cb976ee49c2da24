#!/usr/bin/env python3
"""f16x3 training-forward linear (dal3_tr_linear_x3) against the fp32-MFMA one (dal3_tr_linear) and float64, per layer shape:
max error relative to the output's range, and the launch times."""
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


def case(M, ci, co, act, seg=0):
    g = torch.Generator(device="cuda").manual_seed(ci * 1000 + co)
    a = torch.randn((M, ci), device="cuda", generator=g) * 1.5
    W = torch.randn((co, ci), device="cuda", generator=g) / ci ** 0.5
    sc = sh = None
    if act:
        sc = torch.rand(ci, device="cuda", generator=g) + 0.5
        sh = torch.randn(ci, device="cuda", generator=g) * 0.3
    n_seg = M // seg if seg else 1
    bias = torch.randn((n_seg, co), device="cuda", generator=g).contiguous()
    x = torch.relu(a.double() * sc.double() + sh.double()) if act else a.double()
    ref = x @ W.double().t() + (bias.double().repeat_interleave(seg, 0) if seg else bias.double())
    rng = ref.abs().max().item()
    z32 = train._linear(a, W, ci, ci, co, act=(sc, sh, True) if act else None, bias=bias, seg=seg)
    lay = lib.dal3_tr_linear_x3_layout(M, ci, seg, co, 0, int(act))
    assert lay, (M, ci, co)
    pk = torch.empty(lib.dal3_tr_linear_workspace_bytes(ci, co), dtype=torch.uint8, device="cuda")
    item = (hip.PackItem * 1)(hip.PackItem(hip.ptr(W), W.stride(0), 0, co, ci, lay, hip.ptr(pk)))
    hip.check(lib.dal3_tr_pack_many(item, 1, hip.stream()))
    zx = torch.empty((M, co), device="cuda")

    def run_x3():
        hip.check(lib.dal3_tr_linear_x3(hip.ptr(a), M, ci, a.stride(0), hip.ptr(sc), hip.ptr(sh), 1, hip.ptr(bias), seg, co,
                                        hip.ptr(zx), zx.stride(0), hip.ptr(pk), None, hip.stream()))

    def run_32():
        train._linear(a, W, ci, ci, co, act=(sc, sh, True) if act else None, bias=bias, seg=seg, out=z32)
    run_x3()
    torch.cuda.synchronize()
    e32 = (z32.double() - ref).abs().max().item() / rng
    ex = (zx.double() - ref).abs().max().item() / rng
    t32, tx = events_ms(run_32), events_ms(run_x3)
    gf = 2.0 * M * ci * co / 1e9
    print(f"M={M} {ci:4d}->{co:4d} act={int(act)} seg={seg}: err vs f64  fp32 {e32:.2e}  f16x3 {ex:.2e};  ms fp32 {t32:.3f} ({gf / t32:.0f} GF/ms)  "
          f"f16x3 {tx:.3f} ({gf / tx:.0f} GF/ms)  x{t32 / tx:.2f}", flush=True)


def pool_case(M, ci, co, seg):
    g = torch.Generator(device="cuda").manual_seed(7)
    a = torch.randn((M, ci), device="cuda", generator=g) * 1.5
    W = torch.randn((co, ci), device="cuda", generator=g) / ci ** 0.5
    b = torch.randn(co, device="cuda", generator=g) * 0.1
    sc = torch.rand(ci, device="cuda", generator=g) + 0.5
    sh = torch.randn(ci, device="cuda", generator=g) * 0.3

    class BN:
        scale = torch.rand(co, device="cuda", generator=g) + 0.5
        shift = torch.randn(co, device="cuda", generator=g) * 0.3
    BN.scale[::7] *= -1.0
    g32, a32 = train._linear_pool(a, (sc, sh, True), W, b, BN, seg)
    n_seg = M // seg
    need = lib.dal3_tr_linear_pool_workspace_bytes(ci, co, n_seg)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    gx = torch.empty((n_seg, co), device="cuda")
    ax = torch.empty((n_seg, co), dtype=torch.int32, device="cuda")
    assert lib.dal3_tr_linear_pool_x3_ok(M, ci, seg, co)

    def run_x3():
        hip.check(lib.dal3_tr_linear_pool_x3(hip.ptr(a), M, ci, a.stride(0), hip.ptr(sc), hip.ptr(sh), 1, hip.ptr(W), W.stride(0),
                                             hip.ptr(b), hip.ptr(BN.scale), hip.ptr(BN.shift), seg, co, hip.ptr(gx), hip.ptr(ax),
                                             hip.ptr(ws), need, hip.stream()))
    run_x3()
    torch.cuda.synchronize()
    x = torch.relu(a.double() * sc.double() + sh.double())
    y = torch.relu((x @ W.double().t() + b.double()) * BN.scale.double() + BN.shift.double()).view(n_seg, seg, co)
    ref, rarg = y.max(1)
    rng = ref.abs().max().item()
    print(f"pool M={M} {ci}->{co}: g err vs f64  fp32 {(g32.double() - ref).abs().max().item() / rng:.2e}  f16x3 {(gx.double() - ref).abs().max().item() / rng:.2e}; "
          f"arg == f64's: fp32 {(a32.long() == rarg).float().mean().item():.5f}  f16x3 {(ax.long() == rarg).float().mean().item():.5f}; "
          f"ms fp32 {events_ms(lambda: train._linear_pool(a, (sc, sh, True), W, b, BN, seg)):.3f}  f16x3 {events_ms(run_x3):.3f}", flush=True)


if __name__ == "__main__":
    pool_case(64 * 4096, 128, 1024, 4096)
    pool_case(64 * 512, 256, 512, 512)
    M = 64 * 4096
    case(M, 512, 256, True)
    case(M, 64, 512, True, seg=4096)
    case(M, 128, 1024, True)
    case(M, 256, 512, False)
    case(M, 128, 256, True)
    case(32768, 128, 256, True)
