#!/usr/bin/env python3
"""f16x3 bring-up probe: pooled features / logits of the f16x3 kernels vs the fp32 kernels, and their launch times."""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _common import build_model, synth                      # noqa: E402

hip = importlib.import_module("3dal_pytorch_amd._hip")
lib = hip.lib()


def events_ms(fn, iters=10):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / iters


def head(kind, attr, c, B, M):
    model = build_model(kind, synth.state_dict(kind, seed=23))
    mod = getattr(model, attr)
    rng = np.random.default_rng(5)
    x = torch.from_numpy((rng.standard_normal((B, M, c)) * np.array([2.5, 1.0, 0.8, 0.1, 1, 1, 1, 1][:c])).astype(np.float32)).cuda()
    bcn = hip.bcn(x.transpose(2, 1))
    out = {}
    for name, dt in (("fp32", hip.F32), ("f16x3", hip.F16X3), ("fp16", hip.F16)):
        w = model._cache.get(attr + name, mod, mod.HEAD_KIND, dt)
        f = torch.empty((B, 512), device="cuda")
        ws = torch.empty(max(int(lib.dal3_point_head_pool_workspace_bytes(B, M)), 16), dtype=torch.uint8, device="cuda")

        def run(w=w, f=f, dt=dt, ws=ws):
            hip.check(lib.dal3_point_head_pool(mod.HEAD_KIND, hip.ptr(w), dt, bcn, B, M, None, hip.ptr(f), hip.ptr(ws), ws.numel(),
                                               hip.stream()))
        run()
        torch.cuda.synchronize()
        out[name] = (f.clone(), events_ms(run))
    ref = out["fp32"][0].double()
    for name in ("f16x3", "fp16"):
        e = float((out[name][0].double() - ref).abs().max() / ref.abs().max())
        print(f"{attr} B={B} M={M}: {name} vs fp32 rel err {e:.2e}; ms fp32 {out['fp32'][1]:.3f} {name} {out[name][1]:.3f}")


def whole(kind, B, N):
    """the whole model: f16x3 vs fp32 (logits, masks, boxes) and their step times"""
    bench = importlib.import_module("bench")
    if kind == "static":
        model, inputs, _ = bench.make_static(B, N, torch.device("cuda"), 0)
    else:
        model, inputs = bench.make_dynamic(B, torch.device("cuda"), 0, n_per_frame=N)
    res = {}
    for prec in ("fp32", "f16x3"):
        model.precision = prec
        o = model._run(*inputs)
        torch.cuda.synchronize()
        res[prec] = ({k: o[k].clone() for k in ("logits", "mask", "boxes7")}, events_ms(lambda: model.refine(*inputs), 5))
    a, b = res["fp32"][0], res["f16x3"][0]
    lg = float((a["logits"].double() - b["logits"].double()).abs().max() / a["logits"].abs().max())
    flips = int((a["mask"] != b["mask"]).sum())
    same = (a["mask"] == b["mask"]).all(1)
    bx = float((a["boxes7"][same].double() - b["boxes7"][same].double()).abs().max())
    print(f"{kind} {B}x{N}: logits rel diff {lg:.2e}, mask flips {flips} of {a['mask'].numel()}, boxes abs diff (same-mask crops) {bx:.2e}; "
          f"step ms fp32 {res['fp32'][1]:.3f} f16x3 {res['f16x3'][1]:.3f}")


if __name__ == "__main__":
    head("static_one", "box_est", 3, 64, 512)
    head("static_one", "box_est", 3, 4096, 512)
    head("dynamic", "point_emb", 4, 1024, 2560)
    head("dynamic", "box_emb", 8, 1024, 101)
    whole("static", 64, 1024)
    whole("static", 4096, 1024)
    whole("dynamic", 256, 1024)
