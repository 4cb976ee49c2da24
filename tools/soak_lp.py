#!/usr/bin/env python3
"""Soak test of the 16-bit and f16x3 kernels (persistent workgroups, raw barriers, LDS-DMA ring, hand-written MFMA statements):
many launches at several shapes and both 16-bit types, every output compared bit for bit with the first launch of its
shape. A race or a missing wait state shows up as a rare mismatch.   python tools/soak_lp.py [--launches 300]"""
import argparse
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
synth = importlib.import_module("3dal_pytorch_amd.synth")
sm = importlib.import_module("3dal_pytorch_amd.static_model")
dm = importlib.import_module("3dal_pytorch_amd.dynamic_model")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--launches", type=int, default=300)
    ap.add_argument("--precisions", default="bf16,fp16", help="comma list of bf16, fp16, f16x3 (the split-fp16 family: inline-asm "
                                                               "MFMAs with VGPR accumulators, two-slot / three-slot rings)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    cases = []
    for kind, B, N in (("static_one", 2048, 1024), ("static_two", 300, 700), ("static_one", 64, 4096), ("static_one", 5, 33)):
        pts, init, gt = synth.static_crops(min(B, 256), N, seed=21)
        reps = (B + pts.shape[0] - 1) // pts.shape[0]
        t = lambda a: torch.from_numpy(np.tile(a, (reps,) + (1,) * (a.ndim - 1))[:B]).to(dev)
        cls = sm.StaticModelOneBoxEst if kind == "static_one" else sm.StaticModelTwoBoxEst
        m = cls()
        m.load_state_dict({k: torch.as_tensor(v) for k, v in synth.state_dict(kind, seed=21).items()})
        cases.append((f"{kind} {B}x{N}", m.to(dev).eval(), (t(pts).transpose(2, 1), t(init), t(gt))))
    p, b, i8, _ = synth.dynamic_items(48, seed=21)
    m = dm.DynamicModel()
    m.load_state_dict({k: torch.as_tensor(v) for k, v in synth.state_dict("dynamic", seed=21).items()})
    cases.append(("dynamic 192x5120", m.to(dev).eval(),
                  (torch.from_numpy(np.tile(p, (4, 1, 1))).to(dev).transpose(2, 1),
                   torch.from_numpy(np.tile(b, (4, 1, 1))).to(dev).transpose(2, 1), torch.from_numpy(np.tile(i8, (4, 1))).to(dev))))
    bad = 0
    t0 = time.time()
    for prec in args.precisions.split(","):
        for name, model, inputs in cases:
            model.precision = prec
            keys = ("logits", "mask", "counts", "obj_idx", "boxes7")
            first = model._run(*inputs)
            ref = {k: first[k].clone() for k in keys}
            n_bad = 0
            for _ in range(args.launches):
                out = model._run(*inputs)
                n_bad += sum(0 if torch.equal(out[k], ref[k]) else 1 for k in keys)
            bad += n_bad
            print(f"{prec} {name}: {args.launches} launches, {n_bad} mismatching outputs", flush=True)
    print(f"total mismatches {bad}  ({time.time() - t0:.0f} s)")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
