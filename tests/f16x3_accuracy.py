#!/usr/bin/env python3
"""What a 16-bit MFMA path costs in accuracy, and what a SPLIT one does not (CPU, numpy / torch emulation).

ins_seg's folded layers are applied to synthetic crops with each product formed the way a given arithmetic would form
it — operands rounded to the operand type, products exact, fp32-like accumulation — and the logits compared with a float64
run:   f32     plain float32 (what the fp32-MFMA kernels do)
       f16x3   x = x_hi + x_lo, w = w_hi + w_lo in fp16;  w_hi x_hi + w_hi x_lo + w_lo x_hi  (three fp16 MFMAs per fp32 one)
       bf16x3  the same split in bf16
       f16 / bf16   operands rounded once (the 16-bit kernels of configs C3 / C5)
The first layer (raw coordinates) and the per-crop dconv1 term stay fp32, as in the kernels.
Result (8 crops x 1024 points, max |logit error| / max |logit|):  f32 2.0e-6 | f16x3 1.0e-6 | bf16x3 5.2e-5 | f16 2.7e-3 |
bf16 2.6e-2  ->  the fp16 split is as exact as fp32 arithmetic itself; at 16x the MFMA rate for 3x the MFMAs it is what
dal3_pointmlp_x3.hip builds on. Lives under tests/ because it uses the oracle package (test infrastructure
only); tests/test_host_cpu.py runs it at a small size."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _common import recentred_sd, synth                     # noqa: E402
from oracle import ref_heads as R                           # noqa: E402


def main(B=8, N=1024):
    torch.set_num_threads(8)
    pts_np, _, _ = synth.static_crops(B, N, seed=5)
    t = R.as_torch_sd(recentred_sd("static_one", pts_np[:2], seed=5))

    def fold(conv, bn):
        W, b = t[f"ins_seg.{conv}.weight"].double().squeeze(-1), t[f"ins_seg.{conv}.bias"].double()
        if bn:
            g, be = t[f"ins_seg.{bn}.weight"].double(), t[f"ins_seg.{bn}.bias"].double()
            m, v = t[f"ins_seg.{bn}.running_mean"].double(), t[f"ins_seg.{bn}.running_var"].double()
            s = g / torch.sqrt(v + 1e-5)
            W, b = W * s[:, None], (b - m) * s + be
        return W, b
    names = [("conv1", "bn1"), ("conv2", "bn2"), ("conv3", "bn3"), ("conv4", "bn4"), ("conv5", "bn5"), ("dconv1", "dbn1"),
             ("dconv2", "dbn2"), ("dconv3", "dbn3"), ("dconv4", "dbn4"), ("dconv5", None)]
    Wb = [fold(*n) for n in names]
    x0 = torch.from_numpy(pts_np).double().reshape(B * N, 3)

    def split(a, dt):
        hi = a.to(dt)
        return hi.double(), (a - hi.to(a.dtype)).to(dt).double()

    def mm(a, W, mode):
        if mode == "f64":
            return a @ W.t()
        if mode == "f32":
            return (a.float() @ W.float().t()).double()
        if mode in ("f16", "bf16"):
            dt = torch.float16 if mode == "f16" else torch.bfloat16
            return (a.to(dt).double() @ W.to(dt).double().t()).float().double()
        dt = torch.float16 if mode == "f16x3" else torch.bfloat16
        ah, al = split(a.float(), dt)
        wh, wl = split(W.float(), dt)
        return (ah @ wh.t() + ah @ wl.t() + al @ wh.t()).float().double()

    def net(mode):
        a, outs = x0, []
        for i in range(5):
            W, b = Wb[i]
            a = torch.relu(mm(a, W, "f32" if (i == 0 and mode != "f64") else mode) + b)
            outs.append(a)
        g = a.reshape(B, N, 1024).amax(1)
        W, b = Wb[5]
        per = (mm(g, W[:, 64:], "f32" if mode != "f64" else "f64") + b).repeat_interleave(N, 0)
        a = torch.relu(mm(outs[1], W[:, :64], mode) + per)
        for i in range(6, 9):
            W, b = Wb[i]
            a = torch.relu(mm(a, W, mode) + b)
        W, b = Wb[9]
        return mm(a, W, mode) + b
    ref = net("f64")
    out = {}
    for m in ("f32", "f16x3", "bf16x3", "f16", "bf16"):
        out[m] = float((net(m) - ref).abs().max() / ref.abs().max())
        print(f"{m:7s} max |logit error| / max |logit| = {out[m]:.2e}")
    return out


if __name__ == "__main__":
    main()
