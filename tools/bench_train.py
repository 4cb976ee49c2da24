#!/usr/bin/env python3
"""One training step of the static head as static_train.py:76-86 runs it (forward, criterion, backward, Adam) at the
reference's batch (64 crops x 4096 points): per-point stacks on the HIP training kernels vs the stock-torch composite
on the same GPU.   python tools/bench_train.py [--batch 64] [--points 4096] [--kind static_one]"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
synth = importlib.import_module("3dal_pytorch_amd.synth")
arch = importlib.import_module("3dal_pytorch_amd.arch")
sm = importlib.import_module("3dal_pytorch_amd.static_model")
losses = importlib.import_module("3dal_pytorch_amd.losses")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--lib", default=None, help="another build of the library (variants/NAME.so): A/B runs of the step on one box")
    ap.add_argument("--points", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--backends", default="hip,torch")
    ap.add_argument("--adam", default="default", choices=["default", "fused", "capturable"],
                    help="default: torch.optim.Adam as static_train.py:220 builds it; fused: the same with fused=True; capturable: "
                         "with capturable=True (what a hipGraph capture needs: its step counter lives on the device)")
    ap.add_argument("--graph-optimizer", default="inside", choices=["inside", "outside"],
                    help="inside: Adam(capturable=True) captured with the step; outside: forward + backward captured, the plain Adam "
                         "stepping eagerly behind every replay (graph.CapturedTrainStep optimizer_in_graph=False)")
    ap.add_argument("--graph-only", action="store_true",
                    help="with --graph: skip the eager timed loop (a kernel trace of this run then holds the replays only)")
    ap.add_argument("--graph", action="store_true", help="also time the step captured into a hipGraph (device sampler)")
    ap.add_argument("--kind", default="static_one", choices=["static_one", "dynamic"],
                    help="static_train.py's StaticModelOneBoxEst (64 x 4096) or dynamic_train.py's DynamicModel (5 x 1024 points "
                         "+ 101 boxes per item)")
    ap.add_argument("--sampler", default="numpy", choices=["numpy", "device"],
                    help="object-point sampling in the train-mode forward: the reference's host loop or the GPU kernel")
    args = ap.parse_args()
    if args.lib:
        hip_mod = importlib.import_module("3dal_pytorch_amd._hip")
        hip_mod.LIB_PATH = os.path.abspath(args.lib)        # before the first hip.lib()
    B, N = args.batch, args.points
    dev = torch.device("cuda", 0)
    dynamic = args.kind == "dynamic"
    if dynamic:
        dm = importlib.import_module("3dal_pytorch_amd.dynamic_model")
        if args.points == 4096:
            args.points = 1024                                  # per frame: 5 x 1024 points per item
        p, bx, _, g = synth.dynamic_items(min(B, 16), n_per_frame=args.points, seed=3)
        reps = (B + p.shape[0] - 1) // p.shape[0]
        N = p.shape[1]
        pts = torch.from_numpy(np.tile(p, (reps, 1, 1))[:B]).to(dev).transpose(2, 1)
        init = torch.from_numpy(np.tile(bx, (reps, 1, 1))[:B]).to(dev).transpose(2, 1)      # the box sequence
        gt = torch.from_numpy(np.tile(g, (reps, 1))[:B]).to(dev)
    else:
        p, i, g = synth.static_crops(min(B, 64), N, seed=3)
        reps = (B + p.shape[0] - 1) // p.shape[0]
        pts = torch.from_numpy(np.tile(p, (reps, 1, 1))[:B]).to(dev).transpose(2, 1)
        init = torch.from_numpy(np.tile(i, (reps, 1))[:B]).to(dev)
        gt = torch.from_numpy(np.tile(g, (reps, 1))[:B]).to(dev)
    labels = ((torch.rand((B, N), device=dev) > 0.6).float(), torch.randn((B, 3), device=dev),
              torch.randint(0, 12, (B,), device=dev), 0.1 * torch.randn((B,), device=dev),
              torch.randint(0, 3, (B,), device=dev), 0.3 * torch.randn((B, 3), device=dev))
    crit = losses.DynamicModelLoss() if dynamic else losses.FrustumPointNetLossOneBoxEst()
    out = {"workload": (f"DynamicModel train step, {B} items x {N} pts + 101 boxes" if dynamic else
                        f"StaticModelOneBoxEst train step, {B} crops x {N} pts") + f", fp32, Adam, {args.sampler} sampler",
           "unit": "ms per step"}
    # algorithmic FLOP of the per-point stacks: forward + dgrad + wgrad = 3x forward (nominal formulation)
    mac_pt = sum(ci * co for _, _, ci, co in arch.ins_seg_layers(4 if dynamic else 3)) - 1024 * 512   # per-crop part of dconv1 excluded
    if dynamic:
        mac_obj = sum(ci * co for _, _, ci, co in arch.POINT_EMB["convs"])
        mac_box = sum(ci * co for _, _, ci, co in arch.BOX_EMB["convs"])
        flop = 3 * 2.0 * (B * N * mac_pt + B * 2560 * mac_obj + B * 101 * mac_box)
    else:
        mac_obj = sum(ci * co for _, _, ci, co in arch.STATIC_BOX_EST["convs"])
        flop = 3 * 2.0 * (B * N * mac_pt + B * 512 * mac_obj)
    for backend in args.backends.split(","):
        model = dm.DynamicModel() if dynamic else sm.StaticModelOneBoxEst()
        model.load_state_dict({k: torch.as_tensor(v) for k, v in synth.state_dict(args.kind).items()})
        model = model.to(dev).train()
        model.train_backend = "hip" if backend == "hip_f16x3" else backend
        if backend == "hip_f16x3":                              # the forward's big layers on the f16x3 training kernels
            model.precision = "f16x3"
        model.sampler = args.sampler
        # --adam default: what static_train.py:220 constructs (torch picks its multi-tensor "foreach" path: ~12 launches
        # per step over the 150 tensors); fused: torch.optim.Adam(..., fused=True), ONE multi-tensor launch — the same
        # update, the setting this path is measured and supported with when the optimizer's launches matter
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4, **({"fused": True} if args.adam == "fused" else {"capturable": True} if args.adam == "capturable" else {}))

        def step():
            o = model(pts, init, gt)
            loss = crit(o, *labels)["total_loss"]
            opt.zero_grad()
            loss.backward()
            opt.step()
            return loss
        np.random.seed(0)
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        # median of per-step times (HIP events around each step): a fresh box's first seconds run slower, and a mean
        # over five steps moved by 20 % between two runs of the same binary
        n_eager = 1 if (args.graph and args.graph_only) else args.iters
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(n_eager + 1)]
        marks[0].record()
        t_host = time.perf_counter()
        for i in range(n_eager):
            loss = step()
            marks[i + 1].record()
        host_ms = (time.perf_counter() - t_host) / n_eager * 1e3         # the host's time to ISSUE a step (no sync inside)
        torch.cuda.synchronize()
        per = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(n_eager))
        ms = per[len(per) // 2]
        g_ms = None
        if args.graph and args.sampler == "device":
            # a FRESH model: a parameter that has already taken part in an eager backward keeps a gradient accumulator
            # bound to that stream, which a capture on another stream cannot use
            graph = importlib.import_module("3dal_pytorch_amd.graph")
            del model, opt
            gm = dm.DynamicModel() if dynamic else sm.StaticModelOneBoxEst()
            gm.load_state_dict({k: torch.as_tensor(v) for k, v in synth.state_dict(args.kind).items()})
            gm = gm.to(dev).train()
            gm.train_backend, gm.sampler = ("hip" if backend == "hip_f16x3" else backend), "device"
            if backend == "hip_f16x3":
                gm.precision = "f16x3"
            inside = args.graph_optimizer == "inside"
            gopt = torch.optim.Adam(gm.parameters(), lr=1e-3, weight_decay=1e-4, **({"capturable": True} if inside else {}))
            cap = graph.CapturedTrainStep(gm, gopt, lambda p_, i_, g_: crit(gm(p_, i_, g_), *labels)["total_loss"],
                                          pts, init, gt, optimizer_in_graph=inside)
            for _ in range(2):
                cap(pts, init, gt)
            torch.cuda.synchronize()
            gm_marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.iters + 1)]
            t0 = time.perf_counter()
            gm_marks[0].record()
            for i in range(args.iters):
                cap(pts, init, gt)
                gm_marks[i + 1].record()
            g_issue = (time.perf_counter() - t0) / args.iters * 1e3
            torch.cuda.synchronize()
            g_ms = round((time.perf_counter() - t0) / args.iters * 1e3, 2)
            g_per = sorted(gm_marks[i].elapsed_time(gm_marks[i + 1]) for i in range(args.iters))
            g_extra = {"graph_ms_median_of_events": round(g_per[len(g_per) // 2], 2), "graph_host_issue_ms": round(g_issue, 3),
                       "graph_optimizer": args.graph_optimizer}
            model, opt = gm, gopt
        out[backend] = {"ms": round(ms, 2), "host_issue_ms": round(host_ms, 2), "graph_ms": g_ms, "crops_per_s": round(B / ms * 1e3, 1), "tflops_per_point_stacks": round(flop / ms / 1e9, 1),
                        "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2**30, 2), "loss": round(float(loss), 4),
                        "adam": args.adam}
        if g_ms is not None:
            out[backend].update(g_extra)
        del model, opt
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
    if "torch" in out and "hip" in out:
        out["speedup"] = round(out["torch"]["ms"] / out["hip"]["ms"], 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
