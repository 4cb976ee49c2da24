#!/usr/bin/env python3
"""Latency of one refine() call at small batches: eager (Python + ctypes launches) vs hipGraph replay.
  python tools/bench_latency.py [--points 1024] [--head static|dynamic]
Prints one JSON line with, per batch size, the median wall time per call (synchronised) and the GPU time."""
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
bench = importlib.import_module("bench")
graph = importlib.import_module("3dal_pytorch_amd.graph")


def timed(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    wall = []
    for _ in range(iters):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        wall.append(time.perf_counter() - t0)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    b.synchronize()
    return float(np.median(wall)) * 1e6, a.elapsed_time(b) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=1024)
    ap.add_argument("--head", default="static", choices=["static", "dynamic"])
    ap.add_argument("--precision", default="fp32")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    out = {"head": args.head, "points": args.points, "precision": args.precision, "unit": "us per call", "batches": {}}
    for B in (1, 4, 16, 64, 256):
        if args.head == "static":
            model, inputs, _ = bench.make_static(B, args.points, dev, 0)
        else:
            model, inputs = bench.make_dynamic(B, dev, 0)
        model.precision = args.precision
        eager_wall, eager_gpu = timed(lambda: model.refine(*inputs))
        cap = graph.CapturedRefine(model, *inputs)
        assert torch.equal(cap(*inputs), model.refine(*inputs))
        g_wall, g_gpu = timed(lambda: cap(*inputs))
        replay_wall, replay_gpu = timed(cap.graph.replay)
        out["batches"][B] = {"eager_wall": round(eager_wall, 1), "eager_stream": round(eager_gpu, 1),
                             "graph_wall": round(g_wall, 1), "graph_stream": round(g_gpu, 1),
                             "replay_only_wall": round(replay_wall, 1), "replay_only_stream": round(replay_gpu, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
