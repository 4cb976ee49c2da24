#!/usr/bin/env python3
"""Crop extraction at segment size: F frames x ~180k points x K detections.
  python tools/bench_crops.py [--frames 192] [--points 180000] [--boxes 60]
Prints one JSON line: device time of count+scan+fill (HIP events on the launch stream), the HBM roofline figure
(algorithmic bytes: every sweep point read once per pass = 2 x 12 B, members written once = 24 B + 4 B index),
the end-to-end call including the host set-up (face equations, concatenation, H2D), and the rate of the host
NumPy membership test."""
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
crops = importlib.import_module("3dal_pytorch_amd.crops")
geom = importlib.import_module("3dal_pytorch_amd.geom")
hip = importlib.import_module("3dal_pytorch_amd._hip")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=192)
    ap.add_argument("--points", type=int, default=180000)
    ap.add_argument("--boxes", type=int, default=60)
    ap.add_argument("--distinct", type=int, default=8, help="distinct synthetic frames (replicated to --frames)")
    ap.add_argument("--order", default="shuffled", choices=["shuffled", "range_image"],
                    help="point order inside a sweep: synth.sweep's random order (worst case for the cull: every 64-point "
                         "round touches every detection's neighbourhood) or beam-major / azimuth-minor like a lidar "
                         "range image (what real sweeps look like)")
    print(json.dumps(measure(ap.parse_args())))


def measure(args=None, **kw):
    """-> dict (what main() prints). args: a namespace with frames / points / boxes / distinct / order, or keyword overrides
    of the defaults (bench.py's `next_rows`)"""
    if args is None:
        args = argparse.Namespace(frames=192, points=180000, boxes=60, distinct=8, order="shuffled")
    for k, v in kw.items():
        setattr(args, k, v)
    base = [list(synth.sweep(46, f"b{f}", n_points=args.points, n_boxes=args.boxes)) for f in range(args.distinct)]
    if args.order == "range_image":
        for b in base:
            p = b[0]
            rng = np.linalg.norm(p[:, :2], axis=1)
            beam = np.clip(((np.arctan2(p[:, 2] - 2.0, rng) + 0.35) / 0.4 * 64).astype(np.int64), 0, 63)
            b[0] = np.ascontiguousarray(p[np.lexsort((np.arctan2(p[:, 1], p[:, 0]), beam))])
    sweeps = [torch.from_numpy(base[f % args.distinct][0]).cuda() for f in range(args.frames)]
    dets = [base[f % args.distinct][1] for f in range(args.frames)]
    poses = [base[f % args.distinct][4] for f in range(args.frames)]
    crops.extract_crops(sweeps[:2], dets[:2], poses[:2])                       # warm-up
    torch.cuda.synchronize()
    # (1) the reference-shaped call: a list of sweeps in, per-frame dicts of ragged rows out (concatenation of the
    # sweeps, the plan's host arithmetic and uploads, one host sync for the sizes)
    t0 = time.perf_counter()
    frames = crops.extract_crops(sweeps, dets, poses, return_index=True)
    torch.cuda.synchronize()
    t_call = time.perf_counter() - t0
    members = int(sum(fr["point"].counts().sum() for fr in frames))
    # (2) the planned call (crops.CropPlan): the segment's sweeps already ONE resident tensor, the plan built once;
    # run() = count + starts + fill on the stream, no host work in between; wall time of run() + one synchronise
    F, K = args.frames, args.frames * args.boxes
    d_pts = torch.cat(sweeps)
    n_pts = [int(s.shape[0]) for s in sweeps]
    t0 = time.perf_counter()
    plan = crops.CropPlan(n_pts, dets, poses, return_index=True)
    torch.cuda.synchronize()
    t_plan = time.perf_counter() - t0
    plan.run(d_pts)                                                             # sizes the output (one sync), warm-up
    assert plan.total() == members
    torch.cuda.synchronize()
    walls = []
    for _ in range(10):
        t0 = time.perf_counter()
        plan.run(d_pts)
        torch.cuda.synchronize()
        walls.append(time.perf_counter() - t0)
    t_run = sorted(walls)[len(walls) // 2]
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        plan.run(d_pts)
    b.record()
    b.synchronize()
    ms = a.elapsed_time(b) / 10

    def ev(fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            fn()
        b.record()
        b.synchronize()
        return a.elapsed_time(b) / 10
    ms_count, ms_fill = ev(lambda: plan.count(d_pts)), ev(lambda: plan.fill(d_pts))
    n_total = sum(n_pts)
    nbytes = 2 * 12 * n_total + 28 * members
    # the package's host NumPy membership test (datasets.points_in_rbbox, vectorised over the frame's boxes) on one frame
    datasets = importlib.import_module("3dal_pytorch_amd.datasets")
    boxes7 = crops.waymo_boxes(base[0][1])
    t1 = time.perf_counter()
    datasets.points_in_rbbox(base[0][0], boxes7)
    t_cpu = time.perf_counter() - t1
    return ({"workload": f"{F} frames x {args.points} pts x {args.boxes} detections, {args.order} point order",
                      "members": members,
                      "device_ms": round(ms, 3), "count_and_starts_ms": round(ms_count, 3), "fill_ms": round(ms_fill, 3), "frames_per_s_device": round(F / (ms * 1e-3), 1),
                      "point_box_tests_per_s": round(n_total * args.boxes / (ms * 1e-3) / 1e9, 1),
                      "roofline": {"bound": "hbm", "achieved": round(nbytes / (ms * 1e-3) / 1e9, 1), "peak": 8000.0,
                                   "unit": "GB/s", "frac": round(nbytes / (ms * 1e-3) / 1e9 / 8000.0, 4),
                                   "algorithmic_bytes": nbytes},
                      "planned_call_ms": round(t_run * 1e3, 3), "planned_call_over_kernels": round(t_run * 1e3 / ms, 2),
                      "plan_build_ms": round(t_plan * 1e3, 1),
                      "call_ms_with_host_setup": round(t_call * 1e3, 1),
                      "frames_per_s_call": round(F / t_call, 1),
                      "host_numpy_membership_frames_per_s": round(1.0 / t_cpu, 2)})


if __name__ == "__main__":
    main()
