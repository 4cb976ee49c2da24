#!/usr/bin/env python3
"""One synthetic Waymo-sized segment from its sweeps to the rewritten detections, chained on the device
(3dal_pytorch_amd/segment.py: N2 crop extraction -> N1 crop preparation -> the heads -> N3 write-back, one stream, inputs
resident, ONE synchronisation at the end) — VERDICT r4 #5.

  python tools/bench_pipeline.py [--frames 198] [--points 180000] [--static 64] [--dynamic 40] [--precision fp32] [--iters 5]

The segment is BASELINE.json configs[3]'s: 198 frames; 64 static tracks (every frame) and 40 dynamic tracks of
rng.integers(20,199) frames (4,531 track-frames), i.e. 104 detections in a frame where every track is present; sweeps of
180,000 points in range-image order (beam-major, azimuth-minor), objects consistent across frames (static ones fixed
in the global frame, dynamic ones at constant velocity, the ego vehicle moving).

Prints one JSON object:
  wall_ms        host clock from the first enqueue to the end of the closing synchronise (median of --iters runs)
  issue_ms       host clock until the last enqueue returned (the host's own work per run)
  device_ms      the same chain's duration when the host is OUT of the way: enqueued behind a 60-ms spin kernel, so every
                 launch is already queued when the GPU gets to it — HIP events around the chain
  host_share     (wall_ms - device_ms) / wall_ms: the part of the wall time the GPU spent waiting for the host
  stages         HIP-event time per stage in the normal run (crops | static prep, heads, write-back | dynamic ...)
  items_per_s    (static tracks + dynamic track-frames) / wall
  plan_build_ms  SegmentPlan.__init__ (per-box host arithmetic, uploads): once per segment, before the sweeps are needed
"""
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
segment = importlib.import_module("3dal_pytorch_amd.segment")
sm = importlib.import_module("3dal_pytorch_amd.static_model")
dm = importlib.import_module("3dal_pytorch_amd.dynamic_model")


def make_segment(F, P, n_static, n_dynamic, seed=10922081, order="range_image"):
    """-> (sweeps: list of (P,3) f32 in the vehicle frame, dets: list of (K_f,9) f32 detector-convention boxes, poses:
    list of flat-16 veh_to_global, tracks: [{"kind", "dets": [(frame, k)], "score"}])"""
    rng = np.random.default_rng(seed)
    n_obj = n_static + n_dynamic
    lens = np.random.default_rng(10922081).integers(20, 199, size=n_dynamic)       # bench.py's C4 (c4_segment_sizes)
    lens = np.minimum(lens, F)
    first = np.concatenate([np.zeros(n_static, np.int64), rng.integers(0, F - lens + 1)])
    last = np.concatenate([np.full(n_static, F), first[n_static:] + lens])
    size = np.array(arch.MEAN_SIZE)[np.arange(n_obj) % 3] + rng.normal(0, 0.1, (n_obj, 3))
    c0 = np.concatenate([rng.uniform(-55, 55, (n_obj, 2)), rng.uniform(0.2, 1.2, (n_obj, 1))], 1) + [2.0e4, -1.5e4, 30.0]
    yaw0 = rng.uniform(-np.pi, np.pi, n_obj)
    vel = np.concatenate([np.zeros((n_static, 2)), rng.normal(0, 0.4, (n_dynamic, 2))])      # m per frame
    sweeps, dets, poses = [], [], []
    tracks = [{"kind": "static" if k < n_static else "dynamic", "dets": [], "score": []} for k in range(n_obj)]
    per_obj = (P // 3) // n_obj
    for f in range(F):
        eyaw, et = 0.2 + 0.004 * f, np.array([2.0e4 + 0.3 * f, -1.5e4 + 0.12 * f, 30.0])
        c, s = np.cos(eyaw), np.sin(eyaw)
        pose = np.array([[c, -s, 0, et[0]], [s, c, 0, et[1]], [0, 0, 1, et[2]], [0, 0, 0, 1.0]])
        live = np.nonzero((first <= f) & (f < last))[0]
        ctr_g = c0[live] + np.concatenate([vel[live] * f, np.zeros((len(live), 1))], 1)
        ctr = (ctr_g - et) @ pose[:3, :3]                                              # R^T (c - t): vehicle frame
        yaw_v = yaw0[live] - eyaw
        # detector convention [x,y,z,w,l,h,vx,vy,r2], r2 = -yaw - pi/2 (crops.waymo_boxes inverts it)
        dets.append(np.concatenate([ctr, size[live][:, [1, 0, 2]], np.zeros((len(live), 2)), (-yaw_v - np.pi / 2)[:, None]],
                                   1).astype(np.float32))
        for j, k in enumerate(live):
            tracks[k]["dets"].append((f, j))
            tracks[k]["score"].append(float(0.3 + 0.6 * rng.random()))
        owner = np.repeat(np.arange(len(live)), per_obj)
        loc = rng.uniform(-0.62, 0.62, (len(owner), 3)) * size[live][owner]           # 1.24x the box: in and out
        # det3d's corner routine turns a box clockwise by its yaw (tests/test_gpu_pipeline.py): object points go there
        cy, sy = np.cos(-yaw_v[owner]), np.sin(-yaw_v[owner])
        obj = np.stack([cy * loc[:, 0] - sy * loc[:, 1], sy * loc[:, 0] + cy * loc[:, 1], loc[:, 2]], 1) + ctr[owner]
        clutter = rng.uniform(-75, 75, (P - len(owner), 3)) * [1.0, 1.0, 0.03]
        pts = np.concatenate([obj, clutter]).astype(np.float32)
        if order == "range_image":
            r = np.linalg.norm(pts[:, :2], axis=1)
            beam = np.clip(((np.arctan2(pts[:, 2] - 2.0, r) + 0.35) / 0.4 * 64).astype(np.int64), 0, 63)
            pts = np.ascontiguousarray(pts[np.lexsort((np.arctan2(pts[:, 1], pts[:, 0]), beam))])
        else:
            pts = pts[rng.permutation(P)]
        sweeps.append(pts)
        poses.append(pose.reshape(16))
    return sweeps, dets, poses, tracks


def build_models(precision, dev):
    static = sm.StaticModelOneBoxEst()
    static.load_state_dict({k: torch.as_tensor(v) for k, v in synth.state_dict("static_one").items()})
    dynamic = dm.DynamicModel()
    dynamic.load_state_dict({k: torch.as_tensor(v) for k, v in synth.state_dict("dynamic").items()})
    static, dynamic = static.to(dev).eval(), dynamic.to(dev).eval()
    static.precision = dynamic.precision = precision
    return static, dynamic


def measure(frames=198, points=180000, n_static=64, n_dynamic=40, precision="fp32", iters=5, order="range_image"):
    dev = torch.device("cuda", 0)
    sweeps, dets, poses, tracks = make_segment(frames, points, n_static, n_dynamic, order=order)
    d_pts = torch.cat([torch.from_numpy(s).to(dev) for s in sweeps])
    static, dynamic = build_models(precision, dev)
    t0 = time.perf_counter()
    plan = segment.SegmentPlan([s.shape[0] for s in sweeps], dets, poses, tracks, static, dynamic)
    torch.cuda.synchronize()
    t_plan = time.perf_counter() - t0
    plan.run(d_pts)                                          # sizes the crop buffer (the one read-back), warms everything
    torch.cuda.synchronize()
    members = plan.crop.total()
    assert not plan.overflowed()
    plan.run(d_pts)
    torch.cuda.synchronize()
    walls, issues, stage_ms = [], [], {}
    for _ in range(iters):
        evs = [("start", torch.cuda.Event(enable_timing=True))]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        evs[0][1].record()

        def mark(name):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            evs.append((name, e))
        plan.run(d_pts, marks=mark)
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        walls.append(time.perf_counter() - t0)
        issues.append(t_issue)
        for (_, a), (name, b) in zip(evs[:-1], evs[1:]):
            stage_ms.setdefault(name, []).append(a.elapsed_time(b))
    assert not plan.overflowed()
    # the same chain with the host out of the way: everything is queued while a spin kernel holds the stream
    devs = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        torch.cuda._sleep(int(60e-3 * 2.0e9))                # ~60 ms of spinning (clock64 ticks); the chain queues behind it
        a.record()
        plan.run(d_pts)
        b.record()
        b.synchronize()
        devs.append(a.elapsed_time(b))
    med = lambda v: sorted(v)[len(v) // 2]                  # noqa: E731
    wall, dev_ms = med(walls) * 1e3, med(devs)
    n_items = plan.S + plan.D
    m = plan.wb_static.match.cpu().numpy()
    return {"workload": f"{frames} frames x {points} pts ({order} order), {n_static} static tracks + {n_dynamic} dynamic tracks "
                        f"({plan.D} track-frames), {sum(len(d) for d in dets)} detections, {precision} heads",
            "crop_rows": members, "static_tracks": plan.S, "dynamic_items": plan.D,
            "wall_ms": round(wall, 3), "issue_ms": round(med(issues) * 1e3, 3), "device_ms": round(dev_ms, 3),
            "host_share": round(max(0.0, wall - dev_ms) / wall, 4),
            "stages_ms": {k: round(med(v), 3) for k, v in stage_ms.items()},
            "items_per_s": round(n_items / (wall * 1e-3), 1), "frames_per_s": round(frames / (wall * 1e-3), 1),
            "plan_build_ms": round(t_plan * 1e3, 1), "static_pairs_matched": int((m >= 0).sum()), "static_pairs": int(m.size),
            "iters": iters}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=198)
    ap.add_argument("--points", type=int, default=180000)
    ap.add_argument("--static", type=int, default=64)
    ap.add_argument("--dynamic", type=int, default=40)
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16", "fp16", "f16x3"])
    ap.add_argument("--order", default="range_image", choices=["range_image", "shuffled"])
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    print(json.dumps(measure(a.frames, a.points, a.static, a.dynamic, a.precision, a.iters, a.order)))


if __name__ == "__main__":
    main()
