#!/usr/bin/env python3
"""Crop preparation (N1) and write-back (N3) at batch size: wall time of the Python call (host flattening + H2D +
kernel) and the kernel alone (HIP events around a second call on already-uploaded inputs is not exposed by the
host API, so the kernel figure comes from torch's profiler-free estimate: total call minus a host-only dry run is
not reliable either) — reported: the call, and the achieved bytes/s of the gather against the HBM roofline computed
from the kernel time measured with rocprofv3 (`--kernel-trace`) when run under it.
  python tools/bench_prep_post.py [--tracks 1024]"""
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
prep = importlib.import_module("3dal_pytorch_amd.prep")
post = importlib.import_module("3dal_pytorch_amd.post")


def _events_ms(fn, iters=10, warmup=2):
    for _ in range(warmup):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / iters


def measure(B=1024):
    """-> dict (what main() prints): crop preparation (N1) and write-back (N3) at batch size; bench.py's `next_rows` reads
    the StaticTrackStore / WritebackPlan entries"""
    base = [synth.track(80, t, n_frames=8 + t % 5) for t in range(64)]
    tracks = [base[i % 64] for i in range(B)]
    poses = [synth.pose_veh_to_global(80, tr["token"][int(np.argmax(tr["score"]))]) for tr in tracks]
    out = {}
    for sampler in ("device", "numpy"):
        prep.prepare_static_batch(tracks[:8], poses[:8], n_points=4096, sampler=sampler)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pts, init = prep.prepare_static_batch(tracks, poses, n_points=4096, sampler=sampler)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[f"prepare_static_batch[{sampler}]"] = {"tracks": B, "points_out": B * 4096, "call_ms": round(dt * 1e3, 1),
                                                   "crops_per_s": round(B / dt, 1)}
    # the segment's tracks flattened and uploaded once (StaticTrackStore), then prepared in the drivers' batches of 64
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    store = prep.StaticTrackStore(tracks)
    torch.cuda.synchronize()
    t_store = time.perf_counter() - t0
    prep.prepare_static_batch(store, poses[:64], n_points=4096, sampler="device", first=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(0, B, 64):
        prep.prepare_static_batch(store, poses[k:k + 64], n_points=4096, sampler="device", first=k, item_offset=k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["prepare_static_batch[device, StaticTrackStore, batches of 64]"] = {
        "tracks": B, "store_build_ms": round(t_store * 1e3, 1), "all_batches_ms": round(dt * 1e3, 2),
        "ms_per_batch_of_64": round(dt * 1e3 / max(B // 64, 1), 3), "crops_per_s_excluding_the_build": round(B / dt, 1),
        "crops_per_s_including_the_build": round(B / (dt + t_store), 1)}
    items = [(i % 64, j) for i in range(64) for j in range(len(base[i]["token"]))][:B]
    dposes = [synth.pose_veh_to_global(80, base[t]["token"][j]) for t, j in items]
    prep.prepare_dynamic_batch(base, items[:4], dposes[:4], sampler="device")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    prep.prepare_dynamic_batch(base, items, dposes, sampler="device")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["prepare_dynamic_batch[device]"] = {"items": len(items), "call_ms": round(dt * 1e3, 1), "items_per_s": round(len(items) / dt, 1)}
    # N1 on the device alone: one batch of 64 crops x 4096 points from the resident store (HIP events over 10 calls).
    # Algorithmic bytes per crop: 4096 drawn points read as float64 xyz (24 B, a gather) and written as fp32 xyz (12 B).
    ms4 = _events_ms(lambda: prep.prepare_static_batch(store, poses[:64], n_points=4096, sampler="device", first=0))
    # round 5: a store that holds its tracks' best-frame poses -> the batch is ONE launch (no host arithmetic, no upload)
    store_p = prep.StaticTrackStore(tracks, veh_to_global=poses)
    ms = _events_ms(lambda: prep.prepare_static_batch(store_p, 64, n_points=4096, sampler="device", first=0))
    t0 = time.perf_counter()
    for _ in range(20):
        prep.prepare_static_batch(store_p, 64, n_points=4096, sampler="device", first=0)
    torch.cuda.synchronize()
    call_ms = (time.perf_counter() - t0) / 20 * 1e3
    nbytes = 64 * 4096 * 36
    out["N1_device_batch_of_64"] = {"stream_ms_per_call": round(ms, 4), "algorithmic_bytes": nbytes,
                                    "gb_per_s": round(nbytes / ms / 1e6, 1), "frac_of_8TBps": round(nbytes / ms / 1e6 / 8000.0, 4),
                                    "whole_call_ms": round(call_ms, 4), "stream_ms_with_per_batch_uploads": round(ms4, 4),
                                    "note": "one launch per batch from a store that holds its poses (latency-bound: 64 crops, a gather of 24-B "
                                            "rows from a ragged store); with per-batch pose uploads it is four launches"}
    tr_l, poses_s, dets_s, has_gt = synth.scene(81, n_frames=198, n_tracks=64)
    final = torch.randn((len(tr_l), 7), dtype=torch.float64, device="cuda")
    post.writeback_static(tr_l, poses_s, has_gt, final, dets_s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    post.writeback_static(tr_l, poses_s, has_gt, final, dets_s)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pairs = sum(len(t["token"]) for t in tr_l)
    out["writeback_static"] = {"frames": 198, "tracks": len(tr_l), "pairs": pairs, "call_ms": round(dt * 1e3, 1)}
    # the same with the segment flattened once (post.WritebackPlan): build, then the call that follows the heads
    t0 = time.perf_counter()
    plan = post.WritebackPlan(tr_l, poses_s, has_gt, dets_s, static=True)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    plan.apply(final)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    plan.apply(final)
    torch.cuda.synchronize()
    t_apply = time.perf_counter() - t0
    ms = _events_ms(lambda: plan.launch(final))
    n_det = plan.n_det
    # algorithmic bytes: per (track, frame) pair its pose products (2 x 128 B) + box (56 B) + the frame's detection centres
    # scanned (12 B each, ~n_det / frames per pair) + one 28-B row rewritten; the detection array copied once (2 x 28 B a row)
    nbytes = pairs * (256 + 56 + 28 + 12 * (n_det // 198)) + n_det * 56
    out["writeback_static[WritebackPlan]"] = {"pairs": pairs, "detections": n_det, "plan_build_ms": round(t_build * 1e3, 2),
                                              "apply_call_ms": round(t_apply * 1e3, 3), "stream_ms_per_launch": round(ms, 4),
                                              "algorithmic_bytes": nbytes, "gb_per_s": round(nbytes / ms / 1e6, 1),
                                              "frac_of_8TBps": round(nbytes / ms / 1e6 / 8000.0, 5),
                                              "note": "latency-bound: 6k pairs, two launches"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tracks", type=int, default=1024)
    args = ap.parse_args()
    print(json.dumps(measure(args.tracks)))


if __name__ == "__main__":
    main()
