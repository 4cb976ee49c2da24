#!/usr/bin/env python3
"""The file-level auto-labeling run (3dal_pytorch_amd/eval.py) on a synthetic Waymo-like segment in the reference's
pickle formats: 198 frames, 64 tracks (BASELINE.json C4's shape). Wall time of each stage of the device driver,
next to the reference's own formulation of `test_one_epoch` — its per-item Dataset through a DataLoader feeding
forward() batch by batch and decoding on the host (tools/static_eval.py:255-289, tools/dynamic_eval.py:213-245) —
run with this package's drop-in modules on the same GPU.
  python tools/bench_eval_files.py [--frames 198] [--tracks 64] [--batch 64]"""
import argparse
import importlib
import json
import os
import pickle
import sys
import tempfile
import time

import numpy as np
import torch
from torch.utils.data import DataLoader

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
synth = importlib.import_module("3dal_pytorch_amd.synth")
ev = importlib.import_module("3dal_pytorch_amd.eval")
static_model = importlib.import_module("3dal_pytorch_amd.static_model")
dynamic_model = importlib.import_module("3dal_pytorch_amd.dynamic_model")
datasets = importlib.import_module("3dal_pytorch_amd.datasets")


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return out, time.perf_counter() - t0


def loader_loop(head, model, track, infos, batch):
    """test_one_epoch as the reference writes it: Dataset.__getitem__ per item, collate, .cuda(), forward, host decode"""
    ds = datasets.STATICTRACK(track, infos) if head == "static" else datasets.DYNAMICTRACK(track, infos)
    n = 0
    for data in DataLoader(ds, batch_size=batch, shuffle=False):
        if head == "static":
            _, init_box, bbox_gt, pts = data[:4]
            out = model(pts.transpose(2, 1).float().cuda(), init_box.squeeze(1).float().cuda(), bbox_gt.float().cuda())
        else:
            _, init_box, bbox, bbox_gt, pts = data[:5]
            out = model(pts.transpose(2, 1).float().cuda(), bbox.transpose(2, 1).float().cuda(), bbox_gt.float().cuda())
        hs = out["heading_scores"].cpu().numpy()                       # the host decode's first step: D2H of the heads
        n += hs.shape[0]
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=198)
    ap.add_argument("--tracks", type=int, default=64)
    ap.add_argument("--batch", type=int, default=64)
    args = ap.parse_args()
    root = tempfile.mkdtemp(prefix="dal3_seg_")
    paths, tracks, poses, dets, has_gt = synth.segment_files(root, 90, n_frames=args.frames, n_tracks=args.tracks)
    out = {"segment": {"frames": args.frames, "tracks": args.tracks,
                       "track_frames": sum(len(t["token"]) for t in tracks),
                       "points": int(sum(len(p) for t in tracks for p in t["point"]))}}
    for head, kind, cls in (("static", "static_one", static_model.StaticModelOneBoxEst),
                            ("dynamic", "dynamic", dynamic_model.DynamicModel)):
        ckpt = os.path.join(root, kind + ".pth")
        sd = {k: torch.as_tensor(np.asarray(v)) for k, v in synth.state_dict(kind).items()}
        torch.save({"model_state_dict": sd}, ckpt)
        infos = ev.reorganize_info(pickle.load(open(paths["infos"], "rb")))
        res = {}
        for sampler in ("device", "numpy"):
            ev.run(head, paths[head], paths["infos"], paths["det_annos"], ckpt, batch_size=args.batch, sampler=sampler)
            _, dt = timed(lambda: ev.run(head, paths[head], paths["infos"], paths["det_annos"], ckpt,
                                         batch_size=args.batch, sampler=sampler))
            res[f"run[{sampler}]_s"] = round(dt, 3)
            # stages
            annos = ev.Annos(infos)
            track = pickle.load(open(paths[head], "rb"))
            det_annos = ev.sort_detections(pickle.load(open(paths["det_annos"], "rb")))
            _, t_idx = timed(lambda: ev.token_to_det_index(infos, det_annos, annos))
            if head == "static":
                track = ev.preprocessing(track, annos)
            model = cls(3, 3 if head == "static" else 4)
            model.load_state_dict(sd)
            model = model.cuda()
            refine = ev.refine_static_tracks if head == "static" else ev.refine_dynamic_tracks
            final, t_ref = timed(lambda: refine(model, track, annos, batch_size=args.batch, sampler=sampler))
            _, t_wb = timed(lambda: ev.write_back(track, annos, ev.token_to_det_index(infos, det_annos, annos), final,
                                                  det_annos, head == "static"))
            res[f"stages[{sampler}]_s"] = {"read_annos+index": round(t_idx, 3), "prepare+heads": round(t_ref, 3),
                                           "write_back": round(t_wb, 3), "items": int(final.shape[0])}
        _, dt = timed(lambda: ev.run(head, paths[head], paths["infos"], paths["det_annos"], ckpt, batch_size=args.batch,
                                     sampler="device", precision="bf16"))
        res["run[device,bf16]_s"] = round(dt, 3)
        _, dt = timed(lambda: ev.run(head, paths[head], paths["infos"], paths["det_annos"], ckpt, batch_size=args.batch,
                                     sampler="device", precision="f16x3"))
        res["run[device,f16x3]_s"] = round(dt, 3)
        track = pickle.load(open(paths[head], "rb"))
        if head == "static":
            track = ev.preprocessing(track, ev.Annos(infos))
        model = cls(3, 3 if head == "static" else 4)
        model.load_state_dict(sd)
        model = model.cuda().eval()
        model.sampler = "numpy"
        np.random.seed(ev.SEED)
        n, dt = timed(lambda: loader_loop(head, model, track, infos, args.batch))
        res["reference_formulation_test_one_epoch_s"] = round(dt, 3)
        res["items"] = n
        out[head] = res
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
