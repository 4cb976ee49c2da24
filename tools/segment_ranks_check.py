#!/usr/bin/env python3
"""A rank of the sharded segment chain's self-check (started by tests/test_gpu_pipeline.py through launch.spawn_ranks, or
by `python -m torch.distributed.run --nproc-per-node W tools/segment_ranks_check.py`): every rank builds the same small
segment and plan, runs the chain once on its own (shard=False) and once sharded over the ranks (static tracks and dynamic
track-frames in contiguous ranges, one all-gather of the refined boxes per head), and compares the rewritten detections
bit for bit. DAL3_BENCH_BACKEND=gloo + DAL3_BENCH_SHARE_GPU=1: both ranks on one GPU, boxes staged through host memory."""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")))
    n_dev = torch.cuda.device_count()
    dev = torch.device("cuda", local % n_dev if os.environ.get("DAL3_BENCH_SHARE_GPU") == "1" else local)
    torch.cuda.set_device(dev)
    backend = os.environ.get("DAL3_BENCH_BACKEND", "nccl")
    torch.distributed.init_process_group(backend, rank=rank, world_size=world, **({"device_id": dev} if backend == "nccl" else {}))
    tp = importlib.import_module("test_gpu_pipeline")                      # the test's synthetic segment
    from _common import build_model, synth
    segment = importlib.import_module("3dal_pytorch_amd.segment")
    poses, sweeps, dets, gbox = tp._segment(n_frames=7, n_obj=7, seed=72)
    F, K = len(poses), gbox.shape[0]
    kinds = ["static", "dynamic", "static", "dynamic", "static", "static", "dynamic"]      # 4 static tracks, 21 dynamic items: ragged over 2 or 3 ranks
    scores = [[0.5 + 0.05 * ((f + k) % F) for f in range(F)] for k in range(K)]
    static = build_model("static_one", synth.state_dict("static_one"), device=dev)
    dynamic = build_model("dynamic", synth.state_dict("dynamic"), device=dev)
    plan = segment.SegmentPlan([s.shape[0] for s in sweeps], dets, poses,
                               [{"kind": kinds[k], "dets": [(f, k) for f in range(F)], "score": scores[k]} for k in range(K)],
                               static, dynamic, device=dev, n_static_points=1024, n_per_frame=256, dynamic_batch=4)
    d_pts = torch.from_numpy(np.concatenate(sweeps)).to(dev)
    plan.run(d_pts, shard=False)
    alone = {k: plan.detections(k) for k in ("static", "dynamic")}
    plan.run(d_pts)                                                         # sharded over the process group
    together = {k: plan.detections(k) for k in ("static", "dynamic")}
    ok = all(np.array_equal(alone[k][t], together[k][t]) for k in alone for t in alone[k])
    moved = sum(int((together["static"][t] != plan.crop.boxes[f]).any(1).sum()) for f, t in enumerate(plan.tokens))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
    print(f"segment rank {rank}/{world}: sharded == alone {ok}, static rows rewritten {moved}", flush=True)
    sys.exit(0 if ok and moved > 0 else 5)


if __name__ == "__main__":
    main()
