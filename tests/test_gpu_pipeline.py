"""The widened path end to end on one synthetic segment, every step on the device:
crop extraction from sweeps (N2) -> track dictionaries -> crop preparation (N1) -> static head -> write-back into
the per-frame detections (N3). What is checked is that the pieces COMPOSE — frames, conventions and schemas of
SURVEY.md 8(g) line up — each step's own parity is covered by its own tests. (The association of detections into
tracks is the reference tracker's job: here detection k of every frame is object k.)"""
import importlib

import numpy as np
import pytest
import torch

from _common import build_model, synth

crops = importlib.import_module("3dal_pytorch_amd.crops")
prep = importlib.import_module("3dal_pytorch_amd.prep")
post = importlib.import_module("3dal_pytorch_amd.post")
pytestmark = pytest.mark.gpu


def _segment(n_frames=6, n_obj=5, seed=70):
    poses, sweeps, dets = [], [], []
    gbox = np.concatenate([synth.uniform(seed, "c", (n_obj, 3), -30, 30) * [1, 1, 0.02] + [2.0e4, -1.5e4, 30.0],
                           np.array(synth.arch.MEAN_SIZE)[np.arange(n_obj) % 3],
                           synth.uniform(seed, "y", (n_obj, 1), -3, 3)], 1)              # static objects, global frame
    for f in range(n_frames):
        yaw, t = 0.2 + 0.03 * f, np.array([2.0e4 + 1.2 * f, -1.5e4 + 0.5 * f, 30.0])
        c, s = np.cos(yaw), np.sin(yaw)
        pose = np.array([[c, -s, 0, t[0]], [s, c, 0, t[1]], [0, 0, 1, t[2]], [0, 0, 0, 1.0]])
        poses.append(pose.reshape(16))
        ctr = (gbox[:, :3] - t) @ pose[:3, :3]                                          # R^T (c - t)
        yaw_v = gbox[:, 6] - yaw
        # detector convention: [x,y,z,w,l,h,vx,vy,r2], r2 = -yaw - pi/2 (crops.waymo_boxes inverts this)
        dets.append(np.concatenate([ctr, gbox[:, [4, 3, 5]], np.zeros((n_obj, 2)), (-yaw_v - np.pi / 2)[:, None]],
                                   1).astype(np.float32))
        pts = [synth.uniform(seed, f"clutter{f}", (4000, 3), -60, 60) * [1, 1, 0.05]]
        for k in range(n_obj):
            loc = synth.uniform(seed, f"p{f}_{k}", (150 + 20 * k, 3), -0.48, 0.48) * gbox[k, 3:6]
            # det3d's corner routine turns a box CLOCKWISE by its yaw (box_np_ops.py:146-178 applies p.R with
            # R = [[c,-s],[s,c]]), and the reference hands it Waymo-convention (counter-clockwise) yaws: the region
            # its crop extraction (and its mask labels) actually test is the box mirrored in yaw. Kept, like every
            # other quirk; the synthetic object points are placed in that region.
            cy, sy = np.cos(-yaw_v[k]), np.sin(-yaw_v[k])
            pts.append(np.stack([cy * loc[:, 0] - sy * loc[:, 1], sy * loc[:, 0] + cy * loc[:, 1], loc[:, 2]], 1) + ctr[k])
        sweeps.append(np.concatenate(pts).astype(np.float32))
    return poses, sweeps, dets, gbox


def test_extract_prepare_refine_write_back():
    poses, sweeps, dets, gbox = _segment()
    F, K = len(poses), gbox.shape[0]
    tokens = [f"fr{f}" for f in range(F)]
    frames = crops.extract_crops(sweeps, dets, poses)                                   # N2
    for rec in frames:
        assert np.abs(rec["bbox"][:, :3] - gbox[:, :3]).max() < 5e-3                     # detections land on the objects
        assert all(p.shape[0] >= 150 for p in rec["point"])
    tracks = []
    for k in range(K):                                                                   # trackStatic.pkl schema
        tracks.append({"bbox": [frames[f]["bbox"][k] for f in range(F)],
                       "point": [frames[f]["point"][k].cpu().numpy() for f in range(F)],
                       "score": [0.5 + 0.05 * ((f + k) % F) for f in range(F)], "token": tokens,
                       "match": [f"gt{k}"] * F, "type": [1] * F})
    best = [int(np.argmax(t["score"])) for t in tracks]
    pts, init = prep.prepare_static_batch(tracks, [poses[b] for b in best], n_points=1024, sampler="device")   # N1
    assert float(pts.abs().max()) < 8.0                                                  # box frame: metres, not km
    model = build_model("static_one", synth.state_dict("static_one"))
    boxes = model.refine(pts, init)                                                      # the heads
    det_rows = {tokens[f]: frames[f]["boxes_lidar"].copy() for f in range(F)}
    v2g = {tokens[f]: poses[f] for f in range(F)}
    has_gt = {(k, t): True for k in range(K) for t in tokens}
    new, match = post.writeback_static(tracks, v2g, has_gt, boxes, det_rows)             # N3
    assert (match.reshape(K, F) == np.arange(K)[:, None]).all()                          # object k is row k of every frame
    refined = boxes.double().cpu().numpy()
    for k in range(K):
        m_best = np.reshape(poses[best[k]], [4, 4])
        want_g = m_best[:3, :3] @ refined[k, :3] + m_best[:3, 3]                         # refined centre, global frame
        for f, t in enumerate(tokens):
            m = np.reshape(poses[f], [4, 4])
            got_g = m[:3, :3] @ new[t][k, :3].astype(np.float64) + m[:3, 3]
            assert np.abs(got_g - want_g).max() < 5e-3                                   # fp32 rows at |x| ~ 2e4 m
            assert np.allclose(new[t][k, 3:6], refined[k, 3:6], atol=1e-5)
            yaw_g = new[t][k, 6] + np.arctan2(m[1, 0], m[0, 0])
            want_yaw = refined[k, 6] + np.arctan2(m_best[1, 0], m_best[0, 0])
            assert abs(np.angle(np.exp(1j * (yaw_g - want_yaw)))) < 1e-4


def test_segment_plan_chain_equals_the_step_by_step_run():
    """segment.SegmentPlan: the same segment, chained on the device — crops laid out track-major, the prep kernels
    reading the tracks' rows in place, heads, write-back, nothing downloaded in between — gives, bit for bit, the
    detection rows the step-by-step run (each step handing NumPy track dictionaries to the next, as the reference's
    files do) gives: static tracks through the static head, dynamic tracks (one item per track-frame) through the
    dynamic head. A second run of the plan reuses its buffers and repeats the bits; a plan with too small a crop
    capacity says so."""
    segment = importlib.import_module("3dal_pytorch_amd.segment")
    poses, sweeps, dets, gbox = _segment(n_frames=7, n_obj=6, seed=71)
    F, K = len(poses), gbox.shape[0]
    tokens = [f"frame{f:04d}" for f in range(F)]
    kinds = ["static", "dynamic", "static", "dynamic", "static", "static"]
    scores = [[0.5 + 0.05 * ((f + k) % F) for f in range(F)] for k in range(K)]
    static = build_model("static_one", synth.state_dict("static_one"))
    dynamic = build_model("dynamic", synth.state_dict("dynamic"))
    # ---- step by step
    frames = crops.extract_crops(sweeps, dets, poses)
    tracks = [{"bbox": [frames[f]["bbox"][k] for f in range(F)], "point": [frames[f]["point"][k].cpu().numpy() for f in range(F)],
               "score": scores[k], "token": tokens, "match": [f"gt{k}"] * F, "type": [1] * F} for k in range(K)]
    v2g = {tokens[f]: poses[f] for f in range(F)}
    det_rows = {tokens[f]: frames[f]["boxes_lidar"].copy() for f in range(F)}
    s_tr = [tracks[k] for k in range(K) if kinds[k] == "static"]
    best = [int(np.argmax(t["score"])) for t in s_tr]
    pts, init = prep.prepare_static_batch(s_tr, [poses[b] for b in best], n_points=1024, sampler="device")
    want_s, _ = post.writeback_static(s_tr, v2g, {(i, t): True for i in range(len(s_tr)) for t in tokens},
                                      static.refine(pts, init), det_rows)
    d_tr = [tracks[k] for k in range(K) if kinds[k] == "dynamic"]
    items = [(i, f) for i in range(len(d_tr)) for f in range(F)]
    dp, db, di = prep.prepare_dynamic_batch(d_tr, items, [poses[f] for _, f in items], n_per_frame=256, sampler="device")
    want_d, _ = post.writeback_dynamic(d_tr, v2g, {(i, t): True for i in range(len(d_tr)) for t in tokens},
                                       dynamic.refine(dp, db, di), det_rows)
    # ---- chained
    plan = segment.SegmentPlan([s.shape[0] for s in sweeps], dets, poses,
                               [{"kind": kinds[k], "dets": [(f, k) for f in range(F)], "score": scores[k]} for k in range(K)],
                               static, dynamic, n_static_points=1024, n_per_frame=256, dynamic_batch=5)   # (ragged batches: 5,5,4)
    d_pts = torch.from_numpy(np.concatenate(sweeps)).cuda()
    plan.run(d_pts)
    assert not plan.overflowed()
    got_s, got_d = plan.detections("static"), plan.detections("dynamic")
    moved = 0
    for t in tokens:
        assert np.array_equal(got_s[t], want_s[t]), t
        assert np.array_equal(got_d[t], want_d[t]), t
        moved += int((got_s[t] != det_rows[t]).any(1).sum()) + int((got_d[t] != det_rows[t]).any(1).sum())
    assert moved == K * F                                                             # every tracked detection was rewritten
    res = plan.run(d_pts)
    torch.cuda.synchronize()
    assert bool((res["static"][1].cpu().numpy().reshape(4, F) >= 0).all())
    again = plan.detections("static")
    assert all(np.array_equal(again[t], got_s[t]) for t in tokens)
    small = segment.SegmentPlan([s.shape[0] for s in sweeps], dets, poses,
                                [{"kind": "static", "dets": [(f, 0) for f in range(F)], "score": scores[0]}], static, None,
                                n_static_points=1024, capacity=100)
    small.run(d_pts)
    assert small.overflowed()
    # the overflowing run must not hand the consumers offsets past its 100-row buffer — they are capped at
    # the capacity (the true prefix sums stay in `start`, which is how overflowed() knows), monotone, and what fits
    # in front of the cap is untouched
    off, start = small.crop.offsets.cpu().numpy(), small.crop.start.cpu().numpy()
    order = small.crop.d_order.cpu().numpy()
    assert off.max() == 100 and start[-1] > 100 and (np.diff(off) >= 0).all()
    assert np.array_equal(off[:-1], np.minimum(start[order], 100)) and off[-1] == 100
    small.grow()
    small.run(d_pts)
    assert not small.overflowed() and np.array_equal(small.detections("static")[tokens[0]][0], got_s[tokens[0]][0])


def test_segment_chain_sharded_over_two_ranks_equals_one_rank():
    """SURVEY.md 8(e) for the chain: two ranks (sharing this box's one GPU, the boxes all-gathered over gloo through host
    memory — RCCL cannot connect ranks on one device) shard the static tracks (4 -> 2 + 2) and the dynamic track-frames
    (21 -> 11 + 10), gather the refined boxes once per head and write back; each rank compares its detections with its
    own single-rank run of the same plan, bit for bit (tools/segment_ranks_check.py)."""
    import os
    launch = importlib.import_module("3dal_pytorch_amd.launch")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DAL3_BENCH_SHARE_GPU="1", DAL3_BENCH_BACKEND="gloo")
    rc, out = launch.spawn_ranks(os.path.join(root, "tools", "segment_ranks_check.py"), [], 2, share_gpu=True, env=env, timeout=600)
    assert rc == 0, out[-2000:]
    assert "segment rank 0/2: sharded == alone True" in out and "segment rank 1/2: sharded == alone True" in out, out[-2000:]
