"""Crop preparation kernels through the C ABI vs the fixtures produced by the
reference's real STATICTRACK / DYNAMICTRACK __getitem__ and vs the oracle (oracle/ref_prep.py).
Arithmetic is float64 on both sides; outputs are compared after the fp32 rounding the drivers apply,
within 2 fp32 ulps of the box-frame coordinates (summation order inside the 4x4 product differs)."""
import importlib

import numpy as np
import pytest
import torch

from _common import build_model, golden, synth
from oracle import ref_prep as P

prep = importlib.import_module("3dal_pytorch_amd.prep")
pytestmark = pytest.mark.gpu
TOL = 4e-6      # metres, on box-frame coordinates of magnitude <= ~20 m (fp32 ulp at 16 m is 1.9e-6)


def test_static_crop_prep_vs_reference_dataset():
    g = golden("prep_static")
    tracks = [synth.track(31, i, n_frames=7 + 3 * i) for i in range(3)]
    poses = [synth.pose_veh_to_global(31, tr["token"][int(np.argmax(tr["score"]))]) for tr in tracks]
    # one crop at a time so that the global NumPy stream is consumed exactly as the fixtures did
    for i in range(3):
        np.random.seed(100 + i)
        pts, init = prep.prepare_static_batch([tracks[i]], [poses[i]], n_points=4096, sampler="numpy")
        want = g[f"point{i}"].astype(np.float32)
        got = pts.transpose(2, 1).cpu().numpy()[0]
        assert np.abs(got - want).max() < TOL
        assert np.array_equal(init.cpu().numpy()[0], g[f"init_box{i}"][0].astype(np.float32))
        assert pts.shape == (1, 3, 4096) and pts.stride() == (4096 * 3, 1, 3)       # the callers' layout


def test_static_crop_prep_batched_and_device_sampler():
    tracks = [synth.track(41, i, n_frames=5 + i) for i in range(6)]
    poses = [synth.pose_veh_to_global(41, tr["token"][int(np.argmax(tr["score"]))]) for tr in tracks]
    np.random.seed(7)
    pts, init = prep.prepare_static_batch(tracks, poses, n_points=512, sampler="numpy")
    np.random.seed(7)
    for b, tr in enumerate(tracks):
        box, pt, _ = P.static_crop(np.vstack(tr["point"]), np.vstack(tr["bbox"]), np.stack(tr["score"]), poses[b], 512)
        assert np.abs(pts[b].t().cpu().numpy() - pt.astype(np.float32)).max() < TOL
        assert np.array_equal(init[b].cpu().numpy(), box[0].astype(np.float32))
    # device sampler: every output point is one of the track's points in the box frame; deterministic; sharded == whole
    a, _ = prep.prepare_static_batch(tracks, poses, n_points=512, sampler="device", seed=3)
    b2, _ = prep.prepare_static_batch(tracks, poses, n_points=512, sampler="device", seed=3)
    assert torch.equal(a, b2)
    c, _ = prep.prepare_static_batch(tracks[2:], poses[2:], n_points=512, sampler="device", seed=3, item_offset=2)
    assert torch.equal(a[2:], c)
    for b, tr in enumerate(tracks):
        allp = np.vstack(tr["point"])
        box, full, _ = P.static_crop(allp, np.vstack(tr["bbox"]), np.stack(tr["score"]), poses[b], 8)   # frame only
        pose = np.linalg.inv(np.reshape(poses[b], [4, 4]))
        ref = (pose @ np.concatenate([allp.T, np.ones((1, len(allp)))]))[:3].T - box[:, :3]
        ref = (P.rotz(-box[0, -1]) @ ref.T).T.astype(np.float32)
        got = a[b].t().cpu().numpy()
        d = np.abs(got[:, None, :] - ref[None, :, :]).max(2).min(1)
        assert d.max() < TOL
        assert len(np.unique(got, axis=0)) > 0.5 * min(512, len(allp))               # spread over the track's points


def test_dynamic_item_prep_vs_reference_dataset():
    g = golden("prep_dynamic")
    tracks = [synth.track(32, 10, n_frames=9, empty_every=4), synth.track(32, 11, n_frames=60)]
    k = 0
    while f"index{k}" in g:
        idx = int(g[f"index{k}"])
        t, it = (0, idx) if idx < 9 else (1, idx - 9)
        pose = synth.pose_veh_to_global(31, tracks[t]["token"][it])
        np.random.seed(200 + k)
        pts, box, init = prep.prepare_dynamic_batch(tracks, [(t, it)], [pose], sampler="numpy")
        assert pts.shape == (1, 4, 5120) and box.shape == (1, 8, 101)
        assert np.abs(pts[0].t().cpu().numpy() - g[f"point{k}"]).max() < TOL
        want_box = g[f"bbox{k}"].astype(np.float32)
        assert np.abs(box[0].t().cpu().numpy() - want_box).max() < TOL * max(1.0, np.abs(want_box).max() / 16)
        want_init = g[f"init_box{k}"].astype(np.float32)
        assert np.abs(init[0].cpu().numpy() - want_init).max() <= np.abs(want_init).max() * 2e-7
        k += 1
    assert k == 7


def test_prepared_crops_feed_the_heads():
    """prep -> refine end to end on the device: finite boxes, batch of two calls == one call"""
    tracks = [synth.track(51, i, n_frames=6) for i in range(8)]
    poses = [synth.pose_veh_to_global(51, tr["token"][int(np.argmax(tr["score"]))]) for tr in tracks]
    pts, init = prep.prepare_static_batch(tracks, poses, n_points=1024, sampler="device")
    model = build_model("static_one", synth.state_dict("static_one"))
    boxes = model.refine(pts, init)
    assert boxes.shape == (8, 7) and bool(torch.isfinite(boxes).all())
    dtracks = [synth.track(52, 3, n_frames=12, empty_every=5)]
    items = [(0, i) for i in range(12)]
    dposes = [synth.pose_veh_to_global(52, dtracks[0]["token"][i]) for i in range(12)]
    dp, db, di = prep.prepare_dynamic_batch(dtracks, items, dposes, sampler="device")
    dmodel = build_model("dynamic", synth.state_dict("dynamic"))
    out = dmodel.refine(dp, db, di)
    assert out.shape == (12, 7) and bool(torch.isfinite(out).all())
