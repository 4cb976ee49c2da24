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


_LABELS = ("bbox_gt", "center_label", "heading_class_label", "heading_residuals_label", "size_class_label",
           "size_residual_label")


def test_static_labels_vs_reference_dataset():
    """the training labels of STATICTRACK.__getitem__ (static_model.py:548-566), mask label on the device"""
    g = golden("prep_static")
    tracks = [synth.track(31, i, n_frames=7 + 3 * i) for i in range(3)]
    for i, tr in enumerate(tracks):
        best = int(np.argmax(tr["score"]))
        pose = synth.pose_veh_to_global(31, tr["token"][best])
        gt9 = synth.gt_box_in_vehicle(tr["bbox"][best], pose)
        np.random.seed(100 + i)
        pts, init, lab = prep.prepare_static_batch([tr], [pose], n_points=4096, sampler="numpy", gt_boxes=[gt9])
        assert np.abs(pts.transpose(2, 1).cpu().numpy()[0] - g[f"point{i}"].astype(np.float32)).max() < TOL
        mask = lab["mask_label"].cpu().numpy()[0]
        assert mask.dtype == np.uint8 and np.array_equal(mask, g[f"mask_label{i}"].astype(np.uint8))
        assert 100 < mask.sum() < 4000
        for name in _LABELS:
            assert np.array_equal(lab[name].cpu().numpy()[0], g[f"{name}{i}"]), name


def test_dynamic_labels_vs_reference_dataset():
    """DYNAMICTRACK.__getitem__ labels (dynamic_model.py:455-501): every window frame labelled in its own vehicle
    frame; frames without the matched annotation (d0: frames 2 and 6) and out-of-track frames give zeros"""
    g = golden("prep_dynamic")
    tracks = [synth.track(32, 10, n_frames=9, empty_every=4), synth.track(32, 11, n_frames=60)]

    def pose_of(t, i):
        return synth.pose_veh_to_global(31, tracks[t]["token"][i])

    def gt_of(t, i):
        return None if (t == 0 and i in (2, 6)) else synth.gt_box_in_vehicle(tracks[t]["bbox"][i], pose_of(t, i))
    k = 0
    while f"index{k}" in g:
        idx = int(g[f"index{k}"])
        t, it = (0, idx) if idx < 9 else (1, idx - 9)
        np.random.seed(200 + k)
        pts, box, init, lab = prep.prepare_dynamic_batch(tracks, [(t, it)], [pose_of(t, it)], sampler="numpy",
                                                         gt_of_frame=gt_of, pose_of_frame=pose_of)
        assert np.array_equal(lab["mask_label"].cpu().numpy()[0], g[f"mask_label{k}"].astype(np.uint8)), k
        for name in _LABELS:
            ref = g[f"{name.replace('residuals_label', 'residual_label') if name.startswith('heading') else name}{k}"]
            got = lab[name].cpu().numpy()[0]
            if name in ("center_label", "heading_residuals_label"):     # differences of float64 numbers: 1 ulp of 2e4
                assert np.abs(got - ref).max() < 1e-11, name
            else:
                assert np.array_equal(got, ref), name
        k += 1
    assert k == 7
    with pytest.raises(ValueError):                                      # the item's own frame lacks its annotation
        prep.prepare_dynamic_batch(tracks, [(0, 2)], [pose_of(0, 2)], gt_of_frame=gt_of, pose_of_frame=pose_of)


def test_labels_with_device_sampler_follow_the_points(monkeypatch):
    """device sampler: label n belongs to output point n (the label kernel repeats the prep kernel's draw).
    A table (box-frame point -> label) over ALL the track's points comes from the numpy path with the draw
    replaced by arange; every device-sampled point is then looked up in it by its exact fp32 coordinates."""
    tr = synth.track(61, 0, n_frames=9)
    best = int(np.argmax(tr["score"]))
    pose = synth.pose_veh_to_global(61, tr["token"][best])
    gt9 = synth.gt_box_in_vehicle(tr["bbox"][best], pose)
    n_all = sum(len(p) for p in tr["point"])
    monkeypatch.setattr(np.random, "choice", lambda n, size, replace=True: np.arange(size) % n)
    tpts, _, tlab = prep.prepare_static_batch([tr], [pose], n_points=n_all, sampler="numpy", gt_boxes=[gt9])
    monkeypatch.undo()
    table = {}
    for xyz, m in zip(tpts[0].t().cpu().numpy(), tlab["mask_label"][0].cpu().numpy()):
        assert table.setdefault(xyz.tobytes(), int(m)) == int(m)
    pts, init, lab = prep.prepare_static_batch([tr], [pose], n_points=2048, sampler="device", gt_boxes=[gt9], seed=5)
    got = lab["mask_label"][0].cpu().numpy()
    want = np.array([table[xyz.tobytes()] for xyz in pts[0].t().cpu().numpy()])
    assert np.array_equal(got, want) and 100 < got.sum() < 1950
    # a shard of a batch draws, and therefore labels, what the whole batch does
    two = prep.prepare_static_batch([tr, tr], [pose, pose], n_points=2048, sampler="device", gt_boxes=[gt9, gt9], seed=5)
    one = prep.prepare_static_batch([tr], [pose], n_points=2048, sampler="device", gt_boxes=[gt9], seed=5, item_offset=1)
    assert torch.equal(two[2]["mask_label"][1:], one[2]["mask_label"]) and torch.equal(two[0][1:], one[0])


def test_static_track_store_batches_equal_the_one_shot_call():
    """N1 with the host taken out of the per-batch path: the segment's tracks are flattened and uploaded once
    (StaticTrackStore); batches [first, first+B) prepared from it equal the same tracks prepared from their dicts,
    bit for bit, for both samplers (the NumPy sampler consumes the global stream track by track either way)."""
    tracks = [synth.track(33, t, n_frames=5 + t % 4) for t in range(10)]
    poses = [synth.pose_veh_to_global(33, tr["token"][int(np.argmax(tr["score"]))]) for tr in tracks]
    store = prep.StaticTrackStore(tracks)
    assert len(store) == 10 and store.pts.shape[0] == sum(sum(len(p) for p in tr["point"]) for tr in tracks)
    whole_p, whole_i = prep.prepare_static_batch(tracks, poses, n_points=512, sampler="device", seed=9)
    parts = [prep.prepare_static_batch(store, poses[k:k + 4], n_points=512, sampler="device", seed=9, first=k, item_offset=k)
             for k in range(0, 10, 4)]
    assert torch.equal(torch.cat([p for p, _ in parts]), whole_p) and torch.equal(torch.cat([i for _, i in parts]), whole_i)
    np.random.seed(5)
    a_p, a_i = prep.prepare_static_batch(tracks[3:7], poses[3:7], n_points=300, sampler="numpy")
    np.random.seed(5)
    b_p, b_i = prep.prepare_static_batch(store, poses[3:7], n_points=300, sampler="numpy", first=3)
    assert torch.equal(a_p, b_p) and torch.equal(a_i, b_i)
    with pytest.raises(ValueError):
        prep.prepare_static_batch(store, poses[:4], first=8)
    # round 5: a store that holds its tracks' best-frame poses: a batch is `prepare_static_batch(store, B, first=k)` —
    # one launch, no host arithmetic or upload — and gives the same bits
    store_p = prep.StaticTrackStore(tracks, veh_to_global=poses)
    parts = [prep.prepare_static_batch(store_p, min(4, 10 - k), n_points=512, sampler="device", seed=9, first=k, item_offset=k)
             for k in range(0, 10, 4)]
    assert torch.equal(torch.cat([p for p, _ in parts]), whole_p) and torch.equal(torch.cat([i for _, i in parts]), whole_i)
    with pytest.raises(ValueError):
        prep.prepare_static_batch(store, 4, first=0)                                # a store without poses
    with pytest.raises(ValueError):
        prep.prepare_static_batch(store_p, 4, first=0, sampler="numpy")
