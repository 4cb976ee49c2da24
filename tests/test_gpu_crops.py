"""Points-in-rotated-box and crop extraction on the device (SURVEY.md 8(f) N2, and the labels' test of N1)
through the C ABI, against fixtures produced by the reference's own det3d geometry code and by its
_create_pd_detection (tests/golden/gen_golden.py) and against the oracle (oracle/ref_geom.py).
Membership, counts, order and indices: exact. Global-frame coordinates: float64, |diff| <= 1e-9 m at
|x| ~ 2e4 m (the 4x4 product's summation order differs from BLAS)."""
import importlib

import numpy as np
import pytest
import torch

from _common import golden, synth
from oracle import ref_geom as G

geom = importlib.import_module("3dal_pytorch_amd.geom")
crops = importlib.import_module("3dal_pytorch_amd.crops")
hip = importlib.import_module("3dal_pytorch_amd._hip")
pytestmark = pytest.mark.gpu


def _geom_points(dt):
    pts = synth.sweep(40, "geom", n_points=6000, n_boxes=9)[0].astype(dt)
    pts[:3] = [[np.nan, 0.0, 0.0], [8.0, np.nan, 0.5], [8.0, -4.0, 0.5]]
    return pts


def test_points_in_rbbox_vs_reference():
    g = golden("geom_rbbox")
    for tag, dt in (("f32", np.float32), ("f64", np.float64)):
        got = geom.points_in_rbbox(torch.from_numpy(_geom_points(dt)).cuda(), g[f"boxes_{tag}"])
        assert got.dtype == torch.bool and np.array_equal(got.cpu().numpy(), g[f"inside_{tag}"])
    pts64 = synth.sweep(40, "geom", n_points=6000, n_boxes=9)[0].astype(np.float64) + 1e-9
    got = geom.points_in_rbbox(torch.from_numpy(pts64).cuda(), g["boxes_f32"])        # the Datasets' mixed case
    assert np.array_equal(got.cpu().numpy(), g["inside_mixed"])
    # float32 points against float64 boxes promote to float64 as well
    got = geom.points_in_rbbox(torch.from_numpy(_geom_points(np.float32)).cuda(), g["boxes_f64"])
    assert np.array_equal(got.cpu().numpy(), G.points_in_rbbox(_geom_points(np.float32), g["boxes_f64"]))


def test_points_on_a_face_are_outside_and_rows_may_be_strided():
    box = np.array([[8.0, -4.0, 0.5, 4.0, 2.0, 1.5, 0.0]], np.float32)
    face = np.array([[10.0, -4.0, 0.5], [6.0, -4.0, 0.5], [8.0, -3.0, 0.5], [8.0, -5.0, 0.5], [8.0, -4.0, 1.25],
                     [8.0, -4.0, -0.25], [9.99, -4.0, 0.5]], np.float32)
    assert geom.points_in_rbbox(torch.from_numpy(face).cuda(), box)[:, 0].tolist() == [False] * 6 + [True]
    wide = torch.zeros((7, 5), device="cuda")                                 # (P,5) rows: xyz + two features
    wide[:, :3] = torch.from_numpy(face).cuda()
    assert geom.points_in_rbbox(wide, box)[:, 0].tolist() == [False] * 6 + [True]
    assert geom.points_in_rbbox(wide[:0], box).shape == (0, 1)
    assert geom.points_in_rbbox(wide, box[:0]).shape == (7, 0)


def test_membership_at_sweep_size_vs_oracle():
    pts, box9, _, _, _ = synth.sweep(44, "big", n_points=180000, n_boxes=64)
    boxes = G.waymo_boxes(box9)
    got = geom.points_in_rbbox(torch.from_numpy(pts).cuda(), boxes).cpu().numpy()
    assert np.array_equal(got, G.points_in_rbbox(pts, boxes))
    assert got.sum() > 10000


def test_crop_extraction_vs_reference_create_pd_detection():
    g = golden("crops_extract")
    sweeps, dets, poses = [], [], []
    for f in range(3):
        pts, box9, _, _, pose = synth.sweep(41, f"fr{f}", n_points=12000 + 1000 * f, n_boxes=10 + f)
        sweeps.append(pts)
        dets.append(box9)
        poses.append(pose)
    frames = crops.extract_crops(sweeps, dets, poses, return_index=True)
    for f, rec in enumerate(frames):
        assert np.array_equal(rec["boxes_lidar"], g[f"boxes_lidar{f}"])
        assert np.array_equal(rec["bbox"], g[f"bbox{f}"])
        assert [int(p.shape[0]) for p in rec["point"]] == g[f"count{f}"].tolist()
        got = torch.cat(list(rec["point"])).cpu().numpy()
        assert got.dtype == np.float64 and np.abs(got - g[f"point{f}"]).max() < 1e-9
        # sweep order inside every detection, and the indices are the members
        inside = G.points_in_rbbox(sweeps[f], rec["boxes_lidar"])
        for k, idx in enumerate(rec["index"]):
            assert np.array_equal(idx.cpu().numpy(), np.nonzero(inside[:, k])[0])
    # one frame at a time gives the same as the batch
    solo = crops.extract_crops(sweeps[1:2], dets[1:2], poses[1:2])[0]
    assert all(torch.equal(a, b) for a, b in zip(solo["point"], frames[1]["point"]))


def test_crop_extraction_ragged_frames_vs_oracle():
    """frames of very different sizes, a frame without detections, a sweep shorter than one chunk"""
    spec = [(70000, 40), (300, 3), (5000, 0), (1025, 7), (40000, 25)]
    sweeps, dets, poses = [], [], []
    for f, (n, k) in enumerate(spec):
        pts, box9, _, _, pose = synth.sweep(45, f"rg{f}", n_points=n, n_boxes=max(k, 2))
        sweeps.append(pts)
        dets.append(box9[:k])
        poses.append(pose)
    frames = crops.extract_crops(sweeps, dets, poses)
    for f, rec in enumerate(frames):
        _, boxes_g, pts_g = G.extract_crops(sweeps[f], dets[f], poses[f]) if spec[f][1] else (None, [], [])
        assert len(rec["point"]) == spec[f][1]
        for k in range(spec[f][1]):
            got = rec["point"][k].cpu().numpy()
            assert got.shape == pts_g[k].shape
            if got.size:
                assert np.abs(got - pts_g[k]).max() < 1e-9
            assert np.array_equal(rec["bbox"][k], boxes_g[k])


def test_crop_api_errors():
    lib = hip.lib()
    z = torch.zeros(8, dtype=torch.int64, device="cuda")
    assert lib.dal3_crop_count(None, hip.ptr(z), None, None, hip.ptr(z), 0, 0, 0, None, None, 0, hip.stream()) != 0
    assert "F" in lib.dal3_last_error().decode()
    pts = torch.zeros((10, 3), device="cuda")
    off = torch.tensor([0, 10], dtype=torch.int64, device="cuda")
    planes = torch.zeros((1, 6, 4), dtype=torch.float64, device="cuda")
    boff = torch.tensor([0, 1], dtype=torch.int64, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    sph = torch.zeros((1, 4), device="cuda")
    rc = lib.dal3_crop_count(hip.ptr(pts), hip.ptr(off), hip.ptr(planes), hip.ptr(sph), hip.ptr(boff), 1, 1, 10, hip.ptr(cnt),
                             None, 0, hip.stream())
    assert rc != 0 and "workspace" in lib.dal3_last_error().decode()
    with pytest.raises(RuntimeError):
        geom.points_in_rbbox(torch.zeros((4, 3)), np.zeros((1, 7), np.float32))


def test_crop_extraction_keeps_the_reference_treatment_of_non_finite_points():
    """a NaN coordinate never fails `>= 0`: the reference puts such a point into EVERY detection; the cull in front
    of the exact test must not change that (nor what happens to infinite coordinates)"""
    pts, box9, _, _, pose = synth.sweep(47, "nan", n_points=3000, n_boxes=5)
    pts[10] = [np.nan, 1.0, 2.0]
    pts[20] = [np.inf, 0.0, 0.0]
    pts[30] = [3.0, -np.inf, 0.0]
    pts[2999] = [np.nan, np.nan, np.nan]
    rec = crops.extract_crops([pts], [box9], [pose], return_index=True)[0]
    inside = G.points_in_rbbox(pts, rec["boxes_lidar"])
    assert inside[10].all() and inside[2999].all()
    for k, idx in enumerate(rec["index"]):
        assert np.array_equal(idx.cpu().numpy(), np.nonzero(inside[:, k])[0])
        assert 10 in idx.cpu().numpy() and 2999 in idx.cpu().numpy()
    _, _, pts_g = G.extract_crops(pts, box9, pose)
    for k in range(5):
        assert np.allclose(rec["point"][k].cpu().numpy(), pts_g[k], rtol=0, atol=1e-9, equal_nan=True)


def test_crop_extraction_without_any_detection():
    pts, box9, _, _, pose = synth.sweep(48, "none", n_points=2000, n_boxes=3)
    frames = crops.extract_crops([pts, pts[:100]], [box9[:0], box9[:0]], [pose, pose], return_index=True)
    assert len(frames) == 2 and all(len(f["point"]) == 0 and len(f["index"]) == 0 and list(f["point"]) == [] and
                                    f["point"].numpy_list() == [] and f["bbox"].shape == (0, 7) for f in frames)


def test_crop_extraction_more_than_64_detections_in_a_frame_vs_oracle():
    """the cull rasterises 64 detections at a time (one mask bit each): a frame with 150 goes through three grid
    passes; and a frame whose detections sit far outside the grid's +-80 m (clamped to its border cells)"""
    pts, box9, _, _, pose = synth.sweep(48, "many", n_points=60000, n_boxes=150)
    far_pts, far_box, _, _, far_pose = synth.sweep(48, "far", n_points=9000, n_boxes=6)
    far_pts = far_pts + np.float32([300.0, -250.0, 0.0])
    far_box = far_box.copy()
    far_box[:, :2] += np.float32([300.0, -250.0])
    frames = crops.extract_crops([pts, far_pts], [box9, far_box], [pose, far_pose], return_index=True)
    for (p, b, ps), rec in zip(((pts, box9, pose), (far_pts, far_box, far_pose)), frames):
        inside = G.points_in_rbbox(p, rec["boxes_lidar"])
        assert inside.sum() > 500
        _, _, pts_g = G.extract_crops(p, b, ps)
        for k, idx in enumerate(rec["index"]):
            assert np.array_equal(idx.cpu().numpy(), np.nonzero(inside[:, k])[0]), k
            if pts_g[k].size:
                assert np.abs(rec["point"][k].cpu().numpy() - pts_g[k]).max() < 1e-9


def test_crop_plan_track_major_order_capacity_and_reuse():
    """crops.CropPlan: the planned run (count, device-side starts, fill; no host step in between) with the detections
    laid out TRACK-major (detection k of every frame = object k): equal, row for row, to the per-detection results of
    extract_crops; offsets by output position; a second run of the same plan gives the same bits; with a capacity below
    the total the rows in front of it are intact, nothing is written past it and total() reports the real size."""
    F, K = 4, 9
    sweeps, dets, poses = [], [], []
    for f in range(F):
        pts, box9, _, _, pose = synth.sweep(49, f"pl{f}", n_points=15000 + 700 * f, n_boxes=K)
        sweeps.append(pts)
        dets.append(box9)
        poses.append(pose)
    frames = crops.extract_crops(sweeps, dets, poses)
    want = [[frames[f]["point"][k].cpu().numpy() for f in range(F)] for k in range(K)]      # [object][frame]
    order = np.array([f * K + k for k in range(K) for f in range(F)])                       # output position -> detection
    d_pts = torch.from_numpy(np.concatenate(sweeps)).cuda()
    plan = crops.CropPlan([s.shape[0] for s in sweeps], dets, poses, order=order)
    out, offsets = plan.run(d_pts)
    total = plan.total()
    off = offsets.cpu().numpy()
    flat = np.concatenate([want[k][f] for k in range(K) for f in range(F)])
    assert total == flat.shape[0] == off[-1] and plan.capacity >= total
    assert np.array_equal(np.diff(off), [want[k][f].shape[0] for k in range(K) for f in range(F)])
    assert np.array_equal(out[:total].cpu().numpy(), flat)
    start = plan.start.cpu().numpy()
    assert all(start[order[i]] == off[i] for i in range(F * K))
    first = out[:total].clone()
    out2, _ = plan.run(d_pts)
    assert out2.data_ptr() == out.data_ptr() and torch.equal(out2[:total], first)
    cap = total // 2
    small = crops.CropPlan([s.shape[0] for s in sweeps], dets, poses, order=order, capacity=cap)
    guard = torch.full((cap + 64, 3), -7.0, dtype=torch.float64, device="cuda")
    small.out = guard[:cap]                                                               # rows [cap, cap+64) must stay -7
    o, _ = small.run(d_pts)
    assert small.total() == total > cap
    assert np.array_equal(o.cpu().numpy(), flat[:cap]) and bool((guard[cap:] == -7.0).all())
    with pytest.raises(ValueError):
        crops.CropPlan([s.shape[0] for s in sweeps], dets, poses, order=order[:-1])


@pytest.mark.parametrize("K", [1, 7, 1000, 8192 + 5, 20000])
def test_crop_starts_on_the_device_equal_numpy_prefix_sums_plain_and_capped(K):
    """dal3_crop_starts / dal3_crop_starts_capped (round 6) against NumPy: box_start[order[i]] = rows in front of output
    position i and box_start[K] = the total, always the TRUE prefix sums; out_offsets the same by position — and, for
    the capped entry, min(., capacity) for capacities of 0, inside the range, exactly the total and beyond it (the
    consumers of a buffer of `capacity` rows must never be pointed past it). One to three trips of the one workgroup
    (8192 positions per trip), with and without an output order."""
    lib = hip.lib()
    rng = np.random.default_rng(K)
    counts = rng.integers(0, 400, size=K).astype(np.int64)
    counts[rng.integers(0, K, size=max(K // 10, 1))] = 0
    total = int(counts.sum())
    d_counts = torch.from_numpy(counts).cuda()
    for order in (None, rng.permutation(K).astype(np.int64)):
        d_order = None if order is None else torch.from_numpy(order).cuda()
        by_pos = counts if order is None else counts[order]
        want_off = np.concatenate([[0], np.cumsum(by_pos)])
        want_start = np.empty(K + 1, np.int64)
        want_start[np.arange(K) if order is None else order] = want_off[:-1]
        want_start[K] = total
        start = torch.full((K + 1,), -7, dtype=torch.int64, device="cuda")
        off = torch.full((K + 1,), -7, dtype=torch.int64, device="cuda")
        hip.check(lib.dal3_crop_starts(hip.ptr(d_counts), hip.ptr(d_order), K, hip.ptr(start), hip.ptr(off), hip.stream()))
        assert np.array_equal(start.cpu().numpy(), want_start) and np.array_equal(off.cpu().numpy(), want_off)
        for cap in (0, total // 3, max(total - 1, 0), total, total + 100):
            start.fill_(-7)
            off.fill_(-7)
            hip.check(lib.dal3_crop_starts_capped(hip.ptr(d_counts), hip.ptr(d_order), K, hip.ptr(start), hip.ptr(off), cap,
                                                  hip.stream()))
            assert np.array_equal(start.cpu().numpy(), want_start), cap                     # never capped
            assert np.array_equal(off.cpu().numpy(), np.minimum(want_off, cap)), cap
        # out_offsets is optional
        start.fill_(-7)
        hip.check(lib.dal3_crop_starts_capped(hip.ptr(d_counts), hip.ptr(d_order), K, hip.ptr(start), None, 5, hip.stream()))
        assert np.array_equal(start.cpu().numpy(), want_start)
