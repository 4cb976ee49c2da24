"""Pins oracle/ref_geom.py (points-in-rotated-box and the per-frame crop extraction, SURVEY.md 8(f) N1 labels /
N2) to outputs of the reference's own det3d geometry code and of its `_create_pd_detection`, both run by
tests/golden/gen_golden.py. CPU only, bit for bit."""
import numpy as np

from _common import golden, synth
from oracle import ref_geom as G


def test_planes_and_membership_match_reference():
    g = golden("geom_rbbox")
    for tag, dt in (("f32", np.float32), ("f64", np.float64)):
        boxes = g[f"boxes_{tag}"]
        assert boxes.dtype == dt
        assert np.array_equal(G.box_corners(boxes), g[f"corners_{tag}"])
        n, d = G.box_planes(boxes)
        assert n.dtype == dt and np.array_equal(n, g[f"normal_{tag}"]) and np.array_equal(d, g[f"d_{tag}"])
        pts = synth.sweep(40, "geom", n_points=6000, n_boxes=9)[0].astype(dt)
        pts[:3] = [[np.nan, 0.0, 0.0], [8.0, np.nan, 0.5], [8.0, -4.0, 0.5]]
        inside = G.points_in_rbbox(pts, boxes)
        assert np.array_equal(inside, g[f"inside_{tag}"])
        assert inside[0].all() and inside[1].all()             # a NaN coordinate never fails `>= 0`: inside every box
        assert inside[2, 0] and inside[2].sum() == 1            # the centre of box 0
    pts64 = synth.sweep(40, "geom", n_points=6000, n_boxes=9)[0].astype(np.float64) + 1e-9
    assert np.array_equal(G.points_in_rbbox(pts64, g["boxes_f32"]), g["inside_mixed"])


def test_points_on_a_face_are_outside():
    box = np.array([[8.0, -4.0, 0.5, 4.0, 2.0, 1.5, 0.0]], np.float32)
    face = np.array([[10.0, -4.0, 0.5], [6.0, -4.0, 0.5], [8.0, -3.0, 0.5], [8.0, -5.0, 0.5], [8.0, -4.0, 1.25],
                     [8.0, -4.0, -0.25], [9.99, -4.0, 0.5]], np.float32)
    assert G.points_in_rbbox(face, box)[:, 0].tolist() == [False] * 6 + [True]


def test_crop_extraction_matches_reference_create_pd_detection():
    g = golden("crops_extract")
    for f in range(3):
        pts, box9, scores, labels, pose = synth.sweep(41, f"fr{f}", n_points=12000 + 1000 * f, n_boxes=10 + f)
        boxes_lidar, boxes_g, pts_g = G.extract_crops(pts, box9, pose)
        assert np.array_equal(boxes_lidar, g[f"boxes_lidar{f}"])
        assert np.array_equal(np.stack(boxes_g), g[f"bbox{f}"])
        assert [p.shape[0] for p in pts_g] == g[f"count{f}"].tolist()
        assert np.array_equal(np.concatenate(pts_g), g[f"point{f}"])
        assert g[f"count{f}"][-1] == 0 and g[f"count{f}"][:-1].min() > 20
