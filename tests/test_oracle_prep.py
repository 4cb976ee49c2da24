"""Pins oracle/ref_prep.py (crop preparation) to outputs of the reference's real
STATICTRACK / DYNAMICTRACK __getitem__ (fixtures from tests/golden/gen_golden.py). CPU only."""
import numpy as np

from _common import golden, synth
from oracle import ref_prep as P


def test_static_crop_matches_reference_dataset():
    g = golden("prep_static")
    for i in range(3):
        tr = synth.track(31, i, n_frames=7 + 3 * i)
        best = int(np.argmax(tr["score"]))
        assert str(g[f"token{i}"]) == tr["token"][best]
        np.random.seed(100 + i)
        aux = {}
        pose = synth.pose_veh_to_global(31, tr["token"][best])
        box, pt, _ = P.static_crop(np.vstack(tr["point"]), np.vstack(tr["bbox"]), np.stack(tr["score"]), pose, 4096, aux)
        assert np.array_equal(box, g[f"init_box{i}"])
        assert np.array_equal(pt, g[f"point{i}"])
        assert np.abs(pt).max() < 20.0                    # box-centred despite kilometre-scale global coordinates
        # training labels (static_model.py:548-566) against the matched annotation of the best frame
        gt9 = synth.gt_box_in_vehicle(tr["bbox"][best], pose)
        lab = P.static_labels(aux["point_vehicle"], box, gt9)
        names = ("bbox_gt", "mask_label", "center_label", "heading_class_label", "heading_residuals_label",
                 "size_class_label", "size_residual_label")
        for name, v in zip(names, lab):
            assert np.array_equal(np.asarray(v), g[f"{name}{i}"]), name
        assert 100 < lab[1].sum() < 4000


def test_dynamic_item_matches_reference_dataset():
    g = golden("prep_dynamic")
    tracks = [synth.track(32, 10, n_frames=9, empty_every=4), synth.track(32, 11, n_frames=60)]
    assert int(g["len"]) == 69
    k = 0
    while f"index{k}" in g:
        idx = int(g[f"index{k}"])
        tr, item = (tracks[0], idx) if idx < 9 else (tracks[1], idx - 9)
        np.random.seed(200 + k)
        aux = {}
        pose = synth.pose_veh_to_global(31, tr["token"][item])                                   # pickles: seed 31
        init, bbox, pt, draws = P.dynamic_item(tr["point"], tr["bbox"], item, pose, aux=aux)
        assert np.array_equal(init, g[f"init_box{k}"])
        assert np.array_equal(bbox, g[f"bbox{k}"])
        assert np.array_equal(pt.astype(np.float32), g[f"point{k}"])
        assert np.array_equal(pt[:8], g[f"point64_head{k}"])
        # quirk: a missing frame's "zero" points are still moved by the pose and the re-centring
        for j, d in enumerate(draws):
            if d is None:
                blk = pt[j * 1024:(j + 1) * 1024, :3]
                assert np.ptp(blk, axis=0).max() == 0.0 and np.abs(blk).max() > 0.0
        # training labels (dynamic_model.py:455-501); track d0 lacks its annotation in frames 2 and 6
        def gt_of(i, tr=tr):
            if tr is tracks[0] and i in (2, 6):
                return None
            return synth.gt_box_in_vehicle(tr["bbox"][i], synth.pose_veh_to_global(31, tr["token"][i]))
        lab = P.dynamic_labels(aux["point_item"], aux["bbox_item"], item, len(tr["token"]), pose, gt_of,
                               lambda i, tr=tr: synth.pose_veh_to_global(31, tr["token"][i]))
        names = ("bbox_gt", "mask_label", "center_label", "heading_class_label", "heading_residual_label",
                 "size_class_label", "size_residual_label")
        for name, v in zip(names, lab):
            assert np.array_equal(np.asarray(v), g[f"{name}{k}"]), (name, k)
        k += 1
    assert k == 7
