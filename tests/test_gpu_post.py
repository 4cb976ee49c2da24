"""Write-back kernel through the C ABI vs the det_annos produced by the reference's real
postprocessing(). Float64 transforms on both sides; the stored rows are fp32 and compared within 1 ulp-scale."""
import importlib

import numpy as np
import pytest
import torch

from _common import golden, synth

post = importlib.import_module("3dal_pytorch_amd.post")
pytestmark = pytest.mark.gpu


def test_static_and_dynamic_writeback_vs_reference():
    g = golden("post_writeback")
    tracks, poses, dets, has_gt = synth.scene(33, n_frames=24, n_tracks=9)
    out, match = post.writeback_static(tracks, poses, has_gt, g["final_static"], dets)
    for t in dets:
        want = g[f"static_{t}"]
        assert np.abs(out[t] - want).max() <= 4e-6 * max(1.0, np.abs(want).max()), t
        assert np.array_equal((out[t] != dets[t]).any(1), (want != dets[t]).any(1))       # the same rows changed
    assert (match >= 0).sum() == sum(has_gt.values())
    out, _ = post.writeback_dynamic(tracks, poses, has_gt, torch.from_numpy(g["final_dyn"]).cuda(), dets)
    for t in dets:
        assert np.array_equal(out[t], g[f"dynamic_{t}"]), t          # no transform on this path: exact


def test_writeback_raises_when_a_box_is_missing():
    tracks, poses, dets, has_gt = synth.scene(36, n_frames=10, n_tracks=3)
    tok = tracks[0]["token"][0]
    has_gt[(0, tok)] = True
    dets = dict(dets)
    dets[tok] = dets[tok] + np.float32(5.0)                 # move every detection of that frame away
    with pytest.raises(AssertionError, match="not in det_annos"):
        post.writeback_static(tracks, poses, has_gt, np.zeros((3, 7)), dets)


def test_a_writeback_plan_applied_twice_equals_two_one_shot_calls():
    """round 4: the segment's pairs / poses / detections flattened and uploaded once (post.WritebackPlan); applying it
    to two different sets of refined boxes gives exactly what two one-shot calls give (the detection array is restored
    from its pristine copy in between), and the wrong number of boxes is refused"""
    tracks, poses, dets, has_gt = synth.scene(34, n_frames=24, n_tracks=9)
    g = golden("post_writeback")
    plan = post.WritebackPlan(tracks, poses, has_gt, dets, static=True)
    for final in (g["final_static"], g["final_static"][::-1].copy() if len(g["final_static"]) == plan.n_final else g["final_static"]):
        try:
            want, wm = post.writeback_static(tracks, poses, has_gt, final, dets)
        except AssertionError:
            with pytest.raises(AssertionError, match="not in det_annos"):
                plan.apply(final)
            continue
        got, gm = plan.apply(final)
        assert np.array_equal(gm, wm)
        for t in dets:
            assert np.array_equal(got[t], want[t]), t
    with pytest.raises(ValueError):
        plan.apply(np.zeros((plan.n_final + 1, 7)))
