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
