"""Pins oracle/ref_post.py (write-back) to the det_annos returned by the reference's real
postprocessing() on a synthetic segment (fixture from tests/golden/gen_golden.py). CPU only."""
import numpy as np

from _common import golden, synth
from oracle import ref_post as W


def _scene():
    tracks, poses, dets, has_gt = synth.scene(33, n_frames=24, n_tracks=9)
    gt_by_token = [{t: has_gt[(k, t)] for t in tr["token"]} for k, tr in enumerate(tracks)]
    return tracks, poses, dets, has_gt, gt_by_token


def test_static_writeback_matches_reference():
    g = golden("post_writeback")
    tracks, poses, dets, has_gt, gt_by_token = _scene()
    work = {t: d.copy() for t, d in dets.items()}
    for i, tr in enumerate(tracks):                        # the oracle takes has_gt per token for ONE track at a time
        W.static_writeback([tr], poses, gt_by_token[i], g["final_static"][[i]], work)
    changed = 0
    for t in dets:
        assert np.array_equal(work[t], g[f"static_{t}"]), t
        changed += int((work[t] != dets[t]).any(1).sum())
    assert changed > 50


def test_dynamic_writeback_matches_reference():
    g = golden("post_writeback")
    tracks, poses, dets, has_gt, gt_by_token = _scene()
    work = {t: d.copy() for t, d in dets.items()}
    index = 0
    for i, tr in enumerate(tracks):
        n = len(tr["token"])
        W.dynamic_writeback([tr], poses, gt_by_token[i], g["final_dyn"][index:index + n], work)
        index += n
    for t in dets:
        assert np.array_equal(work[t], g[f"dynamic_{t}"]), t
