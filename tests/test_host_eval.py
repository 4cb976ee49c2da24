"""Host logic of the file-level drivers (3dal_pytorch_amd/eval.py) on a synthetic segment in the reference's pickle
formats; CPU only — the device stages are covered by tests/test_gpu_eval_files.py."""
import builtins
import importlib
import pickle

import numpy as np
import pytest

from _common import synth

ev = importlib.import_module("3dal_pytorch_amd.eval")


@pytest.fixture()
def segment(tmp_path):
    return synth.segment_files(str(tmp_path), 77, n_frames=12, n_tracks=7)


def test_formats_round_trip_and_indexing(segment):
    paths, tracks, poses, dets, has_gt = segment
    raw = pickle.load(open(paths["det_annos"], "rb"))
    assert [d["frame_id"] for d in raw] != sorted(d["frame_id"] for d in raw)           # written out of order
    det_annos = ev.sort_detections(raw)
    assert [d["frame_id"] for d in det_annos] == [f"segment-synth0001_with_camera_labels_{f:03d}" for f in range(12)]
    infos = ev.reorganize_info(pickle.load(open(paths["infos"], "rb")))
    assert list(infos) == [f"fr_{f}" for f in range(12)] and infos["fr_3"]["token"] == "fr_3"
    annos = ev.Annos(infos)
    idx = ev.token_to_det_index(infos, det_annos, annos)
    for tok, i in idx.items():
        assert det_annos[i]["metadata"]["token"] == tok
        assert np.array_equal(det_annos[i]["boxes_lidar"], dets[tok])
        assert np.array_equal(annos.pose(tok), poses[tok])
    for (k, tok), has in has_gt.items():
        box = annos.gt_box(tok, tracks[k]["match"][-1])
        assert (box is not None) == bool(has)
        if has:
            assert box.shape == (9,) and box.dtype == np.float32


def test_annotation_files_are_read_once(segment, monkeypatch):
    paths, tracks, *_ = segment
    infos = ev.reorganize_info(pickle.load(open(paths["infos"], "rb")))
    opened = []
    real_open = builtins.open
    monkeypatch.setattr(builtins, "open", lambda f, *a, **k: (opened.append(str(f)), real_open(f, *a, **k))[1])
    annos = ev.Annos(infos)
    for _ in range(3):
        for tok in infos:
            annos(tok), annos.pose(tok), annos.gt_box(tok, "gt_0")
    assert sorted(opened) == sorted(i["anno_path"] for i in infos.values())


def test_preprocessing_drops_tracks_without_best_frame_annotation(segment):
    paths, tracks, poses, dets, has_gt = segment
    annos = ev.Annos(ev.reorganize_info(pickle.load(open(paths["infos"], "rb"))))
    track = pickle.load(open(paths["static"], "rb"))
    ids = list(track)
    kept = ev.preprocessing(track, annos)
    want = [ids[k] for k, tr in enumerate(tracks) if has_gt[(k, tr["token"][int(np.argmax(tr["score"]))])]]
    assert list(kept) == want and 0 < len(want) < len(ids)


def test_dynamic_item_order_and_sampler_checks(segment):
    paths, tracks, poses, dets, has_gt = segment
    annos = ev.Annos(ev.reorganize_info(pickle.load(open(paths["infos"], "rb"))))
    track = pickle.load(open(paths["dynamic"], "rb"))
    items = ev._dynamic_items(track, annos)
    assert [(t, i) for t, i, _ in items] == [(k, j) for k, tr in enumerate(tracks) for j in range(len(tr["token"]))]
    assert [h for _, _, h in items] == [bool(has_gt[(k, tr["token"][j])]) for k, tr in enumerate(tracks)
                                        for j in range(len(tr["token"]))]
    with pytest.raises(ValueError, match="unknown sampler"):
        ev._check_sampler("host", None)
