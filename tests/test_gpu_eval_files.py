"""The file-level drivers (3dal_pytorch_amd/eval.py = tools/static_eval.py + tools/dynamic_eval.py of the reference) on
one synthetic segment written in the reference's pickle formats. Each stage the driver strings
together is checked against the oracle ON THE DRIVER'S OWN STREAM: the inputs it hands to the heads are recorded,
the oracle re-draws them from the recorded NumPy generator states (so the order in which the driver consumes the
global stream is pinned draw for draw, including the reference's substitution of dynamic items that lack their
annotation), the heads' boxes are compared crop by crop, and the rewritten det_annos against oracle/ref_post.py."""
import importlib
import pickle

import numpy as np
import pytest
import torch

from _common import rel_err, synth
from oracle import ref_heads as R
from oracle import ref_post as W
from oracle import ref_prep as P

ev = importlib.import_module("3dal_pytorch_amd.eval")
static_model = importlib.import_module("3dal_pytorch_amd.static_model")
dynamic_model = importlib.import_module("3dal_pytorch_amd.dynamic_model")
pytestmark = pytest.mark.gpu
TOL_PREP = 4e-6         # metres (tests/test_gpu_prep.py)
TOL = 1e-4              # BASELINE.json north_star, relative, on the refined boxes


def _segment(tmp_path, kind, seed=77):
    paths, tracks, poses, dets, has_gt = synth.segment_files(str(tmp_path), seed, n_frames=12, n_tracks=7)
    sd = synth.state_dict(kind)
    ckpt = str(tmp_path / f"{kind}.pth")
    torch.save({"epoch": 1, "model_state_dict": {k: torch.as_tensor(np.asarray(v)) for k, v in sd.items()}}, ckpt)
    return paths, tracks, poses, dets, has_gt, sd, ckpt


def _record(monkeypatch, cls):
    """wrap cls.refine: per call keep the inputs, the boxes and the NumPy stream state at entry and exit"""
    calls = []
    orig = cls.refine

    def refine(self, *args):
        entry = np.random.get_state()
        out = orig(self, *args)
        calls.append({"args": [a.detach().cpu() for a in args], "boxes": out.detach().cpu().numpy(), "entry": entry,
                      "exit": np.random.get_state()})
        return out
    monkeypatch.setattr(cls, "refine", refine)
    return calls


def _same_state(a, b):
    return a[0] == b[0] and np.array_equal(a[1], b[1]) and a[2:] == b[2:]


def _check_written(result_path, det_annos, dets, want):
    with open(result_path, "rb") as f:
        saved = pickle.load(f)
    assert [d["frame_id"] for d in saved] == sorted(d["frame_id"] for d in saved)          # sort_detections
    changed = 0
    for d, mem in zip(saved, det_annos):
        tok = d["metadata"]["token"]
        assert np.array_equal(d["boxes_lidar"], mem["boxes_lidar"])
        assert np.abs(d["boxes_lidar"] - want[tok]).max() <= 4e-6 * max(1.0, np.abs(want[tok]).max()), tok
        assert np.array_equal((d["boxes_lidar"] != dets[tok]).any(1), (want[tok] != dets[tok]).any(1))
        changed += int((d["boxes_lidar"] != dets[tok]).any(1).sum())
    return changed


def test_static_eval_files_numpy_stream(tmp_path, monkeypatch):
    paths, tracks, poses, dets, has_gt, sd, ckpt = _segment(tmp_path, "static_one")
    calls = _record(monkeypatch, static_model.StaticModelOneBoxEst)
    final, det_annos = ev.run("static", paths["static"], paths["infos"], paths["det_annos"], ckpt, "one_box_est",
                              batch_size=3, sampler="numpy")
    # preprocessing (static_eval.py:26-44): tracks whose best-score frame lacks the annotation are gone
    kept = [k for k, tr in enumerate(tracks) if has_gt[(k, tr["token"][int(np.argmax(tr["score"]))])]]
    assert 2 <= len(kept) < len(tracks) and final.shape == (len(kept), 7)
    assert len(calls) == -(-len(kept) // 3)
    tsd = R.as_torch_sd(sd)
    np.random.seed(ev.SEED)
    state, row, agree = np.random.get_state(), 0, 0
    for call in calls:
        pts, init = call["args"]
        B = pts.shape[0]
        np.random.set_state(state)
        for b in range(B):                                              # STATICTRACK.__getitem__, in order
            tr = tracks[kept[row + b]]
            pose = poses[tr["token"][int(np.argmax(tr["score"]))]]
            box, pt, _ = P.static_crop(np.vstack(tr["point"]), np.vstack(tr["bbox"]), np.stack(tr["score"]), pose, 4096)
            assert np.abs(pts[b].t().numpy() - pt.astype(np.float32)).max() < TOL_PREP
            assert np.array_equal(init[b].numpy(), box[0].astype(np.float32))
        assert _same_state(np.random.get_state(), call["entry"])       # the batch's draws, then the heads'
        want = R.static_one_forward(tsd, pts, init)
        want7 = R.decode_static(want, init, False)
        for b in range(B):
            if rel_err(call["boxes"][b], want7[b]) < TOL:
                agree += 1
        if _same_state(np.random.get_state(), call["exit"]):
            assert rel_err(call["boxes"], want7) < TOL                 # same draws => every crop of the batch agrees
        assert np.array_equal(final[row:row + B], call["boxes"].astype(np.float64))
        state, row = call["exit"], row + B
    assert agree >= 0.8 * len(kept)                                     # (a mask flip at a ~0 margin redraws a crop)
    # write-back vs the oracle, fed with the driver's boxes
    work = {t: d.copy() for t, d in dets.items()}
    for i, k in enumerate(kept):
        W.static_writeback([tracks[k]], poses, {t: has_gt[(k, t)] for t in tracks[k]["token"]}, final[[i]], work)
    changed = _check_written(tmp_path / "static" / "box" / "one_box_est.pkl", det_annos, dets, work)
    assert changed == sum(has_gt[(k, t)] for k in kept for t in tracks[k]["token"])
    assert (tmp_path / "static" / "log" / "eval" / "one_box_est.txt").exists()


def test_dynamic_eval_files_numpy_stream(tmp_path, monkeypatch):
    paths, tracks, poses, dets, has_gt, sd, ckpt = _segment(tmp_path, "dynamic")
    calls = _record(monkeypatch, dynamic_model.DynamicModel)
    final, det_annos = ev.run("dynamic", paths["dynamic"], paths["infos"], paths["det_annos"], ckpt, batch_size=16,
                              sampler="numpy")
    items = [(k, j) for k, tr in enumerate(tracks) for j in range(len(tr["token"]))]
    n = len(items)
    assert final.shape == (n, 7) and sum(not has_gt[(k, tracks[k]["token"][j])] for k, j in items) >= 3
    np.random.seed(ev.SEED)
    state, row = np.random.get_state(), 0

    def getitem(index):                                                 # DYNAMICTRACK.__getitem__ incl. :487-489
        k, j = items[index]
        tr = tracks[k]
        init, bbox, point, _ = P.dynamic_item(tr["point"], tr["bbox"], j, poses[tr["token"][j]])
        if not has_gt[(k, tr["token"][j])]:
            return getitem(np.random.randint(n))
        return init, bbox, point
    for call in calls:
        pts, box, init = call["args"]
        B = pts.shape[0]
        np.random.set_state(state)
        for b in range(B):
            want_init, want_box, want_pt = getitem(row + b)
            # (an absent or empty frame is 1024 ZERO points carried through the pose: coordinates of kilometres)
            assert (np.abs(pts[b].t().numpy() - want_pt) < TOL_PREP * np.maximum(1.0, np.abs(want_pt) / 16)).all()
            wb = want_box.astype(np.float32)
            assert np.abs(box[b].t().numpy() - wb).max() < TOL_PREP * max(1.0, np.abs(wb).max() / 16)
            wi = want_init.astype(np.float32)
            assert np.abs(init[b].numpy() - wi).max() <= np.abs(wi).max() * 2e-7
        assert _same_state(np.random.get_state(), call["entry"])
        assert np.array_equal(final[row:row + B], call["boxes"].astype(np.float64))
        state, row = call["exit"], row + B
    assert row == n and np.isfinite(final).all()
    work = {t: d.copy() for t, d in dets.items()}
    index = 0
    for k, tr in enumerate(tracks):
        m = len(tr["token"])
        W.dynamic_writeback([tr], poses, {t: has_gt[(k, t)] for t in tr["token"]}, final[index:index + m], work)
        index += m
    changed = _check_written(tmp_path / "dynamic" / "box" / "box.pkl", det_annos, dets, work)
    assert changed == sum(has_gt.values())


def test_dynamic_heads_on_prepared_items_vs_oracle(tmp_path, monkeypatch):
    """the heads stage of the dynamic driver: the recorded inputs through the oracle, from the recorded stream state"""
    paths, tracks, poses, dets, has_gt, sd, ckpt = _segment(tmp_path, "dynamic")
    calls = _record(monkeypatch, dynamic_model.DynamicModel)
    ev.run("dynamic", paths["dynamic"], paths["infos"], paths["det_annos"], ckpt, batch_size=8, sampler="numpy")
    tsd = R.as_torch_sd(sd)
    agree = total = 0
    for call in calls[:3]:
        pts, box, init = call["args"]
        np.random.set_state(call["entry"])
        want7 = R.decode_dynamic(R.dynamic_forward(tsd, pts, box), init)
        agree += sum(rel_err(call["boxes"][b], want7[b]) < TOL for b in range(pts.shape[0]))
        total += pts.shape[0]
    assert agree >= 0.8 * total


@pytest.mark.parametrize("head", ["static", "dynamic"])
def test_device_sampler_cli_and_shards(tmp_path, monkeypatch, head):
    """`--sampler device`: repeatable, and two ranks' contiguous shards concatenate to the one-rank result bit for bit"""
    kind = "static_two" if head == "static" else "dynamic"
    paths, tracks, poses, dets, has_gt, sd, ckpt = _segment(tmp_path, kind)
    out = tmp_path / "out.pkl"
    argv = [head, "--track", paths[head], "--infos", paths["infos"], "--model_path", ckpt, "--det_annos",
            paths["det_annos"], "--batch_size", "5", "--sampler", "device", "--result", str(out)]
    if head == "static":
        argv += ["--model_type", "two_box_est"]
    ev.main(argv)
    first = pickle.load(open(out, "rb"))
    ev.main(argv)
    again = pickle.load(open(out, "rb"))
    moved = 0
    for a, b in zip(first, again):
        assert np.array_equal(a["boxes_lidar"], b["boxes_lidar"]) and np.isfinite(a["boxes_lidar"]).all()
        moved += int((a["boxes_lidar"] != dets[a["metadata"]["token"]]).any(1).sum())
    assert moved > 10
    # shards
    infos = ev.reorganize_info(pickle.load(open(paths["infos"], "rb")))
    annos = ev.Annos(infos)
    track = pickle.load(open(paths[head], "rb"))
    if head == "static":
        track = ev.preprocessing(track, annos)
        model = static_model.StaticModelTwoBoxEst(3, 3)
        refine = ev.refine_static_tracks
    else:
        model = dynamic_model.DynamicModel(3, 4)
        refine = ev.refine_dynamic_tracks
    model.load_state_dict(torch.load(ckpt)["model_state_dict"])
    model = model.cuda()
    whole = refine(model, track, annos, batch_size=5, sampler="device")
    parts = []
    monkeypatch.setattr(ev.sharding, "all_gather_boxes", lambda local, n, group=None: local)
    for rank in range(2):
        monkeypatch.setattr(ev, "_world", lambda group, r=rank: (r, 2))
        parts.append(refine(model, track, annos, batch_size=4, sampler="device"))
    assert parts[0].shape[0] + parts[1].shape[0] == whole.shape[0] and parts[1].shape[0] > 0
    assert np.array_equal(np.concatenate(parts), whole)
    with pytest.raises(ValueError, match="cannot be sharded"):
        refine(model, track, annos, sampler="numpy")
