"""The heads' behaviour on non-finite and degenerate inputs, pinned against the oracle (= the reference's own behaviour).

Reference: a NaN coordinate makes conv1's 64 outputs NaN at that point; torch's relu and torch.max propagate NaN, so the
crop's global feature is NaN in all 1024 channels (tools/static_model.py:279-284), the repeat + cat carries it to every
point (:286-289), all logits of the crop are NaN, `logits[...,0] < logits[...,1]` is False everywhere (:59), the crop
has no object points, `gather_object_pts` leaves its row at zero and draws nothing (:29-47), and the box is estimated
from 512 all-zero points. A +-Inf coordinate gives +-Inf after conv1 and Inf - Inf = NaN in conv2 for any but contrived
weights — the same outcome (asserted on the oracle below). The library's contract (include/dal3.h): a crop with ANY
non-finite coordinate gets exactly that outcome, in every kernel family and precision; the other crops of the batch are
untouched. The dynamic head's box window: an item with a non-finite box entry gets a NaN box embedding, hence a NaN
box prediction and a NaN refined box, as in the reference."""
import ctypes as C
import importlib

import numpy as np
import pytest
import torch

from _common import build_model, recentred_sd, rel_err, synth
from oracle import ref_heads as R

hip = importlib.import_module("3dal_pytorch_amd._hip")
pytestmark = pytest.mark.gpu
TOL = 1e-4


def _poison_static(pts_np):
    p = pts_np.copy()
    p[1, 7, 0] = np.nan
    p[3, 0, 2] = np.inf
    p[4, p.shape[1] - 1, 1] = -np.inf
    return p, (1, 3, 4)


@pytest.mark.parametrize("B,N", [(6, 512), (40, 1024)])          # the latency family (96 tiles) and the throughput family
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_static_crops_with_non_finite_points_match_the_reference(B, N, prec):
    pts_np, init_np, gt_np = synth.static_crops(B, N, seed=77)
    sd = recentred_sd("static_one", pts_np[:2], seed=77)
    pts_np, bad = _poison_static(pts_np)
    tsd = R.as_torch_sd(sd)
    np.random.seed(5)
    with np.errstate(invalid="ignore"):
        want = R.static_one_forward(tsd, torch.from_numpy(pts_np).transpose(2, 1), torch.from_numpy(init_np))
        want_boxes = R.decode_static(want, torch.from_numpy(init_np), False)
    wl = want["logits"].numpy()
    for b in bad:                                             # the reference itself: NaN everywhere, empty mask (Inf included)
        assert np.isnan(wl[b]).all() and not want["mask"][b].any()
    good = [b for b in range(B) if b not in bad]
    assert np.isfinite(wl[good]).all()

    model = build_model("static_one", sd)
    model.precision = prec
    model.sampler = "numpy"
    np.random.seed(5)
    out = model(torch.from_numpy(pts_np).cuda().transpose(2, 1), torch.from_numpy(init_np).cuda(), torch.from_numpy(gt_np).cuda())
    lg = out["logits"].cpu().numpy()
    assert np.array_equal(np.isnan(lg), np.isnan(wl))
    assert not out["mask"][list(bad)].any()
    assert (model.last["counts"].cpu().numpy()[list(bad)] == 0).all()
    boxes = model.last["boxes7"].cpu().numpy()
    assert np.isfinite(boxes).all()
    if prec == "fp32":
        assert rel_err(lg[good], wl[good]) < TOL
        assert np.array_equal(out["mask"].cpu().numpy(), want["mask"].numpy())
        for k in ("center", "heading_scores", "size_scores", "size_residuals"):
            assert rel_err(out[k].cpu().numpy(), want[k].numpy()) < TOL, k
        assert rel_err(boxes, want_boxes) < TOL
    else:                                                     # 16-bit arithmetic: the flagged crops' boxes come from zero points, exactly
        assert rel_err(boxes[list(bad)], want_boxes[list(bad)]) < 5e-2
    # the crops beside a poisoned one are the crops of a clean batch, bit for bit
    clean = synth.static_crops(B, N, seed=77)[0]
    np.random.seed(5)
    if prec == "fp32":
        model.sampler = "device"
        a = model(torch.from_numpy(pts_np).cuda().transpose(2, 1), torch.from_numpy(init_np).cuda(), None)["logits"][good].clone()
        b = model(torch.from_numpy(clean).cuda().transpose(2, 1), torch.from_numpy(init_np).cuda(), None)["logits"][good]
        assert torch.equal(a, b)


def test_all_zero_crops_and_single_point_crops_are_ordinary_inputs():
    B, N = 5, 256
    pts_np, init_np, _ = synth.static_crops(B, N, seed=78)
    pts_np[0] = 0.0                                           # a crop of N identical points at the origin
    pts_np[2, 1:] = pts_np[2, 0]                              # a crop that is one point repeated
    sd = recentred_sd("static_one", pts_np[3:], seed=78)
    np.random.seed(6)
    want = R.static_one_forward(R.as_torch_sd(sd), torch.from_numpy(pts_np).transpose(2, 1), torch.from_numpy(init_np))
    model = build_model("static_one", sd)
    model.sampler = "numpy"
    np.random.seed(6)
    out = model(torch.from_numpy(pts_np).cuda().transpose(2, 1), torch.from_numpy(init_np).cuda(), None)
    assert rel_err(out["logits"].cpu().numpy(), want["logits"].numpy()) < TOL
    # (identical points give identical margins: whichever side they fall on, all of them do)
    m = out["mask"].cpu().numpy()
    assert m[0].all() or not m[0].any()
    assert m[2].all() or not m[2].any()
    if np.array_equal(m, want["mask"].numpy()):
        for k in ("center", "heading_scores", "size_scores", "size_residuals"):
            assert rel_err(out[k].cpu().numpy(), want[k].numpy()) < TOL, k
    assert torch.isfinite(model.last["boxes7"]).all()


def test_dynamic_items_with_non_finite_points_or_boxes_match_the_reference():
    B = 4
    p, bx, i8, gt = synth.dynamic_items(B, n_per_frame=128, seed=79)
    sd = recentred_sd("dynamic", p[:1], seed=79)
    p, bx = p.copy(), bx.copy()
    p[1, 300, 3] = np.nan                                     # a NaN time stamp in item 1's points
    bx[2, 50, 4] = np.inf                                     # an Inf size in item 2's box window
    np.random.seed(8)
    with np.errstate(invalid="ignore"):
        want = R.dynamic_forward(R.as_torch_sd(sd), torch.from_numpy(p).transpose(2, 1), torch.from_numpy(bx).transpose(2, 1))
        want_boxes = R.decode_dynamic(want, torch.from_numpy(i8))
    wl = want["logits"].numpy()
    assert np.isnan(wl[1]).all() and not want["mask"][1].any()
    assert np.isnan(want_boxes[2]).all() and np.isfinite(want_boxes[[0, 1, 3]]).all()
    model = build_model("dynamic", sd)
    model.sampler = "numpy"
    np.random.seed(8)
    out = model(torch.from_numpy(p).cuda().transpose(2, 1), torch.from_numpy(bx).cuda().transpose(2, 1), None)
    np.random.seed(8)
    boxes = model.refine(torch.from_numpy(p).cuda().transpose(2, 1), torch.from_numpy(bx).cuda().transpose(2, 1),
                         torch.from_numpy(i8).cuda()).cpu().numpy()
    lg = out["logits"].cpu().numpy()
    assert np.array_equal(np.isnan(lg), np.isnan(wl))
    assert np.array_equal(out["mask"].cpu().numpy(), want["mask"].numpy())
    assert rel_err(lg[[0, 2, 3]], wl[[0, 2, 3]]) < TOL
    assert np.array_equal(np.isnan(boxes), np.isnan(want_boxes))
    assert rel_err(boxes[[0, 1, 3]], want_boxes[[0, 1, 3]]) < TOL
    for k in ("center", "heading_scores", "size_scores"):
        got, ref = out[k].cpu().numpy(), want[k].numpy()
        assert np.array_equal(np.isnan(got), np.isnan(ref)), k
        assert rel_err(got[[0, 1, 3]], ref[[0, 1, 3]]) < TOL, k


def test_empty_batches_and_empty_crops_are_rejected_with_einval():
    lib = hip.lib()
    model = build_model("static_one", synth.state_dict("static_one"))
    w = model._cache.get("ins_seg", model.ins_seg, hip.HEAD_INS_SEG)
    x = torch.zeros((2, 8, 3), device="cuda").transpose(2, 1)
    ws = torch.empty(lib.dal3_ins_seg_workspace_bytes(2), dtype=torch.uint8, device="cuda")
    lg, mk = torch.empty((2, 8, 2), device="cuda"), torch.empty((2, 8), dtype=torch.uint8, device="cuda")
    for B, N in ((2, 0), (0, 8), (-1, 8)):
        rc = lib.dal3_ins_seg_forward(hip.ptr(w), hip.F32, 3, hip.bcn(x), B, N, hip.ptr(lg), hip.ptr(mk), None, hip.ptr(ws),
                                      ws.numel(), hip.stream())
        assert rc == hip.EINVAL and b"positive" in lib.dal3_last_error()
    a = hip.StaticArgs()
    a.B, a.N = 3, 0
    a.workspace, a.workspace_bytes = hip.ptr(ws), ws.numel()
    assert lib.dal3_static_forward(C.byref(a), hip.PHASE_ALL, hip.stream()) == hip.EINVAL
    with pytest.raises(RuntimeError):
        model(torch.zeros((2, 3, 0), device="cuda"), torch.zeros((2, 7), device="cuda"), None)
