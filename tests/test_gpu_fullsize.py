"""BASELINE.json's full sizes. Two kinds of check:
  - against the ORACLE on a stratified sample of the full-size run's rows (round 4): the oracle does ~80 static crops
    of 1024 points per second on the GPU box's cores (bench.py's cpu_baseline), so 256 of C2's 4096 crops cost a few
    seconds — logits at 1e-4, and the box parameters teacher-forced on the product's own drawn points at 1e-4;
  - size-independent properties over the whole batch: a slice of the big batch equals the same crops run alone, bit
    for bit; logits are invariant under a permutation of a crop's points; every output is finite; the standalone
    max-pool equals the fused one."""
import importlib

import numpy as np
import pytest
import torch

from _common import build_model, confident, recentred_sd, rel_err, synth
from oracle import ref_heads as R

TOL = 1e-4                                                  # BASELINE.json north_star: <= 1e-4 rel on fp32 box parameters

hip = importlib.import_module("3dal_pytorch_amd._hip")
pytestmark = pytest.mark.gpu


def _stratified_rows(B, n_rows, edge):
    """rows of a B-item launch that meet every part of it: the first and the last `edge` items (the first and the last
    round of workgroups, the ragged end of the persistent kernels' work lists) and an odd-stride walk over everything
    in between (odd: every residue mod 8 — the XCD a workgroup lands on — occurs)"""
    mid = n_rows - 2 * edge
    stride = ((B - 2 * edge) // mid) | 1
    rows = set(range(edge)) | set(range(B - edge, B)) | {edge + (stride * k) % (B - 2 * edge) for k in range(mid)}
    rows = np.array(sorted(rows))
    assert len(rows) == n_rows and len({int(r) % 8 for r in rows}) == 8 and rows[n_rows // 2] > B // 3
    return rows


def test_static_c2_full_size_vs_oracle_on_a_stratified_sample():
    """BASELINE.json configs[1] as bench.py runs it: StaticModelOneBoxEst, 4096 DISTINCT crops x 1024 points, fp32, the
    device sampler, one launch. 256 of its rows against the oracle: logits (1e-4 of their range), the mask wherever the
    margin is not within 1e-4 of a tie, and — the oracle's box estimator fed the very points the device drew for that
    row (`forced`) — centre, all 39 box parameters and the decoded (B,7) boxes per parameter group at 1e-4."""
    B, N = 4096, 1024
    pts_np, init_np, gt_np = synth.static_crops(B, N, seed=41)
    assert len({pts_np[i, :4].tobytes() for i in range(0, B, 7)}) == len(range(0, B, 7))      # distinct crops, not tilings
    sd = recentred_sd("static_one", pts_np[:8], 41)
    pts_np[::9] *= np.float32(0.85)                         # some crops with few segmented points (< 512: drawn with replacement)
    model = build_model("static_one", sd)
    o = model._run(torch.from_numpy(pts_np).cuda().transpose(2, 1), torch.from_numpy(init_np).cuda(),
                   torch.from_numpy(gt_np).cuda())
    rows = _stratified_rows(B, 256, 64)
    got = {k: o[k][torch.from_numpy(rows).cuda()].cpu().numpy() for k in ("logits", "mask", "counts", "obj_idx", "bp1", "c1", "boxes7")}
    counts = got["counts"]
    assert (counts < 512).any() and (counts >= 512).any()
    tsd = R.as_torch_sd(sd)
    p_t, i_t = torch.from_numpy(pts_np[rows]).transpose(2, 1), torch.from_numpy(init_np[rows])
    want = R.static_one_forward(tsd, p_t, i_t, forced=(torch.from_numpy(got["obj_idx"].astype(np.int64)), counts))
    wl = want["logits"].numpy()
    assert rel_err(got["logits"], wl) < TOL
    ok = confident(wl[:, :, 1] - wl[:, :, 0], np.abs(wl).max())
    assert ok.mean() > 0.99 and np.array_equal(got["mask"].astype(bool)[ok], want["mask"].numpy()[ok])
    # the device's draws are a legal outcome of gather_object_pts: drawn from ITS segmented points, count = their number
    gm = got["mask"].astype(bool)
    assert np.array_equal(counts, gm.sum(1))
    for r in np.nonzero(counts > 0)[0][::16]:
        assert gm[r][got["obj_idx"][r]].all()
    want_tail = np.concatenate([want["heading_scores"].numpy(), want["heading_residuals_normalized"].numpy(),
                                want["size_scores"].numpy(), want["size_residuals_normalized"].numpy().reshape(len(rows), 9)], 1)
    assert rel_err(got["bp1"][:, 3:], want_tail) < TOL
    assert rel_err(got["c1"], want["center"].numpy()) < TOL
    assert rel_err(got["boxes7"], R.decode_static(want, i_t, False)) < TOL


def test_dynamic_fp32_full_size_vs_oracle_on_a_stratified_sample():
    """DynamicModel at C3's item shape and count in the reference's arithmetic (1024 items x 5120 points + 101 boxes,
    fp32): 32 rows of the one launch against the oracle, teacher-forced on the device's draws."""
    B = 1024
    p, bx, i8, _ = synth.dynamic_items(B, seed=43)
    sd = recentred_sd("dynamic", p[:2], 43)
    model = build_model("dynamic", sd)
    o = model._run(torch.from_numpy(p).cuda().transpose(2, 1), torch.from_numpy(bx).cuda().transpose(2, 1),
                   init_box8=torch.from_numpy(i8).cuda())
    rows = _stratified_rows(B, 32, 8)
    sel = torch.from_numpy(rows).cuda()
    got = {k: o[k][sel].cpu().numpy() for k in ("logits", "mask", "counts", "obj_idx", "embedding", "bp", "boxes7")}
    want = R.dynamic_forward(R.as_torch_sd(sd), torch.from_numpy(p[rows]).transpose(2, 1), torch.from_numpy(bx[rows]).transpose(2, 1),
                             forced=(torch.from_numpy(got["obj_idx"].astype(np.int64)), got["counts"]))
    wl = want["logits"].numpy()
    assert rel_err(got["logits"], wl) < TOL
    ok = confident(wl[:, :, 1] - wl[:, :, 0], np.abs(wl).max())
    assert ok.mean() > 0.99 and np.array_equal(got["mask"].astype(bool)[ok], want["mask"].numpy()[ok])
    assert np.array_equal(got["counts"], got["mask"].astype(bool).sum(1))
    assert rel_err(got["embedding"], torch.cat([want["_point_e"], want["_box_e"]], 1).numpy()) < TOL
    want_bp = np.concatenate([want["center"].numpy(), want["heading_scores"].numpy(), want["heading_residuals_normalized"].numpy(),
                              want["size_scores"].numpy(), want["size_residuals_normalized"].numpy().reshape(len(rows), 9)], 1)
    assert rel_err(got["bp"], want_bp) < TOL
    assert rel_err(got["boxes7"], R.decode_dynamic(want, torch.from_numpy(i8[rows]))) < TOL


def test_static_c2_full_batch_properties():
    B, N = 4096, 1024
    pts_np, init_np, gt_np = synth.static_crops(B, N, seed=4)               # 4096 distinct crops
    model = build_model("static_two", synth.state_dict("static_two", seed=4))
    pts, init, gt = (torch.from_numpy(a).cuda() for a in (pts_np, init_np, gt_np))
    full = model._run(pts.transpose(2, 1), init, gt)
    for k in ("logits", "bp1", "bp2", "boxes7"):
        assert bool(torch.isfinite(full[k]).all()), k
    lo, hi = 1000, 1064
    model.item_offset = lo
    part = model._run(pts[lo:hi].transpose(2, 1), init[lo:hi], gt[lo:hi])
    model.item_offset = 0
    for k in ("logits", "mask", "obj_idx", "bp1", "box_one", "bp2", "boxes7"):
        assert torch.equal(part[k], full[k][lo:hi]), k
    perm = torch.from_numpy(np.argsort(synth.uniform(4, "perm", (N,)))).cuda()
    permuted = model._run(pts[:64][:, perm].transpose(2, 1), init[:64], gt[:64])
    assert torch.equal(permuted["logits"], full["logits"][:64][:, perm])


def test_dynamic_c3_shape_properties():
    B = 64                                                  # the C3 item shape (5 x 1024 points, 101 boxes)
    p, bx, i8, _ = synth.dynamic_items(B, seed=8)
    model = build_model("dynamic", synth.state_dict("dynamic", seed=8))
    dp, db, di = torch.from_numpy(p).cuda(), torch.from_numpy(bx).cuda(), torch.from_numpy(i8).cuda()
    full = model._run(dp.transpose(2, 1), db.transpose(2, 1), init_box8=di)
    assert bool(torch.isfinite(full["boxes7"]).all()) and full["logits"].shape == (B, 5120, 2)
    model.item_offset = 40
    part = model._run(dp[40:56].transpose(2, 1), db[40:56].transpose(2, 1), init_box8=di[40:56])
    model.item_offset = 0
    for k in ("logits", "embedding", "bp", "boxes7"):
        assert torch.equal(part[k], full[k][40:56]), k
    # the box window is max-pooled: reversing the 101 boxes cannot change the box embedding
    rev = model._run(dp.transpose(2, 1), db.flip(1).transpose(2, 1), init_box8=di)
    assert torch.equal(rev["embedding"][:, 256:], full["embedding"][:, 256:])


def test_dynamic_c3_full_size_bf16():
    """BASELINE.json configs[2] at its full size: 1024 items x 5120 points + 101 boxes, bf16. Finite; a shard run on
    its own (with its item offset) equals the same rows of the whole job bit for bit; a launch so small that every
    persistent workgroup sees a single group gives the same bits as the 1024-item launch."""
    B = 1024
    p, bx, i8, _ = synth.dynamic_items(B, seed=31)
    model = build_model("dynamic", recentred_sd("dynamic", p[:2], 31))
    model.precision = "bf16"
    dp, db, di = torch.from_numpy(p).cuda(), torch.from_numpy(bx).cuda(), torch.from_numpy(i8).cuda()
    full = model._run(dp.transpose(2, 1), db.transpose(2, 1), init_box8=di)
    assert full["logits"].shape == (B, 5120, 2)
    for k in ("logits", "embedding", "bp", "boxes7"):
        assert bool(torch.isfinite(full[k]).all()), k
    counts = full["counts"].cpu().numpy()
    assert counts.min() >= 0 and counts.max() <= 5120 and 0.2 < (full["mask"].float().mean().item()) < 0.8
    for lo, hi in ((0, 128), (384, 512), (1000, 1024), (517, 521)):      # ranks 0 and 3 of 8, a ragged tail, 4 items
        model.item_offset = lo
        part = model._run(dp[lo:hi].transpose(2, 1), db[lo:hi].transpose(2, 1), init_box8=di[lo:hi])
        for k in ("logits", "mask", "counts", "obj_idx", "embedding", "bp", "boxes7"):
            assert torch.equal(part[k], full[k][lo:hi]), (k, lo)
    model.item_offset = 0
    again = model._run(dp.transpose(2, 1), db.transpose(2, 1), init_box8=di)
    assert torch.equal(again["boxes7"], full["boxes7"]) and torch.equal(again["logits"], full["logits"])


@pytest.mark.parametrize("prec", ["fp16", "bf16"])
def test_static_c5_full_size_lowprec(prec):
    """BASELINE.json configs[4] per-GPU size: 2048 crops x 4096 points through the 16-bit MFMA kernels. Finite; shard
    == whole bit for bit; single-group launches == the persistent launch; logits invariant under a permutation of
    a crop's points."""
    B, N = 2048, 4096
    pts_np, init_np, gt_np = synth.static_crops(B, N, seed=32)                  # 2048 DISTINCT crops (not tilings)
    assert len({pts_np[i, :2].tobytes() for i in range(B)}) == B
    sd = recentred_sd("static_one", pts_np[:4], 32)
    # every ninth crop is shrunk so that only a few dozen of its points are segmented (the head then skips whole
    # tiles of copies; factor found with the oracle: 0.85 leaves 10..170 points of these crops), some to none at all
    pts_np[::9] *= np.float32(0.85)
    pts_np[4::27] *= np.float32(0.02)
    model = build_model("static_one", sd)
    model.precision = prec
    pts, init, gt = (torch.from_numpy(a).cuda() for a in (pts_np, init_np, gt_np))
    full = model._run(pts.transpose(2, 1), init, gt)
    for k in ("logits", "bp1", "boxes7"):
        assert bool(torch.isfinite(full[k]).all()), k
    counts = full["counts"].cpu().numpy()
    assert ((counts > 0) & (counts < 256)).any() and (counts == 0).any() and (counts > 512).any()
    for lo, hi in ((0, 256), (1792, 2048), (1001, 1003)):
        model.item_offset = lo
        part = model._run(pts[lo:hi].transpose(2, 1), init[lo:hi], gt[lo:hi])
        for k in ("logits", "mask", "counts", "obj_idx", "bp1", "boxes7"):
            assert torch.equal(part[k], full[k][lo:hi]), (k, lo)
    model.item_offset = 0
    perm = torch.from_numpy(np.argsort(synth.uniform(32, "perm", (N,)))).cuda()
    permuted = model._run(pts[:32][:, perm].transpose(2, 1), init[:32], gt[:32])
    assert torch.equal(permuted["logits"], full["logits"][:32][:, perm])


# ---------------------------------------------------------------------------------------------------------------------
# The two 16-bit BASELINE configurations at THEIR OWN SIZE against the oracle. The reference is fp32
# only, so the bars are tests/test_gpu_lowprec.py's BARS (each the measured worst case of the small-size cases with at
# most 2x headroom), applied here to a stratified sample of the full-size launch, plus a bar on the fraction of mask
# bits that differ from the exact-fp32 path's over the WHOLE batch. Inputs are stored in bf16 as BASELINE.json says
# ("bf16 storage"); the oracle is given the stored values, so both sides compute on identical inputs. The box
# estimators are teacher-forced on the exact-fp32 path's mask (`mask_override`: the device sampler, keyed on the item
# and the count, then draws the same points, which the oracle is handed through `forced`).
import emu16                                            # noqa: E402  (the model of 16-bit arithmetic)
from test_gpu_lowprec import BARS                       # noqa: E402  (the one table of 16-bit bars)

FULL_MEASURED = []


def _bar16(key, prec, value, where):
    value = float(value)
    FULL_MEASURED.append({"bar": key, "prec": prec, "value": value, "limit": BARS[key][prec], "where": where})
    import json
    import os
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump(FULL_MEASURED, open(os.path.join(out, "lowprec_fullsize_measured.json"), "w"), indent=1)
    assert value < BARS[key][prec], (key, prec, value, BARS[key][prec], where)


def _model16(key, prec, hip_err, model_err, where, n_bits=0):
    """the HIP path's error on these rows <= emu16.MODEL_SLACK x the error of the arithmetic itself (tests/emu16.py) on
    the SAME rows (+ four bits for mask-flip fractions)"""
    hip_err, model_err = float(hip_err), float(model_err)
    limit = (emu16.RMS_SLACK if key.endswith("_rms") else emu16.MODEL_SLACK) * model_err + (4.0 / n_bits if n_bits else 0.0)
    FULL_MEASURED.append({"bar": key + "_vs_model", "prec": prec, "value": hip_err, "model": model_err, "limit": limit,
                          "ratio": hip_err / model_err if model_err else None, "where": where})
    assert hip_err <= limit, (key, prec, "HIP", hip_err, "model of the arithmetic", model_err, where)


def _flips(a, b):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    return float(((a[..., 0] < a[..., 1]) != (b[..., 0] < b[..., 1])).float().mean())


def test_dynamic_c3_full_size_bf16_vs_oracle_on_a_stratified_sample():
    """BASELINE.json configs[2] as bench.py --config C3 runs it: DynamicModel, 1024 items x 5120 points + 101 boxes,
    bf16 storage and bf16 MFMA arithmetic, one launch; 32 stratified rows against the oracle."""
    B, prec = 1024, "bf16"
    p, bx, i8, _ = synth.dynamic_items(B, seed=44)
    sd = recentred_sd("dynamic", p[:2], 44)
    model = build_model("dynamic", sd)
    dp = torch.from_numpy(p).cuda().to(torch.bfloat16).transpose(2, 1)
    db = torch.from_numpy(bx).cuda().to(torch.bfloat16).transpose(2, 1)
    di = torch.from_numpy(i8).cuda()
    ref = model._run(dp, db, init_box8=di)                                    # the exact-fp32 arithmetic on the stored inputs
    model.precision = prec
    free = model._run(dp, db, init_box8=di)
    _bar16("mask_flip", prec, (free["mask"] != ref["mask"]).float().mean().item(), "C3 full size, all 1024 x 5120 bits vs fp32 path")
    forced = model._run(dp, db, init_box8=di, mask_override=ref["mask"])
    assert torch.equal(forced["obj_idx"], ref["obj_idx"]) and torch.equal(forced["counts"], ref["counts"])
    rows = _stratified_rows(B, 32, 8)
    sel = torch.from_numpy(rows).cuda()
    got = {k: forced[k][sel].cpu().numpy() for k in ("counts", "obj_idx", "embedding", "bp", "boxes7")}
    p_t, b_t = dp[sel].float().cpu(), db[sel].float().cpu()                    # the stored (bf16) values, as the oracle's fp32 input
    want = R.dynamic_forward(R.as_torch_sd(sd), p_t, b_t, forced=(torch.from_numpy(got["obj_idx"].astype(np.int64)), got["counts"]))
    wl = want["logits"].numpy()
    _bar16("logits", prec, rel_err(free["logits"][sel].cpu().numpy(), wl), "C3 full size, 32 rows vs oracle")
    tsd = R.as_torch_sd(sd)
    emu_lg = emu16.ins_seg(tsd, p_t, prec).numpy()                             # the same 32 rows in modelled bf16 arithmetic
    _model16("logits", prec, rel_err(free["logits"][sel].cpu().numpy(), wl), rel_err(emu_lg, wl), "C3 full size, 32 rows")
    _model16("logits_rms", prec, emu16.rms(free["logits"][sel].cpu(), wl), emu16.rms(emu_lg, wl), "C3 full size, 32 rows")
    _model16("mask_flip", prec, _flips(free["logits"][sel].cpu(), wl), _flips(emu_lg, wl), "C3 full size, 32 rows", n_bits=32 * 5120)
    margin = wl[:, :, 1] - wl[:, :, 0]
    sure = np.abs(margin) > 2 * BARS["logits"][prec] * np.abs(wl).max()
    assert sure.mean() > 0.5 and np.array_equal(free["mask"][sel].cpu().numpy().astype(bool)[sure], (margin > 0)[sure])
    _bar16("box", prec, rel_err(got["embedding"], torch.cat([want["_point_e"], want["_box_e"]], 1).numpy()), "C3 full size embedding vs oracle")
    emu_emb = torch.cat([emu16.embedding(tsd, want["_object_pts"].float(), "point_emb", prec), emu16.embedding(tsd, b_t, "box_emb", prec)], 1)
    want_emb = torch.cat([want["_point_e"], want["_box_e"]], 1).numpy()
    _model16("box", prec, rel_err(got["embedding"], want_emb), rel_err(emu_emb.numpy(), want_emb), "C3 full size embedding, 32 rows")
    want_bp = np.concatenate([want["center"].numpy(), want["heading_scores"].numpy(), want["heading_residuals_normalized"].numpy(),
                              want["size_scores"].numpy(), want["size_residuals_normalized"].numpy().reshape(len(rows), 9)], 1)
    _bar16("box_tail", prec, rel_err(got["bp"], want_bp), "C3 full size bp vs oracle")
    _model16("box_tail", prec, rel_err(got["bp"], want_bp), rel_err(R.dynamic_box_est(tsd, emu_emb).numpy(), want_bp), "C3 full size bp, 32 rows")
    same = (got["bp"][:, 3:15].argmax(1) == want_bp[:, 3:15].argmax(1)) & (got["bp"][:, 27:30].argmax(1) == want_bp[:, 27:30].argmax(1))
    assert same.mean() > 0.7                                                  # (a near-tie class may flip: a different bin centre)
    wb7 = R.decode_dynamic(want, torch.from_numpy(i8[rows]))
    _bar16("box_tail", prec, np.abs(got["boxes7"][same] - wb7[same]).max() / np.abs(wb7).max(), "C3 full size boxes7 vs oracle")


def test_static_c5_full_size_fp16_vs_oracle_on_a_stratified_sample():
    """BASELINE.json configs[4] per-GPU size as bench.py --config C5 runs it: StaticModelOneBoxEst, 2048 DISTINCT crops
    x 4096 points, bf16 storage, fp16 MFMA shared MLP, one launch; 48 stratified rows against the oracle."""
    B, N, prec = 2048, 4096, "fp16"
    pts_np, init_np, gt_np = synth.static_crops(B, N, seed=45)
    assert len({pts_np[i, :2].tobytes() for i in range(B)}) == B               # distinct crops, not tilings
    sd = recentred_sd("static_one", pts_np[:4], 45)
    pts_np[::9] *= np.float32(0.85)                                            # some crops with few segmented points
    model = build_model("static_one", sd)
    pts = torch.from_numpy(pts_np).cuda().to(torch.bfloat16).transpose(2, 1)
    init, gt = torch.from_numpy(init_np).cuda(), torch.from_numpy(gt_np).cuda()
    ref = model._run(pts, init, gt)
    model.precision = prec
    free = model._run(pts, init, gt)
    _bar16("mask_flip", prec, (free["mask"] != ref["mask"]).float().mean().item(), "C5 full size, all 2048 x 4096 bits vs fp32 path")
    forced = model._run(pts, init, gt, mask_override=ref["mask"])
    assert torch.equal(forced["obj_idx"], ref["obj_idx"])
    rows = _stratified_rows(B, 48, 12)
    sel = torch.from_numpy(rows).cuda()
    got = {k: forced[k][sel].cpu().numpy() for k in ("counts", "obj_idx", "bp1", "boxes7")}
    assert (got["counts"] < 512).any() and (got["counts"] >= 512).any()
    p_t, i_t = pts[sel].float().cpu(), torch.from_numpy(init_np[rows])
    want = R.static_one_forward(R.as_torch_sd(sd), p_t, i_t, forced=(torch.from_numpy(got["obj_idx"].astype(np.int64)), got["counts"]))
    wl = want["logits"].numpy()
    _bar16("logits", prec, rel_err(free["logits"][sel].cpu().numpy(), wl), "C5 full size, 48 rows vs oracle")
    tsd = R.as_torch_sd(sd)
    emu_lg = emu16.ins_seg(tsd, p_t, prec).numpy()                             # the same 48 rows in modelled fp16 arithmetic
    _model16("logits", prec, rel_err(free["logits"][sel].cpu().numpy(), wl), rel_err(emu_lg, wl), "C5 full size, 48 rows")
    _model16("logits_rms", prec, emu16.rms(free["logits"][sel].cpu(), wl), emu16.rms(emu_lg, wl), "C5 full size, 48 rows")
    _model16("mask_flip", prec, _flips(free["logits"][sel].cpu(), wl), _flips(emu_lg, wl), "C5 full size, 48 rows", n_bits=48 * 4096)
    margin = wl[:, :, 1] - wl[:, :, 0]
    sure = np.abs(margin) > 2 * BARS["logits"][prec] * np.abs(wl).max()
    assert sure.mean() > 0.5 and np.array_equal(free["mask"][sel].cpu().numpy().astype(bool)[sure], (margin > 0)[sure])
    wbp = np.concatenate([want["center_boxnet"].numpy(), want["heading_scores"].numpy(), want["heading_residuals_normalized"].numpy(),
                          want["size_scores"].numpy(), want["size_residuals_normalized"].numpy().reshape(len(rows), 9)], 1)
    _bar16("box", prec, rel_err(got["bp1"], wbp), "C5 full size bp1 vs oracle")
    _model16("box", prec, rel_err(got["bp1"], wbp), rel_err(emu16.static_box_est(tsd, want["_object_pts"].float(), prec).numpy(), wbp),
             "C5 full size bp1, 48 rows")
    same = (got["bp1"][:, 3:15].argmax(1) == wbp[:, 3:15].argmax(1)) & (got["bp1"][:, 27:30].argmax(1) == wbp[:, 27:30].argmax(1))
    assert same.mean() > 0.7
    wb7 = R.decode_static(want, i_t, False)
    _bar16("box_tail", prec, np.abs(got["boxes7"][same] - wb7[same]).max() / np.abs(wb7).max(), "C5 full size boxes7 vs oracle")


def test_fused_max_equals_standalone_maxpool():
    """global feature of ins_seg (max fused into conv5's epilogue) == dal3_maxpool_n over a materialised conv5
    output computed layer by layer with dal3_shared_mlp_layer"""
    import ctypes as C
    B, N = 4, 1024
    pts_np, _, _ = synth.static_crops(B, N, seed=2)
    sd = synth.state_dict("static_one", seed=2)
    model = build_model("static_one", sd)
    lib = hip.lib()
    x = torch.from_numpy(pts_np).cuda().transpose(2, 1)
    w = model._cache.get("ins_seg", model.ins_seg, hip.HEAD_INS_SEG)
    g = torch.zeros((B, 1024), device="cuda")
    hip.check(lib.dal3_ins_seg_encode(hip.ptr(w), hip.F32, 3, hip.bcn(x), B, N, hip.ptr(g), hip.stream()))
    cur = x
    for name, bn, ci, co in [("conv1", "bn1", 3, 64), ("conv2", "bn2", 64, 64), ("conv3", "bn3", 64, 64),
                             ("conv4", "bn4", 64, 128), ("conv5", "bn5", 128, 1024)]:
        L = hip.layer_struct(getattr(model.ins_seg, name), getattr(model.ins_seg, bn))
        ws = torch.empty(lib.dal3_shared_mlp_layer_workspace_bytes(ci, co), dtype=torch.uint8, device="cuda")
        y = torch.empty((B, N, co), device="cuda")
        hip.check(lib.dal3_shared_mlp_layer(C.byref(L), 1, hip.bcn(cur), B, N, hip.ptr(y), hip.ptr(ws), ws.numel(),
                                            hip.stream()))
        cur = y.transpose(2, 1)
    conv5 = cur.contiguous()                                 # (B,1024,N)
    out = torch.empty((B, 1024), device="cuda")
    hip.check(lib.dal3_maxpool_n(hip.ptr(conv5), B * 1024, N, hip.ptr(out), hip.stream()))
    # both are fp32 FMA chains over the same folded weights; only the position of the bias add differs
    assert float((out - g).abs().max() / g.abs().max()) < 2e-6


def test_c4_mixed_segment_sharded_over_8_equals_whole_job():
    """BASELINE.json configs[3]: one synthetic 198-frame segment = 64 static crops at N=4096 plus 40 dynamic tracks
    of rng.integers(20,199) frames (~4,400 track-frames), contiguous index ranges over 8 ranks, static and dynamic
    batches back to back. The 8 shards are run one after another on this GPU with the ranks' item offsets;
    their concatenation must equal the one-rank job bit for bit (what the all-gather would return)."""
    dal3_dist = importlib.import_module("3dal_pytorch_amd.dist")
    rng = np.random.default_rng(10922081)
    n_static, n_dynamic = 64, int(rng.integers(20, 199, size=40).sum())
    assert 3000 < n_dynamic < 6000
    sp, si, sg = synth.static_crops(n_static, 4096, seed=12)
    static = build_model("static_one", synth.state_dict("static_one", seed=12))
    s_in = [torch.from_numpy(a).cuda() for a in (sp, si, sg)]
    dp, db, di, _ = synth.dynamic_items(n_dynamic, seed=13)
    dynamic = build_model("dynamic", synth.state_dict("dynamic", seed=13))
    d_in = [torch.from_numpy(a).cuda() for a in (dp, db, di)]

    def run_static(lo, hi):
        static.item_offset = lo
        return static.refine(s_in[0][lo:hi].transpose(2, 1), s_in[1][lo:hi], s_in[2][lo:hi])

    def run_dynamic(lo, hi):
        dynamic.item_offset = lo
        return dynamic.refine(d_in[0][lo:hi].transpose(2, 1), d_in[1][lo:hi].transpose(2, 1), d_in[2][lo:hi])

    for n, run in ((n_static, run_static), (n_dynamic, run_dynamic)):
        whole = run(0, n)
        assert whole.shape == (n, 7) and bool(torch.isfinite(whole).all())
        parts = [run(*dal3_dist.shard_range(n, r, 8)) for r in range(8)]
        assert sum(p.shape[0] for p in parts) == n
        assert torch.equal(torch.cat(parts), whole)
