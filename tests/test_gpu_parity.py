"""Parity of the HIP path (through the C ABI of lib3dal_hip.so) with the oracle and with the
golden vectors generated from the real reference. Tolerance: BASELINE.json north_star —
<= 1e-4 relative on fp32 outputs (measured: ~1e-6); masks, indices and max-pool are exact."""
import ctypes as C
import importlib

import numpy as np
import pytest
import torch

from _common import (build_model, confident, dynamic_case, golden, positions_from_indices, rel_err,
                     static_case, synth)
from oracle import ref_heads as R

hip = importlib.import_module("3dal_pytorch_amd._hip")
heads = importlib.import_module("3dal_pytorch_amd._heads")
pytestmark = pytest.mark.gpu
TOL = 1e-4
MARGIN_FACTOR = 20          # how many times the measured logit error the fixtures' smallest |margin| must be (f16x3: 10, test_gpu_x3.py)


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x)).cuda()


# ------------------------------------------------------------------------------- max-pool (exact)
@pytest.mark.parametrize("rows,n", [(8 * 1024, 1024), (1000, 1000), (37, 1023), (5, 3), (64, 5120), (3, 101)])
def test_maxpool_exact(rows, n):
    x = synth.normal(1, f"mp{rows}x{n}", (rows, n)).astype(np.float32)
    xd, out = dev(x), torch.empty(rows, device="cuda")
    hip.check(hip.lib().dal3_maxpool_n(hip.ptr(xd), rows, n, hip.ptr(out), hip.stream()))
    assert np.array_equal(out.cpu().numpy(), x.max(1))


def test_maxpool_bcn_shape_and_negatives():
    x = -np.abs(synth.normal(2, "neg", (4, 1024, 260))).astype(np.float32) - 1.0   # all negative
    xd, out = dev(x), torch.empty((4, 1024), device="cuda")
    hip.check(hip.lib().dal3_maxpool_n(hip.ptr(xd), 4 * 1024, 260, hip.ptr(out), hip.stream()))
    assert np.array_equal(out.cpu().numpy(), x.max(2))


# ------------------------------------------------------------------------------- one layer
@pytest.mark.parametrize("c_in,c_out,n", [(3, 64, 512), (4, 64, 100), (8, 64, 101), (64, 128, 300),
                                           (128, 1024, 64), (256, 512, 33)])
def test_shared_mlp_layer(c_in, c_out, n):
    B = 3
    sd = {
        "p.c.weight": synth.uniform(9, "w", (c_out, c_in, 1), -0.3, 0.3).astype(np.float32),
        "p.c.bias": synth.uniform(9, "b", (c_out,), -0.1, 0.1).astype(np.float32),
        "p.bn.weight": synth.uniform(9, "g", (c_out,), 0.5, 1.5).astype(np.float32),
        "p.bn.bias": synth.normal(9, "be", (c_out,), 0, 0.2).astype(np.float32),
        "p.bn.running_mean": synth.normal(9, "m", (c_out,), 0, 0.5).astype(np.float32),
        "p.bn.running_var": synth.uniform(9, "v", (c_out,), 0.5, 2.0).astype(np.float32),
    }
    x = synth.normal(9, "x", (B, n, c_in)).astype(np.float32)           # point-major storage
    want = R._cbr(R.as_torch_sd(sd), "p", "c", "bn", torch.from_numpy(x).transpose(2, 1)).numpy()
    t = {k: dev(v) for k, v in sd.items()}
    L = hip.Layer(hip.ptr(t["p.c.weight"]), hip.ptr(t["p.c.bias"]), hip.ptr(t["p.bn.weight"]),
                  hip.ptr(t["p.bn.bias"]), hip.ptr(t["p.bn.running_mean"]), hip.ptr(t["p.bn.running_var"]),
                  c_in, c_out)
    lib = hip.lib()
    ws = torch.empty(lib.dal3_shared_mlp_layer_workspace_bytes(c_in, c_out), dtype=torch.uint8, device="cuda")
    xd = dev(x).transpose(2, 1)
    y = torch.empty((B, n, c_out), device="cuda")
    hip.check(lib.dal3_shared_mlp_layer(C.byref(L), 1, hip.bcn(xd), B, n, hip.ptr(y), hip.ptr(ws), ws.numel(),
                                        hip.stream()))
    assert rel_err(y.cpu().numpy().transpose(0, 2, 1), want) < 1e-5
    # contiguous (B,C,N) input takes the same path through the strides
    xc = dev(np.ascontiguousarray(x.transpose(0, 2, 1)))
    y2 = torch.empty_like(y)
    hip.check(lib.dal3_shared_mlp_layer(C.byref(L), 1, hip.bcn(xc), B, n, hip.ptr(y2), hip.ptr(ws), ws.numel(),
                                        hip.stream()))
    assert torch.equal(y, y2)


# ------------------------------------------------------------------------------- ins_seg
def _ins_seg(model, pts):
    lib = hip.lib()
    B, c_in, N = pts.shape
    w = model._cache.get("ins_seg", model.ins_seg, hip.HEAD_INS_SEG)
    ws = torch.empty(lib.dal3_ins_seg_workspace_bytes(B), dtype=torch.uint8, device="cuda")
    logits = torch.empty((B, N, 2), device="cuda")
    mask = torch.empty((B, N), dtype=torch.uint8, device="cuda")
    g = torch.empty((B, 1024), device="cuda")
    hip.check(lib.dal3_ins_seg_forward(hip.ptr(w), hip.F32, c_in, hip.bcn(pts), B, N, hip.ptr(logits), hip.ptr(mask),
                                       hip.ptr(g), hip.ptr(ws), ws.numel(), hip.stream()))
    return logits.cpu().numpy(), mask.cpu().numpy().astype(bool), g.cpu().numpy()


@pytest.mark.parametrize("tag,b,n", [("static_one_b4_n1024", 4, 1024), ("static_one_b1_n512", 1, 512)])
def test_ins_seg_static_vs_reference_golden(tag, b, n):
    g = golden(tag)
    sd, pts, _, _ = static_case("static_one", b, n, g)
    model = build_model("static_one", sd)
    logits, mask, gf = _ins_seg(model, pts.cuda())
    assert rel_err(gf, g["global_feat"]) < TOL
    assert rel_err(logits, g["logits"]) < TOL
    assert np.array_equal(mask, g["mask"])           # fixture margins are >> the fp32 error


def test_ins_seg_dynamic_vs_reference_golden():
    g = golden("dynamic_b2")
    sd, pts, _, _, _ = dynamic_case(2, g)
    model = build_model("dynamic", sd)
    logits, mask, _ = _ins_seg(model, pts.cuda())
    assert rel_err(logits, g["logits"]) < TOL
    margin = g["logits"][:, :, 1] - g["logits"][:, :, 0]
    ok = confident(margin, np.abs(g["logits"]).max())
    assert np.array_equal(mask[ok], g["mask"][ok])


@pytest.mark.parametrize("n", [1, 5, 31, 33, 96, 700, 1000, 4096])
def test_ins_seg_ragged_n_vs_oracle(n):
    B = 3
    pts_np, _, _ = synth.static_crops(B, n, seed=n)
    sd = synth.state_dict("static_one", seed=n)
    want = R.ins_seg(R.as_torch_sd(sd), torch.from_numpy(pts_np).transpose(2, 1)).numpy()
    model = build_model("static_one", sd)
    logits, _, _ = _ins_seg(model, dev(pts_np).transpose(2, 1))
    assert rel_err(logits, want) < TOL
    # truly contiguous (B,3,N) input
    logits2, _, _ = _ins_seg(model, dev(np.ascontiguousarray(pts_np.transpose(0, 2, 1))))
    assert np.array_equal(logits, logits2)


def test_ins_seg_point_permutation_is_bitwise_invariant():
    """per-point work + exact max: permuting a crop's points permutes its logits bit for bit"""
    B, N = 8, 1024
    pts_np, _, _ = synth.static_crops(B, N, seed=77)
    model = build_model("static_one", synth.state_dict("static_one"))
    perm = np.argsort(synth.uniform(5, "perm", (N,)))
    a, _, ga = _ins_seg(model, dev(pts_np).transpose(2, 1))
    b, _, gb = _ins_seg(model, dev(pts_np[:, perm]).transpose(2, 1))
    assert np.array_equal(ga, gb)
    assert np.array_equal(a[:, perm], b)


# ------------------------------------------------------------------------------- gather
def _gather(mask, pts, M, sampler, choice=None, seed=1, item_offset=0):
    lib = hip.lib()
    B, C_, N = pts.shape
    ws = torch.empty(lib.dal3_gather_workspace_bytes(B, N), dtype=torch.uint8, device="cuda")
    counts = torch.empty(B, dtype=torch.int32, device="cuda")
    idx = torch.empty((B, M), dtype=torch.int32, device="cuda")
    obj = torch.empty((B, M, C_), device="cuda")
    md = dev(mask.astype(np.uint8))
    ch = dev(choice.astype(np.int32)) if choice is not None else None
    hip.check(lib.dal3_mask_compact_sample(hip.ptr(md), hip.bcn(pts), B, N, C_, M, sampler, hip.ptr(ch), seed,
                                           item_offset, hip.ptr(counts), hip.ptr(idx), hip.ptr(obj), hip.ptr(ws),
                                           ws.numel(), hip.stream()))
    return counts.cpu().numpy(), idx.cpu().numpy(), obj.cpu().numpy()


def test_gather_choice_mode_reproduces_reference_draws():
    g = golden("gather_rng")
    pts = dev(synth.static_crops(len(g["counts"]), 1024, seed=5)[0]).transpose(2, 1)
    np.random.seed(12345)
    choice = heads.numpy_choice(g["counts"], 512)
    counts, idx, obj = _gather(g["mask"], pts, 512, hip.SAMPLER_CHOICE, choice)
    assert np.array_equal(counts, g["counts"])
    assert np.array_equal(idx, g["indices"])
    assert np.array_equal(obj.transpose(0, 2, 1), g["object_pts"])
    assert not obj[0].any()                                              # count 0 -> zero row


def test_gather_device_sampler_properties():
    g = golden("gather_rng")
    pts_np = synth.static_crops(len(g["counts"]), 1024, seed=5)[0]
    pts = dev(pts_np).transpose(2, 1)
    counts, idx, obj = _gather(g["mask"], pts, 512, hip.SAMPLER_DEVICE, seed=42)
    assert np.array_equal(counts, g["counts"])
    for row, c in enumerate(g["counts"]):
        pos = set(np.nonzero(g["mask"][row])[0].tolist())
        if c == 0:
            assert not obj[row].any()
            continue
        assert np.array_equal(obj[row], pts_np[row][idx[row]])
        if c < 512:
            assert set(idx[row].tolist()) == pos                         # every segmented point kept
        else:
            assert len(set(idx[row].tolist())) == 512 and set(idx[row].tolist()) <= pos
    # deterministic, seed-dependent, and keyed on the GLOBAL item index (shard == whole job)
    _, idx2, _ = _gather(g["mask"], pts, 512, hip.SAMPLER_DEVICE, seed=42)
    assert np.array_equal(idx, idx2)
    _, idx3, _ = _gather(g["mask"], pts, 512, hip.SAMPLER_DEVICE, seed=43)
    assert not np.array_equal(idx[5], idx3[5])
    _, idx4, _ = _gather(g["mask"][4:], pts[4:], 512, hip.SAMPLER_DEVICE, seed=42, item_offset=4)
    assert np.array_equal(idx[4:], idx4)


def _hash_key(seed, item, i):
    """csrc/dal3_misc.hip hash_key (a splitmix64 finaliser of seed, global item index and rank), top 32 bits"""
    m = (1 << 64) - 1
    z = (seed ^ (item * 0x9E3779B97F4A7C15) ^ ((i << 32) | i)) & m
    z = (z + 0x9E3779B97F4A7C15) & m
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
    return ((z ^ (z >> 31)) & m) >> 32


def test_gather_device_sampler_takes_the_smallest_keys():
    """count >= M: exactly the M segmented points with the smallest hash keys (keyed on seed, GLOBAL item index and
    the point's rank among the segmented ones) — restated on the host, for two point counts, with
    an item offset, and with M = 2560 of 5120 (the dynamic head's shape)"""
    for B, N, M, seed, off in ((6, 1024, 512, 42, 0), (3, 5120, 2560, 10922081, 17)):
        mask = synth.uniform(8, f"dsk{N}", (B, N)) < 0.8
        mask[0] = True
        pts = dev(synth.static_crops(1, N, seed=5)[0]).expand(B, N, 3).transpose(2, 1)
        counts, idx, _ = _gather(mask, pts, M, hip.SAMPLER_DEVICE, seed=seed, item_offset=off)
        for b in range(B):
            pos = np.nonzero(mask[b])[0]
            assert counts[b] == len(pos) >= M
            keys = np.array([_hash_key(seed, off + b, i) for i in range(len(pos))], dtype=np.uint64)
            assert len(np.unique(keys)) == len(keys)                     # no ties in these draws
            order = np.argsort(keys, kind="stable")[:M]
            want = np.concatenate([pos[np.sort(order[:-1])], pos[order[-1:]]])    # keys below the M-th in index order,
            assert np.array_equal(idx[b], want), (N, b)                         # then the M-th itself (the "ties")


def test_gather_device_sampler_is_uniform():
    """count = 1024, M = 512: every point is kept with probability 1/2 over many items"""
    B, N = 512, 1024
    mask = np.ones((B, N), bool)
    pts = dev(synth.static_crops(1, N, seed=5)[0]).expand(B, N, 3).transpose(2, 1)
    _, idx, _ = _gather(mask, pts, 512, hip.SAMPLER_DEVICE, seed=7)
    freq = np.bincount(idx.ravel(), minlength=N) / B
    assert abs(freq.mean() - 0.5) < 1e-9 and freq.std() < 0.04 and freq.min() > 0.35 and freq.max() < 0.65


@pytest.mark.parametrize("N,M,C", [(1024, 512, 3), (4096, 512, 3), (5120, 2560, 4), (700, 512, 3), (33, 512, 3), (7936, 512, 3),
                                   (7937, 512, 3), (300, 101, 8)])
def test_the_lds_sampler_equals_the_global_memory_sampler_bitwise(N, M, C):
    """round 6: items of up to 7936 points are compacted and sampled with their positions and hash keys in LDS (one block
    scan over contiguous pieces instead of one per 256 mask bytes, keys hashed once); DAL3_BCN_NO_LDS_SAMPLER selects the
    original kernel. Same definition, same order: counts, indices and gathered points must agree bit for bit — masks of
    every density incl. empty, a single point, count == M - 1 / M / M + 1 and all points; the device sampler with two seeds
    and an item offset, and the CHOICE sampler; N = 7937 takes the original kernel either way (the dispatch edge)."""
    rng = np.random.default_rng(N * 31 + M)
    B = 24
    dens = rng.random(B)
    mask = rng.random((B, N)) < dens[:, None]
    mask[0] = False
    mask[1] = False
    mask[1, N // 2] = True
    mask[2] = True
    for row, k in ((3, M - 1), (4, M), (5, M + 1)):
        mask[row] = False
        if 0 < k <= N:
            mask[row, rng.permutation(N)[:k]] = True
    pts_np = rng.standard_normal((B, N, C)).astype(np.float32)
    pts = dev(pts_np).transpose(2, 1)
    choice = rng.integers(-3, N + 5, size=(B, M))
    for sampler, kw in ((hip.SAMPLER_DEVICE, dict(seed=42)), (hip.SAMPLER_DEVICE, dict(seed=10922081, item_offset=1000)),
                        (hip.SAMPLER_CHOICE, dict(choice=choice))):
        got = _gather(mask, pts, M, sampler, **kw)
        hip.DISPATCH_FLAGS = hip.BCN_NO_LDS_SAMPLER
        try:
            want = _gather(mask, pts, M, sampler, **kw)
        finally:
            hip.DISPATCH_FLAGS = 0
        for a, b, name in zip(got, want, ("counts", "obj_idx", "obj_pts")):
            assert np.array_equal(a, b), (name, sampler, N)
        assert np.array_equal(got[0], mask.sum(1))
        assert (got[1] >= 0).all() and (got[1] < N).all()
        live = got[0] > 0
        assert np.array_equal(got[2][live], np.stack([pts_np[r][got[1][r]] for r in np.nonzero(live)[0]]))


# ------------------------------------------------------------------------------- full models
ONE_KEYS = ["logits", "center_boxnet", "heading_scores", "heading_residuals_normalized",
            "heading_residuals", "size_scores", "size_residuals_normalized", "size_residuals", "center"]


@pytest.mark.parametrize("tag,b,n", [("static_one_b4_n1024", 4, 1024), ("static_one_b1_n512", 1, 512)])
def test_static_one_forward_vs_reference_golden(tag, b, n):
    g = golden(tag)
    sd, pts, init, gt = static_case("static_one", b, n, g)
    model = build_model("static_one", sd)
    model.sampler = "numpy"
    np.random.seed(int(g["rng_seed"]))
    out = model(pts.cuda(), init.cuda(), gt.cuda())
    assert set(out) == set(ONE_KEYS + ["mask"])
    assert out["mask"].dtype == torch.bool and out["mask"].is_cuda
    assert np.array_equal(out["mask"].cpu().numpy(), g["mask"])
    assert np.array_equal(model.last["obj_idx"].cpu().numpy(), g["indices"])
    for k in ONE_KEYS:
        assert tuple(out[k].shape) == g[k].shape and out[k].dtype == torch.float32, k
        assert rel_err(out[k].cpu().numpy(), g[k]) < TOL, k
    assert rel_err(model.last["boxes7"].cpu().numpy(), g["boxes7"]) < TOL


def test_static_two_forward_vs_reference_golden():
    g = golden("static_two_b4_n1024")
    sd, pts, init, gt = static_case("static_two", 4, 1024, g)
    model = build_model("static_two", sd)
    model.sampler = "numpy"
    np.random.seed(int(g["rng_seed"]))
    out = model(pts.cuda(), init.cuda(), gt.cuda())
    ref_keys = [k for k in g if k not in ("rng_seed", "margin_mean", "in_sum", "indices", "boxes7")]
    assert set(out) == set(ref_keys)
    assert np.array_equal(model.last["obj_idx"].cpu().numpy(), g["indices"])
    for k in ref_keys:
        v = out[k].cpu().numpy()
        assert v.shape == g[k].shape, k
        if g[k].dtype == np.bool_ or g[k].dtype == np.int64:
            assert out[k].dtype == (torch.bool if g[k].dtype == np.bool_ else torch.int64), k
            assert np.array_equal(v, g[k]), k
        else:
            assert rel_err(v, g[k]) < TOL, k
    assert rel_err(model.last["boxes7"].cpu().numpy(), g["boxes7"]) < TOL


def test_dynamic_forward_vs_reference_golden():
    g = golden("dynamic_b2")
    sd, pts, box, init8, gt = dynamic_case(2, g)
    model = build_model("dynamic", sd)
    # teacher-force the reference's mask and draws (a flip of a near-tie point would change the subset)
    choice = np.stack([positions_from_indices(g["mask"][i], g["indices"][i]) for i in range(2)])
    o = model._run(pts.cuda(), box.cuda(), init_box8=init8.cuda(), choice=torch.from_numpy(choice),
                   mask_override=torch.from_numpy(g["mask"]))
    assert np.array_equal(o["obj_idx"].cpu().numpy(), g["indices"])
    assert rel_err(o["embedding"][:, :256].cpu().numpy(), g["point_e"]) < TOL
    assert rel_err(o["embedding"][:, 256:].cpu().numpy(), g["box_e"]) < TOL
    bp = o["bp"].cpu().numpy()
    assert rel_err(bp[:, 0:3], g["center"]) < TOL
    assert rel_err(bp[:, 3:15], g["heading_scores"]) < TOL
    assert rel_err(bp[:, 27:30], g["size_scores"]) < TOL
    assert rel_err(o["hr"].cpu().numpy(), g["heading_residuals"]) < TOL
    assert rel_err(o["sr"].cpu().numpy(), g["size_residuals"]) < TOL
    assert rel_err(o["boxes7"].cpu().numpy(), g["boxes7"]) < TOL
    # and the free-running module API: keys, dtypes, and agreement wherever the mask is not a near-tie
    model.sampler = "numpy"
    np.random.seed(int(g["rng_seed"]))
    out = model(pts.cuda(), box.cuda(), gt.cuda())
    assert set(out) == {"logits", "mask", "center", "heading_scores", "heading_residuals_normalized",
                        "heading_residuals", "size_scores", "size_residuals_normalized", "size_residuals"}
    assert rel_err(out["logits"].cpu().numpy(), g["logits"]) < TOL
    # the fixture's weights are centred on the widest gap between sorted margins (synth.widest_gap_centre): every
    # point is >= min_abs_margin from the tie, far above the fp32 error of the logits, so the free-running mask,
    # the draws and every box parameter must agree — unconditionally
    err = np.abs(out["logits"].cpu().numpy() - g["logits"]).max()
    assert float(g["min_abs_margin"]) > MARGIN_FACTOR * err, (float(g["min_abs_margin"]), err)
    assert np.array_equal(out["mask"].cpu().numpy(), g["mask"])
    assert np.array_equal(model.last["obj_idx"].cpu().numpy(), g["indices"])
    for k in ("center", "heading_scores", "heading_residuals_normalized", "heading_residuals", "size_scores",
              "size_residuals_normalized", "size_residuals"):
        assert out[k].shape == g[k].shape and rel_err(out[k].cpu().numpy(), g[k]) < TOL, k
    np.random.seed(int(g["rng_seed"]))
    assert rel_err(model.refine(pts.cuda(), box.cuda(), init8.cuda()).cpu().numpy(), g["boxes7"]) < TOL


# ------------------------------------------------------------------------------- bigger batches vs oracle
def test_static_one_b64_teacher_forced_vs_oracle():
    B, N = 64, 1024
    pts_np, init_np, gt_np = synth.static_crops(B, N, seed=3)
    pts_np[5] *= 50.0                                                     # an all-background-ish outlier crop
    sd = synth.state_dict("static_one", seed=3)
    tsd = R.as_torch_sd(sd)
    pts_t = torch.from_numpy(pts_np).transpose(2, 1)
    lg = R.ins_seg(tsd, pts_t)
    sd = synth.recentre_seg_bias(sd, float((lg[:, :, 1] - lg[:, :, 0]).mean()))
    tsd = R.as_torch_sd(sd)
    np.random.seed(4)
    want = R.static_one_forward(tsd, pts_t, torch.from_numpy(init_np))
    wmask = want["mask"].numpy()
    counts = wmask.sum(1)
    assert (counts == 0).any() or (counts < 512).any()
    choice = np.stack([positions_from_indices(wmask[i], want["_indices"][i].numpy()) if counts[i] else
                       np.zeros(512, np.int64) for i in range(B)])
    model = build_model("static_one", sd)
    o = model._run(dev(pts_np).transpose(2, 1), dev(init_np), dev(gt_np), choice=torch.from_numpy(choice),
                   mask_override=want["mask"])
    assert rel_err(o["logits"].cpu().numpy(), want["logits"].numpy()) < TOL
    margin = (want["logits"][:, :, 1] - want["logits"][:, :, 0]).numpy()
    ok = confident(margin, np.abs(want["logits"].numpy()).max())
    lg_d = o["logits"]                                # (the kernel's own mask was overwritten by the override)
    seg = (lg_d[:, :, 0] < lg_d[:, :, 1]).cpu().numpy()
    assert np.array_equal(seg[ok], wmask[ok])
    want_tail = np.concatenate([want["heading_scores"].numpy(), want["heading_residuals_normalized"].numpy(),
                                want["size_scores"].numpy(), want["size_residuals_normalized"].numpy().reshape(B, 9)], 1)
    assert rel_err(o["bp1"].cpu().numpy()[:, 3:], want_tail) < TOL
    assert rel_err(o["c1"].cpu().numpy(), want["center"].numpy()) < TOL
    assert rel_err(o["boxes7"].cpu().numpy(), R.decode_static(want, torch.from_numpy(init_np), False)) < TOL


@pytest.mark.parametrize("B,N,cuts", [(48, 1024, (0, 16, 48)), (1070, 96, (0, 130, 1000, 1070))])
def test_shard_equals_whole_job_bitwise(B, N, cuts):
    """object crops are independent: rows [a,b) run alone (item_offset=a) == the same rows of the full run, at a
    small and at a four-digit, ragged batch size"""
    pts_np, init_np, gt_np = synth.static_crops(B, N, seed=12)
    sd = synth.state_dict("static_two", seed=12)
    model = build_model("static_two", sd)
    full = model.refine(dev(pts_np).transpose(2, 1), dev(init_np), dev(gt_np)).cpu().numpy()
    parts = []
    for a, b in zip(cuts, cuts[1:]):
        model.item_offset = a
        parts.append(model.refine(dev(pts_np[a:b]).transpose(2, 1), dev(init_np[a:b]), dev(gt_np[a:b])).cpu().numpy())
    model.item_offset = 0
    assert np.array_equal(np.concatenate(parts), full)
    assert np.isfinite(full).all()


# ------------------------------------------------------------------------------- decode / recentre
def test_decode_boxes_vs_oracle_including_wrap():
    B = 256
    bp = synth.normal(21, "bp", (B, 39)).astype(np.float32)
    bp[:12, 3:15] = -5.0
    for i in range(12):                                                   # force every heading class once
        bp[i, 3 + i] = 5.0
        bp[i, 15 + i] = 0.9                                               # classes >= 6 wrap past pi
    init = synth.static_crops(B, 4, seed=2)[1]
    c, hs, hrn, hr, ss, srn, sr = R.parse_box_pred(torch.from_numpy(bp))
    out = {"heading_scores": hs, "heading_residuals": hr, "size_scores": ss, "size_residuals": sr,
           "center": c + torch.from_numpy(init)[:, :3]}
    want = R.decode_static(out, torch.from_numpy(init), two_stage=False)
    bpd, initd = dev(bp), dev(init)
    hrd, srd, cd, b7 = (torch.empty((B, 12), device="cuda"), torch.empty((B, 9), device="cuda"),
                        torch.empty((B, 3), device="cuda"), torch.empty((B, 7), device="cuda"))
    hip.check(hip.lib().dal3_decode_boxes(hip.ptr(bpd), B, hip.ptr(initd), 7, 0, None, 0,
                                          C.c_void_p(initd.data_ptr() + 24), 7, hip.ptr(hrd), hip.ptr(srd),
                                          hip.ptr(cd), hip.ptr(b7), hip.stream()))
    assert np.array_equal(hrd.cpu().numpy(), hr.numpy())
    assert np.array_equal(srd.cpu().numpy().reshape(B, 3, 3), sr.numpy())
    assert np.array_equal(cd.cpu().numpy(), out["center"].numpy())
    assert np.abs(b7.cpu().numpy() - want).max() < 1e-6
    assert (want[:12, 6] - init[:12, 6] <= np.pi + 1e-6).all()


def test_recenter_rotz_vs_oracle():
    g = golden("static_two_b4_n1024")
    sd, pts, init, gt = static_case("static_two", 4, 1024, g)
    np.random.seed(int(g["rng_seed"]))
    want = R.static_two_forward(R.as_torch_sd(sd), pts, init, gt)
    obj = dev(want["_object_pts"].numpy().transpose(0, 2, 1))
    out = torch.empty_like(obj)
    hcl = torch.empty(4, dtype=torch.int64, device="cuda")
    hrl = torch.empty(4, device="cuda")
    init_d, one_d, gt_d = init.cuda(), dev(g["box_one"]), gt.cuda()      # keep alive: ptr() holds no reference
    hip.check(hip.lib().dal3_recenter_rotz(hip.ptr(obj), 4, 512, hip.ptr(init_d), hip.ptr(one_d), hip.ptr(gt_d),
                                           hip.ptr(out), hip.ptr(hcl), hip.ptr(hrl), hip.stream()))
    assert rel_err(out.cpu().numpy().transpose(0, 2, 1), want["_object_pts_two"].numpy()) < 1e-5
    assert np.array_equal(hcl.cpu().numpy(), g["heading_class_label_two"])
    assert np.abs(hrl.cpu().numpy() - g["heading_residuals_label_two"]).max() < 1e-6


# ------------------------------------------------------------------------------- API behaviour
def test_errors_are_loud():
    lib = hip.lib()
    x = torch.zeros((2, 64, 3), device="cuda").transpose(2, 1)
    assert lib.dal3_ins_seg_forward(None, hip.F32, 3, hip.bcn(x), 2, 64, None, None, None, None, 0, hip.stream()) == hip.EINVAL
    model = build_model("static_one", synth.state_dict("static_one"))
    with pytest.raises(RuntimeError):
        model(torch.zeros((2, 4, 64), device="cuda"), torch.zeros((2, 7), device="cuda"), None)
    a = hip.StaticArgs()
    a.B, a.N = 2, 64
    a.workspace, a.workspace_bytes = hip.ptr(torch.empty(16, dtype=torch.uint8, device="cuda")), 16
    assert lib.dal3_static_forward(C.byref(a), hip.PHASE_ALL, hip.stream()) == hip.EWORKSPACE
    assert b"workspace" in lib.dal3_last_error()


def test_packed_cache_tracks_parameter_updates():
    g = golden("static_one_b1_n512")
    sd, pts, init, gt = static_case("static_one", 1, 512, g)
    model = build_model("static_one", synth.state_dict("static_one", seed=999))
    a = model(pts.cuda(), init.cuda(), gt.cuda())["logits"].clone()
    model.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in sd.items()})
    b = model(pts.cuda(), init.cuda(), gt.cuda())["logits"]
    assert not torch.equal(a, b)
    assert rel_err(b.cpu().numpy(), g["logits"]) < TOL


def test_skipping_duplicate_object_points_is_exact():
    """device sampler: crops with fewer than 512 segmented points get copies as filler and the head skips them;
    feeding the very same indices through the CHOICE path (which computes all 512) must give identical bits"""
    B, N = 48, 1024
    pts_np, init_np, gt_np = synth.static_crops(B, N, seed=15)
    sd = synth.state_dict("static_two", seed=15)
    lg = R.ins_seg(R.as_torch_sd(sd), torch.from_numpy(pts_np[:8]).transpose(2, 1))
    sd = synth.recentre_seg_bias(sd, float((lg[:, :, 1] - lg[:, :, 0]).mean()))
    model = build_model("static_two", sd)
    pts, init, gt = dev(pts_np).transpose(2, 1), dev(init_np), dev(gt_np)
    a = model._run(pts, init, gt)                                         # device sampler, duplicates skipped
    counts = a["counts"].cpu().numpy()
    assert ((counts > 0) & (counts < 512)).any() and (counts >= 512).any()
    mask = a["mask"].cpu().numpy().astype(bool)
    idx = a["obj_idx"].cpu().numpy()
    choice = np.stack([positions_from_indices(mask[i], idx[i]) if counts[i] else np.zeros(512, np.int64)
                       for i in range(B)])
    b = model._run(pts, init, gt, choice=torch.from_numpy(choice))       # same points, nothing skipped
    assert torch.equal(a["obj_idx"], b["obj_idx"])
    for k in ("bp1", "box_one", "bp2", "boxes7"):
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("n", [1, 7, 40])
def test_whole_static_forward_on_tiny_crops_vs_oracle(n):
    """crops with fewer points than one MFMA tile / than the 512 object points: the whole forward incl. sampling with
    replacement and decode, teacher-forced on the oracle's draws"""
    B = 3
    pts_np, init_np, _ = synth.static_crops(B, n, seed=100 + n)
    sd = synth.state_dict("static_one", seed=100 + n)
    np.random.seed(9)
    want = R.static_one_forward(R.as_torch_sd(sd), torch.from_numpy(pts_np).transpose(2, 1), torch.from_numpy(init_np))
    model = build_model("static_one", sd)
    model.sampler = "numpy"
    np.random.seed(9)
    got = model(dev(pts_np).transpose(2, 1), dev(init_np), None)
    assert np.array_equal(got["mask"].cpu().numpy(), want["mask"].numpy())
    assert rel_err(got["logits"].cpu().numpy(), want["logits"].numpy()) < TOL
    for k in ("center", "heading_scores", "size_scores", "size_residuals"):
        assert rel_err(got[k].cpu().numpy(), want[k].numpy()) < TOL, k


@pytest.mark.parametrize("prec", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("kind,head,c,B,M", [("static_one", "box_est", 3, 300, 512), ("dynamic", "point_emb", 4, 70, 2560),
                                             ("dynamic", "box_emb", 8, 200, 101),
                                             # round 6, the mid-size route (guided FIRST run): about one tile per wave (the
                                             # reference's eval batch, static_eval.py:299), fewer tiles than the chip holds
                                             # waves, and a list a few times the grid (first runs of 1, the rest by cursor)
                                             ("static_one", "box_est", 3, 64, 512), ("static_one", "box_est", 3, 40, 512),
                                             ("static_one", "box_est", 3, 150, 512), ("dynamic", "point_emb", 4, 64, 2560)])
def test_point_head_on_the_live_tile_worklist_equals_the_per_tile_launch_bitwise(kind, head, c, B, M, prec):
    """round 3: large jobs run the point heads as persistent waves over the compacted list of tiles that hold distinct
    points (dal3_point_head_pool with a workspace) instead of one workgroup per (item, tile) (without one). Same
    per-point arithmetic, same atomicMax combine: the pooled features must agree bit for bit — with counts of distinct
    points on and around every tile boundary, empty items, and no counts at all (every point distinct)."""
    lib = hip.lib()
    model = build_model(kind, synth.state_dict(kind, seed=23))
    mod = getattr(model, head)
    dt = hip.DTYPES[prec]
    w = model._cache.get(head, mod, mod.HEAD_KIND, dt)
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.standard_normal((B, M, c)).astype(np.float32)).cuda()
    special = np.array([0, 1, 31, 32, 33, 63, 64, 65, M - 1, M, M + 7, -3])
    d_np = rng.integers(0, M + 1, size=B).astype(np.int32)
    d_np[:len(special)] = special
    # a real sampler repeats the first `count` points beyond them: what the kernels rely on when they skip
    xs = x.clone()
    for b in range(B):
        k = int(min(max(d_np[b], 1), M))
        xs[b, k:] = xs[b, torch.arange(M - k) % k]
    ws = torch.empty(int(lib.dal3_point_head_pool_workspace_bytes(B, M)), dtype=torch.uint8, device="cuda")
    assert lib.dal3_point_head_pool_workspace_bytes(B, M) >= B * ((M + 31) // 32) * 16
    bcn = hip.bcn(xs.transpose(2, 1))
    for distinct in (torch.from_numpy(d_np).cuda(), None):
        f_list, f_tile = torch.empty((B, 512), device="cuda"), torch.empty((B, 512), device="cuda")
        hip.check(lib.dal3_point_head_pool(mod.HEAD_KIND, hip.ptr(w), dt, bcn, B, M, hip.ptr(distinct), hip.ptr(f_list),
                                           hip.ptr(ws), ws.numel(), hip.stream()))
        hip.check(lib.dal3_point_head_pool(mod.HEAD_KIND, hip.ptr(w), dt, bcn, B, M, hip.ptr(distinct), hip.ptr(f_tile),
                                           None, 0, hip.stream()))
        # ... and with the workspace given but DAL3_BCN_NO_WORKLIST in the view's flags (the per-call dispatch hint)
        f_flag = torch.empty((B, 512), device="cuda")
        bcn_flag = hip.bcn(xs.transpose(2, 1))
        bcn_flag.flags = hip.BCN_NO_WORKLIST
        hip.check(lib.dal3_point_head_pool(mod.HEAD_KIND, hip.ptr(w), dt, bcn_flag, B, M, hip.ptr(distinct), hip.ptr(f_flag),
                                           hip.ptr(ws), ws.numel(), hip.stream()))
        torch.cuda.synchronize()
        assert torch.equal(f_list, f_tile) and torch.equal(f_list, f_flag)
        assert float(f_list.abs().max()) > 0
    # skipping the copies changes nothing: the run with counts equals the run that computes every point
    f_all = torch.empty((B, 512), device="cuda")
    hip.check(lib.dal3_point_head_pool(mod.HEAD_KIND, hip.ptr(w), dt, bcn, B, M, None, hip.ptr(f_all), hip.ptr(ws),
                                       ws.numel(), hip.stream()))
    f_cnt = torch.empty((B, 512), device="cuda")
    hip.check(lib.dal3_point_head_pool(mod.HEAD_KIND, hip.ptr(w), dt, bcn, B, M, hip.ptr(torch.from_numpy(d_np).cuda()),
                                       hip.ptr(f_cnt), hip.ptr(ws), ws.numel(), hip.stream()))
    torch.cuda.synchronize()
    assert torch.equal(f_all, f_cnt)
