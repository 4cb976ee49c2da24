"""Pins the oracle (oracle/ref_heads.py) to outputs of the real reference: the fixtures in
tests/golden/ were produced by tests/golden/gen_golden.py from a live import of
/root/reference/tools/{static,dynamic}_model.py. CPU only; runs everywhere."""
import numpy as np
import pytest
import torch

from _common import arch, dynamic_case, golden, rel_err, static_case
from oracle import ref_heads as R

TOL = 2e-6   # oracle vs reference, same torch-CPU kernels, differing only in op grouping

DICT_KEYS_ONE = ["logits", "center_boxnet", "heading_scores", "heading_residuals_normalized",
                 "heading_residuals", "size_scores", "size_residuals_normalized", "size_residuals", "center"]


@pytest.mark.parametrize("tag,b,n", [("static_one_b4_n1024", 4, 1024), ("static_one_b1_n512", 1, 512)])
def test_static_one_matches_reference(tag, b, n):
    g = golden(tag)
    sd, pts, init, _ = static_case("static_one", b, n, g)
    sd = R.as_torch_sd(sd)
    np.random.seed(int(g["rng_seed"]))
    out = R.static_one_forward(sd, pts, init)
    assert np.array_equal(out["mask"].numpy(), g["mask"])
    assert np.array_equal(out["_indices"].numpy(), g["indices"])       # same NumPy stream order
    assert np.array_equal(out["_object_pts"].numpy(), g["object_pts"])
    for k in DICT_KEYS_ONE:
        assert out[k].shape == g[k].shape, k
        assert rel_err(out[k].numpy(), g[k]) < TOL, k
    _, gf = R.ins_seg(sd, pts, want_global=True)
    assert rel_err(gf.numpy(), g["global_feat"]) < TOL
    assert rel_err(R.decode_static(out, init, two_stage=False), g["boxes7"]) < TOL


def test_static_two_matches_reference():
    g = golden("static_two_b4_n1024")
    sd, pts, init, gt = static_case("static_two", 4, 1024, g)
    sd = R.as_torch_sd(sd)
    np.random.seed(int(g["rng_seed"]))
    out = R.static_two_forward(sd, pts, init, gt)
    assert np.array_equal(out["_indices"].numpy(), g["indices"])
    for k, v in out.items():
        if k.startswith("_"):
            continue
        ref = g[k]
        assert tuple(v.shape) == ref.shape, k
        if v.dtype in (torch.bool, torch.int64):
            assert np.array_equal(v.numpy(), ref), k
        else:
            assert rel_err(v.numpy(), ref) < 5e-6, k
    assert out["heading_class_label_two"].dtype == torch.int64
    assert rel_err(R.decode_static(out, init, two_stage=True), g["boxes7"]) < 5e-6


def test_dynamic_matches_reference():
    g = golden("dynamic_b2")
    sd, pts, box, init8, _ = dynamic_case(2, g)
    sd = R.as_torch_sd(sd)
    np.random.seed(int(g["rng_seed"]))
    out = R.dynamic_forward(sd, pts, box)
    assert np.array_equal(out["mask"].numpy(), g["mask"])
    assert np.array_equal(out["_indices"].numpy(), g["indices"])
    for k in ["logits", "center", "heading_scores", "heading_residuals_normalized", "heading_residuals",
              "size_scores", "size_residuals_normalized", "size_residuals"]:
        assert rel_err(out[k].numpy(), g[k]) < TOL, k
    assert rel_err(out["_point_e"].numpy(), g["point_e"]) < TOL
    assert rel_err(out["_box_e"].numpy(), g["box_e"]) < TOL
    assert rel_err(R.decode_dynamic(out, init8), g["boxes7"]) < TOL


def test_gather_rng_call_order():
    """counts {0,1,300,511,512,700,N}: the zero row consumes no RNG and stays zero; the stream
    position after the call equals the reference's."""
    from _common import synth
    g = golden("gather_rng")
    pts = torch.from_numpy(synth.static_crops(len(g["counts"]), 1024, seed=5)[0]).transpose(2, 1)
    np.random.seed(12345)
    obj, idx = R.gather_object_pts(pts, torch.from_numpy(g["mask"]), 512)
    assert np.random.randint(0, 1 << 30) == int(g["next_draw"])
    assert np.array_equal(idx.numpy(), g["indices"])
    assert np.array_equal(obj.numpy(), g["object_pts"])
    assert not obj[0].any() and not idx[0].any()
    # count < M: every positive index appears at least once (top-up draws only duplicates)
    for row, c in enumerate(g["counts"]):
        if 0 < c < 512:
            assert set(idx[row].tolist()) == set(np.nonzero(g["mask"][row])[0].tolist())
        if c >= 512:
            assert len(set(idx[row].tolist())) == 512


def test_class_tables():
    g = golden("class_tables")
    for a, (cid, res) in zip(g["angles"], g["a2c"]):
        c, r = R.angle2class(float(a), 12)
        assert c == int(cid) and abs(r - res) < 1e-12
    for ci in range(12):
        for rj, r in enumerate(g["res"]):
            assert abs(R.class2angle(ci, float(r), 12) - g["c2a"][ci, rj]) < 1e-12
    for s, cid, res in zip(g["sizes"], g["s2c_cls"], g["s2c_res"]):
        c, r = R.size2class(s)
        assert c == int(cid) and np.allclose(r, res, atol=1e-12)
    for c in range(3):
        assert np.allclose(R.class2size(c, np.array([0.1, -0.2, 0.3])), g["c2s"][c], atol=1e-12)


@pytest.mark.parametrize("kind", ["static_one", "static_two", "dynamic"])
def test_state_dict_key_set_matches_reference(kind):
    g = golden("state_dict_keys")
    specs = arch.model_param_specs(kind)
    assert [k for k, _ in specs] == list(g[kind + "_keys"])
    assert [str(tuple(s)) for _, s in specs] == list(g[kind + "_shapes"])


def test_fold_bn_equals_unfolded():
    g = golden("static_one_b1_n512")
    sd, pts, _, _ = static_case("static_one", 1, 512, g)
    sd = R.as_torch_sd(sd)
    w, b = R.fold_bn(sd, "ins_seg", "conv1", "bn1")
    y = torch.relu(torch.matmul(w, pts) + b[None, :, None])
    y_ref = R._cbr(sd, "ins_seg", "conv1", "bn1", pts)
    assert rel_err(y.numpy(), y_ref.numpy()) < 1e-6
