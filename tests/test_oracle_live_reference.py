"""Oracle vs a LIVE import of the reference (only where /root/reference exists; skipped on the
GPU box). Widens the golden fixtures: other seeds, ragged N, an all-background crop."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from _common import ROOT, rel_err, synth
from oracle import ref_heads as R

REF = os.environ.get("DAL3_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "tools")),
                                reason="reference checkout not present")


@pytest.fixture(scope="module")
def ref():
    spec = importlib.util.spec_from_file_location("gen_golden", os.path.join(ROOT, "tests/golden/gen_golden.py"))
    gg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gg)
    mods = gg.import_reference()
    torch.set_grad_enabled(False)
    return gg, mods


@pytest.mark.parametrize("n,seed", [(700, 3), (1024, 4), (96, 5)])
def test_static_one_live(ref, n, seed):
    gg, (sm, dm, se, de, ut) = ref
    pts_np, init_np, gt_np = synth.static_crops(3, n, seed=seed)
    pts_np[2] *= 100.0                                 # drives one crop far off-distribution
    pts = torch.from_numpy(pts_np).transpose(2, 1)
    init, gt = torch.from_numpy(init_np), torch.from_numpy(gt_np)
    model = sm.StaticModelOneBoxEst(3, 3)
    sd, _ = gg.centred_sd("static_one", model, pts, seed)
    np.random.seed(99)
    want = model(pts, init, gt)
    np.random.seed(99)
    got = R.static_one_forward(R.as_torch_sd(sd), pts, init)
    assert np.array_equal(got["mask"].numpy(), want["mask"].numpy())
    for k, v in want.items():
        if k != "mask":
            assert rel_err(got[k].numpy(), v.numpy()) < 1e-6, k


def test_static_two_live(ref):
    gg, (sm, dm, se, de, ut) = ref
    pts_np, init_np, gt_np = synth.static_crops(3, 640, seed=8)
    pts = torch.from_numpy(pts_np).transpose(2, 1)
    init, gt = torch.from_numpy(init_np), torch.from_numpy(gt_np)
    model = sm.StaticModelTwoBoxEst(3, 3)
    sd, _ = gg.centred_sd("static_two", model, pts, 8)
    np.random.seed(5)
    want = model(pts, init, gt)
    np.random.seed(5)
    got = R.static_two_forward(R.as_torch_sd(sd), pts, init, gt)
    for k, v in want.items():
        if v.dtype in (torch.bool, torch.int64):
            assert np.array_equal(got[k].numpy(), v.numpy()), k
        else:
            assert rel_err(got[k].numpy(), v.numpy()) < 5e-6, k


def test_dynamic_live(ref):
    gg, (sm, dm, se, de, ut) = ref
    pts_np, box_np, init8_np, gt_np = synth.dynamic_items(2, n_per_frame=256, seed=21)
    pts = torch.from_numpy(pts_np).transpose(2, 1)
    box = torch.from_numpy(box_np).transpose(2, 1)
    model = dm.DynamicModel(3, 4)
    sd, _ = gg.centred_sd("dynamic", model, pts, 21)
    np.random.seed(6)
    want = model(pts, box, torch.from_numpy(gt_np))
    np.random.seed(6)
    got = R.dynamic_forward(R.as_torch_sd(sd), pts, box)
    for k, v in want.items():
        if k == "mask":
            assert np.array_equal(got[k].numpy(), v.numpy())
        else:
            assert rel_err(got[k].numpy(), v.numpy()) < 1e-6, k


# ---------------------------------------------------------------------------------- geometry, losses, datasets
@pytest.fixture(scope="module")
def ref_geometry(ref):
    gg, _ = ref
    return gg.import_reference_geometry()                  # det3d geometry.py / box_np_ops.py / waymo_common.py, numba as identity


@pytest.mark.parametrize("seed,dt", [(51, np.float32), (52, np.float64), (53, np.float32)])
def test_points_in_rbbox_live(ref_geometry, seed, dt):
    """other seeds than the fixture; the oracle AND the product's host-side plane code against the reference's own"""
    import importlib
    from oracle import ref_geom as G
    geo, ops, wc = ref_geometry
    pts, box9, _, _, _ = synth.sweep(seed, "live", n_points=3000, n_boxes=7)
    boxes = np.concatenate([box9[:, :3], box9[:, [4, 3, 5]], (-box9[:, -1:] - np.pi / 2)], 1).astype(dt)
    pts = pts.astype(dt)
    want = ops.points_in_rbbox(pts, boxes)
    assert np.array_equal(G.points_in_rbbox(pts, boxes), want)
    product = importlib.import_module("3dal_pytorch_amd.datasets")
    assert np.array_equal(product.points_in_rbbox(pts, boxes), want)
    geom = importlib.import_module("3dal_pytorch_amd.geom")
    nv, d = geo.surface_equ_3d_jitv2(ops.corner_to_surfaces_3d(
        ops.center_to_corner_box3d(boxes[:, :3], boxes[:, 3:6], boxes[:, -1]))[:, :, :3, :])
    planes = geom.box_planes(boxes)
    assert np.array_equal(planes[..., :3], nv) and np.array_equal(planes[..., 3], d)


def test_loss_modules_live(ref):
    """the product's criteria against the reference's on another seed (values and one gradient)"""
    import importlib
    gg, (sm, dm, se, de, ut) = ref
    losses = importlib.import_module("3dal_pytorch_amd.losses")
    torch.set_grad_enabled(True)
    try:
        for theirs, ours, two in ((sm.FrustumPointNetLossOneBoxEst(), losses.FrustumPointNetLossOneBoxEst(), False),
                                  (sm.FrustumPointNetLossTwoBoxEst(), losses.FrustumPointNetLossTwoBoxEst(), True),
                                  (dm.DynamicModelLoss(), losses.DynamicModelLoss(), False)):
            out_np, labels_np = synth.loss_case(77, two_stage=two, batch=5, n_pts=40)
            res = []
            for crit in (theirs, ours):
                out_t = {k: torch.from_numpy(v).requires_grad_(v.dtype == np.float32) for k, v in out_np.items()}
                l = crit(out_t, *[torch.from_numpy(a) for a in labels_np], w_box=0.7)
                g, = torch.autograd.grad(l["total_loss"], [out_t["logits"]])
                res.append(({k: float(v.detach()) for k, v in l.items()}, g.numpy()))
            assert res[0][0].keys() == res[1][0].keys()
            for k, v in res[0][0].items():
                assert abs(res[1][0][k] - v) <= 1e-6 * abs(v), k
            assert np.allclose(res[1][1], res[0][1], rtol=1e-5, atol=1e-9)
    finally:
        torch.set_grad_enabled(False)
