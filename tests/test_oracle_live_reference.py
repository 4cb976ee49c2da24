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
