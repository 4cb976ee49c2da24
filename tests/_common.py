"""Shared helpers for the parity tests: rebuild the deterministic inputs/weights the golden
vectors were generated on (3dal_pytorch_amd/synth.py) and load the fixtures."""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pkg = importlib.import_module("3dal_pytorch_amd")
synth = importlib.import_module("3dal_pytorch_amd.synth")
arch = importlib.import_module("3dal_pytorch_amd.arch")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def static_case(kind, batch, n_pts, g):
    """-> (numpy state_dict re-centred as the fixture was, pts (B,3,N) view of point-major
    storage, init_box, bbox_gt) as CPU torch tensors."""
    pts_np, init_np, gt_np = synth.static_crops(batch, n_pts)
    s = pts_np.astype(np.float64).sum() + init_np.astype(np.float64).sum()
    assert abs(s - float(g["in_sum"])) < 1e-9, "synthetic input generator drifted from the fixtures"
    sd = synth.recentre_seg_bias(synth.state_dict(kind), float(g["margin_mean"]))
    pts = torch.from_numpy(pts_np).transpose(2, 1)
    return sd, pts, torch.from_numpy(init_np), torch.from_numpy(gt_np)


def dynamic_case(batch, g):
    pts_np, box_np, init8_np, gt_np = synth.dynamic_items(batch)
    s = pts_np.astype(np.float64).sum() + box_np.astype(np.float64).sum()
    assert abs(s - float(g["in_sum"])) < 1e-9, "synthetic input generator drifted from the fixtures"
    sd = synth.recentre_seg_bias(synth.state_dict("dynamic"), float(g["margin_mean"]))
    return (sd, torch.from_numpy(pts_np).transpose(2, 1), torch.from_numpy(box_np).transpose(2, 1),
            torch.from_numpy(init8_np), torch.from_numpy(gt_np))


def recentred_sd(kind, sample_pts_np, seed):
    """synth weights of `kind` with the segmentation bias re-centred (oracle logits of a small sample of the
    inputs) so that about half the points are segmented; with the raw random init every point lands in one class"""
    from oracle import ref_heads as R
    sd = synth.state_dict(kind, seed=seed)
    lg = R.ins_seg(R.as_torch_sd(sd), torch.from_numpy(np.ascontiguousarray(sample_pts_np)).transpose(2, 1))
    return synth.recentre_seg_bias(sd, float((lg[:, :, 1] - lg[:, :, 0]).mean()))


# column groups whose entries share a unit: a (.., 7) box is [centre (m) | size (m) | yaw (rad)], a (.., 39) box_pred is
# [centre | heading scores | normalised heading residuals | size scores | normalised size residuals]
BOX7_GROUPS = ((0, 3), (3, 6), (6, 7))
BOX_PRED_GROUPS = ((0, 3), (3, 15), (15, 27), (27, 30), (30, 39))


def rel_err(a, b):
    """max |a-b| / max|b| : the '<= 1e-4 relative' measure of BASELINE.json's north_star — taken PER PARAMETER GROUP
    for box tensors (normalising a (B,7) box by its largest entry mixes metres and radians, so 1e-4 of a
    10 m length allowed 1e-3 rad of yaw): the result is the largest of the groups' own relative errors. Everything
    else is one group."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    groups = {7: BOX7_GROUPS, 39: BOX_PRED_GROUPS}.get(b.shape[-1] if b.ndim >= 1 else -1)
    if groups is None or a.shape != b.shape:
        return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
    return max(float(np.abs(a[..., lo:hi] - b[..., lo:hi]).max() / max(np.abs(b[..., lo:hi]).max(), 1e-30))
               for lo, hi in groups)


# ------------------------------------------------------------------------------- GPU helpers
def build_model(kind, sd_np, device="cuda"):
    """The product nn.Module for 'static_one' | 'static_two' | 'dynamic' with the given weights."""
    sm = importlib.import_module("3dal_pytorch_amd.static_model")
    dm = importlib.import_module("3dal_pytorch_amd.dynamic_model")
    m = {"static_one": sm.StaticModelOneBoxEst, "static_two": sm.StaticModelTwoBoxEst,
         "dynamic": dm.DynamicModel}[kind]()
    m.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in sd_np.items()}, strict=True)
    return m.to(device).eval()


def positions_from_indices(mask_row, idx_row):
    """oracle point indices -> positions in the ordered list of segmented points (the `choice`
    the C ABI's DAL3_SAMPLER_CHOICE takes)."""
    pos = np.nonzero(np.asarray(mask_row))[0]
    lut = -np.ones(len(mask_row), np.int64)
    lut[pos] = np.arange(len(pos))
    return lut[np.asarray(idx_row)]


def confident(margin, scale, tol=1e-4):
    """points whose segmentation margin is far enough from the tie for a 1e-4-relative error in
    the logits not to flip the mask"""
    return np.abs(margin) > tol * scale
