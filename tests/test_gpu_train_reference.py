"""N4 pinned to the REFERENCE's training step: tests/golden/train_step_*.npz hold one step of the real
StaticModelOneBoxEst / DynamicModel in train mode — forward, the reference's criterion, total_loss.backward(),
Adam — run by tests/golden/gen_train_step.py from the imported reference code (in float64, with the float32 run's
Dropout draw and NumPy draws recorded). Here the drop-in modules take the same step on the MI355X with the HIP
training kernels (train_backend "hip"): same weights, same inputs, the recorded Dropout multiplier and NumPy seed.
Bar: forward outputs, every loss term and the running statistics within 1e-4 (relative to the tensor's largest
entry); so are the gradients of everything behind the last ReLU of ins_seg (dconv5) and of the box heads.
The other gradients: within max(1e-4, 1.5 x the reference's own float32 error on that tensor, 4 / (B*N)).
Why not 1e-4 throughout: the step is not a smooth function. A ReLU input or two pooled candidates within float32
rounding of each other fall on one side in float64 and on the other in a float32 run, and ONE such activation among
the B*N points of this small batch moves the gradients in front of it by ~1/(B*N) of their scale. The reference's
own float32 CPU run is 2e-3 .. 2e-2 away from its float64 run on the static fixture for exactly this reason
(`f32_noise`), and this path lands on the same values there to three digits; on the dynamic fixture it is this
path's rounding that flips one gate of dconv2 (one row of its weight gradient, 6e-3; everything else in that tensor
1e-5). That the difference is such events and not semantics is pinned where it can be: in float64 on the CPU this
package's composite reproduces the fixture to 1e-6 (tests/test_host_dropin_train.py), and layer by layer on
tie-free data the HIP kernels match float64 autograd to 1e-4 (tests/test_gpu_train.py)."""
import importlib

import numpy as np
import pytest
import torch

from _common import build_model, golden, synth

losses = importlib.import_module("3dal_pytorch_amd.losses")
pytestmark = pytest.mark.gpu
TOL = 1e-4


def _case(kind, g):
    if kind in ("static_one", "static_two"):
        B, N = 8, 256
        seed = 41 if kind == "static_one" else 44
        pts, init, gt = synth.static_crops(B, N, seed=seed)
        labels = synth.loss_case(seed, batch=B, n_pts=N)[1]
        inp = dict(pts=pts, init=init, gt=gt)
    else:
        B, n_per = 4, 64
        pts, box, init8, gt = synth.dynamic_items(B, n_per_frame=n_per, seed=42)
        labels = synth.loss_case(42, batch=B, n_pts=5 * n_per)[1]
        inp = dict(pts=pts, box=box, gt=gt)
    s = sum(np.asarray(v, np.float64).sum() for v in inp.values())
    assert abs(s - float(g["in_sum"])) < 1e-9, "synthetic input generator drifted from the fixture"
    sd = synth.recentre_seg_bias(synth.state_dict(kind, seed=43), float(g["margin_shift"]))
    return inp, labels, sd


def _rel(a, ref, ref_max):
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    return float(np.abs(a - ref).max() / max(ref_max, 1e-30))


@pytest.mark.parametrize("kind", ["static_one", "static_two", "dynamic"])
def test_one_training_step_matches_the_reference(kind):
    g = golden("train_step_" + kind)
    inp, labels, sd = _case(kind, g)
    model = build_model(kind, sd).train()
    assert model.train_backend == "hip"
    model.sampler = "numpy"                                              # the reference's draws, in its order
    keep = np.unpackbits(g["drop_keep"], axis=1).astype(np.float32)
    model.drop_mask = torch.from_numpy(keep / (1.0 - model.ins_seg.dropout.p)).cuda()
    crit = (losses.FrustumPointNetLossOneBoxEst() if kind == "static_one" else
            losses.FrustumPointNetLossTwoBoxEst() if kind == "static_two" else losses.DynamicModelLoss())
    opt = torch.optim.Adam(model.parameters(), lr=float(g["lr"]))
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()     # noqa: E731
    np.random.seed(int(g["np_seed"]))
    if kind != "dynamic":
        out = model(dev(inp["pts"]).transpose(2, 1), dev(inp["init"]), dev(inp["gt"]))
    else:
        out = model(dev(inp["pts"]).transpose(2, 1), dev(inp["box"]).transpose(2, 1), dev(inp["gt"]))
    # forward: logits, the mask (hence the draws), the box parameters
    lg = out["logits"].detach().cpu().numpy()
    err = np.abs(lg - g["ref_logits"]).max()
    assert err / np.abs(g["ref_logits"]).max() < TOL
    assert float(g["min_abs_margin"]) > 20 * err
    assert np.array_equal(out["mask"].cpu().numpy(), g["mask"])
    for k in [k[8:] for k in g if k.startswith("ref_out_")]:
        ref = g["ref_out_" + k]
        if ref.dtype == np.int64:
            assert np.array_equal(out[k].cpu().numpy(), ref), k
        else:
            assert _rel(out[k].detach().cpu().numpy(), ref, max(np.abs(ref).max(), 1e-3)) < TOL, k
    # criterion: every entry of the reference's loss dict
    ls = crit(out, *[dev(a) for a in labels])
    for k in ls:
        assert abs(float(ls[k]) - float(g["ref_loss_" + k])) <= TOL * max(abs(float(g["ref_loss_" + k])), 1.0), k
    opt.zero_grad()
    ls["total_loss"].backward()
    params = dict(model.named_parameters())
    names = [k[len("ref_grad_"):] for k in g if k.startswith("ref_grad_")]
    assert len(names) >= 17
    noise = dict(zip([str(k) for k in g["f32_noise_keys"]], [float(v) for v in g["f32_noise"]]))
    worst = {}
    for name in names:
        got = synth.fixture_sample(params[name].grad.detach().cpu().numpy())
        ref, ref_max = g["ref_grad_" + name], float(g["refmax_grad_" + name])
        if ref_max < 1e-9:                                             # analytically zero (a shift that a BN removes)
            assert np.abs(got).max() < 1e-5, name
            continue
        worst[name] = _rel(got, ref, ref_max)
    table = {k: (round(v, 7), round(noise["grad_" + k], 7)) for k, v in sorted(worst.items(), key=lambda t: -t[1])}
    n_points = out["logits"].shape[0] * out["logits"].shape[1]
    # no ReLU / arg-max event in front of these (TwoBoxEst: stage two sits behind stage one's decoded box, so only
    # its last layer qualifies)
    smooth = (("ins_seg.dconv5.", "box_est.", "point_emb.fc", "box_emb.fc") if kind != "static_two" else
              ("ins_seg.dconv5.", "box_est_one.", "box_est_two.fc3"))
    bad = {k: v for k, v in table.items()
           if v[0] >= (TOL if k.startswith(smooth) else max(TOL, 1.5 * v[1], 4.0 / n_points))}
    import json, os
    if os.path.isdir("gpurun_out"):                                    # on the GPU box: keep the table for DESIGN.md
        json.dump(table, open(f"gpurun_out/train_ref_{kind}.json", "w"), indent=1)
    assert not bad, (bad, table)
    print("\n[train step vs reference] gradient error (this path, the reference's own float32 run):", table)
    # BatchNorm running statistics after the forward
    sdm = model.state_dict()
    for k in g:
        if k.startswith("ref_rm_") or k.startswith("ref_rv_"):
            key = k[7:] + (".running_mean" if k.startswith("ref_rm_") else ".running_var")
            ref = g[k]
            assert _rel(sdm[key].cpu().numpy(), ref, np.abs(ref).max()) < TOL, key
    # Adam's first step moves every weight by lr * g / (|g| + eps): sign-like, so an entry whose gradient is within
    # the gradient's own error band of zero can land on either side. Compare where the reference's gradient is clear
    # of that band.
    opt.step()
    after = dict(model.named_parameters())
    for name in names:
        ref_g, ref_max = g["ref_grad_" + name], float(g["refmax_grad_" + name])
        if ref_max < 1e-9:
            continue
        band = TOL if name.startswith(smooth) else max(TOL, 1.5 * noise["grad_" + name], 4.0 / n_points)
        clear = np.abs(ref_g) > 3 * band * ref_max
        new = synth.fixture_sample(after[name].detach().cpu().numpy())
        assert clear.mean() > 0.25, (name, clear.mean())
        assert np.abs(new - g["ref_new_" + name])[clear].max() < 2e-2 * float(g["lr"]) + 1e-6 * float(g["refmax_new_" + name]), name
