#!/usr/bin/env python3
"""Generate tests/golden/train_step_*.npz: ONE training step of the REAL reference (jacky121298/3DAL_PyTorch) —
`model.train()`, forward, criterion, `total_loss.backward()`, `optimizer.step()` exactly as the loop of
tools/static_train.py:65-88 / tools/dynamic_train.py does it — on the deterministic inputs of
3dal_pytorch_amd/synth.py. Run only where /root/reference exists:
    python tests/golden/gen_train_step.py

The reference's modules are imported (never copied) through the shim of gen_golden.py. Two runs per model:
  * as the reference runs (float32 CPU): kept as `f32_*`, it shows the reference's own rounding noise;
  * the SAME code in float64 (torch default dtype float64, `Tensor.float()` made a no-op cast to double for the
    duration of the run — the reference hard-codes `.float()` on the gathered object points): the values the HIP
    training path is compared with (tests/test_gpu_train_reference.py), `ref_*`.
What makes the step reproducible elsewhere is stored with it: the Dropout multiplier of ins_seg (bit-packed; the
reference draws it from torch's CPU generator), the object-point indices that gather_object_pts drew from the NumPy
stream (np.random.seed(s) right before forward), and the segmentation-bias shift that puts the decision threshold
into the widest gap between sorted margins (so that no implementation flips a mask bit and with it the draws).
"""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
synth = importlib.import_module("3dal_pytorch_amd.synth")
import gen_golden as G                                        # noqa: E402  (the import shim lives there)

LR = 1e-3                                                     # static_train.py:219 / dynamic_train.py: Adam(lr=0.001)
PICK = {                                                      # parameters whose gradient / updated value is stored
    "static_one": ["ins_seg.conv1.weight", "ins_seg.bn1.weight", "ins_seg.bn1.bias", "ins_seg.conv3.weight",
                   "ins_seg.conv5.weight", "ins_seg.bn5.weight", "ins_seg.bn5.bias", "ins_seg.dconv1.weight",
                   "ins_seg.dbn1.bias", "ins_seg.dconv2.weight", "ins_seg.dconv4.weight", "ins_seg.dbn4.weight",
                   "ins_seg.dconv5.weight", "ins_seg.dconv5.bias", "box_est.conv1.weight", "box_est.conv4.weight",
                   "box_est.bn4.weight", "box_est.fc1.weight", "box_est.fcbn1.weight", "box_est.fc2.weight",
                   "box_est.fc3.weight", "box_est.fc3.bias"],
    "static_two": ["ins_seg.conv2.weight", "ins_seg.conv5.weight", "ins_seg.dconv1.weight", "ins_seg.dconv5.weight",
                   "ins_seg.dconv5.bias", "box_est_one.conv1.weight", "box_est_one.conv4.weight", "box_est_one.fc1.weight",
                   "box_est_one.fc3.weight", "box_est_one.fc3.bias", "box_est_two.conv1.weight", "box_est_two.conv3.weight",
                   "box_est_two.bn4.bias", "box_est_two.fc2.weight", "box_est_two.fcbn2.weight", "box_est_two.fc3.weight",
                   "box_est_two.fc3.bias"],
    "dynamic": ["ins_seg.conv1.weight", "ins_seg.conv5.weight", "ins_seg.bn5.bias", "ins_seg.dconv1.weight",
                "ins_seg.dconv3.weight", "ins_seg.dconv5.weight", "ins_seg.dconv5.bias", "point_emb.conv1.weight",
                "point_emb.conv4.weight", "point_emb.fc1.weight", "point_emb.fcbn2.bias", "box_emb.conv1.weight",
                "box_emb.conv4.weight", "box_emb.fc2.weight", "box_est.fc1.weight", "box_est.fcbn1.weight",
                "box_est.fc3.weight", "box_est.fc3.bias"],
}
STATS = {"static_two": ["ins_seg.bn3", "ins_seg.dbn1", "box_est_one.bn1", "box_est_one.fcbn1", "box_est_two.bn4",
                        "box_est_two.fcbn2"],
         "static_one": ["ins_seg.bn1", "ins_seg.bn5", "ins_seg.dbn2", "ins_seg.dbn4", "box_est.bn2", "box_est.bn4",
                        "box_est.fcbn1", "box_est.fcbn2"],
         "dynamic": ["ins_seg.bn2", "ins_seg.bn5", "ins_seg.dbn4", "point_emb.bn4", "point_emb.fcbn1", "box_emb.bn3",
                     "box_emb.fcbn2", "box_est.fcbn2"]}


# The LARGE fixtures (VERDICT r2 / ADVICE r2): the same step at B*N >= 65,536 points. ONE activation that falls on the other
# side of a ReLU (or of a pooling tie) in a float32 run moves the gradients in front of it by ~1/(B*N) of their scale;
# at the small fixtures' 1,280 - 2,048 points that alone is 5e-4 ... 8e-4 and their gate had to allow for it, at 65,536
# points it is 1.5e-5 and the gate is max(1e-4, 1.5 x the reference's own float32 error) with nothing else in it.
# kind -> (model kind, items, points per item (per frame for the dynamic head), synth seed)
BIG = {"static_one_big": ("static_one", 16, 4096, 51), "static_two_big": ("static_two", 16, 4096, 54),
       "dynamic_big": ("dynamic", 13, 1024, 52)}


def big_case(kind):
    base, B, n, seed = BIG[kind]
    if base != "dynamic":
        pts, init, gt = synth.static_crops(B, n, seed=seed)
        return dict(pts=pts, init=init, gt=gt), synth.loss_case(seed, batch=B, n_pts=n)[1]
    pts, box, init8, gt = synth.dynamic_items(B, n_per_frame=n, seed=seed)
    return dict(pts=pts, box=box, gt=gt), synth.loss_case(seed, batch=B, n_pts=5 * n)[1]


def big_keep(kind, n_points):
    """the Dropout keep pattern of a large fixture: a function of the seed (synth's hash), not torch's generator — the
    float64 run is given a forced pattern anyway (torch draws another one for float64 tensors), and a (65536, 128)
    pattern regenerated from four numbers need not be stored"""
    return synth.uniform(BIG[kind][3], "dropout_keep", (n_points, 128)) >= 0.5


def case(kind):
    """inputs (numpy, float32) and labels of the step"""
    if kind in BIG:
        return big_case(kind)
    if kind in ("static_one", "static_two"):
        B, N = 8, 256
        seed = 41 if kind == "static_one" else 44
        pts, init, gt = synth.static_crops(B, N, seed=seed)
        labels = synth.loss_case(seed, batch=B, n_pts=N)[1]
        return dict(pts=pts, init=init, gt=gt), labels
    B, n_per = 4, 64
    pts, box, init8, gt = synth.dynamic_items(B, n_per_frame=n_per, seed=42)
    labels = synth.loss_case(42, batch=B, n_pts=5 * n_per)[1]
    return dict(pts=pts, box=box, gt=gt), labels


def one_step(kind, mods, sd_np, inp, labels, dtype, torch_seed, np_seed, force_keep=None):
    """forward + criterion + backward + Adam step of the reference module in `dtype`; returns a dict of numpy arrays.
    force_keep (B*N,128) bool: replace nn.Dropout's own draw by this keep pattern (torch draws a different pattern
    for float64 tensors from the same seed, so the float64 run is given the float32 run's)"""
    sm, dm = mods
    full = kind
    kind = BIG[kind][0] if kind in BIG else kind
    saved_float, saved_default = torch.Tensor.float, torch.get_default_dtype()
    if dtype == torch.float64:
        torch.set_default_dtype(torch.float64)
        torch.Tensor.float = lambda self, *a, **k: self.double()
    try:
        model = (sm.StaticModelOneBoxEst(3, 3) if kind == "static_one" else
                 sm.StaticModelTwoBoxEst(3, 3) if kind == "static_two" else dm.DynamicModel(3, 4))
        model.load_state_dict({k: torch.as_tensor(v) for k, v in sd_np.items()}, strict=True)
        model = model.to(dtype).train()
        crit = (sm.FrustumPointNetLossOneBoxEst() if kind == "static_one" else
                sm.FrustumPointNetLossTwoBoxEst() if kind == "static_two" else dm.DynamicModelLoss())
        opt = torch.optim.Adam(model.parameters(), lr=LR)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dtype)          # noqa: E731
        drop = {}
        def hook(m, i, o):
            if force_keep is not None:
                B_, C_, N_ = i[0].shape
                keep_t = torch.from_numpy(force_keep).reshape(B_, N_, C_).permute(0, 2, 1).to(i[0].dtype)
                o = i[0] * keep_t / (1.0 - m.p)
            drop.update(x=i[0].detach(), y=o.detach())
            return o
        h = model.ins_seg.dropout.register_forward_hook(hook)
        # the layers torch.max pools over points (ins_seg.bn5, the point stacks' bn4): how close the two largest values
        # of an (item, channel) row come — a pooling tie within float32 rounding reroutes that row's gradient
        gaps, pool_hooks = [], []
        def gap_hook(m, i, o):
            top = torch.topk(o.detach().double(), 2, dim=2).values
            live = top[:, :, 0] > 0                                            # (a row whose maximum is <= 0 is gated off by the ReLU)
            if bool(live.any()):
                gaps.append(float(((top[:, :, 0] - top[:, :, 1]) / top[:, :, 0].abs().clamp_min(1e-30))[live].min()))
        for name, mod in model.named_modules():
            if name in ("ins_seg.bn5", "box_est.bn4", "box_est_one.bn4", "box_est_two.bn4", "point_emb.bn4", "box_emb.bn4"):
                pool_hooks.append(mod.register_forward_hook(gap_hook))
        torch.manual_seed(torch_seed)
        np.random.seed(np_seed)
        if kind != "dynamic":
            out = model(t(inp["pts"]).transpose(2, 1), t(inp["init"]), t(inp["gt"]))
        else:
            out = model(t(inp["pts"]).transpose(2, 1), t(inp["box"]).transpose(2, 1), t(inp["gt"]))
        h.remove()
        for ph in pool_hooks:
            ph.remove()
        lab = [torch.from_numpy(a).to(dtype) if a.dtype == np.float32 else torch.from_numpy(a) for a in labels]
        losses = crit(out, *lab)
        opt.zero_grad()
        losses["total_loss"].backward()
        params = dict(model.named_parameters())
        res = {"logits": out["logits"].detach().numpy(), "mask": out["mask"].numpy(),
               "min_pool_gap": np.float64(min(gaps) if gaps else np.inf)}
        for k, v in losses.items():
            res["loss_" + k] = np.float64(v.detach())
        for k in (("center", "heading_scores", "size_scores", "heading_residuals_normalized", "size_residuals_normalized")
                  if kind != "static_two" else
                  ("center_one", "box_one", "heading_scores_one", "size_residuals_normalized_one", "center_two",
                   "heading_scores_two", "size_scores_two", "heading_residuals_normalized_two",
                   "size_residuals_normalized_two", "heading_class_label_two", "heading_residuals_label_two")):
            res["out_" + k] = out[k].detach().numpy()
        del full
        for name in PICK[kind]:
            res["grad_" + name] = params[name].grad.numpy().copy()
        # the Dropout multiplier as the reference drew it: kept where output != 0 OR input == 0 (a dropped zero and a
        # kept zero are the same thing); (B,128,N) -> point-major (B*N,128)
        keep = ((drop["y"] != 0) | (drop["x"] == 0)).permute(0, 2, 1).reshape(-1, 128).numpy()
        if force_keep is not None:
            keep = force_keep
        res["keep"] = keep
        res["drop_keep"] = np.packbits(keep, axis=1)
        opt.step()
        sd = model.state_dict()
        for name in PICK[kind]:
            res["new_" + name] = sd[name].numpy().copy()
        for bn in STATS[kind]:
            res["rm_" + bn] = sd[bn + ".running_mean"].numpy().copy()
            res["rv_" + bn] = sd[bn + ".running_var"].numpy().copy()
        return res
    finally:
        torch.Tensor.float = saved_float
        torch.set_default_dtype(saved_default)


def replay_indices(kind, mods, pts_np, mask, np_seed):
    """the object-point indices gather_object_pts drew in that forward (same seed, same mask => same draws)"""
    sm, dm = mods
    np.random.seed(np_seed)
    pts = torch.from_numpy(pts_np).transpose(2, 1)
    if kind != "dynamic":
        return sm.gather_object_pts(pts[:, :3, :], torch.from_numpy(mask), sm.NUM_OBJECT_POINT)[1].numpy()
    return dm.gather_object_pts(pts[:, :4, :], torch.from_numpy(mask), dm.NUM_FRAME * dm.NUM_OBJECT_POINT)[1].numpy()


def main():
    sm, dm, _, _, _ = G.import_reference()
    torch.set_grad_enabled(True)
    torch.set_num_threads(4)
    for kind, torch_seed, np_seed in (("static_one", 101, 202), ("dynamic", 103, 204), ("static_two", 105, 206)):
        inp, labels = case(kind)
        sd = synth.state_dict(kind, seed=43)
        # pass 0 (float32, as the reference runs): its Dropout draw is THE draw of this step.
        # pass 1 (float64, that draw forced): train-mode margins under this draw and these batch statistics ->
        # threshold in the widest gap; the shift moves every margin by the same constant (nothing else depends on
        # dconv5.bias, and a dropped zero equals a kept zero, so the pattern read off pass 0 stays valid).
        p0 = one_step(kind, (sm, dm), sd, inp, labels, torch.float32, torch_seed, np_seed)
        keep = p0["keep"]
        r0 = one_step(kind, (sm, dm), sd, inp, labels, torch.float64, torch_seed, np_seed, force_keep=keep)
        thr, half = synth.widest_gap_centre(r0["logits"][:, :, 1] - r0["logits"][:, :, 0])
        sd = synth.recentre_seg_bias(sd, thr)
        ref = one_step(kind, (sm, dm), sd, inp, labels, torch.float64, torch_seed, np_seed, force_keep=keep)
        f32 = one_step(kind, (sm, dm), sd, inp, labels, torch.float32, torch_seed, np_seed, force_keep=keep)
        nat = one_step(kind, (sm, dm), sd, inp, labels, torch.float32, torch_seed, np_seed)          # no forcing at all
        # where the natural float32 run kept a NON-zero activation, the forced pattern keeps it too (and vice versa)
        assert float(np.abs(nat["logits"] - f32["logits"]).max()) == 0.0, "forcing the draw changed the float32 run"
        margin = ref["logits"][:, :, 1] - ref["logits"][:, :, 0]
        assert np.array_equal(ref["mask"], f32["mask"]), "the float32 run of the reference flipped a mask bit"
        idx = replay_indices(kind, (sm, dm), inp["pts"], ref["mask"], np_seed)
        out = {"margin_shift": np.float64(thr), "min_abs_margin": np.float64(np.abs(margin).min()),
               "torch_seed": torch_seed, "np_seed": np_seed, "lr": LR, "indices": idx.astype(np.int32),
               "drop_keep": ref["drop_keep"], "mask": ref["mask"],
               "in_sum": np.float64(sum(np.asarray(v, np.float64).sum() for v in inp.values()))}
        for k, v in ref.items():
            if k in ("drop_keep", "mask", "keep"):
                continue
            if k.startswith(("grad_", "new_")):                # large tensors: a fixed sample + the tensor's max
                out["ref_" + k] = synth.fixture_sample(v).astype(np.float32)
                out["refmax_" + k] = np.float64(np.abs(v).max())
            else:
                out["ref_" + k] = v.astype(np.float32) if v.dtype == np.float64 and v.ndim else v
        noise = {}
        for k, v in f32.items():
            if k.startswith(("grad_", "loss_", "new_")) or k == "logits":
                a, b = np.asarray(v, np.float64), np.asarray(ref[k], np.float64)
                noise[k] = float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
        out["f32_noise_keys"] = np.array(list(noise))
        out["f32_noise"] = np.array([noise[k] for k in noise])
        for k in ("loss_total_loss",):
            out["f32_" + k] = f32[k]
        np.savez_compressed(os.path.join(HERE, f"train_step_{kind}.npz"), **out)
        worst = max(noise, key=noise.get)
        print(kind, "loss", float(ref["loss_total_loss"]), "min|margin|", float(np.abs(margin).min()), "counts", ref["mask"].sum(1),
              "| reference fp32-vs-fp64 worst:", worst, f"{noise[worst]:.2e}")


def main_big(only=None):
    """the large fixtures: float64 step with the synthetic Dropout pattern (twice: threshold search, then the step) and
    the float32 step with the same pattern (the reference's own rounding noise at this size)"""
    sm, dm, _, _, _ = G.import_reference()
    torch.set_grad_enabled(True)
    torch.set_num_threads(8)
    for kind, torch_seed, np_seed in (("static_one_big", 111, 212), ("dynamic_big", 113, 214), ("static_two_big", 115, 216)):
        if only and kind not in only:
            continue
        base, B, n, seed = BIG[kind]
        inp, labels = case(kind)
        n_points = B * (n if base != "dynamic" else 5 * n)
        keep = big_keep(kind, n_points)
        sd = synth.state_dict(base, seed=seed + 2)
        r0 = one_step(kind, (sm, dm), sd, inp, labels, torch.float64, torch_seed, np_seed, force_keep=keep)
        thr, half = synth.widest_gap_centre(r0["logits"][:, :, 1] - r0["logits"][:, :, 0], window=0.02)
        del r0
        sd = synth.recentre_seg_bias(sd, thr)
        ref = one_step(kind, (sm, dm), sd, inp, labels, torch.float64, torch_seed, np_seed, force_keep=keep)
        f32 = one_step(kind, (sm, dm), sd, inp, labels, torch.float32, torch_seed, np_seed, force_keep=keep)
        margin = ref["logits"][:, :, 1] - ref["logits"][:, :, 0]
        assert np.array_equal(ref["mask"], f32["mask"]), "the float32 run of the reference flipped a mask bit"
        out = {"margin_shift": np.float64(thr), "min_abs_margin": np.float64(np.abs(margin).min()),
               "min_pool_gap": ref["min_pool_gap"], "weights_seed": seed + 2,
               "torch_seed": torch_seed, "np_seed": np_seed, "lr": LR, "mask_bits": np.packbits(ref["mask"], axis=1),
               "in_sum": np.float64(sum(np.asarray(v, np.float64).sum() for v in inp.values())),
               "ref_logits": synth.fixture_sample(ref["logits"]).astype(np.float32),
               "refmax_logits": np.float64(np.abs(ref["logits"]).max())}
        for k, v in ref.items():
            if k in ("drop_keep", "mask", "keep", "logits", "min_pool_gap"):
                continue
            if k.startswith(("grad_", "new_")):
                out["ref_" + k] = synth.fixture_sample(v).astype(np.float32)
                out["refmax_" + k] = np.float64(np.abs(v).max())
            else:
                out["ref_" + k] = v.astype(np.float32) if v.dtype == np.float64 and v.ndim else v
        noise = {}
        for k, v in f32.items():
            if k.startswith(("grad_", "loss_", "new_")) or k == "logits":
                a, b = np.asarray(v, np.float64), np.asarray(ref[k], np.float64)
                noise[k] = float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
        out["f32_noise_keys"] = np.array(list(noise))
        out["f32_noise"] = np.array([noise[k] for k in noise])
        np.savez_compressed(os.path.join(HERE, f"train_step_{kind}.npz"), **out)
        worst = max((k for k in noise if k.startswith("grad_")), key=noise.get)
        print(kind, "loss", float(ref["loss_total_loss"]), "min|margin|", float(np.abs(margin).min()), "min pool gap",
              float(ref["min_pool_gap"]), "counts", ref["mask"].sum(1), "| reference fp32-vs-fp64 worst gradient:", worst,
              f"{noise[worst]:.2e}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "big":
        main_big(sys.argv[2:])
    else:
        main()
