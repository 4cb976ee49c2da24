"""What the reference's drivers import from the model modules besides the models — the Dataset classes and the
loss modules — in the drop-in: checked on CPU against fixtures made by the reference's own classes
(tests/golden/gen_golden.py: the real STATICTRACK / DYNAMICTRACK __getitem__ and the real criteria)."""
import importlib
import os
import pickle

import numpy as np
import pytest
import torch

from _common import golden, synth

datasets = importlib.import_module("3dal_pytorch_amd.datasets")
losses = importlib.import_module("3dal_pytorch_amd.losses")
sm = importlib.import_module("3dal_pytorch_amd.static_model")
dm = importlib.import_module("3dal_pytorch_amd.dynamic_model")


def _write_annos(tmp_path, tracks, drop=()):
    """the annotation pickles gen_golden.py fed to the reference (poses: seed 31)"""
    infos = {}
    for tr in tracks.values():
        for f, tok in enumerate(tr["token"]):
            pose = synth.pose_veh_to_global(31, tok)
            objs = [{"name": "other", "box": np.zeros(9, np.float32)}]
            if tok not in drop:
                objs.append({"name": tr["match"][-1], "box": synth.gt_box_in_vehicle(tr["bbox"][f], pose)})
            path = os.path.join(tmp_path, tok + ".pkl")
            with open(path, "wb") as fh:
                pickle.dump({"veh_to_global": pose, "objects": objs}, fh)
            infos[tok] = {"anno_path": path}
    return infos


def test_statictrack_items_equal_the_reference_datasets(tmp_path):
    g = golden("prep_static")
    tracks = {f"s{t}": synth.track(31, t, n_frames=7 + 3 * t) for t in range(3)}
    ds = datasets.STATICTRACK(tracks, _write_annos(tmp_path, tracks), npoints=4096)
    assert len(ds) == 3
    names = ("mask_label", "center_label", "heading_class_label", "heading_residuals_label", "size_class_label",
             "size_residual_label")
    for i in range(3):
        np.random.seed(100 + i)
        item = ds[i]
        assert len(item) == 11 and item[0] == f"s{i}" and item[4] == str(g[f"token{i}"])
        assert item[1].dtype == torch.float64 and np.array_equal(item[1].numpy(), g[f"init_box{i}"])
        assert item[2].dtype == torch.float32 and np.array_equal(item[2].numpy(), g[f"bbox_gt{i}"])
        assert np.array_equal(item[3].numpy(), g[f"point{i}"])
        for name, v in zip(names, item[5:]):
            assert np.array_equal(np.asarray(v), g[f"{name}{i}"]), name
    # DataLoader collation gives the batch layout static_eval.py:261-267 unpacks
    np.random.seed(5)
    batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=3, shuffle=False)))
    assert batch[1].shape == (3, 1, 7) and batch[3].shape == (3, 4096, 3) and batch[5].shape == (3, 4096)


def test_dynamictrack_items_equal_the_reference_datasets(tmp_path):
    g = golden("prep_dynamic")
    tracks = {"d0": synth.track(32, 10, n_frames=9, empty_every=4), "d1": synth.track(32, 11, n_frames=60)}
    ds = datasets.DYNAMICTRACK(tracks, _write_annos(tmp_path, tracks, drop=("tok_10_2", "tok_10_6")), npoints=1024)
    assert len(ds) == int(g["len"]) == 69 and ds.r == 2 and ds.s == 50
    names = ("mask_label", "center_label", "heading_class_label", "heading_residual_label", "size_class_label",
             "size_residual_label")
    k = 0
    while f"index{k}" in g:
        np.random.seed(200 + k)
        item = ds[int(g[f"index{k}"])]
        assert len(item) == 12
        assert np.array_equal(item[1].numpy(), g[f"init_box{k}"]) and np.array_equal(item[2].numpy(), g[f"bbox{k}"])
        assert np.array_equal(item[3].numpy(), g[f"bbox_gt{k}"])
        assert np.array_equal(item[4].numpy().astype(np.float32), g[f"point{k}"])
        assert np.array_equal(item[4].numpy()[:8], g[f"point64_head{k}"])
        for name, v in zip(names, item[6:]):
            assert np.array_equal(np.asarray(v), g[f"{name}{k}"]), (name, k)
        k += 1
    assert k == 7
    # an item whose own frame lacks the annotation is replaced by a random other item, as in the reference
    np.random.seed(3)
    other = ds[2]
    assert other[5] != "tok_10_2"


def test_loss_modules_equal_the_reference_criteria():
    g = golden("losses")
    for tag, crit, two in (("one", losses.FrustumPointNetLossOneBoxEst(), False),
                           ("two", losses.FrustumPointNetLossTwoBoxEst(), True), ("dyn", losses.DynamicModelLoss(), False)):
        out_np, labels_np = synth.loss_case(36, two_stage=two)
        out_t = {k: torch.from_numpy(v).requires_grad_(v.dtype == np.float32) for k, v in out_np.items()}
        for w_box in (1.0, 0.3):
            got = crit(out_t, *[torch.from_numpy(a) for a in labels_np], w_box=w_box)
            want = {k[len(f"{tag}_w{w_box}_"):]: v for k, v in g.items() if k.startswith(f"{tag}_w{w_box}_")}
            assert set(got) == set(want)
            for k, v in want.items():
                assert abs(float(got[k].detach()) - float(v)) <= 1e-6 * abs(float(v)), (tag, w_box, k)
        grads = torch.autograd.grad(got["total_loss"], [out_t["logits"], out_t["center_two" if two else "center"],
                                                       out_t["size_residuals_normalized_two" if two else
                                                             "size_residuals_normalized"]])
        for name, gr in zip(("dlogits", "dcenter", "dsrn"), grads):
            assert np.allclose(gr.numpy(), g[f"{tag}_{name}"], rtol=1e-5, atol=1e-8), (tag, name)


def test_drivers_imports_resolve_in_the_dropin():
    """the exact import lines of static_eval.py:9-10, static_train.py:13-15, dynamic_train.py:13-15"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); "
            "from static_model import STATICTRACK; "
            "from static_model import StaticModelOneBoxEst, StaticModelTwoBoxEst; "
            "from static_model import FrustumPointNetLossOneBoxEst, FrustumPointNetLossTwoBoxEst; "
            "from dynamic_model import DYNAMICTRACK; from dynamic_model import DynamicModel; "
            "from dynamic_model import DynamicModelLoss; print('ok')" % os.path.join(root, "3dal_pytorch_amd", "dropin"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp")
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr


# ------------------------------------------------------------------ the reference's training step, in float64
@pytest.mark.parametrize("kind", ["static_one", "static_two", "dynamic", "static_one_big"])
def test_composite_in_float64_reproduces_the_reference_training_step(kind):
    """tests/golden/train_step_*.npz: one step of the REAL reference (forward, its criterion, backward; float64 run of
    the imported code with the float32 run's Dropout and NumPy draws, tests/golden/gen_train_step.py). This package's
    train-mode composite + criteria, run in float64 on the CPU with the same draws, must give the same logits, the
    same loss terms and the same gradients to 1e-6 — the semantic pin of the training path. (What float32 arithmetic
    adds on top — isolated ReLU gates / pooled arg-maxima that fall on the other side — is measured on the GPU in
    tests/test_gpu_train_reference.py.)"""
    from _common import golden
    g = golden("train_step_" + kind)
    big = kind.endswith("_big")            # the 16 x 4096 fixture (B*N = 65,536): one of the three is re-run here, ~1 min of CPU
    if big:
        torch.set_num_threads(8)
        kind, B, N, seed = "static_one", 16, 4096, 51
        pts, init, gt = synth.static_crops(B, N, seed=seed)
        labels = synth.loss_case(seed, batch=B, n_pts=N)[1]
        model, crit = sm.StaticModelOneBoxEst(), losses.FrustumPointNetLossOneBoxEst()
    elif kind in ("static_one", "static_two"):
        B, N = 8, 256
        seed = 41 if kind == "static_one" else 44
        pts, init, gt = synth.static_crops(B, N, seed=seed)
        labels = synth.loss_case(seed, batch=B, n_pts=N)[1]
        model, crit = ((sm.StaticModelOneBoxEst(), losses.FrustumPointNetLossOneBoxEst()) if kind == "static_one" else
                       (sm.StaticModelTwoBoxEst(), losses.FrustumPointNetLossTwoBoxEst()))
    else:
        B, N = 4, 320
        pts, box, _, gt = synth.dynamic_items(B, n_per_frame=64, seed=42)
        labels = synth.loss_case(42, batch=B, n_pts=N)[1]
        model, crit = dm.DynamicModel(), losses.DynamicModelLoss()
    sd = synth.recentre_seg_bias(synth.state_dict(kind, seed=int(g["weights_seed"]) if big else 43), float(g["margin_shift"]))
    model.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    model = model.double().train()
    model.train_backend, model.sampler = "torch", "numpy"
    keep_np = ((synth.uniform(seed, "dropout_keep", (B * N, 128)) >= 0.5) if big else
               np.unpackbits(g["drop_keep"], axis=1)).astype(np.float64)
    keep = torch.from_numpy(keep_np).reshape(B, N, 128).permute(0, 2, 1)
    model.ins_seg.dropout.register_forward_hook(lambda m, i, o: i[0] * keep / (1.0 - m.p))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).double()          # noqa: E731
    np.random.seed(int(g["np_seed"]))
    if kind != "dynamic":
        out = model(t(pts).transpose(2, 1), t(init), t(gt))
    else:
        out = model(t(pts).transpose(2, 1), t(box).transpose(2, 1), t(gt))
    if big:
        assert np.array_equal(np.packbits(out["mask"].numpy(), axis=1), g["mask_bits"])
    else:
        assert np.array_equal(out["mask"].numpy(), g["mask"])
    for k in g:
        if k.startswith("ref_out_"):
            v, ref = out[k[8:]].detach().numpy(), g[k]
            if ref.dtype == np.int64:
                assert np.array_equal(v, ref), k
            else:
                assert np.abs(v - ref).max() < 1e-6 * max(np.abs(ref).max(), 1.0), k
    if big:
        assert np.abs(synth.fixture_sample(out["logits"].detach().numpy()) - g["ref_logits"]).max() < 1e-6 * float(g["refmax_logits"])
    else:
        assert np.abs(out["logits"].detach().numpy() - g["ref_logits"]).max() < 1e-6 * np.abs(g["ref_logits"]).max()
    ls = crit(out, *[t(a) if a.dtype == np.float32 else torch.from_numpy(a) for a in labels])
    for k, v in ls.items():
        assert abs(float(v.detach()) - float(g["ref_loss_" + k])) < 1e-6 * max(1.0, abs(float(g["ref_loss_" + k]))), k
    ls["total_loss"].backward()
    params = dict(model.named_parameters())
    n = 0
    for k in g:
        if not k.startswith("ref_grad_"):
            continue
        name, mx = k[9:], float(g["refmax_grad_" + k[9:]])
        got = synth.fixture_sample(params[name].grad.numpy())
        if mx < 1e-9:
            assert np.abs(got).max() < 1e-9, name
            continue
        assert np.abs(got - g[k]).max() < 1e-6 * mx, (name, np.abs(got - g[k]).max() / mx)
        n += 1
    assert n >= 15
