#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (jacky121298/3DAL_PyTorch).

Run only where /root/reference exists (the build container):
    python tests/golden/gen_golden.py

The reference's tools/static_model.py, tools/dynamic_model.py, tools/static_eval.py and
tools/dynamic_eval.py are imported (never copied) through the shim of SURVEY.md 8(c) — stubs
for the un-vendored det3d / fpointnet_train imports, `np.float`, and a no-op Tensor.cuda()
because forward() hard-codes .cuda() — and run on CPU on the deterministic inputs and weights
of 3dal_pytorch_amd/synth.py. Only inputs' checksums and the reference's OUTPUTS are stored;
the inputs are rebuilt from synth.py wherever the fixtures are used.
"""
import importlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
synth = importlib.import_module("3dal_pytorch_amd.synth")

REF = os.environ.get("DAL3_REFERENCE", "/root/reference")


def import_reference():
    np.float = float
    for name in ["det3d", "det3d.core", "det3d.core.bbox", "det3d.core.bbox.box_np_ops",
                 "fpointnet_train", "fpointnet_train.provider_fpointnet"]:
        mod = types.ModuleType(name)
        mod.__path__ = []
        sys.modules[name] = mod
    sys.modules["det3d.core.bbox"].box_np_ops = sys.modules["det3d.core.bbox.box_np_ops"]
    sys.modules["fpointnet_train"].provider_fpointnet = sys.modules["fpointnet_train.provider_fpointnet"]
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    sys.path.insert(0, os.path.join(REF, "tools"))
    import static_model, dynamic_model, static_eval, dynamic_eval, utils  # noqa: E401
    return static_model, dynamic_model, static_eval, dynamic_eval, utils


def _identity_jit(*a, **k):
    """stand-in for numba.njit / numba.jit: `@njit` and `@jit(nopython=True)` both leave the function as it is"""
    if len(a) == 1 and callable(a[0]) and not k:
        return a[0]
    return lambda f: f


def _load_file(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def import_reference_geometry():
    """The reference's det3d/core/bbox/{geometry,box_np_ops}.py and det3d/datasets/waymo/waymo_common.py, loaded
    from their files (the det3d package __init__ needs torchvision/numba). numba is absent: its decorators become
    identities and the njit bodies run as the plain Python they are. The sinks of _create_pd_detection that are
    not captured (Waymo protobuf objects, IoU matching against GT on a CUDA op) are attribute bags / zeros."""
    nb = types.ModuleType("numba")
    nb.njit = nb.jit = _identity_jit
    nb.prange = range
    sys.modules["numba"] = nb
    geo = _load_file("det3d.core.bbox.geometry", os.path.join(REF, "det3d/core/bbox/geometry.py"))
    stub = sys.modules["det3d.core.bbox.box_np_ops"]                 # the object the Dataset modules already hold
    ops = _load_file("det3d.core.bbox.box_np_ops", os.path.join(REF, "det3d/core/bbox/box_np_ops.py"))
    stub.points_in_rbbox = ops.points_in_rbbox
    sys.modules["det3d.core.bbox"].box_np_ops = ops

    class Bag:
        def __init__(self, *a, **k):
            self.objects = []

        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            b = Bag()
            object.__setattr__(self, name, b)
            return b

        def CopyFrom(self, other):
            pass

        def SerializeToString(self):
            return b""
    for name in ["pyquaternion", "nuscenes", "nuscenes.utils", "nuscenes.utils.geometry_utils", "pcdet", "pcdet.ops",
                 "pcdet.ops.iou3d_nms", "pcdet.ops.iou3d_nms.iou3d_nms_utils", "waymo_open_dataset",
                 "waymo_open_dataset.protos", "waymo_open_dataset.label_pb2", "waymo_open_dataset.protos.metrics_pb2"]:
        mod = types.ModuleType(name)
        mod.__path__ = []
        sys.modules[name] = mod
    sys.modules["pyquaternion"].Quaternion = Bag
    sys.modules["nuscenes.utils.geometry_utils"].transform_matrix = None
    sys.modules["pcdet.ops.iou3d_nms.iou3d_nms_utils"].boxes_iou3d_gpu = lambda a, b: torch.zeros((a.shape[0], b.shape[0]))
    lab, met = sys.modules["waymo_open_dataset.label_pb2"], sys.modules["waymo_open_dataset.protos.metrics_pb2"]
    lab.Label = Bag
    Bag.Box = Bag
    met.Object = met.Objects = Bag
    sys.modules["waymo_open_dataset"].label_pb2 = lab
    sys.modules["waymo_open_dataset.protos"].metrics_pb2 = met
    wc = _load_file("det3d.datasets.waymo.waymo_common", os.path.join(REF, "det3d/datasets/waymo/waymo_common.py"))
    return geo, ops, wc


def load(model, sd_np):
    sd = {k: torch.as_tensor(v) for k, v in sd_np.items()}
    missing = model.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return model.eval()


def centred_sd(kind, model, pts, weight_seed, widest_gap=False):
    """Raw synth weights -> measure mean(logit1-logit0) with the reference -> re-centre. widest_gap: centre on the
    middle of the widest gap between sorted margins near the median instead (synth.widest_gap_centre): every point
    then keeps a margin far above fp32 rounding and the fixture's mask is reproducible by any correct fp32
    implementation, free-running."""
    sd = synth.state_dict(kind, weight_seed)
    load(model, sd)
    with torch.no_grad():
        lg = model.ins_seg(pts)
    margin = (lg[:, :, 1] - lg[:, :, 0]).numpy()
    mm = synth.widest_gap_centre(margin)[0] if widest_gap else float(margin.mean())
    sd = synth.recentre_seg_bias(sd, mm)
    load(model, sd)
    return sd, mm


def tonp(d):
    out = {}
    for k, v in d.items():
        out[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    return out


def capture_global(model):
    store = {}

    def hook(_m, _i, o):
        store["g"] = torch.relu(o).max(2)[0].detach().clone()
    h = model.ins_seg.bn5.register_forward_hook(hook)
    return store, h


def main():
    sm, dm, se, de, ut = import_reference()
    torch.set_grad_enabled(False)
    torch.manual_seed(0)
    out_dir = HERE

    # ------------------------------------------------------------------ static, one box est
    for tag, B, N, rng_seed in (("static_one_b4_n1024", 4, 1024, 7), ("static_one_b1_n512", 1, 512, 11)):
        pts_np, init_np, gt_np = synth.static_crops(B, N)
        pts = torch.from_numpy(pts_np).transpose(2, 1)          # logical (B,3,N), physical (B,N,3)
        init, gt = torch.from_numpy(init_np), torch.from_numpy(gt_np)
        model = sm.StaticModelOneBoxEst(3, 3)
        sd, mm = centred_sd("static_one", model, pts, synth.SEED)
        store, h = capture_global(model)
        np.random.seed(rng_seed)
        out = model(pts, init, gt)
        h.remove()
        # replay the gather alone to record indices / object points (same seed => same draw)
        np.random.seed(rng_seed)
        obj, idx = sm.gather_object_pts(pts[:, :3, :], out["mask"], sm.NUM_OBJECT_POINT)
        box_pred = model.box_est(obj.float())
        # refined boxes through the reference's own eval loop
        np.random.seed(rng_seed)
        batch = (None, init[:, None, :].double(), gt.double(), torch.from_numpy(pts_np).double(),
                 None, None, None, None, None, None, None)
        boxes7 = se.test_one_epoch(model, [batch], None)
        o = tonp(out)
        margin = o["logits"][:, :, 1] - o["logits"][:, :, 0]
        np.savez_compressed(
            os.path.join(out_dir, tag + ".npz"), rng_seed=rng_seed, margin_mean=mm,
            in_sum=np.float64(pts_np.astype(np.float64).sum() + init_np.astype(np.float64).sum()),
            global_feat=store["g"].numpy(), indices=idx.numpy().astype(np.int32),
            object_pts=obj.numpy(), box_pred=box_pred.numpy(), boxes7=boxes7,
            min_abs_margin=np.abs(margin).min(), **o)
        print(tag, "counts", o["mask"].sum(1), "min|margin|", np.abs(margin).min(),
              "logit scale", np.abs(o["logits"]).mean())

    # ------------------------------------------------------------------ static, two box est
    tag, B, N, rng_seed = "static_two_b4_n1024", 4, 1024, 13
    pts_np, init_np, gt_np = synth.static_crops(B, N)
    pts = torch.from_numpy(pts_np).transpose(2, 1)
    init, gt = torch.from_numpy(init_np), torch.from_numpy(gt_np)
    model = sm.StaticModelTwoBoxEst(3, 3)
    sd, mm = centred_sd("static_two", model, pts, synth.SEED)
    np.random.seed(rng_seed)
    out = model(pts, init, gt)
    np.random.seed(rng_seed)
    obj, idx = sm.gather_object_pts(pts[:, :3, :], out["mask"], sm.NUM_OBJECT_POINT)
    np.random.seed(rng_seed)
    batch = (None, init[:, None, :].double(), gt.double(), torch.from_numpy(pts_np).double(),
             None, None, None, None, None, None, None)
    boxes7 = se.test_one_epoch(model, [batch], None)
    o = tonp(out)
    np.savez_compressed(
        os.path.join(out_dir, tag + ".npz"), rng_seed=rng_seed, margin_mean=mm,
        in_sum=np.float64(pts_np.astype(np.float64).sum() + init_np.astype(np.float64).sum()),
        indices=idx.numpy().astype(np.int32), boxes7=boxes7, **o)
    print(tag, "counts", o["mask"].sum(1), "labels", o["heading_class_label_two"])

    # ------------------------------------------------------------------ dynamic
    tag, B, rng_seed = "dynamic_b2", 2, 17
    pts_np, box_np, init8_np, gt_np = synth.dynamic_items(B)
    pts = torch.from_numpy(pts_np).transpose(2, 1)              # (B,4,5120)
    box = torch.from_numpy(box_np).transpose(2, 1)              # (B,8,101)
    model = dm.DynamicModel(3, 4)
    sd, mm = centred_sd("dynamic", model, pts, synth.SEED, widest_gap=True)
    np.random.seed(rng_seed)
    out = model(pts, box, torch.from_numpy(gt_np))
    np.random.seed(rng_seed)
    obj, idx = dm.gather_object_pts(pts[:, :4, :], out["mask"], dm.NUM_FRAME * dm.NUM_OBJECT_POINT)
    point_e = model.point_emb(obj.float())
    box_e = model.box_emb(box)
    np.random.seed(rng_seed)
    batch = (None, torch.from_numpy(init8_np).double(), torch.from_numpy(box_np).double(),
             torch.from_numpy(gt_np).double(), torch.from_numpy(pts_np).double(),
             None, None, None, None, None, None, None)
    boxes7 = de.test_one_epoch(model, [batch])
    o = tonp(out)
    np.savez_compressed(
        os.path.join(out_dir, tag + ".npz"), rng_seed=rng_seed, margin_mean=mm,
        in_sum=np.float64(pts_np.astype(np.float64).sum() + box_np.astype(np.float64).sum()),
        indices=idx.numpy().astype(np.int32), point_e=point_e.numpy(), box_e=box_e.numpy(),
        boxes7=boxes7, min_abs_margin=np.abs(o["logits"][:, :, 1] - o["logits"][:, :, 0]).min(), **o)
    print(tag, "counts", o["mask"].sum(1), "min|margin|", np.abs(o["logits"][:, :, 1] - o["logits"][:, :, 0]).min(),
          "logit scale", np.abs(o["logits"]).max())

    # ------------------------------------------------------------------ gather: RNG call order
    N, M = 1024, 512
    counts = [0, 1, 300, 511, 512, 700, N]
    mask = np.zeros((len(counts), N), bool)
    for i, c in enumerate(counts):
        order = np.argsort(synth.uniform(3, f"gmask{i}", (N,)))
        mask[i, order[:c]] = True
    gp = torch.from_numpy(synth.static_crops(len(counts), N, seed=5)[0]).transpose(2, 1)
    np.random.seed(12345)
    obj, idx = sm.gather_object_pts(gp, torch.from_numpy(mask), M)
    after = np.random.randint(0, 1 << 30)                      # pins how much of the stream was consumed
    np.savez_compressed(os.path.join(out_dir, "gather_rng.npz"), counts=np.array(counts), mask=mask,
                        indices=idx.numpy().astype(np.int32), object_pts=obj.numpy(), next_draw=after)

    # ------------------------------------------------------------------ class/angle/size tables
    angles = np.array([0.0, np.pi / 12, np.pi / 12 - 1e-9, np.pi / 12 + 1e-9, 1.0, np.pi, 2 * np.pi - 1e-6,
                       -0.3, -np.pi, 5.5, 7.0, -7.0])
    a2c = np.array([ut.angle2class(a, 12) for a in angles])
    cls = np.arange(12)
    res = np.linspace(-0.26, 0.26, 12)
    c2a = np.array([[ut.class2angle(c, r, 12) for r in res] for c in cls])
    sizes = np.array([[4.5, 1.9, 1.6], [9.0, 2.5, 3.0], [2.2, 0.9, 1.5], [6.5, 2.2, 2.3], [0.5, 0.5, 0.5]])
    s2c = [ut.size2class(s) for s in sizes]
    np.savez_compressed(os.path.join(out_dir, "class_tables.npz"), angles=angles, a2c=a2c, res=res, c2a=c2a,
                        sizes=sizes, s2c_cls=np.array([c for c, _ in s2c]),
                        s2c_res=np.array([r for _, r in s2c]),
                        c2s=np.array([ut.class2size(c, np.array([0.1, -0.2, 0.3])) for c in range(3)]))

    # ------------------------------------------------------------------ crop preparation (N1): the real Datasets
    # ------------------------------------------------------------------ points-in-rotated-box (N1 labels, N2)
    geo, ops, wc = import_reference_geometry()
    g = {}
    for dt, tag in ((np.float32, "f32"), (np.float64, "f64")):
        pts, box9, _, _, _ = synth.sweep(40, "geom", n_points=6000, n_boxes=9)
        boxes = np.concatenate([box9[:, :3], box9[:, [4, 3, 5]], (-box9[:, -1:] - np.pi / 2)], 1).astype(dt)
        boxes[1, 6], boxes[2, 6], boxes[3, 6] = np.pi, -np.pi / 2, 1e-3                # special yaws
        pts = pts.astype(dt)
        pts[:3] = [[np.nan, 0.0, 0.0], [8.0, np.nan, 0.5], [8.0, -4.0, 0.5]]           # NaN counts as inside
        corners = ops.center_to_corner_box3d(boxes[:, :3], boxes[:, 3:6], boxes[:, -1])
        nv, d = geo.surface_equ_3d_jitv2(ops.corner_to_surfaces_3d(corners)[:, :, :3, :])
        g.update({f"boxes_{tag}": boxes, f"corners_{tag}": corners, f"normal_{tag}": nv, f"d_{tag}": d,
                  f"inside_{tag}": ops.points_in_rbbox(pts, boxes), f"nan_rows_{tag}": np.arange(3)})
    # the mixed case of the Datasets: float64 points against a float32 annotation box
    pts64 = synth.sweep(40, "geom", n_points=6000, n_boxes=9)[0].astype(np.float64) + 1e-9
    g["inside_mixed"] = ops.points_in_rbbox(pts64, g["boxes_f32"])
    np.savez_compressed(os.path.join(out_dir, "geom_rbbox.npz"), **g)
    print("geometry fixture: inside counts f32", g["inside_f32"].sum(0), "mixed", g["inside_mixed"].sum(0))

    # ------------------------------------------------------------------ crop extraction (N2): _create_pd_detection
    import pickle
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        res = os.path.join(tmp, "val")
        os.makedirs(res)
        dets, infos = {}, {}
        for f in range(3):
            pts, box9, scores, labels, pose = synth.sweep(41, f"fr{f}", n_points=12000 + 1000 * f, n_boxes=10 + f)
            tok = f"seg0_frame_{f}"
            with open(os.path.join(tmp, f"anno{f}.pkl"), "wb") as fh:
                pickle.dump({"veh_to_global": pose, "objects": [], "scene_name": "seg0", "frame_name": f"seg0_{1000 + f}",
                             "frame_id": f}, fh)
            with open(os.path.join(tmp, f"lidar{f}.pkl"), "wb") as fh:
                pickle.dump({"lidars": {"points_xyz": pts}}, fh)
            infos[tok] = {"anno_path": os.path.join(tmp, f"anno{f}.pkl"), "path": os.path.join(tmp, f"lidar{f}.pkl"),
                          "timestamp": 1000.0 + f}
            dets[tok] = {"box3d_lidar": torch.from_numpy(box9.copy()), "scores": torch.from_numpy(scores),
                         "label_preds": torch.from_numpy(labels), "tracking_ids": list(range(box9.shape[0]))}
        wc._create_pd_detection(dets, infos, res, tracking=True)
        with open(os.path.join(res, "trackData.pkl"), "rb") as fh:
            td = pickle.load(fh)
        with open(os.path.join(res, "det_annos.pkl"), "rb") as fh:
            da = pickle.load(fh)
        c = {}
        for f, tok in enumerate(dets):
            c[f"boxes_lidar{f}"] = da[f]["boxes_lidar"]
            c[f"bbox{f}"] = np.stack(td[tok]["bbox"])
            c[f"count{f}"] = np.array([p.shape[0] for p in td[tok]["point"]])
            c[f"point{f}"] = np.concatenate(td[tok]["point"])
            c[f"type{f}"] = np.array(td[tok]["type"])
        np.savez_compressed(os.path.join(out_dir, "crops_extract.npz"), **c)
        print("crop-extraction fixture: points per detection", [c[f"count{f}"].tolist() for f in range(3)])
    with tempfile.TemporaryDirectory() as tmp:
        def make_infos(tracks):
            infos = {}
            for tid, tr in tracks.items():
                for f, tok in enumerate(tr["token"]):
                    path = os.path.join(tmp, tok + ".pkl")
                    pose = synth.pose_veh_to_global(31, tok)
                    objs = [{"name": "other", "box": np.zeros(9, np.float32)}]
                    if not (tok.startswith("tok_10_") and tok.endswith(("_2", "_6"))):   # d0 lacks its GT in 2 frames
                        objs.append({"name": tr["match"][-1], "box": synth.gt_box_in_vehicle(tr["bbox"][f], pose)})
                    with open(path, "wb") as fh:
                        pickle.dump({"veh_to_global": pose, "objects": objs}, fh)
                    infos[tok] = {"anno_path": path}
            return infos
        # static: three tracks, 4096-point resample
        tracks = {f"s{t}": synth.track(31, t, n_frames=7 + 3 * t) for t in range(3)}
        ds = sm.STATICTRACK(tracks, make_infos(tracks), npoints=4096)
        out = {}
        for i in range(len(ds)):
            np.random.seed(100 + i)
            item = ds[i]
            out[f"init_box{i}"] = item[1].numpy()
            out[f"point{i}"] = item[3].numpy()
            out[f"token{i}"] = np.array(item[4])
            out[f"bbox_gt{i}"] = item[2].numpy()
            for name, v in zip(("mask_label", "center_label", "heading_class_label", "heading_residuals_label",
                                "size_class_label", "size_residual_label"), item[5:11]):
                out[f"{name}{i}"] = np.asarray(v)
        np.savez_compressed(os.path.join(out_dir, "prep_static.npz"), **out)
        # dynamic: two tracks (one with empty frames), items at the borders and in the middle
        dtracks = {"d0": synth.track(32, 10, n_frames=9, empty_every=4), "d1": synth.track(32, 11, n_frames=60)}
        dds = dm.DYNAMICTRACK(dtracks, make_infos(dtracks), npoints=1024)
        out = {"len": len(dds)}
        for k, idx in enumerate([0, 1, 4, 8, 9, 9 + 30, 9 + 59]):
            np.random.seed(200 + k)
            item = dds[idx]
            out[f"index{k}"] = idx
            out[f"init_box{k}"] = item[1].numpy()
            out[f"bbox{k}"] = item[2].numpy()
            out[f"point{k}"] = item[4].numpy().astype(np.float32)          # what the driver's .float() keeps
            out[f"point64_head{k}"] = item[4].numpy()[:8]
            out[f"bbox_gt{k}"] = item[3].numpy()
            for name, v in zip(("mask_label", "center_label", "heading_class_label", "heading_residual_label",
                                "size_class_label", "size_residual_label"), item[6:12]):
                out[f"{name}{k}"] = np.asarray(v)
        np.savez_compressed(os.path.join(out_dir, "prep_dynamic.npz"), **out)
        print("prep fixtures: static", len(ds), "dynamic items", len(dds))

        # -------------------------------------------------------------- write-back (N3): the real postprocessing()
        # IoU metrics need the un-vendored provider_fpointnet: stubbed, not captured; det_annos (the product) is.
        import logging
        se.compute_box3d_iou = de.compute_box3d_iou = lambda *a: (np.zeros(1), np.zeros(1))
        tracks_l, poses_s, dets_s, has_gt = synth.scene(33, n_frames=24, n_tracks=9)
        infos, token2idx, det_annos = {}, {}, []
        for f, (tok, pose) in enumerate(poses_s.items()):
            objs = [{"name": f"gt_{k}", "box": np.concatenate([tr["bbox"][tr["token"].index(tok)][:6], [0.0, 0.0],
                                                                  tr["bbox"][tr["token"].index(tok)][6:]]).astype(np.float32)}
                    for k, tr in enumerate(tracks_l) if tok in tr["token"] and has_gt[(k, tok)]]
            path = os.path.join(tmp, tok + ".pkl")
            with open(path, "wb") as fh:
                pickle.dump({"veh_to_global": pose, "objects": objs}, fh)
            infos[tok] = {"anno_path": path}
            token2idx[tok] = f
            det_annos.append({"boxes_lidar": dets_s[tok].copy(), "frame_id": tok})
        track_dict = {f"trk{k}": tr for k, tr in enumerate(tracks_l)}
        n_static = len(tracks_l)
        final_static = np.stack([np.concatenate([synth.normal(34, f"fs{k}", (3,), 0.0, 2.0),
                                                 np.array(synth.arch.MEAN_SIZE[k % 3]) + 0.2,
                                                 synth.uniform(34, f"fsy{k}", (1,), -3.0, 3.0)]) for k in range(n_static)])
        out_s = se.postprocessing(track_dict, infos, token2idx, final_static.copy(),
                                  [dict(d, boxes_lidar=d["boxes_lidar"].copy()) for d in det_annos],
                                  os.path.join(tmp, "res_s.pkl"), logging.getLogger("gen"))[3]
        n_dyn = sum(len(tr["token"]) for tr in tracks_l)
        final_dyn = synth.normal(35, "fd", (n_dyn, 7), 0.0, 5.0)
        out_d = de.postprocessing(track_dict, infos, token2idx, final_dyn.copy(),
                                  [dict(d, boxes_lidar=d["boxes_lidar"].copy()) for d in det_annos],
                                  os.path.join(tmp, "res_d.pkl"), logging.getLogger("gen"))[3]
        np.savez_compressed(os.path.join(out_dir, "post_writeback.npz"), final_static=final_static, final_dyn=final_dyn,
                            **{f"static_{d['frame_id']}": d["boxes_lidar"] for d in out_s},
                            **{f"dynamic_{d['frame_id']}": d["boxes_lidar"] for d in out_d})
        changed = sum(int((a["boxes_lidar"] != b["boxes_lidar"]).any(1).sum()) for a, b in zip(out_s, det_annos))
        print("write-back fixtures: frames", len(det_annos), "static rows changed", changed)

    # ------------------------------------------------------------------ loss modules (N4): the reference's criteria
    torch.set_grad_enabled(True)
    lz = {}
    for tag, crit, two in (("one", sm.FrustumPointNetLossOneBoxEst(), False), ("two", sm.FrustumPointNetLossTwoBoxEst(), True),
                           ("dyn", dm.DynamicModelLoss(), False)):
        out_np, labels_np = synth.loss_case(36, two_stage=two)
        out_t = {k: torch.from_numpy(v).requires_grad_(v.dtype == np.float32) for k, v in out_np.items()}
        for w_box in (1.0, 0.3):
            losses = crit(out_t, *[torch.from_numpy(a) for a in labels_np], w_box=w_box)
            for k, v in losses.items():
                lz[f"{tag}_w{w_box}_{k}"] = v.detach().numpy()
        grads = torch.autograd.grad(losses["total_loss"], [out_t["logits"], out_t["center_two" if two else "center"],
                                                          out_t["size_residuals_normalized_two" if two else
                                                                "size_residuals_normalized"]])
        lz[f"{tag}_dlogits"], lz[f"{tag}_dcenter"], lz[f"{tag}_dsrn"] = (g.numpy() for g in grads)
    np.savez_compressed(os.path.join(out_dir, "losses.npz"), **lz)
    torch.set_grad_enabled(False)
    print("loss fixtures:", {k: float(v) for k, v in lz.items() if k.endswith("total_loss")})

    # ------------------------------------------------------------------ state_dict key pin
    keys = {}
    for kind, ctor in (("static_one", lambda: sm.StaticModelOneBoxEst(3, 3)),
                       ("static_two", lambda: sm.StaticModelTwoBoxEst(3, 3)),
                       ("dynamic", lambda: dm.DynamicModel(3, 4))):
        m = ctor()
        keys[kind + "_keys"] = np.array(list(m.state_dict().keys()))
        keys[kind + "_shapes"] = np.array([str(tuple(v.shape)) for v in m.state_dict().values()])
    np.savez_compressed(os.path.join(out_dir, "state_dict_keys.npz"), **keys)
    print("done")


if __name__ == "__main__":
    main()
