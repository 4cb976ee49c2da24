"""Deterministic synthetic weights and object crops (no dataset, no checkpoint, no torch RNG).

Every value is a pure function of (seed, tensor name, element index) through splitmix64, so the
golden-vector generator (tests/golden/gen_golden.py, run where /root/reference exists), the
parity tests and bench.py (run on the GPU box, where it does not) rebuild identical tensors.

Shapes follow SURVEY.md 8(d): crops are box-frame point scatters (static_model.py:529-572
hands forward() box-centred, box-aligned xyz), dynamic items carry the 0.1*(j-2) time channel
(dynamic_model.py:432-437) and a 101-box window with 0.1*(j-50) in channel 7 (:441-447).
"""
import zlib

import numpy as np

from . import arch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
SEED = 10922081            # the reference's own fixSeed value (static_eval.py:303)


def _splitmix(z):
    with np.errstate(over="ignore"):
        z = (z + np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _stream(seed, name, n, lane=0):
    """n uint64 words for (seed, name, lane)."""
    key = (np.uint64(seed) << np.uint64(32)) ^ np.uint64(zlib.crc32(name.encode())) \
        ^ (np.uint64(lane) << np.uint64(56))
    base = _splitmix(np.array([key], dtype=np.uint64))[0]
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + base
    return _splitmix(idx)


def uniform(seed, name, shape, lo=0.0, hi=1.0, lane=0):
    n = int(np.prod(shape)) if len(shape) else 1
    u = (_stream(seed, name, n, lane) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    return (lo + (hi - lo) * u).reshape(shape)


def normal(seed, name, shape, mean=0.0, std=1.0):
    u1 = uniform(seed, name, shape, lane=1)
    u2 = uniform(seed, name, shape, lane=2)
    z = np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)
    return mean + std * z


# --------------------------------------------------------------------------- weights
def state_dict(model, seed=SEED):
    """Random-init state_dict (numpy) with the reference key set of
    'static_one' | 'static_two' | 'dynamic'. Conv/Linear: He-uniform; BN running stats are
    randomised so that folding is exercised (SURVEY.md 8(c) golden-vector recipe)."""
    sd = {}
    for key, shape in arch.model_param_specs(model):
        leaf = key.rsplit(".", 1)[1]
        mod = key.rsplit(".", 2)[-2]
        is_bn = "bn" in mod
        if leaf == "num_batches_tracked":
            sd[key] = np.array(1000, dtype=np.int64)
        elif is_bn and leaf == "weight":
            sd[key] = uniform(seed, key, shape, 0.5, 1.5).astype(np.float32)
        elif is_bn and leaf == "bias":
            sd[key] = normal(seed, key, shape, 0.0, 0.2).astype(np.float32)
        elif leaf == "running_mean":
            sd[key] = normal(seed, key, shape, 0.0, 0.5).astype(np.float32)
        elif leaf == "running_var":
            sd[key] = uniform(seed, key, shape, 0.5, 2.0).astype(np.float32)
        elif leaf == "weight":
            fan_in = shape[1]
            a = np.sqrt(6.0 / fan_in)
            sd[key] = uniform(seed, key, shape, -a, a).astype(np.float32)
        else:  # conv / linear bias
            sd[key] = uniform(seed, key, shape, -0.1, 0.1).astype(np.float32)
    return sd


def recentre_seg_bias(sd, margin_mean):
    """Shift ins_seg.dconv5.bias[1] so the mean of (logit1 - logit0) is zero and roughly half
    the points are segmented as object (with raw random init every point lands in one class)."""
    sd = dict(sd)
    b = sd["ins_seg.dconv5.bias"].copy()
    b[1] -= np.float32(margin_mean)
    sd["ins_seg.dconv5.bias"] = b
    return sd


def widest_gap_centre(margin, window=0.1):
    """(threshold, half_gap): a decision threshold for the segmentation margin logit1 - logit0 that sits in the
    middle of the WIDEST gap between consecutive sorted margins among the central `window` quantiles. Re-centring
    the bias there (recentre_seg_bias(sd, threshold)) still segments about half the points, and leaves every point
    at least half_gap away from the tie — so that a last-bit difference between two fp32 implementations cannot
    flip a mask bit and the free-running comparisons of the parity tests are unconditional."""
    m = np.sort(np.asarray(margin, np.float64).ravel())
    n = len(m)
    lo, hi = int(n * (0.5 - window / 2)), max(int(n * (0.5 + window / 2)), int(n * (0.5 - window / 2)) + 2)
    hi = min(hi, n)
    gaps = np.diff(m[lo:hi])
    i = int(np.argmax(gaps))
    return float(0.5 * (m[lo + i] + m[lo + i + 1])), float(0.5 * gaps[i])


def fixture_sample(a, limit=16384, keep=8192):
    """what a fixture stores of a large tensor: all of it up to `limit` elements, else `keep` elements at fixed
    pseudo-random flat positions (a function of the size only) — the generator and the test apply the same map"""
    a = np.asarray(a)
    if a.size <= limit:
        return a.reshape(-1)
    idx = np.sort(np.random.default_rng(a.size).choice(a.size, keep, replace=False))
    return a.reshape(-1)[idx]


# --------------------------------------------------------------------------- inputs
def static_crops(batch, n_pts, seed=SEED, first=0):
    """pts (batch, n_pts, 3) fp32 point-major, init_box (batch, 7), bbox_gt (batch, 7).
    Items are keyed on their global index `first + i`, so a shard of a larger job generates
    exactly the rows the whole job would."""
    pts = np.empty((batch, n_pts, 3), np.float32)
    init = np.empty((batch, 7), np.float32)
    gt = np.empty((batch, 7), np.float32)
    for i in range(batch):
        g = first + i
        pts[i] = _crop_xyz(seed, f"s{g}", n_pts)
        init[i], gt[i] = _boxes(seed, f"s{g}", g)
    return pts, init, gt


def dynamic_items(batch, n_per_frame=1024, seed=SEED, first=0, n_box=101):
    """pts (batch, 5*n_per_frame, 4), box (batch, n_box, 8), init_box (batch, 8), bbox_gt (batch, 7)."""
    n = arch.NUM_FRAME * n_per_frame
    pts = np.empty((batch, n, 4), np.float32)
    box = np.empty((batch, n_box, 8), np.float32)
    init = np.empty((batch, 8), np.float32)
    gt = np.empty((batch, 7), np.float32)
    half = n_box // 2
    for i in range(batch):
        g = first + i
        for f in range(arch.NUM_FRAME):
            sl = slice(f * n_per_frame, (f + 1) * n_per_frame)
            pts[i, sl, :3] = _crop_xyz(seed, f"d{g}f{f}", n_per_frame)
            pts[i, sl, 3] = np.float32(0.1 * (f - 2))
        ib, gtb = _boxes(seed, f"d{g}", g)
        step = normal(seed, f"d{g}walk", (n_box, 3), 0.0, 0.05)
        ctr = np.cumsum(step, axis=0)
        ctr -= ctr[half]
        yaw = np.cumsum(normal(seed, f"d{g}yaw", (n_box,), 0.0, 0.01))
        yaw -= yaw[half]
        box[i, :, 0:3] = ctr
        box[i, :, 3:6] = ib[3:6]
        box[i, :, 6] = yaw
        box[i, :, 7] = 0.1 * (np.arange(n_box) - half)
        init[i, :7] = ib                        # (x,y,z,l,w,h,yaw,t): yaw at [-2] (dynamic_eval.py:239)
        init[i, 7] = 0.0                        # t of the centre box = 0.1*(50-50)
        gt[i] = gtb
    return pts, box, init, gt


def _crop_xyz(seed, tag, n):
    u = uniform(seed, tag + "sel", (n,))
    obj = uniform(seed, tag + "obj", (n, 3), -1.0, 1.0) * np.array([2.4, 0.9, 0.75])
    clut = uniform(seed, tag + "clu", (n, 3), -1.0, 1.0) * np.array([7.2, 2.7, 2.25])
    return np.where((u < 0.4)[:, None], obj, clut).astype(np.float32)


def _boxes(seed, tag, g):
    c = normal(seed, tag + "c", (3,), 0.0, 0.3)
    size = np.array(arch.MEAN_SIZE[g % 3]) + normal(seed, tag + "s", (3,), 0.0, 0.1)
    yaw = uniform(seed, tag + "y", (1,), -np.pi, np.pi)
    init = np.concatenate([c, size, yaw]).astype(np.float32)
    gt = init + np.concatenate([normal(seed, tag + "g", (6,), 0.0, 0.1),
                                normal(seed, tag + "gy", (1,), 0.0, 0.2)]).astype(np.float32)
    return init, gt.astype(np.float32)


# --------------------------------------------------------------------------- tracks (crop preparation, N1)
def pose_veh_to_global(seed, tag):
    """flat 16 (row-major 4x4) rigid transform like Waymo's veh_to_global: yaw + a translation of kilometres
    (which is why the preparation is done in float64)"""
    yaw = uniform(seed, tag + "yaw", (1,), -np.pi, np.pi)[0]
    t = uniform(seed, tag + "t", (3,), -2.0e4, 2.0e4) * np.array([1.0, 1.0, 0.01])
    c, s = np.cos(yaw), np.sin(yaw)
    m = np.array([[c, -s, 0.0, t[0]], [s, c, 0.0, t[1]], [0.0, 0.0, 1.0, t[2]], [0.0, 0.0, 0.0, 1.0]])
    return m.reshape(16)


def track(seed, tid, n_frames, max_pts=400, empty_every=0):
    """One synthetic track in the reference's schema (SURVEY.md 8(g) `track.pkl`): per-frame lists of global-frame
    boxes (7,), global-frame point arrays (k,3) float64 (some empty when empty_every > 0), scores, tokens."""
    tag = f"trk{tid}"
    base = uniform(seed, tag + "c", (3,), -2.0e4, 2.0e4) * np.array([1.0, 1.0, 0.01])
    vel = normal(seed, tag + "v", (3,), 0.0, 0.5) * np.array([1.0, 1.0, 0.0])
    size = np.array(arch.MEAN_SIZE[tid % 3]) + normal(seed, tag + "s", (3,), 0.0, 0.1)
    yaw0 = uniform(seed, tag + "y", (1,), -np.pi, np.pi)[0]
    out = {"bbox": [], "point": [], "score": [], "token": [], "match": []}
    for f in range(n_frames):
        c = base + vel * f + normal(seed, f"{tag}f{f}n", (3,), 0.0, 0.05)
        yaw = yaw0 + 0.01 * f
        out["bbox"].append(np.concatenate([c, size, [yaw]]))
        k = 0 if (empty_every and f % empty_every == empty_every - 1) else \
            int(20 + uniform(seed, f"{tag}f{f}k", (1,))[0] * (max_pts - 20))
        local = uniform(seed, f"{tag}f{f}p", (k, 3), -1.0, 1.0) * (size / 2 * 1.3)
        cy, sy = np.cos(yaw), np.sin(yaw)
        rot = np.array([[cy, -sy, 0.0], [sy, cy, 0.0], [0.0, 0.0, 1.0]])
        out["point"].append(local @ rot.T + c)
        out["score"].append(float(uniform(seed, f"{tag}f{f}s", (1,))[0]))
        out["token"].append(f"tok_{tid}_{f}")
        out["match"].append(f"gt_{tid}")
    return out


def scene(seed, n_frames, n_tracks, n_clutter=5):
    """A segment for the write-back step (N3): tracks that SHARE frames. Returns
    tracks (list of track dicts whose tokens are 'fr_<f>' and whose boxes lie near the ego path),
    poses {token: flat-16 veh_to_global}, dets {token: (n,7) float32 vehicle-frame detections = every track's box
    of that frame plus clutter, in a shuffled order}, has_gt {(track, token): bool}."""
    poses, dets = {}, {}
    ego0 = uniform(seed, "ego0", (3,), -2.0e4, 2.0e4) * np.array([1.0, 1.0, 0.01])
    for f in range(n_frames):
        yaw = 0.3 + 0.02 * f
        t = ego0 + np.array([1.5 * f, 0.4 * f, 0.0])
        c, s = np.cos(yaw), np.sin(yaw)
        poses[f"fr_{f}"] = np.array([[c, -s, 0.0, t[0]], [s, c, 0.0, t[1]], [0.0, 0.0, 1.0, t[2]],
                                     [0.0, 0.0, 0.0, 1.0]]).reshape(16)
    tracks, has_gt = [], {}
    for k in range(n_tracks):
        f0 = int(uniform(seed, f"sc{k}f0", (1,))[0] * (n_frames // 2))
        f1 = min(n_frames, f0 + 7 + int(uniform(seed, f"sc{k}f1", (1,))[0] * n_frames))
        rel = uniform(seed, f"sc{k}rel", (3,), -1.0, 1.0) * np.array([40.0, 15.0, 0.5])
        size = np.array(arch.MEAN_SIZE[k % 3]) + normal(seed, f"sc{k}s", (3,), 0.0, 0.1)
        yaw0 = uniform(seed, f"sc{k}y", (1,), -np.pi, np.pi)[0]
        tr = {"bbox": [], "point": [], "score": [], "token": [], "match": [], "type": []}
        for f in range(f0, f1):
            ctr = ego0 + rel + np.array([1.5 * f0, 0.4 * f0, 0.0]) + normal(seed, f"sc{k}n{f}", (3,), 0.0, 0.03)
            tr["bbox"].append(np.concatenate([ctr, size, [yaw0 + 0.005 * f]]))
            tr["point"].append(np.zeros((0, 3)))
            tr["score"].append(float(uniform(seed, f"sc{k}sc{f}", (1,))[0]))
            tr["token"].append(f"fr_{f}")
            tr["match"].append(f"gt_{k}")
            tr["type"].append(1 if k % 2 == 0 else 4)
            has_gt[(k, f"fr_{f}")] = uniform(seed, f"sc{k}g{f}", (1,))[0] > 0.15
        tracks.append(tr)
    for f in range(n_frames):
        tok = f"fr_{f}"
        pose = np.linalg.inv(poses[tok].reshape(4, 4))
        rows = []
        for tr in tracks:
            if tok in tr["token"]:
                b = tr["bbox"][tr["token"].index(tok)]
                ctr = pose[:3, :3] @ b[:3] + pose[:3, 3]
                rows.append(np.concatenate([ctr, b[3:6], [b[6] + np.arctan2(pose[1, 0], pose[0, 0])]]))
        for c in range(n_clutter):
            rows.append(np.concatenate([uniform(seed, f"cl{f}_{c}", (3,), -60.0, 60.0),
                                        uniform(seed, f"cls{f}_{c}", (3,), 1.0, 5.0),
                                        uniform(seed, f"cly{f}_{c}", (1,), -np.pi, np.pi)]))
        order = np.argsort(uniform(seed, f"ord{f}", (len(rows),)))
        dets[tok] = np.stack(rows)[order].astype(np.float32)
    return tracks, poses, dets, has_gt


def sweep(seed, tag, n_points=20000, n_boxes=12):
    """One synthetic lidar sweep with detections, in the formats either side of the crop extraction (SURVEY.md 8(f)
    N2 / 8(g)): points_xyz (P,3) float32 vehicle frame; box3d_lidar (K,9) float32 in the DETECTOR's convention
    [x,y,z,w,l,h,vx,vy,r2] (waymo_common.py:105-111 turns it into [x,y,z,l,w,h,-r2-pi/2]); scores (K,), labels (K,)
    in {0,1,2}; flat-16 veh_to_global. About an eighth of the points fall inside some box; box 0 is axis-aligned
    with a few points exactly on its faces (the `>= 0` edge of the test); the last box is empty."""
    t = f"swp{tag}"
    cls = (uniform(seed, t + "cls", (n_boxes,)) * 3).astype(np.int64)
    centre = uniform(seed, t + "c", (n_boxes, 3), -60.0, 60.0) * np.array([1.0, 1.0, 0.02])
    size = np.array(arch.MEAN_SIZE)[cls % 3] + normal(seed, t + "s", (n_boxes, 3), 0.0, 0.15)
    yaw = uniform(seed, t + "y", (n_boxes,), -np.pi, np.pi)
    centre[0], size[0], yaw[0] = [8.0, -4.0, 0.5], [4.0, 2.0, 1.5], 0.0
    centre[-1] = [500.0, 500.0, 0.0]                                              # far away: no points
    n_obj = n_points // 3
    owner = (uniform(seed, t + "o", (n_obj,)) * (n_boxes - 1)).astype(np.int64)
    local = uniform(seed, t + "p", (n_obj, 3), -0.65, 0.65) * size[owner]             # 1.3x the box: in and out
    c, s = np.cos(yaw[owner]), np.sin(yaw[owner])
    obj = np.stack([c * local[:, 0] - s * local[:, 1], s * local[:, 0] + c * local[:, 1], local[:, 2]], 1) + centre[owner]
    clutter = uniform(seed, t + "g", (n_points - n_obj - 6, 3), -75.0, 75.0) * np.array([1.0, 1.0, 0.03])
    edge = np.array([[10.0, -4.0, 0.5], [6.0, -4.0, 0.5], [8.0, -3.0, 0.5], [8.0, -5.0, 0.5], [8.0, -4.0, 1.25],
                     [8.0, -4.0, -0.25]])                                          # on the six faces of box 0
    pts = np.concatenate([obj, clutter, edge]).astype(np.float32)
    pts = pts[np.argsort(uniform(seed, t + "sh", (pts.shape[0],)))]
    r2 = -yaw - np.pi / 2
    box = np.concatenate([centre, size[:, [1, 0, 2]], normal(seed, t + "v", (n_boxes, 2), 0.0, 1.0), r2[:, None]], 1)
    scores = uniform(seed, t + "sc", (n_boxes,), 0.1, 1.0).astype(np.float32)
    return pts, box.astype(np.float32), scores, cls, pose_veh_to_global(seed, t)


def gt_box_in_vehicle(box_global, veh_to_global):
    """A ground-truth annotation for a synthetic track frame, in the annotation files' format (SURVEY.md 8(g):
    float32 (9,) [cx,cy,cz,l,w,h,vx,vy,heading], VEHICLE frame): the track's global box moved into the frame's
    vehicle frame and shrunk by 10 % so that some of the track's points fall outside it."""
    m = np.reshape(veh_to_global, [4, 4])
    c = m[:3, :3].T @ (np.asarray(box_global[:3]) - m[:3, 3])
    yaw = box_global[6] - np.arctan2(m[1, 0], m[0, 0])
    return np.concatenate([c, 0.9 * np.asarray(box_global[3:6]), [0.0, 0.0], [yaw]]).astype(np.float32)


def segment_files(root, seed, n_frames=12, n_tracks=5, max_pts=300, scene_name="synth0001"):
    """One synthetic segment written in the reference's on-disk formats under `root`:
    annos/<token>.pkl, infos.pkl (list of info dicts), det_annos.pkl (list, deliberately NOT in frame order),
    trackStatic.pkl and trackDynamic.pkl (the same tracks: {track id: per-frame lists}). Tracks come from scene()
    with points added around each box (every 5th frame of odd tracks empty); the frames where scene() says the
    matched annotation is missing have no such object in their annos file. Returns the paths and the pieces."""
    import os
    import pickle
    tracks, poses, dets, has_gt = scene(seed, n_frames, n_tracks)
    for k, tr in enumerate(tracks):
        for j, b in enumerate(tr["bbox"]):
            n = 0 if (k % 2 == 1 and j % 5 == 4) else int(40 + uniform(seed, f"sf{k}_{j}k", (1,))[0] * (max_pts - 40))
            local = uniform(seed, f"sf{k}_{j}p", (n, 3), -1.0, 1.0) * (b[3:6] / 2 * 1.2)
            cy, sy = np.cos(b[6]), np.sin(b[6])
            rot = np.array([[cy, -sy, 0.0], [sy, cy, 0.0], [0.0, 0.0, 1.0]])
            tr["point"][j] = local @ rot.T + b[:3]
    os.makedirs(os.path.join(root, "annos"), exist_ok=True)
    infos = []
    for f in range(n_frames):
        tok = f"fr_{f}"
        objs = [{"id": k, "name": tr["match"][0], "label": tr["type"][0], "num_points": 10,
                 "box": gt_box_in_vehicle(tr["bbox"][tr["token"].index(tok)], poses[tok])}
                for k, tr in enumerate(tracks) if tok in tr["token"] and has_gt[(k, tok)]]
        path = os.path.join(root, "annos", tok + ".pkl")
        with open(path, "wb") as fh:
            pickle.dump({"scene_name": scene_name, "frame_name": f"{scene_name}_{f}", "frame_id": f,
                         "veh_to_global": poses[tok], "objects": objs}, fh)
        infos.append({"path": os.path.join(root, "lidar", tok + ".pkl"), "anno_path": path, "token": tok,
                      "timestamp": 0.1 * f, "sweeps": []})
    order = np.argsort(uniform(seed, "sf_order", (n_frames,)))
    det_annos = [{"name": np.array(["VEHICLE"] * len(dets[f"fr_{f}"])), "score": np.full(len(dets[f"fr_{f}"]), 0.5),
                  "boxes_lidar": dets[f"fr_{f}"].copy(),
                  "frame_id": f"segment-{scene_name}_with_camera_labels_{f:03d}", "metadata": {"token": f"fr_{f}"}}
                 for f in order]
    track = {f"id{k:02d}": tr for k, tr in enumerate(tracks)}
    paths = {"infos": os.path.join(root, "infos.pkl"), "det_annos": os.path.join(root, "det_annos.pkl"),
             "static": os.path.join(root, "trackStatic.pkl"), "dynamic": os.path.join(root, "trackDynamic.pkl")}
    for key, obj in (("infos", infos), ("det_annos", det_annos), ("static", track), ("dynamic", track)):
        with open(paths[key], "wb") as fh:
            pickle.dump(obj, fh)
    return paths, tracks, poses, dets, has_gt


def loss_case(seed, two_stage=False, batch=6, n_pts=64):
    """Synthetic model outputs + labels for the loss modules (tools/static_model.py:348-517): returns
    (output dict of float32 arrays, labels tuple in the criterion's argument order)."""
    t = f"loss{int(two_stage)}"
    f32 = lambda name, shape, std=1.0: normal(seed, t + name, shape, 0.0, std).astype(np.float32)   # noqa: E731
    out = {"logits": f32("lg", (batch, n_pts, 2))}
    for tag in (("_one", "_two") if two_stage else ("",)):
        out["center" + tag] = f32("c" + tag, (batch, 3), 2.0)
        out["heading_scores" + tag] = f32("hs" + tag, (batch, 12))
        out["heading_residuals_normalized" + tag] = f32("hrn" + tag, (batch, 12), 0.7)
        out["heading_residuals" + tag] = out["heading_residuals_normalized" + tag] * np.float32(np.pi / 12)
        out["size_scores" + tag] = f32("ss" + tag, (batch, 3))
        out["size_residuals_normalized" + tag] = f32("srn" + tag, (batch, 3, 3), 0.5)
        out["size_residuals" + tag] = out["size_residuals_normalized" + tag] * np.array(arch.MEAN_SIZE, np.float32)[None]
    if two_stage:
        out["heading_class_label_two"] = (uniform(seed, t + "hcl2", (batch,)) * 12).astype(np.int64)
        out["heading_residuals_label_two"] = f32("hrl2", (batch,), 0.1)
    labels = ((uniform(seed, t + "ml", (batch, n_pts)) > 0.6).astype(np.float32), f32("cl", (batch, 3), 2.0),
              (uniform(seed, t + "hcl", (batch,)) * 12).astype(np.int64), f32("hrl", (batch,), 0.1),
              (uniform(seed, t + "scl", (batch,)) * 3).astype(np.int64), f32("srl", (batch, 3), 0.3))
    return out, labels
