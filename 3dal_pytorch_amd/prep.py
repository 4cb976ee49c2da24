"""Crop preparation on the device: the per-item work of the reference's Dataset classes
`STATICTRACK.__getitem__` (tools/static_model.py:529-572) and `DYNAMICTRACK.__getitem__`
(tools/dynamic_model.py:419-509) — minus pickle I/O and minus the training labels — batched into one
lib3dal_hip.so kernel per batch (dal3_static_crop_prep / dal3_dynamic_item_prep). With the matched annotation
boxes given (`gt_boxes` / `gt_of_frame`) the training labels of those methods come back too: the per-point
mask label from a second kernel over the same draws (dal3_*_labels: det3d's points_in_rbbox), the per-item
centre / heading / size labels from O(B) host arithmetic (tools/utils.py:53-67).

Tracks use the reference's schema: dict with per-frame lists 'bbox' (7,) global frame,
'point' (k,3) float64 global frame, 'score'. The host side only concatenates arrays, inverts the 4x4 poses,
moves ONE box per static track to the vehicle frame and (sampler="numpy") draws the resampling indices from the
global NumPy stream in the reference's order; every per-point operation runs on the GPU in float64.
"""
import numpy as np
import torch

from . import _hip, arch, geom

_MEAN_SIZE = np.array(arch.MEAN_SIZE)
_N_HEADING = 12


def _angle2class(angle, num_class=_N_HEADING):
    """tools/utils.py:53-60"""
    angle = angle % (2 * np.pi)
    per = 2 * np.pi / float(num_class)
    shifted = (angle + per / 2) % (2 * np.pi)
    cid = int(shifted / per)
    return cid, shifted - (cid * per + per / 2)


def _size2class(lwh):
    """tools/utils.py:62-67"""
    cid = int(np.argmin(np.linalg.norm(lwh[np.newaxis, ...] - _MEAN_SIZE, axis=1)))
    return cid, lwh - _MEAN_SIZE[cid]


def _item_labels(bbox_gt, heading_ref, center_ref, dev):
    """per-item labels from the (B,7) float32 annotation boxes; heading_ref (B,) / center_ref (B,3) or None"""
    hc, hr, sc, sr = [], [], [], []
    for g, h in zip(bbox_gt, heading_ref):
        c, r = _angle2class(g[-1] - h)
        hc.append(c)
        hr.append(r)
        c, r = _size2class(g[3:6])
        sc.append(c)
        sr.append(r)
    center = bbox_gt[:, :3] if center_ref is None else bbox_gt[:, :3] - center_ref
    return {"bbox_gt": torch.from_numpy(np.ascontiguousarray(bbox_gt)).to(dev),
            "center_label": torch.from_numpy(np.ascontiguousarray(center)).to(dev),
            "heading_class_label": torch.tensor(hc, dtype=torch.int64, device=dev),
            "heading_residuals_label": torch.tensor(np.array(hr, dtype=np.float64), device=dev),
            "size_class_label": torch.tensor(sc, dtype=torch.int64, device=dev),
            "size_residual_label": torch.from_numpy(np.stack(sr)).to(dev)}


def _transform_box(box, pose):
    """static_model.py:574-588 for one box / one pose (float64)."""
    heading = box[..., -1] + np.arctan2(pose[1, 0], pose[0, 0])
    center = np.einsum("...ij,...nj->...ni", pose[0:3, 0:3], box[..., 0:3]) + np.expand_dims(pose[0:3, 3], axis=-2)
    return np.concatenate([center, box[..., 3:6], heading[..., None]], axis=-1)


def _dev(a, device, dtype):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(device)


def _transform_boxes(box, pose):
    """_transform_box for (B,7) boxes with one pose EACH, (B,4,4): vectorised, same float64 operations per box"""
    heading = box[:, -1] + np.arctan2(pose[:, 1, 0], pose[:, 0, 0])
    center = np.einsum("bij,bj->bi", pose[:, 0:3, 0:3], box[:, 0:3]) + pose[:, 0:3, 3]
    return np.concatenate([center, box[:, 3:6], heading[:, None]], axis=-1)


def _upload(arr, dev):
    """host array -> device through a pinned staging buffer (one copy into it, one asynchronous DMA out of it)"""
    arr = np.ascontiguousarray(arr)
    if dev.type != "cuda" or arr.nbytes < (1 << 20):
        return torch.from_numpy(arr).to(dev)
    stage = torch.empty(arr.shape, dtype=torch.from_numpy(arr[:0]).dtype, pin_memory=True)
    stage.numpy()[...] = arr
    return stage.to(dev, non_blocking=True)


class StaticTrackStore:
    """Every static track of a segment resident on the device, flattened ONCE: the global-frame points of all frames
    of all tracks in one float64 array (one concatenate over the frames, no per-track intermediate), the per-track
    offsets into it, and each track's best-score frame with its box. `prepare_static_batch(store, poses, first=k)`
    then prepares tracks [k, k+B) with O(B) host arithmetic and no upload (the drivers walk a segment's tracks in
    order, static_eval.py:256-267 through the DataLoader). What the reference does per ITEM instead: stack the
    track's frames, `np.argmax` the scores, invert the pose, move the box (static_model.py:530-544)."""

    def __init__(self, tracks, device="cuda", veh_to_global=None):
        """veh_to_global (optional): the flat-16 pose of every track's best-score frame. With it the per-track inverse
        poses and vehicle-frame boxes are computed and uploaded HERE, once, and `prepare_static_batch(store, B, first=k)`
        is a single kernel launch with no host arithmetic and no upload (round 5)."""
        dev = torch.device(device)
        frames, counts = [], []
        for tr in tracks:
            n = 0
            for p in tr["point"]:
                p = np.asarray(p, np.float64).reshape(-1, 3)
                frames.append(p)
                n += p.shape[0]
            counts.append(n)
        self.tracks = tracks
        self.offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        self.n_points = np.asarray(counts, np.int64)
        self.best = np.array([int(np.argmax(np.stack(tr["score"]))) for tr in tracks], np.int64)
        self.best_box = (np.stack([np.asarray(tr["bbox"][b], np.float64).reshape(7) for tr, b in zip(tracks, self.best)])
                         if tracks else np.zeros((0, 7)))
        flat = np.concatenate(frames) if frames else np.zeros((0, 3))
        self.pts = _upload(flat, dev)
        self.d_offsets = torch.from_numpy(self.offsets).to(dev)
        self.device = dev
        self.d_pose = self.d_box = None
        if veh_to_global is not None:
            if len(veh_to_global) != len(tracks):
                raise ValueError("StaticTrackStore: one best-frame pose per track")
            pose = np.linalg.inv(np.reshape(np.asarray(veh_to_global, np.float64), [len(tracks), 4, 4]))
            self.d_pose = _dev(pose.reshape(len(tracks), 16), dev, np.float64)
            self.d_box = _dev(_transform_boxes(self.best_box, pose), dev, np.float64)

    def __len__(self):
        return len(self.tracks)


def prepare_static_batch(tracks, veh_to_global, n_points=4096, sampler="numpy", seed=10922081, item_offset=0,
                         device="cuda", gt_boxes=None, first=0):
    """tracks: list of track dicts, or a StaticTrackStore (then tracks [first, first + len(veh_to_global)) of it are
    prepared and nothing is uploaded); veh_to_global: list of flat-16 poses of each track's BEST-score frame
    (annos['veh_to_global'], static_model.py:538) — or, for a store built WITH its poses, just the batch size (int):
    the call is then one kernel launch (device sampler, no labels). Returns (pts (B,3,N) fp32 view of point-major storage,
    init_box (B,7) fp32) — exactly what static_eval.py:265-266 feeds forward().
    gt_boxes: optional list of the matched annotation's float32 (9,) `box` of that frame (static_model.py:550-553);
    then a third value is returned, the labels of static_model.py:548-566 as a dict of device tensors:
    bbox_gt (B,7), mask_label (B,N) u8, center_label, heading_class_label, heading_residuals_label,
    size_class_label, size_residual_label."""
    store = tracks if isinstance(tracks, StaticTrackStore) else StaticTrackStore(tracks, device)
    if not isinstance(tracks, StaticTrackStore):
        first = 0
    resident = isinstance(veh_to_global, (int, np.integer))
    B = int(veh_to_global) if resident else len(veh_to_global)
    if first < 0 or first + B > len(store):
        raise ValueError(f"prepare_static_batch: tracks [{first}, {first + B}) are not in the store of {len(store)}")
    dev = store.device
    if resident:
        if store.d_pose is None or sampler != "device" or gt_boxes is not None:
            raise ValueError("prepare_static_batch(store, B): the store must hold its poses (StaticTrackStore(..., veh_to_global=)), "
                             "sampler='device', no labels")
        out = torch.empty((B, n_points, 3), dtype=torch.float32, device=dev)
        init = torch.empty((B, 7), dtype=torch.float32, device=dev)
        _hip.check(_hip.lib().dal3_static_crop_prep(_hip.ptr(store.pts), _hip.ptr(store.d_offsets[first:first + B + 1]), None,
                                                    _hip.ptr(store.d_pose[first:first + B]), _hip.ptr(store.d_box[first:first + B]),
                                                    B, n_points, seed, item_offset, _hip.ptr(out), _hip.ptr(init), _hip.stream()))
        return out.transpose(2, 1), init
    pose = np.linalg.inv(np.reshape(np.asarray(veh_to_global, np.float64), [B, 4, 4]))          # one batched inverse
    boxes = _transform_boxes(store.best_box[first:first + B], pose)
    choice = None
    if sampler == "numpy":                                                      # static_model.py:546, track by track
        choice = np.empty((B, n_points), np.int32)
        for b in range(B):
            choice[b] = np.random.choice(int(store.n_points[first + b]), n_points, replace=True)
    d_pts = store.pts
    d_off = store.d_offsets[first:first + B + 1]
    d_pose = _dev(pose.reshape(B, 16), dev, np.float64)
    d_box = _dev(boxes, dev, np.float64)
    d_choice = _dev(choice, dev, np.int32) if choice is not None else None
    out = torch.empty((B, n_points, 3), dtype=torch.float32, device=dev)
    init = torch.empty((B, 7), dtype=torch.float32, device=dev)
    _hip.check(_hip.lib().dal3_static_crop_prep(_hip.ptr(d_pts), _hip.ptr(d_off), _hip.ptr(d_choice), _hip.ptr(d_pose),
                                                _hip.ptr(d_box), B, n_points, seed, item_offset, _hip.ptr(out),
                                                _hip.ptr(init), _hip.stream()))
    if gt_boxes is None:
        return out.transpose(2, 1), init
    bbox_gt = np.stack([np.asarray(g)[[0, 1, 2, 3, 4, 5, -1]] for g in gt_boxes])          # float32, as stored
    d_planes = geom.planes_to_device(geom.box_planes(bbox_gt), dev)
    mask = torch.empty((B, n_points), dtype=torch.uint8, device=dev)
    _hip.check(_hip.lib().dal3_static_crop_labels(_hip.ptr(d_pts), _hip.ptr(d_off), _hip.ptr(d_choice), _hip.ptr(d_pose), B,
                                                  n_points, seed, item_offset, _hip.ptr(d_planes), _hip.ptr(mask),
                                                  _hip.stream()))
    labels = _item_labels(bbox_gt, [b[-1] for b in boxes], None, dev)
    labels["mask_label"] = mask
    return out.transpose(2, 1), init, labels


class TrackStore:
    """All frames of a list of tracks resident on the device, uploaded once: the global-frame points of every frame
    stacked (float64), the frame offsets into them, the per-frame boxes, and each track's first frame. Pass it to
    prepare_dynamic_batch in place of the list of tracks when many batches draw from the same tracks
    (eval.refine_dynamic_tracks does)."""

    def __init__(self, tracks, device="cuda"):
        dev = torch.device(device)
        frame_pts, frame_off, boxes, track_first = [], [0], [], [0]
        for tr in tracks:
            for p, bx in zip(tr["point"], tr["bbox"]):
                p = np.asarray(p, np.float64).reshape(-1, 3)
                frame_pts.append(p)
                frame_off.append(frame_off[-1] + p.shape[0])
                boxes.append(np.asarray(bx, np.float64).reshape(7))
            track_first.append(track_first[-1] + len(tr["point"]))
        self.tracks = tracks
        self.pts = _dev(np.vstack(frame_pts) if frame_pts else np.zeros((0, 3)), dev, np.float64)
        self.frame_off = _dev(np.array(frame_off), dev, np.int64)
        self.boxes = _dev(np.stack(boxes), dev, np.float64)
        self.track_first = _dev(np.array(track_first), dev, np.int64)


def prepare_dynamic_batch(tracks, items, veh_to_global, n_per_frame=1024, r=2, s=50, sampler="numpy", seed=10922081,
                          item_offset=0, device="cuda", gt_of_frame=None, pose_of_frame=None):
    """tracks: list of track dicts (or a TrackStore of them); items: list of (track_index, frame_index);
    veh_to_global: flat-16 pose of each item's own frame (dynamic_model.py:449-451). Returns (pts (B,4,5*n) view,
    box (B,8,2s+1) view, init_box (B,8)) as dynamic_eval.py:222-223 builds them.
    Labels (dynamic_model.py:455-501) when gt_of_frame(track_index, frame) -> float32 (9,) annotation box or None
    and pose_of_frame(track_index, frame) -> flat-16 veh_to_global are given: a fourth return value, dict with
    bbox_gt, mask_label (B,5*n) u8, center_label, heading_class_label, heading_residuals_label, size_class_label,
    size_residual_label. Every item's own frame must have its annotation (the reference redraws another item
    otherwise, dynamic_model.py:487-489; that choice is the caller's)."""
    B = len(items)
    dev = torch.device(device)
    store = tracks if isinstance(tracks, TrackStore) else TrackStore(tracks, dev)
    tracks = store.tracks
    poses = np.linalg.inv(np.reshape(np.asarray(veh_to_global, np.float64), [B, 4, 4])).reshape(B, 16)
    choice = None
    if sampler == "numpy":
        choice = np.zeros((B, 2 * r + 1, n_per_frame), np.int32)
        for b, (t, it) in enumerate(items):
            n_frames = len(tracks[t]["point"])
            for j, i in enumerate(range(it - r, it + r + 1)):                     # dynamic_model.py:430-437
                if 0 <= i < n_frames and len(tracks[t]["point"][i]) > 0:
                    choice[b, j] = np.random.choice(len(tracks[t]["point"][i]), n_per_frame, replace=True)
    d_pts, d_foff, d_boxes, d_first = store.pts, store.frame_off, store.boxes, store.track_first
    d_it = _dev(np.array([t for t, _ in items]), dev, np.int32)
    d_if = _dev(np.array([i for _, i in items]), dev, np.int32)
    d_pose = _dev(poses, dev, np.float64)
    d_choice = _dev(choice, dev, np.int32) if choice is not None else None
    n = (2 * r + 1) * n_per_frame
    pts = torch.empty((B, n, 4), dtype=torch.float32, device=dev)
    box = torch.empty((B, 2 * s + 1, 8), dtype=torch.float32, device=dev)
    init = torch.empty((B, 8), dtype=torch.float32, device=dev)
    _hip.check(_hip.lib().dal3_dynamic_item_prep(_hip.ptr(d_pts), _hip.ptr(d_foff), _hip.ptr(d_boxes), _hip.ptr(d_first),
                                                 _hip.ptr(d_it), _hip.ptr(d_if), _hip.ptr(d_choice), _hip.ptr(d_pose), B,
                                                 n_per_frame, r, s, seed, item_offset, _hip.ptr(pts), _hip.ptr(box),
                                                 _hip.ptr(init), _hip.stream()))
    if gt_of_frame is None:
        return pts.transpose(2, 1), box.transpose(2, 1), init
    w = 2 * r + 1
    xform = np.zeros((B, w, 16))
    win_box = np.zeros((B, w, 7), np.float32)
    valid = np.zeros((B, w), np.uint8)
    bbox_gt, centre_box = [], []
    for b, (t, it) in enumerate(items):
        pose = poses[b].reshape(4, 4)
        pose_back = np.linalg.inv(pose)                                          # dynamic_model.py:481
        for j, i in enumerate(range(it - r, it + r + 1)):
            g = gt_of_frame(t, i) if 0 <= i < len(tracks[t]["bbox"]) else None
            if g is None:
                continue
            valid[b, j] = 1
            win_box[b, j] = np.asarray(g)[[0, 1, 2, 3, 4, 5, -1]]
            xform[b, j] = (np.linalg.inv(np.reshape(pose_of_frame(t, i), [4, 4])) @ pose_back).reshape(16)
        if not valid[b, r]:
            raise ValueError(f"item {b} (track {t}, frame {it}) has no matched annotation in its own frame")
        bbox_gt.append(win_box[b, r].copy())
        cb = np.asarray(tracks[t]["bbox"][it], np.float64)
        centre_box.append(_transform_box(cb[None, :], pose)[0])
    bbox_gt, centre_box = np.stack(bbox_gt), np.stack(centre_box)
    d_planes = geom.planes_to_device(geom.box_planes(win_box.reshape(B * w, 7)), dev)
    d_xform = _dev(xform, dev, np.float64)
    d_valid = _dev(valid, dev, np.uint8)
    mask = torch.empty((B, n), dtype=torch.uint8, device=dev)
    _hip.check(_hip.lib().dal3_dynamic_item_labels(_hip.ptr(d_pts), _hip.ptr(d_foff), _hip.ptr(d_first), _hip.ptr(d_it),
                                                   _hip.ptr(d_if), _hip.ptr(d_choice), _hip.ptr(d_pose), B, n_per_frame, r,
                                                   seed, item_offset, _hip.ptr(d_xform), _hip.ptr(d_planes),
                                                   _hip.ptr(d_valid), _hip.ptr(mask), _hip.stream()))
    labels = _item_labels(bbox_gt, centre_box[:, 6], centre_box[:, :3], dev)
    labels["mask_label"] = mask
    return pts.transpose(2, 1), box.transpose(2, 1), init, labels
