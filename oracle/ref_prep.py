"""ORACLE — CPU restatement of the crop preparation that runs right before the heads:
what `STATICTRACK.__getitem__` (tools/static_model.py:529-572) and `DYNAMICTRACK.__getitem__`
(tools/dynamic_model.py:419-509) hand to `forward()`, minus file I/O and minus the training labels.

TEST INFRASTRUCTURE ONLY (same rules as oracle/ref_heads.py).

Parity pin: tests/golden/prep_static.npz and prep_dynamic.npz hold the outputs of the REAL Dataset classes
(tests/golden/gen_golden.py builds synthetic tracks + annotation pickles in a temp directory and calls the
reference's __getitem__ under np.random.seed); tests/test_oracle_prep.py checks this file against them.
The training LABELS of those methods (mask / centre / heading / size: static_model.py:548-566,
dynamic_model.py:455-501) are restated by static_labels / dynamic_labels and pinned to the same fixtures; their
mask goes through det3d's points_in_rbbox, which gen_golden.py runs from the reference's files with numba's
decorators as identities (see oracle/ref_geom.py).

All arithmetic is NumPy float64, as in the reference; the drivers cast to fp32 afterwards
(`pts.transpose(2,1).float()`, static_eval.py:265).
"""
import numpy as np

from oracle import ref_geom
from oracle.ref_heads import NUM_HEADING_BIN, angle2class, size2class


def rotz(angle):
    """static_model.py:590-598"""
    c, s = np.cos(angle), np.sin(angle)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


def transform_box(box, pose):
    """static_model.py:574-588: (..., 7) upright boxes moved by the rigid 4x4 `pose`."""
    heading = box[..., -1] + np.arctan2(pose[1, 0], pose[0, 0])
    # same contraction (and therefore the same float64 summation order) as the reference's einsum
    center = np.einsum("...ij,...nj->...ni", pose[0:3, 0:3], box[..., 0:3]) + np.expand_dims(pose[0:3, 3], axis=-2)
    return np.concatenate([center, box[..., 3:6], heading[..., None]], axis=-1)


def static_crop(points_global, boxes, scores, veh_to_global, n_points, aux=None):
    """static_model.py:529-546, 568-572. points_global (P,3): all frames' points of one track stacked, global
    frame; boxes (F,7), scores (F,): per-frame detections; veh_to_global: flat 16 of the best-score frame.
    Draws `np.random.choice(P, n_points, replace=True)` from the global stream exactly as the reference does.
    Returns init_box (1,7) [vehicle frame of the best frame] and point (n_points,3) [box-centred, box-aligned]."""
    best = int(np.argmax(scores))
    pose = np.linalg.inv(np.reshape(veh_to_global, [4, 4]))
    bbox = transform_box(boxes[best][None, :], pose)
    pt = (pose @ np.concatenate([points_global.T, np.ones((1, points_global.shape[0]))], axis=0))[:3, :].T
    choice = np.random.choice(pt.shape[0], n_points, replace=True)
    pt = pt[choice, :]
    if aux is not None:
        aux["point_vehicle"] = pt                      # what the labels are computed on (static_labels)
    pt = pt - bbox[:, :3]
    pt = (rotz(-bbox[0, -1]) @ pt.T).T
    return bbox, pt, choice


def dynamic_item(frame_points, frame_boxes, item_idx, veh_to_global, n_points=1024, r=2, s=50, aux=None):
    """dynamic_model.py:429-447, 449-453, 490-507. frame_points: list of (P_f,3) global-frame arrays (possibly
    empty), frame_boxes: list of (7,) global boxes, one per frame of the track. Quirks preserved: missing /
    empty frames contribute ZERO points that are still moved by the pose and the re-centring; missing boxes are
    zero rows that are still pose-transformed; points are rotated by -yaw of the centre box, boxes are only
    translated (their yaw made relative); init_box is the centre box BEFORE re-centring.
    Returns init_box (8,), bbox (2s+1, 8), point ((2r+1)*n_points, 4), and the list of draws."""
    n_frames = len(frame_points)
    point = np.zeros((0, 4))
    draws = []
    for j, i in enumerate(range(item_idx - r, item_idx + r + 1)):
        t = np.full((n_points, 1), 0.1 * (j - r))
        if 0 <= i < n_frames and len(frame_points[i]) > 0:
            choice = np.random.choice(len(frame_points[i]), n_points, replace=True)
            draws.append(choice)
            point = np.vstack([point, np.hstack([np.copy(frame_points[i][choice]), t])])
        else:
            draws.append(None)
            point = np.vstack([point, np.hstack([np.zeros((n_points, 3)), t])])
    bbox = np.zeros((0, 8))
    for j, i in enumerate(range(item_idx - s, item_idx + s + 1)):
        row = np.zeros((1, 7)) if (i < 0 or i >= len(frame_boxes)) else np.copy(frame_boxes[i].reshape((1, 7)))
        bbox = np.vstack([bbox, np.hstack([row, np.full((1, 1), 0.1 * (j - s))])])
    pose = np.linalg.inv(np.reshape(veh_to_global, [4, 4]))
    bbox[:, :7] = transform_box(bbox[:, :7], pose)
    point[:, :3] = (pose @ np.concatenate([point[:, :3].T, np.ones((1, point.shape[0]))], axis=0)).T[:, :3]
    init_box = np.copy(bbox[s])
    if aux is not None:
        aux["point_item"], aux["bbox_item"] = np.copy(point), np.copy(bbox)      # inputs of dynamic_labels
    point[:, :3] = point[:, :3] - bbox[s, :3]
    point[:, :3] = (rotz(-bbox[s, -2]) @ point[:, :3].T).T
    bbox[:, :3] = bbox[:, :3] - bbox[s, :3]
    bbox[:, -2] = bbox[:, -2] - bbox[s, -2]
    return init_box, bbox, point, draws


def static_labels(point_vehicle, init_box, gt_box9):
    """static_model.py:548-566. point_vehicle (N,3) float64: the resampled points in the best frame's VEHICLE
    frame (static_crop's `pt[choice]` before re-centring); init_box (1,7) float64; gt_box9: the matched
    annotation's float32 (9,) box. Returns bbox_gt (7,) float32, mask_label (N,) float64, center_label,
    heading class / residual, size class / residual."""
    bbox_gt = gt_box9[[0, 1, 2, 3, 4, 5, -1]]
    mask = ref_geom.points_in_rbbox(point_vehicle, bbox_gt[np.newaxis, ...]).astype(float).squeeze()
    hc, hr = angle2class(bbox_gt[-1] - init_box[0, -1], NUM_HEADING_BIN)
    sc, sr = size2class(bbox_gt[3:6])
    return bbox_gt, mask, bbox_gt[:3], hc, hr, sc, sr


def dynamic_labels(point_item, bbox_item, item_idx, n_frames, pose_item_v2g, gt_of_frame, pose_of_frame, n_points=1024,
                   r=2, s=50):
    """dynamic_model.py:455-501. point_item ((2r+1)*n,4) / bbox_item (2s+1,8): the item AFTER the pose transform
    and BEFORE the re-centring (i.e. in the centre frame's vehicle frame); gt_of_frame(i) -> float32 (9,) box or
    None; pose_of_frame(i) -> flat-16 veh_to_global of track frame i. A window frame's points are moved into THAT
    frame's vehicle frame (`_pose @ inv(pose) @ p`) and tested against that frame's annotation."""
    pose = np.linalg.inv(np.reshape(pose_item_v2g, [4, 4]))
    mask = np.zeros((0, n_points))
    bbox_gt = None
    for j, i in enumerate(range(item_idx - r, item_idx + r + 1)):
        row = np.zeros((1, n_points))
        if 0 <= i < n_frames and gt_of_frame(i) is not None:
            bbox_t = gt_of_frame(i)[[0, 1, 2, 3, 4, 5, -1]]
            if i == item_idx:
                bbox_gt = np.copy(bbox_t)
            _pose = np.linalg.inv(np.reshape(pose_of_frame(i), [4, 4]))
            p = np.copy(point_item[j * n_points:(j + 1) * n_points, :3]).T
            p = _pose @ np.linalg.inv(pose) @ np.vstack([p, np.ones((1, p.shape[1]))])
            row = ref_geom.points_in_rbbox(p.T[:, :3], bbox_t[np.newaxis, ...]).reshape((1, n_points))
        mask = np.vstack([mask, row])
    hc, hr = angle2class(bbox_gt[-1] - bbox_item[s, -2], NUM_HEADING_BIN)
    sc, sr = size2class(bbox_gt[3:6])
    return bbox_gt, mask.flatten().astype(float), bbox_gt[:3] - bbox_item[s, :3], hc, hr, sc, sr
