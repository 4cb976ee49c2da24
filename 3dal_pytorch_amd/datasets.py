"""`STATICTRACK` / `DYNAMICTRACK`: the Dataset classes the reference's eval AND train drivers import from the model
modules (`from static_model import STATICTRACK`, static_eval.py:9, static_train.py:13; `from dynamic_model import
DYNAMICTRACK`, dynamic_eval.py / dynamic_train.py) — so a drop-in for those modules has to carry them.

They are the host-side, one-item-at-a-time path a `DataLoader` drives (tools/static_model.py:519-598,
tools/dynamic_model.py:399-540): same constructor, `__len__`, the same 11- / 12-tuple from `__getitem__` (same
dtypes, same consumption of the global NumPy stream, same quirks — see prep.py's notes), plus `transform_box` and
`rotz`. Differences: annotation pickles are cached per Dataset instead of being re-read for every item (up to six
reads per dynamic item in the reference), and a track without its matched annotation raises a KeyError that says
so (the reference dies on an unbound local). For throughput, `prep.prepare_static_batch` /
`prepare_dynamic_batch` do the same work for a whole batch on the GPU; this module is the API-compatible path.

`points_in_rbbox` here is det3d's (box_np_ops.py:641-647) on the host with NumPy — a DataLoader worker has no GPU;
the device version is geom.points_in_rbbox.
"""
import pickle

import numpy as np
import torch
from torch.utils.data import Dataset

from . import arch, geom

NUM_HEADING_BIN = arch.NUM_HEADING_BIN
_MEAN_SIZE = np.array(arch.MEAN_SIZE)


def points_in_rbbox(points, rbbox):
    """(P,>=3), (K,7) -> (P,K) bool; a point is outside a box as soon as one face gives n.p + d >= 0"""
    pl = geom.box_planes(np.asarray(rbbox))
    p = np.asarray(points)[:, :3]
    s = (p[:, None, None, 0] * pl[None, :, :, 0] + p[:, None, None, 1] * pl[None, :, :, 1]
         + p[:, None, None, 2] * pl[None, :, :, 2] + pl[None, :, :, 3])
    return ~(s >= 0).any(axis=2)


def angle2class(angle, num_class):
    """tools/utils.py:53-60"""
    angle = angle % (2 * np.pi)
    per = 2 * np.pi / float(num_class)
    shifted = (angle + per / 2) % (2 * np.pi)
    cid = int(shifted / per)
    return cid, shifted - (cid * per + per / 2)


def size2class(lwh):
    """tools/utils.py:62-67"""
    cid = np.argmin(np.linalg.norm(lwh[np.newaxis, ...] - _MEAN_SIZE, axis=1))
    return cid, lwh - _MEAN_SIZE[cid]


class _TrackDataset(Dataset):
    def __init__(self, track, infos, npoints):
        self.trackID = list(track.keys())
        self.track = list(track.values())
        self.npoints = npoints
        self.infos = infos
        self._annos = {}

    def _anno(self, token):
        a = self._annos.get(token)
        if a is None:
            with open(self.infos[token]["anno_path"], "rb") as f:
                a = pickle.load(f)
            self._annos[token] = a
        return a

    @staticmethod
    def _matched_box(annos, name):
        for obj in annos["objects"]:
            if obj["name"] == name:
                return obj["box"][[0, 1, 2, 3, 4, 5, -1]]
        return None

    def transform_box(self, box, pose):
        """(...,7) upright boxes moved by the rigid 4x4 `pose` (static_model.py:574-588)"""
        heading = box[..., -1] + np.arctan2(pose[..., 1, 0], pose[..., 0, 0])
        center = np.einsum("...ij,...nj->...ni", pose[..., 0:3, 0:3], box[..., 0:3]) + np.expand_dims(pose[..., 0:3, 3], axis=-2)
        return np.concatenate([center, box[..., 3:6], heading[..., np.newaxis]], axis=-1)

    def rotz(self, angle):
        c, s = np.cos(angle), np.sin(angle)
        return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])

    @staticmethod
    def _to_vehicle(pose, xyz):
        return (pose @ np.concatenate([xyz.T, np.ones((1, xyz.shape[0]))], axis=0))


class STATICTRACK(_TrackDataset):
    """one item per static track: all frames' points in the best-score frame's box frame (static_model.py:519-572)"""

    def __init__(self, track, infos, npoints=4096):
        super().__init__(track, infos, npoints)

    def __len__(self):
        return len(self.track)

    def __getitem__(self, index):
        tr = self.track[index]
        score = np.stack(tr["score"])
        best = np.argmax(score)
        token = tr["token"][best]
        annos = self._anno(token)
        pose = np.linalg.inv(np.reshape(annos["veh_to_global"], [4, 4]))
        bbox = self.transform_box(np.vstack(tr["bbox"])[best][np.newaxis, ...], pose)
        point = self._to_vehicle(pose, np.vstack(tr["point"]))[:3, :].T
        point = point[np.random.choice(point.shape[0], self.npoints, replace=True), :]

        bbox_gt = self._matched_box(annos, tr["match"][-1])
        if bbox_gt is None:
            raise KeyError(f"track {self.trackID[index]}: annotation {tr['match'][-1]!r} is not in frame {token}")
        mask_label = points_in_rbbox(point, bbox_gt[np.newaxis, ...]).astype(float).squeeze()
        heading_class_label, heading_residuals_label = angle2class(bbox_gt[-1] - bbox[0, -1], NUM_HEADING_BIN)
        size_class_label, size_residual_label = size2class(bbox_gt[3:6])

        point = point - bbox[:, :3]
        point = (self.rotz(-bbox[0, -1]) @ point.T).T
        return (self.trackID[index], torch.from_numpy(bbox), torch.from_numpy(bbox_gt), torch.from_numpy(point), token,
                mask_label, bbox_gt[:3], heading_class_label, heading_residuals_label, size_class_label,
                size_residual_label)


class DYNAMICTRACK(_TrackDataset):
    """one item per (track, frame): a 5-frame point window and a 101-box window (dynamic_model.py:399-509)"""

    def __init__(self, track, infos, npoints=1024):
        super().__init__(track, infos, npoints)
        self.heads = [0]
        for t in self.track:
            self.heads.append(self.heads[-1] + len(t["point"]))
        self.len = self.heads[-1]
        self.r = 2
        self.s = 50

    def __len__(self):
        return self.len

    def __getitem__(self, index):
        track_idx = int(np.searchsorted(self.heads, index, side="right")) - 1
        item_idx = index - self.heads[track_idx]
        tr = self.track[track_idx]
        n, r, s = self.npoints, self.r, self.s
        token = tr["token"][item_idx]

        blocks = []
        for j, i in enumerate(range(item_idx - r, item_idx + r + 1)):
            xyz = np.zeros((n, 3))
            if 0 <= i < len(tr["point"]) and len(tr["point"][i]) > 0:
                xyz = np.copy(tr["point"][i][np.random.choice(len(tr["point"][i]), n, replace=True)])
            blocks.append(np.hstack([xyz, np.full((n, 1), 0.1 * (j - r))]))
        point = np.vstack([np.zeros((0, 4))] + blocks)
        rows = []
        for j, i in enumerate(range(item_idx - s, item_idx + s + 1)):
            row = np.zeros((1, 7)) if (i < 0 or i >= len(tr["bbox"])) else np.copy(tr["bbox"][i].reshape((1, 7)))
            rows.append(np.hstack([row, np.full((1, 1), 0.1 * (j - s))]))
        bbox = np.vstack([np.zeros((0, 8))] + rows)

        pose = np.linalg.inv(np.reshape(self._anno(token)["veh_to_global"], [4, 4]))
        bbox[:, :7] = self.transform_box(bbox[:, :7], pose)
        point[:, :3] = self._to_vehicle(pose, point[:, :3]).T[:, :3]

        bbox_gt = []
        labels = []
        for j, i in enumerate(range(item_idx - r, item_idx + r + 1)):
            row = np.zeros((1, n))
            if 0 <= i < len(tr["bbox"]):
                annos = self._anno(tr["token"][i])
                bbox_t = self._matched_box(annos, tr["match"][-1])
                if bbox_t is not None:
                    if i == item_idx:
                        bbox_gt = np.copy(bbox_t)
                    frame_pose = np.linalg.inv(np.reshape(annos["veh_to_global"], [4, 4]))
                    p = np.copy(point[j * n:(j + 1) * n, :3]).T
                    p = frame_pose @ np.linalg.inv(pose) @ np.vstack([p, np.ones((1, p.shape[1]))])
                    row = points_in_rbbox(p.T[:, :3], bbox_t[np.newaxis, ...]).reshape((1, n))
            labels.append(row)
        mask_label = np.vstack([np.zeros((0, n))] + labels).flatten().astype(float)
        if len(bbox_gt) == 0:                                   # the item's own frame lacks the annotation: redraw
            return self.__getitem__(np.random.randint(self.__len__()))

        init_box = np.copy(bbox[s])
        center_label = bbox_gt[:3] - bbox[s, :3]
        heading_class_label, heading_residual_label = angle2class(bbox_gt[-1] - bbox[s, -2], NUM_HEADING_BIN)
        size_class_label, size_residual_label = size2class(bbox_gt[3:6])

        point[:, :3] = point[:, :3] - bbox[s, :3]
        point[:, :3] = (self.rotz(-bbox[s, -2]) @ point[:, :3].T).T
        bbox[:, :3] = bbox[:, :3] - bbox[s, :3]
        bbox[:, -2] = bbox[:, -2] - bbox[s, -2]
        return (self.trackID[track_idx], torch.from_numpy(init_box), torch.from_numpy(bbox), torch.from_numpy(bbox_gt),
                torch.from_numpy(point), token, mask_label, center_label, heading_class_label, heading_residual_label,
                size_class_label, size_residual_label)
