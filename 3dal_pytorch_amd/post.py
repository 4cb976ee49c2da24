"""Write-back of refined boxes into the per-frame detections: the det_annos update of the
reference's `postprocessing` (tools/static_eval.py:62-167, tools/dynamic_eval.py:43-141) on the device through
dal3_writeback_boxes. The host side flattens (track, frame) pairs and the frames' detection arrays — once per segment
(WritebackPlan) —; the box transforms, the 0.1 m centre match and the overwrite run on the GPU in float64. The IoU metrics that
function also logs depend on an un-vendored module and are out of scope.
"""
import numpy as np
import torch

from . import _hip


class WritebackPlan:
    """Everything about a segment's write-back that does not depend on the refined boxes, flattened and uploaded ONCE
    (round 4; as StaticTrackStore did for crop preparation): the (track, frame) pairs with their pose products, the
    frames' detection arrays side by side, the per-pair detection ranges and the ground-truth gate. The tracks, poses
    and detections of a segment are known before the heads run, so a caller builds the plan while the GPU is busy and
    the call that follows the heads is one kernel launch and one download: 10 ms of host flattening per call -> the
    plan's build, off the critical path.

        plan = WritebackPlan(tracks, veh_to_global, has_gt, dets, static=True)
        new_dets, match = plan.apply(final_bboxes)       # any number of times

    The detection array on the device is restored from a pristine copy before every apply (the kernel overwrites it)."""

    def __init__(self, tracks, veh_to_global, has_gt, dets, static, device="cuda"):
        self.static = bool(static)
        self.tokens = list(dets.keys())
        self.start, off = {}, 0
        for t in self.tokens:
            self.start[t] = off
            off += len(dets[t])
        self.lens = {t: len(dets[t]) for t in self.tokens}
        det_all = np.concatenate([np.asarray(dets[t], np.float32).reshape(-1, 7) for t in self.tokens], 0)
        inv = {t: np.linalg.inv(np.reshape(veh_to_global[t], [4, 4])).reshape(16) for t in self.tokens}
        f_idx, p_best, p_inv, tbox, d_start, d_cnt, act = [], [], [], [], [], [], []
        index = 0
        for i, tr in enumerate(tracks):
            best = tr["token"][int(np.argmax(np.stack(tr["score"])))]
            for j, t in enumerate(tr["token"]):
                f_idx.append(i if static else index + j)
                p_best.append(np.asarray(veh_to_global[best], np.float64).reshape(16))
                p_inv.append(inv[t])
                tbox.append(np.asarray(tr["bbox"][j], np.float64).reshape(7))
                d_start.append(self.start[t])
                d_cnt.append(self.lens[t])
                act.append(1 if has_gt[(i, t)] else 0)
            index += len(tr["token"])
        self.P = len(f_idx)
        self.n_final = len(tracks) if static else index
        self.act = np.asarray(act, np.uint8)
        self.dev = dev = torch.device(device)

        def up(a, dt):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
        self.n_det = det_all.shape[0]
        self.d_det0 = up(det_all, np.float32)                          # pristine
        self.d_det = torch.empty_like(self.d_det0)
        self.d_fi, self.d_pi, self.d_tb = up(f_idx, np.int32), up(np.stack(p_inv), np.float64), up(np.stack(tbox), np.float64)
        self.d_pb = up(np.stack(p_best), np.float64) if static else None
        self.d_ds, self.d_dc, self.d_act = up(d_start, np.int64), up(d_cnt, np.int32), up(self.act, np.uint8)
        self.match = torch.empty(self.P, dtype=torch.int32, device=dev)
        self.owner = torch.empty(self.n_det, dtype=torch.int32, device=dev)

    def launch(self, final_bboxes):
        """enqueue only (current stream): the refined boxes -> self.d_det / self.match on the device"""
        final = final_bboxes.to(device=self.dev, dtype=torch.float64).contiguous() if torch.is_tensor(final_bboxes) \
            else torch.from_numpy(np.ascontiguousarray(final_bboxes, dtype=np.float64)).to(self.dev)
        if final.shape != (self.n_final, 7):
            raise ValueError(f"final_bboxes must be ({self.n_final}, 7), got {tuple(final.shape)}")
        self.d_det.copy_(self.d_det0)
        _hip.check(_hip.lib().dal3_writeback_boxes(_hip.ptr(final), _hip.ptr(self.d_fi), _hip.ptr(self.d_pb), _hip.ptr(self.d_pi),
                                                   _hip.ptr(self.d_tb), _hip.ptr(self.d_det), _hip.ptr(self.d_ds),
                                                   _hip.ptr(self.d_dc), _hip.ptr(self.d_act), self.P, self.n_det,
                                                   _hip.ptr(self.match), _hip.ptr(self.owner), _hip.stream()))

    def apply(self, final_bboxes):
        self.launch(final_bboxes)
        out = self.d_det.cpu().numpy()
        m = self.match.cpu().numpy()
        if bool(((m < 0) & (self.act != 0)).any()):
            raise AssertionError("Bounding box not in det_annos.")          # the reference's assert (static_eval.py:155)
        return {t: out[self.start[t]:self.start[t] + self.lens[t]] for t in self.tokens}, m


def _run(tracks, veh_to_global, has_gt, final_bboxes, dets, static, device):
    return WritebackPlan(tracks, veh_to_global, has_gt, dets, static, device).apply(final_bboxes)


def writeback_static(tracks, veh_to_global, has_gt, final_bboxes, dets, device="cuda"):
    """final_bboxes (n_tracks,7): one refined box per track in its best-score frame's vehicle frame
    (static_eval.py:75-87). dets {token: (n,7)} -> updated copy; has_gt {(track_index, token): bool}."""
    return _run(tracks, veh_to_global, has_gt, final_bboxes, dets, True, device)


def writeback_dynamic(tracks, veh_to_global, has_gt, final_bboxes, dets, device="cuda"):
    """final_bboxes (sum of track lengths, 7): one refined box per track-frame, already in that frame
    (dynamic_eval.py:64)."""
    return _run(tracks, veh_to_global, has_gt, final_bboxes, dets, False, device)
