"""Write-back of refined boxes into the per-frame detections: the det_annos update of the
reference's `postprocessing` (tools/static_eval.py:62-167, tools/dynamic_eval.py:43-141) on the device through
dal3_writeback_boxes. The host side flattens (track, frame) pairs and the frames' detection arrays; the
box transforms, the 0.1 m centre match and the overwrite run on the GPU in float64. The IoU metrics that
function also logs depend on an un-vendored module and are out of scope.
"""
import numpy as np
import torch

from . import _hip


def _run(tracks, veh_to_global, has_gt, final_bboxes, dets, static, device):
    tokens = list(dets.keys())
    start, off = {}, 0
    for t in tokens:
        start[t] = off
        off += len(dets[t])
    det_all = np.concatenate([np.asarray(dets[t], np.float32).reshape(-1, 7) for t in tokens], 0)
    inv = {t: np.linalg.inv(np.reshape(veh_to_global[t], [4, 4])).reshape(16) for t in tokens}
    f_idx, p_best, p_inv, tbox, d_start, d_cnt, act = [], [], [], [], [], [], []
    index = 0
    for i, tr in enumerate(tracks):
        best = tr["token"][int(np.argmax(np.stack(tr["score"])))]
        for j, t in enumerate(tr["token"]):
            f_idx.append(i if static else index + j)
            p_best.append(np.asarray(veh_to_global[best], np.float64).reshape(16))
            p_inv.append(inv[t])
            tbox.append(np.asarray(tr["bbox"][j], np.float64).reshape(7))
            d_start.append(start[t])
            d_cnt.append(len(dets[t]))
            act.append(1 if has_gt[(i, t)] else 0)
        index += len(tr["token"])
    P = len(f_idx)
    dev = torch.device(device)

    def up(a, dt):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
    final = final_bboxes.to(device=dev, dtype=torch.float64).contiguous() if torch.is_tensor(final_bboxes) \
        else up(final_bboxes, np.float64)
    d_det = up(det_all, np.float32)
    d_fi, d_pb, d_pi, d_tb = up(f_idx, np.int32), up(np.stack(p_best), np.float64), up(np.stack(p_inv), np.float64), \
        up(np.stack(tbox), np.float64)
    d_ds, d_dc, d_act = up(d_start, np.int64), up(d_cnt, np.int32), up(act, np.uint8)
    match = torch.empty(P, dtype=torch.int32, device=dev)
    owner = torch.empty(det_all.shape[0], dtype=torch.int32, device=dev)
    _hip.check(_hip.lib().dal3_writeback_boxes(_hip.ptr(final), _hip.ptr(d_fi), _hip.ptr(d_pb) if static else None,
                                               _hip.ptr(d_pi), _hip.ptr(d_tb), _hip.ptr(d_det), _hip.ptr(d_ds),
                                               _hip.ptr(d_dc), _hip.ptr(d_act), P, det_all.shape[0], _hip.ptr(match),
                                               _hip.ptr(owner), _hip.stream()))
    out = d_det.cpu().numpy()
    m = match.cpu().numpy()
    missing = [k for k in range(P) if act[k] and m[k] < 0]
    if missing:
        raise AssertionError("Bounding box not in det_annos.")          # the reference's assert (static_eval.py:155)
    return {t: out[start[t]:start[t] + len(dets[t])] for t in tokens}, m


def writeback_static(tracks, veh_to_global, has_gt, final_bboxes, dets, device="cuda"):
    """final_bboxes (n_tracks,7): one refined box per track in its best-score frame's vehicle frame
    (static_eval.py:75-87). dets {token: (n,7)} -> updated copy; has_gt {(track_index, token): bool}."""
    return _run(tracks, veh_to_global, has_gt, final_bboxes, dets, True, device)


def writeback_dynamic(tracks, veh_to_global, has_gt, final_bboxes, dets, device="cuda"):
    """final_bboxes (sum of track lengths, 7): one refined box per track-frame, already in that frame
    (dynamic_eval.py:64)."""
    return _run(tracks, veh_to_global, has_gt, final_bboxes, dets, False, device)
