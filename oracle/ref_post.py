"""ORACLE — CPU restatement of the write-back that follows the heads: the part of
`postprocessing` in tools/static_eval.py:62-167 and tools/dynamic_eval.py:43-141 that carries the refined boxes
into every frame of their track and overwrites the matching detection in `det_annos`.

TEST INFRASTRUCTURE ONLY (same rules as oracle/ref_heads.py).

Parity pin: tests/golden/post_*.npz come from the reference's own `postprocessing` run on synthetic tracks,
pickles and det_annos (tests/golden/gen_golden.py), with `compute_box3d_iou` stubbed: the IoU metrics it logs
depend on the un-vendored fpointnet_train.provider_fpointnet and are NOT pinned; the returned
det_annos — the product of the step — is.
"""
import numpy as np

from .ref_prep import transform_box


def static_writeback(tracks, veh_to_global, has_gt, final_bboxes, det_boxes):
    """static_eval.py:71-87,148-155. tracks: list of track dicts ('bbox' global (7,), 'score', 'token');
    veh_to_global[token] flat 16; has_gt[token] bool (the reference skips frames without the matched GT object);
    final_bboxes (n_tracks,7) refined boxes in each track's best-frame vehicle frame; det_boxes[token] (n,7)
    detections of that frame, modified IN PLACE, sequentially, like the reference."""
    for i, tr in enumerate(tracks):
        bbox = np.vstack(tr["bbox"])
        score = np.stack(tr["score"])
        best_pose = np.reshape(veh_to_global[tr["token"][int(np.argmax(score))]], [4, 4])
        for j, t in enumerate(tr["token"]):
            pose = np.linalg.inv(np.reshape(veh_to_global[t], [4, 4]))
            bbox[j] = transform_box(bbox[[j], ...], pose).squeeze()
            final = transform_box(transform_box(final_bboxes[[i], :], best_pose), pose)
            if not has_gt[t]:
                continue
            _overwrite_first_match(det_boxes[t], bbox[j, :3], final)


def dynamic_writeback(tracks, veh_to_global, has_gt, final_bboxes, det_boxes):
    """dynamic_eval.py:53-64,121-129: one refined box per track-frame, already in that frame's vehicle frame."""
    index = 0
    for tr in tracks:
        bbox = np.vstack(tr["bbox"])
        for j, t in enumerate(tr["token"]):
            pose = np.linalg.inv(np.reshape(veh_to_global[t], [4, 4]))
            bbox[j] = transform_box(bbox[[j], ...], pose).squeeze()
            if has_gt[t]:
                _overwrite_first_match(det_boxes[t], bbox[j, :3], final_bboxes[[index + j], :])
        index += bbox.shape[0]


def _overwrite_first_match(det, centre, final):
    for k, arr in enumerate(det):
        if np.linalg.norm(arr[:3] - centre) < 0.1:
            det[k, :] = final.squeeze()
            return
    raise AssertionError("Bounding box not in det_annos.")
