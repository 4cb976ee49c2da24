"""One segment from its sweeps to the rewritten detections, chained on the device (round 5; SURVEY.md 8(f) N2 -> N1 ->
8(a) heads -> N3 in ONE stream, no host round trip in between):

    sweeps (resident, one flat tensor)
      -> crops.CropPlan.run           count, starts, fill: every tracked detection's points, global frame, TRACK-major
                                      (waymo_common.py:139-203)
      -> dal3_static_crop_prep        one launch per batch, reading the tracks' rows IN PLACE from the crop output
         dal3_dynamic_item_prep       (static_model.py:529-572, dynamic_model.py:419-509)
      -> model.refine                 the heads + decode (static_model.py:117-146 / dynamic_model.py:121-155 + *_eval)
      -> post.WritebackPlan.launch    the refined boxes into the frames' detection arrays (static_eval.py:62-167,
                                      dynamic_eval.py:43-141)

What is host work is what does not depend on a point: the tracker's association (which detection of which frame
belongs to which track — the reference's tracker, out of scope, hands it over as lists), the detections' face equations,
the poses' inverses, each track's best-score frame and its box, the (track, frame) pairs of the write-back. All of it is
done ONCE per segment in `SegmentPlan.__init__` (before the sweeps are touched: off the critical path) and uploaded;
`run()` only enqueues kernels and returns device tensors. The one size that depends on the points — the number of rows
the crops hold — is bounded by a capacity (first run: read back once; later runs: checked after the fact, `overflowed()`).

Several GPUs: under an initialised process group `run()` shards the HEADS' items — static tracks and
dynamic track-frames, contiguous index ranges per rank, the device sampler keyed on the global item index — and closes
each head with ONE all-gather of the refined (n,7) boxes (dist.all_gather_boxes, RCCL); crop extraction (0.7 ms for a
segment) and the write-back run replicated on every rank, so every rank ends with the same detection arrays, bit for
bit the single-rank ones (tests/test_gpu_pipeline.py: two ranks sharing the test box's GPU over gloo).

The reference's file formats (pickles of SURVEY.md 8(g)) stay with eval.py; this module is the device-resident core.
"""
import numpy as np
import torch

from . import _hip, arch, crops, post, prep
from . import dist as sharding


class SegmentPlan:
    """tracks: list of dicts {"kind": "static" | "dynamic", "dets": [(frame, k), ...] in frame order, "score": [...]}
    — detection k of frame `frame` (as numbered in `detections[frame]`) belongs to the track. Every detection may
    belong to at most one track; untracked detections are extracted too (the reference does) and left alone.
    detections: per frame (K_f, 7|9) float32 detector boxes; veh_to_global: per frame flat-16 poses; n_pts: points per
    sweep. static_model / dynamic_model: the eval-mode product modules (either may be None when no track is of that
    kind)."""

    def __init__(self, n_pts, detections, veh_to_global, tracks, static_model=None, dynamic_model=None, device="cuda",
                 n_static_points=4096, n_per_frame=1024,
                 dynamic_batch=1024, seed=10922081, capacity=None):
        dev = torch.device(device)
        F = len(n_pts)
        n_box = [int(np.asarray(d).shape[0]) for d in detections]
        boff = np.concatenate([[0], np.cumsum(n_box)]).astype(np.int64)
        K = int(boff[-1])
        self.static_tracks = [t for t in tracks if t["kind"] == "static"]
        self.dynamic_tracks = [t for t in tracks if t["kind"] == "dynamic"]
        if (self.static_tracks and static_model is None) or (self.dynamic_tracks and dynamic_model is None):
            raise ValueError("SegmentPlan: a model for every kind of track present")
        self.static_model, self.dynamic_model = static_model, dynamic_model
        self.n_static_points, self.n_per_frame, self.dynamic_batch, self.seed = n_static_points, n_per_frame, dynamic_batch, seed
        # ---- output order of the crops: static tracks, then dynamic tracks, each track's detections in frame order, then
        # whatever no track claims
        order, taken = [], np.zeros(K, bool)
        s_first, d_frame_first = [0], [0]
        for t in self.static_tracks + self.dynamic_tracks:
            ids = [int(boff[f] + k) for f, k in t["dets"]]
            if any(taken[i] for i in ids):
                raise ValueError("SegmentPlan: a detection belongs to two tracks")
            taken[ids] = True
            order += ids
            if t["kind"] == "static":
                s_first.append(len(order))
            else:
                d_frame_first.append(d_frame_first[-1] + len(ids))
        n_s_pos = s_first[-1]
        order += [i for i in range(K) if not taken[i]]
        self.crop = crops.CropPlan(n_pts, detections, veh_to_global, dev, order=np.asarray(order, np.int64), capacity=capacity)
        boxes_lidar, poses = self.crop.boxes, self.crop.poses                      # Waymo convention, per frame; 4x4
        gbox = [crops.transform_box(boxes_lidar[f], poses[f]) for f in range(F)]    # trackData 'bbox': global frame
        tokens = [f"frame{f:04d}" for f in range(F)]
        v2g = {tokens[f]: poses[f].reshape(16) for f in range(F)}
        dets = {tokens[f]: boxes_lidar[f] for f in range(F)}

        def as_track(t):                                                           # the reference's track schema (8(g))
            return {"token": [tokens[f] for f, _ in t["dets"]], "score": list(t["score"]),
                    "bbox": [gbox[f][k] for f, k in t["dets"]]}
        # ---- static tracks: rows of track i = out[offsets[s_first[i]] : offsets[s_first[i+1]]]
        self.S = len(self.static_tracks)
        if self.S:
            self.d_s_pos = torch.from_numpy(np.asarray(s_first, np.int64)).to(dev)
            best = [int(np.argmax(np.stack(t["score"]))) for t in self.static_tracks]
            pose = np.linalg.inv(np.stack([poses[t["dets"][b][0]] for t, b in zip(self.static_tracks, best)]))
            best_box = np.stack([gbox[t["dets"][b][0]][t["dets"][b][1]] for t, b in zip(self.static_tracks, best)])
            self.d_s_pose = torch.from_numpy(np.ascontiguousarray(pose.reshape(self.S, 16))).to(dev)
            self.d_s_box = torch.from_numpy(np.ascontiguousarray(prep._transform_boxes(best_box, pose))).to(dev)
            st = [as_track(t) for t in self.static_tracks]
            self.wb_static = post.WritebackPlan(st, v2g, {(i, tok): True for i, t in enumerate(st) for tok in t["token"]},
                                                dets, True, dev)
            self.s_pts = torch.empty((self.S, n_static_points, 3), dtype=torch.float32, device=dev)
            self.s_init = torch.empty((self.S, 7), dtype=torch.float32, device=dev)
        # ---- dynamic tracks: one item per (track, frame); frame i of the flattened frame list = output position n_s_pos + i
        self.D = d_frame_first[-1]
        if self.D:
            self.d_pos0 = n_s_pos
            it, fr, ipose, dbox = [], [], [], []
            for ti, t in enumerate(self.dynamic_tracks):
                for j, (f, k) in enumerate(t["dets"]):
                    it.append(ti)
                    fr.append(j)
                    ipose.append(np.linalg.inv(poses[f]).reshape(16))
                    dbox.append(gbox[f][k])
            self.d_it = torch.tensor(it, dtype=torch.int32, device=dev)
            self.d_if = torch.tensor(fr, dtype=torch.int32, device=dev)
            self.d_ipose = torch.from_numpy(np.stack(ipose)).to(dev)
            self.d_dbox = torch.from_numpy(np.stack(dbox).astype(np.float64)).to(dev)
            self.d_tfirst = torch.from_numpy(np.asarray(d_frame_first, np.int64)).to(dev)
            dt = [as_track(t) for t in self.dynamic_tracks]
            self.wb_dynamic = post.WritebackPlan(dt, v2g, {(i, tok): True for i, t in enumerate(dt) for tok in t["token"]},
                                                 dets, False, dev)
            Bd = min(dynamic_batch, self.D)
            n = arch.NUM_FRAME * n_per_frame
            self.d_pts = torch.empty((Bd, n, 4), dtype=torch.float32, device=dev)
            self.d_box = torch.empty((Bd, 101, 8), dtype=torch.float32, device=dev)
            self.d_init = torch.empty((Bd, 8), dtype=torch.float32, device=dev)
            self.d_final = torch.empty((self.D, 7), dtype=torch.float32, device=dev)
        self.dev, self.tokens = dev, tokens

    def run(self, d_pts, marks=None, group=None, shard=True):
        """Enqueue the whole chain on the current stream. d_pts: the segment's sweeps, one (sum P_f, 3) float32 CUDA
        tensor. Returns {"static": (det rows (n_det,7) fp32, match (P,) i32), "dynamic": (...)} — device tensors, valid
        once the stream has run; nothing here waits for the GPU (except the very first run of a plan without a
        capacity, which reads the crops' size back once). marks: optional callable(name) invoked between the stages
        (the bench records an event there).
        A run whose crops outgrew the buffer (`overflowed()`, one host read AFTER the run) is not a result: the last
        detections' crops were cut short or empty (their offsets are capped at the capacity, so nothing is read past
        the buffer — dal3_crop_starts_capped), their tracks' boxes and the rows written back for them are wrong;
        `grow()` and run again before using any of it."""
        lib, st = _hip.lib(), _hip.stream
        mark = marks or (lambda name: None)
        out, offsets = self.crop.run(d_pts)
        mark("crops")
        res = {}
        rank, world = 0, 1
        if shard and torch.distributed.is_available() and torch.distributed.is_initialized():       # (shard=False: this rank does it all)
            rank, world = torch.distributed.get_rank(group), torch.distributed.get_world_size(group)
        if self.S:
            s_off = offsets.index_select(0, self.d_s_pos)                          # (S+1) row offsets of the static tracks
            lo, hi = sharding.shard_range(self.S, rank, world)
            n = hi - lo
            if n > 0:
                _hip.check(lib.dal3_static_crop_prep(_hip.ptr(out), _hip.ptr(s_off[lo:hi + 1]), None, _hip.ptr(self.d_s_pose[lo:hi]),
                                                     _hip.ptr(self.d_s_box[lo:hi]), n, self.n_static_points, self.seed, lo,
                                                     _hip.ptr(self.s_pts), _hip.ptr(self.s_init), st()))
                mark("static_prep")
                self.static_model.item_offset = lo
                boxes = self.static_model.refine(self.s_pts[:n].transpose(2, 1), self.s_init[:n])
                self.static_model.item_offset = 0
            else:
                boxes = torch.zeros((0, 7), dtype=torch.float32, device=self.dev)
            if world > 1:
                boxes = sharding.all_gather_boxes(boxes, self.S, group)
            mark("static_heads")
            self.wb_static.launch(boxes)
            mark("static_writeback")
            res["static"] = (self.wb_static.d_det, self.wb_static.match)
        if self.D:
            f_off = offsets[self.d_pos0:self.d_pos0 + self.D + 1]                  # a view: the dynamic frames' row offsets
            Bd = self.d_pts.shape[0]
            lo, hi = sharding.shard_range(self.D, rank, world)
            for b0 in range(lo, hi, Bd):
                n = min(Bd, hi - b0)
                _hip.check(lib.dal3_dynamic_item_prep(_hip.ptr(out), _hip.ptr(f_off), _hip.ptr(self.d_dbox), _hip.ptr(self.d_tfirst),
                                                      _hip.ptr(self.d_it[b0:b0 + n]), _hip.ptr(self.d_if[b0:b0 + n]), None,
                                                      _hip.ptr(self.d_ipose[b0:b0 + n]), n, self.n_per_frame, 2, 50, self.seed, b0,
                                                      _hip.ptr(self.d_pts), _hip.ptr(self.d_box), _hip.ptr(self.d_init), st()))
                self.dynamic_model.item_offset = b0
                self.d_final[b0:b0 + n] = self.dynamic_model.refine(self.d_pts[:n].transpose(2, 1), self.d_box[:n].transpose(2, 1),
                                                                     self.d_init[:n])
            self.dynamic_model.item_offset = 0
            final = self.d_final if world == 1 else sharding.all_gather_boxes(self.d_final[lo:hi], self.D, group)
            mark("dynamic_prep_heads")
            self.wb_dynamic.launch(final)
            mark("dynamic_writeback")
            res["dynamic"] = (self.wb_dynamic.d_det, self.wb_dynamic.match)
        return res

    def overflowed(self):
        """did the last run's crops exceed the output capacity (host sync)? Then `grow()` and run again."""
        return self.crop.total() > self.crop.capacity

    def grow(self):
        self.crop.capacity = int(self.crop.total() * 1.25) + 64
        self.crop.out = torch.empty((self.crop.capacity, 3), dtype=torch.float64, device=self.dev)

    def detections(self, which):
        """{token: (n,7) fp32} of the last run (a download): what eval.py would pickle"""
        wb = self.wb_static if which == "static" else self.wb_dynamic
        out = wb.d_det.cpu().numpy()
        return {t: out[wb.start[t]:wb.start[t] + wb.lens[t]] for t in wb.tokens}
