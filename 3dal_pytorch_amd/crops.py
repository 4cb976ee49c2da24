"""Crop extraction from full sweeps on the device: the track-data part of
`_create_pd_detection` (det3d/datasets/waymo/waymo_common.py:67-218) — for every tracked detection of every frame,
the sweep's points inside its rotated box, moved to the global frame — for a whole batch of frames in three
launches (count, scan, fill) instead of one Python iteration per detection.

In: per frame, the sweep `points_xyz` (P,3) float32 (lidar pickle), the detector's boxes
`box3d_lidar` (K,7|9) float32 and the frame's flat-16 `veh_to_global`. Out: the `trackData` fields the downstream
tracker files hold: 'bbox' (7,) global frame, 'point' (k,3) float64 global frame, plus `boxes_lidar` (the det_annos
rows). Proto serialisation, uuid assignment and the IoU match against ground truth stay with the reference.
"""
import numpy as np
import torch

from . import _hip, geom


def waymo_boxes(box3d_lidar):
    """detector boxes [x,y,z,w,l,h,(vx,vy,)r2] -> Waymo [x,y,z,l,w,h,r1 = -r2 - pi/2] (waymo_common.py:105-111)"""
    b = np.array(box3d_lidar, copy=True)
    b[:, -1] = -b[:, -1] - np.pi / 2
    return b[:, [0, 1, 2, 4, 3, 5, -1]]


def transform_box(box, pose):
    """waymo_common.py:52-65 for (K,7) boxes and one 4x4 pose; O(K) host work"""
    heading = box[..., -1] + np.arctan2(pose[1, 0], pose[0, 0])
    center = np.einsum("...ij,...nj->...ni", pose[0:3, 0:3], box[..., 0:3]) + np.expand_dims(pose[0:3, 3], axis=-2)
    return np.concatenate([center, box[..., 3:6], heading[..., np.newaxis]], axis=-1)


class RaggedRows:
    """The rows [start[k], start[k+1]) of one flat device tensor, for k in [k0, k1): behaves like the list of per-
    detection arrays the reference builds (len, indexing, iteration, slicing), but a view object is made only when
    an element is asked for. Materialising the 11,520 views of one segment eagerly (torch.tensor_split) cost 64 ms
    of host time — ten times the three kernels that computed them."""

    def __init__(self, flat, start, k0, k1):
        self.flat, self.start, self.k0, self.k1 = flat, start, k0, k1

    def __len__(self):
        return self.k1 - self.k0

    def __getitem__(self, i):
        if isinstance(i, slice):
            lo, hi, step = i.indices(len(self))
            if step != 1:
                return [self[j] for j in range(lo, hi, step)]
            return RaggedRows(self.flat, self.start, self.k0 + lo, self.k0 + max(lo, hi))
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        k = self.k0 + i
        return self.flat[int(self.start[k]):int(self.start[k + 1])]

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def counts(self):
        """rows per element, as a NumPy array (no device access)"""
        return np.diff(self.start[self.k0:self.k1 + 1])

    def numpy_list(self):
        """what the trackData pickles hold: a Python list of NumPy arrays (one device->host copy for all of them)"""
        lo, hi = int(self.start[self.k0]), int(self.start[self.k1])
        host = self.flat[lo:hi].cpu().numpy()
        cuts = (self.start[self.k0 + 1:self.k1] - lo).astype(np.int64)
        return np.split(host, cuts) if len(self) else []


class CropPlan:
    """Everything about a segment's crop extraction that does not depend on the sweeps' points, prepared and uploaded
    ONCE (round 5; as post.WritebackPlan and prep.StaticTrackStore do for their steps): the detections in Waymo
    convention, their face equations and cull balls (the O(#boxes) NumPy arithmetic the reference does per detection,
    geom.py), the poses, the frame offsets, the workspace and the output order. `run(points)` then is three kernel
    launches on the current stream — count, starts (dal3_crop_starts: the prefix sums stay on the device), fill — with
    no host work and no synchronisation in between: the detections of a segment are known before its sweeps are
    touched, so the plan is built off the critical path.

        plan = CropPlan(n_pts, detections, veh_to_global, order=track_major)     # once per segment
        out, offsets = plan.run(flat_points)                                      # enqueue only
        total = plan.total()                                                      # (first host sync, when needed)

    n_pts: points per frame; order: optional permutation of the K_total detections (numbered frame by frame) giving
    their order in the output — e.g. track-major, so that a track's rows are contiguous and prep.py's kernels can read
    them in place; capacity: rows of the output buffer (default: sized by the first run, which then synchronises once;
    rows past the capacity are dropped by the kernel, the returned offsets are capped at it, and `total()` tells)."""

    def __init__(self, n_pts, detections, veh_to_global, device="cuda", order=None, capacity=None, return_index=False):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("extract_crops runs on the GPU only (lib3dal_hip.so has no CPU fallback)")
        F = len(n_pts)
        if not (F == len(detections) == len(veh_to_global)) or F == 0:
            raise ValueError("extract_crops: need the same non-zero number of sweeps, detection sets and poses")
        self.dev, self.F = dev, F
        self.boxes = [waymo_boxes(np.asarray(d, dtype=np.float32).reshape(-1, np.asarray(d).shape[-1])) for d in detections]
        self.poses = [np.reshape(np.asarray(p, dtype=np.float64), [4, 4]) for p in veh_to_global]
        all_boxes = np.concatenate(self.boxes) if self.boxes else np.zeros((0, 7), np.float32)
        self.n_pts = [int(n) for n in n_pts]
        self.n_box = [int(b.shape[0]) for b in self.boxes]
        self.K, self.max_pts = sum(self.n_box), max(self.n_pts)
        K = self.K
        self.d_poff = torch.from_numpy(np.concatenate([[0], np.cumsum(self.n_pts)]).astype(np.int64)).to(dev)
        self.d_boff = torch.from_numpy(np.concatenate([[0], np.cumsum(self.n_box)]).astype(np.int64)).to(dev)
        self.d_planes = geom.planes_to_device(geom.box_planes(all_boxes), dev)       # per-box arithmetic: one call for every frame
        self.d_sph = torch.from_numpy(geom.cull_spheres(all_boxes)).to(dev)
        self.d_pose = torch.from_numpy(np.stack(self.poses).reshape(F, 16)).to(dev)
        self.d_order = None
        if order is not None:
            order = np.asarray(order, np.int64)
            if order.shape != (K,) or not np.array_equal(np.sort(order), np.arange(K)):
                raise ValueError("CropPlan: order must be a permutation of the detections")
            self.d_order = torch.from_numpy(order).to(dev)
        lib = _hip.lib()
        self.ws = torch.empty(max(int(lib.dal3_crop_workspace_bytes(K, self.max_pts)), 4), dtype=torch.uint8, device=dev)
        self.counts = torch.zeros(max(K, 1), dtype=torch.int64, device=dev)
        self.start = torch.zeros(K + 1, dtype=torch.int64, device=dev)           # by detection; [K] = total
        self.offsets = torch.zeros(K + 1, dtype=torch.int64, device=dev)         # by output position
        self.capacity = int(capacity) if capacity else 0
        self.out = torch.empty((self.capacity, 3), dtype=torch.float64, device=dev) if self.capacity else None
        self.idx = None
        self.return_index = return_index

    def count(self, d_pts):
        lib = _hip.lib()
        _hip.check(lib.dal3_crop_count(_hip.ptr(d_pts), _hip.ptr(self.d_poff), _hip.ptr(self.d_planes), _hip.ptr(self.d_sph),
                                       _hip.ptr(self.d_boff), self.F, self.K, self.max_pts, _hip.ptr(self.counts),
                                       _hip.ptr(self.ws), self.ws.numel(), _hip.stream()))
        # with a buffer of fixed capacity the offsets handed on to the consumers are capped at it (the fill drops the rows
        # past it): an overflowing run yields short or empty crops for the last detections, never a read past the buffer
        # (ADVICE r5); self.start keeps the true prefix sums — total() / SegmentPlan.overflowed() tell
        if self.capacity:
            _hip.check(lib.dal3_crop_starts_capped(_hip.ptr(self.counts), _hip.ptr(self.d_order), self.K, _hip.ptr(self.start),
                                                   _hip.ptr(self.offsets), self.capacity, _hip.stream()))
        else:
            _hip.check(lib.dal3_crop_starts(_hip.ptr(self.counts), _hip.ptr(self.d_order), self.K, _hip.ptr(self.start),
                                            _hip.ptr(self.offsets), _hip.stream()))

    def fill(self, d_pts):
        if self.return_index and (self.idx is None or self.idx.numel() < self.capacity):
            self.idx = torch.empty(max(self.capacity, 1), dtype=torch.int32, device=self.dev)
        _hip.check(_hip.lib().dal3_crop_fill(_hip.ptr(d_pts), _hip.ptr(self.d_poff), _hip.ptr(self.d_planes), _hip.ptr(self.d_sph),
                                             _hip.ptr(self.d_boff), self.F, self.K, self.max_pts, _hip.ptr(self.d_pose),
                                             _hip.ptr(self.counts), _hip.ptr(self.start), _hip.ptr(self.out),
                                             _hip.ptr(self.idx) if self.return_index else None, self.capacity,
                                             _hip.ptr(self.ws), self.ws.numel(), _hip.stream()))

    def total(self):
        """members over all detections of the last run (a host synchronisation)"""
        return int(self.start[self.K].item())

    def run(self, d_pts):
        """d_pts: the segment's sweeps as ONE (sum P_f, 3) float32 CUDA tensor, frame after frame. Enqueues count,
        starts and fill; returns (out (capacity,3) float64 global-frame rows, offsets (K+1) int64 by output position),
        both device tensors that the NEXT run overwrites. Without a capacity (first run) the total is read back once
        to size the buffer (+12 % headroom for later runs of the same plan)."""
        if d_pts.dtype != torch.float32 or d_pts.dim() != 2 or d_pts.shape[1] != 3 or not d_pts.is_contiguous() \
                or d_pts.shape[0] != sum(self.n_pts):
            raise ValueError("CropPlan.run: points must be the contiguous (sum of n_pts, 3) float32 tensor of the segment")
        self.count(d_pts)
        if not self.capacity:
            self.capacity = max(int(self.total() * 1.125) + 64, 64)
            self.out = torch.empty((self.capacity, 3), dtype=torch.float64, device=self.dev)
        self.fill(d_pts)
        return self.out, self.offsets


def extract_crops(sweeps, detections, veh_to_global, device="cuda", return_index=False):
    """sweeps: list of (P_f,3) float32 arrays or CUDA tensors; detections: list of (K_f,7|9) float32 detector
    boxes; veh_to_global: list of flat-16 poses. Returns a list (one entry per frame) of dicts
      'boxes_lidar' (K_f,7) float32 NumPy — Waymo-convention boxes, vehicle frame (det_annos rows)
      'bbox'        (K_f,7) float64 NumPy — the same boxes in the global frame (trackData 'bbox')
      'point'       K_f CUDA float64 tensors (k,3), global frame, in sweep order (trackData 'point'): a RaggedRows
                    sequence over ONE flat tensor (index / iterate / slice it like a list; .numpy_list() for pickles)
      'index'       (return_index) likewise K_f CUDA int32 tensors: which sweep points they are
    """
    dev = torch.device(device)
    n_pts = [int(s.shape[0]) for s in sweeps]
    plan = CropPlan(n_pts, detections, veh_to_global, dev, return_index=return_index)
    d_pts = torch.cat([(s if torch.is_tensor(s) else torch.from_numpy(np.ascontiguousarray(s, dtype=np.float32))).to(dev)
                       .to(torch.float32).reshape(-1, 3) for s in sweeps]).contiguous()
    plan.count(d_pts)
    h_start = plan.start.cpu().numpy()                              # the one host sync: sizes of the ragged outputs
    plan.capacity = max(int(h_start[-1]), 1)
    plan.out = torch.empty((plan.capacity, 3), dtype=torch.float64, device=dev)
    plan.fill(d_pts)
    out, idx, boxes, poses, n_box, F = plan.out, plan.idx, plan.boxes, plan.poses, plan.n_box, plan.F
    frames, k = [], 0
    for f in range(F):
        rec = {"boxes_lidar": boxes[f], "bbox": transform_box(boxes[f], poses[f]),
               "point": RaggedRows(out, h_start, k, k + n_box[f])}
        if return_index:
            rec["index"] = RaggedRows(idx, h_start, k, k + n_box[f])
        frames.append(rec)
        k += n_box[f]
    return frames
