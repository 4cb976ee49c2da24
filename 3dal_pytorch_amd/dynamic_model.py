"""Drop-in for DynamicModel of the reference's tools/dynamic_model.py (:109-155), on MI355X.

Same constructor, `.r`, `.s`, `forward(pts, box, bbox_gt) -> dict` and state_dict key set. Eval
mode runs in lib3dal_hip.so through dal3_dynamic_forward; train mode runs a stock-torch
composite (not accelerated). `.refine(pts, box, init_box)` additionally returns the (B,7)
refined boxes of dynamic_eval.py:226-242 without leaving the GPU.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from . import _hip, arch
from .datasets import DYNAMICTRACK                                  # noqa: F401  (the drivers import it from here)
from .losses import DynamicModelLoss, huber_loss                    # noqa: F401
from ._heads import (BoxEmbedding, DynamicPointNetEstimation as PointNetEstimation, PackedCache, PackedModelMixin,
                     PointEmbedding, PointNetInstanceSeg, Workspace, as_f32, as_points, dtype_of, numpy_choice, rows_contiguous)
from . import train as _train
from .static_model import _box_pred, _mask_and_gather, _parse, _seg_logits, _train_arith

NUM_HEADING_BIN = arch.NUM_HEADING_BIN
NUM_SIZE_CLUSTER = arch.NUM_SIZE_CLUSTER
NUM_OBJECT_POINT = arch.NUM_OBJECT_POINT
NUM_POINT = 1024                                   # dynamic_model.py:15
NUM_FRAME = arch.NUM_FRAME
MEAN_SIZE_ARR = np.array(arch.MEAN_SIZE)
_M = NUM_FRAME * NUM_OBJECT_POINT


class DynamicModel(PackedModelMixin, nn.Module):
    def __init__(self, n_classes=3, n_channel=4):
        super().__init__()
        if n_channel != 4:
            raise ValueError("the dynamic head takes xyz+time points (n_channel=4, dynamic_eval.py:292)")
        self.r = 2
        self.s = 50
        self.n_classes = n_classes
        self.n_channel = n_channel
        self.ins_seg = PointNetInstanceSeg(n_classes=n_classes, n_channel=n_channel)
        self.point_emb = PointEmbedding(n_classes=n_classes)
        self.box_emb = BoxEmbedding(n_classes=n_classes)
        self.box_est = PointNetEstimation(n_classes=n_classes)
        self.sampler = "device"
        self.train_backend = "hip"                   # train-mode per-point stacks: "hip" (train.py) or "torch"
        self.precision = "fp32"                      # "bf16" / "fp16": 16-bit MFMA operands (configs C3/C5)
        self.seed = 10922081
        self.item_offset = 0
        self._cache = PackedCache()
        self._ws = Workspace()
        self.last = {}

    def _run(self, pts, box, init_box8=None, choice=None, mask_override=None):
        lib = _hip.lib()
        pts = as_points(pts, "pts")
        box = as_points(box, "box")
        if init_box8 is not None:
            init_box8 = rows_contiguous(as_f32(init_box8, "init_box"))
        if pts.dim() != 3 or pts.shape[1] != 4:
            raise RuntimeError(f"pts must be (B,4,N), got {tuple(pts.shape)}")
        B, _, N = pts.shape
        if box.dim() != 3 or box.shape[0] != B or box.shape[1] != 8:
            raise RuntimeError(f"box must be (B,8,n_box), got {tuple(box.shape)}")
        n_box = box.shape[2]
        dev = pts.device
        f32 = dict(dtype=torch.float32, device=dev)
        o = {
            "logits": torch.empty((B, N, 2), **f32),
            "mask": torch.empty((B, N), dtype=torch.uint8, device=dev),
            "embedding": torch.empty((B, 384), **f32), "bp": torch.empty((B, 39), **f32),
            "hr": torch.empty((B, 12), **f32), "sr": torch.empty((B, 3, 3), **f32),
            "boxes7": torch.empty((B, 7), **f32),
            "counts": torch.empty((B,), dtype=torch.int32, device=dev),
            "obj_idx": torch.empty((B, _M), dtype=torch.int32, device=dev),
        }
        ws = self._ws.get(lib.dal3_dynamic_workspace_bytes(B, N, n_box), dev)
        a = _hip.DynamicArgs()
        a.B, a.N, a.n_box = B, N, n_box
        a.seed, a.item_offset = self.seed, self.item_offset
        dt = a.dtype = dtype_of(self.precision)
        a.pts, a.box = _hip.bcn(pts), _hip.bcn(box)
        a.init_box8 = _hip.ptr(init_box8)
        a.w_ins_seg = _hip.ptr(self._cache.get("ins_seg", self.ins_seg, _hip.HEAD_INS_SEG, dt))
        a.w_point_emb = _hip.ptr(self._cache.get("pe", self.point_emb, _hip.HEAD_POINT_EMB, dt))
        a.w_box_emb = _hip.ptr(self._cache.get("be", self.box_emb, _hip.HEAD_BOX_EMB, dt))
        a.w_box_est = _hip.ptr(self._cache.get("est", self.box_est, _hip.HEAD_DYNAMIC_BOX_EST, dt))
        a.logits, a.mask = _hip.ptr(o["logits"]), _hip.ptr(o["mask"])
        a.embedding, a.box_pred = _hip.ptr(o["embedding"]), _hip.ptr(o["bp"])
        a.heading_residuals, a.size_residuals = _hip.ptr(o["hr"]), _hip.ptr(o["sr"])
        a.boxes7, a.counts, a.obj_idx = _hip.ptr(o["boxes7"]), _hip.ptr(o["counts"]), _hip.ptr(o["obj_idx"])
        a.workspace, a.workspace_bytes = _hip.ptr(ws), ws.numel()
        st = _hip.stream()
        if choice is None and self.sampler == "device" and mask_override is None:
            a.sampler = _hip.SAMPLER_DEVICE
            _hip.check(lib.dal3_dynamic_forward(C.byref(a), _hip.PHASE_ALL, st))
        else:
            _hip.check(lib.dal3_dynamic_forward(C.byref(a), _hip.PHASE_SEG, st))
            if mask_override is not None:
                o["mask"].copy_(mask_override.to(device=dev, dtype=torch.uint8))
                _hip.check(lib.dal3_segment_counts(_hip.ptr(o["mask"]), B, N, _hip.ptr(o["counts"]), st))
            if choice is None and self.sampler == "numpy":
                choice = torch.from_numpy(numpy_choice(o["counts"].cpu().numpy(), _M))
            elif choice is None and self.sampler != "device":
                raise ValueError(f"unknown sampler {self.sampler!r}")
            if choice is not None:
                choice = choice.to(device=dev, dtype=torch.int32).contiguous()
                a.sampler, a.choice = _hip.SAMPLER_CHOICE, _hip.ptr(choice)
            else:
                a.sampler = _hip.SAMPLER_DEVICE
            _hip.check(lib.dal3_dynamic_forward(C.byref(a), _hip.PHASE_BOX, st))
        o["_keep"] = (pts, box, init_box8, choice, ws)
        return o

    def forward(self, pts, box, bbox_gt):
        if self.training:
            with _train.arithmetic(_train_arith(self)):
                return _train_forward(self, pts, box)
        o = self._run(pts, box)
        bp = o["bp"]
        B = bp.shape[0]
        self.last = {k: o[k] for k in ("counts", "obj_idx", "embedding")}
        return {
            "logits": o["logits"], "mask": o["mask"].view(torch.bool), "center": bp[:, 0:3],
            "heading_scores": bp[:, 3:15], "heading_residuals_normalized": bp[:, 15:27],
            "heading_residuals": o["hr"], "size_scores": bp[:, 27:30],
            "size_residuals_normalized": bp[:, 30:39].view(B, 3, 3), "size_residuals": o["sr"],
        }

    def refine(self, pts, box, init_box):
        """(B,7) fp32 refined boxes on the device: forward + dynamic_eval.py:226-242
        (centre += init_box[:, :3], yaw += init_box[:, -2]; init_box is (B,8))."""
        if self.training:
            raise RuntimeError("refine() is the eval-mode path; call model.eval() first")
        return self._run(pts, box, init_box8=init_box)["boxes7"]


def _train_forward(m, pts, box):
    # the stacks (ins_seg, point_emb, box_emb) and the FC tails on the HIP training kernels when train_backend == "hip"
    logits = _seg_logits(m, pts)
    obj, mask = _mask_and_gather(pts, logits, _M, 4, m)
    emb = torch.cat([_box_pred(m, m.point_emb, obj), _box_pred(m, m.box_emb, box if box.dtype == torch.float64 else box.float())], dim=1)
    c, hs, hrn, hr, ss, srn, sr = _parse(_box_pred(m, m.box_est, emb))
    return {"logits": logits, "mask": mask, "center": c, "heading_scores": hs,
            "heading_residuals_normalized": hrn, "heading_residuals": hr, "size_scores": ss,
            "size_residuals_normalized": srn, "size_residuals": sr}
