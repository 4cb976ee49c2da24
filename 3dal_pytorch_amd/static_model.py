"""Drop-in for the model classes of the reference's tools/static_model.py, running on MI355X.

Same constructors, `.name`, `forward(pts, init_box, bbox_gt) -> dict` (same keys, shapes and
dtypes) and state_dict key set as StaticModelOneBoxEst (static_model.py:108-146) and
StaticModelTwoBoxEst (:148-239). In eval mode forward() runs entirely in lib3dal_hip.so through
dal3_static_forward (include/dal3.h); in train mode it runs a stock-torch composite so that
tools/static_train.py keeps working (not accelerated, SURVEY.md 8(a) note T).

Extras the reference does not have:
  .sampler      "device" (default; counter-based RNG on the GPU, no host round trip) or "numpy"
                (the reference's np.random draws in its exact order -> bit-reproducible eval runs)
  .refine(pts, init_box, bbox_gt=None) -> (B,7) refined boxes on the device, i.e. forward +
                the decode loop of static_eval.py:269-288 without leaving the GPU.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from . import _hip, arch
from . import train as _train
from .datasets import STATICTRACK                                   # noqa: F401  (the drivers import it from here)
from .losses import FrustumPointNetLossOneBoxEst, FrustumPointNetLossTwoBoxEst, huber_loss   # noqa: F401
from ._heads import (PackedCache, PackedModelMixin, PointNetInstanceSeg, StaticPointNetEstimation as PointNetEstimation,
                     Workspace, as_f32, as_points, dtype_of, numpy_choice, rows_contiguous)

NUM_HEADING_BIN = arch.NUM_HEADING_BIN
NUM_SIZE_CLUSTER = arch.NUM_SIZE_CLUSTER
NUM_OBJECT_POINT = arch.NUM_OBJECT_POINT
NUM_POINT = 4096                                   # static_model.py:15
MEAN_SIZE_ARR = np.array(arch.MEAN_SIZE)


class _StaticBase(PackedModelMixin, nn.Module):
    two_stage = False

    def __init__(self, n_classes=3, n_channel=3):
        super().__init__()
        if n_channel != 3:
            raise ValueError("the static heads take xyz points (n_channel=3, static_eval.py:345)")
        self.n_classes = n_classes
        self.n_channel = n_channel
        self.sampler = "device"
        self.train_backend = "hip"                   # train-mode per-point stacks: "hip" (train.py) or "torch"
        self.precision = "fp32"                      # "bf16" / "fp16": 16-bit MFMA operands (configs C3/C5)
        self.seed = 10922081
        self.item_offset = 0
        self._cache = PackedCache()
        self._ws = Workspace()
        self.last = {}                              # side outputs of the last eval forward (boxes7, counts, obj_idx)

    # ------------------------------------------------------------------ HIP path
    def _run(self, pts, init_box, bbox_gt, choice=None, mask_override=None):
        lib = _hip.lib()
        pts = as_points(pts, "pts")
        init_box = rows_contiguous(as_f32(init_box, "init_box"))
        if bbox_gt is not None:
            bbox_gt = rows_contiguous(as_f32(bbox_gt, "bbox_gt"))
        if pts.dim() != 3 or pts.shape[1] != 3:
            raise RuntimeError(f"pts must be (B,3,N), got {tuple(pts.shape)}")
        B, _, N = pts.shape
        if init_box.shape != (B, 7):
            raise RuntimeError(f"init_box must be (B,7), got {tuple(init_box.shape)}")
        dev = pts.device
        two = self.two_stage
        f32 = dict(dtype=torch.float32, device=dev)
        o = {
            "logits": torch.empty((B, N, 2), **f32),
            "mask": torch.empty((B, N), dtype=torch.uint8, device=dev),
            "bp1": torch.empty((B, 39), **f32), "hr1": torch.empty((B, 12), **f32),
            "sr1": torch.empty((B, 3, 3), **f32), "c1": torch.empty((B, 3), **f32),
            "boxes7": torch.empty((B, 7), **f32),
            "counts": torch.empty((B,), dtype=torch.int32, device=dev),
            "obj_idx": torch.empty((B, NUM_OBJECT_POINT), dtype=torch.int32, device=dev),
        }
        if two:
            o.update({
                "box_one": torch.empty((B, 7), **f32), "bp2": torch.empty((B, 39), **f32),
                "hr2": torch.empty((B, 12), **f32), "sr2": torch.empty((B, 3, 3), **f32),
                "c2": torch.empty((B, 3), **f32),
                "hcl": torch.zeros((B,), dtype=torch.int64, device=dev), "hrl": torch.zeros((B,), **f32),
            })
        ws_bytes = lib.dal3_static_workspace_bytes(B, N, int(two))
        ws = self._ws.get(ws_bytes, dev)
        a = _hip.StaticArgs()
        a.B, a.N, a.two_stage = B, N, int(two)
        a.seed, a.item_offset = self.seed, self.item_offset
        dt = a.dtype = dtype_of(self.precision)
        a.pts = _hip.bcn(pts)
        a.init_box, a.bbox_gt = _hip.ptr(init_box), _hip.ptr(bbox_gt)
        a.w_ins_seg = _hip.ptr(self._cache.get("ins_seg", self.ins_seg, _hip.HEAD_INS_SEG, dt))
        if two:
            a.w_box_est_one = _hip.ptr(self._cache.get("one", self.box_est_one, _hip.HEAD_STATIC_BOX_EST, dt))
            a.w_box_est_two = _hip.ptr(self._cache.get("two", self.box_est_two, _hip.HEAD_STATIC_BOX_EST, dt))
        else:
            a.w_box_est_one = _hip.ptr(self._cache.get("one", self.box_est, _hip.HEAD_STATIC_BOX_EST, dt))
        a.logits, a.mask = _hip.ptr(o["logits"]), _hip.ptr(o["mask"])
        a.box_pred_one, a.heading_residuals_one = _hip.ptr(o["bp1"]), _hip.ptr(o["hr1"])
        a.size_residuals_one, a.center_one = _hip.ptr(o["sr1"]), _hip.ptr(o["c1"])
        if two:
            a.box_one, a.box_pred_two = _hip.ptr(o["box_one"]), _hip.ptr(o["bp2"])
            a.heading_residuals_two, a.size_residuals_two = _hip.ptr(o["hr2"]), _hip.ptr(o["sr2"])
            a.center_two = _hip.ptr(o["c2"])
            a.heading_class_label_two, a.heading_residuals_label_two = _hip.ptr(o["hcl"]), _hip.ptr(o["hrl"])
        a.boxes7, a.counts, a.obj_idx = _hip.ptr(o["boxes7"]), _hip.ptr(o["counts"]), _hip.ptr(o["obj_idx"])
        a.workspace, a.workspace_bytes = _hip.ptr(ws), ws.numel()

        st = _hip.stream()
        if choice is None and self.sampler == "device" and mask_override is None:
            a.sampler = _hip.SAMPLER_DEVICE
            _hip.check(lib.dal3_static_forward(C.byref(a), _hip.PHASE_ALL, st))
        else:
            _hip.check(lib.dal3_static_forward(C.byref(a), _hip.PHASE_SEG, st))
            if mask_override is not None:               # teacher forcing (tests)
                o["mask"].copy_(mask_override.to(device=dev, dtype=torch.uint8))
                _hip.check(lib.dal3_segment_counts(_hip.ptr(o["mask"]), B, N, _hip.ptr(o["counts"]), st))
            if choice is None and self.sampler == "numpy":
                choice = torch.from_numpy(numpy_choice(o["counts"].cpu().numpy(), NUM_OBJECT_POINT))
            elif choice is None and self.sampler != "device":
                raise ValueError(f"unknown sampler {self.sampler!r}")
            if choice is not None:
                choice = choice.to(device=dev, dtype=torch.int32).contiguous()
                a.sampler, a.choice = _hip.SAMPLER_CHOICE, _hip.ptr(choice)
            else:
                a.sampler = _hip.SAMPLER_DEVICE
            _hip.check(lib.dal3_static_forward(C.byref(a), _hip.PHASE_BOX, st))
        o["_keep"] = (pts, init_box, bbox_gt, choice, ws)   # keep inputs alive until the stream catches up
        return o

    def refine(self, pts, init_box, bbox_gt=None):
        """(B,7) fp32 refined boxes [cx,cy,cz,l,w,h,yaw] on the device (forward + static_eval.py:269-288)."""
        if self.training:
            raise RuntimeError("refine() is the eval-mode path; call model.eval() first")
        return self._run(pts, init_box, bbox_gt)["boxes7"]


class StaticModelOneBoxEst(_StaticBase):
    """tools/static_model.py:108-146."""
    two_stage = False

    def __init__(self, n_classes=3, n_channel=3):
        super().__init__(n_classes, n_channel)
        self.name = "one_box_est"
        self.ins_seg = PointNetInstanceSeg(n_classes=n_classes, n_channel=n_channel)
        self.box_est = PointNetEstimation(n_classes=n_classes)

    def forward(self, pts, init_box, bbox_gt):
        if self.training:
            with _train.arithmetic(_train_arith(self)):
                return _train_forward_one(self, pts, init_box)
        o = self._run(pts, init_box, bbox_gt)
        bp = o["bp1"]
        B = bp.shape[0]
        self.last = {k: o[k] for k in ("boxes7", "counts", "obj_idx")}
        return {
            "logits": o["logits"], "mask": o["mask"].view(torch.bool),
            "center_boxnet": bp[:, 0:3], "heading_scores": bp[:, 3:15],
            "heading_residuals_normalized": bp[:, 15:27], "heading_residuals": o["hr1"],
            "size_scores": bp[:, 27:30], "size_residuals_normalized": bp[:, 30:39].view(B, 3, 3),
            "size_residuals": o["sr1"], "center": o["c1"],
        }


class StaticModelTwoBoxEst(_StaticBase):
    """tools/static_model.py:148-239."""
    two_stage = True

    def __init__(self, n_classes=3, n_channel=3):
        super().__init__(n_classes, n_channel)
        self.name = "two_box_est"
        self.ins_seg = PointNetInstanceSeg(n_classes=n_classes, n_channel=n_channel)
        self.box_est_one = PointNetEstimation(n_classes=n_classes)
        self.box_est_two = PointNetEstimation(n_classes=n_classes)

    def forward(self, pts, init_box, bbox_gt):
        if self.training:
            with _train.arithmetic(_train_arith(self)):
                return _train_forward_two(self, pts, init_box, bbox_gt)
        o = self._run(pts, init_box, bbox_gt)
        b1, b2 = o["bp1"], o["bp2"]
        B = b1.shape[0]
        self.last = {k: o[k] for k in ("boxes7", "counts", "obj_idx")}
        return {
            "logits": o["logits"], "mask": o["mask"].view(torch.bool),
            "heading_scores_one": b1[:, 3:15], "heading_residuals_normalized_one": b1[:, 15:27],
            "heading_residuals_one": o["hr1"], "size_scores_one": b1[:, 27:30],
            "size_residuals_normalized_one": b1[:, 30:39].view(B, 3, 3), "size_residuals_one": o["sr1"],
            "center_one": b1[:, 0:3], "box_one": o["box_one"],
            "heading_scores_two": b2[:, 3:15], "heading_residuals_normalized_two": b2[:, 15:27],
            "heading_residuals_two": o["hr2"], "size_scores_two": b2[:, 27:30],
            "size_residuals_normalized_two": b2[:, 30:39].view(B, 3, 3), "size_residuals_two": o["sr2"],
            "center_two": b2[:, 0:3],
            "heading_class_label_two": o["hcl"], "heading_residuals_label_two": o["hrl"],
            "center": b2[:, 0:3], "heading_scores": b2[:, 3:15], "heading_residuals": o["hr2"],
            "size_scores": b2[:, 27:30], "size_residuals": o["sr2"],
        }


# ---------------------------------------------------------------------- train mode (stock torch)
def _mask_and_gather(pts, logits, n_obj, n_ch, model=None):
    """point_cloud_masking + gather_object_pts (static_model.py:23-62). Default: the reference's NumPy draws — index
    bookkeeping on the host (one device->host sync and a Python iteration per sample), the gather itself on the
    tensor's device. With model.sampler == "device" on the GPU: the eval path's compaction + sampling kernel
    (dal3_mask_compact_sample), no host round trip; the sampled points carry no gradient either way."""
    mask = logits[:, :, 0] < logits[:, :, 1]
    if model is not None and getattr(model, "sampler", "numpy") == "device" and pts.is_cuda:
        lib = _hip.lib()
        B, _, N = pts.shape
        p32 = as_f32(pts.detach(), "pts")
        m8 = mask.to(torch.uint8).contiguous()
        counts = torch.empty((B,), dtype=torch.int32, device=pts.device)
        idx = torch.empty((B, n_obj), dtype=torch.int32, device=pts.device)
        obj = torch.empty((B, n_obj, n_ch), dtype=torch.float32, device=pts.device)
        need = lib.dal3_gather_workspace_bytes(B, N)
        ws = torch.empty(max(int(need), 8), dtype=torch.uint8, device=pts.device)
        # a fresh draw per training step, as the reference's np.random gives: the key carries a draw counter that
        # lives in DEVICE memory and is bumped by an ordinary op — captured into a hipGraph (graph.CapturedTrainStep)
        # it still advances on every replay, where a host-computed seed would be frozen into the kernel's arguments
        # (eval keeps the fixed key so that shards of a job reproduce the whole job)
        step = _train.draw_step(model, pts.device)
        step.add_(1)
        _hip.check(lib.dal3_mask_compact_sample_step(_hip.ptr(m8), _hip.bcn(p32), B, N, n_ch, n_obj, model.seed, _hip.ptr(step),
                                                     model.item_offset, _hip.ptr(counts), _hip.ptr(idx), _hip.ptr(obj),
                                                     _hip.ptr(ws), ws.numel(), _hip.stream()))
        return obj.transpose(2, 1), mask
    counts = mask.sum(1).cpu().numpy()
    choice = numpy_choice(counts, n_obj)
    # (float32 in the reference whatever comes in; following a float64 input lets the composite be run in float64
    # against the reference's float64 training step, tests/test_host_dropin_train.py)
    obj = torch.zeros((pts.shape[0], n_ch, n_obj), dtype=pts.dtype if pts.dtype == torch.float64 else torch.float32,
                      device=pts.device)
    for i, k in enumerate(counts):
        if k > 0:
            pos = torch.nonzero(mask[i]).squeeze(1)
            idx = pos[torch.from_numpy(choice[i]).long().to(pos.device)]
            obj[i] = pts[i, :n_ch, idx]
    return obj, mask


_MEAN_SIZE_ON = {}


def _mean_size(device, dtype=torch.float32):
    """MEAN_SIZE_ARR as a device tensor, uploaded once per device (an upload inside a step would also break hipGraph
    capture of the step)"""
    t = _MEAN_SIZE_ON.get((device, dtype))
    if t is None:
        t = _MEAN_SIZE_ON[(device, dtype)] = torch.tensor(arch.MEAN_SIZE, dtype=dtype, device=device)
    return t


class _ParseBoxPred(torch.autograd.Function):
    """parse_output_to_tensors (tools/static_model.py:64-92) as ONE autograd node: the seven outputs are slices / scaled
    slices of the (B,39) box_pred, and its backward puts their gradients back side by side with one concatenation —
    sliced with stock ops, autograd answers every slice with a zero-filled (B,39) tensor, a copy into it and an add
    (sixteen launches per estimate in a training step)."""

    @staticmethod
    def forward(ctx, box_pred, mean):
        B, dev = box_pred.shape[0], box_pred.device
        ctx.meta = (B, box_pred.dtype, dev)
        if box_pred.is_cuda and box_pred.dtype == torch.float32 and box_pred.stride(1) == 1:
            # one launch (dal3_parse_box_pred) for the seven tensors. Copies, not views: an output of a custom Function that
            # is a VIEW of its input makes autograd rebase the view on the Function's node (and aborted the process in a
            # later backward on this build)
            outs = [torch.empty(s_, dtype=torch.float32, device=dev)
                    for s_ in ((B, 3), (B, 12), (B, 12), (B, 12), (B, 3), (B, 3, 3), (B, 3, 3))]
            _hip.check(_hip.lib().dal3_parse_box_pred(_hip.ptr(box_pred), box_pred.stride(0), B, *[_hip.ptr(t) for t in outs],
                                                      _hip.stream()))
            ctx.hip = True
            return tuple(outs)
        ctx.hip = False
        ctx.save_for_backward(mean)
        srn = box_pred[:, 30:39].contiguous().view(B, 3, 3)
        c, hs, hrn, ss = (t.contiguous() for t in torch.split(box_pred, [3, 12, 12, 3, 9], 1)[:4])
        return (c, hs, hrn, hrn * (np.pi / NUM_HEADING_BIN), ss, srn, srn * mean[None])

    @staticmethod
    def backward(ctx, gc, ghs, ghrn, ghr, gss, gsrn, gsr):
        B, dtype, dev = ctx.meta
        if ctx.hip:
            gs = [None if t is None else (t if (t.is_contiguous() and t.dtype == torch.float32) else t.contiguous().float())
                  for t in (gc, ghs, ghrn, ghr, gss, gsrn, gsr)]
            g = torch.empty((B, 39), dtype=torch.float32, device=dev)
            _hip.check(_hip.lib().dal3_parse_box_pred_backward(*[_hip.ptr(t) for t in gs], B, _hip.ptr(g), _hip.stream()))
            return g, None
        (mean,) = ctx.saved_tensors

        def z(n):
            return torch.zeros((B, n), dtype=dtype, device=dev)
        if ghr is not None:
            ghrn = ghr * (np.pi / NUM_HEADING_BIN) if ghrn is None else ghrn + ghr * (np.pi / NUM_HEADING_BIN)
        if gsr is not None:
            gsrn = gsr * mean[None] if gsrn is None else gsrn + gsr * mean[None]
        parts = [gc if gc is not None else z(3), ghs if ghs is not None else z(12), ghrn if ghrn is not None else z(12),
                 gss if gss is not None else z(3), gsrn.reshape(B, 9) if gsrn is not None else z(9)]
        return torch.cat(parts, 1), None


def _parse(box_pred):
    mean = _mean_size(box_pred.device, box_pred.dtype)
    if box_pred.requires_grad:
        return _ParseBoxPred.apply(box_pred, mean)
    B = box_pred.shape[0]
    hrn = box_pred[:, 15:27]
    srn = box_pred[:, 30:39].contiguous().view(B, 3, 3)
    return (box_pred[:, 0:3], box_pred[:, 3:15], hrn, hrn * (np.pi / NUM_HEADING_BIN),
            box_pred[:, 27:30], srn, srn * mean[None])


def _hip_training(m, pts):
    """train_backend "hip" (default): the per-point stacks run on lib3dal_hip.so's training kernels (train.py);
    "torch": the stock composite. The HIP kernels need CUDA tensors; any B and N."""
    backend = getattr(m, "train_backend", "hip")
    if backend not in ("hip", "torch"):
        raise ValueError(f"unknown train_backend {backend!r}")
    return backend == "hip" and _train.supported(pts)


def _train_arith(m):
    """model.precision in train mode: "f16x3" runs the forward's big layers on the f16x3 training kernels (train.ARITH);
    anything else trains in exact fp32 (the 16-bit eval precisions have no training path)"""
    return "f16x3" if getattr(m, "precision", "fp32") == "f16x3" else "fp32"


def _seg_logits(m, pts):
    if _hip_training(m, pts):
        # m.drop_mask: optional (B*N,128) multiplier that replaces the random Dropout draw of this forward (parity
        # tests replay the reference's own draw with it); None = draw on the device
        return _train.ins_seg_train_forward(m.ins_seg, pts.float(), p_drop=m.ins_seg.dropout.p,
                                            drop_mask=getattr(m, "drop_mask", None))
    return m.ins_seg(pts)


def _tail(m, head, x):
    """the per-item FC layers of a head: on the HIP training kernels (rows = items), stock torch ops for the "torch"
    backend or on the CPU"""
    if getattr(m, "train_backend", "hip") == "hip" and _train.fc_tail_supported(x):
        return _train.fc_tail_train_forward(head, x)
    return head.tail(x)


def _box_pred(m, head, obj):
    if not head.TABLE["convs"]:                        # the dynamic box estimator: FC layers only
        return _tail(m, head, obj)
    if _hip_training(m, obj):
        return _tail(m, head, _train.point_stack_train_forward(head, obj))
    return head(obj)


def _train_forward_one(m, pts, init_box):
    logits = _seg_logits(m, pts)
    obj, mask = _mask_and_gather(pts, logits, NUM_OBJECT_POINT, 3, m)
    c, hs, hrn, hr, ss, srn, sr = _parse(_box_pred(m, m.box_est, obj))
    return {"logits": logits, "mask": mask, "center_boxnet": c, "heading_scores": hs,
            "heading_residuals_normalized": hrn, "heading_residuals": hr, "size_scores": ss,
            "size_residuals_normalized": srn, "size_residuals": sr, "center": c + init_box[:, :3]}


def _train_forward_two(m, pts, init_box, bbox_gt):
    logits = _seg_logits(m, pts)
    obj, mask = _mask_and_gather(pts, logits, NUM_OBJECT_POINT, 3, m)
    c1, hs1, hrn1, hr1, ss1, srn1, sr1 = _parse(_box_pred(m, m.box_est_one, obj))
    c1 = c1 + init_box[:, :3]
    with torch.no_grad():
        B = c1.shape[0]
        ar = torch.arange(B, device=c1.device)
        hc = hs1.argmax(1)
        sc = ss1.argmax(1)
        mean = _mean_size(c1.device).double()
        size = mean[sc] + sr1[ar, sc].double()
        ang = hc.double() * (2 * np.pi / NUM_HEADING_BIN) + hr1[ar, hc].double()
        ang = torch.where(ang > np.pi, ang - 2 * np.pi, ang) + init_box[:, -1].double()
        box_one = torch.cat([c1.double(), size, ang[:, None]], 1).to(c1.dtype)        # (.float() in the reference)
        y0, y1 = init_box[:, -1], -box_one[:, -1]
        x, y, z = obj[:, 0], obj[:, 1], obj[:, 2]
        px = torch.cos(y0)[:, None] * x - torch.sin(y0)[:, None] * y + init_box[:, 0:1] - box_one[:, 0:1]
        py = torch.sin(y0)[:, None] * x + torch.cos(y0)[:, None] * y + init_box[:, 1:2] - box_one[:, 1:2]
        pz = z + init_box[:, 2:3] - box_one[:, 2:3]
        obj2 = torch.stack([torch.cos(y1)[:, None] * px - torch.sin(y1)[:, None] * py,
                            torch.sin(y1)[:, None] * px + torch.cos(y1)[:, None] * py, pz], 1)
        two_pi = 2 * np.pi
        per = two_pi / NUM_HEADING_BIN
        shifted = torch.remainder(torch.remainder(bbox_gt[:, -1] - box_one[:, -1], two_pi) + per / 2, two_pi)
        hcl = (shifted / per).long()
        hrl = shifted - (hcl.to(shifted.dtype) * per + per / 2)
    c2, hs2, hrn2, hr2, ss2, srn2, sr2 = _parse(_box_pred(m, m.box_est_two, obj2))
    c2 = c2 + c1
    return {"logits": logits, "mask": mask, "heading_scores_one": hs1,
            "heading_residuals_normalized_one": hrn1, "heading_residuals_one": hr1, "size_scores_one": ss1,
            "size_residuals_normalized_one": srn1, "size_residuals_one": sr1, "center_one": c1,
            "box_one": box_one, "heading_scores_two": hs2, "heading_residuals_normalized_two": hrn2,
            "heading_residuals_two": hr2, "size_scores_two": ss2, "size_residuals_normalized_two": srn2,
            "size_residuals_two": sr2, "center_two": c2, "heading_class_label_two": hcl,
            "heading_residuals_label_two": hrl, "center": c2, "heading_scores": hs2,
            "heading_residuals": hr2, "size_scores": ss2, "size_residuals": sr2}
