"""The loss modules the reference's train drivers import next to the models: `FrustumPointNetLossOneBoxEst`,
`FrustumPointNetLossTwoBoxEst` (tools/static_model.py:348-517) and `DynamicModelLoss` (tools/dynamic_model.py:321-398),
with `huber_loss` (:341-346). Same call signature, same keys in the returned dict, same weights
(mask + w_box * (10 center + heading class + size class + 20 heading residual + 20 size residual), per stage).

Stock torch ops on whatever device the outputs live on (the reference hard-codes `.cuda()` on its one-hot
tables): next to the per-point stacks this is O(B) work — plus the mask term over the (B*N, 2) logits, which on the
GPU is one pass of lib3dal_hip.so (`_SegCE`: loss and gradient together) instead of six stock kernels, and the five
box terms of an estimate with their gradients are one launch (`_BoxTerms`) instead of ~60.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import arch


def huber_loss(error, delta=1.0):
    """mean of 0.5 q^2 + delta (|e| - q), q = min(|e|, delta)"""
    abs_error = torch.abs(error)
    quadratic = torch.clamp(abs_error, max=delta)
    return torch.mean(0.5 * quadratic ** 2 + delta * (abs_error - quadratic))


_CONST = {}


def _consts(dev, dtype=torch.float32):
    """one-hot tables and MEAN_SIZE_ARR on the device, built once (also keeps the criterion hipGraph-capturable)"""
    c = _CONST.get((dev, dtype))
    if c is None:
        c = _CONST[(dev, dtype)] = (torch.eye(arch.NUM_HEADING_BIN, device=dev, dtype=dtype),
                                    torch.eye(arch.NUM_SIZE_CLUSTER, device=dev, dtype=dtype),
                                    torch.tensor(arch.MEAN_SIZE, dtype=dtype, device=dev).view(1, arch.NUM_SIZE_CLUSTER, 3))
    return c


class _SegCE(torch.autograd.Function):
    """the mask term on lib3dal_hip.so (dal3_tr_seg_ce): loss and softmax - onehot in one pass over the logits"""

    @staticmethod
    def forward(ctx, logits, labels):
        from . import _hip
        lib = _hip.lib()
        lg = logits.detach().reshape(-1, 2).contiguous()
        lab = labels.reshape(-1).contiguous()
        M = lg.shape[0]
        loss = torch.empty(1, dtype=torch.float32, device=lg.device)
        d = torch.empty_like(lg)
        need = lib.dal3_tr_seg_ce_workspace_bytes(M)
        ws = torch.empty(need, dtype=torch.uint8, device=lg.device)
        _hip.check(lib.dal3_tr_seg_ce(_hip.ptr(lg), _hip.ptr(lab), int(lab.dtype == torch.int64), M, _hip.ptr(loss),
                                      _hip.ptr(d), _hip.ptr(ws), need, _hip.stream()))
        ctx.save_for_backward(d)
        ctx.shape, ctx.M = logits.shape, M
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        return (d * (g / ctx.M)).view(ctx.shape), None


def _mask_loss(logits, mask_label):
    if logits.is_cuda and logits.dtype == torch.float32 and mask_label.is_cuda and mask_label.dtype in (torch.float32,
                                                                                                      torch.int64):
        return _SegCE.apply(logits, mask_label)
    return F.nll_loss(F.log_softmax(logits.view(-1, 2), dim=1), mask_label.view(-1).long())


class _BoxTerms(torch.autograd.Function):
    """the five box terms of one estimate and their gradients in one launch of lib3dal_hip.so (dal3_tr_box_loss)"""

    @staticmethod
    def forward(ctx, center, hs, hrn, ss, srn, center_label, hcl, hrl, scl, srl):
        from . import _hip
        f = lambda t: t.detach().contiguous().float()                       # noqa: E731
        B, dev = center.shape[0], center.device
        ins = (f(center), f(center_label), f(hs), f(hrn), hcl.long().contiguous(), f(hrl), f(ss), f(srn),
               scl.long().contiguous(), f(srl))
        losses = torch.empty(5, dtype=torch.float32, device=dev)
        g = [torch.empty(s, dtype=torch.float32, device=dev) for s in ((B, 3), (B, 12), (B, 12), (B, 3), (B, 3, 3))]
        _hip.check(_hip.lib().dal3_tr_box_loss(*[_hip.ptr(t) for t in ins], B, _hip.ptr(losses), *[_hip.ptr(t) for t in g],
                                               _hip.stream()))
        ctx.save_for_backward(*g)
        return losses[0], losses[1], losses[2], losses[3], losses[4]

    @staticmethod
    def backward(ctx, gc, ghc, ghr, gsc, gsr):
        g = ctx.saved_tensors
        return (g[0] * gc, g[1] * ghc, g[2] * ghr, g[3] * gsc, g[4] * gsr, None, None, None, None, None)


def _box_terms(center, heading_scores, heading_residuals_normalized, size_scores, size_residuals_normalized, center_label,
               heading_class_label, heading_residuals_label, size_class_label, size_residuals_label):
    """(center, heading class, heading residual, size class, size residual) losses of ONE box estimate"""
    dev = center.device
    if center.is_cuda and center.dtype == torch.float32 and all(
            t.is_cuda for t in (heading_scores, center_label, heading_class_label, heading_residuals_label,
                                size_class_label, size_residuals_label)):
        return _BoxTerms.apply(center, heading_scores, heading_residuals_normalized, size_scores, size_residuals_normalized,
                               center_label, heading_class_label, heading_residuals_label, size_class_label,
                               size_residuals_label)
    center_loss = huber_loss(torch.norm(center - center_label, dim=1), delta=2.0)
    hcl, scl = heading_class_label.long(), size_class_label.long()
    heading_class_loss = F.nll_loss(F.log_softmax(heading_scores, dim=1), hcl)
    eye_h, eye_s, mean_size = _consts(dev, center.dtype)
    h_onehot = eye_h[hcl]
    h_pred = torch.sum(heading_residuals_normalized * h_onehot, dim=1)
    heading_res_loss = huber_loss(h_pred - heading_residuals_label / (np.pi / arch.NUM_HEADING_BIN), delta=1.0)
    size_class_loss = F.nll_loss(F.log_softmax(size_scores, dim=1), scl)
    s_onehot = eye_s[scl].view(-1, arch.NUM_SIZE_CLUSTER, 1).repeat(1, 1, 3)
    s_pred = torch.sum(size_residuals_normalized * s_onehot, dim=1)
    mean_size_label = torch.sum(s_onehot * mean_size, dim=1)
    size_res_loss = huber_loss(torch.norm(size_residuals_label / mean_size_label - s_pred, dim=1), delta=1.0)
    return center_loss, heading_class_loss, heading_res_loss, size_class_loss, size_res_loss


_WEIGHTS = {}


def _weighted(terms, w_box):
    """the five box terms of an estimate (centre, heading class, heading residual, size class, size residual) with the
    criterion's weights on them — w_box * (10, 1, 20, 1, 20) — as ONE (5,) tensor: a stack and a multiplication instead of
    nine scalar kernels forward and as many backward (the weight vector is built once per value of w_box: an upload
    inside a step would also break hipGraph capture)"""
    t = torch.stack(list(terms))
    key = (float(w_box), t.device, t.dtype)
    w = _WEIGHTS.get(key)
    if w is None:
        w = _WEIGHTS[key] = torch.tensor([10.0, 1.0, 20.0, 1.0, 20.0], dtype=t.dtype, device=t.device) * float(w_box)
    return t * w


class _OneBoxLoss(nn.Module):
    def forward(self, output, mask_label, center_label, heading_class_label, heading_residuals_label, size_class_label,
                size_residuals_label, w_box=1.0):
        mask_loss = _mask_loss(output["logits"], mask_label)
        w = _weighted(_box_terms(output["center"], output["heading_scores"], output["heading_residuals_normalized"],
                                 output["size_scores"], output["size_residuals_normalized"], center_label,
                                 heading_class_label, heading_residuals_label, size_class_label, size_residuals_label), w_box)
        total = mask_loss + w.sum()
        return {"total_loss": total, "mask_loss": mask_loss, "center_loss": w[0], "heading_class_loss": w[1],
                "size_class_loss": w[3], "heading_residuals_normalized_loss": w[2], "size_residuals_normalized_loss": w[4]}


class FrustumPointNetLossOneBoxEst(_OneBoxLoss):
    """tools/static_model.py:348-428"""


class DynamicModelLoss(_OneBoxLoss):
    """tools/dynamic_model.py:321-398 (the same terms on DynamicModel's output dict)"""


class FrustumPointNetLossTwoBoxEst(nn.Module):
    """tools/static_model.py:430-517: both box estimates; stage two is scored against the stage-two heading labels
    the model itself emits (`heading_class_label_two`, `heading_residuals_label_two`)."""

    def forward(self, output, mask_label, center_label, heading_class_label, heading_residuals_label, size_class_label,
                size_residuals_label, w_box=1.0):
        mask_loss = _mask_loss(output["logits"], mask_label)
        one = _box_terms(output["center_one"], output["heading_scores_one"], output["heading_residuals_normalized_one"],
                         output["size_scores_one"], output["size_residuals_normalized_one"], center_label,
                         heading_class_label, heading_residuals_label, size_class_label, size_residuals_label)
        two = _box_terms(output["center_two"], output["heading_scores_two"], output["heading_residuals_normalized_two"],
                         output["size_scores_two"], output["size_residuals_normalized_two"], center_label,
                         output["heading_class_label_two"], output["heading_residuals_label_two"], size_class_label,
                         size_residuals_label)
        w1, w2 = _weighted(one, w_box), _weighted(two, w_box)
        total = mask_loss + (w1 + w2).sum()
        out = {"total_loss": total, "mask_loss": mask_loss}
        for tag, w in (("one", w1), ("two", w2)):
            out.update({f"center_loss_{tag}": w[0], f"heading_class_loss_{tag}": w[1], f"size_class_loss_{tag}": w[3],
                        f"heading_residuals_normalized_loss_{tag}": w[2], f"size_residuals_normalized_loss_{tag}": w[4]})
        return out
