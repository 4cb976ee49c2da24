"""ORACLE — CPU restatement of the 3DAL static/dynamic auto-labeling heads (eval-mode forward).

TEST INFRASTRUCTURE ONLY. Nothing in the product package imports this file. Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and only as the checker /
the reported CPU baseline. The product path (3dal_pytorch_amd/) runs on lib3dal_hip.so and
fails loudly when that library is missing.

What it restates (reference = jacky121298/3DAL_PyTorch, paths relative to /root/reference):
  tools/static_model.py   gather_object_pts :23-49, point_cloud_masking :51-62,
                          parse_output_to_tensors :64-96, rotz :98-106,
                          StaticModelOneBoxEst.forward :117-146, StaticModelTwoBoxEst.forward
                          :158-239, PointNetInstanceSeg.forward :271-296,
                          PointNetEstimation.forward :320-339
  tools/dynamic_model.py  DynamicModel.forward :121-155, PointEmbedding.forward :234-249,
                          BoxEmbedding.forward :271-286, PointNetEstimation.forward :300-312
  tools/utils.py          angle2class :53-60, size2class :62-67, class2angle :69-75,
                          class2size :77-79
  tools/static_eval.py    box decode in test_one_epoch :269-288
  tools/dynamic_eval.py   box decode in test_one_epoch :226-242

Formulation: the reference's own op sequence (no BN folding, materialised repeat+cat, the
host NumPy mask/gather loop on the legacy global MT19937 stream), written functionally over a
state_dict of plain tensors, fp32, torch-CPU kernels. It is therefore also the "reference
formulation" CPU baseline that bench.py times.

Parity pin: the reference ships NO tests or golden vectors for this path. The
oracle is pinned against outputs of the reference itself: tests/golden/*.npz were produced by
tests/golden/gen_golden.py, which imports /root/reference/tools/{static,dynamic}_model.py
(with the import shim of SURVEY.md 8(c)) and runs its forward on the inputs of
3dal_pytorch_amd/synth.py. tests/test_oracle_golden.py checks this file against those vectors
everywhere, and tests/test_oracle_live_reference.py checks it against a live import of the
reference wherever /root/reference exists.
"""
import numpy as np
import torch
import torch.nn.functional as F

NUM_HEADING_BIN = 12
NUM_SIZE_CLUSTER = 3
NUM_OBJECT_POINT = 512
NUM_FRAME = 5
MEAN_SIZE_ARR = np.array([[4.8, 1.8, 1.5], [10.0, 2.6, 3.2], [2.0, 1.0, 1.6]])
_EPS = 1e-5


def as_torch_sd(sd):
    """numpy/torch state_dict -> dict of CPU torch tensors."""
    out = {}
    for k, v in sd.items():
        out[k] = torch.as_tensor(np.asarray(v)) if not torch.is_tensor(v) else v.detach().cpu()
    return out


# ----------------------------------------------------------------------------- layer pieces
def _pointwise(sd, p, conv, x):
    """Conv1d(kernel=1) over (B, C, N): y[b,o,n] = sum_c W[o,c] x[b,c,n] + bias[o].
    Uses the same torch-CPU convolution kernel the reference's nn.Conv1d dispatches to."""
    return F.conv1d(x, sd[f"{p}.{conv}.weight"], sd[f"{p}.{conv}.bias"])


def _bn(sd, p, bn, x):
    """BatchNorm1d in eval mode: (x - running_mean) / sqrt(running_var + eps) * gamma + beta."""
    return F.batch_norm(x, sd[f"{p}.{bn}.running_mean"], sd[f"{p}.{bn}.running_var"],
                        sd[f"{p}.{bn}.weight"], sd[f"{p}.{bn}.bias"], False, 0.0, _EPS)


def _cbr(sd, p, conv, bn, x):
    return torch.relu(_bn(sd, p, bn, _pointwise(sd, p, conv, x)))


def _fbr(sd, p, fc, bn, x):
    y = F.linear(x, sd[f"{p}.{fc}.weight"], sd[f"{p}.{fc}.bias"])
    return torch.relu(_bn(sd, p, bn, y))


# ----------------------------------------------------------------------------- sub-networks
def ins_seg(sd, pts, p="ins_seg", want_global=False):
    """static_model.py:271-296 / dynamic_model.py:187-212. pts (B, Cin, N) -> logits (B, N, 2)."""
    n = pts.shape[2]
    o1 = _cbr(sd, p, "conv1", "bn1", pts)
    o2 = _cbr(sd, p, "conv2", "bn2", o1)
    o3 = _cbr(sd, p, "conv3", "bn3", o2)
    o4 = _cbr(sd, p, "conv4", "bn4", o3)
    o5 = _cbr(sd, p, "conv5", "bn5", o4)
    g = torch.max(o5, 2, keepdim=True)[0]                      # (B, 1024, 1)
    cat = torch.cat([o2, g.repeat(1, 1, n)], 1)                # (B, 1088, N)
    x = _cbr(sd, p, "dconv1", "dbn1", cat)
    x = _cbr(sd, p, "dconv2", "dbn2", x)
    x = _cbr(sd, p, "dconv3", "dbn3", x)
    x = _cbr(sd, p, "dconv4", "dbn4", x)
    x = _pointwise(sd, p, "dconv5", x)                         # dropout = identity in eval
    logits = x.transpose(2, 1).contiguous()
    return (logits, g[:, :, 0]) if want_global else logits


def shared_mlp_max(sd, p, x):
    """conv1..4 (+BN+ReLU) then channel-wise max over the point axis: (B,C,M) -> (B,512)."""
    for i in (1, 2, 3, 4):
        x = _cbr(sd, p, f"conv{i}", f"bn{i}", x)
    return torch.max(x, 2)[0]


def static_box_est(sd, obj, p="box_est"):
    """static_model.py:320-339: (B,3,M) -> (B,39)."""
    g = shared_mlp_max(sd, p, obj)
    x = _fbr(sd, p, "fc1", "fcbn1", g)
    x = _fbr(sd, p, "fc2", "fcbn2", x)
    return F.linear(x, sd[f"{p}.fc3.weight"], sd[f"{p}.fc3.bias"])


def embedding(sd, x, p):
    """PointEmbedding / BoxEmbedding (dynamic_model.py:234-249, 271-286)."""
    g = shared_mlp_max(sd, p, x)
    x = _fbr(sd, p, "fc1", "fcbn1", g)
    return _fbr(sd, p, "fc2", "fcbn2", x)


def dynamic_box_est(sd, emb, p="box_est"):
    """dynamic_model.py:300-312: (B,384) -> (B,39)."""
    x = _fbr(sd, p, "fc1", "fcbn1", emb)
    x = _fbr(sd, p, "fc2", "fcbn2", x)
    return F.linear(x, sd[f"{p}.fc3.weight"], sd[f"{p}.fc3.bias"])


# ----------------------------------------------------------------------------- mask + gather
def segment_mask(logits):
    """static_model.py:59: strict '<', ties -> background."""
    return logits[:, :, 0] < logits[:, :, 1]


def gather_object_pts(pts, mask, n_obj):
    """static_model.py:23-49. Consumes the GLOBAL legacy NumPy stream in the reference's exact
    per-sample order: choice (without replacement if count >= n_obj, else the top-up draw with
    replacement), then shuffle; a sample with no positive point consumes nothing and stays zero.
    Returns object_pts (B, C, n_obj) fp32 and indices (B, n_obj) int64."""
    bs, c = pts.shape[0], pts.shape[1]
    obj = torch.zeros((bs, c, n_obj))
    idx = torch.zeros((bs, n_obj), dtype=torch.int64)
    for i in range(bs):
        pos = torch.nonzero(mask[i]).squeeze(1)
        k = len(pos)
        if k == 0:
            continue
        if k >= n_obj:
            choice = np.random.choice(k, n_obj, replace=False)
        else:
            extra = np.random.choice(k, n_obj - k, replace=True)
            choice = np.concatenate((np.arange(k), extra))
        np.random.shuffle(choice)
        idx[i] = pos[choice]
        obj[i] = pts[i][:, idx[i]]
    return obj, idx


def take_object_pts(pts, idx, counts):
    """Teacher-forced gather: rows with count 0 stay zero."""
    bs, c = pts.shape[0], pts.shape[1]
    obj = torch.zeros((bs, c, idx.shape[1]))
    for i in range(bs):
        if counts[i] > 0:
            obj[i] = pts[i][:, idx[i]]
    return obj


# ----------------------------------------------------------------------------- parse / decode
def parse_box_pred(box_pred):
    """static_model.py:64-96 (dynamic twin :65-97). box_pred (B,39) -> 7 tensors."""
    bs = box_pred.shape[0]
    nh, ns = NUM_HEADING_BIN, NUM_SIZE_CLUSTER
    center = box_pred[:, 0:3]
    hs = box_pred[:, 3:3 + nh]
    hrn = box_pred[:, 3 + nh:3 + 2 * nh]
    hr = hrn * (np.pi / nh)
    ss = box_pred[:, 3 + 2 * nh:3 + 2 * nh + ns]
    srn = box_pred[:, 3 + 2 * nh + ns:3 + 2 * nh + 4 * ns].contiguous().view(bs, ns, 3)
    sr = srn * torch.from_numpy(MEAN_SIZE_ARR).float()[None]
    return center, hs, hrn, hr, ss, srn, sr


def angle2class(angle, num_class):
    """utils.py:53-60 (works on python floats and 0-dim tensors alike)."""
    angle = angle % (2 * np.pi)
    per = 2 * np.pi / float(num_class)
    shifted = (angle + per / 2) % (2 * np.pi)
    cid = int(shifted / per)
    return cid, shifted - (cid * per + per / 2)


def class2angle(cls, residual, num_class):
    """utils.py:69-75 with to_label_format=True."""
    ang = cls * (2 * np.pi / float(num_class)) + residual
    if ang > np.pi:
        ang = ang - 2 * np.pi
    return ang


def class2size(cls, residual):
    """utils.py:77-79."""
    return MEAN_SIZE_ARR[cls] + residual


def size2class(lwh):
    """utils.py:62-67."""
    d = np.linalg.norm(lwh[np.newaxis, ...] - MEAN_SIZE_ARR, axis=1)
    cid = int(np.argmin(d))
    return cid, lwh - MEAN_SIZE_ARR[cid]


def _decode_size_angle(hs, hr, ss, sr):
    """argmax class -> (size (B,3), angle (B,)) in float64 as the eval drivers do on the host."""
    hs, hr, ss, sr = (t.detach().numpy() for t in (hs, hr, ss, sr))
    bs = hs.shape[0]
    hc = np.argmax(hs, 1)
    sc = np.argmax(ss, 1)
    size = np.zeros((bs, 3))
    ang = np.zeros((bs,))
    for i in range(bs):
        size[i] = class2size(sc[i], sr[i, sc[i], :])
        ang[i] = class2angle(hc[i], hr[i, hc[i]], NUM_HEADING_BIN)
    return size, ang


def rotz(a):
    """static_model.py:98-106: a is a 0-dim fp32 tensor; cos/sin are taken in fp32."""
    c, s = torch.cos(a), torch.sin(a)
    return torch.tensor([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


# ----------------------------------------------------------------------------- full forwards
def static_one_forward(sd, pts, init_box, forced=None):
    """StaticModelOneBoxEst.forward (static_model.py:117-146). pts (B,3,N) logical layout.
    forced=(idx, counts) teacher-forces the object-point indices instead of drawing them."""
    logits = ins_seg(sd, pts)
    mask = segment_mask(logits)
    if forced is None:
        obj, idx = gather_object_pts(pts[:, :3, :], mask, NUM_OBJECT_POINT)
    else:
        idx = forced[0]
        obj = take_object_pts(pts[:, :3, :], idx, forced[1])
    box_pred = static_box_est(sd, obj.float(), "box_est")
    c, hs, hrn, hr, ss, srn, sr = parse_box_pred(box_pred)
    return {
        "logits": logits, "mask": mask, "center_boxnet": c, "heading_scores": hs,
        "heading_residuals_normalized": hrn, "heading_residuals": hr, "size_scores": ss,
        "size_residuals_normalized": srn, "size_residuals": sr, "center": c + init_box[:, :3],
        "_indices": idx, "_object_pts": obj,
    }


def static_two_forward(sd, pts, init_box, bbox_gt, forced=None):
    """StaticModelTwoBoxEst.forward (static_model.py:158-239), including its quirk that
    center_one (predicted in the init-box-aligned frame) is added to init_box[:, :3] un-rotated."""
    logits = ins_seg(sd, pts)
    mask = segment_mask(logits)
    if forced is None:
        obj, idx = gather_object_pts(pts[:, :3, :], mask, NUM_OBJECT_POINT)
    else:
        idx = forced[0]
        obj = take_object_pts(pts[:, :3, :], idx, forced[1])
    obj = obj.float()
    obj2 = obj.clone()
    bp1 = static_box_est(sd, obj, "box_est_one")
    c1, hs1, hrn1, hr1, ss1, srn1, sr1 = parse_box_pred(bp1)
    c1 = c1 + init_box[:, :3]
    size, ang = _decode_size_angle(hs1, hr1, ss1, sr1)
    ang = ang + init_box.numpy()[:, -1].astype(np.float64)
    box_one = torch.from_numpy(np.concatenate((c1.numpy(), size, ang[:, None]), 1)).float()
    bs = pts.shape[0]
    hcl = np.zeros((bs,))
    hrl = np.zeros((bs,))
    for i in range(bs):
        o = rotz(init_box[i, -1]) @ obj2[i]
        o = o + init_box[i, :3][:, None]
        o = o - box_one[i, :3][:, None]
        obj2[i] = rotz(-box_one[i, -1]) @ o
        hcl[i], hrl[i] = angle2class(bbox_gt[i, -1] - box_one[i, -1], NUM_HEADING_BIN)
    bp2 = static_box_est(sd, obj2, "box_est_two")
    c2, hs2, hrn2, hr2, ss2, srn2, sr2 = parse_box_pred(bp2)
    c2 = c2 + c1
    return {
        "logits": logits, "mask": mask,
        "heading_scores_one": hs1, "heading_residuals_normalized_one": hrn1,
        "heading_residuals_one": hr1, "size_scores_one": ss1,
        "size_residuals_normalized_one": srn1, "size_residuals_one": sr1,
        "center_one": c1, "box_one": box_one,
        "heading_scores_two": hs2, "heading_residuals_normalized_two": hrn2,
        "heading_residuals_two": hr2, "size_scores_two": ss2,
        "size_residuals_normalized_two": srn2, "size_residuals_two": sr2, "center_two": c2,
        "heading_class_label_two": torch.from_numpy(hcl).long(),
        "heading_residuals_label_two": torch.from_numpy(hrl).float(),
        "center": c2, "heading_scores": hs2, "heading_residuals": hr2,
        "size_scores": ss2, "size_residuals": sr2,
        "_indices": idx, "_object_pts": obj, "_object_pts_two": obj2,
    }


def dynamic_forward(sd, pts, box, forced=None):
    """DynamicModel.forward (dynamic_model.py:121-155). pts (B,4,N), box (B,8,n_box)."""
    m = NUM_FRAME * NUM_OBJECT_POINT
    logits = ins_seg(sd, pts)
    mask = segment_mask(logits)
    if forced is None:
        obj, idx = gather_object_pts(pts[:, :4, :], mask, m)
    else:
        idx = forced[0]
        obj = take_object_pts(pts[:, :4, :], idx, forced[1])
    pe = embedding(sd, obj.float(), "point_emb")
    be = embedding(sd, box, "box_emb")
    box_pred = dynamic_box_est(sd, torch.cat([pe, be], 1), "box_est")
    c, hs, hrn, hr, ss, srn, sr = parse_box_pred(box_pred)
    return {
        "logits": logits, "mask": mask, "center": c, "heading_scores": hs,
        "heading_residuals_normalized": hrn, "heading_residuals": hr, "size_scores": ss,
        "size_residuals_normalized": srn, "size_residuals": sr,
        "_indices": idx, "_object_pts": obj, "_point_e": pe, "_box_e": be,
    }


# ----------------------------------------------------------------------------- refined boxes
def decode_static(out, init_box, two_stage):
    """static_eval.py:269-288 -> (B,7) float64 [cx,cy,cz,l,w,h,yaw]."""
    size, ang = _decode_size_angle(out["heading_scores"], out["heading_residuals"],
                                   out["size_scores"], out["size_residuals"])
    base = out["box_one"][:, -1] if two_stage else init_box[:, -1]
    ang = ang + base.detach().numpy().astype(np.float64)
    return np.concatenate((out["center"].detach().numpy(), size, ang[:, None]), 1)


def decode_dynamic(out, init_box8):
    """dynamic_eval.py:226-242 -> (B,7); yaw base is init_box[:, -2], centre adds init_box[:, :3]
    (in float32, in place on the numpy centre, as the driver does)."""
    size, ang = _decode_size_angle(out["heading_scores"], out["heading_residuals"],
                                   out["size_scores"], out["size_residuals"])
    ib = init_box8.detach().numpy()
    ang = ang + ib[:, -2].astype(np.float64)
    center = out["center"].detach().numpy().copy()
    center += ib[:, :3]
    return np.concatenate((center, size, ang[:, None]), 1)


# ----------------------------------------------------------------------------- folded form
def fold_bn(sd, p, layer, bn):
    """Eval-mode BN folded into the preceding affine map: W' = W*g/sqrt(v+eps),
    b' = (b-mean)*g/sqrt(v+eps)+beta. Used by tests to check the product's packed weights."""
    w = sd[f"{p}.{layer}.weight"]
    w = w[:, :, 0] if w.dim() == 3 else w
    b = sd[f"{p}.{layer}.bias"]
    if bn is None:
        return w.clone(), b.clone()
    s = sd[f"{p}.{bn}.weight"] / torch.sqrt(sd[f"{p}.{bn}.running_var"] + _EPS)
    return w * s[:, None], (b - sd[f"{p}.{bn}.running_mean"]) * s + sd[f"{p}.{bn}.bias"]
