"""Host-side pieces shared by static_model.py and dynamic_model.py: the parameter containers
(same attribute names as the reference so state_dicts load with strict=True), the packed-weight
cache, the NumPy-stream sampler, and train-mode composites.

Eval-mode math lives in lib3dal_hip.so; torch is used here for parameters, device buffers and
the stream (plumbing). The train-mode `forward_train` methods run stock torch ops because
training needs batch-statistics BN, live dropout and autograd (SURVEY.md 8(a) note T); they are
never used when `self.training` is False.
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _hip, arch


def _conv_bn_stack(mod, specs, conv_cls):
    for name, bn, ci, co in specs:
        setattr(mod, name, conv_cls(ci, co))
    for name, bn, ci, co in specs:
        if bn:
            setattr(mod, bn, nn.BatchNorm1d(co))


def _conv1d(ci, co):
    return nn.Conv1d(ci, co, 1)


class PointNetInstanceSeg(nn.Module):
    """Parameters of tools/static_model.py:241-269 / tools/dynamic_model.py:157-185."""

    def __init__(self, n_classes=3, n_channel=3):
        super().__init__()
        self.n_channel = n_channel
        layers = arch.ins_seg_layers(n_channel)
        _conv_bn_stack(self, layers[:5], _conv1d)
        for name, bn, ci, co in layers[5:9]:
            setattr(self, name, _conv1d(ci, co))
        self.dropout = nn.Dropout(p=0.5)
        self.dconv5 = _conv1d(128, 2)
        for name, bn, ci, co in layers[5:9]:
            setattr(self, bn, nn.BatchNorm1d(co))

    def pairs(self):
        return [(getattr(self, n), getattr(self, b) if b else None)
                for n, b, _, _ in arch.ins_seg_layers(self.n_channel)]

    def forward(self, pts):   # train-mode composite (stock torch ops, autograd)
        n = pts.size(2)
        o1 = F.relu(self.bn1(self.conv1(pts)))
        o2 = F.relu(self.bn2(self.conv2(o1)))
        o3 = F.relu(self.bn3(self.conv3(o2)))
        o4 = F.relu(self.bn4(self.conv4(o3)))
        o5 = F.relu(self.bn5(self.conv5(o4)))
        g = torch.max(o5, 2, keepdim=True)[0]
        x = torch.cat([o2, g.expand(-1, -1, n)], 1)
        x = F.relu(self.dbn1(self.dconv1(x)))
        x = F.relu(self.dbn2(self.dconv2(x)))
        x = F.relu(self.dbn3(self.dconv3(x)))
        x = F.relu(self.dbn4(self.dconv4(x)))
        x = self.dconv5(self.dropout(x))
        return x.transpose(2, 1).contiguous()


class _PointHead(nn.Module):
    """conv1..4 + bn1..4 + fc* + fcbn* with the reference's registration order."""
    TABLE = None
    HEAD_KIND = None

    def __init__(self, n_classes=3):
        super().__init__()
        _conv_bn_stack(self, self.TABLE["convs"], _conv1d)
        _conv_bn_stack(self, self.TABLE["fcs"], nn.Linear)

    def pairs(self):
        return [(getattr(self, n), getattr(self, b) if b else None)
                for n, b, _, _ in self.TABLE["convs"] + self.TABLE["fcs"]]

    def forward(self, x):     # train-mode composite
        for name, bn, _, _ in self.TABLE["convs"]:
            x = F.relu(getattr(self, bn)(getattr(self, name)(x)))
        if self.TABLE["convs"]:
            x = torch.max(x, 2)[0]
        return self.tail(x)

    def tail(self, x):        # the per-item FC layers after the max over points
        for name, bn, _, _ in self.TABLE["fcs"]:
            x = getattr(self, name)(x)
            if bn:
                x = F.relu(getattr(self, bn)(x))
        return x


class StaticPointNetEstimation(_PointHead):
    """tools/static_model.py:298-318."""
    TABLE = arch.STATIC_BOX_EST
    HEAD_KIND = _hip.HEAD_STATIC_BOX_EST


class PointEmbedding(_PointHead):
    """tools/dynamic_model.py:214-232."""
    TABLE = arch.POINT_EMB
    HEAD_KIND = _hip.HEAD_POINT_EMB


class BoxEmbedding(_PointHead):
    """tools/dynamic_model.py:251-269."""
    TABLE = arch.BOX_EMB
    HEAD_KIND = _hip.HEAD_BOX_EMB


class DynamicPointNetEstimation(_PointHead):
    """tools/dynamic_model.py:288-298."""
    TABLE = arch.DYNAMIC_BOX_EST
    HEAD_KIND = _hip.HEAD_DYNAMIC_BOX_EST


# Bumped whenever ANY module registers a parameter, a buffer or a SUBMODULE (assigning a new nn.Parameter to `conv.weight`
# goes through register_parameter, `head.conv1 = nn.Conv1d(...)` through register_module): a PackedCache's cached tensor
# lists carry the value they were built at and are rebuilt when it moved — one integer compare per forward instead of a
# walk over the module tree. What still bypasses it: direct writes into `module._parameters[...]` / `_modules[...]`
# (no hook fires) — call `model.invalidate_packed()` after those, as after `.data` writes.
_REGISTRATION_EPOCH = [0]


def _bump_registration_epoch(*_a, **_k):
    _REGISTRATION_EPOCH[0] += 1


nn.modules.module.register_module_parameter_registration_hook(_bump_registration_epoch)
nn.modules.module.register_module_buffer_registration_hook(_bump_registration_epoch)
nn.modules.module.register_module_module_registration_hook(_bump_registration_epoch)


class PackedCache:
    """Folded + fragment-ordered device weights are a derived cache of the nn.Parameters. A blob is rebuilt when
    any tensor of its module changed identity — (data_ptr, version counter) per tensor, compared as a tuple, plus the
    shapes and devices taken when the tensor list was built — which covers load_state_dict, .cuda()/.to(), optimizer
    steps and every in-place op on the tensor itself. It does NOT see writes that bypass the version counter:
    `p.data.copy_()`, `p.data.mul_()`, EMA updates through `.data`, writes through raw pointers. After such an update
    call `model.invalidate_packed()`; the model classes also call it from load_state_dict and _apply.
    DAL3_CHECK_PACKED=1 (debug) additionally checksums every tensor on the device at each use and rebuilds on a
    mismatch (one device->host sync per forward).

    Cost per forward (round 3): the module tree is walked when the cache is (re)built or a parameter / buffer was
    registered anywhere since (`_REGISTRATION_EPOCH`), not on every call — the per-call check reads data_ptr and _version
    of the cached list (153 tensors for StaticModelTwoBoxEst: ~25 us instead of ~190 us of a 145 us B = 1 call)."""

    def __init__(self):
        self._stamp = {}
        self._blob = {}
        self._src = {}
        self._lists = {}                                    # id(module) -> (epoch, module, tensors, static part of the stamp)
        self._check = os.environ.get("DAL3_CHECK_PACKED") == "1"

    @staticmethod
    def _tensors(module):
        return list(module.parameters()) + list(module.buffers())

    def _list_of(self, module):
        ent = self._lists.get(id(module))
        if ent is None or ent[0] != _REGISTRATION_EPOCH[0] or ent[1] is not module:
            ts = self._tensors(module)
            ent = (_REGISTRATION_EPOCH[0], module, ts, tuple((tuple(t.shape), str(t.device)) for t in ts))
            self._lists[id(module)] = ent
        return ent[2], ent[3]

    def _stamp_of(self, module, dtype):
        ts, static = self._list_of(module)
        stamp = (static, tuple((t.data_ptr(), t._version) for t in ts), dtype)
        if self._check:
            with torch.no_grad():
                stamp += (tuple(float(t.detach().double().sum()) for t in ts),)
        return stamp

    def get(self, key, module, head_kind, dtype=_hip.F32):
        stamp = self._stamp_of(module, dtype)
        if self._stamp.get(key) != stamp:
            dev = next(module.parameters()).device
            self._blob[key] = _hip.pack(head_kind, module.pairs(), dev, dtype)
            self._stamp[key] = stamp
            self._src[key] = (module, dtype)
        return self._blob[key]

    def current(self):
        """True when every packed blob still matches the tensors it was built from"""
        return all(self._stamp_of(mod, dt) == self._stamp[key] for key, (mod, dt) in self._src.items())

    def invalidate(self):
        """forget every packed blob: the next forward re-folds and re-packs from the current parameters"""
        self._stamp.clear()
        self._blob.clear()
        self._src.clear()
        self._lists.clear()

    def snapshot(self):
        """(stamps, blobs) as they are now: what a captured hipGraph must keep alive and compare against"""
        return dict(self._stamp), dict(self._blob)


class PackedModelMixin:
    """invalidate_packed() + the hooks that call it; mixed into the three model classes"""

    def invalidate_packed(self):
        self._cache.invalidate()

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        self._cache.invalidate()
        return out

    def train(self, mode=True):
        # the HIP training kernels update running statistics through raw pointers (no version bump): whatever was
        # packed before a train/eval switch is not trusted after it
        out = super().train(mode)
        if hasattr(self, "_cache"):
            self._cache.invalidate()
        return out

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        if hasattr(self, "_cache"):
            self._cache.invalidate()
        return out


def dtype_of(precision):
    """'fp32' (default, exact-fp32 MFMA) | 'bf16' | 'fp16' (16-bit MFMA operands, fp32 accumulate)."""
    try:
        return _hip.DTYPES[precision]
    except KeyError:
        raise ValueError(f"unknown precision {precision!r}; use one of {sorted(_hip.DTYPES)}") from None


def numpy_choice(counts, m):
    """The reference's per-sample draws (gather_object_pts, static_model.py:36-47) on the global
    legacy NumPy stream, in order; rows with count 0 consume nothing. Returns (B,m) int32 positions
    into each sample's ordered list of segmented points."""
    out = np.zeros((len(counts), m), np.int32)
    for i, k in enumerate(counts):
        k = int(k)
        if k == 0:
            continue
        if k >= m:
            choice = np.random.choice(k, m, replace=False)
        else:
            choice = np.concatenate((np.arange(k), np.random.choice(k, m - k, replace=True)))
        np.random.shuffle(choice)
        out[i] = choice
    return out


class Workspace:
    """One growing device buffer per module, handed to the library as the caller-owned workspace."""

    def __init__(self):
        self.buf = None

    def get(self, nbytes, device):
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return self.buf


def as_f32(t, what):
    _hip.require_gpu(t, what)
    return t if t.dtype == torch.float32 else t.float()


def as_points(t, what):
    """points / box windows as the library reads them: fp32, or bf16 / fp16 storage IN PLACE (widened exactly inside the
    kernels' loads — no fp32 copy; include/dal3.h dal3_bcn.dtype); anything else (float64 from a Dataset) becomes fp32"""
    _hip.require_gpu(t, what)
    return t if t.dtype in _hip.STORAGE else t.float()


def rows_contiguous(t):
    """(B,K) fp32 with contiguous rows (the library takes these as plain (B,K) arrays)."""
    return t if t.is_contiguous() else t.contiguous()
