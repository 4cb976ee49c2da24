"""ctypes binding of lib3dal_hip.so (include/dal3.h). No torch math here: tensors are only a way
to own device memory (data_ptr) and to find the current HIP stream.

The library is REQUIRED: every eval-mode forward of the model classes goes through it, and
loading raises if it has not been built (python -c "import __graft_entry__ as g; g.build()" or
`make -C 3dal_pytorch_amd/csrc`). There is no CPU / eager fallback.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib3dal_hip.so")

OK, EINVAL, EWORKSPACE, EHIP = 0, -1, -2, -3
F32, BF16, F16, F16X3 = 0, 1, 2, 3
DTYPES = {"fp32": F32, "bf16": BF16, "fp16": F16, "f16x3": F16X3}
HEAD_INS_SEG, HEAD_STATIC_BOX_EST, HEAD_POINT_EMB, HEAD_BOX_EMB, HEAD_DYNAMIC_BOX_EST = range(5)
SAMPLER_DEVICE, SAMPLER_CHOICE = 0, 1
PHASE_SEG, PHASE_BOX, PHASE_ALL = 1, 2, 3

vp = C.c_void_p


class Layer(C.Structure):
    _fields_ = [("weight", vp), ("bias", vp), ("bn_weight", vp), ("bn_bias", vp), ("bn_mean", vp),
                ("bn_var", vp), ("c_in", C.c_int32), ("c_out", C.c_int32)]


class PackItem(C.Structure):
    _fields_ = [("W", vp), ("ldw", C.c_int64), ("transpose_w", C.c_int32), ("c_out", C.c_int32), ("c_in", C.c_int32),
                ("mtb", C.c_int32), ("out", vp)]


class WgradPart(C.Structure):
    """dal3_tr_wgrad_part"""
    _fields_ = [("part", vp), ("n_slices", C.c_int64), ("n", C.c_int64), ("dW", vp)]


class BCN(C.Structure):
    _fields_ = [("data", vp), ("stride_b", C.c_int64), ("stride_c", C.c_int64), ("stride_n", C.c_int64),
                ("dtype", C.c_int32), ("flags", C.c_int32)]


class StaticArgs(C.Structure):
    _fields_ = [("B", C.c_int32), ("N", C.c_int32), ("two_stage", C.c_int32), ("sampler", C.c_int32),
                ("dtype", C.c_int32), ("reserved", C.c_int32), ("seed", C.c_uint64), ("item_offset", C.c_int64), ("pts", BCN),
                ("init_box", vp), ("bbox_gt", vp), ("choice", vp),
                ("w_ins_seg", vp), ("w_box_est_one", vp), ("w_box_est_two", vp),
                ("logits", vp), ("mask", vp),
                ("box_pred_one", vp), ("heading_residuals_one", vp), ("size_residuals_one", vp),
                ("center_one", vp), ("box_one", vp),
                ("box_pred_two", vp), ("heading_residuals_two", vp), ("size_residuals_two", vp),
                ("center_two", vp), ("heading_class_label_two", vp), ("heading_residuals_label_two", vp),
                ("boxes7", vp), ("counts", vp), ("obj_idx", vp),
                ("workspace", vp), ("workspace_bytes", C.c_size_t)]


class DynamicArgs(C.Structure):
    _fields_ = [("B", C.c_int32), ("N", C.c_int32), ("n_box", C.c_int32), ("sampler", C.c_int32),
                ("dtype", C.c_int32), ("reserved", C.c_int32), ("seed", C.c_uint64), ("item_offset", C.c_int64), ("pts", BCN), ("box", BCN),
                ("init_box8", vp), ("choice", vp),
                ("w_ins_seg", vp), ("w_point_emb", vp), ("w_box_emb", vp), ("w_box_est", vp),
                ("logits", vp), ("mask", vp), ("embedding", vp), ("box_pred", vp),
                ("heading_residuals", vp), ("size_residuals", vp), ("boxes7", vp),
                ("counts", vp), ("obj_idx", vp),
                ("workspace", vp), ("workspace_bytes", C.c_size_t)]


# every symbol include/dal3.h declares: (restype, argtypes)
_i, _i64, _u64, _sz = C.c_int, C.c_int64, C.c_uint64, C.c_size_t
SIGNATURES = {
    "dal3_version": (_i, []),
    "dal3_last_error": (C.c_char_p, []),
    "dal3_mean_size": (C.POINTER(C.c_float), []),
    "dal3_pack_weights": (_i, [_i, C.POINTER(Layer), _i, _i, vp, C.POINTER(_sz), vp]),
    "dal3_ins_seg_workspace_bytes": (_sz, [_i]),
    "dal3_ins_seg_forward": (_i, [vp, _i, _i, BCN, _i, _i, vp, vp, vp, vp, _sz, vp]),
    "dal3_ins_seg_encode": (_i, [vp, _i, _i, BCN, _i, _i, vp, vp]),
    "dal3_ins_seg_global_bias": (_i, [vp, _i, vp, _i, vp, vp]),
    "dal3_ins_seg_decode": (_i, [vp, _i, _i, BCN, _i, _i, vp, vp, vp, vp]),
    "dal3_gather_workspace_bytes": (_sz, [_i, _i]),
    "dal3_segment_counts": (_i, [vp, _i, _i, vp, vp]),
    "dal3_mask_compact_sample": (_i, [vp, BCN, _i, _i, _i, _i, _i, vp, _u64, _i64, vp, vp, vp, vp, _sz, vp]),
    "dal3_mask_compact_sample_step": (_i, [vp, BCN, _i, _i, _i, _i, _u64, vp, _i64, vp, vp, vp, vp, _sz, vp]),
    "dal3_point_head_workspace_bytes": (_sz, [_i]),
    "dal3_point_head_forward": (_i, [_i, vp, _i, BCN, _i, _i, vp, _i64, vp, _sz, vp]),
    "dal3_point_head_pool_workspace_bytes": (_sz, [_i, _i]),
    "dal3_point_head_pool": (_i, [_i, vp, _i, BCN, _i, _i, vp, vp, vp, _sz, vp]),
    "dal3_dynamic_box_est_forward": (_i, [vp, vp, _i, vp, vp, _sz, vp]),
    "dal3_decode_boxes": (_i, [vp, _i, vp, _i64, _i, vp, _i64, vp, _i64, vp, vp, vp, vp, vp]),
    "dal3_recenter_rotz": (_i, [vp, _i, _i, vp, vp, vp, vp, vp, vp, vp]),
    "dal3_static_crop_prep": (_i, [vp, vp, vp, vp, vp, _i, _i, _u64, _i64, vp, vp, vp]),
    "dal3_dynamic_item_prep": (_i, [vp, vp, vp, vp, vp, vp, vp, vp, _i, _i, _i, _i, _u64, _i64, vp, vp, vp, vp]),
    "dal3_writeback_boxes": (_i, [vp, vp, vp, vp, vp, vp, vp, vp, vp, _i, _i64, vp, vp, vp]),
    "dal3_static_crop_labels": (_i, [vp, vp, vp, vp, _i, _i, _u64, _i64, vp, vp, vp]),
    "dal3_dynamic_item_labels": (_i, [vp, vp, vp, vp, vp, vp, vp, _i, _i, _i, _u64, _i64, vp, vp, vp, vp, vp]),
    "dal3_points_in_boxes": (_i, [vp, _i, _i64, _i64, vp, _i, _i, vp, vp]),
    "dal3_crop_workspace_bytes": (_sz, [_i64, _i64]),
    "dal3_crop_count": (_i, [vp, vp, vp, vp, vp, _i, _i64, _i64, vp, vp, _sz, vp]),
    "dal3_crop_fill": (_i, [vp, vp, vp, vp, vp, _i, _i64, _i64, vp, vp, vp, vp, vp, _i64, vp, _sz, vp]),
    "dal3_crop_starts": (_i, [vp, vp, _i64, vp, vp, vp]),
    "dal3_crop_starts_capped": (_i, [vp, vp, _i64, vp, vp, _i64, vp]),
    "dal3_tr_linear": (_i, [vp, _i64, _i, _i64, vp, vp, _i, vp, _i64, _i, vp, _i64, _i, vp, _i64, _i, vp, _sz, vp]),
    "dal3_tr_linear_workspace_bytes": (_sz, [_i, _i]),
    "dal3_tr_linear_pack_layout": (_i, [_i64, _i, _i64, _i, _i, _i]),
    "dal3_tr_pack_many": (_i, [vp, _i, vp]),
    "dal3_tr_linear_prepacked": (_i, [vp, _i64, _i, _i64, vp, vp, _i, vp, _i64, _i, vp, _i64, _i, vp, _i64, _i, vp, vp]),
    "dal3_tr_linear_x3_layout": (_i, [_i64, _i, _i64, _i, _i, _i]),
    "dal3_tr_linear_x3": (_i, [vp, _i64, _i, _i64, vp, vp, _i, vp, _i64, _i, vp, _i64, vp, vp, vp]),
    "dal3_tr_bnbwd_apply_amax": (_i, [vp, _i64, _i, _i64, vp, _i64, vp, vp, _i64, vp, vp, vp, vp, vp, vp, vp, vp, _i64, vp, vp]),
    "dal3_tr_colred_workspace_bytes": (_sz, [_i64, _i]),
    "dal3_tr_act_colsum": (_i, [vp, _i64, _i, _i64, vp, vp, _i, vp, _i64, vp, _sz, vp, vp]),
    "dal3_tr_colred": (_i, [vp, _i64, _i, _i64, _i, vp, _i64, vp, vp, _i64, vp, vp, vp, vp, vp, _sz, vp, vp]),
    "dal3_tr_pool_coef": (_i, [vp, vp, vp, vp, vp, vp, _i, _i, _i64, vp, vp, vp]),
    "dal3_tr_head2_forward": (_i, [vp, _i64, _i, _i64, vp, vp, _i, vp, _i64, _u64, vp, C.c_float, vp, _i64, vp, vp, vp]),
    "dal3_tr_head2_dgrad": (_i, [vp, _i64, _i, vp, _i64, _u64, vp, C.c_float, vp, _i64, vp, _i64, vp]),
    "dal3_tr_head2_dgrad_bnbwd": (_i, [vp, _i64, _i, vp, _i64, _u64, vp, C.c_float, vp, _i64, vp, _i64, vp, _i64, vp, vp, vp, vp, vp,
                                       vp, vp, vp, vp, vp, vp, _sz, vp]),
    "dal3_tr_head2_wgrad_workspace_bytes": (_sz, [_i64]),
    "dal3_tr_head2_wgrad": (_i, [vp, vp, _i64, _i, _i64, vp, vp, _i, vp, _i64, _u64, vp, C.c_float, vp, _sz, vp, vp]),
    "dal3_tr_conv1_workspace_bytes": (_sz, [_i64, _i]),
    "dal3_tr_conv1_bn_stats": (_i, [vp, _i64, _i64, _i, _i64, vp, _i64, vp, _i, vp, _i64, vp, vp, vp, vp, C.c_float, C.c_float, vp, vp,
                                    vp, vp, vp, _sz, vp]),
    "dal3_tr_conv1_wgrad": (_i, [vp, _i64, vp, _i64, _i, _i64, _i, vp, _sz, vp, vp]),
    "dal3_tr_gather_at": (_i, [vp, _i64, vp, _i64, _i, _i, vp, vp]),
    "dal3_tr_pool_zarg": (_i, [vp, vp, _i64, vp, _i64, vp, _i, _i, _i, _i, vp, vp]),
    "dal3_tr_pool_moments": (_i, [vp, _i64, vp, vp, vp, _i64, _i, _i, vp, vp]),
    "dal3_tr_pool_gv_workspace_bytes": (_sz, [_i]),
    "dal3_tr_pool_gv": (_i, [vp, vp, _i64, vp, _i, _i, vp, vp, vp, _sz, vp]),
    "dal3_tr_pool_dw": (_i, [vp, vp, _i64, vp, vp, vp, _i64, _i, vp, _i, _i, vp, vp]),
    "dal3_tr_pool_sparse": (_i, [vp, vp, vp, _i64, vp, _i64, _i, _i, _i, _i, vp, _i64, vp, vp]),
    "dal3_tr_box_loss": (_i, [vp] * 10 + [_i] + [vp] * 7),
    "dal3_tr_seg_ce_workspace_bytes": (_sz, [_i64]),
    "dal3_tr_seg_ce": (_i, [vp, vp, _i, _i64, vp, vp, vp, _sz, vp]),
    "dal3_tr_linear_red_workspace_bytes": (_sz, [_i64, _i]),
    "dal3_tr_linear_bn_stats": (_i, [vp, _i64, _i, _i64, vp, vp, _i, vp, _i64, vp, _i64, _i, vp, _i64, vp, _i64, vp, vp, vp, vp,
                                     C.c_float, C.c_float, vp, vp, vp, vp, vp, _sz, vp]),
    "dal3_tr_linear_bnbwd_sums": (_i, [vp, _i64, _i, _i64, vp, _i64, _i, vp, _i64, vp, _i64, vp, _i64, vp, vp, vp, vp, vp,
                                       vp, vp, vp, vp, vp, vp, _sz, vp]),
    "dal3_tr_bn_stats": (_i, [vp, _i64, _i, _i64, vp, vp, vp, vp, C.c_float, C.c_float, vp, vp, vp, vp, vp, _sz, vp]),
    "dal3_tr_bnbwd_sums": (_i, [vp, _i64, _i, _i64, vp, _i64, vp, vp, _i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, _sz,
                                vp]),
    "dal3_tr_bn_finalize": (_i, [vp, _i, _i64, vp, vp, vp, vp, C.c_float, C.c_float, vp, vp, vp, vp, vp]),
    "dal3_tr_bnbwd_coef": (_i, [vp, _i, _i64, vp, vp, vp, vp, vp, vp, vp, vp]),
    "dal3_parse_box_pred": (_i, [vp, _i64, _i64, vp, vp, vp, vp, vp, vp, vp, vp]),
    "dal3_parse_box_pred_backward": (_i, [vp, vp, vp, vp, vp, vp, vp, _i64, vp, vp]),
    "dal3_tr_fc_max_rows": (_i, []),
    "dal3_tr_fc_max_act_cin": (_i, []),
    "dal3_tr_fc_forward": (_i, [vp, _i64, _i, _i64, vp, vp, _i, vp, _i64, _i, vp, _i, vp, _i64, vp, vp, vp, vp, C.c_float, C.c_float, vp, vp, vp, vp, vp]),
    "dal3_tr_fc_backward_w": (_i, [vp, _i64, _i64, _i, vp, _i64, vp, vp, vp, vp, vp, vp, vp, vp, _i, _i64, vp, vp, _i, vp, _i64, vp, _i64,
                                   vp, vp]),
    "dal3_tr_bnbwd_apply": (_i, [vp, _i64, _i, _i64, vp, _i64, vp, vp, _i64, vp, vp, vp, vp, vp, vp, vp, vp, _i64, vp]),
    "dal3_tr_bnbwd_apply_segsum_workspace_bytes": (_sz, [_i64, _i]),
    "dal3_tr_bnbwd_apply_segsum": (_i, [vp, _i64, _i, _i64, vp, _i64, vp, vp, vp, vp, vp, vp, vp, vp, _i64, _i64, vp, vp, _sz, vp]),
    "dal3_tr_wgrad_workspace_bytes": (_sz, [_i64, _i, _i]),
    "dal3_tr_wgrad": (_i, [vp, _i64, vp, _i64, vp, vp, _i, _i64, _i, _i, vp, _sz, vp, vp]),
    "dal3_tr_wgrad_final_many": (_i, [C.POINTER(WgradPart), _i, vp]),
    "dal3_tr_wgrad_x3_workspace_bytes": (_sz, [_i64, _i, _i]),
    "dal3_tr_wgrad_x3": (_i, [vp, _i64, vp, _i64, vp, vp, _i, vp, _i64, _i, _i, vp, _sz, vp, vp]),
    "dal3_tr_segmax": (_i, [vp, _i64, _i64, _i, vp, vp, vp, vp, _i64, vp, _sz, vp]),
    "dal3_tr_act_dropout": (_i, [vp, _i64, _i, _i64, vp, vp, _i, vp, _i64, _u64, vp, C.c_float, vp, _i64, vp]),
    "dal3_tr_linear_pool_workspace_bytes": (_sz, [_i, _i, _i64]),
    "dal3_tr_linear_pool": (_i, [vp, _i64, _i, _i64, vp, vp, _i, vp, _i64, vp, vp, vp, _i64, _i, vp, vp, vp, _sz, vp]),
    "dal3_tr_linear_pool_x3": (_i, [vp, _i64, _i, _i64, vp, vp, _i, vp, _i64, vp, vp, vp, _i64, _i, vp, vp, vp, _sz, vp]),
    "dal3_tr_linear_pool_x3_ok": (_i, [_i64, _i, _i64, _i]),
    "dal3_tr_segsum": (_i, [vp, _i64, _i64, _i, vp, _i64, vp]),
    "dal3_maxpool_n": (_i, [vp, _i64, _i64, vp, vp]),
    "dal3_maxpool_n_dtype": (_i, [vp, _i, _i64, _i64, vp, vp]),
    "dal3_shared_mlp_layer": (_i, [C.POINTER(Layer), _i, BCN, _i, _i, vp, vp, _sz, vp]),
    "dal3_shared_mlp_layer_workspace_bytes": (_sz, [_i, _i]),
    "dal3_static_workspace_bytes": (_sz, [_i, _i, _i]),
    "dal3_static_forward": (_i, [C.POINTER(StaticArgs), _i, vp]),
    "dal3_dynamic_workspace_bytes": (_sz, [_i, _i, _i]),
    "dal3_dynamic_forward": (_i, [C.POINTER(DynamicArgs), _i, vp]),
}

_lib = None


def lib():
    """The loaded library; raises (loudly) if it was never built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build the HIP library first "
                "(`make -C 3dal_pytorch_amd/csrc` or __graft_entry__.build()). "
                "There is no CPU/eager fallback for the eval-mode forward.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        from . import arch
        lib_mean = [handle.dal3_mean_size()[i] for i in range(9)]
        want = [C.c_float(v).value for row in arch.MEAN_SIZE for v in row]
        if lib_mean != want:                                # two copies of one table: they may not drift apart
            raise RuntimeError(f"lib3dal_hip.so was built with MEAN_SIZE {lib_mean}, arch.MEAN_SIZE is {want}: rebuild the library")
        _lib = handle
    return _lib


def check(rc):
    if rc != OK:
        raise RuntimeError(f"lib3dal_hip error {rc}: {lib().dal3_last_error().decode()}")


def stream():
    return vp(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return vp(t.data_ptr()) if t is not None else vp(None)


def require_gpu(t, what):
    if not t.is_cuda:
        raise RuntimeError(
            f"{what} lives on {t.device}: the eval-mode forward runs only on an MI355X through "
            "lib3dal_hip.so (no CPU fallback). Move the model and its inputs to the GPU.")


STORAGE = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}
BCN_NO_SMALL_JOB_KERNELS, BCN_NO_WORKLIST, BCN_NO_LDS_SAMPLER = 1, 2, 4
# dal3_bcn.flags of every view bcn() builds (include/dal3.h, DAL3_BCN_*). 0 in normal use; A/B measurements and the
# tests that pin "both kernel families give the same bits" set it around a call (binding-side, the library has no switch).
DISPATCH_FLAGS = 0


def bcn(t):
    """logical (B,C,N) tensor (fp32, or bf16 / fp16 storage read in place) with arbitrary element strides -> dal3_bcn
    (no copy)."""
    assert t.dim() == 3 and t.dtype in STORAGE
    sb, sc, sn = t.stride()
    return BCN(ptr(t), sb, sc, sn, STORAGE[t.dtype], DISPATCH_FLAGS)


def layer_struct(conv, bn):
    """nn.Conv1d(k=1)/nn.Linear (+ optional nn.BatchNorm1d) -> dal3_layer of raw device pointers."""
    w = conv.weight
    c_out, c_in = w.shape[0], w.shape[1]
    if not w.is_contiguous() or w.dtype != torch.float32:
        raise RuntimeError("weights must be contiguous fp32")
    L = Layer(ptr(w), ptr(conv.bias), None, None, None, None, c_in, c_out)
    if bn is not None:
        L.bn_weight, L.bn_bias = ptr(bn.weight), ptr(bn.bias)
        L.bn_mean, L.bn_var = ptr(bn.running_mean), ptr(bn.running_var)
    return L


F16_MAX = 65504.0


def check_f16x3_range(pairs):
    """DAL3_F16X3 (include/dal3.h): a folded weight beyond fp16's largest finite value cannot be split into two halves.
    The shared-MLP kernels are built without NaN semantics (a NaN weight is not guaranteed to reach the outputs), so the
    range is checked HERE, once per packing (one device->host sync per weight change, not per forward), and refused."""
    with torch.no_grad():
        worst = None
        for conv, bn in pairs:
            w = conv.weight.detach().reshape(conv.weight.shape[0], -1).abs().amax(1)
            if bn is not None:
                w = w * (bn.weight.detach() / torch.sqrt(bn.running_var.detach() + 1e-5)).abs()
            m = w.max()
            worst = m if worst is None else torch.maximum(worst, m)
        if worst is not None and not float(worst) < F16_MAX:
            raise ValueError(f"precision 'f16x3': a folded weight of magnitude {float(worst):.3g} is beyond fp16's range "
                             f"({F16_MAX:g}); this head cannot run on the f16x3 kernels (use precision 'fp32')")


def pack(head_kind, pairs, device, dtype=F32):
    """pairs: [(conv_or_linear, bn_or_None), ...] in forward order -> packed uint8 device tensor."""
    if dtype == F16X3:
        check_f16x3_range(pairs)
    arr = (Layer * len(pairs))(*[layer_struct(c, b) for c, b in pairs])
    need = _sz(0)
    check(lib().dal3_pack_weights(head_kind, arr, len(pairs), dtype, None, C.byref(need), None))
    buf = torch.empty(need.value, dtype=torch.uint8, device=device)
    assert buf.data_ptr() % 256 == 0
    check(lib().dal3_pack_weights(head_kind, arr, len(pairs), dtype, ptr(buf), C.byref(need), stream()))
    return buf
