"""Layer tables of the Frustum-PointNet auto-labeling heads.

One place that states every layer shape on the hot path, so that the nn.Module
mirrors, the weight packer, the synthetic-weight generator and the tests agree.

Shapes restate the reference constructors:
  static  ins_seg   tools/static_model.py:241-269   box_est  :298-318
  dynamic ins_seg   tools/dynamic_model.py:157-185  point_emb :214-232
          box_emb   :251-269                         box_est  :288-298
"""

NUM_HEADING_BIN = 12          # static_model.py:12 / dynamic_model.py:12
NUM_SIZE_CLUSTER = 3          # static_model.py:13 / dynamic_model.py:13
NUM_OBJECT_POINT = 512        # static_model.py:14 / dynamic_model.py:14
NUM_FRAME = 5                 # dynamic_model.py:16
BOX_DIM = 3 + NUM_HEADING_BIN * 2 + NUM_SIZE_CLUSTER * 4      # 39 (not the docstring's 59)
MEAN_SIZE = ((4.8, 1.8, 1.5), (10.0, 2.6, 3.2), (2.0, 1.0, 1.6))  # static_model.py:17-21
BN_EPS = 1e-5                 # nn.BatchNorm1d default


def ins_seg_layers(n_channel):
    """(conv, bn, c_in, c_out) in forward order; bn None => no BN / no ReLU."""
    return [
        ("conv1", "bn1", n_channel, 64),
        ("conv2", "bn2", 64, 64),
        ("conv3", "bn3", 64, 64),
        ("conv4", "bn4", 64, 128),
        ("conv5", "bn5", 128, 1024),
        ("dconv1", "dbn1", 1088, 512),
        ("dconv2", "dbn2", 512, 256),
        ("dconv3", "dbn3", 256, 128),
        ("dconv4", "dbn4", 128, 128),
        ("dconv5", None, 128, 2),
    ]


# per-point shared MLP (Conv1d k=1) followed by FC layers (Linear)
STATIC_BOX_EST = {
    "convs": [("conv1", "bn1", 3, 128), ("conv2", "bn2", 128, 128),
              ("conv3", "bn3", 128, 256), ("conv4", "bn4", 256, 512)],
    "fcs": [("fc1", "fcbn1", 512, 512), ("fc2", "fcbn2", 512, 256), ("fc3", None, 256, BOX_DIM)],
}
POINT_EMB = {
    "convs": [("conv1", "bn1", 4, 64), ("conv2", "bn2", 64, 128),
              ("conv3", "bn3", 128, 256), ("conv4", "bn4", 256, 512)],
    "fcs": [("fc1", "fcbn1", 512, 512), ("fc2", "fcbn2", 512, 256)],
}
BOX_EMB = {
    "convs": [("conv1", "bn1", 8, 64), ("conv2", "bn2", 64, 64),
              ("conv3", "bn3", 64, 128), ("conv4", "bn4", 128, 512)],
    "fcs": [("fc1", "fcbn1", 512, 128), ("fc2", "fcbn2", 128, 128)],
}
DYNAMIC_BOX_EST = {
    "convs": [],
    "fcs": [("fc1", "fcbn1", 384, 128), ("fc2", "fcbn2", 128, 128), ("fc3", None, 128, BOX_DIM)],
}


def module_param_specs(kind, n_channel=None):
    """Ordered list of (state_dict key, shape) for one sub-module, in the order
    torch registers them for the reference classes (all convs/fcs first, then all
    BNs; see the constructors cited in the module docstring)."""
    if kind == "ins_seg":
        layers = ins_seg_layers(n_channel)
        enc, dec = layers[:5], layers[5:]
        out = []
        for name, _, ci, co in enc:
            out += [(f"{name}.weight", (co, ci, 1)), (f"{name}.bias", (co,))]
        for _, bn, _, co in enc:
            out += _bn_specs(bn, co)
        for name, _, ci, co in dec:
            out += [(f"{name}.weight", (co, ci, 1)), (f"{name}.bias", (co,))]
        for _, bn, _, co in dec:
            if bn:
                out += _bn_specs(bn, co)
        return out
    table = {"static_box_est": STATIC_BOX_EST, "point_emb": POINT_EMB,
             "box_emb": BOX_EMB, "dynamic_box_est": DYNAMIC_BOX_EST}[kind]
    out = []
    for name, _, ci, co in table["convs"]:
        out += [(f"{name}.weight", (co, ci, 1)), (f"{name}.bias", (co,))]
    for _, bn, _, co in table["convs"]:
        out += _bn_specs(bn, co)
    for name, _, ci, co in table["fcs"]:
        out += [(f"{name}.weight", (co, ci)), (f"{name}.bias", (co,))]
    for _, bn, _, co in table["fcs"]:
        if bn:
            out += _bn_specs(bn, co)
    return out


def _bn_specs(bn, c):
    return [(f"{bn}.weight", (c,)), (f"{bn}.bias", (c,)), (f"{bn}.running_mean", (c,)),
            (f"{bn}.running_var", (c,)), (f"{bn}.num_batches_tracked", ())]


def model_param_specs(model):
    """Full state_dict key/shape list for 'static_one' | 'static_two' | 'dynamic'."""
    def pref(p, specs):
        return [(f"{p}.{k}", s) for k, s in specs]
    if model == "static_one":
        return pref("ins_seg", module_param_specs("ins_seg", 3)) + \
            pref("box_est", module_param_specs("static_box_est"))
    if model == "static_two":
        return pref("ins_seg", module_param_specs("ins_seg", 3)) + \
            pref("box_est_one", module_param_specs("static_box_est")) + \
            pref("box_est_two", module_param_specs("static_box_est"))
    if model == "dynamic":
        return pref("ins_seg", module_param_specs("ins_seg", 4)) + \
            pref("point_emb", module_param_specs("point_emb")) + \
            pref("box_emb", module_param_specs("box_emb")) + \
            pref("box_est", module_param_specs("dynamic_box_est"))
    raise ValueError(model)


# Algorithmic work per item in MAC (SURVEY.md 8(a) "algorithmic" column: dconv1 split into a
# per-point 64->512 part and a per-crop 1024->512 part).
def ins_seg_mac(n_channel, n_pts):
    per_pt = n_channel * 64 + 64 * 64 + 64 * 64 + 64 * 128 + 128 * 1024 \
        + 64 * 512 + 512 * 256 + 256 * 128 + 128 * 128 + 128 * 2
    return per_pt * n_pts + 1024 * 512


def head_mac(table, n_pts):
    per_pt = sum(ci * co for _, _, ci, co in table["convs"])
    per_item = sum(ci * co for _, _, ci, co in table["fcs"])
    return per_pt * n_pts + per_item


def static_one_flop(n_pts, m=NUM_OBJECT_POINT):
    return 2 * (ins_seg_mac(3, n_pts) + head_mac(STATIC_BOX_EST, m))


def static_two_flop(n_pts, m=NUM_OBJECT_POINT):
    return 2 * (ins_seg_mac(3, n_pts) + 2 * head_mac(STATIC_BOX_EST, m))


def dynamic_flop(n_pts=NUM_FRAME * 1024, m=NUM_FRAME * NUM_OBJECT_POINT, n_box=101):
    return 2 * (ins_seg_mac(4, n_pts) + head_mac(POINT_EMB, m) + head_mac(BOX_EMB, n_box)
                + head_mac(DYNAMIC_BOX_EST, 0))


# Executed work (what the kernels really issue), reported NEXT to the algorithmic figures above and never in their
# place: the first layer's K is padded to the MFMA's k-step (3 -> 4), the decode kernel recomputes
# conv1-conv2 instead of reading a 1 GB activation back, ragged N is padded to the wave's tile — and the point heads
# SKIP object points that are copies (device sampler: only the first min(count, M) sampled points are distinct).
def _pad(n, g):
    return (n + g - 1) // g * g


def ins_seg_encode_mac(n_channel, executed=False):
    c = _pad(n_channel, 2) if executed else n_channel
    return c * 64 + 64 * 64 + 64 * 64 + 64 * 128 + 128 * 1024


def ins_seg_decode_mac(n_channel, executed=False):
    mac = 64 * 512 + 512 * 256 + 256 * 128 + 128 * 128 + 128 * 2
    if executed:
        mac += _pad(n_channel, 2) * 64 + 64 * 64
    return mac


def head_point_mac(table, executed=False):
    convs = table["convs"]
    mac = sum(ci * co for _, _, ci, co in convs[1:])
    c0 = convs[0][2]
    return mac + (_pad(c0, 2) if executed else c0) * convs[0][3]


def head_executed_points(counts, m, granule):
    """object points a point head really processes given each item's number of segmented points: the first
    min(count, m) (at least one) rounded up to the skip granule (32 points per wave in the fp32 kernel, 256 per
    workgroup in the 16-bit one), never more than the padded m"""
    total = 0
    for c in counts:
        d = min(max(int(c), 1), m)
        total += min(_pad(d, granule), _pad(m, granule))
    return total
