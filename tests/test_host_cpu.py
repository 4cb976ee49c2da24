"""CPU-side checks: the C-ABI library loads and exports every symbol include/dal3.h declares,
the nn.Module mirrors carry the reference's state_dict key set, host-side sampler/sharding logic.
No GPU compute here."""
import ctypes
import importlib
import os
import re

import numpy as np
import pytest
import torch

from _common import ROOT, arch, golden, rel_err, static_case, synth
from oracle import ref_heads as R

graft_entry = importlib.import_module("__graft_entry__")

hip = importlib.import_module("3dal_pytorch_amd._hip")
static_model = importlib.import_module("3dal_pytorch_amd.static_model")
dynamic_model = importlib.import_module("3dal_pytorch_amd.dynamic_model")
heads = importlib.import_module("3dal_pytorch_amd._heads")


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "dal3.h")).read()
    declared = set(re.findall(r"\b(dal3_[a-z0-9_]+)\s*\(", header))
    assert declared == set(hip.SIGNATURES), declared ^ set(hip.SIGNATURES)
    lib = ctypes.CDLL(hip.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert hip.lib().dal3_version() == graft_entry.header_version()


def test_build_entry_runs_and_agrees_with_the_header():
    """The driver's own door: __graft_entry__.build() (make + import + version check against include/dal3.h). With
    the objects already built, make is a no-op; from a clean tree this is the full gfx950 cross-compile."""
    graft_entry.build()
    assert graft_entry.header_version() >= 120


def test_argument_errors_are_reported_without_a_gpu():
    lib = hip.lib()
    n = ctypes.c_size_t(0)
    assert lib.dal3_pack_weights(99, None, 0, hip.F32, None, ctypes.byref(n), None) == hip.EINVAL
    assert b"head_kind" in lib.dal3_last_error()
    assert lib.dal3_pack_weights(hip.HEAD_INS_SEG, None, 0, hip.F32, None, ctypes.byref(n), None) == 0
    assert n.value > 3_000_000                       # ~3.6 MB of folded fp32 weights
    assert lib.dal3_static_forward(None, 3, None) == hip.EINVAL
    assert lib.dal3_static_workspace_bytes(8, 1024, 0) > 8 * 1024 * 4


def test_batches_beyond_the_32_bit_grid_and_offset_limits_are_refused_without_a_gpu():
    """B * ceil(N / 32) >= 2^31 used to become a truncated `(unsigned)` grid. Every entry that takes
    (B, N) / (B, M) now returns DAL3_EINVAL before anything is carved or launched (so this runs without a GPU; the
    pointers below are never dereferenced): too many tiles, too many items, too many points per item — and the
    largest job inside the limits still gets past this check (it then fails on its NULL workspace / null pointer)."""
    lib = hip.lib()
    header = open(os.path.join(ROOT, "include", "dal3.h")).read()
    lim = {k: int(v) for k, v in re.findall(r"#define\s+(DAL3_MAX_[A-Z_]+)\s+(\d+)", header)}
    assert lim == {"DAL3_MAX_ITEMS": 1 << 24, "DAL3_MAX_POINTS_PER_ITEM": 1 << 24, "DAL3_MAX_TILES": (1 << 31) - 1}
    fake = ctypes.c_void_p(0x1000)                       # non-NULL, never read: every call returns before a launch
    x = hip.BCN(fake, 1, 1, 1, hip.F32, 0)
    big_B, big_N = 1 << 24, 1 << 13                      # 2^24 items x 2^13 points = 2^32 tiles of 32 points
    calls = {
        "ins_seg_forward": lambda B, N: lib.dal3_ins_seg_forward(fake, hip.F32, 3, x, B, N, fake, fake, None, fake, 1 << 40, None),
        "ins_seg_encode": lambda B, N: lib.dal3_ins_seg_encode(fake, hip.F32, 3, x, B, N, fake, None),
        "ins_seg_decode": lambda B, N: lib.dal3_ins_seg_decode(fake, hip.F32, 3, x, B, N, fake, fake, fake, None),
        "segment_counts": lambda B, N: lib.dal3_segment_counts(fake, B, N, fake, None),
        "mask_compact_sample": lambda B, N: lib.dal3_mask_compact_sample(fake, x, B, N, 3, 512, hip.SAMPLER_DEVICE, None, 0, 0, fake,
                                                                         fake, fake, fake, 1 << 40, None),
        "point_head_forward": lambda B, N: lib.dal3_point_head_forward(hip.HEAD_STATIC_BOX_EST, fake, hip.F32, x, B, N, fake, 39, fake,
                                                                       1 << 40, None),
        "point_head_pool": lambda B, N: lib.dal3_point_head_pool(hip.HEAD_STATIC_BOX_EST, fake, hip.F32, x, B, N, None, fake, None, 0, None),
    }
    for name, call in calls.items():
        for B, N, word in ((big_B, big_N, b"DAL3_MAX_TILES"), (big_B + 1, 32, b"DAL3_MAX_ITEMS"), (2, (1 << 24) + 1, b"DAL3_MAX_POINTS_PER_ITEM"),
                           (0, 32, b"positive"), (4, -1, b"positive")):
            assert call(B, N) == hip.EINVAL, (name, B, N)
            assert word in lib.dal3_last_error(), (name, B, N, lib.dal3_last_error())
    sa = hip.StaticArgs()
    sa.B, sa.N, sa.workspace = big_B, big_N, fake
    assert lib.dal3_static_forward(ctypes.byref(sa), hip.PHASE_ALL, None) == hip.EINVAL and b"DAL3_MAX_TILES" in lib.dal3_last_error()
    da = hip.DynamicArgs()
    da.B, da.N, da.n_box, da.workspace = 4, 5120, (1 << 24) + 1, fake
    assert lib.dal3_dynamic_forward(ctypes.byref(da), hip.PHASE_ALL, None) == hip.EINVAL and b"n_box" in lib.dal3_last_error()
    # inside the limits the check lets the call through to the next one: the workspace is too small (no launch either)
    assert lib.dal3_ins_seg_forward(fake, hip.F32, 3, x, 1 << 16, 1 << 13, fake, fake, None, fake, 16, None) == hip.EWORKSPACE
    # dal3_crop_starts_capped: a negative capacity is an argument error
    assert lib.dal3_crop_starts_capped(fake, None, 4, fake, fake, -1, None) == hip.EINVAL


@pytest.mark.parametrize("kind,ctor", [
    ("static_one", lambda: static_model.StaticModelOneBoxEst(3, 3)),
    ("static_two", lambda: static_model.StaticModelTwoBoxEst(3, 3)),
    ("dynamic", lambda: dynamic_model.DynamicModel(3, 4)),
])
def test_state_dict_keys_match_reference(kind, ctor):
    g = golden("state_dict_keys")
    m = ctor()
    sd = m.state_dict()
    assert list(sd.keys()) == list(g[kind + "_keys"])
    assert [str(tuple(v.shape)) for v in sd.values()] == list(g[kind + "_shapes"])
    # a reference-shaped checkpoint loads strictly
    m.load_state_dict({k: torch.as_tensor(v) for k, v in synth.state_dict(kind).items()}, strict=True)


def test_model_attributes_and_constants():
    one, two, dyn = (static_model.StaticModelOneBoxEst(), static_model.StaticModelTwoBoxEst(),
                     dynamic_model.DynamicModel())
    assert one.name == "one_box_est" and two.name == "two_box_est"
    assert (dyn.r, dyn.s) == (2, 50)
    assert static_model.NUM_POINT == 4096 and dynamic_model.NUM_POINT == 1024
    assert static_model.NUM_OBJECT_POINT == 512 and dynamic_model.NUM_FRAME == 5
    assert np.allclose(static_model.MEAN_SIZE_ARR, R.MEAN_SIZE_ARR)


def test_eval_forward_refuses_cpu_tensors():
    m = static_model.StaticModelOneBoxEst().eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 64), torch.zeros(1, 7), torch.zeros(1, 7))


def test_submodule_composites_equal_oracle_in_eval():
    """The stock-torch composites used in train mode share the parameter containers; with BN in
    eval mode they must equal the oracle (guards the layer wiring of _heads.py)."""
    g = golden("static_one_b1_n512")
    sd, pts, init, _ = static_case("static_one", 1, 512, g)
    m = static_model.StaticModelOneBoxEst()
    m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    m.eval()
    with torch.no_grad():
        logits = m.ins_seg(pts)
        box = m.box_est(torch.from_numpy(g["object_pts"]))
    assert rel_err(logits.numpy(), g["logits"]) < 1e-6
    assert rel_err(box.numpy(), g["box_pred"]) < 1e-6


def test_numpy_choice_follows_reference_stream():
    g = golden("gather_rng")
    np.random.seed(12345)
    ch = heads.numpy_choice(g["counts"], 512)
    assert np.random.randint(0, 1 << 30) == int(g["next_draw"])
    for row, c in enumerate(g["counts"]):
        if c:
            pos = np.nonzero(g["mask"][row])[0]
            assert np.array_equal(pos[ch[row]], g["indices"][row])


def test_flop_model_matches_survey():
    assert arch.static_one_flop(1024) == 2 * 462_956_288
    assert arch.static_two_flop(1024) == 2 * 555_830_784
    assert arch.static_one_flop(4096) == 2 * 1_571_628_800
    assert arch.dynamic_flop() == 2 * 2_298_154_368


def test_dropin_modules_resolve():
    """`from static_model import ...` as the reference's drivers write it"""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); "
            "from static_model import StaticModelOneBoxEst, StaticModelTwoBoxEst, NUM_POINT, MEAN_SIZE_ARR; "
            "from dynamic_model import DynamicModel, NUM_FRAME; "
            "m = StaticModelTwoBoxEst(3, 3); assert m.name == 'two_box_est' and NUM_POINT == 4096; "
            "assert DynamicModel(3, 4).s == 50 and NUM_FRAME == 5; print('ok')"
            % os.path.join(ROOT, "3dal_pytorch_amd", "dropin"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp")
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr


def test_header_is_plain_c():
    """include/dal3.h is the boundary for non-Python hosts: it must compile as C99 on its own"""
    import shutil
    import subprocess
    import tempfile
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "t.c")
        with open(src, "w") as f:
            f.write('#include "dal3.h"\nint main(void) { dal3_static_args a = {0}; dal3_dynamic_args d = {0}; (void)a; (void)d; '
                    'return dal3_version() < 0; }\n')
        out = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I",
                              os.path.join(ROOT, "include"), src], capture_output=True, text=True)
        assert out.returncode == 0, out.stderr


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """no CPU/eager fallback: with the shared library absent every entry point raises, naming the build step"""
    monkeypatch.setattr(hip, "_lib", None)
    monkeypatch.setattr(hip, "LIB_PATH", str(tmp_path / "lib3dal_hip.so"))
    with pytest.raises(RuntimeError, match="build the HIP library"):
        hip.lib()
    with pytest.raises(RuntimeError, match="no CPU/eager fallback"):
        hip.lib().dal3_last_error()


def test_oracle_is_test_infrastructure_only():
    """nothing shipped or measured routes through oracle/: the package, the tools and the examples never name it in
    code; bench.py imports it in its cpu_baseline leg only, __graft_entry__ in build() (import check) and smoke()"""
    import ast
    import glob
    pat = re.compile(r"\boracle\b")

    def code_mentions(path):
        """names of the top-level functions (None = module level) whose code imports the oracle package"""
        tree = ast.parse(open(path).read())
        hits = []

        def visit(node, owner):
            for n in ast.iter_child_nodes(node):
                here = n.name if (isinstance(n, ast.FunctionDef) and owner is None) else owner
                names = []
                if isinstance(n, ast.Import):
                    names = [a.name for a in n.names]
                elif isinstance(n, ast.ImportFrom):
                    names = [n.module or ""]
                elif isinstance(n, ast.Call) and n.args and isinstance(n.args[0], ast.Constant) \
                        and isinstance(n.args[0].value, str) and getattr(n.func, "attr", "") == "import_module":
                    names = [n.args[0].value]
                if any(pat.match(x.split(".")[0]) for x in names):
                    hits.append(here)
                visit(n, here)
        visit(tree, None)
        return hits
    for path in glob.glob(os.path.join(ROOT, "3dal_pytorch_amd", "**", "*.py"), recursive=True) + \
            glob.glob(os.path.join(ROOT, "tools", "*.py")):
        assert not code_mentions(path), path
    assert set(code_mentions(os.path.join(ROOT, "bench.py"))) == {"cpu_baseline"}
    assert set(code_mentions(os.path.join(ROOT, "__graft_entry__.py"))) <= {"build", "smoke"}


def test_packed_cache_stamps_follow_parameter_identity():
    """host logic of the derived weight cache: the stamp is a per-tensor tuple (pointer, version, shape, device);
    in-place ops and load_state_dict change it, a write through .data does not (hence invalidate_packed())"""
    heads = importlib.import_module("3dal_pytorch_amd._heads")
    sm = importlib.import_module("3dal_pytorch_amd.static_model")
    m = sm.StaticModelOneBoxEst()
    cache = heads.PackedCache()
    s0 = cache._stamp_of(m.box_est, 0)
    assert s0 == cache._stamp_of(m.box_est, 0) and s0 != cache._stamp_of(m.box_est, 1)
    with torch.no_grad():
        m.box_est.fc3.bias.add_(1.0)
    s1 = cache._stamp_of(m.box_est, 0)
    assert s1 != s0
    m.box_est.fc3.bias.data.add_(1.0)
    assert cache._stamp_of(m.box_est, 0) == s1            # invisible: the documented case for invalidate_packed()
    m.load_state_dict(m.state_dict())
    assert cache._stamp_of(m.box_est, 0) != s1
    # a submodule replaced AFTER the tensor list was built at the current epoch: the new conv is constructed
    # first (its parameters bump the epoch), a stamp is taken (list rebuilt with the OLD conv), then it is assigned
    new_fc = torch.nn.Linear(m.box_est.fc3.in_features, m.box_est.fc3.out_features)
    s2 = cache._stamp_of(m.box_est, 0)
    m.box_est.fc3 = new_fc
    assert cache._stamp_of(m.box_est, 0) != s2
    # the hooks empty the model's own cache
    m._cache._stamp["x"], m._cache._blob["x"], m._cache._src["x"] = (1,), None, (m.box_est, 0)
    m.load_state_dict(m.state_dict())
    assert not m._cache._stamp and not m._cache._blob
    m._cache._stamp["x"] = (1,)
    m.eval()
    assert not m._cache._stamp
    m._cache._stamp["x"] = (1,)
    m.float()
    assert not m._cache._stamp
    m._cache._stamp["x"] = (1,)
    m.invalidate_packed()
    assert not m._cache._stamp


def test_box_gatherer_without_a_process_group_passes_the_boxes_through():
    """one rank, no torch.distributed: submit/collect hand back the very tensors, in order, `keep` in flight, and a
    third submit without a collect is refused (two slots)"""
    import importlib
    dist = importlib.import_module("3dal_pytorch_amd.dist")
    g = dist.BoxGatherer(5, torch.device("cpu"))
    assert not g.active
    a, b = torch.arange(35.0).view(5, 7), torch.arange(35.0, 70.0).view(5, 7)
    g.submit(a)
    assert g.collect(keep=1) is None
    g.submit(b)
    got = g.collect(keep=1)
    assert got.data_ptr() == a.data_ptr() and torch.equal(got, a)
    g.submit(a)
    with pytest.raises(RuntimeError):
        g.submit(b)
    assert torch.equal(g.collect(keep=0), b) and torch.equal(g.collect(keep=0), a) and g.collect(keep=0) is None


def test_f16x3_split_is_as_exact_as_fp32_arithmetic_in_emulation():
    """tests/f16x3_accuracy.py (every product of ins_seg formed as the named arithmetic forms it, logits against float64):
    the (hi, lo) fp16 split is within a factor of two of plain float32 — the premise of the f16x3 kernels — while one rounding
    to fp16 costs three digits and a bf16 split one and a half"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("f16x3_accuracy", os.path.join(ROOT, "tests", "f16x3_accuracy.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    e = mod.main(B=2, N=512)
    assert e["f16x3"] < 2 * e["f32"] and e["f16x3"] < 5e-6
    assert e["f16"] > 100 * e["f16x3"] and e["bf16x3"] > 5 * e["f16x3"]


def test_f16x3_training_kernels_shape_rules_are_host_side():
    """which calls take the f16x3 training kernels is decided on the host (no GPU needed): the rules include/dal3.h states"""
    lib = hip.lib()
    lay = lib.dal3_tr_linear_x3_layout                       # (M, c_in, seg, c_out, accumulate, has_act)
    assert lay(262144, 512, 0, 256, 0, 1) == 0x108 and lay(262144, 64, 4096, 512, 0, 1) == 0x108
    for bad in ((262144, 512, 0, 256, 1, 1), (262144, 512, 0, 128, 0, 1), (262144, 96, 0, 256, 0, 1), (262144 + 32, 512, 0, 256, 0, 1),
                (2048, 512, 0, 256, 0, 1), (262144, 64, 4000, 512, 0, 1), (262144, 2112, 0, 256, 0, 1)):
        assert lay(*bad) == 0, bad
    assert lib.dal3_tr_linear_pool_x3_ok(262144, 128, 4096, 1024) == 1 and lib.dal3_tr_linear_pool_x3_ok(262144, 128, 4000, 1024) == 0
    wg = lib.dal3_tr_wgrad_x3_workspace_bytes
    assert wg(262144, 256, 512) > 0 and wg(262144, 128, 256) > 0 and wg(262144, 128, 128) > 0 and wg(262144, 512, 64) > 0
    assert wg(262144, 64, 64) == 0 and wg(262144, 128, 64) == 0 and wg(4096, 256, 512) == 0
    # a slice's partial sums fit the workspace the function reports: slices x c_out x c_in floats, at most ~2 workgroups per CU
    assert wg(262144, 256, 512) % (256 * 512 * 4) == 0 and wg(262144, 256, 512) // (256 * 512 * 4) <= 1024
