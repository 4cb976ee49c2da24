"""The training forward's big layers on the f16x3 engine (dal3_train_x3.hip; model.precision = "f16x3" in train mode,
train.arithmetic("f16x3")): each kernel against float64 at the fp32 kernels' accuracy, the pooled layer's argmax against
the fp32 kernel's, and ONE WHOLE TRAINING STEP against the reference's own (the big fixtures of
test_gpu_train_reference.py: forward, criterion, gradients gated at the forward's decisions, BatchNorm statistics, Adam
step) with the f16x3 kernels in the forward."""
import importlib

import numpy as np
import pytest
import torch

import test_gpu_train_reference as R
from _common import build_model

hip = importlib.import_module("3dal_pytorch_amd._hip")
train = importlib.import_module("3dal_pytorch_amd.train")
pytestmark = pytest.mark.gpu


def _x3_linear(a, W, act, bias, seg):
    lib = hip.lib()
    M, ci = a.shape
    co = W.shape[0]
    lay = lib.dal3_tr_linear_x3_layout(M, ci, seg, co, 0, int(act is not None))
    assert lay == 0x108
    pk = torch.empty(lib.dal3_tr_linear_workspace_bytes(ci, co), dtype=torch.uint8, device="cuda")
    item = (hip.PackItem * 1)(hip.PackItem(hip.ptr(W), W.stride(0), 0, co, ci, lay, hip.ptr(pk)))
    hip.check(lib.dal3_tr_pack_many(item, 1, hip.stream()))
    z = torch.empty((M, co), device="cuda")
    sc, sh, relu = act if act is not None else (None, None, False)
    hip.check(lib.dal3_tr_linear_x3(hip.ptr(a), M, ci, a.stride(0), hip.ptr(sc), hip.ptr(sh), int(relu), hip.ptr(bias), seg, co,
                                    hip.ptr(z), z.stride(0), hip.ptr(pk), None, hip.stream()))
    return z


@pytest.mark.parametrize("M,ci,co,act,relu,seg,with_bias", [
    (8192, 512, 256, True, True, 0, True),          # dconv2
    (8192, 64, 512, True, True, 4096, True),        # dconv1's per-point part: a bias ROW per crop
    (4096, 64, 256, False, False, 0, False),        # two k-tiles: first and last pair are the same one
    (12288, 128, 768, True, False, 256, True),      # three output blocks, affine without ReLU, a row per group
    (70 * 256, 256, 512, True, True, 0, True),      # more groups than workgroups per block can be dealt evenly
    (4096, 2048, 256, True, True, 0, True),         # the widest input the LDS budget admits (64 k-tiles)
])
def test_linear_x3_against_float64(M, ci, co, act, relu, seg, with_bias):
    g = torch.Generator(device="cuda").manual_seed(M + ci + co)
    lda = ci + 8                                              # a row stride that is not the row length
    a = (torch.randn((M, lda), device="cuda", generator=g) * 1.5)[:, :ci]
    W = torch.randn((co, ci), device="cuda", generator=g) / ci ** 0.5
    sc = torch.rand(ci, device="cuda", generator=g) + 0.5
    sc[::5] *= -1.0
    sh = torch.randn(ci, device="cuda", generator=g) * 0.3
    n_row = M // seg if seg else 1
    bias = torch.randn((n_row, co), device="cuda", generator=g) if with_bias else None
    x = a.double()
    if act:
        x = x * sc.double() + sh.double()
        if relu:
            x = torch.relu(x)
    ref = x @ W.double().t()
    if with_bias:
        ref = ref + (bias.double().repeat_interleave(seg, 0) if seg else bias.double())
    z = _x3_linear(a, W, (sc, sh, relu) if act else None, bias, seg)
    z32 = train._linear(a, W, ci, ci, co, act=(sc, sh, relu) if act else None, bias=bias, seg=seg)
    rng = float(ref.abs().max())
    ex, e32 = float((z.double() - ref).abs().max()) / rng, float((z32.double() - ref).abs().max()) / rng
    assert ex < 2e-6, (ex, e32)                                 # (measured 2.5e-7 .. 7e-7; the fp32 kernel 3e-7 .. 1e-6)
    assert ex < 2 * e32 + 2e-7


def test_linear_x3_layout_rules():
    lib = hip.lib()
    ok = lambda *a: lib.dal3_tr_linear_x3_layout(*a)          # noqa: E731   (M, c_in, seg, c_out, accumulate, has_act)
    assert ok(262144, 512, 0, 256, 0, 1) == 0x108 and ok(262144, 64, 4096, 512, 0, 1) == 0x108
    assert ok(262144, 512, 0, 256, 1, 1) == 0                 # accumulate: the fp32 kernel
    assert ok(262144, 512, 0, 128, 0, 1) == 0                 # c_out % 256
    assert ok(262144, 96, 0, 256, 0, 1) == 0                  # c_in % 64
    assert ok(262144 + 32, 512, 0, 256, 0, 1) == 0            # M % 256
    assert ok(2048, 512, 0, 256, 0, 1) == 0                   # few rows: the small-M kernels
    assert ok(262144, 64, 4000, 512, 0, 1) == 0               # a group of 256 points must lie in one segment
    assert ok(262144, 2048, 0, 256, 0, 1) == 0x108 and ok(262144, 2112, 0, 256, 0, 1) == 0   # LDS: scale + shift beside ring and stages
    # a call that does not qualify is refused, not mis-computed
    a = torch.zeros((512, 64), device="cuda")
    z = torch.empty((512, 256), device="cuda")
    pk = torch.empty(lib.dal3_tr_linear_workspace_bytes(64, 256), dtype=torch.uint8, device="cuda")
    rc = lib.dal3_tr_linear_x3(hip.ptr(a), 512, 64, 64, None, None, 0, None, 0, 256, hip.ptr(z), 256, hip.ptr(pk), None, hip.stream())
    assert rc == hip.EINVAL if hasattr(hip, "EINVAL") else rc != 0


@pytest.mark.parametrize("M,ci,co,seg", [(16 * 4096, 128, 1024, 4096), (64 * 512, 256, 512, 512), (4096, 64, 256, 256)])
def test_linear_pool_x3_against_the_fp32_kernel(M, ci, co, seg):
    g = torch.Generator(device="cuda").manual_seed(co)
    a = torch.randn((M, ci), device="cuda", generator=g) * 1.5
    W = torch.randn((co, ci), device="cuda", generator=g) / ci ** 0.5
    b = torch.randn(co, device="cuda", generator=g) * 0.1
    sc = torch.rand(ci, device="cuda", generator=g) + 0.5
    sh = torch.randn(ci, device="cuda", generator=g) * 0.3

    class BN:
        scale = torch.rand(co, device="cuda", generator=g) + 0.5
        shift = torch.randn(co, device="cuda", generator=g) * 0.3
    BN.scale[::7] *= -1.0
    BN.shift[::11] = -50.0                                    # channels that ReLU clamps everywhere: max 0 at the FIRST point
    g32, a32 = train._linear_pool(a, (sc, sh, True), W, b, BN, seg)
    with train.arithmetic("f16x3"):
        assert hip.lib().dal3_tr_linear_pool_x3_ok(M, ci, seg, co)
        gx, ax = train._linear_pool(a, (sc, sh, True), W, b, BN, seg)
    n_seg = M // seg
    x = torch.relu(a.double() * sc.double() + sh.double())
    y = torch.relu((x @ W.double().t() + b.double()) * BN.scale.double() + BN.shift.double()).view(n_seg, seg, co)
    ref, rarg = y.max(1)
    rng = float(ref.abs().max())
    assert float((gx.double() - ref).abs().max()) / rng < 2e-6
    clamped = ref == 0
    assert bool((ax[clamped] == 0).all()) and bool((gx[clamped] == 0).all())
    # the argmax may differ from float64's only between points whose values tie within the arithmetic's error
    differ = ax.long() != rarg
    assert float(differ.float().mean()) < 1e-3
    if bool(differ.any()):
        picked = y.gather(1, ax.long()[:, None, :].clamp(0, seg - 1))[:, 0]
        assert float((picked - ref)[differ].abs().max()) / rng < 2e-6
    assert float((a32.long() != ax.long()).float().mean()) < 1e-3


@pytest.fixture
def x3_train(monkeypatch, tmp_path):
    def bm(kind, sd, device="cuda"):
        m = build_model(kind, sd, device)
        m.precision = "f16x3"
        return m
    monkeypatch.setattr(R, "build_model", bm)
    monkeypatch.chdir(tmp_path)                               # (the borrowed test writes its table under ./gpurun_out when there is one)
    calls = {"x3": 0, "pool": 0}
    lib = hip.lib()
    real_lin, real_pool = lib.dal3_tr_linear_x3, lib.dal3_tr_linear_pool_x3

    class Counting:                                           # the f16x3 kernels really ran in the borrowed test's forward
        def __init__(self, fn, key):
            self.fn, self.key = fn, key

        def __call__(self, *a):
            calls[self.key] += 1
            return self.fn(*a)
    monkeypatch.setattr(lib, "dal3_tr_linear_x3", Counting(real_lin, "x3"), raising=False)
    monkeypatch.setattr(lib, "dal3_tr_linear_pool_x3", Counting(real_pool, "pool"), raising=False)
    return calls


@pytest.mark.parametrize("kind", ["static_one_big", "static_two_big", "dynamic_big"])
def test_one_training_step_matches_the_reference_f16x3(x3_train, kind):
    R.test_one_training_step_matches_the_reference(kind)
    assert x3_train["x3"] >= 2 and x3_train["pool"] >= 1, x3_train


def test_x3_training_kernels_are_reproducible_bit_for_bit():
    """persistent workgroups, a two-slot LDS-DMA ring, per-wave LDS stages, queue drains at segment opens: a race or a
    missing wait shows up as a launch that differs from the first one"""
    g = torch.Generator(device="cuda").manual_seed(3)
    for M, ci, co, seg in ((32768, 512, 256, 0), (16384, 64, 512, 4096)):
        a = torch.randn((M, ci), device="cuda", generator=g)
        W = torch.randn((co, ci), device="cuda", generator=g) / ci ** 0.5
        sc, sh = torch.rand(ci, device="cuda", generator=g) + 0.5, torch.randn(ci, device="cuda", generator=g) * 0.3
        bias = torch.randn((M // seg if seg else 1, co), device="cuda", generator=g)
        first = _x3_linear(a, W, (sc, sh, True), bias, seg).clone()
        for _ in range(40):
            assert torch.equal(_x3_linear(a, W, (sc, sh, True), bias, seg), first)
    M, ci, co, seg = 32768, 128, 1024, 4096
    a = torch.randn((M, ci), device="cuda", generator=g)
    W = torch.randn((co, ci), device="cuda", generator=g) / ci ** 0.5
    b = torch.randn(co, device="cuda", generator=g) * 0.1
    sc, sh = torch.rand(ci, device="cuda", generator=g) + 0.5, torch.randn(ci, device="cuda", generator=g) * 0.3

    class BN:
        scale = torch.rand(co, device="cuda", generator=g) + 0.5
        shift = torch.randn(co, device="cuda", generator=g) * 0.3
    with train.arithmetic("f16x3"):
        g0, a0 = (t.clone() for t in train._linear_pool(a, (sc, sh, True), W, b, BN, seg))
        for _ in range(40):
            g1, a1 = train._linear_pool(a, (sc, sh, True), W, b, BN, seg)
            assert torch.equal(g1, g0) and torch.equal(a1, a0)


@pytest.mark.parametrize("M,K,co,mag", [(8192, 256, 512, 3e-6), (8192, 128, 256, 1e-9), (4096, 256, 256, 40.0)])
def test_dgrad_x3_with_the_operands_amax(M, K, co, mag):
    """a dgrad's operand (dz) lies far below fp16's range: dal3_tr_bnbwd_apply_amax leaves the bits of its largest |value| in
    64 device words, dal3_tr_linear_x3 scales by a power of two around the products. Here the word is filled the same way
    (atomicMax of the bit patterns == the maximum, for non-negative floats) and the result held to the fp32 kernel's
    accuracy, for magnitudes from 1e-9 to 40 and a log-uniform spread of eight decades inside the tensor."""
    lib = hip.lib()
    g = torch.Generator(device="cuda").manual_seed(K + co)
    spread = torch.exp(torch.rand((M, 1), device="cuda", generator=g) * -18.0)       # per-point scale, 1 .. 1.5e-8
    dz = torch.randn((M, K), device="cuda", generator=g) * spread * mag
    W = torch.randn((K, co), device="cuda", generator=g) / K ** 0.5                 # the layer's (c_out = K, c_in = co) weight
    ref = dz.double() @ W.double()
    z32 = train._linear(dz, W, W.stride(0), K, co, transpose=True)
    lay = lib.dal3_tr_linear_x3_layout(M, K, 0, co, 0, 0)
    assert lay == 0x108
    pk = torch.empty(lib.dal3_tr_linear_workspace_bytes(K, co), dtype=torch.uint8, device="cuda")
    item = (hip.PackItem * 1)(hip.PackItem(hip.ptr(W), W.stride(0), 1, co, K, lay, hip.ptr(pk)))
    hip.check(lib.dal3_tr_pack_many(item, 1, hip.stream()))
    amax = torch.zeros(64, dtype=torch.int32, device="cuda")
    amax[17] = dz.abs().max().reshape(1).view(torch.int32)[0]
    zx = train._linear(dz, W, W.stride(0), K, co, transpose=True, packed=train._X3Image(pk), amax=amax)
    rng = float(ref.abs().max())
    ex, e32 = float((zx.double() - ref).abs().max()) / rng, float((z32.double() - ref).abs().max()) / rng
    assert ex < 2e-6 and ex < 2 * e32 + 2e-7, (ex, e32)
    # rows two to four decades below the tensor's maximum keep their RELATIVE accuracy too (the split holds 22 bits down to
    # 2^-17 of the maximum, 11 bits down to 2^-28: include/dal3.h); further down only the absolute bound above holds
    small = (spread[:, 0] < 1e-2) & (spread[:, 0] > 1e-4)
    rel = ((zx.double() - ref)[small].abs().max(1)[0] / ref[small].abs().max(1)[0]).max()
    assert float(rel) < 1e-5, float(rel)


def test_bnbwd_apply_amax_is_the_maximum():
    """the word dal3_tr_bnbwd_apply_amax fills == the bit pattern of max |dz| of the dz it writes (and dz is what
    dal3_tr_bnbwd_apply writes)"""
    M, C = 8192, 256
    g = torch.Generator(device="cuda").manual_seed(5)
    z = torch.randn((M, C), device="cuda", generator=g)
    bn = train._BN(z, torch.rand(C, device="cuda", generator=g) + 0.5, torch.randn(C, device="cuda", generator=g), None, None, rows=M)
    da = torch.randn((M, C), device="cuda", generator=g) * 1e-6
    dz0 = bn.backward(z, da=da)[0]
    words = torch.zeros(64, dtype=torch.int32, device="cuda")
    dz1 = bn.backward(z, da=da, amax=words)[0]
    assert torch.equal(dz0, dz1)
    assert int(words.max()) == int(dz1.abs().max().view(torch.int32)) and int((words > 0).sum()) == 64


@pytest.mark.parametrize("M,co,ci,act,mag", [(16384, 256, 512, True, 3e-6), (16384, 128, 256, True, 1e-9), (8192, 128, 128, False, 2.0),
                                             (8192 + 32, 512, 64, True, 1e-5), (24576, 384, 256, True, 1e-4)])
def test_wgrad_x3_against_float64(M, co, ci, act, mag):
    """dW = dz^T act(a) with both operands split in fp16 halves (staged through LDS, read back transposed): the fp32 kernel's
    accuracy for dz at gradient magnitudes with its amax words; point counts that leave a ragged last slice; row strides
    that are not the row length"""
    lib = hip.lib()
    g = torch.Generator(device="cuda").manual_seed(co + ci + M)
    spread = torch.exp(torch.rand((M, 1), device="cuda", generator=g) * -9.0)
    dz = (torch.randn((M, co + 8), device="cuda", generator=g) * spread * mag)[:, :co]
    a = (torch.randn((M, ci + 4), device="cuda", generator=g) * 1.5)[:, :ci]
    sc = torch.rand(ci, device="cuda", generator=g) + 0.5
    sc[::3] *= -1.0
    sh = torch.randn(ci, device="cuda", generator=g) * 0.3
    x = torch.relu(a.double() * sc.double() + sh.double()) if act else a.double()
    ref = dz.double().t() @ x
    amax = torch.zeros(64, dtype=torch.int32, device="cuda")
    amax[40] = dz.abs().max().reshape(1).view(torch.int32)[0]
    assert lib.dal3_tr_wgrad_x3_workspace_bytes(M, co, ci) > 0
    wx = train._wgrad(dz, a, co, ci, (sc, sh, True) if act else None, amax=amax)
    w32 = train._wgrad(dz, a, co, ci, (sc, sh, True) if act else None)
    rng = float(ref.abs().max())
    ex, e32 = float((wx.double() - ref).abs().max()) / rng, float((w32.double() - ref).abs().max()) / rng
    assert ex < 2e-6 and ex < 2 * e32 + 2e-7, (ex, e32)
    for _ in range(10):                                       # (two LDS stages, one barrier per 16 points: reproducible bit for bit)
        assert torch.equal(train._wgrad(dz, a, co, ci, (sc, sh, True) if act else None, amax=amax), wx)


def test_wgrad_x3_shapes_that_do_not_qualify_take_the_fp32_kernel():
    lib = hip.lib()
    assert lib.dal3_tr_wgrad_x3_workspace_bytes(262144, 64, 64) == 0 and lib.dal3_tr_wgrad_x3_workspace_bytes(262144, 128, 64) == 0
    assert lib.dal3_tr_wgrad_x3_workspace_bytes(4096, 256, 512) == 0            # few points: the fp32 kernel's slices
    dz, a = torch.randn((8192, 64), device="cuda") * 1e-6, torch.randn((8192, 64), device="cuda")
    amax = torch.zeros(64, dtype=torch.int32, device="cuda")
    assert torch.equal(train._wgrad(dz, a, 64, 64, None, amax=amax), train._wgrad(dz, a, 64, 64, None))


def test_captured_f16x3_train_step_matches_eager_steps():
    """forward + criterion + backward + Adam as one hipGraph with the f16x3 kernels in it (linear, pooled, dgrad with its amax
    words zeroed inside the graph, wgrad): a warm-up step + one replay leave the parameters where two eager steps leave them"""
    graph = importlib.import_module("3dal_pytorch_amd.graph")
    losses = importlib.import_module("3dal_pytorch_amd.losses")
    from _common import synth
    B, N = 4, 4096                                           # 16,384 points: the f16x3 training kernels qualify
    p, i, g = (torch.from_numpy(a).cuda() for a in synth.static_crops(B, N, seed=6))
    pts = p.transpose(2, 1)
    gen = torch.Generator(device="cuda").manual_seed(1)
    labels = ((torch.rand((B, N), device="cuda", generator=gen) > 0.6).float(), torch.randn((B, 3), device="cuda", generator=gen),
              torch.randint(0, 12, (B,), device="cuda", generator=gen), 0.1 * torch.randn((B,), device="cuda", generator=gen),
              torch.randint(0, 3, (B,), device="cuda", generator=gen), 0.3 * torch.randn((B, 3), device="cuda", generator=gen))
    crit = losses.FrustumPointNetLossOneBoxEst()
    runs = {}
    lib = hip.lib()
    for mode in ("graph", "eager"):
        model = build_model("static_one", synth.state_dict("static_one", seed=6)).train()
        model.precision, model.sampler = "f16x3", "device"
        model.ins_seg.dropout.p = 0.0
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True)

        def step(p_, i_, g_, model=model):
            return crit(model(p_, i_, g_), *labels)["total_loss"]
        if mode == "graph":
            cap = graph.CapturedTrainStep(model, opt, step, pts, i, g, warmup=1)
            last = cap(pts, i, g)
        else:
            for _ in range(2):
                opt.zero_grad(set_to_none=True)
                last = step(pts, i, g)
                last.backward()
                opt.step()
        torch.cuda.synchronize()
        runs[mode] = (float(last.detach()), {k: v.detach().clone() for k, v in model.named_parameters()})
    assert lib.dal3_tr_linear_x3_layout(B * N, 512, 0, 256, 0, 1) == 0x108 and lib.dal3_tr_wgrad_x3_workspace_bytes(B * N, 256, 512) > 0
    assert abs(runs["graph"][0] - runs["eager"][0]) <= 1e-5 * abs(runs["eager"][0])
    for k, v in runs["eager"][1].items():
        assert torch.allclose(runs["graph"][1][k], v, rtol=1e-5, atol=1e-6), k
