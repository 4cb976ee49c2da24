"""The linear kernels that take a column reduction of their own output in the epilogue (round 4:
dal3_tr_linear_bn_stats, dal3_tr_linear_bnbwd_sums; csrc/dal3_train.hip `TrRed`) against the two-call sequences they
replace: the layer's output must be the SAME BITS (the tile is computed transposed — operands swapped — but every output
element is the same chain of fused multiply-adds), the statistics / BatchNorm-backward coefficients the same numbers up
to the order of the float64 additions (1e-6 of each vector's largest entry), the running statistics updated once. The
shapes are the training step's (64 crops x 4096 points) and smaller ones on both sides of the persistent kernel's
minimum size, with and without padding rows (then the library runs the two steps itself: return code 0)."""
import importlib

import pytest
import torch

hip = importlib.import_module("3dal_pytorch_amd._hip")
train = importlib.import_module("3dal_pytorch_amd.train")
pytestmark = pytest.mark.gpu


def _close(a, b, tol=1e-6):
    a, b = a.double(), b.double()
    return float((a - b).abs().max()) <= tol * max(float(b.abs().max()), 1e-30)


def _pack(W, c_in, c_out, transpose, M, seg, has_act):
    return train._prepack([(W, c_in, c_out, transpose, M, seg, False, has_act)], W.device)[0]


#                                    M      c_in c_out seg   rows     fused
@pytest.mark.parametrize("M,c_in,c_out,seg,rows,want", [
    (64 * 4096, 64, 512, 4096, None, 1),          # dconv1: per-crop bias rows, K = 64 (straight-line unit)
    (64 * 4096, 512, 256, 0, None, 1),            # dconv2: the K loop
    (64 * 4096, 128, 128, 0, None, 1),            # dconv4: K = 128
    (64 * 4096, 64, 64, 0, None, 1),              # conv2: the two-output-tile kernel, two waves per SIMD
    (64 * 4096, 64, 128, 0, None, 1),             # conv4
    (32 * 1024, 256, 512, 0, None, 1),            # the box head's last conv at a small batch (2 units per wave: the minimum)
    (64 * 4096, 128, 128, 0, 64 * 4096 - 7, 0),   # padding rows: two steps
    (4096, 64, 128, 0, None, 0),                  # too few units for the persistent kernel: two steps
    (64 * 4096, 256, 64, 0, None, 0),             # K > 64 into 64 channels: no fused instantiation
])
def test_linear_with_fused_batch_statistics_equals_linear_then_bn_stats(M, c_in, c_out, seg, rows, want):
    lib = hip.lib()
    gen = torch.Generator(device="cuda").manual_seed(M + 7 * c_in + c_out)
    a = torch.randn((M, c_in), device="cuda", generator=gen)
    sc = torch.rand(c_in, device="cuda", generator=gen) + 0.5
    sh = torch.randn(c_in, device="cuda", generator=gen) * 0.2
    W = torch.randn((c_out, c_in), device="cuda", generator=gen) / c_in ** 0.5
    bias = torch.randn((M // seg if seg else 1, c_out), device="cuda", generator=gen) * (2.0 if seg else 0.3) + 0.7
    gamma = torch.rand(c_out, device="cuda", generator=gen) + 0.5
    beta = torch.randn(c_out, device="cuda", generator=gen) * 0.3
    rows = rows if rows is not None else M
    pk = _pack(W, c_in, c_out, False, M, seg, True)
    assert pk is not None
    # the two-call reference
    z_ref = train._linear(a, W, c_in, c_in, c_out, act=(sc, sh, True), bias=bias, seg=seg, packed=pk)
    rm1, rv1 = torch.zeros(c_out, device="cuda"), torch.ones(c_out, device="cuda")
    bn_ref = train._BN(z_ref, gamma, beta, rm1, rv1, rows=rows)
    # the fused call
    z = torch.full((M, c_out), float("nan"), device="cuda")
    st = torch.empty((4, c_out), device="cuda")
    rm2, rv2 = torch.zeros(c_out, device="cuda"), torch.ones(c_out, device="cuda")
    need = lib.dal3_tr_linear_red_workspace_bytes(rows, c_out)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    rc = lib.dal3_tr_linear_bn_stats(hip.ptr(a), M, c_in, a.stride(0), hip.ptr(sc), hip.ptr(sh), 1, hip.ptr(W), c_in, hip.ptr(bias),
                                     seg, c_out, hip.ptr(z), z.stride(0), hip.ptr(pk), rows, hip.ptr(gamma), hip.ptr(beta),
                                     hip.ptr(rm2), hip.ptr(rv2), 0.1, 1e-5, hip.ptr(st[0]), hip.ptr(st[1]), hip.ptr(st[2]),
                                     hip.ptr(st[3]), hip.ptr(ws), need, hip.stream())
    assert rc == want, (rc, lib.dal3_last_error())
    torch.cuda.synchronize()
    assert torch.equal(z, z_ref)
    for got, ref in ((st[0], bn_ref.mu), (st[1], bn_ref.rstd), (st[2], bn_ref.scale), (st[3], bn_ref.shift), (rm2, rm1), (rv2, rv1)):
        assert _close(got, ref), float((got - ref).abs().max())
    if want == 0:                                                          # the very same two launches
        assert torch.equal(st[0], bn_ref.mu) and torch.equal(st[1], bn_ref.rstd)
    # against float64 over the tensor itself
    z64 = z[:rows].double()
    assert _close(st[0], z64.mean(0)) and _close(st[1], 1.0 / torch.sqrt(z64.var(0, unbiased=False) + 1e-5))
    # deterministic
    z2, st2 = torch.empty_like(z), torch.empty_like(st)
    lib.dal3_tr_linear_bn_stats(hip.ptr(a), M, c_in, a.stride(0), hip.ptr(sc), hip.ptr(sh), 1, hip.ptr(W), c_in, hip.ptr(bias),
                                seg, c_out, hip.ptr(z2), z2.stride(0), hip.ptr(pk), rows, hip.ptr(gamma), hip.ptr(beta),
                                None, None, 0.1, 1e-5, hip.ptr(st2[0]), hip.ptr(st2[1]), hip.ptr(st2[2]),
                                hip.ptr(st2[3]), hip.ptr(ws), need, hip.stream())
    torch.cuda.synchronize()
    assert torch.equal(st2, st) and torch.equal(z2, z)


#                                    M      K(c_in of the dgrad)  C     rows     fused
@pytest.mark.parametrize("M,K,C,rows,want", [
    (64 * 4096, 256, 512, None, 1),               # dconv2's dgrad -> dconv1's sums
    (64 * 4096, 128, 256, None, 1),               # dconv3's dgrad -> dconv2's sums
    (64 * 4096, 128, 128, None, 1),               # dconv4's dgrad -> dconv3's sums (K = 128: straight-line unit)
    (64 * 4096, 64, 128, None, 1),                # K = 64
    (64 * 4096, 128, 64, None, 0),                # 64 output channels: no fused instantiation, two steps
    (64 * 4096, 128, 256, 64 * 4096 - 40, 0),     # padding rows
])
def test_dgrad_with_fused_bn_backward_sums_equals_dgrad_then_sums(M, K, C, rows, want):
    lib = hip.lib()
    gen = torch.Generator(device="cuda").manual_seed(M + 3 * K + C)
    dz = torch.randn((M, K), device="cuda", generator=gen) * 1e-3
    W = torch.randn((K, C), device="cuda", generator=gen) / K ** 0.5          # the layer above: (c_out_fwd, c_in_fwd) = (K, C)
    bz = torch.randn((M, C), device="cuda", generator=gen) * 1.5 + 0.3         # this layer's pre-BN output
    gamma = torch.rand(C, device="cuda", generator=gen) + 0.5
    beta = torch.randn(C, device="cuda", generator=gen) * 0.3
    rows = rows if rows is not None else M
    bn = train._BN(bz, gamma, beta, None, None, rows=rows)
    pk = _pack(W, K, C, True, M, 0, False)
    da_ref = train._linear(dz, W, C, K, C, transpose=True, packed=pk)
    co_ref = torch.empty((5, C), device="cuda")
    need0 = lib.dal3_tr_colred_workspace_bytes(rows, C)
    ws0 = torch.empty(need0, dtype=torch.uint8, device="cuda")
    hip.check(lib.dal3_tr_bnbwd_sums(hip.ptr(bz), rows, C, bz.stride(0), hip.ptr(da_ref), da_ref.stride(0), None, None, 0,
                                     hip.ptr(bn.scale), hip.ptr(bn.shift), hip.ptr(bn.mu), hip.ptr(bn.rstd), hip.ptr(bn.gamma),
                                     hip.ptr(co_ref[0]), hip.ptr(co_ref[1]), hip.ptr(co_ref[2]), hip.ptr(co_ref[3]),
                                     hip.ptr(co_ref[4]), hip.ptr(ws0), need0, hip.stream()))
    da = torch.full((M, C), float("nan"), device="cuda")
    co = torch.empty((5, C), device="cuda")
    need = lib.dal3_tr_linear_red_workspace_bytes(rows, C)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    rc = lib.dal3_tr_linear_bnbwd_sums(hip.ptr(dz), M, K, dz.stride(0), hip.ptr(W), C, C, hip.ptr(da), da.stride(0), hip.ptr(pk),
                                       rows, hip.ptr(bz), bz.stride(0), hip.ptr(bn.scale), hip.ptr(bn.shift), hip.ptr(bn.mu),
                                       hip.ptr(bn.rstd), hip.ptr(bn.gamma), hip.ptr(co[0]), hip.ptr(co[1]), hip.ptr(co[2]),
                                       hip.ptr(co[3]), hip.ptr(co[4]), hip.ptr(ws), need, hip.stream())
    assert rc == want, (rc, lib.dal3_last_error())
    torch.cuda.synchronize()
    assert torch.equal(da, da_ref)
    for i, name in enumerate(("dgamma", "dbeta", "k1", "k2", "k3")):
        assert _close(co[i], co_ref[i]), (name, float((co[i] - co_ref[i]).abs().max()), float(co_ref[i].abs().max()))
    # float64 over the tensors themselves: dbeta = sum dy, dgamma = sum dy * xhat
    gate = (bz[:rows] * bn.scale + bn.shift) > 0
    dy = torch.where(gate, da[:rows], torch.zeros((), device="cuda")).double()
    xhat = ((bz[:rows] - bn.mu) * bn.rstd).double()
    assert _close(co[1], dy.sum(0), 1e-5) and _close(co[0], (dy * xhat).sum(0), 1e-5)
    # the product path: _BN.dgrad_with_sums + backward(co=...) gives the dz of the unfused backward
    da2, co2 = bn.dgrad_with_sums(bz, dz, W, C, K, pk)
    dz_f, dgam_f, dbet_f = bn.backward(bz, da=da2, co=co2)
    dz_u, dgam_u, dbet_u = bn.backward(bz, da=da_ref)
    assert torch.equal(da2, da_ref) and _close(dz_f, dz_u) and _close(dgam_f, dgam_u) and _close(dbet_f, dbet_u)


@pytest.mark.parametrize("M,Mp", [(64 * 4096, 64 * 4096), (5 * 77, 416), (3, 32)])
def test_logits_layer_with_dropout_as_valu_kernels(M, Mp):
    """dal3_tr_head2_* (round 4): dconv5(Dropout(relu(bn(z)))) forward, its input gradient and its parameter gradients
    without the post-Dropout activation in memory — against the stock float64 composition on an explicit multiplier, and
    with the multiplier re-created from a (seed, step) key: bit-identical to the run on the multiplier dal3_tr_act_dropout
    draws from the same key."""
    lib = hip.lib()
    gen = torch.Generator(device="cuda").manual_seed(M)
    z = torch.randn((Mp, 128), device="cuda", generator=gen)
    sc = torch.rand(128, device="cuda", generator=gen) + 0.5
    sh = torch.randn(128, device="cuda", generator=gen) * 0.3
    W = torch.randn((2, 128), device="cuda", generator=gen) * 0.2
    b = torch.randn(2, device="cuda", generator=gen)
    dl = torch.randn((M, 2), device="cuda", generator=gen) * 1e-3
    step = torch.tensor([3], dtype=torch.int64, device="cuda")
    key = (123456789, step, 0.5)
    # the multiplier that key stands for: act_dropout of ones
    ones = torch.ones((M, 128), device="cuda")
    mult = train._act_dropout(ones, None, key)
    assert 0.4 < float((mult == 0).float().mean()) < 0.6 and float(mult.max()) == 2.0
    a64 = (torch.relu(z[:M].double() * sc.double() + sh.double()) * mult.double())
    want_logits = a64 @ W.double().t() + b.double()
    want_da = (dl.double() @ W.double()) * mult.double()
    want_dW = dl.double().t() @ a64
    want_db = dl.double().sum(0)
    res = {}
    for tag, drop in (("mask", mult), ("key", key)):
        logits = train._head2_forward(z, (sc, sh, True), drop, W, b, M)
        da, dW, db = train._head2_backward(dl, z, (sc, sh, True), drop, W, M)
        torch.cuda.synchronize()
        res[tag] = (logits, da, dW, db)
        assert _close(logits, want_logits, 2e-6)
        assert _close(da[:M], want_da, 2e-6) and (Mp == M or float(da[M:].abs().max()) == 0.0)
        assert _close(dW, want_dW, 2e-6) and _close(db, want_db, 2e-6)
    for x, y in zip(res["mask"], res["key"]):
        assert torch.equal(x, y)
    # the dgrad that also takes the BatchNorm-backward sums of the layer below: the same da bits, the same dz / dgamma / dbeta
    gamma = torch.rand(128, device="cuda", generator=gen) + 0.5
    bn = train._BN(z, gamma, sh, None, None, rows=M)
    da_f, dW_f, db_f, co = train._head2_backward(dl, z, bn.act, key, W, M, bn=bn)
    da_u, dW_u, db_u = train._head2_backward(dl, z, bn.act, key, W, M)
    assert co is not None and torch.equal(da_f, da_u) and torch.equal(dW_f, dW_u)
    dz_f, dgam_f, dbet_f = bn.backward(z, da=da_f, co=co)
    dz_u, dgam_u, dbet_u = bn.backward(z, da=da_u)
    assert _close(dz_f[:M], dz_u[:M]) and _close(dgam_f, dgam_u) and _close(dbet_f, dbet_u)
    # no Dropout at all (p = 0): the plain layer
    logits = train._head2_forward(z, (sc, sh, True), None, W, b, M)
    assert _close(logits, torch.relu(z[:M].double() * sc.double() + sh.double()) @ W.double().t() + b.double(), 2e-6)
    assert lib.dal3_tr_head2_forward(hip.ptr(z), M, 64, 128, None, None, 0, None, 0, 0, None, 0.0, hip.ptr(W), 128, hip.ptr(b),
                                     hip.ptr(logits), hip.stream()) == hip.EINVAL          # C must be 128


@pytest.mark.parametrize("B,N,c_in,c_out", [(64, 4096, 3, 64), (5, 77, 4, 64), (3, 101, 8, 64), (64, 512, 3, 128), (1, 2, 3, 64)])
def test_first_layer_as_valu_kernels(B, N, c_in, c_out):
    """dal3_tr_conv1_bn_stats / dal3_tr_conv1_wgrad (round 4): conv1 of a stack (3, 4 or 8 input channels) on the points as
    they are — output, batch statistics, running statistics and the weight gradient against float64; padding rows of z
    hold the bias and stay out of the statistics"""
    gen = torch.Generator(device="cuda").manual_seed(B * N + c_in)
    pts = torch.randn((B, N, c_in), device="cuda", generator=gen).transpose(2, 1)      # (B, C, N) view of point-major storage
    W = torch.randn((c_out, c_in), device="cuda", generator=gen)
    b = torch.randn(c_out, device="cuda", generator=gen)
    gamma = torch.rand(c_out, device="cuda", generator=gen) + 0.5
    beta = torch.randn(c_out, device="cuda", generator=gen) * 0.3
    rm, rv = torch.zeros(c_out, device="cuda"), torch.ones(c_out, device="cuda")
    rows = train._Rows(pts)
    M, Mp = B * N, rows.shape[0]
    assert rows.x.data_ptr() == pts.data_ptr()                                          # no copy of point-major points
    z, bn = train._conv1_bn(rows, W, b, gamma, beta, (rm, rv))
    x64 = pts.transpose(2, 1).reshape(M, c_in).double()
    z64 = x64 @ W.double().t() + b.double()
    assert z.shape == (Mp, c_out) and _close(z[:M], z64, 2e-6)
    if Mp > M:
        assert torch.equal(z[M:], b.expand(Mp - M, c_out))
    mean, var = z64.mean(0), z64.var(0, unbiased=False)
    assert _close(bn.mu, mean, 2e-6) and _close(bn.rstd, 1.0 / torch.sqrt(var + 1e-5), 2e-6)
    assert _close(bn.scale, gamma.double() / torch.sqrt(var + 1e-5), 2e-6)
    assert _close(rm, 0.1 * mean, 2e-6) and _close(rv, 0.9 + 0.1 * z64.var(0, unbiased=True), 2e-6)
    dz = torch.randn((Mp, c_out), device="cuda", generator=gen) * 1e-3
    dW = train._conv1_wgrad(dz, rows, c_out)
    assert dW.shape == (c_out, c_in) and _close(dW, dz[:M].double().t() @ x64, 2e-6)


def test_deferred_second_stages_of_the_weight_gradients_give_the_same_bits():
    """train.deferred_wgrad_finals (dal3_tr_wgrad_final_many, round 4): the weight gradients of a backward function leave
    their per-slice partial sums behind and ONE launch adds them — the bits of the per-call second stage, for the fp32 and
    the f16x3 kernels, more items than one launch takes (24), and a dW is untouched until the flush"""
    gen = torch.Generator(device="cuda").manual_seed(11)
    M = 16 * 4096
    shapes = [(64, 64), (128, 64), (256, 128), (512, 256), (64, 512)] * 6          # 30 items: two launches
    dzs = [torch.randn((M, co), device="cuda", generator=gen) * 1e-3 for co, _ in shapes[:5]]
    acts = [torch.randn((M, ci), device="cuda", generator=gen) for _, ci in shapes[:5]]
    sc = [torch.rand(ci, device="cuda", generator=gen) + 0.5 for _, ci in shapes[:5]]
    sh = [torch.randn(ci, device="cuda", generator=gen) * 0.1 for _, ci in shapes[:5]]
    want = [train._wgrad(dzs[i % 5], acts[i % 5], co, ci, (sc[i % 5], sh[i % 5], True)) for i, (co, ci) in enumerate(shapes)]
    with train.deferred_wgrad_finals() as later:
        got = [train._wgrad(dzs[i % 5], acts[i % 5], co, ci, (sc[i % 5], sh[i % 5], True), later=True) for i, (co, ci) in enumerate(shapes)]
        now = train._wgrad(dzs[0], acts[0], 64, 64, (sc[0], sh[0], True))       # (not marked: complete when it returns)
        assert torch.equal(now, want[0])
        later.flush()
        assert all(torch.equal(g, w) for g, w in zip(got, want))
    # the f16x3 wgrad takes the same route (amax: the bits of max |dz| in 64 words)
    amax = torch.zeros(64, dtype=torch.int32, device="cuda")
    amax[0] = int(dzs[3].abs().max().view(torch.int32))
    w3 = train._wgrad(dzs[3], acts[3], 512, 256, (sc[3], sh[3], True), amax=amax)
    with train.deferred_wgrad_finals():
        g3 = train._wgrad(dzs[3], acts[3], 512, 256, (sc[3], sh[3], True), amax=amax, later=True)
    assert torch.equal(g3, w3)
    assert train._deferred() is None
