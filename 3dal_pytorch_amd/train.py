"""Training forward/backward of the shared-MLP stacks on the MI355X (first slice).

`loss.backward()` in the reference's train drivers (tools/static_train.py:53-166) spends its time in the per-point
stacks Conv1d(k=1) -> BatchNorm1d (batch statistics) -> ReLU (-> max over points) of `PointNetInstanceSeg`
(tools/static_model.py:271-295) and `PointNetEstimation` (:326-334). Here those stacks are two
`torch.autograd.Function`s whose forward AND backward run on lib3dal_hip.so's training kernels (csrc/dal3_train.hip:
fp32 MFMA linear / dgrad / wgrad over point-major activations, fixed-order batch-statistics and BN-backward
reductions, arg-max pooling); the per-crop tails (three Linear+BN1d layers on (B,512) vectors), Dropout's random
mask and the loss stay stock torch ops — together < 0.1 % of the arithmetic.

Semantics follow torch exactly: biased batch variance for normalisation, unbiased for `running_var`, momentum 0.1,
eps 1e-5 (static_model.py:250-269 uses the defaults), ReLU gradient 0 at 0, the max routes its gradient to one
arg-max point. A Conv1d bias in front of a train-mode BatchNorm has an exactly zero gradient (BN subtracts the
mean); autograd's numerically-noisy ~1e-7 is returned as 0.

Parity: tests/test_gpu_train.py compares outputs, running statistics and every parameter gradient with
torch autograd over the stock-torch composite (`PointNetInstanceSeg.forward`, `_PointHead.forward`), which
tests/test_host_cpu.py ties to the oracle.
"""
import os
import threading

import torch

from . import _hip

_EPS = 1e-5
_MOM = 0.1


_SCRATCH = {}

# Test hook (tests/test_gpu_train_reference.py): when a dict, _InsSeg.forward leaves references to what decides its
# discrete events there — every layer's pre-BN output and BatchNorm affine (the ReLU gates are relu(z*scale+shift) > 0)
# and the pooled points — so that a float64 composite can be evaluated AT THE SAME gates. None in production.
CAPTURE = None

# Arithmetic of the training FORWARD's big layers (c_out % 256 == 0, M % 256 == 0 ...: dal3_tr_linear_x3_layout):
# "fp32" — the exact-fp32 MFMA kernels; "f16x3" — fp16 MFMAs on (hi, lo) split operands, fp32 accumulate
# (dal3_train_x3.hip: the same 1e-6 of the output's range, 1.3-2.5 x faster per layer), and the decoder's dgrads that
# qualify and the decoder's wgrads with them (their dz operand is scaled by a power of two around the products:
# _BN.backward(amax=)). Set by the drop-ins from model.precision; a backward follows what its forward ran on.
# Measurement hook (tools/train_roofline.py): when a list, every launch of a per-point kernel family appends what it moves
# — (family, rows, c_in, c_out, algorithmic FLOP, algorithmic HBM bytes) — in launch order, to be joined with a kernel
# trace of the same step. None in production.
CALLS = None


def _note(family, M, c_in, c_out, flop=0.0, nbytes=0.0, **extra):
    if CALLS is not None:
        CALLS.append(dict(family=family, M=int(M), c_in=int(c_in), c_out=int(c_out), flop=float(flop), bytes=float(nbytes), **extra))


ARITH = "fp32"
_WGRAD_X3 = os.environ.get("DAL3_TRAIN_WGRAD_X3", "1") != "0"      # (A/B switch: 0 keeps wgrad on the fp32 kernel in an f16x3 step)


class arithmetic:
    """with arithmetic("f16x3"): ...  — the training forward inside runs its big layers on the f16x3 kernels"""

    def __init__(self, name):
        if name not in ("fp32", "f16x3"):
            raise ValueError(f"unknown training arithmetic {name!r}")
        self.name = name

    def __enter__(self):
        global ARITH
        self.prev, ARITH = ARITH, self.name
        return self

    def __exit__(self, *exc):
        global ARITH
        ARITH = self.prev
        return False


class _X3Image:
    """a dal3_tr_pack_many image in the f16x3 layout (read by dal3_tr_linear_x3)"""

    def __init__(self, tensor):
        self.tensor = tensor


def _ws(nbytes, dev):
    """grow-only scratch per (device, stream). Every user enqueues on the current stream and is done with the buffer when its
    last kernel has run, so consecutive calls can share it (stream order); a per-call torch.empty costs more host
    time than the small kernels it serves."""
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)       # one buffer per stream: stream order is the only fence
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=dev)
        _SCRATCH[key] = buf
    return buf


def _take_saved(ctx):
    """the forward's buffers, handed over to this backward call: the ctx attribute is cleared, so a step's activations are
    released when its backward returns — like torch's own saved tensors — and not when the caller's `loss` / output dict
    go away, which in the usual loop is AFTER the next forward has allocated its own (2.5 GB held at DynamicModel's batch).
    A second backward over the same graph (retain_graph=True) is therefore not available on this path."""
    saved = ctx.saved
    if saved is None:
        raise RuntimeError("3dal_pytorch_amd.train: backward through the HIP training path a second time (its buffers are "
                           "released by the first; use model.train_backend = 'torch' if you need retain_graph)")
    ctx.saved = None
    return saved


def _linear(a, W, ldw, c_in, c_out, act=None, bias=None, seg=0, transpose=False, out=None, accumulate=False, packed=None,
            amax=None):
    """z (M,c_out) (+)= act(a) @ Wop^T + bias through dal3_tr_linear. packed: this call's weights already in fragment
    order (_prepack: one launch for all the layers of a stack instead of one in front of every call)"""
    M = a.shape[0]
    z = out if out is not None else torch.empty((M, c_out), dtype=torch.float32, device=a.device)
    sc, sh, relu = (act if act is not None else (None, None, False))
    lib = _hip.lib()
    _note("linear_x3" if isinstance(packed, _X3Image) else "linear", M, c_in, c_out, 2.0 * M * c_in * c_out,
          4.0 * M * (c_in + c_out * (2 if accumulate else 1)), transpose=bool(transpose))
    if isinstance(packed, _X3Image):
        # amax: 64 device words whose maximum is the bit pattern of the operand's largest |value| (a dgrad's dz, far below fp16's range: the
        # kernel scales by a power of two around the products); a transposed image is only ever packed for such a caller
        assert amax is not None or not transpose
        _hip.check(lib.dal3_tr_linear_x3(_hip.ptr(a), M, c_in, a.stride(0), _hip.ptr(sc), _hip.ptr(sh), int(relu), _hip.ptr(bias), seg,
                                         c_out, _hip.ptr(z), z.stride(0), _hip.ptr(packed.tensor), _hip.ptr(amax), _hip.stream()))
        return z
    if packed is not None:
        _hip.check(lib.dal3_tr_linear_prepacked(_hip.ptr(a), M, c_in, a.stride(0), _hip.ptr(sc), _hip.ptr(sh), int(relu),
                                                _hip.ptr(W), ldw, int(transpose), _hip.ptr(bias), seg, c_out, _hip.ptr(z),
                                                z.stride(0), int(accumulate), _hip.ptr(packed), _hip.stream()))
        return z
    need = lib.dal3_tr_linear_workspace_bytes(c_in, c_out)
    ws = _ws(need, a.device) if need else None
    _hip.check(lib.dal3_tr_linear(_hip.ptr(a), M, c_in, a.stride(0), _hip.ptr(sc), _hip.ptr(sh), int(relu), _hip.ptr(W), ldw,
                                  int(transpose), _hip.ptr(bias), seg, c_out, _hip.ptr(z), z.stride(0), int(accumulate),
                                  _hip.ptr(ws), need, _hip.stream()))
    return z


def _pad32(n):
    return (n + 31) // 32 * 32


def _prepack(specs, dev):
    """Fragment-order images of the weights of several dal3_tr_linear calls from ONE launch (dal3_tr_pack_many).
    specs: [(W 2-D fp32 with contiguous rows, c_in, c_out, transpose, M, seg, accumulate, has_act)] as each CALL will see
    them (c_in / c_out swapped for a transposed weight). Returns one packed tensor per spec, None where the call reads
    no packed image. The weights must not change between this and the calls (a step's forward and backward: they do not)."""
    lib = _hip.lib()
    # 9th element: a dgrad whose caller supplies the operand's amax (True), or "fp32": never the f16x3 image for this call
    specs = [sp if len(sp) == 9 else (*sp, False) for sp in specs]
    lay = [lib.dal3_tr_linear_pack_layout(M, ci, seg, co, int(acc), int(has_act)) for _, ci, co, _, M, seg, acc, has_act, _d in specs]
    if ARITH == "f16x3":                                    # forward calls (and marked dgrads) that qualify take the f16x3 image instead
        lay = [(lib.dal3_tr_linear_x3_layout(M, ci, seg, co, int(acc), int(has_act)) if ((not tr or dg) and dg != "fp32") else 0) or l
               for l, (_, ci, co, tr, M, seg, acc, has_act, dg) in zip(lay, specs)]
    size = [int(lib.dal3_tr_linear_workspace_bytes(ci, co)) if l else 0 for l, (_, ci, co, *_r) in zip(lay, specs)]
    size = [(n + 255) // 256 * 256 for n in size]
    buf = torch.empty(max(sum(size), 16), dtype=torch.uint8, device=dev)
    out, items, off = [], [], 0
    for l, n, (W, ci, co, tr, *_r) in zip(lay, size, specs):
        if not l:
            out.append(None)
            continue
        view = buf[off:off + n]
        off += n
        out.append(_X3Image(view) if l & 0x100 else view)
        items.append(_hip.PackItem(_hip.ptr(W), W.stride(0), int(tr), co, ci, l, _hip.ptr(view)))
    for i in range(0, len(items), 48):
        chunk = items[i:i + 48]
        arr = (_hip.PackItem * len(chunk))(*chunk)
        _hip.check(lib.dal3_tr_pack_many(arr, len(chunk), _hip.stream()))
    return out


def _zero_grads(shapes, idx, dev):
    """gradients that are analytically zero (a conv / fc bias in front of a train-mode BatchNorm): views of ONE zeroed
    buffer instead of a fill kernel per parameter (twenty of them per step)"""
    sizes = [int(torch.Size(shapes[i]).numel()) for i in idx]
    flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
    out, o = {}, 0
    for i, n in zip(idx, sizes):
        out[i] = flat[o:o + n].view(shapes[i])
        o += n
    return out


def _colred(z, mode, da=None, dg=None, arg=None, seg=0, bn=None, rows=None):
    """float64 column sums (2*C,) on the device over the first `rows` rows of z: see dal3_tr_colred"""
    M, C = z.shape
    M = rows if rows is not None else M
    lib = _hip.lib()
    _note("stats" if mode == 0 else "bwd_sums", M, C, C, 0.0, 4.0 * M * C * (1 if mode == 0 else 2))
    need = lib.dal3_tr_colred_workspace_bytes(M, C)
    ws = _ws(need, z.device)
    out = torch.empty(2 * C, dtype=torch.float64, device=z.device)
    sc, sh, mu, rstd = bn if bn is not None else (None, None, None, None)
    _hip.check(lib.dal3_tr_colred(_hip.ptr(z), M, C, z.stride(0), mode, _hip.ptr(da), da.stride(0) if da is not None else 0,
                                  _hip.ptr(dg), _hip.ptr(arg), seg, _hip.ptr(sc), _hip.ptr(sh), _hip.ptr(mu), _hip.ptr(rstd),
                                  _hip.ptr(ws), need, _hip.ptr(out), _hip.stream()))
    return out


_TLS = threading.local()    # .deferred: inside deferred_wgrad_finals(), the (partial sums, slices, elements, dW) of the weight
#                             gradients so far — per thread (autograd runs a device's backward functions on its own thread)


def _deferred():
    return getattr(_TLS, "deferred", None)


class deferred_wgrad_finals:
    """`with deferred_wgrad_finals() as d:` — every _wgrad(later=True) inside leaves its per-slice partial sums in a buffer of its
    own and returns a dW that is filled when the block ends (or at d.flush(), which must precede any stock op that reads one
    of them): ONE second-stage launch for up to 24 weight gradients instead of one behind each (dal3_tr_wgrad_final_many)."""

    def __enter__(self):
        self.outer, _TLS.deferred = _deferred(), []
        return self

    def flush(self):
        items = _deferred()
        while items:
            chunk, items[:] = items[:24], items[24:]
            arr = (_hip.WgradPart * len(chunk))(*[_hip.WgradPart(_hip.ptr(p), ns, n, _hip.ptr(dW)) for p, ns, n, dW in chunk])
            _hip.check(_hip.lib().dal3_tr_wgrad_final_many(arr, len(chunk), _hip.stream()))

    def __exit__(self, exc_type, exc, tb):
        try:
            if exc_type is None:
                self.flush()
        finally:
            _TLS.deferred = self.outer
        return False


def _wgrad(dz, a, c_out, c_in, act=None, amax=None, later=False):
    """dW (c_out, c_in) = dz^T act(a). amax (64 device words holding the bits of max |dz|, _BN.backward(amax=)): the caller's
    step runs on the f16x3 arithmetic — the layer takes dal3_tr_wgrad_x3 when its shape qualifies. later: inside
    deferred_wgrad_finals() the returned dW is filled when that block ends (the caller only passes it on, as a view)"""
    M = dz.shape[0]
    lib = _hip.lib()
    if amax is not None and _WGRAD_X3:
        need = lib.dal3_tr_wgrad_x3_workspace_bytes(M, c_out, c_in)
        if need:
            _note("wgrad_x3", M, c_in, c_out, 2.0 * M * c_in * c_out, 4.0 * M * (c_in + c_out))
            defer = later and _deferred() is not None
            ws = torch.empty(need, dtype=torch.uint8, device=dz.device) if defer else _ws(need, dz.device)
            dW = torch.empty((c_out, c_in), dtype=torch.float32, device=dz.device)
            sc, sh, relu = (act if act is not None else (None, None, False))
            _hip.check(lib.dal3_tr_wgrad_x3(_hip.ptr(dz), dz.stride(0), _hip.ptr(a), a.stride(0), _hip.ptr(sc), _hip.ptr(sh),
                                            int(relu), _hip.ptr(amax), M, c_out, c_in, _hip.ptr(ws), need,
                                            None if defer else _hip.ptr(dW), _hip.stream()))
            if defer:
                _deferred().append((ws, need // (4 * c_out * c_in), c_out * c_in, dW))
            return dW
    _note("wgrad", M, c_in, c_out, 2.0 * M * c_in * c_out, 4.0 * M * (c_in + c_out))
    need = lib.dal3_tr_wgrad_workspace_bytes(M, c_out, c_in)
    defer = later and _deferred() is not None
    ws = torch.empty(need, dtype=torch.uint8, device=dz.device) if defer else _ws(need, dz.device)
    dW = torch.empty((c_out, c_in), dtype=torch.float32, device=dz.device)
    sc, sh, relu = (act if act is not None else (None, None, False))
    _hip.check(lib.dal3_tr_wgrad(_hip.ptr(dz), dz.stride(0), _hip.ptr(a), a.stride(0), _hip.ptr(sc), _hip.ptr(sh), int(relu),
                                 M, c_out, c_in, _hip.ptr(ws), need, None if defer else _hip.ptr(dW), _hip.stream()))
    if defer:
        _deferred().append((ws, need // (4 * c_out * c_in), c_out * c_in, dW))
    return dW


class _BN:
    """batch statistics of one layer's pre-BN output and everything derived from them"""

    def __init__(self, z, gamma, beta, running_mean, running_var, sums=None, rows=None, lin=None):
        """z: the layer's pre-BN output (M,C), or just its shape (M, C) when `sums` is given. rows: how many of z's
        rows are real (the buffers are padded to a multiple of 32 rows; the padding takes no part in the statistics).
        sums: optional float64 (2*C,) [sum z, sum z^2] obtained without a pass over z (see _moments_through).
        lin: z has NOT been computed yet — (a, W, ldw, c_in, act, bias, seg, packed): z = act(a) W^T + bias is written
        by this call too, its statistics taken in the linear kernel's epilogue (dal3_tr_linear_bn_stats)"""
        M, C = z if isinstance(z, tuple) else z.shape
        M = rows if rows is not None else M
        dev = sums.device if sums is not None else z.device
        st = torch.empty((4, C), dtype=torch.float32, device=dev)
        self.mu, self.rstd, self.scale, self.shift = st[0], st[1], st[2], st[3]
        self.gamma = gamma.contiguous()
        lib = _hip.lib()
        if lin is not None:
            a, W, ldw, c_in, act, bias, seg, packed = lin
            sc, sh, relu = (act if act is not None else (None, None, False))
            need = lib.dal3_tr_linear_red_workspace_bytes(M, C)
            ws = _ws(need, dev)
            rc = lib.dal3_tr_linear_bn_stats(_hip.ptr(a), a.shape[0], c_in, a.stride(0), _hip.ptr(sc), _hip.ptr(sh), int(relu),
                                             _hip.ptr(W), ldw, _hip.ptr(bias), seg, C, _hip.ptr(z), z.stride(0), _hip.ptr(packed), M,
                                             _hip.ptr(self.gamma), _hip.ptr(beta.contiguous()), _hip.ptr(running_mean),
                                             _hip.ptr(running_var), _MOM, _EPS, _hip.ptr(self.mu), _hip.ptr(self.rstd),
                                             _hip.ptr(self.scale), _hip.ptr(self.shift), _hip.ptr(ws), need, _hip.stream())
            if rc < 0:
                _hip.check(rc)
            Ma = a.shape[0]
            if rc == 1:                                     # fused: one kernel did both
                _note("linear+stats", Ma, c_in, C, 2.0 * Ma * c_in * C, 4.0 * Ma * (c_in + C))
            else:
                _note("linear", Ma, c_in, C, 2.0 * Ma * c_in * C, 4.0 * Ma * (c_in + C), transpose=False)
                _note("stats", M, C, C, 0.0, 4.0 * M * C)
        elif sums is None:                                  # reduction + epilogue: two launches
            _note("stats", M, C, C, 0.0, 4.0 * M * C)
            need = lib.dal3_tr_colred_workspace_bytes(M, C)
            ws = _ws(need, dev)
            _hip.check(lib.dal3_tr_bn_stats(_hip.ptr(z), M, C, z.stride(0), _hip.ptr(self.gamma), _hip.ptr(beta.contiguous()),
                                            _hip.ptr(running_mean), _hip.ptr(running_var), _MOM, _EPS, _hip.ptr(self.mu),
                                            _hip.ptr(self.rstd), _hip.ptr(self.scale), _hip.ptr(self.shift), _hip.ptr(ws),
                                            need, _hip.stream()))
        else:
            _hip.check(lib.dal3_tr_bn_finalize(_hip.ptr(sums), C, M, _hip.ptr(self.gamma), _hip.ptr(beta.contiguous()),
                                               _hip.ptr(running_mean), _hip.ptr(running_var), _MOM, _EPS, _hip.ptr(self.mu),
                                               _hip.ptr(self.rstd), _hip.ptr(self.scale), _hip.ptr(self.shift),
                                               _hip.stream()))
        self.M = M

    @property
    def act(self):
        return (self.scale, self.shift, True)

    def dgrad_with_sums(self, z, dz_next, W, ldw, c_in, packed):
        """da = dz_next W (the dgrad of the layer ABOVE, through its own weight: the gradient w.r.t. this layer's relu(bn(z)))
        together with this layer's BatchNorm-backward sums, taken in the dgrad kernel's epilogue
        (dal3_tr_linear_bnbwd_sums): returns (da, co) — pass co on to backward(), which then skips its reduction pass"""
        C = z.shape[1]
        lib = _hip.lib()
        if packed is None or isinstance(packed, _X3Image):                  # no fp32 image for this call: the plain dgrad
            return _linear(dz_next, W, ldw, c_in, C, transpose=True, packed=packed), None
        da = torch.empty((dz_next.shape[0], C), dtype=torch.float32, device=z.device)
        co = torch.empty((5, C), dtype=torch.float32, device=z.device)      # dgamma, dbeta, k1, k2, k3
        need = lib.dal3_tr_linear_red_workspace_bytes(self.M, C)
        ws = _ws(need, z.device)
        rc = lib.dal3_tr_linear_bnbwd_sums(_hip.ptr(dz_next), dz_next.shape[0], c_in, dz_next.stride(0), _hip.ptr(W), ldw, C,
                                           _hip.ptr(da), da.stride(0), _hip.ptr(packed), self.M, _hip.ptr(z), z.stride(0),
                                           _hip.ptr(self.scale), _hip.ptr(self.shift), _hip.ptr(self.mu), _hip.ptr(self.rstd),
                                           _hip.ptr(self.gamma), _hip.ptr(co[0]), _hip.ptr(co[1]), _hip.ptr(co[2]), _hip.ptr(co[3]),
                                           _hip.ptr(co[4]), _hip.ptr(ws), need, _hip.stream())
        if rc < 0:
            _hip.check(rc)
        Mz = dz_next.shape[0]
        if rc == 1:
            _note("linear+bwd_sums", Mz, c_in, C, 2.0 * Mz * c_in * C, 4.0 * Mz * (c_in + 2 * C))
        else:
            _note("linear", Mz, c_in, C, 2.0 * Mz * c_in * C, 4.0 * Mz * (c_in + C), transpose=True)
            _note("bwd_sums", self.M, C, C, 0.0, 8.0 * self.M * C)
        return da, co

    def backward(self, z, da=None, dg=None, arg=None, seg=0, sum_seg=0, amax=None, co=None):
        """(dz, dgamma, dbeta) from the gradient w.r.t. relu(bn(z)); with sum_seg also the column sums of dz over every
        segment of sum_seg rows, (M / sum_seg, C), taken in the same pass (dal3_tr_bnbwd_apply_segsum). co: the sums and
        coefficients, when dgrad_with_sums has already produced them with da"""
        C = z.shape[1]
        M = self.M                                                          # the real rows; padding rows get dz = 0
        lib = _hip.lib()
        _note("apply", M, C, C, 0.0, 12.0 * M * C)
        if co is None:
            _note("bwd_sums", M, C, C, 0.0, 8.0 * M * C if da is not None else 4.0 * M * C)
            co = torch.empty((5, C), dtype=torch.float32, device=z.device)      # dgamma, dbeta, k1, k2, k3
            need = lib.dal3_tr_colred_workspace_bytes(M, C)
            ws = _ws(need, z.device)
            _hip.check(lib.dal3_tr_bnbwd_sums(_hip.ptr(z), M, C, z.stride(0), _hip.ptr(da), da.stride(0) if da is not None else 0,
                                              _hip.ptr(dg), _hip.ptr(arg), seg, _hip.ptr(self.scale), _hip.ptr(self.shift),
                                              _hip.ptr(self.mu), _hip.ptr(self.rstd), _hip.ptr(self.gamma), _hip.ptr(co[0]),
                                              _hip.ptr(co[1]), _hip.ptr(co[2]), _hip.ptr(co[3]), _hip.ptr(co[4]), _hip.ptr(ws),
                                              need, _hip.stream()))
        dz = torch.empty_like(z)
        if sum_seg:
            fused = da is not None and C % 64 == 0 and sum_seg % 128 == 0 and M % sum_seg == 0 and M == z.shape[0]
            if fused:
                sums = torch.empty((M // sum_seg, C), dtype=torch.float32, device=z.device)
                need2 = lib.dal3_tr_bnbwd_apply_segsum_workspace_bytes(M, C)
                ws2 = _ws(need2, z.device)
                _hip.check(lib.dal3_tr_bnbwd_apply_segsum(_hip.ptr(z), M, C, z.stride(0), _hip.ptr(da), da.stride(0),
                                                          _hip.ptr(self.scale), _hip.ptr(self.shift), _hip.ptr(self.mu),
                                                          _hip.ptr(self.rstd), _hip.ptr(co[2]), _hip.ptr(co[3]), _hip.ptr(co[4]),
                                                          _hip.ptr(dz), dz.stride(0), sum_seg, _hip.ptr(sums), _hip.ptr(ws2),
                                                          need2, _hip.stream()))
                return dz, co[0], co[1], sums
        if amax is not None:                            # (64 zeroed int32 words: the bits of max |dz| are atomicMax'ed into them)
            _hip.check(lib.dal3_tr_bnbwd_apply_amax(_hip.ptr(z), M, C, z.stride(0), _hip.ptr(da), da.stride(0) if da is not None else 0,
                                                    _hip.ptr(dg), _hip.ptr(arg), seg, _hip.ptr(self.scale), _hip.ptr(self.shift),
                                                    _hip.ptr(self.mu), _hip.ptr(self.rstd), _hip.ptr(co[2]), _hip.ptr(co[3]),
                                                    _hip.ptr(co[4]), _hip.ptr(dz), dz.stride(0), _hip.ptr(amax), _hip.stream()))
        else:
            _hip.check(lib.dal3_tr_bnbwd_apply(_hip.ptr(z), M, C, z.stride(0), _hip.ptr(da), da.stride(0) if da is not None else 0,
                                               _hip.ptr(dg), _hip.ptr(arg), seg, _hip.ptr(self.scale), _hip.ptr(self.shift),
                                               _hip.ptr(self.mu), _hip.ptr(self.rstd), _hip.ptr(co[2]), _hip.ptr(co[3]),
                                               _hip.ptr(co[4]), _hip.ptr(dz), dz.stride(0), _hip.stream()))
        if z.shape[0] > M:
            dz[M:].zero_()                                                  # (wgrad sums over every row it is given)
        if sum_seg:                                                         # (ragged sizes: the caller takes the separate pass)
            return dz, co[0], co[1], None
        return dz, co[0], co[1]


def _linear_bn(a, W, c_in, c_out, act, bias, seg, packed, gamma, beta, stats, rows):
    """z = act(a) W^T + bias and its _BN (batch statistics, running-statistics update): one library call that takes the
    statistics in the linear kernel's epilogue where the shape allows and runs the two steps itself where not; the f16x3
    images and unpacked calls keep the separate statistics pass"""
    rm, rv = stats if stats is not None else (None, None)
    if packed is None or isinstance(packed, _X3Image):
        z = _linear(a, W, W.shape[1], c_in, c_out, act=act, bias=bias, seg=seg, packed=packed)
        return z, _BN(z, gamma, beta, rm, rv, rows=rows)
    z = torch.empty((a.shape[0], c_out), dtype=torch.float32, device=a.device)
    return z, _BN(z, gamma, beta, rm, rv, rows=rows, lin=(a, W, W.shape[1], c_in, act, bias, seg, packed))


def _segmax(z, bn, seg):
    C = z.shape[1]
    n_seg = bn.M // seg
    g = torch.empty((n_seg, C), dtype=torch.float32, device=z.device)
    arg = torch.empty((n_seg, C), dtype=torch.int32, device=z.device)
    ws = torch.empty(n_seg * C, dtype=torch.int64, device=z.device)
    _hip.check(_hip.lib().dal3_tr_segmax(_hip.ptr(z), z.stride(0), seg, C, _hip.ptr(bn.scale), _hip.ptr(bn.shift), _hip.ptr(g),
                                         _hip.ptr(arg), n_seg, _hip.ptr(ws), ws.numel() * 8, _hip.stream()))
    return g, arg


def _linear_pool(a, act, W, b, bn, seg):
    """g, arg of  max over each segment of relu(bn(act(a) @ W^T + b))  without the layer's output (dal3_tr_linear_pool)"""
    M, c_in = a.shape
    c_out = W.shape[0]
    n_seg = M // seg
    lib = _hip.lib()
    need = lib.dal3_tr_linear_pool_workspace_bytes(c_in, c_out, n_seg)
    ws = _ws(need, a.device)
    g = torch.empty((n_seg, c_out), dtype=torch.float32, device=a.device)
    arg = torch.empty((n_seg, c_out), dtype=torch.int32, device=a.device)
    sc, sh, relu = act
    _note("linear_pool", M, c_in, c_out, 2.0 * M * c_in * c_out, 4.0 * M * c_in)
    pool = lib.dal3_tr_linear_pool
    if ARITH == "f16x3" and lib.dal3_tr_linear_pool_x3_ok(M, c_in, seg, c_out):
        pool = lib.dal3_tr_linear_pool_x3
    _hip.check(pool(_hip.ptr(a), M, c_in, a.stride(0), _hip.ptr(sc), _hip.ptr(sh), int(relu), _hip.ptr(W),
                                       W.stride(0), _hip.ptr(b), _hip.ptr(bn.scale), _hip.ptr(bn.shift), seg, c_out,
                                       _hip.ptr(g), _hip.ptr(arg), _hip.ptr(ws), need, _hip.stream()))
    return g, arg


def draw_step(module, dev):
    """the module's device-resident draw counter (starts at 0 with the module, so runs are repeatable), bumped once
    per random draw of a training forward by an ordinary, graph-capturable op. Not a buffer: it is no part of the
    reference's state_dict."""
    t = module.__dict__.get("_draw_step")
    if t is None or t.device != dev:
        t = torch.zeros(1, dtype=torch.int64, device=dev)
        module.__dict__["_draw_step"] = t
    return t


def _act_dropout(x, act, drop):
    """act(x) * Dropout multiplier in one pass (dal3_tr_act_dropout). drop: None | a (M,C) multiplier tensor |
    (seed, step tensor, p): multiplier re-created from the key, nothing stored"""
    M, C = x.shape
    out = torch.empty((M, C), dtype=torch.float32, device=x.device)
    sc, sh, relu = act if act is not None else (None, None, False)
    _note("act", M, C, C, 0.0, 8.0 * M * C)
    mult, seed, step, p = None, 0, None, 0.0
    if torch.is_tensor(drop):
        mult = drop
    elif drop is not None:
        seed, step, p = drop
    _hip.check(_hip.lib().dal3_tr_act_dropout(_hip.ptr(x), M, C, x.stride(0), _hip.ptr(sc), _hip.ptr(sh), int(relu),
                                              _hip.ptr(mult), mult.stride(0) if mult is not None else 0, seed, _hip.ptr(step),
                                              float(p), _hip.ptr(out), out.stride(0), _hip.stream()))
    return out


def _drop_args(drop):
    """drop (None | multiplier tensor | (seed, step tensor, p)) -> (mult, ldm, seed, step, p) as the C ABI takes them"""
    if torch.is_tensor(drop):
        return drop, drop.stride(0), 0, None, 0.0
    if drop is not None:
        seed, step, p = drop
        return None, 0, seed, step, float(p)
    return None, 0, 0, None, 0.0


def _head2_forward(z, act, drop, W, b, M):
    """logits (M, 2) = dconv5(Dropout(act(z))) without the post-Dropout activation in memory (dal3_tr_head2_forward)"""
    sc, sh, relu = act
    mult, ldm, seed, step, p = _drop_args(drop)
    out = torch.empty((M, 2), dtype=torch.float32, device=z.device)
    _note("head2", M, 128, 2, 0.0, 4.0 * M * 130)
    _hip.check(_hip.lib().dal3_tr_head2_forward(_hip.ptr(z), M, z.shape[1], z.stride(0), _hip.ptr(sc), _hip.ptr(sh), int(relu),
                                                _hip.ptr(mult), ldm, seed, _hip.ptr(step), p, _hip.ptr(W), W.stride(0), _hip.ptr(b),
                                                _hip.ptr(out), _hip.stream()))
    return out


def _head2_backward(dl, z, act, drop, W, M, bn=None):
    """(da (rows of z, 128) = gradient w.r.t. act(z), dW (2, 128), db (2,)[, co]) of the same layer: dal3_tr_head2_dgrad /
    _wgrad. bn: the _BN of z — then the dgrad kernel also takes that layer's BatchNorm-backward sums (co, for
    _BN.backward(co=)) and a fourth value is returned"""
    lib = _hip.lib()
    sc, sh, relu = act
    mult, ldm, seed, step, p = _drop_args(drop)
    C = z.shape[1]
    da = torch.empty((z.shape[0], C), dtype=torch.float32, device=z.device)
    co = None
    if bn is not None and bn.M == M:
        co = torch.empty((5, C), dtype=torch.float32, device=z.device)
        need = lib.dal3_tr_colred_workspace_bytes(M, C)
        ws = _ws(need, z.device)
        _note("head2", M, 2, 128, 0.0, 4.0 * M * 258)
        _hip.check(lib.dal3_tr_head2_dgrad_bnbwd(_hip.ptr(dl), M, C, _hip.ptr(mult), ldm, seed, _hip.ptr(step), p, _hip.ptr(W),
                                                 W.stride(0), _hip.ptr(da), da.stride(0), _hip.ptr(z), z.stride(0), _hip.ptr(bn.scale),
                                                 _hip.ptr(bn.shift), _hip.ptr(bn.mu), _hip.ptr(bn.rstd), _hip.ptr(bn.gamma),
                                                 _hip.ptr(co[0]), _hip.ptr(co[1]), _hip.ptr(co[2]), _hip.ptr(co[3]), _hip.ptr(co[4]),
                                                 _hip.ptr(ws), need, _hip.stream()))
    else:
        _note("head2", M, 2, 128, 0.0, 4.0 * M * 130)
        _hip.check(lib.dal3_tr_head2_dgrad(_hip.ptr(dl), M, C, _hip.ptr(mult), ldm, seed, _hip.ptr(step), p, _hip.ptr(W), W.stride(0),
                                           _hip.ptr(da), da.stride(0), _hip.stream()))
    if z.shape[0] > M:
        da[M:].zero_()
    need = lib.dal3_tr_head2_wgrad_workspace_bytes(M)
    ws = _ws(need, z.device)
    f = torch.empty(258, dtype=torch.float32, device=z.device)                 # dW (2, 128), then db (2): views of one buffer
    _note("head2", M, 128, 2, 0.0, 4.0 * M * 130)
    _hip.check(lib.dal3_tr_head2_wgrad(_hip.ptr(dl), _hip.ptr(z), M, C, z.stride(0), _hip.ptr(sc), _hip.ptr(sh), int(relu), _hip.ptr(mult),
                                       ldm, seed, _hip.ptr(step), p, _hip.ptr(ws), need, _hip.ptr(f), _hip.stream()))
    if bn is not None:
        return da, f[:256].view(2, 128), f[256:258], co
    return da, f[:256].view(2, 128), f[256:258]


def _gather_at(z, arg, seg):
    """z[item*seg + arg[item,c], c] -> (items, C): the pre-BN value at each pooled point"""
    n_seg, C = arg.shape
    out = torch.empty((n_seg, C), dtype=torch.float32, device=z.device)
    _hip.check(_hip.lib().dal3_tr_gather_at(_hip.ptr(z), z.stride(0), _hip.ptr(arg), seg, n_seg, C, _hip.ptr(out), _hip.stream()))
    return out


def _segsum(x, seg, n_seg):
    C = x.shape[1]
    out = torch.empty((n_seg, C), dtype=torch.float32, device=x.device)
    _hip.check(_hip.lib().dal3_tr_segsum(_hip.ptr(x), x.stride(0), seg, C, _hip.ptr(out), n_seg, _hip.stream()))
    return out


def _pad_cols(w, n):
    """(r, c) -> contiguous (r, n), zero-padded"""
    out = torch.zeros((w.shape[0], n), dtype=torch.float32, device=w.device)
    out[:, :w.shape[1]] = w
    return out


def _points_major(pts, c_pad=32):
    """(B,C,N) logical -> (pad32(B*N), c_pad) row-major, zero-padded channels and rows"""
    B, C, N = pts.shape
    a0 = torch.zeros((_pad32(B * N), c_pad), dtype=torch.float32, device=pts.device)
    a0[:B * N, :C] = pts.transpose(2, 1).reshape(B * N, C)
    return a0


class _Rows:
    """the input points as (M, C) rows for the first-layer kernels (dal3_tr_conv1_*): the caller's own storage when it is
    point-major fp32, no (pad32(M), 32) zero-padded copy; .shape / .device as that padded buffer had them"""

    def __init__(self, pts):
        B, C, N = pts.shape
        x = pts.transpose(2, 1).reshape(B * N, C)
        self.x = x if (x.is_contiguous() and x.dtype == torch.float32) else x.contiguous().float()
        self.M, self.C = B * N, C
        self.shape = (_pad32(B * N), C)
        self.device = pts.device


def _conv1_ok(c_in, c_out):
    return 1 <= c_in <= 8 and c_out in (64, 128)


def _conv1_bn(rows, W, b, gamma, beta, stats):
    """z (pad32(M), c_out) = x W^T + b (padding rows: b) and its _BN over the M real rows, one pass (dal3_tr_conv1_bn_stats)"""
    lib = _hip.lib()
    M, Mp, c_in, c_out = rows.M, rows.shape[0], rows.C, W.shape[0]
    z = torch.empty((Mp, c_out), dtype=torch.float32, device=rows.device)
    bn = _BN.__new__(_BN)
    st = torch.empty((4, c_out), dtype=torch.float32, device=rows.device)
    bn.mu, bn.rstd, bn.scale, bn.shift = st[0], st[1], st[2], st[3]
    bn.gamma, bn.M = gamma.contiguous(), M
    rm, rv = stats if stats is not None else (None, None)
    need = lib.dal3_tr_conv1_workspace_bytes(Mp, c_out)
    ws = _ws(need, rows.device)
    _note("conv1", M, c_in, c_out, 2.0 * M * c_in * c_out, 4.0 * M * (c_in + c_out))
    _hip.check(lib.dal3_tr_conv1_bn_stats(_hip.ptr(rows.x), M, Mp, c_in, rows.x.stride(0), _hip.ptr(W), W.stride(0), _hip.ptr(b), c_out,
                                          _hip.ptr(z), z.stride(0), _hip.ptr(bn.gamma), _hip.ptr(beta.contiguous()), _hip.ptr(rm),
                                          _hip.ptr(rv), _MOM, _EPS, _hip.ptr(bn.mu), _hip.ptr(bn.rstd), _hip.ptr(bn.scale),
                                          _hip.ptr(bn.shift), _hip.ptr(ws), need, _hip.stream()))
    return z, bn


def _conv1_wgrad(dz, rows, c_out):
    """dW (c_out, c_in) = dz^T x over the M real rows (dal3_tr_conv1_wgrad)"""
    lib = _hip.lib()
    need = lib.dal3_tr_conv1_workspace_bytes(rows.M, c_out)
    ws = _ws(need, rows.device)
    dW = torch.empty((c_out, rows.C), dtype=torch.float32, device=rows.device)
    _note("conv1", rows.M, rows.C, c_out, 2.0 * rows.M * rows.C * c_out, 4.0 * rows.M * (rows.C + c_out))
    _hip.check(lib.dal3_tr_conv1_wgrad(_hip.ptr(dz), dz.stride(0), _hip.ptr(rows.x), rows.M, rows.C, rows.x.stride(0), c_out,
                                       _hip.ptr(ws), need, _hip.ptr(dW), _hip.stream()))
    return dW


def supported(pts):
    """CUDA tensors of any (B, N): the kernels tile the flattened point axis in 32s, the host pads the row buffers to
    a multiple of 32 and keeps the padding out of every sum (statistics over the real rows, dz = 0 on the rest)"""
    return pts.is_cuda


def _check(pts, what):
    _hip.require_gpu(pts, what)
    if pts.requires_grad:
        raise NotImplementedError(f"{what}: the HIP training path does not produce a gradient for the input points "
                                  "(the reference never asks for one); use train_backend='torch' for that")


def _moments_through(z_prev, bn_prev, W, b):
    """[sum z, sum z^2] of z = W a + b over all points, from the K x K second moments of the layer's INPUT a =
    relu(bn(z_prev)) instead of a pass over the C-channel output (C = 1024, K = 128 for conv5: 1 GB not re-read):
        mean(z) = W mean(a) + b,    var(z)_c = w_c^T Cov(a) w_c.
    Cov(a) is accumulated on CENTRED activations by the MFMA wgrad kernel (no mean^2 to cancel) and the small
    products are float64 (dal3_tr_pool_moments). Returns (sums, a, (Sc, centred), m1): the centred second moments and m1 =
    sum a for the backward shortcut (S = sum a a^T = Sc + m1 m1^T / M)."""
    K = z_prev.shape[1]
    M = bn_prev.M                                                          # real rows (z_prev is padded to 32s)
    # a = relu(bn(z_prev)) and its float64 column sums (fixed order) in ONE pass over the rows (dal3_tr_act_colsum)
    lib = _hip.lib()
    a = torch.empty((z_prev.shape[0], K), dtype=torch.float32, device=z_prev.device)
    sums_a = torch.empty(2 * K, dtype=torch.float64, device=z_prev.device)
    need = lib.dal3_tr_colred_workspace_bytes(M, K)
    ws = _ws(need, z_prev.device)
    sc, sh, relu = bn_prev.act
    _note("act", M, K, K, 0.0, 8.0 * M * K)
    _hip.check(lib.dal3_tr_act_colsum(_hip.ptr(z_prev), M, K, z_prev.stride(0), _hip.ptr(sc), _hip.ptr(sh), int(relu), _hip.ptr(a),
                                      a.stride(0), _hip.ptr(ws), need, _hip.ptr(sums_a), _hip.stream()))
    if a.shape[0] > M:
        a[M:].zero_()
    m1 = sums_a[:K]
    mean_a = m1 / M
    ac = a - mean_a.float()
    if a.shape[0] > M:
        ac[M:].zero_()
    Sc = _wgrad(ac, ac, K, K)
    del ac
    C = W.shape[0]
    if K in (64, 128, 256):                                                 # the float64 algebra in one launch
        sums = torch.empty(2 * C, dtype=torch.float64, device=a.device)
        Wc, bc = W.contiguous(), b.contiguous()
        _hip.check(_hip.lib().dal3_tr_pool_moments(_hip.ptr(Wc), Wc.stride(0), _hip.ptr(bc), _hip.ptr(m1), _hip.ptr(Sc), M, C, K,
                                                   _hip.ptr(sums), _hip.stream()))
    else:
        W64 = W.double()
        mu = W64 @ mean_a + b.double()
        var = ((W64 @ (Sc.double() / M)) * W64).sum(1).clamp_(min=0.0)
        sums = torch.cat([mu * M, (var + mu * mu) * M])
    return sums, a, (Sc, True), m1                                         # (Sc, True): S = Sc + m1 m1^T / M


def _pooled_layer_backward(z_prev, bn_prev, W, b, bn, zarg, g, arg, dg, N, cached=None):
    """Backward of  conv (W,b) -> BN (batch statistics) -> ReLU -> max over the N points of each item  WITHOUT the
    (M x C) gradient tensor of the conv output (C = 1024 for ins_seg: 1 GB at 64 x 4096 points, read three times).

    The max sends gradient to one point per (item, channel), so dy = d relu(bn(z)) is zero except at B*C entries,
    and the BN backward  dz = k1*(dy - k2 - xhat*k3)  is therefore AFFINE in z everywhere else:
        dz[p,c] = A_c + Bc_c * z[p,c] + k1_c*dy[p,c],   A = -k1*k2 + k1*k3*rstd*mu,   Bc = -k1*k3*rstd.
    With z[p] = W a[p] + b (a = the layer's input activation, K channels) both GEMMs of the layer collapse to K x K
    work plus B*C sparse terms:
        da[p]  = a[p] (W^T diag(Bc) W) + (A + Bc*b)^T W            + sum_{c: arg=p} k1_c dy_c W[c,:]
        dW[c]  = A_c m1 + Bc_c (W[c,:] S + b_c m1)                  + k1_c sum_items dy[item,c] a[arg[item,c]]
    with S = sum_p a[p] a[p]^T (a K x K Gram matrix: the wgrad of a K->K layer) and m1 = sum_p a[p]. The K x K
    products are float64 torch matmuls (tiny); the per-point work stays on the MFMA kernels.
    Returns (da (M,K), dW (C,K), dgamma, dbeta)."""
    K = z_prev.shape[1]
    M = bn_prev.M                                                          # real rows
    C = W.shape[0]
    dev = z_prev.device
    if cached is not None:
        a, S, m1 = cached                                                   # from _moments_through in the forward
    else:
        a = _act_dropout(z_prev, bn_prev.act, None)                        # (M,K), materialised once (K << C)
        if a.shape[0] > M:
            a[M:].zero_()
        S = m1 = None
    # the per-channel coefficients and kd = k1 * D in one launch (float64, fixed order of additions)
    nB = arg.shape[0]
    coef = torch.empty((4, C), dtype=torch.float64, device=dev)
    kd = torch.empty((nB, C), dtype=torch.float32, device=dev)
    _hip.check(_hip.lib().dal3_tr_pool_coef(_hip.ptr(dg.contiguous()), _hip.ptr(g.contiguous()), _hip.ptr(zarg.contiguous()),
                                            _hip.ptr(bn.mu), _hip.ptr(bn.rstd), _hip.ptr(bn.gamma), nB, C, M, _hip.ptr(coef),
                                            _hip.ptr(kd), _hip.stream()))
    dbeta, dgamma, A, Bc = coef[0], coef[1], coef[2], coef[3]
    fused = K in (64, 128, 256)                                             # the float64 algebra on lib3dal_hip.so
    Wc, bc = W.contiguous(), b.contiguous()
    if fused:
        G = torch.empty((K, K), dtype=torch.float32, device=dev)
        v = torch.empty(K, dtype=torch.float32, device=dev)
        need = _hip.lib().dal3_tr_pool_gv_workspace_bytes(K)
        gws = _ws(need, dev)
        _hip.check(_hip.lib().dal3_tr_pool_gv(_hip.ptr(coef), _hip.ptr(Wc), Wc.stride(0), _hip.ptr(bc), C, K, _hip.ptr(G),
                                              _hip.ptr(v), _hip.ptr(gws), need, _hip.stream()))
    else:
        W64, b64 = W.double(), b.double()
        G = (W64.t() @ (Bc[:, None] * W64)).float().contiguous()            # (K,K)
        v = ((A + Bc * b64) @ W64).float().contiguous()                     # (K,)
    da = _linear(a, G, K, K, K, transpose=True, bias=v)
    if S is None:
        S = (_wgrad(a, a, K, K), False)                                     # Gram matrix on the MFMA wgrad kernel
        m1 = _colred(a, 0, rows=M)[:K]
    S, centred = S
    # the two sparse terms (one pooled point per item and channel): scatter into da, gather for dW
    if 2 * N + C + 1 <= 16384 and C <= 4096 and K in (64, 128, 256):                      # one call of the library (LDS buckets per item)
        dWs = torch.empty((C, K), dtype=torch.float32, device=dev)
        Wc = W.contiguous()
        _hip.check(_hip.lib().dal3_tr_pool_sparse(_hip.ptr(arg), _hip.ptr(kd), _hip.ptr(Wc), Wc.stride(0), _hip.ptr(a),
                                                  a.stride(0), arg.shape[0], C, K, N, _hip.ptr(da), da.stride(0), _hip.ptr(dWs),
                                                  _hip.stream()))
    else:                                                                   # very long items: stock ops (sorted: deterministic)
        rows = (arg.long() + torch.arange(arg.shape[0], device=dev)[:, None] * N).reshape(-1)
        da.index_put_((rows,), (kd[:, :, None] * W[None]).reshape(-1, K), accumulate=True)
        dWs = (kd[:, :, None] * a[rows].reshape(arg.shape[0], C, K)).sum(0)
    if fused:
        dW = torch.empty((C, K), dtype=torch.float32, device=dev)
        m1c = m1.contiguous()
        _hip.check(_hip.lib().dal3_tr_pool_dw(_hip.ptr(coef), _hip.ptr(Wc), Wc.stride(0), _hip.ptr(bc), _hip.ptr(S), _hip.ptr(m1c),
                                              M, int(centred), _hip.ptr(dWs), C, K, _hip.ptr(dW), _hip.stream()))
    else:
        S64 = S.double() + (m1[:, None] * m1[None, :] / M if centred else 0.0)
        dW = (A[:, None] * m1[None] + Bc[:, None] * (W64 @ S64 + b64[:, None] * m1[None]) + dWs.double()).float()
    dgb = coef[:2].float()                                                  # (dbeta, dgamma) in one conversion
    return da, dW, dgb[1], dgb[0]


class _PointStack(torch.autograd.Function):
    """conv1..4 (+BN+ReLU) and the max over points of a `_PointHead` (static box_est / point_emb): (B,C,N) -> (B,512)"""

    @staticmethod
    def forward(ctx, pts, stats, *params):
        _check(pts, "pts")
        B, C_in, N = pts.shape
        fast1 = _conv1_ok(C_in, params[0].shape[0])           # conv1 on the first-layer kernels: no padded copy of the points
        a0 = a = _Rows(pts.detach()) if fast1 else _points_major(pts.detach())
        Ws, bns, zs = [], [], []
        act = None
        for k in range(4):
            W2 = params[4 * k].detach().reshape(params[4 * k].shape[0], -1)
            Ws.append(_pad_cols(W2, 32) if (k == 0 and not fast1) else W2.contiguous())
        Mp = a0.shape[0]
        # one packing launch for the stack: conv1..4 forward and the dgrads of conv2, conv3 (the pooled layer's goes through
        # its algebraic shortcut) — the weights are the same when the backward runs
        pk = _prepack([(Ws[k], Ws[k].shape[1], Ws[k].shape[0], False, Mp, 0, False, k > 0) for k in range(1 if fast1 else 0, 4)] +
                      [(Ws[k], Ws[k].shape[0], Ws[k].shape[1], True, Mp, 0, False, False) for k in (1, 2)], a0.device)
        if fast1:
            pk = [None] + pk
        for k in range(4):
            W, b, gamma, beta = (p.detach() for p in params[4 * k:4 * k + 4])
            W2 = Ws[k]
            if k == 0 and fast1:
                z, bn = _conv1_bn(a0, W2, b.contiguous(), gamma, beta, stats[0] if stats is not None else None)
            else:
                z, bn = _linear_bn(a, W2, W2.shape[1], W2.shape[0], act, b.contiguous(), 0, pk[k], gamma, beta,
                                   stats[k] if stats is not None else None, B * N)
            bns.append(bn)
            zs.append(z)
            a, act = z, bn.act
        g, arg = _segmax(zs[3], bns[3], N)
        zarg = _gather_at(zs[3], arg, N)
        zs[3] = None                                          # the pooled layer's output is not needed again
        biases = [params[4 * k + 1].detach().contiguous() for k in range(4)]
        # g is the OUTPUT: kept through save_for_backward. As a plain attribute it closes a reference cycle (g -> its
        # grad_fn -> ctx -> g) that only Python's cyclic collector breaks, and until it runs every step's activations
        # stay allocated: a DynamicModel run held 4-9 GB of dead steps and peaked at 13 GB instead of 8.
        ctx.save_for_backward(g)
        ctx.saved = (a0, Ws, bns, zs, arg, N, [tuple(p.shape) for p in params], zarg, biases, {1: pk[4], 2: pk[5]})
        return g

    @staticmethod
    def backward(ctx, dg):
        with deferred_wgrad_finals():                                       # (the weight gradients' second stages: one launch)
            return _PointStack._backward(ctx, dg)

    @staticmethod
    def _backward(ctx, dg):
        a0, Ws, bns, zs, arg, N, shapes, zarg, biases, pkT = _take_saved(ctx)
        (g,) = ctx.saved_tensors
        grads = [None] * 16
        zero = _zero_grads(shapes, [1, 5, 9, 13], a0.device)
        da = co = None
        for k in (3, 2, 1, 0):
            if k == 3:
                da, dW, dgam, dbet = _pooled_layer_backward(zs[2], bns[2], Ws[3], biases[3], bns[3], zarg, g, arg,
                                                            dg, N)
                grads[12] = dW.reshape(shapes[12])
                grads[13] = zero[13]
                grads[14], grads[15] = dgam, dbet
                continue
            dz, dgam, dbet = bns[k].backward(zs[k], da=da, co=co)
            co = None
            src, act = (zs[k - 1], bns[k - 1].act) if k > 0 else (a0, None)
            if isinstance(src, _Rows):
                dW = _conv1_wgrad(dz, src, Ws[k].shape[0])
            else:                                                       # (deferred only where grads[] takes dW as a view of itself)
                dW = _wgrad(dz, src, Ws[k].shape[0], Ws[k].shape[1], act, later=shapes[4 * k][1] == Ws[k].shape[1])
            grads[4 * k] = dW[:, :shapes[4 * k][1]].reshape(shapes[4 * k])
            grads[4 * k + 1] = zero[4 * k + 1]
            grads[4 * k + 2], grads[4 * k + 3] = dgam, dbet
            if k > 0:                                                   # (with the sums of the layer below where the shape allows)
                da, co = bns[k - 1].dgrad_with_sums(zs[k - 1], dz, Ws[k], Ws[k].shape[1], Ws[k].shape[0], pkT[k])
        return (None, None, *grads)


class _InsSeg(torch.autograd.Function):
    """PointNetInstanceSeg in train mode: (B,C,N) -> logits (B,N,2). params: conv1..5 then dconv1..4 as
    (W, b, gamma, beta) each, then dconv5 (W, b); drop: None, a (B*N,128) multiplier (0 or 1/(1-p)), or the key
    (seed, device step counter, p) of the on-device draw."""

    @staticmethod
    def forward(ctx, pts, drop, stats, *params):
        _check(pts, "pts")
        B, C_in, N = pts.shape
        M = B * N
        P = [p.detach() for p in params]
        fast1 = _conv1_ok(C_in, P[0].shape[0])                          # conv1 on the first-layer kernels (dal3_tr_conv1_*)
        a0 = _Rows(pts.detach()) if fast1 else _points_major(pts.detach())
        Mp = a0.shape[0]
        Ws, bns, zs = [], [], []
        # every layer's weight as the linear kernels take it, and ONE launch that puts them all into fragment order — the
        # forward calls and the transposed (dgrad) calls of the backward: 19 images instead of a ~4 us packing launch in
        # front of each of the 19 calls
        W2s = {}
        for k in (0, 1, 2, 3, 6, 7, 8):
            W2 = P[4 * k].reshape(P[4 * k].shape[0], -1)
            W2s[k] = _pad_cols(W2, 32) if (k == 0 and not fast1) else W2.contiguous()
        Wd1 = P[20].reshape(P[20].shape[0], -1).contiguous()            # dconv1 (512, 1088): columns 0..63 per point
        # dconv5 (128 -> 2) with its Dropout: three VALU kernels (dal3_tr_head2_*) when dconv4 has the reference's 128 channels;
        # otherwise through the MFMA kernels on a weight padded from 2 to 32 rows
        head2 = P[32].shape[0] == 128 and P[36].shape[0] == 2
        W5 = P[36].reshape(2, -1).contiguous()
        if not head2:
            W5 = torch.zeros((32, 128), dtype=torch.float32, device=pts.device)
            W5[:2] = P[36].reshape(2, 128)                              # rows padded to a tile
        fw = lambda k: (W2s[k], W2s[k].shape[1], W2s[k].shape[0], False, Mp, 0, False, k > 0)      # noqa: E731
        # (dg: the decoder's dgrads get their operand's amax from the BatchNorm backward in front of them, so they may take
        # the f16x3 image when the arithmetic says so and the shape qualifies — dconv3's and dconv2's do)
        tr = lambda k, acc=False, dg=False: (W2s[k], W2s[k].shape[0], W2s[k].shape[1], True, Mp, 0, acc, False, dg)   # noqa: E731
        order = ["f0", "f1", "f2", "f3", "fd1", "f6", "f7", "f8", "fd5", "td5", "t8", "t7", "t6", "td1", "t3", "t2", "t1"]
        # Two calls stay on the fp32 kernels in an f16x3 step too, because the fp32 kernel takes the reduction that follows in
        # its epilogue and the f16x3 kernel does not: dconv1's forward (64 -> 512, HBM-bound either way: 0.185 ms with its
        # statistics against 0.145 + 0.15 of a separate pass over the 537 MB output) and dconv3's dgrad (128 -> 256: 0.19
        # with dconv2's BatchNorm-backward sums against 0.11 + 0.13)
        specs = [fw(0), fw(1), fw(2), fw(3), (Wd1, 64, 512, False, Mp, N, False, True, "fp32"), fw(6), fw(7), fw(8),
                 (W5, 128, 32, False, Mp, 0, False, False), (W5, 32, 128, True, Mp, 0, False, False), tr(8, dg=True), tr(7, dg="fp32"),
                 tr(6, dg=True),
                 (Wd1, 512, 64, True, Mp, 0, False, False), tr(3), tr(2, True), tr(1)]
        skip = (("fd5", "td5") if head2 else ()) + (("f0",) if fast1 else ())      # (calls that read no packed image)
        specs = [sp for sp, name in zip(specs, order) if name not in skip]
        order = [name for name in order if name not in skip]
        pk = dict(zip(order, _prepack(specs, pts.device)))
        pk["arith"] = ARITH                                             # (the backward runs outside the forward's context)
        a, act = a0, None
        for k in range(4):                                              # conv1..4
            W, b, gamma, beta = P[4 * k:4 * k + 4]
            W2 = W2s[k]
            if k == 0 and fast1:
                z, bn = _conv1_bn(a0, W2, b.contiguous(), gamma, beta, stats[0] if stats is not None else None)
            else:
                z, bn = _linear_bn(a, W2, W2.shape[1], W2.shape[0], act, b.contiguous(), 0, pk[f"f{k}"], gamma, beta,
                                   stats[k] if stats is not None else None, M)
            Ws.append(W2)
            bns.append(bn)
            zs.append(z)
            a, act = z, bn.act
        # conv5 -> bn5 -> ReLU -> max over the crop's points. Its batch statistics come from the 128 x 128 second
        # moments of its INPUT (no pass over the 1024-channel output), so bn5's affine is known before the layer runs
        # and the layer + pooling are ONE kernel: the (B*N, 1024) output — 1 GB at 64 x 4096 points — is never
        # written. (N not a multiple of 32: the unfused pair, a tile must lie inside one crop.)
        W5c, b5c, gamma5, beta5 = P[16:20]
        W5c = W5c.reshape(W5c.shape[0], -1).contiguous()
        b5c = b5c.contiguous()
        sums, a4c, S4, m14 = _moments_through(zs[3], bns[3], W5c, b5c)
        bn5 = _BN((M, W5c.shape[0]), gamma5, beta5, *(stats[4] if stats is not None else (None, None)), sums=sums)
        if N % 32 == 0:
            g, arg = _linear_pool(zs[3], bns[3].act, W5c, b5c, bn5, N)
            zarg = torch.empty_like(g)                                  # the pre-BN value at each pooled point: one small kernel
            _hip.check(_hip.lib().dal3_tr_pool_zarg(_hip.ptr(arg), _hip.ptr(a4c), a4c.stride(0), _hip.ptr(W5c), W5c.stride(0),
                                                    _hip.ptr(b5c), B, W5c.shape[0], W5c.shape[1], N, _hip.ptr(zarg), _hip.stream()))
        else:
            z5 = _linear(zs[3], W5c, W5c.shape[1], W5c.shape[1], W5c.shape[0], act=bns[3].act, bias=b5c)
            g, arg = _segmax(z5, bn5, N)
            zarg = _gather_at(z5, arg, N)
            del z5
        Ws.append(W5c)
        bns.append(bn5)
        zs.append(None)
        # dconv1 on cat([out2, g.expand]): per-point part W[:, :64] out2, per-crop part W[:, 64:] g + b
        gb = torch.zeros(((a0.shape[0] - 1) // N + 1, 512), dtype=torch.float32, device=g.device)  # (padding rows index past B)
        torch.addmm(P[21], g, Wd1[:, 64:].t(), out=gb[:B])             # (B,512)
        z, bn = _linear_bn(zs[1], Wd1, 64, 512, bns[1].act, gb, N, pk["fd1"], P[22], P[23], stats[5] if stats is not None else None, M)
        Ws.append(Wd1)
        bns.append(bn)
        zs.append(z)
        a, act = z, bn.act
        for k in range(6, 9):                                           # dconv2..4
            W, b, gamma, beta = P[4 * k:4 * k + 4]
            W2 = W2s[k]
            z, bn = _linear_bn(a, W2, W2.shape[1], W2.shape[0], act, b.contiguous(), 0, pk[f"f{k}"], gamma, beta,
                               stats[k] if stats is not None else None, M)
            Ws.append(W2)
            bns.append(bn)
            zs.append(z)
            a, act = z, bn.act
        if torch.is_tensor(drop):
            drop = drop.contiguous()
        if head2:                                                       # Dropout sits between dbn4's ReLU and dconv5
            a4 = None
            logits = _head2_forward(zs[8], bns[8].act, drop, W5, P[37].contiguous(), M).reshape(B, N, 2)
        else:
            if torch.is_tensor(drop) and drop.shape[0] < zs[8].shape[0]:     # a supplied multiplier: pad its rows too
                drop = torch.cat([drop, drop.new_zeros((zs[8].shape[0] - drop.shape[0], drop.shape[1]))])
            a4 = _act_dropout(zs[8], bns[8].act, drop)
            b5 = torch.zeros(32, dtype=torch.float32, device=pts.device)
            b5[:2] = P[37]
            logits = _linear(a4, W5, 128, 128, 32, bias=b5, packed=pk["fd5"])[:M, :2].reshape(B, N, 2).contiguous()
        ctx.saved = (a0, Ws, bns, zs, g, arg, a4, drop, W5, N, [tuple(p.shape) for p in params], zarg, P[17].contiguous(),
                     (a4c, S4, m14), pk)
        if CAPTURE is not None:
            CAPTURE["ins_seg"] = {"zs": list(zs), "bns": list(bns), "g": g, "arg": arg, "M": M, "N": N, "B": B}
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        with deferred_wgrad_finals() as later:                              # (the weight gradients' second stages: one launch)
            return _InsSeg._backward(ctx, dlogits, later)

    @staticmethod
    def _backward(ctx, dlogits, later):
        a0, Ws, bns, zs, g, arg, a4, drop, W5, N, shapes, zarg, b_conv5, conv5_cache, pk = _take_saved(ctx)
        Mp = a0.shape[0]
        M = dlogits.shape[0] * dlogits.shape[1]
        dev = a0.device
        grads = [None] * 38
        zero = _zero_grads(shapes, [4 * k + 1 for k in range(9)], dev)
        co = None
        if a4 is None:                                                  # dconv5 + Dropout on the head2 kernels, dbn4's sums with them
            da, dW5, db5, co = _head2_backward(dlogits.reshape(M, 2).contiguous(), zs[8], bns[8].act, drop, W5, M, bn=bns[8])
            grads[36], grads[37] = dW5.reshape(shapes[36]), db5
        else:
            dzl = torch.zeros((Mp, 32), dtype=torch.float32, device=dev)
            dzl[:M, :2] = dlogits.reshape(M, 2)
            grads[36] = _wgrad(dzl, a4, 32, 128)[:2].reshape(shapes[36])
            grads[37] = dzl[:, :2].sum(0)
            da = _linear(dzl, W5, 128, 32, 128, transpose=True, packed=pk["td5"])
            if drop is not None:
                da = _act_dropout(da, None, drop)                       # the same multiplier, re-created from its key
        amaxes = torch.zeros(3 * 64, dtype=torch.int32, device=dlogits.device)
        for k in (8, 7, 6):                                             # dconv4..2
            # f16x3 step: dz's largest |value| comes with it (64 words), for the wgrad and — where its image is the f16x3
            # one — the dgrad of this layer
            amax = amaxes[64 * (8 - k):64 * (9 - k)] if pk.get("arith") == "f16x3" and zs[k].shape[1] % 64 == 0 else None
            dz, dgam, dbet = bns[k].backward(zs[k], da=da, amax=amax, co=co)
            grads[4 * k] = _wgrad(dz, zs[k - 1], Ws[k].shape[0], Ws[k].shape[1], bns[k - 1].act, amax=amax, later=True).reshape(shapes[4 * k])
            grads[4 * k + 1] = zero[4 * k + 1]
            grads[4 * k + 2], grads[4 * k + 3] = dgam, dbet
            if isinstance(pk[f"t{k}"], _X3Image):
                da, co = _linear(dz, Ws[k], Ws[k].shape[1], Ws[k].shape[0], Ws[k].shape[1], transpose=True, packed=pk[f"t{k}"],
                                 amax=amax), None
            else:                                                       # the dgrad and the sums of the layer below in one kernel
                da, co = bns[k - 1].dgrad_with_sums(zs[k - 1], dz, Ws[k], Ws[k].shape[1], Ws[k].shape[0], pk[f"t{k}"])
        # dconv1: per-point part against out2, per-crop part against g
        dz, dgam, dbet, dgb = bns[5].backward(zs[5], da=da, sum_seg=N, co=co)   # dgb (B,512): dz summed over each crop's points
        Wd1 = Ws[5]
        dWa = _wgrad(dz, zs[1], 512, 64, bns[1].act, later=True)
        if dgb is None:
            dgb = _segsum(dz, N, g.shape[0])
        dW1g = dgb.t() @ g
        grads[21] = zero[21]
        grads[22], grads[23] = dgam, dbet
        dg = dgb @ Wd1[:, 64:]                                          # (B,1024)
        da2_dec = _linear(dz, Wd1, Wd1.shape[1], 512, 64, transpose=True, packed=pk["td1"])
        # conv5..1
        da = co = None
        for k in (4, 3, 2, 1, 0):
            if k == 4:                                                  # conv5 -> max: the algebraic shortcut
                da, dW, dgam, dbet = _pooled_layer_backward(zs[3], bns[3], Ws[4], b_conv5, bns[4], zarg, g, arg, dg, N,
                                                            cached=conv5_cache)
                grads[16] = dW.reshape(shapes[16])
                grads[17] = zero[17]
                grads[18], grads[19] = dgam, dbet
                continue
            dz, dgam, dbet = bns[k].backward(zs[k], da=da, co=co)
            co = None
            src, act = (zs[k - 1], bns[k - 1].act) if k > 0 else (a0, None)
            if isinstance(src, _Rows):
                dW = _conv1_wgrad(dz, src, Ws[k].shape[0])
            else:                                                       # (deferred only where grads[] takes dW as a view of itself)
                dW = _wgrad(dz, src, Ws[k].shape[0], Ws[k].shape[1], act, later=shapes[4 * k][1] == Ws[k].shape[1])
            grads[4 * k] = dW[:, :shapes[4 * k][1]].reshape(shapes[4 * k])
            grads[4 * k + 1] = zero[4 * k + 1]
            grads[4 * k + 2], grads[4 * k + 3] = dgam, dbet
            if k == 2:                                                  # out2 also feeds the decoder
                da = _linear(dz, Ws[k], Ws[k].shape[1], Ws[k].shape[0], Ws[k].shape[1], transpose=True, out=da2_dec,
                             accumulate=True, packed=pk["t2"])
            elif k > 0:                                                 # (with the sums of the layer below: conv3's and conv1's)
                da, co = bns[k - 1].dgrad_with_sums(zs[k - 1], dz, Ws[k], Ws[k].shape[1], Ws[k].shape[0], pk[f"t{k}"])
        later.flush()                                                   # (dWa is read now)
        grads[20] = torch.cat([dWa, dW1g], 1).reshape(shapes[20])
        return (None, None, None, *grads)


FC_ROWS_KERNELS = True          # False: every FC tail through the per-point kernels (_FcTail; tests, A/B)


class _FcTail(torch.autograd.Function):
    """The per-item tail of a head in train mode — Linear -> BatchNorm1d (batch statistics over the B items) -> ReLU,
    n_bn times, then an optional last Linear without BN (fc3) — on the same training kernels as the per-point stacks,
    with rows = items: (B, c_in) -> (B, c_out); any B (rows padded to 32s, statistics over the B real ones). params: (W, b, gamma, beta) per BN layer, then (W, b) of the last layer if there is one.
    Reference: static_model.py:336-338, dynamic_model.py:247-248, :284-285, :306-311."""

    @staticmethod
    def forward(ctx, x, stats, n_bn, *params):
        P = [p.detach() for p in params]
        B = x.shape[0]
        a, act = x.detach().contiguous(), None
        if B % 32:
            a = torch.cat([a, a.new_zeros((_pad32(B) - B, a.shape[1]))])
        a_in = a
        Bp = a.shape[0]
        Ws, bns, zs = [P[4 * k].contiguous() for k in range(n_bn)], [], []
        Wp = None
        if len(P) > 4 * n_bn:                                     # the last layer, without a BatchNorm: rows padded to a tile
            W, b = P[4 * n_bn], P[4 * n_bn + 1]
            c_out = W.shape[0]
            cp = (c_out + 31) // 32 * 32
            Wp = torch.zeros((cp, W.shape[1]), dtype=torch.float32, device=W.device)
            Wp[:c_out] = W
            bp = torch.zeros(cp, dtype=torch.float32, device=W.device)
            bp[:c_out] = b
        # one packing launch for the tail: every layer forward and transposed (the backward's dgrads)
        specs = [(Ws[k], Ws[k].shape[1], Ws[k].shape[0], False, Bp, 0, False, k > 0) for k in range(n_bn)]
        specs += [(Ws[k], Ws[k].shape[0], Ws[k].shape[1], True, Bp, 0, False, False) for k in range(n_bn)]
        if Wp is not None:
            specs += [(Wp, Wp.shape[1], Wp.shape[0], False, Bp, 0, False, n_bn > 0), (Wp, Wp.shape[0], Wp.shape[1], True, Bp, 0, False, False)]
        pk = _prepack(specs, a.device)
        for k in range(n_bn):
            W, b, gamma, beta = P[4 * k:4 * k + 4]
            W = Ws[k]
            z = _linear(a, W, W.shape[1], W.shape[1], W.shape[0], act=act, bias=b.contiguous(), packed=pk[k])
            bn = _BN(z, gamma, beta, *(stats[k] if stats is not None else (None, None)), rows=B)
            bns.append(bn)
            zs.append(z)
            a, act = z, bn.act
        last = None
        if Wp is not None:
            out = _linear(a, Wp, Wp.shape[1], Wp.shape[1], cp, act=act, bias=bp, packed=pk[2 * n_bn])[:B, :c_out]
            last = (Wp, c_out)
        else:
            out = _act_dropout(a, act, None)[:B]                  # relu(bn(z)) of the last BN layer
        ctx.saved = (a_in, Ws, bns, zs, last, n_bn, [tuple(p.shape) for p in params], pk)
        return out.contiguous()

    @staticmethod
    def backward(ctx, dout):
        a_in, Ws, bns, zs, last, n_bn, shapes, pk = _take_saved(ctx)
        grads = [None] * len(shapes)
        dev = dout.device
        B, Bp = dout.shape[0], a_in.shape[0]
        zero = _zero_grads(shapes, [4 * k + 1 for k in range(n_bn)], dev)
        if last is not None:
            Wp, c_out = last
            dz = torch.zeros((Bp, Wp.shape[0]), dtype=torch.float32, device=dev)
            dz[:B, :c_out] = dout
            src, act = (zs[-1], bns[-1].act) if n_bn else (a_in, None)
            grads[4 * n_bn] = _wgrad(dz, src, Wp.shape[0], Wp.shape[1], act)[:c_out].reshape(shapes[4 * n_bn])
            grads[4 * n_bn + 1] = dout.sum(0)
            da = _linear(dz, Wp, Wp.shape[1], Wp.shape[0], Wp.shape[1], transpose=True, packed=pk[2 * n_bn + 1])
        elif Bp > B:
            da = torch.cat([dout, dout.new_zeros((Bp - B, dout.shape[1]))])
        else:
            da = dout.contiguous()
        for k in range(n_bn - 1, -1, -1):
            dz, dgam, dbet = bns[k].backward(zs[k], da=da)
            src, act = (zs[k - 1], bns[k - 1].act) if k > 0 else (a_in, None)
            grads[4 * k] = _wgrad(dz, src, Ws[k].shape[0], Ws[k].shape[1], act).reshape(shapes[4 * k])
            grads[4 * k + 1] = zero[4 * k + 1]                                 # a bias in front of a train-mode BN
            grads[4 * k + 2], grads[4 * k + 3] = dgam, dbet
            da = _linear(dz, Ws[k], Ws[k].shape[1], Ws[k].shape[0], Ws[k].shape[1], transpose=True, packed=pk[n_bn + k])
        return (da[:B], None, None, *grads)


class _BNc:
    """the BatchNorm constants of one FC layer as dal3_tr_fc_forward left them (what _BN holds, without its launches)"""

    def __init__(self, st, gamma, M):
        self.mu, self.rstd, self.scale, self.shift = st[0], st[1], st[2], st[3]
        self.gamma, self.M = gamma, M

    @property
    def act(self):
        return (self.scale, self.shift, True)


def _fc_forward(a, act, W, bias, c_out, bn=None, transpose=False):
    """z (B, c_out) = act(a) Wop^T + bias on dal3_tr_fc_forward; bn = (gamma, beta, running_mean, running_var): the layer's
    batch statistics in the same launch -> (z, _BNc)"""
    B, dev = a.shape[0], a.device
    sc, sh, relu = act if act is not None else (None, None, False)
    z = torch.empty((B, c_out), dtype=torch.float32, device=dev)
    c_in = W.shape[0] if transpose else W.shape[1]
    gamma = beta = rm = rv = None
    st = [None] * 4
    if bn is not None:
        gamma, beta, rm, rv = bn
        st = torch.empty((4, c_out), dtype=torch.float32, device=dev)
    _hip.check(_hip.lib().dal3_tr_fc_forward(_hip.ptr(a), B, c_in, a.stride(0), _hip.ptr(sc), _hip.ptr(sh), int(relu), _hip.ptr(W),
                                             W.stride(0), int(transpose), _hip.ptr(bias), c_out, _hip.ptr(z), z.stride(0),
                                             _hip.ptr(gamma), _hip.ptr(beta), _hip.ptr(rm), _hip.ptr(rv), _MOM, _EPS, _hip.ptr(st[0]),
                                             _hip.ptr(st[1]), _hip.ptr(st[2]), _hip.ptr(st[3]), _hip.stream()))
    return (z, _BNc(st, gamma, B)) if bn is not None else z


def _fc_backward_w(da, z, bn, src, act_in, W_shape):
    """of one FC layer: (dz, dW, db, dgamma, dbeta) from the gradient w.r.t. its output (relu(bn(z)), or — bn None — z itself,
    and dz is da) on dal3_tr_fc_backward_w"""
    B, C = da.shape
    dev = da.device
    c_in = src.shape[1]
    sc, sh, relu = act_in if act_in is not None else (None, None, False)
    dW = torch.empty((C, c_in), dtype=torch.float32, device=dev)
    db = torch.empty(C, dtype=torch.float32, device=dev)
    dz = dgb = None
    args = [None] * 7
    if bn is not None:
        dz = torch.empty((B, C), dtype=torch.float32, device=dev)
        dgb = torch.empty((2, C), dtype=torch.float32, device=dev)
        args = [bn.scale, bn.shift, bn.mu, bn.rstd, bn.gamma, dgb[0], dgb[1]]
    _hip.check(_hip.lib().dal3_tr_fc_backward_w(_hip.ptr(da), da.stride(0), B, C, _hip.ptr(z), z.stride(0) if z is not None else 0,
                                                *[_hip.ptr(t) for t in args], _hip.ptr(src), c_in, src.stride(0), _hip.ptr(sc),
                                                _hip.ptr(sh), int(relu), _hip.ptr(dz), dz.stride(0) if dz is not None else 0,
                                                _hip.ptr(dW), dW.stride(0), _hip.ptr(db), _hip.stream()))
    return (dz if bn is not None else da), dW.reshape(W_shape), db, (dgb[0] if bn is not None else None), (dgb[1] if bn is not None else None)


class _FcTailRows(torch.autograd.Function):
    """_FcTail for 2 <= B <= dal3_tr_fc_max_rows() items on the rows-are-items kernels (csrc/dal3_train_fc.hip): one launch per
    layer forward (product, bias, batch statistics, running statistics), two backward (BatchNorm backward + weight gradient;
    input gradient) — no padding, no packed weight images. The same arguments, the same results."""

    @staticmethod
    def forward(ctx, x, stats, n_bn, *params):
        P = [p.detach() for p in params]
        a_in = x.detach()
        a_in = a_in if (a_in.is_contiguous() and a_in.dtype == torch.float32) else a_in.contiguous().float()
        a, act, bns, zs, Ws = a_in, None, [], [], []
        for k in range(n_bn):
            W, b, gamma, beta = (t.contiguous() for t in P[4 * k:4 * k + 4])
            W2 = W.reshape(W.shape[0], -1)
            rm, rv = stats[k] if stats is not None else (None, None)
            z, bn = _fc_forward(a, act, W2, b, W2.shape[0], bn=(gamma, beta, rm, rv))
            Ws.append(W2)
            bns.append(bn)
            zs.append(z)
            a, act = z, bn.act
        last = None
        if len(P) > 4 * n_bn:
            W, b = P[4 * n_bn].contiguous(), P[4 * n_bn + 1].contiguous()
            last = W.reshape(W.shape[0], -1)
            out = _fc_forward(a, act, last, b, last.shape[0])
        else:
            out = _act_dropout(a, act, None)                       # relu(bn(z)) of the last BatchNorm layer
        ctx.saved = (a_in, Ws, bns, zs, last, n_bn, [tuple(p.shape) for p in params])
        return out

    @staticmethod
    def backward(ctx, dout):
        a_in, Ws, bns, zs, last, n_bn, shapes = _take_saved(ctx)
        grads = [None] * len(shapes)
        da = dout if (dout.is_contiguous() and dout.dtype == torch.float32) else dout.contiguous().float()
        if last is not None:
            src, act = (zs[-1], bns[-1].act) if n_bn else (a_in, None)
            _, grads[4 * n_bn], grads[4 * n_bn + 1], _, _ = _fc_backward_w(da, None, None, src, act, shapes[4 * n_bn])
            da = _fc_forward(da, None, last, None, last.shape[1], transpose=True)
        for k in range(n_bn - 1, -1, -1):
            src, act = (zs[k - 1], bns[k - 1].act) if k > 0 else (a_in, None)
            dz, grads[4 * k], grads[4 * k + 1], grads[4 * k + 2], grads[4 * k + 3] = _fc_backward_w(da, zs[k], bns[k], src, act, shapes[4 * k])
            if k > 0 or ctx.needs_input_grad[0]:
                da = _fc_forward(dz, None, Ws[k], None, Ws[k].shape[1], transpose=True)
            else:
                da = None
        return (da, None, None, *grads)


def fc_tail_supported(x):
    return x.is_cuda and x.dim() == 2 and x.shape[1] % 32 == 0


def fc_tail_train_forward(head, x):
    """`_PointHead.tail(x)` in train mode on the HIP training kernels; updates the BatchNorm running
    statistics like torch does"""
    fcs = head.TABLE["fcs"]
    params, bn_names = [], []
    for name, bn, _, _ in fcs:
        lin = getattr(head, name)
        params += [lin.weight, lin.bias]
        if bn:
            b = getattr(head, bn)
            params += [b.weight, b.bias]
            bn_names.append(bn)
    if any(bn is None for _, bn, _, _ in fcs[:-1]):
        raise RuntimeError("fc tail: only the last layer may come without a BatchNorm")
    stats = _bn_stats(head, bn_names)
    # the rows-are-items kernels hold a hidden layer's input-activation constants in LDS: layers behind the first must
    # have c_in <= dal3_tr_fc_max_act_cin() (the reference's tails: 512 / 256); a wider tail takes the per-point kernels
    narrow = all(ci <= _hip.lib().dal3_tr_fc_max_act_cin() for _, _, ci, _ in fcs[1:])
    if FC_ROWS_KERNELS and narrow and 2 <= x.shape[0] <= _hip.lib().dal3_tr_fc_max_rows():
        return _FcTailRows.apply(x, stats, len(bn_names), *params)
    return _FcTail.apply(x, stats, len(bn_names), *params)


def _bn_stats(mod, names):
    bns = [getattr(mod, n) for n in names]
    if bns:
        with torch.no_grad():                               # one launch for the module's counters instead of one each
            torch._foreach_add_([bn.num_batches_tracked for bn in bns], 1)
    return [(bn.running_mean, bn.running_var) for bn in bns]


def ins_seg_train_forward(ins_seg, pts, p_drop=0.5, drop_mask=None):
    """train-mode PointNetInstanceSeg.forward on the HIP training kernels: logits (B,N,2), autograd-connected to the
    module's parameters; updates the BatchNorm running statistics like torch does. drop_mask: optional (B*N,128)
    multiplier replacing the random Dropout draw (tests)."""
    drop = drop_mask
    if drop is None and p_drop > 0:
        # the draw happens inside dal3_tr_act_dropout, keyed on a seed taken from torch's CPU generator (so
        # torch.manual_seed() makes runs repeatable) and a device-side step counter (so a captured step draws afresh
        # on every hipGraph replay); the backward re-creates the multiplier from the same key
        step = draw_step(ins_seg, pts.device)
        step.add_(1)
        # the key carries a SNAPSHOT of the counter, not the live tensor: the backward re-creates the multiplier from the
        # key, and a second train-mode forward of the same module before this graph's backward (loss(model(a)) +
        # loss(model(b)), a recompute) has moved the live counter on by then. The clone is an ordinary device
        # op: captured into a hipGraph it copies the replay's counter value, so a replayed step still draws afresh.
        drop = (int(torch.empty((), dtype=torch.int64).random_().item()) & 0x7FFFFFFFFFFFFFFF, step.clone(), float(p_drop))
    params = []
    for conv, bn in ins_seg.pairs():
        params += [conv.weight, conv.bias] + ([bn.weight, bn.bias] if bn is not None else [])
    stats = _bn_stats(ins_seg, ["bn1", "bn2", "bn3", "bn4", "bn5", "dbn1", "dbn2", "dbn3", "dbn4"])
    return _InsSeg.apply(pts, drop, stats, *params)


def point_stack_train_forward(head, pts):
    """conv1..4 + max of a point head (static box_est, point_emb) in train mode: (B,512) global feature"""
    params = []
    for k in range(1, 5):
        conv, bn = getattr(head, f"conv{k}"), getattr(head, f"bn{k}")
        params += [conv.weight, conv.bias, bn.weight, bn.bias]
    stats = _bn_stats(head, ["bn1", "bn2", "bn3", "bn4"])
    return _PointStack.apply(pts, stats, *params)
