#!/usr/bin/env python3
"""A/B timing of library builds in ONE process (cdna guide rule 24): interleaved rounds of the
three shared-MLP kernels, median and min per build, outputs cross-checked against the first build.

  python tools/ab_kernels.py build_a.so build_b.so ... [--B 4096] [--N 1024] [--rounds 7]
"""
import argparse
import ctypes as C
import importlib
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hip = importlib.import_module("3dal_pytorch_amd._hip")
synth = importlib.import_module("3dal_pytorch_amd.synth")
sm = importlib.import_module("3dal_pytorch_amd.static_model")

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--B", type=int, default=4096)
ap.add_argument("--N", type=int, default=1024)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--dtype", type=int, default=0, help="0 fp32, 1 bf16, 2 fp16")
args = ap.parse_args()
B, N = args.B, args.N
DT = args.dtype
dev = torch.device("cuda:0")


def load(path):
    h = C.CDLL(os.path.abspath(path))
    for name, (res, a) in hip.SIGNATURES.items():
        if not hasattr(h, name):                      # an older build may predate some entry points
            continue
        fn = getattr(h, name)
        fn.restype, fn.argtypes = res, a
    return h


model = sm.StaticModelOneBoxEst()
model.load_state_dict({k: torch.as_tensor(v) for k, v in synth.state_dict("static_one").items()})
model = model.to(dev).eval()
pts_np, init_np, _ = synth.static_crops(min(B, 512), N)
reps = (B + pts_np.shape[0] - 1) // pts_np.shape[0]
pts = torch.from_numpy(pts_np).to(dev).repeat(reps, 1, 1)[:B].contiguous().transpose(2, 1)
obj = pts.transpose(2, 1)[:, :512, :].contiguous().transpose(2, 1)
x, xo = hip.bcn(pts), hip.bcn(obj)
st = hip.stream()
libs = [(os.path.basename(p), load(p)) for p in args.libs]
state = []
for name, lib in libs:
    need = C.c_size_t(0)
    def pack(kind, pairs):
        arr = (hip.Layer * len(pairs))(*[hip.layer_struct(c, b) for c, b in pairs])
        lib.dal3_pack_weights(kind, arr, len(pairs), DT, None, C.byref(need), None)
        buf = torch.zeros(need.value, dtype=torch.uint8, device=dev)
        rc = lib.dal3_pack_weights(kind, arr, len(pairs), DT, hip.ptr(buf), C.byref(need), st)
        assert rc == 0, lib.dal3_last_error()
        return buf
    w_seg = pack(hip.HEAD_INS_SEG, model.ins_seg.pairs())
    w_box = pack(hip.HEAD_STATIC_BOX_EST, model.box_est.pairs())
    g = torch.zeros((B, 1024), device=dev)
    gb = torch.empty((B, 512), device=dev)
    logits = torch.empty((B, N, 2), device=dev)
    mask = torch.empty((B, N), dtype=torch.uint8, device=dev)
    ws = torch.empty(lib.dal3_point_head_workspace_bytes(B), dtype=torch.uint8, device=dev)
    bp = torch.empty((B, 39), device=dev)
    d = dict(w_seg=w_seg, w_box=w_box, g=g, gb=gb, logits=logits, mask=mask, ws=ws, bp=bp)
    d["enc"] = lambda lib=lib, d=d: lib.dal3_ins_seg_encode(hip.ptr(d["w_seg"]), DT, 3, x, B, N, hip.ptr(d["g"]), st)
    d["dec"] = lambda lib=lib, d=d: lib.dal3_ins_seg_decode(hip.ptr(d["w_seg"]), DT, 3, x, B, N, hip.ptr(d["gb"]),
                                                            hip.ptr(d["logits"]), hip.ptr(d["mask"]), st)
    d["head"] = lambda lib=lib, d=d: lib.dal3_point_head_forward(hip.HEAD_STATIC_BOX_EST, hip.ptr(d["w_box"]), DT, xo, B, 512,
                                                                 hip.ptr(d["bp"]), 39, hip.ptr(d["ws"]), d["ws"].numel(), st)
    assert d["enc"]() == 0, lib.dal3_last_error()
    assert lib.dal3_ins_seg_global_bias(hip.ptr(w_seg), DT, hip.ptr(g), B, hip.ptr(gb), st) == 0
    assert d["dec"]() == 0 and d["head"]() == 0, lib.dal3_last_error()
    state.append(d)
torch.cuda.synchronize()
ref = state[0]
for (name, _), d in zip(libs, state):
    same = (torch.equal(d["g"], ref["g"]), torch.equal(d["logits"], ref["logits"]), torch.equal(d["bp"], ref["bp"]))
    dl = (d["logits"] - ref["logits"]).abs().max().item() / ref["logits"].abs().max().item()
    print(f"{name}: bitwise-equal to first (g, logits, box_pred) = {same}, logits rel diff {dl:.2e}")


def timed(fn, iters=5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / iters


res = {(n, k): [] for n, _ in libs for k in ("enc", "dec", "head")}
for r in range(args.rounds):
    for k in ("enc", "dec", "head"):
        for (name, _), d in zip(libs, state):
            res[(name, k)].append(timed(d[k]))
print(f"B={B} N={N}: ms per launch, median (min) over {args.rounds} interleaved rounds")
for name, _ in libs:
    row = "  ".join(f"{k} {statistics.median(res[(name, k)]):7.3f} ({min(res[(name, k)]):7.3f})" for k in ("enc", "dec", "head"))
    tot = sum(statistics.median(res[(name, k)]) for k in ("enc", "dec", "head"))
    print(f"{name:40s} {row}   sum {tot:7.3f}")
