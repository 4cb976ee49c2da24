#!/usr/bin/env python3
"""A/B of the per-crop FC kernel (dal3_ins_seg_global_bias: 1024 -> 512 per crop) between library builds, one process:
  python tools/ab_fc.py build_a.so build_b.so [--B 4096]"""
import ctypes as C, importlib, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hip = importlib.import_module("3dal_pytorch_amd._hip")
synth = importlib.import_module("3dal_pytorch_amd.synth")
sm = importlib.import_module("3dal_pytorch_amd.static_model")
libs_p = [a for a in sys.argv[1:] if a.endswith(".so")]
B = int(sys.argv[sys.argv.index("--B") + 1]) if "--B" in sys.argv else 4096
dev = torch.device("cuda:0")
model = sm.StaticModelOneBoxEst()
model.load_state_dict({k: torch.as_tensor(v) for k, v in synth.state_dict("static_one").items()})
model = model.to(dev).eval()
st = hip.stream()
g = torch.rand((B, 1024), device=dev)
outs, res = [], {}
libs = []
for p in libs_p:
    h = C.CDLL(os.path.abspath(p))
    for name, (r, a) in hip.SIGNATURES.items():
        if hasattr(h, name):
            fn = getattr(h, name); fn.restype, fn.argtypes = r, a
    pairs = model.ins_seg.pairs()
    arr = (hip.Layer * len(pairs))(*[hip.layer_struct(c, b) for c, b in pairs])
    need = C.c_size_t(0)
    h.dal3_pack_weights(hip.HEAD_INS_SEG, arr, len(pairs), 0, None, C.byref(need), None)
    w = torch.zeros(need.value, dtype=torch.uint8, device=dev)
    assert h.dal3_pack_weights(hip.HEAD_INS_SEG, arr, len(pairs), 0, hip.ptr(w), C.byref(need), st) == 0
    gb = torch.empty((B, 512), device=dev)
    libs.append((os.path.basename(p), h, w, gb))
    res[os.path.basename(p)] = []
for r in range(7):
    for name, h, w, gb in libs:
        for _ in range(2):
            h.dal3_ins_seg_global_bias(hip.ptr(w), 0, hip.ptr(g), B, hip.ptr(gb), st)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            h.dal3_ins_seg_global_bias(hip.ptr(w), 0, hip.ptr(g), B, hip.ptr(gb), st)
        b.record(); b.synchronize()
        res[name].append(a.elapsed_time(b) / 10 * 1e3)
print("bitwise equal:", all(torch.equal(libs[0][3], l[3]) for l in libs))
for name in res:
    print(f"{name:20s} {statistics.median(res[name]):8.1f} us  ({2.0 * B * 1024 * 512 / statistics.median(res[name]) / 1e6:6.1f} TFLOP/s)")
