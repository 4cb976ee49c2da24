#!/usr/bin/env python3
"""Diagnostic: run a -DDAL3_STAMP build of the 16-bit encode kernel and print s_memtime ticks per phase per wave.
  bash tools/build_variant.sh stamp "-DDAL3_STAMP" && python tools/stamps_lp_enc.py variants/stamp.so"""
import ctypes as C, importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hip = importlib.import_module("3dal_pytorch_amd._hip")
synth = importlib.import_module("3dal_pytorch_amd.synth")
sm = importlib.import_module("3dal_pytorch_amd.static_model")
lib = C.CDLL(os.path.abspath(sys.argv[1]))
for name, (res, a) in hip.SIGNATURES.items():
    fn = getattr(lib, name); fn.restype, fn.argtypes = res, a
B, N, DT = 4096, 1024, 1
dev = torch.device("cuda:0")
model = sm.StaticModelOneBoxEst()
model.load_state_dict({k: torch.as_tensor(v) for k, v in synth.state_dict("static_one").items()})
model = model.to(dev).eval()
pts = torch.from_numpy(synth.static_crops(256, N)[0]).to(dev).repeat(16, 1, 1).contiguous().transpose(2, 1)
arr = (hip.Layer * 10)(*[hip.layer_struct(c, b) for c, b in model.ins_seg.pairs()])
need = C.c_size_t(0)
lib.dal3_pack_weights(0, arr, 10, DT, None, C.byref(need), None)
w = torch.zeros(need.value, dtype=torch.uint8, device=dev)
assert lib.dal3_pack_weights(0, arr, 10, DT, hip.ptr(w), C.byref(need), hip.stream()) == 0
stamps = torch.zeros(4096 * 4 * 8, dtype=torch.int64, device=dev)
lib.dal3_debug_set_stamps_lp_enc.argtypes = [C.c_void_p]
assert lib.dal3_debug_set_stamps_lp_enc(stamps.data_ptr()) == 0
g = torch.zeros((B, 1024), device=dev)
for _ in range(3):
    g.zero_()
    assert lib.dal3_ins_seg_encode(hip.ptr(w), DT, 3, hip.bcn(pts), B, N, hip.ptr(g), hip.stream()) == 0
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(-1, 8)
s = s[s[:, 4] > 0]
d = np.diff(s[:, :5], axis=1).astype(np.float64)
names = ["ring start, points, conv1 (fp32)", "acquire + conv2..conv4 (512 MFMA)", "conv5 + max (2048 MFMA)", "barrier + flush of the maxima"]
mf = [0, 512, 2048, 0]
tot = (s[:, 4] - s[:, 0]).mean()
print("waves sampled", len(s), " ticks per workgroup (stamp 0 -> 4)", tot)
for i, n in enumerate(names):
    print(f"{n:40s} mean {d[:, i].mean():9.0f}  p10 {np.percentile(d[:, i], 10):8.0f}  p90 {np.percentile(d[:, i], 90):8.0f}"
          f"  share {d[:, i].mean() / tot:6.1%}  ticks/MFMA {d[:, i].mean() / mf[i] if mf[i] else 0:6.1f}")
g = stamps.cpu().numpy().reshape(-1, 4, 8)
seam = (g[256:4096, :, 0] - g[:4096 - 256, :, 4]).astype(np.float64)
seam = seam[(g[256:4096, :, 0] > 0) & (g[:4096 - 256, :, 4] > 0)]
print("seam between consecutive groups of a workgroup: mean", seam.mean(), "p10", np.percentile(seam, 10), "p90", np.percentile(seam, 90))
