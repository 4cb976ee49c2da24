#!/usr/bin/env python3
"""Diagnostic: run a -DDAL3_STAMP build of the decode kernel and print cycles per phase per wave."""
import ctypes as C, importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hip = importlib.import_module("3dal_pytorch_amd._hip")
synth = importlib.import_module("3dal_pytorch_amd.synth")
sm = importlib.import_module("3dal_pytorch_amd.static_model")
lib = C.CDLL(os.path.abspath(sys.argv[1]))
for name, (res, a) in hip.SIGNATURES.items():
    fn = getattr(lib, name); fn.restype, fn.argtypes = res, a
B, N = 4096, 1024
dev = torch.device("cuda:0")
model = sm.StaticModelOneBoxEst()
model.load_state_dict({k: torch.as_tensor(v) for k, v in synth.state_dict("static_one").items()})
model = model.to(dev).eval()
pts = torch.from_numpy(synth.static_crops(256, N)[0]).to(dev).repeat(16, 1, 1).contiguous().transpose(2, 1)
arr = (hip.Layer * 10)(*[hip.layer_struct(c, b) for c, b in model.ins_seg.pairs()])
need = C.c_size_t(0)
lib.dal3_pack_weights(0, arr, 10, 0, None, C.byref(need), None)
w = torch.zeros(need.value, dtype=torch.uint8, device=dev)
assert lib.dal3_pack_weights(0, arr, 10, 0, hip.ptr(w), C.byref(need), hip.stream()) == 0
stamps = torch.zeros(2048 * 4 * 8, dtype=torch.int64, device=dev)
lib.dal3_debug_set_stamps.argtypes = [C.c_void_p]
assert lib.dal3_debug_set_stamps(stamps.data_ptr()) == 0
gb = torch.zeros((B, 512), device=dev)
logits = torch.empty((B, N, 2), device=dev); mask = torch.empty((B, N), dtype=torch.uint8, device=dev)
for _ in range(3):
    assert lib.dal3_ins_seg_decode(hip.ptr(w), 0, 3, hip.bcn(pts), B, N, hip.ptr(gb), hip.ptr(logits), hip.ptr(mask), hip.stream()) == 0
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(-1, 8)
s = s[s[:, 4] > 0]
d = np.diff(s[:, :5], axis=1).astype(np.float64)
names = ["prologue (points, conv1-2)", "main loop dconv1+2 (2600 MFMA)", "relu a2", "dconv3+4 (768 MFMA)", "dconv5 + store"]
ideal = [68 * 64, 2600 * 64, 0, 768 * 64, 0]
print("waves sampled", len(s))
for i, n in enumerate(names):
    print(f"{n:36s} mean {d[:, i].mean():10.0f} cyc  p10 {np.percentile(d[:, i], 10):9.0f}  p90 {np.percentile(d[:, i], 90):9.0f}   MFMA-ideal {ideal[i]}")
tot = (s[:, 4] - s[:, 0]).mean()
print("total per wave", tot, "ideal", 3396 * 64, "ratio", 3396 * 64 / tot)
# gap between consecutive waves on one slot is not visible here; wall/wave from the launch:
