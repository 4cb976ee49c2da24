#!/usr/bin/env python3
"""Diagnostic: run a -DDAL3_STAMP build of the 16-bit decode kernel and print s_memtime ticks per phase per wave.
  bash tools/build_variant.sh stamp "-DDAL3_STAMP" && python tools/stamps_lp.py variants/stamp.so"""
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
stamps = torch.zeros(4096 * 4 * 40, dtype=torch.int64, device=dev)
lib.dal3_debug_set_stamps_lp.argtypes = [C.c_void_p]
assert lib.dal3_debug_set_stamps_lp(stamps.data_ptr()) == 0
gb = torch.zeros((B, 512), device=dev)
logits = torch.empty((B, N, 2), device=dev); mask = torch.empty((B, N), dtype=torch.uint8, device=dev)
for _ in range(3):
    assert lib.dal3_ins_seg_decode(hip.ptr(w), DT, 3, hip.bcn(pts), B, N, hip.ptr(gb), hip.ptr(logits), hip.ptr(mask), hip.stream()) == 0
torch.cuda.synchronize()
allst = stamps.cpu().numpy()
sub = allst[4096 * 4 * 8:].reshape(-1, 32)
s = allst[:4096 * 4 * 8].reshape(-1, 8)
sub = sub[s[:, 6] > 0]
s = s[s[:, 6] > 0]
d = np.diff(s[:, :7], axis=1).astype(np.float64)
names = ["prologue: points, conv1, conv2, dconv1 chunk 0 (24 MFMA)", "main loop dconv1+dconv2 (640 MFMA)",
         "pack a2 -> 16 bit", "dconv3 (128 MFMA)", "dconv4 (64 MFMA)", "dconv5 (16 MFMA) + store"]
mf = [24, 640, 0, 128, 64, 16]
tot = (s[:, 6] - s[:, 0]).mean()
print("waves sampled", len(s), " ticks per group", tot)
for i, n in enumerate(names):
    print(f"{n:58s} mean {d[:, i].mean():9.0f}  p10 {np.percentile(d[:, i], 10):8.0f}  p90 {np.percentile(d[:, i], 90):8.0f}"
          f"  share {d[:, i].mean() / tot:6.1%}  ticks/MFMA {d[:, i].mean() / mf[i] if mf[i] else 0:6.1f}")
# consecutive groups of one workgroup: stamp 0 of group g+256 minus stamp 6 of group g (same wave) = the seam
g = allst[:4096 * 4 * 8].reshape(-1, 4, 8)
seam = (g[256:4096, :, 0] - g[:4096 - 256, :, 6]).astype(np.float64)
seam = seam[(g[256:4096, :, 0] > 0) & (g[:4096 - 256, :, 6] > 0)]
print("seam between consecutive groups of a workgroup: mean", seam.mean(), "p90", np.percentile(seam, 90))
# slot 7: per group, ticks each wave spent inside s_barrier (low 32 bits) and inside the counted vmcnt/lgkmcnt wait
raw = g[:, :, 7]
raw = raw[g[:, :, 6] > 0]
bar, wt = (raw & 0xffffffff).astype(np.float64), (raw >> 32).astype(np.float64)
print("per group: ticks in s_barrier, by wave", bar.reshape(-1, 4).mean(0).round(0), " in the counted wait", wt.reshape(-1, 4).mean(0).round(0))

# sub-phase stamps (LP_SUB): differences along the kernel's order
if sub.any():
    order = [("stamp0", s[:, 0]), ("conv1 fp32 + pack", sub[:, 8]), ("conv2", sub[:, 9]), ("a2 <- bias", sub[:, 10]),
             ("dconv1 chunk 0 + packs", sub[:, 11]), ("acquire + first reads", s[:, 1]), ("main iter 0", sub[:, 12]),
             ("main iter 1", sub[:, 13]), ("main iter 2..7 + burst", s[:, 2]), ("prefetch + pack a2", s[:, 3]),
             ("d3 t0: to 1st MFMA", sub[:, 20]), ("d3 t0: MFMA 1-8", sub[:, 16]), ("d3 t0: MFMA 9-16", sub[:, 17]), ("d3 t0: MFMA 17-24", sub[:, 18]),
             ("dconv3 tile 0 (rest)", sub[:, 0]), ("dconv3 tile 1 (EARLY)", sub[:, 1]), ("dconv3 tile 2", sub[:, 2]),
             ("dconv3 tile 3 (EARLY)", sub[:, 3]), ("publish_gb", s[:, 4]), ("d4 t0: to 1st MFMA", sub[:, 21]), ("d4 t0: MFMA 1-8", sub[:, 19]), ("dconv4 tile 0 (rest)", sub[:, 4]),
             ("dconv4 tile 1", sub[:, 5]), ("dconv4 tile 2", sub[:, 6]), ("dconv4 tile 3", s[:, 5]),
             ("dconv5 (EARLY)", sub[:, 7]), ("store", s[:, 6])]
    for (n0, a), (n1, b) in zip(order[:-1], order[1:]):
        d = (b - a).astype(np.float64)
        print(f"  {n1:34s} mean {d.mean():8.0f}  p10 {np.percentile(d, 10):8.0f}  p90 {np.percentile(d, 90):8.0f}")
