import copy, importlib, sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _common import build_model, synth
import test_gpu_train as T
train = importlib.import_module("3dal_pytorch_amd.train")
B, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4, 256)
model = build_model("static_one", synth.state_dict("static_one", seed=21))
ours = model.ins_seg.train()
ref32 = copy.deepcopy(ours)
ref64 = copy.deepcopy(ours).double()
pts = torch.from_numpy(synth.static_crops(B, N, seed=21)[0]).cuda().transpose(2, 1)
mul = (torch.from_numpy(synth.uniform(21, "drop", (B, N, 128))).cuda() >= 0.5).float() * 2.0
weight = torch.from_numpy(synth.normal(21, "lw", (B, N, 2)).astype(np.float32)).cuda()
w32 = T._ref_ins_seg(ref32, pts, mul.transpose(2, 1)); (w32 * weight).sum().backward()
if len(sys.argv) > 3: torch.cuda.empty_cache(); junk = torch.full((1 << 28,), float('nan'), device='cuda'); del junk
w64 = T._ref_ins_seg(ref64, pts.double(), mul.transpose(2, 1).double()); (w64 * weight.double()).sum().backward()
got = train.ins_seg_train_forward(ours, pts, drop_mask=mul.reshape(B * N, 128)); (got * weight).sum().backward()
print("logits err ours/torch32 vs f64:", float((got.double() - w64).abs().max()), float((w32.double() - w64).abs().max()))
for (name, p), (_, q), (_, r) in zip(ours.named_parameters(), ref32.named_parameters(), ref64.named_parameters()):
    t = r.grad
    sc = float(t.abs().max()) + 1e-30
    print(f"{name:16s} |g|max {sc:10.3e}  ours-f64 {float((p.grad.double()-t).abs().max())/sc:9.2e}  torch32-f64 {float((q.grad.double()-t).abs().max())/sc:9.2e}")
