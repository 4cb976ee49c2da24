"""debug: where do the hip / torch train-mode masks differ (dynamic, B=3) and how close to a tie are those points"""
import importlib, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
from _common import build_model, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 3
res = {}
for backend in ("hip", "torch", "torch64"):
    model = build_model("dynamic", synth.state_dict("dynamic", seed=24)).train()
    model.train_backend = "torch" if backend != "hip" else "hip"
    model.sampler = "numpy"
    model.ins_seg.dropout.p = 0.0
    p, bx, _, g = synth.dynamic_items(B, n_per_frame=256, seed=24)
    args = [torch.from_numpy(p).cuda().transpose(2, 1), torch.from_numpy(bx).cuda().transpose(2, 1), torch.from_numpy(g).cuda()]
    if backend == "torch64":
        model = model.double(); args = [a.double() for a in args]
    np.random.seed(77)
    o = model(*args)
    res[backend] = (o["logits"].detach().double(), o["mask"].clone())
for a in ("hip", "torch"):
    d = res[a][1] != res["torch64"][1]
    lg = res["torch64"][0]
    gap = (lg[..., 0] - lg[..., 1]).abs()
    print(a, "flips vs f64:", int(d.sum()), "gap at flips:", gap[d.bool().reshape(gap.shape)].tolist() if d.any() else [],
          "max logit err", float((res[a][0] - lg).abs().max()), "logit scale", float(lg.abs().max()))
