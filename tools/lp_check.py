#!/usr/bin/env python3
"""Error of the bf16 / fp16 paths against the fp32 path (same weights, same crops) and their speed."""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
synth = importlib.import_module("3dal_pytorch_amd.synth")
sm = importlib.import_module("3dal_pytorch_amd.static_model")
dm = importlib.import_module("3dal_pytorch_amd.dynamic_model")
dev = torch.device("cuda:0")


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


B, N = int(os.environ.get("B", 256)), 1024
pts_np, init_np, gt_np = synth.static_crops(B, N)
model = sm.StaticModelTwoBoxEst()
model.load_state_dict({k: torch.as_tensor(v) for k, v in synth.state_dict("static_two").items()})
model = model.to(dev).eval()
pts = torch.from_numpy(pts_np).to(dev).transpose(2, 1)
init, gt = torch.from_numpy(init_np).to(dev), torch.from_numpy(gt_np).to(dev)
with torch.no_grad():
    lg = model(pts, init, gt)["logits"]
    model.ins_seg.dconv5.bias[1] -= (lg[:, :, 1] - lg[:, :, 0]).mean()
ref = model._run(pts, init, gt)
choice = None
for prec in ("bf16", "fp16"):
    model.precision = prec
    o = model._run(pts, init, gt)
    torch.cuda.synchronize()
    m_ref, m = ref["mask"].bool(), o["mask"].bool()
    print(f"[static {prec}] logits rel {rel(o['logits'], ref['logits']):.3e}  mask agreement {(m == m_ref).float().mean().item():.5f}"
          f"  counts max diff {(o['counts'] - ref['counts']).abs().max().item()}")
    # teacher-force the fp32 mask + the same device draws
    o2 = model._run(pts, init, gt, mask_override=ref["mask"])
    for k in ("bp1", "bp2", "boxes7", "c2"):
        print(f"    teacher-forced {k}: rel {rel(o2[k], ref[k]):.3e}  abs {float((o2[k]-ref[k]).abs().max()):.3e}")
    o3 = model._run(pts, init, gt)
    print("    deterministic:", torch.equal(o3["logits"], o["logits"]), torch.equal(o3["boxes7"], o["boxes7"]))
model.precision = "fp32"

Bd = max(4, B // 16)
p, bx, i8, g7 = synth.dynamic_items(Bd)
dmodel = dm.DynamicModel()
dmodel.load_state_dict({k: torch.as_tensor(v) for k, v in synth.state_dict("dynamic").items()})
dmodel = dmodel.to(dev).eval()
dp, dbx, di8 = torch.from_numpy(p).to(dev).transpose(2, 1), torch.from_numpy(bx).to(dev).transpose(2, 1), torch.from_numpy(i8).to(dev)
with torch.no_grad():
    lg = dmodel(dp, dbx, None)["logits"]
    dmodel.ins_seg.dconv5.bias[1] -= (lg[:, :, 1] - lg[:, :, 0]).mean()
dref = dmodel._run(dp, dbx, init_box8=di8)
for prec in ("bf16", "fp16"):
    dmodel.precision = prec
    o = dmodel._run(dp, dbx, init_box8=di8, mask_override=dref["mask"])
    torch.cuda.synchronize()
    print(f"[dynamic {prec}] logits rel {rel(o['logits'], dref['logits']):.3e}  embedding rel {rel(o['embedding'], dref['embedding']):.3e}"
          f"  box_pred rel {rel(o['bp'], dref['bp']):.3e}  boxes7 abs {float((o['boxes7']-dref['boxes7']).abs().max()):.3e}")

# speed at the bench shape
if os.environ.get("SPEED", "1") == "1":
    B2 = 4096
    pts2 = torch.from_numpy(synth.static_crops(512, N)[0]).to(dev).repeat(8, 1, 1).contiguous().transpose(2, 1)
    init2 = init[:1].repeat(B2, 1).contiguous()
    one = sm.StaticModelOneBoxEst()
    one.load_state_dict({k: torch.as_tensor(v) for k, v in synth.state_dict("static_one").items()})
    one = one.to(dev).eval()
    for prec in ("fp32", "bf16", "fp16"):
        one.precision = prec
        for _ in range(2):
            one.refine(pts2, init2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            one.refine(pts2, init2)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(f"[speed {prec}] {dt*1e3:.2f} ms per 4096x1024 step -> {B2/dt:.0f} crops/s")
