import importlib, sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from _common import build_model, golden, synth
sm = importlib.import_module("3dal_pytorch_amd.static_model")
dmm = importlib.import_module("3dal_pytorch_amd.dynamic_model")
losses = importlib.import_module("3dal_pytorch_amd.losses")
kind = sys.argv[1] if len(sys.argv) > 1 else "dynamic"
g = golden("train_step_" + kind)
if kind == "dynamic":
    B, n_per = 4, 64; N = 5 * n_per
    pts, box, init8, gt = synth.dynamic_items(B, n_per_frame=n_per, seed=42)
    labels = synth.loss_case(42, batch=B, n_pts=N)[1]
else:
    B, N = 8, 256
    pts, init, gt = synth.static_crops(B, N, seed=41)
    labels = synth.loss_case(41, batch=B, n_pts=N)[1]
sd = synth.recentre_seg_bias(synth.state_dict(kind, seed=43), float(g["margin_shift"]))
keep = np.unpackbits(g["drop_keep"], axis=1).astype(np.float32)
_orig = sm._mask_and_gather
def _mg(p, lg, n_obj, n_ch, model=None):
    o, m = _orig(p, lg, n_obj, n_ch, model)
    return o.to(p.dtype), m
sm._mask_and_gather = _mg; dmm._mask_and_gather = _mg
grads = {}
_float = torch.Tensor.float
for tag, dtype in (("hip", torch.float32), ("f64", torch.float64)):
    torch.set_default_dtype(dtype)
    torch.Tensor.float = _float if dtype == torch.float32 else (lambda self, *a, **k: self.double())
    sm._MEAN_SIZE_ON.clear(); losses._CONST.clear()
    model = build_model(kind, sd).to(dtype).train()
    model.sampler = "numpy"
    if tag == "hip":
        model.drop_mask = torch.from_numpy(keep * 2.0).cuda()
    else:
        model.train_backend = "torch"
        keep_t = torch.from_numpy(keep).cuda().to(dtype).reshape(B, N, 128).permute(0, 2, 1)
        model.ins_seg.dropout.register_forward_hook(lambda m, i, o: i[0] * keep_t * 2.0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda().to(dtype)
    np.random.seed(int(g["np_seed"]))
    if kind == "dynamic":
        out = model(t(pts).transpose(2, 1), t(box).transpose(2, 1), t(gt))
        crit = losses.DynamicModelLoss()
    else:
        out = model(t(pts).transpose(2, 1), t(init), t(gt))
        crit = losses.FrustumPointNetLossOneBoxEst()
    ls = crit(out, *[t(a) if a.dtype == np.float32 else torch.from_numpy(a).cuda() for a in labels])
    ls["total_loss"].backward()
    grads[tag] = {n: q.grad.detach().double().cpu() for n, q in model.named_parameters() if q.grad is not None}
    print(tag, "loss", float(ls["total_loss"]), "ref", float(g["ref_loss_total_loss"]))
torch.set_default_dtype(torch.float32)
torch.Tensor.float = _float
for n in grads["f64"]:
    a, b = grads["hip"][n], grads["f64"][n]
    mx = float(b.abs().max())
    if mx < 1e-12: continue
    e = (a - b).abs()
    line = f"{n:28s} {float(e.max()) / mx:.2e}"
    if e.dim() >= 2 and float(e.max()) / mx > 2e-4:
        per_row = e.reshape(e.shape[0], -1).max(1).values / mx
        top = per_row.topk(min(4, per_row.numel()))
        line += f"   rows with the largest error: {top.indices.tolist()} {['%.1e' % v for v in top.values.tolist()]}  median row {float(per_row.median()):.1e}"
    print(line)
