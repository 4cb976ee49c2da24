#!/usr/bin/env python3
"""Which lines of the package launch the stock-torch kernels of one training step, by device time:
  python tools/prof_train_ops.py [--kind static_one]     (torch.profiler with stacks; innermost frame inside this repo)"""
import argparse
import collections
import importlib
import os
import sys

import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
synth = importlib.import_module("3dal_pytorch_amd.synth")
sm = importlib.import_module("3dal_pytorch_amd.static_model")
losses = importlib.import_module("3dal_pytorch_amd.losses")
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--points", type=int, default=4096)
args = ap.parse_args()
B, N = args.batch, args.points
dev = torch.device("cuda", 0)
p, i, g = synth.static_crops(B, N, seed=3)
pts, init, gt = torch.from_numpy(p).to(dev).transpose(2, 1), torch.from_numpy(i).to(dev), torch.from_numpy(g).to(dev)
labels = ((torch.rand((B, N), device=dev) > 0.6).float(), torch.randn((B, 3), device=dev), torch.randint(0, 12, (B,), device=dev),
          0.1 * torch.randn((B,), device=dev), torch.randint(0, 3, (B,), device=dev), 0.3 * torch.randn((B, 3), device=dev))
model = sm.StaticModelOneBoxEst()
model.load_state_dict({k: torch.as_tensor(v) for k, v in synth.state_dict("static_one").items()})
model = model.to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
crit = losses.FrustumPointNetLossOneBoxEst()


def step():
    loss = crit(model(pts, init, gt), *labels)["total_loss"]
    opt.zero_grad()
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
by_site = collections.defaultdict(lambda: [0.0, 0])
for ev in prof.events():
    t = getattr(ev, "device_time_total", 0) or getattr(ev, "cuda_time_total", 0)
    if not t or not ev.name.startswith("aten::") or ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
        continue
    site = "?"
    for fr in ev.stack or []:
        if ROOT in fr and "tools/prof_train_ops" not in fr:
            site = fr.replace(ROOT + "/", "")
            break
    k = (ev.name, site)
    by_site[k][0] += t
    by_site[k][1] += 1
tot = sum(v[0] for v in by_site.values())
print(f"stock aten ops: {tot / 1e3:.2f} ms device time in {sum(v[1] for v in by_site.values())} top-level calls")
for (name, site), (t, n) in sorted(by_site.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f"{t:9.1f} us {n:4d} x  {name:28s} {site}")
