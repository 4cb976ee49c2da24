#!/usr/bin/env python3
"""How the oracle's CPU throughput depends on the torch thread count on this host (picks the
thread count bench.py's cpu_baseline uses)."""
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
synth = importlib.import_module("3dal_pytorch_amd.synth")
R = importlib.import_module("oracle.ref_heads")

pts_np, init_np, _ = synth.static_crops(64, 1024)
tsd = R.as_torch_sd(synth.state_dict("static_one"))
pts, init = torch.from_numpy(pts_np).transpose(2, 1), torch.from_numpy(init_np)
print("affinity", len(os.sched_getaffinity(0)))
for n in [int(a) for a in sys.argv[1:]] or [8, 16, 32, 64, 128]:
    torch.set_num_threads(n)
    with torch.no_grad():
        R.static_one_forward(tsd, pts, init)
        t0 = time.perf_counter()
        for _ in range(2):
            R.static_one_forward(tsd, pts, init)
        dt = (time.perf_counter() - t0) / 2
    print(f"threads {n}: {64 / dt:.1f} crops/s", flush=True)
