#!/usr/bin/env python3
"""tools/latency_probe.py B [head] — N refine() calls at one small batch, for `rocprofv3 --kernel-trace --stats`."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
bench = importlib.import_module("bench")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
head = sys.argv[2] if len(sys.argv) > 2 else "static"
dev = torch.device("cuda", 0)
if head == "static":
    model, inputs, _ = bench.make_static(B, 1024, dev, 0)
else:
    model, inputs = bench.make_dynamic(B, dev, 0)
for _ in range(200):
    model.refine(*inputs)
torch.cuda.synchronize()
