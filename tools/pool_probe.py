#!/usr/bin/env python3
"""dal3_tr_linear_pool at conv5's shape (128 -> 1024, 64 x 4096 points), a few launches: the target of tools/pmc_pool.sh"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hip = importlib.import_module("3dal_pytorch_amd._hip")
lib, dev = hip.lib(), torch.device("cuda", 0)
M, ci, co, seg = 64 * 4096, 128, 1024, 4096
a = torch.randn((M, ci), device=dev)
W = torch.randn((co, ci), device=dev) * 0.1
b = torch.randn(co, device=dev)
sc, sh = torch.rand(ci, device=dev) + 0.5, torch.randn(ci, device=dev) * 0.1
osc, osh = torch.rand(co, device=dev) + 0.5, torch.randn(co, device=dev) * 0.1
n_seg = M // seg
g = torch.empty((n_seg, co), device=dev)
arg = torch.empty((n_seg, co), dtype=torch.int32, device=dev)
need = lib.dal3_tr_linear_pool_workspace_bytes(ci, co, n_seg)
ws = torch.empty(need, dtype=torch.uint8, device=dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    hip.check(lib.dal3_tr_linear_pool(hip.ptr(a), M, ci, ci, hip.ptr(sc), hip.ptr(sh), 1, hip.ptr(W), ci, hip.ptr(b), hip.ptr(osc),
                                      hip.ptr(osh), seg, co, hip.ptr(g), hip.ptr(arg), hip.ptr(ws), need, hip.stream()))
torch.cuda.synchronize()
