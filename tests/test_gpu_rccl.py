"""The RCCL leg on the one GPU the test box has: a one-rank "nccl" process group (= RCCL on ROCm) with the collective
forced (DAL3_FORCE_DIST=1), in a child process so that the group does not leak into the other tests. What N > 1
adds on top — ragged tails, ordering — is covered by tests/test_dist_gloo.py on CPU."""
import os
import subprocess
import sys

import pytest

from _common import ROOT

pytestmark = pytest.mark.gpu

CHILD = r"""
import importlib, os, sys
import numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
from _common import build_model, synth
dal3_dist = importlib.import_module("3dal_pytorch_amd.dist")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
model = build_model("static_one", synth.state_dict("static_one", seed=15))
p, i, g = (torch.from_numpy(a).cuda() for a in synth.static_crops(37, 512, seed=15))
want = model.refine(p.transpose(2, 1), i, g).clone()
got = dal3_dist.refine_sharded(model, 37, lambda lo, hi: (p[lo:hi].transpose(2, 1), i[lo:hi], g[lo:hi]))
torch.cuda.synchronize()
assert got.shape == (37, 7) and torch.equal(got, want)
dist.barrier()
dist.destroy_process_group()
print("rccl ok")
"""


def test_one_rank_rccl_all_gather_returns_the_refined_boxes():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", DAL3_FORCE_DIST="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0 and "rccl ok" in out.stdout, out.stderr[-2000:]
