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


def _rehearsal(extra_args, timeout=900):
    """`bench.py --gpus 2` on the ONE GPU of the test box: both ranks on device 0 (DAL3_BENCH_SHARE_GPU=1), the boxes
    gathered over gloo through pinned host memory (DAL3_BENCH_BACKEND=gloo; RCCL cannot connect two ranks that share a
    device). Everything else is the real N > 1 run: the self-launch from a parent that never touches the GPU, device
    selection, sharding by global item index, the real kernels, fences, the overlapped gather, rank 0's stdout relay."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DAL3_BENCH_SHARE_GPU="1", DAL3_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + extra_args, capture_output=True,
                         text=True, env=env, timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0].encode()) <= 4096, out.stdout[-2000:]
    return json.loads(lines[0])


def test_two_rank_rehearsal_on_one_gpu_static_line_proves_its_gather():
    """C2's head at a reduced batch: the gathered (2B,7) boxes hold rank 1's rows, and rank 0 — recomputing rank 1's
    first 64 crops from their global indices with rank 1's sampler key — gets the same bits."""
    rec = _rehearsal(["--steps", "3", "--warmup", "1", "--batch", "512"])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["value"] > 0
    assert rec["rccl"]["backend"] == "gloo" and rec["rccl"]["world_size"] == 2 and rec["rccl"]["ranks_counted"] == 2
    assert rec["rccl"]["ranks_share_gpus"] is True
    assert len(rec["ms_per_step_per_rank"]) == 2 and min(rec["ms_per_step_per_rank"]) > 0
    assert rec["gather_equals_single_rank"] is True, rec["gather_self_check"]
    (c,) = rec["gather_self_check"]["checks"]
    assert (c["head"], c["peer"], c["first_item"], c["rows"]) == ("static", 1, 512, 64)
    # rank 0's roofline is part of the (<= 4 KB) line at every world size
    assert rec["roofline"]["kernel"].startswith("ins_seg_decode") and 0 < rec["roofline"]["frac"] < 1
    assert rec["config"]["items_per_gpu"] == 512
    # round 6: next to the weak-scaling value, the same 512-crop batch SPLIT over the two ranks (256 each, real kernels),
    # gathered, and rank 0's recomputation of rank 1's first rows bit-equal
    st = rec["strong"]
    assert (st["scaling"], st["items"], st["items_per_rank"]) == ("strong", 512, [256, 256])
    assert st["value"] > 0 and st["ms_per_step"] > 0 and st["gather_equals_single_rank"] is True and st["vs_n1"] is None
    assert len(st["ms_per_step_per_rank"]) == 2
    tr = rec["rccl"]["transport"]
    assert tr["backend"] == "gloo" and tr["peer_access"] == ["1", "1"]          # one device, each rank sees itself


def test_one_rank_rccl_bench_line_carries_the_transport_report():
    """`bench.py --gpus 1` with the RCCL path forced (DAL3_FORCE_DIST=1): the line's `rccl.transport` is filled from
    RCCL's own INIT log (one file per rank under gpurun_out/rccl_debug_<port>/) — with one rank there is no peer
    connection to report, but the communicator's size, the parsed file and the peer-access row are there; on a
    multi-GPU node the same code reports `via` (P2P/IPC = xGMI) for every channel."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR")}
    env.update(DAL3_FORCE_DIST="1", MASTER_PORT="29547", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                          "--batch", "256", "--no-extras"], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0].encode()) <= 4096
    rec = json.loads(lines[0])
    tr = rec["rccl"]["transport"]
    assert rec["rccl"]["backend"] == "nccl" and tr["backend"] == "nccl" and tr["peer_access"] == ["1"]
    assert tr["files"] >= 1 and tr["lines"] > 0 and 1 in tr["nranks"], tr
    assert tr["via"] == {} and tr["p2p_only"] is False                          # (no peer with one rank)
    assert rec["strong"]["items_per_rank"] == [256] and rec["strong"]["value"] > 0


def test_two_rank_rehearsal_on_one_gpu_mixed_segment_checks_both_heads():
    """C4 (one segment, static + dynamic heads, strong scaling): one all-gather per head, both self-checked"""
    rec = _rehearsal(["--config", "C4", "--steps", "1", "--warmup", "1", "--no-extras"], timeout=1500)
    assert rec["scaling"] == "strong" and rec["gather_equals_single_rank"] is True, rec["gather_self_check"]
    heads = {(c["head"], c["peer"]): c for c in rec["gather_self_check"]["checks"]}
    assert set(heads) == {("static", 1), ("dynamic", 1)}
    assert heads[("static", 1)]["first_item"] == 32 and heads[("static", 1)]["rows"] == 32
    assert heads[("dynamic", 1)]["rows"] == 64


def test_two_rank_rehearsal_on_one_gpu_f16x3():
    """the same rehearsal on the f16x3 kernels: shard == whole is a property of the arithmetic too (per-point products,
    exact max): rank 0's recomputation of rank 1's rows gives the gathered bits"""
    rec = _rehearsal(["--steps", "3", "--warmup", "1", "--batch", "512", "--precision", "f16x3"])
    assert rec["n_gpus"] == 2 and rec["dtype"].startswith("f16x3") and rec["value"] > 0
    assert rec["gather_equals_single_rank"] is True, rec["gather_self_check"]
    assert rec["roofline"]["kernel"] == "ins_seg_decode_x3_kernel" and rec["roofline"]["peak"] == 2500.0
