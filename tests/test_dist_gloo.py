"""world_size-2 gloo run of the sharding + all-gather path on CPU. The HIP forward cannot run
here, so each rank's `refine` is stood in by the oracle (test infrastructure); what is under
test is shard_range / all_gather_boxes / refine_sharded: ragged tails, empty shards, ordering."""
import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from _common import synth
from oracle import ref_heads as R

dal3_dist = importlib.import_module("3dal_pytorch_amd.dist")


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 8, 9, 4096, 4100):
        for w in (1, 2, 3, 4, 8):
            spans = [dal3_dist.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) == (n + w - 1) // w if n else True


class _OracleModel(torch.nn.Module):
    """stand-in with the product's refine() signature, computing with the oracle on CPU"""

    def __init__(self, sd):
        super().__init__()
        self.sd = R.as_torch_sd(sd)
        self.p = torch.nn.Parameter(torch.zeros(1))
        self.item_offset = 0

    def refine(self, pts, init_box, bbox_gt=None):
        np.random.seed(1000 + self.item_offset)
        out = R.static_one_forward(self.sd, pts, init_box)
        return torch.from_numpy(R.decode_static(out, init_box, False)).float()


def _worker(rank, world, port, n_items, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        model = _OracleModel(synth.state_dict("static_one"))

        def make_shard(lo, hi):
            pts, init, gt = synth.static_crops(hi - lo, 128, first=lo)
            return torch.from_numpy(pts).transpose(2, 1), torch.from_numpy(init), torch.from_numpy(gt)

        out = dal3_dist.refine_sharded(model, n_items, make_shard)
        assert out.shape == (n_items, 7)
        if rank == 0:
            ret.put(out.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_items", [(2, 5), (2, 4), (3, 2)])
def test_sharded_refine_equals_per_shard_concat(world, n_items):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, ret)) for r in range(world)]
    for p in procs:
        p.start()
    got = ret.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # expected: the same shards computed in this process, concatenated in rank order
    model = _OracleModel(synth.state_dict("static_one"))
    want = []
    for r in range(world):
        lo, hi = dal3_dist.shard_range(n_items, r, world)
        if hi > lo:
            pts, init, _ = synth.static_crops(hi - lo, 128, first=lo)
            model.item_offset = lo
            want.append(model.refine(torch.from_numpy(pts).transpose(2, 1), torch.from_numpy(init)).numpy())
    assert np.array_equal(got, np.concatenate(want))
    # and synth shards are slices of the whole job (global item keying)
    whole = synth.static_crops(n_items, 128)[0]
    lo, hi = dal3_dist.shard_range(n_items, world - 1, world)
    if hi > lo:
        assert np.array_equal(whole[lo:hi], synth.static_crops(hi - lo, 128, first=lo)[0])
