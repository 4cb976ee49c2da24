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


def _gatherer_worker(rank, world, port, n_items, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = dal3_dist.shard_range(n_items, rank, world)
        g = dal3_dist.BoxGatherer(n_items, torch.device("cpu"))
        got = []
        for step in range(5):                                   # submit step k, collect step k-1: the overlapped form
            local = torch.arange(lo, hi, dtype=torch.float32)[:, None] * 100 + step + torch.arange(7)[None] * 0.125
            g.submit(local)
            r = g.collect(keep=1)
            if r is not None:
                got.append(r)                                   # kept WITHOUT a clone: collect() returns the caller's own copy
        got.append(g.collect(keep=0))
        assert g.collect(keep=0) is None
        # copy=False hands out a VIEW of the slot's receive buffer: the gather two submits later overwrites it (the
        # lifetime the docstring states); the default copies survive (checked by the caller against every step)
        g.submit(local)
        view = g.collect(keep=0, copy=False)
        before = view.clone()
        g.submit(local + 1)
        g.collect(keep=0)
        g.submit(local + 2)
        g.collect(keep=0)
        assert not torch.equal(view, before) and view.data_ptr() in [r.data_ptr() for r in g.recv]
        # replicated weights / per-rank scalars (what a bench line at N > 1 is built from)
        w = torch.full((3,), float(rank + 1))
        dal3_dist.replicate_(w)
        assert torch.equal(w, torch.ones(3))
        assert dal3_dist.gather_scalars(10.0 + rank, torch.device("cpu")) == [10.0 + r for r in range(world)]
        with pytest.raises(RuntimeError):
            for _ in range(3):
                g.submit(local)
        census = dal3_dist.world_census(torch.device("cpu"))
        if rank == world - 1:
            ret.put((torch.stack(got).numpy(), census))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_items", [(2, 9), (3, 4), (8, 61)])
def test_box_gatherer_pipelines_steps_in_order(world, n_items):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_gatherer_worker, args=(r, world, port, n_items, ret)) for r in range(world)]
    for p in procs:
        p.start()
    got, census = ret.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert census == {"backend": "gloo", "world_size": world, "ranks_counted": world}
    assert got.shape == (5, n_items, 7)
    for step in range(5):
        want = np.arange(n_items, dtype=np.float32)[:, None] * 100 + step + np.arange(7, dtype=np.float32)[None] * 0.125
        assert np.array_equal(got[step], want), step


def test_box_gatherer_without_a_process_group_passes_boxes_through():
    g = dal3_dist.BoxGatherer(5, torch.device("cpu"))
    x = torch.arange(35, dtype=torch.float32).reshape(5, 7)
    g.submit(x)
    assert g.collect(keep=1) is None
    assert torch.equal(g.collect(keep=0), x)
    assert dal3_dist.world_census(torch.device("cpu")) == {"backend": None, "world_size": 1, "ranks_counted": 1}


# ------------------------------------------------------------------ the file-level drivers (eval.py) on two ranks
def _eval_worker(rank, world, port, root, head, ret):
    """refine_static_tracks / refine_dynamic_tracks with the device stages stood in on CPU: crop preparation by a
    deterministic function of the GLOBAL item index (what the device sampler guarantees), the heads by a cheap
    function of the prepared crop. Under test: contiguous sharding of tracks / track-frames, per-batch item offsets,
    the ragged all-gather, and that every rank ends up with the whole result."""
    import pickle
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ev = importlib.import_module("3dal_pytorch_amd.eval")
        infos = ev.reorganize_info(pickle.load(open(os.path.join(root, "infos.pkl"), "rb")))
        annos = ev.Annos(infos)
        track = pickle.load(open(os.path.join(root, "trackStatic.pkl" if head == "static" else "trackDynamic.pkl"), "rb"))

        def fake_static(tracks, poses, n_points, sampler, seed, item_offset, device):
            assert sampler == "device"
            pts = torch.stack([torch.full((3, 8), float(item_offset + b) + 0.25 * len(t["token"])) for b, t in enumerate(tracks)])
            init = torch.stack([torch.as_tensor(np.asarray(t["bbox"][0], np.float32)) for t in tracks])
            return pts, init

        class _Store:
            def __init__(self, tracks, dev):
                self.tracks = tracks

        def fake_dynamic(store, items, poses, n_per_frame, r, s, sampler, seed, item_offset, device):
            assert sampler == "device"
            B = len(items)
            pts = torch.stack([torch.full((4, 8), float(item_offset + b) + 0.5 * t + 0.01 * i) for b, (t, i) in enumerate(items)])
            init = torch.stack([torch.as_tensor(np.append(np.asarray(store.tracks[t]["bbox"][i], np.float32), 0.0)) for t, i in items])
            return pts, torch.zeros((B, 8, 101)), init

        class _Model(torch.nn.Module):
            r, s = 2, 50

            def __init__(self):
                super().__init__()
                self.p = torch.nn.Parameter(torch.zeros(1))
                self.item_offset = 0

            def refine(self, pts, *rest):
                init = rest[-1]
                return torch.cat([init[:, :6], pts[:, 0, :1] + self.item_offset * 0.0], 1).float()

        ev.prep.prepare_static_batch = fake_static
        ev.prep.prepare_dynamic_batch = fake_dynamic
        ev.prep.TrackStore = _Store
        model = _Model()
        if head == "static":
            track = ev.preprocessing(track, annos)
            out = ev.refine_static_tracks(model, track, annos, batch_size=2, n_points=8, sampler="device")
        else:
            out = ev.refine_dynamic_tracks(model, track, annos, batch_size=5, sampler="device")
        if world > 1:
            with pytest.raises(ValueError, match="cannot be sharded"):
                ev.refine_static_tracks(model, track, annos, sampler="numpy")
        if rank == world - 1:                              # the LAST rank reports: it holds the ragged tail
            ret.put(out)
    finally:
        if world > 1:
            dist.destroy_process_group()


@pytest.mark.parametrize("head", ["static", "dynamic"])
def test_eval_drivers_shard_and_gather(tmp_path, head):
    synth.segment_files(str(tmp_path), 77, n_frames=12, n_tracks=7)
    ctx = mp.get_context("spawn")
    results = {}
    for world in (1, 2, 3):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        ret = ctx.Queue()
        procs = [ctx.Process(target=_eval_worker, args=(r, world, port, str(tmp_path), head, ret)) for r in range(world)]
        for p in procs:
            p.start()
        results[world] = ret.get(timeout=240)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    n = results[1].shape[0]
    assert n == (5 if head == "static" else 70) and results[1].shape == (n, 7)      # 7 tracks, 2 dropped / 70 track-frames
    assert len(np.unique(results[1][:, 6])) == n                   # the stand-in encodes the global item index
    for world in (2, 3):
        assert np.array_equal(results[world], results[1]), world


def test_rccl_debug_log_parser_counts_transports_pairs_and_ranks():
    """dist.parse_rccl_debug on recorded RCCL INFO lines (the format of `Channel .. : a[dev] -> b[dev] via <transport>`
    and `... nranks N ... Init COMPLETE`): what bench.py's `rccl.transport` is built from on a multi-GPU node"""
    dal3_dist = importlib.import_module("3dal_pytorch_amd.dist")
    xgmi = "\n".join([
        "node:101:201 [0] NCCL INFO Channel 00/0 : 0[0] -> 1[1] via P2P/IPC",
        "node:101:201 [0] NCCL INFO Channel 01/0 : 0[0] -> 1[1] via P2P/IPC",
        "node:101:201 [0] NCCL INFO Channel 00/0 : 0[c1000] -> 7[e5000] via P2P/direct pointer",
        "node:101:201 [0] NCCL INFO comm 0x55 rank 0 nranks 8 cudaDev 0 busId c1000 commId 0x1 - Init COMPLETE",
        "node:101:201 [0] NCCL INFO  GPU/C1000 (0) --XGMI--> GPU/E5000",
    ])
    got = dal3_dist.parse_rccl_debug([xgmi, "node:102:202 [1] NCCL INFO Channel 00/0 : 1[1] -> 2[2] via P2P/IPC\n"])
    assert got["files"] == 2 and got["via"] == {"P2P/IPC": 3, "P2P/direct pointer": 1} and got["pairs"] == 3
    assert got["nranks"] == [8] and got["xgmi_lines"] == 1 and got["p2p_only"] is True
    shm = dal3_dist.parse_rccl_debug(["n:1:1 [0] NCCL INFO Channel 00 : 0[0] -> 1[1] via SHM/direct/direct\n"])
    assert shm["via"] == {"SHM/direct/direct": 1} and shm["p2p_only"] is False
    assert dal3_dist.parse_rccl_debug([]) == {"files": 0, "lines": 0, "via": {}, "pairs": 0, "nranks": [], "xgmi_lines": 0,
                                              "p2p_only": False}
    assert dal3_dist.peer_access_row("cpu") == [] and dal3_dist.gather_rows([1, 0], 4, "cpu") == [[1, 0, -1, -1]]
