"""Object-sharded multi-GPU refinement: one process per GPU, crops split in contiguous index
ranges, weights replicated, and ONE all-gather of the refined (B,7) boxes per batch over
RCCL/xGMI (torch.distributed backend "nccl" on ROCm). Replaces the reference's pickle-based
all_gather of per-rank results (det3d/torchie/trainer/utils.py:114-154); the heads themselves are
single-GPU in the reference.

No data-path collective exists besides that gather: every crop is independent in eval mode.

Two transports. RCCL ("nccl"): the collective runs on the device buffers. gloo (a rehearsal of the multi-rank
path where RCCL cannot run, e.g. two ranks sharing the one GPU of a box): gloo has no device all-gather, so the
boxes are staged through pinned host memory — same sharding, same buffers, same overlap, real kernels.
"""
import os

import torch
import torch.distributed as dist


def shard_range(n_items, rank, world_size):
    """Contiguous [lo, hi) of rank `rank`: ceil(n/world) items per rank, ragged (possibly empty) tail."""
    per = (n_items + world_size - 1) // world_size
    lo = min(rank * per, n_items)
    return lo, min(lo + per, n_items)


def _active(group=None):
    return dist.is_available() and dist.is_initialized() and (
        dist.get_world_size(group) > 1 or os.environ.get("DAL3_FORCE_DIST") == "1")


def host_staged(device, group=None):
    """True when collectives on tensors of `device` must go through host memory (gloo + a GPU tensor)"""
    return torch.device(device).type == "cuda" and dist.get_backend(group) == "gloo"


def all_gather_boxes(local_boxes, n_items, group=None):
    """local_boxes (hi-lo, 7) of this rank's shard -> (n_items, 7) on every rank. The ragged tail
    is padded to ceil(n/world) rows so that one fixed-size all_gather_into_tensor moves it
    (a 14 KiB-per-rank message at B=4096, W=8: latency-bound, so one collective, not W)."""
    if not _active(group):
        return local_boxes                       # (DAL3_FORCE_DIST=1: run the collective even on one rank)
    world = dist.get_world_size(group)
    per = (n_items + world - 1) // world
    width = local_boxes.shape[1]
    if local_boxes.shape[0] == per:
        send = local_boxes.contiguous()
    else:
        send = local_boxes.new_zeros((per, width))
        send[: local_boxes.shape[0]] = local_boxes
    if host_staged(local_boxes.device, group):
        send_h = send.cpu()                      # synchronises with the stream that produced the boxes
        out_h = send_h.new_empty((world * per, width))
        dist.all_gather_into_tensor(out_h, send_h, group=group)
        return out_h[:n_items].to(local_boxes.device)
    out = local_boxes.new_empty((world * per, width))
    dist.all_gather_into_tensor(out, send, group=group)
    return out[:n_items]


class BoxGatherer:
    """The same all-gather, taken off the critical path: persistent send / receive buffers in two slots and an
    asynchronous all_gather_into_tensor per batch whose result is collected one batch later, so that the collective
    (tens of microseconds of RCCL latency, more when a peer is late) runs beside the next batch's kernels instead
    of between two batches. A 16-bit step of a few milliseconds cannot afford a gather in series with it.

        g = BoxGatherer(n_items, device)
        for batch in batches:
            g.submit(model.refine(*batch))        # enqueue only
            done = g.collect(keep=1)              # (n_items,7) boxes of the PREVIOUS batch, or None
        last = g.collect(keep=0)

    LIFETIME of what collect() returns: by default a fresh tensor (a 28-byte-per-box device copy) that the caller
    may keep. With copy=False it is a VIEW of this object's receive buffer of that slot, overwritten by the
    all-gather submitted `slots` batches later (with the loop above and slots = 2: valid until the second submit()
    after the collect) — for callers that consume the boxes at once and want no extra launch.

    Without an initialised process group it degenerates to passing the local boxes through (no copy either way:
    the tensor is the caller's own)."""

    def __init__(self, n_items, device, width=7, group=None, slots=2):
        self.n_items, self.width, self.group = n_items, width, group
        self.active = _active(group)
        self.world = dist.get_world_size(group) if self.active else 1
        self.per = (n_items + self.world - 1) // self.world
        self.slots = slots
        self.staged = self.active and host_staged(device, group)
        mk = dict(dtype=torch.float32, device=device)
        self.send = [torch.zeros((self.per, width), **mk) for _ in range(slots)] if self.active else None
        self.recv = [torch.empty((self.world * self.per, width), **mk) for _ in range(slots)] if self.active else None
        if self.staged:
            self.send_h = [torch.zeros((self.per, width), dtype=torch.float32).pin_memory() for _ in range(slots)]
            self.recv_h = [torch.empty((self.world * self.per, width), dtype=torch.float32).pin_memory() for _ in range(slots)]
        self.pending = []                        # [(slot | local boxes, work | event)] oldest first
        self.turn = 0

    def submit(self, local_boxes):
        if len(self.pending) == self.slots:
            raise RuntimeError("BoxGatherer: collect() before submitting more batches than there are slots")
        if not self.active:                      # one rank: the boxes themselves, no staging copy
            self.pending.append((local_boxes, None))
            return
        slot = self.turn
        self.turn = (self.turn + 1) % self.slots
        if self.staged:                          # device -> pinned host, asynchronous; the collective waits for it in collect()
            self.send_h[slot][: local_boxes.shape[0]].copy_(local_boxes, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self.pending.append((slot, ev))
            return
        self.send[slot][: local_boxes.shape[0]].copy_(local_boxes)
        work = dist.all_gather_into_tensor(self.recv[slot], self.send[slot], group=self.group, async_op=True)
        self.pending.append((slot, work))

    def collect(self, keep=0, copy=True):
        """Result of the oldest outstanding batch once more than `keep` are in flight, else None."""
        if len(self.pending) <= keep:
            return None
        slot, work = self.pending.pop(0)
        if work is None:
            return slot[: self.n_items]
        if self.staged:
            work.synchronize()                   # that batch's boxes are in host memory (later batches keep the GPU busy)
            dist.all_gather_into_tensor(self.recv_h[slot], self.send_h[slot], group=self.group)
            self.recv[slot].copy_(self.recv_h[slot], non_blocking=True)
        else:
            work.wait()                          # orders the current stream behind the collective; no host sync
        out = self.recv[slot][: self.n_items]
        return out.clone() if copy else out


def world_census(device, group=None):
    """What the communicator itself reports: backend, world size, and the sum of one 1 per rank through an
    all-reduce (= the number of ranks that really took part)."""
    if not (dist.is_available() and dist.is_initialized()):
        return {"backend": None, "world_size": 1, "ranks_counted": 1}
    one = torch.ones(1, dtype=torch.float32, device="cpu" if host_staged(device, group) else device)
    dist.all_reduce(one, group=group)
    return {"backend": dist.get_backend(group), "world_size": dist.get_world_size(group),
            "ranks_counted": int(round(float(one.item())))}


RCCL_VIA = None          # compiled lazily (module import stays cheap)


def parse_rccl_debug(texts):
    """`texts`: the contents of the ranks' RCCL debug files. Returns what they say about the communicator:
    `via` = how many channel connections each transport carries ("P2P/IPC", "P2P/direct pointer", "SHM/direct/direct",
    "NET/Socket/0" ...: the words RCCL prints after "via" in its `Channel 00/0 : 0[0] -> 1[1] via P2P/IPC` lines),
    `pairs` = the distinct (from, to) rank pairs among them, `nranks` = the communicator sizes its "Init COMPLETE" /
    "nranks N" lines report, `xgmi_lines` = lines that mention XGMI (RCCL's topology dump names the link type), and
    (the files are asked for by launch.rccl_debug_to() before torch is imported.)
    `p2p_only` = every connection is a P2P one (on an MI355X node: xGMI; what SURVEY.md 8(e) asks of the one
    all-gather). A pure function of the text: tests/test_dist_gloo.py feeds it a recorded sample."""
    import re
    global RCCL_VIA
    if RCCL_VIA is None:
        RCCL_VIA = (re.compile(r"(\d+)\[[0-9a-fA-F]+\]\s*->\s*(\d+)\[[0-9a-fA-F]+\].*?\bvia\s+(\S+(?: pointer)?)"),
                    re.compile(r"\bnranks\s+(\d+)"))
    via, pairs, nranks, xgmi, lines = {}, set(), set(), 0, 0
    for text in texts:
        for ln in text.splitlines():
            lines += 1
            if "XGMI" in ln.upper():
                xgmi += 1
            m = RCCL_VIA[0].search(ln)
            if m:
                via[m.group(3)] = via.get(m.group(3), 0) + 1
                pairs.add((int(m.group(1)), int(m.group(2))))
            m = RCCL_VIA[1].search(ln)
            if m:
                nranks.add(int(m.group(1)))
    return {"files": len(texts), "lines": lines, "via": dict(sorted(via.items())), "pairs": len(pairs),
            "nranks": sorted(nranks), "xgmi_lines": xgmi,
            "p2p_only": bool(via) and all(k.startswith("P2P") for k in via)}


def peer_access_row(device):
    """this rank's view of HIP peer access: one 0/1 per visible device (1 on the diagonal). Host-side queries only."""
    dev = torch.device(device)
    if dev.type != "cuda":
        return []
    n = torch.cuda.device_count()
    me = dev.index if dev.index is not None else torch.cuda.current_device()
    return [1 if p == me or torch.cuda.can_device_access_peer(me, p) else 0 for p in range(n)]


def gather_rows(row, width, device, group=None):
    """one fixed-width row of small ints per rank -> every rank's rows (world x width), on every rank"""
    row = (list(row) + [-1] * width)[:width]
    if not (dist.is_available() and dist.is_initialized()):
        return [row]
    world = dist.get_world_size(group)
    dev = "cpu" if host_staged(device, group) else device
    mine = torch.tensor(row, dtype=torch.int32, device=dev)
    out = torch.empty(world * width, dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(out, mine, group=group)
    return [[int(v) for v in r] for r in out.cpu().view(world, width)]


def replicate_(tensor, src=0, group=None):
    """Weights are REPLICATED: overwrite `tensor` on every rank with rank `src`'s values (a start-up broadcast, not
    part of the data path). No-op without a process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return tensor
    with torch.no_grad():
        if host_staged(tensor.device, group):
            t = tensor.detach().cpu()
            dist.broadcast(t, src, group=group)
            tensor.copy_(t)
        else:
            dist.broadcast(tensor.detach(), src, group=group)
    return tensor


def gather_scalars(value, device, group=None):
    """one float per rank -> the list of all ranks' values, on every rank (per-rank step times of a bench line)"""
    if not (dist.is_available() and dist.is_initialized()):
        return [float(value)]
    world = dist.get_world_size(group)
    dev = "cpu" if host_staged(device, group) else device
    mine = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    out = torch.empty(world, dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(out, mine, group=group)
    return [float(v) for v in out.cpu()]


def refine_sharded(model, n_items, make_shard, group=None):
    """Run `model.refine` on this rank's shard and gather. make_shard(lo, hi) returns the refine()
    arguments for items [lo, hi) already resident on this rank's GPU. The device sampler is keyed
    on the global item index, so the gathered result equals the single-GPU result bit for bit."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    lo, hi = shard_range(n_items, rank, world)
    if hi > lo:
        saved = model.item_offset
        model.item_offset = lo
        try:
            local = model.refine(*make_shard(lo, hi))
        finally:
            model.item_offset = saved          # the model leaves as it came (a later eager call keys on ITS offset)
    else:
        dev = next(model.parameters()).device
        local = torch.zeros((0, 7), dtype=torch.float32, device=dev)
    return all_gather_boxes(local, n_items, group)
