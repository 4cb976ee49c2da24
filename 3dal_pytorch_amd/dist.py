"""Object-sharded multi-GPU refinement: one process per GPU, crops split in contiguous index
ranges, weights replicated, and ONE all-gather of the refined (B,7) boxes per batch over
RCCL/xGMI (torch.distributed backend "nccl" on ROCm). Replaces the reference's pickle-based
all_gather of per-rank results (det3d/torchie/trainer/utils.py:114-154); the heads themselves are
single-GPU in the reference.

No data-path collective exists besides that gather: every crop is independent in eval mode.
"""
import os

import torch
import torch.distributed as dist


def shard_range(n_items, rank, world_size):
    """Contiguous [lo, hi) of rank `rank`: ceil(n/world) items per rank, ragged (possibly empty) tail."""
    per = (n_items + world_size - 1) // world_size
    lo = min(rank * per, n_items)
    return lo, min(lo + per, n_items)


def all_gather_boxes(local_boxes, n_items, group=None):
    """local_boxes (hi-lo, 7) of this rank's shard -> (n_items, 7) on every rank. The ragged tail
    is padded to ceil(n/world) rows so that one fixed-size all_gather_into_tensor moves it
    (a 14 KiB-per-rank message at B=4096, W=8: latency-bound, so one collective, not W)."""
    if not dist.is_available() or not dist.is_initialized():
        return local_boxes
    if dist.get_world_size(group) == 1 and os.environ.get("DAL3_FORCE_DIST") != "1":
        return local_boxes                       # (DAL3_FORCE_DIST=1: run the collective even on one rank)
    world = dist.get_world_size(group)
    per = (n_items + world - 1) // world
    width = local_boxes.shape[1]
    if local_boxes.shape[0] == per:
        send = local_boxes.contiguous()
    else:
        send = local_boxes.new_zeros((per, width))
        send[: local_boxes.shape[0]] = local_boxes
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

    Without an initialised process group it degenerates to passing the local boxes through."""

    def __init__(self, n_items, device, width=7, group=None, slots=2):
        self.n_items, self.width, self.group = n_items, width, group
        self.active = dist.is_available() and dist.is_initialized() and (
            dist.get_world_size(group) > 1 or os.environ.get("DAL3_FORCE_DIST") == "1")
        self.world = dist.get_world_size(group) if self.active else 1
        self.per = (n_items + self.world - 1) // self.world
        self.slots = slots
        self.send = [torch.zeros((self.per, width), dtype=torch.float32, device=device) for _ in range(slots)] \
            if self.active else None
        self.recv = [torch.empty((self.world * self.per, width), dtype=torch.float32, device=device)
                     for _ in range(slots)] if self.active else None
        self.pending = []                        # [(slot | local boxes, work)] oldest first
        self.turn = 0

    def submit(self, local_boxes):
        if len(self.pending) == self.slots:
            raise RuntimeError("BoxGatherer: collect() before submitting more batches than there are slots")
        if not self.active:                      # one rank: the boxes themselves, no staging copy
            self.pending.append((local_boxes, None))
            return
        slot = self.turn
        self.turn = (self.turn + 1) % self.slots
        self.send[slot][: local_boxes.shape[0]].copy_(local_boxes)
        work = dist.all_gather_into_tensor(self.recv[slot], self.send[slot], group=self.group, async_op=True)
        self.pending.append((slot, work))

    def collect(self, keep=0):
        """Result of the oldest outstanding batch once more than `keep` are in flight, else None."""
        if len(self.pending) <= keep:
            return None
        slot, work = self.pending.pop(0)
        if work is None:
            return slot[: self.n_items]
        work.wait()                              # orders the current stream behind the collective; no host sync
        return self.recv[slot][: self.n_items]


def world_census(device, group=None):
    """What the communicator itself reports: backend, world size, and the sum of one 1 per rank through an
    all-reduce (= the number of ranks that really took part)."""
    if not (dist.is_available() and dist.is_initialized()):
        return {"backend": None, "world_size": 1, "ranks_counted": 1}
    one = torch.ones(1, dtype=torch.float32, device=device)
    dist.all_reduce(one, group=group)
    return {"backend": dist.get_backend(group), "world_size": dist.get_world_size(group),
            "ranks_counted": int(round(float(one.item())))}


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
