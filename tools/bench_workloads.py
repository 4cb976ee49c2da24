"""tools/bench_workloads.py — the synthetic workloads bench.py times (BASELINE.json configs C2..C5 and the reference's
other model classes at those shapes): models with synth.py's weights, inputs resident on the device, and the ONE place
a workload's sharding over ranks is written (workload_shards). No timing here."""
import argparse
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
arch = importlib.import_module("3dal_pytorch_amd.arch")
synth = importlib.import_module("3dal_pytorch_amd.synth")
sm = importlib.import_module("3dal_pytorch_amd.static_model")
dm = importlib.import_module("3dal_pytorch_amd.dynamic_model")
dal3_dist = importlib.import_module("3dal_pytorch_amd.dist")


def recentre(model, fwd):
    """shift the segmentation bias so that about half the points are segmented (synth.py); done with one full-size
    pass so that every profiled launch of a kernel has the same shape"""
    with torch.no_grad():
        lg = fwd()["logits"]
        model.ins_seg.dconv5.bias[1] -= (lg[:, :, 1] - lg[:, :, 0]).mean()
        del lg
    model.invalidate_packed()


def storage_of(precision):
    if precision == "f16x3":                               # fp32 accuracy: fp32-stored points, like the fp32 path
        return torch.float32
    """how a workload's points (and box windows) are STORED on the device: BASELINE.json's 16-bit configurations say
    "bf16 storage" (C3: bf16 arithmetic; C5: bf16 storage, fp16 MFMA), the fp32 ones fp32. The kernels read either in
    place (dal3_bcn.dtype); no fp32 copy of 16-bit points is made."""
    return torch.float32 if precision == "fp32" else torch.bfloat16


def make_static(B, N, dev, first, precision="fp32", two=False):
    pts_np, init_np, gt_np = synth.static_crops(B, N, first=first)
    model = sm.StaticModelTwoBoxEst() if two else sm.StaticModelOneBoxEst()
    sd = synth.state_dict("static_two" if two else "static_one")
    model.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    model = model.to(dev).eval()
    pts = torch.from_numpy(pts_np).to(dev).to(storage_of(precision)).transpose(2, 1)   # the callers' layout (static_eval.py:265)
    init, gt = torch.from_numpy(init_np).to(dev), torch.from_numpy(gt_np).to(dev)
    recentre(model, lambda: model(pts, init, gt))
    replicate_weights(model)
    model.item_offset = first
    model.precision = precision
    return model, (pts, init, gt), (pts_np, init_np, sd)


def static_inputs(first, count, N, dev, precision="fp32"):
    """refine() arguments for the global crops [first, first + count): what the rank that owns them holds"""
    pts_np, init_np, gt_np = synth.static_crops(count, N, first=first)
    return (torch.from_numpy(pts_np).to(dev).to(storage_of(precision)).transpose(2, 1), torch.from_numpy(init_np).to(dev),
            torch.from_numpy(gt_np).to(dev))


def dynamic_inputs(first, count, n_per_frame, dev, precision="fp32"):
    pts_np, box_np, init8_np, _ = synth.dynamic_items(count, n_per_frame=n_per_frame, first=first)
    st = storage_of(precision)
    return (torch.from_numpy(pts_np).to(dev).to(st).transpose(2, 1), torch.from_numpy(box_np).to(dev).to(st).transpose(2, 1),
            torch.from_numpy(init8_np).to(dev))


def replicate_weights(model):
    """Weights are replicated over the ranks. synth's weights are a function of the seed, identical
    everywhere; the one rank-dependent value is the segmentation bias `recentre` shifts by the mean margin of the
    rank's OWN crops — rank 0's is broadcast (start-up, outside every timed region), so that any rank can reproduce
    any other rank's boxes bit for bit (gather_self_check)."""
    dal3_dist.replicate_(model.ins_seg.dconv5.bias)
    model.invalidate_packed()


def make_dynamic(B, dev, first, precision="fp32", n_per_frame=1024):
    pts_np, box_np, init8_np, gt_np = synth.dynamic_items(B, n_per_frame=n_per_frame, first=first)
    model = dm.DynamicModel()
    model.load_state_dict({k: torch.as_tensor(v) for k, v in synth.state_dict("dynamic").items()})
    model = model.to(dev).eval()
    pts = torch.from_numpy(pts_np).to(dev).to(storage_of(precision)).transpose(2, 1)
    box = torch.from_numpy(box_np).to(dev).to(storage_of(precision)).transpose(2, 1)
    init8 = torch.from_numpy(init8_np).to(dev)
    recentre(model, lambda: model(pts, box, None))
    replicate_weights(model)
    model.item_offset = first
    model.precision = precision
    return model, (pts, box, init8)


class Part:
    """one head of a workload on this rank: `run()` -> this rank's (n_local,7) boxes; `shard(rank)` -> (lo, n) of any
    rank's contiguous range of the head's n_total items; `inputs_for(first, count)` -> refine() arguments of global
    items [first, first + count) (what gather_self_check recomputes a peer's rows from)"""

    def __init__(self, name, model, inputs, n_local, n_total, shard, inputs_for):
        self.name, self.model, self.inputs = name, model, inputs
        self.n_local, self.n_total, self.shard, self.inputs_for = n_local, n_total, shard, inputs_for

    def run(self):
        return self.model.refine(*self.inputs)[: self.n_local]

    def __iter__(self):                                    # (fn, n_local, n_total), the shape older call sites unpack
        return iter((self.run, self.n_local, self.n_total))


class Workload:
    """what one rank does per step: `parts` = one Part per head"""

    def __init__(self):
        self.parts = []
        self.gatherers = None


def c4_segment_sizes():
    """198 frames; 64 static tracks -> 64 crops at N=4096; 40 dynamic tracks with lengths
    rng.integers(20,199) -> one item per track-frame"""
    lens = np.random.default_rng(10922081).integers(20, 199, size=40)
    return 64, int(lens.sum())


def workload_shards(args, world):
    """[(head, items of the whole job, rank -> (first item, count))] of the configured workload: the ONE place a bench
    workload's sharding is written. build_workload (the GPU run) and plumbing_only (the CPU rehearsal of the N > 1 path,
    world 8 in tests/test_launch_cpu.py) both read it. C4 is a fixed segment split in contiguous index ranges (strong
    scaling, ragged last rank); every other config gives each rank its own B items (weak scaling)."""
    def span(n):
        return lambda r: (lambda lo, hi: (lo, hi - lo))(*dal3_dist.shard_range(n, r, world))
    if args.config == "C4":
        n_static, n_dyn = c4_segment_sizes()
        return [("static", n_static, span(n_static)), ("dynamic", n_dyn, span(n_dyn))]
    static = args.head == "static"
    B = args.batch or (4096 if static else 1024)
    return [("static" if static else "dynamic", B * world, lambda r: (r * B, B))]


def build_workload(args, dev, rank, world):
    wl = Workload()
    prec = args.precision
    if args.config == "C4":
        # contiguous index sharding, static and dynamic batches back to back, one all-gather per head. The segment is
        # fixed: strong scaling.
        (_, n_static, s_span), (_, n_dyn, d_span) = workload_shards(args, world)
        (s_lo, s_n), (d_lo, d_n) = s_span(rank), d_span(rank)
        s_hi, d_hi = s_lo + s_n, d_lo + d_n
        smodel, sin, _ = make_static(max(s_hi - s_lo, 1), 4096, dev, s_lo, prec)
        dmodel, din = make_dynamic(max(d_hi - d_lo, 1), dev, d_lo, prec)
        wl.parts = [Part("static", smodel, sin, s_hi - s_lo, n_static, s_span,
                         lambda first, count: static_inputs(first, count, 4096, dev, prec)),
                    Part("dynamic", dmodel, din, d_hi - d_lo, n_dyn, d_span,
                         lambda first, count: dynamic_inputs(first, count, 1024, dev, prec))]
        wl.model, wl.inputs, wl.host, wl.static = smodel, None, None, False
        wl.B, wl.N = (s_hi - s_lo) + (d_hi - d_lo), 0
        wl.n_total = n_static + n_dyn
        wl.flop_item = (n_static * arch.static_one_flop(4096) + n_dyn * arch.dynamic_flop(5120)) / wl.n_total
        wl.scaling = "strong"
        wl.desc = (f"one synthetic segment: {n_dyn} dynamic items (40 tracks) x 5120 pts + 64 static crops x 4096 pts, "
                   f"{prec}, both heads back to back (BASELINE.json configs[3])")
        return wl
    static = args.head == "static"
    B = args.batch or (4096 if static else 1024)
    N = args.points if static else 5 * args.points
    first = rank * B                                            # weak scaling: B items per GPU
    two = static and getattr(args, "two_stage", False)
    if static:
        model, inputs, host = make_static(B, N, dev, first, prec, two=two)
        flop_item = arch.static_two_flop(N) if two else arch.static_one_flop(N)
        desc = f"StaticModel{'Two' if two else 'One'}BoxEst forward+decode, {B} crops x {N} pts per GPU, {prec}" + \
            (" (BASELINE.json configs[1])" if (B, N, prec) == (4096, 1024, "fp32") else
             " (BASELINE.json configs[4] shape)" if (N, prec) == (4096, "fp16") else "")
    else:
        model, inputs = make_dynamic(B, dev, first, prec, args.points)
        host = None
        flop_item = arch.dynamic_flop(N)
        desc = (f"DynamicModel forward+decode, {B} items x {N} pts + 101 boxes per GPU, {prec} arithmetic"
                + (" (BASELINE.json configs[2])" if (B, N, prec) == (1024, 5120, "bf16") else ""))
    wl.parts = [Part("static" if static else "dynamic", model, inputs, B, B * world, workload_shards(args, world)[0][2],
                     (lambda first, count: static_inputs(first, count, N, dev, prec)) if static else
                     (lambda first, count: dynamic_inputs(first, count, args.points, dev, prec)))]
    wl.model, wl.inputs, wl.host, wl.static = model, inputs, host, static
    wl.B, wl.N, wl.n_total, wl.flop_item, wl.scaling, wl.desc = B, N, B * world, flop_item, "weak", desc
    return wl


def two_stage(args):
    return bool(getattr(args, "two_stage", False))


def apply_config(args):
    if args.config == "C3":
        args.head, args.precision, args.batch, args.points = "dynamic", "bf16", 1024, 1024
    elif args.config == "C5":
        args.head, args.precision, args.batch, args.points = "static", "fp16", 2048, 4096
    elif args.config == "C2":
        args.head, args.precision, args.batch, args.points = "static", "fp32", 4096, 1024
    elif args.config == "TwoBoxEst":                       # the reference's second static model class, C2's shape
        args.head, args.precision, args.batch, args.points, args.two_stage = "static", "fp32", 4096, 1024, True
    elif args.config == "Dynamic_fp32":                    # the dynamic head in the reference's own arithmetic, C3's shape
        args.head, args.precision, args.batch, args.points = "dynamic", "fp32", 1024, 1024
    elif args.config == "C4_f16x3":                        # the mixed segment on the split-fp16 kernels
        args.config, args.precision = "C4", "f16x3"
    elif args.config == "Dynamic_f16x3":                   # the same, split-fp16 arithmetic (fp32 accuracy: profiles/LEDGER_r01_r03.md 5.4)
        args.head, args.precision, args.batch, args.points = "dynamic", "f16x3", 1024, 1024
    elif args.config == "TwoBoxEst_f16x3":
        args.head, args.precision, args.batch, args.points, args.two_stage = "static", "f16x3", 4096, 1024, True

