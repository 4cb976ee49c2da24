#!/usr/bin/env python3
"""tools/bench_shares.py — the mid-size-job regime on ONE GPU: the step at the sizes a strong-scaling share or the
reference's own eval loop hands the heads, each as a fraction of the large-batch rate.

  python tools/bench_shares.py [--out gpurun_out/share_efficiency.json] [--lib variants/NAME.so] [--only static,ref,dynamic,c4]

Shapes ("next round" 1):
  static   C2's crops, B in {64, 256, 512, 1024, 2048, 4096} x 1024 points, fp32 — B = 512 is C2's share at 8 ranks
  ref      the reference's eval batch: 64 crops x 4096 points (static_eval.py:299,338), fp32
  dynamic  64 items x 5 x 1024 points + 101 boxes (dynamic_eval.py:252,289), fp32 and bf16, next to 1024 items
  c4       rank 0's share of the mixed segment at world 8 (8 static crops x 4096 + ceil(n_dyn / 8) dynamic items)
           next to the whole segment on one GPU

Per shape: ms per step (HIP events around back-to-back refine() calls on the launch stream, median of groups; the
same step as a hipGraph replay where the host would otherwise be the limit), items/s, every kernel's ms through its
own C-ABI entry (tools/bench_kernels.kernel_table), and `of_large_batch_rate` = (algorithmic FLOP / ms) over the same
ratio of the family's largest shape — 1.0 means the share costs exactly its proportional part of the big batch, which
is what strong scaling over W ranks needs from each rank. One JSON document; nothing on the bench line."""
import argparse
import importlib
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
hip = importlib.import_module("3dal_pytorch_amd._hip")


def step_ms(fn, iters, groups=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(groups):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        b.synchronize()
        out.append(a.elapsed_time(b) / iters)
    return statistics.median(out), min(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "share_efficiency.json"))
    ap.add_argument("--lib", default=None, help="another build of the library (variants/NAME.so): A/B runs")
    ap.add_argument("--only", default="static,ref,dynamic,c4")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--no-kernels", action="store_true")
    args = ap.parse_args()
    if args.lib:
        hip.LIB_PATH = os.path.abspath(args.lib)         # before the first hip.lib(): this process runs on that build
    import bench_workloads as W
    from bench_kernels import kernel_table
    arch = importlib.import_module("3dal_pytorch_amd.arch")
    graph = importlib.import_module("3dal_pytorch_amd.graph")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    only = set(args.only.split(","))
    doc = {"library": os.path.relpath(hip.LIB_PATH, ROOT), "device": torch.cuda.get_device_name(0), "unit": "ms per step",
           "timing": f"median of 5 groups of {args.iters} back-to-back refine() calls, HIP events on the launch stream; "
                     "graph_ms = the same step as a hipGraph replay", "families": {}}

    def measure(model, inputs, static, B, N, flop_item, iters):
        fn = lambda: model.refine(*inputs)          # noqa: E731
        ms, ms_min = step_ms(fn, iters)
        rec = {"items": B, "points_per_item": N, "ms": round(ms, 4), "ms_min": round(ms_min, 4)}
        if ms < 3.0:                                # small steps: the host's launches may be the limit; replay a graph too
            cap = graph.CapturedRefine(model, *inputs)
            g_ms, g_min = step_ms(cap.graph.replay, iters)
            rec["graph_ms"] = round(g_ms, 4)
            del cap
        best = min(rec["ms"], rec.get("graph_ms", rec["ms"]))
        rec["items_per_s"] = round(B / best * 1e3, 1)
        rec["algorithmic_gflop"] = round(flop_item * B / 1e9, 2)
        rec["tflops_algorithmic"] = round(flop_item * B / best / 1e9, 2)
        if not args.no_kernels:
            kr, mean_cnt = kernel_table(model, inputs, static, B, N, iters=10)
            rec["kernels_ms"] = {k: v["ms"] for k, v in kr.items()}
            rec["kernels_sum_ms"] = round(sum(v["ms"] for v in kr.values()), 4)
            rec["mean_segmented_points_per_item"] = round(mean_cnt, 1)
        return rec

    def family(name, shapes, ref_key):
        ref = shapes[ref_key]
        for k, r in shapes.items():
            r["of_large_batch_rate"] = round(r["tflops_algorithmic"] / ref["tflops_algorithmic"], 4)
            if "kernels_ms" in r and "kernels_ms" in ref:
                scale = r["algorithmic_gflop"] / ref["algorithmic_gflop"]
                r["kernels_x_proportional"] = {kk: round(v / (ref["kernels_ms"][kk.replace("point_head_kernel", "point_head_pers_kernel")] * scale), 2)
                                               for kk, v in r["kernels_ms"].items()
                                               if kk.replace("point_head_kernel", "point_head_pers_kernel") in ref["kernels_ms"]}
        doc["families"][name] = {"large_batch": ref_key, "shapes": shapes}

    if "static" in only or "ref" in only:
        shapes = {}
        sizes = [(4096, 1024)]
        if "static" in only:
            sizes += [(2048, 1024), (1024, 1024), (512, 1024), (256, 1024), (64, 1024)]
        if "ref" in only:
            sizes += [(64, 4096)]
        for B, N in sizes:
            model, inputs, _ = W.make_static(B, N, dev, 0)
            shapes[f"{B}x{N}"] = measure(model, inputs, True, B, N, arch.static_one_flop(N), args.iters if B <= 1024 else 8)
            del model, inputs
            torch.cuda.empty_cache()
        family("static_fp32", shapes, "4096x1024")
    if "dynamic" in only:
        for prec in ("fp32", "bf16"):
            shapes = {}
            for B in (1024, 256, 128, 64):            # (128 = C3's share at 8 ranks)
                model, inputs = W.make_dynamic(B, dev, 0, prec)
                shapes[f"{B}x5120"] = measure(model, inputs, False, B, 5120, arch.dynamic_flop(5120), args.iters if B <= 256 else 8)
                del model, inputs
                torch.cuda.empty_cache()
            family("dynamic_" + prec, shapes, "1024x5120")
    if "c4" in only:
        n_static, n_dyn = W.c4_segment_sizes()
        dist = importlib.import_module("3dal_pytorch_amd.dist")
        rows = {}
        for label, world in (("whole_segment", 1), ("rank0_of_8", 8)):
            (s_lo, s_hi), (d_lo, d_hi) = dist.shard_range(n_static, 0, world), dist.shard_range(n_dyn, 0, world)
            smodel, sin, _ = W.make_static(s_hi - s_lo, 4096, dev, s_lo)
            dmodel, din = W.make_dynamic(d_hi - d_lo, dev, d_lo)

            def both():
                smodel.refine(*sin)
                dmodel.refine(*din)
            ms, ms_min = step_ms(both, args.iters if world > 1 else 8)
            flop = (s_hi - s_lo) * arch.static_one_flop(4096) + (d_hi - d_lo) * arch.dynamic_flop(5120)
            rec = {"static_crops": s_hi - s_lo, "dynamic_items": d_hi - d_lo, "ms": round(ms, 4), "ms_min": round(ms_min, 4),
                   "items_per_s": round(((s_hi - s_lo) + (d_hi - d_lo)) / ms * 1e3, 1),
                   "algorithmic_gflop": round(flop / 1e9, 2), "tflops_algorithmic": round(flop / ms / 1e9, 2)}
            if not args.no_kernels:
                ks, _ = kernel_table(smodel, sin, True, s_hi - s_lo, 4096, iters=10)
                kd, _ = kernel_table(dmodel, din, False, d_hi - d_lo, 5120, iters=10)
                rec["kernels_ms"] = {"static:" + k: v["ms"] for k, v in ks.items()}
                rec["kernels_ms"].update({"dynamic:" + k: v["ms"] for k, v in kd.items()})
                rec["kernels_sum_ms"] = round(sum(rec["kernels_ms"].values()), 4)
            rows[label] = rec
            del smodel, sin, dmodel, din
            torch.cuda.empty_cache()
        for r in rows.values():
            r["of_large_batch_rate"] = round(r["tflops_algorithmic"] / rows["whole_segment"]["tflops_algorithmic"], 4)
        doc["families"]["c4_fp32"] = {"large_batch": "whole_segment", "shapes": rows}
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(doc, f, indent=1)
    # a short table on stderr (stdout stays empty: this tool writes a file)
    for fam, d in doc["families"].items():
        for k, r in d["shapes"].items():
            sys.stderr.write(f"{fam:14s} {k:14s} {r['ms']:9.4f} ms  graph {r.get('graph_ms', float('nan')):9.4f}  "
                             f"{r['of_large_batch_rate']:.3f} of large  {json.dumps(r.get('kernels_ms', {}))}\n")


if __name__ == "__main__":
    main()
