#!/usr/bin/env python3
"""tools/bench_extras.py — everything bench.py measures BESIDE its headline, written to a file, never printed on the
bench line (the line that carried all of this grew past what the driver parses).

  python tools/bench_extras.py [--out gpurun_out/bench_full.json] [--steps 20] [--only lowprec,f16x3,maxpool,...]

One GPU, C2's workload (4096 crops x 1024 points, fp32) as the anchor. Legs (each guarded: a leg that fails leaves
`<leg>_error` and the others still run):
  headline        the same timed steps bench.py prints, with the full per-kernel table
  lowprec         the same workload on the bf16 / fp16 MFMA kernels + what that costs against the exact-fp32 path
  f16x3           the fp32 formulation on split-fp16 operands
  maxpool         the standalone N-axis max-pool kernel vs the 8 TB/s HBM roof (fp32 and bf16 rows)
  torch_gpu       the reference's formulation in stock PyTorch-ROCm ops on the same GPU
  next_rows       SURVEY 8(f) N1 / N2 / N3, each alone (the chained measurement is tools/bench_pipeline.py)
  configs         BASELINE.json's other configurations and the reference's other model classes, whole-path rates
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench                                                  # noqa: E402  (time_steps: the one step definition)
from bench_kernels import (MFMA_PEAK_TFLOPS, DNAME, kernel_table, roofline_of, executed_gflop_per_step,   # noqa: E402
                           maxpool_roofline)
from bench_workloads import build_workload, apply_config     # noqa: E402

arch = importlib.import_module("3dal_pytorch_amd.arch")
sm = importlib.import_module("3dal_pytorch_amd.static_model")
time_steps = bench.time_steps


def torch_gpu_baseline(model, inputs, sample=256, iters=3):
    """The reference's own formulation on this GPU: stock PyTorch-ROCm ops (Conv1d/BatchNorm1d/Linear/max through
    MIOpen / rocBLAS, the per-sample NumPy gather loop with its device->host syncs, materialised repeat+cat) — the
    eval-mode run of the train-mode composite in 3dal_pytorch_amd/static_model.py, which mirrors
    tools/static_model.py:117-146 op for op — plus an on-device decode. What a user gets from the reference
    unchanged on an MI355X; reported beside the HIP path, never as `value`."""
    pts, init, _ = inputs
    pts, init = pts[:sample], init[:sample]
    mean = torch.tensor(arch.MEAN_SIZE, device=pts.device)

    def run():
        with torch.no_grad():
            o = sm._train_forward_one(model, pts, init)
            hc, sc = o["heading_scores"].argmax(1), o["size_scores"].argmax(1)
            ar = torch.arange(pts.shape[0], device=pts.device)
            ang = hc.float() * (2 * np.pi / 12) + o["heading_residuals"][ar, hc]
            ang = torch.where(ang > np.pi, ang - 2 * np.pi, ang) + init[:, -1]
            return torch.cat([o["center"], mean[sc] + o["size_residuals"][ar, sc], ang[:, None]], 1)
    saved = (model.train_backend, model.sampler)
    model.train_backend, model.sampler = "torch", "numpy"            # stock ops and the reference's host sampling loop
    try:
        np.random.seed(0)
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
    finally:
        model.train_backend, model.sampler = saved
    return {"value": round(pts.shape[0] / dt, 1), "unit": "object-crops/s", "kind": "port",
            "sample": f"stock PyTorch-ROCm ops (torch {torch.__version__}), reference formulation incl. the host gather "
                      f"loop, {iters} x (B={pts.shape[0]}, N={pts.shape[2]}) fp32 on the same GPU"}


def accuracy_vs_fp32_path(model, inputs, prec):
    """What `prec` costs on THIS input, next to its throughput: the same launch in the exact-fp32 arithmetic
    (1e-6 from the reference's PyTorch-CPU forward, tests/test_gpu_parity.py) is the yardstick. Three runs of the whole
    path: fp32; `prec` free-running (its own mask, its own draws); `prec` with the fp32 run's mask forced
    (mask_override: the device sampler, keyed on the item and the count, then draws the very same points), which
    isolates the box estimator's error from the discrete effect of a flipped point. Box error per parameter group —
    centre (m, absolute), size (relative to the largest size), yaw (rad, absolute) — on the crops whose decoded classes
    agree, with their count (box_err)."""
    keep = model.precision
    with torch.no_grad():
        model.precision = "fp32"
        ref = model._run(*inputs)
        model.precision = prec
        got = model._run(*inputs)
        forced = model._run(*inputs, mask_override=ref["mask"])
    model.precision = keep
    assert torch.equal(forced["obj_idx"], ref["obj_idx"])
    B = ref["mask"].shape[0]
    same = (ref["mask"] == got["mask"]).all(1)
    flipped = int((ref["mask"] != got["mask"]).sum())
    bp = "bp1" if "bp1" in ref else "bp"

    def classes(o):
        return o[bp][:, 3:15].argmax(1), o[bp][:, 27:30].argmax(1)

    def box_err(o):
        """boxes of run `o` against the fp32 run's: on the crops whose heading AND size classes agree (the decoded box is
        continuous in the 39 parameters there), and how many do — a flipped class is a different bin centre / mean size,
        i.e. a discrete event like a flipped mask bit, counted, not averaged"""
        (h0, s0), (h1, s1) = classes(ref), classes(o)
        same = (h0 == h1) & (s0 == s1)
        a, b = o["boxes7"][same].double(), ref["boxes7"][same].double()
        d = (a - b).abs()
        return {"crops_with_the_same_heading_and_size_class": int(same.sum()), "crops": int(same.numel()),
                "on_those": {"centre_m_max_abs": round(d[:, :3].max().item(), 6),
                             "centre_m_median_abs": round(d[:, :3].max(1).values.median().item(), 6),
                             "size_max_rel": round((d[:, 3:6].max() / b[:, 3:6].abs().max()).item(), 6),
                             "yaw_rad_max_abs": round(d[:, 6].max().item(), 6)}}
    return {"logits_max_rel": round(((ref["logits"] - got["logits"]).abs().max() / ref["logits"].abs().max()).item(), 6),
            "mask_bits_flipped": flipped, "mask_bits": int(ref["mask"].numel()),
            "mask_agreement": round(1.0 - flipped / ref["mask"].numel(), 6),
            "crops_with_identical_mask": int(same.sum()), "crops": B,
            "box_params_max_rel_fp32_mask_forced": round(((ref[bp] - forced[bp]).abs().max() / ref[bp].abs().max()).item(), 6),
            "boxes7_fp32_mask_forced": box_err(forced), "boxes7_free_running": box_err(got)}


def next_rows():
    """SURVEY.md 8(f)'s rows either side of the heads, measured in THIS run: N1 crop preparation from the
    resident StaticTrackStore, N2 crop extraction from full sweeps, N3 write-back of the refined boxes with the segment
    flattened once (post.WritebackPlan). Per row: stream time of the device part (HIP events), the algorithmic bytes it
    moves, GB/s and the fraction of the 8 TB/s HBM roof, and the whole call with its host part. The measuring code is
    tools/bench_prep_post.py and tools/bench_crops.py (`measure()`); none of these rows is bandwidth-bound at a
    segment's size — they are launch- and gather-bound, which is what the fractions say."""
    out = {}
    try:
        pp = importlib.import_module("bench_prep_post").measure(1024)
        n1, st = pp["N1_device_batch_of_64"], pp["prepare_static_batch[device, StaticTrackStore, batches of 64]"]
        out["N1"] = {"what": "prepare_static_batch: 64 tracks -> (64,3,4096) crops, from the resident StaticTrackStore",
                     "kernel_ms": n1["stream_ms_per_call"], "algorithmic_bytes": n1["algorithmic_bytes"], "gb_per_s": n1["gb_per_s"],
                     "frac_of_hbm_8TBps": n1["frac_of_8TBps"], "whole_call_ms": n1["whole_call_ms"],
                     "stream_ms_with_per_batch_uploads": n1["stream_ms_with_per_batch_uploads"],
                     "one_time_store_build_ms_1024_tracks": st["store_build_ms"], "note": n1["note"]}
        wb = pp["writeback_static[WritebackPlan]"]
        out["N3"] = {"what": f"writeback of {wb['pairs']} (track, frame) pairs into {wb['detections']} detections of a 198-frame segment",
                     "kernel_ms": wb["stream_ms_per_launch"], "algorithmic_bytes": wb["algorithmic_bytes"], "gb_per_s": wb["gb_per_s"],
                     "frac_of_hbm_8TBps": wb["frac_of_8TBps"], "whole_call_ms": wb["apply_call_ms"],
                     "one_time_plan_build_ms": wb["plan_build_ms"], "one_shot_call_ms": pp["writeback_static"]["call_ms"],
                     "note": wb["note"]}
    except Exception as e:                                  # (a row that cannot run must not take the headline with it)
        out["N1_N3_error"] = repr(e)
    try:
        for order in ("range_image", "shuffled"):
            c = importlib.import_module("bench_crops").measure(order=order)
            out["N2" if order == "range_image" else "N2_shuffled_points"] = {
                "what": "extract_crops: " + c["workload"], "kernel_ms": c["device_ms"],
                "algorithmic_bytes": c["roofline"]["algorithmic_bytes"], "gb_per_s": c["roofline"]["achieved"],
                "frac_of_hbm_8TBps": c["roofline"]["frac"], "whole_call_ms": c["call_ms_with_host_setup"],
                "point_box_tests_per_s_e9": c["point_box_tests_per_s"],
                "planned_call_ms": c["planned_call_ms"],
                "note": "latency-bound, not HBM- or VALU-bound (profiles/r05_pmc_crops.txt): a grid lookup and a point's own "
                        "candidates' face tests, eight waves per SIMD hiding each other's LDS / memory latency"}
    except Exception as e:
        out["N2_error"] = repr(e)
    return out


def other_config(name, dev, steps):
    """one of BASELINE.json's other configurations on this GPU: whole-path rate (same step definition)"""
    ns = argparse.Namespace(config=name, head="static", precision="fp32", batch=0, points=1024, two_stage=False)
    apply_config(ns)
    wl = build_workload(ns, dev, 0, 1)
    dt, per_step, _ = time_steps(wl, dev, steps, 5, False)
    value = wl.n_total * steps / dt
    peak = MFMA_PEAK_TFLOPS[ns.precision]
    ms = dt / steps * 1e3
    r = {"workload": wl.desc, "value": round(value, 1), "unit": "items/s", "ms_per_step": round(ms, 3),
         "ms_per_step_min": round(per_step[0], 3), "steps": steps, "dtype": DNAME[ns.precision],
         "algorithmic_gflop_per_item": round(wl.flop_item / 1e9, 4),
         "whole_path_tflops_algorithmic": round(value * wl.flop_item / 1e12, 1),
         "whole_path_mfma_frac_algorithmic": round(value * wl.flop_item / 1e12 / peak, 4)}
    if not name.startswith("C4"):
        kr, _ = kernel_table(wl.model, wl.inputs, wl.static, wl.B, wl.N, iters=max(3, min(steps, 5)))
        r["whole_path_mfma_frac_executed"] = round(executed_gflop_per_step(kr, wl.static, wl.B) / ms / peak, 4)
        r["roofline"] = roofline_of(kr, peak, ns.precision, wl.B, wl.N)
        if ns.precision in ("bf16", "fp16"):                # a 16-bit rate is half a result without its error on the same input
            try:
                r["vs_exact_fp32_path"] = accuracy_vs_fp32_path(wl.model, wl.inputs, ns.precision)
            except Exception as e:                          # (a side metric must not take the rate with it)
                r["vs_exact_fp32_path_error"] = repr(e)
    del wl
    torch.cuda.empty_cache()
    return r



def lowprec_leg(wl, dev, steps):
    """the same workload on the 16-bit MFMA path (BASELINE.json configs C3/C5 arithmetic)"""
    out = {}
    for prec in ("bf16", "fp16"):
        wl.model.precision = prec
        d, _, _ = time_steps(wl, dev, steps, 5, False)
        d /= steps
        k2, _ = kernel_table(wl.model, wl.inputs, True, wl.B, wl.N, iters=max(3, min(steps, 10)))
        out[prec] = {"value": round(wl.B / d, 1), "unit": "object-crops/s", "ms_per_step": round(d * 1e3, 3),
                     "whole_path_tflops_algorithmic": round(wl.B / d * wl.flop_item / 1e12, 1),
                     "whole_path_mfma_frac_executed": round(
                         executed_gflop_per_step(k2, True, wl.B) / (d * 1e3) / MFMA_PEAK_TFLOPS[prec], 4),
                     "roofline": roofline_of(k2, MFMA_PEAK_TFLOPS[prec], prec, wl.B, wl.N),
                     "kernels": k2}
        try:
            out[prec]["vs_exact_fp32_path"] = accuracy_vs_fp32_path(wl.model, wl.inputs, prec)
        except Exception as e:
            out[prec]["vs_exact_fp32_path_error"] = repr(e)
    wl.model.precision = "fp32"
    return out


def f16x3_leg(wl, dev, steps):
    """the fp32 formulation on the fp16 MFMA: every operand as an (hi, lo) fp16 pair, three MFMAs per product, fp32
    accumulate (profiles/LEDGER_r01_r03.md 5.4); its distance from the exact-fp32 path on this very input"""
    model, inputs, B = wl.model, wl.inputs, wl.B
    with torch.no_grad():
        model.precision = "fp32"
        ref = model(*inputs)
        model.precision = "f16x3"
        got = model(*inputs)
    d, _, _ = time_steps(wl, dev, steps, 5, False)
    d /= steps
    k3, _ = kernel_table(model, inputs, True, B, wl.N, iters=max(3, min(steps, 10)))
    model.precision = "fp32"
    same = (ref["mask"] == got["mask"]).all(1)
    lg = (ref["logits"] - got["logits"]).abs().max().item() / ref["logits"].abs().max().item()
    bx = {k: round(((ref[k] - got[k])[same].abs().max() / ref[k].abs().max()).item(), 9)
          for k in ref if k not in ("logits", "mask") and torch.is_tensor(ref[k]) and ref[k].is_floating_point()
          and ref[k].shape[0] == B}
    return {"value": round(B / d, 1), "unit": "object-crops/s", "ms_per_step": round(d * 1e3, 3), "dtype": DNAME["f16x3"],
            "vs_exact_fp32_path": {"logits_max_rel": round(lg, 9), "mask_bits_flipped": int((ref["mask"] != got["mask"]).sum()),
                                   "mask_bits": int(ref["mask"].numel()), "crops_with_identical_mask": int(same.sum()),
                                   "outputs_max_rel_on_those": bx},
            "whole_path_tflops_algorithmic": round(B / d * wl.flop_item / 1e12, 1),
            "x_fp32_mfma_peak": round(B / d * wl.flop_item / 1e12 / MFMA_PEAK_TFLOPS["fp32"], 3),
            "whole_path_mfma_frac_executed": round(executed_gflop_per_step(k3, True, B) / (d * 1e3) / MFMA_PEAK_TFLOPS["f16x3"], 4),
            "roofline": roofline_of(k3, MFMA_PEAK_TFLOPS["f16x3"], "f16x3", B, wl.N), "kernels": k3}


LEGS = ("headline", "lowprec", "f16x3", "maxpool", "torch_gpu", "next_rows", "configs")
OTHER = (("C3", 10), ("C5", 10), ("TwoBoxEst", 5), ("TwoBoxEst_f16x3", 5), ("Dynamic_fp32", 5), ("Dynamic_f16x3", 5),
         ("C4", 3), ("C4_f16x3", 3))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "bench_full.json"))
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--only", default=",".join(LEGS), help="comma-separated legs: " + ", ".join(LEGS))
    ap.add_argument("--configs", default=",".join(n for n, _ in OTHER), help="which of the `configs` leg's workloads")
    ap.add_argument("--maxpool-iters", type=int, default=5)
    ap.add_argument("--maxpool-storage", default="fp32,bf16", help="row storage(s) of the maxpool leg (profiling runs take one)")
    a = ap.parse_args()
    legs = set(a.only.split(","))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ns = argparse.Namespace(config="C2", head="static", precision="fp32", batch=0, points=1024, two_stage=False)
    apply_config(ns)
    wl = build_workload(ns, dev, 0, 1) if legs & {"headline", "lowprec", "f16x3", "torch_gpu"} else None
    rec = {"written_by": "tools/bench_extras.py", "taken": time.strftime("%Y-%m-%d %H:%M:%S"), "legs": sorted(legs)}

    def leg(name, fn):
        if name not in legs:
            return
        try:
            t0 = time.perf_counter()
            rec[name] = fn()
            rec.setdefault("leg_seconds", {})[name] = round(time.perf_counter() - t0, 1)
        except Exception as e:                              # a side measurement never takes the others with it
            rec[name + "_error"] = repr(e)

    def headline():
        dt, per_step, _ = time_steps(wl, dev, a.steps, a.warmup, False)
        kr, mean_count = kernel_table(wl.model, wl.inputs, True, wl.B, wl.N, iters=10)
        ms = dt / a.steps * 1e3
        return {"workload": wl.desc, "value": round(wl.B * a.steps / dt, 1), "ms_per_step": round(ms, 3),
                "ms_per_step_median": round(per_step[len(per_step) // 2], 3), "ms_per_step_min": round(per_step[0], 3),
                "roofline": roofline_of(kr, MFMA_PEAK_TFLOPS["fp32"], "fp32", wl.B, wl.N), "kernels": kr,
                "mean_segmented_points_per_item": round(mean_count, 1),
                "executed_gflop_per_step": round(executed_gflop_per_step(kr, True, wl.B), 1),
                "algorithmic_gflop_per_step": round(wl.B * wl.flop_item / 1e9, 1),
                "whole_path_mfma_frac_executed": round(executed_gflop_per_step(kr, True, wl.B) / ms / MFMA_PEAK_TFLOPS["fp32"], 4)}
    leg("headline", headline)
    leg("lowprec", lambda: lowprec_leg(wl, dev, a.steps))
    leg("f16x3", lambda: f16x3_leg(wl, dev, a.steps))
    leg("maxpool", lambda: {st: maxpool_roofline(dev, iters=a.maxpool_iters, dtype={"fp32": torch.float32, "bf16": torch.bfloat16,
                                                                                       "fp16": torch.float16}[st])
                            for st in a.maxpool_storage.split(",")})
    leg("torch_gpu", lambda: torch_gpu_baseline(wl.model, wl.inputs))
    del wl
    torch.cuda.empty_cache()
    leg("next_rows", next_rows)
    leg("configs", lambda: {n: other_config(n, dev, st) for n, st in OTHER if n in a.configs.split(",")})
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(rec, f, indent=1, allow_nan=False)
    print(f"bench_extras: wrote {a.out} ({os.path.getsize(a.out)} bytes)")


if __name__ == "__main__":
    main()
