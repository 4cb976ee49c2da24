#!/usr/bin/env python3
"""Every kernel of ONE training step against its roofline (profiles/r04_train_roofline.json).

Two passes over the same program:
  rocprofv3 --kernel-trace --output-format csv -d OUT -o kt -- python3 tools/train_roofline.py --record OUT/calls.json [--backend hip_f16x3]
      runs the step of tools/bench_train.py (StaticModelOneBoxEst, 64 crops x 4096 points, Adam, device sampler), and while
      its LAST step runs, train.CALLS collects what every per-point launch moves (family, shape, algorithmic FLOP, bytes);
  python3 tools/train_roofline.py --join OUT/calls.json OUT/.../kt_kernel_trace.csv > profiles/r04_train_roofline.json
      matches the last step's dispatches to those records, family by family in launch order, and prices each kernel:
      MFMA-bound kernels as algorithmic FLOP / time against the fp32 MFMA peak (157.3 TFLOP/s; the f16x3 kernels against
      the same fp32-equivalent figure, since they replace fp32 MFMAs), HBM-bound ones as algorithmic bytes / time
      against 8 TB/s; a kernel's bound is the larger of the two fractions' denominators (the roof it is nearer to)."""
import argparse
import csv
import importlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MFMA_PEAK, HBM_PEAK = 157.3e12, 8.0e12

FAMILY_OF_KERNEL = [("linear_pool", r"tr_linear_pool|tr_linear_x3_kernel<(true|false), true>"), ("linear+stats", r"tr_linear_pers_kernel<\d, \d, \d, \d, 1>"),
                    ("linear+bwd_sums", r"tr_linear_pers_kernel<\d, \d, \d, \d, 2>"), ("linear_x3", r"tr_linear_x3"),
                    ("linear", r"tr_linear_(pers|ring)_kernel|tr_linear_kernel"), ("wgrad_x3", r"tr_wgrad_x3"),
                    ("wgrad", r"tr_wgrad_kernel"), ("stats", r"tr_colred_kernel<0"), ("bwd_sums", r"tr_colred_kernel<1"),
                    ("apply", r"tr_bnbwd_apply"), ("act", r"tr_act_dropout|tr_colred_kernel<2"),
                    ("head2", r"tr_head2_(fwd|dgrad|dgrad_sums|wgrad)_kernel"), ("conv1", r"tr_conv1_(fwd|wgrad)_kernel")]


def family(name):
    for fam, pat in FAMILY_OF_KERNEL:
        if re.search(pat, name):
            return fam
    return None


def record(path, backend, iters):
    import numpy as np
    import torch
    synth = importlib.import_module("3dal_pytorch_amd.synth")
    sm = importlib.import_module("3dal_pytorch_amd.static_model")
    losses = importlib.import_module("3dal_pytorch_amd.losses")
    train = importlib.import_module("3dal_pytorch_amd.train")
    B, N, dev = 64, 4096, torch.device("cuda", 0)
    p, i, g = synth.static_crops(B, N, seed=3)
    pts, init, gt = torch.from_numpy(p).to(dev).transpose(2, 1), torch.from_numpy(i).to(dev), torch.from_numpy(g).to(dev)
    labels = ((torch.rand((B, N), device=dev) > 0.6).float(), torch.randn((B, 3), device=dev),
              torch.randint(0, 12, (B,), device=dev), 0.1 * torch.randn((B,), device=dev),
              torch.randint(0, 3, (B,), device=dev), 0.3 * torch.randn((B, 3), device=dev))
    crit = losses.FrustumPointNetLossOneBoxEst()
    model = sm.StaticModelOneBoxEst()
    model.load_state_dict({k: torch.as_tensor(v) for k, v in synth.state_dict("static_one").items()})
    model = model.to(dev).train()
    model.sampler = "device"
    if backend == "hip_f16x3":
        model.precision = "f16x3"
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)

    def step():
        loss = crit(model(pts, init, gt), *labels)["total_loss"]
        opt.zero_grad()
        loss.backward()
        opt.step()
    np.random.seed(0)
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
    train.CALLS = []
    step()                                                   # the recorded step: the LAST one in the trace
    torch.cuda.synchronize()
    json.dump({"backend": backend, "workload": f"StaticModelOneBoxEst train step, {B} crops x {N} pts, Adam, device sampler",
               "calls": train.CALLS}, open(path, "w"))
    train.CALLS = None


def join(calls_path, trace_path):
    rec = json.load(open(calls_path))
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(trace_path)))
    # the last step: from the last sampler launch that is followed by a whole step's worth of kernels backwards — simply
    # take, per family, the LAST len(calls of that family) dispatches of the trace
    by_fam = {}
    for s, e, n in rows:
        f = family(n)
        if f:
            by_fam.setdefault(f, []).append((s, e, n))
    out, tot = [], {"mfma": 0.0, "hbm": 0.0}
    calls_by_fam = {}
    for c in rec["calls"]:
        calls_by_fam.setdefault(c["family"], []).append(c)
    problems = []
    for fam, calls in calls_by_fam.items():
        disp = by_fam.get(fam, [])
        if len(disp) < len(calls):
            problems.append(f"{fam}: {len(calls)} recorded calls, {len(disp)} dispatches in the trace")
            continue
        for c, (s, e, n) in zip(calls, disp[-len(calls):]):
            t = (e - s) * 1e-9
            fm, fh = c["flop"] / t / MFMA_PEAK, c["bytes"] / t / HBM_PEAK
            bound = "mfma" if fm >= fh else "hbm"
            tot[bound] += t
            out.append({"family": fam, "kernel": re.sub(r"\(.*", "", n)[:60], "rows": c["M"], "c_in": c["c_in"], "c_out": c["c_out"],
                        "us": round(t * 1e6, 1), "algorithmic_gflop": round(c["flop"] / 1e9, 2), "algorithmic_mb": round(c["bytes"] / 1e6, 1),
                        "tflops": round(c["flop"] / t / 1e12, 1), "tb_per_s": round(c["bytes"] / t / 1e12, 2), "bound": bound,
                        "frac_of_roof": round(max(fm, fh), 3)})
    out.sort(key=lambda r: -r["us"])
    listed = sum(r["us"] for r in out)
    summary = {"workload": rec["workload"], "backend": rec["backend"], "peaks": {"fp32_mfma_tflops": 157.3, "hbm_tb_per_s": 8.0},
               "kernels_listed": len(out), "listed_us": round(listed, 1),
               "mfma_bound_us": round(tot["mfma"] * 1e6, 1), "hbm_bound_us": round(tot["hbm"] * 1e6, 1),
               "time_weighted_frac_of_roof": round(sum(r["us"] * r["frac_of_roof"] for r in out) / max(listed, 1e-9), 3),
               "not_listed": "second stages of the reductions, the pooled layer's sparse / float64 kernels, criterion, sampler, "
                             "optimizer and stock-torch glue (tools/train_timeline.py gives their share)",
               "problems": problems, "kernels": out}
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--record")
    ap.add_argument("--join", nargs=2)
    ap.add_argument("--backend", default="hip", choices=["hip", "hip_f16x3"])
    ap.add_argument("--iters", type=int, default=8)
    a = ap.parse_args()
    if a.record:
        record(a.record, a.backend, a.iters)
    else:
        join(*a.join)
