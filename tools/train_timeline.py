#!/usr/bin/env python3
"""Where a training step's wall time goes, from a rocprofv3 kernel trace of tools/bench_train.py (raw
`*_kernel_trace.csv`: start / end stamps per dispatch). The last `--steps` steps are cut at the sampler kernel that opens
each forward; per step: wall = first start .. next step's first start, busy = union of kernel intervals, idle = wall - busy
(launch gaps: the GPU waiting for the host or for a dependent dispatch), and kernel time by family.
    python tools/train_timeline.py gpurun_out/prof_kt_train/kt_kernel_trace.csv [--steps 8] [--list]"""
import argparse
import csv
import json
import re
from collections import defaultdict

FAMILIES = [("gemm: linear", r"tr_linear_(pers|ring|x3)|tr_linear_kernel"), ("gemm: pooled conv5", r"tr_linear_pool"),
            ("gemm: wgrad", r"tr_wgrad_(kernel|x3)"), ("wgrad second stage", r"tr_wgrad_final"),
            ("narrow layers (conv1, dconv5) on VALU", r"tr_head2_|tr_conv1_"), ("fc tails (rows = items)", r"tr_fc_"),
            ("bn: forward statistics", r"tr_colred_kernel<0|tr_stats"), ("bn: backward sums", r"tr_colred_kernel<1"),
            ("bn: reduction second stages", r"tr_colred_final|tr_blocksum_final|tr_segsum_final"),
            ("bn: backward apply", r"tr_bnbwd_apply"), ("dropout / activation", r"tr_act|tr_colred_kernel<2"),
            ("pooled layer glue", r"tr_pool_|tr_segmax"), ("criterion, parse_output_to_tensors", r"tr_seg_ce|tr_box_loss|parse_box_pred"),
            ("packing", r"tr_pack"), ("sampler", r"compact_sample|mask_"),
            ("torch: optimizer", r"multi_tensor_apply|fused_adam|FusedAdam"), ("torch: fill", r"FillFunctor"),
            ("torch: copy", r"direct_copy|copyBuffer"), ("torch: other", r"at::native|Cijk_|rocblas")]


def family(name):
    for fam, pat in FAMILIES:
        if re.search(pat, name):
            return fam
    return "other: " + name.split("(")[0][:40]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--marker", default="compact_sample", help="kernel that opens a step's forward")
    ap.add_argument("--list", action="store_true", help="print one step's dispatches in order with the gap in front of each")
    a = ap.parse_args()
    rows = []
    for r in csv.DictReader(open(a.trace)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    opens = [i for i, r in enumerate(rows) if a.marker in r[2]]
    # a step may hold several sampler launches (one per head): keep openings at least 2 ms apart
    cuts = []
    for i in opens:
        if not cuts or rows[i][0] - rows[cuts[-1]][0] > 2_000_000:
            cuts.append(i)
    cuts = cuts[-(a.steps + 1):]
    steps = []
    for lo, hi in zip(cuts, cuts[1:]):
        seg = rows[lo:hi]
        wall = rows[hi][0] - seg[0][0]
        busy, end = 0, seg[0][0]
        for s, e, _ in seg:
            if e > end:
                busy += e - max(s, end)
                end = e
        fam = defaultdict(lambda: [0, 0])
        for s, e, n in seg:
            f = fam[family(n)]
            f[0] += e - s
            f[1] += 1
        steps.append((wall, busy, len(seg), fam, seg))
    n = len(steps)
    wall = sum(s[0] for s in steps) / n / 1e6
    busy = sum(s[1] for s in steps) / n / 1e6
    out = {"steps": n, "wall_ms": round(wall, 3), "busy_ms": round(busy, 3), "idle_ms": round(wall - busy, 3),
           "dispatches_per_step": round(sum(s[2] for s in steps) / n, 1), "families_ms_per_step": {}}
    tot = defaultdict(lambda: [0, 0])
    for s in steps:
        for k, (t, c) in s[3].items():
            tot[k][0] += t
            tot[k][1] += c
    for k, (t, c) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
        out["families_ms_per_step"][k] = {"ms": round(t / n / 1e6, 3), "launches": round(c / n, 1)}
    print(json.dumps(out, indent=1))
    if a.list:
        seg = steps[-1][4]
        prev = seg[0][0]
        for s, e, nm in seg:
            print(f"{(s - seg[0][0]) / 1e3:9.1f} us  gap {max(s - prev, 0) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  {nm[:110]}")
            prev = max(prev, e)


if __name__ == "__main__":
    main()
