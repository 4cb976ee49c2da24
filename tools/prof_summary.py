#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/prof_*) into the small summaries kept under profiles/.

  python tools/prof_summary.py <round-tag> <gpurun_out dir>

Writes profiles/<tag>_kernel_stats.csv (the --kernel-trace --stats table, our kernels only),
profiles/<tag>_pmc.json (per-kernel counter averages over the full-size launches; FETCH_SIZE is
doubled per MI355X_MICROARCH.md "HBM": gfx950 tallies 128-B requests at 64 B for wide streams)
and profiles/traffic.json (HBM bytes per launch, read by bench.py's roofline.traffic)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

tag, src = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = os.path.join(root, "profiles")
os.makedirs(out_dir, exist_ok=True)
OURS = ("ins_seg_", "point_head_", "maxpool_rows", "fc_kernel", "fc39_decode", "compact_sample", "decode_boxes",
        "segment_counts", "recenter_kernel", "pack_", "generic_layer", "tr_", "fill_words", "lat_kernel", "nonfinite_rows",
        "crop_", "item_prep", "writeback")


def short(name):
    return name.split("(")[0].replace("void ", "").split("<")[0]


for sub, sfx in (("prof_kt", ""), ("prof_kt_bf16", "_bf16"), ("prof_kt_f16x3", "_f16x3"), ("prof_kt_c3", "_c3"), ("prof_kt_c5", "_c5"),
                 ("prof_kt_maxpool", "_maxpool"), ("prof_kt_maxpool_bf16", "_maxpool_bf16"), ("prof_kt_b64", "_b64"), ("prof_kt_b64n1024", "_b64n1024"), ("prof_kt_b512", "_b512"),
                 ("prof_kt_train", "_train"), ("prof_kt_train_x3", "_train_x3"), ("prof_kt_pipeline", "_pipeline")):
    stats = glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True)
    if not stats:
        continue
    rows = list(csv.DictReader(open(stats[0])))
    name = (f"{tag}_train_kernel_stats.csv" if sfx == "_train" else f"{tag}_train_kernel_stats_f16x3.csv" if sfx == "_train_x3"
            else f"{tag}_kernel_stats{sfx}.csv")
    with open(os.path.join(out_dir, name), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            if sfx.startswith("_train") or any(k in r["Name"] for k in OURS) or float(r["Percentage"]) > 0.5:
                w.writerow([r["Name"][:160], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                            r["MinNs"], r["MaxNs"], r["StdDev"]])

def collect(dirs, keep=lambda d, k: True):
    acc = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        for fn in glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(fn)):
                k = short(r["Kernel_Name"])
                if not any(o in k for o in OURS) or not keep(d, k):
                    continue
                acc[k][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"]),
                                                  int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return acc


def summarise(acc):
    out = {}
    for k, ctrs in acc.items():
        s = {}
        for c, vals in ctrs.items():
            gmax = max(g for g, _, _ in vals)
            full = [(v, t) for g, v, t in vals if g == gmax]      # full-size launches only
            s[c] = sum(v for v, _ in full) / len(full)
            s.setdefault("launches", len(full))
            s[f"avg_ns_under_{c}"] = sum(t for _, t in full) / len(full)
        out[k] = s
    return out


# the 16-bit configurations at their own shapes (bench.py --config C3 / C5): <tag>_pmc_c3.json, <tag>_pmc_c5.json
for cfg in ("c3", "c5", "f16x3"):
    acc = collect([f"prof_mfma_{cfg}"] + (["prof_insts_f16x3"] if cfg == "f16x3" else []))
    if acc:
        json.dump(summarise(acc), open(os.path.join(out_dir, f"{tag}_pmc_{cfg}.json"), "w"), indent=1, sort_keys=True)
# the max-pool on bf16 rows: its own file (same kernel name as the fp32 run)
acc = collect(["prof_fetch_maxpool_bf16", "prof_write_maxpool_bf16"])
if acc:
    sm = summarise(acc)
    for k, s in sm.items():
        if "FETCH_SIZE" in s and "WRITE_SIZE" in s:
            s["hbm_read_bytes_corrected"] = s["FETCH_SIZE"] * 1024 * 2
            s["hbm_write_bytes"] = s["WRITE_SIZE"] * 1024
    json.dump(sm, open(os.path.join(out_dir, f"{tag}_pmc_maxpool_bf16.json"), "w"), indent=1, sort_keys=True)

pmc = defaultdict(lambda: defaultdict(list))
for d in ("prof_fetch", "prof_write", "prof_mfma", "prof_fetch_bf16", "prof_write_bf16", "prof_mfma_bf16",
          "prof_fetch_f16x3", "prof_write_f16x3", "prof_fetch_maxpool", "prof_write_maxpool"):
    for fn in glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            k = short(r["Kernel_Name"])
            if not any(o in k for o in OURS):
                continue
            if d.endswith("_bf16") and "_lp_" not in k:
                continue                                          # the fp32 passes already hold the shared small kernels
            if d.endswith("_f16x3") and "_x3_" not in k:
                continue
            pmc[k][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"]),
                                              int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
summary, traffic = {}, {}
for k, ctrs in pmc.items():
    s = {}
    for c, vals in ctrs.items():
        gmax = max(g for g, _, _ in vals)
        full = [(v, t) for g, v, t in vals if g == gmax]          # full-size launches only
        s[c] = sum(v for v, _ in full) / len(full)
        s.setdefault("launches", len(full))
        s[f"avg_ns_under_{c}"] = sum(t for _, t in full) / len(full)
    if "FETCH_SIZE" in s and "WRITE_SIZE" in s:
        # rocprofv3 reports both in KiB; FETCH_SIZE x2 on gfx950 for wide coalesced streams
        s["hbm_read_bytes_corrected"] = s["FETCH_SIZE"] * 1024 * 2
        s["hbm_write_bytes"] = s["WRITE_SIZE"] * 1024
        traffic[k] = round(s["hbm_read_bytes_corrected"] + s["hbm_write_bytes"])
    summary[k] = s
json.dump(summary, open(os.path.join(out_dir, f"{tag}_pmc.json"), "w"), indent=1, sort_keys=True)
import datetime                                              # noqa: E402
traffic["_shape"] = {"precisions": ["fp32", "bf16", "f16x3"], "B": 4096, "N": 1024}   # what the PMC passes ran (`_lp_` rows: bf16, `_x3_`: f16x3)
traffic["_taken"] = f"{tag}, {datetime.date.today().isoformat()}, tools/profile_round.sh"
json.dump(traffic, open(os.path.join(out_dir, "traffic.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(summary, indent=1, sort_keys=True))
