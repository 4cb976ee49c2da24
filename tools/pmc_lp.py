#!/usr/bin/env python3
"""Per-kernel counter averages from tools/pmc_lp.sh (DIR/p*/): python tools/pmc_lp.py DIR [--json OUT.json]
The averages are over the full-size launches of each kernel (largest grid), warm-up launches of the same size included
(they run the same code on the same data)."""
import csv
import glob
import json
import sys
from collections import defaultdict

src = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "gpurun_out/pmc_lp"
MATCH = sys.argv[sys.argv.index("--match") + 1].split(",") if "--match" in sys.argv else []     # further kernel-name substrings
acc = defaultdict(lambda: defaultdict(list))
for fn in glob.glob(src + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        if "_lp_kernel" in k or "_x3_kernel" in k or k in ("ins_seg_decode_kernel", "ins_seg_encode_kernel", "point_head_kernel") \
                or any(m in k for m in MATCH):
            acc[k][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
            if "Start_Timestamp" in r and "End_Timestamp" in r:     # the dispatch's own duration, ns (serialised by the profiler)
                acc[k]["_ns"].append((int(r["Grid_Size"]), float(r["End_Timestamp"]) - float(r["Start_Timestamp"])))
summary = {}
for k, ctrs in sorted(acc.items()):
    v = {}
    for c, vals in ctrs.items():
        gmax = max(g for g, _ in vals)
        full = [x for g, x in vals if g == gmax]
        v[c] = sum(full) / len(full)
    summary[k] = dict(v, launches=len([1 for g, _ in ctrs["_ns"] if g == max(g for g, _ in ctrs["_ns"])]) if "_ns" in ctrs else None)
    if "SQ_WAVE_CYCLES" in v:
        summary[k]["ratios"] = {c + "/SQ_WAVE_CYCLES": round(v[c] / v["SQ_WAVE_CYCLES"], 4)
                                for c in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS", "SQ_BUSY_CYCLES") if c in v}
    if "SQ_INSTS_MFMA" in v and v["SQ_INSTS_MFMA"]:
        summary[k].setdefault("ratios", {})["VALU_per_MFMA"] = round((v.get("SQ_INSTS_VALU", 0) - v["SQ_INSTS_MFMA"]) / v["SQ_INSTS_MFMA"], 3)
        summary[k]["ratios"]["LDS_per_MFMA"] = round(v.get("SQ_INSTS_LDS", 0) / v["SQ_INSTS_MFMA"], 3)
    if "GRBM_GUI_ACTIVE" in v and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
        cyc_ = v["GRBM_GUI_ACTIVE"] / 8
        summary[k].setdefault("ratios", {}).update(mfma_busy=round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc_), 4),
                                                   clock_ghz=round(cyc_ / v["_ns"], 3) if "_ns" in v else None)
    print(k)
    for c in sorted(v):
        print(f"   {c:32s} {v[c]:16.0f}")
    if "FETCH_SIZE" in v or "WRITE_SIZE" in v:               # KiB; FETCH_SIZE x 2 on gfx950 for wide streams (MI355X_MICROARCH.md)
        summary[k]["hbm_bytes"] = {"read_corrected": v.get("FETCH_SIZE", 0) * 2048, "write": v.get("WRITE_SIZE", 0) * 1024}
    if "SQ_ACTIVE_INST_VALU" in v and "GRBM_GUI_ACTIVE" in v:
        # SQ_ACTIVE_INST_VALU counts per SIMD (1024 on the chip), GRBM_GUI_ACTIVE sums the 8 XCDs
        summary[k].setdefault("ratios", {})["valu_active_per_simd_cycle"] = round(v["SQ_ACTIVE_INST_VALU"] / (1024 * v["GRBM_GUI_ACTIVE"] / 8), 4)
    if "SQ_WAVE_CYCLES" in v:
        w = v["SQ_WAVE_CYCLES"]
        for c in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS"):
            if c in v:
                print(f"   {c} / WAVE_CYCLES = {v[c] / w:.3f}")
    if "GRBM_GUI_ACTIVE" in v and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
        cyc = v["GRBM_GUI_ACTIVE"] / 8
        if "_ns" in v:                                      # MI355X_MICROARCH.md 'DVFS give-back': effective clock
            print(f"   effective clock = GRBM_GUI_ACTIVE / 8 / duration = {cyc / v['_ns']:.3f} GHz over {v['_ns'] / 1e6:.3f} ms")
        print(f"   MFMA busy = {v['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc):.3f}; VALU active/SIMD-cycle = "
              f"{v.get('SQ_ACTIVE_INST_VALU', 0) / (1024 * cyc):.3f}; coexec = {v.get('SQ_VALU_MFMA_COEXEC_CYCLES', 0) / (1024 * cyc):.3f}")
if "--json" in sys.argv:
    json.dump(summary, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1, sort_keys=True)
