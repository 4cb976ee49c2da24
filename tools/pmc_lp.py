#!/usr/bin/env python3
"""Per-kernel counter averages from tools/pmc_lp.sh (gpurun_out/pmc_lp/p*/): python tools/pmc_lp.py [dir]"""
import csv
import glob
import sys
from collections import defaultdict

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_lp"
acc = defaultdict(lambda: defaultdict(list))
for fn in glob.glob(src + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        if "_lp_kernel" in k or "_x3_kernel" in k or k in ("ins_seg_decode_kernel", "ins_seg_encode_kernel", "point_head_kernel"):
            acc[k][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
            if "Start_Timestamp" in r and "End_Timestamp" in r:     # the dispatch's own duration, ns (serialised by the profiler)
                acc[k]["_ns"].append((int(r["Grid_Size"]), float(r["End_Timestamp"]) - float(r["Start_Timestamp"])))
for k, ctrs in sorted(acc.items()):
    v = {}
    for c, vals in ctrs.items():
        gmax = max(g for g, _ in vals)
        full = [x for g, x in vals if g == gmax]
        v[c] = sum(full) / len(full)
    print(k)
    for c in sorted(v):
        print(f"   {c:32s} {v[c]:16.0f}")
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
