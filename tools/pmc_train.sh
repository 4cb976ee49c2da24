#!/bin/bash
# the training step's kernels under two rocprofv3 --pmc passes: instruction mix (VALU per MFMA — on the fp32 MFMA the two
# share the SIMD's FMA lanes, SQ_VALU_MFMA_COEXEC_CYCLES is 0) and matrix-pipe busy share per kernel
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_train
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
. $R/tools/_pmc_lib.sh
P="python3 $R/tools/bench_train.py --backends ${1:-hip} --sampler device --iters 3"
pmc_pass $O/p1 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAIT_ANY -- $P
pmc_pass $O/p2 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -- $P
python3 - <<PY
import csv, glob, re
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for fn in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        if not k.startswith("tr_"):
            continue
        k += " g" + r["Grid_Size"]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[k]["ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
rows = []
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    if m.get("SQ_INSTS_MFMA", 0) < 1e5:
        continue
    cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8
    rows.append((m["ns"] * len(c["SQ_INSTS_MFMA"]), k, m["ns"] / 1e3, (m["SQ_INSTS_VALU"] - m["SQ_INSTS_MFMA"]) / m["SQ_INSTS_MFMA"],
                 m["SQ_INSTS_SALU"] / m["SQ_INSTS_MFMA"], m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * cyc) if cyc else 0,
                 cyc / m["ns"] if cyc else 0, len(c["SQ_INSTS_MFMA"])))
print(f"{'kernel':84s} {'us':>7s} {'valu/mfma':>9s} {'salu/mfma':>9s} {'mfma busy':>9s} {'GHz':>5s} {'n':>3s}")
for _, k, us, vm, sm, busy, ghz, n in sorted(rows, reverse=True):
    print(f"{k[:84]:84s} {us:7.1f} {vm:9.2f} {sm:9.2f} {busy:9.3f} {ghz:5.2f} {n:3d}")
PY
exit $PROF_RC
