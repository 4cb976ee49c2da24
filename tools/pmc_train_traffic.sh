#!/bin/bash
# HBM-side traffic of the training step's kernels: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes over
# tools/bench_train.py (KiB; FETCH_SIZE x 2 on gfx950 for wide coalesced reads, MI355X_MICROARCH.md), per kernel and grid,
# next to the kernel's time: a pass that moves more than its tensors (algorithmic bytes: r04_train_roofline.json) re-reads.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_train_traffic
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
. $R/tools/_pmc_lib.sh
P="python3 $R/tools/bench_train.py --backends ${1:-hip} --sampler device --iters 3"
pmc_pass $O/p1 FETCH_SIZE -- $P
pmc_pass $O/p2 WRITE_SIZE -- $P
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
    rd, wr = m.get("FETCH_SIZE", 0) * 1024 * 2, m.get("WRITE_SIZE", 0) * 1024
    if rd + wr < 16e6:
        continue
    rows.append((m["ns"] * len(c["ns"]), k, m["ns"] / 1e3, rd / 1e6, wr / 1e6, (rd + wr) / m["ns"] / 1e3, len(c["ns"]) // 2))
print(f"{'kernel (grid)':84s} {'us':>7s} {'read MB':>8s} {'write MB':>8s} {'TB/s':>5s} {'n':>3s}")
for _, k, us, rd, wr, tbs, n in sorted(rows, reverse=True):
    print(f"{k[:84]:84s} {us:7.1f} {rd:8.1f} {wr:8.1f} {tbs:5.2f} {n:3d}")
PY
exit $PROF_RC
