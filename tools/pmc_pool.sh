#!/bin/bash
# where the cycles of the pooled layer's kernel go: kernel trace + three rocprofv3 --pmc passes over tools/pool_probe.py
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_pool
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
. $R/tools/_pmc_lib.sh
P="python3 $R/tools/pool_probe.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o c -- $P > /dev/null 2>&1
pmc_pass $O/p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -- $P
pmc_pass $O/p2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU -- $P
pmc_pass $O/p4 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE -- $P
pmc_pass $O/p5 SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM -- $P
grep -h "tr_linear_pool\|tr_pack\|unpack\|fill" $O/kt/c_kernel_stats.csv | cut -c1-160
python3 - <<PY
import csv, glob
from collections import defaultdict
acc = defaultdict(list)
for fn in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "tr_linear_pool" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc["_ns:" + fn.split("/")[-3 if "/p" in fn else -2]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k in sorted(acc):
    v = acc[k][1:] or acc[k]
    print(f"{k:36s} {sum(v) / len(v):16.0f}")
PY
exit $PROF_RC
