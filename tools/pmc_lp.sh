#!/bin/bash
# tools/pmc_lp.sh OUTDIR BENCH_ARGS... — where a kernel's cycles go: four separate rocprofv3 --pmc passes (counters
# only, no trace domain, the program right behind `--`) over `bench.py --no-extras BENCH_ARGS` (run ON the GPU box);
# tools/pmc_lp.py prints per-kernel ratios / writes the JSON summary. E.g.
#   bash tools/pmc_lp.sh gpurun_out/pmc_c3 --config C3 --steps 3 --warmup 1
# The output directory is cleared first and every pass's exit code and CSV are checked (a failed pass must
# not be summarised from stale files).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$1; shift
case $O in /*) ;; *) O=$R/$O ;; esac
rm -rf "$O"; mkdir -p "$O"
cd /tmp; export TMPDIR=/tmp
P="python3 $R/bench.py --no-extras $*"
rc=0
pass() {   # pass NAME COUNTERS...
  local n=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $O/$n -o c -- $P > $O/$n.log 2>&1
  local e=$?
  if [ $e -ne 0 ] || [ -z "$(find $O/$n -name '*counter_collection.csv' 2>/dev/null | head -1)" ]; then
    echo "pmc_lp.sh: pass $n FAILED (rc $e; see $O/$n.log)"; rc=1
  fi
}
pass p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
pass p2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU
pass p3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
pass p4 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE
exit $rc
