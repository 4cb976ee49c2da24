#!/bin/bash
# tools/pmc_lp.sh — where the cycles of the 16-bit kernels go: four separate rocprofv3 --pmc passes over
# `bench.py --precision bf16 --no-extras` (run ON the GPU box); tools/pmc_lp.py prints per-kernel ratios.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_lp
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
P="python3 $R/bench.py --no-extras --precision ${1:-bf16} --steps 3 --warmup 1"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/p1 -o c -- $P > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $O/p2 -o c -- $P > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/p3 -o c -- $P > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/p4 -o c -- $P > /dev/null 2>&1
ls $O
