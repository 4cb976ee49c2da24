#!/bin/bash
# tools/profile_round.sh — run ON the GPU box (gpurun -- 'bash tools/profile_round.sh'): the bench line plus the
# rocprofv3 passes whose summaries tools/prof_summary.py condenses into profiles/. Counters are collected in their
# own passes (never together with a trace domain), as /opt/skills/guides/MI355X_MICROARCH.md prescribes.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kt -o kt -- python3 $R/bench.py --no-extras --steps 10 --warmup 2 > $O/bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kt_bf16 -o kt -- python3 $R/bench.py --no-extras --precision bf16 --steps 10 --warmup 2 > $O/bench_bf16_under_rocprof.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -o c -- python3 $R/bench.py --no-extras --steps 3 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write -o c -- python3 $R/bench.py --no-extras --steps 3 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/prof_mfma -o c -- python3 $R/bench.py --no-extras --steps 3 --warmup 1 > /dev/null 2>&1
ls $O
