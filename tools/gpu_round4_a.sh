#!/bin/bash
# round 4, first GPU pass: the whole -m gpu suite, then the training step under the kernel trace (raw trace kept: the
# timeline analysis of tools/train_timeline.py needs start/end stamps, not the --stats summary)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python3 -m pytest $R/tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc $?" >> $O/gputest.log
tail -5 $O/gputest.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kt_train -o kt -- python3 $R/tools/bench_train.py --backends hip --sampler device --iters 10 > $O/bench_train_under_rocprof.json 2>/dev/null
python3 $R/tools/bench_train.py --sampler device --backends hip,hip_f16x3 > $O/bench_train.json 2>/dev/null
python3 $R/tools/bench_train.py --sampler device --backends hip,hip_f16x3 --adam fused > $O/bench_train_fused.json 2>/dev/null
cat $O/bench_train.json $O/bench_train_fused.json
ls $O/prof_kt_train/*
