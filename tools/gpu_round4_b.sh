#!/bin/bash
# round 4: the training tests, then the training step (plain, under the kernel trace)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python3 -m pytest $R/tests/test_gpu_train.py $R/tests/test_gpu_train_reference.py $R/tests/test_gpu_train_x3.py $R/tests/test_gpu_train_fused.py $R/tests/test_gpu_graph.py -x -q > $O/gputest_train.log 2>&1; echo "pytest rc $?" >> $O/gputest_train.log
tail -5 $O/gputest_train.log
python3 $R/tools/bench_train.py --sampler device --backends hip,hip_f16x3 > $O/bench_train.json 2>/dev/null
cat $O/bench_train.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kt_train -o kt -- python3 $R/tools/bench_train.py --backends hip --sampler device --iters 10 > $O/bench_train_under_rocprof.json 2>/dev/null
python3 $R/tools/train_timeline.py $O/prof_kt_train/kt_kernel_trace.csv --steps 8 > $O/train_timeline.json
cat $O/train_timeline.json | head -60
