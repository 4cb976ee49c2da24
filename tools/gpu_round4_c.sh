#!/bin/bash
# round 4: the training step under the kernel trace — timeline by family, one step's dispatch list, per-kernel roofline join
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kt_train -o kt -- python3 $R/tools/bench_train.py --backends hip --sampler device --iters 10 > $O/bench_train_under_rocprof.json 2>/dev/null
python3 $R/tools/train_timeline.py $O/prof_kt_train/kt_kernel_trace.csv --steps 8 > $O/train_timeline.json
python3 $R/tools/train_timeline.py $O/prof_kt_train/kt_kernel_trace.csv --steps 8 --list > $O/train_step_list.txt
cat $O/train_timeline.json | head -70
rocprofv3 --kernel-trace --output-format csv -d $O/prof_kt_roof -o kt -- python3 $R/tools/train_roofline.py --record $O/train_calls.json --backend hip > /dev/null 2>&1
python3 $R/tools/train_roofline.py --join $O/train_calls.json $O/prof_kt_roof/kt_kernel_trace.csv > $O/train_roofline.json
head -14 $O/train_roofline.json
