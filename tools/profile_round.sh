#!/bin/bash
# tools/profile_round.sh [TAG] — run ON the GPU box (gpurun -- 'bash tools/profile_round.sh r04'): the bench line plus
# the rocprofv3 passes whose summaries tools/prof_summary.py condenses into profiles/. Counters are collected in their
# own passes (never together with a trace domain), as /opt/skills/guides/MI355X_MICROARCH.md prescribes. The program
# after `--` is python3 itself (no env / bash -c hop: the profiler's preload has initialised the GPU by then).
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
KT="--kernel-trace --stats --output-format csv"
rocprofv3 $KT -d $O/prof_kt -o kt -- python3 $R/bench.py --no-extras --steps 10 --warmup 2 > $O/bench_under_rocprof.json 2>/dev/null
rocprofv3 $KT -d $O/prof_kt_bf16 -o kt -- python3 $R/bench.py --no-extras --precision bf16 --steps 10 --warmup 2 > $O/bench_bf16_under_rocprof.json 2>/dev/null
rocprofv3 $KT -d $O/prof_kt_f16x3 -o kt -- python3 $R/bench.py --no-extras --precision f16x3 --steps 10 --warmup 2 > $O/bench_f16x3_under_rocprof.json 2>/dev/null
rocprofv3 $KT -d $O/prof_kt_c3 -o kt -- python3 $R/bench.py --no-extras --config C3 --steps 10 --warmup 2 > $O/bench_c3_under_rocprof.json 2>/dev/null
rocprofv3 $KT -d $O/prof_kt_c5 -o kt -- python3 $R/bench.py --no-extras --config C5 --steps 10 --warmup 2 > $O/bench_c5_under_rocprof.json 2>/dev/null
# one training step (64 x 4096, fp32, Adam, device sampler): kernel trace of tools/bench_train.py
rocprofv3 $KT -d $O/prof_kt_train -o kt -- python3 $R/tools/bench_train.py --backends hip --sampler device --iters 10 > $O/bench_train_under_rocprof.json 2>/dev/null
rocprofv3 $KT -d $O/prof_kt_train_x3 -o kt -- python3 $R/tools/bench_train.py --backends hip_f16x3 --sampler device --iters 10 > /dev/null 2>&1
python3 $R/tools/bench_train.py --sampler device --backends hip,hip_f16x3,torch > $O/bench_train.json 2>/dev/null
python3 $R/tools/bench_train.py --sampler device --backends hip,hip_f16x3 --adam fused > $O/bench_train_fused_adam.json 2>/dev/null
# the training step kernel by kernel against its rooflines (tools/train_roofline.py: record under the trace, then join),
# and where the step's time goes by kernel family (tools/train_timeline.py)
for BK in hip hip_f16x3; do
  rocprofv3 --kernel-trace --output-format csv -d $O/prof_kt_trroof_$BK -o kt -- python3 $R/tools/train_roofline.py --record $O/train_calls_$BK.json --backend $BK > /dev/null 2>&1
  python3 $R/tools/train_roofline.py --join $O/train_calls_$BK.json $(ls $O/prof_kt_trroof_$BK/*/kt_kernel_trace.csv $O/prof_kt_trroof_$BK/kt_kernel_trace.csv 2>/dev/null | head -1) > $O/train_roofline_$BK.json 2>$O/train_roofline_$BK.err
done
python3 $R/tools/train_timeline.py $(ls $O/prof_kt_train/*/kt_kernel_trace.csv $O/prof_kt_train/kt_kernel_trace.csv 2>/dev/null | head -1) --steps 8 > $O/train_timeline.json 2>/dev/null
python3 $R/tools/train_timeline.py $(ls $O/prof_kt_train_x3/*/kt_kernel_trace.csv $O/prof_kt_train_x3/kt_kernel_trace.csv 2>/dev/null | head -1) --steps 8 > $O/train_timeline_f16x3.json 2>/dev/null
# instruction mix and matrix-pipe busy share of the training step's MFMA kernels (two --pmc passes of their own), and of the pooled layer alone
bash $R/tools/pmc_train.sh hip > $O/train_pmc.txt 2>/dev/null
bash $R/tools/pmc_pool.sh > $O/pool_pmc.txt 2>/dev/null
bash $R/tools/pmc_train_traffic.sh hip > $O/train_traffic.txt 2>/dev/null
python3 $R/tools/bench_train.py --kind dynamic --sampler device --backends hip,hip_f16x3,torch > $O/bench_train_dynamic.json 2>/dev/null
python3 $R/tools/trx_probe.py > $O/trx_probe.txt 2>/dev/null
python3 $R/tools/wgx_probe.py >> $O/trx_probe.txt 2>/dev/null
python3 $R/tools/bench_latency.py > $O/bench_latency.json 2>/dev/null
# the reference's own eval batch (64 crops x 4096 points): where the small kernels between the three big ones show
rocprofv3 $KT -d $O/prof_kt_b64 -o kt -- python3 $R/bench.py --no-extras --batch 64 --points 4096 --steps 20 --warmup 3 > $O/bench_b64_under_rocprof.json 2>/dev/null
# the N > 1 path rehearsed on this box's one GPU: two ranks on device 0, boxes gathered over gloo through pinned host memory
DAL3_BENCH_SHARE_GPU=1 DAL3_BENCH_BACKEND=gloo python3 $R/bench.py --gpus 2 --steps 10 --warmup 3 > $O/bench_rehearsal_2ranks.json 2>/dev/null
DAL3_BENCH_SHARE_GPU=1 DAL3_BENCH_BACKEND=gloo python3 $R/bench.py --gpus 2 --config C4 --steps 2 --warmup 1 --no-extras > $O/bench_rehearsal_2ranks_c4.json 2>/dev/null
# the HBM-roofline kernel on its own: kernel trace, then FETCH_SIZE and WRITE_SIZE in separate passes (fp32 rows, then bf16 rows)
rocprofv3 $KT -d $O/prof_kt_maxpool -o kt -- python3 $R/bench.py --only-maxpool --steps 10 > $O/bench_maxpool_under_rocprof.json 2>/dev/null
rocprofv3 $KT -d $O/prof_kt_maxpool_bf16 -o kt -- python3 $R/bench.py --only-maxpool --maxpool-storage bf16 --steps 10 > $O/bench_maxpool_bf16_under_rocprof.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch_maxpool_bf16 -o c -- python3 $R/bench.py --only-maxpool --maxpool-storage bf16 --steps 3 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write_maxpool_bf16 -o c -- python3 $R/bench.py --only-maxpool --maxpool-storage bf16 --steps 3 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch_maxpool -o c -- python3 $R/bench.py --only-maxpool --steps 3 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write_maxpool -o c -- python3 $R/bench.py --only-maxpool --steps 3 > /dev/null 2>&1
for P in fp32 bf16 f16x3; do
  S=""; [ $P = bf16 ] && S="_bf16"; [ $P = f16x3 ] && S="_f16x3"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch$S -o c -- python3 $R/bench.py --no-extras --precision $P --steps 3 --warmup 1 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write$S -o c -- python3 $R/bench.py --no-extras --precision $P --steps 3 --warmup 1 > /dev/null 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/prof_mfma$S -o c -- python3 $R/bench.py --no-extras --precision $P --steps 3 --warmup 1 > /dev/null 2>&1
done
# the 16-bit configurations at their own shapes: MFMA-busy and the clock the chip holds (GRBM_GUI_ACTIVE / 8 / kernel time)
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $O/prof_insts_f16x3 -o c -- python3 $R/bench.py --no-extras --precision f16x3 --steps 3 --warmup 1 > /dev/null 2>&1
for CFG in C3 C5; do
  L=$(echo $CFG | tr A-Z a-z)
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/prof_mfma_$L -o c -- python3 $R/bench.py --no-extras --config $CFG --steps 3 --warmup 1 > /dev/null 2>&1
done
# the cpu_baseline leg at several thread counts on this box's host (why bench.py pins it to 32 of the affinity cores)
python3 $R/bench.py --cpu-sweep 8 16 32 64 128 > $O/cpu_threads.json 2>/dev/null
python3 $R/tools/prof_summary.py $TAG $O > $O/prof_summary.log 2>&1
cp $R/profiles/${TAG}_* $R/profiles/traffic.json $O/ 2>/dev/null
ls $O
