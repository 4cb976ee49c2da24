#!/bin/bash
# tools/profile_round.sh [TAG] — run ON the GPU box (gpurun -- 'bash tools/profile_round.sh r05'): the bench line, the
# side measurements (tools/bench_extras.py -> a file) and the rocprofv3 passes whose summaries tools/prof_summary.py
# condenses into profiles/. Counters are collected in passes of their own (never together with a trace domain), as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes; the program behind `--` is python3 itself. Every pass goes through
# tools/_pmc_lib.sh: output directory cleared first, exit code and CSV checked, a failed pass named on stdout and in the
# script's exit code — tools/collect_profiles.sh refuses a run whose log holds such a line.
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
. $R/tools/_pmc_lib.sh
B="python3 $R/bench.py"
X="python3 $R/tools/bench_extras.py"
# ---- the driver's command, as the driver runs it; then everything that is NOT on the line
$B --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cp $O/bench_full.json $O/bench_full_headline.json
$X --out $O/bench_extras.json > $O/bench_extras.log 2>&1
# ---- kernel traces (rocprofv3 --kernel-trace --stats) of the timed steps alone
trace_pass $O/prof_kt -- $B --no-extras --steps 10 --warmup 2;                          cp $O/prof_kt.out $O/bench_under_rocprof.json
trace_pass $O/prof_kt_bf16 -- $B --no-extras --precision bf16 --steps 10 --warmup 2;    cp $O/prof_kt_bf16.out $O/bench_bf16_under_rocprof.json
trace_pass $O/prof_kt_f16x3 -- $B --no-extras --precision f16x3 --steps 10 --warmup 2;  cp $O/prof_kt_f16x3.out $O/bench_f16x3_under_rocprof.json
trace_pass $O/prof_kt_c3 -- $B --no-extras --config C3 --steps 10 --warmup 2;           cp $O/prof_kt_c3.out $O/bench_c3_under_rocprof.json
trace_pass $O/prof_kt_c5 -- $B --no-extras --config C5 --steps 10 --warmup 2;           cp $O/prof_kt_c5.out $O/bench_c5_under_rocprof.json
trace_pass $O/prof_kt_b64 -- $B --no-extras --batch 64 --points 4096 --steps 20 --warmup 3; cp $O/prof_kt_b64.out $O/bench_b64_under_rocprof.json
trace_pass $O/prof_kt_b64n1024 -- $B --no-extras --batch 64 --points 1024 --steps 20 --warmup 3; cp $O/prof_kt_b64n1024.out $O/bench_b64n1024_under_rocprof.json
trace_pass $O/prof_kt_b512 -- $B --no-extras --batch 512 --points 1024 --steps 20 --warmup 3; cp $O/prof_kt_b512.out $O/bench_b512_under_rocprof.json
trace_pass $O/prof_kt_maxpool -- $X --only maxpool --maxpool-storage fp32 --maxpool-iters 10 --out $O/bench_maxpool_under_rocprof.json
trace_pass $O/prof_kt_maxpool_bf16 -- $X --only maxpool --maxpool-storage bf16 --maxpool-iters 10 --out $O/bench_maxpool_bf16_under_rocprof.json
# ---- HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and MFMA-busy + clock per precision at C2's shape
for P in fp32 bf16 f16x3; do
  S=""; [ $P = bf16 ] && S="_bf16"; [ $P = f16x3 ] && S="_f16x3"
  pmc_pass $O/prof_fetch$S FETCH_SIZE -- $B --no-extras --precision $P --steps 3 --warmup 1
  pmc_pass $O/prof_write$S WRITE_SIZE -- $B --no-extras --precision $P --steps 3 --warmup 1
  pmc_pass $O/prof_mfma$S SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- $B --no-extras --precision $P --steps 3 --warmup 1
done
pmc_pass $O/prof_insts_f16x3 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU -- $B --no-extras --precision f16x3 --steps 3 --warmup 1
for CFG in C3 C5; do
  L=$(echo $CFG | tr A-Z a-z)
  pmc_pass $O/prof_mfma_$L SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- $B --no-extras --config $CFG --steps 3 --warmup 1
done
for ST in fp32 bf16; do
  S=""; [ $ST = bf16 ] && S="_bf16"
  pmc_pass $O/prof_fetch_maxpool$S FETCH_SIZE -- $X --only maxpool --maxpool-storage $ST --maxpool-iters 3 --out $O/_mp.json
  pmc_pass $O/prof_write_maxpool$S WRITE_SIZE -- $X --only maxpool --maxpool-storage $ST --maxpool-iters 3 --out $O/_mp.json
done
# ---- the 16-bit decode kernel at C3's own shape: the four counter passes of tools/pmc_lp.sh
bash $R/tools/pmc_lp.sh gpurun_out/pmc_c3 --config C3 --steps 3 --warmup 1 || PROF_RC=1
python3 $R/tools/pmc_lp.py $O/pmc_c3 --json $O/${TAG}_pmc_c3_detail.json > $O/${TAG}_pmc_c3.txt 2>&1
# ---- one training step (64 x 4096, fp32, Adam, device sampler)
T="python3 $R/tools/bench_train.py"
trace_pass $O/prof_kt_train -- $T --backends hip --sampler device --iters 10;        cp $O/prof_kt_train.out $O/bench_train_under_rocprof.json
trace_pass $O/prof_kt_train_x3 -- $T --backends hip_f16x3 --sampler device --iters 10
$T --sampler device --backends hip,hip_f16x3,torch > $O/bench_train.json 2>/dev/null
$T --sampler device --backends hip,hip_f16x3 --adam fused > $O/bench_train_fused_adam.json 2>/dev/null
$T --sampler device --backends hip,hip_f16x3 --graph --graph-optimizer outside > $O/bench_train_graph.json 2>/dev/null
$T --kind dynamic --sampler device --backends hip,hip_f16x3,torch > $O/bench_train_dynamic.json 2>/dev/null
for BK in hip hip_f16x3; do
  rm -rf $O/prof_kt_trroof_$BK
  rocprofv3 --kernel-trace --output-format csv -d $O/prof_kt_trroof_$BK -o kt -- python3 $R/tools/train_roofline.py --record $O/train_calls_$BK.json --backend $BK > /dev/null 2>&1 || { echo "PROFILE PASS FAILED: train_roofline $BK"; PROF_RC=1; }
  python3 $R/tools/train_roofline.py --join $O/train_calls_$BK.json $(ls $O/prof_kt_trroof_$BK/*/kt_kernel_trace.csv $O/prof_kt_trroof_$BK/kt_kernel_trace.csv 2>/dev/null | head -1) > $O/train_roofline_$BK.json 2>$O/train_roofline_$BK.err
done
python3 $R/tools/train_timeline.py $(ls $O/prof_kt_train/*/kt_kernel_trace.csv $O/prof_kt_train/kt_kernel_trace.csv 2>/dev/null | head -1) --steps 8 > $O/train_timeline.json 2>/dev/null
python3 $R/tools/train_timeline.py $(ls $O/prof_kt_train_x3/*/kt_kernel_trace.csv $O/prof_kt_train_x3/kt_kernel_trace.csv 2>/dev/null | head -1) --steps 8 > $O/train_timeline_f16x3.json 2>/dev/null
bash $R/tools/pmc_train.sh hip > $O/train_pmc.txt 2>/dev/null || PROF_RC=1
bash $R/tools/pmc_train_traffic.sh hip > $O/train_traffic.txt 2>/dev/null || PROF_RC=1
# ---- the rows either side of the path: each alone, and chained on the device
python3 $R/tools/bench_crops.py --order range_image > $O/crops_range_image.json 2>/dev/null
python3 $R/tools/bench_crops.py --order shuffled > $O/crops_shuffled.json 2>/dev/null
# (what bounds the crop kernels: vector-ALU activity, waits and HBM bytes per launch, counter passes of their own)
C="python3 $R/tools/bench_crops.py --order range_image --frames 192"
pmc_pass $O/pmc_crops/p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -- $C
pmc_pass $O/pmc_crops/p2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD -- $C
pmc_pass $O/pmc_crops/p3 SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -- $C
pmc_pass $O/pmc_crops/p4 FETCH_SIZE -- $C
pmc_pass $O/pmc_crops/p5 WRITE_SIZE -- $C
python3 $R/tools/pmc_lp.py $O/pmc_crops --match crop_ --json $O/${TAG}_pmc_crops.json > $O/${TAG}_pmc_crops.txt 2>&1
python3 $R/tools/bench_pipeline.py > $O/pipeline_fp32.json 2>/dev/null
python3 $R/tools/bench_pipeline.py --precision bf16 > $O/pipeline_bf16.json 2>/dev/null
trace_pass $O/prof_kt_pipeline -- python3 $R/tools/bench_pipeline.py --precision bf16 --iters 3
python3 $R/tools/bench_latency.py > $O/bench_latency.json 2>/dev/null
# ---- the mid-size regime (round 6): every strong-scaling share and the reference's eval batch as a fraction of the
# large-batch rate; where the decode kernel's HBM-side reads come from (old vs XCD-contiguous block map), when the A/B build exists
python3 $R/tools/bench_shares.py --out $O/share_efficiency.json 2> $O/share_efficiency.log
[ -f $R/variants/decnoxcd.so ] && { bash $R/tools/pmc_decode_traffic.sh gpurun_out/pmc_dec 3dal_pytorch_amd/lib3dal_hip.so variants/decnoxcd.so > /dev/null 2>&1 || PROF_RC=1; }
# ---- the N > 1 path rehearsed on this box's one GPU (two ranks on device 0, boxes gathered over gloo)
DAL3_BENCH_SHARE_GPU=1 DAL3_BENCH_BACKEND=gloo $B --gpus 2 --steps 10 --warmup 3 > $O/bench_rehearsal_2ranks.json 2>/dev/null
DAL3_BENCH_SHARE_GPU=1 DAL3_BENCH_BACKEND=gloo $B --gpus 2 --config C4 --steps 2 --warmup 1 --no-extras > $O/bench_rehearsal_2ranks_c4.json 2>/dev/null
$B --cpu-sweep 8 16 32 64 128 > $O/cpu_threads.json 2>/dev/null
python3 $R/tools/prof_summary.py $TAG $O > $O/prof_summary.log 2>&1
cp -n $R/profiles/${TAG}_* $R/profiles/traffic.json $O/ 2>/dev/null   # (-n: never over what this run wrote into $O itself)
echo "profile_round: PROF_RC=$PROF_RC" | tee $O/profile_round.status
exit $PROF_RC
