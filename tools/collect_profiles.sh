#!/bin/bash
# tools/collect_profiles.sh [TAG] — after `gpurun -- 'bash tools/profile_round.sh TAG'`: copy what that run left under
# gpurun_out/ (scratch) into profiles/ (tracked) under the round's names.
set -e
TAG=${1:-r04}
cd "$(dirname "$0")/.."
O=gpurun_out
for f in bench bench_under_rocprof bench_bf16_under_rocprof bench_f16x3_under_rocprof bench_c3_under_rocprof bench_c5_under_rocprof \
         bench_b64_under_rocprof bench_maxpool_under_rocprof bench_maxpool_bf16_under_rocprof bench_latency bench_train \
         bench_train_dynamic bench_train_under_rocprof bench_train_fused_adam bench_rehearsal_2ranks bench_rehearsal_2ranks_c4 \
         train_timeline train_timeline_f16x3 cpu_threads; do
  [ -s $O/$f.json ] && cp $O/$f.json profiles/${TAG}_$f.json
done
for f in $O/${TAG}_kernel_stats*.csv $O/${TAG}_train_kernel_stats*.csv $O/${TAG}_pmc*.json $O/traffic.json; do
  [ -s $f ] && cp $f profiles/
done
[ -s $O/train_roofline_hip.json ] && cp $O/train_roofline_hip.json profiles/${TAG}_train_roofline.json
[ -s $O/train_roofline_hip_f16x3.json ] && cp $O/train_roofline_hip_f16x3.json profiles/${TAG}_train_roofline_f16x3.json
[ -s $O/train_pmc.txt ] && cp $O/train_pmc.txt profiles/${TAG}_train_pmc.txt
[ -s $O/pool_pmc.txt ] && cp $O/pool_pmc.txt profiles/${TAG}_pool_pmc.txt
[ -s $O/train_traffic.txt ] && cp $O/train_traffic.txt profiles/${TAG}_train_traffic.txt
[ -s $O/trx_probe.txt ] && cp $O/trx_probe.txt profiles/${TAG}_trx_probe.txt
[ -s $O/pmc_x3.txt ] && cp $O/pmc_x3.txt profiles/${TAG}_pmc_f16x3_detail.txt
ls profiles | grep "^${TAG}_" | wc -l
