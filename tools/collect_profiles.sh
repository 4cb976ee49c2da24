#!/bin/bash
# tools/collect_profiles.sh [TAG] — after `gpurun -- 'bash tools/profile_round.sh TAG'`: copy what that run left under
# gpurun_out/ (scratch) into profiles/ (tracked) under the round's names. Refuses a run that reported a failed pass.
set -e
TAG=${1:-r06}
cd "$(dirname "$0")/.."
O=gpurun_out
grep -q "PROF_RC=0" $O/profile_round.status || { echo "collect_profiles: the profile round did not finish clean ($O/profile_round.status)"; exit 1; }
for f in bench bench_extras bench_under_rocprof bench_bf16_under_rocprof bench_f16x3_under_rocprof bench_c3_under_rocprof bench_c5_under_rocprof \
         bench_b64_under_rocprof bench_b64n1024_under_rocprof bench_b512_under_rocprof share_efficiency bench_maxpool_under_rocprof bench_maxpool_bf16_under_rocprof bench_latency bench_train \
         bench_train_dynamic bench_train_under_rocprof bench_train_fused_adam bench_train_graph bench_rehearsal_2ranks bench_rehearsal_2ranks_c4 \
         train_timeline train_timeline_f16x3 cpu_threads crops_range_image crops_shuffled pipeline_fp32 pipeline_bf16; do
  [ -s $O/$f.json ] && cp $O/$f.json profiles/${TAG}_$f.json
done
[ -s $O/bench_full_headline.json ] && cp $O/bench_full_headline.json profiles/${TAG}_bench_full.json
for f in $O/${TAG}_kernel_stats*.csv $O/${TAG}_train_kernel_stats*.csv $O/${TAG}_pmc*.json $O/${TAG}_pmc_c3.txt $O/${TAG}_pmc_crops.txt $O/traffic.json; do
  [ -s $f ] && cp $f profiles/
done
[ -s $O/train_roofline_hip.json ] && cp $O/train_roofline_hip.json profiles/${TAG}_train_roofline.json
[ -s $O/train_roofline_hip_f16x3.json ] && cp $O/train_roofline_hip_f16x3.json profiles/${TAG}_train_roofline_f16x3.json
[ -s $O/pmc_dec/summary.txt ] && cp $O/pmc_dec/summary.txt profiles/${TAG}_pmc_decode_traffic.txt
[ -s $O/train_pmc.txt ] && cp $O/train_pmc.txt profiles/${TAG}_train_pmc.txt
[ -s $O/train_traffic.txt ] && cp $O/train_traffic.txt profiles/${TAG}_train_traffic.txt
ls profiles | grep "^${TAG}_" | wc -l
