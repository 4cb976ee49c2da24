set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/bench_shares.py --lib variants/r05.so --out gpurun_out/shares_r05.json 2> gpurun_out/shares_r05.log
python tools/bench_shares.py --out gpurun_out/shares_new1.json 2> gpurun_out/shares_new1.log
for B in 8 16 24 32 48 64 96 128 256 512; do
  python tools/ab_kernels.py variants/thrall.so variants/latall.so 3dal_pytorch_amd/lib3dal_hip.so variants/r05.so --B $B --N 1024 --rounds 5
done > gpurun_out/ab_families.log 2>&1
tail -30 gpurun_out/shares_new1.log
