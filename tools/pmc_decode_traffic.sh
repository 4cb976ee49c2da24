#!/bin/bash
# tools/pmc_decode_traffic.sh OUTDIR LIB... — where the fp32 decode kernel's HBM-side reads come from:
# per library build (the shipped one and variants/decnoxcd.so = the old block -> (crop, tile) map), three separate
# rocprofv3 --pmc passes (counters only, no trace domain, python3 right behind `--`) over tools/ab_kernels.py at
# 4096 x 1024: FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum. Run ON the GPU box; the summary
# (per build and kernel: bytes per launch, FETCH_SIZE x 2 per MI355X_MICROARCH.md, L2 hit rate) goes to OUTDIR/summary.txt.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$1; shift
case $O in /*) ;; *) O=$R/$O ;; esac
rm -rf "$O"; mkdir -p "$O"
cd /tmp; export TMPDIR=/tmp
. "$R/tools/_pmc_lib.sh"
for LIB in "$@"; do
  T=$(basename "$LIB" .so)
  pmc_pass "$O/${T}_fetch" FETCH_SIZE -- python3 "$R/tools/ab_kernels.py" "$R/$LIB" --B 4096 --N 1024 --rounds 2
  pmc_pass "$O/${T}_write" WRITE_SIZE -- python3 "$R/tools/ab_kernels.py" "$R/$LIB" --B 4096 --N 1024 --rounds 2
  pmc_pass "$O/${T}_tcc" TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum -- python3 "$R/tools/ab_kernels.py" "$R/$LIB" --B 4096 --N 1024 --rounds 2
done
python3 - "$O" "$@" > "$O/summary.txt" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out, libs = sys.argv[1], sys.argv[2:]
print("# tools/pmc_decode_traffic.sh: HBM-side traffic per launch at 4096 x 1024, fp32 kernels (separate --pmc passes)")
print("# algorithmic reads of ins_seg_decode_kernel: points 50.3 MB + per-crop dconv1 terms 8.4 MB = 58.7 MB; writes: logits 33.6 + mask 4.2 MB")
for lib in libs:
    t = os.path.basename(lib)[:-3]
    acc = defaultdict(lambda: defaultdict(list))
    for fn in glob.glob(f"{out}/{t}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            acc[k][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    print(f"\n== {lib}")
    for k in ("ins_seg_decode_kernel", "ins_seg_encode_kernel", "point_head_pers_kernel", "point_head_kernel"):
        if k not in acc:
            continue
        v = {}
        for c, vals in acc[k].items():
            g = max(x for x, _ in vals)
            full = [y for x, y in vals if x == g]
            v[c] = sum(full) / len(full)
        line = f"{k:26s}"
        if "FETCH_SIZE" in v:
            line += f" read {v['FETCH_SIZE'] * 2048 / 1e6:8.1f} MB (FETCH_SIZE {v['FETCH_SIZE']:.0f} KiB x 2)"
        if "WRITE_SIZE" in v:
            line += f"  write {v['WRITE_SIZE'] * 1024 / 1e6:7.1f} MB"
        if "TCC_HIT_sum" in v:
            line += f"  L2 hit {v['TCC_HIT_sum'] / (v['TCC_HIT_sum'] + v['TCC_MISS_sum']):.4f} (hit {v['TCC_HIT_sum']:.3e} miss {v['TCC_MISS_sum']:.3e}) EA_RDREQ {v.get('TCC_EA0_RDREQ_sum', float('nan')):.3e}"
        print(line)
PY
cat "$O/summary.txt"
exit $PROF_RC
