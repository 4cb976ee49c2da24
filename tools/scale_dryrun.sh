#!/bin/bash
# tools/scale_dryrun.sh — the N > 1 bench path without GPUs: for N in 2 4 8, `bench.py --gpus N
# --plumbing-only --config C2|C4` on gloo — the self-launch, the process group, the workloads' real item counts and
# rank -> range maps, the overlapped gather — and a check of rank 0's ONE line: strict JSON, <= 4096 bytes, n_gpus == N,
# rccl.ranks_counted == N, rccl.transport (N peer-access rows), C2's `strong` record (4096 crops in N ranges),
# gather_equals_single_rank true, N per-rank step times. UNMEASURED ON HARDWARE: no multi-GPU
# node has been available to this repo; the first `bench.py --gpus 8` on one prints the same line with RCCL in `rccl`.
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
export DAL3_BENCH_BACKEND=gloo OMP_NUM_THREADS=1
unset RANK LOCAL_RANK WORLD_SIZE MASTER_ADDR MASTER_PORT || true
for N in ${@:-2 4 8}; do
  for CFG in C2 C4; do
    LINE=$(python3 "$R/bench.py" --gpus "$N" --plumbing-only --config "$CFG")
    N=$N CFG=$CFG python3 - "$LINE" <<'PY'
import json, os, sys
line, n = sys.argv[1], int(os.environ["N"])
assert "\n" not in line and len(line.encode()) <= 4096, len(line)
def bad(c):
    raise SystemExit(f"non-strict constant {c}")
r = json.loads(line, parse_constant=bad)
assert r["n_gpus"] == n and r["rccl"]["ranks_counted"] == n and r["rccl"]["world_size"] == n, r["rccl"]
assert r["gather_equals_single_rank"] is True and r["gathered_ok"] is True
assert len(r["ms_per_step_per_rank"]) == n
tr = r["rccl"]["transport"]
assert tr["backend"] == "gloo" and len(tr["peer_access"]) == n, tr
if os.environ["CFG"] == "C2":      # the strong record beside the weak value: ONE 4096-crop batch split in ranges of ceil(4096 / N)
    st = r["strong"]
    assert st["scaling"] == "strong" and st["items"] == 4096 and st["items_per_rank"] == [4096 // n] * n and st["gathered_ok"] is True, st
else:
    assert r["strong"] is None
print(f"ok N={n} {os.environ['CFG']}: {len(line)} bytes, heads {[(h['head'], h['items_per_rank']) for h in r['config']['heads']]}")
PY
  done
done
