#!/usr/bin/env python3
"""Diagnostic: what sits between consecutive MFMAs of a kernel in a hipcc -S listing.
  python tools/isa_gaps.py file.s KERNEL_SUBSTRING [first_mfma last_mfma]
prints one line per MFMA: its index, destination, and the instructions (by class) between it and the next one."""
import re, sys
src, pat = sys.argv[1], sys.argv[2]
lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, 10 ** 9)
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and pat in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
def cls(op):
    if op.startswith("v_mfma"): return "MFMA"
    if op.startswith("v_accvgpr"): return "acc"
    if op.startswith("ds_"): return "ds"
    if op.startswith("global_load_lds"): return "dma"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_barrier"): return "BAR"
    if op.startswith(("s_cbranch", "s_branch")): return "br"
    if op.startswith("s_"): return "s"
    if op.startswith("v_"): return "v"
    return "?"
n = -1
cur, head = [], None
def flush():
    if head is not None and lo <= n <= hi:
        c = {}
        seq = []
        for k, t in cur:
            c[k] = c.get(k, 0) + 1
            if k in ("wait", "nop", "BAR", "br"): seq.append(t)
        print(f"{n:5d} {head:60s} " + " ".join(f"{k}:{v}" for k, v in sorted(c.items())) + ("   | " + "; ".join(seq) if seq else ""))
for l in lines[start + 1:end]:
    t = l.strip()
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
        if t.endswith(":") and lo <= n <= hi: cur.append(("lbl", t))
        continue
    t = t.split(";")[0].strip()
    op = t.split()[0]
    k = cls(op)
    if k == "MFMA":
        flush()
        n += 1
        head, cur = " ".join(t.split()[:3]), []
    else:
        cur.append((k, t))
flush()
print("MFMAs:", n + 1)
