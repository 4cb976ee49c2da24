#!/usr/bin/env python3
"""Static instruction mix of every MFMA kernel in a device assembly listing (hipcc -S --cuda-device-only): MFMAs, other
VALU, 64-bit address arithmetic, AGPR moves, float64 operations — whole function, prologue and epilogue included. On the
fp32 MFMA every other VALU instruction takes matrix time (SQ_VALU_MFMA_COEXEC_CYCLES is 0): python tools/isa_mix.py file.s"""
import collections
import re
import sys

funcs, cur = {}, None
for line in open(sys.argv[1]):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        cur = m.group(1)
        funcs[cur] = []
    elif cur:
        funcs[cur].append(line)
for name, lines in funcs.items():
    ops = collections.Counter(l.split()[0] for l in lines if l.startswith("\t") and not l.startswith("\t.") and l.split())
    mf = sum(v for k, v in ops.items() if k.startswith("v_mfma"))
    if mf < 64:
        continue
    valu = sum(v for k, v in ops.items() if k.startswith("v_") and not k.startswith("v_mfma"))
    a64 = sum(ops[k] for k in ("v_lshl_add_u64", "v_add_co_u32_e32", "v_addc_co_u32_e32", "v_add_co_u32_e64", "v_addc_co_u32_e64"))
    agpr = ops["v_accvgpr_read_b32"] + ops["v_accvgpr_write_b32"] + ops["v_accvgpr_mov_b32"]
    f64 = sum(v for k, v in ops.items() if "f64" in k)
    short = re.sub(r"^_Z\d+", "", name)[:64]
    print(f"{short:64s} mfma {mf:5d}  valu {valu:5d} ({valu / mf:4.2f}/mfma)  addr64 {a64:4d}  agpr moves {agpr:4d}  f64 {f64:4d}  "
          f"vmem {sum(v for k, v in ops.items() if k.startswith(('global_', 'buffer_', 'flat_'))):4d}")
