"""tools/bench_kernels.py — per-kernel accounting for bench.py: every kernel of one step through its own C-ABI entry,
timed with HIP events on the launch stream, with its algorithmic and executed work; the `roofline`
object of the dominant kernel; the standalone N-axis max-pool against the HBM roof."""
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
hip = importlib.import_module("3dal_pytorch_amd._hip")
arch = importlib.import_module("3dal_pytorch_amd.arch")

# dense peaks per arithmetic dtype (MI355X_MICROARCH.md "Chip-level parameters": f32 MFMA = v_mfma_f32_32x32x2_f32,
# exact f32; bf16/fp16 ~2.5 PF dense)
MFMA_PEAK_TFLOPS = {"fp32": 157.3, "bf16": 2500.0, "fp16": 2500.0, "f16x3": 2500.0}   # f16x3 executes on the fp16 MFMA
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E spec
DNAME = {"fp32": "f32", "bf16": "bf16", "fp16": "f16", "f16x3": "f16x3 (fp16 MFMA on (hi, lo) split operands, fp32 accumulate)"}
# f16x3: three fp16 MFMAs per multiply-accumulate of the fp32 formulation (w_hi x_hi + w_hi x_lo + w_lo x_hi): the EXECUTED
# work of its MFMA layers is 3 x the algorithmic one, against the fp16 peak; `frac_algorithmic` is then at most 1/3
EXEC_MULT = {"fp32": 1, "bf16": 1, "fp16": 1, "f16x3": 3}


def events_ms(fn, iters, warmup=2):
    """average duration of fn() in ms, HIP events on torch's current stream (= the launch stream)"""
    for _ in range(warmup):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / iters


# ---------------------------------------------------------------------------------------- per-kernel accounting
def kernel_table(model, inputs, static, B, N, iters):
    """Every kernel of one step through its own C-ABI entry, HIP events on the launch stream. Per launch:
    algorithmic GFLOP (what `frac` is computed from) and executed GFLOP (padding, the decode
    kernel's recompute of conv1-2, and — the other way — the object points the point head skips as copies)."""
    lib = hip.lib()
    prec = model.precision
    dt = hip.DTYPES[prec]
    peak = MFMA_PEAK_TFLOPS[prec]
    dev = inputs[0].device
    c_in = 3 if static else 4
    M = arch.NUM_OBJECT_POINT * (1 if static else arch.NUM_FRAME)
    pts = inputs[0]
    x = hip.bcn(pts)
    w = model._cache.get("ins_seg", model.ins_seg, hip.HEAD_INS_SEG, dt)
    g = torch.zeros((B, 1024), device=dev)
    gb = torch.empty((B, 512), device=dev)
    logits = torch.empty((B, N, 2), device=dev)
    mask = torch.empty((B, N), dtype=torch.uint8, device=dev)
    counts = torch.empty((B,), dtype=torch.int32, device=dev)
    idx = torch.empty((B, M), dtype=torch.int32, device=dev)
    obj = torch.empty((B, M, c_in), device=dev)
    feat = torch.empty((B, 512), device=dev)
    gws = torch.empty(max(int(lib.dal3_gather_workspace_bytes(B, N)), 8), dtype=torch.uint8, device=dev)
    st = hip.stream

    def enc():
        hip.check(lib.dal3_ins_seg_encode(hip.ptr(w), dt, c_in, x, B, N, hip.ptr(g), st()))

    def fc():
        hip.check(lib.dal3_ins_seg_global_bias(hip.ptr(w), dt, hip.ptr(g), B, hip.ptr(gb), st()))

    def dec():
        hip.check(lib.dal3_ins_seg_decode(hip.ptr(w), dt, c_in, x, B, N, hip.ptr(gb), hip.ptr(logits), hip.ptr(mask), st()))

    def samp():
        hip.check(lib.dal3_mask_compact_sample(hip.ptr(mask), x, B, N, c_in, M, hip.SAMPLER_DEVICE, None, model.seed,
                                               model.item_offset, hip.ptr(counts), hip.ptr(idx), hip.ptr(obj),
                                               hip.ptr(gws), gws.numel(), st()))
    if static and getattr(model, "two_stage", False):       # (stage two runs on the re-centred copies of the same points)
        heads = [("box_est_one", "one", model.box_est_one, arch.STATIC_BOX_EST, obj.transpose(2, 1), M, counts),
                 ("box_est_two", "two", model.box_est_two, arch.STATIC_BOX_EST, obj.transpose(2, 1), M, counts)]
    elif static:
        heads = [("box_est", "one", model.box_est, arch.STATIC_BOX_EST, obj.transpose(2, 1), M, counts)]
    else:
        heads = [("point_emb", "pe", model.point_emb, arch.POINT_EMB, obj.transpose(2, 1), M, counts),
                 ("box_emb", "be", model.box_emb, arch.BOX_EMB, inputs[1], inputs[1].shape[2], None)]
    enc(), fc(), dec(), samp()
    torch.cuda.synchronize()
    cnt = counts.cpu().numpy()
    out = {}

    def row(name, t, alg_mac, exe_mac, note=None):
        if "fc_kernel" not in name:
            exe_mac = exe_mac * EXEC_MULT[prec]
        tf = 2.0 * alg_mac / (t * 1e-3) / 1e12
        tfe = 2.0 * exe_mac / (t * 1e-3) / 1e12
        # frac_executed = what the silicon did; frac_algorithmic = the reference's work over the same time (above 1 where
        # the kernel skips work the reference formulation does: the point head's duplicated object points)
        r = {"ms": round(t, 4), "algorithmic_gflop": round(2.0 * alg_mac / 1e9, 2),
             "executed_gflop": round(2.0 * exe_mac / 1e9, 2), "tflops_executed": round(tfe, 2),
             "frac_executed": round(tfe / peak, 4), "tflops_algorithmic": round(tf, 2),
             "frac_algorithmic": round(tf / peak, 4)}
        if prec == "f16x3":                                # what the fp32 formulation's work runs at, next to the fp32 MFMA's peak
            r["x_fp32_mfma_peak"] = round(tf / MFMA_PEAK_TFLOPS["fp32"], 3)
        if note:
            r["note"] = note
        out[name] = r
    lp = prec != "fp32"
    sfx = "_x3_kernel" if prec == "f16x3" else "_lp_kernel" if lp else "_kernel"
    row("ins_seg_encode" + sfx, events_ms(enc, iters), arch.ins_seg_encode_mac(c_in) * B * N,
        arch.ins_seg_encode_mac(c_in, True) * B * N)
    row("fc_kernel[dconv1 global term]", events_ms(fc, iters), 1024 * 512 * B, 1024 * 512 * arch._pad(B, 32))
    row("ins_seg_decode" + sfx, events_ms(dec, iters), arch.ins_seg_decode_mac(c_in) * B * N,
        arch.ins_seg_decode_mac(c_in, True) * B * N)
    t = events_ms(samp, iters)
    out["compact_sample_kernel"] = {"ms": round(t, 4), "algorithmic_gflop": 0.0, "executed_gflop": 0.0,
                                    "bytes": int(B * N + B * M * (4 + 4 * c_in)),
                                    "note": "mask -> ordered positives -> M sampled points; integer work"}
    for name, key, mod, table, hx, m, distinct in heads:
        hw = model._cache.get(key, mod, mod.HEAD_KIND, dt)
        hxb = hip.bcn(hx)

        pws = torch.empty(max(int(lib.dal3_point_head_pool_workspace_bytes(B, m)), 16), dtype=torch.uint8, device=dev)

        def pool(hw=hw, hxb=hxb, m=m, distinct=distinct, kind=mod.HEAD_KIND, pws=pws):
            hip.check(lib.dal3_point_head_pool(kind, hip.ptr(hw), dt, hxb, B, m, hip.ptr(distinct), hip.ptr(feat),
                                               hip.ptr(pws), pws.numel(), st()))
        t = events_ms(pool, iters)
        granule = 256 if lp else 32
        exe_pts = arch.head_executed_points(cnt, m, granule) if distinct is not None else B * arch._pad(m, granule)
        pers = not lp and B * ((m + 31) // 32) > 512            # fp32 throughput family: persistent waves over the live-tile worklist
        row(f"point_head{'_pers' if pers else ''}{sfx}[{name}]", t, arch.head_point_mac(table) * B * m, arch.head_point_mac(table, True) * exe_pts,
            note=f"{exe_pts / (B * m):.3f} of the {m} object points per item are computed"
                 + (" (copies skipped)" if distinct is not None else " (padding)"))
    return out, float(cnt.mean())


def maxpool_roofline(dev, iters, dtype=torch.float32):
    """The standalone N-axis max-pool (the HBM-roofline kernel) timed three ways: `achieved` from single launches, each
    between two device fences with HIP events around that one launch; `back_to_back` from HIP events around `iters`
    launches queued behind each other; `host_clock` from the host's clock around a fenced run of launches. (VERDICT r2:
    rocprofv3's kernel trace reads ~6 % longer per launch than the events do in the same process; three clocks that
    agree with each other say which side the difference is on — profiles/LEDGER_r01_r03.md 5.)"""
    rows, n = 4096 * 1024, 1024
    es = torch.empty((), dtype=dtype).element_size()        # bytes = B*C*N*s + B*C*s, s = 4 (fp32) / 2 (bf16, fp16)
    try:
        x = torch.empty((rows, n), device=dev, dtype=dtype)
    except RuntimeError:
        rows = 1024 * 1024
        x = torch.empty((rows, n), device=dev, dtype=dtype)
    x.normal_()
    out = torch.empty(rows, device=dev, dtype=dtype)
    lib = hip.lib()

    def run():
        hip.check(lib.dal3_maxpool_n_dtype(hip.ptr(x), hip.STORAGE[dtype], rows, n, hip.ptr(out), hip.stream()))
    t_b2b = events_ms(run, iters)
    single = []
    for _ in range(max(iters, 5)):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        run()
        b.record()
        b.synchronize()
        single.append(a.elapsed_time(b))
    single.sort()
    t = single[len(single) // 2]
    # a third clock, independent of HIP events and of the profiler: the host's, around a fenced run of launches
    n_wall = 4 * max(iters, 5)
    torch.cuda.synchronize()
    w0 = time.perf_counter()
    for _ in range(n_wall):
        run()
    torch.cuda.synchronize()
    t_wall = (time.perf_counter() - w0) / n_wall * 1e3
    nbytes = rows * n * es + rows * es
    gbs = nbytes / (t * 1e-3) / 1e9
    gbs_b2b = nbytes / (t_b2b * 1e-3) / 1e9
    ok = bool(torch.equal(out[:4096], x[:4096].max(1)[0]))
    del x
    return {"kernel": "maxpool_rows_kernel", "shape": [rows // 1024, 1024, n], "storage": str(dtype).replace("torch.", ""),
            "bound": "hbm", "ms": round(t, 4),
            "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
            "timing": "median of single launches, each between two device fences (= the kernel's own duration)",
            "ms_min": round(single[0], 4),
            "back_to_back": {"ms_per_launch": round(t_b2b, 4), "achieved": round(gbs_b2b, 1),
                             "frac": round(gbs_b2b / HBM_PEAK_GBS, 4),
                             "note": f"{iters} launches queued behind each other, HIP events around the run"},
            "host_clock": {"ms_per_launch": round(t_wall, 4), "launches": n_wall,
                           "note": "time.perf_counter around a fenced run of launches (includes one launch latency + one sync)"},
            "algorithmic_bytes": nbytes, "exact": ok}



def committed_traffic(kernel, precision, B, N):
    """HBM bytes per launch of `kernel` from the COMMITTED rocprofv3 counter passes (profiles/traffic.json, written by
    tools/prof_summary.py from separate --pmc FETCH_SIZE / WRITE_SIZE passes of this very command, FETCH_SIZE x 2 per
    MI355X_MICROARCH.md) — only when those passes were taken at this precision and shape; otherwise None. Counters
    cannot be read from inside the timed process, so this number is never measured in the run that prints it: the
    line names its file (`traffic_from`), and the PMC ratios (MFMA-busy, clock) stay in profiles/."""
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tfile):
        return None, None
    t = json.load(open(tfile))
    shape = t.get("_shape", {})
    kernel = kernel.split("[")[0]
    if precision in shape.get("precisions", []) and (shape.get("B"), shape.get("N")) == (B, N) and kernel in t:
        return t[kernel], "profiles/traffic.json (" + t.get("_taken", "an earlier run").split(",")[0] + ")"
    return None, None


def roofline_of(kr, peak, precision, B, N):
    """the `roofline` object for the dominant MFMA kernel of a kernel table: `achieved` = ALGORITHMIC FLOP per launch
     / the launch's average duration measured live (HIP events on the launch stream)"""
    dom = max((k for k in kr if "frac_algorithmic" in kr[k]), key=lambda k: kr[k]["ms"])
    traffic, src = committed_traffic(dom, precision, B, N)
    return {"kernel": dom, "bound": "mfma", "achieved": kr[dom]["tflops_algorithmic"], "peak": peak, "unit": "TFLOP/s",
            "frac": kr[dom]["frac_algorithmic"], "traffic": traffic, "traffic_from": src,
            "ms_per_launch": kr[dom]["ms"], "algorithmic_gflop_per_launch": kr[dom]["algorithmic_gflop"],
            "executed_gflop_per_launch": kr[dom]["executed_gflop"], "frac_executed": kr[dom]["frac_executed"]}


def executed_gflop_per_step(kr, static, B):
    """every kernel's executed GFLOP + the per-item FC tails (which have no row of their own)"""
    return sum(kr[k]["executed_gflop"] for k in kr) + 2.0 * B * sum(
        ci * co for t in ([arch.STATIC_BOX_EST] if static else [arch.POINT_EMB, arch.BOX_EMB, arch.DYNAMIC_BOX_EST])
        for _, _, ci, co in t["fcs"]) / 1e9

