#!/usr/bin/env python3
"""bench.py — object-crops/sec through the refinement heads on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2|C3|C4|C5] [--precision fp32|bf16|fp16]

With --gpus N > 1 and no launcher in the environment, the process starts the N ranks itself
(3dal_pytorch_amd/launch.py: `python -m torch.distributed.run ... bench.py ...` as a child, before
this process has made any HIP call) and relays rank 0's JSON line; started under
`python -m torch.distributed.run` it is a rank. Either way: one process per GPU over RCCL.

Workload (config.workload): BASELINE.json configs[1] — StaticModelOneBoxEst, 4096 crops x 1024
points, fp32, per GPU, synthetic crops and random-init weights from 3dal_pytorch_amd/synth.py
(no dataset / checkpoint is reachable). One step = one pass of the hot path over the batch:
ins_seg -> mask -> object-point sampling -> box estimator -> decode to (B,7) boxes, inputs
already resident in HBM; with N > 1 every rank refines its own 4096 crops (weak scaling, crops
are independent) and ONE RCCL all-gather of the (N*4096, 7) boxes closes the step. The gather is
asynchronous and collected one step later (dist.BoxGatherer), so it runs beside the next batch's
kernels; every gather is finished inside the timed region.

The JSON line also carries
  roofline      the dominant kernel (an MFMA shared-MLP kernel) timed live with HIP events on the
                launch stream: algorithmic FLOP per launch / average duration vs the MFMA peak of
                the arithmetic type (157.3 TFLOP/s f32, 2.5 PFLOP/s bf16/f16 dense)
  kernels       every kernel of the step with algorithmic AND executed GFLOP per launch
  maxpool       the standalone N-axis max-pool kernel on (4096,1024,1024) fp32 vs 8 TB/s HBM
  configs       BASELINE.json's other configurations on this GPU (C3 dynamic bf16, C5 static N=4096 fp16 MFMA,
                C4 the mixed segment), each a whole-path rate
  rccl          what the communicator reports (world size, ranks counted by an all-reduce) and the
                all-gather's own latency
  cpu_baseline  the oracle (reference-formulation torch-CPU port) on this box's host cores, on
                a bounded sample of the same workload
"""
import argparse
import gc
import ctypes as C
import importlib
import json
import os
import sys
import time

# RCCL shares device buffers between the ranks of a node through HIP IPC handles. The host driver of this pool (ROCm 7.2
# user space on the MI355X boxes) implements only the dmabuf flavour; with the legacy flavour (the runtime's default)
# hipIpcGetMemHandle fails with "invalid argument" as soon as two ranks connect. The pool exports the variable itself;
# setdefault keeps a value the caller exported (a driver that wants the legacy mode sets HSA_ENABLE_IPC_MODE_LEGACY=1)
# and only fills it in for a shell that lost it. It must be set before the HSA runtime loads, hence at import time.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np                                           # noqa: E402
import torch                                                 # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
hip = importlib.import_module("3dal_pytorch_amd._hip")
arch = importlib.import_module("3dal_pytorch_amd.arch")
synth = importlib.import_module("3dal_pytorch_amd.synth")
sm = importlib.import_module("3dal_pytorch_amd.static_model")
dm = importlib.import_module("3dal_pytorch_amd.dynamic_model")
dal3_dist = importlib.import_module("3dal_pytorch_amd.dist")
launch = importlib.import_module("3dal_pytorch_amd.launch")
dal3_graph = importlib.import_module("3dal_pytorch_amd.graph")

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


def recentre(model, fwd):
    """shift the segmentation bias so that about half the points are segmented (synth.py); done with one full-size
    pass so that every profiled launch of a kernel has the same shape"""
    with torch.no_grad():
        lg = fwd()["logits"]
        model.ins_seg.dconv5.bias[1] -= (lg[:, :, 1] - lg[:, :, 0]).mean()
        del lg
    model.invalidate_packed()


def storage_of(precision):
    if precision == "f16x3":                               # fp32 accuracy: fp32-stored points, like the fp32 path
        return torch.float32
    """how a workload's points (and box windows) are STORED on the device: BASELINE.json's 16-bit configurations say
    "bf16 storage" (C3: bf16 arithmetic; C5: bf16 storage, fp16 MFMA), the fp32 ones fp32. The kernels read either in
    place (dal3_bcn.dtype); no fp32 copy of 16-bit points is made."""
    return torch.float32 if precision == "fp32" else torch.bfloat16


def make_static(B, N, dev, first, precision="fp32", two=False):
    pts_np, init_np, gt_np = synth.static_crops(B, N, first=first)
    model = sm.StaticModelTwoBoxEst() if two else sm.StaticModelOneBoxEst()
    sd = synth.state_dict("static_two" if two else "static_one")
    model.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    model = model.to(dev).eval()
    pts = torch.from_numpy(pts_np).to(dev).to(storage_of(precision)).transpose(2, 1)   # the callers' layout (static_eval.py:265)
    init, gt = torch.from_numpy(init_np).to(dev), torch.from_numpy(gt_np).to(dev)
    recentre(model, lambda: model(pts, init, gt))
    replicate_weights(model)
    model.item_offset = first
    model.precision = precision
    return model, (pts, init, gt), (pts_np, init_np, sd)


def static_inputs(first, count, N, dev, precision="fp32"):
    """refine() arguments for the global crops [first, first + count): what the rank that owns them holds"""
    pts_np, init_np, gt_np = synth.static_crops(count, N, first=first)
    return (torch.from_numpy(pts_np).to(dev).to(storage_of(precision)).transpose(2, 1), torch.from_numpy(init_np).to(dev),
            torch.from_numpy(gt_np).to(dev))


def dynamic_inputs(first, count, n_per_frame, dev, precision="fp32"):
    pts_np, box_np, init8_np, _ = synth.dynamic_items(count, n_per_frame=n_per_frame, first=first)
    st = storage_of(precision)
    return (torch.from_numpy(pts_np).to(dev).to(st).transpose(2, 1), torch.from_numpy(box_np).to(dev).to(st).transpose(2, 1),
            torch.from_numpy(init8_np).to(dev))


def replicate_weights(model):
    """Weights are replicated over the ranks. synth's weights are a function of the seed, identical
    everywhere; the one rank-dependent value is the segmentation bias `recentre` shifts by the mean margin of the
    rank's OWN crops — rank 0's is broadcast (start-up, outside every timed region), so that any rank can reproduce
    any other rank's boxes bit for bit (gather_self_check)."""
    dal3_dist.replicate_(model.ins_seg.dconv5.bias)
    model.invalidate_packed()


def make_dynamic(B, dev, first, precision="fp32", n_per_frame=1024):
    pts_np, box_np, init8_np, gt_np = synth.dynamic_items(B, n_per_frame=n_per_frame, first=first)
    model = dm.DynamicModel()
    model.load_state_dict({k: torch.as_tensor(v) for k, v in synth.state_dict("dynamic").items()})
    model = model.to(dev).eval()
    pts = torch.from_numpy(pts_np).to(dev).to(storage_of(precision)).transpose(2, 1)
    box = torch.from_numpy(box_np).to(dev).to(storage_of(precision)).transpose(2, 1)
    init8 = torch.from_numpy(init8_np).to(dev)
    recentre(model, lambda: model(pts, box, None))
    replicate_weights(model)
    model.item_offset = first
    model.precision = precision
    return model, (pts, box, init8)


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


def cpu_baseline(host, budget_s=20.0, threads=None):
    """the oracle (reference formulation) on the host cores, B=64 sample of the same crops"""
    R = importlib.import_module("oracle.ref_heads")
    pts_np, init_np, sd = host
    # `bench.py --cpu-sweep 8 16 32 64 128` on the GPU box (256 logical cores): 8 thr 64, 16 thr 76, 32 thr 80,
    # 64 thr 53, 128 thr 27 crops/s -> the port saturates at 32 threads; more only adds contention
    avail = len(os.sched_getaffinity(0))
    n = threads or min(avail, 32)
    torch.set_num_threads(n)
    sample = 64
    tsd = R.as_torch_sd(sd)
    pts = torch.from_numpy(pts_np[:sample]).transpose(2, 1)
    init = torch.from_numpy(init_np[:sample])
    with torch.no_grad():
        np.random.seed(0)
        R.decode_static(R.static_one_forward(tsd, pts, init), init, False)        # warm-up
        t0 = time.perf_counter()
        it = 0
        while it < 3 or (time.perf_counter() - t0 < budget_s and it < 50):
            R.decode_static(R.static_one_forward(tsd, pts, init), init, False)
            it += 1
        dt = (time.perf_counter() - t0) / it
    return {"value": round(sample / dt, 2), "unit": "object-crops/s", "cores": n, "cores_available": avail,
            "kind": "port",
            "sample": f"oracle/ref_heads.py static_one_forward+decode, {it} x (B={sample}, N={pts.shape[2]}) fp32, "
                      f"torch {torch.__version__} CPU kernels, {n} of {avail} host threads (the port saturates there)"}


def torch_gpu_baseline(model, inputs, sample=256, iters=3):
    """The reference's own formulation on this GPU: stock PyTorch-ROCm ops (Conv1d/BatchNorm1d/Linear/max through
    MIOpen / rocBLAS, the per-sample NumPy gather loop with its device->host syncs, materialised repeat+cat) — the
    eval-mode run of the train-mode composite in 3dal_pytorch_amd/static_model.py, which mirrors
    tools/static_model.py:117-146 op for op — plus an on-device decode. What a user gets from the reference
    unchanged on an MI355X; reported beside the HIP path, never as `value`."""
    pts, init, _ = inputs
    pts, init = pts[:sample], init[:sample]
    mean = torch.tensor(arch.MEAN_SIZE, device=pts.device)

    def run():
        with torch.no_grad():
            o = sm._train_forward_one(model, pts, init)
            hc, sc = o["heading_scores"].argmax(1), o["size_scores"].argmax(1)
            ar = torch.arange(pts.shape[0], device=pts.device)
            ang = hc.float() * (2 * np.pi / 12) + o["heading_residuals"][ar, hc]
            ang = torch.where(ang > np.pi, ang - 2 * np.pi, ang) + init[:, -1]
            return torch.cat([o["center"], mean[sc] + o["size_residuals"][ar, sc], ang[:, None]], 1)
    saved = (model.train_backend, model.sampler)
    model.train_backend, model.sampler = "torch", "numpy"            # stock ops and the reference's host sampling loop
    try:
        np.random.seed(0)
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
    finally:
        model.train_backend, model.sampler = saved
    return {"value": round(pts.shape[0] / dt, 1), "unit": "object-crops/s", "kind": "port",
            "sample": f"stock PyTorch-ROCm ops (torch {torch.__version__}), reference formulation incl. the host gather "
                      f"loop, {iters} x (B={pts.shape[0]}, N={pts.shape[2]}) fp32 on the same GPU"}


def committed_profile(kernel, precision, B, N, tag=""):
    """HBM bytes per launch and PMC ratios from the COMMITTED rocprofv3 passes (profiles/*.json, written by
    tools/prof_summary.py on an earlier run of this very command) — only when that profile was taken at this
    precision and shape; always labelled with its file, never presented as measured in this run."""
    out = {"traffic": None, "pmc": None}
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tfile):
        t = json.load(open(tfile))
        shape = t.get("_shape", {"precisions": ["fp32"], "B": 4096, "N": 1024})
        if precision in shape.get("precisions", []) and (shape.get("B"), shape.get("N")) == (B, N) and kernel in t:
            out["traffic"] = t[kernel]
            out["traffic_source"] = {"file": "profiles/traffic.json", "taken": t.get("_taken", "round 1"),
                                     "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, "
                                            "FETCH_SIZE x 2 per MI355X_MICROARCH.md, per launch; not measured in this run"}
    import glob
    # tag "": the passes of the default command (fp32 and bf16 at 4096 x 1024); "_c3" / "_c5": passes of --config C3 / C5
    pfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_pmc{tag}.json")))
    pshape = (t if os.path.exists(tfile) else {}).get("_shape", {"precisions": ["fp32"], "B": 4096, "N": 1024})
    if pfiles and (tag or (precision in pshape.get("precisions", []) and (pshape.get("B"), pshape.get("N")) == (B, N))):
        pfile = pfiles[-1]                                          # the newest round's passes (tools/profile_round.sh)
        d = json.load(open(pfile)).get(kernel.split("[")[0], {})
        if "GRBM_GUI_ACTIVE" in d and "SQ_VALU_MFMA_BUSY_CYCLES" in d:
            cyc = d["GRBM_GUI_ACTIVE"] / 8.0                        # the counter sums the 8 XCDs
            out["pmc"] = {"source": f"profiles/{os.path.basename(pfile)} (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES "
                                    "GRBM_GUI_ACTIVE in a pass of its own; an earlier run of this command, not this run)",
                          "mfma_busy": round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), 4),
                          "clock_ghz": round(cyc / d["avg_ns_under_GRBM_GUI_ACTIVE"], 3)}
    return out


def roofline_of(kr, peak, precision, B, N, tag=""):
    """the `roofline` object for the dominant MFMA kernel of a kernel table: `achieved` = ALGORITHMIC FLOP per launch
     / the launch's average duration measured live (HIP events on the launch stream); the executed
    rate beside it; HBM traffic and PMC ratios from the committed rocprofv3 passes when they were taken at this shape"""
    dom = max((k for k in kr if "frac_algorithmic" in kr[k]), key=lambda k: kr[k]["ms"])
    prof = committed_profile(dom, precision, B, N, tag)
    r = {"kernel": dom, "bound": "mfma", "achieved": kr[dom]["tflops_algorithmic"], "peak": peak, "unit": "TFLOP/s",
         "frac": kr[dom]["frac_algorithmic"], "traffic": prof["traffic"], "ms_per_launch": kr[dom]["ms"],
         "algorithmic_gflop_per_launch": kr[dom]["algorithmic_gflop"],
         "executed_gflop_per_launch": kr[dom]["executed_gflop"], "frac_executed": kr[dom]["frac_executed"],
         "pmc": prof["pmc"]}
    if "traffic_source" in prof:
        r["traffic_source"] = prof["traffic_source"]
    return r


def executed_gflop_per_step(kr, static, B):
    """every kernel's executed GFLOP + the per-item FC tails (which have no row of their own)"""
    return sum(kr[k]["executed_gflop"] for k in kr) + 2.0 * B * sum(
        ci * co for t in ([arch.STATIC_BOX_EST] if static else [arch.POINT_EMB, arch.BOX_EMB, arch.DYNAMIC_BOX_EST])
        for _, _, ci, co in t["fcs"]) / 1e9


# ---------------------------------------------------------------------------------------- workloads
class Part:
    """one head of a workload on this rank: `run()` -> this rank's (n_local,7) boxes; `shard(rank)` -> (lo, n) of any
    rank's contiguous range of the head's n_total items; `inputs_for(first, count)` -> refine() arguments of global
    items [first, first + count) (what gather_self_check recomputes a peer's rows from)"""

    def __init__(self, name, model, inputs, n_local, n_total, shard, inputs_for):
        self.name, self.model, self.inputs = name, model, inputs
        self.n_local, self.n_total, self.shard, self.inputs_for = n_local, n_total, shard, inputs_for

    def run(self):
        return self.model.refine(*self.inputs)[: self.n_local]

    def __iter__(self):                                    # (fn, n_local, n_total), the shape older call sites unpack
        return iter((self.run, self.n_local, self.n_total))


class Workload:
    """what one rank does per step: `parts` = one Part per head"""

    def __init__(self):
        self.parts = []
        self.gatherers = None


def c4_segment_sizes():
    """198 frames; 64 static tracks -> 64 crops at N=4096; 40 dynamic tracks with lengths
    rng.integers(20,199) -> one item per track-frame"""
    lens = np.random.default_rng(10922081).integers(20, 199, size=40)
    return 64, int(lens.sum())


def workload_shards(args, world):
    """[(head, items of the whole job, rank -> (first item, count))] of the configured workload: the ONE place a bench
    workload's sharding is written. build_workload (the GPU run) and plumbing_only (the CPU rehearsal of the N > 1 path,
    world 8 in tests/test_launch_cpu.py) both read it. C4 is a fixed segment split in contiguous index ranges (strong
    scaling, ragged last rank); every other config gives each rank its own B items (weak scaling)."""
    def span(n):
        return lambda r: (lambda lo, hi: (lo, hi - lo))(*dal3_dist.shard_range(n, r, world))
    if args.config == "C4":
        n_static, n_dyn = c4_segment_sizes()
        return [("static", n_static, span(n_static)), ("dynamic", n_dyn, span(n_dyn))]
    static = args.head == "static"
    B = args.batch or (4096 if static else 1024)
    return [("static" if static else "dynamic", B * world, lambda r: (r * B, B))]


def build_workload(args, dev, rank, world):
    wl = Workload()
    prec = args.precision
    if args.config == "C4":
        # contiguous index sharding, static and dynamic batches back to back, one all-gather per head. The segment is
        # fixed: strong scaling.
        (_, n_static, s_span), (_, n_dyn, d_span) = workload_shards(args, world)
        (s_lo, s_n), (d_lo, d_n) = s_span(rank), d_span(rank)
        s_hi, d_hi = s_lo + s_n, d_lo + d_n
        smodel, sin, _ = make_static(max(s_hi - s_lo, 1), 4096, dev, s_lo, prec)
        dmodel, din = make_dynamic(max(d_hi - d_lo, 1), dev, d_lo, prec)
        wl.parts = [Part("static", smodel, sin, s_hi - s_lo, n_static, s_span,
                         lambda first, count: static_inputs(first, count, 4096, dev, prec)),
                    Part("dynamic", dmodel, din, d_hi - d_lo, n_dyn, d_span,
                         lambda first, count: dynamic_inputs(first, count, 1024, dev, prec))]
        wl.model, wl.inputs, wl.host, wl.static = smodel, None, None, False
        wl.B, wl.N = (s_hi - s_lo) + (d_hi - d_lo), 0
        wl.n_total = n_static + n_dyn
        wl.flop_item = (n_static * arch.static_one_flop(4096) + n_dyn * arch.dynamic_flop(5120)) / wl.n_total
        wl.scaling = "strong"
        wl.desc = (f"one synthetic segment: {n_dyn} dynamic items (40 tracks) x 5120 pts + 64 static crops x 4096 pts, "
                   f"{prec}, both heads back to back (BASELINE.json configs[3])")
        return wl
    static = args.head == "static"
    B = args.batch or (4096 if static else 1024)
    N = args.points if static else 5 * args.points
    first = rank * B                                            # weak scaling: B items per GPU
    two = static and getattr(args, "two_stage", False)
    if static:
        model, inputs, host = make_static(B, N, dev, first, prec, two=two)
        flop_item = arch.static_two_flop(N) if two else arch.static_one_flop(N)
        desc = f"StaticModel{'Two' if two else 'One'}BoxEst forward+decode, {B} crops x {N} pts per GPU, {prec}" + \
            (" (BASELINE.json configs[1])" if (B, N, prec) == (4096, 1024, "fp32") else
             " (BASELINE.json configs[4] shape)" if (N, prec) == (4096, "fp16") else "")
    else:
        model, inputs = make_dynamic(B, dev, first, prec, args.points)
        host = None
        flop_item = arch.dynamic_flop(N)
        desc = (f"DynamicModel forward+decode, {B} items x {N} pts + 101 boxes per GPU, {prec} arithmetic"
                + (" (BASELINE.json configs[2])" if (B, N, prec) == (1024, 5120, "bf16") else ""))
    wl.parts = [Part("static" if static else "dynamic", model, inputs, B, B * world, workload_shards(args, world)[0][2],
                     (lambda first, count: static_inputs(first, count, N, dev, prec)) if static else
                     (lambda first, count: dynamic_inputs(first, count, args.points, dev, prec)))]
    wl.model, wl.inputs, wl.host, wl.static = model, inputs, host, static
    wl.B, wl.N, wl.n_total, wl.flop_item, wl.scaling, wl.desc = B, N, B * world, flop_item, "weak", desc
    return wl


def time_steps(wl, dev, steps, warmup, use_dist, overlap=True):
    """W untimed + exactly K timed steps between two fences (barrier + device synchronize); returns (seconds,
    per-step event times, last complete result). With a process group every step ends in one all-gather per head;
    with `overlap` it is collected one step later, and the last one before the closing fence."""
    if wl.gatherers is None:
        wl.gatherers = [dal3_dist.BoxGatherer(n_total, dev) for _, _, n_total in wl.parts]

    def fence():
        if use_dist:
            torch.distributed.barrier()
        torch.cuda.synchronize()
    last = [None] * len(wl.parts)
    pipe = getattr(wl, "pipe", None)

    def step(final=False):
        if pipe is not None:                                # consecutive steps on alternating streams (one head, no ranks)
            pipe.submit(*wl.inputs)
            for got in pipe.collect(keep=0 if final else len(pipe.streams) - 1):
                last[0] = got
            return
        for i, (fn, _, _) in enumerate(wl.parts):
            wl.gatherers[i].submit(fn())
            got = wl.gatherers[i].collect(keep=1 if (overlap and not final) else 0)
            if got is not None:
                last[i] = got
    gc_was = gc.isenabled()
    if os.environ.get("DAL3_BENCH_GC") != "1":              # (=1: leave the collector on, to reproduce the stall)
        # As timeit does: no cyclic-GC pass inside the timed region. A generation-2 pass over the process's objects is a
        # 40 ms host stall; where it falls depends on allocation counts (even on the script's path), and when it falls on
        # the first timed step — the queue is empty right behind the fence — the GPU waits for it: 20 steps read 28.7 ms
        # per step instead of 26.3 (median and minimum unaffected). Collected BEFORE the warm-up: 40 ms of idling is
        # enough for the chip to drop its clocks, and the 16-bit steps take five steps to get them back.
        # DAL3_BENCH_DEBUG=1 prints the host/event timeline.
        gc.collect()
        gc.disable()
    for _ in range(warmup):
        step(final=True)
    fence()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    host = []
    for i in range(steps):
        step(final=(i == steps - 1))
        marks[i + 1].record()
        host.append(time.perf_counter() - t0)
    t_issued = time.perf_counter() - t0
    fence()
    dt = time.perf_counter() - t0
    if gc_was:
        gc.enable()
    if os.environ.get("DAL3_BENCH_DEBUG") == "1":
        ev = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
        sys.stderr.write(f"[time_steps] issued by {t_issued * 1e3:.1f} ms, fence returned at {dt * 1e3:.1f} ms, events sum "
                         f"{sum(ev):.1f} ms, host per step (ms) {[round(h * 1e3, 1) for h in host]}, events {[round(e, 1) for e in ev]}\n")
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
    boxes = torch.cat(last)
    assert boxes.shape == (wl.n_total, 7) and bool(torch.isfinite(boxes).all())
    wl.last_boxes = last                                    # per head: the (n_total,7) boxes of the last timed step
    wl.rank_seconds = [dt]
    if use_dist:                                            # the job's time is the slowest rank's; every rank's is kept
        wl.rank_seconds = dal3_dist.gather_scalars(dt, dev)
        dt = max(wl.rank_seconds)
    return dt, per_step, boxes


def gather_self_check(wl, dev, rank, world, rows=64):
    """SURVEY.md 8(e) determinism check, inside the bench run: rank 0 recomputes the first `rows` items of OTHER
    ranks' shards (rank 1 and the last rank) from scratch — the peer's synthetic inputs regenerated from their
    global indices, the replicated weights, the sampler keyed with the peer's item_offset — and compares them with
    the rows the all-gather delivered, bit for bit. One entry per head and peer; `equal` is their conjunction."""
    if world < 2:
        return None
    out = {"rows_per_peer": rows, "checks": []}
    if rank == 0:
        for part, gathered in zip(wl.parts, wl.last_boxes):
            for peer in sorted({1, world - 1}):
                lo, n = part.shard(peer)
                n = min(n, rows)
                if n <= 0:
                    out["checks"].append({"head": part.name, "peer": peer, "rows": 0, "equal": None})
                    continue
                saved = part.model.item_offset
                part.model.item_offset = lo
                try:
                    mine = part.model.refine(*part.inputs_for(lo, n))[:n]
                finally:
                    part.model.item_offset = saved
                theirs = gathered[lo:lo + n]
                eq = bool(torch.equal(mine, theirs))
                c = {"head": part.name, "peer": peer, "first_item": lo, "rows": n, "equal": eq}
                if not eq:
                    c["max_abs_diff"] = float((mine - theirs).abs().max())
                out["checks"].append(c)
        done = [c["equal"] for c in out["checks"] if c["equal"] is not None]
        out["equal"] = bool(done) and all(done)
    return out


def gather_latency(wl, dev, iters=50):
    """the all-gather on its own: HIP events around `iters` synchronous gathers of this rank's boxes (per head)"""
    outs = []
    for (fn, n_local, n_total) in wl.parts:
        local = torch.zeros((n_local, 7), device=dev)
        outs.append(events_ms(lambda: dal3_dist.all_gather_boxes(local, n_total), iters, warmup=5))
    return [round(t * 1e3, 1) for t in outs]


def accuracy_vs_fp32_path(model, inputs, prec):
    """What `prec` costs on THIS input, next to its throughput: the same launch in the exact-fp32 arithmetic
    (1e-6 from the reference's PyTorch-CPU forward, tests/test_gpu_parity.py) is the yardstick. Three runs of the whole
    path: fp32; `prec` free-running (its own mask, its own draws); `prec` with the fp32 run's mask forced
    (mask_override: the device sampler, keyed on the item and the count, then draws the very same points), which
    isolates the box estimator's error from the discrete effect of a flipped point. Box error per parameter group —
    centre (m, absolute), size (relative to the largest size), yaw (rad, absolute) — on the crops whose decoded classes
    agree, with their count (box_err)."""
    keep = model.precision
    with torch.no_grad():
        model.precision = "fp32"
        ref = model._run(*inputs)
        model.precision = prec
        got = model._run(*inputs)
        forced = model._run(*inputs, mask_override=ref["mask"])
    model.precision = keep
    assert torch.equal(forced["obj_idx"], ref["obj_idx"])
    B = ref["mask"].shape[0]
    same = (ref["mask"] == got["mask"]).all(1)
    flipped = int((ref["mask"] != got["mask"]).sum())
    bp = "bp1" if "bp1" in ref else "bp"

    def classes(o):
        return o[bp][:, 3:15].argmax(1), o[bp][:, 27:30].argmax(1)

    def box_err(o):
        """boxes of run `o` against the fp32 run's: on the crops whose heading AND size classes agree (the decoded box is
        continuous in the 39 parameters there), and how many do — a flipped class is a different bin centre / mean size,
        i.e. a discrete event like a flipped mask bit, counted, not averaged"""
        (h0, s0), (h1, s1) = classes(ref), classes(o)
        same = (h0 == h1) & (s0 == s1)
        a, b = o["boxes7"][same].double(), ref["boxes7"][same].double()
        d = (a - b).abs()
        return {"crops_with_the_same_heading_and_size_class": int(same.sum()), "crops": int(same.numel()),
                "on_those": {"centre_m_max_abs": round(d[:, :3].max().item(), 6),
                             "centre_m_median_abs": round(d[:, :3].max(1).values.median().item(), 6),
                             "size_max_rel": round((d[:, 3:6].max() / b[:, 3:6].abs().max()).item(), 6),
                             "yaw_rad_max_abs": round(d[:, 6].max().item(), 6)}}
    return {"logits_max_rel": round(((ref["logits"] - got["logits"]).abs().max() / ref["logits"].abs().max()).item(), 6),
            "mask_bits_flipped": flipped, "mask_bits": int(ref["mask"].numel()),
            "mask_agreement": round(1.0 - flipped / ref["mask"].numel(), 6),
            "crops_with_identical_mask": int(same.sum()), "crops": B,
            "box_params_max_rel_fp32_mask_forced": round(((ref[bp] - forced[bp]).abs().max() / ref[bp].abs().max()).item(), 6),
            "boxes7_fp32_mask_forced": box_err(forced), "boxes7_free_running": box_err(got)}


def next_rows():
    """SURVEY.md 8(f)'s rows either side of the heads, measured in THIS run: N1 crop preparation from the
    resident StaticTrackStore, N2 crop extraction from full sweeps, N3 write-back of the refined boxes with the segment
    flattened once (post.WritebackPlan). Per row: stream time of the device part (HIP events), the algorithmic bytes it
    moves, GB/s and the fraction of the 8 TB/s HBM roof, and the whole call with its host part. The measuring code is
    tools/bench_prep_post.py and tools/bench_crops.py (`measure()`); none of these rows is bandwidth-bound at a
    segment's size — they are launch- and gather-bound, which is what the fractions say."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    out = {}
    try:
        pp = importlib.import_module("bench_prep_post").measure(1024)
        n1, st = pp["N1_device_batch_of_64"], pp["prepare_static_batch[device, StaticTrackStore, batches of 64]"]
        out["N1"] = {"what": "prepare_static_batch: 64 tracks -> (64,3,4096) crops, from the resident StaticTrackStore",
                     "kernel_ms": n1["stream_ms_per_call"], "algorithmic_bytes": n1["algorithmic_bytes"], "gb_per_s": n1["gb_per_s"],
                     "frac_of_hbm_8TBps": n1["frac_of_8TBps"], "whole_call_ms": st["ms_per_batch_of_64"],
                     "one_time_store_build_ms_1024_tracks": st["store_build_ms"], "note": n1["note"]}
        wb = pp["writeback_static[WritebackPlan]"]
        out["N3"] = {"what": f"writeback of {wb['pairs']} (track, frame) pairs into {wb['detections']} detections of a 198-frame segment",
                     "kernel_ms": wb["stream_ms_per_launch"], "algorithmic_bytes": wb["algorithmic_bytes"], "gb_per_s": wb["gb_per_s"],
                     "frac_of_hbm_8TBps": wb["frac_of_8TBps"], "whole_call_ms": wb["apply_call_ms"],
                     "one_time_plan_build_ms": wb["plan_build_ms"], "one_shot_call_ms": pp["writeback_static"]["call_ms"],
                     "note": wb["note"]}
    except Exception as e:                                  # (a row that cannot run must not take the headline with it)
        out["N1_N3_error"] = repr(e)
    try:
        for order in ("range_image", "shuffled"):
            c = importlib.import_module("bench_crops").measure(order=order)
            out["N2" if order == "range_image" else "N2_shuffled_points"] = {
                "what": "extract_crops: " + c["workload"], "kernel_ms": c["device_ms"],
                "algorithmic_bytes": c["roofline"]["algorithmic_bytes"], "gb_per_s": c["roofline"]["achieved"],
                "frac_of_hbm_8TBps": c["roofline"]["frac"], "whole_call_ms": c["call_ms_with_host_setup"],
                "point_box_tests_per_s_e9": c["point_box_tests_per_s"],
                "note": "VALU-bound (six plane tests per candidate pair behind a sphere cull), not HBM-bound"}
    except Exception as e:
        out["N2_error"] = repr(e)
    return out


def other_config(name, dev, steps):
    """one of BASELINE.json's other configurations on this GPU: whole-path rate (same step definition)"""
    ns = argparse.Namespace(config=name, head="static", precision="fp32", batch=0, points=1024, two_stage=False)
    apply_config(ns)
    wl = build_workload(ns, dev, 0, 1)
    dt, per_step, _ = time_steps(wl, dev, steps, 5, False)
    value = wl.n_total * steps / dt
    peak = MFMA_PEAK_TFLOPS[ns.precision]
    ms = dt / steps * 1e3
    r = {"workload": wl.desc, "value": round(value, 1), "unit": "items/s", "ms_per_step": round(ms, 3),
         "ms_per_step_min": round(per_step[0], 3), "steps": steps, "dtype": DNAME[ns.precision],
         "algorithmic_gflop_per_item": round(wl.flop_item / 1e9, 4),
         "whole_path_tflops_algorithmic": round(value * wl.flop_item / 1e12, 1),
         "whole_path_mfma_frac_algorithmic": round(value * wl.flop_item / 1e12 / peak, 4)}
    if not name.startswith("C4"):
        kr, _ = kernel_table(wl.model, wl.inputs, wl.static, wl.B, wl.N, iters=max(3, min(steps, 5)))
        r["whole_path_mfma_frac_executed"] = round(executed_gflop_per_step(kr, wl.static, wl.B) / ms / peak, 4)
        r["roofline"] = roofline_of(kr, peak, ns.precision, wl.B, wl.N, tag="_" + name.lower() if name in ("C3", "C5") else "")
        if ns.precision in ("bf16", "fp16"):                # a 16-bit rate is half a result without its error on the same input
            r["vs_exact_fp32_path"] = accuracy_vs_fp32_path(wl.model, wl.inputs, ns.precision)
    del wl
    torch.cuda.empty_cache()
    return r


def two_stage(args):
    return bool(getattr(args, "two_stage", False))


def apply_config(args):
    if args.config == "C3":
        args.head, args.precision, args.batch, args.points = "dynamic", "bf16", 1024, 1024
    elif args.config == "C5":
        args.head, args.precision, args.batch, args.points = "static", "fp16", 2048, 4096
    elif args.config == "C2":
        args.head, args.precision, args.batch, args.points = "static", "fp32", 4096, 1024
    elif args.config == "TwoBoxEst":                       # the reference's second static model class, C2's shape
        args.head, args.precision, args.batch, args.points, args.two_stage = "static", "fp32", 4096, 1024, True
    elif args.config == "Dynamic_fp32":                    # the dynamic head in the reference's own arithmetic, C3's shape
        args.head, args.precision, args.batch, args.points = "dynamic", "fp32", 1024, 1024
    elif args.config == "C4_f16x3":                        # the mixed segment on the split-fp16 kernels
        args.config, args.precision = "C4", "f16x3"
    elif args.config == "Dynamic_f16x3":                   # the same, split-fp16 arithmetic (fp32 accuracy: profiles/LEDGER_r01_r03.md 5.4)
        args.head, args.precision, args.batch, args.points = "dynamic", "f16x3", 1024, 1024
    elif args.config == "TwoBoxEst_f16x3":
        args.head, args.precision, args.batch, args.points, args.two_stage = "static", "f16x3", 4096, 1024, True


def plumbing_only(args, rank, world):
    """No GPU work: every rank makes the boxes a refine() of its shard would return (a function of the global item
    index), the launcher / process-group / gather path runs on DAL3_BENCH_BACKEND (gloo on CPU), and rank 0 prints a
    line whose `value` is null — with the fields a real N > 1 line carries (per-rank step times, the replicated
    weight, the self-check of other ranks' rows). Without --config: one 37-item head (ragged over any world size).
    With --config C2 / C3 / C4 / C5: that workload's heads, item counts and rank -> range map (workload_shards, the
    same function the GPU run shards with) — what tests/test_launch_cpu.py drives at world sizes 2 and 8."""
    backend = os.environ.get("DAL3_BENCH_BACKEND", "gloo")
    torch.distributed.init_process_group(backend, rank=rank, world_size=world)
    cpu = torch.device("cpu")
    if args.config is None:
        n = 37
        shards = [("stub", n, lambda r: (lambda lo, hi: (lo, hi - lo))(*dal3_dist.shard_range(n, r, world)))]
        scaling = "strong"
    else:
        apply_config(args)
        shards = workload_shards(args, world)
        scaling = "strong" if args.config == "C4" else "weak"

    class _Stub:                                            # refine() stand-in: boxes = f(global item index, bias)
        item_offset = 0

        def __init__(self, salt):
            self.salt = salt
            self.bias = torch.tensor([1.0 + rank])          # rank-dependent until replicated

        def refine(self, idx):
            return idx[:, None] * 10 + torch.arange(7, dtype=torch.float32)[None] + self.bias + self.salt

    def idx(first, count):
        return (torch.arange(first, first + count, dtype=torch.float32),)
    wl = Workload()
    for k, (name, n_total, span) in enumerate(shards):
        stub = _Stub(100.0 * k)
        dal3_dist.replicate_(stub.bias)
        lo, cnt = span(rank)
        wl.parts.append(Part(name, stub, idx(lo, cnt), cnt, n_total, span, idx))
    gatherers = [dal3_dist.BoxGatherer(p.n_total, cpu) for p in wl.parts]
    got = [None] * len(wl.parts)
    t0 = time.perf_counter()
    for _ in range(3):                                      # the step loop of time_steps: submit, collect one step later
        for i, p in enumerate(wl.parts):
            gatherers[i].submit(p.run())
            r = gatherers[i].collect(keep=1)
            got[i] = r if r is not None else got[i]
    got = [g.collect(keep=0) for g in gatherers]
    wl.last_boxes = got
    rank_ms = [round(t * 1e3, 3) for t in dal3_dist.gather_scalars(time.perf_counter() - t0, cpu)]
    ok = True
    for k, (p, g) in enumerate(zip(wl.parts, got)):
        want = torch.arange(p.n_total, dtype=torch.float32)[:, None] * 10 + torch.arange(7, dtype=torch.float32)[None] + 1.0 + 100.0 * k
        ok = ok and g.shape == (p.n_total, 7) and bool(torch.equal(g, want))
        ok = ok and bool(torch.equal(dal3_dist.all_gather_boxes(p.run(), p.n_total), want))
    check = gather_self_check(wl, cpu, rank, world, rows=5)
    census = dal3_dist.world_census(cpu)
    if os.environ.get("DAL3_BENCH_FAIL_RANK") == str(rank):          # the launcher's failure path, for its test
        sys.exit(3)
    if os.environ.get("DAL3_BENCH_HANG_RANK") == str(rank):          # a rank that never comes back, for the launcher's timeout
        time.sleep(3600)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "object-crops/sec through static+dynamic refinement heads", "value": None,
                          "plumbing_only": True, "n_gpus": world, "gathered_ok": ok, "rccl": census, "scaling": scaling,
                          "config": {"workload": args.config or "stub", "heads": [
                              {"head": p.name, "items": p.n_total, "items_per_rank": [p.shard(r)[1] for r in range(world)]}
                              for p in wl.parts]},
                          "ms_per_step_per_rank": rank_ms, "gather_equals_single_rank": check["equal"] if check else None,
                          "gather_self_check": check}), flush=True)
    sys.exit(0 if ok else 4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)           # >= 20 iterations after 5 warm-ups
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--head", default="static", choices=["static", "dynamic"])
    ap.add_argument("--batch", type=int, default=0, help="items per GPU (default: 4096 static, 1024 dynamic)")
    ap.add_argument("--points", type=int, default=1024)
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16", "fp16", "f16x3"],
                    help="arithmetic of the shared-MLP kernels (fp32 = the reference's; bf16/fp16 = configs C3/C5)")
    ap.add_argument("--no-extras", action="store_true", help="skip roofline / maxpool / configs / cpu_baseline legs")
    ap.add_argument("--only-maxpool", action="store_true", help="run only the standalone max-pool kernel (profiling)")
    ap.add_argument("--maxpool-storage", default="fp32", choices=["fp32", "bf16", "fp16"], help="row storage for --only-maxpool")
    ap.add_argument("--serial-gather", action="store_true", help="wait for each step's all-gather inside the step")
    ap.add_argument("--streams", type=int, default=1,
                    help="run consecutive steps on this many HIP streams (graph.StreamPipe): small batches, one GPU")
    ap.add_argument("--two-stage", action="store_true", help="static head: StaticModelTwoBoxEst instead of OneBoxEst")
    ap.add_argument("--config", default=None, choices=["C2", "C3", "C4", "C5", "TwoBoxEst", "Dynamic_fp32", "TwoBoxEst_f16x3", "Dynamic_f16x3", "C4_f16x3"],
                    help="BASELINE.json configs by name: C2 = the default (static, 4096 x 1024, fp32); C3 = dynamic head, "
                         "1024 items x 5 x 1024 pts, bf16; C4 = one segment (64 static crops x 4096 pts + 40 dynamic tracks), sharded "
                         "over the GPUs (strong scaling); C5 = static, N=4096, 2048 crops per GPU, fp16 MFMA; TwoBoxEst = StaticModelTwoBoxEst "
                         "at C2's shape; Dynamic_fp32 = DynamicModel at C3's shape in fp32")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="launcher + process group + gather only, no GPU work (CPU test of the N > 1 start-up path)")
    ap.add_argument("--cpu-sweep", type=int, nargs="+", default=None, metavar="THREADS",
                    help="run only the cpu_baseline leg at these torch thread counts (no GPU work) and print one JSON line")
    args = ap.parse_args()
    if args.cpu_sweep:
        pts_np, init_np, _ = synth.static_crops(64, args.points)
        host = (pts_np, init_np, synth.state_dict("static_one"))
        print(json.dumps({"cpu_baseline_sweep": [cpu_baseline(host, budget_s=6.0, threads=t) for t in args.cpu_sweep],
                          "affinity": len(os.sched_getaffinity(0))}))
        return

    # ---- N > 1 without a launcher: start the ranks as children; this process never touches the GPU
    share_gpu = os.environ.get("DAL3_BENCH_SHARE_GPU") == "1"      # rehearsal: ranks share devices (1-GPU box)
    if args.gpus > 1 and not launch.under_launcher():
        rc, out = launch.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus, need_gpus=not args.plumbing_only,
                                     share_gpu=share_gpu)
        line = launch.relay_json_line(out)
        if line:
            print(line, flush=True)
        sys.exit(rc if rc else (0 if line else 1))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started {world} ranks")
    if args.plumbing_only:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        plumbing_only(args, rank, world)
    apply_config(args)
    n_dev = torch.cuda.device_count()
    if share_gpu and n_dev > 0:
        local = local % n_dev
    if n_dev <= local:                                      # a rank started by a launcher on a box with too few GPUs: the same
        sys.stderr.write(f"bench.py: needs {max(args.gpus, local + 1)} GPUs, this machine shows {n_dev}\n")   # line and exit code
        sys.exit(2)                                         # as the self-launching parent gives (launch.spawn_ranks)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # DAL3_FORCE_DIST=1 runs the RCCL path even with one rank (exercises init + all-gather on a 1-GPU box)
    use_dist = world > 1 or os.environ.get("DAL3_FORCE_DIST") == "1"
    # stdout carries exactly one JSON line. RCCL prints its version banner to the C-level stdout (and flushes it at
    # exit), so while anything but that line can be written, file descriptor 1 points at stderr.
    real_stdout = os.dup(1)
    sys.stdout.flush()
    os.dup2(2, 1)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # "nccl" = RCCL over xGMI, the product transport. DAL3_BENCH_BACKEND=gloo: the rehearsal transport (boxes
        # staged through pinned host memory, dist.BoxGatherer) for boxes where RCCL cannot connect the ranks, e.g. two
        # ranks on ONE GPU with DAL3_BENCH_SHARE_GPU=1 — launcher, sharding, fences, overlap and self-check as in the real run.
        backend = os.environ.get("DAL3_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            torch.distributed.init_process_group(backend, rank=rank, world_size=world)

    if args.only_maxpool:
        rec = {"maxpool": maxpool_roofline(dev, iters=args.steps, dtype={"fp32": torch.float32, "bf16": torch.bfloat16,
                                                                            "fp16": torch.float16}[args.maxpool_storage])}
        os.write(real_stdout, (json.dumps(rec) + "\n").encode())
        return

    wl = build_workload(args, dev, rank, world)
    if args.streams > 1:
        if use_dist or args.config == "C4":
            sys.exit("bench.py: --streams is for single-GPU, single-head runs")
        wl.pipe = dal3_graph.StreamPipe(wl.model, depth=args.streams)
        wl.desc += f", consecutive steps on {args.streams} HIP streams"
    peak = MFMA_PEAK_TFLOPS[args.precision]
    dt, per_step, _ = time_steps(wl, dev, args.steps, args.warmup, use_dist, overlap=not args.serial_gather)
    ms_per_step = dt / args.steps * 1e3
    value = wl.n_total * args.steps / dt
    mixed = args.config == "C4"
    rec = {
        "metric": "object-crops/sec through static+dynamic refinement heads",
        "value": round(value, 1), "unit": "object-crops/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "ms_per_step_median": round(per_step[len(per_step) // 2], 3), "ms_per_step_min": round(per_step[0], 3),
        "ms_per_step_max": round(per_step[-1], 3),
        "higher_is_better": True,
        "scaling": wl.scaling, "vs_baseline": None, "dtype": DNAME[args.precision], "data": "synthetic",
        "config": {"workload": wl.desc, "items_per_gpu": wl.B, "points_per_item": wl.N, "sampler": wl.model.sampler,
                   "parallelism": (f"object-sharded x{world}, one all-gather of (B,7) boxes" + (" per head" if mixed else "")
                                   + (", collected one step later (overlapped)" if not args.serial_gather else ", serial"))
                   if use_dist else "single GPU",
                   "algorithmic_gflop_per_item": round(wl.flop_item / 1e9, 4)},
        "whole_path_tflops_algorithmic": round(value * wl.flop_item / 1e12, 2),
        "whole_path_mfma_frac_algorithmic": round(value / world * wl.flop_item / 1e12 / peak, 4),
    }
    if use_dist:
        census = dal3_dist.world_census(dev)
        rank_ms = [round(t / args.steps * 1e3, 3) for t in wl.rank_seconds]
        rec["ms_per_step_per_rank"] = rank_ms                   # each rank's own clock between the two fences
        rec["ms_per_step_rank_min"], rec["ms_per_step_rank_max"] = min(rank_ms), max(rank_ms)
        check = gather_self_check(wl, dev, rank, world)
        rec["gather_equals_single_rank"] = check["equal"] if check and rank == 0 else None
        rec["gather_self_check"] = check
        rec["rccl"] = dict(census, allgather_us=gather_latency(wl, dev),
                           message_bytes_per_rank=[((n_total + world - 1) // world) * 28 for _, _, n_total in wl.parts],
                           transport="RCCL on the device buffers" if census["backend"] == "nccl" else
                           f"{census['backend']} (rehearsal): boxes staged through pinned host memory",
                           ranks_share_gpus=share_gpu,
                           rccl_version=".".join(str(v) for v in torch.cuda.nccl.version()))
        rec["rccl_world_size"] = census["world_size"]
        if not args.no_extras:
            # the same steps with the gather waited for inside each step: what the overlap is worth
            dt2, _, _ = time_steps(wl, dev, args.steps, 1, use_dist, overlap=args.serial_gather)
            rec["rccl"]["ms_per_step_" + ("overlapped" if args.serial_gather else "serial_gather")] = round(dt2 / args.steps * 1e3, 3)
    if rank == 0 and not args.no_extras and not mixed:
        # per-kernel table and roofline: on rank 0 at every world size (the other ranks wait at the closing barrier)
        B, N, model, inputs, static = wl.B, wl.N, wl.model, wl.inputs, wl.static
        kr, mean_count = kernel_table(model, inputs, static, B, N, iters=max(3, min(args.steps, 10)))
        rec["roofline"] = roofline_of(kr, peak, args.precision, B, N,
                                      tag="_" + args.config.lower() if args.config in ("C3", "C5") else
                                      "_f16x3" if (args.precision, B, N) == ("f16x3", 4096, 1024) else "")
        rec["kernels"] = kr
        rec["mean_segmented_points_per_item"] = round(mean_count, 1)
        exe = executed_gflop_per_step(kr, static, B)
        rec["executed_gflop_per_step"] = round(exe, 1)
        rec["algorithmic_gflop_per_step"] = round(B * wl.flop_item / 1e9, 1)
        # the headline whole-path fraction: what the silicon executed per rank-0 step over the MFMA peak
        rec["whole_path_mfma_frac_executed"] = round(exe / (wl.rank_seconds[0] / args.steps * 1e3) / peak, 4)   # GFLOP/ms = TFLOP/s
    if rank == 0 and world == 1 and not args.no_extras and not mixed:
        if static and args.precision == "fp32" and not two_stage(args):
            # the same workload on the 16-bit MFMA path (BASELINE.json configs C3/C5 arithmetic): reported beside
            # the fp32 headline, never as `value`
            rec["lowprec"] = {}
            for prec in ("bf16", "fp16"):
                model.precision = prec
                d, _, _ = time_steps(wl, dev, args.steps, 5, False)
                d /= args.steps
                k2, _ = kernel_table(model, inputs, static, B, N, iters=max(3, min(args.steps, 10)))
                rec["lowprec"][prec] = {"value": round(B / d, 1), "unit": "object-crops/s", "ms_per_step": round(d * 1e3, 3),
                                        "whole_path_tflops_algorithmic": round(B / d * wl.flop_item / 1e12, 1),
                                        "whole_path_mfma_frac_executed": round(
                                            executed_gflop_per_step(k2, static, B) / (d * 1e3) / MFMA_PEAK_TFLOPS[prec], 4),
                                        "roofline": roofline_of(k2, MFMA_PEAK_TFLOPS[prec], prec, B, N),
                                        "vs_exact_fp32_path": accuracy_vs_fp32_path(model, inputs, prec),
                                        "kernels": k2}
            # the fp32 formulation on the fp16 MFMA: every operand as an (hi, lo) fp16 pair, three MFMAs per product,
            # fp32 accumulate (profiles/LEDGER_r01_r03.md 5.4). Its distance from the exact-fp32 path is measured here on this very
            # input, beside its step; `value` stays the exact-fp32 path's.
            with torch.no_grad():
                model.precision = args.precision
                ref = model(*inputs)
                model.precision = "f16x3"
                got = model(*inputs)
            d, _, _ = time_steps(wl, dev, args.steps, 5, False)
            d /= args.steps
            k3, _ = kernel_table(model, inputs, static, B, N, iters=max(3, min(args.steps, 10)))
            same = (ref["mask"] == got["mask"]).all(1)
            lg = (ref["logits"] - got["logits"]).abs().max().item() / ref["logits"].abs().max().item()
            bx = {k: round(((ref[k] - got[k])[same].abs().max() / ref[k].abs().max()).item(), 9)
                  for k in ref if k not in ("logits", "mask") and torch.is_tensor(ref[k]) and ref[k].is_floating_point()
                  and ref[k].shape[0] == B}
            rec["f16x3"] = {"value": round(B / d, 1), "unit": "object-crops/s", "ms_per_step": round(d * 1e3, 3),
                            "dtype": DNAME["f16x3"],
                            "vs_exact_fp32_path": {
                                "logits_max_rel": round(lg, 9), "mask_bits_flipped": int((ref["mask"] != got["mask"]).sum()),
                                "mask_bits": int(ref["mask"].numel()), "crops_with_identical_mask": int(same.sum()),
                                "outputs_max_rel_on_those": bx,
                                "note": "the exact-fp32 path itself is 1e-6 from the reference's PyTorch-CPU forward; parity tests: "
                                        "tests/test_gpu_x3.py (the reference's golden vectors at the fp32 tolerance 1e-4)"},
                            "whole_path_tflops_algorithmic": round(B / d * wl.flop_item / 1e12, 1),
                            "x_fp32_mfma_peak": round(B / d * wl.flop_item / 1e12 / MFMA_PEAK_TFLOPS["fp32"], 3),
                            "whole_path_mfma_frac_executed": round(
                                executed_gflop_per_step(k3, static, B) / (d * 1e3) / MFMA_PEAK_TFLOPS["f16x3"], 4),
                            "roofline": roofline_of(k3, MFMA_PEAK_TFLOPS["f16x3"], "f16x3", B, N, tag="_f16x3"),
                            "kernels": k3}
            # the same three numbers at the top level of the line, next to `value` (which stays the exact-fp32 path's)
            rec["f16x3_value"], rec["f16x3_ms_per_step"] = rec["f16x3"]["value"], rec["f16x3"]["ms_per_step"]
            rec["f16x3_logits_max_rel_vs_exact_fp32_path"] = rec["f16x3"]["vs_exact_fp32_path"]["logits_max_rel"]
            model.precision = args.precision
        rec["maxpool"] = maxpool_roofline(dev, iters=5)
        rec["maxpool_bf16"] = maxpool_roofline(dev, iters=5, dtype=torch.bfloat16)     # the same rows in 2-byte storage (C3 / C5)
        if static and args.precision == "fp32" and (B, N) == (4096, 1024) and not two_stage(args):
            rec["torch_gpu_baseline"] = torch_gpu_baseline(model, inputs)
            rec["cpu_baseline"] = cpu_baseline(wl.host)
            del wl, model, inputs
            torch.cuda.empty_cache()
            # BASELINE.json's other configurations and the reference's other two model classes in its own arithmetic,
            # driver-timed in the same run (the metric is "static+dynamic heads")
            rec["next_rows"] = next_rows()
            rec["configs"] = {"C2": "this line's `value`"}
            for name, st in (("C3", 10), ("C5", 10), ("TwoBoxEst", 5), ("TwoBoxEst_f16x3", 5), ("Dynamic_fp32", 5),
                                 ("Dynamic_f16x3", 5), ("C4", 3), ("C4_f16x3", 3)):
                rec["configs"][name] = other_config(name, dev, st)
        elif static:
            rec["cpu_baseline"] = cpu_baseline(wl.host)
    if use_dist:
        torch.distributed.barrier()
        torch.cuda.synchronize()
        torch.distributed.destroy_process_group()
    sys.stdout.flush()
    C.CDLL(None).fflush(None)                                   # whatever C stdio still holds goes to stderr too
    if rank == 0:
        os.write(real_stdout, (json.dumps(rec) + "\n").encode())
    os.close(real_stdout)


if __name__ == "__main__":
    main()
