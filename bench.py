#!/usr/bin/env python3
"""bench.py — object-crops/sec through the refinement heads on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--head static|dynamic]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (config.workload): BASELINE.json configs[1] — StaticModelOneBoxEst, 4096 crops x 1024
points, fp32, per GPU, synthetic crops and random-init weights from 3dal_pytorch_amd/synth.py
(no dataset / checkpoint is reachable). One step = one pass of the hot path over the batch:
ins_seg -> mask -> object-point sampling -> box estimator -> decode to (B,7) boxes, inputs
already resident in HBM; with N > 1 every rank refines its own 4096 crops (weak scaling, crops
are independent) and one RCCL all-gather of the (N*4096, 7) boxes closes the step.

The JSON line also carries
  roofline      the dominant kernel (an fp32-MFMA shared-MLP kernel) timed live with HIP events
                on the launch stream: algorithmic FLOP per launch / average duration vs the
                157.3 TFLOP/s f32 MFMA peak of gfx950
  maxpool       the standalone N-axis max-pool kernel on (4096,1024,1024) fp32 vs 8 TB/s HBM
  cpu_baseline  the oracle (reference-formulation torch-CPU port) on this box's host cores, on
                a bounded sample of the same workload
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC: what RCCL needs between ranks on this driver

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

F32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, exact f32
# dense peaks per arithmetic dtype (MI355X_MICROARCH.md "Chip-level parameters"; bf16/fp16 ~2.5 PF dense)
MFMA_PEAK_TFLOPS = {"fp32": 157.3, "bf16": 2500.0, "fp16": 2500.0}
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E spec


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


def make_static(B, N, dev, first):
    pts_np, init_np, gt_np = synth.static_crops(B, N, first=first)
    model = sm.StaticModelOneBoxEst()
    sd = synth.state_dict("static_one")
    model.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    model = model.to(dev).eval()
    pts = torch.from_numpy(pts_np).to(dev).transpose(2, 1)          # the callers' layout (static_eval.py:265)
    init, gt = torch.from_numpy(init_np).to(dev), torch.from_numpy(gt_np).to(dev)
    # re-centre the segmentation bias so that about half the points are segmented (synth.py);
    # done with one full-size pass so that every profiled launch of a kernel has the same shape
    with torch.no_grad():
        lg = model(pts, init, gt)["logits"]
        model.ins_seg.dconv5.bias[1] -= (lg[:, :, 1] - lg[:, :, 0]).mean()
        del lg
    model.item_offset = first
    return model, (pts, init, gt), (pts_np, init_np, sd)


def make_dynamic(B, dev, first):
    pts_np, box_np, init8_np, gt_np = synth.dynamic_items(B, first=first)
    model = dm.DynamicModel()
    model.load_state_dict({k: torch.as_tensor(v) for k, v in synth.state_dict("dynamic").items()})
    model = model.to(dev).eval()
    pts = torch.from_numpy(pts_np).to(dev).transpose(2, 1)
    box = torch.from_numpy(box_np).to(dev).transpose(2, 1)
    init8 = torch.from_numpy(init8_np).to(dev)
    with torch.no_grad():
        lg = model(pts, box, None)["logits"]
        model.ins_seg.dconv5.bias[1] -= (lg[:, :, 1] - lg[:, :, 0]).mean()
        del lg
    model.item_offset = first
    return model, (pts, box, init8)


def kernel_rooflines(model, pts, c_in, B, N, iters):
    """per-kernel HIP-event timing of the two shared-MLP kernels of ins_seg (per-kernel C-ABI entries)"""
    lib = hip.lib()
    dt = hip.DTYPES[model.precision]
    peak = MFMA_PEAK_TFLOPS[model.precision]
    w = model._cache.get("ins_seg", model.ins_seg, hip.HEAD_INS_SEG, dt)
    g = torch.zeros((B, 1024), device=pts.device)
    gb = torch.empty((B, 512), device=pts.device)
    logits = torch.empty((B, N, 2), device=pts.device)
    mask = torch.empty((B, N), dtype=torch.uint8, device=pts.device)
    x = hip.bcn(pts)

    def enc():
        hip.check(lib.dal3_ins_seg_encode(hip.ptr(w), dt, c_in, x, B, N, hip.ptr(g), hip.stream()))

    def dec():
        hip.check(lib.dal3_ins_seg_decode(hip.ptr(w), dt, c_in, x, B, N, hip.ptr(gb), hip.ptr(logits), hip.ptr(mask),
                                          hip.stream()))
    enc()
    hip.check(lib.dal3_ins_seg_global_bias(hip.ptr(w), dt, hip.ptr(g), B, hip.ptr(gb), hip.stream()))
    t_enc = events_ms(enc, iters)
    t_dec = events_ms(dec, iters)
    mac_enc = c_in * 64 + 64 * 64 * 2 + 64 * 128 + 128 * 1024
    mac_dec = 64 * 512 + 512 * 256 + 256 * 128 + 128 * 128 + 128 * 2
    out = {}
    for name, t, mac in (("ins_seg_encode_kernel", t_enc, mac_enc), ("ins_seg_decode_kernel", t_dec, mac_dec)):
        tf = 2.0 * mac * B * N / (t * 1e-3) / 1e12
        out[name] = {"ms": round(t, 4), "algorithmic_gflop": round(2.0 * mac * B * N / 1e9, 2),
                     "tflops": round(tf, 2), "frac": round(tf / peak, 4)}
    return out


def maxpool_roofline(dev, iters):
    rows, n = 4096 * 1024, 1024
    try:
        x = torch.empty((rows, n), device=dev)
    except RuntimeError:
        rows = 1024 * 1024
        x = torch.empty((rows, n), device=dev)
    x.normal_()
    out = torch.empty(rows, device=dev)
    lib = hip.lib()
    t = events_ms(lambda: hip.check(lib.dal3_maxpool_n(hip.ptr(x), rows, n, hip.ptr(out), hip.stream())), iters)
    nbytes = rows * n * 4 + rows * 4
    gbs = nbytes / (t * 1e-3) / 1e9
    ok = bool(torch.equal(out[:4096], x[:4096].max(1)[0]))
    del x
    return {"kernel": "maxpool_rows_kernel", "shape": [rows // 1024, 1024, n], "bound": "hbm", "ms": round(t, 4),
            "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
            "exact": ok}


def cpu_baseline(host, budget_s=20.0, threads=None):
    """the oracle (reference formulation) on the host cores, B=64 sample of the same crops"""
    R = importlib.import_module("oracle.ref_heads")
    pts_np, init_np, sd = host
    # `bench.py --cpu-sweep 8 16 32 64 128` on the GPU box (256 logical cores): 8 thr 64, 16 thr 76, 32 thr 80,
    # 64 thr 53, 128 thr 27 crops/s -> the port saturates at 32 threads; more only adds contention
    n = threads or min(len(os.sched_getaffinity(0)), 32)
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
    return {"value": round(sample / dt, 2), "unit": "object-crops/s", "cores": n, "kind": "port",
            "sample": f"oracle/ref_heads.py static_one_forward+decode, {it} x (B={sample}, N={pts.shape[2]}) fp32, "
                      f"torch {torch.__version__} CPU kernels, {n} threads"}


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)           # >= 20 iterations after 5 warm-ups
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--head", default="static", choices=["static", "dynamic"])
    ap.add_argument("--batch", type=int, default=0, help="items per GPU (default: 4096 static, 1024 dynamic)")
    ap.add_argument("--points", type=int, default=1024)
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16", "fp16"],
                    help="arithmetic of the shared-MLP kernels (fp32 = the reference's; bf16/fp16 = configs C3/C5)")
    ap.add_argument("--no-extras", action="store_true", help="skip roofline / maxpool / cpu_baseline legs")
    ap.add_argument("--config", default=None, choices=["C2", "C3", "C4", "C5"],
                    help="BASELINE.json configs by name: C2 = the default (static, 4096 x 1024, fp32); C3 = dynamic head, "
                         "1024 items x 5 x 1024 pts, bf16; C4 = one segment (64 static crops x 4096 pts + 40 dynamic tracks), sharded "
                         "over the GPUs (strong scaling); C5 = static, N=4096, 2048 crops per GPU, fp16 MFMA")
    ap.add_argument("--cpu-sweep", type=int, nargs="+", default=None, metavar="THREADS",
                    help="run only the cpu_baseline leg at these torch thread counts (no GPU work) and print one JSON line")
    args = ap.parse_args()
    if args.cpu_sweep:
        pts_np, init_np, _ = synth.static_crops(64, args.points)
        host = (pts_np, init_np, synth.state_dict("static_one"))
        print(json.dumps({"cpu_baseline_sweep": [cpu_baseline(host, budget_s=6.0, threads=t) for t in args.cpu_sweep],
                          "affinity": len(os.sched_getaffinity(0))}))
        return
    if args.config == "C3":
        args.head, args.precision, args.batch, args.points = "dynamic", "bf16", 1024, 1024
    elif args.config == "C5":
        args.head, args.precision, args.batch, args.points = "static", "fp16", 2048, 4096

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
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
        torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    static = args.head == "static"
    mixed = args.config == "C4"
    if mixed:
        # 198 frames; 64 static tracks -> 64 crops at N=4096; 40 dynamic tracks with lengths
        # rng.integers(20,199) -> one item per track-frame; contiguous index sharding, static and dynamic batches
        # back to back, one all-gather per head. The segment is fixed: strong scaling.
        lens = np.random.default_rng(10922081).integers(20, 199, size=40)
        n_static, n_dyn = 64, int(lens.sum())
        s_lo, s_hi = dal3_dist.shard_range(n_static, rank, world)
        d_lo, d_hi = dal3_dist.shard_range(n_dyn, rank, world)
        smodel, sin, _ = make_static(max(s_hi - s_lo, 1), 4096, dev, s_lo)
        dmodel, din = make_dynamic(max(d_hi - d_lo, 1), dev, d_lo)
        smodel.precision = dmodel.precision = args.precision
        B, N, static, host, model = (s_hi - s_lo) + (d_hi - d_lo), 0, False, None, smodel
        flop_item = (n_static * arch.static_one_flop(4096) + n_dyn * arch.dynamic_flop(5120)) / (n_static + n_dyn)
        n_total = n_static + n_dyn

        def step_fn():
            a = smodel.refine(*sin)[:s_hi - s_lo]
            a = dal3_dist.all_gather_boxes(a, n_static) if use_dist else a
            b = dmodel.refine(*din)[:d_hi - d_lo]
            b = dal3_dist.all_gather_boxes(b, n_dyn) if use_dist else b
            return torch.cat([a, b])
    else:
        B = args.batch or (4096 if static else 1024)
        N = args.points if static else 5 * args.points
        first = rank * B                                            # weak scaling: B items per GPU
        if static:
            model, inputs, host = make_static(B, N, dev, first)
            step_fn = lambda: model.refine(*inputs)                 # noqa: E731
            flop_item = arch.static_one_flop(N)
        else:
            model, inputs = make_dynamic(B, dev, first)
            step_fn = lambda: model.refine(*inputs)                 # noqa: E731
            flop_item = arch.dynamic_flop(N)
            host = None
        n_total = B * world
        model.precision = args.precision
    peak = MFMA_PEAK_TFLOPS[args.precision]
    dname = {"fp32": "f32", "bf16": "bf16", "fp16": "f16"}[args.precision]

    def step():
        boxes = step_fn()
        return dal3_dist.all_gather_boxes(boxes, n_total) if (use_dist and not mixed) else boxes

    def fence():
        if use_dist:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # per-step HIP events on the launch stream ride along (median / min); `value` comes from the
    # wall clock around all K steps, fenced on both sides
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        boxes = step()
        marks[i + 1].record()
    fence()
    dt = time.perf_counter() - t0
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    assert boxes.shape == (n_total, 7) and bool(torch.isfinite(boxes).all())
    if use_dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    ms_per_step = dt / args.steps * 1e3
    value = n_total * args.steps / dt
    rec = {
        "metric": "object-crops/sec through static+dynamic refinement heads",
        "value": round(value, 1), "unit": "object-crops/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "ms_per_step_median": round(per_step[len(per_step) // 2], 3), "ms_per_step_min": round(per_step[0], 3),
        "higher_is_better": True,
        "scaling": "strong" if mixed else "weak", "vs_baseline": None, "dtype": dname, "data": "synthetic",
        "config": {"workload": (f"one synthetic segment: {n_total - 64} dynamic items (40 tracks) x 5120 pts + 64 static "
                                f"crops x 4096 pts, {args.precision}, both heads back to back "
                                "(BASELINE.json configs[3])") if mixed else
                               (f"StaticModelOneBoxEst forward+decode, {B} crops x {N} pts per GPU, {args.precision}"
                                + (" (BASELINE.json configs[1])" if (B, N) == (4096, 1024) else "")) if static else
                               (f"DynamicModel forward+decode, {B} items x {N} pts + 101 boxes per GPU, {args.precision} "
                                "arithmetic (BASELINE.json configs[2] shape)"),
                   "items_per_gpu": B, "points_per_item": N, "sampler": model.sampler,
                   "parallelism": (f"object-sharded x{world}, one all-gather of (B,7) boxes" + (" per head" if mixed else ""))
                   if world > 1 else "single GPU",
                   "algorithmic_gflop_per_item": round(flop_item / 1e9, 4)},
        "whole_path_tflops": round(value * flop_item / 1e12, 2),
        "whole_path_mfma_frac": round(value / world * flop_item / 1e12 / peak, 4),
    }
    if rank == 0 and world == 1 and not args.no_extras and not mixed:
        kr = kernel_rooflines(model, inputs[0], 3 if static else 4, B, N, iters=max(3, min(args.steps, 10)))
        dom = max(kr, key=lambda k: kr[k]["ms"])
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")          # HBM bytes per launch from rocprofv3 --pmc
        if os.path.exists(tfile):
            traffic = json.load(open(tfile)).get(dom)
        pmc = None                                                      # MFMA-busy and clock from the committed PMC passes
        pfile = os.path.join(ROOT, "profiles", "r01_pmc.json")
        if args.precision == "fp32" and os.path.exists(pfile):
            d = json.load(open(pfile)).get(dom, {})
            if "GRBM_GUI_ACTIVE" in d and "SQ_VALU_MFMA_BUSY_CYCLES" in d:
                cyc = d["GRBM_GUI_ACTIVE"] / 8.0                        # the counter sums the 8 XCDs
                pmc = {"source": "profiles/r01_pmc.json (rocprofv3 --pmc, separate passes)",
                       "mfma_busy": round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), 4),
                       "clock_ghz": round(cyc / d["avg_ns_under_GRBM_GUI_ACTIVE"], 3)}
        rec["roofline"] = {"kernel": dom + ("" if args.precision == "fp32" else "_lp"), "bound": "mfma",
                           "achieved": kr[dom]["tflops"], "peak": peak, "unit": "TFLOP/s", "frac": kr[dom]["frac"],
                           "traffic": traffic, "ms_per_launch": kr[dom]["ms"],
                           "algorithmic_gflop_per_launch": kr[dom]["algorithmic_gflop"], "pmc": pmc}
        rec["kernels"] = kr
        if static and args.precision == "fp32":
            # the same workload on the 16-bit MFMA path (BASELINE.json configs C3/C5 arithmetic): reported beside
            # the fp32 headline, never as `value`
            rec["lowprec"] = {}
            for prec in ("bf16", "fp16"):
                model.precision = prec
                for _ in range(2):
                    step_fn()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    step_fn()
                torch.cuda.synchronize()
                d = (time.perf_counter() - t1) / args.steps
                k2 = kernel_rooflines(model, inputs[0], 3, B, N, iters=max(3, min(args.steps, 10)))
                d2 = max(k2, key=lambda k: k2[k]["ms"])
                rec["lowprec"][prec] = {"value": round(B / d, 1), "unit": "object-crops/s", "ms_per_step": round(d * 1e3, 3),
                                        "whole_path_tflops": round(B / d * flop_item / 1e12, 1),
                                        "roofline": {"kernel": d2 + "_lp", "bound": "mfma", "achieved": k2[d2]["tflops"],
                                                     "peak": MFMA_PEAK_TFLOPS[prec], "unit": "TFLOP/s",
                                                     "frac": k2[d2]["frac"], "ms_per_launch": k2[d2]["ms"]},
                                        "kernels": k2}
            model.precision = args.precision
        rec["maxpool"] = maxpool_roofline(dev, iters=5)
        if static:
            if args.precision == "fp32":
                rec["torch_gpu_baseline"] = torch_gpu_baseline(model, inputs)
            rec["cpu_baseline"] = cpu_baseline(host)
    if use_dist:
        fence()
        torch.distributed.destroy_process_group()
    sys.stdout.flush()
    C.CDLL(None).fflush(None)                                   # whatever C stdio still holds goes to stderr too
    if rank == 0:
        os.write(real_stdout, (json.dumps(rec) + "\n").encode())
    os.close(real_stdout)


if __name__ == "__main__":
    main()
