#!/usr/bin/env python3
"""bench.py — object-crops/sec through the refinement heads on MI355X (BASELINE.json's metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2|C3|C4|C5] [--precision fp32|bf16|fp16|f16x3]

Workload (`config.workload`): BASELINE.json configs[1] — StaticModelOneBoxEst, 4096 crops x 1024 points, fp32, per GPU;
synthetic crops and random-init weights (3dal_pytorch_amd/synth.py; no dataset or checkpoint is reachable). One step =
one pass of the hot path over the batch: ins_seg -> mask -> object-point sampling -> box estimator -> decode to (B,7)
boxes, inputs resident in HBM. With N > 1 every rank refines its own 4096 crops (weak scaling, crops are independent)
and ONE RCCL all-gather of the (N*4096, 7) boxes closes the step; it is collected one step later (dist.BoxGatherer) so
it runs beside the next batch's kernels, and every gather is finished inside the timed region. Started with --gpus N > 1
and no launcher in the environment, the process starts the ranks itself as children (3dal_pytorch_amd/launch.py) and
relays rank 0's line; under `python -m torch.distributed.run` it is a rank.

stdout carries ONE JSON line of at most LINE_CAP bytes (tests/test_launch_cpu.py holds it to that): the contract's
fields, `roofline` (the step's dominant kernel, timed live with HIP events on the launch stream: algorithmic FLOP per
launch / average duration vs the MFMA peak of the arithmetic type), `cpu_baseline` (the oracle on this box's host
cores, a bounded sample of the same workload) and, for N > 1, `rccl`. The same record with the whole per-kernel table
goes to gpurun_out/bench_full.json. Everything else this repo measures (16-bit and f16x3 runs of the workload, the
max-pool's HBM roofline, the other configurations, the rows either side of the path) is tools/bench_extras.py's, which
writes a file and prints nothing here (`--extras` runs it after the line is out).
"""
import argparse
import ctypes as C
import gc
import importlib
import json
import os
import subprocess
import sys
import time

# RCCL shares device buffers between the ranks of a node through HIP IPC handles; this pool's host driver implements
# only the dmabuf flavour (with the legacy one hipIpcGetMemHandle fails as soon as two ranks connect). The pool exports
# the variable itself; setdefault keeps a caller's value. It must be set before the HSA runtime loads.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
launch = importlib.import_module("3dal_pytorch_amd.launch")     # (imports no torch)


def rccl_debug_dir():
    """where the ranks' RCCL logs go: gpurun_out/ under the repo, or the system's temporary directory when the repo
    cannot be written (every rank of a node makes the same choice: same file system, same user)"""
    name = "rccl_debug_" + os.environ.get("MASTER_PORT", "29533")
    for base in (os.path.join(ROOT, "gpurun_out"), os.path.join(os.environ.get("TMPDIR", "/tmp"), "dal3_bench")):
        try:
            os.makedirs(os.path.join(base, name), exist_ok=True)
            if os.access(os.path.join(base, name), os.W_OK):
                return os.path.join(base, name)
        except OSError:
            continue
    return None


# A rank of an RCCL job (not the self-launching parent, not a gloo rehearsal): RCCL's own INIT / P2P log goes to one
# file per rank, parsed into the line's `rccl.transport` after the run. RCCL reads the variables when `import torch`
# loads it, so this happens here, before that import.
RCCL_RANK = ((int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("DAL3_FORCE_DIST") == "1")
             and os.environ.get("DAL3_BENCH_BACKEND", "nccl") == "nccl" and "--plumbing-only" not in sys.argv)
T_START = time.time()
RCCL_DEBUG_DIR = rccl_debug_dir() if RCCL_RANK else None
if RCCL_DEBUG_DIR:
    launch.rccl_debug_to(RCCL_DEBUG_DIR)

import numpy as np                                           # noqa: E402
import torch                                                 # noqa: E402

from bench_workloads import (Part, Workload, apply_config, build_workload, workload_shards,      # noqa: E402,F401
                             make_static, make_dynamic, static_inputs, dynamic_inputs)
from bench_kernels import MFMA_PEAK_TFLOPS, DNAME, events_ms, kernel_table, roofline_of      # noqa: E402
dal3_dist = importlib.import_module("3dal_pytorch_amd.dist")

LINE_CAP = 4096            # bytes of the one stdout line; r04's 21.7 KB line was not parsed by the driver
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")
OPTIONAL = ("gather_self_check", "ms_per_step_per_rank", "ms_per_step_max", "ms_per_step_min", "ms_per_step_median", "rccl", "strong")
METRIC = "object-crops/sec through static+dynamic refinement heads"


def compact_line(rec):
    """the one stdout line: strict JSON (no NaN/Infinity), at most LINE_CAP bytes — optional keys are dropped, most
    dispensable first, until it fits (the full record is in gpurun_out/bench_full.json either way)"""
    rec = dict(rec)
    line = json.dumps(rec, allow_nan=False, separators=(", ", ": "))
    for k in OPTIONAL:
        if len(line.encode()) <= LINE_CAP:
            break
        if rec.pop(k, None) is not None:
            rec.setdefault("dropped_for_size", []).append(k)
        line = json.dumps(rec, allow_nan=False, separators=(", ", ": "))
    assert len(line.encode()) <= LINE_CAP and "\n" not in line, len(line)
    return line


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

    def step(final=False):
        for i, (fn, _, _) in enumerate(wl.parts):
            wl.gatherers[i].submit(fn())
            got = wl.gatherers[i].collect(keep=1 if (overlap and not final) else 0)
            if got is not None:
                last[i] = got
    gc_was = gc.isenabled()
    if os.environ.get("DAL3_BENCH_GC") != "1":              # (=1: leave the collector on, to reproduce the stall)
        # As timeit does: no cyclic-GC pass inside the timed region (a generation-2 pass is a 40 ms host stall; when it
        # falls right behind the fence the GPU waits for it). Collected BEFORE the warm-up: 40 ms of idling drops the clocks.
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


def strong_shards(args, world):
    """the STRONG-scaling split of the configured single-head workload: ONE batch of the N = 1 size —
    C2: 4096 crops — in contiguous ranges, rank r gets [r * ceil(B / W), ...). Returns (head, items, rank -> (first, count))."""
    static = args.head == "static"
    n_total = args.batch or (4096 if static else 1024)
    return ("static" if static else "dynamic", n_total,
            lambda r: (lambda lo, hi: (lo, hi - lo))(*dal3_dist.shard_range(n_total, r, world)))


def strong_leg(args, dev, rank, world, use_dist):
    """The same step with the batch of the N = 1 run SPLIT over the ranks instead of repeated on each (VERDICT r5 #2:
    `value` above is weak scaling and says nothing about per-rank launch overhead or the gather; this record does).
    Same fences, same K, the max over ranks; `vs_n1` stays null: the N = 1 figure is another run's line (the driver
    divides the lines it collected itself)."""
    name, n_total, span = strong_shards(args, world)
    lo, n = span(rank)
    prec, static = args.precision, args.head == "static"
    if static:
        model, inputs, _ = make_static(max(n, 1), args.points, dev, lo, prec, two=getattr(args, "two_stage", False))
        inputs_for = lambda first, count: static_inputs(first, count, args.points, dev, prec)      # noqa: E731
    else:
        model, inputs = make_dynamic(max(n, 1), dev, lo, prec, args.points)
        inputs_for = lambda first, count: dynamic_inputs(first, count, args.points, dev, prec)     # noqa: E731
    wl = Workload()
    wl.parts = [Part(name, model, inputs, n, n_total, span, inputs_for)]
    wl.n_total = n_total
    dt, per_step, _ = time_steps(wl, dev, args.steps, args.warmup, use_dist, overlap=not args.serial_gather)
    check = gather_self_check(wl, dev, rank, world)
    return {"scaling": "strong", "items": n_total, "items_per_rank": [span(r)[1] for r in range(world)],
            "value": round(n_total * args.steps / dt, 1), "unit": "object-crops/s", "ms_per_step": round(dt / args.steps * 1e3, 3),
            "ms_per_step_per_rank": [round(t / args.steps * 1e3, 3) for t in wl.rank_seconds],
            "gather_equals_single_rank": check["equal"] if check and rank == 0 else None, "vs_n1": None}


def transport_report(dev, backend, debug_dir):
    """which wires the collective used: every rank's HIP peer-access row (gathered; call before the group is destroyed)
    and what RCCL's own INIT / P2P log says (parsed by rank 0 after the run). On gloo: the rows only."""
    rows = dal3_dist.gather_rows(dal3_dist.peer_access_row(dev), 16, dev)
    n_dev = torch.cuda.device_count() if dev.type == "cuda" else 0
    return {"backend": backend, "peer_access": ["".join(str(v) if v >= 0 else "" for v in r[:n_dev]) for r in rows],
            "debug_dir": (os.path.relpath(debug_dir, ROOT) if debug_dir.startswith(ROOT) else debug_dir) if debug_dir else None}


def transport_from_logs(tr):
    """rank 0, after destroy_process_group: the ranks' RCCL debug files -> `via` counts etc. (dist.parse_rccl_debug)"""
    d = tr.get("debug_dir")
    if not d:
        return tr
    texts = []
    try:
        for f in sorted(os.listdir(os.path.join(ROOT, d))):                 # (os.path.join keeps an absolute d as it is)
            path = os.path.join(ROOT, d, f)
            if os.path.getmtime(path) < T_START - 5.0:                      # an earlier job's log under the same port
                continue
            with open(path, errors="replace") as fh:
                texts.append(fh.read())
    except OSError:
        pass
    return dict(tr, **dal3_dist.parse_rccl_debug(texts))


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
    strong = None
    if scaling == "weak":                                   # the split the GPU run's `strong` record uses, gathered once
        name, n_strong, sspan = strong_shards(args, world)
        lo, cnt = sspan(rank)
        sstub = _Stub(7000.0)
        dal3_dist.replicate_(sstub.bias)
        g = dal3_dist.all_gather_boxes(sstub.refine(*idx(lo, cnt)), n_strong)
        want = torch.arange(n_strong, dtype=torch.float32)[:, None] * 10 + torch.arange(7, dtype=torch.float32)[None] + 1.0 + 7000.0
        s_ok = g.shape == (n_strong, 7) and bool(torch.equal(g, want))
        ok = ok and s_ok
        strong = {"scaling": "strong", "items": n_strong, "items_per_rank": [sspan(r)[1] for r in range(world)], "value": None,
                  "unit": "object-crops/s", "ms_per_step": None, "gathered_ok": s_ok, "vs_n1": None}
    census["transport"] = transport_report(cpu, backend, None)
    if os.environ.get("DAL3_BENCH_FAIL_RANK") == str(rank):          # the launcher's failure path, for its test
        sys.exit(3)
    if os.environ.get("DAL3_BENCH_HANG_RANK") == str(rank):          # a rank that never comes back, for the launcher's timeout
        time.sleep(3600)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
    if rank == 0:
        print(compact_line({"metric": METRIC, "value": None, "unit": "object-crops/s", "n_gpus": world, "steps": 3, "warmup": 0,
                            "ms_per_step": max(rank_ms) / 3, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
                            "dtype": None, "data": "none (plumbing only)", "plumbing_only": True, "gathered_ok": ok,
                            "config": {"workload": args.config or "stub", "heads": [
                                {"head": p.name, "items": p.n_total, "items_per_rank": [p.shard(r)[1] for r in range(world)]}
                                for p in wl.parts]},
                            "roofline": None, "cpu_baseline": None, "rccl": census, "strong": strong, "ms_per_step_per_rank": rank_ms,
                            "gather_equals_single_rank": check["equal"] if check else None, "gather_self_check": check}),
              flush=True)
    sys.exit(0 if ok else 4)


def write_full(rec):
    """the whole record (per-kernel table included), for profiles/: a file, never stdout"""
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_full.json"), "w") as f:
            json.dump(rec, f, indent=1)
    except OSError as e:
        sys.stderr.write(f"bench.py: bench_full.json not written: {e}\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)           # >= 20 iterations after 5 warm-ups
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default=None, choices=["C2", "C3", "C4", "C5", "TwoBoxEst", "Dynamic_fp32", "TwoBoxEst_f16x3",
                                                       "Dynamic_f16x3", "C4_f16x3"],
                    help="BASELINE.json configs by name: C2 = the default (static, 4096 x 1024, fp32); C3 = dynamic, 1024 items x 5 x "
                         "1024 pts, bf16; C4 = one segment sharded over the GPUs (strong scaling); C5 = static, 2048 x 4096, fp16 MFMA")
    ap.add_argument("--head", default="static", choices=["static", "dynamic"])
    ap.add_argument("--batch", type=int, default=0, help="items per GPU (default: 4096 static, 1024 dynamic)")
    ap.add_argument("--points", type=int, default=1024)
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16", "fp16", "f16x3"])
    ap.add_argument("--two-stage", action="store_true", help="static head: StaticModelTwoBoxEst instead of OneBoxEst")
    ap.add_argument("--serial-gather", action="store_true", help="wait for each step's all-gather inside the step")
    ap.add_argument("--no-extras", action="store_true", help="timed steps only: no roofline / cpu_baseline legs (profiling runs)")
    ap.add_argument("--extras", action="store_true", help="after the line: run tools/bench_extras.py (writes gpurun_out/bench_extras.json)")
    ap.add_argument("--plumbing-only", action="store_true", help="launcher + process group + gather only, no GPU work (CPU tests)")
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of host time the cpu_baseline leg may take")
    ap.add_argument("--cpu-sweep", type=int, nargs="+", default=None, metavar="THREADS", help="only the cpu_baseline leg at these thread counts")
    args = ap.parse_args()
    if args.cpu_sweep:
        synth = importlib.import_module("3dal_pytorch_amd.synth")
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
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started {world} ranks")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if args.plumbing_only:
        plumbing_only(args, rank, world)
    apply_config(args)
    n_dev = torch.cuda.device_count()
    if share_gpu and n_dev > 0:
        local = local % n_dev
    if n_dev <= local:                                      # a rank on a box with too few GPUs: the line and exit code the
        sys.stderr.write(f"bench.py: needs {max(args.gpus, local + 1)} GPUs, this machine shows {n_dev}\n")   # self-launching
        sys.exit(2)                                         # parent gives (launch.spawn_ranks)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or os.environ.get("DAL3_FORCE_DIST") == "1"      # =1: the RCCL path with one rank (1-GPU box)
    # stdout carries exactly one JSON line. RCCL prints its banner to the C-level stdout (and flushes it at exit), so
    # while anything but that line can be written, file descriptor 1 points at stderr.
    real_stdout = os.dup(1)
    sys.stdout.flush()
    os.dup2(2, 1)
    if use_dist:
        # "nccl" = RCCL over xGMI, the product transport. DAL3_BENCH_BACKEND=gloo: the rehearsal transport (boxes staged
        # through pinned host memory) for boxes where RCCL cannot connect the ranks, e.g. two ranks on ONE GPU.
        backend = os.environ.get("DAL3_BENCH_BACKEND", "nccl")
        debug_dir = RCCL_DEBUG_DIR                          # RCCL's own account of the transport, one file per rank
        torch.distributed.init_process_group(backend, rank=rank, world_size=world, **({"device_id": dev} if backend == "nccl" else {}))

    wl = build_workload(args, dev, rank, world)
    dt, per_step, _ = time_steps(wl, dev, args.steps, args.warmup, use_dist, overlap=not args.serial_gather)
    value = wl.n_total * args.steps / dt
    mixed = args.config == "C4"
    rec = {"metric": METRIC, "value": round(value, 1), "unit": "object-crops/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
           "ms_per_step_median": round(per_step[len(per_step) // 2], 3), "ms_per_step_min": round(per_step[0], 3),
           "ms_per_step_max": round(per_step[-1], 3), "higher_is_better": True, "scaling": wl.scaling, "vs_baseline": None,
           "dtype": DNAME[args.precision], "data": "synthetic",
           "config": {"workload": wl.desc, "items_per_gpu": wl.B, "points_per_item": wl.N, "sampler": wl.model.sampler,
                      "parallelism": (f"object-sharded x{world}, one all-gather of (B,7) boxes" + (" per head" if mixed else "")
                                      + (", collected one step later" if not args.serial_gather else ", serial"))
                      if use_dist else "single GPU", "algorithmic_gflop_per_item": round(wl.flop_item / 1e9, 4)},
           "roofline": None, "cpu_baseline": None}
    full = {}
    if use_dist:
        census = dal3_dist.world_census(dev)
        rec["ms_per_step_per_rank"] = [round(t / args.steps * 1e3, 3) for t in wl.rank_seconds]   # each rank's own clock
        check = gather_self_check(wl, dev, rank, world)
        rec["gather_equals_single_rank"] = check["equal"] if check and rank == 0 else None
        rec["gather_self_check"] = check
        rec["rccl"] = dict(census, allgather_us=gather_latency(wl, dev), ranks_share_gpus=share_gpu,
                           message_bytes_per_rank=[((n_total + world - 1) // world) * 28 for _, _, n_total in wl.parts],
                           rccl_version=".".join(str(v) for v in torch.cuda.nccl.version()),
                           transport=transport_report(dev, backend, debug_dir))
        # ONE batch of the N = 1 size split over the ranks (C4 is that already)
        rec["strong"] = None if mixed else strong_leg(args, dev, rank, world, use_dist)
    if rank == 0 and not args.no_extras and not mixed:
        # the dominant kernel, live: every kernel of the step through its own entry, HIP events on the launch stream
        kr, mean_count = kernel_table(wl.model, wl.inputs, wl.static, wl.B, wl.N, iters=max(3, min(args.steps, 10)))
        rec["roofline"] = roofline_of(kr, MFMA_PEAK_TFLOPS[args.precision], args.precision, wl.B, wl.N)
        full = {"kernels": kr, "mean_segmented_points_per_item": round(mean_count, 1)}
        if world == 1 and wl.static:
            rec["cpu_baseline"] = cpu_baseline(wl.host, budget_s=args.cpu_budget)
    if use_dist:
        torch.distributed.barrier()
        torch.cuda.synchronize()
        torch.distributed.destroy_process_group()
    sys.stdout.flush()
    C.CDLL(None).fflush(None)                                   # whatever C stdio still holds goes to stderr too
    if rank == 0 and use_dist:                              # (after destroy_process_group: every rank's log is complete)
        rec["rccl"]["transport"] = transport_from_logs(rec["rccl"]["transport"])
    if rank == 0:
        os.write(real_stdout, (compact_line(rec) + "\n").encode())
        write_full(dict(rec, **full))
    os.close(real_stdout)
    if rank == 0 and args.extras and world == 1:               # a child process, after the line is out; prints nothing here
        del wl
        torch.cuda.empty_cache()
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_extras.py"), "--out",
                        os.path.join(ROOT, "gpurun_out", "bench_extras.json")], stdout=sys.stderr, check=False)


if __name__ == "__main__":
    main()
