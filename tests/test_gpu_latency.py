"""The latency kernels (csrc/dal3_latency.hip: small jobs, one 16-wave workgroup per 32-point tile, activations
through LDS) against the throughput kernels (one wave per tile, activations in registers): the dispatch between the
two families must be invisible — bit-identical logits, masks, counts, drawn indices, box parameters and refined boxes.
The throughput family's outputs come from the same calls with DAL3_BCN_NO_SMALL_JOB_KERNELS in dal3_bcn.flags (the
dispatch is a function of the job size and of that per-call bit; the library reads no environment); both are pinned to the oracle elsewhere (tests/test_gpu_parity.py runs the small fixtures through
the latency family now, the large ones through the throughput family)."""
import numpy as np
import pytest
import torch

import importlib

from _common import build_model, synth

hip = importlib.import_module("3dal_pytorch_amd._hip")

pytestmark = pytest.mark.gpu

CASES = [("static_one", 1, 1024), ("static_one", 3, 700), ("static_two", 2, 33), ("static_one", 16, 1024),
         ("dynamic", 2, 0), ("static_one", 5, 4096)]
# round 6: the throughput decode kernel gives XCD x (workgroup ids = x mod 8) a contiguous range of logical blocks
# (csrc/dal3_pointmlp.hip). N = 600 is five 128-point workgroups per crop, so B = 17 .. 24 makes grids of 85 .. 120 blocks:
# every residue of G mod 8 (5, 2, 7, 4, 1, 6, 3, 0) — the ragged split of the map — against the latency family, which has
# no such map. (323 .. 456 tiles of 32 points: the default dispatch takes the latency kernels, the flag the throughput ones.)
CASES += [("static_one", B, 600) for B in range(17, 25)]


def _run(kind, B, N):
    """every output of one forward + decode, as NumPy arrays"""
    model = build_model(kind, synth.state_dict(kind, seed=51))
    if kind == "dynamic":
        p, bx, i8, _ = synth.dynamic_items(B, n_per_frame=256, seed=52)
        o = model._run(torch.from_numpy(p).cuda().transpose(2, 1), torch.from_numpy(bx).cuda().transpose(2, 1),
                       init_box8=torch.from_numpy(i8).cuda())
    else:
        p, i, g = synth.static_crops(B, N, seed=52)
        p[0, : max(N // 3, 1)] *= 0.85                       # a crop with few segmented points (tiles of copies are skipped)
        o = model._run(torch.from_numpy(p).cuda().transpose(2, 1), torch.from_numpy(i).cuda(), torch.from_numpy(g).cuda())
    return {k: v.cpu().numpy() for k, v in o.items() if torch.is_tensor(v)}


def test_latency_family_equals_throughput_family_bitwise():
    want = {}
    hip.DISPATCH_FLAGS = hip.BCN_NO_SMALL_JOB_KERNELS        # dal3_bcn.flags: the throughput family for every kernel of the call
    try:
        for kind, B, N in CASES:
            want[kind, B, N] = _run(kind, B, N)
    finally:
        hip.DISPATCH_FLAGS = 0
    n = 0
    for kind, B, N in CASES:
        got = _run(kind, B, N)
        for k, v in got.items():
            w = want[kind, B, N][k]
            assert v.shape == w.shape and v.dtype == w.dtype, (kind, B, N, k)
            assert np.array_equal(v, w, equal_nan=True), (kind, B, N, k, float(np.abs(v.astype(np.float64) - w).max()))
            n += 1
    assert n >= 50


def test_unknown_dispatch_flag_bits_are_rejected():
    p, _, _ = synth.static_crops(1, 64, seed=1)
    v = hip.bcn(torch.from_numpy(p).cuda().transpose(2, 1))
    v.flags = 8
    g = torch.zeros((1, 1024), device="cuda")
    assert hip.lib().dal3_ins_seg_encode(hip.ptr(g), hip.F32, 3, v, 1, 64, hip.ptr(g), hip.stream()) == hip.EINVAL
    assert b"flags" in hip.lib().dal3_last_error()


def test_small_job_latency_is_below_the_single_wave_chain():
    """B = 1 crop of 1024 points: GPU time of one refine() replayed from a hipGraph (the host's launch calls out of
    the picture; HIP events over 200 replays, best of 3). With the throughput kernels the three single-wave chains alone
    took 258 us (encode 72 + decode 102 + point head 84 at 2.4 GHz) and the call 318 us; the latency family brought
    the call to ~150 us. The bound leaves room for a slow box, not for the old path."""
    import importlib
    graph = importlib.import_module("3dal_pytorch_amd.graph")
    model = build_model("static_one", synth.state_dict("static_one", seed=51))
    p, i, g = (torch.from_numpy(a).cuda() for a in synth.static_crops(1, 1024, seed=52))
    cap = graph.CapturedRefine(model, p.transpose(2, 1), i, g)
    assert torch.equal(cap(p.transpose(2, 1), i, g), model.refine(p.transpose(2, 1), i, g))
    best = 1e9
    for _ in range(3):
        for _ in range(20):
            cap.graph.replay()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(200):
            cap.graph.replay()
        b.record()
        b.synchronize()
        best = min(best, a.elapsed_time(b) * 5.0)
    print(f"\n[latency] B=1 x 1024 points: {best:.1f} us per refine() (hipGraph replay)")
    assert best < 240.0, best
