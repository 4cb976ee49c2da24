"""hipGraph capture of refine(): replays must reproduce the eager path bit for bit, also on new inputs."""
import importlib

import pytest
import torch

from _common import build_model, synth

graph = importlib.import_module("3dal_pytorch_amd.graph")
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", ["static_one", "static_two"])
def test_captured_static_refine_equals_eager(kind):
    B, N = 16, 1024
    model = build_model(kind, synth.state_dict(kind, seed=3))
    p, i, g = (torch.from_numpy(a).cuda() for a in synth.static_crops(B, N, seed=3))
    cap = graph.CapturedRefine(model, p.transpose(2, 1), i, g)
    assert torch.equal(cap(p.transpose(2, 1), i, g), model.refine(p.transpose(2, 1), i, g))
    p2, i2, g2 = (torch.from_numpy(a).cuda() for a in synth.static_crops(B, N, seed=4))
    want = model.refine(p2.transpose(2, 1), i2, g2).clone()
    assert torch.equal(cap(p2.transpose(2, 1), i2, g2), want)
    assert not torch.equal(want, model.refine(p.transpose(2, 1), i, g))


def test_captured_dynamic_refine_equals_eager():
    B = 4
    model = build_model("dynamic", synth.state_dict("dynamic", seed=5))
    p, bx, i8, _ = synth.dynamic_items(B, seed=5)
    dp, db, di = torch.from_numpy(p).cuda().transpose(2, 1), torch.from_numpy(bx).cuda().transpose(2, 1), torch.from_numpy(i8).cuda()
    cap = graph.CapturedRefine(model, dp, db, di)
    for _ in range(3):
        assert torch.equal(cap(dp, db, di), model.refine(dp, db, di))


def test_captured_graph_survives_workspace_growth_and_follows_weight_updates():
    """The graph bakes in raw pointers: (1) an eager call with a larger batch re-allocates the model's workspace —
    the capture runs on its own and must be unaffected; (2) new weights (load_state_dict, an in-place update, or a
    `.data` write followed by invalidate_packed()) must not be answered with the stale packed blobs."""
    B, N = 8, 512
    model = build_model("static_one", synth.state_dict("static_one", seed=3))
    p, i, g = (torch.from_numpy(a).cuda() for a in synth.static_crops(B, N, seed=3))
    cap = graph.CapturedRefine(model, p.transpose(2, 1), i, g)
    first = cap(p.transpose(2, 1), i, g).clone()
    big = [torch.from_numpy(a).cuda() for a in synth.static_crops(512, 2048, seed=5)]
    model.refine(big[0].transpose(2, 1), big[1], big[2])          # grows model._ws: a new buffer
    junk = torch.full((64 << 20,), 7, dtype=torch.uint8, device="cuda")   # whatever was freed is likely reused here
    assert torch.equal(cap(p.transpose(2, 1), i, g), first)
    del junk
    assert cap.recaptures == 0
    # (2a) load_state_dict
    sd2 = synth.state_dict("static_one", seed=4)
    model.load_state_dict({k: torch.as_tensor(v) for k, v in sd2.items()})
    want = model.refine(p.transpose(2, 1), i, g).clone()
    assert not torch.equal(want, first)
    assert torch.equal(cap(p.transpose(2, 1), i, g), want) and cap.recaptures == 1
    # (2b) a write through .data does not bump the version counter: invalidate_packed() is the documented hook
    with torch.no_grad():
        model.box_est.fc3.bias.data.add_(0.25)
    model.invalidate_packed()
    want = model.refine(p.transpose(2, 1), i, g).clone()
    assert torch.equal(cap(p.transpose(2, 1), i, g), want) and cap.recaptures == 2
    # (2c) an ordinary in-place update (what an optimizer step is) is seen without any call
    with torch.no_grad():
        model.box_est.fc3.bias.add_(0.25)
    want2 = model.refine(p.transpose(2, 1), i, g).clone()
    assert not torch.equal(want2, want)
    assert torch.equal(cap(p.transpose(2, 1), i, g), want2) and cap.recaptures == 3


def test_data_writes_need_invalidate_packed_and_debug_mode_finds_them(monkeypatch):
    model = build_model("static_one", synth.state_dict("static_one", seed=3))
    p, i, g = (torch.from_numpy(a).cuda() for a in synth.static_crops(4, 512, seed=3))
    a = model.refine(p.transpose(2, 1), i, g).clone()
    model.box_est.fc3.bias.data.add_(0.5)                        # invisible to the version counter
    model.invalidate_packed()
    b = model.refine(p.transpose(2, 1), i, g).clone()
    assert not torch.equal(a, b) and float((b[:, :3] - a[:, :3] - 0.5).abs().max()) < 1e-5
    # DAL3_CHECK_PACKED=1: the cache checksums the tensors on the device and repacks by itself
    monkeypatch.setenv("DAL3_CHECK_PACKED", "1")
    m2 = build_model("static_one", synth.state_dict("static_one", seed=3))
    assert torch.equal(m2.refine(p.transpose(2, 1), i, g), a)
    m2.box_est.fc3.bias.data.add_(0.5)
    assert torch.equal(m2.refine(p.transpose(2, 1), i, g), b)


def test_capture_refuses_the_numpy_sampler_and_train_mode():
    model = build_model("static_one", synth.state_dict("static_one"))
    p, i, g = (torch.from_numpy(a).cuda() for a in synth.static_crops(2, 256))
    model.sampler = "numpy"
    with pytest.raises(RuntimeError):
        graph.CapturedRefine(model, p.transpose(2, 1), i, g)
    model.sampler = "device"
    model.train()
    with pytest.raises(RuntimeError):
        graph.CapturedRefine(model, p.transpose(2, 1), i, g)


@pytest.mark.parametrize("inside", [True, False])
def test_captured_train_step_matches_eager_steps(inside):
    """forward + criterion + backward + Adam as one hipGraph (inside), or forward + criterion + backward as the graph
    with the plain Adam stepping eagerly behind each replay (round 5): warm-up step + replays leave the parameters where
    the same number of eager steps leave them (Dropout off: its draw differs between the two runs' RNG positions)"""
    losses = importlib.import_module("3dal_pytorch_amd.losses")
    B, N = 8, 1024
    p, i, g = (torch.from_numpy(a).cuda() for a in synth.static_crops(B, N, seed=6))
    pts = p.transpose(2, 1)
    gen = torch.Generator(device="cuda").manual_seed(1)
    labels = ((torch.rand((B, N), device="cuda", generator=gen) > 0.6).float(), torch.randn((B, 3), device="cuda", generator=gen),
              torch.randint(0, 12, (B,), device="cuda", generator=gen), 0.1 * torch.randn((B,), device="cuda", generator=gen),
              torch.randint(0, 3, (B,), device="cuda", generator=gen), 0.3 * torch.randn((B, 3), device="cuda", generator=gen))
    crit = losses.FrustumPointNetLossOneBoxEst()
    runs = {}
    for mode in ("graph", "eager"):
        model = build_model("static_one", synth.state_dict("static_one", seed=6)).train()
        model.ins_seg.dropout.p = 0.0
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, **({"capturable": True} if inside else {}))

        def step(p_, i_, g_, model=model):
            return crit(model(p_, i_, g_), *labels)["total_loss"]
        if mode == "graph" and inside:
            cap = graph.CapturedTrainStep(model, opt, step, pts, i, g, warmup=1)      # one real step; capturing runs nothing
            last = cap(pts, i, g)                                                     # the second step
        elif mode == "graph":
            # the warm-up (default: 3 iterations) runs forward + backward only, no optimizer step, and leaves no trace:
            # BN running statistics and the draw counters are put back — two replays = two eager steps
            cap = graph.CapturedTrainStep(model, opt, step, pts, i, g, optimizer_in_graph=False)
            cap(pts, i, g)
            last = cap(pts, i, g)
        else:
            for _ in range(2):
                opt.zero_grad(set_to_none=True)
                last = step(pts, i, g)
                last.backward()
                opt.step()
        torch.cuda.synchronize()
        runs[mode] = (float(last.detach()), {k: v.detach().clone() for k, v in model.named_parameters()},
                      {k: v.detach().clone() for k, v in model.named_buffers()})
    assert abs(runs["graph"][0] - runs["eager"][0]) <= 1e-5 * abs(runs["eager"][0])
    for k, v in runs["eager"][1].items():
        assert torch.allclose(runs["graph"][1][k], v, rtol=1e-5, atol=1e-6), k
    for k, v in runs["eager"][2].items():                                             # BN running statistics, num_batches_tracked
        assert torch.allclose(runs["graph"][2][k].double(), v.double(), rtol=1e-5, atol=1e-6), k


def test_a_captured_training_draw_advances_on_every_replay():
    """The device sampler's key and the Dropout key carry a draw counter that lives in device memory and is bumped by a
    captured op: a hipGraph replay of a training forward draws fresh object points / a fresh Dropout pattern each
    time (a host-computed seed would be frozen into the kernel arguments at capture)."""
    sm = importlib.import_module("3dal_pytorch_amd.static_model")
    train = importlib.import_module("3dal_pytorch_amd.train")
    model = build_model("static_one", synth.state_dict("static_one", seed=3)).train()
    B, N = 4, 2048
    pts = torch.from_numpy(synth.static_crops(B, N, seed=3)[0]).cuda().transpose(2, 1)
    logits = torch.zeros((B, N, 2), device="cuda")
    logits[:, : N // 2 + 300, 1] = 1.0                                  # 1324 segmented points per crop: a real subset draw
    x = torch.randn((B * N, 128), device="cuda")

    def draw():
        obj, _ = sm._mask_and_gather(pts, logits, 512, 3, model)
        key = (99, train.draw_step(model.ins_seg, pts.device), 0.5)
        train.draw_step(model.ins_seg, pts.device).add_(1)
        return obj.clone(), train._act_dropout(x, None, key)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        draw()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        obj, dropped = draw()
    outs = []
    for _ in range(3):
        g.replay()
        outs.append((obj.clone(), dropped.clone(), int(train.draw_step(model, pts.device)), int(train.draw_step(model.ins_seg, pts.device))))
    assert outs[1][2] == outs[0][2] + 1 and outs[2][3] == outs[1][3] + 1
    assert not torch.equal(outs[0][0], outs[1][0]) and not torch.equal(outs[1][0], outs[2][0])       # new object points
    assert not torch.equal(outs[0][1], outs[1][1]) and not torch.equal(outs[1][1], outs[2][1])       # new Dropout pattern


def test_stream_pipe_returns_what_sequential_calls_return():
    """graph.StreamPipe: consecutive refine() calls on two streams with a workspace each; the boxes come back in
    submission order and are bitwise what the same calls give one after the other (different batches, different
    item offsets, a batch size that changes in between)"""
    model = build_model("static_one", synth.state_dict("static_one", seed=33))
    batches = []
    for k, (B, N) in enumerate([(8, 1024), (8, 1024), (5, 512), (8, 1024), (16, 2048), (8, 1024)]):
        p, i, _ = synth.static_crops(B, N, seed=40 + k)
        batches.append((torch.from_numpy(p).cuda().transpose(2, 1), torch.from_numpy(i).cuda()))
    want = []
    for k, b in enumerate(batches):
        model.item_offset = 100 * k
        want.append(model.refine(*b).clone())
    pipe = graph.StreamPipe(model, depth=2)
    got = []
    for k, b in enumerate(batches):
        model.item_offset = 100 * k
        pipe.submit(*b)
        got += pipe.collect(keep=1)
    got += pipe.collect()
    torch.cuda.synchronize()
    assert len(got) == len(want) and all(torch.equal(a, b) for a, b in zip(got, want))
    model.train()
    with pytest.raises(RuntimeError):
        graph.StreamPipe(model)
