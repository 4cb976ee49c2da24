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


def test_captured_train_step_matches_eager_steps():
    """forward + criterion + backward + Adam as one hipGraph: warm-up step + one replay leave the parameters where two
    eager steps leave them (Dropout off: its draw differs between the two runs' RNG positions)"""
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
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True)

        def step(p_, i_, g_, model=model):
            return crit(model(p_, i_, g_), *labels)["total_loss"]
        if mode == "graph":
            cap = graph.CapturedTrainStep(model, opt, step, pts, i, g, warmup=1)      # one real step; capturing runs nothing
            last = cap(pts, i, g)                                                     # the second step
        else:
            for _ in range(2):
                opt.zero_grad(set_to_none=True)
                last = step(pts, i, g)
                last.backward()
                opt.step()
        torch.cuda.synchronize()
        runs[mode] = (float(last.detach()), {k: v.detach().clone() for k, v in model.named_parameters()})
    assert abs(runs["graph"][0] - runs["eager"][0]) <= 1e-5 * abs(runs["eager"][0])
    for k, v in runs["eager"][1].items():
        assert torch.allclose(runs["graph"][1][k], v, rtol=1e-5, atol=1e-6), k
