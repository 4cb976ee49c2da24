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
