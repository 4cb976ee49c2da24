"""SURVEY.md 8(b) "Threading": the library keeps no mutable global state — launches go to the stream they are given,
the error string is thread-local. Two host threads drive two models on two streams at once; each must get what it
gets alone, and an error raised in one thread must not show up in the other's dal3_last_error()."""
import importlib
import threading

import pytest
import torch

from _common import build_model, synth

hip = importlib.import_module("3dal_pytorch_amd._hip")
pytestmark = pytest.mark.gpu


def test_two_threads_two_streams_give_the_sequential_results():
    static = build_model("static_two", synth.state_dict("static_two", seed=41))
    dynamic = build_model("dynamic", synth.state_dict("dynamic", seed=42))
    p, i, g = (torch.from_numpy(a).cuda() for a in synth.static_crops(64, 1024, seed=41))
    dp, db, di, _ = synth.dynamic_items(8, seed=42)
    dp, db, di = torch.from_numpy(dp).cuda(), torch.from_numpy(db).cuda(), torch.from_numpy(di).cuda()
    want_s = static.refine(p.transpose(2, 1), i, g).clone()
    want_d = dynamic.refine(dp.transpose(2, 1), db.transpose(2, 1), di).clone()
    torch.cuda.synchronize()
    out, errs = {}, []

    def work(name, fn):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                for _ in range(5):
                    res = fn()
                stream.synchronize()
            out[name] = res.clone()
        except Exception as e:                               # noqa: BLE001
            errs.append((name, e))

    threads = [threading.Thread(target=work, args=("s", lambda: static.refine(p.transpose(2, 1), i, g))),
               threading.Thread(target=work, args=("d", lambda: dynamic.refine(dp.transpose(2, 1), db.transpose(2, 1), di)))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    assert torch.equal(out["s"], want_s) and torch.equal(out["d"], want_d)


def test_error_string_is_thread_local():
    lib = hip.lib()
    seen = {}

    def bad():
        rc = lib.dal3_maxpool_n(None, 0, 0, None, hip.stream())
        seen["bad"] = (rc, lib.dal3_last_error().decode())

    def good():
        x = torch.rand((8, 64), device="cuda")
        o = torch.empty(8, device="cuda")
        rc = lib.dal3_maxpool_n(hip.ptr(x), 8, 64, hip.ptr(o), hip.stream())
        torch.cuda.synchronize()
        seen["good"] = (rc, lib.dal3_last_error().decode(), bool(torch.equal(o, x.max(1)[0])))

    t1 = threading.Thread(target=bad)
    t1.start()
    t1.join()
    t2 = threading.Thread(target=good)
    t2.start()
    t2.join()
    assert seen["bad"][0] != 0 and "maxpool_n" in seen["bad"][1]
    assert seen["good"][0] == 0 and "maxpool_n" not in seen["good"][1] and seen["good"][2]
