"""Repro / regression probe for the hipGraph memset-node problem found in round 2 (ROCm 7.0, MI355X).

A captured refine() was replayed correctly, then — after a larger EAGER call and a 64 MiB allocation in the same
process — replays came back wrong in about 40 % of the processes, persistently, until the graph was recorded again;
eager calls were never affected. This script caught it: the global-feature accumulator `g`, which the library zeroed
with hipMemsetAsync (a memset NODE inside the graph), was filled with an arbitrary 32-bit pattern instead of 0 on
replay (the atomicMax of the encode kernel then keeps the garbage). With the zero-fill as a kernel
(fill_words_kernel, csrc/dal3_misc.hip) 0 of 16 processes fail. Run it a dozen times:
    for k in $(seq 12); do python tools/repro_graph_memset.py | grep "^ok"; done
"""
import importlib, sys, torch, ctypes, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from _common import build_model, synth
graph = importlib.import_module("3dal_pytorch_amd.graph")
hip = importlib.import_module("3dal_pytorch_amd._hip")
hiprt = ctypes.CDLL("libamdhip64.so")
def peek(ptr, n):
    buf = np.empty(n, np.uint8)
    rc = hiprt.hipMemcpy(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(ptr), ctypes.c_size_t(n), 2)
    assert rc == 0, rc
    return buf
B, N = 8, 512
hold = [torch.empty(3 << 20, dtype=torch.uint8, device="cuda")]
model = build_model("static_one", synth.state_dict("static_one", seed=3))
p, i, g = (torch.from_numpy(a).cuda() for a in synth.static_crops(B, N, seed=3))
ptrs = []
orig = model._run
def spy(*a, **k):
    o = orig(*a, **k)
    ptrs.append({kk: (v.data_ptr(), v.numel() * v.element_size()) for kk, v in o.items() if torch.is_tensor(v)})
    return o
model._run = spy
cap = graph.CapturedRefine(model, p.transpose(2, 1), i, g)
model._run = orig
cp = ptrs[-1]
first = cap(p.transpose(2, 1), i, g).clone()
big = [torch.from_numpy(a).cuda() for a in synth.static_crops(512, 2048, seed=5)]
model.refine(big[0].transpose(2, 1), big[1], big[2])
junk = torch.full((64 << 20,), 7, dtype=torch.uint8, device="cuda")
r = cap(p.transpose(2, 1), i, g).clone()
ok = torch.equal(r, first)
print("ok", ok, "| ptrs: cap.ws", hex(cap._ws.buf.data_ptr()), "model.ws", hex(model._ws.buf.data_ptr()), "junk", hex(junk.data_ptr()),
      "inputs", [hex(t.data_ptr()) for t in cap.inputs], "blobs", [hex(v.data_ptr()) for v in cap._blobs.values()], "out0", hex(cp["logits"][0]))
if not ok:
    print("   counts", peek(cp["counts"][0], 32).view(np.int32), "mask sum", int(peek(cp["mask"][0], B * N).sum()))
    lg = torch.from_numpy(peek(cp["logits"][0], B * N * 8).view(np.float32).reshape(B, N, 2).copy())
    g_r = cap._ws.buf[:B * 4096].view(torch.float32).view(B, 1024).clone()
    e = model.forward(p.transpose(2, 1), i, g)["logits"].cpu()
    print("   logits vs eager maxdiff", float((lg - e).abs().max()), "replay logits absmax", float(lg.abs().max()), "eager absmax", float(e.abs().max()))
    gf = torch.zeros((B, 1024), device="cuda")
    w = model._cache.get("ins_seg", model.ins_seg, hip.HEAD_INS_SEG, 0)
    hip.check(hip.lib().dal3_ins_seg_encode(hip.ptr(w), 0, 3, hip.bcn(cap.inputs[0]), B, N, hip.ptr(gf), hip.stream()))
    print("   g (ws head) vs eager encode maxdiff", float((g_r - gf).abs().max()), "g absmax", float(g_r.abs().max()), float(gf.abs().max()))
    for k, v in cap._blobs.items():
        print("   blob", k, "frac7", float((v == 7).float().mean()), "is cache's", model._cache._blob[k] is v)
    r2 = cap(p.transpose(2, 1), i, g).clone()
    print("   replay again ok", torch.equal(r2, first))
    cap._ws.buf.zero_()
    r2 = cap(p.transpose(2, 1), i, g).clone()
    g_r = cap._ws.buf[:B * 4096].view(torch.float32).view(B, 1024).clone()
    print("   replay after zeroing the workspace by hand ok", torch.equal(r2, first), "g absmax", float(g_r.abs().max()), "g vs eager", float((g_r - gf).abs().max()))
    cap._ws.buf.fill_(0x7f)
    r2 = cap(p.transpose(2, 1), i, g).clone()
    g_r = cap._ws.buf[:B * 4096].view(torch.float32).view(B, 1024).clone()
    print("   replay after filling the workspace with 0x7f ok", torch.equal(r2, first), "g absmax", float(g_r.abs().max()))
    cap._capture()
    r3 = cap(p.transpose(2, 1), i, g).clone()
    print("   after re-capture ok", torch.equal(r3, first))
