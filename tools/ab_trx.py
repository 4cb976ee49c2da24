import sys, os, importlib, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
hip = importlib.import_module("3dal_pytorch_amd._hip")
def load(path):
    h = C.CDLL(os.path.abspath(path))
    for name, (res, a) in hip.SIGNATURES.items():
        if hasattr(h, name):
            fn = getattr(h, name); fn.restype, fn.argtypes = res, a
    return h
libs = [(p, load(p)) for p in sys.argv[1:]]
def ms(fn, it=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / it
M = 262144
for ci, co in ((512, 256), (128, 1024), (64, 512)):
    a = torch.randn((M, ci), device="cuda"); W = torch.randn((co, ci), device="cuda") / ci ** 0.5
    sc = torch.rand(ci, device="cuda") + 0.5; sh = torch.randn(ci, device="cuda") * 0.3
    bias = torch.randn((1, co), device="cuda"); z = torch.empty((M, co), device="cuda")
    out = []
    for name, lib in libs:
        pk = torch.empty(lib.dal3_tr_linear_workspace_bytes(ci, co), dtype=torch.uint8, device="cuda")
        item = (hip.PackItem * 1)(hip.PackItem(hip.ptr(W), W.stride(0), 0, co, ci, 0x108, hip.ptr(pk)))
        assert lib.dal3_tr_pack_many(item, 1, hip.stream()) == 0
        def run(lib=lib, pk=pk):
            assert lib.dal3_tr_linear_x3(hip.ptr(a), M, ci, a.stride(0), hip.ptr(sc), hip.ptr(sh), 1, hip.ptr(bias), 0, co, hip.ptr(z), z.stride(0), hip.ptr(pk), None, hip.stream()) == 0
        out.append(f"{os.path.basename(name)} {ms(run):.3f}")
    print(ci, co, " | ".join(out), flush=True)
