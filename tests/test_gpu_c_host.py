"""The C ABI from a plain C host: examples/c_host_static_refine.c (no Python, no torch in the process) is built with
gcc, fed the same weights and crops as the Python module through a file, and must return the same refined boxes bit
for bit — same library, same device sampler key."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest
import torch

from _common import ROOT, build_model, synth

pytestmark = pytest.mark.gpu


def test_c_host_program_matches_the_python_module(tmp_path):
    if shutil.which("gcc") is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("needs gcc and the HIP runtime headers")
    exe = str(tmp_path / "c_host")
    pkg = os.path.join(ROOT, "3dal_pytorch_amd")
    build = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
                            "-I", "/opt/rocm/include", os.path.join(ROOT, "examples", "c_host_static_refine.c"), "-L", pkg,
                            "-l:lib3dal_hip.so", "-L", "/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{pkg}",
                            "-Wl,-rpath,/opt/rocm/lib", "-o", exe], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr

    B, N = 24, 700
    sd = synth.state_dict("static_one", seed=14)
    pts_np, init_np, _ = synth.static_crops(B, N, seed=14)
    layers = [(f"ins_seg.{c}", f"ins_seg.{b}" if b else None) for c, b in
              [("conv1", "bn1"), ("conv2", "bn2"), ("conv3", "bn3"), ("conv4", "bn4"), ("conv5", "bn5"), ("dconv1", "dbn1"),
               ("dconv2", "dbn2"), ("dconv3", "dbn3"), ("dconv4", "dbn4"), ("dconv5", None)]]
    layers += [(f"box_est.{c}", f"box_est.{b}" if b else None) for c, b in
               [("conv1", "bn1"), ("conv2", "bn2"), ("conv3", "bn3"), ("conv4", "bn4"), ("fc1", "fcbn1"), ("fc2", "fcbn2"),
                ("fc3", None)]]
    inp = str(tmp_path / "in.bin")
    with open(inp, "wb") as f:
        f.write(struct.pack("<3i", B, N, len(layers)))
        for conv, bn in layers:
            w = np.asarray(sd[conv + ".weight"], np.float32)
            w = w.reshape(w.shape[0], -1)
            f.write(struct.pack("<3i", w.shape[1], w.shape[0], 1 if bn else 0))
            f.write(w.tobytes())
            f.write(np.asarray(sd[conv + ".bias"], np.float32).tobytes())
            if bn:
                for part in ("weight", "bias", "running_mean", "running_var"):
                    f.write(np.asarray(sd[f"{bn}.{part}"], np.float32).tobytes())
        f.write(pts_np.astype(np.float32).tobytes())
        f.write(init_np.astype(np.float32).tobytes())
    outp = str(tmp_path / "boxes.bin")
    run = subprocess.run([exe, inp, outp], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stderr
    got = np.fromfile(outp, np.float32).reshape(B, 7)

    model = build_model("static_one", sd)
    want = model.refine(torch.from_numpy(pts_np).cuda().transpose(2, 1), torch.from_numpy(init_np).cuda()).cpu().numpy()
    assert np.array_equal(got, want)
    assert "refined 24 crops x 700 points" in run.stdout
