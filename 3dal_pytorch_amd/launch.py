"""One process per GPU, started from a parent that never touches the GPU.

`bench.py --gpus N` (and any other entry point that shards crops over ranks) can be started two
ways: already under `python -m torch.distributed.run` (RANK / WORLD_SIZE in the environment), or
as a plain `python script.py --gpus N`. In the second case `spawn_ranks` starts the N ranks as
children (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
--master-port <free> script.py ...`) BEFORE anything in the parent has made a HIP call — the
parent only counts devices (torch.cuda.device_count() does not initialise the GPU on this image),
relays rank 0's stdout line and exits with the children's exit code. It never exec()s: replacing
a process image after HIP initialisation takes the machine down on this pool.

The heads themselves are single-GPU in the reference; what this replaces is its launcher usage for
sharded inference: `--launcher pytorch` + LOCAL_RANK from torch.distributed.launch
(tools/dist_test.py:59-72) and the pickle all_gather that closes it (tools/dist_test.py:184,
det3d/torchie/trainer/utils.py:114-154).
"""
import os
import socket
import subprocess
import sys


def under_launcher():
    """True when this process is a rank started by torch.distributed.run / torchrun."""
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def visible_gpus():
    """Number of GPUs this process could use, without creating a HIP context."""
    import torch
    return torch.cuda.device_count()


def spawn_ranks(script, argv, nproc, need_gpus=True, env=None, timeout=None):
    """Start `nproc` ranks of `script argv...` and wait. Returns (exit code, stdout of the job).
    stderr of the ranks goes straight to this process's stderr. A rank that fails makes
    torch.distributed.run tear the others down and return non-zero; that code is passed on."""
    if need_gpus:
        have = visible_gpus()
        if have < nproc:
            sys.stderr.write(f"{os.path.basename(script)}: needs {nproc} GPUs, this machine shows {have}\n")
            return 2, ""
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: what RCCL needs between ranks on this driver
    e.setdefault("OMP_NUM_THREADS", "4")                     # torch.distributed.run would set 1 and warn
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script] + list(argv)
    p = subprocess.run(cmd, stdout=subprocess.PIPE, env=e, text=True, timeout=timeout)
    return p.returncode, p.stdout


def relay_json_line(stdout):
    """The last line of the job's stdout that is a JSON object (rank 0 prints exactly one)."""
    for line in reversed(stdout.splitlines()):
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            return line
    return None
