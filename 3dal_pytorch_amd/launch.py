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
import signal
import socket
import subprocess
import sys
import threading


def rccl_debug_to(directory):
    """Ask RCCL for its INIT / P2P log, one file per process under `directory`, so that a multi-GPU run can PROVE which
    transport carried the all-gather (xGMI peer-to-peer vs host shared memory vs sockets; dist.parse_rccl_debug reads
    the files). RCCL takes the variables when its library is LOADED, i.e. with `import torch`: call this before that
    import (measured on the pool: set between the import and init_process_group, no file is written). This module
    imports torch nowhere at its top for that reason. The level is raised to INFO unless the caller asked for more
    (TRACE); a caller's NCCL_DEBUG_SUBSYS is kept; the file pattern is this function's. Returns the file pattern."""
    os.makedirs(directory, exist_ok=True)
    if os.environ.get("NCCL_DEBUG", "").upper() not in ("INFO", "TRACE"):
        os.environ["NCCL_DEBUG"] = "INFO"
    os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,P2P,GRAPH")
    os.environ["NCCL_DEBUG_FILE"] = os.path.join(directory, "rccl.%h.%p.log")
    return os.environ["NCCL_DEBUG_FILE"]


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


def _pump(stream, sink, keep):
    """copy a child's pipe to `sink` line by line as it arrives (a hung job shows what it printed) and keep the text"""
    for line in iter(stream.readline, ""):
        keep.append(line)
        if sink is not None:
            sink.write(line)
            sink.flush()
    stream.close()


def _kill_group(p):
    """end the launcher AND its ranks: they share the session started for them. SIGTERM, then SIGKILL."""
    for sig, wait in ((signal.SIGTERM, 10.0), (signal.SIGKILL, 5.0)):
        try:
            os.killpg(p.pid, sig)
        except ProcessLookupError:
            return
        try:
            p.wait(timeout=wait)
            # the agent is gone; ranks it had no time to reap die with the group's next signal
        except subprocess.TimeoutExpired:
            continue
    try:
        os.killpg(p.pid, signal.SIGKILL)
    except ProcessLookupError:
        pass


def spawn_ranks(script, argv, nproc, need_gpus=True, env=None, timeout=None, share_gpu=False):
    """Start `nproc` ranks of `script argv...` and wait. Returns (exit code, stdout of the job).
    stderr of the ranks is passed through to this process's stderr as it is written. A rank that fails makes
    torch.distributed.run tear the others down and return non-zero; that code is passed on. The job runs in a
    session of its own: on `timeout` (exit code 124) or KeyboardInterrupt the WHOLE process group is ended — killing
    only the launcher would orphan the ranks with the GPUs and the rendezvous port in their hands. The rendezvous
    port is picked free and the launch retried when another job took it in between (free_port closes its socket
    before torch.distributed.run binds it).
    share_gpu: the ranks may share devices (rank -> device LOCAL_RANK % device_count; a rehearsal on a 1-GPU box)."""
    if need_gpus:
        have = visible_gpus()
        if have < (1 if share_gpu else nproc):
            sys.stderr.write(f"{os.path.basename(script)}: needs {nproc} GPUs, this machine shows {have}\n")
            return 2, ""
    e = dict(os.environ if env is None else env)
    # dmabuf IPC between the ranks' HIP runtimes: the only flavour this pool's host driver implements (see bench.py);
    # a value the caller exported wins
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    e.setdefault("OMP_NUM_THREADS", "4")                     # torch.distributed.run would set 1 and warn
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    rc, out = 1, []
    for attempt in range(3):
        port = free_port()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), script] + list(argv)
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, text=True, bufsize=1,
                             start_new_session=True)
        out, err = [], []
        pumps = [threading.Thread(target=_pump, args=(p.stdout, None, out), daemon=True),
                 threading.Thread(target=_pump, args=(p.stderr, sys.stderr, err), daemon=True)]
        for t in pumps:
            t.start()
        try:
            rc = p.wait(timeout=timeout)
        except subprocess.TimeoutExpired:
            sys.stderr.write(f"{os.path.basename(script)}: {nproc} ranks still running after {timeout} s, ending them\n")
            _kill_group(p)
            rc = 124
        except KeyboardInterrupt:
            _kill_group(p)
            raise
        for t in pumps:
            t.join(timeout=5.0)
        if not rendezvous_port_was_taken(rc, port, err, out):
            break
        sys.stderr.write(f"{os.path.basename(script)}: rendezvous port {port} taken by another job, retrying ({attempt + 1}/3)\n")
    return rc, "".join(out)


def rendezvous_port_was_taken(rc, port, stderr_lines, stdout_lines):
    """True only for the one failure a re-launch can cure: torch.distributed.run could not bind ITS rendezvous store to
    the port picked for it (free_port closes its socket before the launcher binds — another job can take the port in
    between). c10d reports that as one line naming the port ("... port: 29500 ... EADDRINUSE ... address already in
    use"), before any rank has started, so the job has printed nothing. An 'address already in use' from anything else — a
    rank's own server, gloo's secondary sockets, a different port — is the job's own failure: it is passed on, not
    re-run (a blanket match re-ran whole multi-rank jobs up to three times)."""
    if rc in (0, 124) or any(ln.strip() for ln in stdout_lines):
        return False
    tag = f"port: {port}"
    return any(tag in ln and ("EADDRINUSE" in ln or "address already in use" in ln.lower()) for ln in stderr_lines)


def relay_json_line(stdout):
    """The last line of the job's stdout that is a JSON object (rank 0 prints exactly one)."""
    for line in reversed(stdout.splitlines()):
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            return line
    return None
