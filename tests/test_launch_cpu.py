"""The N > 1 start-up path of bench.py on CPU: `python bench.py --gpus 2` with no launcher in the environment must
spawn its ranks itself (3dal_pytorch_amd/launch.py) from a parent that never touches a GPU, relay rank 0's one JSON
line and pass the ranks' exit code on. The hot path cannot run here, so the ranks run `--plumbing-only`: launcher,
process group (gloo), BoxGatherer / all_gather_boxes and the census, no kernels."""
import importlib
import json
import os
import subprocess
import sys

from _common import ROOT

launch = importlib.import_module("3dal_pytorch_amd.launch")
BENCH = os.path.join(ROOT, "bench.py")


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DAL3_BENCH_BACKEND="gloo", OMP_NUM_THREADS="2")
    env.update(extra)
    return env


def _strict(line):
    """the driver's view of the one stdout line: bounded, strictly parseable (no NaN / Infinity), contract keys present"""
    bench = importlib.import_module("bench")
    assert len(line.encode()) <= bench.LINE_CAP == 4096, len(line)

    def no_constants(c):
        raise AssertionError(f"non-strict JSON constant {c} in the bench line")
    rec = json.loads(line, parse_constant=no_constants)
    missing = [k for k in bench.REQUIRED if k not in rec]
    assert not missing, missing
    return rec


def test_compact_line_drops_optional_keys_until_it_fits_and_refuses_nan():
    """the line can never again outgrow the driver's parser. A record with the contract's keys and
    oversized optional ones comes out <= 4 KB with every required key kept; NaN is an error, not 'NaN' in the line"""
    bench = importlib.import_module("bench")
    rec = {k: 1 for k in bench.REQUIRED}
    rec["config"] = {"workload": "w" * 300}
    rec["roofline"] = {"kernel": "k", "bound": "mfma", "achieved": 138.3, "peak": 157.3, "unit": "TFLOP/s", "frac": 0.879, "traffic": None}
    rec["cpu_baseline"] = {"value": 80.0, "unit": "object-crops/s", "cores": 32, "kind": "port", "sample": "s" * 300}
    rec["gather_self_check"] = {"checks": [{"head": "static", "peer": i, "equal": True, "pad": "x" * 200} for i in range(40)]}
    rec["ms_per_step_per_rank"] = [25.123] * 8
    rec["rccl"] = {"backend": "nccl", "world_size": 8, "ranks_counted": 8}
    out = _strict(bench.compact_line(rec))
    assert out["dropped_for_size"] == ["gather_self_check"] and out["rccl"]["ranks_counted"] == 8
    import pytest
    with pytest.raises(ValueError):
        bench.compact_line(dict(rec, value=float("nan")))


def test_bench_without_enough_gpus_fails_cleanly():
    """no GPU here: a plain `bench.py --gpus 2` says what it needs and exits non-zero without starting anything"""
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True,
                         text=True, env=_env(), timeout=300)
    assert out.returncode != 0
    assert "needs 2 GPUs" in out.stderr, out.stderr[-1000:]
    assert out.stdout.strip() == ""


def test_eight_ranks_on_a_box_without_eight_gpus_exit_2_with_one_line():
    """what the driver's `bench.py --gpus 8` meets on a smaller box: exit code 2, ONE line on stderr, nothing on stdout —
    from the self-launching parent, and from a rank that a launcher started (LOCAL_RANK beyond the devices)"""
    out = subprocess.run([sys.executable, BENCH, "--gpus", "8"], capture_output=True, text=True, env=_env(), timeout=300)
    assert out.returncode == 2 and out.stdout.strip() == ""
    assert [ln for ln in out.stderr.splitlines() if ln.strip()] == ["bench.py: needs 8 GPUs, this machine shows 0"], out.stderr
    env = _env(RANK="7", LOCAL_RANK="7", WORLD_SIZE="8", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(launch.free_port()))
    out = subprocess.run([sys.executable, BENCH, "--gpus", "8"], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 2 and out.stdout.strip() == ""
    assert "bench.py: needs 8 GPUs, this machine shows 0" in out.stderr


def test_eight_ranks_c2_weak_and_c4_ragged_on_gloo():
    """the 8-rank job as the driver will start it (`bench.py --gpus 8`), rehearsed on gloo with the
    workloads' real item counts and rank -> range maps (bench.workload_shards — the function the GPU run shards with):
    C2 = 4096 crops per rank (weak), C4 = one segment of 64 static crops (8 per rank) + 4,531 dynamic track-frames
    (567 per rank, 562 on the last). Every rank takes part, the gathered boxes equal the single-process result, and
    rank 0's recomputation of rank 1's and rank 7's first rows equals what the gather delivered."""
    for config, want_heads in (("C2", [("static", 8 * 4096, [4096] * 8)]),
                               ("C4", [("static", 64, [8] * 8), ("dynamic", 4531, [567] * 7 + [562])])):
        out = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--plumbing-only", "--config", config],
                             capture_output=True, text=True, env=_env(OMP_NUM_THREADS="1"), timeout=900)
        assert out.returncode == 0, out.stderr[-3000:]
        lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, out.stdout
        rec = _strict(lines[0])
        assert rec["n_gpus"] == 8 and rec["gathered_ok"] is True
        assert {k: rec["rccl"][k] for k in ("backend", "world_size", "ranks_counted")} == {"backend": "gloo", "world_size": 8, "ranks_counted": 8}
        # round 6: which wires carried the gather (here: gloo, no HIP devices -> eight empty peer-access rows, no RCCL log)
        assert rec["rccl"]["transport"] == {"backend": "gloo", "peer_access": [""] * 8, "debug_dir": None}
        assert rec["scaling"] == ("strong" if config == "C4" else "weak")
        # ... and the STRONG record next to C2's weak `value`: ONE 4096-crop batch in contiguous ranges of 512;
        # C4 is strong scaling already
        if config == "C2":
            st = rec["strong"]
            assert (st["scaling"], st["items"], st["items_per_rank"]) == ("strong", 4096, [512] * 8)
            assert st["gathered_ok"] is True and st["vs_n1"] is None and st["unit"] == "object-crops/s"
        else:
            assert rec["strong"] is None
        heads = [(h["head"], h["items"], h["items_per_rank"]) for h in rec["config"]["heads"]]
        assert heads == want_heads, heads
        assert len(rec["ms_per_step_per_rank"]) == 8
        assert rec["gather_equals_single_rank"] is True
        peers = sorted((c["head"], c["peer"]) for c in rec["gather_self_check"]["checks"])
        assert peers == sorted((h, p) for h, _, _ in want_heads for p in (1, 7))
        assert all(c["equal"] and c["rows"] == 5 for c in rec["gather_self_check"]["checks"])


def test_rendezvous_retry_only_for_the_launchers_own_port():
    """only torch.distributed.run failing to bind the port picked for it is re-launched; any other 'address
    already in use' (a rank's own server, another port, a job that already printed) is the job's failure"""
    line = "[E] The server socket has failed to listen on any local network address. port: 29611, useIpv6: 0, code: -98, name: EADDRINUSE, message: address already in use\n"
    assert launch.rendezvous_port_was_taken(1, 29611, [line], [])
    assert not launch.rendezvous_port_was_taken(1, 29612, [line], [])                     # some other port
    assert not launch.rendezvous_port_was_taken(1, 29611, ["OSError: [Errno 98] Address already in use\n"], [])
    assert not launch.rendezvous_port_was_taken(1, 29611, [line], ['{"value": 1}\n'])      # the job had got going
    assert not launch.rendezvous_port_was_taken(0, 29611, [line], []) and not launch.rendezvous_port_was_taken(124, 29611, [line], [])


def test_bench_self_launches_two_ranks_and_relays_one_line():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--plumbing-only"], capture_output=True, text=True,
                         env=_env(), timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    rec = _strict(lines[0])
    assert rec["n_gpus"] == 2 and rec["plumbing_only"] and rec["gathered_ok"]
    assert {k: rec["rccl"][k] for k in ("backend", "world_size", "ranks_counted")} == {"backend": "gloo", "world_size": 2, "ranks_counted": 2}
    assert rec["rccl"]["transport"]["backend"] == "gloo" and rec["strong"] is None       # (the 37-item stub is split already)
    # what a real N > 1 line carries: every rank's own step time, and rank 0's recomputation of rank 1's first rows
    # (its inputs regenerated from their global indices, the replicated weight) equal to what the gather delivered
    assert len(rec["ms_per_step_per_rank"]) == 2
    assert rec["gather_equals_single_rank"] is True
    assert rec["gather_self_check"]["checks"] == [{"head": "stub", "peer": 1, "first_item": 19, "rows": 5, "equal": True}]


def test_two_ranks_c2_carry_the_strong_record_and_a_ragged_split():
    """the N > 1 line holds, next to the weak-scaling `value`, the same batch SPLIT over the ranks. World 2
    with C2's 4096 crops (2048 each) and, through --batch 37, a ragged split (19 + 18)."""
    for extra, want in ((["--config", "C2"], (4096, [2048, 2048])), (["--head", "static", "--batch", "37", "--config", "C2"], None)):
        out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--plumbing-only"] + extra, capture_output=True, text=True,
                             env=_env(), timeout=600)
        assert out.returncode == 0, out.stderr[-3000:]
        rec = _strict([ln for ln in out.stdout.splitlines() if ln.strip()][0])
        st = rec["strong"]
        assert st["scaling"] == "strong" and st["gathered_ok"] is True and st["value"] is None
        if want:
            assert (st["items"], st["items_per_rank"]) == want
    bench = importlib.import_module("bench")

    class A:
        head, batch = "dynamic", 37
    name, n, span = bench.strong_shards(A, 2)
    assert (name, n, span(0), span(1)) == ("dynamic", 37, (0, 19), (19, 18))
    A.batch = 0
    assert bench.strong_shards(A, 8)[1] == 1024 and bench.strong_shards(A, 8)[2](7) == (896, 128)


def test_a_failing_rank_fails_the_launcher():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--plumbing-only"], capture_output=True, text=True,
                         env=_env(DAL3_BENCH_FAIL_RANK="1"), timeout=600)
    assert out.returncode != 0


def test_under_a_launcher_the_process_is_a_rank():
    """what the driver does for N > 1: torch.distributed.run starts bench.py; it must not spawn again"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(launch.free_port()), BENCH, "--gpus", "2", "--plumbing-only"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=_env(), timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads(launch.relay_json_line(out.stdout))
    assert rec["rccl"]["ranks_counted"] == 2


def test_relay_picks_the_json_line():
    assert launch.relay_json_line("NCCL version 2.x\n{\"a\": 1}\ntrailing\n") == '{"a": 1}'
    assert launch.relay_json_line("nothing here\n") is None


def test_launcher_timeout_ends_the_whole_process_group():
    """a timeout that kills only torch.distributed.run orphans the ranks (GPUs and the rendezvous port stay
    taken). spawn_ranks starts the job in its own session and ends the group: no rank survives, exit code 124."""
    import time
    env = _env(DAL3_BENCH_HANG_RANK="1")
    t0 = time.time()
    rc, out = launch.spawn_ranks(BENCH, ["--gpus", "2", "--plumbing-only"], 2, need_gpus=False, env=env, timeout=45)
    assert rc == 124 and time.time() - t0 < 120
    time.sleep(1.0)
    left = subprocess.run(["pgrep", "-f", "bench.py --gpus 2 --plumbing-only"], capture_output=True, text=True).stdout.split()
    assert left == [], f"ranks left behind: {left}"
