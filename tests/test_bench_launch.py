"""bench.py as a launcher: `--gpus N` without WORLD_SIZE starts its own ranks before anything touches the GPU, never
reports an n_gpus other than the world size it ran with, and refuses when the GPUs are not there."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(extra)
    return env


def test_refuses_more_ranks_than_gpus():
    import torch
    if torch.cuda.device_count() >= 4:
        pytest.skip("this box has the GPUs")
    res = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                         env=_env(), timeout=300)
    assert res.returncode == 2 and "GPU(s) visible" in res.stderr and not res.stdout.strip()


def test_world_size_mismatch_is_an_error():
    res = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                         env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), timeout=300)
    assert res.returncode != 0 and "WORLD_SIZE=1" in (res.stderr + res.stdout)


def test_failing_ranks_fail_the_launcher_with_labelled_stderr():
    """Self-spawned ranks: a rank that exits non-zero makes the parent exit non-zero, the siblings are terminated instead
    of being waited for, and every line of a child's stderr carries its rank.  (Without a GPU every rank stops at
    "bench.py needs an MI355X"; on a GPU box the WORLD_SIZE check of the workload is what fails.)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU: there the ranks would run")
    res = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                         env=_env(FIAT_AMD_BENCH_BACKEND="gloo"), timeout=600)
    assert res.returncode != 0 and not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert "[rank 0] " in res.stderr or "[rank 1] " in res.stderr
    assert "needs an MI355X" in res.stderr and "the other ranks have" in res.stderr


# ---- the guarded all-gather leg, rehearsed on CPU: every ordering of the ranks' deaths leaves ONE line and status != 0
REHEARSAL = os.path.join(ROOT, "tests", "native", "guard_rehearsal.py")


def _port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rehearse(launcher, world=2, **hooks):
    if launcher == "spawn":
        cmd = [sys.executable, "-c", "import sys; sys.path.insert(0, %r); import bench; "
               "sys.exit(bench.spawn_ranks(%d, cmd=[sys.executable, %r], need_gpus=False))" % (ROOT, world, REHEARSAL)]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
               "127.0.0.1", "--master-port", str(_port()), REHEARSAL]
    res = subprocess.run(cmd, capture_output=True, text=True, env=_env(**dict({"GUARD_TIMEOUT": "6"}, **hooks)), timeout=300)
    lines = [json.loads(ln) for ln in res.stdout.splitlines() if ln.startswith("{")]
    return res, lines


@pytest.mark.parametrize("launcher", ["spawn", "torchrun"])
def test_guarded_leg_success_prints_one_line(launcher):
    res, lines = _rehearse(launcher)
    assert res.returncode == 0, res.stderr[-2000:]
    assert len(lines) == 1 and lines[0]["allgather"] == {"sum": 2.0} and lines[0]["value"] == 1.0


@pytest.mark.parametrize("launcher", ["spawn", "torchrun"])
@pytest.mark.parametrize("stuck", [0, 1])
def test_guarded_leg_hung_rank_leaves_the_line_and_fails(launcher, stuck):
    """A rank that never joins the exchange, while rank 0 reaches the leg 3 s after its peer (its oracle check): rank 0's
    watchdog is the first to fire whichever rank hangs, so the line always says "no result within"."""
    res, lines = _rehearse(launcher, GUARD_HANG=str(stuck), GUARD_RANK0_EXTRA="3")
    assert res.returncode != 0, res.stderr[-2000:]
    assert len(lines) == 1, (res.stdout[-2000:], res.stderr[-2000:])
    assert lines[0]["value"] == 1.0 and "no result within 6" in lines[0]["allgather"]["error"]


@pytest.mark.parametrize("launcher", ["spawn", "torchrun"])
def test_guarded_leg_dead_peer_leaves_the_line_and_fails(launcher):
    """The peer dies inside the leg: rank 0, blocked in the collective, either sees the connection drop or is SIGTERMed by
    the launcher -- both end in the line with an error inside and a non-zero status."""
    res, lines = _rehearse(launcher, GUARD_CRASH="1")
    assert res.returncode != 0, res.stderr[-2000:]
    assert len(lines) == 1, (res.stdout[-2000:], res.stderr[-2000:])
    assert lines[0]["value"] == 1.0 and lines[0]["allgather"]["error"]


def test_guarded_leg_sigterm_inside_a_blocked_collective():
    """The pure launcher-kill ordering, without help from gloo: three ranks, rank 2 hangs OUTSIDE any collective (so no
    connection drops), rank 1 dies; the parent's SIGTERM reaches rank 0 while its main thread is blocked in all_reduce and
    the wake-up pipe's watcher thread prints the line."""
    res, lines = _rehearse("spawn", world=3, GUARD_CRASH="1", GUARD_HANG="2", GUARD_TIMEOUT="60")
    assert res.returncode != 0, res.stderr[-2000:]
    assert len(lines) == 1, (res.stdout[-2000:], res.stderr[-2000:])
    assert lines[0]["allgather"]["error"] and "rank 0 terminated during the all-gather leg" in res.stderr \
        or "Connection" in lines[0]["allgather"]["error"]


def test_guarded_leg_orderly_failure():
    res, lines = _rehearse("spawn", GUARD_RAISE="0")
    assert res.returncode != 0 and len(lines) == 1 and "exchange refused" in lines[0]["allgather"]["error"]


@pytest.mark.gpu
@pytest.mark.parametrize("stuck", [1, 0])
def test_stuck_allgather_leg_is_a_failure_not_a_success(stuck):
    """A rank that never joins the exchange: rank 0's watchdog (always the first to fire) still prints the compute-only line,
    with the error inside, and the job exits NON-ZERO -- a hung collective must not look like success to the driver."""
    res = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2000", "--check", "50",
                          "--no-cpu-baseline", "--allgather-timeout", "20"], capture_output=True, text=True,
                         env=_env(FIAT_AMD_BENCH_BACKEND="gloo", FIAT_AMD_BENCH_FORCE_GATHER_TIMEOUT=str(stuck)), timeout=900)
    assert res.returncode != 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (res.stdout[-2000:], res.stderr[-3000:])
    line = json.loads(lines[0])
    assert line["value"] > 0 and "no result within 20" in line["allgather"]["error"]
    assert "[rank " in res.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("hook", ["FIAT_AMD_BENCH_FORCE_GATHER_TIMEOUT", "FIAT_AMD_BENCH_FORCE_GATHER_CRASH"])
def test_torchrun_launch_with_a_failing_peer_keeps_the_line(hook):
    """The driver's launch line with rank 1 hanging / dying inside the exchange: torch.distributed.run SIGTERMs the siblings
    of the first failed rank -- the compute-only line must be on stdout regardless."""
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(_port()), BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--batch", "2000", "--check", "50", "--allgather-timeout", "20"], capture_output=True, text=True,
                         env=_env(FIAT_AMD_BENCH_BACKEND="gloo", **{hook: "1"}), timeout=900)
    assert res.returncode != 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (res.stdout[-2000:], res.stderr[-3000:])
    line = json.loads(lines[0])
    assert line["value"] > 0 and line["n_gpus"] == 2 and line["allgather"]["error"]
    if hook.endswith("TIMEOUT"):
        assert "no result within 20" in line["allgather"]["error"]


@pytest.mark.gpu
def test_allgather_verify_two_ranks():
    """--allgather-verify: both ranks tabulate the same requests and compare every gathered block with their own."""
    res = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "3000", "--check", "50",
                          "--no-cpu-baseline", "--allgather-verify"], capture_output=True, text=True,
                         env=_env(FIAT_AMD_BENCH_BACKEND="gloo"), timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    v = line["allgather"]["verify"]
    assert v["ok"] and v["requests"] == 3000 and all(v["one_shot_blocks_equal"]) and all(v["chunked_blocks_equal"])
    assert line["allgather"]["free_bytes_min_over_ranks"] > 0 and line["ms_per_step_cold"] > 0
    assert line["roofline"]["mfma"]["frac"] > 0


@pytest.mark.gpu
def test_bare_two_rank_launch_on_one_gpu():
    """FIAT_AMD_BENCH_BACKEND=gloo lets two ranks share the one GPU of this box: the parent spawns them, both tabulate their
    own block, the with-gather leg runs through fiat_amd/distributed.py, rank 0 prints ONE line with n_gpus = 2."""
    res = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4000",
                          "--allgather-chunks", "4"], capture_output=True, text=True,
                         env=_env(FIAT_AMD_BENCH_BACKEND="gloo"), timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["world_size"] == 2 and line["scaling"] == "weak"
    assert line["config"]["launch"] == "self-spawned ranks"
    assert line["value"] > 0 and line["max_rel_err_vs_oracle"] < 1e-12
    ag = line["allgather"]
    assert "error" not in ag, ag
    assert ag["impl"] == "torch" and ag["all_finite"] and ag["chunks"] == 4
    assert ag["gathered_bytes_per_gpu"] == 2 * 4000 * 4 * 20 * 23 * 8


@pytest.mark.gpu
def test_two_ranks_with_the_staging_ring():
    """Outputs too large to replicate travel through the ring of staging buffers (forced here on a small batch)."""
    res = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "3000", "--check", "100",
                          "--allgather-mode", "ring", "--allgather-chunks", "4"], capture_output=True, text=True,
                         env=_env(FIAT_AMD_BENCH_BACKEND="gloo"), timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    ag = line["allgather"]
    assert "error" not in ag, ag
    assert line["n_gpus"] == 2 and ag["mode"].startswith("staging ring") and ag["all_finite"] and ag["value_with_allgather"] > 0


@pytest.mark.gpu
def test_driver_style_torchrun_launch_two_ranks():
    """The driver's launch line for N > 1 -- python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ... -- rehearsed with two gloo ranks sharing this box's GPU: ONE line from rank 0."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "3000",
                          "--check", "200"], capture_output=True, text=True, env=_env(FIAT_AMD_BENCH_BACKEND="gloo"), timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["launch"] == "torch.distributed.run" and line["value"] > 0
    assert line["max_rel_err_vs_oracle"] < 1e-12 and "error" not in line.get("allgather", {})
