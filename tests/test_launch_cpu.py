"""`python bench.py --gpus N` must start its own ranks.  The launch logic lives in
fedmlp_amd/launch.py; tests/launch_probe.py runs it with gloo on CPU, world size 2."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "tests", "launch_probe.py")


def _env():
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    return e


def test_self_launch_world2_prints_one_json_line():
    r = subprocess.run([sys.executable, PROBE, "--gpus", "2"], capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["sum"] == 3.0 and out["local_rank"] == "0"


def test_self_launch_propagates_failure():
    r = subprocess.run([sys.executable, PROBE, "--gpus", "2", "--fail-rank", "1"], capture_output=True, text=True,
                       env=_env(), timeout=300)
    assert r.returncode != 0


def test_single_rank_needs_no_launcher():
    r = subprocess.run([sys.executable, PROBE, "--gpus", "1"], capture_output=True, text=True, env=_env(), timeout=120)
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_bench_uses_the_launcher_before_touching_the_gpu():
    """bench.py must branch into spawn_ranks before any torch.cuda call and must not assert on WORLD_SIZE."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def main():"):]
    assert "spawn_ranks(" in body
    assert body.index("spawn_ranks(") < body.index("torch.cuda.")
    assert "assert world == args.gpus" not in src


def test_rank_command_is_the_drivers_command():
    from fedmlp_amd.launch import rank_command
    cmd = rank_command("bench.py", ["--gpus", "4", "--steps", "5"], 4, port=29511)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=4" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-5:] == ["bench.py", "--gpus", "4", "--steps", "5"]
