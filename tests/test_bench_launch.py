"""bench.py's launch path and shard arithmetic, on the CPU: `python bench.py --gpus 2` from a plain shell must start
its own rank processes (the parent never imports torch), shard the step's samples the way renderer.spp_shard does,
and relay rank 0's line.  --dry-run renders nothing (there is no CPU render path); it exercises the plumbing only."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import ray_tracing_amd as rt  # noqa: E402


def run_bench(*args):
    env = dict(os.environ)
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("LOCAL_RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("spp,world", [(1024, 1), (1024, 2), (1024, 8), (100, 8), (7, 8), (1000, 3)])
def test_shard_matches_renderer_and_partitions_the_samples(spp, world):
    got = [bench.shard(spp, r, world) for r in range(world)]
    assert got == [rt.spp_shard(spp, r, world) for r in range(world)]
    pos = 0
    for b, c in got:
        assert b == pos and c >= 0
        pos += c
    assert pos == spp
    assert max(c for _, c in got) - min(c for _, c in got) <= 1


def test_launcher_command_is_the_drivers_form():
    cmd = bench.launcher_command(4, ["--gpus", "4", "--steps", "2"], port=29999)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=4" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "2"] and cmd[-5].endswith("bench.py")


def test_parent_does_not_import_torch_before_launching():
    # the parent of --gpus N must not initialise anything GPU-related: torch is imported inside main() only, after the launch branch
    src = open(os.path.join(ROOT, "bench.py")).read()
    head, tail = src.split("def main():", 1)
    assert "\nimport torch" not in head and "\nfrom torch" not in head
    assert tail.index("self_launch(a, sys.argv[1:])") < tail.index("import torch")


def test_self_launch_two_ranks_strong_scaling_dry_run():
    r = run_bench("--gpus", "2", "--dry-run", "--spp", "1024")
    assert r["dry_run"] and r["n_gpus"] == 2 and r["scaling"] == "strong" and r["counts_ok"]
    assert r["ranges"] == [[0, 512], [512, 512]] and r["spp_per_step"] == 1024


def test_self_launch_uneven_shards_and_weak_flag():
    r = run_bench("--gpus", "3", "--dry-run", "--spp", "100")
    assert r["ranges"] == [[0, 33], [33, 33], [66, 34]] and r["counts_ok"]
    r = run_bench("--gpus", "2", "--dry-run", "--spp", "64", "--weak")
    assert r["scaling"] == "weak" and r["ranges"] == [[0, 64], [64, 64]] and r["spp_per_step"] == 128


def test_single_rank_runs_in_process():
    r = run_bench("--dry-run")
    assert r["n_gpus"] == 1 and r["ranges"] == [[0, 1024]]
