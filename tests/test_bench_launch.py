"""bench.py's launch plumbing on the CPU (no GPU): `python bench.py --gpus 2` must start its two ranks itself
(torch.distributed.run children of a parent that never touches a GPU), rendezvous on 127.0.0.1, reduce the tile-sharded
frame to rank 0 and print ONE JSON line whose n_gpus is the number of ranks the communicator formed."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)                                   # as the driver calls it: no launcher environment
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                       # rank 0 prints one line, the other ranks none
    return json.loads(lines[0])


def test_gpus_2_without_a_launcher_spawns_its_own_ranks():
    out = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dryrun-cpu"])
    assert out["n_gpus"] == 2 and out["dryrun"] is True and out["steps"] == 3
    assert out["frame_assembled"] is True                  # both shards arrived on rank 0 through the reduce
    assert out["config"]["parallelism"] == "tile-shard x2" and out["scaling"] == "strong"


def test_gpus_1_runs_in_process_and_three_ranks_partition_the_frame():
    assert _run(["--dryrun-cpu", "--steps", "1"])["n_gpus"] == 1
    out = _run(["--gpus", "3", "--steps", "1", "--dryrun-cpu", "--width", "100", "--height", "70"])   # clipped tiles
    assert out["n_gpus"] == 3 and out["frame_assembled"] is True


def test_the_parent_of_a_multi_rank_run_never_imports_torch():
    """The spawning parent must not have initialised a GPU (an exec/fork from such a process takes the box down):
    it leaves main() before `import torch`."""
    src = open(BENCH).read()
    spawn = src.index("sys.exit(spawn_ranks(")
    assert spawn < src.index("import torch  #")
    head = src[:src.index("def main(")]
    assert "import torch" not in head                      # nothing at module level either


def test_usable_cpus_is_the_thread_count_the_cpu_baseline_reports():
    sys.path.insert(0, ROOT)
    import bench
    n = bench.usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
