"""bench.py --gpus N without a launcher around it must START N ranks (one process per GPU over RCCL), or fail loudly -- never
measure one GPU and report it under another N (VERDICT r02 missing #2)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launcher_command_is_one_process_per_gpu_on_localhost():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launcher_command(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]      # the ranks see the caller's arguments


def test_more_gpus_than_the_node_has_fails_loudly():
    """No GPU in this container: `--gpus 2` must exit non-zero with a message, and print no JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert "needs 2 visible GPUs" in r.stderr and "nothing was measured" in r.stderr
    assert "{" not in r.stdout


def test_world_size_must_match_gpus():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=4" in (r.stderr + r.stdout)
