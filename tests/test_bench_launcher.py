"""bench.py's launch contract without a GPU: a WORLD_SIZE that disagrees with --gpus is refused."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_refuses_world_size_mismatch():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "refusing" in r.stderr
