"""bench.py's launcher on a box without enough GPUs: it must refuse before starting ranks."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_launcher_refuses_without_devices():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("box has GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 2
    assert "--oversubscribe" in p.stderr
