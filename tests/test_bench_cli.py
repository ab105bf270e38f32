"""bench.py launcher logic (CPU, no GPU call is reached): `--gpus N` never silently runs on fewer ranks."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_gpus_without_launcher_refuses_when_devices_are_missing():
    """No launcher, --gpus 2, fewer than 2 visible GPUs: non-zero exit and no JSON line (round 1 printed n_gpus=1)."""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("2+ GPUs visible: the self-launch would really start")
    cp = run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert cp.returncode != 0
    assert "only" in cp.stderr and "GPU" in cp.stderr
    assert "n_gpus" not in cp.stdout


def test_world_size_mismatch_is_refused():
    """A launcher that started 1 rank for --gpus 4 must not produce a line."""
    cp = run(["--gpus", "4", "--steps", "1", "--warmup", "0"],
             {"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_PORT": "29999", "MASTER_ADDR": "127.0.0.1"})
    assert cp.returncode != 0
    assert "WORLD_SIZE=1" in (cp.stderr + cp.stdout)
    assert "n_gpus" not in cp.stdout
