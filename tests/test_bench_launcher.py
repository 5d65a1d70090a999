"""`python bench.py --gpus N` without a launcher starts its own ranks: the spawn logic at N = 2 on CPU / gloo with
the GPU legs left out (QUIP_BENCH_SELFTEST=1), and the N = 1 / already-launched cases, which must not spawn."""
import json
import os
import subprocess
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_plain_python_gpus2_spawns_two_ranks_and_prints_one_line():
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                        env=_env(QUIP_BENCH_SELFTEST="1"), capture_output=True, text=True, timeout=300)
    assert pr.returncode == 0, pr.stderr[-2000:]
    lines = [l for l in pr.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, pr.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["max_over_ranks"] == 2.0                      # the MAX over both ranks arrived at rank 0
    assert "torch.distributed.run" in pr.stderr              # ... through the launcher bench.py started itself


def test_no_spawn_when_launched_or_single():
    sys.path.insert(0, ROOT)
    import bench
    one = types.SimpleNamespace(gpus=1)
    two = types.SimpleNamespace(gpus=2)
    saved = {k: os.environ.pop(k, None) for k in ("WORLD_SIZE", "RANK")}
    try:
        assert bench.respawn_if_needed(one, argv=[]) is None
        os.environ["WORLD_SIZE"], os.environ["RANK"] = "2", "0"
        assert bench.respawn_if_needed(two, argv=[]) is None      # torchrun / the driver already set the rank environment
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
