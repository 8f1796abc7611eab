"""CPU: `python bench.py --gpus N` starts its own N ranks (the driver calls the plain form, not torch.distributed.run) and relays rank 0's
JSON line; the BASELINE.json configurations map to the documented per-GPU shards."""
import importlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_self_launch_two_ranks_over_gloo():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-launch"], env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec == {"selftest": True, "world": 2, "sum": 3.0}


def test_config_presets_follow_baseline_json():
    bench = importlib.import_module("bench")
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        assert len(json.load(f)["configs"]) == len(bench.CONFIGS) == 5
    a = bench.parse_args([])
    assert (a.model, a.batch, a.size, a.dtype, a.gpus) == ("runet", 16, 256, "f32", 1)          # the headline configuration
    a = bench.parse_args(["--config", "3"])
    assert (a.batch, a.size, a.dtype) == (4, 512, "bf16")
    a = bench.parse_args(["--config", "5"])
    assert (a.batch, a.size, a.dtype, a.graph) == (1, 1024, "fp16", True)
    a = bench.parse_args(["--config", "4"])
    assert a.model == "deeplab"
    a = bench.parse_args(["--config", "1", "--batch", "8"])
    assert (a.batch, a.size) == (8, 64)
