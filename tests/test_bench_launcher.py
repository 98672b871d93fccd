"""`python bench.py --gpus N` must start itself: with WORLD_SIZE unset and --gpus > 1 the script
re-executes under torch.distributed.run, one rank per GPU, and rank 0's JSON line is the last line of stdout.  No GPU
here: LS_BENCH_DRY_RUN makes the ranks rendezvous over gloo and stop before touching a device."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_spawns_its_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["LS_BENCH_DRY_RUN"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line == {"dry_run": True, "n_gpus": 2, "ranks_seen": 2, "steps": 4, "warmup": 1}


def test_bench_config_presets_name_baseline_configs_verbatim():
    sys.path.insert(0, ROOT)
    import bench
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        names = json.load(f)["configs"]
    assert [c["name"] for c in bench.BASELINE_CONFIGS] == names
