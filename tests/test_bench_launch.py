"""CPU: `python bench.py --gpus N` can never silently be a one-rank run.  The N > 1 path self-launches N
ranks through torch.distributed.run when no launcher environment is present (reference launcher: T/train_bert_base.py:40-50),
reports the world size the ranks actually observed, and refuses a launcher that started a different number of ranks."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return env


def test_gpus2_self_launches_two_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check", "--backend", "gloo"], capture_output=True, text=True,
                       timeout=300, env=_clean_env())
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["ranks_joined"] == 2 and out["launch_check"] is True


def test_world_size_mismatch_fails_loudly():
    env = dict(_clean_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert "WORLD_SIZE=1" in (r.stderr + r.stdout)
