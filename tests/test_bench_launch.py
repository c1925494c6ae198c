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


import pytest  # noqa: E402


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_two_rank_line_on_one_gpu_carries_the_reduce_trace():
    """The N > 1 code path of bench.py end to end on one MI355X: two gloo ranks sharing cuda:0 (RCCL refuses two ranks on one device),
    pooled negatives, bucketed gradient reduction from the backward callbacks.  The line must say what ran (2 ranks, weak scaling,
    whole-job throughput) and carry the per-bucket issue / join trace a scaling run is read with."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--share-device", "--backend", "gloo", "--bert", "tiny", "--batch", "16", "--steps", "3",
                        "--warmup", "2", "--no-secondary", "--no-cpu-baseline", "--sweep"], capture_output=True, text=True, timeout=500, env=_clean_env())
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["world_size_observed"] == 2 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 32 and "pooled-negatives" in out["config"]["parallelism"]
    assert abs(out["value"] - 32 * 1e3 / out["ms_per_step"]) < 1e-2 * out["value"]
    tr = out["gradient_reduce_trace"]
    assert "error" not in tr and len(tr["buckets"]) >= 3            # recommender group, encoder layers, closing sweep
    issued = [b["issued_at_ms"] for b in tr["buckets"]]
    assert issued == sorted(issued) and tr["join_begin_ms"] >= issued[-1] and tr["join_end_ms"] >= tr["join_begin_ms"]
    assert out["config"]["gemm8p_reserve_cus"] == 0      # gloo ranks: no RCCL ring kernel to leave CUs for (train_step.TrainStep.reserve_cus)
    assert "exposed_join_ms" in tr and tr["pooled_scoring"]["Nc"] == 2 * 16 * 21          # scored against BOTH ranks' item vectors
    # --sweep (schema: INTEGRATION.md): over gloo on one device the RCCL-communicator leg is reported as skipped, the torch.distributed leg
    # runs with the reduction overlapped and not; every timed configuration carries its own trace
    sw = out["sweep"]
    assert any(c.get("comm") == "rccl" and "skipped" in c for c in sw)
    timed = [c for c in sw if "ms_per_step" in c]
    assert {(c["comm"], c["overlap_reduce"], c["reserve_cus"]) for c in timed} == {("torch.distributed", True, 0), ("torch.distributed", False, 0)}
    for c in timed:
        assert "error" not in c and c["ms_per_step"] > 0 and "error" not in c["trace"] and "exposed_join_ms" in c["trace"]
        assert abs(c["user_seq_per_s"] - 32 * 1e3 / c["ms_per_step"]) < 1e-2 * c["user_seq_per_s"]
    no_overlap = [c for c in timed if not c["overlap_reduce"]][0]["trace"]
    # without overlap nothing is issued from the backward pass: one closing sweep per arena, at the join
    assert len(no_overlap["buckets"]) <= 2 and all("issued_at_ms" in b for b in no_overlap["buckets"])


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_watchdog_prints_the_headline_alone():
    """N > 1: when the explanatory passes behind the timed region do not finish in MOREC_BENCH_WATCHDOG_S (here: a limit they cannot meet), rank 0
    prints the measured headline with `post_headline` saying so and every rank exits 0 -- a hang in a first real multi-GPU run must not take
    the measured number with it."""
    env = dict(_clean_env(), MOREC_BENCH_WATCHDOG_S="0.001")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--share-device", "--backend", "gloo", "--bert", "tiny", "--batch", "16", "--steps", "3",
                        "--warmup", "2", "--no-secondary", "--no-cpu-baseline"], capture_output=True, text=True, timeout=500, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["world_size_observed"] == 2 and out["value"] > 0 and "watchdog" in out["post_headline"]
    assert abs(out["value"] - 32 * 1e3 / out["ms_per_step"]) < 1e-2 * out["value"]
