"""The path ``INTEGRATION.md`` §1 advertises, at world size 2: the reference's own loop -- ``DistributedDataParallel(model,
device_ids=[local_rank], output_device=local_rank, find_unused_parameters=True)`` (``T/run.py:148``), ``optim.AdamW`` over the two
parameter groups of ``T/run.py:150-162``, ``loss.backward()`` / ``optimizer.step()`` (``T/run.py:243-247``) -- over the drop-in
``Model``, whose forward / backward are the autograd shells around the HIP kernels.  Two ``gloo`` ranks share ``cuda:0`` (RCCL refuses
two ranks on one device; DDP's bucketing, hooks and unused-parameter bookkeeping are transport independent).

* rank-local negatives (the reference's arithmetic): DDP averages the ranks' gradients, so two ranks x B is the single process
  that scores each half batch against its OWN items and averages the two losses;
* pooled negatives (``--pool_negatives``): two ranks x B is the single-process step at batch 2 B.

The second test drives ``python -m idvs.morec_amd.run`` itself under ``torch.distributed.run --nproc-per-node 2`` (both optimisation
paths)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _groups(model, lr=1e-3, flr=5e-4):
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    return [{"params": [p for n, p in named if "bert_model" in n], "lr": flr, "weight_decay": 0.02},          # T/run.py:150-162
            {"params": [p for n, p in named if "bert_model" not in n], "lr": lr, "weight_decay": 0.01}]


def _freeze_pooler(model):
    for n, p in model.named_parameters():      # T/run.py:67-75: the pooler is frozen (and unused: find_unused_parameters=True)
        if ".pooler." in n:
            p.requires_grad = False


def _worker(rank, world, port, q, dtype, pool):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    from test_train_step_ddp_gpu import _build
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model, ids, lm, content = _build(dtype, "text")
    model.pool_negatives = bool(pool)
    _freeze_pooler(model)
    ddp = DDP(model, device_ids=[0], output_device=0, find_unused_parameters=True)      # T/run.py:148
    opt = torch.optim.AdamW(_groups(model))
    B = ids.shape[0] // world
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda")
    my_ids, my_lm = ids[rank * B:(rank + 1) * B], lm[rank * B:(rank + 1) * B]
    flat = my_ids.reshape(-1)
    losses = []
    for _ in range(2):
        opt.zero_grad()
        loss = ddp(dev(flat), dev(content[flat]), dev(my_lm), 0)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    tot = torch.tensor(losses, dtype=torch.float64)
    dist.all_reduce(tot)                      # mean over ranks of what each rank's forward returned
    tot /= world
    if rank == 0:
        q.put((tot.tolist(), {k: v.detach().float().cpu().numpy() for k, v in model.state_dict().items()}))
    dist.barrier()
    dist.destroy_process_group()


def _single(dtype, pool):
    from test_train_step_ddp_gpu import _build
    model, ids, lm, content = _build(dtype, "text")
    _freeze_pooler(model)
    opt = torch.optim.AdamW(_groups(model))
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda")
    B = ids.shape[0] // 2
    losses = []
    for _ in range(2):
        opt.zero_grad()
        if pool:      # one process, batch 2 B
            flat = ids.reshape(-1)
            loss = model(dev(flat), dev(content[flat]), dev(lm), 0)
        else:         # each half batch against its own items, averaged (what DDP's gradient MEAN over two ranks computes)
            loss = 0.0
            for r in range(2):
                flat = ids[r * B:(r + 1) * B].reshape(-1)
                loss = loss + 0.5 * model(dev(flat), dev(content[flat]), dev(lm[r * B:(r + 1) * B]), 0)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    return model, losses


@pytest.mark.timeout(600)
@pytest.mark.parametrize("dtype,pool", [("fp32", False), ("fp32", True), ("bf16", True)])
def test_ddp_wrapped_model_two_ranks_equal_single_process(dtype, pool):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + os.getpid() % 90
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, dtype, pool)) for r in range(2)]
    for pr in procs:
        pr.start()
    losses2, sd2 = q.get(timeout=500)
    for pr in procs:
        pr.join(120)
        assert pr.exitcode == 0
    model, losses1 = _single(dtype, pool)
    tol_l = 2e-5 if dtype == "fp32" else 2e-2
    assert all(abs(a - b) < tol_l * max(1.0, abs(b)) for a, b in zip(losses2, losses1)), (losses2, losses1)
    worst, lr = 0.0, 1e-3
    for k, v in model.state_dict().items():
        if "pooler" in k or k.endswith(("key.bias", "w_K.bias")):
            continue      # key biases: the true gradient is zero (softmax shift invariance), Adam amplifies the rounding noise
        worst = max(worst, float(np.abs(v.detach().float().cpu().numpy() - sd2[k]).max()))
    # two Adam steps move a weight by <= 2 lr; sign flips of eps-dominated gradient elements may cost a fraction of that
    assert worst < (0.2 * lr if dtype == "fp32" else 2.5 * lr), worst
    print(f"DDP drop-in, {dtype}, pooled={pool}: 2-rank {losses2} vs single {losses1}; worst param diff {worst:.2e}")


@pytest.mark.timeout(900)
@pytest.mark.parametrize("fused", [False, True])
def test_run_driver_under_torchrun_two_ranks(fused, tmp_path):
    """``python -m idvs.morec_amd.run`` under ``torch.distributed.run --nproc-per-node 2`` (the reference's launcher shape,
    ``T/train_bert_base.py:40-50``): process-group set-up, DistributedSampler shards, DDP wrap (``run.py`` wraps only when the world is
    larger than one) or the fused step with pooled negatives, the all-gather of eval metrics, rank-0 checkpointing -- end to end over gloo
    on one device."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MOREC_DIST_BACKEND="gloo", MOREC_DEVICE_INDEX="0", PYTHONPATH=ROOT + os.pathsep + env.get("PYTHONPATH", ""))
    port = 29500 + (os.getpid() + (7 if fused else 0)) % 150
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "idvs.morec_amd.run",
           "--synthetic", "400", "--synthetic_items", "200", "--item_tower", "modal", "--bert_model_load", "bert_tiny",
           "--freeze_paras_before", "0", "--batch_size", "16", "--embedding_dim", "64", "--lr", "1e-3", "--fine_tune_lr", "1e-4",
           "--epoch", "1", "--max_steps", "5", "--collate_workers", "0", "--checkpoint_root", str(tmp_path / "ckpt"), "--pool_negatives"]
    if fused:
        cmd.append("--fused_step")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=800, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    log = r.stdout + r.stderr
    mt = re.search(r"epoch 1: (\d+) steps, mean loss ([0-9.]+)", log)
    assert mt and int(mt.group(1)) == 5 and np.isfinite(float(mt.group(2))), log[-2000:]
    assert re.search(r"max eval Hit10 [0-9.]+", log)
