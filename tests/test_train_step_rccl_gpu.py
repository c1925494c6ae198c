"""-m gpu: the RCCL code paths of the data-parallel step EXECUTED on one MI355X.  A one-rank ``nccl`` (= RCCL) process group
makes every collective the identity, so ``TrainStep(force_collectives=True)`` -- which then issues the pooled exchange
(``all_gather_into_tensor`` x 2), the fp32 ``reduce_scatter_tensor`` of dE, the bucketed ``all_reduce(async_op=True)`` from the
backward-pass callbacks and the closing sweep -- must reproduce the plain single-process step: same losses, same parameters
after two AdamW steps.  The same through the C-ABI's own communicator (``morec_comm_*``: ncclAllGather / ncclReduceScatter /
ncclAllReduce on the compute stream).  Reference: DDP's NCCL all-reduce, ``T/run.py:148,321``; exchange: SURVEY.md §8(e).
Each case runs in a child process (the process group must not leak into the other tests of the session)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(port, q, dtype, tower, comm, overlap):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        os.environ["MOREC_OVERLAP_REDUCE"] = "1" if overlap else "0"
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch.distributed as dist
        from test_train_step_ddp_gpu import _build
        from idvs.morec_amd.train_step import TrainStep
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        assert dist.get_backend() == "nccl"
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda")   # noqa: E731
        out = {}
        for name, kw in (("plain", {}), ("coll", dict(force_collectives=True, comm=comm))):
            model, ids, lm, content = _build(dtype, tower)
            ts = TrainStep(model, lr=1e-3, fine_tune_lr=5e-4, l2_weight=0.01, fine_tune_l2_weight=0.02, pool_negatives=True, **kw)
            flat = ids.reshape(-1)
            items = flat if tower == "id" else content[flat]
            losses = []
            for _ in range(2):
                loss = ts.step(dev(flat), dev(items), dev(lm))
                losses.append(float(ts.global_loss(loss)))
            torch.cuda.synchronize()
            out[name] = (losses, {k: v.detach().float().cpu().numpy() for k, v in model.state_dict().items()}, len(ts._reduced),
                         sorted(str(k) for k in ts.buckets), ts.comm is not None)
        dist.destroy_process_group()
        q.put(("ok", out))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put(("err", f"{type(e).__name__}: {e}\n{traceback.format_exc()}"))


@pytest.mark.timeout(600)
@pytest.mark.parametrize("tower,dtype,comm,overlap", [("text", "fp32", "", True), ("text", "bf16", "", True), ("text", "fp32", "", False),
                                                      ("text", "fp32", "rccl", True), ("text", "bf16", "rccl", True),
                                                      ("id", "fp32", "", True), ("swin", "fp32", "rccl", True)])
def test_one_rank_rccl_collectives_are_the_identity(tower, dtype, comm, overlap):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    pr = ctx.Process(target=_worker, args=(29700 + os.getpid() % 200, q, dtype, tower, comm, overlap))
    pr.start()
    status, out = q.get(timeout=500)
    pr.join(120)
    assert status == "ok", out
    (l0, sd0, n0, _, c0), (l1, sd1, n1, buckets, c1) = out["plain"], out["coll"]
    assert n0 == 0 and not c0
    assert c1 == (comm == "rccl")
    if tower == "text":
        assert n1 == (3 + 1 + 1 if overlap else 2), n1       # 3 layer buckets + head + closing sweep / one sweep per group
    else:
        assert n1 >= 1
    # a one-rank SUM is the identity: the only difference is the fp32 hand-over of dE (pooled path) instead of the compute dtype
    tol = 1e-6 if dtype == "fp32" else 2e-2
    assert all(abs(a - b) <= tol * max(1.0, abs(a)) for a, b in zip(l0, l1)), (l0, l1)
    # bf16: every sum of the step has a fixed order -> the two runs agree to the last bit or two.  fp32 parity mode: its weight
    # gradients are split-K fp32 ATOMIC sums (order varies from launch to launch), and two Adam steps turn a last-bit difference of an
    # eps-dominated gradient element into a fraction of 2 lr -- the bound of tests/test_train_step_ddp_gpu.py (key biases excluded
    # there too: their true gradient is zero, softmax shift invariance)
    lr = 1e-3
    worst = max(float(np.abs(sd0[k] - sd1[k]).max()) for k in sd0
                if "pooler" not in k and not k.endswith(("key.bias", "k_proj.bias", "w_K.bias")))
    assert worst <= (0.2 * lr if dtype == "fp32" else 1e-6), worst
    print(f"{tower} {dtype} comm={comm or 'torch.distributed(nccl)'} overlap={overlap}: losses {l0} vs {l1}; worst parameter difference {worst:.2e}; "
          f"{n1} reduced slices, buckets {buckets}")


def test_morec_comm_one_rank_semantics():
    """morec_comm_* directly on a one-rank communicator: all-gather / reduce-scatter / all-reduce are copies, stream-ordered."""
    from idvs.morec_amd.comm import MorecComm
    c = MorecComm(rank=0, world=1)
    x = torch.randn(1000, 7, device="cuda")
    assert torch.equal(c.all_gather(x), x)
    assert torch.equal(c.all_gather(x.to(torch.bfloat16)), x.to(torch.bfloat16))
    assert torch.equal(c.reduce_scatter_sum(x), x)
    y = x.clone()
    c.all_reduce_sum_(y)
    assert torch.equal(y, x)
    c.close()
