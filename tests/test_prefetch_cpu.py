"""CPU: ``run.BatchPrefetcher`` -- the collate thread that stands in for the reference's ``DataLoader(num_workers=12, pin_memory=True)``
(``T/run.py:111-124``): batches arrive in order, exactly once, built ahead of the consumer; a failure in the worker surfaces in the
consumer; closing early does not hang."""
import threading
import time

import numpy as np
import pytest
import torch

from idvs.morec_amd.run import BatchPrefetcher


def test_order_and_overlap():
    built = []

    def make(idx):
        built.append(idx[0])
        return torch.tensor(idx), None, (torch.ones(2), torch.zeros(1))

    batches = [[i, i + 100] for i in range(7)]
    f = BatchPrefetcher(make, batches, depth=2, pin=False)
    time.sleep(0.3)
    assert 2 <= len(built) <= 3          # depth 2 in the queue (+ one being offered): the worker runs AHEAD, and no further
    got = [(b, t[0].tolist()) for b, t in f]
    assert got == [(i, batches[i]) for i in range(7)] and built == list(range(7))
    f.close()
    assert not f.t.is_alive()


def test_worker_exception_reaches_the_consumer():
    def make(idx):
        if idx[0] == 2:
            raise ValueError("bad batch")
        return (torch.tensor(idx),)

    f = BatchPrefetcher(make, [[0], [1], [2], [3]], depth=1, pin=False)
    seen = []
    with pytest.raises(ValueError, match="bad batch"):
        for b, t in f:
            seen.append(b)
    assert seen == [0, 1]
    f.close()


def test_close_with_a_full_queue_does_not_hang():
    f = BatchPrefetcher(lambda idx: (torch.tensor(idx),), [[i] for i in range(100)], depth=1, pin=False)
    it = iter(f)
    next(it)
    t0 = time.time()
    f.close()
    assert time.time() - t0 < 3 and not f.t.is_alive()
    assert threading.active_count() < 50


def _make_for_workers(idx):
    a = torch.from_numpy(np.asarray(idx, dtype=np.int64))
    return a, a.float() * 2.0, torch.ones(len(idx), 3), (a.int(), None)       # nested tuple with a None entry, as the token packing has


def test_batches_from_worker_processes_arrive_in_order():
    """``run.py --collate_workers N``: one DataLoader "sample" is one whole collated batch (``batch_size=None``), built in worker processes
    (T/run.py:111-124 uses ``DataLoader(num_workers=12)``); order, ragged last batch and ``None`` entries survive the trip."""
    from idvs.morec_amd.run import _BatchSet
    batches = [list(range(i * 4, i * 4 + 4)) for i in range(9)] + [[100, 101]]
    loader = torch.utils.data.DataLoader(_BatchSet(_make_for_workers, batches), batch_size=None, shuffle=False, num_workers=2, prefetch_factor=2)
    got = list(loader)
    assert len(got) == len(batches)
    for idx, (a, b, c, pack) in zip(batches, got):
        assert a.tolist() == idx and b.tolist() == [2.0 * v for v in idx] and c.shape == (len(idx), 3)
        assert pack[0].dtype == torch.int32 and pack[0].tolist() == idx and pack[1] is None
