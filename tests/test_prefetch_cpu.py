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


def _maker(n_users=37, n_items=50, S=6, T=5, attrs=(("title", 0, 10),), seed=0):
    from idvs.morec_amd.run import BatchMaker
    rng = np.random.default_rng(seed)
    users_train = {u: [int(v) for v in rng.integers(1, n_items + 1, int(rng.integers(2, S + 2)))] for u in range(n_users)}
    width = sum(w for _, _, w in attrs)
    content = np.zeros((n_items + 1, width), dtype=np.int64)
    for _, a0, aw in attrs:
        h = aw // 2
        lens = rng.integers(1, h + 1, n_items)
        valid = np.arange(h)[None, :] < lens[:, None]
        content[1:, a0:a0 + h] = np.where(valid, rng.integers(1, 99, (n_items, h)), 0)
        content[1:, a0 + h:a0 + aw] = valid
    return BatchMaker(list(users_train), users_train, content, S, True, text_attrs=list(attrs), pin=False), users_train, content


def test_batch_maker_pickles_and_builds_one_packing_per_attribute():
    """``run.BatchMaker`` (round 5): the host side of a batch as a picklable object over numpy tables (DataLoader workers under fork,
    forkserver or spawn); with several text attributes (``--news_attributes title,abstract``) one unpadded-layout packing per attribute."""
    import pickle
    from idvs.morec_amd import engine
    mk, users_train, content = _maker(attrs=(("title", 0, 10), ("abstract", 10, 16)))
    mk2 = pickle.loads(pickle.dumps(mk))
    for m in (mk, mk2):
        ids, items, lm, pack = m([0, 5, 9])
        assert ids.shape == (3, 7) and items.shape == (3, 7, 26) and lm.shape == (3, 6)
        assert isinstance(pack, tuple) and len(pack) == 2 and all(len(p) == 4 for p in pack)
        rows = items.view(-1, 26)
        for (name, a0, aw), p in zip(m.text_attrs, pack):
            h = aw // 2
            want = engine.token_packing_host(rows[:, a0 + h:a0 + aw], rows[:, a0:a0 + h], pin=False)
            assert all(torch.equal(x, y) for x, y in zip(p, want))
            lens = (rows[:, a0 + h:a0 + aw] != 0).sum(1).clamp(min=1)
            assert int(p[0][-1]) == int(lens.sum()) and p[1].numel() == int(lens.sum())
    single, *_ = _maker()
    pack1 = single([1, 2])[3]
    assert len(pack1) == 4 and pack1[0].dtype == torch.int32          # one attribute: the packing tuple itself, as before


def test_persistent_loader_follows_the_epoch_sampler():
    """One DataLoader for the whole run (``persistent_workers``): a sample key is (epoch, b) and the WORKER derives the epoch's index
    batches (``epoch_batches``: torch's DistributedSampler + batching), so batches match the in-process collate for every epoch, in order,
    without anything being sent to the workers between epochs."""
    from idvs.morec_amd.data_utils import epoch_batches
    from idvs.morec_amd.run import _EpochBatchSet, _EpochSampler, _with_next
    mk, users_train, content = _maker()
    sampler = _EpochSampler()
    loader = torch.utils.data.DataLoader(_EpochBatchSet(mk, len(users_train), 8, 2, 1), batch_size=None, sampler=sampler, num_workers=2,
                                         prefetch_factor=2, persistent_workers=True, multiprocessing_context="fork")
    for ep in (1, 2, 5):
        batches = epoch_batches(len(users_train), 8, 2, 1, ep)
        sampler.set_epoch(ep, len(batches))
        got = list(loader)
        assert len(got) == len(batches)
        for idx, (ids, items, lm, pack) in zip(batches, got):
            w_ids, w_items, w_lm, w_pack = mk(idx)
            assert torch.equal(ids, w_ids) and torch.equal(items, w_items) and torch.equal(lm, w_lm)
            assert all(torch.equal(x, y) for x, y in zip(pack, w_pack))
    # an epoch cut short (--max_steps) leaves the loader usable
    sampler.set_epoch(7, 3)
    it = iter(loader)
    next(it)
    del it
    sampler.set_epoch(8, 2)
    assert len(list(loader)) == 2
    pairs = list(_with_next(iter([10, 11, 12])))
    assert pairs == [(10, 11), (11, 12), (12, None)] and list(_with_next(iter([]))) == []
