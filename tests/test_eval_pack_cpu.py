"""CPU: the vectorised packing of evaluation users (``data_utils/metrics.py::PackedEvalUsers``) against the per-user loop of the
reference's ``eval_model`` (``T/data_utils/metrics.py:92-102``: right-aligned input sequence, mask, target, history to be masked)."""
import numpy as np
import torch

from idvs.morec_amd.data_utils.metrics import PackedEvalUsers


def _loop(user_history, eval_seq, users, S):
    U = len(users)
    idx, lm = np.zeros((U, S), dtype=np.int64), np.zeros((U, S), dtype=np.float32)
    hmax = max(1, max(len(user_history[u]) for u in users))
    hist, target = np.full((U, hmax), -1, dtype=np.int32), np.zeros(U, dtype=np.int32)
    for r, u in enumerate(users):
        seq = eval_seq[u]
        toks = seq[:-1]
        idx[r, S - len(toks):] = toks
        lm[r, S - len(toks):] = 1
        h = np.asarray(user_history[u])
        hist[r, :len(h)] = h
        target[r] = seq[-1]
    return idx, lm, hist, target


def test_packed_eval_users_equals_the_loop():
    rng = np.random.default_rng(0)
    S, U = 7, 83
    eval_seq = {u: [int(v) for v in rng.integers(1, 100, rng.integers(1, S + 2))] for u in range(U)}
    hist = {u: torch.LongTensor(eval_seq[u][:-1] + ([int(rng.integers(1, 100))] if u % 5 == 0 else [])) for u in range(U)}
    users = [int(u) for u in rng.permutation(U)[:61]]
    p = PackedEvalUsers(hist, eval_seq, users, S)
    idx, lm, h, tgt = _loop(hist, eval_seq, users, S)
    assert np.array_equal(p.idx, idx) and np.array_equal(p.lm, lm) and np.array_equal(p.hist, h) and np.array_equal(p.target, tgt)
    i2, l2, h2, t2 = p.slice(10, 30)          # a chunk carries its own history width; dropped columns are padding only
    assert np.array_equal(i2, idx[10:30]) and np.array_equal(l2, lm[10:30]) and np.array_equal(t2, tgt[10:30])
    assert np.array_equal(h2, h[10:30, :h2.shape[1]]) and (h[10:30, h2.shape[1]:] == -1).all()


def test_packed_eval_users_edge_cases():
    p = PackedEvalUsers({0: [], 1: [3]}, {0: [5], 1: [3, 9]}, [0, 1], 4)      # a user with only a target: empty input, empty history
    assert p.idx.tolist() == [[0, 0, 0, 0], [0, 0, 0, 3]] and p.lm.sum() == 1 and p.target.tolist() == [5, 9]
    assert p.hist.tolist() == [[-1], [3]]
    e = PackedEvalUsers({}, {}, [], 4)
    assert e.idx.shape == (0, 4) and e.hist.shape == (0, 1)
