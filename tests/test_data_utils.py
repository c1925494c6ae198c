"""CPU tests: the product's preprocessing / collation against goldens captured from the reference."""
import os

import numpy as np
import torch

from idvs.morec_amd.data_utils import BuildTrainDataset, collate_train_batch, read_behaviors, read_news
from morec_oracle import bookkeeping as bk


def test_read_behaviors_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "g2_read_behaviors.npz"))
    a, b, c = read_news(os.path.join(golden_dir, "g2_items.tsv"))
    item_num, id2dic, tr, va, te, hv, ht, name2id, pop = read_behaviors(os.path.join(golden_dir, "g2_users.tsv"), a, b, c,
                                                                         int(g["S"]), int(g["min_seq_len"]), None)
    assert item_num == int(g["item_num"]) and len(tr) == int(g["n_users"])
    assert np.array_equal(np.asarray(pop, dtype=np.float64), g["pop"])          # bit-exact float64
    assert name2id == dict(zip(g["names"].tolist(), g["name_ids"].tolist()))
    for u in range(len(tr)):
        assert np.array_equal(tr[u], g[f"train.{u}"]) and np.array_equal(va[u], g[f"valid.{u}"])
        assert np.array_equal(te[u], g[f"test.{u}"])
        assert np.array_equal(hv[u].numpy(), g[f"hv.{u}"]) and np.array_equal(ht[u].numpy(), g[f"ht.{u}"])


def test_train_dataset_and_vector_collate(golden_dir):
    g = np.load(os.path.join(golden_dir, "g1_g4_id_tower.npz"))
    for case in "abde":
        ids, lm = g[f"{case}.ids"], g[f"{case}.log_mask"]
        S = int(g[f"{case}.S"])
        u2seq = {u: [int(v) for v in ids[u][-(int(lm[u].sum()) + 1):]] for u in range(ids.shape[0])}
        content = np.arange(int(g[f"{case}.item_num"]) + 1)
        ds = BuildTrainDataset(u2seq, content, int(g[f"{case}.item_num"]), S, use_modal=False)
        for u in range(ids.shape[0]):
            i, it, m = ds[u]
            assert np.array_equal(i.numpy(), ids[u]) and np.array_equal(m.numpy(), lm[u]) and np.array_equal(it.numpy(), ids[u])
            oi, om = bk.collate_train_sample(u2seq[u], S)
            assert np.array_equal(oi, ids[u]) and np.array_equal(om, lm[u])
        bi, bit, bm = collate_train_batch(u2seq, list(range(ids.shape[0])), content, S, False)
        assert np.array_equal(bi.numpy(), ids) and np.array_equal(bm.numpy(), lm)
        table = np.arange((int(g[f"{case}.item_num"]) + 1) * 4).reshape(-1, 4)
        _, bit2, _ = collate_train_batch(u2seq, list(range(ids.shape[0])), table, S, True)
        assert np.array_equal(bit2.numpy(), table[ids])


def test_token_packing_matches_python_loop():
    """Unpadded-layout bookkeeping (engine.token_packing) against an explicit loop: ragged titles, a full-length title and the
    all-[PAD] padding item (keeps its first position)."""
    import numpy as np
    import torch
    from idvs.morec_amd.engine import token_packing
    rng = np.random.default_rng(5)
    Nc, T = 37, 30
    lens = rng.integers(0, T + 1, Nc)
    lens[0], lens[1], lens[2] = 0, T, 1
    mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.int64)
    cu, tok = token_packing(torch.from_numpy(mask))
    ref_cu, ref_tok = [0], []
    for s in range(Nc):
        L = max(int(lens[s]), 1)
        ref_tok += [s * T + t for t in range(L)]
        ref_cu.append(ref_cu[-1] + L)
    assert cu.dtype == torch.int32 and tok.dtype == torch.int32
    assert cu.tolist() == ref_cu and tok.tolist() == ref_tok


def test_collate_bce_batch_bookkeeping():
    """BCE-variant batch (bce_text/main-end2end/data_utils/dataset.py:25-49): left padding, one negative per INPUT position, never
    an item of the user's own sequence, zeros on the padding and on the last slot; log_mask has len(seq) - 1 ones."""
    import numpy as np
    from idvs.morec_amd.data_utils import collate_bce_batch
    u2seq = {0: [3, 4, 5], 1: [7, 8, 9, 10, 11], 2: [2, 6, 2, 6, 9, 1]}
    S, item_num = 5, 12
    items, lm = collate_bce_batch(u2seq, [0, 1, 2], np.arange(item_num + 1), S, item_num, False, np.random.default_rng(3))
    assert items.shape == (3, S + 1, 2) and lm.shape == (3, S)
    for r, u in enumerate([0, 1, 2]):
        seq = u2seq[u]
        n = len(seq)
        assert items[r, :, 0].tolist() == [0] * (S + 1 - n) + seq
        neg = items[r, :, 1].tolist()
        assert neg[:S + 1 - n] == [0] * (S + 1 - n) and neg[-1] == 0
        assert all(1 <= v <= item_num and v not in seq for v in neg[S + 1 - n:-1])
        assert lm[r].tolist() == [0.0] * (S + 1 - n) + [1.0] * (n - 1)
