"""Integer bookkeeping of the in-batch loss -- numpy restatement (bit-exact target).

TEST INFRASTRUCTURE ONLY (see package docstring).  Symbols: B users, S = max_seq_len,
Nc = B*(S+1) item slots (logit columns), Nr = B*S logit rows.
"""
from __future__ import annotations

import numpy as np


def collate_train_sample(seq, max_seq_len: int):
    """One training sample.  Follows ``T/data_utils/dataset.py:24-36``
    (``BuildTrainDataset.__getitem__``): left-pad the id sequence with item 0 to S+1 slots;
    ``log_mask = [0]*pad + [1]*(len-1)`` (length S).

    Returns (ids int64[S+1], log_mask float32[S]).
    """
    L = max_seq_len + 1
    seq = list(seq)
    assert 1 <= len(seq) <= L
    pad = L - len(seq)
    ids = np.asarray([0] * pad + seq, dtype=np.int64)
    log_mask = np.asarray([0] * pad + [1] * (len(seq) - 1), dtype=np.float32)
    return ids, log_mask


def ce_labels(bs: int, max_seq_len: int) -> np.ndarray:
    """Target column of every logit row.  ``T/model/model.py:45-48``:
    ``ce_label[i*S + (j-1)] = i*S + i + j`` for j = 1..S, i.e. column ``i*(S+1) + j``."""
    i = np.repeat(np.arange(bs, dtype=np.int64), max_seq_len)
    j = np.tile(np.arange(1, max_seq_len + 1, dtype=np.int64), bs)
    return i * (max_seq_len + 1) + j


def column_valid(log_mask: np.ndarray) -> np.ndarray:
    """``T/model/model.py:51-52``: a column (item slot) is valid iff
    ``cat(log_mask, ones[B,1], dim=1).view(-1) != 0``.  Returns bool[Nc]."""
    bs = log_mask.shape[0]
    ext = np.concatenate([log_mask, np.ones((bs, 1), dtype=log_mask.dtype)], axis=1)
    return ext.reshape(-1) != 0


def reject_mask(sample_items_id: np.ndarray, bs: int, max_seq_len: int,
                pool_ids: np.ndarray | None = None, col_offset: int = 0) -> np.ndarray:
    """No-false-negative mask.  ``T/model/model.py:54-63``: for user i every column whose item id
    is one of user i's S+1 slot ids (padding id 0 included) is rejected, except that for row j
    of user i the positive column ``i*(S+1)+j+1`` is restored.

    ``pool_ids``/``col_offset`` generalise to a pooled negative set: columns are
    ``pool_ids`` (default: the local ids) and the local positives sit at ``col_offset + ...``.
    Returns bool[B, S, Ncols]; True = overwritten with -1e4.
    """
    S = max_seq_len
    ids = np.asarray(sample_items_id).reshape(bs, S + 1)
    cols = ids.reshape(-1) if pool_ids is None else np.asarray(pool_ids).reshape(-1)
    member = (cols[None, None, :] == ids[:, :, None]).any(axis=1)  # [B, Ncols]
    mask = np.repeat(member[:, None, :], S, axis=1).copy()          # [B, S, Ncols]
    ii, jj = np.meshgrid(np.arange(bs), np.arange(S), indexing="ij")
    mask[ii, jj, col_offset + ii * (S + 1) + jj + 1] = False
    return mask


def valid_rows(log_mask: np.ndarray) -> np.ndarray:
    """``T/model/model.py:65``: indices of rows with ``log_mask != 0`` (row-major over [B, S])."""
    return np.nonzero(np.asarray(log_mask).reshape(-1) != 0)[0].astype(np.int64)


def log_pop(pop_prob_list: np.ndarray, sample_items_id: np.ndarray) -> np.ndarray:
    """``T/model/model.py:32-33``: ``log(FloatTensor(pop_prob_list)[ids])`` -- float32 table, float32 log."""
    table = np.asarray(pop_prob_list, dtype=np.float64).astype(np.float32)
    return np.log(table[np.asarray(sample_items_id).reshape(-1)]).astype(np.float32)


def pooled_targets(bs: int, max_seq_len: int, rank: int) -> np.ndarray:
    """Pooled-negative extension (not in the reference): rank r's positives are its
    own columns, offset by ``r * B*(S+1)`` in the rank-major pooled column space."""
    return ce_labels(bs, max_seq_len) + rank * bs * (max_seq_len + 1)
