"""Preprocessing restatement (integer bookkeeping, bit-exact target) -- TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import numpy as np


def read_news_ref(news_path: str):
    """``read_news`` ``T/data_utils/preprocess.py:84-98``: item names numbered 1.. in file order."""
    name_to_id, id_to_name = {}, {}
    with open(news_path, "r") as f:
        for k, line in enumerate(f, start=1):
            name = line.strip("\n").split("\t")[0]
            name_to_id[name] = k
            id_to_name[k] = name
    return name_to_id, id_to_name


def read_behaviors_ref(behaviors_path: str, before_name_to_id: dict, max_seq_len: int, min_seq_len: int):
    """``read_behaviors`` ``T/data_utils/preprocess.py:5-81``:
    keep users with >= min_seq_len items (:19); keep the LAST S+3 items (:21); renumber items that
    occur densely 1..item_num in original id order (:30-40); per user train = seq[:-2],
    valid = seq[-(S+2):-1], test = seq[-(S+1):] (:53-55); pop_prob from TRAIN counts only,
    normalised in float64, with a leading 1 for the padding id (:60-61,71-76)."""
    n_before = len(before_name_to_id)
    counts = np.zeros(n_before + 1, dtype=np.int64)
    user_seqs = []
    with open(behaviors_path, "r") as f:
        for line in f:
            parts = line.strip("\n").split("\t")
            names = parts[1].split(" ")
            if len(names) < min_seq_len:
                continue
            names = names[-(max_seq_len + 3):]
            ids = [before_name_to_id[n] for n in names]
            user_seqs.append(ids)
            for i in ids:
                counts[i] += 1
    before_to_now = {}
    nxt = 1
    for before_id in range(1, n_before + 1):
        if counts[before_id] != 0:
            before_to_now[before_id] = nxt
            nxt += 1
    item_num = len(before_to_now)
    users_train, users_valid, users_test, hist_valid, hist_test = {}, {}, {}, {}, {}
    train_counts = np.zeros(item_num + 1, dtype=np.int64)
    for uid, ids in enumerate(user_seqs):
        seq = [before_to_now[i] for i in ids]
        users_train[uid] = seq[:-2]
        users_valid[uid] = seq[-(max_seq_len + 2):-1]
        users_test[uid] = seq[-(max_seq_len + 1):]
        for i in seq[:-2]:
            train_counts[i] += 1
        hist_valid[uid] = np.asarray(seq[:-2], dtype=np.int64)
        hist_test[uid] = np.asarray(seq[:-1], dtype=np.int64)
    powered = np.power(train_counts.astype(np.float64), 1.0)
    pop = powered[1:] / np.sum(powered[1:])
    pop_prob_list = np.append([1], pop)
    return dict(item_num=item_num, before_to_now=before_to_now, users_train=users_train, users_valid=users_valid,
                users_test=users_test, hist_valid=hist_valid, hist_test=hist_test, pop_prob_list=pop_prob_list)
