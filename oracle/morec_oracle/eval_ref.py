"""Evaluation restatement (HR@10 / nDCG@10) -- TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations


import numpy as np


def eval_ranks(scores: np.ndarray, histories, targets: np.ndarray) -> np.ndarray:
    """``eval_model`` ``T/data_utils/metrics.py:96-102`` + ``metrics_topK`` ``:49-57``: per user set
    history scores to -inf, drop column 0, rank (1-based) of the target among all item_num items in
    descending score order.  With tie-free scores rank = 1 + #(items scoring strictly higher).
    scores float32[U, item_num+1]; targets are item ids (1-based).  Returns int64[U]."""
    U = scores.shape[0]
    ranks = np.zeros(U, dtype=np.int64)
    for u in range(U):
        s = scores[u].astype(np.float32).copy()
        s[np.asarray(histories[u], dtype=np.int64)] = -np.inf
        s = s[1:]
        t = s[targets[u] - 1]
        ranks[u] = 1 + int(np.sum(s > t))
    return ranks


def hit_ndcg_at_k(ranks: np.ndarray, k: int = 10):
    """``metrics_topK``: Hit = [rank <= k]; nDCG = 1/log2(rank+1) when hit, else 0; means over users."""
    hit = (ranks <= k).astype(np.float64)
    ndcg = np.where(ranks <= k, 1.0 / np.log2(ranks.astype(np.float64) + 1.0), 0.0)
    return float(hit.mean()), float(ndcg.mean())
