"""Datasets / samplers with the reference's class names (``T/data_utils/dataset.py``)."""
from __future__ import annotations

import math

import numpy as np
import torch
from torch.utils.data import Dataset


class BuildTrainDataset(Dataset):
    """``T/data_utils/dataset.py:10-36``: left-pad the user's train sequence with item 0 to S + 1 slots;
    ``log_mask`` marks the S input positions that are real; modal runs gather the token rows."""

    def __init__(self, u2seq, item_content, item_num, max_seq_len, use_modal):
        self.u2seq, self.item_content, self.item_num = u2seq, item_content, item_num
        self.max_seq_len = max_seq_len + 1
        self.use_modal = use_modal

    def __len__(self):
        return len(self.u2seq)

    def __getitem__(self, user_id):
        seq = self.u2seq[user_id]
        pad = self.max_seq_len - len(seq)
        ids = torch.LongTensor([0] * pad + list(seq))
        log_mask = torch.FloatTensor([0] * pad + [1] * (len(seq) - 1))
        items = torch.LongTensor(self.item_content[ids]) if self.use_modal else ids
        return ids, items, log_mask


def collate_train_batch(u2seq, users, item_content, max_seq_len, use_modal):
    """Vectorised equivalent of DataLoader(BuildTrainDataset) default collation for a list of users:
    (ids int64 [B, S+1], items int64 [B, S+1, 2T] | [B, S+1], log_mask float32 [B, S])."""
    B, L = len(users), max_seq_len + 1
    ids = np.zeros((B, L), dtype=np.int64)
    log_mask = np.zeros((B, max_seq_len), dtype=np.float32)
    for r, u in enumerate(users):
        seq = u2seq[u]
        ids[r, L - len(seq):] = seq
        log_mask[r, L - len(seq):] = 1.0
    if use_modal and hasattr(item_content, "device_batch"):      # vision catalogue in the reference's LMDB (data_utils/images.py)
        return torch.from_numpy(ids), item_content.device_batch(ids, "cuda"), torch.from_numpy(log_mask)
    items = np.asarray(item_content)[ids] if use_modal else ids
    items = torch.from_numpy(np.ascontiguousarray(items))
    if items.dtype != torch.uint8:          # uint8 = decoded images of the vision variant (normalised on the device)
        items = items.long()
    return torch.from_numpy(ids), items, torch.from_numpy(log_mask)


def collate_bce_batch(u2seq, users, item_content, max_seq_len, item_num, use_modal, rng):
    """Batch of the BCE variant (``bce_text/main-end2end/data_utils/dataset.py:25-49``): per user the left-padded sequence and,
    for every input position, one negative drawn uniformly from the items the user has not interacted with (rejection
    sampling, as the reference does); returns (sample_items [B, S+1, 2(, 2T)], log_mask [B, S])."""
    B, L = len(users), max_seq_len + 1
    items = np.zeros((B, L, 2), dtype=np.int64)
    log_mask = np.zeros((B, max_seq_len), dtype=np.float32)
    for r, u in enumerate(users):
        seq = u2seq[u]
        n = len(seq)
        items[r, L - n:, 0] = seq
        log_mask[r, L - n:] = 1.0
        seen = set(seq)
        for j in range(n - 1):
            neg = int(rng.integers(1, item_num + 1))
            while neg in seen:
                neg = int(rng.integers(1, item_num + 1))
            items[r, L - n + j, 1] = neg
    out = np.asarray(item_content)[items] if use_modal else items
    return torch.from_numpy(np.ascontiguousarray(out)).long(), torch.from_numpy(log_mask)


class BuildEvalDataset(Dataset):
    """``T/data_utils/dataset.py:39-65`` (inputs are pre-computed item EMBEDDINGS; label is one-hot over items)."""

    def __init__(self, u2seq, item_content, max_seq_len, item_num):
        self.u2seq, self.item_content, self.item_num = u2seq, item_content, item_num
        self.max_seq_len = max_seq_len + 1

    def __len__(self):
        return len(self.u2seq)

    def __getitem__(self, user_id):
        seq = self.u2seq[user_id]
        tokens, target = seq[:-1], seq[-1]
        pad = self.max_seq_len - len(seq)
        log_mask = [0] * pad + [1] * len(tokens)
        labels = np.zeros(self.item_num)
        labels[target - 1] = 1.0
        return torch.LongTensor([user_id]), self.item_content[[0] * pad + list(tokens)], torch.FloatTensor(log_mask), labels


class SequentialDistributedSampler(torch.utils.data.sampler.Sampler):
    """``T/data_utils/dataset.py:68-94``: contiguous shards, padded with the last index to a multiple of
    batch_size * num_replicas."""

    def __init__(self, dataset, batch_size, rank=None, num_replicas=None):
        if num_replicas is None:
            num_replicas = torch.distributed.get_world_size()
        if rank is None:
            rank = torch.distributed.get_rank()
        self.dataset, self.num_replicas, self.rank, self.batch_size = dataset, num_replicas, rank, batch_size
        self.num_samples = int(math.ceil(len(dataset) * 1.0 / batch_size / num_replicas)) * batch_size
        self.total_size = self.num_samples * num_replicas

    def __iter__(self):
        idx = list(range(len(self.dataset)))
        idx += [idx[-1]] * (self.total_size - len(idx))
        return iter(idx[self.rank * self.num_samples:(self.rank + 1) * self.num_samples])

    def __len__(self):
        return self.num_samples


def epoch_batches(n_samples: int, batch_size: int, world: int, rank: int, epoch: int):
    """The index batches one rank sees in one epoch under the reference's sampler + loader (``T/run.py:114,123-124,230``):
    ``torch.utils.data.DistributedSampler`` itself (shuffle with seed 0 + epoch, PAD to a multiple of ``world`` by repeating the
    head of the permutation, rank r takes every world-th index) cut into ``batch_size`` pieces by a ``DataLoader`` without
    ``drop_last`` -- the LAST BATCH IS SHORT, so the batch size is not a constant of the step."""
    from torch.utils.data.distributed import DistributedSampler
    sampler = DistributedSampler(range(n_samples), num_replicas=world, rank=rank, shuffle=True, seed=0, drop_last=False)
    sampler.set_epoch(epoch)
    idx = list(iter(sampler))
    return [idx[i:i + batch_size] for i in range(0, len(idx), batch_size)]
