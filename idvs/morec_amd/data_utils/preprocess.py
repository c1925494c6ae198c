"""Preprocessing with the reference's function names and return tuples (``T/data_utils/preprocess.py``).
The integer bookkeeping (dense item re-numbering, train/valid/test splits, popularity table) is parity-critical
and pinned bit-exactly against goldens captured from the reference (tests/test_data_utils.py)."""
from __future__ import annotations

import numpy as np
import torch


def read_news(news_path):
    """``T/data_utils/preprocess.py:84-98``: item names -> ids 1.. in file order; one extra 'mask sentence' entry."""
    item_id_to_dic, item_id_to_name, item_name_to_id = {}, {}, {}
    nxt = 1
    with open(news_path, "r") as fh:
        for line in fh:
            name = line.strip("\n").split("\t")[0]
            item_name_to_id[name] = nxt
            item_id_to_dic[nxt] = name
            item_id_to_name[nxt] = name
            nxt += 1
    item_id_to_dic[nxt] = "this is a mask sentence"
    return item_id_to_dic, item_name_to_id, item_id_to_name


def read_news_bert(news_path, args, tokenizer):
    """``T/data_utils/preprocess.py:101-128``: tokenise title / abstract / body to fixed length (max_length padding)."""
    item_id_to_dic, item_id_to_name, item_name_to_id = {}, {}, {}
    nxt = 1
    with open(news_path, "r") as fh:
        for line in fh:
            doc_name, title, abstract = line.strip("\n").split("\t")[:3]
            enc = []
            for attr, text, n in (("title", title, args.num_words_title), ("abstract", abstract, args.num_words_abstract),
                                  ("body", "", args.num_words_body)):
                enc.append(tokenizer(text.lower(), max_length=n, padding="max_length", truncation=True)
                           if attr in args.news_attributes else [])
            item_name_to_id[doc_name] = nxt
            item_id_to_name[nxt] = doc_name
            item_id_to_dic[nxt] = enc
            nxt += 1
    return item_id_to_dic, item_name_to_id, item_id_to_name


def read_behaviors(behaviors_path, before_item_id_to_dic, before_item_name_to_id, before_item_id_to_name, max_seq_len,
                   min_seq_len, Log_file=None):
    """``T/data_utils/preprocess.py:5-81``.  Returns the reference's 9-tuple:
    (item_num, item_id_to_dic, users_train, users_valid, users_test, users_history_for_valid,
     users_history_for_test, item_name_to_id, pop_prob_list)."""
    n_before = len(before_item_name_to_id)
    seen = np.zeros(n_before + 1, dtype=np.int64)
    raw = {}
    with open(behaviors_path, "r") as fh:
        for line in fh:
            fields = line.strip("\n").split("\t")
            names = fields[1].split(" ")
            if len(names) < min_seq_len:                      # :19
                continue
            tail = [before_item_name_to_id[n] for n in names[-(max_seq_len + 3):]]   # :21 keep the LAST S+3
            raw[fields[0]] = tail
            np.add.at(seen, tail, 1)
    remap, item_id_to_dic, item_name_to_id = {}, {}, {}
    nxt = 1
    for old in range(1, n_before + 1):                        # :30-40 dense renumbering in original id order
        if seen[old]:
            remap[old] = nxt
            item_id_to_dic[nxt] = before_item_id_to_dic[old]
            item_name_to_id[before_item_id_to_name[old]] = nxt
            nxt += 1
    item_num = len(remap)
    users_train, users_valid, users_test, hist_valid, hist_test = {}, {}, {}, {}, {}
    train_counts = np.zeros(item_num + 1, dtype=np.int64)
    for uid, tail in enumerate(raw.values()):
        seq = [remap[i] for i in tail]
        users_train[uid] = seq[:-2]                           # :53-55
        users_valid[uid] = seq[-(max_seq_len + 2):-1]
        users_test[uid] = seq[-(max_seq_len + 1):]
        np.add.at(train_counts, seq[:-2], 1)                  # :60-61 popularity from TRAIN interactions only
        hist_valid[uid] = torch.LongTensor(np.array(seq[:-2]))
        hist_test[uid] = torch.LongTensor(np.array(seq[:-1]))
    powered = np.power(train_counts.astype(np.float64), 1.0)
    pop_prob_list = np.append([1], powered[1:] / np.sum(powered[1:]))   # :71-76, slot 0 := 1 so log(pop[0]) = 0
    if Log_file is not None:
        Log_file.info("##### items after clearing %d, users %d #####" % (item_num, len(users_train)))
    return item_num, item_id_to_dic, users_train, users_valid, users_test, hist_valid, hist_test, item_name_to_id, \
        pop_prob_list


def get_doc_input_bert(item_id_to_content, args):
    """``T/data_utils/preprocess.py:131-172``: int32 [item_num + 1, n_words] id and mask tables per attribute
    (row 0 = padding item, all zero)."""
    item_num = len(item_id_to_content) + 1
    out = []
    for k, (attr, n) in enumerate((("title", args.num_words_title), ("abstract", args.num_words_abstract),
                                   ("body", args.num_words_body))):
        if attr not in args.news_attributes:
            out += [None, None]
            continue
        ids = np.zeros((item_num, n), dtype="int32")
        mask = np.zeros((item_num, n), dtype="int32")
        for item_id in range(1, item_num):
            enc = item_id_to_content[item_id][k]
            ids[item_id] = enc["input_ids"]
            mask[item_id] = enc["attention_mask"]
        out += [ids, mask]
    return tuple(out)


def read_images(images_path):
    """``V/data_utils/preprocess.py:88-101``: one item name per line (``v<int>``) -> ids 1.. in file order and the LMDB key of each
    (the ascii decimal of the name's integer, as ``dataset/HM/build_lmdb_hm.py:44-50`` wrote it)."""
    item_id_to_keys, item_name_to_id, item_id_to_name = {}, {}, {}
    index = 1
    with open(images_path, "r") as f:
        for line in f:
            image_name = line.strip("\n").split("\t")[0]
            item_name_to_id[image_name] = index
            item_id_to_name[index] = image_name
            item_id_to_keys[index] = u"{}".format(int(image_name.replace("v", ""))).encode("ascii")
            index += 1
    return item_id_to_keys, item_name_to_id, item_id_to_name
