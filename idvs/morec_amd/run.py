"""Training driver with the shape of ``T/run.py``: ``train(args, use_modal, local_rank)``, ``run_eval``,
``setup_seed`` and a ``__main__`` that works under ``torchrun`` on ROCm (RCCL behind backend ``'nccl'``).

    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m idvs.morec_amd.run --item_tower modal \
        --bert_model_load bert_base_uncased --synthetic 60000 --batch_size 128 --embedding_dim 512 --fused_step --pool_negatives

Two optimisation paths: the reference's (DDP + torch.optim.AdamW on the drop-in ``Model``; ``T/run.py:148-162,241-247``
without the fp16 GradScaler, which bf16 does not need) and ``--fused_step`` (``train_step.TrainStep``).
Checkpoints: reference format (``epoch-N.pt``), saved by rank 0 when the validation Hit@10 improves (modal runs only, as
``T/run.py:265-267``), resumed with ``--load_ckpt_name epoch-N.pt`` (``T/run.py:130-139,193-195``; early stopping is switched off
on a resumed run, like there); ``--mode test`` evaluates a checkpoint on the test split (``T/run_test.py:115-128``).  Both
optimisation paths read and write the SAME optimizer state (``TrainStep.optimizer_state_dict``)."""
from __future__ import annotations

import logging
import os
import queue
import random
import threading
import time

import numpy as np
import torch
import torch.distributed as dist
import torch.optim as optim
from torch.nn.parallel import DistributedDataParallel as DDP

from .data_utils import (collate_bce_batch, collate_train_batch, epoch_batches, eval_model, get_checkpoint, get_item_embeddings,
                         load_model, read_behaviors, read_news, save_model)
from .model import BceModel, BertShape, HipBertModel, Model
from .model.swin import HipSwinForImageClassification
from .swin_engine import SwinShape
from .parameters import parse_args
from . import engine
from .train_step import TrainStep

Log = logging.getLogger("morec")


def setup_seed(seed):   # T/run.py:307-314
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


def synthetic_dataset(n_users, n_items, S, T, seed=12345, full_len=False, extra_T=()):
    """MIND-shaped synthetic data in the structures ``read_behaviors`` / ``get_doc_input_bert`` return.
    ``full_len``: every user has raw history S + 3 (train sequence of S + 1 items: the throughput shape of bench.py).
    ``extra_T``: token counts of further text attributes (abstract, body: ``--news_attributes title,abstract``); their [ids | mask]
    blocks follow the title's in the item rows, as ``get_doc_input_bert`` + ``T/run.py:86-90`` lay them out."""
    rng = np.random.default_rng(seed)
    w = 1.0 / np.arange(1, n_items + 1)
    w /= w.sum()
    perm = rng.permutation(n_items) + 1
    lens = np.full(n_users, S + 3) if full_len else rng.integers(5, S + 4, n_users)
    users_train, users_valid, users_test, hist_valid, hist_test = {}, {}, {}, {}, {}
    counts = np.zeros(n_items + 1)
    for u in range(n_users):
        seq = [int(v) for v in perm[rng.choice(n_items, size=int(lens[u]), p=w)]]
        users_train[u], users_valid[u], users_test[u] = seq[:-2], seq[-(S + 2):-1], seq[-(S + 1):]
        np.add.at(counts, seq[:-2], 1)
        hist_valid[u], hist_test[u] = torch.LongTensor(seq[:-2]), torch.LongTensor(seq[:-1])
    counts[1:] += 1e-9
    pop = np.append([1], counts[1:] / counts[1:].sum())
    content = np.zeros((n_items + 1, 2 * T), dtype=np.int64)
    tl = rng.integers(8, T + 1, n_items)
    toks = rng.integers(1000, 30522, (n_items, T))
    valid = np.arange(T)[None, :] < tl[:, None]
    toks = np.where(valid, toks, 0)
    toks[:, 0] = 101
    toks[np.arange(n_items), tl - 1] = 102
    content[1:, :T], content[1:, T:] = toks, valid
    for Tx in extra_T:
        blk = np.zeros((n_items + 1, 2 * Tx), dtype=np.int64)
        tl = rng.integers(8, Tx + 1, n_items)
        toks = rng.integers(1000, 30522, (n_items, Tx))
        valid = np.arange(Tx)[None, :] < tl[:, None]
        toks = np.where(valid, toks, 0)
        toks[:, 0] = 101
        toks[np.arange(n_items), tl - 1] = 102
        blk[1:, :Tx], blk[1:, Tx:] = toks, valid
        content = np.concatenate((content, blk), axis=1)
    return n_items, content, users_train, users_valid, users_test, hist_valid, hist_test, pop


class _BatchSet(torch.utils.data.Dataset):
    """One DataLoader "sample" = one collated batch (``batch_size=None``): the loader's worker processes run ``make(batch_idx)``."""

    def __init__(self, make, batches):
        self.make, self.batches = make, batches

    def __len__(self):
        return len(self.batches)

    def __getitem__(self, i):
        return self.make(self.batches[i])


class BatchMaker:
    """Host side of one batch -- the work of ``T/run.py:111-124``'s DataLoader workers -- as a PICKLABLE object: it holds the user
    lists, the item table (numpy) and flags, no closure over the model or the stepper, so DataLoader worker processes can be started by
    fork, forkserver or spawn alike.  ``text_attrs`` (fused step, text tower): [(name, first column, width)] of the [ids | mask] blocks of
    the item rows -- the collate then also prepares the unpadded token layout's index vectors (``engine.token_packing_host``), one tuple
    for a single attribute, one per attribute otherwise.  ``pin``: False inside worker processes (they must not touch the HIP runtime;
    the loader's pin thread page-locks), None = page-lock when a GPU is present."""

    def __init__(self, users, users_train, item_content, S, use_modal, *, bce=False, item_num=0, neg_seed=0, text_attrs=None, pad_to=0, pin=None):
        self.users, self.users_train, self.item_content, self.S, self.use_modal = users, users_train, item_content, S, use_modal
        self.bce, self.item_num, self.neg_seed = bce, item_num, neg_seed
        self.text_attrs, self.pad_to, self.pin = list(text_attrs or []), pad_to, pin
        self._neg_rng = None

    def __call__(self, batch_idx):
        batch_users = [self.users[i] for i in batch_idx]
        if self.bce:      # bce_text/main-end2end/run.py:224-237
            if self._neg_rng is None:
                self._neg_rng = np.random.default_rng(self.neg_seed)
            items, log_mask = collate_bce_batch(self.users_train, batch_users, self.item_content, self.S, self.item_num, self.use_modal, self._neg_rng)
            return None, items, log_mask, None
        ids, items, log_mask = collate_train_batch(self.users_train, batch_users, self.item_content, self.S, self.use_modal)
        pack = None
        if self.text_attrs:      # the collate's share of the unpadded token layout (no host sync in the step)
            rows = items.view(-1, items.size(-1))
            packs = []
            for _, a0, aw in self.text_attrs:
                h = aw // 2
                packs.append(engine.token_packing_host(rows[:, a0 + h:a0 + aw], rows[:, a0:a0 + h], pad_to=self.pad_to, pin=self.pin))
            pack = packs[0] if len(packs) == 1 else (None if any(x is None for x in packs) else tuple(packs))
        return ids, items, log_mask, pack


class _EpochBatchSet(torch.utils.data.Dataset):
    """Dataset of a loader that lives for the whole run (``persistent_workers``): a sample key is ``(epoch, b)`` and the worker derives
    that epoch's index batches itself (``epoch_batches``: the reference's DistributedSampler + batching, cached per epoch), so nothing has
    to be sent to the workers when an epoch starts."""

    def __init__(self, make, n_users, batch_size, world, rank):
        self.make, self.n_users, self.batch_size, self.world, self.rank = make, n_users, batch_size, world, rank
        self._ep, self._batches = None, None

    def __getitem__(self, key):
        ep, b = key
        if ep != self._ep:
            self._batches, self._ep = epoch_batches(self.n_users, self.batch_size, self.world, self.rank, ep), ep
        return self.make(self._batches[b])


class _EpochSampler:
    def __init__(self):
        self.keys = []

    def set_epoch(self, epoch, n_batches):
        self.keys = [(epoch, b) for b in range(n_batches)]

    def __iter__(self):
        return iter(self.keys)

    def __len__(self):
        return len(self.keys)


def _with_next(it):
    """(item, next item or None) pairs: one element of look-ahead over any iterator."""
    it = iter(it)
    cur = next(it, None)
    while cur is not None:
        nxt = next(it, None)
        yield cur, nxt
        cur = nxt


def _to_device(t, dev):
    """Tensors of a (possibly nested) tuple to the device, asynchronously; None entries stay."""
    if t is None:
        return None
    if isinstance(t, (tuple, list)):
        return tuple(_to_device(x, dev) for x in t)
    return t.to(dev, non_blocking=True)


class BatchPrefetcher:
    """Batches built AHEAD of the device by a worker thread: collate (numpy gathers out of the item table), the unpadded token layout's
    index vectors and page-locking -- what the reference gets from ``DataLoader(num_workers=12, pin_memory=True)`` (``T/run.py:111-124``;
    ``V/run.py:93-94``).  One thread is enough here: a batch is a few numpy gathers (they release the GIL), the step itself is launched
    asynchronously, so the host only has to stay ``depth`` batches ahead of the GPU.  ``make(batch_idx)`` returns a tuple of CPU tensors
    (or ``None`` entries); iteration yields ``(b, tuple)`` in order; an exception in the worker is re-raised in the consumer."""

    def __init__(self, make, batches, depth: int = 2, pin: bool = True):
        self.make, self.batches, self.pin = make, batches, pin and torch.cuda.is_available()
        self.q = queue.Queue(maxsize=max(1, depth))
        self.stop = threading.Event()
        self.wait_s, self.make_s, self.n = 0.0, 0.0, 0      # consumer time blocked on an empty queue; worker time building batches
        self.t = threading.Thread(target=self._run, name="morec-prefetch", daemon=True)
        self.t.start()

    def _pin(self, t):
        if isinstance(t, tuple):
            return tuple(self._pin(x) for x in t)
        return t.pin_memory() if (self.pin and isinstance(t, torch.Tensor) and not t.is_pinned()) else t

    def _put(self, item):
        while not self.stop.is_set():
            try:
                self.q.put(item, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def _run(self):
        try:
            for b, idx in enumerate(self.batches):
                t_ = time.perf_counter()
                item = (b, tuple(self._pin(x) for x in self.make(idx)))
                self.make_s += time.perf_counter() - t_
                self.n += 1
                if not self._put(item):
                    return
        except BaseException as e:  # noqa: BLE001 -- handed to the consumer
            self._put(e)
            return
        self._put(None)

    def __iter__(self):
        while True:
            t_ = time.perf_counter()
            item = self.q.get()
            self.wait_s += time.perf_counter() - t_
            if item is None:
                return
            if isinstance(item, BaseException):
                raise item
            yield item

    def close(self):
        self.stop.set()
        self.t.join(timeout=5)


def run_eval(model, item_content, user_history, users_eval, batch_size, item_num, use_modal, args, mode, local_rank):
    t0 = time.time()
    item_embeddings = get_item_embeddings(model, item_content, batch_size, args, use_modal, local_rank)
    hit10 = eval_model(model, user_history, users_eval, item_embeddings, batch_size, args, item_num, Log, mode, local_rank)
    Log.info("eval %.1f s, Hit10 %.5f" % (time.time() - t0, hit10 * 100))
    return hit10


def model_dir_candidates(args, world: int = 1):
    """The reference's checkpoint directory, label for label, first; then the other spellings a checkpoint may sit under.
    TEXT (``T/run.py:326-337``): ``checkpoint_<dir_label>/cpt_<model>_ed_<D>_bs_<batch_size * gpus>_lr_<lr>_Flr_<Flr>_L2_<l2>_FL2_<Fl2>`` with
    ``model = bert_model_load``, ``dir_label = <item_tower>_<model>_freeze_<n>``; the ID tower: ``model = id``, ``dir_label = <item_tower>``
    (batch size times the GPU count there too).  VISION (``V/run.py:307-324``): ``model = CV_model_load`` WITHOUT its ``.pth``, the
    label starts ``<model>-<freeze_paras_before>_ed_...``; V's ID branch does not multiply the batch size by the GPU count.
    ``world`` = number of ranks (the reference uses ``torch.cuda.device_count()``)."""
    tail = f"_lr_{args.lr}_Flr_{args.fine_tune_lr}_L2_{args.l2_weight}_FL2_{args.fine_tune_l2_weight}"
    root = args.checkpoint_root
    vision = "modal" in args.item_tower and getattr(args, "CV_model_load", "None") != "None"
    out = []
    if vision:
        enc = args.CV_model_load.replace(".pth", "")
        dir_label = f"{args.item_tower}_{enc}_freeze_{args.freeze_paras_before}"
        out.append((dir_label, f"{enc}-{args.freeze_paras_before}_ed_{args.embedding_dim}_bs_{args.batch_size * world}" + tail))
        # what earlier versions of THIS driver wrote (the text run's label with the vision tower's name, '.pth' kept)
        old = args.CV_model_load
        out.append((f"{args.item_tower}_{old}_freeze_{args.freeze_paras_before}", f"{old}_ed_{args.embedding_dim}_bs_{args.batch_size * world}" + tail))
    elif "modal" in args.item_tower:
        enc = args.bert_model_load
        out.append((f"{args.item_tower}_{enc}_freeze_{args.freeze_paras_before}", f"{enc}_ed_{args.embedding_dim}_bs_{args.batch_size * world}" + tail))
    else:
        out.append((str(args.item_tower), f"id_ed_{args.embedding_dim}_bs_{args.batch_size * world}" + tail))      # T/run.py:331-337
        if world > 1:
            out.append((str(args.item_tower), f"id_ed_{args.embedding_dim}_bs_{args.batch_size}" + tail))          # V/run.py:318-323
    return [os.path.join(root, "checkpoint_" + d, "cpt_" + l) for d, l in out]


def model_dir_of(args, world: int = 1):
    """Where this run WRITES its checkpoints: the reference's own directory for the tower (``model_dir_candidates``)."""
    return model_dir_candidates(args, world)[0]


def normalize_pretrained_bert_keys(sd: dict) -> dict:
    """Key names of a stock ``bert-*`` checkpoint file -> ``BertModel.state_dict()`` names, as ``from_pretrained`` does at load time
    (``T/run.py:51-53``): the ``bert.`` prefix of ``BertForPreTraining`` files is dropped, the TF-era LayerNorm names ``*.LayerNorm.gamma``
    / ``*.LayerNorm.beta`` (what ``bert-base-uncased/pytorch_model.bin`` holds) become ``weight`` / ``bias``, and the pre-training
    heads (``cls.*``) and the ``position_ids`` buffer are left out."""
    out = {}
    for k, v in sd.items():
        if k.startswith("bert."):
            k = k[len("bert."):]
        if k.startswith("cls.") or k.endswith("embeddings.position_ids"):
            continue
        if k.endswith("LayerNorm.gamma"):
            k = k[: -len("gamma")] + "weight"
        elif k.endswith("LayerNorm.beta"):
            k = k[: -len("beta")] + "bias"
        out[k] = v
    return out


def _load_pretrained_text_tower(bert, args):
    """``BertModel.from_pretrained(bert_model_load)`` (``T/run.py:51-53``): weights from ``<pretrained_dir>/<bert_model_load>/
    pytorch_model.bin`` (or ``model.safetensors``) when that directory exists; otherwise the tower keeps its random
    initialisation and says so -- there are no checkpoints in this image (``pretrained_models/`` ships configs only)."""
    d = os.path.join(args.pretrained_dir, args.bert_model_load)
    for fn in ("pytorch_model.bin", "model.safetensors"):
        path = os.path.join(d, fn)
        if os.path.exists(path):
            if fn.endswith(".bin"):
                sd = torch.load(path, map_location="cpu", weights_only=True)
            else:
                from safetensors.torch import load_file
                sd = load_file(path)
            missing, unexpected = bert.load_state_dict(normalize_pretrained_bert_keys(sd), strict=False)
            missing = [k for k in missing if not k.startswith("pooler.")]      # the pooler is frozen and unused (T/run.py:67-69)
            if missing:     # a tower that silently keeps part of its random initialisation is a different experiment: refuse
                raise SystemExit(f"text tower: {path} does not provide {len(missing)} parameter(s) of the tower, e.g. {missing[:4]}")
            Log.info("text tower: %s loaded (%d unexpected keys ignored)" % (path, len(unexpected)))
            return True
    Log.warning("text tower: no pretrained weights under %s -- RANDOM initialisation" % d)
    return False


def train(args, use_modal, local_rank):
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    S, T = args.max_seq_len, args.num_words_title
    vision = use_modal and args.CV_model_load != "None"
    if not vision and hasattr(args, "CV_model_load"):
        del args.CV_model_load                # `Model` picks the vision tower by the presence of this attribute (V/model/model.py:24-29)
    if args.synthetic > 0:
        extra_T = [w for n, w in (("abstract", args.num_words_abstract), ("body", args.num_words_body))
                   if use_modal and not vision and n in args.news_attributes]
        item_num, content, users_train, users_valid, users_test, hist_valid, hist_test, pop = synthetic_dataset(
            args.synthetic, args.synthetic_items, S, T, full_len=bool(getattr(args, "synthetic_full_len", False)), extra_T=extra_T)
        item_content = content if use_modal else np.arange(item_num + 1)
        if vision:      # decoded uint8 images; a real run passes --images_npy (what the LMDB of V/data_utils holds after Resize)
            R = args.CV_resize
            item_content = (np.load(args.images_npy, mmap_mode="r") if args.images_npy != "None" else
                            np.random.default_rng(4321).integers(0, 256, (item_num + 1, R, R, 3), dtype=np.uint8))
            assert item_content.shape == (item_num + 1, R, R, 3) and item_content.dtype == np.uint8
    elif use_modal and not vision:                                       # T/run.py:28-98: tokenizer -> fixed-length id / mask tables
        from transformers import BertTokenizer
        from .data_utils import get_doc_input_bert, read_news_bert
        tok_dir = os.path.join(args.pretrained_dir, args.bert_model_load)
        if not os.path.exists(os.path.join(tok_dir, "vocab.txt")):
            raise SystemExit(f"modal run on real data: no tokenizer vocabulary under {tok_dir} (--pretrained_dir / --bert_model_load)")
        tokenizer = BertTokenizer.from_pretrained(tok_dir)
        a, b, c = read_news_bert(os.path.join(args.root_data_dir, args.dataset, args.news), args, tokenizer)
        item_num, item_id_to_dic, users_train, users_valid, users_test, hist_valid, hist_test, _, pop = read_behaviors(
            os.path.join(args.root_data_dir, args.dataset, args.behaviors), a, b, c, S, args.min_seq_len, Log)
        tables = get_doc_input_bert(item_id_to_dic, args)
        item_content = np.concatenate([x for x in tables if x is not None], axis=1)
    elif vision:                                                         # V/run.py:62-80: item list -> LMDB keys, images from the LMDB
        from .data_utils import LmdbImageStore, LmdbItemImages, read_images
        if args.image_lmdb == "None":
            raise SystemExit("vision run on real data: --image_lmdb <the LMDB of dataset/HM/build_lmdb_hm.py> (or --synthetic N --images_npy ...)")
        a, b, c = read_images(os.path.join(args.root_data_dir, args.dataset, args.news))
        item_num, id2key, users_train, users_valid, users_test, hist_valid, hist_test, _, pop = read_behaviors(
            os.path.join(args.root_data_dir, args.dataset, args.behaviors), a, b, c, S, args.min_seq_len, Log)
        item_content = LmdbItemImages(LmdbImageStore(os.path.join(args.root_data_dir, args.dataset, args.image_lmdb)), id2key, args.CV_resize)
    else:
        a, b, c = read_news(os.path.join(args.root_data_dir, args.dataset, args.news))
        item_num, _, users_train, users_valid, users_test, hist_valid, hist_test, _, pop = read_behaviors(
            os.path.join(args.root_data_dir, args.dataset, args.behaviors), a, b, c, S, args.min_seq_len, Log)
        item_content = np.arange(item_num + 1)
    bert = None
    if vision:
        bert = HipSwinForImageClassification(SwinShape.named(args.CV_model_load), args.embedding_dim)   # V/run.py:47-54
        for index, (name, param) in enumerate(bert.named_parameters()):   # V/run.py:58-60
            if index < args.freeze_paras_before:
                param.requires_grad = False
    elif use_modal:
        shape = BertShape.named(args.bert_model_load)
        args.word_embedding_dim = shape.hidden_size              # T/run.py:55-72
        bert = HipBertModel(shape)
        pooler = {"pooler.dense.weight", "pooler.dense.bias"}
        for index, (name, param) in enumerate(bert.named_parameters()):   # T/run.py:73-75
            if index < args.freeze_paras_before or name in pooler:
                param.requires_grad = False
    if use_modal and not vision:
        _load_pretrained_text_tower(bert, args)
    bce = args.loss == "bce"
    if bce and (args.fused_step or vision):
        raise SystemExit("--loss bce runs on the drop-in autograd path with the text / ID towers (bce_text/main-end2end)")
    model = (BceModel(args, item_num, use_modal, bert) if bce else Model(args, item_num, use_modal, bert, pop)).to(local_rank)
    users = list(users_train.keys())
    model_dir = model_dir_of(args, world)
    ckpt, start_epoch, is_early_stop = None, 0, True
    if "None" not in args.load_ckpt_name:                               # T/run.py:130-139: BEFORE the arenas / DDP are built
        ckpt_path = None
        for cand in model_dir_candidates(args, world):      # the reference's directory first, then the older spellings
            ckpt_path = get_checkpoint(cand, args.load_ckpt_name)
            if ckpt_path is not None:
                break
        if ckpt_path is None:
            raise SystemExit(f"--load_ckpt_name {args.load_ckpt_name}: not found under any of {model_dir_candidates(args, world)}")
        start_epoch = load_model(model, ckpt_path)
        ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=False)
        if ckpt.get("rng_state") is not None:
            torch.set_rng_state(ckpt["rng_state"])
        if ckpt.get("cuda_rng_state") is not None and torch.cuda.is_available():
            torch.cuda.set_rng_state(ckpt["cuda_rng_state"])
        if ckpt.get("morec_drop_calls") is not None:      # the library's counter-based dropout / DropPath streams continue where they stopped
            model._drop_calls = int(ckpt["morec_drop_calls"])
        is_early_stop = False
        Log.info("model loaded from %s (epoch %d)" % (ckpt_path, start_epoch))
    if args.mode == "test":                                             # T/run_test.py:115-128
        if ckpt is None:
            raise SystemExit("--mode test needs --load_ckpt_name epoch-N.pt")
        return run_eval(model, item_content, hist_test, users_test, 512, item_num, use_modal, args, "test", local_rank)
    stepper, optimizer = None, None
    if args.fused_step:
        stepper = TrainStep(model, lr=args.lr, fine_tune_lr=args.fine_tune_lr, l2_weight=args.l2_weight,
                            fine_tune_l2_weight=args.fine_tune_l2_weight, pool_negatives=args.pool_negatives,
                            defer_update=os.environ.get("MOREC_DEFER_UPDATE", "1") != "0",      # (the epoch ends with a device synchronisation before eval / save)
                            graph=(world == 1 and bool(getattr(args, "graph", False))))
        if ckpt is not None and ckpt.get("optimizer") is not None:     # T/run.py:193-195
            stepper.load_optimizer_state_dict(ckpt["optimizer"])
        if ckpt is not None and ckpt.get("scaler_state"):              # (the reference never saves its GradScaler: an empty dict there)
            stepper.load_scaler_state_dict(ckpt["scaler_state"])
        wrapped = model
    else:
        wrapped = DDP(model, device_ids=[local_rank], output_device=local_rank, find_unused_parameters=True) if world > 1 else model
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        if vision:   # V/run.py:121-135
            tower = lambda n: "image_net" in n and not ("fc" in n or "classifier" in n)
        else:        # T/run.py:150-162
            tower = lambda n: "bert_model" in n
        groups = [{"params": [p for n, p in named if tower(n)], "lr": args.fine_tune_lr, "weight_decay": args.fine_tune_l2_weight},
                  {"params": [p for n, p in named if not tower(n)], "lr": args.lr, "weight_decay": args.l2_weight}]
        optimizer = optim.AdamW([g for g in groups if g["params"]])   # T/run.py:159-162
        if ckpt is not None and ckpt.get("optimizer") is not None:
            optimizer.load_state_dict(ckpt["optimizer"])
    # T/run.py:210: `scaler = torch.cuda.amp.GradScaler()` -- engaged for the fp16 compute dtype (bf16 / fp32 gradients need no scaling);
    # the fused step keeps the same protocol in its device block (TrainStep.sp)
    scaler = torch.amp.GradScaler("cuda", enabled=True) if (optimizer is not None and args.compute_dtype in ("fp16", "fp16_res32")) else None
    if scaler is not None and ckpt is not None and ckpt.get("scaler_state"):
        scaler.load_state_dict(ckpt["scaler_state"])
    best, step = 0.0, 0
    max_epoch, early_stop_epoch, early_stop_count = 0, args.epoch, 0
    early_stop_gap = 6 if vision else 10                               # T/run.py:221 / V/run.py:185
    on_device = hasattr(item_content, "device_batch")       # LMDB catalogue: the collate itself issues device work (decode -> H2D -> resize)
    n_workers = 0 if (bce or on_device) else max(0, int(getattr(args, "collate_workers", 0)))
    depth = 0 if on_device else int(getattr(args, "prefetch", 2))
    # host side of a batch: a picklable object over numpy tables and flags (no closure over the model / stepper)
    make_batch = BatchMaker(users, users_train, item_content, S, use_modal, bce=bce, item_num=item_num, neg_seed=777 + rank,
                            text_attrs=(stepper.text_attrs if (stepper is not None and use_modal and not vision) else None),
                            pad_to=512 if (stepper is not None and stepper.graph) else 0, pin=False if n_workers > 0 else None)
    loader, sampler = None, None
    if n_workers > 0:
        # T/run.py:111-124: DataLoader worker processes + its pin-memory thread; one "sample" = one whole collated batch.  ONE loader for
        # the run (persistent workers: no re-start per epoch); the start method is explicit -- "fork" by default (the item table is shared
        # copy-on-write; the workers never touch the HIP runtime), MOREC_LOADER_CONTEXT=forkserver | spawn for runtimes that mind a fork
        # of a process holding HIP / RCCL threads (everything the workers need pickles)
        sampler = _EpochSampler()
        loader = torch.utils.data.DataLoader(_EpochBatchSet(make_batch, len(users), args.batch_size, world, rank), batch_size=None, sampler=sampler,
                                             num_workers=n_workers, pin_memory=True, prefetch_factor=max(2, depth // max(1, n_workers)),
                                             persistent_workers=True, multiprocessing_context=os.environ.get("MOREC_LOADER_CONTEXT", "fork"))
    for ep in range(1, args.epoch + 1):
        now_epoch = start_epoch + ep
        model.train()
        # T/run.py:114,123-124,230: DistributedSampler(seed 0 + epoch, padded to a multiple of the world size) + a loader
        # without drop_last -- the last batch of an epoch is short
        batches = epoch_batches(len(users), args.batch_size, world, rank, now_epoch)
        feeder = None
        if loader is not None:
            sampler.set_epoch(now_epoch, len(batches))
            source = enumerate(loader)
        else:
            feeder = BatchPrefetcher(make_batch, batches, depth) if depth > 0 else None
            source = feeder if feeder is not None else ((b_, make_batch(idx_)) for b_, idx_ in enumerate(batches))
        t0, loss_acc, t_mark, n_mark = time.time(), None, None, 0
        b = -1
        # Vision catalogue held as uint8 arrays on the host (--images_npy / synthetic; not the LMDB, whose collate does device work itself):
        # the upload of the NEXT batch's images and their normalising patch im2col run on a second stream under the current step
        # (data_utils.images.DeviceImageFeed: what the reference's DataLoader workers + pin thread overlap, V/run.py:93-94,201-204)
        img_feed, img_ahead = None, None
        if vision and args.fused_step and not on_device and stepper is not None and not stepper.graph and not stepper.dedup_items:
            from .data_utils.images import DeviceImageFeed
            img_feed = DeviceImageFeed(local_rank, args.CV_resize, stepper.swin_shape.patch_size, stepper.dtype)
        for (b, (ids, items, log_mask, pack)), nxt in _with_next(source):
            if b == int(getattr(args, "steady_after", 10)):       # steady-state clock: starts once the first steps are behind us
                torch.cuda.synchronize()
                t_mark, n_mark = time.time(), 0
            n_mark += len(batches[b])
            if bce:
                items, log_mask = items.to(local_rank, non_blocking=True), log_mask.to(local_rank, non_blocking=True)
                items = items.view(-1, items.size(-1)) if use_modal else items.view(-1)
                optimizer.zero_grad()
                loss = wrapped(items, log_mask, local_rank)
                if scaler is not None:
                    scaler.scale(loss).backward()
                    scaler.step(optimizer)
                    scaler.update()
                else:
                    loss.backward()
                    optimizer.step()
                loss_acc = loss.detach() if loss_acc is None else loss_acc + loss.detach()
                step += 1
                if args.max_steps and step >= args.max_steps:
                    break
                continue
            pack = _to_device(pack, local_rank)
            img_slot = None
            if img_feed is not None and items.dtype == torch.uint8 and not items.is_cuda:
                if img_ahead is not None and img_ahead[0] == b:
                    img_slot = img_ahead[1]
                else:
                    img_slot = img_feed.submit(items.view(-1, *items.shape[-3:]), None, None)
                img_ahead = None
                if nxt is not None and nxt[1][1] is not None and nxt[1][1].dtype == torch.uint8:      # the next batch: queued before this step's launches
                    img_ahead = (nxt[0], img_feed.submit(nxt[1][1].view(-1, *nxt[1][1].shape[-3:]), None, None))
                items = img_feed.take(img_slot)
                ids, log_mask = ids.to(local_rank, non_blocking=True), log_mask.to(local_rank, non_blocking=True)
            else:
                ids, items, log_mask = (ids.to(local_rank, non_blocking=True), items.to(local_rank, non_blocking=True),
                                        log_mask.to(local_rank, non_blocking=True))
                if vision:
                    items = items.view(-1, *items.shape[-3:])                # [B*(S+1), R, R, 3] uint8 (V/run.py:203 views to NCHW floats)
                else:
                    items = items.view(-1, items.size(-1)) if use_modal else items.view(-1)
            if args.fused_step:
                loss = stepper.global_loss(stepper.step_graphed(ids.view(-1), items, log_mask, token_packing=pack))    # pooled negatives: a rank's step returns its SHARE; one rank: hipGraph replay per input shape
                if img_slot is not None:
                    img_feed.release(img_slot)
            else:
                optimizer.zero_grad()
                loss = wrapped(ids.view(-1), items, log_mask, local_rank)
                if scaler is not None:      # T/run.py:243-247
                    scaler.scale(loss).backward()
                    scaler.step(optimizer)
                    scaler.update()
                else:
                    loss.backward()
                    optimizer.step()
            loss_acc = loss.detach() if loss_acc is None else loss_acc + loss.detach()
            step += 1
            if args.max_steps and step >= args.max_steps:
                break
        if feeder is not None:
            feeder.close()
        torch.cuda.synchronize()
        t_end = time.time()
        dt = t_end - t0
        mean_loss = float(loss_acc.item()) / max(1, b + 1)
        if torch.isnan(torch.tensor(mean_loss)):                        # T/run.py:249-251
            Log.info("NaN loss, stopping")
            break
        n_seq = sum(len(x) for x in batches[:b + 1]) * world
        Log.info("epoch %d: %d steps, mean loss %.5f, %.1f user-seq/s" % (now_epoch, b + 1, mean_loss, n_seq / dt))
        if t_mark is not None and n_mark > 0:
            steady = n_mark * world / max(t_end - t_mark, 1e-9)
            train.last_steady_rate = steady          # (read by bench.py's run.py-vs-bench comparison)
            Log.info("epoch %d: steady state (after step %d): %.1f user-seq/s, %s" % (now_epoch, int(getattr(args, "steady_after", 10)), steady,
                                                                                    ("%d collate worker processes" % n_workers) if n_workers > 0 else ("collate thread, depth %d" % depth)))
            if feeder is not None and feeder.n:
                Log.info("collate thread: %.2f ms per batch; the training loop waited %.1f ms in total for batches" % (feeder.make_s / feeder.n * 1e3, feeder.wait_s * 1e3))
        if stepper is not None and stepper.sp is not None:
            h_ = stepper.sp.host()
            Log.info("loss scaler: scale %g, %d steps applied, %d skipped" % (h_.loss_scale, h_.step, h_.skipped))
        hit10 = run_eval(wrapped, item_content, hist_valid, users_valid, 512, item_num, use_modal, args, "valid", local_rank)
        need_break = False
        if hit10 > best:                                                # T/run.py:291-304
            best, max_epoch, early_stop_count = hit10, now_epoch, 0
            if use_modal and rank == 0:                                 # T/run.py:265-267: modal runs only, rank 0 only
                save_model(now_epoch, model, model_dir, stepper if stepper is not None else optimizer, torch.get_rng_state(),
                           torch.cuda.get_rng_state() if torch.cuda.is_available() else None,
                           stepper if (stepper is not None and stepper.sp is not None) else scaler, Log,      # loss-scaler state: exact fp16 resumption
                           extra={"morec_drop_calls": int(getattr(model, "_drop_calls", 0))})
        else:
            early_stop_count += 1
            if early_stop_count > early_stop_gap:
                need_break = is_early_stop
                early_stop_epoch = now_epoch
        if need_break or (args.max_steps and step >= args.max_steps):
            break
    Log.info("max eval Hit10 %.5f in epoch %d; early stop in epoch %d" % (best * 100, max_epoch, early_stop_epoch))
    return best


def main(argv=None):
    args = parse_args(argv)
    logging.basicConfig(level=logging.INFO, format="[%(levelname)s %(asctime)s] %(message)s")
    local_rank = args.local_rank if args.local_rank >= 0 else int(os.environ.get("LOCAL_RANK", 0))
    if os.environ.get("MOREC_DEVICE_INDEX") is not None:      # several ranks on ONE device (functional tests of the N > 1 driver over gloo)
        local_rank = int(os.environ["MOREC_DEVICE_INDEX"])
    torch.cuda.set_device(local_rank)
    if int(os.environ.get("WORLD_SIZE", 1)) > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL on ROCm (T/run.py:321: backend 'nccl'); MOREC_DIST_BACKEND=gloo is for tests that put two ranks on one GPU, which RCCL refuses
        dist.init_process_group(backend=os.environ.get("MOREC_DIST_BACKEND", "nccl"))
    setup_seed(12345)
    use_modal = "modal" in args.item_tower
    if dist.is_initialized() and dist.get_rank() != 0:
        Log.setLevel(logging.WARNING)
    best = train(args, use_modal, local_rank)
    Log.info("max eval Hit10 %.5f" % (best * 100))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
