"""Host half of the vision input pipeline: the reference's on-disk image format and the tables the device-side
resize needs.

* ``LmdbImageStore`` reads what ``dataset/HM/build_lmdb_hm.py:13-66`` writes: an LMDB whose values are pickled ``LMDB_Image``
  objects (``channels``, ``size`` = (H, W), ``image`` = raw RGB bytes, ``id``) keyed by the ascii item id, plus ``__keys__`` /
  ``__len__``; ``V/data_utils/dataset.py:61-66,91-97`` opens it read-only and unpickles one record per sequence slot.
  The ``lmdb`` module is optional (it is not part of this image): any object with ``get(key) -> bytes`` can stand in.
* ``resize_table`` / ``pack_images``: what ``ops.image_resize_u8`` (``morec_image_resize_u8``) consumes -- the reference resizes
  each image on the host with ``tv.transforms.Resize((R, R))`` (Pillow BILINEAR), here the DECODED uint8 images cross PCIe at
  their native size and are resampled on the GPU with the same fixed-point taps."""
from __future__ import annotations

import io
import math
import os
import pickle

import numpy as np

PRECISION_BITS = 32 - 8 - 2          # Pillow Resample.c: 8-bit images


class LMDB_Image:
    """Field-for-field the record class of ``dataset/HM/build_lmdb_hm.py:13-22`` / ``V/data_utils/dataset.py:16-25``."""

    def __init__(self, image, id):
        self.channels = image.shape[2]
        self.size = image.shape[:2]
        self.image = image.tobytes()
        self.id = id

    def get_image(self):
        image = np.frombuffer(self.image, dtype=np.uint8)
        return image.reshape(*self.size, self.channels)


class _RecordUnpickler(pickle.Unpickler):
    """The builder pickles ``__main__.LMDB_Image`` (it runs as a script) and the reference unpickles with that name imported into
    ITS ``__main__``; here any ``LMDB_Image`` resolves to the class above, wherever it was defined."""

    def find_class(self, module, name):
        if name == "LMDB_Image":
            return LMDB_Image
        return super().find_class(module, name)


def decode_record(blob: bytes) -> np.ndarray:
    """One LMDB value -> uint8 [H, W, 3] (``IMAGE.get_image()``; the reference then does ``.convert('RGB')`` on an RGB array)."""
    rec = _RecordUnpickler(io.BytesIO(blob)).load()
    img = rec.get_image()
    if img.ndim != 3 or img.shape[2] != 3:
        raise ValueError(f"record {getattr(rec, 'id', '?')}: expected RGB, got shape {img.shape}")
    return img


class LmdbImageStore:
    """``V/data_utils/dataset.py:61-66``: ``lmdb.open(db_path, subdir=isdir, readonly=True, lock=False, readahead=False,
    meminit=False)``; ``__len__`` / ``__keys__`` read once."""

    def __init__(self, db_path=None, backend=None):
        if backend is None:
            try:
                import lmdb
            except ImportError as e:            # noqa: F841
                raise ImportError("the `lmdb` module is needed to open an image database (pip package `lmdb`); it is not part of this "
                                  "image -- pass backend=<object with get(key) -> bytes> or use --images_npy") from None
            self._env = lmdb.open(db_path, subdir=os.path.isdir(db_path), readonly=True, lock=False, readahead=False, meminit=False)
            self._txn = self._env.begin()
            backend = self._txn
        self._kv = backend
        self.length = pickle.loads(self._kv.get(b"__len__"))
        self.keys = pickle.loads(self._kv.get(b"__keys__"))

    def __len__(self):
        return self.length

    def image(self, key: bytes) -> np.ndarray:
        blob = self._kv.get(key)
        if blob is None:
            raise KeyError(key)
        return decode_record(blob)

    def batch(self, keys) -> list:
        """Decoded images of one training batch; ``None`` key (padding slot: the reference leaves an all-zero tensor there,
        ``V/data_utils/dataset.py:88``) -> a 1 x 1 mid-grey image (the slot reaches nothing in the loss)."""
        pad = np.full((1, 1, 3), 128, dtype=np.uint8)
        return [pad if k is None else self.image(k) for k in keys]


def _bilinear(x: float) -> float:
    x = -x if x < 0.0 else x
    return 1.0 - x if x < 1.0 else 0.0


_TABLES = {}


def resize_table(in_size: int, out_size: int) -> np.ndarray:
    """Pillow ``precompute_coeffs`` + ``normalize_coeffs_8bpc`` (BILINEAR, full box) as one int32 array: ``[ksize]`` followed by
    ``out_size`` rows ``(first input index, tap count, taps[ksize])``.  Python floats are IEEE doubles, the arithmetic is written
    in Pillow's order; cached per (in_size, out_size)."""
    key = (int(in_size), int(out_size))
    t = _TABLES.get(key)
    if t is not None:
        return t
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    tab = np.zeros(1 + out_size * (2 + ksize), dtype=np.int32)
    tab[0] = ksize
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_bilinear((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        row = 1 + xx * (2 + ksize)
        tab[row], tab[row + 1] = xmin, xmax
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            tab[row + 2 + x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
    _TABLES[key] = tab
    return tab


def pack_images(images, R: int):
    """list of uint8 [H, W, 3] arrays -> (flat uint8 bytes, int64 meta [n, 5], int32 tables) for ``morec_image_resize_u8``."""
    tabs, tab_off, off = [], {}, 0

    def table(size):
        nonlocal off
        if size not in tab_off:
            t = resize_table(size, R)
            tab_off[size] = off
            tabs.append(t)
            off += t.size
        return tab_off[size]

    meta = np.zeros((len(images), 5), dtype=np.int64)
    pos = 0
    for i, im in enumerate(images):
        if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3:
            raise ValueError(f"image {i}: expected uint8 [H, W, 3], got {im.dtype} {im.shape}")
        H, W = im.shape[:2]
        meta[i] = (pos, H, W, table(W), table(H))
        pos += H * W * 3
    flat = np.empty(pos, dtype=np.uint8)
    for i, im in enumerate(images):
        flat[meta[i, 0]:meta[i, 0] + im.size] = im.reshape(-1)
    return flat, meta, np.concatenate(tabs)


class LmdbItemImages:
    """The item catalogue of a vision run read straight from the reference's LMDB: indexable by item id like the ``item_content``
    arrays of the other towers, but an access decodes the records and resamples them ON THE DEVICE (``ops.image_resize_u8``).
    ``item_id_to_keys``: ``read_images`` / ``read_behaviors`` output (id 0 = padding item, no key)."""

    def __init__(self, store: LmdbImageStore, item_id_to_keys: dict, R: int):
        self.store, self.keys, self.R = store, item_id_to_keys, R
        self.n = max(item_id_to_keys) + 1 if item_id_to_keys else 1

    def __len__(self):
        return self.n

    def device_batch(self, ids, device):
        """item ids (any shape, numpy) -> uint8 [*ids.shape, R, R, 3] on ``device``."""
        from .. import ops
        flat = np.asarray(ids).reshape(-1)
        # id 0 = the padding slot (no image; masked everywhere in the loss); any OTHER id without an LMDB key is a corrupt / mismatched
        # item list and raises KeyError like the reference's ``self.item_id_to_keys[item_id]`` (V/data_utils/dataset.py:93,162)
        imgs = self.store.batch([self.keys[int(i)] if int(i) != 0 else None for i in flat])
        out = ops.image_resize_u8(imgs, self.R, device)
        return out.view(*np.asarray(ids).shape, self.R, self.R, 3)


class PatchRows:
    """A batch of images that has already been through the normalising patch im2col (``morec_swin_patchify_u8``): what
    ``swin_engine.swin_forward`` accepts in place of the pixel tensor.  ``patches`` [n_img (R / p)^2, 3 p^2] in the compute dtype."""

    def __init__(self, patches, n_img):
        self.patches, self.n_img = patches, int(n_img)
        self.device, self.shape = patches.device, (int(n_img),)


class DeviceImageFeed:
    """Device half of the vision input pipeline, off the training step's stream.  The reference's DataLoader workers decode + resize
    the NEXT batch's images while the GPU trains on the current one, and the main loop only uploads tensors (``V/run.py:93-94,201-204``,
    ``V/data_utils/dataset.py:61-99``).  Here the decoded uint8 images cross PCIe and are resampled / normalised on the GPU, so the same
    overlap is a second HIP stream: ``submit`` queues, for batch k + 1, the H2D copy of the packed bytes, ``morec_image_resize_u8`` and
    ``morec_swin_patchify_u8`` on the feed's own stream into one of ``depth`` buffer sets, all under step k; ``take`` makes the step's
    stream wait for that set (one event) and hands out ``PatchRows``; ``release`` (after the step's launches) marks the set reusable.
    Buffers are allocated once per size on the caller's stream and reused -- nothing crosses between the streams' allocator pools."""

    def __init__(self, device, R: int, patch: int, dtype, depth: int = 2):
        import torch
        self.dev, self.R, self.patch, self.dtype = torch.device(device), int(R), int(patch), dtype
        self.stream = torch.cuda.Stream(device=self.dev)
        self.slots = [dict(src=None, imgs=None, patches=None, ready=None, free=None, n=0) for _ in range(max(2, depth))]
        self.k = 0

    def _buf(self, slot, key, shape, dtype):
        import torch
        need = 1
        for v in shape:
            need *= int(v)
        t = slot[key]
        if t is None or t.numel() < need or t.dtype != dtype:
            t = slot[key] = torch.empty(need + need // 8, device=self.dev, dtype=dtype)
            slot["fresh"] = True      # a block the step stream's allocator may just have recycled: see submit()
        return t[:need].view(*shape)

    def submit(self, flat, meta, tabs=None):
        """Queue one batch on the feed's stream.  ``flat, meta, tabs`` = ``pack_images`` output as (page-locked) CPU tensors; with
        ``tabs=None``, ``flat`` is a uint8 [n, R, R, 3] CPU tensor of images that are already R x R (no resize: upload + im2col only)."""
        import torch
        from .. import ops
        slot = self.slots[self.k % len(self.slots)]
        self.k += 1
        n = int(flat.shape[0]) if tabs is None else int(meta.shape[0])
        G = self.R // self.patch
        # buffers first, on the CALLER's stream (allocator pool of the step), then the work on the feed's stream
        imgs = self._buf(slot, "imgs", (n, self.R, self.R, 3), torch.uint8)
        patches = self._buf(slot, "patches", (n * G * G, 3 * self.patch * self.patch), self.dtype)
        src = None if tabs is None else self._buf(slot, "src", (int(flat.numel()),), torch.uint8)
        if slot["free"] is not None:      # the step that read this set last must be done with it
            self.stream.wait_event(slot["free"])
        if slot["free"] is None or slot.pop("fresh", False):
            # first use, or a buffer that has just been (re)allocated from the STEP stream's pool (a later batch outgrew it): the block may
            # be one that kernels still queued on the step stream are reading -- stream-ordered reuse only holds on that stream -- so the
            # feed waits for everything the step stream has queued so far, not only for the set's last reader
            self.stream.wait_stream(torch.cuda.current_stream(self.dev))
        slot.pop("fresh", None)
        with torch.cuda.stream(self.stream):
            if tabs is None:
                imgs.copy_(flat, non_blocking=True)
            else:
                ops.image_resize_u8_packed(flat, meta, tabs, self.R, self.dev, out=imgs, src_buf=src)
            ops.swin_patchify_u8(imgs, self.patch, self.dtype, out=patches)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        slot["ready"], slot["n"], slot["host"] = ev, n, (flat, meta, tabs)      # (the host tensors stay referenced until the set is reused)
        slot["cur"] = (imgs, patches)
        return slot

    def take(self, slot) -> PatchRows:
        import torch
        torch.cuda.current_stream(self.dev).wait_event(slot["ready"])
        return PatchRows(slot["cur"][1], slot["n"])

    def release(self, slot):
        import torch
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev))
        slot["free"] = ev
