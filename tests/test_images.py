"""Vision input pipeline.  CPU: the oracle and the host tables against Pillow's own outputs (golden g16, written by
tests/golden/make_golden_images.py with the routine torchvision's Resize calls), the LMDB record format of
dataset/HM/build_lmdb_hm.py.  -m gpu: the device resampler bit for bit against the same goldens."""
import os
import pickle
import sys

import numpy as np
import pytest
import torch

from helpers import GOLDEN_DIR
from idvs.morec_amd.data_utils import LmdbImageStore, decode_record, pack_images, read_images, resize_table

G = np.load(os.path.join(GOLDEN_DIR, "g16_image_resize.npz"))


def test_oracle_resize_equals_pillow():
    from morec_oracle.image_ref import pil_bilinear_resize
    for i, (H, W, R) in enumerate(G["cases"]):
        np.testing.assert_array_equal(pil_bilinear_resize(G[f"in{i}"], int(R)), G[f"out{i}"])


def test_host_tap_tables_equal_the_oracle():
    from morec_oracle.image_ref import coeffs
    for n_in, n_out in [(37, 24), (24, 24), (11, 24), (200, 56), (640, 224), (224, 224), (100, 224)]:
        tab = resize_table(n_in, n_out)
        b, k = coeffs(n_in, n_out)
        ks = int(tab[0])
        assert ks == k.shape[1]
        rows = tab[1:].reshape(n_out, 2 + ks)
        np.testing.assert_array_equal(rows[:, :2], b)
        np.testing.assert_array_equal(rows[:, 2:], k)


class _HmRecord:          # stands in for the builder's class: pickled under ANOTHER module path, like `__main__.LMDB_Image`
    pass


def _builder_blob(img, item_id):
    """What dataset/HM/build_lmdb_hm.py:47-49 stores: pickle.dumps(LMDB_Image(img, item_id)) with the class living in __main__."""
    import types
    mod = types.ModuleType("__main__hm_builder__")

    class LMDB_Image:
        def __init__(self, image, id):
            self.channels = image.shape[2]
            self.size = image.shape[:2]
            self.image = image.tobytes()
            self.id = id
    LMDB_Image.__module__ = mod.__name__
    LMDB_Image.__qualname__ = "LMDB_Image"
    mod.LMDB_Image = LMDB_Image
    sys.modules[mod.__name__] = mod
    try:
        return pickle.dumps(LMDB_Image(img, item_id))
    finally:
        del sys.modules[mod.__name__]


def test_lmdb_image_records_and_store(tmp_path):
    imgs = {108775015: G["in0"], 108775044: G["in3"], 110065001: G["in5"]}
    kv = {str(k).encode("ascii"): _builder_blob(v, k) for k, v in imgs.items()}
    keys = list(kv)
    kv[b"__keys__"], kv[b"__len__"] = pickle.dumps(keys), pickle.dumps(len(keys))      # build_lmdb_hm.py:58-61
    for k, v in imgs.items():
        np.testing.assert_array_equal(decode_record(kv[str(k).encode("ascii")]), v)   # the builder's module is gone: remapped
    store = LmdbImageStore(backend=kv)
    assert len(store) == 3 and store.keys == keys
    batch = store.batch([keys[1], None, keys[0]])
    np.testing.assert_array_equal(batch[0], G["in3"])
    assert batch[1].shape == (1, 1, 3)
    # item list -> ids and keys (V/data_utils/preprocess.py:88-101)
    p = tmp_path / "items.tsv"
    p.write_text("v108775015\nv108775044\nv110065001\n")
    id2key, name2id, id2name = read_images(str(p))
    assert id2key == {1: b"108775015", 2: b"108775044", 3: b"110065001"} and name2id["v110065001"] == 3 and id2name[2] == "v108775044"
    with pytest.raises(ImportError):
        LmdbImageStore(str(tmp_path / "absent.lmdb"))           # no `lmdb` module in this image: says so instead of guessing


def test_pack_images_layout():
    flat, meta, tabs = pack_images([G["in0"], G["in1"], G["in0"]], 24)
    assert meta.shape == (3, 5) and meta[0, 0] == 0 and meta[1, 0] == G["in0"].size and meta[2, 0] == G["in0"].size + G["in1"].size
    assert (meta[0, 3:] == meta[2, 3:]).all()                    # one table per distinct size
    H, W = G["in0"].shape[:2]
    np.testing.assert_array_equal(tabs[meta[0, 3]:meta[0, 3] + resize_table(W, 24).size], resize_table(W, 24))
    np.testing.assert_array_equal(tabs[meta[0, 4]:meta[0, 4] + resize_table(H, 24).size], resize_table(H, 24))


@pytest.mark.gpu
def test_device_resize_is_pillow_bit_for_bit():
    from idvs.morec_amd import ops
    by_R = {}
    for i, (H, W, R) in enumerate(G["cases"]):
        by_R.setdefault(int(R), []).append(i)
    for R, idx in by_R.items():
        out = ops.image_resize_u8([G[f"in{i}"] for i in idx], R).cpu().numpy()
        for j, i in enumerate(idx):
            np.testing.assert_array_equal(out[j], G[f"out{i}"])


@pytest.mark.gpu
def test_device_resize_random_sizes_vs_oracle_and_into_the_encoder():
    """Random sizes to the launcher's R = 224 (V/parameters.py:36), against the oracle; the result feeds the Swin tower's uint8 entry
    (ToTensor + Normalize fused in patchify): same vectors as the host-normalised float path of V/run.py:201-204."""
    from idvs.morec_amd import ops
    from morec_oracle.image_ref import pil_bilinear_resize, to_tensor_normalize
    rng = np.random.default_rng(5)
    imgs = [rng.integers(0, 256, (int(rng.integers(40, 400)), int(rng.integers(40, 400)), 3), dtype=np.uint8) for _ in range(6)]
    out = ops.image_resize_u8(imgs, 224)
    for j, im in enumerate(imgs):
        np.testing.assert_array_equal(out[j].cpu().numpy(), pil_bilinear_resize(im, 224))
    host = torch.from_numpy(np.stack([to_tensor_normalize(pil_bilinear_resize(im, 224)) for im in imgs]))
    a = ops.swin_patchify_u8(out, 4, torch.float32)
    b = ops.swin_patchify(host.contiguous().cuda(), 4, torch.float32)
    np.testing.assert_array_equal(a.cpu().numpy(), b.cpu().numpy())


@pytest.mark.gpu
def test_device_image_feed_equals_the_inline_pipeline():
    """``DeviceImageFeed`` (H2D + resize + normalising im2col of the NEXT batch on its own stream, double-buffered) hands the step bit for
    bit the patch rows the inline path computes on the step's stream, batch after batch (buffer sets reused, sizes changing), and a
    fused train step fed through it returns the inline step's loss."""
    import types
    from idvs.morec_amd import ops
    from idvs.morec_amd.data_utils.images import DeviceImageFeed, pack_images
    rng = np.random.default_rng(9)
    feed = DeviceImageFeed("cuda", 56, 4, torch.float16)
    batches = [[rng.integers(0, 256, (int(rng.integers(40, 120)), int(rng.integers(40, 120)), 3), dtype=np.uint8) for _ in range(n)]
               for n in (5, 9, 5, 3, 9)]
    packed = [tuple(torch.from_numpy(x).pin_memory() for x in pack_images(b, 56)) for b in batches]
    slot = feed.submit(*packed[0])
    for k in range(len(batches)):
        nxt = feed.submit(*packed[k + 1]) if k + 1 < len(batches) else None      # queued before batch k is consumed, as in the training loop
        rows = feed.take(slot)
        ref = ops.swin_patchify_u8(ops.image_resize_u8_packed(*packed[k], 56, "cuda"), 4, torch.float16)
        assert rows.n_img == len(batches[k]) and torch.equal(rows.patches, ref)
        feed.release(slot)
        slot = nxt
    # images that are already R x R (run.py's catalogue arrays): upload + im2col only
    arr = torch.from_numpy(rng.integers(0, 256, (7, 56, 56, 3), dtype=np.uint8)).pin_memory()
    s2 = feed.submit(arr, None, None)
    rows = feed.take(s2)
    assert torch.equal(rows.patches, ops.swin_patchify_u8(arr.cuda(), 4, torch.float16))
    feed.release(s2)
    # through a fused step
    from idvs.morec_amd.model import Model
    from idvs.morec_amd.model.swin import HipSwinForImageClassification
    from idvs.morec_amd.swin_engine import SwinShape
    from idvs.morec_amd.train_step import TrainStep
    S, D, item_num, B = 4, 64, 30, 3
    vshape = SwinShape.named("swin_micro")
    args = types.SimpleNamespace(max_seq_len=S, embedding_dim=D, num_attention_heads=2, drop_rate=0.0, transformer_block=2,
                                 CV_model_load="swin_micro", compute_dtype="fp16")
    pop = np.full(item_num + 1, 1.0 / item_num)
    pop[0] = 1.0
    ids = torch.from_numpy(rng.integers(1, item_num + 1, (B, S + 1))).cuda()
    imgs = torch.from_numpy(rng.integers(0, 256, (B * (S + 1), vshape.image_size, vshape.image_size, 3), dtype=np.uint8)).pin_memory()
    losses = []
    for fed in (False, True):
        torch.manual_seed(3)
        m = Model(args, item_num, True, HipSwinForImageClassification(vshape, D), pop).cuda().eval()
        ts = TrainStep(m, lr=1e-3, fine_tune_lr=1e-3, l2_weight=0.0, fine_tune_l2_weight=0.0, pool_negatives=False, loss_scale=256.0)
        if fed:
            f2 = DeviceImageFeed("cuda", vshape.image_size, vshape.patch_size, torch.float16)
            sl = f2.submit(imgs, None, None)
            loss = ts.step(ids.view(-1), f2.take(sl), torch.ones(B, S, device="cuda"))
            f2.release(sl)
        else:
            loss = ts.step(ids.view(-1), imgs.cuda(), torch.ones(B, S, device="cuda"))
        losses.append(float(loss))
    assert losses[0] == losses[1], losses
