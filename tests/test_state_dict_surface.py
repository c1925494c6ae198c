"""CPU: the drop-in Model exposes the reference's ``state_dict`` keys (order, names, shapes) and parameter order --
captured from the imported reference (tests/golden/g10_state_dict_keys.json) -- and reference-format checkpoints
round-trip through ``data_utils.utils.save_model`` / ``load_model``."""
import json
import os
import types

import torch

from idvs.morec_amd.data_utils.utils import load_model, save_model
from idvs.morec_amd.model import BertShape, HipBertModel, Model


def _args(**kw):
    d = dict(max_seq_len=20, embedding_dim=512, num_attention_heads=2, drop_rate=0.1, transformer_block=2,
             num_words_title=30, num_words_abstract=50, num_words_body=50, news_attributes=["title"],
             bert_model_load="bert_base_uncased", word_embedding_dim=768, compute_dtype="bf16")
    d.update(kw)
    return types.SimpleNamespace(**d)


def test_state_dict_keys_match_reference(golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "g10_state_dict_keys.json")))
    m = Model(_args(), 100, False, None, [1.0] * 101)
    assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == ref["id"]
    shape = BertShape.named("base")
    shape.num_hidden_layers = 2
    m = Model(_args(), 100, True, HipBertModel(shape), [1.0] * 101)
    assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == ref["modal_base_2layers"]
    assert [k for k, _ in m.named_parameters()] == ref["named_parameters_modal"]   # index-based freezing (T/run.py:73-75)


def test_checkpoint_roundtrip(tmp_path):
    shape = BertShape.named("micro")
    a = _args(embedding_dim=64, word_embedding_dim=64, max_seq_len=6)
    m1 = Model(a, 30, True, HipBertModel(shape), [1.0] * 31)
    opt = torch.optim.AdamW(m1.parameters(), lr=1e-3)
    wrapped = types.SimpleNamespace(module=m1)
    path = save_model(3, wrapped, str(tmp_path), opt, torch.get_rng_state(), None)
    assert os.path.basename(path) == "epoch-3.pt"
    ck = torch.load(path, weights_only=False)
    assert set(ck) == {"model_state_dict", "optimizer", "rng_state", "cuda_rng_state", "scaler_state"}
    ck["model_state_dict"]["bert_encoder.text_encoders.title.bert_model.embeddings.position_ids"] = torch.arange(64)[None]
    torch.save(ck, path)                               # what a transformers-4.20.1 era checkpoint carries
    m2 = Model(a, 30, True, HipBertModel(shape), [1.0] * 31)
    assert load_model(m2, path) == 3
    for (k1, v1), (k2, v2) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)


# ---------------------------------------------------------------------------------------------------------------- vision
def test_vision_state_dict_surface(golden_dir):
    """Keys, shapes and ORDER of the vision ``Model`` (``V/model/model.py``) with a Swin-T tower, as captured from the
    reference + installed HF (``tests/golden/make_golden_vision.py``): checkpoints and the index-based freezing of
    ``V/run.py:58-60`` depend on them."""
    import json
    import os
    import types
    from idvs.morec_amd.model import Model
    from idvs.morec_amd.model.swin import HipSwinForImageClassification
    from idvs.morec_amd.swin_engine import SwinShape
    with open(os.path.join(golden_dir, "g12_vision_keys.json")) as f:
        ref = json.load(f)
    args = types.SimpleNamespace(max_seq_len=10, embedding_dim=2048, num_attention_heads=2, drop_rate=0.1, transformer_block=2,
                                 CV_model_load="swin_tiny")
    kw = ref["swin_tiny_config"]
    shape = SwinShape(image_size=kw["image_size"], patch_size=kw["patch_size"], embed_dim=kw["embed_dim"], depths=tuple(kw["depths"]),
                      num_heads=tuple(kw["num_heads"]), window_size=kw["window_size"], mlp_ratio=kw["mlp_ratio"],
                      layer_norm_eps=kw["layer_norm_eps"], drop_path_rate=kw["drop_path_rate"])
    m = Model(args, 100, True, HipSwinForImageClassification(shape, 2048), [1.0] * 101)
    assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == ref["state_dict"]
    assert [k for k, _ in m.named_parameters()] == ref["named_parameters"]


import torch  # noqa: E402
from idvs.morec_amd.model.swin import HipSwinForImageClassification, remap_legacy_swin_keys  # noqa: E402
from idvs.morec_amd.swin_engine import SwinShape  # noqa: E402


def test_legacy_key_remap():
    shape = SwinShape.named("swin_micro")
    net = HipSwinForImageClassification(shape, 16)
    sd = net.state_dict()
    legacy = {}
    for k, v in sd.items():
        k = k.replace(".attention.q_proj.", ".attention.self.query.").replace(".attention.k_proj.", ".attention.self.key.")
        k = k.replace(".attention.v_proj.", ".attention.self.value.").replace(".attention.o_proj.", ".attention.output.dense.")
        k = k.replace(".attention.relative_position_bias.relative_position_bias_table", ".attention.self.relative_position_bias_table")
        k = k.replace(".mlp.fc1.", ".intermediate.dense.").replace(".mlp.fc2.", ".output.dense.")
        legacy[k] = v
    legacy["swin.encoder.layers.0.blocks.0.attention.self.relative_position_index"] = torch.zeros(49, 49)
    assert list(remap_legacy_swin_keys(legacy)) == list(sd)
