"""Item / user encoders with the reference's class names, constructor arguments, attribute names and
``forward`` signatures (``T/model/encoders.py:7-117``); arithmetic in ``libmorec_hip.so``."""
import os

import torch
import torch.nn as nn
from torch.nn.init import constant_, xavier_normal_

from .. import functional as F_
from ..engine import NO_DROP, DropCfg
from ..ops import FLT_MIN_MASK
from .bert import HipBertModel
from .swin import HipSwinForImageClassification
from .modules import TransformerEncoder


def resolve_dtype(args=None) -> torch.dtype:
    """Compute dtype: ``args.compute_dtype`` > env ``MOREC_DTYPE`` > bf16.  'fp32' = exact-fp32 MFMA (parity mode); 'fp32x3' = fp32
    tensors everywhere, the GEMMs as three bf16 MFMA passes over hi / lo splits of both operands (``resolve_fp32_gemm``)."""
    name = getattr(args, "compute_dtype", None) or os.environ.get("MOREC_DTYPE", "bf16")
    return {"fp32": torch.float32, "float32": torch.float32, "fp32x3": torch.float32, "bf16": torch.bfloat16,
            "bfloat16": torch.bfloat16, "fp16": torch.float16, "float16": torch.float16, "half": torch.float16,
            "fp16_res32": torch.float16, "bf16_res32": torch.bfloat16}[str(name)]


def resolve_res32(args=None) -> bool:
    """``compute_dtype`` "fp16_res32" / "bf16_res32": 16-bit GEMM operands and outputs with an fp32 RESIDUAL STREAM (LayerNorm takes the
    fp32 residual, returns fp32 and a rounded copy for the next GEMM) -- the data flow of the reference's ``torch.cuda.amp.autocast()``
    step (``T/run.py:242``; autocast runs layer_norm and softmax in fp32).  Text / ID towers; the Swin tower keeps 16-bit residuals."""
    name = getattr(args, "compute_dtype", None) or os.environ.get("MOREC_DTYPE", "bf16")
    return str(name).endswith("_res32")


def resolve_fp32_gemm(args=None) -> str:
    """How products of fp32 operands run for this model: ``ops.FP32_GEMM`` value ("exact" | "bf16x3")."""
    name = getattr(args, "compute_dtype", None) or os.environ.get("MOREC_DTYPE", "bf16")
    return "bf16x3" if str(name) == "fp32x3" else "exact"


class User_Encoder(nn.Module):
    def __init__(self, item_num, max_seq_len, item_dim, num_attention_heads, dropout, n_layers, compute_dtype=None, res32=None):
        super().__init__()
        self.transformer_encoder = TransformerEncoder(n_vocab=item_num, n_position=max_seq_len, d_model=item_dim,
                                                      n_heads=num_attention_heads, dropout=dropout, n_layers=n_layers)
        self.compute_dtype = compute_dtype or resolve_dtype()
        self.res32 = resolve_res32() if res32 is None else bool(res32)
        self.apply(self._init_weights)

    def _init_weights(self, module):   # T/model/encoders.py:15-21
        if isinstance(module, nn.Embedding):
            xavier_normal_(module.weight.data)
        elif isinstance(module, nn.Linear):
            xavier_normal_(module.weight.data)
            if module.bias is not None:
                constant_(module.bias.data, 0)

    def encode(self, input_embs, log_mask, drop: DropCfg = NO_DROP):
        """[B, S, D] -> [B, S, D] in the compute dtype (internal path of ``Model.forward``)."""
        names, params = zip(*self.named_parameters())
        te = self.transformer_encoder
        cfg = (names, te.n_heads, te.n_layers, self.compute_dtype, "transformer_encoder.", drop, self.res32)
        return F_.SasrecFn.apply(input_embs, log_mask, cfg, *params)

    def forward(self, input_embs, log_mask, local_rank=None):
        out = self.encode(input_embs, log_mask)
        return out if out.dtype == input_embs.dtype else out.to(input_embs.dtype)


class Text_Encoder(nn.Module):
    def __init__(self, bert_model, item_embedding_dim, word_embedding_dim, compute_dtype=None, res32=None):
        super().__init__()
        self.bert_model = bert_model if isinstance(bert_model, HipBertModel) else HipBertModel.from_hf(bert_model)
        self.fc = nn.Linear(word_embedding_dim, item_embedding_dim)
        self.compute_dtype = compute_dtype or resolve_dtype()
        self.res32 = resolve_res32() if res32 is None else bool(res32)
        self.mask_value = FLT_MIN_MASK   # transformers >= 4.3x eager; set to -10000.0 for 4.20.1 behaviour

    def encode(self, text, drop: DropCfg = NO_DROP):
        c = self.bert_model.config
        named = [(n, p) for n, p in self.named_parameters() if ".pooler." not in n]
        names, params = zip(*named)
        cfg = (names, c.num_attention_heads, c.num_hidden_layers, self.compute_dtype, "", c.layer_norm_eps, self.mask_value,
               drop, self.res32)
        return F_.BertEncoderFn.apply(text, cfg, *params)

    def forward(self, text):
        return self.encode(text).float()


class Bert_Encoder(nn.Module):
    def __init__(self, args, bert_model):
        super().__init__()
        self.args = args
        self.attributes2length = {'title': args.num_words_title * 2, 'abstract': args.num_words_abstract * 2,
                                  'body': args.num_words_body * 2}
        for key in list(self.attributes2length.keys()):
            if key not in args.news_attributes:
                self.attributes2length[key] = 0
        self.attributes2start = {key: sum(list(self.attributes2length.values())[:list(self.attributes2length.keys()).index(key)])
                                 for key in self.attributes2length.keys()}
        assert len(args.news_attributes) > 0
        if 'opt' in args.bert_model_load:
            raise NotImplementedError("OPT mean-pooling encoder (T/model/encoders.py:31-50) is outside the hot path")
        self.text_encoders = nn.ModuleDict({
            'title': Text_Encoder(bert_model, args.embedding_dim, args.word_embedding_dim, resolve_dtype(args), resolve_res32(args))})
        # (the reference iterates a set intersection, i.e. in hash order; the mean over attributes does not depend on it -- a fixed order
        # keeps the fp32 sum of three passes identical across processes)
        self.newsname = [name for name in ('title', 'abstract', 'body') if name in set(args.news_attributes)]

    def encode(self, news, drop: DropCfg = NO_DROP):
        vecs = [self.text_encoders['title'].encode(
            torch.narrow(news, 1, self.attributes2start[name], self.attributes2length[name]).contiguous(), drop.stream(j))
            for j, name in enumerate(self.newsname)]
        return vecs[0] if len(vecs) == 1 else torch.mean(torch.stack(vecs, dim=1), dim=1)

    def forward(self, news):
        return self.encode(news).float()


class Vit_Encoder(nn.Module):
    """``V/model/encoders.py:24-31``: ``GELU(image_net(item_content)[0])`` with ``image_net`` a Swin classifier whose head
    was replaced by ``Linear(num_features, embedding_dim)`` (``V/run.py:47-54``).  BEiT shares the class in the reference;
    only Swin (the benchmarked tower, ``V/train_swin_tiny.py``) is implemented."""

    def __init__(self, image_net, compute_dtype=None):
        super().__init__()
        self.image_net = image_net if isinstance(image_net, HipSwinForImageClassification) \
            else HipSwinForImageClassification.from_hf(image_net)
        self.compute_dtype = compute_dtype or resolve_dtype()

    def encode(self, item_content, drop: DropCfg = NO_DROP):
        named = [(n, p) for n, p in self.named_parameters()]
        names, params = zip(*named)
        cfg = (names, self.image_net.shape, self.compute_dtype, "image_net.", drop, self.training)
        return F_.SwinEncoderFn.apply(item_content, cfg, *params)

    def forward(self, item_content):
        return self.encode(item_content).float()


class IdEmbedding(nn.Module):
    """``nn.Embedding(item_num + 1, D, padding_idx=0)`` (``T/model/model.py:27``) backed by the HIP gather / scatter-add."""

    def __init__(self, num_embeddings, embedding_dim, padding_idx=0, compute_dtype=None):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(num_embeddings, embedding_dim))
        self.padding_idx = padding_idx
        self.compute_dtype = compute_dtype or resolve_dtype()

    def encode(self, ids):
        return F_.IdEmbeddingFn.apply(ids, self.weight, self.compute_dtype)

    def forward(self, ids):
        return F_.IdEmbeddingFn.apply(ids, self.weight, torch.float32)
