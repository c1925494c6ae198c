"""CPU: host logic added in round 2 -- the reference's sampler semantics and the frozen-prefix bookkeeping."""
import torch
from torch.utils.data.distributed import DistributedSampler

from idvs.morec_amd.data_utils import epoch_batches
from idvs.morec_amd.engine import TE, bert_grad_from, bert_needs_grad_buffer


def test_epoch_batches_are_the_reference_sampler_and_loader():
    """T/run.py:114,123-124,230: DistributedSampler(dataset) (shuffle, seed 0 + epoch, padded to a multiple of the world size) cut
    by DataLoader(batch_size, no drop_last): every rank sees ceil(n / world) samples per epoch and a SHORT last batch."""
    n, B, world = 103, 8, 4
    for epoch in (1, 2, 7):
        seen = []
        for rank in range(world):
            batches = epoch_batches(n, B, world, rank, epoch)
            s = DistributedSampler(range(n), num_replicas=world, rank=rank)     # what the reference constructs
            s.set_epoch(epoch)
            want = list(iter(s))
            dl = torch.utils.data.DataLoader(range(n), batch_size=B, sampler=s)
            assert [list(map(int, b)) for b in dl] == batches
            assert sum(len(b) for b in batches) == len(want) == 26 and len(batches[-1]) == 26 % B
            seen += [i for b in batches for i in b]
        assert len(seen) == 104 and set(seen) == set(range(n))                 # padded by repeating one index
    assert epoch_batches(n, B, world, 0, 1) != epoch_batches(n, B, world, 0, 2)    # set_epoch reshuffles


def test_bert_grad_from_follows_the_freeze_index():
    bm = TE + "bert_model."
    emb = [bm + "embeddings.word_embeddings.weight", bm + "embeddings.LayerNorm.bias"]
    lay = lambda l: [bm + f"encoder.layer.{l}.attention.self.query.weight", bm + f"encoder.layer.{l}.output.LayerNorm.bias"]
    head = [TE + "fc.weight", "user_encoder.transformer_encoder.layer_norm.weight"]
    assert bert_grad_from(emb + lay(0) + lay(11) + head, 12) == -1          # --freeze_paras_before 0 (the launcher's value)
    assert bert_grad_from(lay(10) + lay(11) + head, 12) == 10               # 165 = 5 + 16 * 10 (the parser's default)
    assert bert_grad_from(head + [bm + "pooler.dense.weight"], 12) == 12    # nothing inside bert_model trains
    assert not bert_needs_grad_buffer(emb[0], 10) and not bert_needs_grad_buffer(lay(9)[0], 10)
    assert bert_needs_grad_buffer(lay(10)[0], 10) and bert_needs_grad_buffer(head[0], 10) and bert_needs_grad_buffer(emb[0], -1)
    assert not bert_needs_grad_buffer(bm + "pooler.dense.bias", -1)
