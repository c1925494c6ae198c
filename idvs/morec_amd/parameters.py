"""Command-line surface of ``T/parameters.py`` (same flags and defaults) plus what the reference forgot or
what ROCm / torchrun need: ``--news`` (read by ``T/run.py:79,100`` but never declared), ``--local-rank`` (torch >= 2.0
launcher spelling), and the MI355X-path switches."""
import argparse
import os


def build_parser():
    p = argparse.ArgumentParser()
    # ============== data_dir ==============
    p.add_argument("--mode", type=str, default="train")
    p.add_argument("--item_tower", type=str, default="id")
    p.add_argument("--root_data_dir", type=str, default="../")
    p.add_argument("--dataset", type=str, default="MIND-small")
    p.add_argument("--behaviors", type=str, default="behaviors_l5_tr_v.tsv")
    p.add_argument("--news", type=str, default="news.tsv")
    # ============== train parameters ==============
    p.add_argument("--batch_size", type=int, default=64)
    p.add_argument("--epoch", type=int, default=1)
    p.add_argument("--lr", type=float, default=1e-5)
    p.add_argument("--fine_tune_lr", type=float, default=1e-5)
    p.add_argument("--l2_weight", type=float, default=0)
    p.add_argument("--fine_tune_l2_weight", type=float, default=0)
    p.add_argument("--drop_rate", type=float, default=0.1)
    # ============== model parameters ==============
    p.add_argument("--bert_model_load", type=str, default="bert-base-uncased")
    p.add_argument("--freeze_paras_before", type=int, default=165)
    p.add_argument("--word_embedding_dim", type=int, default=768)
    p.add_argument("--embedding_dim", type=int, default=256)
    p.add_argument("--num_attention_heads", type=int, default=2)
    p.add_argument("--transformer_block", type=int, default=2)
    p.add_argument("--max_seq_len", type=int, default=20)
    p.add_argument("--min_seq_len", type=int, default=5)
    # ============== switch and logging setting ==============
    p.add_argument("--num_workers", type=int, default=12)
    p.add_argument("--load_ckpt_name", type=str, default="None")
    p.add_argument("--label_screen", type=str, default="None")
    p.add_argument("--logging_num", type=int, default=8)
    p.add_argument("--testing_num", type=int, default=1)
    p.add_argument("--local_rank", "--local-rank", dest="local_rank", default=int(os.environ.get("LOCAL_RANK", -1)), type=int)
    # ============== news information ==============
    p.add_argument("--num_words_title", type=int, default=30)
    p.add_argument("--num_words_abstract", type=int, default=50)
    p.add_argument("--num_words_body", type=int, default=50)
    p.add_argument("--news_attributes", type=str, default="title")
    # ============== vision variant (V/parameters.py:34-39) ==============
    p.add_argument("--CV_model_load", type=str, default="None", help="swin_tiny | swin_small | swin_base (vision item tower)")
    p.add_argument("--CV_resize", type=int, default=224)
    p.add_argument("--image_lmdb", type=str, default="None", help="LMDB of pickled LMDB_Image records (dataset/HM/build_lmdb_hm.py; "
                   "V/parameters.py --lmdb_data): decoded per batch, resized on the GPU; needs the optional `lmdb` module")
    p.add_argument("--images_npy", type=str, default="None",
                   help="uint8 array [item_num + 1, R, R, 3] of decoded, resized item images (row 0 = padding item); stands in for the "
                        "LMDB reader of V/data_utils/dataset.py, whose lmdb / torchvision dependencies are not part of this package")
    # ============== MI355X path ==============
    p.add_argument("--compute_dtype", type=str, default="fp16", choices=["bf16", "fp16", "fp32", "fp32x3", "fp16_res32", "bf16_res32"],
                   help="fp16 (default) = IEEE half operands on the MFMA, fp32 accumulate, with the reference's GradScaler loss scaling -- the arithmetic of "
                        "its own autocast step (T/run.py:210,242-247; V/run.py likewise) and what bench.py times; bf16 = the same kernels on bf16 operands, no "
                        "loss scaling; fp32 = exact-fp32 MFMA parity mode; fp32x3 = fp32 tensors with every GEMM as three bf16 MFMA passes over hi / lo "
                        "operand splits (fp32 nn.Linear numerics to ~1e-5); fp16_res32 / bf16_res32 = 16-bit GEMMs with an fp32 residual stream "
                        "(LayerNorm in and out fp32): the data flow of torch.cuda.amp.autocast itself, text / ID towers")
    p.add_argument("--pool_negatives", action="store_true", help="pool in-batch negatives over ranks (RCCL all-gather)")
    p.add_argument("--fused_step", action="store_true",
                   help="flat-arena TrainStep (fused AdamW, one gradient all-reduce) instead of DDP + torch.optim.AdamW")
    p.add_argument("--loss", type=str, default="inbatch", choices=["inbatch", "bce"],
                   help="inbatch = debiased in-batch softmax CE (inbatch_sasrec_e2e_*); bce = one sampled negative per position "
                        "(bce_text/main-end2end; drop-in autograd path only)")
    p.add_argument("--synthetic", type=int, default=0, help="N > 0: train on N synthetic MIND-shaped users (no data files)")
    p.add_argument("--synthetic_items", type=int, default=20000)
    p.add_argument("--synthetic_full_len", action="store_true", help="synthetic users all have raw history max_seq_len + 3 (train sequences "
                   "of exactly S + 1 items, no padding): the shape bench.py times (SURVEY.md §8d)")
    p.add_argument("--max_steps", type=int, default=0)
    p.add_argument("--collate_workers", type=int, default=2, help="worker PROCESSES that build the batches (collate + unpadded-layout index vectors) "
                   "through torch's DataLoader with pin_memory, as T/run.py:111-124 does with num_workers=12 -- no GIL shared with the thread that "
                   "launches ~700 kernels per step; 0 = the in-process collate thread (--prefetch)")
    p.add_argument("--prefetch", type=int, default=4, help="batches built ahead of the device by the collate thread (run.BatchPrefetcher; "
                   "T/run.py:111-124 uses DataLoader(num_workers=12, pin_memory=True)); 0 = collate inline on the main thread")
    p.add_argument("--graph", action="store_true", help="--fused_step on one rank: replay the step as a captured hipGraph per input shape "
                   "(TrainStep.step_graphed; unpadded token rows padded to buckets of 512) instead of host launches")
    p.add_argument("--steady_after", type=int, default=10, help="the epoch log also reports user-seq/s over the steps after this many (allocator "
                   "warm-up, first-call set-up and the loss scaler's initial back-off excluded)")
    p.add_argument("--checkpoint_root", type=str, default="./checkpoint", help="parent of checkpoint_<tower>.../cpt_<label>/ (T/run.py:326-331)")
    p.add_argument("--pretrained_dir", type=str, default="../pretrained_models",
                   help="directory holding <bert_model_load>/pytorch_model.bin (T/run.py:29-53 reads ../../pretrained_models/)")
    return p


def parse_args(argv=None):
    args = build_parser().parse_args(argv)
    args.news_attributes = args.news_attributes.split(",")
    return args
