#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out/r4r
python bench.py > gpurun_out/r4r/bench.json 2> gpurun_out/r4r/bench.err; tail -2 gpurun_out/r4r/bench.err
python -m idvs.morec_amd.run --synthetic 25600 --synthetic_items 80000 --synthetic_full_len --item_tower modal --bert_model_load bert_base_uncased \
  --freeze_paras_before 0 --batch_size 128 --embedding_dim 512 --lr 1e-4 --fine_tune_lr 5e-5 --l2_weight 0.01 --fine_tune_l2_weight 0.01 \
  --epoch 1 --max_steps 160 --steady_after 60 --fused_step --compute_dtype fp16 --local_rank 0 > gpurun_out/r4r/run.log 2>&1
grep -n "steady\|user-seq\|scaler\|collate\|Hit10" gpurun_out/r4r/run.log | tail -6
