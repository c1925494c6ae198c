#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out/r4s
RUN="python -m idvs.morec_amd.run --synthetic 25600 --synthetic_items 80000 --synthetic_full_len --item_tower modal --bert_model_load bert_base_uncased \
  --freeze_paras_before 0 --batch_size 128 --embedding_dim 512 --lr 1e-4 --fine_tune_lr 5e-5 --l2_weight 0.01 --fine_tune_l2_weight 0.01 \
  --epoch 1 --max_steps 160 --steady_after 60 --fused_step --compute_dtype fp16 --local_rank 0"
timeout 300 $RUN --collate_workers 2 > gpurun_out/r4s/run_w2.log 2>&1
grep -n "steady\|user-seq\|collate" gpurun_out/r4s/run_w2.log | tail -4
timeout 300 $RUN --collate_workers 4 > gpurun_out/r4s/run_w4.log 2>&1
grep -n "steady\|user-seq\|collate" gpurun_out/r4s/run_w4.log | tail -4
timeout 300 $RUN --collate_workers 0 > gpurun_out/r4s/run_w0.log 2>&1
grep -n "steady\|user-seq\|collate" gpurun_out/r4s/run_w0.log | tail -4
timeout 600 python -m pytest tests/test_eval_gpu.py -x -q -m gpu 2>&1 | tail -4
