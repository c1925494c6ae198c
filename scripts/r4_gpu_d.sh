#!/bin/bash
mkdir -p gpurun_out/r4d
python bench.py --tower swin_tiny --batch 64 --steps 6 --warmup 2 --vision-input u8 --no-secondary --no-cpu-baseline > gpurun_out/r4d/u8.json 2> gpurun_out/r4d/u8.err
tail -3 gpurun_out/r4d/u8.err; head -c 1500 gpurun_out/r4d/u8.json; echo
for pf in 2 4; do
python -m idvs.morec_amd.run --synthetic 25600 --synthetic_items 80000 --synthetic_full_len --item_tower modal --bert_model_load bert_base_uncased \
  --freeze_paras_before 0 --batch_size 128 --embedding_dim 512 --lr 1e-4 --fine_tune_lr 5e-5 --l2_weight 0.01 --fine_tune_l2_weight 0.01 \
  --epoch 1 --max_steps 160 --steady_after 60 --fused_step --compute_dtype fp16 --local_rank 0 --prefetch $pf > gpurun_out/r4d/run_pf$pf.log 2>&1
grep -n "steady\|user-seq\|scaler\|collate" gpurun_out/r4d/run_pf$pf.log | tail -5
done
