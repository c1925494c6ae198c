#!/bin/bash
# round-4 GPU call C: new-path tests, the full default bench line (fp16 headline + secondaries), run.py steady-state rate
mkdir -p gpurun_out/r4c
python -m pytest tests/test_fp16_mode_gpu.py tests/test_eval_gpu.py tests/test_bench_launch.py tests/test_train_step_gpu.py tests/test_train_step_rccl_gpu.py tests/test_train_step_ddp_gpu.py -q -s -m gpu > gpurun_out/r4c/tests.log 2>&1
tail -4 gpurun_out/r4c/tests.log
python bench.py > gpurun_out/r4c/bench.json 2> gpurun_out/r4c/bench.err
tail -2 gpurun_out/r4c/bench.err
for pf in 2 0; do
python -m idvs.morec_amd.run --synthetic 12800 --synthetic_items 80000 --synthetic_full_len --item_tower modal --bert_model_load bert_base_uncased \
  --freeze_paras_before 0 --batch_size 128 --embedding_dim 512 --lr 1e-4 --fine_tune_lr 5e-5 --l2_weight 0.01 --fine_tune_l2_weight 0.01 \
  --epoch 1 --max_steps 70 --fused_step --compute_dtype fp16 --local_rank 0 --prefetch $pf > gpurun_out/r4c/run_pf$pf.log 2>&1
grep -n "steady\|user-seq\|scaler" gpurun_out/r4c/run_pf$pf.log | tail -4
done
