#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out/r4l
timeout 600 python -m pytest tests/test_swin_gpu.py tests/test_bench_mode_parity_vision_gpu.py tests/test_train_step_gpu.py -q -x > gpurun_out/r4l/t.log 2>&1; tail -4 gpurun_out/r4l/t.log
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof; mkdir -p /tmp/prof
R=$GRAFT_REPO_ROOT
MOREC_WGRAD_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o ks -- python $R/bench.py --tower swin_tiny --batch 64 --steps 4 --warmup 2 --no-cpu-baseline --no-secondary > $R/gpurun_out/r4l/swin_line.json 2> /dev/null
python $R/scripts/prof_summary.py /tmp/prof/ks_results.db 11 "r04 swin_tiny after the 16-byte attention stores" > $R/gpurun_out/r4l/swin_kernel_stats.csv
grep "swin_attn" $R/gpurun_out/r4l/swin_kernel_stats.csv
cd $R
python bench.py --tower swin_tiny --batch 64 --steps 6 --warmup 2 --no-secondary --no-cpu-baseline > gpurun_out/r4l/swin.json 2> gpurun_out/r4l/swin.err; grep "timed region" gpurun_out/r4l/swin.err
python bench.py --tower swin_base --batch 32 --steps 4 --warmup 2 --no-secondary --no-cpu-baseline > gpurun_out/r4l/swinb.json 2> gpurun_out/r4l/swinb.err; grep "timed region" gpurun_out/r4l/swinb.err
