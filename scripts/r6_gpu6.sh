O=gpurun_out
python -m pytest tests/test_gemm_bench_shapes_gpu.py tests/test_swin_gpu.py tests/test_gemm_skinny_gpu.py -x -q 2>&1 | tail -3
B="python bench.py --no-secondary --no-cpu-baseline"
for rep in 1 2 3; do
for g in 1 0; do
  MOREC_GEMM2W=$g $B --tower swin_tiny --batch 64 --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('GEMM2W=$g swin_tiny', d['ms_per_step'])"
  MOREC_GEMM2W=$g $B --tower swin_base --batch 32 --steps 4 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('GEMM2W=$g swin_base', d['ms_per_step'])"
done
done
