B="python bench.py --no-secondary --no-cpu-baseline --steps 20 --warmup 5"
for rep in 1 2 3; do
for g in 1 0 2; do
  MOREC_GEMM2W=$g $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('GEMM2W=$g text fp16', d['ms_per_step'], d['roofline']['frac'], d['roofline']['gemm_ms_per_step'])"
done
done
