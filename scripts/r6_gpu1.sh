#!/bin/bash
# round 6, GPU call 1: tests of the changed areas, shadow/setprio experiment, res32 lazy A/B, default bench
O=gpurun_out; mkdir -p $O
python -m pytest tests/test_layernorm_res32_gpu.py tests/test_eval_gpu.py tests/test_fp16_mode_gpu.py tests/test_deterministic_gpu.py -x -q > $O/g1_tests.txt 2>&1
tail -5 $O/g1_tests.txt
for rep in 1 2; do
for L in hip nosp exp1 exp1_nosp exph exph_nosp; do
  MOREC_HIP_LIB=$PWD/idvs/morec_amd/libmorec_$L.so timeout 120 python scripts/shadow_bench.py 2>&1 | grep -v amdgpu.ids
done
done > $O/g1_shadow.txt 2>&1
B="python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline"
for rep in 1 2; do
  $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp16', d['ms_per_step'], d['roofline']['frac'])"
  MOREC_RES32_LAZY=1 $B --dtype fp16_res32 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('res32 lazy', d['ms_per_step'])"
  MOREC_RES32_LAZY=0 $B --dtype fp16_res32 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('res32 eager', d['ms_per_step'])"
done > $O/g1_res32_ab.txt 2>&1
cat $O/g1_res32_ab.txt
python bench.py > $O/g1_bench_default.json 2> $O/g1_bench_default.err
tail -1 $O/g1_bench_default.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('default', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('at_sustained_clock'))
print('sustained', d.get('sustained'))
print('eval', d.get('eval',{}).get('encode_all_items'))
print('res32', d.get('fp16_res32_mode'))
"
