#!/bin/bash
# round 6, GPU call 21: the whole GPU suite, smoke(), the default bench line
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
python -m pytest tests/ -x -q -m gpu > $O/g21_tests.txt 2>&1; tail -4 $O/g21_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
python bench.py > $O/g21_bench_default.json 2> $O/g21_bench_default.err
tail -1 $O/g21_bench_default.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('default', d['value'], d['ms_per_step'], d['roofline']['frac'])
for k in ('fp16_res32_mode','bf16_mode','vision_swin_tiny','vision_swin_base','vision_u8_pipeline','id_tower','bert_tiny','fp32_parity_mode','fp32x3_mode','padded_token_layout','with_item_dedup'):
    v=d.get(k); print(k, v.get('ms_per_step') if isinstance(v,dict) else v)
print('eval', d.get('eval',{}).get('encode_all_items',{}).get('seconds'), d.get('eval',{}).get('rank_users',{}).get('seconds'))
print('cpu', d.get('cpu_baseline'))
"
