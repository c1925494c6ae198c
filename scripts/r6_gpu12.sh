for i in 1 2; do
MOREC_DETERMINISTIC=1 python -m pytest tests/test_swin_gpu.py -q -s -k "g13 or g15" 2>&1 | grep -E "^g1[35]|passed|failed"
done
for i in 1 2; do
MOREC_DETERMINISTIC=1 python -m pytest tests/test_bench_mode_parity_vision_gpu.py -q -s 2>&1 | grep -E "vision|passed|failed" | cut -c1-400
done
python -m pytest tests/test_bench_mode_parity_vision_gpu.py -q -s 2>&1 | grep -E "vision|passed|failed" | cut -c1-400
