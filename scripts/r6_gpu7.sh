python -m pytest tests/test_swin_gpu.py -x -q 2>&1 | tail -2
for rep in 1 2; do
for L in scratch_libs/libmorec_base.so idvs/morec_amd/libmorec_hip.so; do
  echo "== $L"
  MOREC_HIP_LIB=$PWD/$L python scripts/swin_attn_bench.py 704 2>&1 | grep -v amdgpu.ids
done
done
