for L in idvs/morec_amd/libmorec_hip.so scratch_libs/libmorec_2w_nosp.so; do
for fill in randn zero; do
  echo "== $L fill=$fill"
  SB_FILL=$fill MOREC_HIP_LIB=$PWD/$L python scripts/gemm2w_check.py time 2>&1 | grep -E "M=54919 N=3072|M=137984 N=1536" | sed 's/   runs .*//'
done
done
