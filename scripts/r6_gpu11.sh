python scripts/gemm2w_check.py check 2>&1 | tail -4
for epi in 1 0; do
for f in randn zero; do
  echo "== MOREC_GEMM2W_EPI=$epi fill=$f"
  MOREC_GEMM2W_EPI=$epi SB_FILL=$f python scripts/gemm2w_check.py time 2>&1 | grep -E "M=54919 N=3072|M=137984 N=1536|M=34496 N=3072|M=68992 N=2048" | sed "s/   runs .*//"
done
done
