for st in 0 40 80 120 1073741864 1073741904; do
  echo "== MOREC_GEMM2W_STAGGER=$st"
  MOREC_GEMM2W_STAGGER=$st timeout 300 python scripts/gemm2w_check.py time 2>&1 | grep -E "K=768 (gelu|dmul|plain)|K=384 gelu|K=3072" | cut -c1-110
done
