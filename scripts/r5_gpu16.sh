#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for m in 1 4 8 16 1 8; do
  echo "== wpw_min $m (704 images)"; MOREC_SWIN_WPW_MIN=$m timeout 200 python scripts/swin_attn_bench.py 704 2>&1 | grep "stage [23]"
done
for m in 1 4 8 16; do
  echo "== wpw_min $m (352 images: Swin-B batch; widths differ, window counts per stage are Swin-B's)"; MOREC_SWIN_WPW_MIN=$m timeout 200 python scripts/swin_attn_bench.py 352 2>&1 | grep "stage"
done
