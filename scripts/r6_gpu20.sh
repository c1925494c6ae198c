#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
python scripts/det_probe2.py 6 2>&1 | grep -v amdgpu.ids > $O/g20_det2.txt
cat $O/g20_det2.txt | cut -c1-250
