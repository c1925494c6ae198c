#!/bin/bash
# rebuild gemm8p.o with the resource-usage remarks and print VGPR / spill counts per kernel variant
cd /root/repo/idvs/morec_amd/csrc || exit 1
touch gemm8p.hip
make build/gemm8p.o EXTRA="-Rpass-analysis=kernel-resource-usage" 2>&1 | grep -E "error|Function Name| VGPRs:|VGPRs Spill|SGPRs Spill|ScratchSize" | sed 's/.*remark: *//; s/\[-Rpass.*//' | paste - - - - - | sed 's/Function Name: _ZN12_GLOBAL__N_113gemm8p_kernel//'
