#!/bin/bash
# dump the ISA of one gemm8p kernel variant (default: bf16 ACT 0) to /tmp/g8/k.s and list its waits / VMEM ops / barriers
mkdir -p /tmp/g8
V=${1:-I4bf16Li0ELb0E}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I/root/repo/include -I/root/repo/idvs/morec_amd/csrc -Wno-unused-value -S --cuda-device-only /root/repo/idvs/morec_amd/csrc/gemm8p.hip -o /tmp/g8/gemm8p.s 2>/dev/null
L=$(grep -n "^_ZN12_GLOBAL__N_113gemm8p_kernel${V}EEv8GemmArgs:" /tmp/g8/gemm8p.s | cut -d: -f1)
awk -v s=$L 'NR>=s' /tmp/g8/gemm8p.s | awk '/^\.Lfunc_end/{exit} {print}' > /tmp/g8/k.s
wc -l /tmp/g8/k.s
grep -n "s_waitcnt vmcnt\|buffer_load\|buffer_store\|global_load\|global_store\|scratch_\|Loop Header\|s_endpgm" /tmp/g8/k.s | grep -v "lds$" | awk '{print $1,$2,$3,$4}'
