#!/bin/bash
# Experiment builds of the eight-phase GEMM main loop (never the product library): gemm8p.hip recompiled with -D flags and linked with the
# product's other objects into idvs/morec_amd/libmorec_<tag>.so; scripts/shadow_bench.py runs them through MOREC_HIP_LIB.
#   bash scripts/shadow_build.sh <tag> <flags...>      e.g.  shadow_build.sh exp1_nosp -DG8_EXP_SHADOW=1 -DG8_NO_SETPRIO
set -e
TAG=$1; shift
cd "$(dirname "$0")/../idvs/morec_amd/csrc"
make -j8 > /dev/null
mkdir -p build_exp
/opt/rocm/bin/hipcc "$@" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../../include -Wno-unused-value -c gemm8p.hip -o build_exp/gemm8p_$TAG.o
OBJS=$(ls build/*.o | grep -v "build/gemm8p.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS build_exp/gemm8p_$TAG.o -o ../libmorec_$TAG.so
echo built libmorec_$TAG.so
