#!/bin/bash
# diagnosis builds of kernels_linear.hip with one or more components of the GEMM half-step removed (never the production library): scratch/libv_<name>.so
# usage: scratch/gemm_variants.sh GQ_NO_MFMA GQ_NO_MFMA+GQ_NO_EXPAND ...
set -e
cd "$(dirname "$0")/../mllm_amd/csrc"
mkdir -p /tmp/vobj
for v in "$@"; do
  flags=$(echo "$v" | sed 's/+/ -D/g')
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -D$flags -c kernels_linear.hip -o /tmp/vobj/lin_$v.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch/libv_$v.so /tmp/vobj/lin_$v.o $(ls _obj/*.o | grep -v kernels_linear) -L/opt/rocm/lib -lrccl ) &
done
wait
