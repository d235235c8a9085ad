#!/bin/bash
# rebuilds only kernels_decode.hip with extra -D flags and links a scratch .so: scratch/lib_$2.so
set -e
cd /root/repo/mllm_amd/csrc
mkdir -p /tmp/vd_$2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC $1 -c kernels_decode.hip -o /tmp/vd_$2/kernels_decode.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/scratch/lib_$2.so /tmp/vd_$2/kernels_decode.o _obj/runtime.hip.o _obj/kernels_elem.hip.o _obj/kernels_linear.hip.o _obj/kernels_attn.hip.o _obj/engine.hip.o _obj/host_quantize.cpp.o -fopenmp -lgomp
