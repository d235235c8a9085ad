#!/bin/bash
# one source file (default kernels_decode) recompiled with extra -D flags and linked with the current objects into scratch/tmp_so/<name>.so: scratch/variant.sh <name> "<flags>" [file]
set -e
cd "$(dirname "$0")/.."
mkdir -p scratch/tmp_so /tmp/var_$1
F=${3:-kernels_decode}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC $2 -c mllm_amd/csrc/$F.hip -o /tmp/var_$1/$F.hip.o
OBJS=$(ls mllm_amd/csrc/_obj/*.o | grep -v "/$F.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/tmp_so/$1.so $OBJS /tmp/var_$1/$F.hip.o -L/opt/rocm/lib -lrccl
