#!/bin/bash
# diagnostic build with stamps in the gate|up GEMV only (scratch .so, never the production library)
set -e
cd "$(dirname "$0")/../mllm_amd/csrc"
rm -rf /tmp/stampobj; mkdir -p /tmp/stampobj
for f in runtime kernels_elem kernels_linear kernels_attn kernels_decode kernels_sample kernels_image kernels_n4 moe engine; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DMLLM_HIP_STAMPS_GUB $1 -c $f.hip -o /tmp/stampobj/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libmllm_hip_stamps.so /tmp/stampobj/*.o -L/opt/rocm/lib -lrccl
