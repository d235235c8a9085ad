#!/bin/bash
# scratch .so with extra -D flags ($1) -> scratch/lib_$2.so  (never the production library)
set -e
cd /root/repo/mllm_amd/csrc
mkdir -p /tmp/varobj_$2
for f in runtime kernels_elem kernels_linear kernels_attn kernels_decode engine; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC $1 -c $f.hip -o /tmp/varobj_$2/$f.o &
done
g++ -std=c++17 -O2 -mavx2 -mf16c -mfma -ffp-contract=off -fopenmp -fPIC -c host_quantize.cpp -o /tmp/varobj_$2/hq.o
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/scratch/lib_$2.so /tmp/varobj_$2/*.o -fopenmp -lgomp
