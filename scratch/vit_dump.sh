#!/bin/bash
# bring-up: the reference's ViTModel through the adapter with per-Op dumps (tiny shape, one image)
set -e
rm -rf gpurun_out/vitdump && mkdir -p gpurun_out/vitdump /tmp/vd
python - <<'P'
from mllm_amd import synth, synthfile as w
c = synth.vit_tiny(); print(w.vit_file(c, "/tmp/vd")); synth.vit_images(c, 1).tofile("/tmp/vd/img.f32")
P
MLLM_HIP_DUMP_DIR=gpurun_out/vitdump oracle/_ref/ref_hip_vit --model /tmp/vd/vit-h256-f512-b2-p16-i64-c16-q4k-qd1.mllm --img /tmp/vd/img.f32 --n 1 --threads 2 --out /tmp/vd --cfg 256,4,512,2,16,64,16
ls gpurun_out/vitdump | wc -l
