#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 120 python scratch/gemm_occ.py mllm_amd/libmllm_hip.so full
for v in k8_NO_BAR k8_NO_DMA k8_NO_P1 k8_NO_P2 k8_NO_P1DK8_NO_P2; do timeout -k 10 120 python scratch/gemm_occ.py scratch/tmp_so/$v.so; done
