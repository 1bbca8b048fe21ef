#!/bin/bash
# round 4, batch 14: ReLU after the bf16 pack (v_pk_max_i16) against the fp32 form, same box, C2 / C4 / C3
cd /root/repo
bash profiles/scripts/gpu_ab.sh "RG_LIB=reagent_amd/lib_norelu/libreagent_hip.so" "RG_X=late_relu"
AB_CONFIG=c4 bash profiles/scripts/gpu_ab.sh "RG_LIB=reagent_amd/lib_norelu/libreagent_hip.so" "RG_X=late_relu" 2>&1 | tail -4
AB_CONFIG=c3 bash profiles/scripts/gpu_ab.sh "RG_LIB=reagent_amd/lib_norelu/libreagent_hip.so" "RG_X=late_relu" 2>&1 | tail -4
timeout 600 python -m pytest tests/test_fused_mlp.py tests/test_baseline_shapes.py -m gpu -x -q 2>&1 | tail -3
