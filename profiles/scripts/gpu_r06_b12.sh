#!/bin/bash
# round 6, call 12: split-bf16 kernels — the backward's first step requested early (RG_BWD_SIGNS_EARLY, x3 form) and
# x3_tile_kloop with four chunks per request group (RG_X3_TILE_G=4: the thin output layer's / dx layer's L2 chain)
cd /root/repo; L=/root/repo/reagent_amd; O=/root/repo/gpurun_out/r06_mb; mkdir -p $O
cd profiles/microbench
for v in x3_phases x3_phases_g4; do ./$v 0 > $O/${v}_save0.txt; echo "== $v"; grep -E "forward|output layer|avg|x tile" $O/${v}_save0.txt; done
cd /root/repo
timeout 900 python -m pytest tests/test_fused_mlp.py tests/test_baseline_shapes.py tests/test_sac_trainer.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED" | tail -3
RG_LIB=$L/lib_x3g4/libreagent_hip.so timeout 900 python -m pytest tests/test_fused_mlp.py tests/test_baseline_shapes.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED" | tail -3
AB_NO_PREFLIGHT=1 AB_CONFIG=c2 AB_PREC=bf16x3 bash profiles/scripts/gpu_ab6.sh "RG_LIB=$L/lib_early0/libreagent_hip.so" "-" "RG_LIB=$L/lib_x3g4/libreagent_hip.so" 2>&1 | sed "s#$L/##g"
AB_NO_PREFLIGHT=1 AB_CONFIG=c4 AB_PREC=bf16x3 bash profiles/scripts/gpu_ab6.sh "RG_LIB=$L/lib_early0/libreagent_hip.so" "-" "RG_LIB=$L/lib_x3g4/libreagent_hip.so" 2>&1 | sed "s#$L/##g"
