#!/bin/bash
# round 6, call 9: the saving forward's fragment stores split between PACK and after the LDS store (RG_SAVE_SPLIT = 2 / 0 against 4)
cd /root/repo; L=/root/repo/reagent_amd; O=/root/repo/gpurun_out/r06_mb; mkdir -p $O
cd profiles/microbench
for v in fwd_phases fwd_phases_split2 fwd_phases_split0; do ./$v 1 > $O/${v}_save1.txt; echo "== $v"; head -17 $O/${v}_save1.txt | grep -E "forward|pack|store|mainloop|wait"; done
cd /root/repo
RG_LIB=$L/lib_split2/libreagent_hip.so timeout 600 python -m pytest tests/test_fused_mlp.py tests/test_baseline_shapes.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED" | tail -3
AB_NO_PREFLIGHT=1 AB_CONFIG=c2 bash profiles/scripts/gpu_ab6.sh "-" "RG_LIB=$L/lib_split2/libreagent_hip.so" "RG_LIB=$L/lib_split0/libreagent_hip.so" 2>&1 | sed "s#$L/##g"
AB_NO_PREFLIGHT=1 AB_CONFIG=c4 AB_REPS=1 bash profiles/scripts/gpu_ab6.sh "-" "RG_LIB=$L/lib_split2/libreagent_hip.so" "RG_LIB=$L/lib_split0/libreagent_hip.so" 2>&1 | sed "s#$L/##g"
