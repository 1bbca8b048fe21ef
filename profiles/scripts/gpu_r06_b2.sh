#!/bin/bash
# round 6, call 2: C3 grouped forward — whole-tile path (new lib) against round 5's per-row-tile loop (lib_r5grp), bf16; GPU tests of the grouped engine
cd /root/repo
timeout 900 python -m pytest tests/test_qrdqn_trainer.py tests/test_baseline_shapes.py tests/test_full_size.py -m gpu -q --no-header -p no:cacheprovider -k "qrdqn or c3 or grouped or qr" 2>&1 | grep -E "passed|failed|^FAILED|^E  " | tail -8
AB_CONFIG=c3 bash profiles/scripts/gpu_ab6.sh "RG_LIB=/root/repo/reagent_amd/lib_r5grp/libreagent_hip.so" "-" 2>&1 | sed "s#/root/repo/reagent_amd/##g"
AB_NO_PREFLIGHT=1 AB_CONFIG=c3 AB_REPS=1 bash profiles/scripts/gpu_ab6.sh "RG_LIB=/root/repo/reagent_amd/lib_ring4/libreagent_hip.so" "RG_LIB=/root/repo/reagent_amd/lib_ring16/libreagent_hip.so" 2>&1 | sed "s#/root/repo/reagent_amd/##g"
