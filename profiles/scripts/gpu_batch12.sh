#!/bin/bash
cd /root/repo; L=/root/repo/reagent_amd
timeout 600 python -m pytest tests/test_fused_mlp.py tests/test_dqn_trainer.py tests/test_sac_trainer.py tests/test_graph_replay.py tests/test_predictor.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED" | tail -5
bash profiles/scripts/gpu_ab.sh "RG_LIB=$L/lib_noswap/libreagent_hip.so" "RG_LIB=$L/lib/libreagent_hip.so" 2>&1 | sed "s#$L/##g"
AB_CONFIG=c4 bash profiles/scripts/gpu_ab.sh "RG_LIB=$L/lib_noswap/libreagent_hip.so" "RG_LIB=$L/lib/libreagent_hip.so" 2>&1 | sed "s#$L/##g"
