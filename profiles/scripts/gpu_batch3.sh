#!/bin/bash
cd /root/repo; L=/root/repo/reagent_amd
bash profiles/scripts/gpu_ab.sh "RG_LIB=$L/lib/libreagent_hip.so" "RG_LIB=$L/lib_s0/libreagent_hip.so" "RG_LIB=$L/lib_t128/libreagent_hip.so" "RG_LIB=$L/lib_t32/libreagent_hip.so" 2>&1 | sed "s#$L/##g"
AB_CONFIG=c4 bash profiles/scripts/gpu_ab.sh "RG_LIB=$L/lib/libreagent_hip.so" "RG_LIB=$L/lib_s0/libreagent_hip.so" 2>&1 | sed "s#$L/##g"
timeout 600 python -m pytest tests/test_fused_mlp.py tests/test_dqn_trainer.py tests/test_sac_trainer.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -3
