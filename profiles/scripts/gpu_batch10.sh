#!/bin/bash
cd /root/repo
bash profiles/scripts/gpu_ab.sh "RG_WGRAD_ORDER=0" "RG_WGRAD_ORDER=1" "RG_WGRAD_ORDER=1 RG_WGRAD_THIN=128" "RG_WGRAD_ORDER=1 RG_WGRAD_THIN=256" "RG_WGRAD_ORDER=1 RG_WGRAD_THIN=96" 2>&1
AB_PREC=bf16x3 bash profiles/scripts/gpu_ab.sh "RG_WGRAD_ORDER=0" "RG_WGRAD_ORDER=1 RG_WGRAD_THIN=128" 2>&1
AB_CONFIG=c4 bash profiles/scripts/gpu_ab.sh "RG_WGRAD_ORDER=0" "RG_WGRAD_ORDER=1 RG_WGRAD_THIN=128" 2>&1
timeout 600 python -m pytest tests/test_fused_mlp.py tests/test_dqn_trainer.py tests/test_sac_trainer.py tests/test_baseline_shapes.py tests/test_qrdqn_trainer.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -3
