#!/bin/bash
cd /root/repo
bash profiles/scripts/gpu_ab.sh "RG_WGRAD_PLAN=legacy" "RG_WGRAD_PLAN=balanced" "RG_WGRAD_TOTAL=320" "RG_WGRAD_TOTAL=384" "RG_WGRAD_TOTAL=512" "RG_WGRAD_UNSHARED=20" "RG_WGRAD_UNSHARED=36" 2>&1
AB_PREC=bf16x3 bash profiles/scripts/gpu_ab.sh "RG_WGRAD_PLAN=legacy" "RG_WGRAD_PLAN=balanced" 2>&1
AB_CONFIG=c4 bash profiles/scripts/gpu_ab.sh "RG_WGRAD_PLAN=legacy" "RG_WGRAD_PLAN=balanced" 2>&1
timeout 600 python -m pytest tests/test_fused_mlp.py tests/test_dqn_trainer.py tests/test_sac_trainer.py tests/test_baseline_shapes.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -3
