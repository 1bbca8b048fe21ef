#!/bin/bash
# round 6, call 10: schedule ticks of a step in one launch (RG_DEFER_TICKS=0/1) — GPU tests of the scheduled paths, C4 / C3 / C2 steps
cd /root/repo
timeout 900 python -m pytest tests/test_sac_trainer.py tests/test_td3_trainer.py tests/test_graph_replay.py tests/test_dqn_trainer.py tests/test_optimizers.py tests/test_checkpoint_resume.py tests/test_crr_trainer.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED|^E  " | tail -6
AB_NO_PREFLIGHT=1 AB_CONFIG=c4 bash profiles/scripts/gpu_ab6.sh "RG_DEFER_TICKS=0" "-" 2>&1
AB_NO_PREFLIGHT=1 AB_CONFIG=c4 AB_EXTRA="--launch graph" bash profiles/scripts/gpu_ab6.sh "RG_DEFER_TICKS=0" "-" 2>&1
