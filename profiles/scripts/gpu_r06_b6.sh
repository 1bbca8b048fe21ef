#!/bin/bash
# round 6, call 6: the one-launch policy sampler (GPU tests, C4 A/B), the feed micro-benchmark with both operands from LDS,
# phase stamps of the forward kernels (saved under profiles/microbench/out/r06)
cd /root/repo; L=/root/repo/reagent_amd; O=/root/repo/gpurun_out/r06_mb; mkdir -p $O
timeout 900 python -m pytest tests/test_replay_buffer.py tests/test_sac_trainer.py tests/test_td3_trainer.py tests/test_graph_replay.py tests/test_full_size.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED|^E  " | tail -8
AB_NO_PREFLIGHT=1 AB_CONFIG=c4 bash profiles/scripts/gpu_ab6.sh "RG_POLICY_SAMPLER=0" "-" 2>&1
AB_NO_PREFLIGHT=1 AB_CONFIG=c4 AB_PREC=bf16x3 AB_REPS=1 bash profiles/scripts/gpu_ab6.sh "RG_POLICY_SAMPLER=0" "-" 2>&1
cd profiles/microbench
./mfma_feed > $O/mfma_feed.txt 2>&1; cat $O/mfma_feed.txt
./fwd_phases 0 > $O/fwd_phases_save0.txt; ./fwd_phases 1 > $O/fwd_phases_save1.txt; head -20 $O/fwd_phases_save0.txt
./grouped_phases_w1 1 > $O/grouped_phases_whole_scatter.txt; ./grouped_phases_w0 1 > $O/grouped_phases_r5loop_scatter.txt
