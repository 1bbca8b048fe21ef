#!/bin/bash
# round 6, call 13: the main loop's weight fragments by raw buffer loads (RG_WFRAG_BUFFER) against global loads
cd /root/repo; L=/root/repo/reagent_amd; O=/root/repo/gpurun_out/r06_mb; mkdir -p $O
cd profiles/microbench
./mfma_feed_dma > $O/mfma_feed_dma.txt 2>&1; head -8 $O/mfma_feed_dma.txt | cut -c1-170
for v in fwd_phases_wfrag0 fwd_phases; do for save in 0 1; do ./$v $save 512 1 > $O/${v}_save${save}_outlds1.txt; echo "== $v save=$save"; grep -E "forward|mainloop\(K=512|avg" $O/${v}_save${save}_outlds1.txt | head -5; done; done
cd /root/repo
timeout 900 python -m pytest tests/test_fused_mlp.py tests/test_baseline_shapes.py tests/test_sac_trainer.py tests/test_dqn_trainer.py tests/test_qrdqn_trainer.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED" | tail -3
AB_NO_PREFLIGHT=1 AB_CONFIG=c2 AB_REPS=3 bash profiles/scripts/gpu_ab6.sh "RG_LIB=$L/lib_wfrag0/libreagent_hip.so" "-" 2>&1 | sed "s#$L/##g"
AB_NO_PREFLIGHT=1 AB_CONFIG=c4 bash profiles/scripts/gpu_ab6.sh "RG_LIB=$L/lib_wfrag0/libreagent_hip.so" "-" 2>&1 | sed "s#$L/##g"
AB_NO_PREFLIGHT=1 AB_CONFIG=c3 bash profiles/scripts/gpu_ab6.sh "RG_LIB=$L/lib_wfrag0/libreagent_hip.so" "-" 2>&1 | sed "s#$L/##g"
