#!/bin/bash
# round 6, call 7: accumulators in AccVGPRs + PACK/LDS store after the barrier (lib_agpr) against the default build — numerics tests
# on the variant, phase stamps, C2 / C4 / C3 steps same box; the feed micro-benchmark with both operands from LDS
cd /root/repo; L=/root/repo/reagent_amd; O=/root/repo/gpurun_out/r06_mb; mkdir -p $O
RG_LIB=$L/lib_agpr/libreagent_hip.so timeout 900 python -m pytest tests/test_fused_mlp.py tests/test_baseline_shapes.py tests/test_dqn_trainer.py tests/test_sac_trainer.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED|^E  " | tail -8
cd profiles/microbench
./fwd_phases 0 > $O/fwd_phases_save0.txt; ./fwd_phases_agpr 0 > $O/fwd_phases_agpr_save0.txt; ./fwd_phases_agpr 1 > $O/fwd_phases_agpr_save1.txt; ./fwd_phases 1 > $O/fwd_phases_save1.txt
for f in fwd_phases_save0 fwd_phases_agpr_save0 fwd_phases_save1 fwd_phases_agpr_save1; do echo "== $f"; head -18 $O/$f.txt; done
./mfma_feed 2>&1 | tail -5 | tee $O/mfma_feed_lds.txt
cd /root/repo
AB_NO_PREFLIGHT=1 AB_CONFIG=c2 bash profiles/scripts/gpu_ab6.sh "-" "RG_LIB=$L/lib_agpr/libreagent_hip.so" 2>&1 | sed "s#$L/##g"
AB_NO_PREFLIGHT=1 AB_CONFIG=c4 AB_REPS=1 bash profiles/scripts/gpu_ab6.sh "-" "RG_LIB=$L/lib_agpr/libreagent_hip.so" 2>&1 | sed "s#$L/##g"
AB_NO_PREFLIGHT=1 AB_CONFIG=c3 AB_REPS=1 bash profiles/scripts/gpu_ab6.sh "-" "RG_LIB=$L/lib_agpr/libreagent_hip.so" 2>&1 | sed "s#$L/##g"
