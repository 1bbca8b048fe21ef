#!/bin/bash
# round 6, call 11: the thin output layer's LDS K loop in bursts of 8 chunks (RG_OUT_LDS_BURST) and the backward's first step
# (sign planes + one-chunk weight fragments requested before the dout tile, RG_BWD_SIGNS_EARLY) against round 6's sources
cd /root/repo; L=/root/repo/reagent_amd; O=/root/repo/gpurun_out/r06_mb; mkdir -p $O
cd profiles/microbench
for v in fwd_phases_burst0 fwd_phases; do ./$v 0 512 1 > $O/${v}_outlds1.txt; echo "== $v (out_lds 1)"; grep -E "forward|output layer|avg" $O/${v}_outlds1.txt; done
for v in bwd_phases_early0 bwd_phases; do ./$v 16 > $O/${v}_out16.txt; echo "== $v"; head -8 $O/${v}_out16.txt; done
cd /root/repo
timeout 900 python -m pytest tests/test_fused_mlp.py tests/test_baseline_shapes.py tests/test_sac_trainer.py tests/test_dqn_trainer.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED" | tail -3
AB_NO_PREFLIGHT=1 AB_CONFIG=c2 bash profiles/scripts/gpu_ab6.sh "RG_LIB=$L/lib_base6/libreagent_hip.so" "-" 2>&1 | sed "s#$L/##g"
AB_NO_PREFLIGHT=1 AB_CONFIG=c4 bash profiles/scripts/gpu_ab6.sh "RG_LIB=$L/lib_base6/libreagent_hip.so" "-" 2>&1 | sed "s#$L/##g"
