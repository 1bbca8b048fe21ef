#!/bin/bash
# round 6, call 5: split-bf16 grouped forward through the whole-unit path — GPU tests of the grouped engine and the fused stacks, then
# C3 / C2 / C4 in bf16x3 against round 5's kernels (lib_r5base), same box
cd /root/repo; L=/root/repo/reagent_amd
timeout 900 python -m pytest tests/test_qrdqn_trainer.py tests/test_baseline_shapes.py tests/test_full_size.py tests/test_fused_mlp.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED|^E  " | tail -8
for lib in lib_r5base lib; do RG_SKIP_PREFLIGHT=1 RG_LIB=$L/$lib/libreagent_hip.so python profiles/microbench/grouped_fwd_time.py bf16x3 2>&1 | grep "us / launch" | sed "s/^/$lib /"; done
AB_NO_PREFLIGHT=1 AB_CONFIG=c3 AB_PREC=bf16x3 bash profiles/scripts/gpu_ab6.sh "RG_LIB=$L/lib_r5base/libreagent_hip.so" "-" 2>&1 | sed "s#$L/##g"
AB_NO_PREFLIGHT=1 AB_CONFIG=c2 AB_PREC=bf16x3 AB_REPS=1 bash profiles/scripts/gpu_ab6.sh "RG_LIB=$L/lib_r5base/libreagent_hip.so" "-" 2>&1 | sed "s#$L/##g"
AB_NO_PREFLIGHT=1 AB_CONFIG=c4 AB_PREC=bf16x3 AB_REPS=1 bash profiles/scripts/gpu_ab6.sh "RG_LIB=$L/lib_r5base/libreagent_hip.so" "-" 2>&1 | sed "s#$L/##g"
