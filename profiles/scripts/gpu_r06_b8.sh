#!/bin/bash
# round 6, call 8: grouped whole-tile path, 2 x 2 tiles per wave against one column tile per wave (lib_split0) — stamps, launches alone, C3 step
cd /root/repo; L=/root/repo/reagent_amd; O=/root/repo/gpurun_out/r06_mb; mkdir -p $O
timeout 900 python -m pytest tests/test_qrdqn_trainer.py tests/test_baseline_shapes.py tests/test_full_size.py -m gpu -q --no-header -p no:cacheprovider -k "qrdqn or c3 or grouped or qr" 2>&1 | grep -E "passed|failed|^FAILED|^E  " | tail -8
cd profiles/microbench
./grouped_phases_w1s0 1 > $O/grouped_phases_whole_1x4_scatter.txt; ./grouped_phases_w1 1 > $O/grouped_phases_whole_2x2_scatter.txt; ./grouped_phases_w1 0 > $O/grouped_phases_whole_2x2_ordered.txt
for f in grouped_phases_whole_1x4_scatter grouped_phases_whole_2x2_scatter grouped_phases_whole_2x2_ordered; do echo "== $f"; grep -E "grouped forward|grouped output|whole-tile|wave [0-9]" $O/$f.txt; done
cd /root/repo
for lib in lib_split0 lib; do RG_SKIP_PREFLIGHT=1 RG_LIB=$L/$lib/libreagent_hip.so python profiles/microbench/grouped_fwd_time.py bf16 2>&1 | grep "us / launch" | sed "s/^/$lib /"; done
AB_NO_PREFLIGHT=1 AB_CONFIG=c3 bash profiles/scripts/gpu_ab6.sh "RG_LIB=$L/lib_split0/libreagent_hip.so" "-" 2>&1 | sed "s#$L/##g"
