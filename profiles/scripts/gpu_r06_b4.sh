#!/bin/bash
# round 6, call 4: the run-time activation tested once per tile (act_apply_n) — phase stamps of both output paths, the launches alone
# against round 5's kernels (lib_r5base), C3 / C2 / C4 steps same box
cd /root/repo/profiles/microbench
for v in w0 w1; do ./grouped_phases_$v 1 | grep -E "grouped forward|grouped output|whole-tile|avg"; done
./fwd_phases 0 | grep -E "forward NW|output layer|avg"
cd /root/repo
L=/root/repo/reagent_amd
for lib in lib_r5base lib_r5grp lib; do RG_SKIP_PREFLIGHT=1 RG_LIB=$L/$lib/libreagent_hip.so python profiles/microbench/grouped_fwd_time.py bf16 2>&1 | grep "us / launch" | sed "s/^/$lib /"; done
AB_NO_PREFLIGHT=1 AB_CONFIG=c3 bash profiles/scripts/gpu_ab6.sh "RG_LIB=$L/lib_r5base/libreagent_hip.so" "RG_LIB=$L/lib_r5grp/libreagent_hip.so" "-" 2>&1 | sed "s#$L/##g"
AB_NO_PREFLIGHT=1 AB_CONFIG=c2 bash profiles/scripts/gpu_ab6.sh "RG_LIB=$L/lib_r5base/libreagent_hip.so" "-" 2>&1 | sed "s#$L/##g"
AB_NO_PREFLIGHT=1 AB_CONFIG=c4 AB_REPS=1 bash profiles/scripts/gpu_ab6.sh "RG_LIB=$L/lib_r5base/libreagent_hip.so" "-" 2>&1 | sed "s#$L/##g"
