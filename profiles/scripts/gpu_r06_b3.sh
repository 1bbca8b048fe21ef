#!/bin/bash
# round 6, call 3: where the grouped forward's time goes — phase stamps (old per-row-tile output loop vs the whole-tile path), and the launches alone
cd /root/repo/profiles/microbench
for v in w0 w1; do for sc in 1 0; do ./grouped_phases_$v $sc; done; done
./fwd_phases 0 | head -20
cd /root/repo
RG_SKIP_PREFLIGHT=1 python profiles/microbench/grouped_fwd_time.py bf16 2>&1 | grep "us / launch"
RG_SKIP_PREFLIGHT=1 RG_LIB=/root/repo/reagent_amd/lib_r5grp/libreagent_hip.so python profiles/microbench/grouped_fwd_time.py bf16 2>&1 | grep "us / launch" | sed 's/^/r5grp /'
