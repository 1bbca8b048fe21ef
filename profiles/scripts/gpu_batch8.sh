#!/bin/bash
cd /root/repo/profiles/microbench
for b in wgrad_phases wgrad_phases_p0 wgrad_phases_pp0; do for e in "RG_WGRAD_PLAN=legacy" "RG_WGRAD_PLAN=balanced" "RG_WGRAD_UNSHARED=58"; do echo "== $b $e"; env $e timeout 100 ./$b | grep -E "wgrad group|layer|compute|wait|barrier|store"; done; done 2>&1 | tee ../../gpurun_out/wgrad_phases_r04b.txt | grep -E "==|wgrad group|layer"
cd /root/repo; L=/root/repo/reagent_amd
bash profiles/scripts/gpu_ab.sh "RG_WGRAD_PLAN=legacy RG_LIB=$L/lib_pp0/libreagent_hip.so" "RG_WGRAD_PLAN=legacy RG_LIB=$L/lib_p0/libreagent_hip.so" "RG_WGRAD_PLAN=legacy" "RG_WGRAD_PLAN=balanced" "RG_WGRAD_UNSHARED=58" "RG_WGRAD_UNSHARED=58 RG_WGRAD_TOTAL=384" 2>&1 | sed "s#$L/##g"
