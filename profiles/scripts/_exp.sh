cd /root/repo
timeout 300 python profiles/microbench/mall_chunks.py bf16 2>&1 | tail -5
timeout 300 python profiles/microbench/mall_chunks.py bf16x3 2>&1 | tail -5
AB_PREC=bf16x3 bash profiles/scripts/gpu_ab.sh "RG_X=base" "RG_LIB=/root/repo/reagent_amd/lib_x3ring2/libreagent_hip.so"
