cd /root/repo
L=/root/repo/reagent_amd
bash profiles/scripts/gpu_ab.sh "RG_X=0" "RG_LIB=$L/lib_wg256/libreagent_hip.so" "RG_LIB=$L/lib_wg384/libreagent_hip.so" "RG_LIB=$L/lib_wg512/libreagent_hip.so"
AB_PREC=bf16x3 bash profiles/scripts/gpu_ab.sh "RG_LIB=$L/lib_x3ring2/libreagent_hip.so" "RG_LIB=$L/lib_x3p0/libreagent_hip.so" "RG_LIB=$L/lib_x3p2/libreagent_hip.so" "RG_LIB=$L/lib_wg256/libreagent_hip.so"
