cd /root/repo
L=/root/repo/reagent_amd
timeout 900 python -m pytest tests/test_fused_mlp.py tests/test_dqn_trainer.py tests/test_sac_trainer.py tests/test_baseline_shapes.py tests/test_full_size.py -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -2
bash profiles/scripts/gpu_ab.sh "RG_LIB=$L/lib_nooutlds/libreagent_hip.so" "RG_X=0"
AB_CONFIG=c4 bash profiles/scripts/gpu_ab.sh "RG_LIB=$L/lib_nooutlds/libreagent_hip.so" "RG_X=0" 2>&1 | cut -c1-80
