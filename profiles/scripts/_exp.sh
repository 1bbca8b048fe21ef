cd /root/repo
L=/root/repo/reagent_amd
RG_LIB=$L/lib_acc/libreagent_hip.so timeout 900 python -m pytest tests/test_fused_mlp.py tests/test_baseline_shapes.py tests/test_dqn_trainer.py -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -3
bash profiles/scripts/gpu_ab.sh "RG_X=0" "RG_LIB=$L/lib_acc/libreagent_hip.so"
