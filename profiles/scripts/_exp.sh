cd /root/repo
timeout 900 python -m pytest tests/test_model_autograd.py tests/test_torch_ops.py tests/test_dqn_trainer.py tests/test_data_parallel.py tests/test_checkpoint_resume.py tests/test_sac_trainer.py tests/test_fused_mlp.py -m gpu -q -x --no-header -p no:cacheprovider > gpurun_out/pt_r03b.log 2>&1; grep -E "passed|failed|error" gpurun_out/pt_r03b.log | tail -3
