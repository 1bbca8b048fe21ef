cd /root/repo
timeout 900 python -m pytest tests/test_full_size.py tests/test_graph_replay.py tests/test_checkpoint_resume.py tests/test_data_parallel.py -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -4
bash profiles/scripts/gpu_default_bench.sh r03e
