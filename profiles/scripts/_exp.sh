cd /root/repo
timeout 900 python -m pytest tests/test_graph_replay.py tests/test_checkpoint_resume.py tests/test_sac_trainer.py tests/test_td3_trainer.py -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | grep -E "passed|failed|rror" | tail -3
for l in eager graph; do python bench.py --config c4 --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-accurate --no-also --no-kernel-profile --launch $l 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('c4 $l', round(r['ms_per_step'],4), r['region_ms'], 'host', round(r['host_enqueue_ms_per_step'],3), r['parity']['max_abs_dlogits'], r['parity']['rel_dloss'])"; done
python profiles/microbench/glue_trace.py c4 bf16 2>/dev/null | tail -6
