cd /root/repo
timeout 900 python -m pytest tests/test_graph_replay.py tests/test_checkpoint_resume.py tests/test_dqn_trainer.py tests/test_qrdqn_trainer.py tests/test_replay_buffer.py -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | grep -E "passed|failed|rror|assert" | tail -5
for rep in 1 2; do for v in "RG_GRAPH_CURSOR=0 graph" "RG_GRAPH_CURSOR=1 graph" "RG_X=1 eager"; do set -- $v; env $1 python bench.py --config c2 --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-accurate --no-also --no-kernel-profile --launch $2 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('c2 $1 $2', round(r['ms_per_step'],4), r['region_ms'], r['parity']['max_abs_dq'], r['parity']['rel_dloss'])"; done; done
TL_LAUNCH="--launch graph" bash profiles/scripts/gpu_timeline.sh c2 bf16 | tail -11
