cd /root/repo
timeout 900 python -m pytest tests/test_sac_trainer.py tests/test_td3_trainer.py tests/test_baseline_shapes.py tests/test_graph_replay.py -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -2
python bench.py --config c4 --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline --launch eager --no-graph > gpurun_out/c4v.json 2>/dev/null
python -c "
import json; r=json.load(open('gpurun_out/c4v.json'))
print('c4', round(r['ms_per_step'],4), [round(x,2) for x in r['region_ms']], {k:v for k,v in r['parity'].items() if k.startswith('max') or k.startswith('rel')})
for k,v in r['per_call_ms_per_step'].items(): print('   %-80s %.4f'%(k[:80],v))"
