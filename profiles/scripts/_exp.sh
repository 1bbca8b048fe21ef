cd /root/repo
timeout 900 python -m pytest tests/test_qrdqn_trainer.py tests/test_baseline_shapes.py tests/test_abi_symbols.py -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | grep -E "passed|failed|rror" | tail -2
for S in 1 0; do
  RG_QR_STREAMS=$S python bench.py --config c3 --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline --launch eager --no-graph > gpurun_out/c3v.json 2>/dev/null
  python -c "
import json; r=json.load(open('gpurun_out/c3v.json'))
print('streams $S', round(r['ms_per_step'],4), [round(x,2) for x in r['region_ms']], 'dquantile', r['parity'].get('max_abs_dquantile'))"
done
bash profiles/scripts/gpu_timeline.sh c3 bf16 RG_QR_STREAMS=1 | head -30
