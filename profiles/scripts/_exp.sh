cd /root/repo
timeout 900 python -m pytest tests/test_qrdqn_trainer.py tests/test_baseline_shapes.py tests/test_fused_mlp.py tests/test_sac_trainer.py -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -2
for S in 0 1; do
  RG_QR_STREAMS=$S python bench.py --config c3 --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-parity --launch eager --no-graph > gpurun_out/c3v.json 2>/dev/null
  python -c "
import json; r=json.load(open('gpurun_out/c3v.json'))
print('streams $S', round(r['ms_per_step'],4), [round(x,2) for x in r['region_ms']])
for k,v in r['per_call_ms_per_step'].items():
    if 'forward' in k or 'backward' in k: print('   %-80s %.4f'%(k[:80],v))"
done
for c in c2 c4; do python bench.py --config $c --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-parity --launch eager --no-graph --no-accurate --no-also 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$c', round(r['ms_per_step'],4))"; done
