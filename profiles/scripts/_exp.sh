cd /root/repo
L=/root/repo/reagent_amd
timeout 900 python -m pytest tests/test_qrdqn_trainer.py tests/test_baseline_shapes.py tests/test_fused_mlp.py -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -2
for S in 0 1; do
for v in nostage ""; do
  if [ -n "$v" ]; then export RG_LIB=$L/lib_$v/libreagent_hip.so; else unset RG_LIB; fi
  RG_QR_STREAMS=$S python bench.py --config c3 --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline --launch eager --no-graph > gpurun_out/c3v.json 2>/dev/null
  python -c "
import json; r=json.load(open('gpurun_out/c3v.json'))
print('streams $S variant [$v]', round(r['ms_per_step'],4), [round(x,2) for x in r['region_ms']], 'parity', {k:v for k,v in (r.get('parity') or {}).items() if k.startswith('max') or k.startswith('rel')})
for k,v in r['per_call_ms_per_step'].items():
    if 'forward' in k or 'backward' in k: print('   %-80s %.4f'%(k[:80],v))"
done; done
