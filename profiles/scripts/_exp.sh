cd /root/repo
bash profiles/scripts/gpu_default_bench.sh r03m | cut -c1-200 | head -8
python -c "
import json; r=json.load(open('gpurun_out/bench_r03m.json'))
print(r['config']['launch']); print(r['accurate']['launch']); print({k:v.get('launch') for k,v in r['also_measured'].items()})"
