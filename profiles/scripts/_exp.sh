cd /root/repo
run() { python bench.py --config c2 --precision $1 --steps 20 --warmup 5 --no-cpu-baseline --no-parity --launch eager --no-graph --no-accurate --no-also --no-kernel-profile $2 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$1 $2', r['ms_per_step'], r['region_ms'])"; }
for rep in 1 2; do run bf16 ""; run bf16 "--prefetch wgrad"; run bf16 "--prefetch all"; done
for rep in 1 2; do run bf16x3 ""; run bf16x3 "--prefetch wgrad"; done
