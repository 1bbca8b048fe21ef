cd /root/repo
for rep in 1 2; do for v in "HIP_FORCE_DEV_KERNARG=0 eager" "HIP_FORCE_DEV_KERNARG=1 eager" "HIP_FORCE_DEV_KERNARG=0 graph" "HIP_FORCE_DEV_KERNARG=1 graph"; do set -- $v; env $1 python bench.py --config c2 --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-accurate --no-also --no-kernel-profile --launch $2 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('c2 $1 $2', round(r['ms_per_step'],4), r['region_ms'], round(r['host_enqueue_ms_per_step'],4))"; done; done
