#!/bin/bash
# rocprofv3 kernel statistics of C2 over a LONG run (300 steps after 20 of warm-up): the 10-step run of gpu_round.sh averages in the
# clock ramp of a cold GPU (first launches 114 us against 83 steady).  bash profiles/scripts/gpu_rocprof_long.sh <tag>
cd /root/repo; OUT=/root/repo/gpurun_out; mkdir -p $OUT; TAG=${1:-long}; export TMPDIR=/tmp
eval "$(timeout 600 python -m reagent_amd.device_preflight | tee /dev/stderr | grep "^export ")"; timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); assert float((x * 2).sum()) == 2 << 20" || { echo "preflight failed: faulty GPU node, aborting"; exit 97; }
for spec in "c2 bf16" "c2 bf16x3"; do
  set -- $spec; prec=$1_$2
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/proflong_${prec}_$TAG -o bench -- python /root/repo/bench.py --config $1 --precision $2 --repeats 1 --steps 300 --warmup 20 --no-cpu-baseline --no-kernel-profile --no-parity --no-accurate --no-also --sustained-steps 0 --launch eager > $OUT/rocproflong_${prec}_$TAG.log 2>&1; echo "rocprof $prec rc=$?")
  for f in $(find $OUT/proflong_${prec}_$TAG -name "*kernel_stats.csv" | head -1); do head -9 $f | cut -c1-150; done
  find $OUT/proflong_${prec}_$TAG -name "*kernel_trace.csv" -delete
  tail -1 $OUT/rocproflong_${prec}_$TAG.log | python -c "
import sys, json
try:
    r = json.loads(sys.stdin.read()); print('bench under rocprof: ms/step', r['ms_per_step'])
except Exception as e: print('no json', e)"
done
