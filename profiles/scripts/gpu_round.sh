#!/bin/bash
# round check in one call: smoke, pytest -m gpu, the driver's default bench command, rocprofv3 kernel stats of C2 bf16 / bf16x3
# (320 steps: a 13-step run averages in the clock ramp of a cold GPU — first launches 114 us against 83 steady)
# bash profiles/scripts/gpu_round.sh <tag> [notest] [norocprof]
cd /root/repo; OUT=/root/repo/gpurun_out; mkdir -p $OUT; TAG=${1:-r03}
eval "$(timeout 600 python -m reagent_amd.device_preflight | tee /dev/stderr | grep "^export ")"; timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); assert float((x * 2).sum()) == 2 << 20" || { echo "preflight failed: faulty GPU node, aborting"; exit 97; }
export TMPDIR=/tmp
{ rocminfo 2>/dev/null | grep -E "Marketing|gfx|Compute Unit" | head -6; nproc; rocm-smi --showclocks 2>/dev/null | head -20; } > $OUT/env_$TAG.log 2>&1
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke_$TAG.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke_$TAG.log
if [ "$2" != "notest" ]; then
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > $OUT/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest_gpu_$TAG.log | tail -3; grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu_$TAG.log | head
fi
bash profiles/scripts/gpu_default_bench.sh $TAG
[ "$3" == "norocprof" ] && exit 0
for spec in "c2 bf16" "c2 bf16x3"; do
  set -- $spec; prec=$1_$2
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${prec}_$TAG -o bench -- python /root/repo/bench.py --config $1 --precision $2 --repeats 1 --steps 300 --warmup 20 --no-cpu-baseline --no-kernel-profile --no-parity --no-accurate --no-also --sustained-steps 0 --launch eager > $OUT/rocprof_${prec}_$TAG.log 2>&1; echo "rocprof $prec rc=$?")
  for f in $(find $OUT/prof_${prec}_$TAG -name "*kernel_stats.csv" | head -1); do head -16 $f | cut -c1-200; done
  find $OUT/prof_${prec}_$TAG -name "*kernel_trace.csv" -delete
done
