#!/bin/bash
# One gpurun call: smoke, GPU parity tests, bench (bf16 + f32), rocprofv3 kernel trace.
# Everything is logged under gpurun_out/ (merged back by gpurun).
set -u
cd /root/repo
# preflight: a node whose first device touch faults (seen once: "Memory access fault by GPU" on tensor.to)
# would otherwise burn the whole GPU budget in core dumps and timeouts
eval "$(timeout 600 python -m reagent_amd.device_preflight | tee /dev/stderr | grep "^export ")"; timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); assert float((x * 2).sum()) == 2 << 20" || { echo "preflight failed: faulty GPU node, aborting"; exit 97; }
OUT=/root/repo/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
TAG=${1:-r01}
{
  echo "== rocminfo =="; rocminfo 2>/dev/null | grep -E "Marketing|gfx|Compute Unit" | head -8
  echo "== nproc =="; nproc
} > $OUT/env_$TAG.log 2>&1
echo "== smoke ==" ; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke_$TAG.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke_$TAG.log; tail -3 $OUT/smoke_$TAG.log
echo "== pytest -m gpu =="; timeout 1200 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider > $OUT/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu_$TAG.log; tail -15 $OUT/pytest_gpu_$TAG.log
echo "== bench bf16 =="; timeout 900 python bench.py --steps 30 --warmup 5 > $OUT/bench_bf16_$TAG.json 2> $OUT/bench_bf16_$TAG.err; echo "rc=$?"; cat $OUT/bench_bf16_$TAG.json; tail -5 $OUT/bench_bf16_$TAG.err
echo "== bench f32 =="; timeout 900 python bench.py --steps 5 --warmup 2 --precision f32 --no-cpu-baseline > $OUT/bench_f32_$TAG.json 2> $OUT/bench_f32_$TAG.err; echo "rc=$?"; cat $OUT/bench_f32_$TAG.json; tail -5 $OUT/bench_f32_$TAG.err
echo "== rocprofv3 kernel trace =="
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o bench -- python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile > $OUT/rocprof_$TAG.log 2>&1; echo "rocprof rc=$?"
cd /root/repo
find $OUT/prof_$TAG -name "*stats*" | head; 
for f in $(find $OUT/prof_$TAG -name "*kernel_stats.csv" | head -1); do head -25 $f; done
# keep the merge-back small: drop the raw per-dispatch trace if it is huge
find $OUT/prof_$TAG -name "*kernel_trace.csv" -size +20M -delete
