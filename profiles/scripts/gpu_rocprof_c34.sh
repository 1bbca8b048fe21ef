#!/bin/bash
# rocprofv3 kernel statistics of C3 and C4 (bf16 and split-bf16), 120 steps after 10 of warm-up, eager launches
cd /root/repo; OUT=/root/repo/gpurun_out; mkdir -p $OUT; TAG=${1:-c34}; export TMPDIR=/tmp
eval "$(timeout 600 python -m reagent_amd.device_preflight | tee /dev/stderr | grep "^export ")"; timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); assert float((x * 2).sum()) == 2 << 20" || { echo "preflight failed: faulty GPU node, aborting"; exit 97; }
for spec in "c3 bf16" "c3 bf16x3" "c4 bf16" "c4 bf16x3"; do
  set -- $spec; prec=$1_$2
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${prec}_$TAG -o bench -- python /root/repo/bench.py --config $1 --precision $2 --repeats 1 --steps 120 --warmup 10 --no-cpu-baseline --no-kernel-profile --no-parity --no-accurate --no-also --sustained-steps 0 --launch eager --no-graph > $OUT/rocprof_${prec}_$TAG.log 2>&1; echo "rocprof $prec rc=$?")
  for f in $(find $OUT/prof_${prec}_$TAG -name "*kernel_stats.csv" | head -1); do head -12 $f | cut -c1-140; done
  find $OUT/prof_${prec}_$TAG -name "*kernel_trace.csv" -delete
done
