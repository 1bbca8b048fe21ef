#!/bin/bash
# The calibration microbenchmarks with their output KEPT: bash profiles/scripts/gpu_microbench.sh [tag]
# (build the binaries first in the build container: bash profiles/microbench/build.sh)
cd /root/repo; OUT=/root/repo/gpurun_out/microbench_${1:-r03}; mkdir -p $OUT
eval "$(timeout 600 python -m reagent_amd.device_preflight | tee /dev/stderr | grep "^export ")"; timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); assert float((x * 2).sum()) == 2 << 20" || { echo "preflight failed: faulty GPU node, aborting"; exit 97; }
{ rocminfo 2>/dev/null | grep -E "Marketing|gfx|Compute Unit|Uuid" | head -8; nproc; rocm-smi --showclocks --showpower 2>/dev/null | grep -vE "^=|^$" | head -20; } > $OUT/env.log 2>&1
cd profiles/microbench
# clocks sampled WHILE a microbenchmark runs (the chip clocks to its power budget under dense MFMA on random data)
sample_clocks() { for i in 1 2 3 4 5 6; do sleep 0.5; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | head -3; done; }
for b in mfma_peak mfma_feed; do
  [ -x ./$b ] || { echo "$b not built"; continue; }
  ( sample_clocks > $OUT/${b}_clocks.log 2>&1 & ) ; timeout 120 ./$b > $OUT/$b.txt 2>&1; echo "$b rc=$?"; tail -14 $OUT/$b.txt
done
for b in fwd_phases wgrad_phases; do
  [ -x ./$b ] || { echo "$b not built"; continue; }
  timeout 120 ./$b > $OUT/$b.txt 2>&1; echo "$b rc=$?"; tail -12 $OUT/$b.txt
done
for b in chain_fwd chain_fwd_k2s8 chain_fwd_k4s4p4 chain_fwd_k2s9; do
  [ -x ./$b ] || continue
  ( sample_clocks > $OUT/${b}_clocks.log 2>&1 & ) ; timeout 60 ./$b 65536 512 > $OUT/$b.txt 2>&1; echo "$b rc=$?"; cat $OUT/$b.txt
done
(cd /tmp && export TMPDIR=/tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o mb -- /root/repo/profiles/microbench/mfma_feed > /dev/null 2>&1; echo "rocprof mfma_feed rc=$?")
find $OUT/prof -name "*kernel_trace.csv" -delete 2>/dev/null; find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -r head -12 | cut -c1-160
