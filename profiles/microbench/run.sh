#!/bin/bash
# build + run the forward phase-timing microbenchmark on the GPU box: bash profiles/microbench/run.sh [save]
set -e
cd /root/repo/profiles/microbench
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../reagent_amd/csrc -I../../include fwd_phases.hip -o fwd_phases -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|ScratchSize" | sort | uniq -c | grep -v ": 0 " || true
cd /root/repo
/usr/local/graft/bin/gpurun --timeout 300 -- "cd profiles/microbench && ./fwd_phases ${1:-0}" 2>&1 | grep -v "^\[gpurun\]"
