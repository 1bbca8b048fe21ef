#!/bin/bash
# rocprofv3 kernel statistics of one bench configuration: profiles/scripts/gpu_prof1.sh <config> <precision> <tag>
cd /root/repo; OUT=/root/repo/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); assert float((x * 2).sum()) == 2 << 20" || { echo "preflight failed"; exit 97; }
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$1_$2_$3 -o bench -- python /root/repo/bench.py --config $1 --precision $2 --repeats 1 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile --no-parity --sustained-steps 0 --launch eager > $OUT/rocprof_$1_$2_$3.log 2>&1; echo "rocprof rc=$?")
find $OUT/prof_$1_$2_$3 -name "*kernel_trace.csv" -size +20M -delete
python - <<PY
import csv, glob
f = glob.glob("$OUT/prof_$1_$2_$3/**/*kernel_stats.csv", recursive=True)[0]
tot = 0
for r in csv.DictReader(open(f)):
    print("%-90s calls %4s avg_us %8.1f total_ms %8.2f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
