#!/bin/bash
# round 6, final tree in ONE call: env, smoke, the whole GPU suite as the driver runs it, PMC passes -> profiles/traffic.json stamped
# for these kernel sources, the driver's bench command (line + full record), rocprofv3 kernel stats of every configuration.
# bash profiles/scripts/gpu_r06_final.sh <tag>      -> gpurun_out/<tag>/
cd /root/repo; TAG=${1:-r06_final}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
eval "$(timeout 600 python -m reagent_amd.device_preflight | tee /dev/stderr | grep "^export ")"; timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); assert float((x * 2).sum()) == 2 << 20" || { echo "preflight failed: faulty GPU node, aborting"; exit 97; }
{ rocminfo 2>/dev/null | grep -E "Marketing|gfx|Compute Unit" | head -6; nproc; rocm-smi --showclocks 2>/dev/null | head -20; } > $OUT/env.log 2>&1
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
if [ "$2" != "notest" ]; then
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest_gpu.txt | tail -2; grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu.txt | head
fi
# ---- PMC passes (separate rocprofv3 runs per counter group, kernel trace only), then the stamp
bash profiles/scripts/gpu_pmc_all.sh $TAG > $OUT/pmc.log 2>&1; echo "pmc rc=$?"; grep -E "rc=" $OUT/pmc.log | tr '\n' ' '; echo
for spec in "c2 bf16" "c2 bf16x3" "c3 bf16" "c4 bf16"; do set -- $spec
  python profiles/pmc_to_traffic.py /root/repo/gpurun_out/pmc_${TAG}_$1_$2 $1 $2 "MI355X, round 6 final kernel sources, profiles/scripts/gpu_r06_final.sh $TAG (profiles/r06_pmc)" > /dev/null 2>> $OUT/pmc.log
done
cp profiles/traffic.json $OUT/traffic.json; mkdir -p $OUT/pmc; for d in /root/repo/gpurun_out/pmc_${TAG}_*; do cp $d/summary.txt $OUT/pmc/$(basename $d).txt 2>/dev/null; done
python -c "import json; t=json.load(open('profiles/traffic.json')); print('traffic.json stamp', t['source_stamp'], len(t['kernels']), 'kernels')"
# ---- the driver's command
t0=$(date +%s.%N)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.txt 2> $OUT/bench_err.txt; echo "bench rc=$? wall=$(python -c "import time,sys; print(round(time.time()-float(sys.argv[1]),1))" $t0) s bytes=$(wc -c < $OUT/bench_line.txt) lines=$(wc -l < $OUT/bench_line.txt)"
cp bench_report.json $OUT/bench_report.json; cat $OUT/bench_line.txt
# ---- rocprofv3 kernel stats per configuration (eager launches; 300 / 130 steps: past the clock ramp of a cold GPU)
for spec in "c2 bf16 300" "c2 bf16x3 300" "c3 bf16 130" "c3 bf16x3 130" "c4 bf16 130" "c4 bf16x3 130"; do
  set -- $spec; name=$1_$2
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o bench -- python /root/repo/bench.py --config $1 --precision $2 --repeats 1 --steps $3 --warmup 20 --no-cpu-baseline --no-kernel-profile --no-parity --no-accurate --no-also --sustained-steps 0 --launch eager > $OUT/rocprof_$name.log 2>&1; echo "rocprof $name rc=$?")
  for f in $(find $OUT/prof_$name -name "*kernel_stats.csv" | head -1); do cp $f $OUT/bench_${name}_kernel_stats_$3steps.csv; head -8 $f | cut -c1-160; done
  rm -rf $OUT/prof_$name
done
