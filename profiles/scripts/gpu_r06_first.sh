#!/bin/bash
# round 6, call 1: the bench exactly as the driver runs it (line size, parse), smoke, the GPU suite
cd /root/repo; OUT=/root/repo/gpurun_out/r06_run1; mkdir -p $OUT
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.txt 2> $OUT/bench_err.txt; echo "bench rc=$? bytes=$(wc -c < $OUT/bench_line.txt) lines=$(wc -l < $OUT/bench_line.txt)"
cp bench_report.json $OUT/bench_report.json
python -c "
import json; l=json.loads(open('$OUT/bench_line.txt').read()); print(json.dumps(l, indent=None)[:4000])"
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $OUT/smoke.log
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.txt
