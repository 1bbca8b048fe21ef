#!/bin/bash
# full GPU check of the build: smoke, pytest -m gpu, bench c2 bf16 / bf16x3 / f32, c3, c4, rocprofv3 kernel stats (bf16 and bf16x3)
cd /root/repo; OUT=/root/repo/gpurun_out; mkdir -p $OUT; TAG=${1:-r02}
eval "$(timeout 600 python -m reagent_amd.device_preflight | tee /dev/stderr | grep "^export ")"; timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); assert float((x * 2).sum()) == 2 << 20" || { echo "preflight failed: faulty GPU node, aborting"; exit 97; }
export TMPDIR=/tmp
{ rocminfo 2>/dev/null | grep -E "Marketing|gfx|Compute Unit" | head -6; nproc; } > $OUT/env_$TAG.log 2>&1
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke_$TAG.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke_$TAG.log
timeout 1500 python -m pytest tests -m gpu -q -s --no-header -p no:cacheprovider > $OUT/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest_gpu_$TAG.log | tail -3; grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu_$TAG.log | head
for spec in "c2 bf16" "c2 bf16x3" "c2 f32" "c3 bf16" "c4 bf16" "c4 bf16x3"; do
  set -- $spec
  extra=""; [ "$1 $2" != "c2 bf16" ] && extra="--no-cpu-baseline"
  timeout 900 python bench.py --config $1 --precision $2 --steps 20 --warmup 5 $extra > $OUT/bench_$1_$2_$TAG.json 2> $OUT/bench_$1_$2_$TAG.err; echo "bench $1 $2 rc=$?"
  python - <<PY
import json
try:
    r=json.load(open("$OUT/bench_$1_$2_$TAG.json"))
    print("  value %.3e ms/step %.3f host %.3f fc_frac %.4f exec %.4f | %s" % (r["value"], r["ms_per_step"], r.get("host_enqueue_ms_per_step",0), r.get("fc_roofline",{}).get("frac",0), r.get("fc_roofline",{}).get("executed_frac",0), r["config"].get("launch")))
    p=r.get("parity") or {}; print("  parity ok:", p.get("ok"), {k:v for k,v in p.items() if k.startswith("max_") or k.startswith("rel_") or k.startswith("frac")}, p.get("error"))
    print("  cpu_baseline:", r.get("cpu_baseline"))
except Exception as e: print("  no json", e)
PY
done
for spec in "c2 bf16" "c2 bf16x3" "c3 bf16"; do
  set -- $spec; prec=$1_$2
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${prec}_$TAG -o bench -- python /root/repo/bench.py --config $1 --precision $2 --repeats 1 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile --no-parity --sustained-steps 0 --launch eager > $OUT/rocprof_${prec}_$TAG.log 2>&1; echo "rocprof $prec rc=$?")
  for f in $(find $OUT/prof_${prec}_$TAG -name "*kernel_stats.csv" | head -1); do head -14 $f | cut -c1-200; done
  find $OUT/prof_${prec}_$TAG -name "*kernel_trace.csv" -size +20M -delete
done
