#!/bin/bash
# round 2, first check: baseline-shape parity tests + bench configs with the parity object
cd /root/repo; OUT=/root/repo/gpurun_out; mkdir -p $OUT; TAG=${1:-r02a}
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); assert float((x * 2).sum()) == 2 << 20" || { echo "preflight failed: faulty GPU node, aborting"; exit 97; }
timeout 900 python -m pytest tests/test_baseline_shapes.py tests/test_bench_cli.py -m gpu -q -s --no-header -p no:cacheprovider -k "not bf16x3" > $OUT/pytest_base_$TAG.log 2>&1; echo "pytest rc=$?"; grep -E "baseline_c|passed|failed|Error" $OUT/pytest_base_$TAG.log | tail -30
for c in c2 c3 c4; do
  timeout 600 python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_${c}_$TAG.json 2> $OUT/bench_${c}_$TAG.err; echo "$c rc=$?"
  python - <<PY
import json
try:
    r=json.load(open("$OUT/bench_${c}_$TAG.json"))
    print("$c value %.3e ms/step %.3f host %.3f fc_frac %.4f launch=%s" % (r["value"], r["ms_per_step"], r.get("host_enqueue_ms_per_step",0), r.get("fc_roofline",{}).get("frac",0), r["config"].get("launch")))
    print("  parity", r.get("parity"))
    for k,v in list(r.get("per_call_ms_per_step",{}).items())[:14]: print("  %-70s %.4f" % (k,v))
except Exception as e: print("no json", e)
PY
  tail -3 $OUT/bench_${c}_$TAG.err
done
timeout 300 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_g2_$TAG.json 2> $OUT/bench_g2_$TAG.err; echo "gpus2 rc=$? (expected non-zero on a 1-GPU box)"; tail -3 $OUT/bench_g2_$TAG.err
