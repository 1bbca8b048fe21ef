#!/bin/bash
cd /root/repo; OUT=/root/repo/gpurun_out; mkdir -p $OUT; TAG=${1:-r02c}
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); assert float((x * 2).sum()) == 2 << 20" || { echo "preflight failed: faulty GPU node, aborting"; exit 97; }
timeout 900 python -m pytest tests/test_fused_mlp.py tests/test_baseline_shapes.py -m gpu -q -s --no-header -p no:cacheprovider > $OUT/pytest_x3_$TAG.log 2>&1; echo "pytest rc=$?"; grep -E "^\[x3|baseline_c|actor weights|passed|failed|Error|assert" $OUT/pytest_x3_$TAG.log | tail -40
for prec in bf16x3 bf16; do
  timeout 600 python bench.py --config c2 --precision $prec --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_c2_${prec}_$TAG.json 2> $OUT/bench_c2_${prec}_$TAG.err; echo "$prec rc=$?"
  python - <<PY
import json
try:
    r=json.load(open("$OUT/bench_c2_${prec}_$TAG.json"))
    print("$prec value %.3e ms/step %.3f host %.3f fc %s" % (r["value"], r["ms_per_step"], r.get("host_enqueue_ms_per_step",0), r.get("fc_roofline")))
    print("  roofline", r.get("roofline")); print("  launch", r["config"].get("launch"))
    print("  parity", r.get("parity"))
    for k,v in list(r.get("per_call_ms_per_step",{}).items())[:12]: print("  %-70s %.4f" % (k,v))
except Exception as e: print("no json", e)
PY
  tail -3 $OUT/bench_c2_${prec}_$TAG.err
done
