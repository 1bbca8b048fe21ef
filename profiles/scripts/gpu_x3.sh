#!/bin/bash
cd /root/repo; OUT=/root/repo/gpurun_out; mkdir -p $OUT; TAG=${1:-x3}
eval "$(timeout 600 python -m reagent_amd.device_preflight | tee /dev/stderr | grep "^export ")"; timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); assert float((x * 2).sum()) == 2 << 20" || { echo "preflight failed: faulty GPU node, aborting"; exit 97; }
timeout 900 python -m pytest tests/test_fused_mlp.py tests/test_baseline_shapes.py tests/test_dueling.py -m gpu -q -s --no-header -p no:cacheprovider > $OUT/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED" $OUT/pytest_$TAG.log | tail -5
for spec in "c2 bf16x3" "c4 bf16x3" "c2 bf16"; do
  set -- $spec
  timeout 900 python bench.py --config $1 --precision $2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_$1_$2_$TAG.json 2> $OUT/bench_$1_$2_$TAG.err; echo "bench $1 $2 rc=$?"
  python - <<PY
import json
try:
    r=json.load(open("$OUT/bench_$1_$2_$TAG.json"))
    print("  value %.3e ms/step %.3f host %.3f fc_frac %.4f dom %.4f | %s" % (r["value"], r["ms_per_step"], r.get("host_enqueue_ms_per_step",0), r.get("fc_roofline",{}).get("frac",0), r.get("roofline",{}).get("frac",0), r["config"].get("launch")))
    p=r.get("parity") or {}; print("  parity ok:", p.get("ok"), {k:v for k,v in p.items() if k.startswith("max_") or k.startswith("rel_") or k.startswith("frac")}, p.get("error"))
    for k,v in list(r.get("per_call_ms_per_step",{}).items())[:8]: print("  %-70s %.4f" % (k,v))
except Exception as e: print("  no json", e)
PY
done
