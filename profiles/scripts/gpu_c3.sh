#!/bin/bash
cd /root/repo; OUT=/root/repo/gpurun_out; mkdir -p $OUT; TAG=${1:-c3}
eval "$(timeout 600 python -m reagent_amd.device_preflight | tee /dev/stderr | grep "^export ")"; timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); assert float((x * 2).sum()) == 2 << 20" || { echo "preflight failed: faulty GPU node, aborting"; exit 97; }
timeout 900 python -m pytest tests/test_qrdqn_trainer.py tests/test_baseline_shapes.py -m gpu -q -s --no-header -p no:cacheprovider -k "qrdqn or c3 or grouped" > $OUT/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; grep -E "baseline_c3|passed|failed|^FAILED|^E  " $OUT/pytest_$TAG.log | tail -12
timeout 900 python bench.py --config c3 --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_c3_$TAG.json 2> $OUT/bench_c3_$TAG.err; echo "bench rc=$?"; tail -3 $OUT/bench_c3_$TAG.err
python - <<PY
import json
r=json.load(open("$OUT/bench_c3_$TAG.json"))
print("  value %.3e ms/step %.3f host %.3f | %s" % (r["value"], r["ms_per_step"], r.get("host_enqueue_ms_per_step",0), r["config"].get("launch")))
print("  parity", r.get("parity"))
for k,v in list(r.get("per_call_ms_per_step",{}).items())[:24]: print("  %-70s %.4f" % (k,v))
PY
