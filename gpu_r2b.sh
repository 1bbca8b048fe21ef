#!/bin/bash
cd /root/repo; OUT=/root/repo/gpurun_out; mkdir -p $OUT; TAG=${1:-r02b}
timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); assert float((x * 2).sum()) == 2 << 20" || { echo "preflight failed: faulty GPU node, aborting"; exit 97; }
timeout 900 python -m pytest tests/test_graph_replay.py tests/test_optimizers.py -m gpu -q -s --no-header -p no:cacheprovider > $OUT/pytest_graph_$TAG.log 2>&1; echo "pytest rc=$?"; tail -30 $OUT/pytest_graph_$TAG.log
timeout 600 python -m pytest tests/test_baseline_shapes.py -m gpu -q -s --no-header -p no:cacheprovider -k "c4 and f32" 2>&1 | grep -E "baseline_c4|actor param|passed|failed" | head -20
for c in c2 c4; do
  timeout 600 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_${c}_$TAG.json 2> $OUT/bench_${c}_$TAG.err; echo "$c rc=$?"
  python - <<PY
import json
try:
    r=json.load(open("$OUT/bench_${c}_$TAG.json"))
    print("$c value %.3e ms/step %.3f host %.3f fc_frac %.4f launch=%s" % (r["value"], r["ms_per_step"], r.get("host_enqueue_ms_per_step",0), r.get("fc_roofline",{}).get("frac",0), r["config"].get("launch")))
    print("  parity ok:", (r.get("parity") or {}).get("ok"), (r.get("parity") or {}).get("error"))
except Exception as e: print("no json", e)
PY
  tail -3 $OUT/bench_${c}_$TAG.err
done
timeout 600 python bench.py --config c2 --steps 20 --warmup 5 --no-cpu-baseline --no-graph --no-kernel-profile --no-parity | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('c2 eager ms/step %.3f host %.3f' % (r['ms_per_step'], r['host_enqueue_ms_per_step']))"
