#!/bin/bash
# round 4, batch 17: C3 after the grouped-space bookkeeping fix (scatter: sliced sums + ballots; count: loads eight at a time)
# and the loss / trunk-bias tails folded into the trunk weight gradient's reduce launch: tests, timeline, bench bf16 / split-bf16
cd /root/repo; OUT=/root/repo/gpurun_out; mkdir -p $OUT
eval "$(timeout 600 python -m reagent_amd.device_preflight | tee /dev/stderr | grep "^export ")"; timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); assert float((x * 2).sum()) == 2 << 20" || { echo "preflight failed: faulty GPU node, aborting"; exit 97; }
timeout 900 python -m pytest tests/test_qrdqn_trainer.py tests/test_baseline_shapes.py tests/test_full_size.py tests/test_graph_replay.py -m gpu -q --no-header -p no:cacheprovider > $OUT/pytest_b17.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^E  " $OUT/pytest_b17.log | tail -12
bash profiles/scripts/gpu_timeline.sh c3 bf16 RG_X=1 > $OUT/tl_c3_b17.txt 2>&1; head -30 $OUT/tl_c3_b17.txt
for rep in 1 2; do
for prec in bf16 bf16x3; do
  timeout 600 python bench.py --config c3 --precision $prec --steps 20 --warmup 5 --no-cpu-baseline --no-accurate --no-also --sustained-steps 0 > $OUT/b17.json 2> $OUT/b17.err || tail -5 $OUT/b17.err
  python - "$prec" <<'PY'
import json, sys
r = json.load(open("/root/repo/gpurun_out/b17.json"))
p = r.get("parity") or {}
lc = r.get("launch_calibration") or {}
print(f"[{sys.argv[1]:8s}] ms/step {r['ms_per_step']:.4f} graph {lc.get('graph_ms_per_step',0):.4f} eager {lc.get('eager_ms_per_step',0):.4f} host {lc.get('eager_host_enqueue_ms_per_step',0):.3f} parity ok {p.get('ok')} dquant {p.get('max_abs_dquantile')}")
PY
done; done
