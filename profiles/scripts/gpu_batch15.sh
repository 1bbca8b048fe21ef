#!/bin/bash
# round 4, batch 15: dense grouped spaces (no padding between the groups: B / 128 tiles) against the padded layout, same box:
# GPU tests of the grouped engine, then C3 in bf16 and split-bf16 with RG_QR_DENSE = 1 / 0 (graph and eager)
cd /root/repo; OUT=/root/repo/gpurun_out; mkdir -p $OUT
eval "$(timeout 600 python -m reagent_amd.device_preflight | tee /dev/stderr | grep "^export ")"; timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); assert float((x * 2).sum()) == 2 << 20" || { echo "preflight failed: faulty GPU node, aborting"; exit 97; }
timeout 900 python -m pytest tests/test_qrdqn_trainer.py tests/test_baseline_shapes.py tests/test_full_size.py -m gpu -q --no-header -p no:cacheprovider -k "qrdqn or c3 or grouped" > $OUT/pytest_b15.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^E  " $OUT/pytest_b15.log | tail -12
for rep in 1 2; do
for prec in bf16 bf16x3; do
for dense in 1 0; do
  RG_QR_DENSE=$dense timeout 600 python bench.py --config c3 --precision $prec --steps 20 --warmup 5 --no-cpu-baseline --no-accurate --no-also --sustained-steps 0 > $OUT/b15.json 2> $OUT/b15.err || tail -5 $OUT/b15.err
  python - "$prec dense=$dense" <<'PY'
import json, sys
r = json.load(open("/root/repo/gpurun_out/b15.json"))
calls = r.get("per_call_ms_per_step", {})
short = {k.split("(")[0].replace("rg_mlp_", "").replace("rg_", "") + ("+save" if ", 1, (" in k else ""): v for k, v in calls.items()}
top = " ".join(f"{k}={v*1e3:.0f}" for k, v in list(short.items())[:9])
p = r.get("parity") or {}
lc = r.get("launch_calibration") or {}
print(f"[{sys.argv[1]:18s}] ms/step {r['ms_per_step']:.4f} graph {lc.get('graph_ms_per_step',0):.4f} eager {lc.get('eager_ms_per_step',0):.4f} parity ok {p.get('ok')} dquant {p.get('max_abs_dquantile')} | {top}")
PY
done; done; done
