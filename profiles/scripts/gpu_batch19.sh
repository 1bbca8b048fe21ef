#!/bin/bash
# round 4, batch 19: the quantile head with byte-offset bisection counts (one VALU operation fewer per search step) and a DPP
# fp64 prefix scan, against the previous build (reagent_amd/lib_prev), same box: head tests, then C3 bf16 eager, per-call times
cd /root/repo; OUT=/root/repo/gpurun_out; mkdir -p $OUT
eval "$(timeout 600 python -m reagent_amd.device_preflight | tee /dev/stderr | grep "^export ")"; timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); assert float((x * 2).sum()) == 2 << 20" || { echo "preflight failed: faulty GPU node, aborting"; exit 97; }
timeout 600 python -m pytest tests/test_qrdqn_trainer.py tests/test_baseline_shapes.py -m gpu -q --no-header -p no:cacheprovider -k "qrdqn or c3 or grouped or compact" 2>&1 | tail -2
for rep in 1 2 3; do
for lib in lib_prev lib; do
  RG_LIB=reagent_amd/$lib/libreagent_hip.so timeout 600 python bench.py --config c3 --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-accurate --no-also --no-parity --sustained-steps 0 --launch eager --no-graph > $OUT/b19.json 2> $OUT/b19.err || tail -5 $OUT/b19.err
  python - "$lib" <<'PY'
import json, sys
r = json.load(open("/root/repo/gpurun_out/b19.json"))
calls = r.get("per_call_ms_per_step", {})
short = {k.split("(")[0].replace("rg_mlp_", "").replace("rg_", "") + ("+save" if ", 1, (" in k else ""): v for k, v in calls.items()}
print(f"[{sys.argv[1]:10s}] ms/step {r['ms_per_step']:.4f} | head {short.get('qr_compact_head',0)*1e3:.1f} us  backward {short.get('backward_fused',0)*1e3:.0f}")
PY
done; done
