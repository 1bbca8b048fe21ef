#!/bin/bash
# round 4, batch 18: the grouped output layer (QR-DQN's 200 quantiles) through the hidden layers' main loop + staged whole-row stores
# (ring 2 / 4 / 8) against the per-row-tile loop, same box, C3 bf16
cd /root/repo; OUT=/root/repo/gpurun_out; mkdir -p $OUT
eval "$(timeout 600 python -m reagent_amd.device_preflight | tee /dev/stderr | grep "^export ")"; timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); assert float((x * 2).sum()) == 2 << 20" || { echo "preflight failed: faulty GPU node, aborting"; exit 97; }
timeout 600 python -m pytest tests/test_qrdqn_trainer.py tests/test_baseline_shapes.py -m gpu -q --no-header -p no:cacheprovider -k "qrdqn or c3 or grouped" 2>&1 | tail -2
for rep in 1 2; do
for lib in lib_out0 lib lib_out2 lib_out8; do
  RG_LIB=reagent_amd/$lib/libreagent_hip.so timeout 600 python bench.py --config c3 --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-accurate --no-also --no-parity --sustained-steps 0 --launch eager --no-graph > $OUT/b18.json 2> $OUT/b18.err || tail -5 $OUT/b18.err
  python - "$lib" <<'PY'
import json, sys
r = json.load(open("/root/repo/gpurun_out/b18.json"))
calls = r.get("per_call_ms_per_step", {})
short = {k.split("(")[0].replace("rg_mlp_", "").replace("rg_", "") + ("+save" if ", 1, (" in k else ""): v for k, v in calls.items()}
top = " ".join(f"{k}={v*1e3:.0f}" for k, v in list(short.items())[:6])
print(f"[{sys.argv[1]:10s}] ms/step {r['ms_per_step']:.4f} | {top}")
PY
done; done
