#!/bin/bash
# round 4, batch 16: with dense grouped spaces, do the two-stream halves of the C3 forward / backward still pay?  graph and eager
cd /root/repo; OUT=/root/repo/gpurun_out; mkdir -p $OUT
eval "$(timeout 600 python -m reagent_amd.device_preflight | tee /dev/stderr | grep "^export ")"; timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); assert float((x * 2).sum()) == 2 << 20" || { echo "preflight failed: faulty GPU node, aborting"; exit 97; }
for rep in 1 2; do
for prec in bf16 bf16x3; do
for cfg in "RG_QR_STREAMS=1 RG_QR_WGRAD_STREAMS=1" "RG_QR_STREAMS=0 RG_QR_WGRAD_STREAMS=1" "RG_QR_STREAMS=1 RG_QR_WGRAD_STREAMS=0" "RG_QR_STREAMS=0 RG_QR_WGRAD_STREAMS=0"; do
  env $cfg timeout 600 python bench.py --config c3 --precision $prec --steps 20 --warmup 5 --no-cpu-baseline --no-accurate --no-also --no-parity --sustained-steps 0 > $OUT/b16.json 2> $OUT/b16.err || tail -5 $OUT/b16.err
  python - "$prec $cfg" <<'PY'
import json, sys
r = json.load(open("/root/repo/gpurun_out/b16.json"))
lc = r.get("launch_calibration") or {}
print(f"[{sys.argv[1]:52s}] ms/step {r['ms_per_step']:.4f} graph {lc.get('graph_ms_per_step',0):.4f} eager {lc.get('eager_ms_per_step',0):.4f} host {lc.get('eager_host_enqueue_ms_per_step',0):.3f}")
PY
done; done; done
