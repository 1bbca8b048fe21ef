#!/bin/bash
# round 4, batch 22: head-wgrad splits x one / two weight-gradient streams, C3 bf16, same box
cd /root/repo; OUT=/root/repo/gpurun_out; mkdir -p $OUT
eval "$(timeout 600 python -m reagent_amd.device_preflight | tee /dev/stderr | grep "^export ")"; timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); assert float((x * 2).sum()) == 2 << 20" || { echo "preflight failed: faulty GPU node, aborting"; exit 97; }
for rep in 1 2; do
for cfg in "RG_QR_HEAD_SPLITS=4 RG_QR_WGRAD_STREAMS=1" "RG_QR_HEAD_SPLITS=4 RG_QR_WGRAD_STREAMS=0" "RG_QR_HEAD_SPLITS=8 RG_QR_WGRAD_STREAMS=0" "RG_QR_HEAD_SPLITS=3 RG_QR_WGRAD_STREAMS=1" "RG_QR_HEAD_SPLITS=6 RG_QR_WGRAD_STREAMS=1"; do
  env $cfg timeout 600 python bench.py --config c3 --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-accurate --no-also --no-parity --sustained-steps 0 --launch eager --no-graph --no-kernel-profile > $OUT/b22.json 2> $OUT/b22.err || tail -5 $OUT/b22.err
  python - "$cfg" <<'PY'
import json, sys
r = json.load(open("/root/repo/gpurun_out/b22.json"))
print(f"[{sys.argv[1]:46s}] ms/step {r['ms_per_step']:.4f}")
PY
done; done
